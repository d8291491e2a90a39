"""A reference-sized second robot: every integration test of the reference (/root/reference/test/controller.jl:99-285) runs on
`ValkyrieRobot.Valkyrie()`; its URDF is not in the tree, so `models.valkyrie()` is a synthetic humanoid of Valkyrie's topology with
fingers: 59 bodies, nv = 64 -- the kernel's limits (include/qpc.h: 64 bodies, 64 velocity coordinates, tree depth 32).  On it:
  * the foot-w.r.t.-palm SpatialAccelerationTask in the reference's five task modes (controller.jl:232-285), atol 1e-8;
  * the hard MomentumRateTask with parametric contacts and five activation probabilities (controller.jl:169-230), atol 1e-3;
  * tau / v̇ / wrenches against the oracle at 1e-8 (the constraint mode and the momentum task);
through the tick (L2: hiprtc-specialised kernel; the ahead-of-time generic kernel under `-m gpu_slow`) and through the reference's
own plug (L0: the reference-layout QP of the Valkyrie standing controller, 171 x 206, in the sparse workgroup form)."""
import ctypes

import numpy as np
import pytest

import emu
import parity
import scenarios as SC
from oracle import oracle as O
from qpcontrol_jl_amd import binding
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd import workloads as W

MODES = ("constraint", "scalar", "scalar_param", "matrix", "matrix_param")
BACKENDS = [pytest.param("emulated", id="emulated"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)]


def gpu_factory(mech, n_basis, floating):
    return binding.Controller(binding.Model(mech), n_basis, floating)


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


def _factory(backend):
    if backend == "emulated":
        emu.build()
        return emu.factory
    return gpu_factory


def test_model_is_at_the_kernels_limits():
    m = M.valkyrie()
    assert (m.nbodies, m.nq, m.nv) == (59, 65, 64) and len(m.contact_points) == 8
    assert 100.0 < m.total_mass() < 160.0


def _spatial(factory, mode, B=4):
    return SC.spatial_acceleration(factory, mode, B=B, seed=71, mech=M.valkyrie(), body="leftFoot", base="rightPalm", name=f"valkyrie_spatial_{mode}")


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mode", MODES)
def test_spatial_acceleration_foot_wrt_palm(backend, mode):
    """controller.jl:232-285 on the Valkyrie-sized robot: the foot's spatial acceleration w.r.t. the palm, expressed in the palm
    frame, equals the desired one in all five task modes (atol 1e-8); in the constraint mode also tau / v̇ against the oracle"""
    g = _spatial(_factory(backend), mode)
    r = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
    assert np.all(r["status"] == 0), r["status"]
    om = O.OracleModel(g.mech)
    body, base = g.mech.findbody("leftFoot"), g.mech.findbody("rightPalm")
    for b in range(g.B):
        om.set_state(g.q[b], g.v[b])
        acc = om.spatial_accelerations(r["vdot"][b])
        rel = om.transform_accel(acc[body] - acc[base], body, base, base)
        np.testing.assert_allclose(rel, g.desired[b], atol=1e-8, rtol=0)
    if mode == "constraint":
        o = _spatial(ora_factory, mode)
        ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
        parity.compare(g, r, o, ro)


def _momentum(factory):
    return SC.momentum_rate(factory, seed=9091, mech=M.valkyrie(), name="valkyrie_momentum_rate")


@pytest.mark.parametrize("backend", BACKENDS)
def test_achievable_momentum_rate(backend):
    """controller.jl:169-230 on the Valkyrie-sized robot: a hard MomentumRateTask, eight parametric contact points, activation
    probabilities 0 .. 1: the achievable desired rate is achieved (atol 1e-3), and the tick agrees with the oracle at 1e-8"""
    g = _momentum(_factory(backend))
    r = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
    assert np.all(r["status"] == 0), r["status"]
    om = O.OracleModel(g.mech)
    for b in range(g.B):
        om.set_state(g.q[b], g.v[b])
        com = om.center_of_mass()
        A, Adv = om.momentum_matrix()
        h = A @ r["vdot"][b] + Adv
        hc = np.concatenate([h[:3] - np.cross(com, h[3:]), h[3:]])
        np.testing.assert_allclose(hc, g.desired[b], atol=1e-3, rtol=0)
    o = _momentum(ora_factory)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
    parity.compare(g, r, o, ro)


@pytest.mark.gpu
@pytest.mark.gpu_slow
def test_generic_kernel_takes_the_valkyrie_tree(monkeypatch):
    """the ahead-of-time kernel (QPC_DISABLE_JIT=1): the tree is 15 deep, inside its ancestor bound of 16"""
    monkeypatch.setenv("QPC_DISABLE_JIT", "1")
    g = _spatial(gpu_factory, "constraint")
    assert g.ctrl.L.qpc_debug_jit_active(g.ctrl.h) == 0
    r = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
    o = _spatial(ora_factory, "constraint")
    parity.compare(g, r, o, o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf))


def valkyrie_standing(factory, B, seed=77, momentum_weight_matrix=None):
    """the StandingController task set (standing.jl:18-56) on the Valkyrie-sized robot: contacts, regularisation, hard foot tasks, the
    soft linear momentum task, the pelvis orientation task, joint tasks off the legs' paths"""
    mech = M.valkyrie()
    fl = mech.findbody("pelvis")
    ctrl = factory(mech, 4, fl)
    rng = np.random.default_rng(seed)
    q0 = np.zeros(mech.nq)
    for j in range(mech.nbodies):
        r = mech.configuration_range(j)
        if mech.joint_type[j] == M.QUAT_FLOATING:
            q0[r[0]] = 1.0
            q0[r[0] + 6] = 1.0  # pelvis height
    for s in ("left", "right"):
        q0[mech.configuration_range(mech.findbody(f"{s}KneePitchLink"))[0]] = 0.9
        q0[mech.configuration_range(mech.findbody(f"{s}HipPitchLink"))[0]] = -0.45
        q0[mech.configuration_range(mech.findbody(f"{s}AnklePitchLink"))[0]] = -0.45
    st = W.standing_setup(ctrl, mech, q0, feet=("leftFoot", "rightFoot"), momentum_weight_matrix=momentum_weight_matrix)
    ctrl.finalize()
    q = np.tile(q0, (B, 1))
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.REVOLUTE:
            q[:, mech.configuration_range(j)[0]] += rng.uniform(-0.1, 0.1, size=B)
    v = rng.normal(size=(B, mech.nv)) * 0.1
    des = W.standing_desired(st, q, v)
    nc = len(mech.contact_points)
    return mech, ctrl, q, v, des, np.full((B, nc), 1e-3), np.full((B, nc), 1e6)


@pytest.mark.parametrize("backend", BACKENDS)
def test_standing_tick_matches_the_oracle(backend):
    mech, c, q, v, des, cw, cm = valkyrie_standing(_factory(backend), 6)
    _, oc, *_ = valkyrie_standing(ora_factory, 1)
    r = c.control_batch(q, v, des, cw, cm)
    ro = oc.tick_batch(q, v, des, cw, cm)
    ok = ro["status"] == 0
    assert ok.sum() >= 4 and np.all(r["status"][ok] == 0)
    np.testing.assert_allclose(r["tau"][ok], ro["tau"][ok], atol=1e-8 * max(1.0, np.abs(ro["tau"][ok]).max()), rtol=0)
    np.testing.assert_allclose(r["vdot"][ok], ro["vdot"][ok], atol=1e-8 * max(1.0, np.abs(ro["vdot"][ok]).max()), rtol=0)


def momentum_weight_matrix():
    """a 6 x 6 symmetric positive definite W that is not diagonal (seeded): `addtask!(controller, MomentumRateTask, W::AbstractMatrix)`"""
    Q = np.random.default_rng(61).normal(size=(6, 6))
    return Q @ Q.T / 6.0 + np.diag([0.5, 0.5, 0.5, 1.0, 1.0, 1.0])


def valkyrie_l0_problems(B=4, matrix=False):
    """the reference-layout QPs of the Valkyrie standing controller as the oracle builds them (what Parametron would hand to the
    optimizer slot, momentum.jl:27,58), with the oracle's solution"""
    mech, oc, q, v, des, cw, cm = valkyrie_standing(ora_factory, B, momentum_weight_matrix=momentum_weight_matrix() if matrix else None)
    Ps, qs, As, ls, us, xs = [], [], [], [], [], []
    for b in range(B):
        P, qq, A, l, u = oc.build_qp(q[b], v[b], des[b], cw[b], cm[b])
        r = oc.tick(q[b], v[b], des[b], cw[b], cm[b])
        if r["status"] == 0:
            Ps.append(P); qs.append(qq); As.append(A); ls.append(l); us.append(u); xs.append(r["x"])
    return np.array(Ps), np.array(qs), np.array(As), np.array(ls), np.array(us), np.array(xs), mech.nv


def _l0(lib, matrix=False):
    P, q, A, l, u, x, nv = valkyrie_l0_problems(matrix=matrix)
    n, m = P.shape[1], A.shape[1]
    assert n > 160 and m > 190  # (171 x 206: beyond what the dense workgroup form holds in LDS)
    if matrix:  # the reference's OBJECTIVE_TERM_MATRIX_WEIGHT mode (test/controller.jl:232-285) on this robot: six slack variables, a 6 x 6 block in P
        assert (n, m) == (174, 209) and np.any(P[0] - np.diag(np.diag(P[0])) != 0.0)
    pr = binding.QpProblem(n, m, (P != 0).any(0), (A != 0).any(0), lib_path=lib)
    Pv, Av = pr.values(P, A)
    info = pr.analyze(l[0], u[0])
    assert info["in_use"] == 1 and info["lds_bytes"] <= 160 * 1024, info
    r = pr.solve(Pv, q, Av, l, u)
    assert (r["status"] == 0).all(), r["status"]
    np.testing.assert_allclose(r["x"][:, :nv], x[:, :nv], atol=1e-8 * max(1.0, np.abs(x[:, :nv]).max()), rtol=0)
    rw = pr.solve(Pv, q, Av, l, u, warm_start=True)
    assert (rw["status"] == 0).all() and (rw["iters"] <= 1).all()
    pr.close()
    return info


def test_l0_plug_on_the_valkyrie_problem_emulated():
    emu.build()
    L = ctypes.CDLL(emu.EMU_LIB)
    L.emu_l0_sp_count.restype = ctypes.c_long
    L.emu_l0_sp_count()
    _l0(emu.EMU_LIB)
    assert L.emu_l0_sp_count() >= 4  # (every solve in the sparse workgroup form)


@pytest.mark.gpu
def test_l0_plug_on_the_valkyrie_problem_gpu():
    _l0(None)


def test_l0_plug_on_the_valkyrie_problem_with_a_matrix_weight_emulated():
    """174 x 209 with a block in P: 158 KB of the 160 KB of LDS -- still the sparse form (the alternative for a P with blocks is the one-lane form)"""
    emu.build()
    info = _l0(emu.EMU_LIB, matrix=True)
    assert 150 * 1024 < info["lds_bytes"] <= 160 * 1024 - 64


@pytest.mark.gpu
def test_l0_plug_on_the_valkyrie_problem_with_a_matrix_weight_gpu():
    _l0(None, matrix=True)
