"""Contact-model edge cases the reference allows but its tests do not exercise: friction bases other
than N = 4 (`MomentumBasedController{N}`, contacts.jl:16-23), frictionless contacts (mu = 0: every
ray of the basis coincides, H_rho has rank 1), and a single supporting foot.  Fused kernel (CPU
emulation / GPU) against the oracle tick; rho itself is not unique in these cases, the contact
wrenches and tau are."""
import numpy as np
import pytest

import emu
import parity
import scenarios as SC
from oracle import oracle as O
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd import workloads as W


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


def make(factory, n_basis=4, mu=None, one_foot=False, B=4, seed=5, dense=False, many=0):
    mech = M.atlas()
    if many:  # `many` contact points (copies of the eight, shrunk towards the feet's centres): beyond 32 contacts / 128 / 256 rho's the
        import copy  # working set's bit masks take a third, a fifth 64-bit word (the warm start's two words are refused there: test_warm_start.py)
        base, pts = list(mech.contact_points), []
        while len(pts) < many:
            cp = copy.deepcopy(base[len(pts) % len(base)])
            cp.position = np.array(cp.position) * np.array([1.0 - 0.05 * (len(pts) // len(base))] * 2 + [1.0])
            pts.append(cp)
        mech.contact_points = pts
        B = 2
    if mu is not None:
        for cp in mech.contact_points:
            cp.mu = mu
    if dense:  # 16 contact points x 4 rays = 64 rho: every lane of the wavefront is a ray (the quad paths at full width)
        import copy
        extra = []
        for cp in mech.contact_points:
            e = copy.deepcopy(cp)
            e.position = np.array(cp.position) * np.array([0.5, 0.6, 1.0])
            extra.append(e)
        mech.contact_points = list(mech.contact_points) + extra
    if one_foot:
        lf = mech.findbody("l_foot")
        mech.contact_points = [cp for cp in mech.contact_points if cp.body == lf]
    floating = mech.findbody("pelvis")
    c = factory(mech, n_basis, floating)
    st = W.standing_setup(c, mech, M.atlas_nominal_configuration(mech))
    c.finalize()
    q, v = W.standing_states(mech, B, seed, dq=0.05, drot=0.02, vstd=0.05)
    des = W.standing_desired(st, q, v)
    nc = len(st.contact_ids)
    return SC.Case(f"edge_N{n_basis}_mu{mu}_{'one' if one_foot else 'two'}{'_dense' if dense else ''}{'_many%d' % many if many else ''}", mech, c, q, v, des, np.full((B, nc), 1e-3),
                   np.full((B, nc), 1e6), floating=floating, n_basis=n_basis)


CASES = [dict(n_basis=3), dict(n_basis=6), dict(n_basis=8), dict(mu=0.0), dict(mu=1e-3), dict(one_foot=True),
         dict(n_basis=5, one_foot=True), dict(dense=True), dict(many=33), dict(many=33, n_basis=8)]  # (132 and 264 rho's)


def _run(factory, kw):
    g, o = make(factory, **kw), make(ora_factory, **kw)
    rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf)
    if "mu" not in kw:
        parity.compare(g, rg, o, ro)
        return
    # (nearly) frictionless cones: the reference-layout problem is so ill-conditioned that the oracle's
    # ADMM stalls at max_iter (as the reference's OSQP would: QPSolveFailure) and the rho-lifting of the
    # KKT certificate loses its footing; the independent dense device path L1 (oracle-evaluated blocks,
    # different assembly and search) is the second opinion instead
    import test_l1_assemble_solve as L1
    r1 = L1.run_l1(g, o)
    concl = (ro["status"] == 0) | (ro["status"] == 2)
    assert np.array_equal((rg["status"] == 0)[concl], (ro["status"] == 0)[concl])
    both = (rg["status"] == 0) & (r1["status"] == 0)
    assert both.sum() >= 3
    np.testing.assert_allclose(rg["vdot"][both], r1["vdot"][both], atol=1e-8, rtol=0)
    np.testing.assert_allclose(rg["cwrench"][both], r1["cwrench"][both], atol=1e-7, rtol=0)


@pytest.mark.parametrize("kw", CASES, ids=[str(k) for k in CASES])
def test_edge_contacts_emulated(kw):
    emu.build()
    emu.race_report()
    _run(emu.factory, kw)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


GPU_SLOW_CASES = (dict(n_basis=6), dict(n_basis=8), dict(mu=0.0), dict(mu=1e-3), dict(n_basis=5, one_foot=True), dict(many=33, n_basis=8))  # (a hiprtc compile and an oracle run each; the fast set keeps the odd basis, the single foot, the dense form and the 33-point case; the frictionless cones run on the CPU emulation and under -m gpu_slow)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [pytest.param(k, marks=pytest.mark.gpu_slow) if k in GPU_SLOW_CASES else k for k in CASES], ids=[str(k) for k in CASES])
def test_edge_contacts_gpu(kw):
    from qpcontrol_jl_amd import binding
    _run(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), kw)
