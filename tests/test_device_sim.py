"""SURVEY.md 8 f-2 on the device: the forward-dynamics / compliant-ground / integration step
(qpcontrol.jl_amd/csrc/qpc_sim.h, qpc_sim_step_batch) and the tick+step rollout (qpc_rollout_batch)
against the oracle's forward dynamics and manifold integration with the Python contact model of
tests/sim.py -- the batched counterpart of the reference notebook's closed-loop run
(notebooks/Standing controller.ipynb cells 9-11).  Tolerance: 1e-9 relative on the new state (fp64,
one 36 x 36 factorization deep)."""
from types import SimpleNamespace

import numpy as np
import pytest

import emu
import sim
from oracle import oracle as O
from qpcontrol_jl_amd import api, binding, models as M, workloads as W
from qpcontrol_jl_amd.workloads import HARD, JOINT

TOL = 1e-9


def gpu_factory(mech, n_basis, floating):
    return binding.Controller(binding.Model(mech), n_basis, floating)


def atlas_case(factory, B, seed=5):
    """Atlas with its 8 contact points; states around the stance with the feet in / near / off the ground"""
    mech = M.atlas()
    c = factory(mech, 4, mech.findbody("pelvis"))
    for cp in mech.contact_points:
        c.add_contact(cp.body, cp.position, cp.normal, cp.mu)
    for j in range(mech.nbodies):
        c.regularize(j, 0.05)
    c.finalize()
    q, v = W.standing_states(mech, B, seed, dq=0.15, drot=0.1, vstd=0.5)
    rng = np.random.default_rng(seed)
    q[:, 6] += rng.uniform(-0.03, 0.03, size=B)  # pelvis height: some feet penetrate, some hover
    tau = rng.normal(scale=30.0, size=(B, mech.nv))
    tau[:, :6] = 0.0
    ground = sim.GroundContact(0.02, k=6.0e4, d=400.0, ct=300.0)
    return mech, c, q, v, tau, ground, mech.contact_points


def tree_case(factory, B, floating, seed=11):
    """random trees with prismatic and FIXED joints, contact points on two bodies"""
    types = ([M.QUAT_FLOATING] if floating else []) + [M.REVOLUTE, M.PRISMATIC, M.FIXED, M.REVOLUTE, M.REVOLUTE, M.FIXED,
                                                       M.PRISMATIC, M.REVOLUTE]
    mech = M.rand_tree(types, seed=seed)
    nb = mech.nbodies
    c = factory(mech, 4, 0 if floating else -1)
    cps = [SimpleNamespace(body=nb - 1, position=np.array([0.1, 0.0, -0.1]), mu=0.6),
           SimpleNamespace(body=nb - 1, position=np.array([-0.1, 0.05, -0.1]), mu=0.6),
           SimpleNamespace(body=nb - 3, position=np.array([0.0, 0.1, 0.2]), mu=1.1)]
    if floating:  # the ABI ties contacts to a floating base (wrench balance)
        for cp in cps:
            c.add_contact(cp.body, cp.position, [0.0, 0.0, 1.0], cp.mu)
    else:
        cps = []
    for j in range(nb):
        c.regularize(j, 0.1)
    c.add_task(JOINT, joint=nb - 1, wkind=HARD)
    c.finalize()
    rng = np.random.default_rng(seed)
    qs, vs = zip(*(M.rand_state(mech, rng, 0.8) for _ in range(B)))
    tau = rng.normal(scale=2.0, size=(B, mech.nv))
    ground = sim.GroundContact(0.3, k=2.0e3, d=30.0, ct=50.0)  # high plane: most points are inside
    return mech, c, np.array(qs), np.array(vs), tau, ground, cps


def params(ground, dt):
    return binding.SimParams(dt, ground.z, ground.k, ground.d, ground.ct)


def check_step(case, dt=1e-3):
    mech, c, q, v, tau, ground, cps = case
    qn, vn, bad = c.sim_step(q, v, tau, params(ground, dt))
    assert not bad.any()
    om = O.OracleModel(mech)
    in_contact = 0
    for b in range(q.shape[0]):
        qr, vr = sim.step(om, mech, ground, q[b], v[b], tau[b], dt, cps)
        np.testing.assert_allclose(vn[b], vr, rtol=TOL, atol=TOL * (1.0 + np.abs(vr).max()))
        np.testing.assert_allclose(qn[b], qr, rtol=TOL, atol=TOL)
        om.set_state(q[b], v[b])
        in_contact += int(np.abs(ground.wrenches(om, mech, cps)).sum() > 0)
    return in_contact


def test_sim_step_atlas_emulated():
    emu.build()
    emu.race_report()
    n = check_step(atlas_case(emu.factory, 6))
    assert n >= 2  # the ground model was exercised
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def _packed_vs_single(factory, B, monkeypatch):
    """two robots per wavefront (lanes 0-31 / 32-63, mechanisms of at most 32 bodies) against one per wavefront
    (QPC_SIM_RPW=1): the same arithmetic in the same order per robot, so bitwise equal -- also for an odd batch, whose
    last robot is stepped by both halves"""
    monkeypatch.delenv("QPC_SIM_RPW", raising=False)
    mech, c, q, v, tau, ground, cps = atlas_case(factory, B)
    a = c.sim_step(q, v, tau, params(ground, 1e-3))
    monkeypatch.setenv("QPC_SIM_RPW", "1")
    mech, c1, q, v, tau, ground, cps = atlas_case(factory, B)
    b = c1.sim_step(q, v, tau, params(ground, 1e-3))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_sim_two_robots_per_wave_emulated(monkeypatch):
    emu.build()
    emu.race_report()
    _packed_vs_single(emu.factory, 7, monkeypatch)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
def test_sim_two_robots_per_wave_gpu(monkeypatch):
    # (QPC_SIM_RPW is an experiment switch: only the -DQPC_EXPERIMENTS build of the library reads it)
    _packed_vs_single(lambda mech, N, fl: binding.Controller(binding.Model(mech, binding.EXPERIMENTS_LIB), N, fl), 1025, monkeypatch)


@pytest.mark.parametrize("floating", [False, True])
def test_sim_step_random_trees_emulated(floating):
    emu.build()
    emu.race_report()
    n = check_step(tree_case(emu.factory, 5, floating), dt=2e-3)
    assert n >= 1 or not floating
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def check_failed_qp(factory):
    """a robot whose QP failed gets zero torque for the step and is counted (tests/sim.py does the same)"""
    mech, c, q, v, tau, ground, cps = atlas_case(factory, 3)
    status = np.array([0, 4, 1], dtype=np.int32)
    qn, vn, bad = c.sim_step(q, v, tau, params(ground, 1e-3), status=status, bad=np.array([0, 2, 0]))
    assert list(bad) == [0, 3, 0]
    om = O.OracleModel(mech)
    for b in range(3):
        qr, vr = sim.step(om, mech, ground, q[b], v[b], tau[b] if status[b] <= 1 else np.zeros(mech.nv), 1e-3, cps)
        np.testing.assert_allclose(vn[b], vr, rtol=TOL, atol=TOL * (1.0 + np.abs(vr).max()))
        np.testing.assert_allclose(qn[b], qr, rtol=TOL, atol=TOL)


def test_sim_failed_qp_emulated():
    emu.build()
    check_failed_qp(emu.factory)


def standing(lib_path, B):
    mech = M.atlas()
    low = api.MomentumBasedController(mech, n_basis=4, floatingjoint="pelvis_to_world", lib_path=lib_path)
    for cp in mech.contact_points:
        pt = low.addcontact(cp.body, cp.position, cp.normal, cp.mu)
        pt.maxnormalforce, pt.weight = 1e6, 1e-3
    ctrl = api.StandingController(low, ["l_foot", "r_foot"], "pelvis", M.atlas_nominal_configuration(mech))
    q, v = W.standing_states(mech, B, 3, dq=0.02, drot=0.01, vstd=0.0)
    v[:, 3] = np.linspace(-0.08, 0.08, B)
    k = 6.0e4
    z = 0.01308714 + mech.total_mass() * 9.81 / (8 * k)
    return mech, low, ctrl, q, v, api.SoftGround(z, k), sim.GroundContact(z, k=k)


def check_rollout(lib_path, B, steps):
    """qpc_rollout_batch == the host loop (device tick, oracle dynamics) of tests/sim.py, step for step"""
    mech, low, ctrl, q, v, ground, ref_ground = standing(lib_path, B)
    r = low.simulate((q, v), steps, ground)
    assert not r["bad"].any() and (r["status"] <= 1).all()

    def tick(qq, vv):
        return dict(tau=ctrl(None, 0.0, (qq, vv), throw_on_failure=False), status=low.status)
    log = sim.rollout(tick, mech, q, v, steps, ground=ref_ground)
    # the closed loop is stable, so the two trajectories stay together at rounding level
    np.testing.assert_allclose(r["q"], log["q"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(r["v"], log["v"], rtol=0, atol=1e-6)


def test_rollout_matches_host_loop_emulated():
    emu.build()
    check_rollout(emu.EMU_LIB, 2, 12)


def test_sim_argument_errors():
    emu.build()
    mech, c, q, v, tau, ground, cps = atlas_case(emu.factory, 1)
    with pytest.raises(binding.QpcError):
        c.sim_step(q, v, tau, binding.SimParams(0.0, 0.0, 1.0, 1.0, 1.0))  # dt must be positive


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_sim_step_atlas_gpu():
    assert check_step(atlas_case(gpu_factory, 64)) >= 8


@pytest.mark.gpu
@pytest.mark.parametrize("floating", [False, True])
def test_sim_step_random_trees_gpu(floating):
    check_step(tree_case(gpu_factory, 32, floating), dt=2e-3)


@pytest.mark.gpu
def test_sim_failed_qp_gpu():
    check_failed_qp(gpu_factory)


@pytest.mark.gpu
def test_rollout_matches_host_loop_gpu():
    check_rollout(None, 8, 40)


@pytest.mark.gpu
def test_rollout_standing_acceptance_gpu():
    """the notebook's acceptance criterion, entirely on the device: 256 pushed robots, 2 s at 1 kHz,
    every QP solves, nobody falls, the centre of mass settles on its reference"""
    mech, low, ctrl, q, v, ground, _ = standing(None, 256)
    r = low.simulate((q, v), 2000, ground)
    assert not r["bad"].any()
    kin = W.batch_kinematics(mech, r["q"], r["v"])
    com = kin["com"]
    assert np.all(np.abs(com[:, :2] - ctrl.comref[:2]) < 0.02)
    assert np.all(np.abs(com[:, 2] - ctrl.comref[2]) < 0.03)
    assert np.all(np.abs(r["v"]).max(axis=1) < 0.3)
    assert r["q"][:, 6].min() > 0.70 and r["q"][:, 6].max() < 0.95


def fuzz_case(factory, seed, B=3, big=False):
    """random mechanism (floating or fixed base, mixed joint types, random tree shape) with random contact
    points, random ground height and gains.  big (tests/offline/fuzz_sim.py): 30 ... 57 joints behind a floating base -- up to 58 bodies --, trees
    only (a chain that long is beyond the depth limit), up to 16 contact points"""
    rng = np.random.default_rng(seed)
    floating = bool(rng.integers(2)) or big
    kinds = [M.REVOLUTE, M.REVOLUTE, M.REVOLUTE, M.PRISMATIC, M.FIXED]
    types = ([M.QUAT_FLOATING] if floating else []) + [kinds[int(k)] for k in rng.integers(len(kinds), size=int(rng.integers(30, 58)) if big else int(rng.integers(3, 14)))]
    mech = M.rand_tree(types, seed=seed, chain=bool(rng.integers(2)) and not big)
    nb = mech.nbodies
    c = factory(mech, 4, 0 if floating else -1)
    cps = []
    if floating:
        for _ in range(int(rng.integers(1, 17)) if big else int(rng.integers(1, 6))):
            cp = SimpleNamespace(body=int(rng.integers(nb)), position=rng.normal(scale=0.2, size=3), mu=float(rng.uniform(0.2, 1.2)))
            cps.append(cp)
            c.add_contact(cp.body, cp.position, [0.0, 0.0, 1.0], cp.mu)
    for j in range(nb):
        c.regularize(j, 0.1)
    c.finalize()
    qs, vs = zip(*(M.rand_state(mech, rng, 1.0) for _ in range(B)))
    tau = rng.normal(scale=3.0, size=(B, mech.nv))
    ground = sim.GroundContact(float(rng.uniform(-0.5, 0.5)), k=float(rng.uniform(1e2, 5e3)), d=float(rng.uniform(0, 50)),
                               ct=float(rng.uniform(0, 80)))
    return mech, c, np.array(qs), np.array(vs), tau, ground, cps


@pytest.mark.parametrize("seed", range(400, 412))
def test_sim_step_fuzz_emulated(seed):
    emu.build()
    emu.race_report()
    check_step(fuzz_case(emu.factory, seed), dt=float(np.random.default_rng(seed).choice([1e-3, 5e-3])))
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
@pytest.mark.parametrize("jit", [pytest.param(True, marks=pytest.mark.gpu_slow), False])
def test_sim_step_fuzz_gpu(jit, monkeypatch):
    """8 random mechanisms through the JIT-specialised kernel (a hiprtc compile each), 24 through the generic one"""
    if not jit:
        monkeypatch.setenv("QPC_DISABLE_JIT", "1")
    for seed in (range(500, 508) if jit else range(520, 544)):
        check_step(fuzz_case(gpu_factory, seed, B=5), dt=1e-3)


@pytest.mark.gpu
def test_sim_generic_kernel_without_jit(monkeypatch):
    """the ahead-of-time generic simulation kernel (QPC_DISABLE_JIT=1) passes the same bar"""
    monkeypatch.setenv("QPC_DISABLE_JIT", "1")
    case = atlas_case(gpu_factory, 16)
    assert case[1].L.qpc_debug_jit_active(case[1].h) == 0
    assert check_step(case) >= 2
    check_step(tree_case(gpu_factory, 8, True), dt=2e-3)


# ------------------------------------------------------------------------------------------ physics
def floating_tree(factory, seed, gravity):
    mech = M.rand_tree([M.QUAT_FLOATING, M.REVOLUTE, M.REVOLUTE, M.PRISMATIC, M.REVOLUTE, M.FIXED, M.REVOLUTE], seed=seed)
    mech.gravity = gravity
    c = factory(mech, 4, 0)
    for j in range(mech.nbodies):
        c.regularize(j, 0.1)
    c.finalize()
    return mech, c


def check_newton_euler(factory):
    """free flight (no ground in reach, arbitrary joint torques): whatever the joints do, the step's v̇ satisfies
    A v̇ + Ȧv = W_gravity -- the whole-robot Newton-Euler equation the controller's wrench balance is built on
    (momentum.jl:162-193) and the reference's free-fall test asserts (test/controller.jl:128-165)"""
    mech, c = floating_tree(factory, 31, (0.3, -0.2, -9.81))
    rng = np.random.default_rng(2)
    B, dt = 6, 1e-3
    qs, vs = zip(*(M.rand_state(mech, rng, 1.0) for _ in range(B)))
    q, v = np.array(qs), np.array(vs)
    tau = rng.normal(scale=2.0, size=(B, mech.nv))
    tau[:, :6] = 0.0  # nothing pushes on the floating base
    qn, vn, bad = c.sim_step(q, v, tau, binding.SimParams(dt, -1e3, 1.0, 1.0, 1.0))
    assert not bad.any()
    om = O.OracleModel(mech)
    g = np.array(mech.gravity)
    for b in range(B):
        om.set_state(q[b], v[b])
        A, Adv = om.momentum_matrix()
        com = om.center_of_mass()
        mtot = mech.total_mass()
        rate = A @ ((vn[b] - v[b]) / dt) + Adv
        wg = np.concatenate([np.cross(com, mtot * g), mtot * g])
        np.testing.assert_allclose(rate, wg, rtol=0, atol=1e-7 * (1.0 + np.abs(wg).max()))


def check_energy_drift(factory):
    """no gravity, no torque, no ground: kinetic energy is conserved by the dynamics; the semi-implicit Euler step
    drifts at first order in dt (halving dt at fixed simulated time roughly halves the drift)"""
    mech, c = floating_tree(factory, 32, (0.0, 0.0, 0.0))
    rng = np.random.default_rng(3)
    q0, v0 = M.rand_state(mech, rng, 0.5)
    om = O.OracleModel(mech)

    def energy(q, v):
        om.set_state(q, v)
        return 0.5 * v @ om.mass_matrix() @ v
    drift = []
    for steps, dt in ((50, 2e-3), (100, 1e-3)):
        q, v = q0[None].copy(), v0[None].copy()
        for _ in range(steps):
            q, v, bad = c.sim_step(q, v, np.zeros((1, mech.nv)), binding.SimParams(dt, -1e3, 1.0, 1.0, 1.0))
            assert not bad.any()
        drift.append(abs(energy(q[0], v[0]) - energy(q0, v0)) / energy(q0, v0))
    assert drift[0] < 0.05 and drift[1] < 0.6 * drift[0] + 1e-9, drift


def test_sim_newton_euler_emulated():
    emu.build()
    check_newton_euler(emu.factory)


def test_sim_energy_drift_emulated():
    emu.build()
    check_energy_drift(emu.factory)


@pytest.mark.gpu
def test_sim_physics_gpu():
    check_newton_euler(gpu_factory)
    check_energy_drift(gpu_factory)


@pytest.mark.gpu
def test_rollout_equals_sequential_calls_large_batch_gpu():
    """a rollout over a batch larger than the resident grid equals the one-call-per-kernel sequence bit for bit"""
    B, steps = 8192 + 3, 3
    mech, low, ctrl, q, v, ground, _ = standing(None, B)
    r = low.simulate((q, v), steps, ground)
    qs, vs = q.copy(), v.copy()
    bad = np.zeros(B, dtype=np.int32)
    for _ in range(steps):
        tau = ctrl(None, 0.0, (qs, vs), throw_on_failure=False)
        qs, vs, bad = low._c.sim_step(qs, vs, tau, ground.params(1e-3), status=low.status, bad=bad)
    assert np.array_equal(r["q"], qs) and np.array_equal(r["v"], vs) and np.array_equal(r["bad"], bad)
    assert np.array_equal(r["tau"], tau)


# ---- a contact point registered on the WORLD body (qpc_add_contact accepts QPC_WORLD) ----------------
def check_world_contact(factory):
    """the simulation step must ignore it (it pushes on nothing that moves) instead of reading LDS in front of the body
    arrays: same new state as the twin controller without that contact, everything finite"""
    def make(with_world):
        mech = M.atlas()
        c = factory(mech, 4, mech.findbody("pelvis"))
        if with_world:
            c.add_contact(M.WORLD, [0.3, 0.1, 0.0], [0.0, 0.0, 1.0], 0.5)
        for cp in mech.contact_points:
            c.add_contact(cp.body, cp.position, cp.normal, cp.mu)
        for j in range(mech.nbodies):
            c.regularize(j, 0.05)
        c.finalize()
        return mech, c
    mech, cw_ = make(True)
    _, c0 = make(False)
    B = 6
    q, v = W.standing_states(mech, B, 9, dq=0.1, drot=0.05, vstd=0.3)
    q[:, 6] -= 0.02
    rng = np.random.default_rng(9)
    tau = rng.normal(scale=20.0, size=(B, mech.nv))
    tau[:, :6] = 0.0
    sp = params(sim.GroundContact(0.02, k=6.0e4, d=400.0, ct=300.0), 1e-3)
    q1, v1, bad1 = cw_.sim_step(q, v, tau, sp)
    q0, v0, bad0 = c0.sim_step(q, v, tau, sp)
    assert np.isfinite(q1).all() and np.isfinite(v1).all()
    assert np.array_equal(q1, q0) and np.array_equal(v1, v0) and np.array_equal(bad1, bad0)


def test_sim_world_contact_emulated():
    emu.build()
    emu.race_report()
    check_world_contact(emu.factory)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
def test_sim_world_contact_gpu():
    check_world_contact(gpu_factory)
