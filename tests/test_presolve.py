"""Null-space presolve of the hard foot tasks (qpc_compile.cpp QpcDims::presolve, qpc_tick.h qpc_presolve / qpc_postsolve): the six
joint accelerations of a leg are eliminated through its foot's SpatialAccelerationTask, the solver runs on 3 free columns and 9 rows
instead of 15 and 21, the eliminated accelerations are restored afterwards.  Same QP, so against the controller built with
QPC_NO_PRESOLVE=1: equal statuses, tau and vdot to 1e-10 of the torque scale, the same working-set changes; a QP whose leg is
near a singularity (straight knee) is solved in the full dimensions BY THE SAME KERNEL, bitwise equal to the plain controller; a task
set the layout rules do not cover is left alone.  iters carries 1 000 000 for a QP that went through the reduced problem."""
import os

import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding
from qpcontrol_jl_amd import models as M
from qpcontrol_jl_amd import workloads as W


def _build(factory, B, config, presolve, fused=False):
    old = os.environ.pop("QPC_NO_PRESOLVE", None)
    if not presolve:
        os.environ["QPC_NO_PRESOLVE"] = "1"
    try:
        c, wl = W.atlas_workload(factory, B, config, fused_laws=fused)
    finally:
        os.environ.pop("QPC_NO_PRESOLVE", None)
        if old is not None:
            os.environ["QPC_NO_PRESOLVE"] = old
    return c, wl


def _tick(c, wl, q=None, v=None):
    q = wl.q if q is None else q
    v = wl.v if v is None else v
    des = W.standing_desired(wl.setup, q, v)
    return c.control_batch(q, v, des, wl.contact_weight, wl.contact_maxnormalforce)


def _compare(a, b):
    # the same QPs are solved; HOW an infeasible one fails (INFEASIBLE / ITERATION_LIMIT on the cycle guard) may depend on rounding
    assert np.array_equal(a["status"] == 0, b["status"] == 0), (a["status"][a["status"] != b["status"]], b["status"][a["status"] != b["status"]])
    assert np.mean(a["status"] != b["status"]) <= 0.03  # (config 5: 6.5 % of the QPs are infeasible; ~1 % end with another failure code)
    ok = a["status"] == 0
    assert ok.sum() >= len(ok) // 2
    scale = np.maximum(1.0, np.abs(b["tau"][ok]).max(axis=1, keepdims=True))
    assert float(np.abs((a["tau"][ok] - b["tau"][ok]) / scale).max()) <= 1e-10
    vs = np.maximum(1.0, np.abs(b["vdot"][ok]).max(axis=1, keepdims=True))
    assert float(np.abs((a["vdot"][ok] - b["vdot"][ok]) / vs).max()) <= 1e-10
    assert np.mean(a["iters"][ok] % 1000 != b["iters"][ok] % 1000) <= 0.05  # same search (ties by rounding aside)


@pytest.mark.parametrize("config", [3, 4, 5])
def test_presolve_equals_full_problem_emulated(config):
    emu.build()
    emu.race_report()
    cp, wl = _build(emu.factory, 48, config, True)
    cf, _ = _build(emu.factory, 48, config, False)
    assert cp.L.qpc_debug_presolve_blocks(cp.h) == 2 and cf.L.qpc_debug_presolve_blocks(cf.h) == 0
    a, b = _tick(cp, wl), _tick(cf, wl)
    _compare(a, b)
    assert np.all(a["iters"] >= 1000000) and np.all(b["iters"] < 1000000)  # every QP went through the reduced problem
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def _leg_det(mech, q, side="l"):
    """determinant of the 6 x 6 block of the foot Jacobian that belongs to the six joints of a leg (world frame, per state)"""
    kin = W.batch_kinematics(mech, q, np.zeros((q.shape[0], mech.nv)))
    cols = []
    for name in ("hpz", "hpx", "hpy", "kny", "aky", "akx"):
        j = mech.findjoint(f"{side}_leg_{name}")
        w = np.einsum("bij,j->bi", kin["R"][j], mech.joint_axis[j])
        cols.append(np.concatenate([w, np.cross(kin["p"][j], w)], axis=1))
    return np.linalg.det(np.stack(cols, axis=2))


def _straight_knees(wl, which):
    """states `which` get the knee angle at which their left leg is singular (the three pitch axes in one plane: found by bisection
    on the determinant of the leg block -- the link offsets put it near, not at, a zero knee angle)"""
    q = wl.q.copy()
    m = wl.mech
    k = m.configuration_range(m.findjoint("l_leg_kny"))[0]
    lo, hi = np.full(q.shape[0], -0.6), np.full(q.shape[0], 0.6)
    qa, qb = q.copy(), q.copy()
    qa[:, k], qb[:, k] = lo, hi
    dlo, dhi = _leg_det(m, qa), _leg_det(m, qb)
    assert np.all(dlo[which] * dhi[which] < 0)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        qa[:, k] = mid
        dm = _leg_det(m, qa)
        left = dlo * dm <= 0
        hi = np.where(left, mid, hi)
        lo = np.where(left, lo, mid); dlo = np.where(left, dlo, dm)
    q[which, k] = (0.5 * (lo + hi))[which]
    return q


def test_singular_leg_is_solved_in_full_dimensions_emulated():
    emu.build()
    emu.race_report()
    cp, wl = _build(emu.factory, 32, 3, True)
    cf, _ = _build(emu.factory, 32, 3, False)
    which = np.arange(32) % 2 == 0
    q = _straight_knees(wl, which)
    a, b = _tick(cp, wl, q=q), _tick(cf, wl, q=q)
    assert np.all(a["iters"][which] < 1000000) and np.all(a["iters"][~which] >= 1000000)
    # the same kernel solved them as assembled: bit for bit what the plain controller does
    assert np.array_equal(a["status"][which], b["status"][which])
    assert np.array_equal(a["tau"][which], b["tau"][which]) and np.array_equal(a["vdot"][which], b["vdot"][which])
    _compare({k: x[~which] for k, x in a.items()}, {k: x[~which] for k, x in b.items()})
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.parametrize("off", [1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-8])
def test_nearly_singular_legs_emulated(off):
    """knee angles `off` away from the singular one: whichever way the conditioning test decides, the answer is the one of the
    full problem (the elimination loses cond(A_e) eps; the test lets a ratio of 1e-4 between the smallest and the largest diagonal entry of R through)"""
    emu.build()
    cp, wl = _build(emu.factory, 16, 3, True)
    cf, _ = _build(emu.factory, 16, 3, False)
    q = _straight_knees(wl, np.ones(16, dtype=bool))
    q[:, wl.mech.configuration_range(wl.mech.findjoint("l_leg_kny"))[0]] += off
    a, b = _tick(cp, wl, q=q), _tick(cf, wl, q=q)
    assert np.array_equal(a["status"] == 0, b["status"] == 0)
    ok = a["status"] == 0
    if ok.any():
        scale = np.maximum(1.0, np.abs(b["tau"][ok]).max(axis=1, keepdims=True))
        err = float(np.abs((a["tau"][ok] - b["tau"][ok]) / scale).max())
        assert err <= 1e-10, (err, np.mean(a["iters"] >= 1000000))


def _random_biped(seed):
    """floating pelvis + two chains of six revolute joints with random axes, offsets and inertias, four contact points per foot"""
    rng = np.random.default_rng(seed)
    m = M.Mechanism()

    def inertia(mass):
        a = rng.normal(size=(3, 3))
        i = (a @ a.T * 0.02 + np.eye(3) * 0.01) * mass
        return (i[0, 0], i[1, 1], i[2, 2], i[0, 1], i[0, 2], i[1, 2])
    m.add_body("pelvis", M.WORLD, "pelvis_to_world", M.QUAT_FLOATING, mass=40.0, com=rng.normal(size=3) * 0.03, inertia=inertia(40.0))
    for s, sg in (("l", 1.0), ("r", -1.0)):
        parent = "pelvis"
        for k in range(6):
            mass = float(rng.uniform(0.5, 6.0))
            m.add_body(f"{s}{k}", parent, f"{s}_j{k}", M.REVOLUTE, axis=rng.normal(size=3),
                       p=np.array([0.05 * rng.normal(), sg * 0.1 * (k == 0) + 0.03 * rng.normal(), -0.15 * rng.uniform(0.3, 1.5)]),
                       mass=mass, com=rng.normal(size=3) * 0.05, inertia=inertia(mass))
            parent = f"{s}{k}"
        foot = m.findbody(f"{s}5")
        for x, y in ((-0.08, 0.05), (-0.08, -0.05), (0.15, 0.05), (0.15, -0.05)):
            m.contact_points.append(M.ContactSpec(foot, np.array([x, y, -0.05]), np.array([0.0, 0.0, 1.0]), 0.7))
    return m


@pytest.mark.parametrize("seed", range(6))
def test_random_bipeds_against_the_oracle_emulated(seed):
    """not only Atlas: random leg geometries, the reduced problem against the full one AND against the oracle"""
    from oracle import oracle as O
    emu.build()
    emu.race_report()
    mech = _random_biped(seed)
    rng = np.random.default_rng(100 + seed)
    B = 12
    q = np.zeros((B, mech.nq)); q[:, 0] = 1.0
    rv = rng.uniform(-0.1, 0.1, size=(B, 3))
    th = np.linalg.norm(rv, axis=1)
    q[:, 0] = np.cos(th / 2); q[:, 1:4] = rv / th[:, None] * np.sin(th / 2)[:, None]
    q[:, 4:7] = rng.normal(size=(B, 3)) * 0.1 + np.array([0.0, 0.0, 0.9])
    q[:, 7:] = rng.uniform(-0.4, 0.4, size=(B, mech.nq - 7))
    v = rng.normal(size=(B, mech.nv)) * 0.2

    def run(factory, presolve):
        old = os.environ.pop("QPC_NO_PRESOLVE", None)
        if not presolve:
            os.environ["QPC_NO_PRESOLVE"] = "1"
        try:
            c = factory(mech, 4, mech.findbody("pelvis"))
            st = W.standing_setup(c, mech, q[0], feet=("l5", "r5"))
            c.finalize()
        finally:
            os.environ.pop("QPC_NO_PRESOLVE", None)
            if old is not None:
                os.environ["QPC_NO_PRESOLVE"] = old
        des = W.standing_desired(st, q, v)
        nc = len(st.contact_ids)
        return c, des, np.full((B, nc), 1e-3), np.full((B, nc), 1e6)
    cp, des, cw, cm = run(emu.factory, True)
    cf, _, _, _ = run(emu.factory, False)
    assert cp.L.qpc_debug_presolve_blocks(cp.h) == 2
    a, b = cp.control_batch(q, v, des, cw, cm), cf.control_batch(q, v, des, cw, cm)
    assert np.array_equal(a["status"] == 0, b["status"] == 0)
    ok = a["status"] == 0
    assert ok.sum() >= B // 2, a["status"]
    scale = np.maximum(1.0, np.abs(b["tau"][ok]).max(axis=1, keepdims=True))
    assert float(np.abs((a["tau"][ok] - b["tau"][ok]) / scale).max()) <= 1e-9
    oc = O.OracleController(O.OracleModel(mech), 4, mech.findbody("pelvis"))
    W.standing_setup(oc, mech, q[0], feet=("l5", "r5"))
    oc.finalize()
    ro = oc.tick_batch(q, v, des, cw, cm)
    both = ok & (ro["status"] == 0)
    assert both.sum() >= 1
    assert float(np.abs(a["tau"][both] - ro["tau"][both]).max()) <= 1e-8 * float(scale.max())
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, pytest.param(3, marks=pytest.mark.gpu_slow)])
def test_random_bipeds_gpu(seed):
    mech = _random_biped(seed)
    rng = np.random.default_rng(200 + seed)
    B = 512
    q = np.zeros((B, mech.nq))
    rv = rng.uniform(-0.1, 0.1, size=(B, 3))
    th = np.linalg.norm(rv, axis=1)
    q[:, 0] = np.cos(th / 2); q[:, 1:4] = rv / th[:, None] * np.sin(th / 2)[:, None]
    q[:, 4:7] = rng.normal(size=(B, 3)) * 0.1 + np.array([0.0, 0.0, 0.9])
    q[:, 7:] = rng.uniform(-0.4, 0.4, size=(B, mech.nq - 7))
    v = rng.normal(size=(B, mech.nv)) * 0.2
    res = []
    for presolve in (True, False):
        old = os.environ.pop("QPC_NO_PRESOLVE", None)
        if not presolve:
            os.environ["QPC_NO_PRESOLVE"] = "1"
        try:
            c = binding.Controller(binding.Model(mech), 4, mech.findbody("pelvis"))
            st = W.standing_setup(c, mech, q[0], feet=("l5", "r5"))
            c.finalize()
        finally:
            os.environ.pop("QPC_NO_PRESOLVE", None)
            if old is not None:
                os.environ["QPC_NO_PRESOLVE"] = old
        assert c.L.qpc_debug_presolve_blocks(c.h) == (2 if presolve else 0)
        nc = len(st.contact_ids)
        res.append(c.control_batch(q, v, W.standing_desired(st, q, v), np.full((B, nc), 1e-3), np.full((B, nc), 1e6)))
    _compare(res[0], res[1])
    assert np.mean(res[0]["iters"] >= 1000000) > 0.9


def test_feet_in_either_order_emulated():
    """the right foot's task registered first: its block's columns come AFTER the other block's (per-block first columns)"""
    emu.build()
    emu.race_report()
    mech = M.atlas()
    q0 = M.atlas_nominal_configuration(mech)
    B = 24
    q, v = W.standing_states(mech, B, 31)
    res = []
    for presolve in (True, False):
        old = os.environ.pop("QPC_NO_PRESOLVE", None)
        if not presolve:
            os.environ["QPC_NO_PRESOLVE"] = "1"
        try:
            c = emu.factory(mech, 4, mech.findbody("pelvis"))
            st = W.standing_setup(c, mech, q0, feet=("r_foot", "l_foot"))
            c.finalize()
        finally:
            os.environ.pop("QPC_NO_PRESOLVE", None)
            if old is not None:
                os.environ["QPC_NO_PRESOLVE"] = old
        assert c.L.qpc_debug_presolve_blocks(c.h) == (2 if presolve else 0)
        res.append(c.control_batch(q, v, W.standing_desired(st, q, v), np.full((B, 8), 1e-3), np.full((B, 8), 1e6)))
    _compare(res[0], res[1])
    assert np.all(res[0]["iters"] >= 1000000)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def test_other_task_orders_are_left_alone_emulated():
    """the layout rules (foot tasks first, hard, six private joint columns each) decide at qpc_finalize"""
    emu.build()
    mech = M.atlas()
    q0 = M.atlas_nominal_configuration(mech)

    def blocks(**kw):
        c = emu.factory(mech, 4, mech.findbody("pelvis"))
        W.standing_setup(c, mech, q0, **kw)
        c.finalize()
        return c.L.qpc_debug_presolve_blocks(c.h)
    assert blocks() == 2
    assert blocks(feet=("l_foot",)) == 1      # one foot task: one block (the other leg's joints are pinned by their joint tasks)
    assert blocks(contacts=False) == 0        # no contacts: no wrench rows to reduce
    # a fixed-base arm: nothing to eliminate
    c, _ = W.atlas_workload(emu.factory, 2, 2)
    assert c.L.qpc_debug_presolve_blocks(c.h) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("config", [3, 5])
def test_presolve_equals_full_problem_gpu(config):
    f = lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl)  # noqa: E731
    cp, wl = _build(f, 4096, config, True)
    cf, _ = _build(f, 4096, config, False)
    assert cp.L.qpc_debug_presolve_blocks(cp.h) == 2 and cf.L.qpc_debug_presolve_blocks(cf.h) == 0
    a, b = _tick(cp, wl), _tick(cf, wl)
    _compare(a, b)
    assert np.all(a["iters"] >= 1000000)


@pytest.mark.gpu
def test_singular_leg_is_solved_in_full_dimensions_gpu():
    f = lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl)  # noqa: E731
    cp, wl = _build(f, 2048, 3, True)
    cf, _ = _build(f, 2048, 3, False)
    which = np.arange(2048) % 4 == 0
    q = _straight_knees(wl, which)
    a, b = _tick(cp, wl, q=q), _tick(cf, wl, q=q)
    assert np.all(a["iters"][which] < 1000000) and np.all(a["iters"][~which] >= 1000000)
    # (bit for bit in the emulator; two compilations of the full-dimension solver on the device differ in their last bits)
    _compare({k: x[which] for k, x in a.items()}, {k: x[which] for k, x in b.items()})
    _compare({k: x[~which] for k, x in a.items()}, {k: x[~which] for k, x in b.items()})


def test_per_qp_momentum_weight_emulated():
    """the folded rows take their weights from the slab (per-QP scalar weight of the momentum task: no static Hessian tables),
    and a matrix weight keeps the presolve off"""
    from qpcontrol_jl_amd.workloads import LINEAR_MOMENTUM_RATE, SCALAR, SCALAR_PER_QP
    emu.build()
    emu.race_report()
    mech = M.atlas()
    q0 = M.atlas_nominal_configuration(mech)

    def build(presolve, wkind, weight):
        old = os.environ.pop("QPC_NO_PRESOLVE", None)
        if not presolve:
            os.environ["QPC_NO_PRESOLVE"] = "1"
        try:
            c = emu.factory(mech, 4, mech.findbody("pelvis"))
            real = c.add_task

            def add_task(kind, *a, **kw):
                if kind == LINEAR_MOMENTUM_RATE:
                    kw["wkind"], kw["weight"] = wkind, weight
                return real(kind, *a, **kw)
            c.add_task = add_task
            st = W.standing_setup(c, mech, q0)
            c.finalize()
        finally:
            os.environ.pop("QPC_NO_PRESOLVE", None)
            if old is not None:
                os.environ["QPC_NO_PRESOLVE"] = old
        return c, st
    B = 16
    q, v = W.standing_states(mech, B, 77)
    rng = np.random.default_rng(5)
    tw = rng.uniform(0.3, 3.0, size=(B, 1))
    cp, st = build(True, SCALAR_PER_QP, None)
    cf, _ = build(False, SCALAR_PER_QP, None)
    assert cp.L.qpc_debug_presolve_blocks(cp.h) == 2 and cp.layout.n_task_weights == 1
    des = W.standing_desired(st, q, v)
    cw, cm = np.full((B, 8), 1e-3), np.full((B, 8), 1e6)
    a = cp.control_batch(q, v, des, cw, cm, task_weights=tw)
    b = cf.control_batch(q, v, des, cw, cm, task_weights=tw)
    _compare(a, b)
    assert np.all(a["iters"] >= 1000000)
    # the weight matters (otherwise the comparison above proves nothing about where it is read from)
    c1, _ = build(True, SCALAR, [1.0])
    r1 = c1.control_batch(q, v, des, cw, cm)
    assert float(np.abs(r1["tau"] - a["tau"]).max()) > 1e-3
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


# ---------------------------------------------------------------------------------------------------------------------
# The presolve must not depend on how the model numbers its bodies nor on the order in which the tasks were registered
# (StandingController iterates a Dict of feet, /root/reference/src/highlevel/standing.jl:35-49: unspecified order): the host
# compiler owns the order of the free columns and of the row blocks and arranges them itself (qpc_compile.cpp pass 1.5).
def _shuffled_atlas(rng):
    """the synthetic Atlas with its bodies listed in another (still topological) order: same tree, same inertias"""
    m0 = M.atlas()
    nb = m0.nbodies
    order, placed = [], set()
    remaining = list(range(nb))
    while remaining:  # random topological order: any body whose parent is placed may come next
        ready = [b for b in remaining if m0.parent[b] < 0 or m0.parent[b] in placed]
        b = ready[int(rng.integers(0, len(ready)))]
        order.append(b); placed.add(b); remaining.remove(b)
    new_of = {old: new for new, old in enumerate(order)}
    m = M.Mechanism()
    for old in order:
        par = m0.parent[old]
        m.add_body(m0.body_names[old], M.WORLD if par < 0 else new_of[par], m0.joint_names[old], m0.joint_type[old], axis=m0.joint_axis[old],
                   R=m0.joint_R[old], p=m0.joint_p[old], mass=m0.mass[old], com=m0.com[old], inertia=m0.inertia[old])
    for cp in m0.contact_points:
        m.contact_points.append(M.ContactSpec(new_of[cp.body], cp.position, cp.normal, cp.mu))
    return m0, m, order


def _map_state(m0, m, q0, v0):
    """states of the original numbering -> the shuffled model's coordinate order"""
    q, v = np.zeros((q0.shape[0], m.nq)), np.zeros((v0.shape[0], m.nv))
    for j, name in enumerate(m.joint_names):
        j0 = m0.findjoint(name)
        q[:, list(m.configuration_range(j))] = q0[:, list(m0.configuration_range(j0))]
        v[:, list(m.velocity_range(j))] = v0[:, list(m0.velocity_range(j0))]
    return q, v


def _standing_shuffled_tasks(ctrl, mech, nominal_q, rng):
    """the StandingController task set with the registration order shuffled: joint tasks, pelvis, momentum and foot tasks mixed,
    right foot before left"""
    from qpcontrol_jl_amd.workloads import ANGULAR, HARD, JOINT, LINEAR_MOMENTUM_RATE, SCALAR, SPATIAL, StandingSetup
    feet = [mech.findbody("r_foot"), mech.findbody("l_foot")]
    st = StandingSetup(mech, feet, mech.findbody("pelvis"))
    for cp in mech.contact_points:
        st.contact_ids.append(ctrl.add_contact(cp.body, cp.position, cp.normal, cp.mu))
    for j in range(mech.nbodies):
        ctrl.regularize(j, 0.05)
    on_foot_path = set()
    for f in feet:
        on_foot_path.update(mech.path_joints(M.WORLD, f))
    jobs = [("foot", f) for f in feet] + [("mom", None), ("pelvis", None)] + \
           [("joint", j) for j in range(mech.nbodies) if mech.joint_type[j] == M.REVOLUTE and j not in on_foot_path]
    jobs = [jobs[i] for i in rng.permutation(len(jobs))]
    off = 0
    for kind, arg in jobs:
        if kind == "foot":
            tid = ctrl.add_task(SPATIAL, base=M.WORLD, target=arg, frame=M.FRAME_DEFAULT, wkind=HARD); st.foot_tasks.append(tid); dim = 6
        elif kind == "mom":
            tid = ctrl.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR, weight=[1.0]); st.linmom_task = tid; dim = 3
        elif kind == "pelvis":
            tid = ctrl.add_task(ANGULAR, base=M.WORLD, target=st.pelvis, frame=M.FRAME_DEFAULT, wkind=HARD); st.pelvis_task = tid; dim = 3
        else:
            tid = ctrl.add_task(JOINT, joint=arg, wkind=HARD); st.joint_tasks[arg] = tid; dim = 1
        st.offsets[tid] = off
        off += dim
    st.n_desired = off
    kin = W.batch_kinematics(mech, nominal_q, np.zeros(mech.nv))
    st.comref = kin["com"][0] - np.array([0.0, 0.0, 0.05])
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.REVOLUTE:
            st.jointrefs[j] = float(nominal_q[mech.configuration_range(j)[0]])
    return st


def _shuffled_vs_plain(factory, B, seed):
    rng = np.random.default_rng(seed)
    cp, wl = W.atlas_workload(factory, B, 3, seed=400 + seed)
    ref = cp.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    m0, m, order = _shuffled_atlas(rng)
    assert order != list(range(m0.nbodies))
    c = factory(m, 4, m.findbody("pelvis"))
    q0n = M.atlas_nominal_configuration(m)
    st = _standing_shuffled_tasks(c, m, q0n, rng)
    c.finalize()
    assert c.L.qpc_debug_presolve_blocks(c.h) == 2  # both foot tasks eliminated whatever the numbering and the task order
    q, v = _map_state(m0, m, wl.q, wl.v)
    des = W.standing_desired(st, q, v)
    r = c.control_batch(q, v, des, wl.contact_weight, wl.contact_maxnormalforce)
    assert np.all(r["iters"] >= 1000000)  # presolved fraction 1.0
    assert np.array_equal(r["status"], ref["status"]) and (ref["status"] == 0).all()
    # tau per joint NAME must agree with the plain model's
    tau_back = np.zeros_like(ref["tau"])
    for j, name in enumerate(m.joint_names):
        tau_back[:, list(m0.velocity_range(m0.findjoint(name)))] = r["tau"][:, list(m.velocity_range(j))]
    scale = np.maximum(1.0, np.abs(ref["tau"]).max(axis=1, keepdims=True))
    assert float(np.abs((tau_back - ref["tau"]) / scale).max()) <= 1e-10


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_presolve_with_shuffled_bodies_and_tasks_emulated(seed):
    emu.build()
    emu.race_report()
    _shuffled_vs_plain(emu.factory, 24, seed)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


@pytest.mark.gpu
def test_presolve_with_shuffled_bodies_and_tasks_gpu():
    _shuffled_vs_plain(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 512, 5)
