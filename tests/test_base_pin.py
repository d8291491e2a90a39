"""A hard Angular / Linear / SpatialAccelerationTask of the floating body w.r.t. the world in its own frame is compiled as a PIN of
the floating joint's accelerations (qpc_compile.cpp pass 1: the Jacobian is rows of [I6 0] for every state, the bias vanishes --
reference semantics src/tasks.jl:31-44, 73-84, 112-123 with RBD's body-frame twist coordinates of a QuaternionFloating joint).
QPC_NO_BASE_PIN=1 keeps the rows in the QP: both forms must give the same tick (1e-9), and the pinned form must really be smaller."""
import os

import numpy as np
import pytest

import emu
from qpcontrol_jl_amd import binding, models as M
from qpcontrol_jl_amd import workloads as W
from qpcontrol_jl_amd.workloads import ANGULAR, HARD, JOINT, LINEAR, SPATIAL


def _with_env(flag, fn):
    old = os.environ.pop("QPC_NO_BASE_PIN", None)
    if flag:
        os.environ["QPC_NO_BASE_PIN"] = "1"
    try:
        return fn()
    finally:
        os.environ.pop("QPC_NO_BASE_PIN", None)
        if old is not None:
            os.environ["QPC_NO_BASE_PIN"] = old


def _atlas(factory, B, config, fused):
    def run(flag):
        c, wl = _with_env(flag, lambda: W.atlas_workload(factory, B, config, fused_laws=fused))
        r = c.control_batch(wl.q, wl.v, None if fused else wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
        return c.layout, r
    (lp, rp), (lr, rr) = run(False), run(True)
    assert (lp.n_free, lp.n_pinned, lp.n_eq_rows) == (15, 21, 21)   # pelvis rows pinned: 18 -> 15 free columns, 24 -> 21 rows
    assert (lr.n_free, lr.n_pinned, lr.n_eq_rows) == (18, 18, 24)
    assert lp.lds_bytes_per_qp < lr.lds_bytes_per_qp
    assert np.array_equal(rp["status"] == 0, rr["status"] == 0)
    ok = rp["status"] == 0
    assert ok.mean() > 0.8
    scale = np.maximum(1.0, np.abs(rr["tau"][ok]).max(axis=1, keepdims=True))
    assert np.abs((rp["tau"][ok] - rr["tau"][ok]) / scale).max() < 1e-9
    assert np.abs(rp["vdot"][ok] - rr["vdot"][ok]).max() < 1e-8


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("config", [3, 5])
def test_pelvis_task_is_a_pin_emulated(config, fused):
    emu.build()
    emu.race_report()
    _atlas(emu.factory, 24, config, fused)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def _tree(factory, kind, flag):
    """a floating tree with contacts whose base carries a hard task of the given kind in its own frame"""
    def make():
        types = [M.QUAT_FLOATING, M.REVOLUTE, M.REVOLUTE, M.PRISMATIC, M.REVOLUTE, M.REVOLUTE]
        mech = M.rand_tree(types, seed=21)
        c = factory(mech, 4, 0)
        for k, b in enumerate((3, 5)):
            c.add_contact(b, [0.1 * k, 0.05, -0.1], [0.0, 0.1, 1.0], 0.8)
        for j in range(mech.nbodies):
            c.regularize(j, 0.2)
        t = c.add_task(kind, base=M.WORLD, target=0, frame=M.FRAME_DEFAULT, wkind=HARD)
        c.add_task(JOINT, joint=2, wkind=HARD)
        c.finalize()
        return mech, c, t
    mech, c, t = _with_env(flag, make)
    rng = np.random.default_rng(4)
    B = 6
    qs, vs = zip(*(M.rand_state(mech, rng, 0.6) for _ in range(B)))
    des = rng.normal(size=(B, c.layout.n_desired))
    r = c.control_batch(np.array(qs), np.array(vs), des, np.full((B, 2), 1e-2), np.full((B, 2), 1e5))
    return c.layout, r, des, c.task_desired_range(t)


@pytest.mark.parametrize("kind,v0", [(ANGULAR, 0), (LINEAR, 3), (SPATIAL, 0)])
def test_base_tasks_of_all_three_kinds_emulated(kind, v0):
    emu.build()
    lp, rp, des, (o, n) = _tree(emu.factory, kind, False)
    lr, rr, _, _ = _tree(emu.factory, kind, True)
    assert lp.n_free == lr.n_free - n and lp.n_eq_rows == lr.n_eq_rows - n
    # (two contact points cannot realise every random base acceleration: such QPs are INFEASIBLE either way)
    assert np.array_equal(rp["status"] == 0, rr["status"] == 0)
    ok = rp["status"] == 0
    assert ok.sum() >= 3
    np.testing.assert_allclose(rp["vdot"][ok, v0:v0 + n], des[ok, o:o + n], rtol=0, atol=1e-12)  # pinned: exactly the desired value
    np.testing.assert_allclose(rp["tau"][ok], rr["tau"][ok], rtol=0, atol=1e-8)
    np.testing.assert_allclose(rp["vdot"][ok], rr["vdot"][ok], rtol=0, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, pytest.param(True, marks=pytest.mark.gpu_slow)])
def test_pelvis_task_is_a_pin_gpu(fused):
    _atlas(lambda mech, N, fl: binding.Controller(binding.Model(mech, binding.EXPERIMENTS_LIB), N, fl), 512, 3, fused)


def test_floating_joint_on_a_moving_parent_is_not_pinned_emulated():
    """the pin needs the floating joint to hang on the world: on a moving platform the platform's joints enter the task's Jacobian, so
    the task stays a row block of the QP -- and the tick still agrees with the oracle"""
    from oracle import oracle as O
    emu.build()
    mech = M.rand_tree([M.REVOLUTE, M.QUAT_FLOATING, M.REVOLUTE, M.REVOLUTE], seed=5, chain=True)

    def make(factory):
        c = factory(mech, 4, 1)
        for k, b in enumerate((2, 3)):
            c.add_contact(b, [0.1 * k, 0.05, -0.1], [0.0, 0.1, 1.0], 0.8)
        for j in range(mech.nbodies):
            c.regularize(j, 0.2)
        c.add_task(ANGULAR, base=M.WORLD, target=1, frame=M.FRAME_DEFAULT, wkind=HARD)
        c.add_task(JOINT, joint=0, wkind=HARD)
        return c, c.finalize()
    c, lay = make(emu.factory)
    assert lay.n_pinned == 1  # the platform's joint task only
    oc, _ = make(lambda m, N, fl: O.OracleController(O.OracleModel(m), N, fl))
    rng = np.random.default_rng(1)
    B = 4
    qs, vs = zip(*(M.rand_state(mech, rng, 0.5) for _ in range(B)))
    des = rng.normal(size=(B, lay.n_desired)) * 0.3
    args = (np.array(qs), np.array(vs), des, np.full((B, 2), 1e-2), np.full((B, 2), 1e5))
    r, ro = c.control_batch(*args), oc.tick_batch(*args)
    assert (r["status"] == 0).all() and (ro["status"] == 0).all()
    np.testing.assert_allclose(r["tau"], ro["tau"], rtol=0, atol=1e-8)
