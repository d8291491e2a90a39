"""regularize!(c, joint, weight) with a Parametron `Parameter` as the weight (/root/reference/src/lowlevel/momentum.jl:128-131 takes an
untyped `weight`; a Parameter is re-evaluated every tick) = `qpc_regularize_per_qp`: the weight is entry i of the per-QP
`task_weights` record.  Checked against the oracle (whose reference-layout P gets 2 (w_static + w_per_qp) on the diagonal) at 1e-8,
against the static form (per-QP weights that happen to be the constant reproduce `qpc_regularize`) at 1e-10, and the three fields of
QPSolveFailure (exceptions.jl:1-11) through qpc_status_triple."""
import numpy as np
import pytest

import emu
import parity
import scenarios as SC
from oracle import oracle as O
from qpcontrol_jl_amd import api, binding, models as M
from qpcontrol_jl_amd import workloads as W


def _ofactory(mech, N, fl):
    return O.OracleController(O.OracleModel(mech), N, fl)


def _gpu_factory(mech, N, fl):
    return binding.Controller(binding.Model(mech), N, fl)


def _case(factory, B, static_too=True, seed=31):
    """Atlas config 3 (contacts, floating base) whose arm and back joints carry per-QP regularisation weights on top of the
    static 0.05 of the standing controller; a leg joint (a column the null-space presolve eliminates) as well"""
    mech = M.atlas()
    fl = mech.findbody("pelvis")
    c = factory(mech, 4, fl)
    st = W.standing_setup(c, mech, M.atlas_nominal_configuration(mech))
    joints = [mech.findjoint(n) for n in ("l_leg_kny", "r_leg_hpx", "back_bkz", "l_arm_shx")]
    idx = [c.regularize_per_qp(j) for j in joints]
    assert idx == list(range(len(joints)))
    c.finalize()
    q, v = W.standing_states(mech, B, seed)
    des = W.standing_desired(st, q, v)
    nc = len(st.contact_ids)
    rng = np.random.default_rng(seed + 1)
    tw = rng.uniform(0.0, 0.5, size=(B, len(joints)))
    tw[0] = 0.0  # no extra weight at all: the plain standing controller
    return SC.Case("atlas_reg_per_qp", mech, c, q, v, des, np.full((B, nc), 1e-3), np.full((B, nc), 1e6), tw=tw, floating=fl), joints


def _tick(case):
    return case.ctrl.control_batch(case.q, case.v, case.desired, case.cweight, case.cmaxf, task_weights=case.tw)


def _check_vs_oracle(factory, B):
    g, _ = _case(factory, B)
    o, _ = _case(_ofactory, B)
    rg = _tick(g)
    ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw)
    n = parity.compare(g, rg, o, ro)
    assert n >= B // 2, n
    # the weights matter: the same states without them give other torques
    g0, _ = _case(factory, B)
    r0 = g0.ctrl.control_batch(g0.q, g0.v, g0.desired, g0.cweight, g0.cmaxf, task_weights=np.zeros_like(g0.tw))
    assert np.abs(r0["tau"][1:] - rg["tau"][1:]).max() > 1e-3
    np.testing.assert_allclose(r0["tau"][0], rg["tau"][0], atol=1e-10)


def _check_vs_static(factory, B=6):
    """per-QP weights equal to a constant == qpc_regularize with that constant added"""
    g, joints = _case(factory, B)
    w = 0.125
    rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, task_weights=np.full_like(g.tw, w))
    mech = M.atlas()
    c = factory(mech, 4, g.floating)
    W.standing_setup(c, mech, M.atlas_nominal_configuration(mech))
    for j in joints:
        c.regularize(j, w)
    c.finalize()
    rs = c.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf)
    assert np.array_equal(rg["status"], rs["status"]) and (rs["status"] == 0).all()
    scale = np.maximum(1.0, np.abs(rs["tau"]).max(axis=1, keepdims=True))
    assert np.abs((rg["tau"] - rs["tau"]) / scale).max() < 1e-10
    assert np.abs(rg["vdot"] - rs["vdot"]).max() < 1e-9


def test_regularize_per_qp_emulated_vs_oracle():
    emu.build()
    emu.race_report()
    _check_vs_oracle(emu.factory, 8)
    races, uninit, outside, first = emu.race_report()
    assert (races, uninit, outside) == (0, 0, 0), first


def test_regularize_per_qp_equals_static_emulated():
    emu.build()
    _check_vs_static(emu.factory)


def test_regularize_per_qp_rejects_duplicates_and_late_calls():
    mech = M.atlas()
    c = emu.factory(mech, 4, mech.findbody("pelvis"))
    j = mech.findjoint("l_leg_kny")
    assert c.regularize_per_qp(j) == 0
    with pytest.raises(binding.QpcError):
        c.regularize_per_qp(j)
    c.finalize()
    with pytest.raises(binding.QpcError):
        c.regularize_per_qp(mech.findjoint("r_leg_kny"))


def test_status_triple_is_the_reference_failure_triple():
    """exceptions.jl:1-11 / momentum.jl:83-91: checkstatus accepts OPTIMAL + FEASIBLE_POINT and ALMOST_OPTIMAL + UNKNOWN_RESULT_STATUS"""
    assert binding.status_triple(0) == ("OPTIMAL", "FEASIBLE_POINT", "FEASIBLE_POINT")
    assert binding.status_triple(1) == ("ALMOST_OPTIMAL", "UNKNOWN_RESULT_STATUS", "UNKNOWN_RESULT_STATUS")
    assert binding.status_triple(2) == ("INFEASIBLE", "NO_SOLUTION", "INFEASIBILITY_CERTIFICATE")
    assert binding.status_triple(3) == ("DUAL_INFEASIBLE", "INFEASIBILITY_CERTIFICATE", "NO_SOLUTION")
    assert binding.status_triple(4)[0] == "ITERATION_LIMIT" and binding.status_triple(5)[0] == "NUMERICAL_ERROR"
    assert binding.status_triple(17)[0] == "UNKNOWN"
    for code in range(6):  # the first field is qpc_status_string
        assert binding.status_triple(code)[0] == binding.STATUS_NAMES[code]
    e = api.QPSolveFailure(3, 2)
    assert (e.terminationstatus, e.primalstatus, e.dualstatus) == ("INFEASIBLE", "NO_SOLUTION", "INFEASIBILITY_CERTIFICATE")
    assert "Termination status: INFEASIBLE" in str(e) and "Dual status: INFEASIBILITY_CERTIFICATE" in str(e)
    # the MOI twin of the optimizer reports the same triple
    from qpcontrol_jl_amd import moi
    for code, t in moi._STATUS.items():
        assert t == binding.status_triple(code)


@pytest.mark.gpu
def test_regularize_per_qp_gpu_vs_oracle():
    _check_vs_oracle(_gpu_factory, 64)


@pytest.mark.gpu
def test_regularize_per_qp_equals_static_gpu():
    _check_vs_static(_gpu_factory, 32)
