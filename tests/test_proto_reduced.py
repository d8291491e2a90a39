"""The reduced-form algorithm of the HIP kernels (numpy prototype, tests/proto_reduced.py) against
the oracle's reference-layout OSQP path on every scenario: two formulations, two algorithms, one
answer.  Unique quantities only: v̇ (when regularised), contact wrenches, tau, status."""
import numpy as np
import pytest

import proto_reduced as PR
import scenarios as SC
from oracle import oracle as O


def factory(mech, N, fl):
    return PR.Recorder(O.OracleController(O.OracleModel(mech), N, fl), mech, N, fl)


CASES = [c.name for c in SC.all_cases(lambda *a: _Null(), small=True)] if False else None


def _cases():
    return list(SC.all_cases(factory, small=True))


@pytest.fixture(scope="module")
def cases():
    return {c.name: c for c in _cases()}


NAMES = ["fixed_base_hard", "fixed_base_soft", "acrobot_config1", "acrobot_point", "free_fall", "momentum_rate",
         "spatial_constraint", "spatial_scalar", "spatial_scalar_param", "spatial_matrix", "spatial_matrix_param",
         "mixed_tasks", "mixed_tasks_fixed", "atlas_config2", "atlas_config3", "atlas_config5"]


@pytest.mark.parametrize("name", NAMES)
def test_prototype_matches_oracle(cases, name):
    case = cases[name]
    rec = case.ctrl
    n_ok = 0
    for b in range(case.B):
        args = (case.q[b], case.v[b], case.desired[b], case.cweight[b], case.cmaxf[b])
        kw = dict(tw=None if case.tw is None else case.tw[b], surf=None if case.surf is None else case.surf[b])
        ro = rec.inner.tick(*args, **kw)
        rp = PR.tick(rec, *args, **kw)
        assert (ro["status"] == 0) == (rp["status"] == 0), (name, b, ro["status"], rp["status"])
        if ro["status"] != 0:
            continue
        n_ok += 1
        if np.isfinite(case.tol_tau):
            np.testing.assert_allclose(rp["tau"], ro["tau"], atol=case.tol_tau, rtol=0)
            np.testing.assert_allclose(rp["cwrench"], ro["cwrench"], atol=10 * case.tol_tau * 10, rtol=0)
        else:  # non-unique v̇: compare the achieved task acceleration instead
            om = rec.inner.model
            om.set_state(case.q[b], case.v[b])
            J, bias = rec.inner.task_eval(0)
            np.testing.assert_allclose(J @ rp["vdot"] + bias, case.desired[b], atol=1e-8)
            np.testing.assert_allclose(J @ ro["vdot"] + bias, case.desired[b], atol=1e-8)
    assert n_ok > 0


def test_config1_closed_form(cases):
    """Acrobot: v̇ = [0, d]; tau = ID(q, v, v̇)."""
    case = cases["acrobot_config1"]
    rp = PR.tick(case.ctrl, case.q[0], case.v[0], case.desired[0], (), ())
    np.testing.assert_allclose(rp["vdot"], [0.0, 0.7], atol=1e-12)
    om = case.ctrl.inner.model
    om.set_state(case.q[0], case.v[0])
    np.testing.assert_allclose(rp["tau"], om.inverse_dynamics([0.0, 0.7]), atol=1e-12)
