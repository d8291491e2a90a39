"""Shared comparison of a backend's batched tick result with the oracle's."""
import numpy as np

import kkt_check


def compare(g, rg, o, ro, check_floating_zero=True):
    """g/o: scenario Cases of the backend under test / of the oracle; rg/ro their results.
    * where the oracle is conclusive (OPTIMAL or INFEASIBLE) the statuses must agree and the
      unique quantities (tau, v̇ when regularised, contact wrenches) must match to g.tol_tau;
    * where the oracle's ADMM (like the reference's OSQP) stalls at max_iter, the backend's
      OPTIMAL answer must carry an independent KKT certificate instead."""
    so, sg = ro["status"], rg["status"]
    conclusive = (so == 0) | (so == 2)
    assert np.array_equal((sg == 0)[conclusive], (so == 0)[conclusive]), (sg, so)
    ok = so == 0
    certified = np.where(~conclusive & (sg == 0))[0]
    assert ok.any() or len(certified), "nothing to compare: no OPTIMAL element on either side"
    for b in certified:
        c = kkt_check.certify(o.ctrl, o.q[b], o.v[b], o.desired[b], o.cweight[b], o.cmaxf[b], rg["vdot"][b],
                              rg["cwrench"][b], None if o.tw is None else o.tw[b], None if o.surf is None else o.surf[b])
        assert c["ok"], (g.name, b, c["primal"], c["stationarity"])
    if not ok.any():
        pass  # every OPTIMAL answer of the backend was certified independently above
    elif np.isfinite(g.tol_tau):
        np.testing.assert_allclose(rg["tau"][ok], ro["tau"][ok], atol=g.tol_tau, rtol=0)
        np.testing.assert_allclose(rg["cwrench"][ok], ro["cwrench"][ok], atol=100 * g.tol_tau, rtol=0)
        np.testing.assert_allclose(rg["vdot"][ok], ro["vdot"][ok], atol=g.tol_tau, rtol=0)
    else:  # unregularised soft modes: v̇ is not unique; the achieved task acceleration is
        om = o.ctrl.model
        for b in range(g.B):
            om.set_state(o.q[b], o.v[b])
            J, bias = o.ctrl.task_eval(0)
            np.testing.assert_allclose(J @ rg["vdot"][b] + bias, g.desired[b], atol=1e-8)
    if check_floating_zero and g.floating >= 0:
        fr = list(g.mech.velocity_range(g.floating))
        assert np.all(rg["tau"][:, fr] == 0.0)  # zero_floating_joint_torques!, momentum.jl:93-97
    return int(ok.sum())
