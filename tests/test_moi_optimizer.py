"""Solver-level plug: the MOI-optimizer twin (qpcontrol.jl_amd/moi.py, mirrored by
`QPCOptimizer <: MOI.AbstractOptimizer` in julia/QPControlHIPMOI.jl) driven the way Parametron drives its optimizer
(momentum.jl:27,58): copy the 143 x 178 reference-layout Atlas problem ONCE, then per tick rewrite the coefficients in
place (same sparsity pattern) and optimize -- a 20-tick closed-loop sequence, every tick against the oracle's solve of
the same problem; statuses through checkstatus (momentum.jl:83-91)."""
import numpy as np
import pytest

import emu
from oracle import oracle as O
from qpcontrol_jl_amd import moi
from qpcontrol_jl_amd import workloads as W


def _replay(lib_path, n_ticks=20, **opt_kw):
    oc, wl = W.atlas_workload(lambda mech, N, fl: O.OracleController(O.OracleModel(mech), N, fl), 4, 3)
    om = oc.model
    nv = wl.mech.nv
    q, v = wl.q[0].copy(), wl.v[0].copy()
    cw, cm = wl.contact_weight[0], wl.contact_maxnormalforce[0]
    opt = moi.QPCOptimizer(lib_path=lib_path, **opt_kw)
    assert opt.is_empty() and opt.termination_status() == "OPTIMIZE_NOT_CALLED"
    # structural sparsity pattern (Parametron's terms): union of the nonzeros over random states -- an entry that
    # happens to vanish at one state is still a term (1549 stored nonzeros in A, 63 in P)
    rng = np.random.default_rng(0)
    pat_P, pat_A = np.zeros((143, 143), bool), np.zeros((178, 143), bool)
    from qpcontrol_jl_amd import models as M
    for _ in range(6):
        qr, vr = M.rand_state(wl.mech, rng)
        Pr, _, Ar, _, _ = oc.build_qp(qr, vr, rng.normal(size=wl.desired.shape[1]), cw, cm)
        pat_P |= Pr != 0.0
        pat_A |= Ar != 0.0
    assert pat_A.sum() <= 1549 and pat_P.sum() == 63
    worst, total_iters = 0.0, 0
    for k in range(n_ticks):
        des = W.standing_desired(wl.setup, q[None], v[None])[0]
        P, qv, A, l, u = oc.build_qp(q, v, des, cw, cm)
        if k == 0:
            assert P.shape == (143, 143) and A.shape == (178, 143)
            opt.copy_to(P, qv, A, l, u, pattern_P=pat_P, pattern_A=pat_A)  # MOI.copy_to: once
        else:
            opt.set_objective(P, qv, strict_pattern=True)     # update!: coefficients only, fixed sparsity
            opt.set_constraints(A, l, u, strict_pattern=True)
        opt.optimize()
        total_iters += opt.iters
        opt.checkstatus()
        assert (opt.termination_status(), opt.primal_status()) == ("OPTIMAL", "FEASIBLE_POINT")
        r = oc.tick(q, v, des, cw, cm)
        assert r["status"] == 0
        x = opt.variable_primal()
        worst = max(worst, np.abs(x[:nv] - r["x"][:nv]).max())
        np.testing.assert_allclose(x[:nv], r["x"][:nv], atol=1e-7)  # v̇ is unique (regularised); rho of the N = 4 basis is not
        # the optimizer's point is primal feasible for the reference-layout problem and no worse in objective
        Ax = A @ x
        assert np.all(Ax >= l - 1e-6 * (1 + np.abs(l))) and np.all(Ax <= u + 1e-6 * (1 + np.abs(u)))
        assert opt.objective_value() <= 0.5 * r["x"] @ P @ r["x"] + qv @ r["x"] + 1e-6
        # closed loop at 1 kHz with the oracle's RNEA-free integration of the QP's own v̇
        v = v + 1e-3 * x[:nv]
        q = om.integrate(q, v, 1e-3)
    assert opt.n_optimize == n_ticks
    return worst, total_iters


def test_moi_twin_replays_20_ticks_emulated():
    """the prepared entry (qpc_qp_setup once, qpc_qp_solve per tick), warm-started from the previous tick; the same ticks through the
    dense cold entry need more working-set changes"""
    emu.build()
    _, warm_iters = _replay(emu.EMU_LIB)
    _, cold_iters = _replay(emu.EMU_LIB, prepared=False)
    assert warm_iters <= cold_iters, (warm_iters, cold_iters)


def test_moi_twin_reports_failure_like_checkstatus():
    emu.build()
    opt = moi.QPCOptimizer(lib_path=emu.EMU_LIB)
    opt.copy_to(np.eye(2), np.zeros(2), np.array([[1.0, 0.0], [1.0, 0.0]]), np.array([1.0, -1e30]), np.array([1e30, 0.0]))
    opt.optimize()
    assert opt.termination_status() != "OPTIMAL"
    with pytest.raises(moi.QPSolveFailure):
        opt.checkstatus()
    with pytest.raises(ValueError):
        opt.set_constraints(A=np.array([[1.0, 2.0], [1.0, 0.0]]), strict_pattern=True)  # outside the copied pattern
    opt.empty()
    assert opt.is_empty()


@pytest.mark.gpu
def test_moi_twin_replays_20_ticks_gpu():
    worst, iters = _replay(None)
    print(f"max |v̇_L0 - v̇_oracle| over 20 ticks: {worst:.2e}, {iters} working-set changes (prepared entry, warm-started)")
