"""Solver-level plug: the Python twin of `QPCOptimizer <: MOI.AbstractOptimizer`
(qpcontrol.jl_amd/julia/QPControlHIP.jl).

The reference hands its optimizer to Parametron (`MomentumBasedController{N}(mechanism, optimizer)`,
src/lowlevel/momentum.jl:15-16,27; built as in test/runtests.jl:35-43).  Parametron copies the problem to the
optimizer ONCE (`MOI.copy_to`), then before every `MOI.optimize!` rewrites objective and constraint COEFFICIENTS in
place -- same sparsity pattern, new numbers (momentum.jl:58 `solve!` -> `update!`) -- and afterwards reads
`VariablePrimal`, `TerminationStatus`, `PrimalStatus`, `DualStatus` (momentum.jl:62-64, 83-91).

This class keeps that life cycle and maps it onto the PREPARED L0 entry of the C ABI (include/qpc.h):
  MOI.copy_to   -> qpc_qp_setup   (the sparsity pattern: compressed columns / rows and the structure of A H^-1 A', once)
  MOI.optimize! -> qpc_qp_solve   (B = 1: the values in the pattern's order -- 13 KB for the 143 x 178 Atlas problem --, the
                                   working set warm-started from the previous tick's multipliers, as OSQP warm-starts)
  MOI.empty!    -> qpc_qp_free
`prepared=False` keeps the dense entry `qpc_qp_solve_batch` (368 KB per solve, cold).
    min 1/2 x'Px + q'x   s.t.  l <= A x <= u.
It is plumbing / parity (assembly stays on the host); the performance path is the L2 tick."""
from __future__ import annotations

import numpy as np

from . import binding

# MOI.TerminationStatusCode / ResultStatusCode names used by checkstatus (momentum.jl:83-91)
_STATUS = {
    0: ("OPTIMAL", "FEASIBLE_POINT", "FEASIBLE_POINT"),
    1: ("ALMOST_OPTIMAL", "UNKNOWN_RESULT_STATUS", "UNKNOWN_RESULT_STATUS"),
    2: ("INFEASIBLE", "NO_SOLUTION", "INFEASIBILITY_CERTIFICATE"),
    3: ("DUAL_INFEASIBLE", "INFEASIBILITY_CERTIFICATE", "NO_SOLUTION"),
    4: ("ITERATION_LIMIT", "UNKNOWN_RESULT_STATUS", "UNKNOWN_RESULT_STATUS"),
    5: ("NUMERICAL_ERROR", "UNKNOWN_RESULT_STATUS", "UNKNOWN_RESULT_STATUS"),
}


class QPCOptimizer:
    """Dense QP optimizer with MOI's copy-once / modify-coefficients / optimize / query life cycle."""

    def __init__(self, device_id=0, lib_path=None, prepared=True, warm_start=True, **settings):
        self.device_id, self.lib_path, self.settings = device_id, lib_path, settings
        self.prepared, self.warm_start = prepared, warm_start
        self._problem = None
        self.empty()

    # ---- MOI.is_empty / MOI.empty!
    def is_empty(self):
        return self.n == 0

    def empty(self):
        if self._problem is not None:
            self._problem.close()
            self._problem = None
        self.n = self.m = 0
        self.P = self.q = self.A = self.l = self.u = None
        self._pattern_P = self._pattern_A = None
        self.x = self.y = None
        self.raw_status = None
        self.n_optimize = 0

    # ---- MOI.copy_to: the problem with its (fixed) sparsity pattern
    def copy_to(self, P, q, A, l, u, pattern_P=None, pattern_A=None):
        """pattern_*: boolean masks of the STRUCTURAL nonzeros (the terms present in the MOI functions, explicit zeros
        included); default: the numerically nonzero entries of P / A."""
        P, A = np.array(P, dtype=np.float64), np.array(A, dtype=np.float64)
        self.n, self.m = P.shape[0], A.shape[0]
        assert P.shape == (self.n, self.n) and A.shape == (self.m, self.n)
        self.P, self.q, self.A = P, np.array(q, dtype=np.float64), A
        self.l, self.u = np.array(l, dtype=np.float64), np.array(u, dtype=np.float64)
        self._pattern_P = (P != 0.0) if pattern_P is None else np.asarray(pattern_P, dtype=bool)
        self._pattern_A = (A != 0.0) if pattern_A is None else np.asarray(pattern_A, dtype=bool)
        self.x = self.y = None
        self.raw_status = None
        if self._problem is not None:
            self._problem.close()
            self._problem = None
        if self.prepared and self.m > 0:
            self._problem = binding.QpProblem(self.n, self.m, self._pattern_P, self._pattern_A, device_id=self.device_id,
                                              lib_path=self.lib_path)
            self._problem.analyze(self.l, self.u)  # (the constraint sets are known at copy_to: the plan of the sparse solver now, not at the first tick)

    # ---- MOI.set(ObjectiveFunction) / MOI.set(ConstraintFunction / ConstraintSet): coefficient updates, pattern fixed
    def set_objective(self, P=None, q=None, strict_pattern=False):
        if P is not None:
            P = np.asarray(P, dtype=np.float64)
            if strict_pattern and np.any((P != 0.0) & ~self._pattern_P):
                raise ValueError("objective update outside the sparsity pattern given to copy_to")
            self.P[...] = P
        if q is not None:
            self.q[...] = q

    def set_constraints(self, A=None, l=None, u=None, strict_pattern=False):
        if A is not None:
            A = np.asarray(A, dtype=np.float64)
            if strict_pattern and np.any((A != 0.0) & ~self._pattern_A):
                raise ValueError("constraint update outside the sparsity pattern given to copy_to")
            self.A[...] = A
        if l is not None:
            self.l[...] = l
        if u is not None:
            self.u[...] = u

    def modify_coefficient(self, row, col, value):
        """MOI.modify(ScalarCoefficientChange): one entry of A"""
        self.A[row, col] = value

    # ---- MOI.optimize!
    def optimize(self):
        if self.is_empty():
            raise RuntimeError("optimize on an empty optimizer")
        if self._problem is not None:
            outside = ((self.P != 0.0) & ~(self._pattern_P | self._pattern_P.T)).any() or ((self.A != 0.0) & ~self._pattern_A).any()
            if outside:  # a coefficient outside the copied pattern (MOI would need a new copy_to): the dense entry takes it
                r = binding.qp_solve_batch(self.P, self.q, self.A, self.l, self.u, device_id=self.device_id, lib_path=self.lib_path,
                                           **self.settings)
            else:
                Pv, Av = self._problem.values(self.P, self.A)
                r = self._problem.solve(Pv, self.q, Av, self.l, self.u, warm_start=self.warm_start and self.n_optimize > 0,
                                        **self.settings)
        else:
            r = binding.qp_solve_batch(self.P, self.q, self.A, self.l, self.u, device_id=self.device_id, lib_path=self.lib_path,
                                       **self.settings)
        self.x, self.y = r["x"][0], r["y"][0]
        self.raw_status = int(r["status"][0])
        self.iters = int(r["iters"][0])
        self.n_optimize += 1

    # ---- MOI.get
    def variable_primal(self):
        return self.x

    def constraint_dual(self):
        return self.y

    def termination_status(self):
        return "OPTIMIZE_NOT_CALLED" if self.raw_status is None else _STATUS[self.raw_status][0]

    def primal_status(self):
        return "NO_SOLUTION" if self.raw_status is None else _STATUS[self.raw_status][1]

    def dual_status(self):
        return "NO_SOLUTION" if self.raw_status is None else _STATUS[self.raw_status][2]

    def objective_value(self):
        return 0.5 * self.x @ self.P @ self.x + self.q @ self.x

    def checkstatus(self):
        """momentum.jl:83-91: OPTIMAL + FEASIBLE_POINT, or ALMOST_OPTIMAL + UNKNOWN_RESULT_STATUS; else QPSolveFailure"""
        t, p = self.termination_status(), self.primal_status()
        ok = (t == "OPTIMAL" and p == "FEASIBLE_POINT") or (t == "ALMOST_OPTIMAL" and p == "UNKNOWN_RESULT_STATUS")
        if not ok:
            raise QPSolveFailure(t, p, self.dual_status())


class QPSolveFailure(RuntimeError):
    """exceptions.jl:1-11"""

    def __init__(self, termination, primal, dual):
        super().__init__(f"QP solve unsuccessful. Termination status: {termination}. Primal status: {primal}. Dual status: {dual}.")
        self.termination, self.primal, self.dual = termination, primal, dual
