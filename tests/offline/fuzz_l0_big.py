#!/usr/bin/env python3
"""Seeded fuzzing of the solver-level plug NEAR ITS LIMITS: structured QPs with 120 ... 256 variables and rows (the workgroup forms' dimension
limit), many equality rows, sparse A -- plans that fill LDS to the last kilobyte, fronts of every width, working sets that do or do not fit
beside the plan -- through the prepared entry (cold and warm-started) against the oracle's OSQP restatement.
  python tests/offline/fuzz_l0_big.py <seed> <count> [gpu]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding  # noqa: E402
from oracle import oracle as O  # noqa: E402
import emu  # noqa: E402
from test_l0_dense_qp import _random_structured_qp  # noqa: E402



def gen(rng):
    """one problem: (P, q, A, l, u, n, m, ne)"""
    if rng.uniform() < 0.3:
        n = int(rng.integers(120, 257)); m = int(rng.integers(120, 257))
        ne = int(rng.uniform(0.3, 0.95) * min(n, m))
        P, q, A, l, u = _random_structured_qp(rng, n, m, ne)
    else:
        # block-arrow structure (what a controller's QP looks like: contacts that do not touch each other, a few rows that couple them all):
        # groups of variables with their own equality rows, border equality rows over one variable of every group, boxes on some variables
        gs = int(rng.integers(3, 13)); ng = int(rng.integers(120 // gs, 256 // gs + 1)); n = ng * gs
        ge = int(rng.integers(1, gs)); nbord = int(rng.integers(0, 7))
        rows, lo, hi = [], [], []
        x0 = rng.normal(size=n)
        for g in range(ng):
            for _ in range(ge):
                r = np.zeros(n); r[g * gs:(g + 1) * gs] = rng.normal(size=gs) * (rng.uniform(size=gs) < 0.7); r[g * gs + int(rng.integers(0, gs))] += 1.0
                rows.append(r); lo.append(r @ x0); hi.append(r @ x0)
        for _ in range(nbord):
            r = np.zeros(n); r[np.arange(ng) * gs + rng.integers(0, gs, size=ng)] = rng.normal(size=ng)
            rows.append(r); lo.append(r @ x0); hi.append(r @ x0)
        mi = min(256 - len(rows), int(rng.integers(8, 100)))
        for i in rng.permutation(n)[:max(mi, 0)]:
            r = np.zeros(n); r[i] = 1.0
            w = rng.uniform(0.0, 0.5, size=2)
            rows.append(r); lo.append(x0[i] - w[0] if rng.uniform() < 0.8 else -1e30); hi.append(x0[i] + w[1] if rng.uniform() < 0.8 else 1e30)
        A = np.array(rows); l = np.array(lo); u = np.array(hi); m = A.shape[0]; ne = ng * ge + nbord
        P = np.diag(rng.uniform(0.05, 2.0, size=n))
        if rng.uniform() < 0.4:  # a few dense blocks (matrix-weighted tasks)
            for g in rng.permutation(ng)[:3]:
                bs = min(gs, int(rng.integers(2, 7))); idx = np.arange(g * gs, g * gs + bs)
                M_ = rng.normal(size=(bs, bs)); P[np.ix_(idx, idx)] = M_ @ M_.T / bs + np.diag(rng.uniform(0.1, 1.0, size=bs))
        q = rng.normal(size=n) * 2.0
        perm = rng.permutation(m); A, l, u = A[perm], l[perm], u[perm]
    return P, q, A, l, u, n, m, ne


def main():
    LIB = None if (len(sys.argv) > 3 and sys.argv[3] == "gpu") else emu.build()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    bad = planned = solved = 0
    for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 50):
        P, q, A, l, u, n, m, ne = gen(rng)
        ro = O.qp_solve(P, q, A, l, u)
        pr = binding.QpProblem(n, m, P != 0, A != 0, lib_path=LIB)
        Pv, Av = pr.values(P, A)
        t0_ = time.perf_counter()
        rp = pr.solve(Pv[0], q, Av[0], l, u)
        rw = pr.solve(Pv[0], q, Av[0], l, u, warm_start=True)
        if time.perf_counter() - t0_ > 5.0:
            print(f"(trial {trial}: n {n} m {m} ne {ne}, {time.perf_counter() - t0_:.1f} s for the two solves, iterations {int(rp['iters'][0])} / {int(rw['iters'][0])}, plan {pr.plan_info()['in_use']})", flush=True)
        info = pr.plan_info()
        pr.close()
        planned += info["in_use"]
        msg = []
        if rw["status"][0] != rp["status"][0] and not (rp["status"][0] == 4 and rw["status"][0] == 0):
            msg.append(f"status cold {rp['status'][0]} warm {rw['status'][0]}")
        if ro["status"] == 0:
            if rp["status"][0] == 0:
                solved += 1
                e = np.abs(rp["x"][0] - ro["x"]).max()
                ew = np.abs(rw["x"][0] - ro["x"]).max()
                if not (e < 1e-7 and ew < 1e-7):
                    # the oracle's ADMM is not exact on an ill-conditioned problem (seed 15, trial 272: 2e-7 of infeasibility times multipliers of 650):
                    # the device's point stands if it satisfies the KKT conditions by itself
                    def certified(r):
                        xx, yy = r["x"][0], r["y"][0]
                        Ax = A @ xx
                        sc = 1.0 + max(np.abs(q).max(), np.abs(yy).max())
                        feas = max(np.maximum(l - Ax, 0).max(), np.maximum(Ax - u, 0).max()) <= 1e-8 * (1.0 + np.abs(Ax).max())
                        stat = np.abs(P @ xx + q + A.T @ yy).max() <= 1e-8 * sc
                        act_lo, act_hi = np.abs(Ax - l) <= 1e-7 * (1 + np.abs(l)), np.abs(Ax - u) <= 1e-7 * (1 + np.abs(u))
                        signs = not np.any((yy > 1e-8 * sc) & ~act_hi) and not np.any((yy < -1e-8 * sc) & ~act_lo)
                        return feas and stat and signs
                    if not (certified(rp) and certified(rw)):
                        msg.append(f"x differs from the oracle's by {e:.2e} (warm {ew:.2e}) and does not certify itself")
            elif rp["status"][0] not in (3, 4):  # NUMERICAL_ERROR / ITERATION_LIMIT are honest; INFEASIBLE on a solvable problem is not
                msg.append(f"oracle solved it, status {rp['status'][0]}")
        elif ro["status"] == 2 and rp["status"][0] == 0:
            msg.append("oracle INFEASIBLE, device OPTIMAL")
        if msg:
            bad += 1
            print(f"trial {trial} n {n} m {m} ne {ne} plan {info['in_use']} lds {info['lds_bytes']}: " + "; ".join(msg), flush=True)
    print(f"done, bad = {bad} | plans in use {planned} | solved and compared {solved}", flush=True)


if __name__ == "__main__":
    main()
