#!/usr/bin/env python3
"""Seeded fuzzing of L0 beyond the suite: random structured QPs (diagonal P with zero-cost variables, or with dense blocks of up to six variables; sparse A of several densities,
equality rows anywhere, one-sided rows, structural zeros in the pattern) through the dense entry, the prepared entry and the prepared
entry warm-started, against the oracle's OSQP restatement.
  python tests/offline/fuzz_l0.py <seed> <count> [gpu]     (default: the CPU emulation; QPC_EMU_WG_LDS=3000 moves the matrix to global memory)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding  # noqa: E402
from oracle import oracle as O  # noqa: E402
import emu  # noqa: E402

LIB = None if (len(sys.argv) > 3 and sys.argv[3] == "gpu") else emu.build()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = known = limit = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 200):
    big = rng.uniform() < 0.1
    n = int(rng.integers(1, 150 if big else 40)); m = int(rng.integers(1, 200 if big else 60))
    ne = int(rng.integers(0, min(n, m) + 1)) if rng.uniform() < 0.8 else 0
    dens = rng.choice([0.05, 0.15, 0.4, 1.0])
    P = np.diag(rng.uniform(0.05, 2.0, size=n))
    if rng.uniform() < 0.3:
        P[rng.integers(0, n), :] = 0  # a zero-cost variable
    if rng.uniform() < 0.3 and n >= 4:  # dense positive definite blocks on the diagonal (matrix-weighted tasks, momentum.jl:113-117): the sparse form's block-diagonal P
        free = [i for i in rng.permutation(n) if P[i, i] > 0.0]
        for _ in range(int(rng.integers(1, 4))):
            bs = int(rng.integers(2, 7))
            if len(free) < bs:
                break
            idx = np.array(sorted(free[:bs])); free = free[bs:]
            M = rng.normal(size=(bs, bs))
            P[np.ix_(idx, idx)] = M @ M.T / bs + np.diag(rng.uniform(0.1, 1.0, size=bs))
    q = rng.normal(size=n)
    q[np.diag(P) == 0.0] = 0.0  # (a zero-cost variable with a linear term and no row is an unbounded QP, not a test)
    A = rng.normal(size=(m, n)) * (rng.uniform(size=(m, n)) < dens)
    if rng.uniform() < 0.5:
        A[np.arange(m), rng.integers(0, n, size=m)] += rng.uniform(0.5, 1.5, size=m)
    x0 = rng.normal(size=n)
    b = A @ x0
    l, u = b - rng.uniform(0.0, 0.6, size=m), b + rng.uniform(0.0, 0.6, size=m)
    l[:ne] = u[:ne] = b[:ne]
    k = rng.integers(0, 3)
    if k == 0: l[ne::3] = -1e30
    if k == 1: u[ne::2] = 1e30
    perm = rng.permutation(m)  # equality rows anywhere
    A, l, u = A[perm], l[perm], u[perm]
    ro = O.qp_solve(P, q, A, l, u)
    rd = binding.qp_solve_batch(P, q, A, l, u, lib_path=LIB)
    pat = (A != 0) | (rng.uniform(size=A.shape) < 0.02)
    pr = binding.QpProblem(n, m, np.eye(n, dtype=bool) | (P != 0), pat, lib_path=LIB)
    Pv, Av = pr.values(P, A)
    rp = pr.solve(Pv, q, Av, l, u)
    rw = pr.solve(Pv, q, Av, l, u, warm_start=True)
    pr.close()
    ok = True
    msg = []
    blocks = bool(np.any(P - np.diag(np.diag(P)) != 0.0))
    # (a P with blocks goes to the one-lane form behind the dense entry -- the general method, weaker on zero-cost variables --: there the
    #  prepared entry, the sparse workgroup form, is held against the oracle alone)
    zero_cost = bool(np.any(np.diag(P) == 0.0))
    # (... and with a zero-cost variable -- A H^-1 A' conditioned like 1 / eps, the known limit below -- the two entries eliminate the
    #  equality rows in different orders: which of them still resolves such a problem is luck, what counts is that neither is silently wrong)
    if (rd["status"][0] != rp["status"][0] and not blocks and not (zero_cost and 0 in (rd["status"][0], rp["status"][0]))) or (rw["status"][0] != rp["status"][0] and not (rp["status"][0] == 4 and rw["status"][0] == 0)):  # (a search that ended at its iteration limit goes on from where it stopped when it is warm-started: 4 then 0 is no contradiction)
        ok = False; msg.append(f"status dense {rd['status'][0]} prep {rp['status'][0]} warm {rw['status'][0]}")
    if ro["status"] == 0:
        if rp["status"][0] != 0:
            # known limit of the range-space method: a zero-cost variable shared by several equality rows makes
            # A H^-1 A' ill-conditioned like 1 / eps; such a QP may end INFEASIBLE / NUMERICAL_ERROR -- reported, never silently wrong
            if np.any(np.diag(P) == 0.0):
                known += 1
            elif rp["status"][0] == 4 and rd["status"][0] == 4 and m - ne > 128:
                limit += 1  # (more than 128 dense inequality rows: the working-set search ends at its iteration limit -- reported as such)
            else:
                ok = False; msg.append(f"oracle optimal, kernel {rp['status'][0]}")
        else:
            # (zero-cost variables can leave x non-unique: what is checked is that the point is feasible and as good as the oracle's;
            #  x itself where the problem is strictly convex)
            fo = 0.5 * ro["x"] @ P @ ro["x"] + q @ ro["x"]
            def certified(r):
                # the oracle's ADMM point may sit 1e-10 outside a row whose multiplier is 1e4 and be "better" by 1e-6 (seed 7208, trial 521): a
                # device point that differs stands if it satisfies the KKT conditions by itself -- feasible, stationary with its own
                # multipliers, multipliers only on active rows and of the right sign
                xx, yy = r["x"][0], r["y"][0]
                Ax = A @ xx
                sc = 1.0 + max(np.abs(q).max(initial=0.0), np.abs(yy).max(initial=0.0))
                feas = max(np.max(l - Ax, initial=0.0), np.max(Ax - u, initial=0.0)) <= 1e-9 * (1.0 + np.abs(Ax).max(initial=0.0))
                stat = np.abs(P @ xx + q + A.T @ yy).max(initial=0.0) <= 1e-9 * sc
                act_lo, act_hi = np.abs(Ax - l) <= 1e-7 * (1 + np.abs(l)), np.abs(Ax - u) <= 1e-7 * (1 + np.abs(u))
                return bool(feas and stat and not np.any((yy > 1e-9 * sc) & ~act_hi) and not np.any((yy < -1e-9 * sc) & ~act_lo))
            for name, r in (("dense", rd), ("prep", rp), ("warm", rw)):
                if r["status"][0] != 0:
                    continue  # (a disagreement of the statuses is judged above)
                if certified(r):
                    continue  # (a KKT point of a convex QP is a minimiser: no feasible point has a lower objective, whatever the oracle's says;
                              #  where P is positive definite it is THE minimiser)
                xx = r["x"][0]
                Ax = A @ xx
                viol = max(np.max(l - Ax, initial=0.0), np.max(Ax - u, initial=0.0))
                f = 0.5 * xx @ P @ xx + q @ xx
                if viol > 1e-7 * (1 + np.abs(Ax).max(initial=0.0)):
                    ok = False; msg.append(f"{name} infeasible by {viol:.2e}")
                Ao = A @ ro["x"]
                viol_o = max(np.max(l - Ao, initial=0.0), np.max(Ao - u, initial=0.0))  # (the oracle's ADMM point may sit 1e-8 outside: then it can be "better")
                if viol_o < 1e-10 and f > fo + 1e-7 * (1 + abs(fo)):
                    ok = False; msg.append(f"{name} objective {f:.9e} > oracle {fo:.9e}")
                if np.all(np.diag(P) > 0) and viol_o < 1e-10:
                    e = np.abs(xx - ro["x"]).max()
                    if e > 2e-7:
                        ok = False; msg.append(f"{name} x err {e:.2e}")
    if not ok:
        bad += 1
        print("trial", trial, "n m ne", n, m, ne, "dens", dens, "oracle", ro["status"], msg, flush=True)
print("done, bad =", bad, "| not solved because of a zero-cost variable (known limit):", known, "| iteration limit on dense inequality-only problems:", limit)
