#!/usr/bin/env python3
"""Latency of both entries on the Valkyrie-sized humanoid (models.valkyrie(): 59 bodies, nv = 64, the kernel's limits; the
StandingController task set with 8 contact points): the tick (qpc_control_batch, host pointers, B = 1 and 2 048) and the solver-level
plug (qpc_qp_setup / qpc_qp_solve on its 171 x 206 reference-layout QP, built by the oracle-independent... the problems come from
tests/test_valkyrie.py, which uses the ORACLE to build them: this tool is test infrastructure, not product).  One JSON line.
    python tools/valkyrie_latency.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding  # noqa: E402
import test_valkyrie as TV  # noqa: E402


def main():
    out = {"robot": "models.valkyrie(): 59 bodies, nv 64", "unit": "us"}
    mech, c, q, v, des, cw, cm = TV.valkyrie_standing(TV.gpu_factory, 2048)
    lay = c.layout
    out["tick"] = {"lds_bytes_per_qp": int(lay.lds_bytes_per_qp), "n_rho": int(lay.n_rho), "n_eq_rows": int(lay.n_eq_rows)}
    for B in (1, 2048):
        a = tuple(binding.pinned_array(c.L, x[:B].shape) for x in (q, v, des, cw, cm))
        for d, x in zip(a, (q, v, des, cw, cm)):
            d[...] = x[:B]
        mk = lambda shape, dt=np.float64: binding.pinned_array(c.L, shape, dt)  # noqa: E731
        o = dict(tau=mk((B, lay.nv)), vdot=mk((B, lay.nv)), cwrench_flat=mk((B, 6 * lay.n_contacts)), status=mk((B,), np.int32), iters=mk((B,), np.int32))
        for _ in range(10):
            r = c.control_batch(*a, out=o)
        ts = []
        for _ in range(200 if B == 1 else 20):
            t0 = time.perf_counter()
            r = c.control_batch(*a, out=o)
            ts.append(time.perf_counter() - t0)
        out["tick"][str(B)] = {"median_us": float(np.median(ts) * 1e6), "solves_per_s": float(B / np.median(ts)), "optimal_fraction": float((r["status"] <= 1).mean()),
                               "mean_iters": float((r["iters"] % 1000).mean())}
    P, qq, A, l, u, x, nv = TV.valkyrie_l0_problems(8)
    pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0))
    Pv, Av = pr.values(P, A)
    info = pr.analyze(l[0], u[0])
    lc, uc = np.clip(l, -1e20, 1e20), np.clip(u, -1e20, 1e20)
    per = []
    for b in range(P.shape[0]):
        ar = tuple(np.ascontiguousarray(z[b:b + 1]) for z in (Pv, qq, Av, lc, uc))
        ob = pr.solve_raw(*ar)
        tc, tw = [], []
        for _ in range(10):
            t0 = time.perf_counter(); r = pr.solve_raw(*ar, out=ob); tc.append(time.perf_counter() - t0)
        it = int(r["iters"][0]); err = float(np.abs(r["x"][0, :nv] - x[b, :nv]).max())
        for _ in range(10):
            t0 = time.perf_counter(); rw = pr.solve_raw(*ar, out=ob, warm_start=True); tw.append(time.perf_counter() - t0)
        per.append({"iters": it, "cold_median_us": float(np.median(tc) * 1e6), "warm_median_us": float(np.median(tw) * 1e6), "max_abs_err_vdot_vs_oracle": err,
                    "status": int(r["status"][0])})
    out["l0"] = {"n": int(P.shape[1]), "m": int(A.shape[1]), "plan": info, "B1_per_problem": per,
                 "B1_cold_median_over_problems_us": float(np.median([p["cold_median_us"] for p in per])),
                 "B1_cold_worst_problem_us": float(max(p["cold_median_us"] for p in per)),
                 "B1_warm_median_over_problems_us": float(np.median([p["warm_median_us"] for p in per]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
