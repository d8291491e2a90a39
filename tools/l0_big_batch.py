#!/usr/bin/env python3
"""One-off robustness check of the solver-level plug on a BIG batch with a few QPs the sparse form leaves (other equality rows): 65 536
Atlas problems, three of them odd, host pointers; the dense forms behind the sparse one run over chunks of 2 048 QPs, so the device memory they
need does not grow with the batch.   python tools/l0_big_batch.py [B]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
import bench_latency as BL  # noqa: E402
from qpcontrol_jl_amd import binding  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
P, q, A, l, u, x, nv = BL.load_l0_fixture("diag")
k = P.shape[0]
idx = np.arange(B) % k
pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0))
Pv, Av = pr.values(P, A)
lc, uc = np.clip(l, -1e20, 1e20)[idx].copy(), np.clip(u, -1e20, 1e20)[idx].copy()
row = int(np.where(l[0] < u[0] - 1.0)[0][0])
odd = [7, B // 2 + 1, B - 3]
for b in odd:
    uc[b, row] = lc[b, row] = 0.5 * (max(l[idx[b], row], -10.0) + min(u[idx[b], row], 10.0))
a = (np.ascontiguousarray(Pv[idx]), np.ascontiguousarray(q[idx]), np.ascontiguousarray(Av[idx]), lc, uc)
t0 = time.perf_counter()
r = pr.solve_raw(*a)
t1 = time.perf_counter()
r2 = pr.solve_raw(*a, out=r)
t2 = time.perf_counter()
plain = ~np.isin(np.arange(B), odd)
print(json.dumps({"B": B, "first_call_s": t1 - t0, "second_call_s": t2 - t1, "problems_per_s": B / (t2 - t1),
                  "optimal_of_the_plain": int((r["status"][plain] == 0).sum()), "plain": int(plain.sum()), "status_of_the_odd": [int(r["status"][b]) for b in odd],
                  "max_abs_err_vdot_vs_fixture": float(np.abs(r["x"][plain][:, :nv] - x[idx][plain][:, :nv]).max())}))
