#!/usr/bin/env python3
"""Where the host side of a B = 1 qpc_qp_solve spends its time: the experiments library with QPC_L0_HOST_PROF=1 prints the time between
the stages of one call (stderr); this script makes a few warm calls on the closed-loop fixture and also times an empty call through
ctypes (B = 0 returns after the argument checks).    QPC_LIB=.../libqpcontrol_hip_exp.so QPC_L0_HOST_PROF=1 python tools/l0_host_prof.py"""
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

P, q, A, l, u, x, nv = BL.load_l0_fixture("closed_loop")
pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0))
Pv, Av = pr.values(P, A)
lc, uc = np.clip(l, -1e20, 1e20), np.clip(u, -1e20, 1e20)
pr.analyze(lc[0], uc[0])
args = [tuple(np.ascontiguousarray(z[k:k + 1]) for z in (Pv, q, Av, lc, uc)) for k in range(8)]
o = pr.solve_raw(*args[0])
for k in range(1, 8):
    sys.stderr.write(f"--- tick {k}\n")
    t0 = time.perf_counter()
    pr.solve_raw(*args[k], out=o, warm_start=True)
    sys.stderr.write(f"--- call through ctypes: {(time.perf_counter() - t0) * 1e6:.1f} us\n")
empty = tuple(z[:0] for z in args[0])
oe = dict(x=o["x"][:0], y=o["y"][:0], status=o["status"][:0], iters=o["iters"][:0])
t = []
for _ in range(200):
    t0 = time.perf_counter()
    pr.solve_raw(*empty, out=oe)
    t.append(time.perf_counter() - t0)
print(f"empty call (B = 0) through ctypes + solve_raw: median {np.median(t) * 1e6:.2f} us")
