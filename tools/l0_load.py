#!/usr/bin/env python3
"""The prepared L0 entry on batches of ONE problem (the committed Atlas problem that needs no working-set change), B = 1 ... 2048:
host time per call; with a -DQPC_WG_PROF library the kernel prints its phase times and the shader clock under that load."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import qpc_pkg; qpc_pkg.load()
from qpcontrol_jl_amd import binding
import bench_latency as BL
P, q, A, l, u, x, nv = BL.load_l0_fixture()
pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0), lib_path=sys.argv[1])
Pv, Av = pr.values(P, A)
which = sys.argv[2] if len(sys.argv) > 2 else "0"  # a problem index, or "mixed"
for B in (1, 16, 64, 256, 512, 2048):
    idx = np.arange(B) % P.shape[0] if which == "mixed" else np.full(B, int(which))
    a = (Pv[idx], q[idx], Av[idx], l[idx], u[idx])
    pr.solve(*a)
    t0 = time.perf_counter(); r = pr.solve(*a); dt = time.perf_counter() - t0
    print("problem", which, "B", B, "host ms", round(dt * 1e3, 3), flush=True)
