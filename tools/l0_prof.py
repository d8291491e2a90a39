#!/usr/bin/env python3
"""Phase times of the L0 workgroup kernel (csrc/qpc_qp_wg.h) on the committed 143 x 178 Atlas problems: the kernel prints its
stamps (set-up, S build, elimination, sweeps, Goldfarb-Idnani, equality multipliers, refinement; microseconds) when the
library is built with -DQPC_WG_PROF:
  cd qpcontrol.jl_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -DQPC_WG_PROF \
     -o ../../tests/_build/libqpc_prof.so -lhiprtc qpc_runtime.hip -x hip qpc_compile.cpp -x hip qpc_capi_build.cpp -x hip qpc_jit.cpp
  python tools/l0_prof.py tests/_build/libqpc_prof.so"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding  # noqa: E402
import bench_latency as BL  # noqa: E402

P, q, A, l, u, x, nv = BL.load_l0_fixture(sys.argv[2] if len(sys.argv) > 2 else "diag")
for b in (() if os.environ.get("QPC_PROF_SKIP_DENSE") else (0, 2, 6)):  # no working-set change / five sweeps / forty Goldfarb-Idnani steps
    for rep in range(2):
        r = binding.qp_solve_batch(P[b], q[b], A[b], l[b], u[b], lib_path=sys.argv[1])
    print("problem", b, "iters", r["iters"], "status", r["status"], flush=True)
pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0), lib_path=sys.argv[1])
Pv, Av = pr.values(P, A)
for b in (0, 2, 6):  # the same through the prepared entry, cold and warm
    for warm in (False, False, True):
        r = pr.solve(Pv[b], q[b], Av[b], l[b], u[b], warm_start=warm)
    print("prepared problem", b, "iters (warm)", r["iters"], "status", r["status"], flush=True)
