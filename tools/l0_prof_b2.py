import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import qpc_pkg; qpc_pkg.load()
from qpcontrol_jl_amd import binding
import bench_latency as BL, numpy as np
P, q, A, l, u, x, nv = BL.load_l0_fixture()
pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0), lib_path=sys.argv[1])
Pv, Av = pr.values(P, A)
idx = np.array([2, 2, 2])
for rep in range(2):
    r = pr.solve(Pv[idx], q[idx], Av[idx], l[idx], u[idx])
print(r["iters"], r["status"])
