#!/usr/bin/env python3
"""B = 1 latency of the prepared L0 entry (qpc_qp_setup / qpc_qp_solve) on the first problem of each committed fixture:
the diagonal-P Atlas problem and the one with a matrix-weighted MomentumRateTask (a 6 x 6 block in P:
/root/reference/src/lowlevel/momentum.jl:113-117, test/controller.jl:232-285).  One JSON line.
    python tools/l0_cliff.py [calls]"""
import json
import sys
import time

import numpy as np

import bench_latency as BL
from qpcontrol_jl_amd import binding


def main(calls=3):
    out = {"metric": "qpc_qp_solve at B = 1, host pointers, synchronous (median over calls)", "unit": "us", "calls": calls}
    for which in ("diag", "matrix"):
        P, q, A, l, u, x, nv = BL.load_l0_fixture(which)
        pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0))
        Pv, Av = pr.values(P, A)
        a = (Pv[0], q[0], Av[0], l[0], u[0])
        r = pr.solve(*a)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            r = pr.solve(*a)
            ts.append(time.perf_counter() - t0)
        out[which] = {"n": int(P.shape[1]), "m": int(A.shape[1]), "nnz_P_pattern": int(len(pr.P_rows)), "status": int(r["status"][0]),
                      "iters": int(r["iters"][0]), "median_us": float(np.median(ts) * 1e6),
                      "max_abs_err_vdot_vs_fixture": float(np.abs(r["x"][0, :nv] - x[0, :nv]).max())}
        pr.close()
    out["matrix_over_diag"] = out["matrix"]["median_us"] / out["diag"]["median_us"]
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
