#!/usr/bin/env python3
"""Latency of one controller tick through the host-pointer entry point (qpc_control_batch: H2D, kernel, D2H,
synchronous) for small batches -- the regime of the reference itself (ONE robot, one tick per millisecond).
python tools/bench_latency.py  -> one JSON line (median / p99 microseconds per call; not the headline metric)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W  # noqa: E402


def main():
    def factory(mech, n_basis, floating):
        return binding.Controller(binding.Model(mech), n_basis, floating)
    ctrl, wl = W.atlas_workload(factory, 2048, 3)
    out = {"metric": "tick latency, host pointers in / out (Atlas config 3)", "unit": "us", "batches": {}}
    for B in (1, 8, 64, 512, 2048):
        a = (wl.q[:B], wl.v[:B], wl.desired[:B], wl.contact_weight[:B], wl.contact_maxnormalforce[:B])
        for _ in range(20):
            ctrl.control_batch(*a)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            r = ctrl.control_batch(*a)
            ts.append(time.perf_counter() - t0)
        assert (r["status"] <= 1).all()
        ts = np.array(ts) * 1e6
        out["batches"][str(B)] = {"median_us": float(np.median(ts)), "p99_us": float(np.percentile(ts, 99)),
                                  "solves_per_s": float(B / np.median(ts) * 1e6)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
