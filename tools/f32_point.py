#!/usr/bin/env python3
"""The fp32 tick on BASELINE config 5 (32 768 QPs, device-resident) and config 3, beside the fp64 tick on the same states:
rate, solved fraction, torque error against fp64.  python tools/f32_point.py -> one JSON line (a subset of bench.py's secondary points)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W  # noqa: E402
import bench  # noqa: E402

out = []
tau = {}
for config, B, steps, f32 in ((5, 32768, 20, False), (5, 32768, 20, True), (3, 32768, 20, True)):
    run = bench.DeviceRun(torch, binding, W, B, config, {3: 2025, 5: 2027}[config], 0, f32=f32)
    el, kms, nl = run.timed(steps, 3)
    st = run.status.cpu().numpy()
    pt = {"config": config, "batch": B, "dtype": "f32" if f32 else "f64", "solves_per_s": B * steps / el, "kernel_ms": kms,
          "optimal_fraction": float((st <= 1).mean())}
    if config == 5:
        tau[f32] = (run.tau.cpu().numpy(), st)
        if len(tau) == 2:
            (t64, s64), (t32, s32) = tau[False], tau[True]
            both = (s64 <= 1) & (s32 <= 1)
            err = (np.abs(t32 - t64) / np.maximum(1.0, np.abs(t64).max(axis=1, keepdims=True))).max(axis=1)
            pt["vs_f64"] = {"rel_tau_err_median": float(np.median(err[both])), "rel_tau_err_p999": float(np.percentile(err[both], 99.9)),
                            "rel_tau_err_max": float(err[both].max())}
    out.append(pt)
    del run
print(json.dumps(out))
