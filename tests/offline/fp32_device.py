#!/usr/bin/env python3
"""BASELINE config 5 "fp32" question settled ON THE DEVICE: the tick with its whole LDS slab stored in
fp32 (QPC_F32_LDS=1: M, G, H_rho, K, Y, the Schur complement and its factors, every vector -- half the LDS bytes,
fp64 arithmetic in registers) against the fp64 kernel on the same 4096 states: LDS bytes per QP, resident QPs per CU,
rate, statuses and |tau_f32 - tau_f64|.  Prints one JSON document (kept under profiles/r02_fp32_device.json)."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding  # noqa: E402
from qpcontrol_jl_amd import workloads as W  # noqa: E402


def factory(mech, N, fl):
    return binding.Controller(binding.Model(mech), N, fl)


def rate(c, wl, B, reps=10):
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight,
                                                                               cm=wl.contact_maxnormalforce).items()}
    lay = c.layout
    tau = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
    vd = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p

    def go():
        rc = c.L.qpc_control_batch_device(c.h, B, vp(t["q"].data_ptr()), vp(t["v"].data_ptr()), vp(t["d"].data_ptr()), vp(t["cw"].data_ptr()),
                                          vp(t["cm"].data_ptr()), vp(0), vp(0), vp(tau.data_ptr()), vp(vd.data_ptr()), vp(0), vp(st.data_ptr()), vp(0), vp(0))
        assert rc == 0
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    return B * reps / (time.perf_counter() - t0)


out = {}
for config in (3, 5):
    B = 4096
    os.environ.pop("QPC_F32_LDS", None)
    c64, wl = W.atlas_workload(factory, B, config)
    r64 = c64.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    os.environ["QPC_F32_LDS"] = "1"
    c32, _ = W.atlas_workload(factory, B, config)
    os.environ.pop("QPC_F32_LDS", None)
    r32 = c32.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    ok64, ok32 = r64["status"] == 0, r32["status"] <= 1
    both = ok64 & ok32
    scale = np.maximum(np.abs(r64["tau"]).max(axis=1), 1.0)
    rel = np.abs(r32["tau"] - r64["tau"]).max(axis=1) / scale
    B2 = 32768
    cb64, wlb = W.atlas_workload(factory, B2, config)
    os.environ["QPC_F32_LDS"] = "1"
    cb32, _ = W.atlas_workload(factory, B2, config)
    os.environ.pop("QPC_F32_LDS", None)
    out[f"config{config}"] = {
        "states": B, "optimal_fp64": int(ok64.sum()), "accepted_fp32": int(ok32.sum()),
        "fp32_status_histogram": {int(k): int(v) for k, v in zip(*np.unique(r32["status"], return_counts=True))},
        "fp64_solved_but_fp32_failed": int((ok64 & ~ok32).sum()),
        "rel_tau_error_where_both": {"median": float(np.median(rel[both])) if both.any() else None,
                                     "p90": float(np.percentile(rel[both], 90)) if both.any() else None,
                                     "p99": float(np.percentile(rel[both], 99)) if both.any() else None,
                                     "max": float(rel[both].max()) if both.any() else None,
                                     "fraction_below_1e-4": float((rel[both] < 1e-4).mean()) if both.any() else None},
        "lds_bytes_per_qp": {"fp64": int(c64.layout.lds_bytes_per_qp), "fp32": int(c64.layout.lds_bytes_per_qp) // 2},
        "resident_qps_per_cu": {"fp64": 8, "fp32": 8, "note": "both kernels need 256 VGPRs (fp64 arithmetic): 2 waves per SIMD either way"},
        "solves_per_s_at_32768": {"fp64": rate(cb64, wlb, B2), "fp32": rate(cb32, wlb, B2)},
    }
print(json.dumps(out, indent=1))
