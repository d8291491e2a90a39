#!/usr/bin/env python3
"""The host-pointer entry (qpc_control_batch, what the Julia shim binds) with the StandingController laws fused into the tick
(standing.jl:58-89 on the device: `desired` is not an input any more) against the input-record form, page-locked caller arrays
used in place: how much of the PCIe-bound rate the smaller record (712 B instead of 1 000 B in) buys.  One JSON line.
    python tools/bench_host_fused.py [B=32768]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.cuda.init()
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
out = {"B": B}
for fused in (False, True):
    def factory(mech, n_basis, floating):
        return binding.Controller(binding.Model(mech), n_basis, floating)
    c, wl = W.atlas_workload(factory, B, 3, seed=2025, fused_laws=fused)
    lay, L = c.layout, c.L
    ins = dict(q=wl.q, v=wl.v, cw=wl.contact_weight, cm=wl.contact_maxnormalforce)
    if not fused:
        ins["d"] = wl.desired
    arr = {k: binding.pinned_array(L, a.shape) for k, a in ins.items()}
    for k, a in ins.items():
        arr[k][...] = a
    mk = lambda shape, dt=np.float64: binding.pinned_array(L, shape, dt)  # noqa: E731
    o = dict(tau=mk((B, lay.nv)), vdot=mk((B, lay.nv)), cwrench_flat=mk((B, 6 * lay.n_contacts)), status=mk((B,), np.int32),
             iters=mk((B,), np.int32))
    for _ in range(3):
        c.control_batch(arr["q"], arr["v"], arr.get("d"), arr["cw"], arr["cm"], out=o)
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        r = c.control_batch(arr["q"], arr["v"], arr.get("d"), arr["cw"], arr["cm"], out=o)
    dt = time.perf_counter() - t0
    out["fused_laws" if fused else "input_record"] = {"solves_per_s": B * reps / dt, "optimal_fraction": float((o["status"] == 0).mean()),
                                                     "input_bytes_per_qp": int(8 * sum(a.shape[1] for a in ins.values()))}
print(json.dumps(out))
