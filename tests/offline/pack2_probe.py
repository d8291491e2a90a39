#!/usr/bin/env python3
"""Two QPs per wavefront (qpc_set_qps_per_wavefront(2), csrc/qpc_device.h QPC_PACK2) against one: same statuses, tau to rounding
(the packed kernel builds its Schur complement with dot products instead of the matrix core), and the rate of both.
python tests/offline/pack2_probe.py [f32|f64] [config] [B]   (GPU box)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
    config = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 32768

    def factory(pack):
        def f(mech, N, fl):
            c = binding.Controller(binding.Model(mech), N, fl)
            if prec == "f32":
                c.set_precision(1)
            if pack:
                c.set_qps_per_wavefront(2)
            return c
        return f
    out = {"precision": prec, "config": config, "batch": B}
    res = {}
    for pack in (False, True):
        c, wl = W.atlas_workload(factory(pack), B, config)
        a = [binding.pinned_array(c.L, x.shape) for x in (wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)]
        for d, x in zip(a, (wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)):
            d[...] = x
        r = c.control_batch(*a)
        t0 = time.perf_counter()
        for _ in range(5):
            r = c.control_batch(*a)
        el = (time.perf_counter() - t0) / 5
        res[pack] = {k: np.array(v) for k, v in r.items()}
        out["packed" if pack else "plain"] = {"host_path_solves_per_s": B / el, "optimal_fraction": float((r["status"] <= 1).mean()),
                                               "lds_bytes_per_qp": int(c.layout.lds_bytes_per_qp)}
    a, b = res[False], res[True]
    out["status_equal_fraction"] = float((a["status"] == b["status"]).mean())
    ok = (a["status"] <= 1) & (b["status"] <= 1)
    scale = np.maximum(1.0, np.abs(a["tau"]).max(axis=1, keepdims=True))
    err = (np.abs(a["tau"] - b["tau"]) / scale).max(axis=1)
    out["rel_tau_diff"] = {"median": float(np.median(err[ok])), "p999": float(np.percentile(err[ok], 99.9)), "max": float(err[ok].max())}
    out["iters_equal_fraction"] = float((a["iters"][ok] == b["iters"][ok]).mean())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
