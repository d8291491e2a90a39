#!/usr/bin/env python3
"""Sweep of the host-pointer pipeline (qpc_control_batch with host arrays, PCIe-inclusive): chunk size x slots per device
x {pageable, pinned} caller arrays, against the device-resident rate of the same controller.  One JSON line per point.

    python tools/bench_host_path.py [--batch 32768] [--reps 5]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (initialise torch's HIP runtime first, as bench.py does)

torch.cuda.init()
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding  # noqa: E402
from qpcontrol_jl_amd import workloads as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32768)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--slots", type=int, nargs="*", default=[2, 3, 4, 6])
    ap.add_argument("--chunks", type=int, nargs="*", default=[2048, 4096, 8192, 16384])
    args = ap.parse_args()
    B = args.batch
    for slots in args.slots:
        real = binding.Controller.finalize

        def fin(self, device_id=0, slots=slots):
            self.set_host_pipeline(0, slots, 0)
            return real(self, device_id)
        binding.Controller.finalize = fin
        try:
            t0 = time.perf_counter()
            c, wl = W.atlas_workload(lambda m, N, fl: binding.Controller(binding.Model(m), N, fl), B, 3)
            t_setup = time.perf_counter() - t0
        finally:
            binding.Controller.finalize = real
        L, lay = c.L, c.layout
        ins = dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight, cm=wl.contact_maxnormalforce)
        for mode in ("pageable", "pinned"):
            if mode == "pinned":
                arr = {k: binding.pinned_array(L, a.shape) for k, a in ins.items()}
                for k, a in ins.items():
                    arr[k][...] = a
                mk = lambda shape, dt=np.float64: binding.pinned_array(L, shape, dt)  # noqa: E731
            else:
                arr = {k: np.ascontiguousarray(a) for k, a in ins.items()}
                mk = lambda shape, dt=np.float64: np.zeros(shape, dtype=dt)  # noqa: E731
            out = dict(tau=mk((B, lay.nv)), vdot=mk((B, lay.nv)), cwrench_flat=mk((B, 6 * lay.n_contacts)),
                       status=mk((B,), np.int32), iters=mk((B,), np.int32))
            for a in out.values():
                a[...] = 0  # touch every page once
            for chunk in args.chunks:
                c.set_host_pipeline(chunk_qps=chunk)
                for _ in range(2):
                    c.control_batch(arr["q"], arr["v"], arr["d"], arr["cw"], arr["cm"], out=out)
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    c.control_batch(arr["q"], arr["v"], arr["d"], arr["cw"], arr["cm"], out=out)
                dt = (time.perf_counter() - t0) / args.reps
                assert (out["status"] == 0).all()
                print(json.dumps({"slots": slots, "mode": mode, "chunk": chunk, "B": B, "ms_per_call": 1e3 * dt,
                                  "solves_per_s": B / dt, "setup_s": round(t_setup, 2)}), flush=True)
        # zero-copy: the kernel reads / writes pinned host memory directly over PCIe (one launch for the whole batch)
        if slots == args.slots[0]:
            arr = {k: binding.pinned_array(L, a.shape) for k, a in ins.items()}
            for k, a in ins.items():
                arr[k][...] = a
            out = dict(tau=binding.pinned_array(L, (B, lay.nv)), vdot=binding.pinned_array(L, (B, lay.nv)),
                       cwrench=binding.pinned_array(L, (B, 6 * lay.n_contacts)), status=binding.pinned_array(L, (B,), np.int32),
                       iters=binding.pinned_array(L, (B,), np.int32))
            vp = ctypes.c_void_p
            P = lambda a: vp(a.ctypes.data)  # noqa: E731
            for what in ("zerocopy_inout", "zerocopy_out_only"):
                if what == "zerocopy_out_only":
                    dev = torch.device("cuda", 0)
                    tin = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in ins.items()}
                    pin = {k: vp(t.data_ptr()) for k, t in tin.items()}
                else:
                    pin = {k: P(a) for k, a in arr.items()}

                def call():
                    rc = L.qpc_control_batch_device(c.h, B, pin["q"], pin["v"], pin["d"], pin["cw"], pin["cm"], vp(0), vp(0), P(out["tau"]),
                                                    P(out["vdot"]), P(out["cwrench"]), P(out["status"]), P(out["iters"]), vp(0))
                    assert rc == 0
                    torch.cuda.synchronize()
                for _ in range(2):
                    call()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    call()
                dt = (time.perf_counter() - t0) / args.reps
                assert (out["status"] == 0).all()
                print(json.dumps({"slots": 0, "mode": what, "chunk": B, "B": B, "ms_per_call": 1e3 * dt, "solves_per_s": B / dt, "setup_s": 0}), flush=True)
        del c


if __name__ == "__main__":
    main()
