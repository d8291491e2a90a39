#!/usr/bin/env python3
"""In-kernel section timers (diagnostic build -DQPC_STAMPS, libqpcontrol_hip_stamps.so): mean shader
cycles per QP spent in each section of the tick.  Shares only -- the stamped build is never timed
end-to-end (cdna_hip_programming.md section 7, In-kernel stamps)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["kinematics", "assembly", "schur_build", "cholesky", "x_inverse", "refactor_rho", "residuals", "kkt_solve",
         "update", "logic", "final_refine", "rnea", "k_input", "k_joint", "k_world", "k_subspace_inertia", "k_twist",
         "k_bias_crb", "sweep_kr", "sweep_y", "sweep_c22", "sweep_chol", "sweep_back", "sweep_test"]


def main():
    import torch
    import qpc_pkg
    qpc_pkg.load()
    from qpcontrol_jl_amd import binding, workloads as W
    # the JIT-specialised kernel is compiled with the timers when QPC_JIT_STAMPS=1
    os.environ["QPC_JIT_STAMPS"] = "1"
    os.environ["QPC_REQUIRE_JIT"] = "1"
    lib = binding.DEFAULT_LIB
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
    B = 8192

    def factory(mech, N, fl):
        return binding.Controller(binding.Model(mech, lib), N, fl)
    ctrl, wl = W.atlas_workload(factory, B, cfg)
    prof = torch.zeros((B, len(NAMES)), dtype=torch.int64, device="cuda")
    ctrl.L.qpc_debug_set_profile_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ctrl.L.qpc_debug_set_profile_buffer(ctrl.h, ctypes.c_void_p(prof.data_ptr()))
    r = ctrl.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    tot = p.sum(1)
    out = {"config": cfg, "B": B, "mean_cycles_per_qp": float(tot.mean()), "p99_cycles": float(np.percentile(tot, 99)),
           "mean_iters": float((r["iters"] % 1000).mean()),
           "sections": {n: {"mean_cycles": float(p[:, i].mean()), "share": float(p[:, i].sum() / tot.sum())}
                        for i, n in enumerate(NAMES)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
