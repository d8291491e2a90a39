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
         "k_bias_crb", "sweep_kr", "sweep_y", "sweep_c22", "sweep_chol", "sweep_back", "sweep_test",
         "l_u0", "l_guard", "l_gi_check", "l_gi_step", "polish_prox", "polish_back", "accept", "rnea_reload",
         "asm_pre", "asm_items", "asm_rows", "xinv_x", "xinv_mfma", "rnea_sv", "rnea_bw", "presolve_qr", "presolve_reduce",
         "postsolve"]


def main():
    import torch
    torch.cuda.init()
    import qpc_pkg
    qpc_pkg.load()
    from qpcontrol_jl_amd import binding, workloads as W
    # the JIT-specialised kernel is compiled with the timers when QPC_JIT_STAMPS=1
    os.environ["QPC_JIT_STAMPS"] = "1"
    os.environ["QPC_REQUIRE_JIT"] = "1"
    lib = binding.DEFAULT_LIB
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
    B = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 8192  # 1024: one wave per SIMD (a QP alone)

    def factory(mech, N, fl):
        return binding.Controller(binding.Model(mech, lib), N, fl)
    fused = "--fused-laws" in sys.argv  # StandingController laws evaluated in the kernel (no `desired` input)
    ctrl, wl = W.atlas_workload(factory, B, cfg, fused_laws=fused)
    prof = torch.zeros((B, len(NAMES)), dtype=torch.int64, device="cuda")
    ctrl.L.qpc_debug_set_profile_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ctrl.L.qpc_debug_set_profile_buffer(ctrl.h, ctypes.c_void_p(prof.data_ptr()))
    # the section counters are written by the *_device entry (device-resident arrays on the caller's stream)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(np.ascontiguousarray(a)).to(dev) for k, a in dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight,
                                                                               cm=wl.contact_maxnormalforce).items()}
    lay = ctrl.layout
    tau = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
    vd = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
    st = torch.zeros((B,), dtype=torch.int32, device=dev)
    its = torch.zeros((B,), dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p
    rc = ctrl.L.qpc_control_batch_device(ctrl.h, B, vp(t["q"].data_ptr()), vp(t["v"].data_ptr()), vp(0) if fused else vp(t["d"].data_ptr()),
                                         vp(t["cw"].data_ptr()) if lay.n_contacts else vp(0), vp(t["cm"].data_ptr()) if lay.n_contacts else vp(0),
                                         vp(0), vp(0), vp(tau.data_ptr()), vp(vd.data_ptr()), vp(0), vp(st.data_ptr()), vp(its.data_ptr()), vp(0))
    assert rc == 0
    torch.cuda.synchronize()
    r = {"iters": its.cpu().numpy()}
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(np.float64)
    tot = p.sum(1)
    out = {"config": cfg, "B": B, "fused_laws": fused, "mean_cycles_per_qp": float(tot.mean()), "p99_cycles": float(np.percentile(tot, 99)),
           "mean_iters": float((r["iters"] % 1000).mean()),
           "sections": {n: {"mean_cycles": float(p[:, i].mean()), "share": float(p[:, i].sum() / tot.sum())}
                        for i, n in enumerate(NAMES)}}
    if "--by-ticket" in sys.argv:  # how a section's cycles depend on when the QP was drawn (eighths of the batch, in ticket order)
        names = [a for a in sys.argv if a in NAMES] or ["k_input"]
        out["by_ticket_eighth"] = {n: [float(c.mean()) for c in np.array_split(p[:, NAMES.index(n)], 8)] for n in names}
        out["by_ticket_eighth"]["total"] = [float(c.mean()) for c in np.array_split(tot, 8)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
