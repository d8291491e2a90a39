#!/usr/bin/env python3
"""Closed-loop throughput on the device: B pushed Atlas robots, K x (StandingController
tick with fused PD laws -> forward-dynamics step on compliant ground), everything resident in HBM, one
stream, no host synchronisation inside the rollout (qpc_rollout_batch_device).
python tools/bench_rollout.py [B=32768] [K=50]  -> one JSON line (robot-steps/s; not the headline metric)"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import api, binding, models as M, workloads as W  # noqa: E402


def main():
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    mech = M.atlas()
    low = api.MomentumBasedController(mech, n_basis=4, floatingjoint="pelvis_to_world")
    for cp in mech.contact_points:
        pt = low.addcontact(cp.body, cp.position, cp.normal, cp.mu)
        pt.maxnormalforce, pt.weight = 1e6, 1e-3
    ctrl = api.StandingController(low, ["l_foot", "r_foot"], "pelvis", M.atlas_nominal_configuration(mech))
    warm = os.environ.get("QPC_ROLLOUT_WARM") == "1"  # qpc_rollout_batch_device warm-starts its ticks when the controller allows it
    if warm:
        low._c.enable_warm_start()
    lay = low.finalize(0)
    q, v = W.standing_states(mech, B, 3, dq=0.02, drot=0.01, vstd=0.0)
    v[:, 3] = np.linspace(-0.08, 0.08, B)
    k = 6.0e4
    sp = binding.SimParams(1e-3, 0.01308714 + mech.total_mass() * 9.81 / (8 * k), k, 400.0, 300.0)
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)  # noqa: E731
    dq, dv = t(q), t(v)
    cw, cm = t(np.full((B, lay.n_contacts), 1e-3)), t(np.full((B, lay.n_contacts), 1e6))
    tau, vd = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev), torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
    status, bad = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p
    p = lambda x: vp(x.data_ptr())  # noqa: E731
    L, h = low._c.L, low._c.h
    stream = torch.cuda.current_stream(dev)

    def run(n):
        rc = L.qpc_rollout_batch_device(h, B, n, p(dq), p(dv), vp(0), p(cw), p(cm), vp(0), vp(0), ctypes.byref(sp), p(tau), p(vd),
                                        p(status), p(bad), vp(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(L.qpc_last_error().decode())

    run(3)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    qf = dq.cpu().numpy()
    print(json.dumps({"metric": "closed-loop robot-steps/s (tick + simulation step, Atlas standing, fused PD laws)",
                      "value": B * K / el, "robots": B, "steps": K, "ms_per_step": 1e3 * el / K, "failed_qps": int(bad.sum().item()),
                      "pelvis_z_min": float(qf[:, 6].min()), "pelvis_z_max": float(qf[:, 6].max()),
                      "simulated_seconds_per_wall_second": B * K * sp.dt / el, "warm_started_ticks": warm}))


if __name__ == "__main__":
    main()
