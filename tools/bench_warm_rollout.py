#!/usr/bin/env python3
"""Warm-started ticks in a rollout with CHANGING contact sets: B standing Atlas robots in closed loop
(tick with fused StandingController laws -> forward-dynamics step on compliant ground); every `period` steps the right
foot's four contact points are capped at a small maxnormalforce (the rho's of that foot go to their UPPER bounds and the
left foot takes the load) and released again -- the batched stand-in for lifting a foot.  The same rollout is run with
cold-started ticks (qpc_control_batch_device) and with warm-started ones (qpc_control_batch_device_warm: the working set
of each robot carried from tick to tick, the counterpart of the reference's persistent OSQP workspace, momentum.jl:58).

    python tools/bench_warm_rollout.py [B=8192] [steps=400] [period=100]   -> one JSON document"""
import ctypes
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
from qpcontrol_jl_amd import api, binding, models as M, workloads as W  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    period = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    mech = M.atlas()
    low = api.MomentumBasedController(mech, n_basis=4, floatingjoint="pelvis_to_world")
    for cp in mech.contact_points:
        pt = low.addcontact(cp.body, cp.position, cp.normal, cp.mu)
        pt.maxnormalforce, pt.weight = 1e6, 1e-3
    api.StandingController(low, ["l_foot", "r_foot"], "pelvis", M.atlas_nominal_configuration(mech))
    low._c.enable_warm_start()
    lay = low.finalize(0)
    L, h = low._c.L, low._c.h
    assert L.qpc_debug_warm_active(h) == 1
    q0, v0 = W.standing_states(mech, B, 3, dq=0.02, drot=0.01, vstd=0.0)
    k = 6.0e4
    sp = binding.SimParams(1e-3, 0.01308714 + mech.total_mass() * 9.81 / (8 * k), k, 400.0, 300.0)
    dev = torch.device("cuda", 0)
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)  # noqa: E731
    nc = lay.n_contacts
    rfoot = mech.findbody("r_foot")
    right = np.array([cp.body == rfoot for cp in mech.contact_points])
    cap = 0.45 * mech.total_mass() * 9.81 / right.sum()  # the right foot may carry ~45 % of the weight in the capped phases
    cm_free = np.full((B, nc), 1e6)
    cm_cap = cm_free.copy()
    cm_cap[:, right] = cap
    cw = t(np.full((B, nc), 1e-3))
    cms = [t(cm_free), t(cm_cap)]
    vp = ctypes.c_void_p
    p = lambda x: vp(x.data_ptr())  # noqa: E731
    stream = torch.cuda.current_stream(dev)
    out = {}
    finals = {}
    for mode in ("cold", "warm"):
        dq, dv = t(q0), t(v0)
        tau = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
        vd = torch.zeros((B, lay.nv), dtype=torch.float64, device=dev)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        iters = torch.zeros(B, dtype=torch.int32, device=dev)
        bad = torch.zeros(B, dtype=torch.int32, device=dev)
        ws = torch.zeros((B, 2), dtype=torch.int64, device=dev)
        it_sum = torch.zeros((), dtype=torch.float64, device=dev)
        it_switch = torch.zeros((), dtype=torch.float64, device=dev)
        n_switch = 0
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for s in range(K):
            cm = cms[(s // period) % 2]
            if mode == "cold":
                rc = L.qpc_control_batch_device(h, B, p(dq), p(dv), vp(0), p(cw), p(cm), vp(0), vp(0), p(tau), p(vd), vp(0), p(status),
                                                p(iters), vp(stream.cuda_stream))
            else:
                rc = L.qpc_control_batch_device_warm(h, 0, B, p(dq), p(dv), vp(0), p(cw), p(cm), vp(0), vp(0), p(tau), p(vd), vp(0),
                                                     p(status), p(iters), p(ws), vp(stream.cuda_stream))
            assert rc == 0, L.qpc_last_error()
            it_sum += (iters % 1000).double().mean()
            if s % period == 0 and s > 0:
                it_switch += (iters % 1000).double().mean()
                n_switch += 1
            rc = L.qpc_sim_step_batch_device(h, B, p(dq), p(dv), p(tau), p(status), ctypes.byref(sp), p(bad), vp(stream.cuda_stream))
            assert rc == 0
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        qf = dq.cpu().numpy()
        finals[mode] = (qf, tau.cpu().numpy())
        out[mode] = {"ms_per_step": 1e3 * el / K, "robot_steps_per_s": B * K / el, "mean_working_set_changes_per_tick": float(it_sum.item()) / K,
                     "mean_changes_on_switch_ticks": float(it_switch.item()) / max(n_switch, 1), "failed_qps": int(bad.sum().item()),
                     "pelvis_z_min": float(qf[:, 6].min()), "pelvis_z_max": float(qf[:, 6].max())}
    out["robots"], out["steps"], out["period"], out["right_foot_cap_N_per_contact"] = B, K, period, cap
    out["speedup_warm_over_cold"] = out["cold"]["ms_per_step"] / out["warm"]["ms_per_step"]
    out["max_abs_dq_between_modes"] = float(np.abs(finals["cold"][0] - finals["warm"][0]).max())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
