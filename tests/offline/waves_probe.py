#!/usr/bin/env python3
"""What does a THIRD wave per SIMD buy a tick-like instruction stream?  The Atlas tick cannot answer it (its 20 KB slab allows eight
wavefronts per CU whatever the register budget), so this probe runs the same kernel code on a smaller robot whose slab fits twelve
times into a CU: a legs-only biped (pelvis + two 6-joint legs, two contact points per foot), standing task set.  With the slab as
dynamic shared memory the SAME binary (168 registers, QPC_EXP_WAVES=3) runs at 8 or 12 wavefronts per CU by padding the
LDS request (QPC_EXP_LDS_PAD) -- the ratio is the gain of the third wave, free of the register effect.
  QPC_EXP_DYN_LDS=1 QPC_EXP_WAVES=3 QPC_EXP_BLOCKS_PER_CU=12 python tests/offline/waves_probe.py
  QPC_EXP_DYN_LDS=1 QPC_EXP_WAVES=3 QPC_EXP_BLOCKS_PER_CU=8 QPC_EXP_LDS_PAD=<bytes to 20 KB> python tests/offline/waves_probe.py"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, models as M, workloads as W  # noqa: E402


def biped():
    """the pelvis and legs of models.atlas() (13 bodies, nv 18), the trunk mass lumped into the pelvis, 2 contact points per foot"""
    a = M.atlas()
    m = M.Mechanism()
    keep = [i for i, n in enumerate(a.body_names) if n == "pelvis" or "_leg_" in a.joint_names[i]]
    for i in keep:
        par = a.parent[i] if a.parent[i] < 0 else a.body_names[a.parent[i]]
        mass = a.mass[i] + (100.0 if a.body_names[i] == "pelvis" else 0.0)
        m.add_body(a.body_names[i], par, a.joint_names[i], a.joint_type[i], a.joint_axis[i], a.joint_R[i], a.joint_p[i], mass,
                   a.com[i], a.inertia[i] * (mass / a.mass[i]))
    for cp in a.contact_points:
        if cp.position[1] > 0:
            continue
        m.contact_points.append(M.ContactSpec(m.findbody(a.body_names[cp.body]), cp.position * np.array([1.0, 0.0, 1.0]), cp.normal, cp.mu))
    return m


def workload(factory, B, seed=7):
    mech = biped()
    c = factory(mech, 4, mech.findbody("pelvis"))
    q0 = M.atlas_nominal_configuration(mech)
    st = W.standing_setup(c, mech, q0)
    lay = c.finalize()
    q, v = W.standing_states(mech, B, seed)
    des = W.standing_desired(st, q, v)
    nc = len(st.contact_ids)
    return c, lay, mech, q, v, des, np.full((B, nc), 1e-3), np.full((B, nc), 1e6)


def main():
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
    # QPC_PROBE_F32=1 / QPC_PROBE_PACK2=1: the fp32 tick / two QPs per wavefront on this biped (13 bodies, nv 18, 16 rays: every phase
    # fits the 32 lanes of a half-wave -- the best case of the half-wave packing)
    def factory(m, N, fl):
        ctl = binding.Controller(binding.Model(m), N, fl)
        if os.environ.get("QPC_PROBE_F32") == "1":
            ctl.set_precision(1)
        if os.environ.get("QPC_PROBE_PACK2") == "1":
            ctl.set_qps_per_wavefront(2)
        return ctl
    c, lay, mech, q_, v_, des, cw_, cm_ = workload(factory, B)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    q, v, d, cw, cm = t(q_), t(v_), t(des), t(cw_), t(cm_)
    tau = torch.empty((B, lay.nv), dtype=torch.float64, device=dev); vd = torch.empty_like(tau)
    status = torch.empty(B, dtype=torch.int32, device=dev); iters = torch.empty(B, dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p
    p = lambda x: vp(x.data_ptr())  # noqa: E731

    def step():
        rc = c.L.qpc_control_batch_device(c.h, B, p(q), p(v), p(d), p(cw), p(cm), vp(0), vp(0), p(tau), p(vd), vp(0), p(status), p(iters),
                                          vp(torch.cuda.current_stream(dev).cuda_stream))
        assert rc == 0, c.L.qpc_last_error()
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    s = status.cpu().numpy(); it = iters.cpu().numpy() % 1000
    env = {k: os.environ[k] for k in os.environ if k.startswith("QPC_EXP_") or k.startswith("QPC_PROBE_")}
    print(json.dumps({"solves_per_s": B * n / el, "ms_per_launch": 1e3 * el / n, "B": B, "n_free": lay.n_free, "n_eq_rows": lay.n_eq_rows,
                      "lds_bytes_per_qp": lay.lds_bytes_per_qp, "status_counts": np.bincount(s, minlength=6).tolist(),
                      "mean_working_set_changes": float(it.mean()), "env": env}))


if __name__ == "__main__":
    main()
