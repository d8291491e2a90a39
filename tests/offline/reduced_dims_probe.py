#!/usr/bin/env python3
"""How fast would the tick be if the hard foot tasks were eliminated before the solve (null-space presolve)?
A stand-in with the SOLVER DIMENSIONS of that reduced problem: the Atlas config-3 controller with the twelve leg joints pinned by hard
JointAccelerationTasks instead of the two foot tasks -- free columns = the three base linear accelerations, task rows = the three slack
rows, six wrench rows, 32 rays; same kinematics, same contact assembly, same working-set search machinery.  Not the same QPs (other
optima, other change counts), so only the per-QP cost structure carries over.  python tests/offline/reduced_dims_probe.py [B=32768]"""
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
from qpcontrol_jl_amd.workloads import ANGULAR, HARD, JOINT, LINEAR_MOMENTUM_RATE, SCALAR  # noqa: E402


def main():
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    mech = M.atlas()
    c = binding.Controller(binding.Model(mech), 4, mech.findbody("pelvis"))
    for cp in mech.contact_points:
        c.add_contact(cp.body, cp.position, cp.normal, cp.mu)
    for j in range(mech.nbodies):
        c.regularize(j, 0.05)
    c.add_task(LINEAR_MOMENTUM_RATE, wkind=SCALAR, weight=[1.0])
    c.add_task(ANGULAR, base=M.WORLD, target=mech.findbody("pelvis"), frame=M.FRAME_DEFAULT, wkind=HARD)
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.REVOLUTE:
            c.add_task(JOINT, joint=j, wkind=HARD)
    lay = c.finalize()
    # states / set-points of the config-3 workload; the leg joints get the PD accelerations of the other joints' law
    ref, wl = W.atlas_workload(lambda m, N, fl: binding.Controller(binding.Model(m), N, fl), B, 3)
    st = wl.setup
    des = np.zeros((B, lay.n_desired))
    des[:, 0:3] = wl.desired[:, st.offsets[st.linmom_task]:st.offsets[st.linmom_task] + 3]
    des[:, 3:6] = wl.desired[:, st.offsets[st.pelvis_task]:st.offsets[st.pelvis_task] + 3]
    k = 6
    for j in range(mech.nbodies):
        if mech.joint_type[j] == M.REVOLUTE:
            qi, vi = wl.q[:, mech.configuration_range(j)[0]], wl.v[:, mech.velocity_range(j)[0]]
            des[:, k] = -st.jointgains[0] * (qi - st.jointrefs[j]) - st.jointgains[1] * vi
            k += 1
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
    q, v, d, cw, cm = t(wl.q), t(wl.v), t(des), t(wl.contact_weight), t(wl.contact_maxnormalforce)
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
    print(json.dumps({"solves_per_s": B * n / el, "ms_per_launch": 1e3 * el / n, "n_free": lay.n_free, "n_eq_rows": lay.n_eq_rows,
                      "lds_bytes_per_qp": lay.lds_bytes_per_qp, "status_counts": np.bincount(s, minlength=6).tolist(),
                      "mean_working_set_changes": float(it.mean())}))


if __name__ == "__main__":
    main()
