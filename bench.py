#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the batched whole-body-QP control path on MI355X.

One "step" = one pass of the hot path (qpc_control_batch_device: state -> task/contact assembly
-> QP solve -> v̇, contact wrenches -> inverse dynamics -> tau) over one batch of synthetic Atlas
states with inputs already resident in HBM.  Workload = BASELINE.json config 3 (Atlas, 8
ContactPoints, linearised friction cones, fp64) at 32768 QPs per GPU, so that `--gpus 8`
reproduces config 4 (B = 262144 sharded over 8 GPUs, no collective in the data path) exactly.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SOLVE = 1968  # in 8*(37+36+36+16) + out 8*(36+36+48) + 8
ALG_FLOPS_PER_SOLVE = 4.6e5  # one reduced-KKT factor+solve + assembly + FK/RNEA
# HBM traffic per solve measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes
# (profiles/r01_v35_pmc_*.csv, r01_v35_traffic_summary.json; tools/profile_round.sh): 185389 KiB read + 259537 KiB written per
# 32768-QP launch.  ~5.6 KB each way per solve is the deliberate lane-private spill slot (22 x 32 lanes x 8 B), the rest
# register spills of the 256-VGPR kernel; the slot (inertias,
# bias wrenches, motion subspaces parked in HBM during the solve so that 8 QPs fit the LDS of a CU,
# DESIGN.md 4); the slots total 23 MB and live in L2 / Infinity Cache, TCC->EA requests count them anyway.
# 8-byte-per-lane accesses: the gfx950 "x2 for 16-B wide reads" correction of MI355X_MICROARCH.md
# does not apply.
MEASURED_TRAFFIC_BYTES_PER_SOLVE = (185389.1 + 259537.1) * 1024 / 32768
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md (spec; 6.29e12 measured)
FP64_PEAK = 78.6e12  # FLOP/s vector fp64 (datasheet)


def _cpu_worker(rank, B_sample, budget_s):
    """One host core: the oracle's reference-layout tick on its own slice of the workload for
    `budget_s` seconds.  Runs in a child process (no torch, no HIP)."""
    import qpc_pkg
    qpc_pkg.load()
    from qpcontrol_jl_amd import workloads as W
    from oracle import oracle as O

    def factory(mech, N, fl):
        return O.OracleController(O.OracleModel(mech, native=True), N, fl)

    oc, wl = W.atlas_workload(factory, B_sample, 3, seed=2025 + 104729 * rank)
    oc.set_settings(O.test_settings(polish=0))
    done, iters, chunk = 0, 0, 16
    t0 = time.perf_counter()
    while done < B_sample and time.perf_counter() - t0 < budget_s:
        s = slice(done, done + chunk)
        r = oc.tick_batch(wl.q[s], wl.v[s], wl.desired[s], wl.contact_weight[s], wl.contact_maxnormalforce[s])
        iters += int(r["iters"].sum())
        done += len(r["iters"])
    sec = time.perf_counter() - t0
    # the same path at the looser tolerances of the reference's notebook (Standing controller.ipynb:64-71,
    # eps 1e-5): what a user of the reference would run interactively; a third of the budget
    oc.set_settings(O.notebook_settings(polish=0))
    nb_done, nb_iters, off = 0, 0, 0
    t1 = time.perf_counter()
    while off < B_sample and time.perf_counter() - t1 < budget_s / 3:
        s = slice(off, off + chunk)
        r = oc.tick_batch(wl.q[s], wl.v[s], wl.desired[s], wl.contact_weight[s], wl.contact_maxnormalforce[s])
        nb_iters += int(r["iters"].sum())
        nb_done += len(r["iters"])
        off += chunk
    print(json.dumps({"done": done, "iters": iters, "seconds": sec, "nb_done": nb_done, "nb_iters": nb_iters,
                      "nb_seconds": time.perf_counter() - t1}))


def cpu_baseline(budget_s=12.0, max_workers=64):
    """The oracle (CPU restatement of the reference path: 143x178 reference-layout problem,
    coefficient rewrite per tick, OSQP-algorithm ADMM with the test-suite settings of
    test/runtests.jl:35-43, cold start, RNEA) timed on the host cores of the box: one independent
    controller per process over disjoint slices, bounded by `budget_s` seconds.
    `value` is the all-cores rate, `single_core` the rate of one of those processes.
    The workers are child processes of this script (`--cpu-worker`): they never touch torch or HIP."""
    import subprocess
    from oracle import oracle as O
    O.build(native=True)  # build once, before the workers load the library
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:  # a container's CPU quota (cgroup v2) is the real number of cores this job may keep busy
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = min(avail, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    n = max(1, min(avail, max_workers))
    B_sample = max(64, int(120 * budget_s))  # more than one core can finish in the budget
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(r), str(B_sample), str(budget_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(n)]
    res = []
    for p in procs:
        out, _ = p.communicate(timeout=budget_s * 6 + 120)
        if p.returncode == 0 and out.strip():
            res.append(json.loads(out.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    if not res:
        raise RuntimeError("cpu_baseline: no worker finished")
    done = sum(r["done"] for r in res)
    iters = sum(r["iters"] for r in res)
    busy = max(r["seconds"] for r in res)
    return {"value": done / busy, "unit": "QP solves/s", "cores": len(res), "kind": "port",
            "single_core": res[0]["done"] / res[0]["seconds"],
            "notebook_settings": {"value": sum(r["nb_done"] for r in res) / max(max(r["nb_seconds"] for r in res), 1e-9),
                                  "mean_admm_iterations": sum(r["nb_iters"] for r in res) / max(sum(r["nb_done"] for r in res), 1),
                                  "note": "same workload and cores at the notebook's OSQP tolerances (eps_abs = eps_rel = 1e-5): "
                                          "tau then agrees with the fp64 optimum to ~1e-3 only"},
            "sample": f"{done} QPs of the same Atlas config-3 workload over {len(res)} processes (one controller each, disjoint "
                      f"seeded slices), OSQP restatement eps_abs=1e-8 eps_rel=1e-16 cold start, mean "
                      f"{iters / max(done, 1):.0f} ADMM iterations, {busy:.1f} s of solving ({wall:.1f} s with process start-up); "
                      f"host has {os.cpu_count()} cores"}


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=32768)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 5])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused-laws", action="store_true",
                    help="SURVEY 8 f-1: StandingController PD laws (standing.jl:58-89) evaluated in the kernel, no `desired` "
                         "input (a pure (q, v) -> tau map); default is the reference boundary with desireds as inputs")
    ap.add_argument("--host-path", action="store_true",
                    help="also time the host-pointer entry point (PCIe-inclusive; reported in config, never `value`)")
    args = ap.parse_args()

    import torch
    import qpc_pkg
    qpc_pkg.load()
    from qpcontrol_jl_amd import binding
    from qpcontrol_jl_amd import workloads as W

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # QPC_BENCH_TEST_ONE_DEVICE=1 (test hook): every rank uses cuda:0 and the process group is gloo, so that the
    # multi-rank code path can be exercised on a single-GPU box; the numbers of such a run mean nothing
    one_device = os.environ.get("QPC_BENCH_TEST_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B = args.batch_per_gpu

    def factory(mech, n_basis, floating):
        return binding.Controller(binding.Model(mech), n_basis, floating)

    # every rank builds its own contiguous shard [rank*B, (rank+1)*B) of the global batch
    import importlib
    mech_ctrl = None
    ctrl, wl = None, None

    def make():
        nonlocal ctrl, wl
        real_finalize = binding.Controller.finalize

        def fin(self, device_id=0):
            return real_finalize(self, local_rank)
        binding.Controller.finalize = fin
        try:
            ctrl, wl = W.atlas_workload(factory, B, args.config, seed={2: 2024, 3: 2025, 5: 2027}[args.config] + 7919 * rank,
                                        fused_laws=args.fused_laws)
        finally:
            binding.Controller.finalize = real_finalize
    make()
    lay = ctrl.layout
    dev = torch.device("cuda", local_rank)

    def dev_t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)

    q, v, des = dev_t(wl.q), dev_t(wl.v), (None if args.fused_laws else dev_t(wl.desired))
    cw, cm = dev_t(wl.contact_weight), dev_t(wl.contact_maxnormalforce)
    tau = torch.empty((B, lay.nv), dtype=torch.float64, device=dev)
    vd = torch.empty((B, lay.nv), dtype=torch.float64, device=dev)
    cwr = torch.empty((B, max(lay.n_contacts, 1) * 6), dtype=torch.float64, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    iters = torch.empty(B, dtype=torch.int32, device=dev)
    L = ctrl.L
    vp = ctypes.c_void_p

    def ptr(t):
        return vp(t.data_ptr()) if t is not None and t.numel() else vp(0)

    stream = torch.cuda.current_stream(dev)

    def step():
        rc = L.qpc_control_batch_device(ctrl.h, B, ptr(q), ptr(v), ptr(des), ptr(cw) if lay.n_contacts else vp(0),
                                        ptr(cm) if lay.n_contacts else vp(0), vp(0), vp(0), ptr(tau), ptr(vd),
                                        ptr(cwr) if lay.n_contacts else vp(0), ptr(status), ptr(iters),
                                        vp(stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(L.qpc_last_error().decode())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    L.qpc_timing_enable(ctrl.h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_ms, nl = ctypes.c_double(0), ctypes.c_int64(0)
    L.qpc_timing_read(ctrl.h, ctypes.byref(avg_ms), ctypes.byref(nl))
    L.qpc_timing_enable(ctrl.h, 0)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = status.cpu().numpy()
    it = iters.cpu().numpy() % 1000
    fp64_measured = ctypes.c_double(0.0)  # vector-fp64 FMA peak measured on this box, outside the timed region
    if rank == 0 and hasattr(L, "qpc_debug_measure_fp64_peak"):
        L.qpc_debug_measure_fp64_peak.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
        if L.qpc_debug_measure_fp64_peak(local_rank, ctypes.byref(fp64_measured)) != 0:
            fp64_measured = ctypes.c_double(0.0)
    n_opt = int((st == 0).sum())

    host_rate = None
    if args.host_path and world == 1:
        hdes = None if args.fused_laws else wl.desired
        ctrl.control_batch(wl.q, wl.v, hdes, wl.contact_weight, wl.contact_maxnormalforce)
        th = time.perf_counter()
        for _ in range(3):
            ctrl.control_batch(wl.q, wl.v, hdes, wl.contact_weight, wl.contact_maxnormalforce)
        host_rate = 3 * B / (time.perf_counter() - th)

    if rank == 0:
        total = B * world * args.steps
        value = total / elapsed
        kern_s = avg_ms.value * 1e-3
        # compulsory bytes at the boundary = inputs + outputs (+ status, iters)
        nc_ = int(lay.n_contacts)
        alg_bytes = 8 * (lay.nq + lay.nv + (0 if args.fused_laws else lay.n_desired) + 2 * nc_) + 8 * (2 * lay.nv + 6 * nc_) + 8
        assert args.config != 3 or args.fused_laws or alg_bytes == ALG_BYTES_PER_SOLVE
        ach_bytes = alg_bytes * B / kern_s if kern_s > 0 else 0.0
        out = {
            "metric": "QP solves/sec (Atlas ~30-DoF whole-body QP, batched)",
            "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"atlas_config{args.config}: synthetic Atlas (nq 37, nv 36), StandingController task set, "
                                   + ("8 ContactPoints x 4-ray friction cones, floating base wrench balance "
                                      "(143 vars / 114 eq / 64 ineq in the reference layout)" if args.config != 2
                                      else "no contacts (39 vars / 36 eq)"),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"shard{world} (independent QPs, no collective)",
                       "optimal_fraction": n_opt / B, "mean_active_set_iters": float(it.mean()),
                       "lds_bytes_per_qp": int(lay.lds_bytes_per_qp),
                       "desired": "device PD laws (standing.jl:58-89 fused)" if args.fused_laws else "input record",
                       "host_pointer_path_solves_per_s": host_rate},
            "roofline": {"bound": "hbm", "achieved": ach_bytes / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach_bytes / HBM_PEAK,
                         "traffic": MEASURED_TRAFFIC_BYTES_PER_SOLVE * B if args.config == 3 else None,
                         "traffic_source": "profiles/r01_v35_traffic_summary.json (rocprofv3 PMC FETCH_SIZE + WRITE_SIZE, separate passes), bytes per launch; includes the HBM spill slots (DESIGN.md 4)",
                         "kernel": "qpc_tick_kernel", "kernel_avg_ms": avg_ms.value, "launches": int(nl.value),
                         "algorithmic_bytes_per_solve": alg_bytes,
                         "note": "path is fp64-VALU / LDS-latency bound, not HBM bound (SURVEY.md 8d); "
                                 "fp64 fraction reported beside it",
                         "fp64": {"achieved_tflops": ALG_FLOPS_PER_SOLVE * B / kern_s / 1e12 if kern_s > 0 else 0.0,
                                  "peak_tflops": FP64_PEAK / 1e12,
                                  "frac": (ALG_FLOPS_PER_SOLVE * B / kern_s) / FP64_PEAK if kern_s > 0 else 0.0,
                                  "algorithmic_flops_per_solve": ALG_FLOPS_PER_SOLVE,
                                  "peak_measured_tflops": fp64_measured.value or None,
                                  "frac_of_measured_peak": ((ALG_FLOPS_PER_SOLVE * B / kern_s) / (fp64_measured.value * 1e12)
                                                            if kern_s > 0 and fp64_measured.value > 0 else None)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
