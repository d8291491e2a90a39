#!/usr/bin/env python3
"""bench.py -- QP solves/sec of the batched whole-body-QP control path on MI355X.

One "step" = one pass of the hot path (qpc_control_batch_device: state -> task/contact assembly
-> QP solve -> v̇, contact wrenches -> inverse dynamics -> tau) over one batch of synthetic Atlas
states with inputs already resident in HBM.  Workload = BASELINE.json config 3 (Atlas, 8
ContactPoints, linearised friction cones, fp64) at 32768 QPs per GPU, so that `--gpus 8`
reproduces config 4 (B = 262144 sharded over 8 GPUs, no collective in the data path) exactly.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: this process starts the N ranks itself -- launch_ranks --
                                                          and adds config.in_library_devices: one process driving N devices)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Beside the headline (device-resident, timed region = K launches):
  config.secondary   the literal BASELINE batches on the same GPU, measured after the timed region:
                     config 2 @ 1024, config 3 @ 8192, config 5 @ 32768, config 3 @ 262144
  config.host_pointer_path_*   the host-pointer entry point the Julia shim binds (PCIe inclusive; never `value`)
  config.per_rank    (N > 1) every rank's kernel time and wall time per step
  config.closed_loop controller tick + simulation step on the device, robot-steps/s
  roofline           HBM (algorithmic bytes) + fp64 (algorithmic flops) fractions of the tick kernel, live HIP events
  cpu_baseline       the oracle (CPU restatement of the reference path) on the host cores.  `value` = the regime the reference
                     itself runs on Atlas: notebook tolerances (1e-5), closed loop at 1 kHz, OSQP warm-started from the
                     previous tick, all cores.  Beside it: `notebook_cold`, `tight_cold` (test-suite tolerances 1e-8, cold: the
                     setting the 1e-8 parity is judged against), `tight_warm_closed_loop`
  config.l0_plugin   the solver-level plug (b-1) on the committed Atlas problems (143 x 178; 146 x 181 with a matrix-weighted task): latency at
                     B = 1 and rate at 8 192 of the prepared entry qpc_qp_setup / qpc_qp_solve (cold and warm-started) that QPCOptimizer
                     binds, and of the dense entry qpc_qp_solve_batch (tools/bench_latency.py --l0); the CPU beside it is
                     cpu_baseline.l0_fixture_osqp: the oracle's OSQP on the same problems, measured in the same run;
                     .closed_loop: 95 consecutive ticks of one closed loop, each solve warm-started from the tick before (the regime the
                     reference runs the slot in; the CPU on the same sequence: cpu_baseline.l0_fixture_osqp.closed_loop_sequence);
                     solves_per_s_B8192_device_resident_*: qpc_qp_solve_device (values, bounds and results resident in HBM)
  config.tick_latency  one qpc_control_batch call on 1 / 8 / 64 states, host pointers, synchronous, timed at the C ABI (the reference's regime: one robot)
  cpu_baseline.accuracy_config5   tau of the oracle's OSQP at the notebook's tolerances and of the fp32 tick against the fp64 tick, same states
  roofline_l0        the sparse workgroup kernel of that entry (qpc_qp_sp_kernel) at 8 192 problems per launch, from live HIP events
  degraded           true when a secondary measurement failed or solved < 90 % of its QPs (the line itself stays alive)
"""
import argparse
import ctypes
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SOLVE = 1968  # in 8*(37+36+36+16) + out 8*(36+36+48) + 8
ALG_FLOPS_PER_SOLVE = 4.6e5  # one reduced-KKT factor+solve + assembly + FK/RNEA
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md (spec; 6.29e12 measured)
FP64_PEAK = 78.6e12  # FLOP/s vector fp64 (datasheet)


# ------------------------------------------------------------------------------------------ CPU baseline (child processes)
def _cpu_worker(rank, B_sample, budget_s):
    """One host core: the oracle's reference-layout tick on its own slice of the workload, no torch, no HIP.  Four legs:
      nbw  the regime the reference itself runs on Atlas (notebooks/Standing controller.ipynb:68-71, 202-203): ONE model ticked
           every millisecond along a closed-loop trajectory, OSQP at the notebook's tolerances (eps_abs = eps_rel = 1e-5),
           warm-started from the previous tick (momentum.jl:58 calls solve! on the same model) -- budget_s / 3;
      nbc  the same tolerances, cold start over the batch -- budget_s / 4;
      tc   the test-suite tolerances (test/runtests.jl:35-43: eps_abs 1e-8, eps_rel 1e-16, max_iter 20000), cold -- budget_s / 2;
      tw   the test-suite tolerances along the closed loop, warm-started -- budget_s / 3."""
    import qpc_pkg
    qpc_pkg.load()
    from qpcontrol_jl_amd import workloads as W
    from oracle import oracle as O

    def factory(mech, N, fl):
        return O.OracleController(O.OracleModel(mech, native=True), N, fl)

    oc, wl = W.atlas_workload(factory, B_sample, 3, seed=2025 + 104729 * rank)
    om = oc.model
    chunk = 16

    def cold(budget):
        done, iters = 0, []
        t0 = time.perf_counter()
        while done < B_sample and time.perf_counter() - t0 < budget:
            s = slice(done, done + chunk)
            r = oc.tick_batch(wl.q[s], wl.v[s], wl.desired[s], wl.contact_weight[s], wl.contact_maxnormalforce[s])
            iters += [int(x) for x in r["iters"]]
            done += len(r["iters"])
        return {"done": done, "iters": iters, "seconds": time.perf_counter() - t0}

    def warm(budget):
        # closed-loop 1 kHz trajectories (tick -> semi-implicit Euler with the QP's own v̇), desireds from the StandingController
        # laws at every step; the first tick of a trajectory is cold and not counted; solve time only
        done, iters, sec, traj = 0, [], 0.0, 0
        t1 = time.perf_counter()
        while time.perf_counter() - t1 < budget and traj < B_sample:
            q, v = wl.q[traj].copy(), wl.v[traj].copy()
            for k in range(40):
                des = W.standing_desired(wl.setup, q[None], v[None])[0]
                ta = time.perf_counter()
                r = oc.tick(q, v, des, wl.contact_weight[traj], wl.contact_maxnormalforce[traj], warm=(k > 0))
                tb = time.perf_counter()
                if r["status"] > 1:
                    break
                if k > 0:
                    done += 1; iters.append(int(r["info"].iters)); sec += tb - ta
                v = v + 1e-3 * r["vdot"]
                q = om.integrate(q, v, 1e-3)
            traj += 1
        return {"done": done, "iters": iters, "seconds": sec}

    out = {}
    oc.set_settings(O.notebook_settings(polish=0))
    out["nbw"] = warm(budget_s / 3)
    out["nbc"] = cold(budget_s / 4)
    oc.set_settings(O.test_settings(polish=0))
    out["tc"] = cold(budget_s / 2)
    out["tw"] = warm(budget_s / 3)
    print(json.dumps(out))


def _histogram(iters, max_iter=20000):
    edges = [0, 100, 250, 500, 1000, 2500, 5000, 10000, max_iter]
    it = np.asarray(iters)
    h = {f"{lo}-{hi}": int(((it >= lo) & (it < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])}
    h[f">={max_iter} (iteration limit)"] = int((it >= max_iter).sum())
    return h


def cpu_baseline(budget_s=12.0, max_workers=64):
    """The oracle (CPU restatement of the reference path: 143 x 178 reference-layout problem, coefficient rewrite per tick,
    OSQP-algorithm ADMM, RNEA) timed on the host cores of the box: one independent controller per process over disjoint
    slices, four bounded legs (_cpu_worker).  `value` is the all-cores rate in the regime the reference itself
    runs on Atlas -- notebook tolerances, warm-started closed loop; the tight-tolerance legs, against which
    the 1e-8 parity is judged, stand beside it.  The workers are child processes (`--cpu-worker`): no torch, no HIP."""
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
        out, _ = p.communicate(timeout=budget_s * 8 + 180)
        if p.returncode == 0 and out.strip():
            res.append(json.loads(out.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    if not res:
        raise RuntimeError("cpu_baseline: no worker finished")

    def leg(key, closed_loop):
        rs = [r[key] for r in res]
        iters = [i for r in rs for i in r["iters"]]
        done = sum(r["done"] for r in rs)
        if closed_loop:  # per-process solve time: the rates of the processes add up
            rate = sum(r["done"] / r["seconds"] for r in rs if r["seconds"] > 0)
        else:
            rate = done / max(max(r["seconds"] for r in rs), 1e-9)
        single = rs[0]["done"] / rs[0]["seconds"] if rs[0]["seconds"] > 0 else None
        return {"value": rate, "single_core": single, "ticks": done,
                "admm_iterations": {"mean": float(np.mean(iters)) if iters else None, "median": float(np.median(iters)) if iters else None,
                                    "histogram": _histogram(iters) if iters else None}}

    nbw, nbc, tc, tw = leg("nbw", True), leg("nbc", False), leg("tc", False), leg("tw", True)
    nbw["note"] = ("the regime the reference runs on Atlas (notebooks/Standing controller.ipynb:68-71, 202-203): ONE controller ticked at "
                   "1 kHz along closed-loop trajectories (tick -> Euler step with the QP's v̇, StandingController laws re-evaluated), OSQP "
                   "at eps_abs = eps_rel = 1e-5 warm-started from the previous tick as momentum.jl:58 does; solve time only; tau then "
                   "agrees with the fp64 optimum to ~1e-3 only")
    nbc["note"] = "notebook tolerances, cold start over the batch (no previous tick)"
    tc["note"] = ("test-suite tolerances (test/runtests.jl:35-43: eps_abs 1e-8, eps_rel 1e-16, max_iter 20000), cold start: the setting the "
                  "1e-8 parity of the GPU path is judged against; a quarter of the cold solves end at the iteration limit")
    tw["note"] = "test-suite tolerances along the closed loop, warm-started"
    return {"value": nbw["value"], "unit": "QP solves/s", "cores": len(res), "kind": "port",
            "regime": "notebook tolerances (1e-5), warm-started closed loop, all cores -- the reference's own Atlas regime",
            "single_core": nbw["single_core"],
            "notebook_warm_closed_loop": nbw, "notebook_cold": nbc, "tight_cold": tc, "tight_warm_closed_loop": tw,
            "sample": f"{nbw['ticks']} warm-started closed-loop ticks (value) + {nbc['ticks']} / {tc['ticks']} / {tw['ticks']} ticks of the other "
                      f"legs, all on the same Atlas config-3 workload over {len(res)} processes (one controller each, disjoint seeded "
                      f"slices); {wall:.1f} s for all four legs with process start-up; host has {os.cpu_count()} cores"}


# ------------------------------------------------------------------------------------------ device-side helpers
class DeviceRun:
    """A controller + one workload resident in HBM (torch tensors), launched through qpc_control_batch_device."""

    def __init__(self, torch, binding, W, B, config, seed, device_index, fused_laws=False, f32=False, pack2=False, rotate=1):
        self.torch, self.B = torch, B
        self.rotate, self.nstep = max(1, rotate), 0
        real_finalize = binding.Controller.finalize

        def fin(self_, device_id=0):
            return real_finalize(self_, device_index)
        binding.Controller.finalize = fin
        try:
            def factory(m, N, fl):
                c = binding.Controller(binding.Model(m), N, fl)
                if f32:
                    c.set_precision(1)  # the fp32 variant of the tick (qpc_set_precision): BASELINE config 5's stated precision
                if pack2:
                    c.set_qps_per_wavefront(2)  # two QPs per wavefront (lanes 0-31 / 32-63)
                return c
            self.ctrl, self.wl = W.atlas_workload(factory, B, config, seed=seed, fused_laws=fused_laws)
        finally:
            binding.Controller.finalize = real_finalize
        self.lay = lay = self.ctrl.layout
        self.dev = dev = torch.device("cuda", device_index)
        wl = self.wl

        def dev_t(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        self.q, self.v = dev_t(wl.q), dev_t(wl.v)
        self.des = None if fused_laws else dev_t(wl.desired)
        self.cw, self.cm = dev_t(wl.contact_weight), dev_t(wl.contact_maxnormalforce)
        # rotate > 1 (measurement only, --rotate-inputs): that many distinct copies of the input record, taken in turn -- consecutive
        # launches then read different memory, so that no cache between the shader and HBM (the 256 MB MALL) can serve a launch from the
        # previous one's reads
        self.in_sets = [(self.q, self.v, self.des, self.cw, self.cm)]
        for _ in range(self.rotate - 1):
            self.in_sets.append(tuple(None if t is None else t.clone() for t in self.in_sets[0]))
        self.tau = torch.empty((B, lay.nv), dtype=torch.float64, device=dev)
        self.vd = torch.empty((B, lay.nv), dtype=torch.float64, device=dev)
        self.cwr = torch.empty((B, max(lay.n_contacts, 1) * 6), dtype=torch.float64, device=dev)
        self.status = torch.empty(B, dtype=torch.int32, device=dev)
        self.iters = torch.empty(B, dtype=torch.int32, device=dev)
        self.stream = torch.cuda.current_stream(dev)

    def step(self):
        vp = ctypes.c_void_p
        L, lay = self.ctrl.L, self.lay

        def ptr(t):
            return vp(t.data_ptr()) if t is not None and t.numel() else vp(0)
        nc = lay.n_contacts
        q_, v_, des_, cw_, cm_ = self.in_sets[self.nstep % self.rotate]
        self.nstep += 1
        rc = L.qpc_control_batch_device(self.ctrl.h, self.B, ptr(q_), ptr(v_), ptr(des_), ptr(cw_) if nc else vp(0),
                                        ptr(cm_) if nc else vp(0), vp(0), vp(0), ptr(self.tau), ptr(self.vd),
                                        ptr(self.cwr) if nc else vp(0), ptr(self.status), ptr(self.iters), vp(self.stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(L.qpc_last_error().decode())

    def timed(self, steps, warmup):
        """(wall seconds for `steps` launches, average kernel ms from HIP events on the launch stream, launches)"""
        L = self.ctrl.L
        for _ in range(warmup):
            self.step()
        self.torch.cuda.synchronize(self.dev)
        L.qpc_timing_enable(self.ctrl.h, 1)
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.torch.cuda.synchronize(self.dev)
        el = time.perf_counter() - t0
        avg_ms, nl = ctypes.c_double(0), ctypes.c_int64(0)
        L.qpc_timing_read(self.ctrl.h, ctypes.byref(avg_ms), ctypes.byref(nl))
        L.qpc_timing_enable(self.ctrl.h, 0)
        return el, avg_ms.value, int(nl.value)


_ACCURACY_SAMPLE = {}


def accuracy_leg():
    """cpu_baseline leg: what the reference's own regime delivers in ACCURACY on BASELINE config 5 -- the oracle's OSQP at the notebook's
    tolerances (eps_abs = eps_rel = 1e-5, Standing controller.ipynb:68-71; cold start) on a sample of the states the fp32 tick was run on,
    its tau against the fp64 tick's (exact to 1e-8 against the oracle at the test suite's tolerances: tests/test_gpu_parity.py), in the
    same measure as `vs_f64` (relative to max(1, |tau|_inf)), with the fp32 tick's error on the SAME sample beside it.  An answer either
    of them calls OPTIMAL is what `checkstatus` (momentum.jl:83-91) lets through."""
    S = _ACCURACY_SAMPLE
    if not S:
        return None
    try:
        from oracle import oracle as O
        from qpcontrol_jl_amd import workloads as W

        def ora_factory(mech, N, fl):
            return O.OracleController(O.OracleModel(mech), N, fl)
        oc, _ = W.atlas_workload(ora_factory, 1, S["config"])
        oc.set_settings(O.notebook_settings(polish=0))
        t0 = time.perf_counter()
        ro = oc.tick_batch(S["q"], S["v"], S["desired"], S["cw"], S["cm"])
        sec = time.perf_counter() - t0
        ok64 = S["s64"] <= 1
        scale = np.maximum(1.0, np.abs(S["tau64"]).max(axis=1, keepdims=True))
        eo = (np.abs(ro["tau"] - S["tau64"]) / scale).max(axis=1)

        def dist(e):
            if not len(e):
                return None
            return {"qps": int(len(e)), "rel_tau_err_median": float(np.median(e)), "rel_tau_err_p99": float(np.percentile(e, 99)),
                    "rel_tau_err_p999": float(np.percentile(e, 99.9)), "rel_tau_err_max": float(e.max()),
                    "fraction_beyond_1e-4": float((e > 1e-4).mean()), "fraction_beyond_1e-3": float((e > 1e-3).mean())}
        return {"sample": f"the first {len(eo)} states of BASELINE config 5 (seed {S['seed']}), one core, {sec:.1f} s",
                "reference_regime_osqp_eps_1e-5_status_optimal": dist(eo[(ro["status"] == 0) & ok64]),
                "fp32_tick_status_optimal_same_sample": dist(S["err32"][(S["s32"] == 0) & ok64]),
                "note": "tau against the fp64 tick of the same states, relative to max(1, |tau|_inf); the reference runs Atlas at these tolerances and "
                        "accepts every such answer (momentum.jl:83-91)"}
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def secondary_points(torch, binding, W, device_index):
    """The literal BASELINE.json batches on this GPU (device-resident inputs, same kernel), outside the headline's timed
    region: config 2 @ 1024, config 3 @ 8192, config 5 @ 32768, config 3 @ 262144."""
    out = []
    tau5 = {}
    for config, B, steps, f32 in ((2, 1024, 50, False), (3, 8192, 30, False), (5, 32768, 20, False), (3, 262144, 5, False),
                                  (5, 32768, 20, True), (5, 8192, 30, True), (3, 32768, 20, True)):
        run = DeviceRun(torch, binding, W, B, config, {2: 2024, 3: 2025, 5: 2027}[config], device_index, f32=f32)
        el, kms, nl = run.timed(steps, 3)
        st = run.status.cpu().numpy()
        pt = {"config": config, "batch": B, "dtype": "f32" if f32 else "f64", "solves_per_s": B * steps / el, "ms_per_launch": 1e3 * el / steps,
              "kernel_ms": kms, "optimal_fraction": float((st <= 1).mean()),
              "mean_active_set_iters": float((run.iters.cpu().numpy() % 1000).mean()), "lds_bytes_per_qp": int(run.lay.lds_bytes_per_qp)}
        if config == 5 and B == 32768:  # BASELINE config 5 in both precisions on the same states: what fp32 costs in accuracy
            tau5[f32] = (run.tau.cpu().numpy(), st)
            if len(tau5) == 2:
                (t64, s64), (t32, s32) = tau5[False], tau5[True]
                both = (s64 <= 1) & (s32 <= 1)
                err = (np.abs(t32 - t64) / np.maximum(1.0, np.abs(t64).max(axis=1, keepdims=True))).max(axis=1)
                pt["vs_f64"] = {"rel_tau_err_median": float(np.median(err[both])), "rel_tau_err_p999": float(np.percentile(err[both], 99.9)),
                                "rel_tau_err_max": float(err[both].max()), "optimal_in_f64_not_in_f32": float(((s64 <= 1) & ~(s32 <= 1)).mean()),
                                "optimal_in_f32_not_in_f64": float((~(s64 <= 1) & (s32 <= 1)).mean())}
                # the same over the QPs the fp32 tick reports as OPTIMAL (status 0 only: what `checkstatus`, momentum.jl:83-91, licenses without a remark)
                opt = (s32 == 0) & (s64 <= 1)
                if opt.any():
                    pt["vs_f64"]["of_status_optimal_only"] = {
                        "qps": int(opt.sum()), "rel_tau_err_median": float(np.median(err[opt])), "rel_tau_err_p99": float(np.percentile(err[opt], 99)),
                        "rel_tau_err_p999": float(np.percentile(err[opt], 99.9)), "rel_tau_err_max": float(err[opt].max()),
                        "beyond_1e-4": int((err[opt] > 1e-4).sum()), "beyond_1e-3": int((err[opt] > 1e-3).sum())}
                pt["vs_f64"]["unresolved_fraction"] = float(((s64 <= 1) & ~(s32 <= 1)).mean())
                # a sample of the same states for cpu_baseline's accuracy leg (what the REFERENCE's own regime delivers on them)
                ns = 1024
                wl = run.wl
                _ACCURACY_SAMPLE.update(config=5, seed=2027, q=wl.q[:ns].copy(), v=wl.v[:ns].copy(), desired=wl.desired[:ns].copy(), cw=wl.contact_weight[:ns].copy(),
                                        cm=wl.contact_maxnormalforce[:ns].copy(), tau64=t64[:ns].copy(), s64=s64[:ns].copy(), err32=err[:ns].copy(), s32=s32[:ns].copy())
        out.append(pt)
        del run
    return out


def ordered_point(torch, binding, W, device_index):
    """Cost-ordered tickets (qpc_ticket_order_device / qpc_control_batch_device_ordered): config 3 at BASELINE's literal batch, ONE launch
    at a time, the QPs handed out by descending working-set changes of the previous tick of the same batch (what a control loop has:
    momentum.jl:58 solves the same model every tick).  Outputs are bit-identical to the batch order."""
    vp = ctypes.c_void_p
    B, steps = 8192, 30
    run = DeviceRun(torch, binding, W, B, 3, 2025, device_index)
    L = run.ctrl.L
    if not hasattr(L, "qpc_control_batch_device_ordered"):
        return None
    order = torch.zeros(B, dtype=torch.int32, device=run.dev)
    p = lambda t: vp(t.data_ptr())  # noqa: E731

    def launch(o):
        rc = L.qpc_control_batch_device_ordered(run.ctrl.h, 0, 0, B, p(run.q), p(run.v), p(run.des), p(run.cw), p(run.cm), vp(0), vp(0), p(run.tau),
                                                p(run.vd), p(run.cwr), p(run.status), p(run.iters), o, vp(run.stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(L.qpc_last_error().decode())
    launch(vp(0))
    tau0 = run.tau.clone()
    if L.qpc_ticket_order_device(run.ctrl.h, 0, B, p(run.iters), 1000, p(order), vp(run.stream.cuda_stream)) != 0:
        raise RuntimeError(L.qpc_last_error().decode())
    out = {"config": 3, "batch": B, "launches": steps}
    for name, o in (("batch_order", vp(0)), ("cost_order", p(order))):
        for _ in range(3):
            launch(o)
        torch.cuda.synchronize(run.dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            launch(o)
        torch.cuda.synchronize(run.dev)
        out[name + "_solves_per_s"] = B * steps / (time.perf_counter() - t0)
    out["bitwise_equal"] = bool(torch.equal(run.tau, tau0))
    return out


def pipelined_points(torch, binding, W, device_index):
    """Two INDEPENDENT batches in flight on two launch slots / streams of the device (qpc_control_batch_device_slot): the workgroups of
    one launch fill the ~110 us tail of the other.  Not the headline (a step there is one launch at a time): what a caller with several
    independent batches gets.  Config 3 at BASELINE's literal batch (8 192) and at the headline batch (32 768); distinct output arrays."""
    import ctypes as ct
    out = []
    vp = ct.c_void_p
    for B, steps in ((8192, 60), (32768, 30)):
        run = DeviceRun(torch, binding, W, B, 3, 2025, device_index)
        L, lay, dev = run.ctrl.L, run.lay, run.dev
        if not hasattr(L, "qpc_control_batch_device_slot"):
            return None
        sets = []
        for k in range(2):
            sets.append(dict(stream=torch.cuda.Stream(dev), tau=torch.empty_like(run.tau), vd=torch.empty_like(run.vd), cwr=torch.empty_like(run.cwr),
                             status=torch.empty_like(run.status), iters=torch.empty_like(run.iters)))
        torch.cuda.synchronize(dev)

        def launch(k):
            o = sets[k]
            p = lambda t: vp(t.data_ptr())  # noqa: E731
            rc = L.qpc_control_batch_device_slot(run.ctrl.h, 0, k, B, p(run.q), p(run.v), p(run.des), p(run.cw), p(run.cm), vp(0), vp(0), p(o["tau"]),
                                                 p(o["vd"]), p(o["cwr"]), p(o["status"]), p(o["iters"]), vp(o["stream"].cuda_stream))
            if rc != 0:
                raise RuntimeError(L.qpc_last_error().decode())
        for i in range(4):
            launch(i & 1)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            launch(i & 1)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        same = all(bool(torch.equal(sets[0][k], sets[1][k])) for k in ("tau", "status"))
        out.append({"config": 3, "batch": B, "launches": steps, "slots": 2, "solves_per_s": B * steps / el, "ms_per_launch": 1e3 * el / steps,
                    "optimal_fraction": float((sets[0]["status"] == 0).float().mean().item()), "both_slots_bitwise_equal": same})
        del run, sets
    return out


def closed_loop_point():
    """SURVEY.md 8 f-2, outside the headline's timed region: tools/bench_rollout.py (32 768 robots x 50 controller ticks with
    fused PD laws + simulation steps on the device) in a child process; robot-steps/s or None"""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_rollout.py")], capture_output=True, text=True, timeout=240)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"robot_steps_per_s": d["value"], "robots": d["robots"], "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                "failed_qps": d["failed_qps"]}
    except Exception as e:  # noqa: BLE001 -- a diagnostic extra must never take the bench line down
        return {"error": repr(e)}


L0_ALG_NOTE = ("algorithmic bytes per problem at the boundary of qpc_qp_solve: in 8 (nnz P + n + nnz A + 2 m), out 8 (n + m) + 8 (status, iters); "
               "the path is bound by the LATENCY of its dependent steps (pivots, LDS round trips), not by memory or arithmetic: DESIGN.md 7d")


def l0_plugin_point():
    """SURVEY.md 8 b-1, outside the timed region: the solver-level plug on the committed Atlas problems (what unmodified QPControl.jl
    hands to its optimizer slot: 143 x 178 with a diagonal P, 146 x 181 with a matrix-weighted MomentumRateTask) -- the prepared entry
    QPCOptimizer binds (qpc_qp_setup / qpc_qp_solve, cold and warm-started) and the dense entry qpc_qp_solve_batch;
    tools/bench_latency.py --l0 in a child process.  -> (config.l0_plugin, roofline_l0)"""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_latency.py"), "--l0", "1", "8192"], capture_output=True, text=True,
                           timeout=400)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        pr, pm = d.get("prepared", {}), d.get("prepared_matrix_weight", {})
        big = pr.get("8192", {})
        dense = {"latency_us_B1": d["1"]["median_us"], "latency_us_B1_median_over_problems": d.get("B1_median_over_problems_us"),
                 "latency_us_B1_worst_problem": d.get("B1_worst_problem_us"), "solves_per_s_B8192": d["8192"]["solves_per_s"],
                 "max_abs_err_vdot_vs_fixture": max(d["1"]["max_abs_err_vdot_vs_fixture"], d["8192"]["max_abs_err_vdot_vs_fixture"])}
        point = {"n": d["n"], "m": d["m"],
                 # the optimizer's life cycle (qpc_qp_setup once, qpc_qp_solve per tick: values only, warm start): what QPCOptimizer binds
                 "entry": "qpc_qp_setup / qpc_qp_solve (pattern once, values per call; host pointers, synchronous, B = 1 unless stated)",
                 "plan": pr.get("plan"),
                 "latency_us_B1_cold_median_over_problems": pr.get("B1_cold_median_over_problems_us"),
                 "latency_us_B1_cold_worst_problem": pr.get("B1_cold_worst_problem_us"),
                 "latency_us_B1_warm_median_over_problems": pr.get("B1_warm_median_over_problems_us"),
                 "latency_us_B1_warm_worst_problem": pr.get("B1_warm_worst_problem_us"),
                 "warm_means": pr.get("warm_means"),
                 "solves_per_s_B8192_cold": big.get("cold_solves_per_s"), "solves_per_s_B8192_warm": big.get("warm_solves_per_s"),
                 "solves_per_s_B8192_page_locked_cold": big.get("page_locked_cold_solves_per_s"), "solves_per_s_B8192_page_locked_warm": big.get("page_locked_warm_solves_per_s"),
                 # values, bounds and results resident in HBM (qpc_qp_solve_device): the rate of the path without PCIe
                 "solves_per_s_B8192_device_resident_cold": big.get("device_resident_cold_solves_per_s"),
                 "solves_per_s_B8192_device_resident_warm": big.get("device_resident_warm_solves_per_s"),
                 "max_abs_err_vdot_vs_fixture": big.get("max_abs_err_vdot_vs_fixture"),
                 "matrix_weighted_task_146x181": {"plan": pm.get("plan"),
                                                  "latency_us_B1_cold_median_over_problems": pm.get("B1_cold_median_over_problems_us"),
                                                  "latency_us_B1_cold_worst_problem": pm.get("B1_cold_worst_problem_us"),
                                                  "latency_us_B1_warm_median_over_problems": pm.get("B1_warm_median_over_problems_us"),
                                                  "max_abs_err_vdot_vs_fixture": max([p["max_abs_err_vdot_vs_fixture"] for p in pm.get("B1_per_problem", [])] or [None])},
                 # ONE robot, 1 kHz, every solve warm-started from the tick before: the regime the reference runs the slot in (cpu_baseline.l0_fixture_osqp.closed_loop_sequence beside it)
                 "closed_loop": d.get("closed_loop"),
                 "dense_entry_qpc_qp_solve_batch": dense,
                 "reference_cpu": "cpu_baseline.l0_fixture_osqp: the oracle's OSQP on the same committed problems, one core, measured in this run"}
        roof = None
        if big.get("cold_kernel_ms"):
            n, m, nnzP, nnzA = d["n"], d["m"], 63, 1004
            alg = 8 * (nnzP + n + nnzA + 2 * m) + 8 * (n + m) + 8
            ach = alg * 8192 / (big["cold_kernel_ms"] * 1e-3)
            madds = (pr.get("plan") or {}).get("elimination_madds")
            roof = {"bound": "hbm", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach / HBM_PEAK, "traffic": None,
                    "kernel": "qpc_qp_sp_kernel", "kernel_ms": big["cold_kernel_ms"], "kernel_ms_warm": big.get("warm_kernel_ms"), "problems_per_launch": 8192,
                    "problems_per_s_of_device_time": 8192 / (big["cold_kernel_ms"] * 1e-3), "algorithmic_bytes_per_problem": alg,
                    "elimination_multiply_adds_per_problem": madds,
                    "dense_elimination_multiply_adds_per_problem": (pr.get("plan") or {}).get("dense_elimination_madds"),
                    "timing": "HIP events on the problem's stream around the kernels of one qpc_qp_solve (qpc_qp_enable_timing), live in this run",
                    "note": L0_ALG_NOTE}
        return point, roof
    except Exception as e:  # noqa: BLE001 -- a diagnostic extra must never take the bench line down
        return {"error": repr(e)}, None


def tick_latency_point():
    """The tick in the reference's own regime -- ONE robot, one call per tick: qpc_control_batch on host pointers, synchronous, timed at the
    C ABI (tools/bench_latency.py in a child process, outside the timed region).  Microseconds per call for 1 / 8 / 64 states."""
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_latency.py")], capture_output=True, text=True, timeout=300)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out = {"entry": "qpc_control_batch, host pointers in / out, synchronous, one call; Atlas config 3; caller arrays pageable / page-locked"}
        for mode in ("pageable", "pinned"):
            out[mode] = {f"B{B}": {"median_us": v["median_us"], "p99_us": v["p99_us"]} for B, v in d["batches_" + mode].items() if B in ("1", "8", "64")}
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def l0_cpu_leg():
    """cpu_baseline leg for the solver-level plug: the oracle's OSQP restatement on the SAME committed problems the device entry is
    timed on (tests/golden/l0), one core, setup once and then update + solve per call as `Parametron.solve!` does (momentum.jl:58):
    microseconds per solve at the notebook's tolerances (Standing controller.ipynb:68-71) and at the test suite's (test/runtests.jl:35-43),
    cold and warm-started from the same problem's previous solution (how tools/bench_latency.py times the device's warm start)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_latency as BL
        from oracle import oracle as O
        out = {"unit": "us per solve, one core", "warm_means": "the same problem solved again from its previous solution"}
        for which in ("diag", "matrix"):
            P, q, A, l, u, x, nv = BL.load_l0_fixture(which)
            res = {}
            for name, st in (("notebook_tolerances", O.notebook_settings(0)), ("test_suite_tolerances", O.test_settings(0))):
                for warm in (False, True):
                    r = O.qp_time(P, q, A, l, u, st, reps=3, warm=warm)
                    res[name + ("_warm" if warm else "_cold")] = {"median_over_problems": float(np.median(r["us"])), "worst_problem": float(r["us"].max()),
                                                                  "best_problem": float(r["us"].min()), "admm_iterations": [int(i) for i in r["iters"]]}
            out["atlas_143x178" if which == "diag" else "atlas_matrix_weight_146x181"] = res
        # the regime the reference runs its optimizer in: consecutive ticks of ONE closed loop, each solve warm-started from the tick before
        # (the same committed sequence config.l0_plugin.closed_loop times the device on)
        P, q, A, l, u, x, nv = BL.load_l0_fixture("closed_loop")
        cl = {"ticks": int(P.shape[0] - 1), "what": "tests/golden/l0/atlas_closed_loop_143x178.npz: every problem solved once from the previous problem's solution (update + solve)"}
        for name, st in (("notebook_tolerances", O.notebook_settings(0)), ("test_suite_tolerances", O.test_settings(0))):
            r = O.qp_time(P, q, A, l, u, st, reps=1, warm=2)
            us, it = r["us"][1:], r["iters"][1:]
            scale = np.maximum(1.0, np.abs(x[1:, :nv]).max(axis=1))
            cl[name] = {"median_us": float(np.median(us)), "p90_us": float(np.percentile(us, 90)), "worst_tick_us": float(us.max()), "mean_us": float(us.mean()),
                        "admm_iterations": {"mean": float(it.mean()), "median": float(np.median(it)), "max": int(it.max())},
                        "not_optimal": int((r["status"][1:] != 0).sum())}
        out["closed_loop_sequence"] = cl
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def host_pointer_path(binding, run, reps=5):
    """PCIe-inclusive rate of the host-pointer entry qpc_control_batch on the headline workload (never `value`):
    with the caller's arrays page-locked (qpc_host_alloc / qpc_host_register, what the Julia shim does with the buffers
    it owns: the kernel reads and writes them in place) and with plain pageable arrays (staged in chunks)."""
    c, wl, lay, B = run.ctrl, run.wl, run.lay, run.B
    L = c.L
    ins = dict(q=wl.q, v=wl.v, d=wl.desired, cw=wl.contact_weight, cm=wl.contact_maxnormalforce)
    res = {}
    for mode in ("pinned", "pageable"):
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
            a[...] = 0
        for _ in range(2):
            c.control_batch(arr["q"], arr["v"], arr["d"], arr["cw"], arr["cm"], out=out)
        t0 = time.perf_counter()
        for _ in range(reps):
            c.control_batch(arr["q"], arr["v"], arr["d"], arr["cw"], arr["cm"], out=out)
        res[mode] = B * reps / (time.perf_counter() - t0)
        res[mode + "_optimal_fraction"] = float((out["status"] == 0).mean())
    return res


def measured_traffic(fingerprint):
    """HBM bytes per launch of the tick kernel from the newest rocprofv3 PMC summary under profiles/ whose kernel
    fingerprint equals the built kernel's (tools/profile_round.sh + tools/summarize_pmc.py); None if there is none."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic_summary.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("kernel_fingerprint") == fingerprint and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            best = (f, d)
    if best is None:
        return None, "no profile under profiles/ carries the fingerprint of the built kernel (" + fingerprint + "): traffic not quoted"
    f, d = best
    qps = d.get("qps_per_launch", 32768)
    src = f"{os.path.relpath(f, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB as reported on gfx950), {qps} QPs per launch, kernel fingerprint {fingerprint} == built kernel"
    fetch, write = d["FETCH_SIZE"]["per_launch_KiB"] * 1024.0, d["WRITE_SIZE"]["per_launch_KiB"] * 1024.0
    fr, wr = d.get("FETCH_SIZE_inputs_rotated"), d.get("WRITE_SIZE_inputs_rotated")
    if not fr or not fr.get("launches"):
        return fetch + write, src + "; no pass with rotated inputs in this profile: FETCH as counted (identical inputs of consecutive launches may have been served by the memory-side cache)"
    # the plain pass counted FEWER bytes fetched than the kernel must read.  The second pair of passes cycles the inputs through
    # 9 distinct copies (295 MB, more than the 256 MB memory-side cache), so that no launch can be served from its predecessor's reads:
    must_read = (ALG_BYTES_PER_SOLVE - 968) * qps  # the input half of the algorithmic bytes (1 000 B in, 968 B out per solve)
    fetch_r = fr["per_launch_KiB"] * 1024.0
    write_r = wr["per_launch_KiB"] * 1024.0 if wr and wr.get("launches") else write
    if fetch_r >= 0.97 * must_read:
        return fetch_r + write_r, (src + f"; inputs ROTATED through 9 copies: FETCH {fetch_r / 1e6:.1f} MB per launch >= the {must_read / 1e6:.1f} MB the kernel must read "
                                   f"(plain pass, identical inputs every launch: {fetch / 1e6:.1f} MB -- the shortfall was reuse between launches in the memory-side cache, "
                                   f"not under-counting: no x2 correction); traffic = FETCH + WRITE of the rotated passes")
    # ... it did not.  tools/fetch_calib.hip read KNOWN byte counts in the tick's access pattern (profiles/r05_fetch_calibration.json): FETCH_SIZE is
    # TCC_EA0_RDREQ x 64 B while every request is a 128-B line -- exactly half the bytes for streams of 16 and 8 B per lane and for aligned
    # records alike -- and unaligned per-QP records fetched by wavefronts on different XCDs bring their shared lines in more than once
    # (x 1.38 for 296-B records, x 1.94 for 64-B ones).  So: FETCH doubled, and what then exceeds the compulsory bytes is partial lines.
    calib = ""
    try:
        cj = json.load(open(os.path.join(ROOT, "profiles", "r05_fetch_calibration.json")))["kernels"]
        calib = (f"; calibration on known byte counts (profiles/r05_fetch_calibration.json): counted / known = {cj['calib_stream16']['factor']:.2f} for a 16 B/lane stream, "
                 f"{cj['calib_stream8']['factor']:.2f} for 8 B/lane, {cj['calib_records<64>']['factor']:.2f} for aligned 512-B records, i.e. every 128-B request tallied at 64 B; "
                 f"unaligned records drawn by wavefronts on different XCDs fetch shared lines more than once: 2 x counted / known = {2 * cj['calib_records<37>']['factor']:.2f} for 296-B records, "
                 f"{2 * cj['calib_records<8>']['factor']:.2f} for 64-B records")
    except (OSError, ValueError, KeyError):
        pass
    return 2.0 * fetch_r + write_r, (src + f"; inputs ROTATED through 9 copies: FETCH still {fetch_r / 1e6:.1f} MB per launch (plain pass {fetch / 1e6:.1f} MB) < the {must_read / 1e6:.1f} MB the kernel must read: "
                                     f"not reuse between launches, the counter under-reports -- FETCH doubled as MI355X_MICROARCH.md prescribes" + calib +
                                     f"; traffic = 2 FETCH + WRITE of the rotated passes = {(2.0 * fetch_r + write_r) / 1e6:.1f} MB per launch against {ALG_BYTES_PER_SOLVE * qps / 1e6:.1f} MB algorithmic "
                                     f"(the excess is partial 128-B lines of the per-QP records; at < 2 % of the HBM rate it costs no time)")


def measured_issue(fingerprint):
    """What actually bounds the tick: the instructions it issues.  Per-solve instruction counts from the newest
    rocprofv3 SQ counter summary under profiles/ taken on the built kernel (same fingerprint rule as the traffic), and the
    share of a QP slot (one QP on each of the two waves of a SIMD) that their issue time accounts for."""
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(f))
            t = json.load(open(f.replace("_pmc_summary.json", "_traffic_summary.json")))
        except (OSError, ValueError):
            continue
        pl = d.get("per_launch", {})
        if t.get("kernel_fingerprint") != fingerprint or "SQ_INSTS_VALU" not in pl:
            continue
        n = float(d.get("qps_per_launch", 32768))
        valu, lds, salu = pl["SQ_INSTS_VALU"] / n, pl.get("SQ_INSTS_LDS", 0.0) / n, pl.get("SQ_INSTS_SALU", 0.0) / n
        out = {"valu_per_solve": valu, "lds_per_solve": lds, "salu_per_solve": salu, "source": os.path.relpath(f, ROOT)}
        if "SQ_ACTIVE_INST_ANY" in pl and "SQ_WAVE_CYCLES" in pl and pl["SQ_WAVE_CYCLES"] > 0:
            # both counters are in quad-cycles summed over waves: the share of its life a wave spends executing an instruction;
            # the two waves of a SIMD do not overlap each other's instructions (calibrated), so twice that share is
            # how full the SIMD's issue is
            share = pl["SQ_ACTIVE_INST_ANY"] / pl["SQ_WAVE_CYCLES"]
            out["wave_active_share"] = share
            out["simd_issue_busy"] = min(1.0, 2.0 * share)
        if "valu_lane_utilisation" in pl:
            out["valu_lane_utilisation"] = pl["valu_lane_utilisation"]
            out["valu_active_lanes_per_instruction"] = 64.0 * pl["valu_lane_utilisation"]
        out["note"] = ("the tick is bound by the number of instructions it issues: every added VALU instruction costs 8 cycles of QP-slot "
                       "time whether dependent or not (DESIGN.md 7b)")
        return out
    return None


# ------------------------------------------------------------------------------------------ main
def in_library_leg(n_dev, B, steps, warmup, config, ids=None, lib_path=None):
    """BASELINE config 4's split the way `QPControlHIP.jl`'s `finalize!(...; devices = ...)` does it: ONE process,
    qpc_finalize_devices(ctrl, N, ids), a device-resident shard of B QPs on every device, the same timed loop
    (W untimed + K timed steps; a step = one launch per device through qpc_control_batch_device_on, then every device synchronised).
    ids: the devices (default 0 .. N-1; the test hook QPC_BENCH_TEST_ONE_DEVICE=1 puts every shard on device 0).  lib_path: another
    build of the library -- the CPU test runs this leg on the emulation's fake devices (numpy arrays instead of torch tensors).  -> dict"""
    torch = None
    if lib_path is None:
        import torch
    import qpc_pkg
    qpc_pkg.load()
    from qpcontrol_jl_amd import binding
    from qpcontrol_jl_amd import workloads as W
    if ids is None:
        # (the test hook of a one-GPU box: qpc_finalize_devices takes every device once, so the leg runs on device 0 alone)
        ids = [0] if os.environ.get("QPC_BENCH_TEST_ONE_DEVICE") == "1" else list(range(n_dev))
    n_dev = len(ids)
    real_finalize = binding.Controller.finalize

    def fin(self_, device_id=0):
        return real_finalize(self_, devices=ids)
    binding.Controller.finalize = fin
    try:
        ctrl, wl0 = W.atlas_workload(lambda m, N, fl: binding.Controller(binding.Model(m, lib_path), N, fl), B, config, seed={2: 2024, 3: 2025, 5: 2027}[config])
    finally:
        binding.Controller.finalize = real_finalize
    L, lay = ctrl.L, ctrl.layout
    vp = ctypes.c_void_p
    shards = []
    for g, dev_id in enumerate(ids):
        if g == 0:
            wl = wl0
        else:  # every shard its own seeded states (shard g of the global batch)
            q, v = W.standing_states(wl0.mech, B, {2: 2024, 3: 2025, 5: 2027}[config] + 7919 * g)
            wl = W.Workload(wl0.name, wl0.mech, wl0.setup, q, v, W.standing_desired(wl0.setup, q, v), wl0.contact_weight, wl0.contact_maxnormalforce,
                            wl0.floating_body)
        if torch is not None:
            dev = torch.device("cuda", dev_id)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
            mk = lambda shape, dt: torch.empty(shape, dtype=torch.float64 if dt == np.float64 else torch.int32, device=dev)  # noqa: E731
        else:
            t = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
            mk = lambda shape, dt: np.zeros(shape, dtype=dt)  # noqa: E731
        sh = dict(q=t(wl.q), v=t(wl.v), des=t(wl.desired), cw=t(wl.contact_weight), cm=t(wl.contact_maxnormalforce),
                  tau=mk((B, lay.nv), np.float64), vd=mk((B, lay.nv), np.float64), cwr=mk((B, max(lay.n_contacts, 1) * 6), np.float64),
                  status=mk((B,), np.int32), iters=mk((B,), np.int32))
        sh["stream"] = torch.cuda.Stream(device=dev) if torch is not None else None
        shards.append(sh)

    def step():
        nc = lay.n_contacts
        for g, sh in enumerate(shards):
            p = (lambda x: vp(x.data_ptr()) if x.numel() else vp(0)) if torch is not None else (lambda x: vp(x.ctypes.data) if x.size else vp(0))  # noqa: E731
            rc = L.qpc_control_batch_device_on(ctrl.h, g, B, p(sh["q"]), p(sh["v"]), p(sh["des"]), p(sh["cw"]) if nc else vp(0), p(sh["cm"]) if nc else vp(0),
                                               vp(0), vp(0), p(sh["tau"]), p(sh["vd"]), p(sh["cwr"]) if nc else vp(0), p(sh["status"]), p(sh["iters"]),
                                               vp(sh["stream"].cuda_stream) if sh["stream"] is not None else vp(0))
            if rc != 0:
                raise RuntimeError(L.qpc_last_error().decode())

    def sync():
        for sh in shards:
            if sh["stream"] is not None:
                sh["stream"].synchronize()
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    el = time.perf_counter() - t0
    opt = [float(((sh["status"].cpu().numpy() if torch is not None else sh["status"]) == 0).mean()) for sh in shards]
    return {"n_devices": n_dev, "device_ids": ids, "batch_per_device": B, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps,
            "solves_per_s": n_dev * B * steps / el, "optimal_fraction_per_device": opt,
            "entry": "one process: qpc_finalize_devices + qpc_control_batch_device_on per device (device-resident shards, one stream per device)"}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: the parent -- before anything here has touched the GPU -- starts N fresh child
    processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR = 127.0.0.1 / MASTER_PORT in their environment: the same
    contract torch.distributed.run fulfils), lets rank 0's JSON line through, then adds the in-library leg (one process, N devices)
    to it.  Exit code: non-zero when a child fails."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out0 or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(rcs) or line is None:
        sys.stderr.write(f"bench.py: ranks exited with {rcs}\n")
        if out0:
            sys.stdout.write(out0)
        raise SystemExit(1)
    d = json.loads(line)
    # the in-library leg in a process of its own (this one must stay clear of the GPU: it may be asked to launch again)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--in-library-leg", str(args.gpus), str(args.batch_per_gpu), str(args.steps), str(args.warmup),
                        str(args.config)], capture_output=True, text=True)
    try:
        d["config"]["in_library_devices"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception:  # noqa: BLE001
        d["config"]["in_library_devices"] = {"error": (r.stderr or "no output")[-400:]}
        d["degraded"] = True
        d.setdefault("degraded_why", []).append("in_library_devices leg failed")
    print(json.dumps(d))


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
    if len(sys.argv) >= 7 and sys.argv[1] == "--in-library-leg":
        print(json.dumps(in_library_leg(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=32768)
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 5])
    ap.add_argument("--f32", action="store_true", help="the fp32 variant of the tick (qpc_set_precision): not the headline (tau 1e-4 instead of 1e-8)")
    ap.add_argument("--pack2", action="store_true", help="two QPs per wavefront (qpc_set_qps_per_wavefront(2))")
    ap.add_argument("--rotate-inputs", type=int, default=1, help="measurement only: cycle through that many distinct copies of the input record "
                    "(9 copies of config 3 at 32 768 are 295 MB: more than the 256 MB MALL) -- tools/profile_round.sh's traffic pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the literal-BASELINE-batch points and the host-pointer path")
    ap.add_argument("--fused-laws", action="store_true",
                    help="SURVEY 8 f-1: StandingController PD laws (standing.jl:58-89) evaluated in the kernel, no `desired` "
                         "input (a pure (q, v) -> tau map); default is the reference boundary with desireds as inputs")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])  # (no launcher: this process starts the ranks itself, before it touches the GPU)

    import torch
    import qpc_pkg
    qpc_pkg.load()
    from qpcontrol_jl_amd import binding
    from qpcontrol_jl_amd import workloads as W

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE = {world} but --gpus {args.gpus}")
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
    # every rank builds its own contiguous shard [rank*B, (rank+1)*B) of the global batch
    run = DeviceRun(torch, binding, W, B, args.config, {2: 2024, 3: 2025, 5: 2027}[args.config] + 7919 * rank, local_rank,
                    fused_laws=args.fused_laws, f32=args.f32, pack2=args.pack2, rotate=args.rotate_inputs)
    ctrl, lay, L = run.ctrl, run.lay, run.ctrl.L
    dev = run.dev

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        run.step()
    barrier()
    L.qpc_timing_enable(ctrl.h, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.step()
    barrier()
    elapsed = time.perf_counter() - t0
    avg_ms, nl = ctypes.c_double(0), ctypes.c_int64(0)
    L.qpc_timing_read(ctrl.h, ctypes.byref(avg_ms), ctypes.byref(nl))
    L.qpc_timing_enable(ctrl.h, 0)
    per_rank = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else dev)
        mine = torch.tensor([avg_ms.value, elapsed * 1e3 / args.steps], dtype=torch.float64, device="cpu" if one_device else dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)  # (every rank's kernel time and wall time per step: a straggler GPU shows up here)
        per_rank = [{"rank": r, "kernel_avg_ms": float(x[0]), "ms_per_step": float(x[1])} for r, x in enumerate(allr)]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = run.status.cpu().numpy()
    it_raw = run.iters.cpu().numpy()
    it = it_raw % 1000
    presolved = float((it_raw >= 1000000).mean())  # QPs solved in the reduced dimensions of the null-space presolve
    fp64_measured = ctypes.c_double(0.0)  # vector-fp64 FMA peak measured on this box, outside the timed region
    if rank == 0 and hasattr(L, "qpc_debug_measure_fp64_peak"):
        L.qpc_debug_measure_fp64_peak.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
        if L.qpc_debug_measure_fp64_peak(local_rank, ctypes.byref(fp64_measured)) != 0:
            fp64_measured = ctypes.c_double(0.0)
    n_opt = int((st == 0).sum())

    host_rates, secondary, closed_loop, pipelined, l0_plugin, ordered, roofline_l0 = None, None, None, None, None, None, None
    tick_latency = None
    if rank == 0 and world == 1 and not args.no_secondary:
        # diagnostic extras, measured after the headline's timed region: none of them may take the bench line down
        if not args.fused_laws:
            try:
                host_rates = host_pointer_path(binding, run)
            except Exception as e:  # noqa: BLE001
                host_rates = {"error": repr(e)}
        try:
            secondary = secondary_points(torch, binding, W, local_rank)
        except Exception as e:  # noqa: BLE001
            secondary = {"error": repr(e)}
        try:
            pipelined = pipelined_points(torch, binding, W, local_rank)
        except Exception as e:  # noqa: BLE001
            pipelined = {"error": repr(e)}
        try:
            ordered = ordered_point(torch, binding, W, local_rank)
        except Exception as e:  # noqa: BLE001
            ordered = {"error": repr(e)}
        closed_loop = closed_loop_point()
        l0_plugin, roofline_l0 = l0_plugin_point()
        tick_latency = tick_latency_point()

    if rank == 0:
        total = B * world * args.steps
        value = total / elapsed
        kern_s = avg_ms.value * 1e-3
        # compulsory bytes at the boundary = inputs + outputs (+ status, iters)
        nc_ = int(lay.n_contacts)
        alg_bytes = 8 * (lay.nq + lay.nv + (0 if args.fused_laws else lay.n_desired) + 2 * nc_) + 8 * (2 * lay.nv + 6 * nc_) + 8
        assert args.config != 3 or args.fused_laws or alg_bytes == ALG_BYTES_PER_SOLVE
        ach_bytes = alg_bytes * B / kern_s if kern_s > 0 else 0.0
        fingerprint = L.qpc_kernel_fingerprint().decode() if hasattr(L, "qpc_kernel_fingerprint") else "unknown"
        traffic, traffic_source = (None, "measured for config 3 at 32768 QPs per launch only")
        if args.config == 3 and not args.fused_laws and not args.f32:
            traffic, traffic_source = measured_traffic(fingerprint)
            if traffic is not None and B != 32768:
                traffic = traffic * B / 32768.0
        out = {
            "metric": "QP solves/sec (Atlas ~30-DoF whole-body QP, batched)",
            "value": value, "unit": "QP solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.f32 else "f64", "data": "synthetic",
            "config": {"workload": f"atlas_config{args.config}: synthetic Atlas (nq 37, nv 36), StandingController task set, "
                                   + ("8 ContactPoints x 4-ray friction cones, floating base wrench balance "
                                      "(143 vars / 114 eq / 64 ineq in the reference layout)" if args.config != 2
                                      else "no contacts (39 vars / 36 eq)"),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"shard{world} (independent QPs, no collective)",
                       "per_rank": per_rank,
                       "optimal_fraction": n_opt / B, "mean_active_set_iters": float(it.mean()),
                       "presolved_fraction": presolved,
                       "lds_bytes_per_qp": int(lay.lds_bytes_per_qp),
                       "desired": "device PD laws (standing.jl:58-89 fused)" if args.fused_laws else "input record",
                       "host_pointer_path_solves_per_s": host_rates.get("pinned") if host_rates else None,
                       "host_pointer_path_pageable_solves_per_s": host_rates.get("pageable") if host_rates else None,
                       "host_pointer_path_optimal_fraction": ([host_rates.get("pinned_optimal_fraction"), host_rates.get("pageable_optimal_fraction")]
                                                              if host_rates else None),
                       "host_pointer_path_error": host_rates.get("error") if host_rates else None,
                       "host_pointer_path_note": "qpc_control_batch on this workload, PCIe inclusive, synchronous calls: page-locked caller "
                                                 "arrays are read / written in place by the kernel (what QPControlHIP.jl does with the "
                                                 "buffers it owns); pageable arrays are staged chunk by chunk by worker threads",
                       "secondary": secondary,
                       "pipelined": pipelined,
                       "cost_ordered_tickets": ordered,
                       "closed_loop": closed_loop,
                       "l0_plugin": l0_plugin,
                       "tick_latency": tick_latency,
                       "rotate_inputs": args.rotate_inputs,
                       "kernel_fingerprint": fingerprint},
            "roofline": {"bound": "hbm", "achieved": ach_bytes / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach_bytes / HBM_PEAK,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "qpc_tick_kernel_jit", "kernel_avg_ms": avg_ms.value, "launches": int(nl.value),
                         "algorithmic_bytes_per_solve": alg_bytes,
                         "note": "path is bound by instruction issue (fp64 VALU + LDS + scalar), not by HBM (SURVEY.md 8d, DESIGN.md 7b); "
                                 "fp64 fraction and the instruction counts reported beside it",
                         "issue": measured_issue(fingerprint) if args.config == 3 and not args.fused_laws and not args.f32 else None,
                         "fp64": {"achieved_tflops": ALG_FLOPS_PER_SOLVE * B / kern_s / 1e12 if kern_s > 0 else 0.0,
                                  "peak_tflops": FP64_PEAK / 1e12,
                                  "frac": (ALG_FLOPS_PER_SOLVE * B / kern_s) / FP64_PEAK if kern_s > 0 else 0.0,
                                  "algorithmic_flops_per_solve": ALG_FLOPS_PER_SOLVE,
                                  "peak_measured_tflops": fp64_measured.value or None,
                                  "frac_of_measured_peak": ((ALG_FLOPS_PER_SOLVE * B / kern_s) / (fp64_measured.value * 1e12)
                                                            if kern_s > 0 and fp64_measured.value > 0 else None)}},
        }
        out["roofline_l0"] = roofline_l0
        # a regression in a secondary path must not hide in a JSON field: flagged at the top level
        why = []
        if n_opt / B < 0.9:
            why.append(f"headline optimal_fraction {n_opt / B:.3f}")
        for name, x in (("host_pointer_path", host_rates), ("secondary", secondary), ("pipelined", pipelined), ("closed_loop", closed_loop),
                        ("l0_plugin", l0_plugin), ("cost_ordered_tickets", ordered)):
            if isinstance(x, dict) and x.get("error"):
                why.append(f"{name}: {x['error']}")
        if isinstance(host_rates, dict):
            why += [f"host_pointer_path {k} {host_rates[k]:.3f}" for k in ("pinned_optimal_fraction", "pageable_optimal_fraction")
                    if k in host_rates and host_rates[k] < 0.9]
        for name, pts in (("secondary", secondary), ("pipelined", pipelined)):
            if isinstance(pts, list):  # config 5 holds infeasible QPs by construction (93.5 % feasible): its floor is lower
                why += [f"{name} config {x['config']} @ {x['batch']}: optimal_fraction {x['optimal_fraction']:.3f}" for x in pts
                        if x["optimal_fraction"] < (0.85 if x["config"] == 5 else 0.9)]
        if isinstance(closed_loop, dict) and closed_loop.get("failed_qps", 0) > 0.1 * closed_loop.get("robots", 1) * closed_loop.get("steps", 1):
            why.append(f"closed_loop failed_qps {closed_loop['failed_qps']}")
        out["degraded"] = bool(why)
        if why:
            out["degraded_why"] = why
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["l0_fixture_osqp"] = l0_cpu_leg()
            acc = accuracy_leg()
            if acc is not None:
                out["cpu_baseline"]["accuracy_config5"] = acc
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
