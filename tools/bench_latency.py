#!/usr/bin/env python3
"""Latency of one controller tick through the host-pointer entry point (qpc_control_batch: H2D, kernel, D2H,
synchronous) for small batches -- the regime of the reference itself (ONE robot, one tick per millisecond).
python tools/bench_latency.py  -> one JSON line (median / p99 microseconds per call; not the headline metric)
python tools/bench_latency.py --l0 [B ...]  -> the same for the solver-level plug (b-1: qpc_qp_solve_batch, what the reference's
    own optimizer slot sees: the dense 143 x 178 Atlas problem of tests/golden/l0/atlas_143x178.npz), latency at B = 1 and rate at
    larger batches"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W  # noqa: E402


L0_FIXTURES = {"diag": "atlas_143x178.npz",                   # the standing controller as the notebook builds it: P diagonal
               "matrix": "atlas_matrix_weight_146x181.npz",   # ... with a matrix-weighted MomentumRateTask: a 6 x 6 block in P
               "closed_loop": "atlas_closed_loop_143x178.npz"}  # 96 CONSECUTIVE ticks of one robot at 1 kHz (a push at tick 32, a contact lost at 64)


def load_l0_fixture(which="diag"):
    """The committed reference-layout problems (inputs + expected x), densified: P[k,n,n], q, A[k,m,n], l, u, x, nv"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "l0", L0_FIXTURES[which]))
    n, m, k = int(g["n"]), int(g["m"]), g["q"].shape[0]
    P, A = np.zeros((k, n, n)), np.zeros((k, m, n))
    P[:, g["P_idx"][:, 0], g["P_idx"][:, 1]] = g["P_val"]
    A[:, g["A_idx"][:, 0], g["A_idx"][:, 1]] = g["A_val"]
    return P, g["q"], A, g["l"], g["u"], g["x"], int(g["nv"])


def l0_main(batches):
    try:  # the device-resident leg hands torch tensors to the library: torch's HIP runtime has to be initialised first (as in bench.py)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001
        pass
    P, q, A, l, u, x, nv = load_l0_fixture()
    k = P.shape[0]
    out = {"metric": "L0 qpc_qp_solve_batch on the dense 143 x 178 Atlas reference-layout QP (host pointers, synchronous, through ctypes)",
           "unit": "us", "n": int(P.shape[1]), "m": int(A.shape[1])}
    # B = 1 on each of the committed problems by itself: the latency depends on the working-set iterations the QP needs
    per_qp = []
    for b in range(k):
        a = (P[b], q[b], A[b], l[b], u[b])
        r = binding.qp_solve_batch(*a)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            r = binding.qp_solve_batch(*a)
            ts.append(time.perf_counter() - t0)
        per_qp.append({"iters": int(r["iters"][0]), "median_us": float(np.median(ts) * 1e6)})
    out["B1_per_problem"] = per_qp
    out["B1_median_over_problems_us"] = float(np.median([p["median_us"] for p in per_qp]))
    out["B1_worst_problem_us"] = float(max(p["median_us"] for p in per_qp))
    for B in batches:
        idx = np.arange(B) % k
        a = (P[idx], q[idx], A[idx], l[idx], u[idx])
        r = binding.qp_solve_batch(*a)  # first call: workspace allocation, module load
        assert (r["status"] == 0).all(), r["status"]
        err = float(np.abs(r["x"][:, :nv] - x[idx][:, :nv]).max())
        reps = 20 if B == 1 else 3
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = binding.qp_solve_batch(*a)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        out[str(B)] = {"median_us": float(np.median(ts)), "max_us": float(ts.max()), "solves_per_s": float(B / np.median(ts) * 1e6),
                       "max_abs_err_vdot_vs_fixture": err}
    out["prepared"] = prepared_entry(P, q, A, l, u, x, nv, batches)
    # the problems with a matrix-weighted MomentumRateTask (a 6 x 6 block in P: momentum.jl:113-117, test/controller.jl:232-285)
    Pm, qm, Am, lm, um, xm, nvm = load_l0_fixture("matrix")
    out["prepared_matrix_weight"] = prepared_entry(Pm, qm, Am, lm, um, xm, nvm, [b for b in batches if b == 1])
    try:
        out["closed_loop"] = closed_loop_entry()
    except Exception as e:  # noqa: BLE001
        out["closed_loop"] = {"error": repr(e)}
    print(json.dumps(out))


def prepared_entry(P, q, A, l, u, x, nv, batches):
    """the PREPARED entry (qpc_qp_setup once, qpc_qp_solve per call: values only, 13 KB per QP), cold and warm-started from the previous
    call's multipliers -- what QPCOptimizer (julia/QPControlHIPMOI.jl, moi.py) does every tick.  "warm" here = the same problem solved
    again from its own multipliers: the best case of a warm start (bench.py times the CPU's OSQP in the same way beside it)"""
    k = P.shape[0]
    pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0))
    Pv, Av = pr.values(P, A)
    pr.analyze(l[0], u[0])  # (what an optimizer does once after `MOI.copy_to`: the first solve is then not the slow one)
    prep = {"n": int(P.shape[1]), "m": int(A.shape[1]), "plan": pr.plan_info(), "B1_per_problem": [],
            "warm_means": "the same problem solved again from its own multipliers"}
    # (timed: the call itself on arrays prepared beforehand -- contiguous, bounds clipped, outputs preallocated, pointers converted once
    #  (`QpProblem.bind_call`): what a caller with its own buffers pays, the Julia shim among them; `QpProblem.solve` clips and allocates
    #  on every call, a millisecond at B = 8 192, and numpy's `.ctypes` conversions alone cost `solve_raw` 10 us per call)
    lc, uc = np.clip(l, -1e20, 1e20), np.clip(u, -1e20, 1e20)
    for b in range(k):
        a = tuple(np.ascontiguousarray(z[b:b + 1]) for z in (Pv, q, Av, lc, uc))
        r = rw = o = pr.solve_raw(*a)
        cold, warm = pr.bind_call(*a, o), pr.bind_call(*a, o, warm_start=True)  # (pointers converted once: the call is the C ABI's)
        tc, tw = [], []
        for _ in range(10):
            t0 = time.perf_counter()
            rc = cold()
            tc.append(time.perf_counter() - t0)
            assert rc == 0
        itc, xc = int(r["iters"][0]), r["x"][0, :nv].copy()
        assert r["status"][0] == 0
        for _ in range(10):
            t0 = time.perf_counter()
            rc = warm()
            tw.append(time.perf_counter() - t0)
            assert rc == 0
        assert rw["status"][0] == 0
        prep["B1_per_problem"].append({"iters": itc, "cold_median_us": float(np.median(tc) * 1e6),
                                       "warm_iters": int(rw["iters"][0]), "warm_median_us": float(np.median(tw) * 1e6),
                                       "max_abs_err_vdot_vs_fixture": float(np.abs(xc - x[b, :nv]).max())})
    prep["B1_cold_median_over_problems_us"] = float(np.median([p["cold_median_us"] for p in prep["B1_per_problem"]]))
    prep["B1_cold_worst_problem_us"] = float(max(p["cold_median_us"] for p in prep["B1_per_problem"]))
    prep["B1_warm_median_over_problems_us"] = float(np.median([p["warm_median_us"] for p in prep["B1_per_problem"]]))
    prep["B1_warm_worst_problem_us"] = float(max(p["warm_median_us"] for p in prep["B1_per_problem"]))
    for B in batches:
        if B == 1:
            continue
        idx = np.arange(B) % k
        res, kc, kw = {}, [], []
        for mode in ("pageable", "page_locked"):  # the caller's arrays: plain numpy, or qpc_host_alloc'ed (what QPControlHIP.jl does with the buffers it owns)
            mk = (lambda shape, dt=np.float64: binding.pinned_array(pr.L, shape, dt)) if mode == "page_locked" else (lambda shape, dt=np.float64: np.zeros(shape, dtype=dt))
            a = []
            for z in (Pv, q, Av, lc, uc):
                zz = mk((B,) + z.shape[1:])
                zz[...] = z[idx]
                a.append(zz)
            o = dict(x=mk((B, pr.n)), y=mk((B, max(pr.m, 1))), status=mk((B,), np.int32), iters=mk((B,), np.int32))
            r = pr.solve_raw(*a, out=o)
            assert (r["status"] == 0).all(), r["status"]
            err = float(np.abs(r["x"][:, :nv] - x[idx][:, :nv]).max())
            tc, tw = [], []
            for _ in range(5):
                t0 = time.perf_counter()
                pr.solve_raw(*a, out=o)
                tc.append(time.perf_counter() - t0)
            for _ in range(5):
                t0 = time.perf_counter()
                pr.solve_raw(*a, out=o, warm_start=True)
                tw.append(time.perf_counter() - t0)
            res[mode] = {"cold_solves_per_s": float(B / np.median(tc)), "warm_solves_per_s": float(B / np.median(tw)), "max_abs_err_vdot_vs_fixture": err}
            if mode == "pageable":
                pr.enable_timing(True)  # the kernel by itself: one launch for the whole batch between two HIP events (a timed call does not go in chunks)
                for _ in range(3):
                    pr.solve_raw(*a, out=o)
                    kc.append(pr.last_kernel_ms())
                for _ in range(3):
                    pr.solve_raw(*a, out=o, warm_start=True)
                    kw.append(pr.last_kernel_ms())
                pr.enable_timing(False)
        dres = {}
        if hasattr(pr.L, "qpc_qp_solve_device"):  # the same batch with values, bounds and results resident in HBM (qpc_qp_solve_device)
            try:
                import torch
                dev = torch.device("cuda", 0)
                di = [torch.from_numpy(np.ascontiguousarray(z[idx])).to(dev) for z in (Pv, q, Av, lc, uc)]
                do = [torch.zeros((B, pr.n), dtype=torch.float64, device=dev), torch.zeros((B, max(pr.m, 1)), dtype=torch.float64, device=dev),
                      torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)]
                strm = torch.cuda.Stream(dev)
                torch.cuda.synchronize(dev)
                call = lambda warm: pr.solve_device(B, *(t.data_ptr() for t in di), *(t.data_ptr() for t in do), warm_start=warm, stream=strm.cuda_stream)  # noqa: E731
                call(False)
                assert (do[2].cpu().numpy() == 0).all()
                derr = float(np.abs(do[0].cpu().numpy()[:, :nv] - x[idx][:, :nv]).max())
                tc, tw = [], []
                for _ in range(5):
                    t0 = time.perf_counter()
                    call(False)
                    tc.append(time.perf_counter() - t0)
                for _ in range(5):
                    t0 = time.perf_counter()
                    call(True)
                    tw.append(time.perf_counter() - t0)
                dres = {"device_resident_cold_solves_per_s": float(B / np.median(tc)), "device_resident_warm_solves_per_s": float(B / np.median(tw)),
                        "device_resident_max_abs_err_vdot_vs_fixture": derr}
                del di, do
            except Exception as e:  # noqa: BLE001
                dres = {"device_resident_error": repr(e)}
        prep[str(B)] = {**dres, "cold_solves_per_s": res["pageable"]["cold_solves_per_s"], "warm_solves_per_s": res["pageable"]["warm_solves_per_s"],
                        "page_locked_cold_solves_per_s": res["page_locked"]["cold_solves_per_s"],
                        "page_locked_warm_solves_per_s": res["page_locked"]["warm_solves_per_s"],
                        "cold_kernel_ms": float(np.median(kc)), "warm_kernel_ms": float(np.median(kw)),  # HIP events around the kernel
                        "max_abs_err_vdot_vs_fixture": max(res["pageable"]["max_abs_err_vdot_vs_fixture"], res["page_locked"]["max_abs_err_vdot_vs_fixture"])}
    pr.close()
    return prep


def closed_loop_entry(reps=5):
    """The reference's own regime for its optimizer slot: ONE robot, a tick every millisecond, every solve warm-started from the tick before
    (Standing controller.ipynb; momentum.jl:58).  The committed sequence of consecutive problems through qpc_qp_setup / qpc_qp_analyze once and
    qpc_qp_solve(warm_start = 1) per tick, B = 1, host pointers, synchronous; per tick the median over `reps` passes (a pass starts cold)."""
    P, q, A, l, u, x, nv = load_l0_fixture("closed_loop")
    K = P.shape[0]
    pr = binding.QpProblem(P.shape[1], A.shape[1], (P != 0).any(0), (A != 0).any(0))
    Pv, Av = pr.values(P, A)
    lc, uc = np.clip(l, -1e20, 1e20), np.clip(u, -1e20, 1e20)
    pr.analyze(lc[0], uc[0])
    args = [tuple(np.ascontiguousarray(z[k:k + 1]) for z in (Pv, q, Av, lc, uc)) for k in range(K)]
    o = pr.solve_raw(*args[0])
    calls = [pr.bind_call(*args[k], o, warm_start=k > 0) for k in range(K)]  # (pointers converted once: the call is the C ABI's, as a Julia `ccall` is)
    t = np.zeros((reps, K))
    iters = np.zeros(K, dtype=int)
    err = 0.0
    for r in range(reps):
        for k in range(K):
            t0 = time.perf_counter()
            rc = calls[k]()
            t[r, k] = time.perf_counter() - t0
            assert rc == 0
            assert o["status"][0] == 0, (k, o["status"][0])
            iters[k] = int(o["iters"][0])
            err = max(err, float(np.abs(o["x"][0, :nv] - x[k, :nv]).max()))
    us = np.median(t, axis=0)[1:] * 1e6  # (the first tick of a pass is the cold one)
    info = pr.plan_info()
    pr.close()
    return {"ticks": int(K - 1), "what": "consecutive ticks of one closed loop (tests/golden/l0/atlas_closed_loop_143x178.npz), each warm-started from the tick before",
            "median_us": float(np.median(us)), "p90_us": float(np.percentile(us, 90)), "worst_tick_us": float(us.max()), "mean_us": float(us.mean()),
            "working_set_iterations_per_tick": {"mean": float(iters[1:].mean()), "max": int(iters[1:].max()), "ticks_without_a_change": int((iters[1:] == 0).sum())},
            "first_tick_cold_us": float(np.median(t, axis=0)[0] * 1e6), "max_abs_err_vdot_vs_fixture": err, "plan_in_use": info["in_use"]}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--l0-closed-loop":
        print(json.dumps(closed_loop_entry()))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--l0":
        return l0_main([int(a) for a in sys.argv[2:]] or [1, 64, 8192])
    def factory(mech, n_basis, floating):
        return binding.Controller(binding.Model(mech), n_basis, floating)
    ctrl, wl = W.atlas_workload(factory, 2048, 3)
    out = {"metric": "tick latency, host pointers in / out, synchronous call (Atlas config 3): at the C ABI (pointers converted once); through_control_batch_*: through the Python method, numpy conversions included", "unit": "us"}
    lay = ctrl.layout
    for mode in ("pageable", "pinned"):
        res = {}
        for B in (1, 8, 64, 512, 2048):
            src = (wl.q[:B], wl.v[:B], wl.desired[:B], wl.contact_weight[:B], wl.contact_maxnormalforce[:B])
            if mode == "pinned":  # page-locked, preallocated: the kernel reads / writes the caller's arrays in place
                a = tuple(binding.pinned_array(ctrl.L, x.shape) for x in src)
                for d, x in zip(a, src):
                    d[...] = x
                mk = lambda shape, dt=np.float64: binding.pinned_array(ctrl.L, shape, dt)  # noqa: E731
            else:
                a = tuple(np.ascontiguousarray(x) for x in src)
                mk = lambda shape, dt=np.float64: np.zeros(shape, dtype=dt)  # noqa: E731
            o = dict(tau=mk((B, lay.nv)), vdot=mk((B, lay.nv)), cwrench_flat=mk((B, 6 * lay.n_contacts)), status=mk((B,), np.int32),
                     iters=mk((B,), np.int32))
            for _ in range(20):
                ctrl.control_batch(*a, out=o)
            ts = []
            for _ in range(300):
                t0 = time.perf_counter()
                r = ctrl.control_batch(*a, out=o)
                ts.append(time.perf_counter() - t0)
            assert (r["status"] <= 1).all()
            ts = np.array(ts) * 1e6
            # the same call at the C ABI: pointers converted once (Controller.bind_call) -- what a C caller or a Julia `ccall` pays
            call = ctrl.bind_call(*a, o)
            tc = []
            for _ in range(300):
                t0 = time.perf_counter()
                rc = call()
                tc.append(time.perf_counter() - t0)
                assert rc == 0
            assert (o["status"] <= 1).all()
            tc = np.array(tc) * 1e6
            res[str(B)] = {"median_us": float(np.median(tc)), "p99_us": float(np.percentile(tc, 99)),
                           "solves_per_s": float(B / np.median(tc) * 1e6),
                           "through_control_batch_median_us": float(np.median(ts)), "through_control_batch_p99_us": float(np.percentile(ts, 99))}
        out["batches_" + mode] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
