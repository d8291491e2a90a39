#!/usr/bin/env python3
"""Large-sample validation of the HIP path against the oracle (not part of the test-suite: minutes of
CPU time).  N seeded Atlas states per config, the oracle ticks spread over worker processes; every
disagreement is classified with the independent KKT certificate (tests/kkt_check.py).

    python tests/offline/validate_large.py [N=4096] [configs=3,5]
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(config, seed, lo, hi, out, N):
    import qpc_pkg
    qpc_pkg.load()
    from oracle import oracle as O
    from qpcontrol_jl_amd import workloads as W
    # the seeded workload depends on its batch size: generate all N states, tick this worker's slice
    oc, wl = W.atlas_workload(lambda m, Nb, fl: O.OracleController(O.OracleModel(m), Nb, fl), N, config, seed=seed)
    s = slice(lo, hi)
    r = oc.tick_batch(wl.q[s], wl.v[s], wl.desired[s], wl.contact_weight[s], wl.contact_maxnormalforce[s])
    np.savez(out, tau=r["tau"], vdot=r["vdot"], cwrench=r["cwrench"], status=r["status"], iters=r["iters"])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7]))
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    configs = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "3,5").split(",")]
    import qpc_pkg
    qpc_pkg.load()
    import kkt_check
    from oracle import oracle as O
    from qpcontrol_jl_amd import binding, workloads as W
    O.build()
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    nw = max(1, min(ncpu, 32))
    report = {}
    for config in configs:
        seed = 90000 + config
        gc, wl = W.atlas_workload(lambda m, Nb, fl: binding.Controller(binding.Model(m), Nb, fl), N, config, seed=seed)
        rg = gc.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
        bounds = np.linspace(0, N, nw + 1).astype(int)
        procs = []
        for k in range(nw):
            out = f"/tmp/validate_{config}_{k}.npz"
            procs.append((out, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(config), str(seed),
                                                 str(bounds[k]), str(bounds[k + 1]), out, str(N)])))
        parts = []
        for out, p in procs:
            assert p.wait() == 0
            parts.append(dict(np.load(out)))
        ro = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        oko, okg = ro["status"] == 0, rg["status"] == 0
        stalled = oko & (ro["iters"] >= 20000)
        good = oko & ~stalled & okg
        err = np.abs(rg["tau"] - ro["tau"]).max(1)
        oc, _ = W.atlas_workload(lambda m, Nb, fl: O.OracleController(O.OracleModel(m), Nb, fl), 1, config)
        unresolved = []
        for b in np.where((oko & ~okg) | (~oko & okg & (ro["status"] != 4)) | (stalled & okg & (err > 1e-8)) | (good & (err > 1e-8)))[0]:
            if okg[b]:
                c = kkt_check.certify(oc, wl.q[b], wl.v[b], wl.desired[b], wl.contact_weight[b], wl.contact_maxnormalforce[b],
                                      rg["vdot"][b], rg["cwrench"][b])
                if c["ok"]:
                    continue
                unresolved.append((int(b), int(rg["status"][b]), int(ro["status"][b]), float(err[b]), c["primal"], c["stationarity"]))
            else:
                unresolved.append((int(b), int(rg["status"][b]), int(ro["status"][b]), float(err[b]), None, None))
        report[config] = dict(N=N, oracle_status=np.bincount(ro["status"], minlength=6).tolist(),
                              kernel_status=np.bincount(rg["status"], minlength=6).tolist(), oracle_stalled=int(stalled.sum()),
                              compared=int(good.sum()), max_tau_err=float(err[good].max()) if good.any() else None,
                              kernel_missed=int((oko & ~okg).sum()), unresolved=unresolved)
        report[config]["unresolved"] = report[config]["unresolved"][:20] + (["..."] if len(unresolved) > 20 else [])
        report[config]["n_unresolved"] = len(unresolved)
        print(json.dumps({config: report[config]}))
    return report


if __name__ == "__main__":
    main()
