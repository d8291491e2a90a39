#!/usr/bin/env python3
"""GPU diagnostic: where does a warm-started tick differ from the cold one (config 5, neighbour's working set)?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg
qpc_pkg.load()
import test_warm_start as T
from oracle import oracle as O
from qpcontrol_jl_amd import binding, workloads as W

B, config = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 5
c, wl = T._make(lambda m, N, fl: binding.Controller(binding.Model(m), N, fl), B, config)
args = (wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)
cold = c.control_batch(*args)
cold2 = c.control_batch(*args)
print("cold deterministic:", np.array_equal(cold["tau"], cold2["tau"]))
ok = cold["status"] == 0
ws = np.zeros((B, 2), dtype=np.uint64)
r0 = c.control_batch(*args, working_set=ws)
print("warm entry from zeros == cold bitwise:", np.array_equal(r0["tau"], cold["tau"]), np.abs(r0["tau"] - cold["tau"]).max())
for name, w in (("exact", ws.copy()), ("neighbour", np.roll(ws, 1, axis=0).copy()),
                ("garbage", np.random.default_rng(config).integers(0, 2 ** 63, size=(B, 2), dtype=np.uint64))):
    w_in = w.copy()
    r = c.control_batch(*args, working_set=w)
    d = np.abs(r["tau"] - cold["tau"]).max(1)
    bad = np.where(ok & (d > 1e-9))[0]
    print(name, "status equal", np.array_equal(r["status"], cold["status"]), "offenders", len(bad), "max", d[ok].max(),
          "iters cold/warm", (cold["iters"][ok] % 1000).mean(), (r["iters"][ok] % 1000).mean())
    oc, _ = W.atlas_workload(lambda m, N, fl: O.OracleController(O.OracleModel(m), N, fl), 1, config)
    for b in bad[:6]:
        ro = oc.tick(wl.q[b], wl.v[b], wl.desired[b], wl.contact_weight[b], wl.contact_maxnormalforce[b])
        print("  b", b, "d", d[b], "iters", cold["iters"][b], r["iters"][b], "n_active", int((wl.contact_maxnormalforce[b] > 0).sum()),
              "|tau|", np.abs(cold["tau"][b]).max(), "|f|", np.abs(cold["cwrench"][b]).max(), "oracle st", ro["status"],
              "cold-ora", np.abs(cold["tau"][b] - ro["tau"]).max(), "warm-ora", np.abs(r["tau"][b] - ro["tau"]).max(),
              "ws_in", hex(int(w_in[b, 0])), hex(int(w_in[b, 1])), "ws_out", hex(int(w[b, 0])), hex(int(w[b, 1])), "ws_cold", hex(int(ws[b, 0])), hex(int(ws[b, 1])))
