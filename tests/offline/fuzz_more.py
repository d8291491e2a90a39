#!/usr/bin/env python3
"""More seeds of tests/test_fuzz_controllers.py than the test-suite runs (random mechanisms, task sets and states against the
oracle):  python tests/offline/fuzz_more.py [emu|gpu] [first=1000] [count=400]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import qpc_pkg  # noqa: E402

qpc_pkg.load()
import emu  # noqa: E402
import parity  # noqa: E402
import test_fuzz_controllers as T  # noqa: E402
from qpcontrol_jl_amd import binding  # noqa: E402

backend = sys.argv[1] if len(sys.argv) > 1 else "emu"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 400
if backend == "emu":
    emu.build()
    factory = emu.factory
else:
    factory = lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl)  # noqa: E731
bad = []
for seed in range(first, first + count):
    try:
        g, o = T.build(factory, seed, B=4), T.build(T.ora_factory, seed, B=4)
        rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
        ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
        so, sg = ro["status"], rg["status"]
        concl = (so == 0) | (so == 2)
        assert np.array_equal((sg == 0)[concl], (so == 0)[concl]), (sg, so)
        if (so == 0).any() or (~concl & (sg == 0)).any():
            parity.compare(g, rg, o, ro)
    except Exception as e:  # noqa: BLE001
        bad.append((seed, repr(e)[:300]))
    if (seed - first) % 10 == 9:
        print("seeds", first, "..", seed, "failures", len(bad), flush=True)
print("failures:", len(bad), bad[:5])
if backend == "emu":
    print("races / uninitialised / outside:", emu.race_report()[:3])
