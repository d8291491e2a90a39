#!/usr/bin/env python3
"""More seeds of the simulation step's fuzz (tests/test_device_sim.py::fuzz_case: the device step against the host's articulated-body step)
than the suite runs, and near the kernel's limits:  python tests/offline/fuzz_sim.py [emu|gpu] [first=3000] [count=200] [big]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import qpc_pkg  # noqa: E402

qpc_pkg.load()
import emu  # noqa: E402
import test_device_sim as T  # noqa: E402

backend = sys.argv[1] if len(sys.argv) > 1 else "emu"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
big = len(sys.argv) > 4 and sys.argv[4] == "big"
if backend == "emu":
    emu.build()
    emu.race_report()
    factory = emu.factory
else:
    factory = T.gpu_factory
bad = []
for seed in range(first, first + count):
    try:
        T.check_step(T.fuzz_case(factory, seed, B=3, big=big), dt=float(np.random.default_rng(seed).choice([1e-3, 5e-3])))
    except Exception as e:  # noqa: BLE001
        bad.append((seed, repr(e)[:300]))
    if (seed - first) % 20 == 19:
        print("seeds", first, "..", seed, "failures", len(bad), flush=True)
print("failures:", len(bad), bad[:5])
if backend == "emu":
    print("races / uninitialised / outside:", emu.race_report()[:3])
