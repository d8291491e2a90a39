#!/usr/bin/env python3
"""Offline fuzzing on the GPU: tests/test_fuzz_controllers.build over many seeds, HIP path vs oracle.
python tests/offline/fuzz_gpu.py [first_seed=1000] [count=200]   (QPC_DISABLE_JIT=1 for the generic kernel: no compile time)"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
import parity  # noqa: E402
import test_fuzz_controllers as TF  # noqa: E402
from qpcontrol_jl_amd import binding  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cnt, worst, failures = collections.Counter(), 0.0, []
    for seed in range(first, first + count):
        g = TF.build(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), seed, B=4)
        o = TF.build(TF.ora_factory, seed, B=4)
        rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
        ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
        for a, b in zip(rg["status"], ro["status"]):
            cnt[(int(a), int(b))] += 1
        so, sg = ro["status"], rg["status"]
        concl = (so == 0) | (so == 2)
        try:
            assert np.array_equal((sg == 0)[concl], (so == 0)[concl]), (sg, so)
            if (so == 0).any() or (~concl & (sg == 0)).any():
                parity.compare(g, rg, o, ro)
            both = (so == 0) & (sg == 0)
            if both.any():
                worst = max(worst, float(np.abs(rg["tau"] - ro["tau"])[both].max()))
        except AssertionError as e:
            failures.append((seed, str(e)[:200]))
    print("status pairs (kernel, oracle):", dict(cnt))
    print("max |tau - tau_oracle| where both optimal:", worst)
    print("failures:", failures)


if __name__ == "__main__":
    main()
