#!/usr/bin/env python3
"""More seeds of tests/test_fuzz_controllers.py than the test-suite runs (random mechanisms, task sets and states against the
oracle):  python tests/offline/fuzz_more.py [emu|gpu] [first=1000] [count=400] [big]
big: mechanisms near the kernels' limits (test_fuzz_controllers.build(..., big=True))"""
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
big = len(sys.argv) > 4 and sys.argv[4] == "big"
if backend == "emu":
    emu.build()
    factory = emu.factory
else:
    factory = lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl)  # noqa: E731
bad = []
for seed in range(first, first + count):
    try:
        g, o = T.build(factory, seed, B=4, big=big), T.build(T.ora_factory, seed, B=4, big=big)
        rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
        ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
        so, sg = ro["status"], rg["status"]
        if big:
            # torques of hundreds: 1e-8 of the torque scale (the suite's mechanisms have |tau| of order one to a hundred and use 1e-8 absolute);
            # and the oracle's INFEASIBLE is OSQP's heuristic, which misfires on badly scaled QPs (tests/test_gpu_parity.py): an OPTIMAL answer
            # of the kernel there must carry the independent KKT certificate, and the state is then compared as one the oracle did not resolve
            g.tol_tau = 1e-8 * max(1.0, float(np.abs(ro["tau"]).max()))
            import kkt_check
            for b in np.where((so == 2) & (sg == 0))[0]:
                c = kkt_check.certify(o.ctrl, o.q[b], o.v[b], o.desired[b], o.cweight[b], o.cmaxf[b], rg["vdot"][b], rg["cwrench"][b],
                                      None if o.tw is None else o.tw[b], None if o.surf is None else o.surf[b])
                assert c["ok"], ("oracle INFEASIBLE, kernel OPTIMAL without a certificate", b, c["primal"], c["stationarity"])
                so[b] = 4
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
