#!/usr/bin/env python3
"""Every scenario of tests/scenarios.py that takes a batch size, at a larger B than the test-suite
uses, HIP path vs oracle (tests/parity.py rules).  python tests/offline/validate_scenarios.py [B=256]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
import parity  # noqa: E402
import scenarios as SC  # noqa: E402
from oracle import oracle as O  # noqa: E402
from qpcontrol_jl_amd import binding  # noqa: E402


def gpu_factory(mech, n_basis, floating):
    return binding.Controller(binding.Model(mech), n_basis, floating)


def ora_factory(mech, n_basis, floating):
    return O.OracleController(O.OracleModel(mech), n_basis, floating)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    makers = {"fixed_base_hard": lambda f: SC.fixed_base_joint_space(f, True, B=B), "fixed_base_soft": lambda f: SC.fixed_base_joint_space(f, False, B=B),
              "acrobot_point": lambda f: SC.acrobot_point_task(f, B=B), "mixed_tasks": lambda f: SC.mixed_tasks(f, B=B),
              "mixed_tasks_fixed": lambda f: SC.mixed_tasks(f, B=B, floating=False, seed=78)}
    for name, mk in makers.items():
        g, o = mk(gpu_factory), mk(ora_factory)
        rg = g.ctrl.control_batch(g.q, g.v, g.desired, g.cweight, g.cmaxf, g.tw, g.surf)
        ro = o.ctrl.tick_batch(o.q, o.v, o.desired, o.cweight, o.cmaxf, o.tw, o.surf)
        n = parity.compare(g, rg, o, ro)
        import numpy as np
        print(name, "B", g.B, "oracle optimal", n, "kernel statuses", np.bincount(rg["status"], minlength=5).tolist(),
              "oracle statuses", np.bincount(ro["status"], minlength=5).tolist())


if __name__ == "__main__":
    main()
