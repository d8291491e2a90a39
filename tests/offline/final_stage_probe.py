#!/usr/bin/env python3
"""Accuracy of the final stage of the tick WITHOUT its confirming refinement sweep of the full KKT system (emulator builds with
-DQPC_EXP_NOCONFIRM, compared with the default build): python tests/offline/final_stage_probe.py [B=512] [config=3]

Why the sweep stays (512 config-3 states, |dtau| / max(1, |tau|_max), round 3):
  no confirming sweep                                   median 1.2e-10   p99 5.6e-9   max 8.6e-9
  ... proximal sweeps until they stall (-DQPC_EXP_PROX2)        1.2e-10       5.6e-9       8.6e-9
  ... + one or two steps of iterative refinement of the 6 x 6 wrench-space solve (residual of B22 lam_w + G rho = u0
      through the explicit inverse T22 again)                   4.1e-11       6.3e-10      1.75e-9
  ... + X = L11^-1 applied as an operator for u0 = c2 - L21 (X c1) and lam_1 = X'(X c1 - L21' lam_w) instead of the
      explicit S11^-1 = X'X and Z = S11^-1 S12                  3.9e-11       6.3e-10      1.74e-9
The floor that is left is in the DATA of the reduced system (B22 = S22 - L21 L21' cancels), which only a residual of the full
system -- computed from M, G and H, i.e. the confirming sweep -- can see.  The two experimental variants were removed again.

Asked once more after the presolve with folded rows (no leading block: B22 = S22 exactly, no Z): no confirming sweep 8e-11 / 5.7e-9 /
1.3e-8; with the wrench-space refinement step 6e-13 / 6.5e-11 / 1.1e-10 on config 3 -- much better, as predicted -- but 3.8e-5 at the
worst single-contact state of config 5, and relative to a torque scale of 900 N m even 1e-10 is 1e-7 N m.  Skipping the sweep only
where the true residuals are small does not separate the cases: at 1e-14 a third of the QPs skip it (errors <= 3e-12) and the other
two thirds pay for the test; at 1e-13 half skip it with errors up to 1e-10.  The sweep stays unconditional."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W  # noqa: E402


def build(tag, defs):
    out = f"/tmp/libqpc_emul_{tag}.so"
    src = ["emul.cpp", "../../qpcontrol.jl_amd/csrc/qpc_compile.cpp", "../../qpcontrol.jl_amd/csrc/qpc_capi_build.cpp"]
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w"] + defs + ["-o", out] + src + ["-lpthread"],
                   check=True, cwd=os.path.join(ROOT, "tests", "emul"))
    return out


def run(lib, B, config, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c, wl = W.atlas_workload(lambda m, N, fl: binding.Controller(binding.Model(m, lib), N, fl), B, config)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return c.control_batch(wl.q, wl.v, wl.desired, wl.contact_weight, wl.contact_maxnormalforce)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    config = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    plain, nocf = build("plain", []), build("nocf", ["-DQPC_EXP_NOCONFIRM"])
    ref = run(plain, B, config, {})
    ok = ref["status"] == 0
    scale = np.maximum(1.0, np.abs(ref["tau"]).max(axis=1))
    nocf2 = build("nocf2", ["-DQPC_EXP_NOCONFIRM", "-DQPC_EXP_PROX2"])
    for name, lib, env in (("noconfirm", nocf, {}), ("noconfirm+prox2", nocf2, {})):
        r = run(lib, B, config, env)
        same = np.array_equal(r["status"], ref["status"])
        e = np.abs(r["tau"] - ref["tau"]).max(axis=1)[ok] / scale[ok]
        print(f"{name:20s} statuses equal {same}  |dtau|/scale: median {np.median(e):.2e}  p99 {np.percentile(e, 99):.2e}  max {e.max():.2e}")


if __name__ == "__main__":
    main()
