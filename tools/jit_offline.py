#!/usr/bin/env python3
"""Offline compile of the JIT-specialised tick kernel of the Atlas config-3 controller (no GPU needed):
writes the translation unit, compiles it with hipcc for gfx950 with the same options as qpc_jit.cpp and prints
the register / LDS / scratch usage of every kernel.  Extra hipcc options after `--`.
python tools/jit_offline.py [config=3] [outdir=/tmp/jit] [-- -mllvm -foo ...]"""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qpc_pkg  # noqa: E402

qpc_pkg.load()
from qpcontrol_jl_amd import binding, workloads as W  # noqa: E402


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        k = argv.index("--")
        argv, extra = argv[:k], argv[k + 1:]
    config = int(argv[0]) if argv else 3
    out = argv[1] if len(argv) > 1 else "/tmp/jit"
    os.makedirs(out, exist_ok=True)
    real = binding.Controller.finalize
    holder = {}

    def fin(self, device_id=0, devices=None):  # no device here: stop before qpc_finalize
        holder["c"] = self
        raise StopIteration
    binding.Controller.finalize = fin
    try:
        W.atlas_workload(lambda mech, N, fl: binding.Controller(binding.Model(mech), N, fl), 4, config,
                         fused_laws=os.environ.get("QPC_OFFLINE_FUSED") == "1")
    except StopIteration:
        pass
    binding.Controller.finalize = real
    c = holder["c"]
    if os.environ.get("QPC_OFFLINE_WARM") == "1":
        c.enable_warm_start()
    f32 = os.environ.get("QPC_OFFLINE_F32") == "1"  # the fp32 variant of the tick (qpc_set_precision)
    if f32:
        c.set_precision(1)
    if os.environ.get("QPC_OFFLINE_PACK2") == "1":  # two QPs per wavefront (qpc_set_qps_per_wavefront)
        c.set_qps_per_wavefront(2)
    src = os.path.join(out, "tick_jit.hip")
    c.L.qpc_debug_jit_source.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    rc = c.L.qpc_debug_jit_source(c.h, src.encode())
    assert rc == 0, c.L.qpc_last_error()
    # hiprtc pre-declares the device builtins and has no <stdint.h>: swap the four typedef lines for the headers
    lines = open(src).read().split("\n")
    assert lines[0].startswith("typedef int int32_t;") and lines[3].startswith("typedef unsigned long long uint64_t;")
    open(src, "w").write("#include <hip/hip_runtime.h>\n#include <stdint.h>\n" + "\n".join(lines[4:]))
    csrc = os.path.join(ROOT, "qpcontrol.jl_amd", "csrc")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp",
           "-I", csrc, "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj",
           "-o", os.path.join(out, "tick_jit.o"), src] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=out)
    if r.returncode != 0:
        print(r.stderr[-4000:])
        sys.exit(1)
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            print(name)
        for key in ("VGPRs:", "AGPRs", "SGPRs:", "ScratchSize", "Occupancy", "LDS Size", "VGPR Spill", "SGPR Spill"):
            if key in line and name:
                print("   ", line.split("remark:")[-1].strip())


if __name__ == "__main__":
    main()
