"""In-tree build of libqpcontrol_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libqpcontrol_hip.so")
SOURCES = ["qpc_runtime.hip", "qpc_compile.cpp", "qpc_capi_build.cpp"]
DEPS = SOURCES + ["qpc_tick.h", "qpc_program.h", "qpc_compile.h", "../../include/qpc.h"]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    if not force and up_to_date():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-Wno-unused-result", "-o", OUT] + list(extra)
    for s in SOURCES:
        if s.endswith(".cpp"):
            cmd += ["-x", "hip", os.path.join(CSRC, s)]
        else:
            cmd += [os.path.join(CSRC, s)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else [])
    print(OUT)
