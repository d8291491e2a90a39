#!/bin/bash
# libqpc_prof.so: the product library with the L0 translation unit rebuilt with phase stamps (-DQPC_SP_PROF -DQPC_WG_PROF: the
# kernels print the microseconds of their phases for QP 0).  Run from the repo root after `python qpcontrol.jl_amd/build.py`.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/tests/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -DQPC_SP_PROF $EXTRA -c -o $ROOT/tests/_build/l0_prof.o $ROOT/qpcontrol.jl_amd/csrc/qpc_l0.hip
P=$ROOT/qpcontrol.jl_amd/build/prod
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o $ROOT/tests/_build/libqpc_prof.so $P/qpc_runtime.hip.o $ROOT/tests/_build/l0_prof.o $P/qpc_compile.cpp.o $P/qpc_capi_build.cpp.o $P/qpc_jit.cpp.o -lhiprtc
echo $ROOT/tests/_build/libqpc_prof.so
