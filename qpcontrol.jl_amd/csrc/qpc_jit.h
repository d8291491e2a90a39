// qpc_jit.h -- JIT specialisation of the tick kernel on the frozen task program (hiprtc).
#pragma once
#include <string>
#include <vector>

#include "qpc_env.h"
#include "qpc_program.h"

// source of the specialised translation unit (QpcDims baked in as a constexpr object)
// warm: also emit qpc_tick_kernel_jit_warm (the warm-started variant, qpc_enable_warm_start)
#define QPC_F32_WAVES_PER_SIMD 3 // register budget of the fp32 tick kernel (qpc_jit.cpp) = workgroups per CU / 4 of its launches (qpc_runtime.hip)
// f32: the fp32 variant of the tick (qpc_set_precision): `real` = float, single-precision literals, LDS slab of floats, no simulation
// step in the module
// pack2: two QPs per wavefront (qpc_set_qps_per_wavefront; qpc_device.h QPC_PACK2): lanes 0-31 / 32-63, two LDS slabs per workgroup
std::string qpc_jit_source(const QpcDims &D, const QpcDims &DS, bool warm, bool f32 = false, bool pack2 = false);
// compile for `arch` (e.g. "gfx950"); returns 0 and fills `code` on success, else fills `log`
// disk_hit (optional): set to the file the code object was READ from (QPC_JIT_CACHE_DIR), empty when it was compiled or came from memory
int qpc_jit_compile(const QpcDims &D, const QpcDims &DS, const std::string &arch, bool warm, bool f32, bool pack2, std::vector<char> &code, std::string &log,
                    std::string *disk_hit = nullptr);
// a code object read from the disk cache would not load (another ROCm, a damaged file): the file is removed and the in-memory cache
// dropped, so that the next qpc_jit_compile compiles afresh
void qpc_jit_forget(const std::string &disk_path);
// waves per SIMD the tick kernel of this precision / packing is compiled for (= workgroups per CU / 4 of its persistent grid)
int qpc_jit_tick_waves(bool f32, bool pack2);
// robots per wavefront of the JIT-specialised simulation step (qpc_sim.h): 2 when a robot fits half a wave and two
// LDS slabs fit a workgroup (QPC_SIM_RPW=1 forces one)
int qpc_jit_sim_rpw(const QpcDims &D);
// experiment switch QPC_EXP_DYN_LDS=1: the tick kernels take their LDS slab as dynamic shared memory (the launch passes the bytes)
bool qpc_jit_dynamic_lds();
