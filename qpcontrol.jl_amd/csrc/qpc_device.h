// qpc_device.h -- device-side definitions of the macros qpc_tick.h is written against (the CPU
// emulation in tests/emul/emul.cpp provides lane-serial counterparts).  gfx950, wave64.
#pragma once

#define QPC_FN __device__ __forceinline__
#define QPC_PHASE(lane) { const int lane = (int)(threadIdx.x & 63);
#define QPC_END_PHASE } __syncthreads();
#define QPC_SINCOS(x, s, c) sincos((x), &(s), &(c))
#define QPC_SQRT(x) sqrt((double)(x))
#define QPC_FABS(x) fabs((double)(x))
#define QPC_UNROLL_FULL _Pragma("unroll")
// lane-private storage that persists across phases (registers on the device)
#define QPC_LANE_ARRAY(name, n) double name[n]
#define QPC_LA(name, lane) name
#define QPC_LANE_INTS(name, n) int name[n]
#define QPC_LI(name, lane) name
#define QPC_CTZ64(x) (__ffsll((unsigned long long)(x)) - 1)
#ifdef QPC_JIT
#define QPC_UNROLL_JIT _Pragma("unroll")
#define QPC_UNROLL _Pragma("unroll")
#else
#define QPC_UNROLL _Pragma("unroll 4")
#endif

namespace qpc {
// wave-uniform reductions over the 64 lanes (butterfly over ds_bpermute/DPP; no LDS traffic)
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double y = __shfl_xor(x, o, 64); x = y > x ? y : x; }
  return x;
}
__device__ __forceinline__ double wave_min(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double y = __shfl_xor(x, o, 64); x = y < x ? y : x; }
  return x;
}
__device__ __forceinline__ int wave_min_int(int x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int y = __shfl_xor(x, o, 64); x = y < x ? y : x; }
  return x;
}
} // namespace qpc

// reductions that straddle a phase boundary: *_PUT inside the phase (every lane, exactly once),
// *_GET after QPC_END_PHASE.  `name` is a double (and `iname` an int) declared before the phase.
#define QPC_RED_MAX_PUT(slot, name, val) name = qpc::wave_max(val)
#define QPC_RED_MAX_GET(slot, name) (name)
#define QPC_RED_ARGMIN_PUT(slot, name, iname, val, idx) \
  do { double v__ = (val); name = qpc::wave_min(v__); iname = qpc::wave_min_int(v__ == name ? (int)(idx) : 0x3fffffff); } while (0)
#define QPC_RED_ARGMIN_GET(slot, name, iname) ((void)0)
