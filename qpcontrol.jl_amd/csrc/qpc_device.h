// qpc_device.h -- device-side definitions of the macros qpc_tick.h is written against (the CPU
// emulation in tests/emul/emul.cpp provides lane-serial counterparts).  gfx950, wave64.
#pragma once

#include "qpc_real.h" // `real`: double, or a float class in the fp32 module (QPC_F32, its own hiprtc translation unit)
#define QPC_FN __device__ __forceinline__
#define QPC_LAMBDA_INLINE __attribute__((always_inline))
// The lane index is made OPAQUE once per QP (per function scope that runs phases: QPC_LANE_SCOPE).  Without it the compiler
// treats everything derived from the lane (table addresses, lane-dependent constants of the unrolled factorization phases,
// ...) as invariant of the persistent per-QP loop, hoists dozens of such values out of it and, out of registers, spills
// them -- to be reloaded from scratch in the hot phases (measured: 77 spilled VGPRs, 136 MB of extra HBM reads per
// 32768-QP launch).  Opaque per QP the values are recomputed once per QP and shared by its phases: 2 spilled VGPRs,
// 25.6 M solves/s; opaque per PHASE (-DQPC_LANE_PER_PHASE) nothing is spilled (205 VGPRs before the register-resident RNEA
// inputs) but every phase redoes its address arithmetic (+18 % VALU instructions): 24.2 M.
// QPC_PACK2: TWO QPs per wavefront.  The tick uses at most ~36 of its 64 lanes at a time (measured: 56 % of the lanes of an average
// vector instruction are active, profiles/r04_*_lane_utilisation), and the kernel is bound by the NUMBER of instructions it issues: so
// lanes 0-31 work one QP and lanes 32-63 another, each on its own LDS slab, and one instruction stream advances both.  The source is
// written for 64 "virtual" lanes per QP; a half-wave runs a phase in up to two TRIPS (virtual lane = physical lane + 32 trip), and a
// phase whose virtual lanes 32.. have nothing to do (bodies, rays, rows: all <= 32 for Atlas) takes one.  What used to be
// wave-uniform (statuses, iteration counts, tolerances from reductions) is uniform per HALF: data-dependent branches diverge between
// the two QPs and the hardware's exec mask serialises them -- single-wave workgroups have no barrier instruction to deadlock on.
#ifdef QPC_PACK2
#define QPC_W 32
#else
#define QPC_W 64
#endif
__device__ __forceinline__ int qpc_opaque_lane() {
  int l = (int)(threadIdx.x & (QPC_W - 1));
  asm volatile("" : "+v"(l));
  return l;
}
#if defined(QPC_PACK2)
#define QPC_TRIPS 2
#define QPC_LANE_SCOPE const int qpc_lane_q = qpc_opaque_lane();
#define QPC_LANE_IN_SCOPE 1
// per-lane block for the virtual lanes below `nact` (a compile-time bound of the lanes that have work in this phase)
#define QPC_PHASE_N(lane, nact) { _Pragma("unroll") for (int qpc_trip = 0; qpc_trip < QPC_TRIPS; qpc_trip++) { if (QPC_W * qpc_trip >= (nact)) break; const int lane = qpc_lane_q + QPC_W * qpc_trip;
#define QPC_PHASE(lane) QPC_PHASE_N(lane, 64)
#elif defined(QPC_NO_OPAQUE_LANE)
#define QPC_PHASE(lane) { const int lane = (int)(threadIdx.x & 63);
#define QPC_LANE_SCOPE
#elif defined(QPC_LANE_PER_PHASE)
#define QPC_PHASE(lane) { const int lane = qpc_opaque_lane();
#define QPC_LANE_SCOPE
#else
#define QPC_PHASE(lane) { const int lane = qpc_lane_q;
#define QPC_LANE_SCOPE const int qpc_lane_q = qpc_opaque_lane();
#define QPC_LANE_IN_SCOPE 1
#endif
#if defined(QPC_LANE_IN_SCOPE) && !defined(QPC_MACRO_LANE_TID)
#define QPC_MACRO_LANE qpc_lane_q
#else
#define QPC_MACRO_LANE ((int)(threadIdx.x & 63))
#endif
#ifndef QPC_PACK2
#define QPC_PHASE_N(lane, nact) QPC_PHASE(lane)
#define QPC_BRACE_END }
#else
#define QPC_BRACE_END } }
#endif
#define QPC_END_PHASE QPC_BRACE_END __syncthreads();
// SOFT end of a phase: the following block only depends on this one through the lane's own registers, DPP exchanges or LDS
// words the SAME lane wrote -- no cross-lane LDS traffic in either direction until the next hard end.  Nothing is emitted: the
// compiler may interleave the two blocks (loads of the next one hoisted over the arithmetic of this one).  The emulator keeps the
// race detector's phase number across a soft end, so a cross-lane dependency inside such a group is reported as a race.
// Opt-in (-DQPC_SOFT_PHASES): measured on MI355X with the fused sweeps of the search joined this way, 28.93 against 28.99 M
// solves/s with hard ends -- the compiler does not turn the freedom into overlap, so the default stays the conservative form.
#ifdef QPC_SOFT_PHASES
#define QPC_END_PHASE_SOFT QPC_BRACE_END
#else
#define QPC_END_PHASE_SOFT QPC_END_PHASE
#endif
// end of a block after which nothing is exchanged through LDS until the next hard end (registers and DPP only): never a fence
#define QPC_END_PHASE_REG QPC_BRACE_END
// hard boundary without a block (after a group of soft-joined blocks, before code that reads other lanes' LDS words)
#define QPC_PHASE_SYNC __syncthreads();
#include "qpc_math.h" // qpc_sincos, qpc_atan2: straight-line elementary functions (no scratch, few registers)
#ifdef QPC_F32
#define QPC_SINCOS(x, s, c) qpc_sincosf((float)(real)(x), (s).v, (c).v)
#define QPC_SQRT(x) real(__builtin_sqrtf((float)(real)(x)))
#define QPC_FABS(x) real(__builtin_fabsf((float)(real)(x)))
#define QPC_ATAN2(y, x) real(atan2f((float)(real)(y), (float)(real)(x)))
#else
#define QPC_SINCOS(x, s, c) qpc_sincos((double)(x), (s), (c))
#define QPC_SQRT(x) sqrt((double)(x))
#define QPC_FABS(x) fabs((double)(x))
#define QPC_ATAN2(y, x) qpc_atan2((double)(y), (double)(x))
#endif
// Pivot reciprocals on the serial chains of the small factorizations: hardware seed (v_rcp_f64 /
// v_rsq_f64, ~24 bits) + two Newton steps (~1 ulp) instead of the IEEE sqrt + divide sequences
// (about a third of the dependent instructions per pivot).
__device__ __forceinline__ double qpc_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
}
__device__ __forceinline__ double qpc_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = fma(y, 0.5 * fma(-x * y, y, 1.0), y);
  y = fma(y, 0.5 * fma(-x * y, y, 1.0), y);
  return y;
}
// one Newton step (~2^-45): the reciprocals of the working-set SEARCH (pivots of the 6 x 6 wrench-space solve, K rows) -- the
// search only has to pick the working set; the final solve refines against the true KKT residuals
__device__ __forceinline__ double qpc_rcp1(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
}
// fp32: v_rcp_f32 / v_rsq_f32 are good to 1 ulp; one Newton step where the fp64 path takes two
__device__ __forceinline__ float qpc_rcpf(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  return fmaf(y, fmaf(-x, y, 1.0f), y);
}
__device__ __forceinline__ float qpc_rsqrtf(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  return fmaf(y, 0.5f * fmaf(-x * y, y, 1.0f), y);
}
#ifdef QPC_F32
#define QPC_RCP_SEARCH(x) real(__builtin_amdgcn_rcpf((float)(real)(x)))
#define QPC_RCP(x) real(qpc_rcpf((float)(real)(x)))
#define QPC_RSQRT(x) real(qpc_rsqrtf((float)(real)(x)))
#else
#ifdef QPC_RCP_SEARCH_FULL // (measured: 28.80 M solves/s with two steps, 29.01 M with one; same working-set changes)
#define QPC_RCP_SEARCH(x) qpc_rcp((double)(x))
#else
#define QPC_RCP_SEARCH(x) qpc_rcp1((double)(x))
#endif
#define QPC_RCP(x) qpc_rcp((double)(x))
#define QPC_RSQRT(x) qpc_rsqrt((double)(x))
#endif
#define QPC_UNROLL_FULL _Pragma("unroll")
// lane-private storage that persists across phases (registers on the device)
#ifdef QPC_PACK2 // one copy per trip (the trip loop is unrolled: constant indices)
#define QPC_LANE_ARRAY(name, n) real name[QPC_TRIPS][n]
#define QPC_LA(name, lane) name[(lane) >> 5]
#define QPC_LANE_INTS(name, n) int name[QPC_TRIPS][n]
#define QPC_LI(name, lane) name[(lane) >> 5]
#else
#define QPC_LANE_ARRAY(name, n) real name[n]
#define QPC_LA(name, lane) name
#define QPC_LANE_INTS(name, n) int name[n]
#define QPC_LI(name, lane) name
#endif
#define QPC_CTZ64(x) (__ffsll((unsigned long long)(x)) - 1)
#define QPC_POPC64(x) __popcll((unsigned long long)(x))
#ifdef QPC_JIT
#define QPC_UNROLL_JIT _Pragma("unroll")
#define QPC_UNROLL _Pragma("unroll")
#else
#define QPC_UNROLL _Pragma("unroll 4")
#endif

namespace qpc {
// Wave-uniform reductions over the 64 lanes through DPP row shifts / row broadcasts: VALU only
// (3 short instructions per step for a double: two v_mov_dpp halves + one v_max/min_f64), against
// the 12 dependent ds_bpermute round trips of a __shfl_xor butterfly -- the reductions sit on the
// critical path of every active-set sweep.  All 64 lanes must be active (phase level only).
// DPP controls: row_shr:n = 0x110 + n, row_bcast:15 = 0x142 (row_mask 0xa), row_bcast:31 = 0x143
// (row_mask 0xc); lanes without a valid source keep `old` = their own value (neutral for max/min).
// The result is complete in lane 63 and broadcast with v_readlane.
#define QPC_DPP_STEPS(OP)                                                                          \
  OP(0x111, 0xf) OP(0x112, 0xf) OP(0x114, 0xf) OP(0x118, 0xf) OP(0x142, 0xa) OP(0x143, 0xc)
__device__ __forceinline__ double wave_max(double x) {
#define QPC_DPP_OP(ctrl, rmask)                                                                    \
  {                                                                                                \
    const int lo__ = __double2loint(x), hi__ = __double2hiint(x);                                  \
    const int tl__ = __builtin_amdgcn_update_dpp(lo__, lo__, ctrl, rmask, 0xf, false);             \
    const int th__ = __builtin_amdgcn_update_dpp(hi__, hi__, ctrl, rmask, 0xf, false);             \
    x = fmax(x, __hiloint2double(th__, tl__));                                                     \
  }
  QPC_DPP_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
// Maximum of NON-NEGATIVE doubles to 20 mantissa bits: the bit pattern of a non-negative double orders like an
// unsigned integer, so the maximum of the HIGH words (one v_mov_dpp + one v_max_u32 per step instead of two moves and a
// v_max_f64) is the true maximum rounded down by at most 2^-20 relative.  Every max reduction of the tick is a SCALE of
// a tolerance or a stopping test, for which that is exact enough; arg-min reductions stay exact (wave_min).
__device__ __forceinline__ double wave_max_scale(double x) {
  unsigned h = (unsigned)__double2hiint(x) & 0x7fffffffu; // callers pass magnitudes; a stray sign bit must not win
#define QPC_DPP_OP(ctrl, rmask)                                                                    \
  {                                                                                                \
    const unsigned t__ = (unsigned)__builtin_amdgcn_update_dpp((int)h, (int)h, ctrl, rmask, 0xf, false); \
    h = t__ > h ? t__ : h;                                                                         \
  }
  QPC_DPP_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __hiloint2double(__builtin_amdgcn_readlane((int)h, 63), 0);
}
__device__ __forceinline__ double wave_min(double x) {
#define QPC_DPP_OP(ctrl, rmask)                                                                    \
  {                                                                                                \
    const int lo__ = __double2loint(x), hi__ = __double2hiint(x);                                  \
    const int tl__ = __builtin_amdgcn_update_dpp(lo__, lo__, ctrl, rmask, 0xf, false);             \
    const int th__ = __builtin_amdgcn_update_dpp(hi__, hi__, ctrl, rmask, 0xf, false);             \
    x = fmin(x, __hiloint2double(th__, tl__));                                                     \
  }
  QPC_DPP_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
// fp32: one DPP move and one v_max / v_min per step (a float is one register)
__device__ __forceinline__ float wave_max(float x) {
#define QPC_DPP_OP(ctrl, rmask)                                                                    \
  {                                                                                                \
    const int b__ = __float_as_int(x);                                                             \
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(b__, b__, ctrl, rmask, 0xf, false)));  \
  }
  QPC_DPP_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_max_scale(float x) { return wave_max(__builtin_fabsf(x)); }
__device__ __forceinline__ float wave_min(float x) {
#define QPC_DPP_OP(ctrl, rmask)                                                                    \
  {                                                                                                \
    const int b__ = __float_as_int(x);                                                             \
    x = fminf(x, __int_as_float(__builtin_amdgcn_update_dpp(b__, b__, ctrl, rmask, 0xf, false)));  \
  }
  QPC_DPP_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ int wave_min_int(int x) {
#define QPC_DPP_OP(ctrl, rmask)                                                                    \
  {                                                                                                \
    const int t__ = __builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false);                    \
    x = t__ < x ? t__ : x;                                                                         \
  }
  QPC_DPP_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __builtin_amdgcn_readlane(x, 63);
}
#undef QPC_DPP_STEPS
} // namespace qpc

// reductions that straddle a phase boundary: *_PUT inside the phase (every lane, exactly once),
// *_GET after QPC_END_PHASE.  `name` is a double (and `iname` an int) declared before the phase.
#ifdef QPC_F32
#define QPC_RAW(x) ((float)(real)(x))
#else
#define QPC_RAW(x) (x)
#endif
#ifdef QPC_PACK2
// reductions over the 32 lanes of a half: DPP row shifts inside the rows of 16, row_bcast:15 into the second row of the half, the
// result read from lane 31 / 63.  A PUT runs once per trip: the first assigns, the second combines.
namespace qpc {
#define QPC_H_STEPS(OP) OP(0x111, 0xf) OP(0x112, 0xf) OP(0x114, 0xf) OP(0x118, 0xf) OP(0x142, 0xa)
__device__ __forceinline__ int half_pick(int v) {
  const int a = __builtin_amdgcn_readlane(v, 31), b = __builtin_amdgcn_readlane(v, 63);
  return (threadIdx.x & 32) ? b : a;
}
__device__ __forceinline__ float half_max(float x) {
#define QPC_DPP_OP(ctrl, rmask) { const int b__ = __float_as_int(x); x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(b__, b__, ctrl, rmask, 0xf, false))); }
  QPC_H_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __int_as_float(half_pick(__float_as_int(x)));
}
__device__ __forceinline__ float half_min(float x) {
#define QPC_DPP_OP(ctrl, rmask) { const int b__ = __float_as_int(x); x = fminf(x, __int_as_float(__builtin_amdgcn_update_dpp(b__, b__, ctrl, rmask, 0xf, false))); }
  QPC_H_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __int_as_float(half_pick(__float_as_int(x)));
}
__device__ __forceinline__ double half_max(double x) {
#define QPC_DPP_OP(ctrl, rmask) { const int lo__ = __double2loint(x), hi__ = __double2hiint(x); \
    x = fmax(x, __hiloint2double(__builtin_amdgcn_update_dpp(hi__, hi__, ctrl, rmask, 0xf, false), __builtin_amdgcn_update_dpp(lo__, lo__, ctrl, rmask, 0xf, false))); }
  QPC_H_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __hiloint2double(half_pick(__double2hiint(x)), half_pick(__double2loint(x)));
}
__device__ __forceinline__ double half_min(double x) {
#define QPC_DPP_OP(ctrl, rmask) { const int lo__ = __double2loint(x), hi__ = __double2hiint(x); \
    x = fmin(x, __hiloint2double(__builtin_amdgcn_update_dpp(hi__, hi__, ctrl, rmask, 0xf, false), __builtin_amdgcn_update_dpp(lo__, lo__, ctrl, rmask, 0xf, false))); }
  QPC_H_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __hiloint2double(half_pick(__double2hiint(x)), half_pick(__double2loint(x)));
}
__device__ __forceinline__ int half_min_int(int x) {
#define QPC_DPP_OP(ctrl, rmask) { const int t__ = __builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false); x = t__ < x ? t__ : x; }
  QPC_H_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return half_pick(x);
}
__device__ __forceinline__ float half_sum(float x) {
#define QPC_DPP_OP(ctrl, rmask) { x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false)); }
  QPC_H_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __int_as_float(half_pick(__float_as_int(x)));
}
__device__ __forceinline__ double half_sum(double x) {
#define QPC_DPP_OP(ctrl, rmask) { const int lo__ = __double2loint(x), hi__ = __double2hiint(x); \
    x += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi__, ctrl, rmask, 0xf, false), __builtin_amdgcn_update_dpp(0, lo__, ctrl, rmask, 0xf, false)); }
  QPC_H_STEPS(QPC_DPP_OP)
#undef QPC_DPP_OP
  return __hiloint2double(half_pick(__double2hiint(x)), half_pick(__double2loint(x)));
}
#undef QPC_H_STEPS
// the 32 predicate bits of this half (bit j = physical lane j of the half)
__device__ __forceinline__ unsigned long long half_ballot(bool pred) {
  return (unsigned long long)((__ballot(pred ? 1 : 0) >> (threadIdx.x & 32)) & 0xffffffffull);
}
} // namespace qpc
#ifdef QPC_F32
#define QPC_HABS(x) __builtin_fabsf(x)
#else
#define QPC_HABS(x) fabs(x)
#endif
#define QPC_RED_MAX_PUT(slot, name, val) { const real r__ = real(qpc::half_max(QPC_HABS(QPC_RAW(val)))); name = (qpc_trip == 0 || r__ > name) ? r__ : name; }
#define QPC_RED_MAX_GET(slot, name) (name)
#define QPC_RED_ARGMIN_PUT(slot, name, iname, val, idx) \
  do { const real v__ = (val); const real m__ = real(qpc::half_min(QPC_RAW(v__))); const int i__ = qpc::half_min_int(v__ == m__ ? (int)(idx) : 0x3fffffff); \
       if (qpc_trip == 0 || m__ < name || (m__ == name && i__ < iname)) { name = m__; iname = i__; } } while (0)
#define QPC_RED_ARGMIN_GET(slot, name, iname) ((void)0)
#define QPC_RED_ARGMIN_LANE_PUT(slot, name, iname, val, idx) \
  do { const real v__ = (val); const real m__ = real(qpc::half_min(QPC_RAW(v__))); const unsigned long long b__ = qpc::half_ballot(v__ == m__); \
       const int i__ = b__ ? (int)__builtin_ctzll(b__) + QPC_W * qpc_trip : 0x3fffffff; \
       if (qpc_trip == 0 || m__ < name) { name = m__; iname = i__; } } while (0)
#define QPC_BALLOT_PUT(slot, name, pred) { const unsigned long long b__ = qpc::half_ballot(pred); name = qpc_trip == 0 ? b__ : (name | (b__ << 32)); }
#define QPC_BALLOT_GET(slot, name) ((void)0)
#else // !QPC_PACK2
#define QPC_RED_MAX_PUT(slot, name, val) name = real(qpc::wave_max_scale(QPC_RAW(val)))
#define QPC_RED_MAX_GET(slot, name) (name)

#define QPC_RED_ARGMIN_PUT(slot, name, iname, val, idx) \
  do { real v__ = (val); name = real(qpc::wave_min(QPC_RAW(v__))); iname = qpc::wave_min_int(v__ == name ? (int)(idx) : 0x3fffffff); } while (0)
#define QPC_RED_ARGMIN_GET(slot, name, iname) ((void)0)
// ... when idx is the lane of every real candidate (other lanes pass a value that can only win if nobody has a candidate, in which
// case the caller discards the index): minimum, then the first lane that holds it -- no second reduction for the index
#define QPC_RED_ARGMIN_LANE_PUT(slot, name, iname, val, idx) \
  do { real v__ = (val); name = real(qpc::wave_min(QPC_RAW(v__))); const unsigned long long b__ = __ballot(v__ == name ? 1 : 0); \
       iname = b__ ? (int)__builtin_ctzll(b__) : 0x3fffffff; } while (0)
// 64-bit lane mask of a predicate (one scalar compare); same PUT / GET protocol
#define QPC_BALLOT_PUT(slot, name, pred) name = __ballot((pred) ? 1 : 0)
#define QPC_BALLOT_GET(slot, name) ((void)0)
#endif // QPC_PACK2

// OUT4[i] = XEXPR evaluated in lane 4 * (lane / 4) + i, i = 0..3: DPP quad_perm broadcasts (no LDS).  Must be executed
// by all 64 lanes (phase level, like the MFMA macro below).
__device__ __forceinline__ double qpc_quad_bcast(double x, int i) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  int l, h;
  switch (i) { // quad_perm:[i,i,i,i] = i * 0x55
    case 0: l = __builtin_amdgcn_update_dpp(lo, lo, 0x00, 0xf, 0xf, false); h = __builtin_amdgcn_update_dpp(hi, hi, 0x00, 0xf, 0xf, false); break;
    case 1: l = __builtin_amdgcn_update_dpp(lo, lo, 0x55, 0xf, 0xf, false); h = __builtin_amdgcn_update_dpp(hi, hi, 0x55, 0xf, 0xf, false); break;
    case 2: l = __builtin_amdgcn_update_dpp(lo, lo, 0xaa, 0xf, 0xf, false); h = __builtin_amdgcn_update_dpp(hi, hi, 0xaa, 0xf, 0xf, false); break;
    default: l = __builtin_amdgcn_update_dpp(lo, lo, 0xff, 0xf, 0xf, false); h = __builtin_amdgcn_update_dpp(hi, hi, 0xff, 0xf, 0xf, false); break;
  }
  return __hiloint2double(h, l);
}
__device__ __forceinline__ float qpc_quad_bcast(float x, int i) {
  const int b = __float_as_int(x);
  int l;
  switch (i) {
    case 0: l = __builtin_amdgcn_update_dpp(b, b, 0x00, 0xf, 0xf, false); break;
    case 1: l = __builtin_amdgcn_update_dpp(b, b, 0x55, 0xf, 0xf, false); break;
    case 2: l = __builtin_amdgcn_update_dpp(b, b, 0xaa, 0xf, 0xf, false); break;
    default: l = __builtin_amdgcn_update_dpp(b, b, 0xff, 0xf, 0xf, false); break;
  }
  return __int_as_float(l);
}
#define QPC_QUAD_GATHER(lane, OUT4, XEXPR)                                                        \
  {                                                                                              \
    const int lane = QPC_MACRO_LANE;                                                             \
    (void)lane;                                                                                  \
    real *out__ = (OUT4);                                                                        \
    const real x__ = (XEXPR);                                                                    \
    out__[0] = real(qpc_quad_bcast(QPC_RAW(x__), 0)); out__[1] = real(qpc_quad_bcast(QPC_RAW(x__), 1)); \
    out__[2] = real(qpc_quad_bcast(QPC_RAW(x__), 2)); out__[3] = real(qpc_quad_bcast(QPC_RAW(x__), 3)); \
  }

// OUT[i] = X[i] of lane 16 * (lane / 16) + K, i = 0 .. N - 1: DPP row_share broadcasts inside the rows of 16 lanes (no LDS).
// Phase level only (all 64 lanes), like QPC_QUAD_GATHER.
__device__ __forceinline__ double qpc_row_share(double x, int k) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  int l, h;
#define QPC_RS_CASE(K) case K: l = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + K, 0xf, 0xf, false); h = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + K, 0xf, 0xf, false); break;
  switch (k & 15) {
    QPC_RS_CASE(0) QPC_RS_CASE(1) QPC_RS_CASE(2) QPC_RS_CASE(3) QPC_RS_CASE(4) QPC_RS_CASE(5) QPC_RS_CASE(6) QPC_RS_CASE(7)
    QPC_RS_CASE(8) QPC_RS_CASE(9) QPC_RS_CASE(10) QPC_RS_CASE(11) QPC_RS_CASE(12) QPC_RS_CASE(13) QPC_RS_CASE(14)
    default: l = __builtin_amdgcn_update_dpp(lo, lo, 0x15f, 0xf, 0xf, false); h = __builtin_amdgcn_update_dpp(hi, hi, 0x15f, 0xf, 0xf, false); break;
  }
#undef QPC_RS_CASE
  return __hiloint2double(h, l);
}
__device__ __forceinline__ float qpc_row_share(float x, int k) {
  const int b = __float_as_int(x);
  int l;
#define QPC_RS_CASE(K) case K: l = __builtin_amdgcn_update_dpp(b, b, 0x150 + K, 0xf, 0xf, false); break;
  switch (k & 15) {
    QPC_RS_CASE(0) QPC_RS_CASE(1) QPC_RS_CASE(2) QPC_RS_CASE(3) QPC_RS_CASE(4) QPC_RS_CASE(5) QPC_RS_CASE(6) QPC_RS_CASE(7)
    QPC_RS_CASE(8) QPC_RS_CASE(9) QPC_RS_CASE(10) QPC_RS_CASE(11) QPC_RS_CASE(12) QPC_RS_CASE(13) QPC_RS_CASE(14)
    default: l = __builtin_amdgcn_update_dpp(b, b, 0x15f, 0xf, 0xf, false); break;
  }
#undef QPC_RS_CASE
  return __int_as_float(l);
}
#define QPC_ROW_SHARE(lane, OUT, N, X, K)                                                         \
  {                                                                                              \
    const int lane = QPC_MACRO_LANE;                                                             \
    (void)lane;                                                                                  \
    real *out__ = (OUT);                                                                         \
    const real *x__ = (X);                                                                       \
    for (int i__ = 0; i__ < (N); i__++) out__[i__] = real(qpc_row_share(QPC_RAW(x__[i__]), (K))); \
  }

// inclusive prefix sum over the lanes of the wave (lane l: x_0 + ... + x_l): Hillis-Steele inside the DPP rows of 16 (row_shr with
// zero fill), then the totals of the rows before (row_bcast:15 / :31) -- 6 steps of two moves and an add per double, no LDS
__device__ __forceinline__ double qpc_wave_prefix_sum(double x) {
#define QPC_PS_STEP(ctrl, rmask)                                                           \
  {                                                                                        \
    const int lo__ = __double2loint(x), hi__ = __double2hiint(x);                          \
    const int tl__ = __builtin_amdgcn_update_dpp(0, lo__, ctrl, rmask, 0xf, false);        \
    const int th__ = __builtin_amdgcn_update_dpp(0, hi__, ctrl, rmask, 0xf, false);        \
    x += __hiloint2double(th__, tl__);                                                     \
  }
  QPC_PS_STEP(0x111, 0xf) QPC_PS_STEP(0x112, 0xf) QPC_PS_STEP(0x114, 0xf) QPC_PS_STEP(0x118, 0xf)
  QPC_PS_STEP(0x142, 0xa) QPC_PS_STEP(0x143, 0xc)
#undef QPC_PS_STEP
  return x;
}
__device__ __forceinline__ float qpc_wave_prefix_sum(float x) {
#define QPC_PS_STEP(ctrl, rmask) { x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false)); }
  QPC_PS_STEP(0x111, 0xf) QPC_PS_STEP(0x112, 0xf) QPC_PS_STEP(0x114, 0xf) QPC_PS_STEP(0x118, 0xf)
  QPC_PS_STEP(0x142, 0xa) QPC_PS_STEP(0x143, 0xc)
#undef QPC_PS_STEP
  return x;
}
__device__ __forceinline__ float qpc_wave_sum(float x) {
  x = qpc_wave_prefix_sum(x);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// sum over the 64 lanes (the last lane's prefix sum), same PUT / GET protocol as the max reductions
__device__ __forceinline__ double qpc_wave_sum(double x) {
  x = qpc_wave_prefix_sum(x);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
#ifdef QPC_PACK2
#define QPC_RED_SUM_PUT(slot, name, val) { const real r__ = real(qpc::half_sum(QPC_RAW(val))); name = qpc_trip == 0 ? r__ : name + r__; }
#else
#define QPC_RED_SUM_PUT(slot, name, val) name = real(qpc_wave_sum(QPC_RAW(val)))
#endif
#define QPC_RED_SUM_GET(slot, name) (name)
// phase level (all 64 lanes): ARR[i] of every lane becomes the sum of ARR[i] over the lanes up to and including it
#define QPC_WAVE_INCL_SCAN(lane, ARR, N)                                                          \
  {                                                                                              \
    const int lane = QPC_MACRO_LANE;                                                             \
    (void)lane;                                                                                  \
    real *a__ = (ARR);                                                                           \
    for (int i__ = 0; i__ < (N); i__++) a__[i__] = real(qpc_wave_prefix_sum(QPC_RAW(a__[i__]))); \
  }

// matrix core: acc(16x16 tile, 4 values per lane) += A(16x4) * B(4x16) with v_mfma_f64_16x16x4_f64 (fp32 tick: v_mfma_f32_16x16x4_f32).
//   lane l supplies A[i = l&15][k = l>>4] (AEXPR) and B[k = l>>4][j = l&15] (BEXPR);
//   acc[r] is C[row = QPC_MFMA_CROW(l, r)][col = l&15]: (l>>4) + 4 r for the f64 instruction (its own C/D map), 4 (l>>4) + r for the
//   f32 one (the standard 16x16 map; cdna_hip_programming.md 3)
// Must be executed by all 64 lanes (EXEC all ones): only use it at phase level, never under a
// lane-dependent branch.
#ifdef QPC_F32
typedef float qpc_v4r __attribute__((ext_vector_type(4)));
#define QPC_MFMA_BUILTIN __builtin_amdgcn_mfma_f32_16x16x4f32
#define QPC_MFMA_CROW(lane, reg) (4 * ((lane) >> 4) + (reg))
#else
typedef double qpc_v4r __attribute__((ext_vector_type(4)));
#define QPC_MFMA_BUILTIN __builtin_amdgcn_mfma_f64_16x16x4f64
#define QPC_MFMA_CROW(lane, reg) (((lane) >> 4) + 4 * (reg))
#endif
#define QPC_MFMA_16x16x4(lane, ACC, AEXPR, BEXPR)                                               \
  {                                                                                              \
    const int lane = QPC_MACRO_LANE;                                                             \
    real *acc__ = (ACC);                                                                         \
    const real a__ = (AEXPR), b__ = (BEXPR);                                                     \
    qpc_v4r c__ = {QPC_RAW(acc__[0]), QPC_RAW(acc__[1]), QPC_RAW(acc__[2]), QPC_RAW(acc__[3])};  \
    c__ = QPC_MFMA_BUILTIN(QPC_RAW(a__), QPC_RAW(b__), c__, 0, 0, 0);                            \
    acc__[0] = real(c__[0]); acc__[1] = real(c__[1]); acc__[2] = real(c__[2]); acc__[3] = real(c__[3]); \
  }
