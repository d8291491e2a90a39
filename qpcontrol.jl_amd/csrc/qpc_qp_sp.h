// qpc_qp_sp.h -- L0 for a PREPARED problem with a PLAN (qpc_qp_sym.h): the sparse form of the workgroup solver of qpc_qp_wg.h, what
// serves the reference's own plug point -- the optimizer slot of `MomentumBasedController{N}(mechanism, optimizer)`,
// /root/reference/src/lowlevel/momentum.jl:15-16,27,58 -- at B = 1 and in batches.
//
// Same method (range-space active set on G = A H^-1 A': ONE elimination of the equality rows leaves their factor, the Schur
// complement C on the inequality rows and the right-hand sides; primal-dual sweeps and Goldfarb-Idnani in the space of the
// inequality rows; refinement on the true KKT residuals), but nothing in it is dense any more except C:
//   * H = P + eps I is block diagonal (blocks of <= 8 variables: the e'We of matrix-weighted tasks, momentum.jl:113-117), inverted
//     block by block, one wavefront per block;
//   * G is built entry by entry of its STRUCTURE, into a store that holds only the structure of its factor: the FRONTS of the
//     elimination tree (small dense rectangles) and C.  For Atlas (143 x 178): 6.2 k doubles instead of 16.1 k;
//   * the elimination runs level by level of the tree: the fronts of a level are factored concurrently, one wavefront each, without
//     a workgroup barrier inside a front; their updates of later fronts and of C are 16 x 16 tile jobs on the matrix core
//     (v_mfma_f64_16x16x4_f64), every target tile summed by one wavefront over all its source columns in a fixed order;
//   * triangular solves with the equality rows' factor go front by front along the same levels;
//   * the working-set iterations are ONE wavefront's work (the other wavefronts wait at a barrier): factor of C_WW, multipliers,
//     A x, ratio tests and the choice of the next row without a single workgroup barrier.
// A QP whose set of equality rows differs from the plan's returns QPC_WG_FALLBACK (the dense workgroup form takes it).
//
// Written like qpc_qp_wg.h as loops over the workgroup / a wavefront's lanes so that the same source runs serially inside the test
// emulation (one "thread"); the matrix-core tile loop has a scalar twin there (same sums, another order).
#pragma once
#include "../../include/qpc.h"
#include "qpc_qp_spview.h"

#ifndef QPC_FN
#error "include qpc_device.h (or the emulator's definitions) first"
#endif
#ifndef QPC_WG_FALLBACK
#define QPC_WG_FALLBACK (-1)
#endif

#ifndef QPC_SP_THREADS
#define QPC_SP_THREADS 512 // threads of the workgroup (device): two wavefronts per SIMD
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define QPC_SP_NT QPC_SP_THREADS
#define QPC_SP_NWAVE (QPC_SP_THREADS / 64)
#define QPC_SP_NLANE 64
#define QPC_SP_TID ((int)threadIdx.x)
#define QPC_SP_SYNC __syncthreads();
#define QPC_SP_WAVE __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))
// inside one wavefront LDS operations execute in program order: what a lane has stored, every lane reads afterwards -- the compiler
// only has to be kept from moving the accesses across this point
#define QPC_SP_WSYNC do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0);
#else
#define QPC_SP_NT 1
#define QPC_SP_NWAVE 1
#define QPC_SP_NLANE 1
#define QPC_SP_TID 0
#define QPC_SP_SYNC
#define QPC_SP_WAVE 0
#define QPC_SP_WSYNC
#endif
#define QPC_SP_LANE (QPC_SP_TID % QPC_SP_NLANE)
#define QPC_SP_FOR(i, n) for (int i = QPC_SP_TID; i < (n); i += QPC_SP_NT)
#define QPC_SP_FOR_WAVE(i, a, b) for (int i = (a) + QPC_SP_WAVE; i < (b); i += QPC_SP_NWAVE)
#define QPC_SP_FOR_LANE(i, n) for (int i = QPC_SP_LANE; i < (n); i += QPC_SP_NLANE)
#define QPC_SP_T0 if (QPC_SP_TID == 0)
#define QPC_SP_L0 if (QPC_SP_LANE == 0)
#if defined(QPC_SP_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define QPC_SP_STAMP(W, k) do { if (threadIdx.x == 0 && (k) < 32) (W).stamps[k] = wall_clock64(); } while (0)
#else
#define QPC_SP_STAMP(W, k) do { } while (0)
#endif

namespace qpc {

#if defined(QPC_SP_PROF) && defined(__HIPCC__)
__device__ unsigned long long qpc_sp_acc[16]; // (profiling build: time inside the pieces of the working-set iterations, QP 0's workgroup)
#endif
#if defined(QPC_SP_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define QPC_SP_ACC_T0() unsigned long long acc_t0__ = wall_clock64()
#define QPC_SP_ACC(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { const unsigned long long n__ = wall_clock64(); qpc_sp_acc[k] += n__ - acc_t0__; acc_t0__ = n__; } } while (0)
#define QPC_SP_ACC_ADD(k, v) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) qpc_sp_acc[k] += (v); } while (0)
#else
#define QPC_SP_ACC_T0() do { } while (0)
#define QPC_SP_ACC(k) do { } while (0)
#define QPC_SP_ACC_ADD(k, v) do { } while (0)
#endif

QPC_FN int sp_cs(int c, int N1) { return c * N1 - (c * (c - 1)) / 2; }      // packed lower triangle, column-major: start of column c
QPC_FN int sp_sidx(int i, int c, int N1) { return sp_cs(c, N1) + i - c; }    // i >= c
QPC_FN int sp_sym(int a, int b, int N1) { return a >= b ? sp_sidx(a, b, N1) : sp_sidx(b, a, N1); }

// reductions over one wavefront (every lane gets the result; all 64 lanes active): DPP row shifts (qpc_device.h) -- a __shfl_xor butterfly
// is twelve dependent LDS-crossbar round trips for a double
QPC_FN double sp_wave_max(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return qpc::wave_max(v);
#else
  return v;
#endif
}
QPC_FN bool sp_wave_any(bool f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ballot(f) != 0ull;
#else
  return f;
#endif
}
// the candidate (idx >= 0) with the largest v, ties to the smallest idx; idx < 0: no candidate.  Every lane gets (v, idx).
QPC_FN void sp_wave_argmax(double &v, int &idx) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double m = qpc::wave_max(idx >= 0 ? v : -1e308);
  const int i = qpc::wave_min_int((idx >= 0 && v == m) ? idx : 0x7fffffff);
  v = m; idx = i == 0x7fffffff ? -1 : i;
#endif
}
QPC_FN void sp_wave_argmin(double &v, int &idx) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double m = qpc::wave_min(idx >= 0 ? v : 1e308);
  const int i = qpc::wave_min_int((idx >= 0 && v == m) ? idx : 0x7fffffff);
  v = m; idx = i == 0x7fffffff ? -1 : i;
#endif
}
// max over the workgroup of N per-thread values at once (every thread gets them); red: N * QPC_SP_NWAVE doubles of LDS.  A NaN counts
// as +infinity (the callers test for "not finite").  Two barriers whatever N.
template <int N>
QPC_FN void sp_reduce_max(double (&v)[N], double *red) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int k = 0; k < N; k++) {
    const double x = v[k] == v[k] ? v[k] : __builtin_inf();
    v[k] = qpc::wave_max(x);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; k++) red[(threadIdx.x >> 6) * N + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; k++) {
    double r = red[k];
    for (int w = 1; w < QPC_SP_NWAVE; w++) { const double o = red[w * N + k]; r = o > r ? o : r; }
    v[k] = r;
  }
  __syncthreads();
#else
  (void)red;
  for (int k = 0; k < N; k++) v[k] = v[k] == v[k] ? v[k] : __builtin_inf();
#endif
}

struct QpSp {
  const int *pl;  // the plan's LDS part
  const int *plg; // the whole plan in global memory
  int n, m, ne, mi, N1, M1, Cb, nfront, nlevel, kw;
  const double *Pv, *q, *Av, *l, *u, *y0;
  // LDS, doubles
  double *sc, *red;
  double *V;                               // the value store: fronts, then C
  double *dinv, *gdiag, *t, *yp;           // by position: ne, m + 1, m + 1, m + 1
  double *zs, *part, *part2;               // ne: the scaled solution of a forward solve; the sparse products' segment sums (two products in flight)
  double *hv, *hm;                         // values of H^-1 and of P, block by block
  double *xu, *x, *r1, *tn;                // n
  double *lo, *hi, *wsf, *tm, *ax, *a0;    // mi
  double *yw, *dinvW, *dW, *tW;            // kw + 1
  int *idxW;
  double *LW;                              // the working set's factor, (k + 1) x (k + 1) packed
  double *sval;                            // A's values in column order
  unsigned long long *stamps;
};
enum { SPS_FLAG = 0, SPS_STATUS, SPS_ITERS, SPS_NW, SPS_CONV, SPS_COUNT };

QPC_FN double sp_C(const QpSp &W, int p, int q) { return W.V[W.Cb + sp_sym(p, q, W.M1)]; } // C by inequality position (mi: the right-hand-side row)

// what the kernel asks of LDS for a plan, in doubles (kw: inequality rows a working set may hold)
#ifdef __HIPCC__
__host__ __device__
#endif
// what the host may ask for as DYNAMIC LDS (doubles): the CU's 160 KB less the kernel's static LDS (qpc_qp_sp_kernel's ticket word) and a margin
#define QPC_SP_LDS_CAP_DOUBLES (160 * 1024 / 8 - 8)
inline int qp_sp_lds_doubles(const int *plan, int kw) {
  const int n = plan[SP_N], m = plan[SP_M], ne = plan[SP_NE], mi = plan[SP_MI];
  return (plan[SP_LDS_INTS] + 1) / 2 + 16 + 32 + plan[SP_NVAL] + 2 * ne + 2 * plan[SP_MAXSEG] + 3 * (m + 1) + 2 * plan[SP_NHV] + 4 * n + 6 * mi + 4 * (kw + 1) + (kw + 2) / 2 +
         (kw + 1) * (kw + 2) / 2 + plan[SP_NNZA] + 2;
}

// ---- LDL' without square roots of the leading nc columns of a tall panel, by ONE wavefront (no workgroup barrier).  Entry (i, c) at
//      S[cb + c * cs1 - (c (c - 1) / 2) * cs2 + i]: a front's rectangle (cs1 = its rows, cs2 = 0) or a packed triangle (cs1 = its
//      dimension - 1 ... see sp_pcol).  The columns keep L D (unscaled), dinv[j] = 1 / d_j, or 0 for a pivot below tau * ref[j] (a
//      dependent row drops out of everything).  Eight columns at a time, as qpc_qp_wg.h does it for the whole matrix: every lane factors
//      the panel's 8 x 8 diagonal block itself, in registers (36 broadcast loads, a chain of pivots); a row below the block is a little
//      triangular solve against multipliers the lane already holds; then the panel's rank-8 update of the later columns, a lane per
//      row, the row's eight panel values loaded once.  What bounds it is LDS latency: every loop has its loads first.
struct SpCols { int cb, cs1, cs2; }; // column c starts (row 0) at cb + c * cs1 - cs2 * (c (c - 1) / 2)
QPC_FN int sp_pcol(const SpCols &C, int c) { return C.cb + c * C.cs1 - C.cs2 * ((c * (c - 1)) / 2); }
QPC_FN SpCols sp_cols_rect(int base, int r) { SpCols C; C.cb = base; C.cs1 = r; C.cs2 = 0; return C; }
QPC_FN SpCols sp_cols_packed(int K1) { SpCols C; C.cb = 0; C.cs1 = K1 - 1; C.cs2 = 1; return C; } // sp_cs(c, K1) - c = c (K1 - 1) - c (c - 1) / 2
#define QPC_SP_NB 8
QPC_FN void sp_ldl_panels(double *S, const SpCols C, int nr, int nc, const double *ref, double tau, double *dinv) {
  for (int jp = 0; jp < nc; jp += QPC_SP_NB) {
    const int nb = nc - jp < QPC_SP_NB ? nc - jp : QPC_SP_NB;
    int pc[QPC_SP_NB];
    double blk[QPC_SP_NB][QPC_SP_NB], mul[QPC_SP_NB][QPC_SP_NB], dv[QPC_SP_NB], rfv[QPC_SP_NB];
#pragma unroll
    for (int c = 0; c < QPC_SP_NB; c++) {
      pc[c] = sp_pcol(C, c < nb ? jp + c : jp); // S[i][jp + c] = S[pc[c] + i]
#pragma unroll
      for (int r = c; r < QPC_SP_NB; r++) {
        const double v = S[pc[c] + jp + (r < nb ? r : 0)]; // (always a load from a valid address, the choice afterwards)
        blk[r][c] = (r < nb && c < nb) ? v : (r == c ? 1.0 : 0.0);
      }
      rfv[c] = ref[c < nb ? jp + c : jp];
    }
    QPC_SP_WSYNC // (every lane has the block: lane 0 may overwrite it)
#pragma unroll
    for (int j = 0; j < QPC_SP_NB; j++) {
      const double d = blk[j][j], rf = rfv[j];
      const bool ok = j < nb && d > tau * (rf > 1e-300 ? rf : 1e-300);
      dv[j] = ok ? 1.0 / d : 0.0;
#pragma unroll
      for (int c = j + 1; c < QPC_SP_NB; c++) {
        const double lc = blk[c][j] * dv[j];
        mul[c][j] = lc;
#pragma unroll
        for (int r = c; r < QPC_SP_NB; r++) blk[r][c] -= blk[r][j] * lc;
      }
    }
    QPC_SP_L0 {
#pragma unroll
      for (int c = 0; c < QPC_SP_NB; c++) {
        if (c < nb) {
          dinv[jp + c] = dv[c];
#pragma unroll
          for (int r = c; r < QPC_SP_NB; r++) if (r < nb) S[pc[c] + jp + r] = blk[r][c];
        }
      }
    }
    // the rows below the block: u L1' = a.  Then, while the row's panel values are in registers, the panel's update of the row's
    // entries in the later columns of this elimination (c2 = jp + nb .. nc - 1, c2 <= i)
    const int c2a = jp + nb;
    for (int i = jp + nb + QPC_SP_LANE; i < nr; i += QPC_SP_NLANE) {
      double a[QPC_SP_NB];
#pragma unroll
      for (int c = 0; c < QPC_SP_NB; c++) a[c] = S[pc[c] + i]; // (a column beyond nb: pc of the first one, value unused)
#pragma unroll
      for (int j = 0; j < QPC_SP_NB; j++) {
#pragma unroll
        for (int c = j + 1; c < QPC_SP_NB; c++) a[c] -= a[j] * mul[c][j];
      }
#pragma unroll
      for (int c = 1; c < QPC_SP_NB; c++) if (c < nb) S[pc[c] + i] = a[c];
    }
    QPC_SP_WSYNC
    if (c2a < nc) {
      for (int i = c2a + QPC_SP_LANE; i < nr; i += QPC_SP_NLANE) {
        double u[QPC_SP_NB];
#pragma unroll
        for (int c = 0; c < QPC_SP_NB; c++) u[c] = S[pc[c] + i] * dv[c]; // (dv of a padding column: 0)
        const int cend = i < nc ? i : nc - 1;
        int c2 = c2a;
        for (; c2 + 1 <= cend; c2 += 2) { // two target columns at a time: all their loads first
          const int t0 = sp_pcol(C, c2) + i, t1 = sp_pcol(C, c2 + 1) + i;
          double m0[QPC_SP_NB], m1[QPC_SP_NB];
#pragma unroll
          for (int c = 0; c < QPC_SP_NB; c++) { m0[c] = S[pc[c] + c2]; m1[c] = S[pc[c] + c2 + 1]; }
          double v0 = S[t0], v1 = S[t1];
#pragma unroll
          for (int c = 0; c < QPC_SP_NB; c++) { v0 -= u[c] * m0[c]; v1 -= u[c] * m1[c]; }
          S[t0] = v0; S[t1] = v1;
        }
        if (c2 <= cend) {
          const int t0 = sp_pcol(C, c2) + i;
          double m0[QPC_SP_NB];
#pragma unroll
          for (int c = 0; c < QPC_SP_NB; c++) m0[c] = S[pc[c] + c2];
          double v0 = S[t0];
#pragma unroll
          for (int c = 0; c < QPC_SP_NB; c++) v0 -= u[c] * m0[c];
          S[t0] = v0;
        }
      }
      QPC_SP_WSYNC
    }
  }
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_SP_NO_REG)
// ---- the same elimination with the panel in REGISTERS (device): lane = row, column c of the row in a[c] (static indices), a pivot
//      row's multipliers broadcast with v_readlane -- no LDS round trip inside the elimination: what bounds the LDS forms is the
//      latency of their dependent loads (~60 ns each), several per pivot.  Straight-line code for KR = 8, 16, 24, 32 columns, NO
//      conditions inside: the columns beyond the real ones are zero, their pivots fail the test and drop out like dependent rows do
//      (a skipped block costs a taken branch, measured at ~40 ns -- more than the block).
// (the routines below are not inlined -- a register allocation of their own --, so their pointers say that they point into LDS: a plain
//  pointer argument is a generic one, and its loads are flat_load instructions with the latency of global memory)
#define QPC_LDS __attribute__((address_space(3)))
QPC_FN double sp_readlane(double v, int lane) { // (lane: a constant)
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// one pass over <= 64 rows.  first: the lanes 0 .. KR-1 hold the diagonal block (lane c = its row c): the pivots are taken,
// mul[j] <- a[j] / d_j (lane c: L_cj), dv (lane j) = 1 / d_j or 0; refv (lane j): the pivot's reference.  Later passes (further rows
// of the same panel) reuse mul.
template <int KR>
QPC_FN void sp_reg_elim(double (&a)[KR], double (&mul)[KR], bool first, double refv, double tau, double &dv) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < KR; j++) {
    if (first) {
      const double d = sp_readlane(a[j], j), rf = sp_readlane(refv, j);
      const bool ok = d > tau * (rf > 1e-300 ? rf : 1e-300);
      const double di = ok ? qpc_rcp(d) : 0.0;
      mul[j] = a[j] * di;
      dv = lane == j ? di : dv;
    }
#pragma unroll
    for (int c = j + 1; c < KR; c++) a[c] -= a[j] * sp_readlane(mul[j], c);
  }
}
// (not inlined: a register allocation of its own -- inlined into the kernel the four sizes cost it 1 268 spilled scalar registers)
template <int KR>
__device__ __attribute__((noinline)) void sp_front_factor_reg(QPC_LDS double *V, int base, int r, int w, QPC_LDS const double *ref, double tau, QPC_LDS double *dinv) {
  const int lane = threadIdx.x & 63;
  double a[KR], mul[KR], dv = 0.0;
  const double refv = lane < w ? ref[lane] : 1.0;
  for (int p0 = 0; p0 < r; p0 += 64) {
    const int i = p0 + lane, ic = i < r ? i : r - 1;
#pragma unroll
    for (int c = 0; c < KR; c++) { const double v = V[base + (c < w ? c : 0) * r + ic]; a[c] = (c < w && i < r) ? v : 0.0; }
    sp_reg_elim<KR>(a, mul, p0 == 0, refv, tau, dv);
#pragma unroll
    for (int c = 1; c < KR; c++) { if (c < w && i < r && i >= c) V[base + c * r + i] = a[c]; }
    if (p0 == 0 && lane < w) dinv[lane] = dv;
  }
}
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_SP_NO_REG)
// the small triangular solves of a front's own block (w <= KR columns, a rectangle at base with r rows) in registers: lane c holds
// row c (forward) or column c (back) of the block, a step is two broadcasts and a multiply-add instead of the LDS round trips of the
// four-column form.  t: the w values at the front's columns, solved in place.
template <int KR>
QPC_FN void sp_front_forward_reg(const double *V, int base, int r, int w, const double *dinv, double *t) {
  const int lane = threadIdx.x & 63, c = lane < w ? lane : 0;
  double row[KR];
#pragma unroll
  for (int j = 0; j < KR - 1; j++) { const double v = V[base + (j < w ? j : 0) * r + c]; row[j] = (j < c && lane < w) ? v : 0.0; } // L_cj d_j (unscaled)
  double tv = lane < w ? t[c] : 0.0;
  const double dv = lane < w ? dinv[c] : 0.0;
#pragma unroll
  for (int j = 0; j < KR - 1; j++) tv -= row[j] * (sp_readlane(tv, j) * sp_readlane(dv, j)); // (rows <= j: row[j] = 0; a column beyond w: dv = 0)
  if (lane < w) t[c] = tv;
}
template <int KR>
QPC_FN void sp_front_back_reg(const double *V, int base, int r, int w, const double *dinv, double *t) {
  const int lane = threadIdx.x & 63, c = lane < w ? lane : 0;
  double col[KR];
#pragma unroll
  for (int j = 1; j < KR; j++) { const double v = V[base + c * r + (j < w ? j : 0)]; col[j] = (j > c && j < w && lane < w) ? v : 0.0; } // S_jc
  double tv = lane < w ? t[c] : 0.0, yv = 0.0;
  const double dv = lane < w ? dinv[c] : 0.0;
#pragma unroll
  for (int j = KR - 1; j >= 0; j--) {
    const double yj = sp_readlane(tv, j) * sp_readlane(dv, j);
    yv = lane == j ? yj : yv;
    if (j > 0) tv -= col[j] * yj;
  }
  if (lane < w) t[c] = yv;
}
#endif
// t <- L^-1 t on the leading nc columns (L unit lower: L_ij = S_ij dinv_j), four columns per step: every lane solves the 4 x 4
// diagonal block itself (ten broadcast loads), then takes the four columns out of its rows below.  One wavefront.
QPC_FN void sp_tri_forward(const double *S, const SpCols C, int nc, const double *dinv, double *t) {
  for (int j = 0; j < nc; j += 4) {
    const int nb = nc - j < 4 ? nc - j : 4;
    const int j1 = nb > 1 ? j + 1 : j, j2 = nb > 2 ? j + 2 : j, j3 = nb > 3 ? j + 3 : j; // (clamped: a short last block re-reads column j)
    const int c0 = sp_pcol(C, j), c1 = sp_pcol(C, j1), c2 = sp_pcol(C, j2), c3 = sp_pcol(C, j3);
    const double t0 = t[j], t1 = t[j1], t2 = t[j2], t3 = t[j3];
    const double d0 = dinv[j], d1 = nb > 1 ? dinv[j1] : 0.0, d2 = nb > 2 ? dinv[j2] : 0.0, d3 = nb > 3 ? dinv[j3] : 0.0;
    const double l10 = S[c0 + j1], l20 = S[c0 + j2], l30 = S[c0 + j3], l21 = S[c1 + j2], l31 = S[c1 + j3], l32 = S[c2 + j3];
    const double f0 = t0 * d0;
    double z1 = t1; z1 -= l10 * f0;
    const double f1 = z1 * d1;
    double z2 = t2; z2 -= l20 * f0; z2 -= l21 * f1;
    const double f2 = z2 * d2;
    double z3 = t3; z3 -= l30 * f0; z3 -= l31 * f1; z3 -= l32 * f2;
    const double f3 = z3 * d3;
    QPC_SP_WSYNC // (everybody has t[j .. j+3])
    for (int i = j + nb + QPC_SP_LANE; i < nc; i += QPC_SP_NLANE) {
      const double a0 = S[c0 + i], a1 = S[c1 + i], a2 = S[c2 + i], a3 = S[c3 + i];
      double v = t[i];
      v -= a0 * f0; v -= a1 * f1; v -= a2 * f2; v -= a3 * f3; // (f of a clamped column is 0)
      t[i] = v;
    }
    QPC_SP_L0 { if (nb > 1) t[j + 1] = z1; if (nb > 2) t[j + 2] = z2; if (nb > 3) t[j + 3] = z3; }
    QPC_SP_WSYNC
  }
}
// out <- L^-T D^-1 t on the leading nc columns; t is used up (out may be t).  Four rows per step.  One wavefront.
QPC_FN void sp_tri_back(const double *S, const SpCols C, int nc, const double *dinv, double *t, double *out) {
  for (int i0 = nc - 1; i0 >= 0; i0 -= 4) {
    const int nb = i0 + 1 < 4 ? i0 + 1 : 4;
    const int i1 = nb > 1 ? i0 - 1 : i0, i2 = nb > 2 ? i0 - 2 : i0, i3 = nb > 3 ? i0 - 3 : i0; // (clamped)
    const double t0 = t[i0], t1 = t[i1], t2 = t[i2], t3 = t[i3];
    const double d0 = dinv[i0], d1 = dinv[i1], d2 = dinv[i2], d3 = dinv[i3];
    const int c1 = sp_pcol(C, i1), c2 = sp_pcol(C, i2), c3 = sp_pcol(C, i3);
    const double u01 = S[c1 + i0], u02 = S[c2 + i0], u03 = S[c3 + i0], u12 = S[c2 + i1], u13 = S[c3 + i1], u23 = S[c3 + i2];
    const double y0 = t0 * d0;
    double v1 = t1; v1 -= u01 * y0;
    const double y1 = nb > 1 ? v1 * d1 : 0.0;
    double v2 = t2; v2 -= u02 * y0; v2 -= u12 * y1;
    const double y2 = nb > 2 ? v2 * d2 : 0.0;
    double v3 = t3; v3 -= u03 * y0; v3 -= u13 * y1; v3 -= u23 * y2;
    const double y3 = nb > 3 ? v3 * d3 : 0.0;
    QPC_SP_WSYNC // (everybody has t[i0 - 3 .. i0])
    for (int j = QPC_SP_LANE; j < i0 + 1 - nb; j += QPC_SP_NLANE) {
      const int cj = sp_pcol(C, j);
      const double a0 = S[cj + i0], a1 = S[cj + i1], a2 = S[cj + i2], a3 = S[cj + i3];
      double v = t[j];
      v -= a0 * y0; v -= a1 * y1; v -= a2 * y2; v -= a3 * y3; // (y of a clamped row is 0)
      t[j] = v;
    }
    QPC_SP_L0 { out[i0] = y0; if (nb > 1) out[i1] = y1; if (nb > 2) out[i2] = y2; if (nb > 3) out[i3] = y3; }
    QPC_SP_WSYNC
  }
}

// a front: its w columns eliminated, the rows below carried along (one wavefront)
QPC_FN void sp_front_factor(double *V, int base, int r, int w, const double *ref, double tau, double *dinv) {
  if (w == 1) { // a single column: its pivot, nothing else (the rows below keep L D)
    const double d = V[base], rf = ref[0];
    const bool ok = d > tau * (rf > 1e-300 ? rf : 1e-300);
    QPC_SP_L0 { dinv[0] = ok ? 1.0 / d : 0.0; }
    return;
  }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_SP_NO_REG)
  if (w <= 8) { sp_front_factor_reg<8>((QPC_LDS double *)V, base, r, w, (QPC_LDS const double *)ref, tau, (QPC_LDS double *)dinv); return; }
  if (w <= 12) { sp_front_factor_reg<12>((QPC_LDS double *)V, base, r, w, (QPC_LDS const double *)ref, tau, (QPC_LDS double *)dinv); return; }
  if (w <= 16) { sp_front_factor_reg<16>((QPC_LDS double *)V, base, r, w, (QPC_LDS const double *)ref, tau, (QPC_LDS double *)dinv); return; }
  if (w <= 20) { sp_front_factor_reg<20>((QPC_LDS double *)V, base, r, w, (QPC_LDS const double *)ref, tau, (QPC_LDS double *)dinv); return; }
  if (w <= 24) { sp_front_factor_reg<24>((QPC_LDS double *)V, base, r, w, (QPC_LDS const double *)ref, tau, (QPC_LDS double *)dinv); return; }
  if (w <= 32) { sp_front_factor_reg<32>((QPC_LDS double *)V, base, r, w, (QPC_LDS const double *)ref, tau, (QPC_LDS double *)dinv); return; }
#endif
  sp_ldl_panels(V, sp_cols_rect(base, r), r, w, ref, tau, dinv);
}

// ---- the update jobs of a level: target tile -= sum over its slots of (source column rows) d^-1 (source column rows)'
// jobs j0 .. j1-1, the wavefronts wave0 .. QPC_SP_NWAVE-1 taking them in turn
QPC_FN void sp_update_jobs(const QpSp &W, int j0, int j1, int wave0) {
  const int *pl = W.pl;
  const int *job = pl + pl[SP_O_JOB], *slot = pl + pl[SP_O_SLOT];
  const short *map = (const short *)(pl + pl[SP_O_MAP]);
  double *V = W.V;
#if defined(__HIP_DEVICE_COMPILE__)
  typedef double sp_v4d __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, kq = lane >> 4, ll = lane & 15;
  // wave0 > 0: the first wave0 wavefronts are busy with fronts -- the jobs go to the others, and (fewer fronts than SIMDs) not to the
  // wavefronts that share a SIMD with one of them: a front's elimination is the critical path, alone on its SIMD it issues twice as fast
  int nwv = 0, mine = -1;
  for (int w = 0; w < QPC_SP_NWAVE; w++) {
    const bool elig = w >= wave0 && (wave0 >= 4 || wave0 == 0 || (w & 3) >= wave0);
    if (w == QPC_SP_WAVE && elig) mine = nwv;
    nwv += elig ? 1 : 0;
  }
  if (nwv == 0) { nwv = 1; mine = QPC_SP_WAVE == QPC_SP_NWAVE - 1 ? 0 : -1; }
  for (int jb = j0 + mine; mine >= 0 && jb < j1; jb += nwv) {
    const int *J = job + jb * SP_JOB_INTS;
    const int tbase = J[0], tstride = J[1], trows = J[2], tcols = J[3], r0 = J[4], c0 = J[5], s0 = J[6], nch = J[7];
    const int col = c0 + ll, colc = col < tcols ? col : tcols - 1;
    const int cbase = tstride ? tbase + colc * tstride : tbase + sp_cs(colc, trows) - colc;
    int addr[4];
    bool okc[4];
    sp_v4d acc;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int row = r0 + kq + 4 * rr;
      okc[rr] = col < tcols && row < trows && row >= col;
      addr[rr] = cbase + (okc[rr] ? row : colc); // (an entry that does not exist: the column's diagonal, value unused)
      acc[rr] = V[addr[rr]];
    }
    for (int ch = 0; ch < nch; ch++) {
      const int *sl = slot + (s0 + 4 * ch + kq) * SP_SLOT_INTS;
      const int cb = sl[0] & 0xfffff, dp = sl[0] >> 20, mp = sl[1];
      const int mpc = mp < 0 ? 0 : mp;
      const int ra = map[mpc * 32 + ll], rb = map[mpc * 32 + 16 + ll];
      const double av = V[cb + (ra < 0 ? 0 : ra)], bv = V[cb + (rb < 0 ? 0 : rb)], dv = W.dinv[dp];
      const double a = (mp < 0 || ra < 0) ? 0.0 : -av, b = (mp < 0 || rb < 0) ? 0.0 : bv * dv;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) if (okc[rr]) V[addr[rr]] = acc[rr];
  }
#else
  for (int jb = j0; jb < j1; jb++) {
    const int *J = job + jb * SP_JOB_INTS;
    const int tbase = J[0], tstride = J[1], trows = J[2], tcols = J[3], r0 = J[4], c0 = J[5], s0 = J[6], nch = J[7];
    for (int lc = c0; lc < c0 + 16 && lc < tcols; lc++)
      for (int lr = (r0 > lc ? r0 : lc); lr < r0 + 16 && lr < trows; lr++) {
        double s = 0.0;
        for (int k = s0; k < s0 + 4 * nch; k++) {
          const int *sl = slot + k * SP_SLOT_INTS;
          if (sl[1] < 0) continue;
          const int ra = map[sl[1] * 32 + lr - r0], rb = map[sl[1] * 32 + 16 + lc - c0], cb = sl[0] & 0xfffff;
          if (ra >= 0 && rb >= 0) s += V[cb + ra] * (V[cb + rb] * W.dinv[sl[0] >> 20]);
        }
        V[tstride ? tbase + lc * tstride + lr : tbase + sp_sidx(lr, lc, trows)] -= s;
      }
  }
#endif
}

// the solves with a front's own block, in place at its columns (one wavefront)
QPC_FN void sp_front_forward(const double *V, int base, int r, int w, const double *dinv, double *t) {
  if (w == 1) return;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_SP_NO_REG)
  if (w <= 8) { sp_front_forward_reg<8>(V, base, r, w, dinv, t); return; }
  if (w <= 16) { sp_front_forward_reg<16>(V, base, r, w, dinv, t); return; }
  if (w <= 32) { sp_front_forward_reg<32>(V, base, r, w, dinv, t); return; }
#endif
  sp_tri_forward(V, sp_cols_rect(base, r), w, dinv, t);
}
QPC_FN void sp_front_back(const double *V, int base, int r, int w, const double *dinv, double *y) {
  if (w == 1) { QPC_SP_L0 { y[0] *= dinv[0]; } return; }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_SP_NO_REG)
  if (w <= 8) { sp_front_back_reg<8>(V, base, r, w, dinv, y); return; }
  if (w <= 16) { sp_front_back_reg<16>(V, base, r, w, dinv, y); return; }
  if (w <= 32) { sp_front_back_reg<32>(V, base, r, w, dinv, y); return; }
#endif
  sp_tri_back(V, sp_cols_rect(base, r), w, dinv, y, y);
}

// ---- the sparse products (qpc_qp_spview.h): a thread per segment of at most SP_SEG terms, then a thread per target
struct SpOpView { int nt, nseg; const int *tpos, *sptr, *segs, *terms; };
QPC_FN SpOpView sp_op(const QpSp &W, int op) {
  const int *pl = W.pl, *O = pl + pl[SP_O_OPS] + op * SP_OP_INTS, *D = pl + pl[SP_O_OPDATA];
  SpOpView V;
  V.nt = O[0]; V.nseg = O[1]; V.tpos = D + O[2]; V.sptr = D + O[3]; V.segs = D + O[4]; V.terms = D + O[5];
  return V;
}
// the segments' sums -> part (a barrier before the targets read them): a segment's packed indices all at once, then all its values
QPC_FN void sp_op_partials(const QpSp &W, const SpOpView &O, const double *val, const double *vec, double *part) {
  (void)W;
  QPC_SP_FOR(sg, O.nseg) {
    const int t0 = O.segs[sg], t1 = O.segs[sg + 1];
    int k[SP_SEG];
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) k[e] = O.terms[t0 + e < t1 ? t0 + e : t0];
    double a[SP_SEG], b[SP_SEG];
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) { a[e] = val[k[e] >> 9]; b[e] = vec[k[e] & 511]; }
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) acc += t0 + e < t1 ? a[e] * b[e] : 0.0;
    part[sg] = acc;
  }
}
QPC_FN double sp_op_sum(const SpOpView &O, int tg, const double *part) {
  double s = 0.0;
  for (int sg = O.sptr[tg]; sg < O.sptr[tg + 1]; sg++) s += part[sg];
  return s;
}

// ---- triangular solves with the equality rows' factor, front by front.  L is unit lower with L_ij = S_ij dinv_j.
// t (by position) <- L^-1 t on the equality positions; with_rows: the inequality positions too (t_I -= L_IE z).  A level's columns
// first take what the earlier levels' solutions leave in them (a sparse product), then every front solves with its own block.
// Barrier-synchronised on entry and on exit.
// the sums of the targets tg0 .. tg0 + w - 1 of a product, by ONE wavefront: a lane per segment, then (after the wavefront's own sync) lane
// c adds up target tg0 + c.  part: scratch of this wavefront's own (the segments of these targets).
QPC_FN double sp_op_wave_sums(const SpOpView &O, int tg0, int w, const double *val, const double *vec, double *part) {
  const int s0 = O.sptr[tg0], s1 = O.sptr[tg0 + w];
  for (int sg = s0 + QPC_SP_LANE; sg < s1; sg += QPC_SP_NLANE) {
    const int t0 = O.segs[sg], t1 = O.segs[sg + 1];
    int k[SP_SEG];
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) k[e] = O.terms[t0 + e < t1 ? t0 + e : t0];
    double a[SP_SEG], b[SP_SEG];
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) { a[e] = val[k[e] >> 9]; b[e] = vec[k[e] & 511]; }
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) acc += t0 + e < t1 ? a[e] * b[e] : 0.0;
    part[sg] = acc;
  }
  QPC_SP_WSYNC
  double s = 0.0;
  if (QPC_SP_LANE < w) s = sp_op_sum(O, tg0 + QPC_SP_LANE, part);
  return s; // (lane c < w: the sum of target tg0 + c; on the host, w = 1 per call -- see the callers)
}
// a single target's sum by one thread (the single-column fronts: a thread each)
QPC_FN double sp_op_thread_sum(const SpOpView &O, int tg, const double *val, const double *vec) {
  double s = 0.0;
  for (int sg = O.sptr[tg]; sg < O.sptr[tg + 1]; sg++) {
    const int t0 = O.segs[sg], t1 = O.segs[sg + 1];
    int k[SP_SEG];
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) k[e] = O.terms[t0 + e < t1 ? t0 + e : t0];
    double a[SP_SEG], b[SP_SEG];
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) { a[e] = val[k[e] >> 9]; b[e] = vec[k[e] & 511]; }
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < SP_SEG; e++) acc += t0 + e < t1 ? a[e] * b[e] : 0.0;
    s += acc;
  }
  return s;
}
QPC_FN void sp_forward(const QpSp &W, double *t, bool with_rows) {
  const int *pl = W.pl;
  const int *lv = pl + pl[SP_O_LEVEL], *ft = pl + pl[SP_O_FRONT];
  for (int l = 0; l < W.nlevel; l++) {
    const SpOpView O = sp_op(W, SP_OP_FWD0 + l);
    QPC_SP_FOR_WAVE(f, lv[2 * l], lv[2 * l + 1]) { // the fronts of several columns: a wavefront each
      const int *F = ft + f * SP_FRONT_INTS;
      const int base = F[0], r = F[1], w = F[2], c0 = F[3], tg0 = F[5];
      if (l > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        for (int cb = 0; cb < w; cb += 64) { const int wc = w - cb < 64 ? w - cb : 64; const double sm = sp_op_wave_sums(O, tg0 + cb, wc, W.V, W.zs, W.part); if ((int)(threadIdx.x & 63) < wc) t[c0 + cb + (threadIdx.x & 63)] -= sm; }
#else
        for (int c = 0; c < w; c++) t[c0 + c] -= sp_op_wave_sums(O, tg0 + c, 1, W.V, W.zs, W.part);
#endif
        QPC_SP_WSYNC
      }
      sp_front_forward(W.V, base, r, w, W.dinv + c0, t + c0);
      QPC_SP_WSYNC
      QPC_SP_FOR_LANE(c, w) { W.zs[c0 + c] = t[c0 + c] * W.dinv[c0 + c]; }
    }
    QPC_SP_FOR(e, lv[2 * l + 2] - lv[2 * l + 1]) { // the single columns: a thread each
      const int *F = ft + (lv[2 * l + 1] + e) * SP_FRONT_INTS;
      const int c0 = F[3];
      double v = t[c0];
      if (l > 0) { v -= sp_op_thread_sum(O, F[5], W.V, W.zs); t[c0] = v; }
      W.zs[c0] = v * W.dinv[c0];
    }
    QPC_SP_SYNC
  }
  if (with_rows) {
    const SpOpView O = sp_op(W, SP_OP_FWDROWS);
    sp_op_partials(W, O, W.V, W.zs, W.part);
    QPC_SP_SYNC
    QPC_SP_FOR(tg, O.nt) { t[O.tpos[tg]] -= sp_op_sum(O, tg, W.part); }
    QPC_SP_SYNC
  }
}
// y_E <- L^-T D^-1 z_E - (the rows below)' y, y given on the inequality positions; z_E is read from y's equality positions, the
// result replaces it.  Barrier-synchronised on entry and on exit.
QPC_FN void sp_back(const QpSp &W, double *y) {
  const int *pl = W.pl;
  const int *lv = pl + pl[SP_O_LEVEL], *ft = pl + pl[SP_O_FRONT];
  for (int l = W.nlevel - 1; l >= 0; l--) {
    const SpOpView O = sp_op(W, SP_OP_FWD0 + W.nlevel + l);
    QPC_SP_FOR_WAVE(f, lv[2 * l], lv[2 * l + 1]) {
      const int *F = ft + f * SP_FRONT_INTS;
      const int base = F[0], r = F[1], w = F[2], c0 = F[3], tg0 = F[5];
#if defined(__HIP_DEVICE_COMPILE__)
      for (int cb = 0; cb < w; cb += 64) { const int wc = w - cb < 64 ? w - cb : 64; const double sm = sp_op_wave_sums(O, tg0 + cb, wc, W.V, y, W.part); if ((int)(threadIdx.x & 63) < wc) y[c0 + cb + (threadIdx.x & 63)] -= sm; }
#else
      for (int c = 0; c < w; c++) y[c0 + c] -= sp_op_wave_sums(O, tg0 + c, 1, W.V, y, W.part);
#endif
      QPC_SP_WSYNC
      sp_front_back(W.V, base, r, w, W.dinv + c0, y + c0);
    }
    QPC_SP_FOR(e, lv[2 * l + 2] - lv[2 * l + 1]) {
      const int *F = ft + (lv[2 * l + 1] + e) * SP_FRONT_INTS;
      const int c0 = F[3];
      y[c0] = (y[c0] - sp_op_thread_sum(O, F[5], W.V, y)) * W.dinv[c0];
    }
    QPC_SP_SYNC
  }
}

#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_SP_NO_REG)
// the working set's factor in registers (k <= KR <= 32 rows, idxW filled): lane i = row i of C_WW, lane k = the right-hand side.
// Leaves LW and dinvW in LDS like the panel form (the Goldfarb-Idnani steps and the refinement solve with them) and, with_rhs, yw.
struct SpWArgs { // (the routine is not inlined)
  QPC_LDS const double *C; int M1;
  QPC_LDS const int *idxW;
  QPC_LDS const double *a0, *wsf, *hi, *lo, *gdiagI;
  QPC_LDS double *tW, *LW, *dinvW, *yw;
};
template <int KR>
__device__ __attribute__((noinline)) void sp_w_factor_reg(const SpWArgs A, int k, double tau, bool with_rhs) {
  const int lane = threadIdx.x & 63, K1 = k + 1;
  const int pi = A.idxW[lane < k ? lane : 0];
  if (with_rhs && lane < k) A.tW[lane] = A.a0[pi] - (A.wsf[pi] > 0.0 ? A.hi[pi] : A.lo[pi]);
  const double refv = lane < k ? A.gdiagI[pi] : 1.0;
  int pc[KR];
#pragma unroll
  for (int c = 0; c < KR; c++) pc[c] = A.idxW[c < k ? c : 0];
  QPC_SP_WSYNC
  double a[KR], mul[KR], dv = 0.0;
#pragma unroll
  for (int c = 0; c < KR; c++) {
    const double cv = A.C[sp_sym(pi, pc[c], A.M1)], rv = with_rhs ? A.tW[c < k ? c : 0] : 0.0;
    a[c] = c < k ? (lane < k ? cv : (lane == k ? rv : 0.0)) : 0.0;
  }
  sp_reg_elim<KR>(a, mul, true, refv, tau, dv);
#pragma unroll
  for (int c = 0; c < KR; c++) { if (c < k && lane <= k && lane >= c) A.LW[sp_cs(c, K1) + lane - c] = a[c]; }
  if (lane < k) A.dinvW[lane] = dv;
  QPC_SP_WSYNC
  if (with_rhs) { // y_W = L^-T D^-1 z: lane c takes its COLUMN of the factor back from LW, the substitution runs in registers
    double col[KR];
    const int cc = lane < k ? lane : 0, cb = sp_cs(cc, K1) - cc;
#pragma unroll
    for (int rr = 1; rr < KR; rr++) { const double v = A.LW[cb + (rr > cc && rr < k ? rr : cc)]; col[rr] = (rr < k && lane < rr) ? v : 0.0; }
    double t = lane < k ? A.LW[cb + k] : 0.0; // z_c (the right-hand-side row)
    double yv = 0.0;
#pragma unroll
    for (int rr = KR - 1; rr >= 0; rr--) {
      const double yr = sp_readlane(t, rr) * sp_readlane(dv, rr); // (a row beyond k: dv = 0)
      yv = lane == rr ? yr : yv;
      if (rr > 0) t -= col[rr] * yr;
    }
    if (lane < k) A.yw[lane] = yv;
    QPC_SP_WSYNC
  }
}
#endif

// Working set in W.wsf (by inequality position) -> idxW (ascending), factor of C_WW in W.LW; with_rhs: the right-hand side a0_W - b_W
// rides along and yw <- the multipliers of the working rows.  Returns the number of working rows, or -1 beyond what LW holds.
// One wavefront.
QPC_FN int sp_w_factor(QpSp &W, double tau, bool with_rhs) {
  const int mi = W.mi, kw = W.kw;
  int k = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  for (int p0 = 0; p0 < mi; p0 += 64) {
    const int p = p0 + (int)(threadIdx.x & 63);
    const bool f = p < mi && W.wsf[p] != 0.0;
    const unsigned long long mask = __ballot(f);
    if (f) { const int r = k + __popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull)); if (r < kw) W.idxW[r] = p; }
    k += __popcll(mask);
  }
#else
  for (int p = 0; p < mi; p++) if (W.wsf[p] != 0.0) { if (k < kw) W.idxW[k] = p; k++; }
#endif
  if (k > kw) return -1;
  QPC_SP_WSYNC
  const int K1 = k + 1;
  if (k == 0) return 0;
  QPC_SP_ACC_T0();
  QPC_SP_ACC_ADD(8, 1); QPC_SP_ACC_ADD(9, k);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_SP_NO_REG)
  if (k <= 32) {
    SpWArgs A;
    A.C = (QPC_LDS const double *)(W.V + W.Cb); A.M1 = W.M1; A.idxW = (QPC_LDS const int *)W.idxW; A.a0 = (QPC_LDS const double *)W.a0;
    A.wsf = (QPC_LDS const double *)W.wsf; A.hi = (QPC_LDS const double *)W.hi; A.lo = (QPC_LDS const double *)W.lo; A.gdiagI = (QPC_LDS const double *)(W.gdiag + W.ne);
    A.tW = (QPC_LDS double *)W.tW; A.LW = (QPC_LDS double *)W.LW; A.dinvW = (QPC_LDS double *)W.dinvW; A.yw = (QPC_LDS double *)W.yw;
    if (k <= 8) sp_w_factor_reg<8>(A, k, tau, with_rhs);
    else if (k <= 12) sp_w_factor_reg<12>(A, k, tau, with_rhs);
    else if (k <= 16) sp_w_factor_reg<16>(A, k, tau, with_rhs);
    else if (k <= 20) sp_w_factor_reg<20>(A, k, tau, with_rhs);
    else if (k <= 24) sp_w_factor_reg<24>(A, k, tau, with_rhs);
    else if (k <= 28) sp_w_factor_reg<28>(A, k, tau, with_rhs);
    else sp_w_factor_reg<32>(A, k, tau, with_rhs);
    QPC_SP_ACC(1);
    return k;
  }
#endif
  // lane i gathers ROW i of C_WW (lane k: the right-hand side), four columns at a time: the index loads, then the loads of C
  for (int i = QPC_SP_LANE; i <= k; i += QPC_SP_NLANE) {
    const int pi = i < k ? W.idxW[i] : 0;
    for (int c0 = 0; c0 < k && c0 <= i; c0 += 4) {
      int pc[4];
      double v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) pc[e] = W.idxW[c0 + e < k ? c0 + e : c0];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const double cv = sp_C(W, pi, pc[e]);
        const double rv = (with_rhs && i == k) ? W.a0[pc[e]] - (W.wsf[pc[e]] > 0.0 ? W.hi[pc[e]] : W.lo[pc[e]]) : 0.0;
        v[e] = i == k ? rv : cv;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) if (c0 + e < k && c0 + e <= i) W.LW[sp_cs(c0 + e, K1) + i - (c0 + e)] = v[e];
    }
    if (i < k) W.tW[i] = W.gdiag[W.ne + pi];
  }
  QPC_SP_WSYNC
  QPC_SP_ACC(0);
  const SpCols LC = sp_cols_packed(K1);
  sp_ldl_panels(W.LW, LC, K1, k, W.tW, tau, W.dinvW);
  QPC_SP_ACC(1);
  if (with_rhs) {
    QPC_SP_FOR_LANE(a, k) { W.tW[a] = W.LW[sp_sidx(k, a, K1)]; }
    QPC_SP_WSYNC
    sp_tri_back(W.LW, LC, k, W.dinvW, W.tW, W.yw);
  }
  QPC_SP_ACC(2);
  return k;
}
// EQP of the working set: multipliers of its rows -> yw, ax = A x for every inequality row.  One wavefront.
QPC_FN int sp_solve_ws(QpSp &W, double tau) {
  const int k = sp_w_factor(W, tau, true);
  if (k < 0) return -1;
  QPC_SP_ACC_T0();
  QPC_SP_FOR_LANE(p, W.mi) {
    double t = W.a0[p];
#pragma unroll 8
    for (int a = 0; a < k; a++) t -= sp_C(W, p, W.idxW[a]) * W.yw[a];
    W.ax[p] = t;
  }
  QPC_SP_WSYNC
  QPC_SP_ACC(3);
  return k;
}

// ---- the working-set iterations (qpc_qp_wg.h: the same sweeps, the same Goldfarb-Idnani steps, the same ties), one wavefront.
//      Leaves status / iters / nw / converged in W.sc.
QPC_FN void sp_iterate(QpSp &W, const qpc_settings &st, double bscale) {
  const double INF = 1e19;
  const int mi = W.mi;
  int status = QPC_STATUS_OPTIMAL, iters = 0;
  const int max_it = st.max_active_set_iters > 0 ? st.max_active_set_iters : 200;
  int nw = sp_solve_ws(W, st.eps_dual_reg);
  if (nw < 0) { // (a warm start beyond what LW holds: start cold)
    QPC_SP_FOR_LANE(p, mi) { W.wsf[p] = 0.0; }
    QPC_SP_WSYNC
    nw = sp_solve_ws(W, st.eps_dual_reg);
  }
  bool converged = false, fallback = false;
  const int max_pdas = 4;
  for (; iters < max_pdas && iters < max_it; iters++) {
    double a_y = 1.0, a_x = 1.0;
    QPC_SP_FOR_LANE(a, nw) { a_y = QPC_FABS(W.yw[a]) > a_y ? QPC_FABS(W.yw[a]) : a_y; }
    QPC_SP_FOR_LANE(p, mi) { a_x = QPC_FABS(W.ax[p]) > a_x ? QPC_FABS(W.ax[p]) : a_x; W.tm[p] = 0.0; }
    const double ysc = sp_wave_max(a_y), xsc = sp_wave_max(a_x);
    const double toly = st.tol_bound * ysc, tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
    bool ch = false;
    QPC_SP_WSYNC
    QPC_SP_FOR_LANE(a, nw) { // wrong-signed multipliers and dependent (dropped) rows leave
      const int p = W.idxW[a];
      const double w = W.wsf[p];
      const double viol = w < 0.0 ? W.yw[a] : -W.yw[a];
      if (viol > toly || W.dinvW[a] == 0.0) ch = true;
      else W.tm[p] = w;
    }
    QPC_SP_WSYNC
    QPC_SP_FOR_LANE(p, mi) {
      if (W.wsf[p] == 0.0) {
        const double lo = W.lo[p] > -INF ? W.lo[p] - W.ax[p] : -1.0, hi = W.hi[p] < INF ? W.ax[p] - W.hi[p] : -1.0;
        if (lo > tolx && lo >= hi) { W.tm[p] = -1.0; ch = true; }
        else if (hi > tolx) { W.tm[p] = 1.0; ch = true; }
      }
    }
    if (!sp_wave_any(ch)) { converged = true; break; }
    QPC_SP_WSYNC
    QPC_SP_FOR_LANE(p, mi) { W.wsf[p] = W.tm[p]; }
    QPC_SP_WSYNC
    nw = sp_solve_ws(W, st.eps_dual_reg);
    if (nw < 0) { fallback = true; break; }
  }
  // ---- Goldfarb-Idnani from what is left of the sweeps' last set once the rows with wrong-signed multipliers are gone
  if (!converged && !fallback) {
    for (int pass = 0; pass <= mi; pass++) {
      double a_y = 1.0;
      QPC_SP_FOR_LANE(a, nw) { a_y = QPC_FABS(W.yw[a]) > a_y ? QPC_FABS(W.yw[a]) : a_y; }
      const double toly = st.tol_bound * sp_wave_max(a_y);
      bool ch = false;
      QPC_SP_FOR_LANE(a, nw) {
        const int p = W.idxW[a];
        const double viol = W.wsf[p] < 0.0 ? W.yw[a] : -W.yw[a];
        if (viol > toly || W.dinvW[a] == 0.0) { W.tm[p] = 0.0; ch = true; } else W.tm[p] = W.wsf[p];
      }
      if (!sp_wave_any(ch)) break;
      QPC_SP_WSYNC
      QPC_SP_FOR_LANE(a, nw) { const int p = W.idxW[a]; W.wsf[p] = W.tm[p]; }
      QPC_SP_WSYNC
      nw = sp_solve_ws(W, st.eps_dual_reg);
      if (nw < 0) { fallback = true; break; }
    }
    for (; !fallback && iters < max_it; iters++) {
      double a_x = 1.0;
      QPC_SP_FOR_LANE(p, mi) { a_x = QPC_FABS(W.ax[p]) > a_x ? QPC_FABS(W.ax[p]) : a_x; }
      const double xsc = sp_wave_max(a_x);
      const double tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
      // the most violated row (ties: the first row; a row's lower bound before its upper one): candidates numbered 2 p + (upper ? 1 : 0)
      double bv = 0.0;
      int bi = -1;
      QPC_SP_FOR_LANE(p, mi) {
        if (W.wsf[p] != 0.0) continue;
        const double lo = W.lo[p] > -INF ? W.lo[p] - W.ax[p] : -1.0, hi = W.hi[p] < INF ? W.ax[p] - W.hi[p] : -1.0;
        if (lo > tolx && (bi < 0 || lo > bv)) { bv = lo; bi = 2 * p; }
        if (hi > tolx && (bi < 0 || hi > bv)) { bv = hi; bi = 2 * p + 1; }
      }
      sp_wave_argmax(bv, bi);
      const int bp = bi >> 1; // (-1 stays -1)
      const double bsg = (bi & 1) ? 1.0 : -1.0;
      const int pr = bp;
      if (pr < 0) { converged = true; break; }
      const double sg = bsg;
      double viol = bv;
      bool infeasible = false;
      int k = nw; // LW is the factor of the current working set
      for (int inner = 0; inner <= W.m; inner++) {
        // r = (C_WW)^-1 C_Wp
        QPC_SP_FOR_LANE(a, k) { W.tW[a] = sp_C(W, W.idxW[a], pr); }
        QPC_SP_WSYNC
        sp_tri_forward(W.LW, sp_cols_packed(k + 1), k, W.dinvW, W.tW);
        sp_tri_back(W.LW, sp_cols_packed(k + 1), k, W.dinvW, W.tW, W.dW);
        double dpart = 0.0;
        QPC_SP_FOR_LANE(a, k) { dpart += sp_C(W, pr, W.idxW[a]) * W.dW[a]; }
        // (the sum over the lanes in lane order: a fixed order)
        double d = W.V[W.Cb + sp_cs(pr, W.M1)];
#if defined(__HIP_DEVICE_COMPILE__)
        for (int off = 32; off > 0; off >>= 1) dpart += __shfl_xor(dpart, off, 64);
#endif
        d -= dpart;
        const bool dependent = !(d > 1e-11 * W.gdiag[W.ne + pr]);
        double t1 = 1e300; // dual ratio test: y_W(t) = y_W - t sg r must keep the sign of every working inequality
        int kd = -1;
        QPC_SP_FOR_LANE(a, k) {
          const double w = W.wsf[W.idxW[a]], sr = sg * W.dW[a];
          if (W.dinvW[a] == 0.0) continue;
          if ((w > 0.0 && sr > 1e-14) || (w < 0.0 && sr < -1e-14)) {
            double t = W.yw[a] / sr;
            t = t > 0.0 ? t : 0.0;
            if (kd < 0 || t < t1) { t1 = t; kd = a; }
          }
        }
        sp_wave_argmin(t1, kd);
        if (kd < 0) t1 = 1e300;
        int outcome; // 0: full step, p enters; 1: partial step, row kd left; 2: infeasible
        if (dependent && kd < 0) outcome = 2;
        else {
          const double t2 = dependent ? 1e300 : viol / d;
          if (t2 <= t1) outcome = 0;
          else {
            outcome = 1;
            QPC_SP_WSYNC
            // y_W -= t1 sg r, row kd leaves, the multipliers close up (through tW: free here)
            QPC_SP_FOR_LANE(a, k) { const int src = a >= kd ? a + 1 : a; W.tW[a] = src < k ? W.yw[src] - t1 * sg * W.dW[src] : 0.0; }
            QPC_SP_WSYNC
            QPC_SP_FOR_LANE(a, k) { if (a + 1 < k) W.yw[a] = W.tW[a]; }
            if (!dependent) viol -= t1 * d;
            QPC_SP_L0 { W.wsf[W.idxW[kd]] = 0.0; }
            QPC_SP_WSYNC
          }
        }
        if (outcome == 2) { infeasible = true; break; }
        if (outcome == 0) break;
        k = sp_w_factor(W, st.eps_dual_reg, false); // a row left: the factor of what stays
        if (k < 0) { fallback = true; break; }
      }
      if (fallback) break;
      if (infeasible) { status = QPC_STATUS_INFEASIBLE; break; }
      QPC_SP_L0 { W.wsf[pr] = sg; }
      QPC_SP_WSYNC
      nw = sp_solve_ws(W, st.eps_dual_reg);
      if (nw < 0) { fallback = true; break; }
    }
  }
  if (!converged && status == QPC_STATUS_OPTIMAL) status = QPC_STATUS_ITERATION_LIMIT;
  QPC_SP_L0 {
    W.sc[SPS_STATUS] = fallback ? (double)QPC_WG_FALLBACK : (double)status;
    W.sc[SPS_ITERS] = (double)iters; W.sc[SPS_NW] = (double)nw; W.sc[SPS_CONV] = converged ? 1.0 : 0.0;
  }
}

// ---- products with A, H, H^-1 (A's compressed columns / rows by position and the blocks' rows from the plan in LDS)
QPC_FN double sp_h_dot(const QpSp &W, const double *hvals, int i, const double *v) { // (row i of the block matrix) . v
  const int *hptr = W.pl + W.pl[SP_O_HPTR], *hcol = W.pl + W.pl[SP_O_HCOL], *hidx = W.pl + W.pl[SP_O_HIDX];
  double t = 0.0;
  for (int k = hptr[i]; k < hptr[i + 1]; k++) t += hvals[hidx[k]] * v[hcol[k]];
  return t;
}

// copies the LDS part of the plan (once per workgroup)
QPC_FN void qp_sp_load_plan(const int *plan_g, int *plan_lds) {
  const int nint = plan_g[SP_LDS_INTS];
  QPC_SP_FOR(i, nint) { plan_lds[i] = plan_g[i]; }
  QPC_SP_SYNC
}

// One QP.  plan_lds: the plan's LDS part (qp_sp_load_plan), lds: what follows it, lds_cap doubles.  Returns the status, or
// QPC_WG_FALLBACK (the set of equality rows is not the plan's; a working set beyond what LDS holds).
QPC_FN int qp_solve_sp(const int *plan_g, const int *plan_lds, const double *Pv, const double *q, const double *Av, const double *l,
                       const double *u, const qpc_settings &st, double *lds, int lds_cap, const double *y0, double *x, double *y,
                       int32_t *iters_out, unsigned long long *stamps) {
  const double INF = 1e19;
  const int *pl = plan_lds;
  QpSp W;
  W.pl = pl; W.plg = plan_g;
  const int n = pl[SP_N], m = pl[SP_M], ne = pl[SP_NE], mi = pl[SP_MI], nnzA = pl[SP_NNZA], nhv = pl[SP_NHV], nval = pl[SP_NVAL];
  W.n = n; W.m = m; W.ne = ne; W.mi = mi; W.N1 = m + 1; W.M1 = mi + 1; W.Cb = pl[SP_CB]; W.nfront = pl[SP_NFRONT]; W.nlevel = pl[SP_NLEVEL];
  W.Pv = Pv; W.q = q; W.Av = Av; W.l = l; W.u = u; W.y0 = y0; W.stamps = stamps;
  QPC_SP_STAMP(W, 0);
  double *p = lds;
  W.sc = p; p += 16; W.red = p; p += 32;
  W.V = p; p += nval;
  W.dinv = p; p += ne; W.gdiag = p; p += m + 1; W.t = p; p += m + 1; W.yp = p; p += m + 1;
  W.zs = p; p += ne; W.part = p; p += pl[SP_MAXSEG]; W.part2 = p; p += pl[SP_MAXSEG];
  W.hv = p; p += nhv; W.hm = p; p += nhv;
  W.xu = p; p += n; W.x = p; p += n; W.r1 = p; p += n; W.tn = p; p += n;
  W.lo = p; p += mi; W.hi = p; p += mi; W.wsf = p; p += mi; W.tm = p; p += mi; W.ax = p; p += mi; W.a0 = p; p += mi;
  W.sval = p; p += nnzA;
  // the working set's factor takes what is left (at most the inequality rows)
  int kw = mi;
  while (kw > 0 && (int)(p - lds) + 4 * (kw + 1) + (kw + 2) / 2 + (kw + 1) * (kw + 2) / 2 > lds_cap) kw--;
  if ((int)(p - lds) + 4 * (kw + 1) + (kw + 2) / 2 + (kw + 1) * (kw + 2) / 2 > lds_cap) return QPC_WG_FALLBACK;
  W.kw = kw;
  W.yw = p; p += kw + 1; W.dinvW = p; p += kw + 1; W.dW = p; p += kw + 1; W.tW = p; p += kw + 1;
  W.idxW = (int *)p; p += (kw + 2) / 2;
  W.LW = p; p += (kw + 1) * (kw + 2) / 2;
  const int *perm = pl + pl[SP_O_PERM], *ipos = pl + pl[SP_O_IPOS];
  // ---- the values; the rows' kinds against the plan's; scales
  QPC_SP_T0 { for (int i = 0; i < SPS_COUNT; i++) W.sc[i] = 0.0; }
  QPC_SP_FOR(o, nnzA) { W.sval[o] = Av[o]; } // (the plan's tables index the caller's order)
  {
    const int *hptr = pl + pl[SP_O_HPTR], *hidx = pl + pl[SP_O_HIDX], *pidx = pl + pl[SP_O_PIDX];
    QPC_SP_FOR(i, n) {
      for (int k = hptr[i]; k < hptr[i + 1]; k++) W.hm[hidx[k]] = pidx[k] >= 0 ? Pv[pidx[k]] : 0.0;
    }
  }
  double a_b = 1.0;
  bool mismatch = false;
  QPC_SP_FOR(i, m) {
    const double li = l[i], ui = u[i];
    const bool lo = li > -INF, hi = ui < INF;
    const bool eq = lo && hi && ui - li <= 1e-12 * (1.0 + QPC_FABS(li));
    if (eq != (ipos[i] < ne)) mismatch = true;
    if (lo) a_b = QPC_FABS(li) > a_b ? QPC_FABS(li) : a_b;
    if (hi) a_b = QPC_FABS(ui) > a_b ? QPC_FABS(ui) : a_b;
    const int pos = ipos[i];
    if (pos >= ne) { W.lo[pos - ne] = li; W.hi[pos - ne] = ui; }
  }
  QPC_SP_SYNC
  double a_h = 0.0;
  {
    const int *hptr = pl + pl[SP_O_HPTR], *hcol = pl + pl[SP_O_HCOL], *hidx = pl + pl[SP_O_HIDX];
    QPC_SP_FOR(i, n) {
      for (int k = hptr[i]; k < hptr[i + 1]; k++) if (hcol[k] == i) { const double a = QPC_FABS(W.hm[hidx[k]]); a_h = a > a_h ? a : a_h; }
    }
  }
  double red3[3] = {a_b, a_h, mismatch ? 1.0 : 0.0};
  sp_reduce_max<3>(red3, W.red);
  const double bscale = red3[0], hmax = red3[1];
  if (red3[2] != 0.0) return QPC_WG_FALLBACK;
  const double eps = st.eps_primal_reg * (hmax > 0.0 ? hmax : 1.0);
  // ---- H^-1, block by block: a single variable by a division, a block by Gauss-Jordan sweeps of one wavefront (lane = entry)
  {
    const int nblk = pl[SP_NBLK];
    const int *bptr = pl + pl[SP_O_BPTR], *bbase = pl + pl[SP_O_BBASE];
    QPC_SP_FOR(b, nblk) { if (bptr[b + 1] - bptr[b] == 1) W.hv[bbase[b]] = 1.0 / (W.hm[bbase[b]] + eps); }
    QPC_SP_FOR_WAVE(b, 0, nblk) {
      const int bs = bptr[b + 1] - bptr[b], hb = bbase[b];
      if (bs == 1) continue;
      QPC_SP_FOR_LANE(e, bs * bs) { W.hv[hb + e] = W.hm[hb + e] + ((e / bs == e % bs) ? eps : 0.0); }
      QPC_SP_WSYNC
      for (int k = 0; k < bs; k++) {
        const double pv = 1.0 / W.hv[hb + k * bs + k];
        double nv[1 + (QPC_SP_MAXBLK * QPC_SP_MAXBLK - 1) / QPC_SP_NLANE];
        int cnt = 0;
        for (int e = QPC_SP_LANE; e < bs * bs; e += QPC_SP_NLANE) {
          const int i = e / bs, j = e % bs;
          const double aik = W.hv[hb + i * bs + k], akj = W.hv[hb + k * bs + j], aij = W.hv[hb + e];
          nv[cnt++] = (i == k && j == k) ? pv : (i == k ? akj * pv : (j == k ? -aik * pv : aij - aik * akj * pv));
        }
        QPC_SP_WSYNC
        cnt = 0;
        for (int e = QPC_SP_LANE; e < bs * bs; e += QPC_SP_NLANE) W.hv[hb + e] = nv[cnt++];
        QPC_SP_WSYNC
      }
    }
  }
  QPC_SP_FOR(e, nval) { W.V[e] = 0.0; }
  QPC_SP_SYNC
  QPC_SP_FOR(i, n) { W.tn[i] = -q[i]; }
  QPC_SP_SYNC
  QPC_SP_FOR(i, n) { W.xu[i] = sp_h_dot(W, W.hv, i, W.tn); }
  QPC_SP_STAMP(W, 1);
  // ---- S = [G ; rhs'] into the store: a thread per SEGMENT (at most SP_SEG products) of a structural entry of G, its products in the
  //      order the host laid down; an entry with several segments is summed from their sums afterwards.  The term lists stream from
  //      global memory: a segment's are loaded up front, eight at a time.
  {
    const int ngseg = pl[SP_NGSEG];
    const int *gseg = plan_g + pl[SP_O_GSEG], *term = plan_g + pl[SP_O_TERM];
    QPC_SP_FOR(sg, ngseg) {
      const int dst = gseg[2 * sg], t0 = gseg[2 * sg + 1], t1 = gseg[2 * sg + 3];
      double acc = 0.0;
      for (int t = t0; t < t1; t += 8) {
        int ka[8], kh[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { const int tt = t + e < t1 ? t + e : t0; ka[e] = term[2 * tt]; kh[e] = term[2 * tt + 1]; }
        double a[8], b[8], h[8];
#pragma unroll
        for (int e = 0; e < 8; e++) { a[e] = W.sval[ka[e] & 0xffff]; b[e] = W.sval[(ka[e] >> 16) & 0xffff]; h[e] = W.hv[kh[e]]; }
#pragma unroll
        for (int e = 0; e < 8; e++) acc += t + e < t1 ? a[e] * b[e] * h[e] : 0.0;
      }
      if (dst >= 0) W.V[dst] = acc; else W.part[-dst - 1] = acc;
    }
  }
  QPC_SP_SYNC // (xu complete)
  {
    const int ngsum = pl[SP_NGSUM];
    const int *gsum = plan_g + pl[SP_O_GSUM];
    QPC_SP_FOR(e, ngsum) {
      double acc = 0.0;
      for (int k = gsum[3 * e + 1]; k < gsum[3 * e + 2]; k++) acc += W.part[k];
      W.V[gsum[3 * e]] = acc;
    }
  }
  QPC_SP_SYNC // (the segments' sums are used up: the products below reuse the array)
  const SpOpView Orow = sp_op(W, SP_OP_ROWDOT);
  const SpOpView Ocol = sp_op(W, SP_OP_COLDOT);
  sp_op_partials(W, Orow, W.sval, W.xu, W.part);
  QPC_SP_SYNC
  {
    const int *rtaddr = plan_g + pl[SP_O_RTADDR];
    QPC_SP_FOR(pos, m) { W.V[rtaddr[pos]] = sp_op_sum(Orow, pos, W.part) - (pos < ne ? l[perm[pos]] : 0.0); }
  }
  QPC_SP_SYNC
  {
    const int *colfront = pl + pl[SP_O_COLFRONT], *ft = pl + pl[SP_O_FRONT];
    QPC_SP_FOR(pos, m) {
      if (pos < ne) { const int *F = ft + colfront[pos] * SP_FRONT_INTS; const int lc = pos - F[3]; W.gdiag[pos] = W.V[F[0] + lc * F[1] + lc]; }
      else W.gdiag[pos] = W.V[W.Cb + sp_cs(pos - ne, W.M1)];
    }
  }
  QPC_SP_SYNC
  QPC_SP_STAMP(W, 2);
  // ---- the elimination, level by level: a level's fronts (a wavefront each), then their updates of later fronts and of C (tile
  //      jobs).  Only the jobs into the NEXT level's fronts have somebody waiting: the others run beside those fronts, on the
  //      wavefronts the fronts leave free (when the level has fewer fronts than the workgroup wavefronts).
  {
    const int *lv = pl + pl[SP_O_LEVEL], *ft = pl + pl[SP_O_FRONT], *jl = pl + pl[SP_O_JOBLV];
    auto fronts = [&](int lev) {
      QPC_SP_FOR_WAVE(f, lv[2 * lev], lv[2 * lev + 1]) {
        const int *F = ft + f * SP_FRONT_INTS;
        sp_front_factor(W.V, F[0], F[1], F[2], W.gdiag + F[3], st.eps_dual_reg, W.dinv + F[3]);
      }
      QPC_SP_FOR(e, lv[2 * lev + 2] - lv[2 * lev + 1]) { // a single column: its pivot, nothing else (the rows below keep L D); a thread each
        const int *F = ft + (lv[2 * lev + 1] + e) * SP_FRONT_INTS;
        const double d = W.V[F[0]], rf = W.gdiag[F[3]];
        W.dinv[F[3]] = d > st.eps_dual_reg * (rf > 1e-300 ? rf : 1e-300) ? 1.0 / d : 0.0;
      }
    };
    if (W.nlevel > 0) fronts(0);
    QPC_SP_SYNC
    QPC_SP_STAMP(W, 3);
    for (int lev = 0; lev < W.nlevel; lev++) {
      sp_update_jobs(W, jl[2 * lev], jl[2 * lev + 1], 0);
      QPC_SP_SYNC
      QPC_SP_STAMP(W, lev < 2 ? 4 + 2 * lev : 8);
      const int nf = lev + 1 < W.nlevel ? lv[2 * lev + 3] - lv[2 * lev + 2] : 0; // (the next level's fronts of several columns: its single columns cost nothing)
      if (lev + 1 < W.nlevel) fronts(lev + 1);
      sp_update_jobs(W, jl[2 * lev + 1], jl[2 * lev + 2], nf < QPC_SP_NWAVE ? nf : 0);
      QPC_SP_SYNC
      QPC_SP_STAMP(W, lev < 2 ? 5 + 2 * lev : 8);
    }
  }
#if defined(QPC_SP_DEBUG_DUMP) && !defined(__HIP_DEVICE_COMPILE__)
  if (const char *dump = getenv("QPC_SP_DUMP")) { // (test emulation only: the store after the elimination)
    FILE *f = fopen(dump, "wb");
    if (f) { fwrite(W.V, 8, nval, f); fwrite(W.dinv, 8, ne, f); fwrite(W.gdiag, 8, m + 1, f); fclose(f); }
  }
#endif
  QPC_SP_FOR(p2, mi) {
    W.a0[p2] = W.V[W.Cb + sp_sidx(mi, p2, W.M1)];
    double w = 0.0;
    if (y0) { // warm start: the rows an earlier solution held at a bound
      const double yv = y0[perm[ne + p2]];
      w = (yv > 0.0 && W.hi[p2] < INF) ? 1.0 : ((yv < 0.0 && W.lo[p2] > -INF) ? -1.0 : 0.0);
    }
    W.wsf[p2] = w;
  }
  QPC_SP_SYNC
  // ---- the working-set iterations: one wavefront's work
  if (QPC_SP_WAVE == 0) sp_iterate(W, st, bscale);
  QPC_SP_SYNC
  int status = (int)W.sc[SPS_STATUS];
  const int iters = (int)W.sc[SPS_ITERS], nw = (int)W.sc[SPS_NW];
  if (status == QPC_WG_FALLBACK) return QPC_WG_FALLBACK;
  QPC_SP_STAMP(W, 9);
  // ---- the multipliers by position: the working rows', then the equality rows' -- L11' yE = D^-1 (zE - (rows of W)' yW)
  double *yp = W.yp;
  {
    const int *colfront = pl + pl[SP_O_COLFRONT], *ft = pl + pl[SP_O_FRONT];
    QPC_SP_FOR(b, m + 1) {
      double v = 0.0;
      if (b < ne) { const int *F = ft + colfront[b] * SP_FRONT_INTS; v = W.V[F[0] + (b - F[3]) * F[1] + F[1] - 1]; } // the right-hand-side row: zE (unscaled)
      yp[b] = v;
    }
  }
  QPC_SP_SYNC
  QPC_SP_FOR(a, nw) { yp[ne + W.idxW[a]] = W.yw[a]; }
  QPC_SP_SYNC
  sp_back(W, yp);
  QPC_SP_STAMP(W, 10);
  // ---- x = xu - H^-1 A' y, then iterative refinement on the true KKT residuals of the final working set
  sp_op_partials(W, Ocol, W.sval, yp, W.part2);
  QPC_SP_SYNC
  QPC_SP_FOR(i, n) { W.tn[i] = sp_op_sum(Ocol, i, W.part2); } // A' y: kept in tn for the first residual
  QPC_SP_SYNC
  QPC_SP_FOR(i, n) { W.x[i] = W.xu[i] - sp_h_dot(W, W.hv, i, W.tn); }
  QPC_SP_SYNC
  QPC_SP_STAMP(W, 15);
  double rp = 0.0, rs = 0.0;
  for (int sweep = 0; sweep <= st.max_refine_iters; sweep++) {
    double a_rs = 0.0, a_rp = 0.0, a_ss = 1.0, a_ps = 1.0;
    sp_op_partials(W, Ocol, W.sval, yp, W.part2);
    sp_op_partials(W, Orow, W.sval, W.x, W.part);
    QPC_SP_SYNC
    QPC_SP_FOR(i, n) { // r1 = -(q + P x + A' y)
      const double aty = sp_op_sum(Ocol, i, W.part2), qi = q[i];
      const double t = qi + sp_h_dot(W, W.hm, i, W.x) + aty;
      W.r1[i] = -t;
      a_rs = (QPC_FABS(t) > a_rs || !(t == t)) ? QPC_FABS(t) : a_rs;
      a_ss = QPC_FABS(qi) > a_ss ? QPC_FABS(qi) : a_ss;
      a_ss = QPC_FABS(aty) > a_ss ? QPC_FABS(aty) : a_ss;
    }
    QPC_SP_FOR(b, m + 1) { // r2 = b_W - A_W x, by position (the rows outside the working set: 0)
      const int p2 = b - ne;
      double t = 0.0;
      if (b < m && (b < ne || W.wsf[p2] != 0.0)) {
        const double bw = b < ne ? l[perm[b]] : (W.wsf[p2] > 0.0 ? W.hi[p2] : W.lo[p2]);
        t = bw - sp_op_sum(Orow, b, W.part);
        a_rp = (QPC_FABS(t) > a_rp || !(t == t)) ? QPC_FABS(t) : a_rp;
        a_ps = QPC_FABS(bw) > a_ps ? QPC_FABS(bw) : a_ps;
      }
      W.t[b] = t;
    }
    QPC_SP_SYNC
    double red4[4] = {a_rs, a_rp, a_ss, a_ps};
    sp_reduce_max<4>(red4, W.red);
    rs = red4[0]; rp = red4[1];
    QPC_SP_STAMP(W, sweep == 0 ? 16 : 22);
    if (sweep == st.max_refine_iters) break;
    // after a correction: residuals at the rounding floor of their own scales (|q|, |A'y|; the working rows' bounds) -- another
    // correction would change nothing that 1e-8 can see (the eps bias of the first solve is 1e-6: one correction takes it to 1e-12)
    if (sweep > 0 && rs <= 1e-12 * red4[2] && rp <= 1e-12 * red4[3]) break;
    // dy = K^-1 (A_W H^-1 r1 - r2) ; dx = H^-1 (r1 - A_W' dy)
    QPC_SP_FOR(i, n) { W.xu[i] = sp_h_dot(W, W.hv, i, W.r1); } // xu is free from here on: H^-1 r1
    QPC_SP_SYNC
    sp_op_partials(W, Orow, W.sval, W.xu, W.part);
    QPC_SP_SYNC
    QPC_SP_FOR(b, m) {
      const int p2 = b - ne;
      if (b < ne || W.wsf[p2] != 0.0) W.t[b] = sp_op_sum(Orow, b, W.part) - W.t[b];
    }
    QPC_SP_SYNC
    // K on (E, W): forward through the fronts (the working rows pulled along), the working set's factor, back through the fronts
    if (sweep == 0) QPC_SP_STAMP(W, 17);
    sp_forward(W, W.t, true);
    if (sweep == 0) QPC_SP_STAMP(W, 18);
    if (QPC_SP_WAVE == 0) { // the working rows' block, then dy on the inequality positions: the working rows' solution, 0 elsewhere
      QPC_SP_FOR_LANE(a, nw) { W.tW[a] = W.t[ne + W.idxW[a]]; }
      QPC_SP_WSYNC
      sp_tri_forward(W.LW, sp_cols_packed(nw + 1), nw, W.dinvW, W.tW);
      sp_tri_back(W.LW, sp_cols_packed(nw + 1), nw, W.dinvW, W.tW, W.dW);
      QPC_SP_FOR_LANE(b, mi + 1) { W.t[ne + b] = 0.0; }
      QPC_SP_WSYNC
      QPC_SP_FOR_LANE(a, nw) { W.t[ne + W.idxW[a]] = W.dW[a]; }
    }
    QPC_SP_SYNC
    if (sweep == 0) QPC_SP_STAMP(W, 19);
    sp_back(W, W.t); // W.t: dy by position
    if (sweep == 0) QPC_SP_STAMP(W, 20);
    double a_big = 0.0, a_dbig = 0.0;
    sp_op_partials(W, Ocol, W.sval, W.t, W.part2);
    QPC_SP_SYNC
    QPC_SP_FOR(i, n) { W.tn[i] = W.r1[i] - sp_op_sum(Ocol, i, W.part2); }
    QPC_SP_SYNC
    QPC_SP_FOR(i, n) {
      const double dx = sp_h_dot(W, W.hv, i, W.tn);
      const double xi = W.x[i] + dx;
      W.x[i] = xi;
      a_big = QPC_FABS(xi) > a_big ? QPC_FABS(xi) : a_big;
      a_dbig = QPC_FABS(dx) > a_dbig ? QPC_FABS(dx) : a_dbig;
    }
    QPC_SP_FOR(b, m) { yp[b] += W.t[b]; }
    QPC_SP_SYNC
    double red2[2] = {a_big, a_dbig};
    sp_reduce_max<2>(red2, W.red);
    if (sweep == 0) QPC_SP_STAMP(W, 21);
    if (sweep > 0 && !(red2[1] > 1e-12 * (1.0 + red2[0]))) break;
  }
  QPC_SP_STAMP(W, 11);
  double a_ysc = 1.0, a_qsc = 1.0;
  QPC_SP_FOR(b, m) { a_ysc = QPC_FABS(yp[b]) > a_ysc ? QPC_FABS(yp[b]) : a_ysc; }
  QPC_SP_FOR(i, n) { a_qsc = QPC_FABS(q[i]) > a_qsc ? QPC_FABS(q[i]) : a_qsc; }
  double redq[2] = {a_ysc, a_qsc};
  sp_reduce_max<2>(redq, W.red);
  const double ysc = redq[0], qsc = redq[1];
  if (status == QPC_STATUS_OPTIMAL) {
    if (!(rp < 1e300) || !(rs < 1e300)) status = QPC_STATUS_NUMERICAL_ERROR; // (a NaN residual was reduced as +infinity)
    else if (rp > 1e-7 * bscale) status = QPC_STATUS_INFEASIBLE;
    else if (rs > 1e-7 * (ysc > qsc ? ysc : qsc)) status = QPC_STATUS_ALMOST_OPTIMAL;
  }
  QPC_SP_FOR(i, n) { x[i] = W.x[i]; }
  QPC_SP_FOR(b, m) { y[perm[b]] = yp[b]; }
  QPC_SP_T0 { if (iters_out) iters_out[0] = iters; }
  QPC_SP_SYNC // (the next QP of this workgroup reuses LDS)
  QPC_SP_STAMP(W, 12);
  return status;
}

} // namespace qpc
