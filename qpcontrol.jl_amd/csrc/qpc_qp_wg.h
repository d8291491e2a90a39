// qpc_qp_wg.h -- L0 of the C ABI on a WORKGROUP: one dense convex QP in OSQP problem form
//       min 1/2 x'Px + q'x   s.t.   l <= A x <= u
// solved by the 512 threads of one workgroup with its working matrices in LDS -- what makes the solver-level plug of the reference
// (`Parametron.Model(optimizer)`, /root/reference/src/lowlevel/momentum.jl:15-16,27,58; `QPCOptimizer` in julia/QPControlHIPMOI.jl)
// usable at B = 1: the one-lane-per-QP form of qpc_qp_dense.h needs 0.57 s for the 143 x 178 Atlas problem (measured, MI355X); the oracle's
// OSQP on one CPU core: 0.15 ms warm-started from the same problem's solution, 0.27 - 0.80 ms cold at the notebook's tolerances,
// 0.44 - 3.4 ms at the test suite's (bench.py cpu_baseline.l0_fixture_osqp measures them in every run).  Since round 5 this DENSE form
// stands behind the dense entry qpc_qp_solve_batch and takes what the sparse form (qpc_qp_sp.h) leaves of a prepared problem.
//
// Same method as qpc_qp_dense.h (range-space active set on G = A H^-1 A', primal-dual multi-change sweeps, Goldfarb-Idnani, iterative
// refinement on the true KKT residuals), for the problems Parametron actually builds for this controller: a DIAGONAL P (regularize!,
// contact weights, scalar task weights: 63 entries for Atlas) and a sparse A given densely (1 004 of 25 454 entries).
//
//   * ONE elimination does the set-up.  The rows are ordered equalities first (E, ne rows: they are in every working set), the others
//     after (I, mi rows), and the right-hand side g0 - b rides along as the last ROW of the matrix S = [G ; rhs'] (packed lower
//     triangle, column-major, in LDS).  Eliminating the first ne columns (LDL' without square roots, in panels of 8 columns: the
//     panel's diagonal block factored by every thread in registers, the trailing triangle updated as 16 x 16 tiles on the matrix
//     core, v_mfma_f64_16x16x4_f64) leaves, in place: L11 D (the factor of G_EE), the
//     rows G_IE L11^-T of the inequality rows, the Schur complement C = G_II - G_IE G_EE^-1 G_EI in the trailing block,
//     zE = L11^-1 (g0_E - b_E) and a0 = A x at the equality-constrained minimiser in the last row.
//   * The working-set iterations live in the space of the inequality rows only: a working set W factors C_WW (its own small packed
//     triangle, the same elimination routine), A x = a0 - C[:, W] y_W.  The equality multipliers are recovered once, at the end.
//   * Two ways in.  A PREPARED problem (qp_solve_wg_prep: qpc_qp_setup / qpc_qp_solve, the optimizer's life cycle) brings the
//     pattern's tables, built once on the host (qpc_qp_prep.h): compressed columns and rows, and for every structurally nonzero G_rs
//     the terms it sums over -- a thread per entry, a fixed order, no atomics; the working set can start from an earlier call's
//     multipliers.  DENSE matrices (qp_solve_wg) are compressed by columns on the way in (a thread per column: coalesced) and
//     G = sum_c h_c a_c a_c' is accumulated column by column with LDS adds; every wave walks every column and adds only into the
//     entries it owns, so the summation order is fixed.  Products with A' walk the compressed columns; products with A the
//     compressed rows (prepared) or its transpose by row POSITION (dense entry: global workspace, coalesced).
//   * The workgroup runs two waves per SIMD: little hides a load but the code itself, so every loop over global or LDS data is
//     written with its loads up front, from addresses that are always valid (a conditional load makes the compiler wait for every
//     load by itself), and per-wave work is reached by a scalar step, never by walking and skipping.
// Anything else -- a dense or matrix-weighted P, a working set beyond QPC_WG_KW inequality rows -- is left to the one-lane kernel: the
// workgroup reports QPC_WG_FALLBACK and the caller runs qp_solve_dense on that QP.  Both are gfx950 device code; neither is a
// fallback to anything else.  When S does not fit beside the vectors in LDS (m beyond ~180) it lives in the global workspace
// (L2-resident, slower, same code).
//
// Written as parallel loops (QPC_WG_FOR: strided over the workgroup; QPC_WG_FOR_WAVE / lanes: a wave per column, lanes down the
// column) separated by barriers, so that the same source runs serially on the host inside the test emulation (tests/emul): one
// "thread", barriers are no-ops.
#pragma once
#include "../../include/qpc.h"
#include "qpc_qp_view.h"

#ifndef QPC_FN
#error "include qpc_device.h (or the emulator's definitions) first"
#endif

#define QPC_WG_FALLBACK (-1) // status: not solved here, run the one-lane kernel
#define QPC_WG_KW 128        // inequality rows of a working set (its factor: (k + 1)(k + 2) / 2 doubles, sized by min(mi, this))
#define QPC_WG_DIM 256       // n, m <= this
#define QPC_WG_LDS_DOUBLES 20480 // what the kernel asks for: the 160 KB of a CU
#define QPC_WG_NB 8          // columns per elimination panel
#define QPC_WG_CB 4          // adjacent trailing columns a wave updates together
#define QPC_WG_THREADS 512   // threads of the workgroup (device)

#if defined(__HIP_DEVICE_COMPILE__)
#define QPC_WG_NT QPC_WG_THREADS // two waves per SIMD: the second one hides part of the first one's LDS latency
#define QPC_WG_NWAVE (QPC_WG_THREADS / 64)
#define QPC_WG_NLANE 64
#define QPC_WG_TID ((int)threadIdx.x)
#define QPC_WG_SYNC __syncthreads();
#define QPC_WG_ADD(ptr, v) atomicAdd((ptr), (v)) // ds_add_f64 / global_atomic_add_f64, result unused: fire and forget, in program order
#else
#define QPC_WG_NT 1
#define QPC_WG_NWAVE 1
#define QPC_WG_NLANE 1
#define QPC_WG_TID 0
#define QPC_WG_SYNC
#define QPC_WG_ADD(ptr, v) (*(ptr) += (v))
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define QPC_WG_WAVE __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) // (told to be a scalar: loops over "my wave's" items branch instead of masking)
#else
#define QPC_WG_WAVE (QPC_WG_TID / QPC_WG_NLANE)
#endif
#define QPC_WG_LANE (QPC_WG_TID % QPC_WG_NLANE)
#define QPC_WG_FOR(i, n) for (int i = QPC_WG_TID; i < (n); i += QPC_WG_NT)
#define QPC_WG_FOR_WAVE(i, n) for (int i = QPC_WG_WAVE; i < (n); i += QPC_WG_NWAVE)
#define QPC_WG_FOR_LANE(i, n) for (int i = QPC_WG_LANE; i < (n); i += QPC_WG_NLANE)
#define QPC_WG_T0 if (QPC_WG_TID == 0)
#if defined(QPC_WG_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define QPC_WG_STAMP(W, k) do { if (threadIdx.x == 0 && (k) < 16) (W).stamps[k] = wall_clock64(); } while (0)
#else
#define QPC_WG_STAMP(W, k) do { } while (0)
#endif

namespace qpc {

#if defined(QPC_WG_PROF) && defined(__HIPCC__)
__device__ unsigned long long qpc_wg_acc[8]; // (profiling build: time inside the elimination, block 0 only)
#endif
#if defined(QPC_WG_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define QPC_WG_ACC_T0() const unsigned long long acc_t0__ = wall_clock64()
#define QPC_WG_ACC(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) qpc_wg_acc[k] += wall_clock64() - acc_t0__; } while (0)
#else
#define QPC_WG_ACC_T0() do { } while (0)
#define QPC_WG_ACC(k) do { } while (0)
#endif

// LDS (or host scratch) handed to qp_solve_wg, in doubles
#ifdef __HIPCC__
__host__ __device__
#endif
inline int qp_wg_lds_doubles(int n, int m) {
  (void)n; (void)m;
  return QPC_WG_LDS_DOUBLES;
}

// column-major packed lower triangle of an N1 x N1 matrix: column c holds its rows c .. N1-1
QPC_FN int wg_cs(int c, int N1) { return c * N1 - (c * (c - 1)) / 2; }
QPC_FN int wg_sidx(int i, int c, int N1) { return wg_cs(c, N1) + i - c; } // i >= c
QPC_FN int wg_sym(int a, int b, int N1) { return a >= b ? wg_sidx(a, b, N1) : wg_sidx(b, a, N1); }

// max over the workgroup of a per-thread value (every thread gets the result).  red: QPC_WG_NWAVE doubles of LDS.
QPC_FN double wg_reduce_max(double v, double *red) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off, 64); v = (o > v || !(o == o)) ? o : v; }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = red[0];
  for (int w = 1; w < QPC_WG_NWAVE; w++) { const double o = red[w]; r = (o > r || !(o == o)) ? o : r; }
  __syncthreads();
  return r;
#else
  (void)red;
  return v;
#endif
}

// S[tgt + e] -= S[src + e] * l for e < len: the lanes of the calling wave, four chunks in flight.  Branch-free up to the stores (a
// chunk past the end re-reads the first one): a conditional load makes the compiler wait for every load by itself.
QPC_FN void wg_col_axpy(double *S, int tgt, int src, int len, double l) {
  for (int e0 = QPC_WG_LANE; e0 < len; e0 += 4 * QPC_WG_NLANE) {
    const int e1 = e0 + QPC_WG_NLANE, e2 = e0 + 2 * QPC_WG_NLANE, e3 = e0 + 3 * QPC_WG_NLANE;
    const bool h1 = e1 < len, h2 = e2 < len, h3 = e3 < len;
    const int c1 = h1 ? e1 : e0, c2 = h2 ? e2 : e0, c3 = h3 ? e3 : e0;
    const double t0 = S[tgt + e0], s0 = S[src + e0], t1 = S[tgt + c1], s1 = S[src + c1];
    const double t2 = S[tgt + c2], s2 = S[src + c2], t3 = S[tgt + c3], s3 = S[src + c3];
    const double r0 = t0 - s0 * l, r1 = t1 - s1 * l, r2 = t2 - s2 * l, r3 = t3 - s3 * l;
    S[tgt + e0] = r0;
    if (h1) S[tgt + e1] = r1;
    if (h2) S[tgt + e2] = r2;
    if (h3) S[tgt + e3] = r3;
  }
}

// One chunk (CH = 1) or two chunks (CH = 2) of 64 rows of a block of QPC_WG_CB adjacent target columns cb .. cb+3 against the NB
// columns of the panel at jp: the panel's values of a row are loaded once and serve the four columns.  Straight-line code: the
// loads of an entry that does not exist (above the diagonal, beyond the block) go to an address that does, its store is skipped.
template <int CH>
QPC_FN void wg_trail_rows(double *S, int e0, int len, const int *src, const int *tgt, int ncol, const double (*lc)[QPC_WG_NB]) {
  int e[CH];
  bool h[CH];
  double a[CH][QPC_WG_NB], t[CH][QPC_WG_CB];
  int idx[CH][QPC_WG_CB];
  bool ok[CH][QPC_WG_CB];
#pragma unroll
  for (int ch = 0; ch < CH; ch++) {
    const int ee = e0 + ch * QPC_WG_NLANE;
    h[ch] = ee < len;
    e[ch] = h[ch] ? ee : e0;
#pragma unroll
    for (int jj = 0; jj < QPC_WG_NB; jj++) a[ch][jj] = S[src[jj] + e[ch]];
#pragma unroll
    for (int cc = 0; cc < QPC_WG_CB; cc++) {
      ok[ch][cc] = h[ch] && cc < ncol && e[ch] >= cc;
      idx[ch][cc] = ok[ch][cc] ? tgt[cc] + e[ch] - cc : tgt[0] + e[ch];
      t[ch][cc] = S[idx[ch][cc]];
    }
  }
#pragma unroll
  for (int ch = 0; ch < CH; ch++)
#pragma unroll
    for (int cc = 0; cc < QPC_WG_CB; cc++) {
      double v = t[ch][cc];
#pragma unroll
      for (int jj = 0; jj < QPC_WG_NB; jj++) v -= a[ch][jj] * lc[cc][jj];
      t[ch][cc] = v;
    }
  // every store unconditional (a conditional one makes the compiler sink the arithmetic into a branch per entry): an entry that does
  // not exist was read from the block's FIRST column at this row (chunk 0's row for a chunk past the end) and gets that entry's new
  // value again -- the same bits, whichever of the two stores lands last
#pragma unroll
  for (int ch = 0; ch < CH; ch++)
#pragma unroll
    for (int cc = 0; cc < QPC_WG_CB; cc++) S[idx[ch][cc]] = ok[ch][cc] ? t[ch][cc] : (h[ch] ? t[ch][0] : t[0][0]);
}

// Eliminate the columns j0 .. j1-1 of S (packed, N1 x N1, the last row is a right-hand side riding along): LDL' without square
// roots -- the columns keep L D (unscaled), dinv[j] = 1 / d_j, or 0 for a pivot below tau * ref[j] (a dependent row: it drops out of
// everything, the "infinite pivot" of qpc_qp_dense.h).  Panels of QPC_WG_NB columns: the panel itself costs two barriers (below), then
// the trailing columns take the whole panel at once, in blocks of QPC_WG_CB adjacent columns per wave (a block from the front with one from the back: every
// pair has about the same length), lanes down the rows: 8 + 4 loads and 4 stores for 32 multiply-adds -- the LDS pipe is what
// bounds this routine.  A short last panel and a short last block are padded with zero factors.
// Barrier-synchronised on entry and on exit.
QPC_FN void wg_ldl_eliminate(double *S, int N1, int j0, int j1, const double *ref, double tau, double *dinv) {
  for (int jp = j0; jp < j1; jp += QPC_WG_NB) {
    const int nb = j1 - jp < QPC_WG_NB ? j1 - jp : QPC_WG_NB;
    // ---- the panel.  Its nb x nb diagonal block is factored by EVERY thread in registers (36 broadcast loads, a chain of pivots):
    //      nobody waits for anybody, and the rows below the block need no barrier between the panel's columns either -- a row's
    //      nb values are a little triangular solve against the block's multipliers, which the thread already holds.  Same
    //      arithmetic in the same order as eliminating the panel column by column.
    QPC_WG_ACC_T0();
    int pc[QPC_WG_NB];
    double blk[QPC_WG_NB][QPC_WG_NB], mul[QPC_WG_NB][QPC_WG_NB], dv[QPC_WG_NB];
#pragma unroll
    for (int c = 0; c < QPC_WG_NB; c++) {
      const int jc = c < nb ? jp + c : jp;
      pc[c] = wg_cs(jc, N1) - jc; // S[i][jc] = S[pc + i]
#pragma unroll
      for (int r = c; r < QPC_WG_NB; r++) {
        const double v = S[pc[c] + jp + (r < nb ? r : 0)]; // (always a load from a valid address, the choice afterwards)
        blk[r][c] = (r < nb && c < nb) ? v : (r == c ? 1.0 : 0.0);
      }
    }
    double rfv[QPC_WG_NB];
#pragma unroll
    for (int c = 0; c < QPC_WG_NB; c++) rfv[c] = ref[c < nb ? jp + c : jp];
    QPC_WG_SYNC // (everybody has the block: thread 0 may overwrite it)
#pragma unroll
    for (int j = 0; j < QPC_WG_NB; j++) {
      const double d = blk[j][j], rf = rfv[j];
      const bool ok = j < nb && d > tau * (rf > 1e-300 ? rf : 1e-300);
      dv[j] = ok ? 1.0 / d : 0.0;
#pragma unroll
      for (int c = j + 1; c < QPC_WG_NB; c++) {
        const double lc = blk[c][j] * dv[j];
        mul[c][j] = lc;
#pragma unroll
        for (int r = c; r < QPC_WG_NB; r++) blk[r][c] -= blk[r][j] * lc;
      }
    }
    QPC_WG_T0 {
#pragma unroll
      for (int c = 0; c < QPC_WG_NB; c++) {
        if (c < nb) {
          dinv[jp + c] = dv[c];
#pragma unroll
          for (int r = c; r < QPC_WG_NB; r++) if (r < nb) S[pc[c] + jp + r] = blk[r][c];
        }
      }
    }
    QPC_WG_FOR(t, N1 - jp - nb) { // the rows below the block (the right-hand-side row among them): u L1' = a
      const int i = jp + nb + t;
      double a[QPC_WG_NB];
#pragma unroll
      for (int c = 0; c < QPC_WG_NB; c++) a[c] = S[pc[c] + i]; // (a column beyond nb: pc of the first one, value unused)
#pragma unroll
      for (int j = 0; j < QPC_WG_NB; j++) {
#pragma unroll
        for (int c = j + 1; c < QPC_WG_NB; c++) a[c] -= a[j] * mul[c][j];
      }
#pragma unroll
      for (int c = 1; c < QPC_WG_NB; c++) if (c < nb) S[pc[c] + i] = a[c];
    }
    QPC_WG_SYNC
    QPC_WG_ACC(N1 > 100 ? 0 : 2);
    const int c0 = jp + nb, R = N1 - 1 - c0; // trailing columns c0 .. N1-2 (column N1-1 would be the corner of the right-hand-side row)
    if (R > 0) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(QPC_WG_NO_MFMA)
      // ---- on the device the update is what it looks like, a GEMM  C -= (panel rows) (panel rows of the columns)' : 16 x 16 tiles of the
      //      trailing triangle on the matrix core, two v_mfma_f64_16x16x4_f64 per tile and panel (k = 8).  A wave takes a tile:
      //      lane l supplies A[i = l & 15][k = l >> 4] and B[k][j = l & 15] and holds C[(l >> 4) + 4 r][l & 15], r = 0 .. 3 (the f64
      //      instruction's own C/D map).  Tiles on the diagonal and at the edges compute all 256 entries and store the ones that
      //      exist (rows and columns beyond the matrix read a clamped address).  ~60 instructions per tile instead of ~1 400 on the vector
      //      unit (which the host emulation below still runs: same sums, another order).
      {
        typedef double wg_v4d __attribute__((ext_vector_type(4)));
        const int lane = threadIdx.x & 63, kq = lane >> 4, ll = lane & 15;
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (a scalar: the tile loop below branches, it does not mask)
        const int k0 = kq < nb ? kq : 0, k1 = kq + 4 < nb ? kq + 4 : 0;          // this lane's two panel columns (k and k + 4)
        const int p0 = wg_cs(jp + k0, N1) - (jp + k0), p1 = wg_cs(jp + k1, N1) - (jp + k1);
        const double d0 = kq < nb ? dinv[jp + k0] : 0.0, d1 = kq + 4 < nb ? dinv[jp + k1] : 0.0; // (written before the barrier above)
        const int ntr = (N1 - c0 + 15) / 16, ntc = (R + 15) / 16, cend = c0 + R;
        // the tiles of the lower triangle, row of tiles by row of tiles (row ti has min(ti + 1, ntc) of them); a wave takes every
        // QPC_WG_NWAVE-th one and steps straight to its next (walking over the others' tiles cost more than the tiles themselves)
#if defined(QPC_WG_PROF)
        const unsigned long long tl0__ = wall_clock64();
#endif
        int ti = 0, tj = wave;
        while (true) {
          while (ti < ntr && tj >= (ti + 1 < ntc ? ti + 1 : ntc)) { tj -= ti + 1 < ntc ? ti + 1 : ntc; ti++; }
          if (ti >= ntr) break;
          {
            const int R0 = c0 + 16 * ti, C0 = c0 + 16 * tj;
            const int rA = R0 + ll < N1 ? R0 + ll : N1 - 1, rB = C0 + ll < N1 ? C0 + ll : N1 - 1;
            const double a0 = -S[p0 + rA], a1 = -S[p1 + rA], b0 = S[p0 + rB] * d0, b1 = S[p1 + rB] * d1;
            const int col = C0 + ll, colc = col < cend ? col : cend - 1;
            const int cbase = wg_cs(colc, N1) - colc;
            int addr[4];
            bool okc[4];
            wg_v4d acc;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int row = R0 + kq + 4 * r;
              okc[r] = col < cend && row < N1 && row >= col;
              addr[r] = cbase + (okc[r] ? row : colc); // (an entry that does not exist: the column's diagonal, value unused)
              acc[r] = S[addr[r]];
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) if (okc[r]) S[addr[r]] = acc[r];
          }
          tj += QPC_WG_NWAVE;
        }
#if defined(QPC_WG_PROF)
        if (threadIdx.x == 0 && blockIdx.x == 0 && N1 > 100) { qpc_wg_acc[4] += wall_clock64() - tl0__; }
#endif
      }
      QPC_WG_SYNC
      QPC_WG_ACC(N1 > 100 ? 1 : 3);
      continue;
#endif
      const int nblk = (R + QPC_WG_CB - 1) / QPC_WG_CB;
#pragma unroll
      for (int jj = 0; jj < QPC_WG_NB; jj++) dv[jj] = jj < nb ? dv[jj] : 0.0; // (the padding columns of a short panel)
      QPC_WG_FOR_WAVE(p, (nblk + 1) / 2) {
        for (int half = 0; half < 2; half++) {
          const int blk = half == 0 ? p : nblk - 1 - p;
          if (half == 1 && blk == p) break; // the middle block of an odd count
          const int cb = c0 + blk * QPC_WG_CB, ncol = c0 + R - cb < QPC_WG_CB ? c0 + R - cb : QPC_WG_CB, len = N1 - cb;
          double lc[QPC_WG_CB][QPC_WG_NB];
          int src[QPC_WG_NB], tgt[QPC_WG_CB];
#pragma unroll
          for (int cc = 0; cc < QPC_WG_CB; cc++) {
            const int c = cc < ncol ? cb + cc : cb;
            tgt[cc] = wg_cs(c, N1);
#pragma unroll
            for (int jj = 0; jj < QPC_WG_NB; jj++) {
              const double v = S[pc[jj] + c] * dv[jj];
              lc[cc][jj] = cc < ncol ? v : 0.0;
            }
          }
#pragma unroll
          for (int jj = 0; jj < QPC_WG_NB; jj++) src[jj] = pc[jj] + cb; // row cb + e of panel column jj: src + e
          for (int e0 = QPC_WG_LANE; e0 < len; e0 += 2 * QPC_WG_NLANE) {
            if (e0 - QPC_WG_LANE + QPC_WG_NLANE < len) wg_trail_rows<2>(S, e0, len, src, tgt, ncol, lc); // (the same for every lane)
            else wg_trail_rows<1>(S, e0, len, src, tgt, ncol, lc);
          }
        }
      }
      QPC_WG_SYNC
      QPC_WG_ACC(N1 > 100 ? 1 : 3);
    }
  }
}
// t <- L^-1 t for the leading nc columns (L unit lower: L_ij = S_ij dinv_j), four columns per barrier: every thread solves the 4 x 4
// diagonal block itself (ten broadcast loads, a chain of six multiply-adds), then takes the four columns out of its rows below.
// Thread 0 stores the block's result after the barrier (the others may still be reading t[j .. j+3] before it).
// Barrier-synchronised on entry and exit.
QPC_FN void wg_ldl_forward(const double *S, int N1, int nc, const double *dinv, double *t) {
  for (int j = 0; j < nc; j += 4) {
    const int nb = nc - j < 4 ? nc - j : 4;
    const int j1 = nb > 1 ? j + 1 : j, j2 = nb > 2 ? j + 2 : j, j3 = nb > 3 ? j + 3 : j; // (clamped: a short last block re-reads column j)
    const int c0 = wg_cs(j, N1), c1 = wg_cs(j1, N1), c2 = wg_cs(j2, N1);
    const double t0 = t[j], t1 = t[j1], t2 = t[j2], t3 = t[j3];
    const double d0 = dinv[j], d1 = nb > 1 ? dinv[j1] : 0.0, d2 = nb > 2 ? dinv[j2] : 0.0, d3 = nb > 3 ? dinv[j3] : 0.0;
    const double l10 = S[c0 + (j1 - j)], l20 = S[c0 + (j2 - j)], l30 = S[c0 + (j3 - j)];
    const double l21 = S[c1 + (j2 - j1)], l31 = S[c1 + (j3 - j1)], l32 = S[c2 + (j3 - j2)];
    const double z0 = t0, f0 = z0 * d0;
    double z1 = t1; z1 -= l10 * f0;
    const double f1 = z1 * d1;
    double z2 = t2; z2 -= l20 * f0; z2 -= l21 * f1;
    const double f2 = z2 * d2;
    double z3 = t3; z3 -= l30 * f0; z3 -= l31 * f1; z3 -= l32 * f2;
    const double f3 = z3 * d3;
    const int c3 = wg_cs(j3, N1);
    QPC_WG_FOR(e, nc - j - nb) {
      const int i = j + nb + e;
      const double a0 = S[c0 + (i - j)], a1 = S[c1 + (i - j1)], a2 = S[c2 + (i - j2)], a3 = S[c3 + (i - j3)];
      double v = t[i];
      v -= a0 * f0; v -= a1 * f1; v -= a2 * f2; v -= a3 * f3; // (f of a clamped column is 0)
      t[i] = v;
    }
    QPC_WG_SYNC
    QPC_WG_T0 { if (nb > 1) t[j + 1] = z1; if (nb > 2) t[j + 2] = z2; if (nb > 3) t[j + 3] = z3; }
  }
  QPC_WG_SYNC
}
// out <- L^-T D^-1 t for the leading nc columns; t is used up (out != t).  Four rows per barrier, like wg_ldl_forward.
// Barrier-synchronised on entry and exit.
QPC_FN void wg_ldl_back(const double *S, int N1, int nc, const double *dinv, double *t, double *out) {
  for (int i0 = nc - 1; i0 >= 0; i0 -= 4) {
    const int nb = i0 + 1 < 4 ? i0 + 1 : 4;
    const int i1 = nb > 1 ? i0 - 1 : i0, i2 = nb > 2 ? i0 - 2 : i0, i3 = nb > 3 ? i0 - 3 : i0; // (clamped)
    const double t0 = t[i0], t1 = t[i1], t2 = t[i2], t3 = t[i3];
    const double d0 = dinv[i0], d1 = dinv[i1], d2 = dinv[i2], d3 = dinv[i3];
    const int c1 = wg_cs(i1, N1), c2 = wg_cs(i2, N1), c3 = wg_cs(i3, N1);
    const double u01 = S[c1 + (i0 - i1)], u02 = S[c2 + (i0 - i2)], u03 = S[c3 + (i0 - i3)]; // S[i0][i1], S[i0][i2], S[i0][i3]
    const double u12 = S[c2 + (i1 - i2)], u13 = S[c3 + (i1 - i3)], u23 = S[c3 + (i2 - i3)];
    const double y0 = t0 * d0;
    double v1 = t1; v1 -= u01 * y0;
    const double y1 = nb > 1 ? v1 * d1 : 0.0;
    double v2 = t2; v2 -= u02 * y0; v2 -= u12 * y1;
    const double y2 = nb > 2 ? v2 * d2 : 0.0;
    double v3 = t3; v3 -= u03 * y0; v3 -= u13 * y1; v3 -= u23 * y2;
    const double y3 = nb > 3 ? v3 * d3 : 0.0;
    QPC_WG_T0 { out[i0] = y0; if (nb > 1) out[i1] = y1; if (nb > 2) out[i2] = y2; if (nb > 3) out[i3] = y3; }
    QPC_WG_FOR(j, i0 + 1 - nb) {
      const int cj = wg_cs(j, N1) - j;
      const double a0 = S[cj + i0], a1 = S[cj + i1], a2 = S[cj + i2], a3 = S[cj + i3];
      double v = t[j];
      v -= a0 * y0; v -= a1 * y1; v -= a2 * y2; v -= a3 * y3; // (y of a clamped row is 0)
      t[j] = v;
    }
    QPC_WG_SYNC
  }
}

struct QpWg {
  int n, m, ne, mi, N1; // N1 = m + 1: dimension of S with its right-hand-side row
  const double *P, *q, *A, *l, *u;
  // global workspace
  double *AT;     // n x m: A' by row position, AT[c * m + pos]
  double *cval;   // nonzeros of A, column by column; crow: the POSITION of the row
  int *crow;
  const double *sval; // the same, staged in LDS during the set-up when it fits (else the global copy)
  const int *srow;
  // LDS
  double *S;      // (LDS, or the global workspace when it does not fit)
  double *LW;     // the working set's factor, (k + 1) x (k + 1) packed
  int *cptr, *perm, *ipos; // column starts; position -> row, row -> position (equality rows first)
  double *hinv, *xu, *x, *r1;              // n
  double *gdiag, *tA;                      // m + 1 (by position): diagonal of G (from the final stage on: y by position); scratch
  double *dinvE, *yE, *tE;                 // ne
  double *wsf, *tm, *ax, *lo, *hi, *a0;    // mi (by inequality position)
  double *yw, *dinvW, *dW, *tW;            // kw + 1
  int *idxW;
  double *sc, *red;
  unsigned long long *stamps;
  // prepared problem (qpc_qp_setup): the pattern's tables instead of the dense arrays (V == nullptr: dense input)
  const QpPrepView *V;
  const double *Pv;  // the caller's P values
  const double *y0;  // warm start: multipliers of an earlier solve by ROW (their signs give the first working set), or nullptr
};

enum { WGS_FLAG = 0, WGS_K, WGS_CHANGES, WGS_T, WGS_P, WGS_VIOL, WGS_SG, WGS_NE, WGS_NNZ, WGS_COUNT };

// t[pos] = sum_c AT[c * m + pos] * w[c]  (a row of A times w; the loads of eight columns in flight)
QPC_FN double wg_row_dot(const QpWg &W, int pos, const double *w) {
  const int n = W.n, m = W.m;
  double t = 0.0;
  if (W.V) { // prepared problem: the compressed row (short)
    const int r = W.perm[pos];
    for (int k = W.V->rptr[r]; k < W.V->rptr[r + 1]; k++) t += W.cval[W.V->ro[k]] * w[W.V->rcol[k]];
    return t;
  }
  int c = 0;
  for (; c + 8 <= n; c += 8) {
    const double *a = W.AT + (long long)c * m + pos;
    const double a0 = a[0], a1 = a[m], a2 = a[2 * m], a3 = a[3 * m], a4 = a[4 * m], a5 = a[5 * m], a6 = a[6 * m], a7 = a[7 * m];
    t += a0 * w[c]; t += a1 * w[c + 1]; t += a2 * w[c + 2]; t += a3 * w[c + 3];
    t += a4 * w[c + 4]; t += a5 * w[c + 5]; t += a6 * w[c + 6]; t += a7 * w[c + 7];
  }
  for (; c < n; c++) t += W.AT[(long long)c * m + pos] * w[c];
  return t;
}
QPC_FN double wg_pdiag(const QpWg &W, int i) {
  if (W.V) { const int k = W.V->pdiag[i]; return k >= 0 ? W.Pv[k] : 0.0; }
  return W.P[(long long)i * W.n + i];
}
// (A' y)_c for y by position (a compressed column: short)
QPC_FN double wg_col_dot(const QpWg &W, int c, const double *ypos) {
  double t = 0.0;
  for (int o = W.cptr[c]; o < W.cptr[c + 1]; o++) t += W.cval[o] * ypos[W.crow[o]];
  return t;
}

// Working set in W.wsf (by inequality position) -> idxW (ascending), factor of C_WW in W.LW.  with_rhs: the right-hand side
// a0_W - b_W rides along and yw <- the multipliers of the working rows.  Returns the number of working inequality rows, or -1 beyond
// what LW holds.
QPC_FN int wg_w_factor(QpWg &W, double tau, bool with_rhs, int kw) {
  const int mi = W.mi, ne = W.ne, N1 = W.N1;
  int k = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  { // one thread per inequality row (mi <= 256): ranks from the waves' ballots
    const int p = threadIdx.x, lane = p & 63, wv = p >> 6;
    const bool f = p < mi && W.wsf[p] != 0.0;
    const unsigned long long mask = __ballot(f);
    int *cnt = (int *)W.red;
    if (lane == 0) cnt[wv] = __popcll(mask);
    __syncthreads();
    int off = 0;
    for (int w = 0; w < QPC_WG_NWAVE; w++) { const int c = cnt[w]; off += w < wv ? c : 0; k += c; }
    if (f) { const int r = off + __popcll(mask & ((1ull << lane) - 1ull)); if (r < kw) W.idxW[r] = p; }
    __syncthreads();
  }
#else
  for (int p = 0; p < mi; p++) if (W.wsf[p] != 0.0) { if (k < kw) W.idxW[k] = p; k++; }
#endif
  if (k > kw) return -1;
  const int K1 = k + 1;
  QPC_WG_FOR_WAVE(b, k) {
    const int pb = W.idxW[b], sb = wg_cs(b, K1);
    QPC_WG_FOR_LANE(e, k - b) { W.LW[sb + e] = W.S[wg_sym(ne + W.idxW[b + e], ne + pb, N1)]; }
  }
  QPC_WG_FOR(a, k) {
    const int pa = W.idxW[a];
    W.tW[a] = W.gdiag[ne + pa];
    W.LW[wg_sidx(k, a, K1)] = with_rhs ? W.a0[pa] - (W.wsf[pa] > 0.0 ? W.hi[pa] : W.lo[pa]) : 0.0;
  }
  QPC_WG_SYNC
  wg_ldl_eliminate(W.LW, K1, 0, k, W.tW, tau, W.dinvW);
  if (with_rhs) {
    QPC_WG_FOR(a, k) { W.tW[a] = W.LW[wg_sidx(k, a, K1)]; }
    QPC_WG_SYNC
    wg_ldl_back(W.LW, K1, k, W.dinvW, W.tW, W.yw);
  }
  return k;
}
// EQP of the working set: multipliers of its inequality rows -> yw, ax = A x for every inequality row
QPC_FN int wg_solve_ws(QpWg &W, double tau, int kw) {
  const int k = wg_w_factor(W, tau, true, kw);
  if (k < 0) return -1;
  QPC_WG_FOR(p, W.mi) {
    double t = W.a0[p];
    for (int a = 0; a < k; a++) t -= W.S[wg_sym(W.ne + p, W.ne + W.idxW[a], W.N1)] * W.yw[a];
    W.ax[p] = t;
  }
  QPC_WG_SYNC
  return k;
}
// (outE, outW) <- K^-1 (tE, tW) with K = G on (E, W): the block factor [L11 ; rows of the working inequality rows ; LW].
// in: W.tE (ne), W.tW (k); outE may be W.tE.  W.tA is scratch.
QPC_FN void wg_block_solve(QpWg &W, int k, double *outE, double *outW) {
  const int ne = W.ne, N1 = W.N1, K1 = k + 1;
  wg_ldl_forward(W.S, N1, ne, W.dinvE, W.tE);
  QPC_WG_FOR(a, k) {
    const int i = ne + W.idxW[a];
    double t = W.tW[a];
    for (int j = 0; j < ne; j++) t -= W.S[wg_sidx(i, j, N1)] * W.dinvE[j] * W.tE[j];
    W.tW[a] = t;
  }
  QPC_WG_SYNC
  wg_ldl_forward(W.LW, K1, k, W.dinvW, W.tW);
  wg_ldl_back(W.LW, K1, k, W.dinvW, W.tW, outW);
  QPC_WG_FOR(j, ne) {
    double t = W.tE[j];
    for (int a = 0; a < k; a++) t -= W.S[wg_sidx(ne + W.idxW[a], j, N1)] * outW[a];
    W.tA[j] = t;
  }
  QPC_WG_SYNC
  wg_ldl_back(W.S, N1, ne, W.dinvE, W.tA, outE);
}

// the solver proper, S placed by the caller (LDS or global)
QPC_FN int qp_solve_wg_core(QpWg &W, const qpc_settings &st, double bscale, int kw, double *x, double *y, int32_t *iters_out) {
  const double INF = 1e19;
  const int n = W.n, m = W.m, ne = W.ne, mi = W.mi, N1 = W.N1;
  // ---- S = [G ; rhs'] by position, column by column of A: G += h_c a_c a_c', rhs += xu_c a_c'.  Every wave walks every column and
  //      adds only into the entries it owns (by the smaller position): a fixed summation order without a barrier.
  QPC_WG_FOR(e, (m + 1) * (m + 2) / 2) { W.S[e] = 0.0; }
  QPC_WG_SYNC
  if (W.V) { // prepared problem: a thread per structurally nonzero G_rs, its terms in the order the host laid down; a thread per row for g0
    const QpPrepView &V = *W.V;
    QPC_WG_FOR(pi, V.npairs) {
      double acc = 0.0;
      for (int t = V.pptr[pi]; t < V.pptr[pi + 1]; t++) acc += W.sval[V.ta[t]] * W.sval[V.tb[t]] * W.hinv[V.tc[t]];
      W.S[wg_sym(W.ipos[V.pr[pi]], W.ipos[V.ps[pi]], N1)] = acc;
    }
    QPC_WG_FOR(r, m) {
      double t = 0.0;
      for (int k = V.rptr[r]; k < V.rptr[r + 1]; k++) t += W.sval[V.ro[k]] * W.xu[V.rcol[k]];
      W.S[wg_sidx(m, W.ipos[r], N1)] = t;
    }
  } else
  for (int cg = 0; cg < n; cg += 4) { // four columns at a time: all their loads first, then their adds, in column order
    int o0[4], kk[4];
    double hh[4], xx[4];
    float rk[4];
    int k2max = 0, kmax = 0;
#pragma unroll
    for (int u4 = 0; u4 < 4; u4++) {
      const bool in = cg + u4 < n;
      const int c = in ? cg + u4 : cg;
      o0[u4] = W.cptr[c];
      kk[u4] = in ? W.cptr[c + 1] - o0[u4] : 0;
      hh[u4] = W.hinv[c]; xx[u4] = W.xu[c];
      rk[u4] = 1.0f / (float)(kk[u4] > 0 ? kk[u4] : 1);
      k2max = kk[u4] * kk[u4] > k2max ? kk[u4] * kk[u4] : k2max;
      kmax = kk[u4] > kmax ? kk[u4] : kmax;
    }
    QPC_WG_FOR_LANE(e, k2max) {
      int pa[4], pb[4];
      double va[4], vb[4];
      bool act[4];
#pragma unroll
      for (int u4 = 0; u4 < 4; u4++) {
        const int k = kk[u4];
        const int i1 = (int)(((float)e + 0.5f) * rk[u4]), i2 = e - i1 * k;
        act[u4] = e < k * k && i2 <= i1;
        const int oa = act[u4] ? o0[u4] + i1 : 0, ob = act[u4] ? o0[u4] + i2 : 0;
        pa[u4] = W.srow[oa]; pb[u4] = W.srow[ob]; va[u4] = W.sval[oa]; vb[u4] = W.sval[ob];
      }
#pragma unroll
      for (int u4 = 0; u4 < 4; u4++) {
        const int hi = pa[u4] > pb[u4] ? pa[u4] : pb[u4], lo = pa[u4] > pb[u4] ? pb[u4] : pa[u4];
        if (act[u4] && lo % QPC_WG_NWAVE == QPC_WG_WAVE) QPC_WG_ADD(&W.S[wg_sidx(hi, lo, N1)], va[u4] * vb[u4] * hh[u4]);
      }
    }
    QPC_WG_FOR_LANE(e, kmax) {
      int pa[4];
      double va[4];
#pragma unroll
      for (int u4 = 0; u4 < 4; u4++) {
        const int oa = e < kk[u4] ? o0[u4] + e : 0;
        pa[u4] = W.srow[oa]; va[u4] = W.sval[oa];
      }
#pragma unroll
      for (int u4 = 0; u4 < 4; u4++)
        if (e < kk[u4] && pa[u4] % QPC_WG_NWAVE == QPC_WG_WAVE) QPC_WG_ADD(&W.S[wg_sidx(m, pa[u4], N1)], va[u4] * xx[u4]);
    }
  }
  QPC_WG_SYNC
  QPC_WG_FOR(b, m) { // the right-hand-side row: g0_E - b_E on the equality rows, g0 on the others
    if (b < ne) W.S[wg_sidx(m, b, N1)] -= W.l[W.perm[b]];
    W.gdiag[b] = W.S[wg_cs(b, N1)];
  }
  QPC_WG_SYNC
  QPC_WG_STAMP(W, 2);
  wg_ldl_eliminate(W.S, N1, 0, ne, W.gdiag, st.eps_dual_reg, W.dinvE);
  QPC_WG_STAMP(W, 3);
  QPC_WG_FOR(p, mi) {
    W.a0[p] = W.S[wg_sidx(m, ne + p, N1)];
    double w = 0.0;
    if (W.y0) { // warm start: the rows an earlier solution held at a bound
      const double y0 = W.y0[W.perm[ne + p]];
      w = (y0 > 0.0 && W.hi[p] < INF) ? 1.0 : ((y0 < 0.0 && W.lo[p] > -INF) ? -1.0 : 0.0);
    }
    W.wsf[p] = w;
  }
  QPC_WG_SYNC
  int status = QPC_STATUS_OPTIMAL, iters = 0;
  const int max_it = st.max_active_set_iters > 0 ? st.max_active_set_iters : 200;
  int nw = wg_solve_ws(W, st.eps_dual_reg, kw); // the equality-only working set (or the warm start's)
  if (nw < 0) { // (a warm start beyond what LW holds: start cold)
    QPC_WG_FOR(p, mi) { W.wsf[p] = 0.0; }
    QPC_WG_SYNC
    nw = wg_solve_ws(W, st.eps_dual_reg, kw);
  }
  bool converged = false;
  const int max_pdas = 4; // (measured on the committed Atlas problems: 4 / 6 / 8 / 12 sweeps before Goldfarb-Idnani takes over -> 31 / 47 / 39 / 49 iterations in total, the worst problem 8 / 21 / 13 / 23; a sweep and a Goldfarb-Idnani step cost about the same)
  // ---- primal-dual multi-change sweeps (qpc_qp_dense.h phase A), on the inequality rows
  for (; iters < max_pdas && iters < max_it; iters++) {
    double a_y = 1.0, a_x = 1.0;
    QPC_WG_FOR(a, nw) { a_y = QPC_FABS(W.yw[a]) > a_y ? QPC_FABS(W.yw[a]) : a_y; }
    QPC_WG_FOR(p, mi) { a_x = QPC_FABS(W.ax[p]) > a_x ? QPC_FABS(W.ax[p]) : a_x; W.tm[p] = 0.0; }
    QPC_WG_T0 { W.sc[WGS_CHANGES] = 0.0; }
    const double ysc = wg_reduce_max(a_y, W.red), xsc = wg_reduce_max(a_x, W.red);
    const double toly = st.tol_bound * ysc, tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
    QPC_WG_FOR(a, nw) { // wrong-signed multipliers and dependent (dropped) rows leave
      const int p = W.idxW[a];
      const double w = W.wsf[p];
      const double viol = w < 0.0 ? W.yw[a] : -W.yw[a];
      if (viol > toly || W.dinvW[a] == 0.0) W.sc[WGS_CHANGES] = 1.0;
      else W.tm[p] = w;
    }
    QPC_WG_SYNC
    QPC_WG_FOR(p, mi) {
      if (W.wsf[p] == 0.0) {
        const double lo = W.lo[p] > -INF ? W.lo[p] - W.ax[p] : -1.0, hi = W.hi[p] < INF ? W.ax[p] - W.hi[p] : -1.0;
        if (lo > tolx && lo >= hi) { W.tm[p] = -1.0; W.sc[WGS_CHANGES] = 1.0; }
        else if (hi > tolx) { W.tm[p] = 1.0; W.sc[WGS_CHANGES] = 1.0; }
      }
    }
    QPC_WG_SYNC
    if (W.sc[WGS_CHANGES] == 0.0) { converged = true; break; }
    QPC_WG_FOR(p, mi) { W.wsf[p] = W.tm[p]; }
    QPC_WG_SYNC
    nw = wg_solve_ws(W, st.eps_dual_reg, kw);
    if (nw < 0) return QPC_WG_FALLBACK;
  }
  QPC_WG_STAMP(W, 4);
  // ---- Goldfarb-Idnani dual active set (finite; qpc_qp_dense.h phase B).  It may start from any working set whose EQP has
  //      multipliers of the right signs (the minimiser subject to those rows AS INEQUALITIES): not from the equality-only set, which
  //      throws the sweeps' work away, but from what is left of their last set once the rows with wrong-signed multipliers have
  //      been dropped (repeatedly: a drop can turn another sign).  The scalar logic of an iteration (most violated row, ratio test,
  //      step) is thread 0's, the factorizations the workgroup's.
  if (!converged) {
    for (int pass = 0; pass <= mi; pass++) {
      double a_y = 1.0;
      QPC_WG_FOR(a, nw) { a_y = QPC_FABS(W.yw[a]) > a_y ? QPC_FABS(W.yw[a]) : a_y; }
      QPC_WG_T0 { W.sc[WGS_CHANGES] = 0.0; }
      const double toly = st.tol_bound * wg_reduce_max(a_y, W.red);
      QPC_WG_FOR(a, nw) {
        const int p = W.idxW[a];
        const double viol = W.wsf[p] < 0.0 ? W.yw[a] : -W.yw[a];
        if (viol > toly || W.dinvW[a] == 0.0) { W.tm[p] = 0.0; W.sc[WGS_CHANGES] = 1.0; } else W.tm[p] = W.wsf[p];
      }
      QPC_WG_SYNC
      if (W.sc[WGS_CHANGES] == 0.0) break;
      QPC_WG_FOR(a, nw) { const int p = W.idxW[a]; W.wsf[p] = W.tm[p]; }
      QPC_WG_SYNC
      nw = wg_solve_ws(W, st.eps_dual_reg, kw);
      if (nw < 0) return QPC_WG_FALLBACK;
    }
    for (; iters < max_it; iters++) {
      QPC_WG_T0 {
        double xsc = 1.0;
        for (int p = 0; p < mi; p++) xsc = QPC_FABS(W.ax[p]) > xsc ? QPC_FABS(W.ax[p]) : xsc;
        const double tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
        int pp = -1;
        double viol = tolx, sg = 0.0;
        for (int p = 0; p < mi; p++) { // the most violated row (positions of a kind are in row order: ties go to the first row, like the one-lane form)
          if (W.wsf[p] != 0.0) continue;
          const double lo = W.lo[p] > -INF ? W.lo[p] - W.ax[p] : -1.0, hi = W.hi[p] < INF ? W.ax[p] - W.hi[p] : -1.0;
          if (lo > viol) { viol = lo; pp = p; sg = -1.0; }
          if (hi > viol) { viol = hi; pp = p; sg = 1.0; }
        }
        W.sc[WGS_P] = (double)pp; W.sc[WGS_VIOL] = viol; W.sc[WGS_SG] = sg;
      }
      QPC_WG_SYNC
      const int pr = (int)W.sc[WGS_P];
      if (pr < 0) { converged = true; break; }
      const double sg = W.sc[WGS_SG];
      bool infeasible = false, bail = false;
      int k = nw; // LW is the factor of the current working set (left by wg_solve_ws)
      for (int inner = 0; inner <= m; inner++) {
        // r = (C_WW)^-1 C_Wp : a_p in terms of the working rows (H^-1 metric, equality rows projected out)
        QPC_WG_FOR(a, k) { W.tW[a] = W.S[wg_sym(ne + W.idxW[a], ne + pr, N1)]; }
        QPC_WG_SYNC
        wg_ldl_forward(W.LW, k + 1, k, W.dinvW, W.tW);
        wg_ldl_back(W.LW, k + 1, k, W.dinvW, W.tW, W.dW);
        QPC_WG_T0 {
          double viol = W.sc[WGS_VIOL];
          double d = W.S[wg_cs(ne + pr, N1)];
          for (int a = 0; a < k; a++) d -= W.S[wg_sym(ne + pr, ne + W.idxW[a], N1)] * W.dW[a];
          const bool dependent = !(d > 1e-11 * W.gdiag[ne + pr]);
          double t1 = 1e300; // dual ratio test: y_W(t) = y_W - t sg r must keep the sign of every working inequality
          int kd = -1;
          for (int a = 0; a < k; a++) {
            const double w = W.wsf[W.idxW[a]], sr = sg * W.dW[a];
            if (W.dinvW[a] == 0.0) continue;
            if ((w > 0.0 && sr > 1e-14) || (w < 0.0 && sr < -1e-14)) {
              double t = W.yw[a] / sr;
              t = t > 0.0 ? t : 0.0;
              if (t < t1) { t1 = t; kd = a; }
            }
          }
          int outcome; // 0: full step, p enters; 1: partial step, row kd left; 2: infeasible
          if (dependent && kd < 0) outcome = 2;
          else {
            const double t2 = dependent ? 1e300 : viol / d;
            if (t2 <= t1) outcome = 0;
            else {
              outcome = 1;
              for (int a = 0; a < k; a++) W.yw[a] -= t1 * sg * W.dW[a];
              if (!dependent) viol -= t1 * d;
              W.wsf[W.idxW[kd]] = 0.0;
              for (int a = kd; a + 1 < k; a++) W.yw[a] = W.yw[a + 1];
            }
          }
          W.sc[WGS_VIOL] = viol; W.sc[WGS_T] = (double)outcome;
        }
        QPC_WG_SYNC
        const int outcome = (int)W.sc[WGS_T];
        if (outcome == 2) { infeasible = true; break; }
        if (outcome == 0) break;
        k = wg_w_factor(W, st.eps_dual_reg, false, kw); // a row left: the factor of what stays
        if (k < 0) { bail = true; break; }
      }
      if (bail) return QPC_WG_FALLBACK;
      if (infeasible) { status = QPC_STATUS_INFEASIBLE; break; }
      QPC_WG_T0 { W.wsf[pr] = sg; }
      QPC_WG_SYNC
      nw = wg_solve_ws(W, st.eps_dual_reg, kw);
      if (nw < 0) return QPC_WG_FALLBACK;
    }
  }
  if (!converged && status == QPC_STATUS_OPTIMAL) status = QPC_STATUS_ITERATION_LIMIT;
  QPC_WG_STAMP(W, 5);
  // ---- the multipliers of the equality rows, once:  L11' yE = D^-1 (zE - (rows of W)' yW)
  QPC_WG_FOR(j, ne) {
    double t = W.S[wg_sidx(m, j, N1)];
    for (int a = 0; a < nw; a++) t -= W.S[wg_sidx(ne + W.idxW[a], j, N1)] * W.yw[a];
    W.tA[j] = t;
  }
  QPC_WG_SYNC
  wg_ldl_back(W.S, N1, ne, W.dinvE, W.tA, W.yE);
  QPC_WG_STAMP(W, 6);
  // ---- x = xu - H^-1 A' y, then iterative refinement on the true KKT residuals of the final working set.  yp: y by position
  //      (the diagonal of G is not needed any more)
  double *yp = W.gdiag;
  QPC_WG_FOR(b, m) { yp[b] = b < ne ? W.yE[b] : 0.0; }
  QPC_WG_SYNC
  QPC_WG_FOR(a, nw) { yp[ne + W.idxW[a]] = W.yw[a]; }
  QPC_WG_SYNC
  QPC_WG_FOR(i, n) { W.x[i] = W.xu[i] - W.hinv[i] * wg_col_dot(W, i, yp); }
  QPC_WG_SYNC
  double rp = 0.0, rs = 0.0;
  for (int sweep = 0; sweep <= st.max_refine_iters; sweep++) {
    double a_rs = 0.0, a_rp = 0.0, a_ss = 1.0, a_ps = 1.0; // residuals and the scales they are measured against
    QPC_WG_FOR(i, n) { // r1 = -(q + P x + A' y)
      const double aty = wg_col_dot(W, i, yp), qi = W.q[i];
      const double t = qi + wg_pdiag(W, i) * W.x[i] + aty;
      W.r1[i] = -t;
      a_rs = (QPC_FABS(t) > a_rs || !(t == t)) ? QPC_FABS(t) : a_rs;
      a_ss = QPC_FABS(qi) > a_ss ? QPC_FABS(qi) : a_ss;
      a_ss = QPC_FABS(aty) > a_ss ? QPC_FABS(aty) : a_ss;
    }
    QPC_WG_FOR(b, m) { // r2 = b_W - A_W x, by position (the rows outside the working set: 0)
      const int p = b - ne;
      double t = 0.0;
      if (b < ne || W.wsf[p] != 0.0) {
        const double bw = b < ne ? W.l[W.perm[b]] : (W.wsf[p] > 0.0 ? W.hi[p] : W.lo[p]);
        t = bw - wg_row_dot(W, b, W.x);
        a_rp = (QPC_FABS(t) > a_rp || !(t == t)) ? QPC_FABS(t) : a_rp;
        a_ps = QPC_FABS(bw) > a_ps ? QPC_FABS(bw) : a_ps;
      }
      W.tA[b] = t;
    }
    QPC_WG_SYNC
    rs = wg_reduce_max(a_rs, W.red);
    rp = wg_reduce_max(a_rp, W.red);
    if (sweep == st.max_refine_iters) break;
    // after a correction: residuals at the rounding floor of their own scales (|q|, |A'y|; the working rows' bounds) -- another
    // correction would change nothing that 1e-8 can see (the eps bias of the first solve is 1e-6: one correction takes it to 1e-12)
    if (sweep > 0) {
      const double ss = wg_reduce_max(a_ss, W.red), ps = wg_reduce_max(a_ps, W.red);
      if (rs <= 1e-12 * ss && rp <= 1e-12 * ps) break;
    }
    // dy = K^-1 (A_W H^-1 r1 - r2) ; dx = H^-1 (r1 - A_W' dy)
    QPC_WG_FOR(i, n) { W.xu[i] = W.hinv[i] * W.r1[i]; }    // xu is free from here on: H^-1 r1
    QPC_WG_SYNC
    QPC_WG_FOR(b, m) {
      const int p = b - ne;
      if (b < ne || W.wsf[p] != 0.0) {
        const double t = wg_row_dot(W, b, W.xu) - W.tA[b];
        if (b < ne) W.tE[b] = t; else W.tm[p] = t;
      }
    }
    QPC_WG_SYNC
    QPC_WG_FOR(a, nw) { W.tW[a] = W.tm[W.idxW[a]]; }
    QPC_WG_SYNC
    wg_block_solve(W, nw, W.tE, W.dW);
    QPC_WG_FOR(b, m) { W.tA[b] = b < ne ? W.tE[b] : 0.0; } // dy by position
    QPC_WG_SYNC
    QPC_WG_FOR(a, nw) { W.tA[ne + W.idxW[a]] = W.dW[a]; }
    QPC_WG_SYNC
    double a_big = 0.0, a_dbig = 0.0;
    QPC_WG_FOR(i, n) {
      const double dx = W.hinv[i] * (W.r1[i] - wg_col_dot(W, i, W.tA));
      const double xi = W.x[i] + dx;
      W.x[i] = xi;
      a_big = QPC_FABS(xi) > a_big ? QPC_FABS(xi) : a_big;
      a_dbig = QPC_FABS(dx) > a_dbig ? QPC_FABS(dx) : a_dbig;
    }
    QPC_WG_FOR(b, m) { yp[b] += W.tA[b]; }
    QPC_WG_SYNC
    const double big = wg_reduce_max(a_big, W.red), dbig = wg_reduce_max(a_dbig, W.red);
    if (sweep > 0 && !(dbig > 1e-12 * (1.0 + big))) break; // (the residuals of this sweep -- before a correction of that size -- are reported)
  }
  QPC_WG_STAMP(W, 7);
  double a_ysc = 1.0, a_qsc = 1.0;
  QPC_WG_FOR(b, m) { a_ysc = QPC_FABS(yp[b]) > a_ysc ? QPC_FABS(yp[b]) : a_ysc; }
  QPC_WG_FOR(i, n) { a_qsc = QPC_FABS(W.q[i]) > a_qsc ? QPC_FABS(W.q[i]) : a_qsc; }
  const double ysc = wg_reduce_max(a_ysc, W.red), qsc = wg_reduce_max(a_qsc, W.red);
  if (status == QPC_STATUS_OPTIMAL) {
    if (!(rp == rp) || !(rs == rs)) status = QPC_STATUS_NUMERICAL_ERROR;
    else if (rp > 1e-7 * bscale) status = QPC_STATUS_INFEASIBLE;
    else if (rs > 1e-7 * (ysc > qsc ? ysc : qsc)) status = QPC_STATUS_ALMOST_OPTIMAL;
  }
  QPC_WG_FOR(i, n) { x[i] = W.x[i]; }
  QPC_WG_FOR(b, m) { y[W.perm[b]] = yp[b]; }
  QPC_WG_T0 { if (iters_out) iters_out[0] = iters; }
  QPC_WG_STAMP(W, 8);
  return status;
}

// returns the status (QPC_STATUS_*), or QPC_WG_FALLBACK when the problem is outside what this kernel handles.
// lds: lds_cap doubles (QPC_WG_LDS_DOUBLES on the device); ws: the global workspace of qp_solve_dense (qp_dense_workspace_doubles).
QPC_FN int qp_solve_wg(int n, int m, const double *P, const double *q, const double *A, const double *l, const double *u,
                       const qpc_settings &st, double *ws, double *lds, int lds_cap, double *x, double *y, int32_t *iters_out) {
  const double INF = 1e19;
  if (n > QPC_WG_DIM || m > QPC_WG_DIM || n < 1 || m < 1) return QPC_WG_FALLBACK;
  QpWg W;
  W.n = n; W.m = m; W.N1 = m + 1; W.P = P; W.q = q; W.A = A; W.l = l; W.u = u;
  W.V = nullptr; W.Pv = nullptr; W.y0 = nullptr;
  const int nS = (m + 1) * (m + 2) / 2;
  // global workspace: [AT n x m | S if it does not fit in LDS | cval | crow | stamps]
  W.AT = ws;
  double *Sg = ws + (long long)n * m;
  W.cval = Sg + nS;
  W.stamps = (unsigned long long *)(ws + qp_dense_workspace_doubles(n, m) - 16);
  QPC_WG_STAMP(W, 0);
  // LDS, first part (what does not depend on the number of equality rows)
  double *p = lds;
  W.sc = p; p += 16; W.red = p; p += 8;
  W.hinv = p; p += n; W.xu = p; p += n; W.x = p; p += n; W.r1 = p; p += n;
  W.gdiag = p; p += m + 1; W.tA = p; p += m + 1;
  W.perm = (int *)p; W.ipos = W.perm + m; p += m;
  W.cptr = (int *)p; p += (n + 2) / 2;
  if ((int)(p - lds) + (n + 2) / 2 > lds_cap) return QPC_WG_FALLBACK;
  int *ccnt = (int *)p; // (scratch of the set-up: nonzeros per column; in the part carved below)
  // ---- structure: P diagonal (as many nonzeros in P as on its diagonal); nonzeros of A per column (a thread per column: coalesced,
  //      eight rows in flight); equality rows
  QPC_WG_T0 { for (int i = 0; i < WGS_COUNT; i++) W.sc[i] = 0.0; }
  double a_h = 0.0, a_b = 1.0, a_nz = 0.0;
  {
    const int nn = n * n;
    for (int e0 = QPC_WG_TID; e0 < nn; e0 += 8 * QPC_WG_NT) {
      double v[8];
#pragma unroll
      for (int u8 = 0; u8 < 8; u8++) { const int e = e0 + u8 * QPC_WG_NT; v[u8] = P[e < nn ? e : e0]; }
#pragma unroll
      for (int u8 = 0; u8 < 8; u8++) a_nz += (e0 + u8 * QPC_WG_NT < nn && v[u8] != 0.0) ? 1.0 : 0.0;
    }
  }
  QPC_WG_FOR(i, n) {
    const double d = P[(long long)i * n + i], a = QPC_FABS(d);
    a_h = a > a_h ? a : a_h;
    a_nz -= d != 0.0 ? 1.0 : 0.0;
    int c = 0, r = 0;
    for (; r + 8 <= m; r += 8) {
      const double *a8 = A + (long long)r * n + i;
      const double v0 = a8[0], v1 = a8[n], v2 = a8[2 * n], v3 = a8[3 * n], v4 = a8[4 * n], v5 = a8[5 * n], v6 = a8[6 * n], v7 = a8[7 * n];
      c += (v0 != 0.0) + (v1 != 0.0) + (v2 != 0.0) + (v3 != 0.0) + (v4 != 0.0) + (v5 != 0.0) + (v6 != 0.0) + (v7 != 0.0);
    }
    for (; r < m; r++) c += A[(long long)r * n + i] != 0.0 ? 1 : 0;
    ccnt[i] = c;
  }
  QPC_WG_FOR(i, m) {
    const double li = l[i], ui = u[i];
    const bool lo = li > -INF, hi = ui < INF;
    W.tA[i] = (lo && hi && ui - li <= 1e-12 * (1.0 + QPC_FABS(li))) ? 1.0 : 0.0;
    if (lo) a_b = QPC_FABS(li) > a_b ? QPC_FABS(li) : a_b;
    if (hi) a_b = QPC_FABS(ui) > a_b ? QPC_FABS(ui) : a_b;
  }
  QPC_WG_SYNC
  const double hmax = wg_reduce_max(a_h, W.red), bscale = wg_reduce_max(a_b, W.red);
  // (a_nz: this thread's count of nonzeros of P minus its diagonal ones; the per-thread values need not vanish one by one, their sum must:
  //  max and min of the partial sums are not enough -- so the threads add them up in LDS, in integers)
  QPC_WG_ADD(&W.sc[WGS_FLAG], a_nz); // (small integers in doubles: exact in any order)
  QPC_WG_FOR(i, m) { // equality rows first: the position of a row = the rows of its kind before it
    int e = 0;
    for (int o = 0; o < i; o++) e += W.tA[o] != 0.0 ? 1 : 0;
    const bool eq = W.tA[i] != 0.0;
    if (i == m - 1) W.sc[WGS_NE] = (double)(e + (eq ? 1 : 0));
    W.gdiag[i] = (double)(eq ? e : -(i - e) - 1); // position among its kind (the others: encoded negative, shifted by ne below)
  }
  QPC_WG_FOR(c, n + 1) { // column starts
    int o = 0;
    for (int c2 = 0; c2 < c && c2 < n; c2++) o += ccnt[c2];
    W.cptr[c] = o;
    if (c == n) W.sc[WGS_NNZ] = (double)o;
  }
  QPC_WG_SYNC
  if (W.sc[WGS_FLAG] != 0.0) return QPC_WG_FALLBACK;
  const int ne = (int)W.sc[WGS_NE], mi = m - ne, nnz = (int)W.sc[WGS_NNZ];
  W.ne = ne; W.mi = mi;
  W.crow = (int *)(W.cval + nnz);
  QPC_WG_FOR(i, m) {
    const double g = W.gdiag[i];
    const int pos = g >= 0.0 ? (int)g : ne + (int)(-g - 1.0);
    W.perm[pos] = i; W.ipos[i] = pos;
  }
  // LDS, second part
  W.lo = p; p += mi; W.hi = p; p += mi;
  double *scratch = p; // from here to the end of LW: free during the set-up (the staged columns of A)
  W.dinvE = p; p += ne; W.yE = p; p += ne; W.tE = p; p += ne;
  W.wsf = p; p += mi; W.tm = p; p += mi; W.ax = p; p += mi; W.a0 = p; p += mi;
  const int kw = mi < QPC_WG_KW ? mi : QPC_WG_KW;
  W.yw = p; p += kw + 1; W.dinvW = p; p += kw + 1; W.dW = p; p += kw + 1; W.tW = p; p += kw + 1;
  W.idxW = (int *)p; p += (kw + 2) / 2;
  W.LW = p; p += (kw + 1) * (kw + 2) / 2;
  if ((int)(p - lds) > lds_cap) return QPC_WG_FALLBACK;
  const bool s_in_lds = (int)(p - lds) + nS <= lds_cap;
  const bool stage = nnz + (nnz + 1) / 2 <= (int)(p - scratch);
  QPC_WG_SYNC // (ccnt lies in what has just been carved: every thread is past reading it; perm / ipos are complete)
  const double eps = st.eps_primal_reg * (hmax > 0.0 ? hmax : 1.0);
  double *lval = scratch;
  int *lrow = (int *)(scratch + nnz);
  QPC_WG_FOR(c, n) { // the compressed columns: row POSITIONS and values, in row order
    int o = W.cptr[c];
    for (int r = 0; r < m; r += 8) {
      double v[8];
#pragma unroll
      for (int u8 = 0; u8 < 8; u8++) v[u8] = A[(long long)(r + u8 < m ? r + u8 : r) * n + c];
#pragma unroll
      for (int u8 = 0; u8 < 8; u8++) {
        if (r + u8 < m && v[u8] != 0.0) {
          const int pos = W.ipos[r + u8];
          W.cval[o] = v[u8]; W.crow[o] = pos;
          if (stage) { lval[o] = v[u8]; lrow[o] = pos; }
          o++;
        }
      }
    }
    const double h = 1.0 / (P[(long long)c * n + c] + eps);
    W.hinv[c] = h;
    W.xu[c] = -q[c] * h;
  }
  QPC_WG_FOR(i, m) {
    const int pos = W.ipos[i];
    if (pos >= ne) { W.lo[pos - ne] = l[i]; W.hi[pos - ne] = u[i]; }
  }
  QPC_WG_FOR(e, n * m) { W.AT[e] = 0.0; }
  QPC_WG_SYNC
  QPC_WG_FOR(c, n) {
    for (int o = W.cptr[c]; o < W.cptr[c + 1]; o++) W.AT[(long long)c * m + W.crow[o]] = W.cval[o];
  }
  W.sval = stage ? lval : W.cval;
  W.srow = stage ? lrow : W.crow;
  QPC_WG_SYNC
  QPC_WG_STAMP(W, 1);
  if (s_in_lds) { W.S = p; return qp_solve_wg_core(W, st, bscale, kw, x, y, iters_out); }
  W.S = Sg;
  return qp_solve_wg_core(W, st, bscale, kw, x, y, iters_out);
}

// The same for a PREPARED problem (qpc_qp_setup: the pattern's tables V, built once on the host): Pv / Av are the caller's values in
// the pattern's order, y0 (or nullptr) the multipliers of an earlier solve for the warm start.  No dense matrix is read or built: the
// set-up is the classification of the rows and a gather of the values.  V's P must be diagonal (the caller checks).
QPC_FN int qp_solve_wg_prep(const QpPrepView &V, const double *Pv, const double *q, const double *Av, const double *l, const double *u,
                            const qpc_settings &st, double *ws, double *lds, int lds_cap, const double *y0, double *x, double *y,
                            int32_t *iters_out) {
  const double INF = 1e19;
  const int n = V.n, m = V.m, nnz = V.nnzA;
  if (n > QPC_WG_DIM || m > QPC_WG_DIM || n < 1 || m < 1) return QPC_WG_FALLBACK;
  QpWg W;
  W.n = n; W.m = m; W.N1 = m + 1; W.P = nullptr; W.q = q; W.A = nullptr; W.l = l; W.u = u;
  W.V = &V; W.Pv = Pv; W.y0 = y0;
  const int nS = (m + 1) * (m + 2) / 2;
  W.AT = ws; // (not used)
  double *Sg = ws + (long long)n * m;
  W.cval = Sg + nS;
  W.crow = (int *)(W.cval + nnz);
  W.stamps = (unsigned long long *)(ws + qp_dense_workspace_doubles(n, m) - 16);
  QPC_WG_STAMP(W, 0);
  double *p = lds;
  W.sc = p; p += 16; W.red = p; p += 8;
  W.hinv = p; p += n; W.xu = p; p += n; W.x = p; p += n; W.r1 = p; p += n;
  W.gdiag = p; p += m + 1; W.tA = p; p += m + 1;
  W.perm = (int *)p; W.ipos = W.perm + m; p += m;
  W.cptr = (int *)p; p += (n + 2) / 2;
  if ((int)(p - lds) > lds_cap) return QPC_WG_FALLBACK;
  QPC_WG_T0 { for (int i = 0; i < WGS_COUNT; i++) W.sc[i] = 0.0; }
  double a_h = 0.0, a_b = 1.0;
  QPC_WG_FOR(i, n) {
    const int k = V.pdiag[i];
    const double a = k >= 0 ? QPC_FABS(Pv[k]) : 0.0;
    a_h = a > a_h ? a : a_h;
  }
  QPC_WG_FOR(c, n + 1) { W.cptr[c] = V.cptr[c]; }
  QPC_WG_FOR(i, m) {
    const double li = l[i], ui = u[i];
    const bool lo = li > -INF, hi = ui < INF;
    W.tA[i] = (lo && hi && ui - li <= 1e-12 * (1.0 + QPC_FABS(li))) ? 1.0 : 0.0;
    if (lo) a_b = QPC_FABS(li) > a_b ? QPC_FABS(li) : a_b;
    if (hi) a_b = QPC_FABS(ui) > a_b ? QPC_FABS(ui) : a_b;
  }
  QPC_WG_SYNC
  const double hmax = wg_reduce_max(a_h, W.red), bscale = wg_reduce_max(a_b, W.red);
  QPC_WG_FOR(i, m) { // equality rows first: the position of a row = the rows of its kind before it
    int e = 0;
    for (int o = 0; o < i; o++) e += W.tA[o] != 0.0 ? 1 : 0;
    const bool eq = W.tA[i] != 0.0;
    if (i == m - 1) W.sc[WGS_NE] = (double)(e + (eq ? 1 : 0));
    W.gdiag[i] = (double)(eq ? e : -(i - e) - 1);
  }
  QPC_WG_SYNC
  const int ne = (int)W.sc[WGS_NE], mi = m - ne;
  W.ne = ne; W.mi = mi;
  QPC_WG_FOR(i, m) {
    const double g = W.gdiag[i];
    const int pos = g >= 0.0 ? (int)g : ne + (int)(-g - 1.0);
    W.perm[pos] = i; W.ipos[i] = pos;
  }
  W.lo = p; p += mi; W.hi = p; p += mi;
  double *scratch = p;
  W.dinvE = p; p += ne; W.yE = p; p += ne; W.tE = p; p += ne;
  W.wsf = p; p += mi; W.tm = p; p += mi; W.ax = p; p += mi; W.a0 = p; p += mi;
  const int kw = mi < QPC_WG_KW ? mi : QPC_WG_KW;
  W.yw = p; p += kw + 1; W.dinvW = p; p += kw + 1; W.dW = p; p += kw + 1; W.tW = p; p += kw + 1;
  W.idxW = (int *)p; p += (kw + 2) / 2;
  W.LW = p; p += (kw + 1) * (kw + 2) / 2;
  if ((int)(p - lds) > lds_cap) return QPC_WG_FALLBACK;
  const bool s_in_lds = (int)(p - lds) + nS <= lds_cap;
  const bool stage = nnz <= (int)(p - scratch);
  QPC_WG_SYNC
  const double eps = st.eps_primal_reg * (hmax > 0.0 ? hmax : 1.0);
  QPC_WG_FOR(o, nnz) { // the values in column order (global: the products of the final stage; LDS: the S build), rows as positions
    const double v = Av[V.cperm[o]];
    W.cval[o] = v; W.crow[o] = W.ipos[V.crow[o]];
    if (stage) scratch[o] = v;
  }
  QPC_WG_FOR(c, n) {
    const int k = V.pdiag[c];
    const double h = 1.0 / ((k >= 0 ? Pv[k] : 0.0) + eps);
    W.hinv[c] = h;
    W.xu[c] = -q[c] * h;
  }
  QPC_WG_FOR(i, m) {
    const int pos = W.ipos[i];
    if (pos >= ne) { W.lo[pos - ne] = l[i]; W.hi[pos - ne] = u[i]; }
  }
  W.sval = stage ? scratch : W.cval;
  W.srow = W.crow;
  QPC_WG_SYNC
  QPC_WG_STAMP(W, 1);
  if (s_in_lds) { W.S = p; return qp_solve_wg_core(W, st, bscale, kw, x, y, iters_out); }
  W.S = Sg;
  return qp_solve_wg_core(W, st, bscale, kw, x, y, iters_out);
}

} // namespace qpc
