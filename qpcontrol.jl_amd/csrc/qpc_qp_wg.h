// qpc_qp_wg.h -- L0 of the C ABI on a WORKGROUP: one dense convex QP in OSQP problem form
//       min 1/2 x'Px + q'x   s.t.   l <= A x <= u
// solved by the 256 threads of one workgroup with its working matrices in LDS -- what makes the solver-level plug of the reference
// (`Parametron.Model(optimizer)`, /root/reference/src/lowlevel/momentum.jl:15-16,27,58; `QPCOptimizer` in julia/QPControlHIPMOI.jl)
// usable at B = 1: the one-lane-per-QP form of qpc_qp_dense.h needs 0.57 s for the 143 x 178 Atlas problem (measured, MI355X), the
// reference's own warm-started OSQP 1.3 ms on a CPU core.
//
// Same method as qpc_qp_dense.h (range-space active set on G = A H^-1 A', primal-dual multi-change sweeps, iterative refinement on
// the true KKT residuals), for the problems Parametron actually builds for this controller: a DIAGONAL P (regularize!, contact
// weights, scalar task weights: 63 entries for Atlas) and a sparse A given densely (1 004 of 25 454 entries).  The kernel finds
// that out by itself:
//   * P diagonal            -> H^-1 is a vector, Y = H^-1 A' is never formed;
//   * rows of A compressed  -> G = A H^-1 A' costs (rows x nnz) multiply-adds instead of (rows^2 x columns);
//   * the working-set matrix S = G_WW (<= QPC_WG_KMAX rows, packed lower triangle) is factored IN LDS by a right-looking Cholesky
//     (one row per thread, three barriers per column, infinite-pivot rule for dependent rows) with the right-hand side riding
//     along, so a solve is one factorization + one back-substitution.
// Anything else -- a dense or matrix-weighted P, more than QPC_WG_NNZ nonzeros, a working set beyond QPC_WG_KMAX rows -- is left to
// the one-lane kernel: the workgroup reports QPC_WG_FALLBACK and the
// caller runs qp_solve_dense on that QP.  Both are gfx950 device code; neither is a fallback to anything else.
//
// Written as parallel loops (QPC_WG_FOR: strided over the workgroup) separated by barriers, so that the same source runs serially on
// the host inside the test emulation (tests/emul): one "thread", barriers are no-ops.
#pragma once
#include "../../include/qpc.h"

#ifndef QPC_FN
#error "include qpc_device.h (or the emulator's definitions) first"
#endif

#define QPC_WG_FALLBACK (-1) // status: not solved here, run the one-lane kernel
#define QPC_WG_KMAX 160      // rows of a working set the LDS factor holds (packed: 12 880 doubles)
#define QPC_WG_NNZ 2048      // nonzeros of A held in LDS
#define QPC_WG_DIM 256       // n, m <= this

#if defined(__HIP_DEVICE_COMPILE__)
#define QPC_WG_NT 256
#define QPC_WG_TID ((int)threadIdx.x)
#define QPC_WG_SYNC __syncthreads();
#else
#define QPC_WG_NT 1
#define QPC_WG_TID 0
#define QPC_WG_SYNC
#endif
#define QPC_WG_FOR(i, n) for (int i = QPC_WG_TID; i < (n); i += QPC_WG_NT)
#define QPC_WG_T0 if (QPC_WG_TID == 0)

namespace qpc {

// LDS (or host scratch) needed by qp_solve_wg, in doubles
#ifdef __HIPCC__
__host__ __device__
#endif
inline int qp_wg_lds_doubles(int n, int m) {
  (void)n; (void)m;
  return QPC_WG_KMAX * (QPC_WG_KMAX + 1) / 2 + QPC_WG_NNZ + QPC_WG_NNZ / 2 + 4 * QPC_WG_DIM + 12 * QPC_WG_DIM + 64 + 8;
}

struct QpWg {
  int n, m;
  const double *P, *q, *A, *l, *u;
  double *G;      // m x m in global memory (the workspace of qp_solve_dense keeps its G at the same place)
  // block elimination of the equality rows (they are in EVERY working set): factored once, the working sets only refactor their
  // active inequality rows.  E = equality rows (ne, first in every compaction), I = the others (mi).
  bool blk;
  int ne, mi;
  double *L21;    // mi x ne in global memory: row i = L11^-1 G_E,i  (the part of an inequality row's factor row that never changes)
  double *Call;   // mi x mi in global memory: Schur complement G_II - L21 L21' (what a working set factors a sub-block of)
  double *zE, *a0; // LDS: L11^-1 (g0_E - b_E);  A x at the equality-constrained minimiser for the inequality rows, g0_I - L21 zE
  int *ipos, *irows; // row -> position among the inequality rows (-1: equality row); position -> row
  // LDS
  double *S;      // packed lower triangle of the working-set matrix / its Cholesky factor
  double *val;    // nonzeros of A, row by row
  int *col, *rowptr;
  double *hinv, *xu, *x, *r1;              // n
  double *g0, *yw, *bw, *ax, *wsf, *sinvd, *tm, *sdiag; // m
  int *idx;       // compacted working set
  double *sc;     // scalars
};

enum { WGS_FLAG = 0, WGS_HMAX, WGS_EPS, WGS_BSCALE, WGS_K, WGS_CHANGES, WGS_YSC, WGS_XSC, WGS_RP, WGS_RS, WGS_BIG, WGS_DBIG, WGS_T, WGS_P, WGS_VIOL, WGS_SG, WGS_MI,
       WGS_COUNT };

QPC_FN int wg_sidx(int i, int j) { return i * (i + 1) / 2 + j; } // packed lower triangle, i >= j

// Factor the working set in W.wsf and solve one system with it: compaction -> idx (equality rows first), S = G_WW, Cholesky with the
// right-hand side riding along, back-substitution -> out (compact).  Right-hand side: pcol < 0: g0_W - b_W (the EQP of the working
// set; b -> W.bw); pcol >= 0: column pcol of G (the Goldfarb-Idnani direction of row pcol).  Returns the number of working rows, or -1
// beyond QPC_WG_KMAX.
// Block mode (W.blk): rows 0 .. ne-1 of S hold L11 for good; a working inequality row gets its first ne factor entries from L21 and
// its remaining ones from the Schur complement Call, so only the columns ne .. k-1 are eliminated.  full_back = false stops the
// back-substitution at column ne: out[ne ..] are the multipliers of the inequality rows, out[0 .. ne-1] still L11^-1 of the right-hand
// side (wg_finish_back completes them).
// zsave: the right-hand side after the forward part (L^-1 rhs), before the back-substitution.
QPC_FN int wg_factor(QpWg &W, double tau, int pcol, double *out, bool full_back = true, double *zsave = nullptr) {
  const int m = W.m;
  QPC_WG_T0 {
    int k = 0;
    for (int i = 0; i < m; i++) if (W.wsf[i] == 2.0) { if (k < QPC_WG_KMAX) W.idx[k] = i; k++; }
    for (int i = 0; i < m; i++) if (W.wsf[i] != 0.0 && W.wsf[i] != 2.0) { if (k < QPC_WG_KMAX) W.idx[k] = i; k++; }
    W.sc[WGS_K] = (double)k;
  }
  QPC_WG_SYNC
  const int k = (int)W.sc[WGS_K];
  if (k > QPC_WG_KMAX) return -1;
  const int j0 = W.blk ? W.ne : 0; // first column to eliminate
  if (W.blk) {
    const int ne = W.ne, pp = pcol >= 0 ? W.ipos[pcol] : -1;
    QPC_WG_FOR(e, (k - ne) * ne) { // the fixed part of the working inequality rows
      const int a = ne + e / ne, b = e % ne;
      W.S[wg_sidx(a, b)] = W.L21[(long long)W.ipos[W.idx[a]] * ne + b];
    }
    QPC_WG_FOR(a0_, k - ne) {
      const int a = ne + a0_, ia = W.idx[a], pa = W.ipos[ia];
      for (int b = ne; b <= a; b++) W.S[wg_sidx(a, b)] = W.Call[(long long)pa * W.mi + W.ipos[W.idx[b]]];
      if (pcol < 0) {
        const double bnd = (W.wsf[ia] == 1.0) ? W.u[ia] : W.l[ia];
        W.bw[a] = bnd;
        out[a] = W.a0[pa] - bnd;
      } else out[a] = W.Call[(long long)pa * W.mi + pp];
      W.sdiag[a] = W.G[(long long)ia * m + ia];
    }
    QPC_WG_FOR(b, ne) {
      out[b] = pcol < 0 ? W.zE[b] : W.L21[(long long)pp * ne + b];
      if (pcol < 0) W.bw[b] = W.l[W.idx[b]];
    }
  } else {
    QPC_WG_FOR(a, k) {
      const int ia = W.idx[a];
      for (int b = 0; b <= a; b++) W.S[wg_sidx(a, b)] = W.G[(long long)ia * m + W.idx[b]];
      if (pcol < 0) {
        const double w = W.wsf[ia];
        const double bnd = (w == 1.0) ? W.u[ia] : W.l[ia]; // lower-active and equality rows sit at l
        W.bw[a] = bnd;
        out[a] = W.g0[ia] - bnd;
      } else out[a] = W.G[(long long)ia * m + pcol];
      W.sdiag[a] = W.G[(long long)ia * m + ia];
    }
  }
  QPC_WG_SYNC
  for (int j = j0; j < k; j++) { // right-looking Cholesky, infinite-pivot rule; out <- L^-1 out on the way
    QPC_WG_T0 {
      const double d = W.S[wg_sidx(j, j)], ref = W.sdiag[j];
      const bool ok = d > tau * (ref > 1e-300 ? ref : 1e-300);
      const double inv = ok ? 1.0 / QPC_SQRT(d) : 0.0;
      W.sinvd[j] = inv;
      out[j] = out[j] * inv;
    }
    QPC_WG_SYNC
    const double inv = W.sinvd[j], yj = out[j];
    QPC_WG_FOR(i0, k - j - 1) {
      const int i = j + 1 + i0;
      const double lij = W.S[wg_sidx(i, j)] * inv;
      W.S[wg_sidx(i, j)] = lij;
      out[i] -= lij * yj;
    }
    QPC_WG_SYNC
    QPC_WG_FOR(i0, k - j - 1) { // trailing update, one row per thread
      const int i = j + 1 + i0;
      const double lij = W.S[wg_sidx(i, j)];
      if (lij != 0.0)
        for (int c = j + 1; c <= i; c++) W.S[wg_sidx(i, c)] -= lij * W.S[wg_sidx(c, j)];
    }
    QPC_WG_SYNC
  }
  if (zsave) {
    QPC_WG_FOR(a, k) { zsave[a] = out[a]; }
    QPC_WG_SYNC
  }
  for (int j = k - 1; j >= (full_back ? 0 : j0); j--) { // L' y = z
    QPC_WG_T0 { out[j] = out[j] * W.sinvd[j]; }
    QPC_WG_SYNC
    const double yj = out[j];
    QPC_WG_FOR(i, j) { out[i] -= W.S[wg_sidx(j, i)] * yj; }
    QPC_WG_SYNC
  }
  return k;
}
// block mode: the equality part of a back-substitution that wg_factor(full_back = false) left at column ne
QPC_FN void wg_finish_back(QpWg &W, double *out) {
  if (!W.blk) return;
  for (int j = W.ne - 1; j >= 0; j--) {
    QPC_WG_T0 { out[j] = out[j] * W.sinvd[j]; }
    QPC_WG_SYNC
    const double yj = out[j];
    QPC_WG_FOR(i, j) { out[i] -= W.S[wg_sidx(j, i)] * yj; }
    QPC_WG_SYNC
  }
}
// One-time set-up of the block mode, right after the equality-only working set has been factored (S rows 0 .. ne-1 = L11, W.zE =
// L11^-1 (g0_E - b_E): the zsave of that factorization).  Every inequality row: its factor row against L11 by forward substitution -- one thread
// per row, nothing shared but read-only L11 -- in chunks that fit behind L11 in the LDS buffer; then the Schur complement.
QPC_FN void wg_block_setup(QpWg &W) {
  const int m = W.m, ne = W.ne, mi = W.mi;
  const int chunk = QPC_WG_KMAX - ne;
  for (int c0 = 0; c0 < mi; c0 += chunk) {
    const int cnt = mi - c0 < chunk ? mi - c0 : chunk;
    QPC_WG_FOR(a, cnt) {
      const int row = ne + a, ir = W.irows[c0 + a];
      double acc0 = W.g0[ir];
      for (int j = 0; j < ne; j++) {
        double t = W.G[(long long)ir * m + W.idx[j]];
        for (int c = 0; c < j; c++) t -= W.S[wg_sidx(row, c)] * W.S[wg_sidx(j, c)];
        t *= W.sinvd[j];
        W.S[wg_sidx(row, j)] = t;
        acc0 -= t * W.zE[j];
      }
      for (int j = 0; j < ne; j++) W.L21[(long long)(c0 + a) * ne + j] = W.S[wg_sidx(row, j)];
      W.a0[c0 + a] = acc0;
    }
    QPC_WG_SYNC
  }
  QPC_WG_FOR(e, mi * mi) {
    const int i = e / mi, c = e % mi;
    if (c <= i) {
      double t = W.G[(long long)W.irows[i] * m + W.irows[c]];
      for (int j = 0; j < ne; j++) t -= W.L21[(long long)i * ne + j] * W.L21[(long long)c * ne + j];
      W.Call[(long long)i * mi + c] = t;
      W.Call[(long long)c * mi + i] = t;
    }
  }
  QPC_WG_SYNC
}
// EQP of the working set: multipliers -> yw (compact), ax = A x for every row
QPC_FN int wg_solve_ws(QpWg &W, double tau, double *zsave = nullptr) {
  const int m = W.m;
  const int k = wg_factor(W, tau, -1, W.yw, !W.blk, zsave);
  if (k < 0) return -1;
  if (W.blk) { // in the space of the inequality rows: A x = a0 - Call[:, W] y_W ; the equality rows hold exactly
    QPC_WG_FOR(i, m) {
      const int pi = W.ipos[i];
      double t = pi < 0 ? W.l[i] : W.a0[pi];
      if (pi >= 0)
        for (int a = W.ne; a < k; a++) t -= W.Call[(long long)pi * W.mi + W.ipos[W.idx[a]]] * W.yw[a];
      W.ax[i] = t;
    }
  } else {
    QPC_WG_FOR(i, m) {
      double t = W.g0[i];
      for (int a = 0; a < k; a++) t -= W.G[(long long)i * m + W.idx[a]] * W.yw[a];
      W.ax[i] = t;
    }
  }
  QPC_WG_SYNC
  return k;
}

// z <- S^-1 z with the factor left in W.S by wg_solve_ws (z compact, length k)
QPC_FN void wg_factor_solve(QpWg &W, double *z, int k) {
  for (int j = 0; j < k; j++) {
    QPC_WG_T0 { z[j] = z[j] * W.sinvd[j]; }
    QPC_WG_SYNC
    const double zj = z[j];
    QPC_WG_FOR(i0, k - j - 1) { const int i = j + 1 + i0; z[i] -= W.S[wg_sidx(i, j)] * zj; }
    QPC_WG_SYNC
  }
  for (int j = k - 1; j >= 0; j--) {
    QPC_WG_T0 { z[j] = z[j] * W.sinvd[j]; }
    QPC_WG_SYNC
    const double zj = z[j];
    QPC_WG_FOR(i, j) { z[i] -= W.S[wg_sidx(j, i)] * zj; }
    QPC_WG_SYNC
  }
}

// returns the status (QPC_STATUS_*), or QPC_WG_FALLBACK when the problem is outside what this kernel handles
QPC_FN int qp_solve_wg(int n, int m, const double *P, const double *q, const double *A, const double *l, const double *u,
                       const qpc_settings &st, double *ws, double *lds, double *x, double *y, int32_t *iters_out) {
  const double INF = 1e19;
  if (n > QPC_WG_DIM || m > QPC_WG_DIM || n < 1) return QPC_WG_FALLBACK;
  QpWg W;
  // global workspace = the one of qp_solve_dense (qp_dense_workspace_doubles): [H n x n | Y n x m | G m x m | S m x m | ...]; G sits
  // where that kernel keeps it, the block mode's L21 (mi x ne) goes where its Y was, Call (mi x mi) where its S was
  W.n = n; W.m = m; W.P = P; W.q = q; W.A = A; W.l = l; W.u = u; W.G = ws + (long long)n * n + (long long)n * m;
  W.L21 = ws + (long long)n * n; W.Call = W.G + (long long)m * m;
  W.blk = false; W.ne = 0; W.mi = m;
  double *p = lds;
  W.S = p; p += QPC_WG_KMAX * (QPC_WG_KMAX + 1) / 2;
  W.val = p; p += QPC_WG_NNZ;
  W.col = (int *)p; p += QPC_WG_NNZ / 2;
  W.hinv = p; p += QPC_WG_DIM; W.xu = p; p += QPC_WG_DIM; W.x = p; p += QPC_WG_DIM; W.r1 = p; p += QPC_WG_DIM;
  W.g0 = p; p += QPC_WG_DIM; W.yw = p; p += QPC_WG_DIM; W.bw = p; p += QPC_WG_DIM; W.ax = p; p += QPC_WG_DIM;
  W.wsf = p; p += QPC_WG_DIM; W.sinvd = p; p += QPC_WG_DIM; W.tm = p; p += QPC_WG_DIM; W.sdiag = p; p += QPC_WG_DIM;
  W.idx = (int *)p; W.ipos = W.idx + QPC_WG_DIM; p += QPC_WG_DIM;         // (two ints per double)
  W.rowptr = (int *)p; W.irows = W.rowptr + QPC_WG_DIM + 1; p += QPC_WG_DIM + 1;
  W.zE = p; p += QPC_WG_DIM; W.a0 = p; p += QPC_WG_DIM;
  W.sc = p; p += 64;
  // ---- structure: diagonal P, nonzeros of A per row
  QPC_WG_T0 { for (int i = 0; i < WGS_COUNT; i++) W.sc[i] = 0.0; }
  QPC_WG_SYNC
  QPC_WG_FOR(e, n * n) {
    const int i = e / n, j = e % n;
    if (i != j && P[e] != 0.0) W.sc[WGS_FLAG] = 1.0; // (every writer writes the same value)
  }
  QPC_WG_FOR(i, m) {
    int c = 0;
    for (int k2 = 0; k2 < n; k2++) c += A[(long long)i * n + k2] != 0.0 ? 1 : 0;
    W.rowptr[i + 1] = c;
  }
  QPC_WG_SYNC
  QPC_WG_T0 {
    W.rowptr[0] = 0;
    for (int i = 0; i < m; i++) W.rowptr[i + 1] += W.rowptr[i];
    if (W.rowptr[m] > QPC_WG_NNZ) W.sc[WGS_FLAG] = 1.0;
    double hmax = 0.0;
    for (int i = 0; i < n; i++) { const double a = QPC_FABS(P[(long long)i * n + i]); hmax = a > hmax ? a : hmax; }
    W.sc[WGS_HMAX] = hmax;
  }
  QPC_WG_SYNC
  if (W.sc[WGS_FLAG] != 0.0) return QPC_WG_FALLBACK;
  const double eps = st.eps_primal_reg * (W.sc[WGS_HMAX] > 0.0 ? W.sc[WGS_HMAX] : 1.0);
  QPC_WG_FOR(i, m) {
    int o = W.rowptr[i];
    for (int k2 = 0; k2 < n; k2++) {
      const double a = A[(long long)i * n + k2];
      if (a != 0.0) { W.val[o] = a; W.col[o] = k2; o++; }
    }
  }
  QPC_WG_FOR(i, n) {
    const double h = 1.0 / (P[(long long)i * n + i] + eps);
    W.hinv[i] = h;
    W.xu[i] = -q[i] * h;
  }
  QPC_WG_SYNC
  // ---- G = A H^-1 A' (lower triangle computed, mirrored), g0 = A xu
  QPC_WG_FOR(e, m * m) {
    const int i = e / m, j = e % m;
    if (j <= i) {
      double s = 0.0;
      for (int o = W.rowptr[j]; o < W.rowptr[j + 1]; o++) { const int c = W.col[o]; s += W.val[o] * A[(long long)i * n + c] * W.hinv[c]; }
      W.G[(long long)i * m + j] = s;
      W.G[(long long)j * m + i] = s;
    }
  }
  QPC_WG_FOR(i, m) {
    double t = 0.0;
    for (int o = W.rowptr[i]; o < W.rowptr[i + 1]; o++) t += W.val[o] * W.xu[W.col[o]];
    W.g0[i] = t;
    const bool lo = l[i] > -INF, hi = u[i] < INF;
    W.wsf[i] = (lo && hi && u[i] - l[i] <= 1e-12 * (1.0 + QPC_FABS(l[i]))) ? 2.0 : 0.0;
  }
  QPC_WG_SYNC
  QPC_WG_T0 {
    double bscale = 1.0;
    for (int i = 0; i < m; i++) {
      if (l[i] > -INF) bscale = QPC_FABS(l[i]) > bscale ? QPC_FABS(l[i]) : bscale;
      if (u[i] < INF) bscale = QPC_FABS(u[i]) > bscale ? QPC_FABS(u[i]) : bscale;
    }
    W.sc[WGS_BSCALE] = bscale;
  }
  QPC_WG_SYNC
  const double bscale = W.sc[WGS_BSCALE];
  int status = QPC_STATUS_OPTIMAL, iters = 0;
  const int max_it = st.max_active_set_iters > 0 ? st.max_active_set_iters : 200;
  int nw = wg_solve_ws(W, st.eps_dual_reg, W.zE); // the equality-only working set (block mode is not on yet); zE = L11^-1 (g0_E - b_E)
  if (nw < 0) return QPC_WG_FALLBACK;
  // block mode: worth it when there are equality rows, and possible when L11 leaves room for a chunk of inequality rows behind it
  QPC_WG_T0 {
    int mi = 0;
    for (int i = 0; i < m; i++) { if (W.wsf[i] == 2.0) W.ipos[i] = -1; else { W.ipos[i] = mi; W.irows[mi] = i; mi++; } }
    W.sc[WGS_MI] = (double)mi;
  }
  QPC_WG_SYNC
  {
    const int mi = (int)W.sc[WGS_MI], ne = m - mi;
    if (ne >= 8 && ne + 16 <= QPC_WG_KMAX && (long long)mi * ne <= (long long)n * m && mi >= 1) {
      W.ne = ne; W.mi = mi;
      wg_block_setup(W);
      W.blk = true;
      nw = wg_solve_ws(W, st.eps_dual_reg);
      if (nw < 0) return QPC_WG_FALLBACK;
    }
  }
  bool converged = false;
  const int max_pdas = 12;
  // ---- primal-dual multi-change sweeps (qpc_qp_dense.h phase A)
  for (; iters < max_pdas && iters < max_it; iters++) {
    QPC_WG_T0 {
      double ysc = 1.0, xsc = 1.0;
      for (int a = 0; a < nw; a++) ysc = QPC_FABS(W.yw[a]) > ysc ? QPC_FABS(W.yw[a]) : ysc;
      for (int i = 0; i < m; i++) xsc = QPC_FABS(W.ax[i]) > xsc ? QPC_FABS(W.ax[i]) : xsc;
      W.sc[WGS_YSC] = ysc; W.sc[WGS_XSC] = xsc; W.sc[WGS_CHANGES] = 0.0;
    }
    QPC_WG_FOR(i, m) { W.tm[i] = 0.0; }
    QPC_WG_SYNC
    const double toly = st.tol_bound * W.sc[WGS_YSC], xsc = W.sc[WGS_XSC], tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
    QPC_WG_FOR(a, nw) { // wrong-signed multipliers and dependent (dropped) inequality rows leave
      const int i = W.idx[a];
      const double w = W.wsf[i];
      const double viol = (w == -1.0) ? W.yw[a] : ((w == 1.0) ? -W.yw[a] : 0.0);
      if (viol > toly || (w != 2.0 && W.sinvd[a] == 0.0)) W.sc[WGS_CHANGES] = 1.0;
      else W.tm[i] = w;
    }
    QPC_WG_SYNC
    QPC_WG_FOR(i, m) {
      if (W.wsf[i] == 0.0) {
        const double lo = l[i] > -INF ? l[i] - W.ax[i] : -1.0, hi = u[i] < INF ? W.ax[i] - u[i] : -1.0;
        if (lo > tolx && lo >= hi) { W.tm[i] = -1.0; W.sc[WGS_CHANGES] = 1.0; }
        else if (hi > tolx) { W.tm[i] = 1.0; W.sc[WGS_CHANGES] = 1.0; }
      }
    }
    QPC_WG_SYNC
    if (W.sc[WGS_CHANGES] == 0.0) { converged = true; break; }
    QPC_WG_FOR(i, m) { W.wsf[i] = W.tm[i]; }
    QPC_WG_SYNC
    nw = wg_solve_ws(W, st.eps_dual_reg);
    if (nw < 0) return QPC_WG_FALLBACK;
  }
  // ---- Goldfarb-Idnani dual active set from the equality-only working set (finite; qpc_qp_dense.h phase B).  The scalar logic of an
  //      iteration (most violated row, ratio test, step) is thread 0's, the factorizations the workgroup's.
  if (!converged) {
    QPC_WG_FOR(i, m) { W.wsf[i] = W.wsf[i] == 2.0 ? 2.0 : 0.0; }
    QPC_WG_SYNC
    nw = wg_solve_ws(W, st.eps_dual_reg);
    if (nw < 0) return QPC_WG_FALLBACK;
    for (; iters < max_it; iters++) {
      QPC_WG_T0 {
        double xsc = 1.0;
        for (int i = 0; i < m; i++) xsc = QPC_FABS(W.ax[i]) > xsc ? QPC_FABS(W.ax[i]) : xsc;
        const double tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
        int pp = -1;
        double viol = tolx, sg = 0.0;
        for (int i = 0; i < m; i++) {
          if (W.wsf[i] != 0.0) continue;
          const double lo = l[i] > -INF ? l[i] - W.ax[i] : -1.0, hi = u[i] < INF ? W.ax[i] - u[i] : -1.0;
          if (lo > viol) { viol = lo; pp = i; sg = -1.0; }
          if (hi > viol) { viol = hi; pp = i; sg = 1.0; }
        }
        W.sc[WGS_P] = (double)pp; W.sc[WGS_VIOL] = viol; W.sc[WGS_SG] = sg;
      }
      QPC_WG_SYNC
      const int pr = (int)W.sc[WGS_P];
      if (pr < 0) { converged = true; break; }
      const double sg = W.sc[WGS_SG];
      bool infeasible = false, bail = false;
      for (int inner = 0; inner <= m; inner++) {
        const int k = wg_factor(W, st.eps_dual_reg, pr, W.tm, !W.blk); // r = (G_WW)^-1 G_Wp : a_p in terms of the working rows (H^-1 metric)
        if (k < 0) { bail = true; break; }
        QPC_WG_T0 {
          double viol = W.sc[WGS_VIOL];
          double d;
          if (W.blk) { // Schur-complement identity: the same number from the inequality block alone
            const int pp = W.ipos[pr];
            d = W.Call[(long long)pp * W.mi + pp];
            for (int a = W.ne; a < k; a++) d -= W.Call[(long long)pp * W.mi + W.ipos[W.idx[a]]] * W.tm[a];
          } else {
            d = W.G[(long long)pr * m + pr];
            for (int a = 0; a < k; a++) d -= W.G[(long long)pr * m + W.idx[a]] * W.tm[a];
          }
          const bool dependent = !(d > 1e-11 * W.G[(long long)pr * m + pr]);
          double t1 = 1e300; // dual ratio test: y_W(t) = y_W - t sg r must keep the sign of every working inequality
          int kd = -1;
          for (int a = 0; a < k; a++) {
            const double w = W.wsf[W.idx[a]], sr = sg * W.tm[a];
            if (w == 2.0 || W.sinvd[a] == 0.0) continue;
            if ((w == 1.0 && sr > 1e-14) || (w == -1.0 && sr < -1e-14)) {
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
              for (int a = 0; a < k; a++) W.yw[a] -= t1 * sg * W.tm[a];
              if (!dependent) viol -= t1 * d;
              W.wsf[W.idx[kd]] = 0.0;
              for (int a = kd; a + 1 < k; a++) W.yw[a] = W.yw[a + 1];
            }
          }
          W.sc[WGS_VIOL] = viol; W.sc[WGS_T] = (double)outcome;
        }
        QPC_WG_SYNC
        const int outcome = (int)W.sc[WGS_T];
        if (outcome == 2) { infeasible = true; break; }
        if (outcome == 0) break;
      }
      if (bail) return QPC_WG_FALLBACK;
      if (infeasible) { status = QPC_STATUS_INFEASIBLE; break; }
      QPC_WG_T0 { W.wsf[pr] = sg; }
      QPC_WG_SYNC
      nw = wg_solve_ws(W, st.eps_dual_reg);
      if (nw < 0) return QPC_WG_FALLBACK;
    }
  }
  if (!converged && status == QPC_STATUS_OPTIMAL) status = QPC_STATUS_ITERATION_LIMIT;
  wg_finish_back(W, W.yw); // block mode: the multipliers of the equality rows, once
  // ---- x = xu - Y_W y_W, then iterative refinement on the true KKT residuals of the final working set
  QPC_WG_FOR(i, n) {
    double t = 0.0;
    for (int a = 0; a < nw; a++) t += A[(long long)W.idx[a] * n + i] * W.yw[a];
    W.x[i] = W.xu[i] - W.hinv[i] * t;
  }
  QPC_WG_SYNC
  double rp = 0.0, rs = 0.0;
  for (int sweep = 0; sweep <= st.max_refine_iters; sweep++) {
    QPC_WG_FOR(i, n) {
      double t = q[i] + P[(long long)i * n + i] * W.x[i];
      for (int a = 0; a < nw; a++) t += A[(long long)W.idx[a] * n + i] * W.yw[a];
      W.r1[i] = -t;
    }
    QPC_WG_FOR(a, nw) {
      const int ia = W.idx[a];
      double t = W.bw[a];
      for (int o = W.rowptr[ia]; o < W.rowptr[ia + 1]; o++) t -= W.val[o] * W.x[W.col[o]];
      W.tm[a] = t; // r2
    }
    QPC_WG_SYNC
    QPC_WG_T0 {
      double a_rs = 0.0, a_rp = 0.0;
      for (int i = 0; i < n; i++) a_rs = QPC_FABS(W.r1[i]) > a_rs || !(W.r1[i] == W.r1[i]) ? QPC_FABS(W.r1[i]) : a_rs;
      for (int a = 0; a < nw; a++) a_rp = QPC_FABS(W.tm[a]) > a_rp || !(W.tm[a] == W.tm[a]) ? QPC_FABS(W.tm[a]) : a_rp;
      W.sc[WGS_RS] = a_rs; W.sc[WGS_RP] = a_rp;
    }
    QPC_WG_SYNC
    rs = W.sc[WGS_RS]; rp = W.sc[WGS_RP];
    if (sweep == st.max_refine_iters) break;
    // dy = S^-1 (Y_W' r1 - r2) ; dx = H^-1 (r1 - A_W' dy)
    QPC_WG_FOR(a, nw) {
      const int ia = W.idx[a];
      double t = -W.tm[a];
      for (int o = W.rowptr[ia]; o < W.rowptr[ia + 1]; o++) { const int c = W.col[o]; t += W.val[o] * W.hinv[c] * W.r1[c]; }
      W.ax[a] = t;
    }
    QPC_WG_SYNC
    wg_factor_solve(W, W.ax, nw);
    QPC_WG_FOR(i, n) {
      double t = W.r1[i];
      for (int a = 0; a < nw; a++) t -= A[(long long)W.idx[a] * n + i] * W.ax[a];
      W.r1[i] = W.hinv[i] * t; // dx
      W.x[i] += W.r1[i];
    }
    QPC_WG_FOR(a, nw) { W.yw[a] += W.ax[a]; }
    QPC_WG_SYNC
    QPC_WG_T0 {
      double big = 0.0, dbig = 0.0;
      for (int i = 0; i < n; i++) {
        big = QPC_FABS(W.x[i]) > big ? QPC_FABS(W.x[i]) : big;
        dbig = QPC_FABS(W.r1[i]) > dbig ? QPC_FABS(W.r1[i]) : dbig;
      }
      W.sc[WGS_BIG] = big; W.sc[WGS_DBIG] = dbig;
    }
    QPC_WG_SYNC
    if (sweep > 0 && !(W.sc[WGS_DBIG] > 1e-13 * (1.0 + W.sc[WGS_BIG]))) sweep = st.max_refine_iters - 1;
  }
  QPC_WG_T0 {
    double ysc = 1.0, qsc = 1.0;
    for (int a = 0; a < nw; a++) ysc = QPC_FABS(W.yw[a]) > ysc ? QPC_FABS(W.yw[a]) : ysc;
    for (int i = 0; i < n; i++) qsc = QPC_FABS(q[i]) > qsc ? QPC_FABS(q[i]) : qsc;
    W.sc[WGS_YSC] = ysc; W.sc[WGS_XSC] = qsc;
  }
  QPC_WG_SYNC
  {
    const double ysc = W.sc[WGS_YSC], qsc = W.sc[WGS_XSC];
    if (status == QPC_STATUS_OPTIMAL) {
      if (!(rp == rp) || !(rs == rs)) status = QPC_STATUS_NUMERICAL_ERROR;
      else if (rp > 1e-7 * bscale) status = QPC_STATUS_INFEASIBLE;
      else if (rs > 1e-7 * (ysc > qsc ? ysc : qsc)) status = QPC_STATUS_ALMOST_OPTIMAL;
    }
  }
  QPC_WG_FOR(i, n) { x[i] = W.x[i]; }
  QPC_WG_FOR(i, m) { y[i] = 0.0; }
  QPC_WG_SYNC
  QPC_WG_FOR(a, nw) { y[W.idx[a]] = W.yw[a]; }
  QPC_WG_T0 { if (iters_out) iters_out[0] = iters; }
  return status;
}

} // namespace qpc
