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
  return QPC_WG_KMAX * (QPC_WG_KMAX + 1) / 2 + QPC_WG_NNZ + QPC_WG_NNZ / 2 + 4 * QPC_WG_DIM + 12 * QPC_WG_DIM + 64;
}

struct QpWg {
  int n, m;
  const double *P, *q, *A, *l, *u;
  double *G;      // m x m in global memory (the workspace of qp_solve_dense keeps its G at the same place)
  // LDS
  double *S;      // packed lower triangle of the working-set matrix / its Cholesky factor
  double *val;    // nonzeros of A, row by row
  int *col, *rowptr;
  double *hinv, *xu, *x, *r1;              // n
  double *g0, *yw, *bw, *ax, *wsf, *sinvd, *tm, *sdiag; // m
  int *idx;       // compacted working set
  double *sc;     // scalars
};

enum { WGS_FLAG = 0, WGS_HMAX, WGS_EPS, WGS_BSCALE, WGS_K, WGS_CHANGES, WGS_YSC, WGS_XSC, WGS_RP, WGS_RS, WGS_BIG, WGS_DBIG, WGS_T, WGS_P, WGS_VIOL, WGS_SG,
       WGS_COUNT };

QPC_FN int wg_sidx(int i, int j) { return i * (i + 1) / 2 + j; } // packed lower triangle, i >= j

// Factor the working set in W.wsf and solve one system with it: compaction -> idx, S = G_WW, Cholesky with the right-hand side riding
// along, back-substitution -> out (compact).  Right-hand side: pcol < 0: g0_W - b_W (the EQP of the working set; b -> W.bw);
// pcol >= 0: column pcol of G (the Goldfarb-Idnani direction of row pcol).  Returns the number of working rows, or -1 beyond QPC_WG_KMAX.
QPC_FN int wg_factor(QpWg &W, double tau, int pcol, double *out) {
  const int m = W.m;
  QPC_WG_T0 {
    int k = 0;
    for (int i = 0; i < m; i++) if (W.wsf[i] != 0.0) { if (k < QPC_WG_KMAX) W.idx[k] = i; k++; }
    W.sc[WGS_K] = (double)k;
  }
  QPC_WG_SYNC
  const int k = (int)W.sc[WGS_K];
  if (k > QPC_WG_KMAX) return -1;
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
  QPC_WG_SYNC
  for (int j = 0; j < k; j++) { // right-looking Cholesky, infinite-pivot rule; yw <- L^-1 yw on the way
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
  for (int j = k - 1; j >= 0; j--) { // L' y = z
    QPC_WG_T0 { out[j] = out[j] * W.sinvd[j]; }
    QPC_WG_SYNC
    const double yj = out[j];
    QPC_WG_FOR(i, j) { out[i] -= W.S[wg_sidx(j, i)] * yj; }
    QPC_WG_SYNC
  }
  return k;
}
// EQP of the working set: multipliers -> yw (compact), ax = A x for every row
QPC_FN int wg_solve_ws(QpWg &W, double tau) {
  const int m = W.m;
  const int k = wg_factor(W, tau, -1, W.yw);
  if (k < 0) return -1;
  QPC_WG_FOR(i, m) {
    double t = W.g0[i];
    for (int a = 0; a < k; a++) t -= W.G[(long long)i * m + W.idx[a]] * W.yw[a];
    W.ax[i] = t;
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
                       const qpc_settings &st, double *Gglobal, double *lds, double *x, double *y, int32_t *iters_out) {
  const double INF = 1e19;
  if (n > QPC_WG_DIM || m > QPC_WG_DIM || n < 1) return QPC_WG_FALLBACK;
  QpWg W;
  W.n = n; W.m = m; W.P = P; W.q = q; W.A = A; W.l = l; W.u = u; W.G = Gglobal;
  double *p = lds;
  W.S = p; p += QPC_WG_KMAX * (QPC_WG_KMAX + 1) / 2;
  W.val = p; p += QPC_WG_NNZ;
  W.col = (int *)p; p += QPC_WG_NNZ / 2;
  W.hinv = p; p += QPC_WG_DIM; W.xu = p; p += QPC_WG_DIM; W.x = p; p += QPC_WG_DIM; W.r1 = p; p += QPC_WG_DIM;
  W.g0 = p; p += QPC_WG_DIM; W.yw = p; p += QPC_WG_DIM; W.bw = p; p += QPC_WG_DIM; W.ax = p; p += QPC_WG_DIM;
  W.wsf = p; p += QPC_WG_DIM; W.sinvd = p; p += QPC_WG_DIM; W.tm = p; p += QPC_WG_DIM; W.sdiag = p; p += QPC_WG_DIM;
  W.idx = (int *)p; p += QPC_WG_DIM; // (two ints per double: idx and rowptr share two blocks)
  W.rowptr = (int *)p; p += QPC_WG_DIM;
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
  int nw = wg_solve_ws(W, st.eps_dual_reg);
  if (nw < 0) return QPC_WG_FALLBACK;
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
        const int k = wg_factor(W, st.eps_dual_reg, pr, W.tm); // r = (G_WW)^-1 G_Wp : a_p in terms of the working rows (H^-1 metric)
        if (k < 0) { bail = true; break; }
        QPC_WG_T0 {
          double viol = W.sc[WGS_VIOL];
          double d = W.G[(long long)pr * m + pr];
          for (int a = 0; a < k; a++) d -= W.G[(long long)pr * m + W.idx[a]] * W.tm[a];
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
