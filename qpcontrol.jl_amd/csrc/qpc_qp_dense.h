// qpc_qp_dense.h -- L0 of the C ABI: one dense convex QP in OSQP problem form
//       min 1/2 x'Px + q'x   s.t.   l <= A x <= u
// solved by ONE LANE (64 independent QPs per wavefront, workspace in HBM).  This is the
// MOI-optimizer-level plug of the reference (`Parametron.Model(optimizer)`, momentum.jl:27,58):
// plumbing and parity for arbitrary problems up to the full 143 x 178 reference layout, not the
// performance path -- the fused tick kernel (qpc_tick.h) is.  Same source runs in the emulator.
//
// Method: range-space active set.  H = P + eps I = L L';  Y = H^-1 A';  G = A Y;  for a working
// set W the KKT solve is  y_W = (G_WW)^-1 (A x_unc - b_W),  x = x_unc - Y_W y_W  (infinite-pivot
// Cholesky for dependent active rows), primal-dual sweeps on the working set with a single-change
// fallback, then iterative refinement on the true residuals (removes the eps bias).
// Sign convention of y: OSQP's (P x + q + A'y = 0, y <= 0 on active lower bounds, >= 0 on upper).
#pragma once
#include "../../include/qpc.h"

#ifndef QPC_FN
#error "include qpc_device.h (or the emulator's definitions) first"
#endif

namespace qpc {

#ifdef __HIPCC__
__host__ __device__
#endif
inline long long qp_dense_workspace_doubles(int n, int m) {
  return (long long)n * n + (long long)n * m + 2LL * m * m + 4LL * n + 8LL * m + 16;
}

// in-place Cholesky of the leading k x k block of S (row-major, ld), infinite-pivot rule;
// invd[i] = 1/L_ii or 0 for a dropped (dependent) row
QPC_FN void dense_chol(double *S, int ld, int k, double tau, double *invd) {
  for (int j = 0; j < k; j++) {
    double sjj = S[j * ld + j], d = sjj;
    for (int t = 0; t < j; t++) d -= S[j * ld + t] * S[j * ld + t];
    bool ok = d > tau * (sjj > 1e-300 ? sjj : 1e-300);
    double ld_ = ok ? QPC_SQRT(d) : 0.0;
    invd[j] = ok ? 1.0 / ld_ : 0.0;
    for (int i = j + 1; i < k; i++) {
      double t2 = S[i * ld + j];
      for (int t = 0; t < j; t++) t2 -= S[i * ld + t] * S[j * ld + t];
      S[i * ld + j] = t2 * invd[j];
    }
  }
}
QPC_FN void dense_chol_solve(const double *S, int ld, int k, const double *invd, double *b) {
  for (int i = 0; i < k; i++) {
    double t = b[i];
    for (int j = 0; j < i; j++) t -= S[i * ld + j] * b[j];
    b[i] = t * invd[i];
  }
  for (int i = k - 1; i >= 0; i--) {
    double t = b[i];
    for (int j = i + 1; j < k; j++) t -= S[j * ld + i] * b[j];
    b[i] = t * invd[i];
  }
}

QPC_FN void qp_solve_dense(int n, int m, const double *P, const double *q, const double *A, const double *l,
                           const double *u, const qpc_settings &st, double *ws, double *x, double *y,
                           int32_t *status_out, int32_t *iters_out) {
  const double INF = 1e19;
  double *H = ws;                      // n*n   Cholesky factor of P + eps I
  double *Y = H + (long long)n * n;    // n*m   H^-1 A'   (column i = H^-1 a_i), stored [i*n + k]
  double *G = Y + (long long)n * m;    // m*m   A H^-1 A'
  double *S = G + (long long)m * m;    // m*m   working-set Schur complement / factor
  double *hinvd = S + (long long)m * m; // n
  double *xu = hinvd + n;              // n   unconstrained minimiser -H^-1 q
  double *r1 = xu + n;                 // n
  double *tn = r1 + n;                 // n
  double *g0 = tn + n;                 // m   A xu
  double *yw = g0 + m;                 // m
  double *sinvd = yw + m;              // m
  double *ax = sinvd + m;              // m
  double *bw = ax + m;                 // m
  double *wsf = bw + m;                // m  working set as doubles: 0 / -1 / +1 / 2
  double *idx = wsf + m;               // m  compacted indices
  double *tm = idx + m;                // m
  // ---- H = P + eps I, Cholesky
  double hmax = 0.0;
  for (int i = 0; i < n; i++) { double a = QPC_FABS(P[i * n + i]); hmax = a > hmax ? a : hmax; }
  const double eps = st.eps_primal_reg * (hmax > 0.0 ? hmax : 1.0);
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) H[i * n + j] = 0.5 * (P[i * n + j] + P[j * n + i]) + (i == j ? eps : 0.0);
  dense_chol(H, n, n, 0.0, hinvd);
  for (int i = 0; i < n; i++) xu[i] = -q[i];
  dense_chol_solve(H, n, n, hinvd, xu);
  for (int i = 0; i < m; i++) {
    double *yi = Y + (long long)i * n;
    for (int k = 0; k < n; k++) yi[k] = A[i * n + k];
    dense_chol_solve(H, n, n, hinvd, yi);
  }
  for (int i = 0; i < m; i++) {
    double t = 0.0;
    for (int k = 0; k < n; k++) t += A[i * n + k] * xu[k];
    g0[i] = t;
    for (int j = 0; j <= i; j++) {
      double s = 0.0;
      for (int k = 0; k < n; k++) s += A[i * n + k] * Y[(long long)j * n + k];
      G[i * m + j] = s; G[j * m + i] = s;
    }
  }
  // ---- initial working set: equalities
  double bscale = 1.0;
  for (int i = 0; i < m; i++) {
    bool lo = l[i] > -INF, hi = u[i] < INF;
    wsf[i] = (lo && hi && u[i] - l[i] <= 1e-12 * (1.0 + QPC_FABS(l[i]))) ? 2.0 : 0.0;
    if (lo) bscale = QPC_FABS(l[i]) > bscale ? QPC_FABS(l[i]) : bscale;
    if (hi) bscale = QPC_FABS(u[i]) > bscale ? QPC_FABS(u[i]) : bscale;
  }
  int status = QPC_STATUS_OPTIMAL, iters = 0, nw = 0;
  const int max_it = st.max_active_set_iters > 0 ? st.max_active_set_iters : 200;
  auto solve_ws = [&]() { // EQP of the current working set -> yw (compact), x, ax ; returns nw
    // the equality rows FIRST: the infinite-pivot rule drops the LATER of two dependent rows, and of an equality row and an active bound that
    // say the same thing it is the bound that may go (in the caller's order an equality row behind such a bound was dropped, the answer violated
    // it and the QP was called INFEASIBLE: tests/offline/fuzz_l0_big.py seed 11)
    int k = 0;
    for (int i = 0; i < m; i++) if (wsf[i] == 2.0) { idx[k] = (double)i; k++; }
    for (int i = 0; i < m; i++) if (wsf[i] != 0.0 && wsf[i] != 2.0) { idx[k] = (double)i; k++; }
    for (int a = 0; a < k; a++) {
      int ia = (int)idx[a];
      for (int b = 0; b <= a; b++) S[a * m + b] = G[ia * m + (int)idx[b]];
      bw[a] = wsf[ia] == -1.0 ? l[ia] : u[ia];
      if (wsf[ia] == 2.0) bw[a] = l[ia];
      yw[a] = g0[ia] - bw[a];
    }
    dense_chol(S, m, k, st.eps_dual_reg, sinvd);
    dense_chol_solve(S, m, k, sinvd, yw);
    for (int i = 0; i < n; i++) x[i] = xu[i];
    for (int a = 0; a < k; a++) {
      const double *yi = Y + (long long)((int)idx[a]) * n;
      for (int i = 0; i < n; i++) x[i] -= yi[i] * yw[a];
    }
    for (int i = 0; i < m; i++) {
      double t = g0[i];
      for (int a = 0; a < k; a++) t -= G[i * m + (int)idx[a]] * yw[a];
      ax[i] = t;
    }
    return k;
  };
  nw = solve_ws();
  bool converged = false;
  const int max_pdas = 12;
  // ---- phase A: primal-dual multi-change sweeps (no convergence guarantee, usually 3-6 sweeps)
  for (; iters < max_pdas && iters < max_it; iters++) {
    double ysc = 1.0, xsc = 1.0;
    for (int a = 0; a < nw; a++) ysc = QPC_FABS(yw[a]) > ysc ? QPC_FABS(yw[a]) : ysc;
    for (int i = 0; i < m; i++) xsc = QPC_FABS(ax[i]) > xsc ? QPC_FABS(ax[i]) : xsc;
    const double toly = st.tol_bound * ysc, tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
    int changes = 0;
    for (int i = 0; i < m; i++) tm[i] = 0.0;
    for (int a = 0; a < nw; a++) { // wrong-signed multipliers and dependent (dropped) inequality rows leave
      int i = (int)idx[a];
      double w = wsf[i];
      double viol = (w == -1.0) ? yw[a] : ((w == 1.0) ? -yw[a] : 0.0);
      if (viol > toly || (w != 2.0 && sinvd[a] == 0.0)) changes++;
      else tm[i] = w;
    }
    for (int i = 0; i < m; i++) {
      if (wsf[i] != 0.0) continue;
      double lo = l[i] > -INF ? l[i] - ax[i] : -1.0, hi = u[i] < INF ? ax[i] - u[i] : -1.0;
      if (lo > tolx && lo >= hi) { tm[i] = -1.0; changes++; }
      else if (hi > tolx) { tm[i] = 1.0; changes++; }
    }
    if (changes == 0) { converged = true; break; }
    for (int i = 0; i < m; i++) wsf[i] = tm[i];
    nw = solve_ws();
  }
  // ---- phase B: Goldfarb-Idnani dual active set from the equality-only working set (finite)
  if (!converged) {
    for (int i = 0; i < m; i++) wsf[i] = wsf[i] == 2.0 ? 2.0 : 0.0;
    nw = solve_ws();
    for (; iters < max_it; iters++) {
      double xsc = 1.0;
      for (int i = 0; i < m; i++) xsc = QPC_FABS(ax[i]) > xsc ? QPC_FABS(ax[i]) : xsc;
      const double tolx = st.tol_bound * (xsc > bscale ? xsc : bscale);
      int p = -1;
      double viol = tolx, sg = 0.0;
      for (int i = 0; i < m; i++) {
        if (wsf[i] != 0.0) continue;
        double lo = l[i] > -INF ? l[i] - ax[i] : -1.0, hi = u[i] < INF ? ax[i] - u[i] : -1.0;
        if (lo > viol) { viol = lo; p = i; sg = -1.0; }
        if (hi > viol) { viol = hi; p = i; sg = 1.0; }
      }
      if (p < 0) { converged = true; break; }
      bool infeasible = false;
      for (int inner = 0; inner <= m; inner++) {
        // r = (G_WW)^-1 G_Wp : a_p in terms of the working rows (H^-1 metric)
        int k = 0; // (the order of solve_ws: equality rows first)
        for (int i = 0; i < m; i++) if (wsf[i] == 2.0) { idx[k] = (double)i; k++; }
        for (int i = 0; i < m; i++) if (wsf[i] != 0.0 && wsf[i] != 2.0) { idx[k] = (double)i; k++; }
        for (int a = 0; a < k; a++) {
          int ia = (int)idx[a];
          for (int b = 0; b <= a; b++) S[a * m + b] = G[ia * m + (int)idx[b]];
          tm[a] = G[ia * m + p];
        }
        dense_chol(S, m, k, st.eps_dual_reg, sinvd);
        dense_chol_solve(S, m, k, sinvd, tm);
        double d = G[p * m + p];
        for (int a = 0; a < k; a++) d -= G[p * m + (int)idx[a]] * tm[a];
        const bool dependent = !(d > 1e-11 * G[p * m + p]);
        // dual ratio test: y_W(t) = y_W - t sg r must keep the sign of every working inequality
        double t1 = 1e300;
        int kd = -1;
        for (int a = 0; a < k; a++) {
          double w = wsf[(int)idx[a]], sr = sg * tm[a];
          if (w == 2.0 || sinvd[a] == 0.0) continue;
          if ((w == 1.0 && sr > 1e-14) || (w == -1.0 && sr < -1e-14)) {
            double t = yw[a] / sr;
            t = t > 0.0 ? t : 0.0;
            if (t < t1) { t1 = t; kd = a; }
          }
        }
        if (dependent && kd < 0) { infeasible = true; break; }
        const double t2 = dependent ? 1e300 : viol / d;
        if (t2 <= t1) break; // full step: p enters
        // partial step: working row kd leaves
        for (int a = 0; a < k; a++) yw[a] -= t1 * sg * tm[a];
        if (!dependent) viol -= t1 * d;
        wsf[(int)idx[kd]] = 0.0;
        for (int a = kd; a + 1 < k; a++) yw[a] = yw[a + 1];
      }
      if (infeasible) { status = QPC_STATUS_INFEASIBLE; break; }
      wsf[p] = sg;
      nw = solve_ws();
    }
  }
  if (!converged && status == QPC_STATUS_OPTIMAL) status = QPC_STATUS_ITERATION_LIMIT;
  // ---- iterative refinement on the true KKT residuals of the final working set
  double rp = 0.0, rs = 0.0;
  for (int sweep = 0; sweep <= st.max_refine_iters; sweep++) {
    for (int i = 0; i < n; i++) {
      double t = q[i];
      for (int k = 0; k < n; k++) t += 0.5 * (P[i * n + k] + P[k * n + i]) * x[k];
      r1[i] = -t;
    }
    for (int a = 0; a < nw; a++) {
      int ia = (int)idx[a];
      for (int i = 0; i < n; i++) r1[i] -= A[ia * n + i] * yw[a];
    }
    rs = 0.0; rp = 0.0;
    for (int i = 0; i < n; i++) rs = QPC_FABS(r1[i]) > rs ? QPC_FABS(r1[i]) : rs;
    for (int a = 0; a < nw; a++) {
      int ia = (int)idx[a];
      double t = bw[a];
      for (int i = 0; i < n; i++) t -= A[ia * n + i] * x[i];
      tm[a] = t; // r2
      if (sinvd[a] != 0.0) rp = QPC_FABS(t) > rp ? QPC_FABS(t) : rp;
      else { // dropped (dependent) row: consistent only if its residual vanishes too
        rp = QPC_FABS(t) > rp ? QPC_FABS(t) : rp;
      }
    }
    if (sweep == st.max_refine_iters) break;
    // dy = S^-1 (Y_W' r1 - r2) ; dx = H^-1 r1 - Y_W dy
    for (int a = 0; a < nw; a++) {
      const double *yi = Y + (long long)((int)idx[a]) * n;
      double t = -tm[a];
      for (int i = 0; i < n; i++) t += yi[i] * r1[i];
      ax[a] = t;
    }
    dense_chol_solve(S, m, nw, sinvd, ax);
    for (int i = 0; i < n; i++) tn[i] = r1[i];
    dense_chol_solve(H, n, n, hinvd, tn);
    for (int a = 0; a < nw; a++) {
      const double *yi = Y + (long long)((int)idx[a]) * n;
      for (int i = 0; i < n; i++) tn[i] -= yi[i] * ax[a];
      yw[a] += ax[a];
    }
    double big = 0.0, dbig = 0.0;
    for (int i = 0; i < n; i++) {
      x[i] += tn[i];
      big = QPC_FABS(x[i]) > big ? QPC_FABS(x[i]) : big;
      dbig = QPC_FABS(tn[i]) > dbig ? QPC_FABS(tn[i]) : dbig;
    }
    if (sweep > 0 && !(dbig > 1e-13 * (1.0 + big))) { sweep = st.max_refine_iters - 1; }
  }
  double ysc = 1.0, qsc = 1.0;
  for (int a = 0; a < nw; a++) ysc = QPC_FABS(yw[a]) > ysc ? QPC_FABS(yw[a]) : ysc;
  for (int i = 0; i < n; i++) qsc = QPC_FABS(q[i]) > qsc ? QPC_FABS(q[i]) : qsc;
  if (status == QPC_STATUS_OPTIMAL) {
    if (!(rp == rp) || !(rs == rs)) status = QPC_STATUS_NUMERICAL_ERROR;
    else if (rp > 1e-7 * bscale) status = QPC_STATUS_INFEASIBLE;
    else if (rs > 1e-7 * (ysc > qsc ? ysc : qsc)) status = QPC_STATUS_ALMOST_OPTIMAL;
  }
  for (int i = 0; i < m; i++) y[i] = 0.0;
  for (int a = 0; a < nw; a++) y[(int)idx[a]] = yw[a];
  status_out[0] = status;
  if (iters_out) iters_out[0] = iters;
}

} // namespace qpc
