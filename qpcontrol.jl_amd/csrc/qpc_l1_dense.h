// qpc_l1_dense.h -- L1 of the C ABI: assembly + solve from HOST-EVALUATED blocks,
// i.e. `solve!` + `value` of the reference (momentum.jl:58-72) with the kinematics taken out.  The
// caller supplies, per QP, what task_error / ContactPoint / add_wrench_balance_constraint! evaluate
// from the state: task Jacobians and biases (src/tasks.jl:31-262), the world-frame momentum matrix
// and right-hand side of the wrench balance (momentum.jl:162-193), and the world pose of every
// contact frame (contacts.jl:61).  ONE LANE assembles the QP densely in the variables
//       x = [v̇ (nv); e (slacks of the soft tasks); rho (nc * N)]
// (f_local and wrench_world of the reference layout are linear maps of rho and are substituted) and
// solves it with the dense active-set code of L0 (qpc_qp_dense.h).  It shares nothing with the
// fused tick kernel (no pins, no folding, no Schur reduction), which is the point: an independent
// device-side assembly to check the fused one against, and a parity path that isolates assembly +
// solve from kinematics.  Plumbing, not the performance path.
#pragma once
#include "qpc_qp_dense.h"

namespace qpc {

// static description of the controller for this path (built by the host per call, tiny)
//   ints   : [0] nv [1] ntask [2] nc [3] N [4] floating(0/1) [5] ne [6] n_desired, then per task
//            {dim, wkind, des_off, slack0 (-1 hard), woff (offset in dbl W area), tw_off}, then per contact {surf_off or -1}
//   doubles: reg[nv], then W area (static weights: 1 or dim*dim per soft task), then mu[nc]
struct L1Prog {
  const int *it;
  const double *dt;
};

#ifdef __HIPCC__
__host__ __device__
#endif
inline int l1_n(int nv, int ne, int nc, int N) { return nv + ne + nc * N; }
#ifdef __HIPCC__
__host__ __device__
#endif
inline int l1_m(int n_desired, int floating, int nc, int N) { return n_desired + (floating ? 6 : 0) + nc * N; }
#ifdef __HIPCC__
__host__ __device__
#endif
inline long long l1_workspace_doubles(int n, int m) {
  return (long long)n * n + n + (long long)m * n + 2LL * m + n + m + qp_dense_workspace_doubles(n, m) + 8;
}

QPC_FN void l1_assemble_solve(const L1Prog &Pg, const double *J, const double *bias, const double *desired,
                              const double *Am, const double *mrhs, const double *cR, const double *cp,
                              const double *cw, const double *cmax, const double *tw, const double *surf,
                              const qpc_settings &st, double *ws, double *vdot, double *cwrench, int32_t *status,
                              int32_t *iters) {
  const int *it = Pg.it;
  const double *dt = Pg.dt;
  const int nv = it[0], ntask = it[1], nc = it[2], N = it[3], floating = it[4], ne = it[5], nd = it[6];
  const int nrho = nc * N, n = l1_n(nv, ne, nc, N), m = l1_m(nd, floating, nc, N);
  const double *reg = dt, *Wst = dt + nv;
  double *P = ws, *q = P + (long long)n * n, *A = q + n, *l = A + (long long)m * n, *u = l + m, *x = u + m, *y = x + n,
         *dws = y + m;
  for (long long i = 0; i < (long long)n * n; i++) P[i] = 0.0;
  for (long long i = 0; i < (long long)m * n; i++) A[i] = 0.0;
  for (int i = 0; i < n; i++) q[i] = 0.0;
  for (int k = 0; k < nv; k++) P[k * n + k] = 2.0 * reg[k]; // regularize! (momentum.jl:128-131), P = 2 x written weights
  const int *tk = it + 7;
  for (int t = 0; t < ntask; t++) { // addtask! (momentum.jl:99-117): hard -> equality rows, soft -> slack e == err, obj e'We
    const int dim = tk[6 * t], wk = tk[6 * t + 1], d0 = tk[6 * t + 2], s0 = tk[6 * t + 3], wo = tk[6 * t + 4], two = tk[6 * t + 5];
    for (int a = 0; a < dim; a++) {
      const int r = d0 + a;
      for (int k = 0; k < nv; k++) A[(long long)r * n + k] = J[(long long)r * nv + k];
      if (s0 >= 0) A[(long long)r * n + nv + s0 + a] = -1.0;
      l[r] = u[r] = desired[r] - bias[r];
    }
    if (s0 >= 0)
      for (int a = 0; a < dim; a++)
        for (int b = 0; b < dim; b++) {
          double w;
          if (wk == QPC_WEIGHT_SCALAR) w = (a == b) ? 2.0 * Wst[wo] : 0.0;
          else if (wk == QPC_WEIGHT_MATRIX) w = Wst[wo + a * dim + b] + Wst[wo + b * dim + a];
          else if (wk == QPC_WEIGHT_SCALAR_PER_QP) w = (a == b) ? 2.0 * tw[two] : 0.0;
          else w = tw[two + a * dim + b] + tw[two + b * dim + a];
          P[(long long)(nv + s0 + a) * n + nv + s0 + b] = w;
        }
  }
  const int R0 = nd, R1 = nd + (floating ? 6 : 0), c0 = nv + ne;
  const double *mu_st = Wst + it[7 + 6 * ntask + nc]; // W area size stored after the contact table
  const int *ck = it + 7 + 6 * ntask;
  for (int c = 0; c < nc; c++) { // ContactPoint (contacts.jl:38-69), objectiveterm (contacts.jl:75-79)
    const double mu = ck[c] >= 0 ? surf[ck[c] + 3] : mu_st[c];
    const double inv = 1.0 / QPC_SQRT(mu * mu + 1.0);
    const double *R = cR + 9 * c, *p = cp + 3 * c;
    double Bm[3][8];
    for (int i = 0; i < N; i++) { // forcebasis(mu, N) (contacts.jl:16-23)
      double sn, cs;
      QPC_SINCOS(i * (6.283185307179586 / N), sn, cs);
      Bm[0][i] = mu * cs * inv; Bm[1][i] = mu * sn * inv; Bm[2][i] = inv;
    }
    for (int i = 0; i < N; i++) {
      const int j = c * N + i;
      double lin[3], ang[3];
      for (int a = 0; a < 3; a++) lin[a] = R[3 * a] * Bm[0][i] + R[3 * a + 1] * Bm[1][i] + R[3 * a + 2] * Bm[2][i];
      ang[0] = p[1] * lin[2] - p[2] * lin[1]; ang[1] = p[2] * lin[0] - p[0] * lin[2]; ang[2] = p[0] * lin[1] - p[1] * lin[0];
      if (floating)
        for (int a = 0; a < 3; a++) { A[(long long)(R0 + a) * n + c0 + j] = -ang[a]; A[(long long)(R0 + 3 + a) * n + c0 + j] = -lin[a]; }
      for (int i2 = 0; i2 < N; i2++)
        P[(long long)(c0 + j) * n + c0 + c * N + i2] = 2.0 * cw[c] * (Bm[0][i] * Bm[0][i2] + Bm[1][i] * Bm[1][i2] + Bm[2][i] * Bm[2][i2]);
      A[(long long)(R1 + j) * n + c0 + j] = 1.0;
      l[R1 + j] = 0.0;
      u[R1 + j] = cmax[c] * inv / N; // maxrho (contacts.jl:57); disable! = 0 (contacts.jl:72)
    }
  }
  if (floating) // add_wrench_balance_constraint! (momentum.jl:162-193), invertible S' dropped
    for (int a = 0; a < 6; a++) {
      for (int k = 0; k < nv; k++) A[(long long)(R0 + a) * n + k] = Am[a * nv + k];
      l[R0 + a] = u[R0 + a] = mrhs[a];
    }
  qp_solve_dense(n, m, P, q, A, l, u, st, dws, x, y, status, iters);
  for (int k = 0; k < nv; k++) vdot[k] = x[k];
  for (int c = 0; c < nc; c++) { // value(wrench_world) (contacts.jl:82-89)
    for (int a = 0; a < 6; a++) {
      double acc = 0.0;
      for (int i = 0; i < N; i++) {
        const int j = c * N + i;
        // wrench column of rho_j: minus the wrench-balance coefficient when floating, else rebuilt
        acc += x[c0 + j] * (floating ? -A[(long long)(R0 + a) * n + c0 + j] : 0.0);
      }
      cwrench[6 * c + a] = acc;
    }
    if (!floating) { // no wrench rows to read the basis back from
      const double mu = ck[c] >= 0 ? surf[ck[c] + 3] : mu_st[c];
      const double inv = 1.0 / QPC_SQRT(mu * mu + 1.0);
      const double *R = cR + 9 * c, *p = cp + 3 * c;
      double f[3] = {0, 0, 0};
      for (int i = 0; i < N; i++) {
        double sn, cs;
        QPC_SINCOS(i * (6.283185307179586 / N), sn, cs);
        const double b[3] = {mu * cs * inv, mu * sn * inv, inv}, r = x[c0 + c * N + i];
        for (int a = 0; a < 3; a++) f[a] += r * (R[3 * a] * b[0] + R[3 * a + 1] * b[1] + R[3 * a + 2] * b[2]);
      }
      cwrench[6 * c + 0] = p[1] * f[2] - p[2] * f[1]; cwrench[6 * c + 1] = p[2] * f[0] - p[0] * f[2];
      cwrench[6 * c + 2] = p[0] * f[1] - p[1] * f[0];
      cwrench[6 * c + 3] = f[0]; cwrench[6 * c + 4] = f[1]; cwrench[6 * c + 5] = f[2];
    }
  }
}

} // namespace qpc
