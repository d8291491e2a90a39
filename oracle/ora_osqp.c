/*
 * ora_osqp.c -- oracle: restatement of the OSQP algorithm (ADMM for convex QPs), the solver
 * behind every reference test and notebook (test/runtests.jl:14-15,35-43; notebooks/Standing
 * controller.ipynb cell 4) and reached through Parametron.solve! at momentum.jl:58.
 *
 * OSQP.jl 0.5.2 / OSQP 0.5.0 are NOT in this image ([recalled]); this follows the published
 * algorithm -- Stellato, Banjac, Goulart, Bemporad, Boyd, "OSQP: an operator splitting solver
 * for quadratic programs", Math. Prog. Comp. 12 (2020): Algorithm 1, sections 3.1 (quasi-definite
 * KKT), 3.4 (termination), 3.5 (infeasibility), 5.1 (Ruiz equilibration), 5.2 (adaptive rho,
 * per-constraint rho with 1e3 on equalities), 4 (polish) -- with the 0.5.0 default constants.
 *
 * Problem form: min 1/2 x'Px + q'x  s.t.  l <= Ax <= u.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "ora_osqp.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OSQP_INFTY 1e20
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_EQ_FACTOR 1e3
#define RHO_TOL 1e-4

void ora_osqp_default_settings(ora_osqp_settings *s) {
  s->rho = 0.1; s->sigma = 1e-6; s->alpha = 1.6;
  s->eps_abs = 1e-3; s->eps_rel = 1e-3; s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
  s->max_iter = 4000; s->scaling = 10; s->adaptive_rho = 1; s->adaptive_rho_interval = 0;
  s->check_termination = 25; s->warm_start = 1; s->adaptive_rho_tolerance = 5.0; s->polish = 0;
}
/* defaultoptimizer(), test/runtests.jl:35-43 */
void ora_osqp_test_settings(ora_osqp_settings *s) {
  ora_osqp_default_settings(s);
  s->eps_abs = 1e-8; s->eps_rel = 1e-16; s->max_iter = 20000; s->adaptive_rho_interval = 25;
}
/* notebooks/Standing controller.ipynb cell 4 (lines 64-71 of the raw file) */
void ora_osqp_notebook_settings(ora_osqp_settings *s) {
  ora_osqp_default_settings(s);
  s->eps_abs = 1e-5; s->eps_rel = 1e-5; s->max_iter = 5000; s->adaptive_rho_interval = 25;
}

static double *dalloc(int n) { return (double *)calloc(n > 0 ? n : 1, sizeof(double)); }
static int *ialloc(int n) { return (int *)calloc(n > 0 ? n : 1, sizeof(int)); }
static double vmaxabs(const double *v, int n) {
  double m = 0;
  for (int i = 0; i < n; i++) { double a = fabs(v[i]); if (a > m) m = a; }
  return m;
}

/* y = A x (CSC) */
static void A_mul(const ora_osqp *w, const double *Ax, const double *x, double *y) {
  for (int i = 0; i < w->m; i++) y[i] = 0;
  for (int j = 0; j < w->n; j++)
    for (int p = w->Ap[j]; p < w->Ap[j + 1]; p++) y[w->Ai[p]] += Ax[p] * x[j];
}
/* x = A' y */
static void At_mul(const ora_osqp *w, const double *Ax, const double *y, double *x) {
  for (int j = 0; j < w->n; j++) {
    double t = 0;
    for (int p = w->Ap[j]; p < w->Ap[j + 1]; p++) t += Ax[p] * y[w->Ai[p]];
    x[j] = t;
  }
}
/* y = P x, P symmetric stored upper */
static void P_mul(const ora_osqp *w, const double *Px, const double *x, double *y) {
  for (int i = 0; i < w->n; i++) y[i] = 0;
  for (int j = 0; j < w->n; j++)
    for (int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
      int i = w->Pi[p];
      y[i] += Px[p] * x[j];
      if (i != j) y[j] += Px[p] * x[i];
    }
}

ora_osqp *ora_osqp_setup(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                         const ora_osqp_settings *st) {
  ora_osqp *w = (ora_osqp *)calloc(1, sizeof(ora_osqp));
  w->n = n; w->m = m; w->st = *st;
  w->Pnnz = Pp[n]; w->Annz = Ap[n];
  w->Pp = ialloc(n + 1); memcpy(w->Pp, Pp, sizeof(int) * (n + 1));
  w->Pi = ialloc(w->Pnnz); memcpy(w->Pi, Pi, sizeof(int) * w->Pnnz);
  w->Ap = ialloc(n + 1); memcpy(w->Ap, Ap, sizeof(int) * (n + 1));
  w->Ai = ialloc(w->Annz); memcpy(w->Ai, Ai, sizeof(int) * w->Annz);
  w->Px0 = dalloc(w->Pnnz); w->Px = dalloc(w->Pnnz);
  w->Ax0 = dalloc(w->Annz); w->Ax = dalloc(w->Annz);
  w->q0 = dalloc(n); w->q = dalloc(n);
  w->l0 = dalloc(m); w->l = dalloc(m); w->u0 = dalloc(m); w->u = dalloc(m);
  w->D = dalloc(n); w->Dinv = dalloc(n); w->E = dalloc(m); w->Einv = dalloc(m);
  w->rho_vec = dalloc(m); w->rho_inv_vec = dalloc(m); w->ctype = ialloc(m);
  w->x = dalloc(n); w->xt = dalloc(n); w->xprev = dalloc(n); w->dx = dalloc(n);
  w->y = dalloc(m); w->z = dalloc(m); w->zt = dalloc(m); w->zprev = dalloc(m); w->dy = dalloc(m);
  w->rhs = dalloc(n + m); w->tmpn = dalloc(n); w->tmpn2 = dalloc(n);
  w->tmpm = dalloc(m); w->tmpm2 = dalloc(m);
  w->rho = st->rho;

  /* ---- KKT pattern [P+sigma I, A'; A, -1/rho] upper triangular, then permute ---- */
  int N = n + m;
  int ntrip = w->Pnnz + n + w->Annz + m;
  int *tr = ialloc(ntrip), *tc = ialloc(ntrip), *tsrc = ialloc(ntrip);
  int t = 0;
  for (int j = 0; j < n; j++)
    for (int p = Pp[j]; p < Pp[j + 1]; p++) { tr[t] = Pi[p]; tc[t] = j; tsrc[t] = p; t++; }
  for (int i = 0; i < n; i++) { tr[t] = i; tc[t] = i; tsrc[t] = w->Pnnz + i; t++; }
  for (int j = 0; j < n; j++)
    for (int p = Ap[j]; p < Ap[j + 1]; p++) { tr[t] = j; tc[t] = n + Ai[p]; tsrc[t] = w->Pnnz + n + p; t++; }
  for (int i = 0; i < m; i++) { tr[t] = n + i; tc[t] = n + i; tsrc[t] = w->Pnnz + n + w->Annz + i; t++; }
  /* unpermuted upper CSC for the ordering */
  {
    int *cnt = ialloc(N + 1), *Kp0 = ialloc(N + 1), *Ki0 = ialloc(ntrip);
    for (int k = 0; k < ntrip; k++) cnt[tc[k] + 1]++;
    for (int j = 0; j < N; j++) Kp0[j + 1] = Kp0[j] + cnt[j + 1];
    int *pos = ialloc(N);
    for (int j = 0; j < N; j++) pos[j] = Kp0[j];
    for (int k = 0; k < ntrip; k++) Ki0[pos[tc[k]]++] = tr[k];
    w->perm = ialloc(N); w->pinv = ialloc(N);
    ora_min_degree(N, Kp0, Ki0, w->perm);
    for (int k = 0; k < N; k++) w->pinv[w->perm[k]] = k;
    free(cnt); free(Kp0); free(Ki0); free(pos);
  }
  /* permuted triplets -> CSC with duplicate merging */
  for (int k = 0; k < ntrip; k++) {
    int r = w->pinv[tr[k]], c = w->pinv[tc[k]];
    if (r > c) { int s = r; r = c; c = s; }
    tr[k] = r; tc[k] = c;
  }
  int *order = ialloc(ntrip);
  for (int k = 0; k < ntrip; k++) order[k] = k;
  /* sort by (col,row): simple counting by col then insertion by row */
  {
    int *cnt = ialloc(N + 1), *start = ialloc(N + 1), *tmp = ialloc(ntrip);
    for (int k = 0; k < ntrip; k++) cnt[tc[k] + 1]++;
    for (int j = 0; j < N; j++) start[j + 1] = start[j] + cnt[j + 1];
    int *pos = ialloc(N);
    for (int j = 0; j < N; j++) pos[j] = start[j];
    for (int k = 0; k < ntrip; k++) tmp[pos[tc[k]]++] = k;
    for (int j = 0; j < N; j++)
      for (int a = start[j] + 1; a < start[j + 1]; a++) {
        int key = tmp[a], b = a - 1;
        while (b >= start[j] && tr[tmp[b]] > tr[key]) { tmp[b + 1] = tmp[b]; b--; }
        tmp[b + 1] = key;
      }
    memcpy(order, tmp, sizeof(int) * ntrip);
    free(cnt); free(start); free(tmp); free(pos);
  }
  w->Kp = ialloc(N + 1); w->Ki = ialloc(ntrip); w->Kx = dalloc(ntrip);
  int *srcToK = ialloc(ntrip);
  int nk = 0, lastr = -1, lastc = -1;
  for (int a = 0; a < ntrip; a++) {
    int k = order[a];
    if (tr[k] != lastr || tc[k] != lastc) {
      w->Ki[nk] = tr[k];
      w->Kp[tc[k] + 1]++;
      lastr = tr[k]; lastc = tc[k];
      nk++;
    }
    srcToK[tsrc[k]] = nk - 1;
  }
  for (int j = 0; j < N; j++) w->Kp[j + 1] += w->Kp[j];
  w->nk = nk;
  w->PtoK = ialloc(w->Pnnz); w->sigtoK = ialloc(n); w->AtoK = ialloc(w->Annz); w->rhotoK = ialloc(m);
  memcpy(w->PtoK, srcToK, sizeof(int) * w->Pnnz);
  memcpy(w->sigtoK, srcToK + w->Pnnz, sizeof(int) * n);
  memcpy(w->AtoK, srcToK + w->Pnnz + n, sizeof(int) * w->Annz);
  memcpy(w->rhotoK, srcToK + w->Pnnz + n + w->Annz, sizeof(int) * m);
  free(tr); free(tc); free(tsrc); free(order); free(srcToK);
  if (ora_ldl_symbolic(&w->F, N, w->Kp, w->Ki) != 0) { ora_osqp_free(w); return NULL; }
  return w;
}

void ora_osqp_free(ora_osqp *w) {
  if (!w) return;
  free(w->Pp); free(w->Pi); free(w->Ap); free(w->Ai);
  free(w->Px0); free(w->Px); free(w->Ax0); free(w->Ax); free(w->q0); free(w->q);
  free(w->l0); free(w->l); free(w->u0); free(w->u);
  free(w->D); free(w->Dinv); free(w->E); free(w->Einv);
  free(w->rho_vec); free(w->rho_inv_vec); free(w->ctype);
  free(w->x); free(w->xt); free(w->xprev); free(w->dx);
  free(w->y); free(w->z); free(w->zt); free(w->zprev); free(w->dy);
  free(w->rhs); free(w->tmpn); free(w->tmpn2); free(w->tmpm); free(w->tmpm2);
  free(w->Kp); free(w->Ki); free(w->Kx); free(w->PtoK); free(w->AtoK); free(w->sigtoK);
  free(w->rhotoK); free(w->perm); free(w->pinv);
  if (w->F.n) ora_ldl_free(&w->F);
  free(w);
}

static double limit_scaling(double v) {
  if (v < MIN_SCALING) return 1.0;
  if (v > MAX_SCALING) return MAX_SCALING;
  return v;
}

/* Ruiz equilibration + cost scaling (paper section 5.1) on w->Px, w->Ax, w->q, w->l, w->u */
static void scale_data(ora_osqp *w) {
  int n = w->n, m = w->m;
  for (int i = 0; i < n; i++) w->D[i] = 1.0;
  for (int i = 0; i < m; i++) w->E[i] = 1.0;
  w->c = 1.0;
  double *dn = w->tmpn, *em = w->tmpm;
  for (int it = 0; it < w->st.scaling; it++) {
    for (int i = 0; i < n; i++) dn[i] = 0;
    for (int i = 0; i < m; i++) em[i] = 0;
    for (int j = 0; j < n; j++) {
      for (int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
        double a = fabs(w->Px[p]);
        int i = w->Pi[p];
        if (a > dn[j]) dn[j] = a;
        if (a > dn[i]) dn[i] = a;
      }
      for (int p = w->Ap[j]; p < w->Ap[j + 1]; p++) {
        double a = fabs(w->Ax[p]);
        if (a > dn[j]) dn[j] = a;
        if (a > em[w->Ai[p]]) em[w->Ai[p]] = a;
      }
    }
    for (int i = 0; i < n; i++) dn[i] = 1.0 / sqrt(limit_scaling(dn[i]));
    for (int i = 0; i < m; i++) em[i] = 1.0 / sqrt(limit_scaling(em[i]));
    for (int j = 0; j < n; j++) {
      for (int p = w->Pp[j]; p < w->Pp[j + 1]; p++) w->Px[p] *= dn[j] * dn[w->Pi[p]];
      for (int p = w->Ap[j]; p < w->Ap[j + 1]; p++) w->Ax[p] *= dn[j] * em[w->Ai[p]];
      w->q[j] *= dn[j];
      w->D[j] *= dn[j];
    }
    for (int i = 0; i < m; i++) w->E[i] *= em[i];
    /* cost scaling */
    for (int i = 0; i < n; i++) dn[i] = 0;
    for (int j = 0; j < n; j++)
      for (int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
        double a = fabs(w->Px[p]);
        int i = w->Pi[p];
        if (a > dn[j]) dn[j] = a;
        if (a > dn[i]) dn[i] = a;
      }
    double cmean = 0;
    for (int i = 0; i < n; i++) cmean += dn[i];
    cmean /= (n > 0 ? n : 1);
    double qn = limit_scaling(vmaxabs(w->q, n));
    double ct = cmean > qn ? cmean : qn;
    ct = 1.0 / limit_scaling(ct);
    for (int p = 0; p < w->Pnnz; p++) w->Px[p] *= ct;
    for (int i = 0; i < n; i++) w->q[i] *= ct;
    w->c *= ct;
  }
  for (int i = 0; i < n; i++) w->Dinv[i] = 1.0 / w->D[i];
  for (int i = 0; i < m; i++) w->Einv[i] = 1.0 / w->E[i];
  w->cinv = 1.0 / w->c;
  for (int i = 0; i < m; i++) {
    w->l[i] = w->l0[i] <= -OSQP_INFTY ? -OSQP_INFTY : w->l0[i] * w->E[i];
    w->u[i] = w->u0[i] >= OSQP_INFTY ? OSQP_INFTY : w->u0[i] * w->E[i];
  }
}

static void set_rho_vec(ora_osqp *w) {
  for (int i = 0; i < w->m; i++) {
    if (w->l[i] <= -OSQP_INFTY && w->u[i] >= OSQP_INFTY) { w->ctype[i] = -1; w->rho_vec[i] = RHO_MIN; }
    else if (w->u[i] - w->l[i] < RHO_TOL) { w->ctype[i] = 1; w->rho_vec[i] = RHO_EQ_FACTOR * w->rho; }
    else { w->ctype[i] = 0; w->rho_vec[i] = w->rho; }
    w->rho_inv_vec[i] = 1.0 / w->rho_vec[i];
  }
}

static int factor_kkt(ora_osqp *w) {
  for (int k = 0; k < w->nk; k++) w->Kx[k] = 0;
  for (int p = 0; p < w->Pnnz; p++) w->Kx[w->PtoK[p]] += w->Px[p];
  for (int i = 0; i < w->n; i++) w->Kx[w->sigtoK[i]] += w->st.sigma;
  for (int p = 0; p < w->Annz; p++) w->Kx[w->AtoK[p]] += w->Ax[p];
  for (int i = 0; i < w->m; i++) w->Kx[w->rhotoK[i]] += -w->rho_inv_vec[i];
  w->factor_count++;
  return ora_ldl_numeric(&w->F, w->Kp, w->Ki, w->Kx);
}

int ora_osqp_update(ora_osqp *w, const double *Px, const double *q, const double *Ax,
                    const double *l, const double *u) {
  /* keep the previous unscaled iterate so a warm start survives the rescaling */
  int n = w->n, m = w->m;
  double *xu = w->tmpn2, *yu = w->tmpm2;
  if (w->have_start) {
    for (int i = 0; i < n; i++) xu[i] = w->x[i] * w->D[i];
    for (int i = 0; i < m; i++) yu[i] = w->y[i] * w->E[i] * w->cinv;
  }
  memcpy(w->Px0, Px, sizeof(double) * w->Pnnz);
  memcpy(w->Ax0, Ax, sizeof(double) * w->Annz);
  memcpy(w->q0, q, sizeof(double) * n);
  for (int i = 0; i < m; i++) {
    w->l0[i] = l[i] < -OSQP_INFTY ? -OSQP_INFTY : l[i];
    w->u0[i] = u[i] > OSQP_INFTY ? OSQP_INFTY : u[i];
  }
  memcpy(w->Px, Px, sizeof(double) * w->Pnnz);
  memcpy(w->Ax, Ax, sizeof(double) * w->Annz);
  memcpy(w->q, q, sizeof(double) * n);
  if (w->st.scaling > 0) scale_data(w);
  else {
    for (int i = 0; i < n; i++) w->D[i] = w->Dinv[i] = 1;
    for (int i = 0; i < m; i++) { w->E[i] = w->Einv[i] = 1; w->l[i] = w->l0[i]; w->u[i] = w->u0[i]; }
    w->c = w->cinv = 1;
  }
  if (w->have_start) {
    for (int i = 0; i < n; i++) w->x[i] = xu[i] * w->Dinv[i];
    for (int i = 0; i < m; i++) w->y[i] = yu[i] * w->Einv[i] * w->c;
    A_mul(w, w->Ax, w->x, w->z);
  }
  w->rho = w->st.rho; /* OSQP keeps the adapted rho between solves; reset keeps ticks independent */
  set_rho_vec(w);
  return factor_kkt(w);
}

void ora_osqp_cold_start(ora_osqp *w) {
  memset(w->x, 0, sizeof(double) * w->n);
  memset(w->y, 0, sizeof(double) * w->m);
  memset(w->z, 0, sizeof(double) * w->m);
  w->have_start = 0;
}

typedef struct {
  double prim, dual, nAx, nz, nPx, nAty, nq;
} resinfo;

/* unscaled residuals and normalisation terms (paper section 3.4, scaled_termination = 0) */
static void residuals(ora_osqp *w, resinfo *r) {
  int n = w->n, m = w->m;
  double *Axv = w->tmpm, *Pxv = w->tmpn, *Aty = w->tmpn2;
  A_mul(w, w->Ax, w->x, Axv);
  r->prim = 0; r->nAx = 0; r->nz = 0;
  for (int i = 0; i < m; i++) {
    double a = fabs(w->Einv[i] * (Axv[i] - w->z[i]));
    if (a > r->prim) r->prim = a;
    a = fabs(w->Einv[i] * Axv[i]); if (a > r->nAx) r->nAx = a;
    a = fabs(w->Einv[i] * w->z[i]); if (a > r->nz) r->nz = a;
  }
  P_mul(w, w->Px, w->x, Pxv);
  At_mul(w, w->Ax, w->y, Aty);
  r->dual = 0; r->nPx = 0; r->nAty = 0; r->nq = 0;
  for (int i = 0; i < n; i++) {
    double a = fabs(w->Dinv[i] * (Pxv[i] + w->q[i] + Aty[i]));
    if (a > r->dual) r->dual = a;
    a = fabs(w->Dinv[i] * Pxv[i]); if (a > r->nPx) r->nPx = a;
    a = fabs(w->Dinv[i] * Aty[i]); if (a > r->nAty) r->nAty = a;
    a = fabs(w->Dinv[i] * w->q[i]); if (a > r->nq) r->nq = a;
  }
  r->dual *= w->cinv; r->nPx *= w->cinv; r->nAty *= w->cinv; r->nq *= w->cinv;
}

static double max3(double a, double b, double c) { double m = a > b ? a : b; return m > c ? m : c; }

static int is_primal_infeasible(ora_osqp *w) {
  int m = w->m, n = w->n;
  double nrm = 0;
  for (int i = 0; i < m; i++) {
    double dyi = w->dy[i];
    if (w->u[i] >= OSQP_INFTY && dyi > 0) dyi = 0; /* project on polar of recession cone */
    if (w->l[i] <= -OSQP_INFTY && dyi < 0) dyi = 0;
    w->tmpm[i] = dyi;
    double a = fabs(w->E[i] * dyi);
    if (a > nrm) nrm = a;
  }
  if (nrm <= w->st.eps_prim_inf) return 0;
  double ineq = 0;
  for (int i = 0; i < m; i++) {
    double dyi = w->tmpm[i];
    if (dyi > 0) ineq += w->u[i] * dyi; else ineq += w->l[i] * dyi;
  }
  if (ineq > -w->st.eps_prim_inf * nrm) return 0;
  At_mul(w, w->Ax, w->tmpm, w->tmpn);
  double an = 0;
  for (int i = 0; i < n; i++) { double a = fabs(w->Dinv[i] * w->tmpn[i]); if (a > an) an = a; }
  return an <= w->st.eps_prim_inf * nrm;
}

static int is_dual_infeasible(ora_osqp *w) {
  int m = w->m, n = w->n;
  double nrm = 0;
  for (int i = 0; i < n; i++) { double a = fabs(w->D[i] * w->dx[i]); if (a > nrm) nrm = a; }
  if (nrm <= w->st.eps_dual_inf) return 0;
  double qdx = 0;
  for (int i = 0; i < n; i++) qdx += w->q[i] * w->dx[i];
  if (qdx * w->cinv > -w->st.eps_dual_inf * nrm) return 0;
  P_mul(w, w->Px, w->dx, w->tmpn);
  double pn = 0;
  for (int i = 0; i < n; i++) { double a = fabs(w->Dinv[i] * w->tmpn[i]); if (a > pn) pn = a; }
  if (pn * w->cinv > w->st.eps_dual_inf * nrm) return 0;
  A_mul(w, w->Ax, w->dx, w->tmpm);
  for (int i = 0; i < m; i++) {
    double a = w->Einv[i] * w->tmpm[i];
    if ((w->u[i] < OSQP_INFTY && a > w->st.eps_dual_inf * nrm) ||
        (w->l[i] > -OSQP_INFTY && a < -w->st.eps_dual_inf * nrm)) return 0;
  }
  return 1;
}

static int check_term(ora_osqp *w, const resinfo *r, double factor) {
  double ea = w->st.eps_abs * factor, er = w->st.eps_rel * factor;
  double eps_prim = ea + er * (r->nAx > r->nz ? r->nAx : r->nz);
  double eps_dual = ea + er * max3(r->nPx, r->nAty, r->nq);
  return r->prim <= eps_prim && r->dual <= eps_dual;
}

/* ---------------------------------------------------------------- polish (paper section 4) */
static int dense_ldl_solve(int N, double *K, double *b) {
  /* LDL^T without pivoting (quasi-definite K), in place: lower part holds L, diagonal D */
  for (int j = 0; j < N; j++) {
    double d = K[j * N + j];
    for (int k = 0; k < j; k++) d -= K[j * N + k] * K[j * N + k] * K[k * N + k];
    if (d == 0.0 || d != d) return -1;
    K[j * N + j] = d;
    for (int i = j + 1; i < N; i++) {
      double t = K[i * N + j];
      for (int k = 0; k < j; k++) t -= K[i * N + k] * K[j * N + k] * K[k * N + k];
      K[i * N + j] = t / d;
    }
  }
  (void)b;
  return 0;
}
static void dense_ldl_apply(int N, const double *K, double *b) {
  for (int i = 0; i < N; i++) { double t = b[i]; for (int k = 0; k < i; k++) t -= K[i * N + k] * b[k]; b[i] = t; }
  for (int i = 0; i < N; i++) b[i] /= K[i * N + i];
  for (int i = N - 1; i >= 0; i--) { double t = b[i]; for (int k = i + 1; k < N; k++) t -= K[k * N + i] * b[k]; b[i] = t; }
}

/* Solve the equality-constrained QP on the active set `act` (+1 upper, -1 lower, 2 equality,
 * 0 inactive) with the regularised KKT + iterative refinement; unscaled data. */
static int polish_solve(ora_osqp *w, const int *act, int refine, double delta, double *x,
                        double *y) {
  int n = w->n, m = w->m, ma = 0;
  int *rows = ialloc(m);
  for (int i = 0; i < m; i++) if (act[i]) rows[ma++] = i;
  int N = n + ma;
  double *K = dalloc(N * N), *Kreg = dalloc(N * N), *rhs = dalloc(N), *sol = dalloc(N), *res = dalloc(N);
  int *rowpos = ialloc(m);
  for (int i = 0; i < m; i++) rowpos[i] = -1;
  for (int a = 0; a < ma; a++) rowpos[rows[a]] = a;
  for (int j = 0; j < n; j++) {
    for (int p = w->Pp[j]; p < w->Pp[j + 1]; p++) {
      int i = w->Pi[p];
      K[i * N + j] = w->Px0[p];
      K[j * N + i] = w->Px0[p];
    }
    for (int p = w->Ap[j]; p < w->Ap[j + 1]; p++) {
      int a = rowpos[w->Ai[p]];
      if (a < 0) continue;
      K[(n + a) * N + j] = w->Ax0[p];
      K[j * N + (n + a)] = w->Ax0[p];
    }
  }
  memcpy(Kreg, K, sizeof(double) * N * N);
  for (int i = 0; i < n; i++) Kreg[i * N + i] += delta;
  for (int a = 0; a < ma; a++) Kreg[(n + a) * N + (n + a)] -= delta;
  for (int i = 0; i < n; i++) rhs[i] = -w->q0[i];
  for (int a = 0; a < ma; a++) rhs[n + a] = act[rows[a]] == 1 ? w->u0[rows[a]] : w->l0[rows[a]];
  int rc = dense_ldl_solve(N, Kreg, NULL);
  if (rc == 0) {
    memcpy(sol, rhs, sizeof(double) * N);
    dense_ldl_apply(N, Kreg, sol);
    for (int it = 0; it < refine; it++) {
      double rn = 0;
      for (int i = 0; i < N; i++) {
        double t = rhs[i];
        for (int k = 0; k < N; k++) t -= K[i * N + k] * sol[k];
        res[i] = t;
        if (fabs(t) > rn) rn = fabs(t);
      }
      if (rn < 1e-300) break;
      dense_ldl_apply(N, Kreg, res);
      double dn = 0;
      for (int i = 0; i < N; i++) { sol[i] += res[i]; if (fabs(res[i]) > dn) dn = fabs(res[i]); }
      if (dn <= 1e-16 * (1.0 + vmaxabs(sol, N))) break;
    }
    memcpy(x, sol, sizeof(double) * n);
    for (int i = 0; i < m; i++) y[i] = 0;
    for (int a = 0; a < ma; a++) y[rows[a]] = sol[n + a];
  }
  free(rows); free(K); free(Kreg); free(rhs); free(sol); free(res); free(rowpos);
  return rc;
}

/* KKT certificate of (x,y) on the unscaled problem */
static void kkt_certificate(ora_osqp *w, const double *x, const double *y, ora_tick_info *info) {
  int n = w->n, m = w->m;
  double *Axv = dalloc(m), *g = dalloc(n), *Aty = dalloc(n);
  A_mul(w, w->Ax0, x, Axv);
  P_mul(w, w->Px0, x, g);
  At_mul(w, w->Ax0, y, Aty);
  double st = 0, pr = 0, cp = 0, obj = 0;
  for (int i = 0; i < n; i++) {
    obj += 0.5 * x[i] * g[i] + w->q0[i] * x[i];
    double a = fabs(g[i] + w->q0[i] + Aty[i]);
    if (a > st) st = a;
  }
  for (int i = 0; i < m; i++) {
    double lo = w->l0[i] - Axv[i], hi = Axv[i] - w->u0[i];
    if (lo > pr) pr = lo;
    if (hi > pr) pr = hi;
    double c = 0;
    if (y[i] > 0) c = w->u0[i] >= OSQP_INFTY ? y[i] : y[i] * fabs(w->u0[i] - Axv[i]);
    else if (y[i] < 0) c = w->l0[i] <= -OSQP_INFTY ? -y[i] : -y[i] * fabs(Axv[i] - w->l0[i]);
    if (c > cp) cp = c;
  }
  info->kkt_stat = st; info->kkt_prim = pr; info->kkt_comp = cp; info->obj = obj;
  free(Axv); free(g); free(Aty);
}

static void kkt_certificate(ora_osqp *w, const double *x, const double *y, ora_tick_info *info);
/* exact mode: active-set clean-up around the polish solve until the KKT conditions hold */
static int exact_polish(ora_osqp *w, double *x, double *y, ora_tick_info *info) {
  int n = w->n, m = w->m;
  int *act = ialloc(m);
  double *Axv = dalloc(m), *xs = dalloc(n), *ys = dalloc(m);
  /* initial guess from the ADMM iterate (unscaled z = A x side, y) */
  A_mul(w, w->Ax0, x, Axv);
  double scale = 1.0 + vmaxabs(Axv, m);
  for (int i = 0; i < m; i++) {
    if (w->u0[i] - w->l0[i] < 1e-12 * scale && w->l0[i] > -OSQP_INFTY) act[i] = 2;
    else if (w->l0[i] > -OSQP_INFTY && Axv[i] - w->l0[i] < -y[i]) act[i] = -1;
    else if (w->u0[i] < OSQP_INFTY && w->u0[i] - Axv[i] < y[i]) act[i] = 1;
    else act[i] = 0;
  }
  int ok = 0;
  for (int round = 0; round < 400 && !ok; round++) {
    if (polish_solve(w, act, 30, 1e-7, xs, ys) != 0) break;
    A_mul(w, w->Ax0, xs, Axv);
    double tolp = 1e-9 * (1.0 + vmaxabs(Axv, m)), told = 1e-9 * (1.0 + vmaxabs(ys, m));
    /* an inconsistent active set leaves a residual on the active rows (the regularised solve is
     * then a least-squares compromise): release the active INEQUALITY row with the largest one */
    int worst = -1, wkind = 0;
    double wv = 0;
    for (int i = 0; i < m; i++) {
      if (act[i] == 1 || act[i] == -1) {
        double res = fabs(Axv[i] - (act[i] == 1 ? w->u0[i] : w->l0[i]));
        if (res > tolp && res / tolp > wv) { wv = res / tolp; worst = i; wkind = 0; }
      }
    }
    if (worst >= 0) { act[worst] = wkind; continue; }
    /* most violated primal constraint / most wrong-signed multiplier */
    for (int i = 0; i < m; i++) {
      if (act[i] == 0) {
        double lo = w->l0[i] - Axv[i], hi = Axv[i] - w->u0[i];
        if (lo > tolp && lo / tolp > wv) { wv = lo / tolp; worst = i; wkind = -1; }
        if (hi > tolp && hi / tolp > wv) { wv = hi / tolp; worst = i; wkind = 1; }
      } else if (act[i] == 1) {
        if (ys[i] < -told && -ys[i] / told > wv) { wv = -ys[i] / told; worst = i; wkind = 0; }
      } else if (act[i] == -1) {
        if (ys[i] > told && ys[i] / told > wv) { wv = ys[i] / told; worst = i; wkind = 0; }
      }
    }
    if (worst < 0) {
      /* accept only a verified KKT point: equality rows and stationarity must hold too */
      ora_tick_info chk;
      kkt_certificate(w, xs, ys, &chk);
      double sx = 1.0 + vmaxabs(ys, m);
      ok = chk.kkt_prim <= 1e-7 * (1.0 + vmaxabs(Axv, m)) && chk.kkt_stat <= 1e-7 * sx;
      if (!ok) break;
    } else act[worst] = wkind;
  }
  if (ok) {
    memcpy(x, xs, sizeof(double) * n);
    memcpy(y, ys, sizeof(double) * m);
  }
  info->polish_ok = ok;
  free(act); free(Axv); free(xs); free(ys);
  return ok;
}

int ora_osqp_solve(ora_osqp *w, double *xo, double *yo, ora_tick_info *info) {
  int n = w->n, m = w->m;
  const ora_osqp_settings *st = &w->st;
  if (!st->warm_start || !w->have_start) ora_osqp_cold_start(w);
  int status = -1, iter = 0;
  resinfo r;
  memset(&r, 0, sizeof(r));
  memset(info, 0, sizeof(*info));
  double alpha = st->alpha;
  for (iter = 1; iter <= st->max_iter; iter++) {
    memcpy(w->xprev, w->x, sizeof(double) * n);
    memcpy(w->zprev, w->z, sizeof(double) * m);
    /* KKT solve: [x~; nu] */
    for (int i = 0; i < n; i++) w->rhs[w->pinv[i]] = st->sigma * w->xprev[i] - w->q[i];
    for (int i = 0; i < m; i++) w->rhs[w->pinv[n + i]] = w->zprev[i] - w->rho_inv_vec[i] * w->y[i];
    ora_ldl_solve(&w->F, w->rhs);
    for (int i = 0; i < n; i++) w->xt[i] = w->rhs[w->pinv[i]];
    for (int i = 0; i < m; i++)
      w->zt[i] = w->zprev[i] + w->rho_inv_vec[i] * (w->rhs[w->pinv[n + i]] - w->y[i]);
    for (int i = 0; i < n; i++) {
      w->x[i] = alpha * w->xt[i] + (1 - alpha) * w->xprev[i];
      w->dx[i] = w->x[i] - w->xprev[i];
    }
    for (int i = 0; i < m; i++) {
      double zr = alpha * w->zt[i] + (1 - alpha) * w->zprev[i];
      double zn = zr + w->rho_inv_vec[i] * w->y[i];
      if (zn < w->l[i]) zn = w->l[i];
      if (zn > w->u[i]) zn = w->u[i];
      w->z[i] = zn;
      w->dy[i] = w->rho_vec[i] * (zr - zn);
      w->y[i] += w->dy[i];
    }
    int checked = 0;
    if (st->check_termination && iter % st->check_termination == 0) {
      residuals(w, &r);
      checked = 1;
      if (check_term(w, &r, 1.0)) { status = QPC_STATUS_OPTIMAL; break; }
      if (is_primal_infeasible(w)) { status = QPC_STATUS_INFEASIBLE; break; }
      if (is_dual_infeasible(w)) { status = QPC_STATUS_DUAL_INFEASIBLE; break; }
    }
    if (st->adaptive_rho && st->adaptive_rho_interval && iter % st->adaptive_rho_interval == 0) {
      if (!checked) residuals(w, &r);
      double pn = r.prim / ((r.nAx > r.nz ? r.nAx : r.nz) + 1e-10);
      double dn = r.dual / (max3(r.nPx, r.nAty, r.nq) + 1e-10);
      double rho_new = w->rho * sqrt(pn / (dn + 1e-10));
      if (rho_new < RHO_MIN) rho_new = RHO_MIN;
      if (rho_new > RHO_MAX) rho_new = RHO_MAX;
      if (rho_new > w->rho * st->adaptive_rho_tolerance || rho_new < w->rho / st->adaptive_rho_tolerance) {
        w->rho = rho_new;
        set_rho_vec(w);
        if (factor_kkt(w) != 0) { status = QPC_STATUS_NUMERICAL_ERROR; break; }
      }
    }
  }
  if (status < 0) {
    iter = st->max_iter;
    residuals(w, &r);
    if (check_term(w, &r, 1.0)) status = QPC_STATUS_OPTIMAL;
    else if (check_term(w, &r, 10.0)) status = QPC_STATUS_ALMOST_OPTIMAL;
    else status = QPC_STATUS_ITERATION_LIMIT;
  }
  w->have_start = 1;
  for (int i = 0; i < n; i++) xo[i] = w->D[i] * w->x[i];
  for (int i = 0; i < m; i++) yo[i] = w->cinv * w->E[i] * w->y[i];
  info->status = status; info->iters = iter; info->prim_res = r.prim; info->dual_res = r.dual;
  if ((status == QPC_STATUS_OPTIMAL || status == QPC_STATUS_ALMOST_OPTIMAL ||
       status == QPC_STATUS_ITERATION_LIMIT) && st->polish) {
    if (st->polish == 1) {
      /* OSQP polish: one shot, delta 1e-6, 3 refinement steps, keep only if it improves */
      int *act = ialloc(m);
      double *Axv = dalloc(m), *xs = dalloc(n), *ys = dalloc(m);
      A_mul(w, w->Ax0, xo, Axv);
      for (int i = 0; i < m; i++) {
        if (Axv[i] - w->l0[i] < -yo[i]) act[i] = -1;
        else if (w->u0[i] - Axv[i] < yo[i]) act[i] = 1;
        else act[i] = 0;
      }
      if (polish_solve(w, act, 3, 1e-6, xs, ys) == 0) {
        ora_tick_info a, b;
        kkt_certificate(w, xo, yo, &a);
        kkt_certificate(w, xs, ys, &b);
        if (b.kkt_stat <= a.kkt_stat + 1e-300 && b.kkt_prim <= a.kkt_prim + 1e-300 + r.prim) {
          memcpy(xo, xs, sizeof(double) * n);
          memcpy(yo, ys, sizeof(double) * m);
          info->polish_ok = 1;
        }
      }
      free(act); free(Axv); free(xs); free(ys);
    } else {
      if (exact_polish(w, xo, yo, info)) info->status = QPC_STATUS_OPTIMAL;
    }
  }
  kkt_certificate(w, xo, yo, info);
  return info->status;
}

/* generic dense-input front end */
int ora_qp_solve(int n, int m, const double *P, const double *q, const double *A, const double *l,
                 const double *u, const ora_osqp_settings *st, double *x, double *y,
                 ora_tick_info *info) {
  int *Pp = ialloc(n + 1), *Ap = ialloc(n + 1);
  int pn = 0, an = 0;
  for (int j = 0; j < n; j++) {
    for (int i = 0; i <= j; i++) if (P[i * n + j] != 0.0 || P[j * n + i] != 0.0 || i == j) pn++;
    Pp[j + 1] = pn;
    for (int i = 0; i < m; i++) if (A[i * n + j] != 0.0) an++;
    Ap[j + 1] = an;
  }
  int *Pi = ialloc(pn), *Ai = ialloc(an);
  double *Px = dalloc(pn), *Ax = dalloc(an);
  pn = an = 0;
  for (int j = 0; j < n; j++) {
    for (int i = 0; i <= j; i++)
      if (P[i * n + j] != 0.0 || P[j * n + i] != 0.0 || i == j) {
        Pi[pn] = i; Px[pn] = 0.5 * (P[i * n + j] + P[j * n + i]); pn++;
      }
    for (int i = 0; i < m; i++) if (A[i * n + j] != 0.0) { Ai[an] = i; Ax[an] = A[i * n + j]; an++; }
  }
  ora_osqp *w = ora_osqp_setup(n, m, Pp, Pi, Ap, Ai, st);
  int rc = QPC_STATUS_NUMERICAL_ERROR;
  if (w) {
    if (ora_osqp_update(w, Px, q, Ax, l, u) == 0) rc = ora_osqp_solve(w, x, y, info);
    else { memset(info, 0, sizeof(*info)); info->status = rc; }
    ora_osqp_free(w);
  }
  free(Pp); free(Ap); free(Pi); free(Ai); free(Px); free(Ax);
  return rc;
}

/* Timing of the OSQP restatement at the solver-level boundary (bench.py's cpu_baseline leg for the L0 plug: the same committed
 * problems the device entry qpc_qp_setup / qpc_qp_solve is timed on): k dense problems that share a sparsity pattern (the union of
 * their nonzeros) -- osqp_setup once, then for every problem `reps` x (osqp_update of all coefficients + osqp_solve), as
 * `Parametron.solve!` does every tick (momentum.jl:58).  warm = 0: every solve from a cold start; warm = 1: from the previous
 * solution of the SAME problem (its first solve, cold, is not counted); warm == 2: the k problems are a SEQUENCE -- problem b > 0 is
 * solved once from the solution of problem b - 1 (values updated in place, as Parametron does every tick: momentum.jl:58).  us[k]: mean microseconds per solve; iters[k], status[k]: of
 * the last one.  Returns 0, or -1 when the set-up fails. */
#include <time.h>
int ora_qp_time(int n, int m, int k, const double *P, const double *q, const double *A, const double *l, const double *u,
                const ora_osqp_settings *st, int reps, int warm, double *us, int *iters, int *status) {
  int *Pp = ialloc(n + 1), *Ap = ialloc(n + 1);
  int pn = 0, an = 0;
  const long long nn = (long long)n * n, mn = (long long)m * n;
#define ANY_P(i, j) ({ int f__ = (i) == (j); for (int b__ = 0; b__ < k && !f__; b__++) f__ = P[b__ * nn + (i) * n + (j)] != 0.0 || P[b__ * nn + (j) * n + (i)] != 0.0; f__; })
#define ANY_A(i, j) ({ int f__ = 0; for (int b__ = 0; b__ < k && !f__; b__++) f__ = A[b__ * mn + (long long)(i) * n + (j)] != 0.0; f__; })
  for (int j = 0; j < n; j++) {
    for (int i = 0; i <= j; i++) if (ANY_P(i, j)) pn++;
    Pp[j + 1] = pn;
    for (int i = 0; i < m; i++) if (ANY_A(i, j)) an++;
    Ap[j + 1] = an;
  }
  int *Pi = ialloc(pn), *Ai = ialloc(an);
  double *Px = dalloc(pn), *Ax = dalloc(an), *x = dalloc(n), *y = dalloc(m > 0 ? m : 1);
  pn = an = 0;
  for (int j = 0; j < n; j++) {
    for (int i = 0; i <= j; i++) if (ANY_P(i, j)) Pi[pn++] = i;
    for (int i = 0; i < m; i++) if (ANY_A(i, j)) Ai[an++] = i;
  }
  ora_osqp_settings s2 = *st;
  s2.warm_start = warm ? 1 : 0;
  ora_osqp *w = ora_osqp_setup(n, m, Pp, Pi, Ap, Ai, &s2);
  int rc = w ? 0 : -1;
  for (int b = 0; b < k && w; b++) {
    for (int e = 0; e < Pp[n]; e++) { int j = 0; while (Pp[j + 1] <= e) j++; Px[e] = 0.5 * (P[b * nn + Pi[e] * n + j] + P[b * nn + j * n + Pi[e]]); }
    for (int e = 0; e < Ap[n]; e++) { int j = 0; while (Ap[j + 1] <= e) j++; Ax[e] = A[b * mn + (long long)Ai[e] * n + j]; }
    ora_tick_info info;
    struct timespec t0, t1;
    if (warm == 2 && b > 0) { // a SEQUENCE (consecutive ticks of one closed loop): this problem from the previous problem's solution, once
      clock_gettime(CLOCK_MONOTONIC, &t0);
      ora_osqp_update(w, Px, q + b * n, Ax, l + b * m, u + b * m);
      status[b] = ora_osqp_solve(w, x, y, &info);
      clock_gettime(CLOCK_MONOTONIC, &t1);
      us[b] = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
      iters[b] = info.iters;
      continue;
    }
    ora_osqp_cold_start(w);
    if (warm) { ora_osqp_update(w, Px, q + b * n, Ax, l + b * m, u + b * m); ora_osqp_solve(w, x, y, &info); }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; r++) {
      if (!warm) ora_osqp_cold_start(w);
      ora_osqp_update(w, Px, q + b * n, Ax, l + b * m, u + b * m);
      status[b] = ora_osqp_solve(w, x, y, &info);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    us[b] = ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3) / (reps > 0 ? reps : 1);
    iters[b] = info.iters;
  }
#undef ANY_P
#undef ANY_A
  if (w) ora_osqp_free(w);
  free(Pp); free(Ap); free(Pi); free(Ai); free(Px); free(Ax); free(x); free(y);
  return rc;
}
