/* ora_osqp.h -- oracle-internal OSQP restatement (see ora_osqp.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORA_OSQP_H
#define ORA_OSQP_H
#include "oracle.h"
#include "ora_ldl.h"

typedef struct ora_osqp {
  int n, m;
  ora_osqp_settings st;
  /* structure (CSC): P upper triangular, A general */
  int *Pp, *Pi, *Ap, *Ai;
  int Pnnz, Annz;
  /* original and scaled data */
  double *Px0, *Ax0, *q0, *l0, *u0;
  double *Px, *Ax, *q, *l, *u;
  double *D, *E, *Dinv, *Einv, c, cinv;
  /* rho */
  double rho;
  double *rho_vec, *rho_inv_vec;
  int *ctype; /* -1 unbounded, 0 inequality, 1 equality */
  /* KKT (permuted, upper CSC) */
  int nk, *Kp, *Ki, *PtoK, *AtoK, *sigtoK, *rhotoK, *perm, *pinv;
  double *Kx;
  ora_ldl F;
  /* iterates (scaled) */
  double *x, *y, *z, *xt, *zt, *xprev, *zprev, *dx, *dy, *rhs, *tmpn, *tmpm, *tmpn2, *tmpm2;
  int have_start;
  int factor_count;
} ora_osqp;

ora_osqp *ora_osqp_setup(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                         const ora_osqp_settings *st);
void ora_osqp_free(ora_osqp *w);
/* new numerical values (same pattern): unscale / update / rescale / refactor */
int ora_osqp_update(ora_osqp *w, const double *Px, const double *q, const double *Ax,
                    const double *l, const double *u);
void ora_osqp_cold_start(ora_osqp *w);
/* returns QPC_STATUS_* ; x[n], y[m] unscaled */
int ora_osqp_solve(ora_osqp *w, double *x, double *y, ora_tick_info *info);
#endif
