/* ora_ldl.h -- oracle-internal sparse LDL^T (see ora_ldl.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORA_LDL_H
#define ORA_LDL_H
#include <stdint.h>

typedef struct ora_ldl {
  int n;
  int *etree, *Lnz, *Lp, *Li;
  double *Lx, *D, *Dinv;
  int *iw;
  double *fw;
} ora_ldl;

void ora_min_degree(int n, const int *Ap, const int *Ai, int *perm);
int ora_ldl_symbolic(ora_ldl *F, int n, const int *Ap, const int *Ai);
int ora_ldl_numeric(ora_ldl *F, const int *Ap, const int *Ai, const double *Ax);
void ora_ldl_solve(const ora_ldl *F, double *x);
void ora_ldl_free(ora_ldl *F);
#endif
