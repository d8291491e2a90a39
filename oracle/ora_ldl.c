/*
 * ora_ldl.c -- oracle: sparse LDL^T of a quasi-definite KKT matrix (upper-triangular CSC input),
 * elimination tree + up-looking numeric factorisation, as in QDLDL (the linear-system solver
 * OSQP 0.5.0 ships; published in Stellato et al., "OSQP: an operator splitting solver for
 * quadratic programs", Math. Prog. Comp. 2020, section 3.1) plus a greedy minimum-degree
 * ordering standing in for AMD.  [recalled: OSQP/QDLDL sources are not in this image.]
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 */
#include "ora_ldl.h"
#include <stdlib.h>
#include <string.h>

/* greedy minimum degree on the symmetric pattern given as upper CSC; perm[new] = old */
void ora_min_degree(int n, const int *Ap, const int *Ai, int *perm) {
  int W = (n + 63) / 64;
  uint64_t *adj = (uint64_t *)calloc((size_t)n * W, sizeof(uint64_t));
  char *done = (char *)calloc(n, 1);
  for (int j = 0; j < n; j++)
    for (int p = Ap[j]; p < Ap[j + 1]; p++) {
      int i = Ai[p];
      if (i == j) continue;
      adj[(size_t)i * W + j / 64] |= 1ull << (j % 64);
      adj[(size_t)j * W + i / 64] |= 1ull << (i % 64);
    }
  for (int step = 0; step < n; step++) {
    int best = -1, bestdeg = n + 1;
    for (int i = 0; i < n; i++) {
      if (done[i]) continue;
      int deg = 0;
      for (int w = 0; w < W; w++) deg += __builtin_popcountll(adj[(size_t)i * W + w]);
      if (deg < bestdeg) { bestdeg = deg; best = i; }
    }
    perm[step] = best;
    done[best] = 1;
    uint64_t *nb = adj + (size_t)best * W;
    /* connect neighbours pairwise, remove `best` from their lists */
    for (int i = 0; i < n; i++) {
      if (done[i] || !(nb[i / 64] >> (i % 64) & 1)) continue;
      uint64_t *ai = adj + (size_t)i * W;
      for (int w = 0; w < W; w++) ai[w] |= nb[w];
      ai[i / 64] &= ~(1ull << (i % 64));
      ai[best / 64] &= ~(1ull << (best % 64));
    }
    /* eliminated node no longer appears anywhere */
    for (int i = 0; i < n; i++)
      if (!done[i]) adj[(size_t)i * W + best / 64] &= ~(1ull << (best % 64));
  }
  free(adj);
  free(done);
}

int ora_ldl_symbolic(ora_ldl *F, int n, const int *Ap, const int *Ai) {
  memset(F, 0, sizeof(*F));
  F->n = n;
  F->etree = (int *)malloc(sizeof(int) * n);
  F->Lnz = (int *)calloc(n, sizeof(int));
  F->Lp = (int *)malloc(sizeof(int) * (n + 1));
  int *work = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) F->etree[i] = -1;
  for (int j = 0; j < n; j++) {
    work[j] = j;
    for (int p = Ap[j]; p < Ap[j + 1]; p++) {
      int i = Ai[p];
      if (i > j) { free(work); return -1; } /* must be upper triangular */
      while (work[i] != j) {
        if (F->etree[i] == -1) F->etree[i] = j;
        F->Lnz[i]++;
        work[i] = j;
        i = F->etree[i];
      }
    }
  }
  F->Lp[0] = 0;
  for (int i = 0; i < n; i++) F->Lp[i + 1] = F->Lp[i] + F->Lnz[i];
  int nnz = F->Lp[n];
  F->Li = (int *)malloc(sizeof(int) * (nnz > 0 ? nnz : 1));
  F->Lx = (double *)malloc(sizeof(double) * (nnz > 0 ? nnz : 1));
  F->D = (double *)malloc(sizeof(double) * n);
  F->Dinv = (double *)malloc(sizeof(double) * n);
  F->iw = (int *)malloc(sizeof(int) * 4 * n);
  F->fw = (double *)malloc(sizeof(double) * n);
  free(work);
  return 0;
}

int ora_ldl_numeric(ora_ldl *F, const int *Ap, const int *Ai, const double *Ax) {
  int n = F->n;
  int *yIdx = F->iw, *elim = F->iw + n, *next = F->iw + 2 * n, *mark = F->iw + 3 * n;
  double *y = F->fw;
  for (int i = 0; i < n; i++) { next[i] = F->Lp[i]; mark[i] = 0; y[i] = 0.0; F->D[i] = 0.0; }
  for (int k = 0; k < n; k++) {
    int nnzY = 0;
    for (int p = Ap[k]; p < Ap[k + 1]; p++) {
      int b = Ai[p];
      if (b == k) { F->D[k] = Ax[p]; continue; }
      y[b] = Ax[p];
      int nx = b;
      if (!mark[nx]) {
        mark[nx] = 1;
        elim[0] = nx;
        int nnzE = 1;
        nx = F->etree[b];
        while (nx != -1 && nx < k) {
          if (mark[nx]) break;
          mark[nx] = 1;
          elim[nnzE++] = nx;
          nx = F->etree[nx];
        }
        while (nnzE) yIdx[nnzY++] = elim[--nnzE];
      }
    }
    for (int i = nnzY - 1; i >= 0; i--) {
      int c = yIdx[i];
      int tmp = next[c];
      double yc = y[c];
      for (int j = F->Lp[c]; j < tmp; j++) y[F->Li[j]] -= F->Lx[j] * yc;
      F->Li[tmp] = k;
      F->Lx[tmp] = yc * F->Dinv[c];
      F->D[k] -= yc * F->Lx[tmp];
      next[c]++;
      y[c] = 0.0;
      mark[c] = 0;
    }
    if (F->D[k] == 0.0) return -1;
    F->Dinv[k] = 1.0 / F->D[k];
  }
  return 0;
}

void ora_ldl_solve(const ora_ldl *F, double *x) {
  int n = F->n;
  for (int i = 0; i < n; i++) {
    double xi = x[i];
    for (int j = F->Lp[i]; j < F->Lp[i + 1]; j++) x[F->Li[j]] -= F->Lx[j] * xi;
  }
  for (int i = 0; i < n; i++) x[i] *= F->Dinv[i];
  for (int i = n - 1; i >= 0; i--) {
    double xi = x[i];
    for (int j = F->Lp[i]; j < F->Lp[i + 1]; j++) xi -= F->Lx[j] * x[F->Li[j]];
    x[i] = xi;
  }
}

void ora_ldl_free(ora_ldl *F) {
  free(F->etree); free(F->Lnz); free(F->Lp); free(F->Li); free(F->Lx);
  free(F->D); free(F->Dinv); free(F->iw); free(F->fw);
  memset(F, 0, sizeof(*F));
}
