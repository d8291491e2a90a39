/*
 * ora_rbd.c -- oracle: rigid-body kinematics/dynamics restated from RigidBodyDynamics.jl 2.2.0
 * semantics (package not present in this image => "recalled", pinned by
 * the property tests in tests/test_oracle_rbd.py).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Reference call sites that consume these quantities:
 *   transform_to_root        src/tasks.jl:33,75,114 ; src/contacts.jl:61 ; momentum.jl:174
 *   geometric_jacobian!      src/tasks.jl:34,76,115
 *   point_jacobian!          src/tasks.jl:157-160
 *   bias_acceleration        src/tasks.jl:37,79,118,163
 *   momentum_matrix!         src/tasks.jl:212 ; momentum.jl:168
 *   momentum_rate_bias       src/tasks.jl:218 ; momentum.jl:169
 *   center_of_mass           src/tasks.jl:209,215 ; momentum.jl:170
 *   inverse_dynamics!        momentum.jl:75
 *
 * Everything is expressed in WORLD coordinates about the world origin; spatial vectors are
 * [angular; linear].
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- small helpers */
static void cross3(const double a[3], const double b[3], double c[3]) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static void mv3(const double R[9], const double x[3], double y[3]) {
  double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  double b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  double c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
static void mtv3(const double R[9], const double x[3], double y[3]) {
  double a = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  double b = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  double c = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
static void mm3(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
/* motion cross product ad(a) b   */
static void crm(const double a[6], const double b[6], double c[6]) {
  double t1[3], t2[3], t3[3];
  cross3(a, b, t1);
  cross3(a, b + 3, t2);
  cross3(a + 3, b, t3);
  c[0] = t1[0]; c[1] = t1[1]; c[2] = t1[2];
  c[3] = t2[0] + t3[0]; c[4] = t2[1] + t3[1]; c[5] = t2[2] + t3[2];
}
/* force cross product (w,v) x* (k,l) = (w x k + v x l, w x l)   */
static void crf(const double a[6], const double f[6], double c[6]) {
  double t1[3], t2[3], t3[3];
  cross3(a, f, t1);
  cross3(a + 3, f + 3, t2);
  cross3(a, f + 3, t3);
  c[0] = t1[0] + t2[0]; c[1] = t1[1] + t2[1]; c[2] = t1[2] + t2[2];
  c[3] = t3[0]; c[4] = t3[1]; c[5] = t3[2];
}
/* momentum = I * twist : ang = J w + h x v ; lin = m v - h x w */
static void inertia_mul(const ora_inertia *I, const double t[6], double out[6]) {
  const double *w = t, *v = t + 3;
  double hv[3], hw[3];
  cross3(I->h, v, hv);
  cross3(I->h, w, hw);
  out[0] = I->J[0] * w[0] + I->J[3] * w[1] + I->J[4] * w[2] + hv[0];
  out[1] = I->J[3] * w[0] + I->J[1] * w[1] + I->J[5] * w[2] + hv[1];
  out[2] = I->J[4] * w[0] + I->J[5] * w[1] + I->J[2] * w[2] + hv[2];
  out[3] = I->m * v[0] - hw[0];
  out[4] = I->m * v[1] - hw[1];
  out[5] = I->m * v[2] - hw[2];
}
static void inertia_add(ora_inertia *a, const ora_inertia *b) {
  a->m += b->m;
  for (int k = 0; k < 3; k++) a->h[k] += b->h[k];
  for (int k = 0; k < 6; k++) a->J[k] += b->J[k];
}

/* ---------------------------------------------------------------- model */
ora_model *ora_model_create(const qpc_model_desc *d) {
  if (!d || d->nbodies <= 0 || d->nbodies > ORA_MAXB) return NULL;
  ora_model *m = (ora_model *)calloc(1, sizeof(ora_model));
  m->nb = d->nbodies;
  int nq = 0, nv = 0;
  for (int i = 0; i < m->nb; i++) {
    m->parent[i] = d->parent[i];
    if (m->parent[i] >= i) { free(m); return NULL; }
    m->jtype[i] = d->joint_type[i];
    memcpy(m->axis[i], d->joint_axis + 3 * i, 24);
    memcpy(m->Rf[i], d->joint_R + 9 * i, 72);
    memcpy(m->pf[i], d->joint_p + 3 * i, 24);
    m->mass[i] = d->mass[i];
    memcpy(m->com[i], d->com + 3 * i, 24);
    memcpy(m->Icom[i], d->inertia + 6 * i, 48);
    int nqj, nvj;
    switch (m->jtype[i]) {
      case QPC_JOINT_REVOLUTE: case QPC_JOINT_PRISMATIC: nqj = 1; nvj = 1; break;
      case QPC_JOINT_QUAT_FLOATING: nqj = 7; nvj = 6; break;
      case QPC_JOINT_FIXED: nqj = 0; nvj = 0; break;
      default: free(m); return NULL;
    }
    m->qoff[i] = nq; m->voff[i] = nv; m->nqj[i] = nqj; m->nvj[i] = nvj;
    for (int k = 0; k < nvj; k++) {
      if (nv + k >= ORA_MAXV) { free(m); return NULL; }
      m->vbody[nv + k] = i;
    }
    nq += nqj; nv += nvj;
    m->mtot += m->mass[i];
  }
  if (nq > ORA_MAXQ) { free(m); return NULL; }
  m->nq = nq; m->nv = nv;
  memcpy(m->g, d->gravity, 24);
  return m;
}
void ora_model_destroy(ora_model *m) { free(m); }
int ora_model_nq(const ora_model *m) { return m->nq; }
int ora_model_nv(const ora_model *m) { return m->nv; }
ora_state *ora_state_create(void) { return (ora_state *)calloc(1, sizeof(ora_state)); }
void ora_state_destroy(ora_state *s) { free(s); }

static void rodrigues(const double a[3], double th, double R[9]) {
  double c = cos(th), s = sin(th), C = 1 - c;
  R[0] = c + a[0] * a[0] * C;        R[1] = a[0] * a[1] * C - a[2] * s; R[2] = a[0] * a[2] * C + a[1] * s;
  R[3] = a[1] * a[0] * C + a[2] * s; R[4] = c + a[1] * a[1] * C;        R[5] = a[1] * a[2] * C - a[0] * s;
  R[6] = a[2] * a[0] * C - a[1] * s; R[7] = a[2] * a[1] * C + a[0] * s; R[8] = c + a[2] * a[2] * C;
}
static void quat_to_R(const double q[4], double R[9]) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* copyto!(state, x) followed by every cache the hot path reads (momentum.jl:57) */
void ora_state_update(const ora_model *m, const double *q, const double *v, ora_state *s) {
  memcpy(s->q, q, sizeof(double) * m->nq);
  memcpy(s->v, v, sizeof(double) * m->nv);
  double msum = 0, csum[3] = {0, 0, 0};
  for (int i = 0; i < m->nb; i++) {
    /* joint transform: after-joint (body) frame -> before-joint frame */
    double Rj[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pj[3] = {0, 0, 0};
    const double *qi = q + m->qoff[i];
    switch (m->jtype[i]) {
      case QPC_JOINT_REVOLUTE: rodrigues(m->axis[i], qi[0], Rj); break;
      case QPC_JOINT_PRISMATIC: for (int k = 0; k < 3; k++) pj[k] = m->axis[i][k] * qi[0]; break;
      case QPC_JOINT_QUAT_FLOATING: quat_to_R(qi, Rj); pj[0] = qi[4]; pj[1] = qi[5]; pj[2] = qi[6]; break;
      default: break;
    }
    double Rl[9], pl[3];
    mm3(m->Rf[i], Rj, Rl);
    mv3(m->Rf[i], pj, pl);
    for (int k = 0; k < 3; k++) pl[k] += m->pf[i][k];
    int par = m->parent[i];
    if (par < 0) {
      memcpy(s->R[i], Rl, 72);
      memcpy(s->p[i], pl, 24);
    } else {
      mm3(s->R[par], Rl, s->R[i]);
      mv3(s->R[par], pl, s->p[i]);
      for (int k = 0; k < 3; k++) s->p[i][k] += s->p[par][k];
    }
    const double *R = s->R[i], *p = s->p[i];
    /* motion subspace columns in world: Ad(H_i) S_local ; joint twist tj = S v_j */
    double tj[6] = {0, 0, 0, 0, 0, 0};
    int v0 = m->voff[i];
    if (m->jtype[i] == QPC_JOINT_REVOLUTE) {
      double w[3], l[3];
      mv3(R, m->axis[i], w);
      cross3(p, w, l);
      for (int k = 0; k < 3; k++) { s->S[v0][k] = w[k]; s->S[v0][3 + k] = l[k]; }
    } else if (m->jtype[i] == QPC_JOINT_PRISMATIC) {
      double l[3];
      mv3(R, m->axis[i], l);
      for (int k = 0; k < 3; k++) { s->S[v0][k] = 0; s->S[v0][3 + k] = l[k]; }
    } else if (m->jtype[i] == QPC_JOINT_QUAT_FLOATING) {
      for (int c = 0; c < 3; c++) {
        double w[3] = {R[c], R[3 + c], R[6 + c]}, l[3];
        cross3(p, w, l);
        for (int k = 0; k < 3; k++) {
          s->S[v0 + c][k] = w[k]; s->S[v0 + c][3 + k] = l[k];
          s->S[v0 + 3 + c][k] = 0; s->S[v0 + 3 + c][3 + k] = w[k];
        }
      }
    }
    for (int c = 0; c < m->nvj[i]; c++)
      for (int k = 0; k < 6; k++) tj[k] += s->S[v0 + c][k] * v[v0 + c];
    /* twist_wrt_world and gravity-free bias acceleration */
    for (int k = 0; k < 6; k++) s->T[i][k] = (par < 0 ? 0.0 : s->T[par][k]) + tj[k];
    double cr[6];
    crm(s->T[i], tj, cr);
    for (int k = 0; k < 6; k++) s->Ab[i][k] = (par < 0 ? 0.0 : s->Ab[par][k]) + cr[k];
    /* body inertia in world about world origin */
    double cw[3];
    mv3(R, m->com[i], cw);
    for (int k = 0; k < 3; k++) cw[k] += p[k];
    const double *I = m->Icom[i];
    double Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], Iw[9], Rt[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = R[3 * b + a];
    mm3(R, Ib, RI);
    mm3(RI, Rt, Iw);
    double mm = m->mass[i], cc = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
    ora_inertia *B = &s->Iw[i];
    B->m = mm;
    for (int k = 0; k < 3; k++) B->h[k] = mm * cw[k];
    B->J[0] = Iw[0] + mm * (cc - cw[0] * cw[0]);
    B->J[1] = Iw[4] + mm * (cc - cw[1] * cw[1]);
    B->J[2] = Iw[8] + mm * (cc - cw[2] * cw[2]);
    B->J[3] = Iw[1] - mm * cw[0] * cw[1];
    B->J[4] = Iw[2] - mm * cw[0] * cw[2];
    B->J[5] = Iw[5] - mm * cw[1] * cw[2];
    msum += mm;
    for (int k = 0; k < 3; k++) csum[k] += mm * cw[k];
  }
  for (int k = 0; k < 3; k++) s->com[k] = csum[k] / msum;
  /* composite rigid body inertias */
  for (int i = 0; i < m->nb; i++) s->Ic[i] = s->Iw[i];
  for (int i = m->nb - 1; i >= 0; i--)
    if (m->parent[i] >= 0) inertia_add(&s->Ic[m->parent[i]], &s->Ic[i]);
  /* momentum matrix: column k = Ic[succ(joint)] * S_k */
  for (int k = 0; k < m->nv; k++) {
    double h[6];
    inertia_mul(&s->Ic[m->vbody[k]], s->S[k], h);
    for (int r = 0; r < 6; r++) s->Amom[r][k] = h[r];
  }
  /* momentum rate bias = sum_i I_i Ab_i + T_i x* (I_i T_i) */
  for (int r = 0; r < 6; r++) s->Adv[r] = 0;
  for (int i = 0; i < m->nb; i++) {
    double h[6], ia[6], c[6];
    inertia_mul(&s->Iw[i], s->T[i], h);
    inertia_mul(&s->Iw[i], s->Ab[i], ia);
    crf(s->T[i], h, c);
    for (int r = 0; r < 6; r++) s->Adv[r] += ia[r] + c[r];
  }
}

void ora_get_transform(const ora_state *s, int body, double R[9], double p[3]) {
  if (body < 0) {
    double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(R, I, 72); p[0] = p[1] = p[2] = 0;
  } else { memcpy(R, s->R[body], 72); memcpy(p, s->p[body], 24); }
}
void ora_get_twist(const ora_state *s, int body, double T[6]) {
  for (int k = 0; k < 6; k++) T[k] = body < 0 ? 0.0 : s->T[body][k];
}
void ora_get_bias(const ora_state *s, int body, double A[6]) {
  for (int k = 0; k < 6; k++) A[k] = body < 0 ? 0.0 : s->Ab[body][k];
}
void ora_get_com(const ora_state *s, double c[3]) { memcpy(c, s->com, 24); }
void ora_get_momentum_matrix(const ora_model *m, const ora_state *s, double *A, double Adv[6]) {
  for (int r = 0; r < 6; r++)
    for (int k = 0; k < m->nv; k++) A[r * m->nv + k] = s->Amom[r][k];
  if (Adv) memcpy(Adv, s->Adv, 48);
}

/* world -> frame transform of a motion vector: w' = R^T w ; v' = R^T (v - p x w) */
static void motion_to_frame(const ora_state *s, int frame, const double in[6], double out[6]) {
  if (frame < 0) { memcpy(out, in, 48); return; }
  const double *R = s->R[frame], *p = s->p[frame];
  double pw[3], t[3];
  cross3(p, in, pw);
  for (int k = 0; k < 3; k++) t[k] = in[3 + k] - pw[k];
  mtv3(R, in, out);
  mtv3(R, t, out + 3);
}

/* geometric_jacobian!(jac, state, path, world_to_frame): columns +-S_j for the joints on the
 * path base -> target (minus when traversed towards the root), SURVEY B.5 */
void ora_geometric_jacobian(const ora_model *m, const ora_state *s, int base, int target,
                            int frame, double *J) {
  int nv = m->nv;
  memset(J, 0, sizeof(double) * 6 * nv);
  /* mark ancestors of target */
  int mark[ORA_MAXB];
  memset(mark, 0, sizeof(mark));
  for (int b = target; b >= 0; b = m->parent[b]) mark[b] = 1;
  int lca = -1;
  for (int b = base; b >= 0; b = m->parent[b])
    if (mark[b]) { lca = b; break; }
  for (int pass = 0; pass < 2; pass++) {
    int start = pass == 0 ? base : target;
    double sign = pass == 0 ? -1.0 : 1.0;
    for (int b = start; b >= 0 && b != lca; b = m->parent[b]) {
      for (int c = 0; c < m->nvj[b]; c++) {
        int k = m->voff[b] + c;
        double col[6];
        motion_to_frame(s, frame, s->S[k], col);
        for (int r = 0; r < 6; r++) J[r * nv + k] = sign * col[r];
      }
    }
  }
}

void ora_relative_twist(const ora_model *m, const ora_state *s, int body, int base, int frame,
                        double T[6]) {
  (void)m;
  double tw[6];
  for (int k = 0; k < 6; k++)
    tw[k] = (body < 0 ? 0.0 : s->T[body][k]) - (base < 0 ? 0.0 : s->T[base][k]);
  motion_to_frame(s, frame, tw, T);
}

/* transform(state, accel, frame) for accel = acceleration of `body` w.r.t. `base` given in
 * world coordinates: Ad(F<-W)[ A - T_F x (T_body - T_base) ]   */
void ora_transform_accel(const ora_model *m, const ora_state *s, const double Aw[6], int body,
                         int base, int frame, double out[6]) {
  (void)m;
  if (frame < 0) { memcpy(out, Aw, 48); return; }
  double trel[6], cr[6], a[6];
  for (int k = 0; k < 6; k++)
    trel[k] = (body < 0 ? 0.0 : s->T[body][k]) - (base < 0 ? 0.0 : s->T[base][k]);
  crm(s->T[frame], trel, cr);
  for (int k = 0; k < 6; k++) a[k] = Aw[k] - cr[k];
  motion_to_frame(s, frame, a, out);
}

void ora_spatial_accelerations(const ora_model *m, const ora_state *s, const double *vd,
                               double *acc) {
  for (int i = 0; i < m->nb; i++) {
    int par = m->parent[i];
    double a[6];
    /* A_i = A_par + S v̇_j + T_i x (S v_j) ; the last term is Ab_i - Ab_par */
    for (int k = 0; k < 6; k++)
      a[k] = (par < 0 ? 0.0 : acc[6 * par + k]) + s->Ab[i][k] - (par < 0 ? 0.0 : s->Ab[par][k]);
    for (int c = 0; c < m->nvj[i]; c++)
      for (int k = 0; k < 6; k++) a[k] += s->S[m->voff[i] + c][k] * vd[m->voff[i] + c];
    memcpy(acc + 6 * i, a, 48);
  }
}

/* inverse_dynamics!(tau, jointwrenches, accelerations, state, v̇, externalwrenches)
 * (momentum.jl:75): RNEA, root acceleration -g, external wrenches in world frame per body. */
void ora_inverse_dynamics(const ora_model *m, const ora_state *s, const double *vd,
                          const double *wext, double *tau) {
  double acc[ORA_MAXB * 6], F[ORA_MAXB][6];
  ora_spatial_accelerations(m, s, vd, acc);
  for (int i = 0; i < m->nb; i++) {
    double a[6], h[6], ia[6], c[6];
    for (int k = 0; k < 3; k++) { a[k] = acc[6 * i + k]; a[3 + k] = acc[6 * i + 3 + k] - m->g[k]; }
    inertia_mul(&s->Iw[i], a, ia);
    inertia_mul(&s->Iw[i], s->T[i], h);
    crf(s->T[i], h, c);
    for (int k = 0; k < 6; k++) F[i][k] = ia[k] + c[k] - (wext ? wext[6 * i + k] : 0.0);
  }
  for (int i = m->nb - 1; i >= 0; i--) {
    for (int c = 0; c < m->nvj[i]; c++) {
      int k = m->voff[i] + c;
      double t = 0;
      for (int r = 0; r < 6; r++) t += s->S[k][r] * F[i][r];
      tau[k] = t;
    }
    if (m->parent[i] >= 0)
      for (int r = 0; r < 6; r++) F[m->parent[i]][r] += F[i][r];
  }
}

/* mass matrix M = sum_i J_i^T I_i J_i via composite inertias: M[j][k] = S_j^T Ic[max] S_k when one
 * joint is an ancestor of the other */
void ora_mass_matrix(const ora_model *m, const ora_state *s, double *M) {
  int nv = m->nv;
  memset(M, 0, sizeof(double) * nv * nv);
  for (int k = 0; k < nv; k++) {
    int bk = m->vbody[k];
    double h[6];
    inertia_mul(&s->Ic[bk], s->S[k], h);
    /* all j on the chain from bk to root (and same-joint columns) */
    for (int b = bk; b >= 0; b = m->parent[b]) {
      for (int c = 0; c < m->nvj[b]; c++) {
        int j = m->voff[b] + c;
        double t = 0;
        for (int r = 0; r < 6; r++) t += s->S[j][r] * h[r];
        M[j * nv + k] = t;
        M[k * nv + j] = t;
      }
    }
  }
}

/* dense Cholesky solve helper */
static int chol_solve(int n, double *A, double *b) {
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (d <= 0) return -1;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double t = A[i * n + j];
      for (int k = 0; k < j; k++) t -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = t / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double t = b[i];
    for (int k = 0; k < i; k++) t -= A[i * n + k] * b[k];
    b[i] = t / A[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double t = b[i];
    for (int k = i + 1; k < n; k++) t -= A[k * n + i] * b[k];
    b[i] = t / A[i * n + i];
  }
  return 0;
}

/* dynamics!(result, state, tau) as used by test/controller.jl:92-96: M v̇ = tau - c(q,v,g,wext) */
int ora_forward_dynamics(const ora_model *m, const ora_state *s, const double *tau,
                         const double *wext, double *vd) {
  int nv = m->nv;
  double *M = (double *)malloc(sizeof(double) * nv * nv);
  double zero[ORA_MAXV] = {0}, c[ORA_MAXV];
  ora_mass_matrix(m, s, M);
  ora_inverse_dynamics(m, s, zero, wext, c);
  for (int k = 0; k < nv; k++) vd[k] = tau[k] - c[k];
  int rc = chol_solve(nv, M, vd);
  free(M);
  return rc;
}

/* configuration update on the manifold: revolute/prismatic q += v dt; floating joint:
 * quat <- quat * exp(w_body dt / 2), pos += R(quat) v_body dt  (first order in the position) */
void ora_integrate(const ora_model *m, double *q, const double *v, double dt) {
  for (int i = 0; i < m->nb; i++) {
    double *qi = q + m->qoff[i];
    const double *vi = v + m->voff[i];
    if (m->jtype[i] == QPC_JOINT_REVOLUTE || m->jtype[i] == QPC_JOINT_PRISMATIC) {
      qi[0] += vi[0] * dt;
    } else if (m->jtype[i] == QPC_JOINT_QUAT_FLOATING) {
      double R[9], dp[3];
      quat_to_R(qi, R);
      mv3(R, vi + 3, dp);
      double th = sqrt(vi[0] * vi[0] + vi[1] * vi[1] + vi[2] * vi[2]) * dt;
      double dq[4];
      if (th < 1e-12) { dq[0] = 1; dq[1] = 0.5 * vi[0] * dt; dq[2] = 0.5 * vi[1] * dt; dq[3] = 0.5 * vi[2] * dt; }
      else {
        double sc = sin(0.5 * th) / (th / dt);
        dq[0] = cos(0.5 * th); dq[1] = sc * vi[0]; dq[2] = sc * vi[1]; dq[3] = sc * vi[2];
      }
      double w = qi[0], x = qi[1], y = qi[2], z = qi[3];
      double nw = w * dq[0] - x * dq[1] - y * dq[2] - z * dq[3];
      double nx = w * dq[1] + x * dq[0] + y * dq[3] - z * dq[2];
      double ny = w * dq[2] - x * dq[3] + y * dq[0] + z * dq[1];
      double nz = w * dq[3] + x * dq[2] - y * dq[1] + z * dq[0];
      double nn = sqrt(nw * nw + nx * nx + ny * ny + nz * nz);
      qi[0] = nw / nn; qi[1] = nx / nn; qi[2] = ny / nn; qi[3] = nz / nn;
      for (int k = 0; k < 3; k++) qi[4 + k] += dp[k] * dt;
    }
  }
}
