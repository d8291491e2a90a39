/*
 * ora_ctrl.c -- oracle: the MomentumBasedController tick restated in the REFERENCE LAYOUT
 * (SURVEY.md Appendix A): variables in creation order, rows in @constraint call order, P = 2x
 * the written objective weights, solved by the OSQP restatement (ora_osqp.c).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows, function by function:
 *   MomentumBasedController ctor          src/lowlevel/momentum.jl:15-33
 *   (controller)(tau, t, x)               src/lowlevel/momentum.jl:41-81
 *   checkstatus                           src/lowlevel/momentum.jl:83-91
 *   addtask! (3 methods) + slack helper   src/lowlevel/momentum.jl:99-126
 *   regularize!                           src/lowlevel/momentum.jl:128-131
 *   addcontact!                           src/lowlevel/momentum.jl:133-148
 *   initialize! / wrench balance          src/lowlevel/momentum.jl:150-193
 *   task_error for the 7 task types       src/tasks.jl:31-44,73-84,112-123,153-171,185-189,204-236,258-262
 *   ContactPoint, forcebasis, z_up        src/contacts.jl:8-23,38-69,75-79
 */
#include "oracle.h"
#include "ora_osqp.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define INF 1e30

typedef struct {
  int kind, base, target, frame, joint, wkind, dim, des_off, tw_off, var0, row0;
  double point[3], W[36];
} ora_task;
typedef struct {
  int body, per_qp, surf_off, var0, row0;
  double pos[3], normal[3], mu;
} ora_contact;

struct ora_ctrl {
  const ora_model *m;
  int N, floating;
  ora_task tasks[ORA_MAXTASK];
  int nt;
  ora_contact contacts[ORA_MAXCONTACT];
  int nc;
  double reg[ORA_MAXV];
  int reg_tw[ORA_MAXV]; /* regularize! with a Parameter weight (momentum.jl:128-131 takes any `weight`): index into tw or -1 */
  int nvar, nrow, neq, nineq, n_desired, n_tw, n_surf, wb_row0, finalized;
  ora_state *s;
  ora_osqp_settings st;
  ora_osqp *osqp;
  /* dense scratch + structural masks */
  double *P, *q, *A, *l, *u;
  char *Amask, *Pmask;
  int *Pp, *Pi, *Ap, *Ai;
  double *Pxv, *Axv, *xsol, *ysol;
};

static void cross3(const double a[3], const double b[3], double c[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static void mv3(const double R[9], const double x[3], double y[3]) {
  double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2], b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2],
         c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
static void mtv3(const double R[9], const double x[3], double y[3]) {
  double a = R[0] * x[0] + R[3] * x[1] + R[6] * x[2], b = R[1] * x[0] + R[4] * x[1] + R[7] * x[2],
         c = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}

static int task_dim_of(const ora_model *m, int kind, int joint) {
  switch (kind) {
    case QPC_TASK_SPATIAL_ACCELERATION: case QPC_TASK_MOMENTUM_RATE: return 6;
    case QPC_TASK_ANGULAR_ACCELERATION: case QPC_TASK_LINEAR_ACCELERATION:
    case QPC_TASK_POINT_ACCELERATION: case QPC_TASK_LINEAR_MOMENTUM_RATE: return 3;
    case QPC_TASK_JOINT_ACCELERATION: return m->nvj[joint];
  }
  return -1;
}

ora_ctrl *ora_ctrl_create(const ora_model *m, int n_basis, int floating_body) {
  if (n_basis < 1 || n_basis > ORA_MAXBASIS) return NULL;
  ora_ctrl *c = (ora_ctrl *)calloc(1, sizeof(ora_ctrl));
  c->m = m; c->N = n_basis; c->floating = floating_body;
  c->nvar = m->nv; /* v̇ = [Variable(qpmodel) for _ = 1:nv], momentum.jl:28 */
  c->s = ora_state_create();
  ora_osqp_test_settings(&c->st);
  c->st.polish = 2;
  return c;
}
void ora_ctrl_destroy(ora_ctrl *c) {
  if (!c) return;
  ora_state_destroy(c->s);
  if (c->osqp) ora_osqp_free(c->osqp);
  free(c->P); free(c->q); free(c->A); free(c->l); free(c->u); free(c->Amask); free(c->Pmask);
  free(c->Pp); free(c->Pi); free(c->Ap); free(c->Ai); free(c->Pxv); free(c->Axv);
  free(c->xsol); free(c->ysol);
  free(c);
}
void ora_set_osqp_settings(ora_ctrl *c, const ora_osqp_settings *s) {
  c->st = *s;
  if (c->osqp) c->osqp->st = *s;
}

/* addtask! : hard -> dim equality rows; soft -> dim slack variables + dim rows e == err */
int ora_add_task(ora_ctrl *c, int kind, int base, int target, int frame, const double *point,
                 int joint, int wkind, const double *weight) {
  if (c->finalized || c->nt >= ORA_MAXTASK) return -1;
  ora_task *t = &c->tasks[c->nt];
  memset(t, 0, sizeof(*t));
  t->kind = kind; t->base = base; t->target = target; t->joint = joint; t->wkind = wkind;
  t->frame = frame == QPC_FRAME_DEFAULT ? target : frame;
  if (kind == QPC_TASK_POINT_ACCELERATION) t->frame = base; /* src/tasks.jl:139-140,154 */
  if (point) memcpy(t->point, point, 24);
  t->dim = task_dim_of(c->m, kind, joint);
  if (t->dim <= 0) return -1;
  t->des_off = c->n_desired; c->n_desired += t->dim;
  t->tw_off = -1; t->var0 = -1;
  if (wkind != QPC_WEIGHT_HARD) {
    t->var0 = c->nvar; c->nvar += t->dim; /* momentum.jl:123 */
    if (wkind == QPC_WEIGHT_SCALAR) t->W[0] = weight[0];
    else if (wkind == QPC_WEIGHT_MATRIX) memcpy(t->W, weight, sizeof(double) * t->dim * t->dim);
    else if (wkind == QPC_WEIGHT_SCALAR_PER_QP) { t->tw_off = c->n_tw; c->n_tw += 1; }
    else if (wkind == QPC_WEIGHT_MATRIX_PER_QP) { t->tw_off = c->n_tw; c->n_tw += t->dim * t->dim; }
    else return -1;
  }
  t->row0 = c->nrow; c->nrow += t->dim; c->neq += t->dim;
  return c->nt++;
}
int ora_regularize(ora_ctrl *c, int joint, double w) {
  if (c->finalized) return -1;
  for (int k = 0; k < c->m->nvj[joint]; k++) c->reg[c->m->voff[joint] + k] += w;
  return 0;
}
/* regularize!(c, joint, weight::Parameter) (momentum.jl:128-131): the weight is re-evaluated every tick; here it is entry
 * `return value` of the per-tick task-weight record.  One parametric weight per joint (static ones still add up). */
int ora_regularize_per_qp(ora_ctrl *c, int joint) {
  if (c->finalized || c->reg_tw[c->m->voff[joint]] > 0) return -1;
  const int off = c->n_tw;
  for (int k = 0; k < c->m->nvj[joint]; k++) c->reg_tw[c->m->voff[joint] + k] = off + 1; /* 0 = none (calloc) */
  c->n_tw += 1;
  return off;
}
/* ContactPoint{N}: rho[N], force_local[3], wrench_world angular[3], linear[3] (contacts.jl:46-48)
 * and 3 + N + N + 3 + 3 rows (contacts.jl:63-67) */
int ora_add_contact(ora_ctrl *c, int body, const double pos[3], const double normal[3], double mu,
                    int per_qp_surface) {
  if (c->finalized || c->nc >= ORA_MAXCONTACT) return -1;
  ora_contact *k = &c->contacts[c->nc];
  memset(k, 0, sizeof(*k));
  k->body = body; k->mu = mu; k->per_qp = per_qp_surface;
  memcpy(k->pos, pos, 24); memcpy(k->normal, normal, 24);
  k->surf_off = -1;
  if (per_qp_surface) { k->surf_off = c->n_surf; c->n_surf += ((per_qp_surface & 1) ? 4 : 0) + ((per_qp_surface & 2) ? 3 : 0); }
  k->var0 = c->nvar; c->nvar += c->N + 9;
  k->row0 = c->nrow; c->nrow += 2 * c->N + 9;
  c->neq += 9; c->nineq += 2 * c->N;
  return c->nc++;
}
int ora_task_dim(const ora_ctrl *c, int task) { return c->tasks[task].dim; }

static void fill_layout(const ora_ctrl *c, qpc_layout *L) {
  memset(L, 0, sizeof(*L));
  L->nq = c->m->nq; L->nv = c->m->nv; L->n_tasks = c->nt; L->n_contacts = c->nc; L->n_basis = c->N;
  L->n_desired = c->n_desired; L->n_task_weights = c->n_tw; L->n_surface = c->n_surf;
  L->ref_n_var = c->nvar; L->ref_n_eq = c->neq; L->ref_n_ineq = c->nineq;
  L->n_desired_inputs = c->n_desired; /* the oracle has no device laws: setdesired! semantics only */
}

/* J (dim x nv) and bias with task_error = J v̇ + bias - desired */
int ora_task_eval(const ora_ctrl *c, const ora_state *s, int ti, double *J, double *bias) {
  const ora_model *m = c->m;
  const ora_task *t = &c->tasks[ti];
  int nv = m->nv;
  double J6[6 * ORA_MAXV], b6[6], dA[6];
  memset(J, 0, sizeof(double) * t->dim * nv);
  switch (t->kind) {
    case QPC_TASK_SPATIAL_ACCELERATION: case QPC_TASK_ANGULAR_ACCELERATION:
    case QPC_TASK_LINEAR_ACCELERATION: case QPC_TASK_POINT_ACCELERATION: {
      ora_geometric_jacobian(m, s, t->base, t->target, t->frame, J6);
      for (int k = 0; k < 6; k++)
        dA[k] = (t->target < 0 ? 0.0 : s->Ab[t->target][k]) - (t->base < 0 ? 0.0 : s->Ab[t->base][k]);
      ora_transform_accel(m, s, dA, t->target, t->base, t->frame, b6);
      if (t->kind == QPC_TASK_SPATIAL_ACCELERATION) {
        memcpy(J, J6, sizeof(double) * 6 * nv); memcpy(bias, b6, 48);
      } else if (t->kind == QPC_TASK_ANGULAR_ACCELERATION) {
        memcpy(J, J6, sizeof(double) * 3 * nv); memcpy(bias, b6, 24);
      } else if (t->kind == QPC_TASK_LINEAR_ACCELERATION) {
        memcpy(J, J6 + 3 * nv, sizeof(double) * 3 * nv); memcpy(bias, b6 + 3, 24);
      } else {
        /* point in the task (= base) frame */
        double pw[3], pF[3], T[6], pd[3], t1[3], t2[3];
        if (t->target >= 0) {
          mv3(s->R[t->target], t->point, pw);
          for (int k = 0; k < 3; k++) pw[k] += s->p[t->target][k];
        } else memcpy(pw, t->point, 24);
        if (t->frame >= 0) {
          double d[3] = {pw[0] - s->p[t->frame][0], pw[1] - s->p[t->frame][1], pw[2] - s->p[t->frame][2]};
          mtv3(s->R[t->frame], d, pF);
        } else memcpy(pF, pw, 24);
        /* point_jacobian: J_lin - p^ J_ang   (src/tasks.jl:157-160) */
        for (int k = 0; k < nv; k++) {
          double ja[3] = {J6[k], J6[nv + k], J6[2 * nv + k]}, pxj[3];
          cross3(pF, ja, pxj);
          for (int r = 0; r < 3; r++) J[r * nv + k] = J6[(3 + r) * nv + k] - pxj[r];
        }
        ora_relative_twist(m, s, t->target, t->base, t->frame, T);
        cross3(T, pF, pd);
        for (int k = 0; k < 3; k++) pd[k] += T[3 + k]; /* point_velocity */
        cross3(T, pd, t1);                             /* omega x pdot */
        cross3(b6, pF, t2);                            /* J̇v_ang x p */
        for (int k = 0; k < 3; k++) bias[k] = t1[k] + t2[k] + b6[3 + k];
      }
      break;
    }
    case QPC_TASK_JOINT_ACCELERATION:
      for (int k = 0; k < t->dim; k++) { J[k * nv + m->voff[t->joint] + k] = 1.0; bias[k] = 0.0; }
      break;
    case QPC_TASK_MOMENTUM_RATE: case QPC_TASK_LINEAR_MOMENTUM_RATE: {
      /* world -> centroidal: pure translation by -com (src/tasks.jl:208-219) */
      const double *cm = s->com;
      for (int k = 0; k < nv; k++) {
        double lin[3] = {s->Amom[3][k], s->Amom[4][k], s->Amom[5][k]}, cxl[3];
        cross3(cm, lin, cxl);
        if (t->kind == QPC_TASK_MOMENTUM_RATE) {
          for (int r = 0; r < 3; r++) { J[r * nv + k] = s->Amom[r][k] - cxl[r]; J[(3 + r) * nv + k] = lin[r]; }
        } else for (int r = 0; r < 3; r++) J[r * nv + k] = lin[r];
      }
      double cxl[3];
      cross3(cm, s->Adv + 3, cxl);
      if (t->kind == QPC_TASK_MOMENTUM_RATE) {
        for (int r = 0; r < 3; r++) { bias[r] = s->Adv[r] - cxl[r]; bias[3 + r] = s->Adv[3 + r]; }
      } else for (int r = 0; r < 3; r++) bias[r] = s->Adv[3 + r];
      break;
    }
    default: return -1;
  }
  return 0;
}

/* Rotations.rotation_between(z, n) (contacts.jl:11) [recalled, SURVEY B.6] */
static void rotation_between_z(const double n[3], double R[9]) {
  double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  double w = nn + n[2], x, y, z;
  if (fabs(w) < 100 * 2.220446049250313e-16) { x = 1; y = 0; z = 0; } /* Rotations.perpendicular_vector((0,0,1)) = (1,0,0): pi about x */
  else { x = -n[1]; y = n[0]; z = 0; }
  double qn = sqrt(w * w + x * x + y * y + z * z);
  w /= qn; x /= qn; y /= qn; z /= qn;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

void ora_contact_eval(const ora_ctrl *c, const ora_state *s, int ci, double normal[3], double mu,
                      double maxnormalforce, double *B, double *maxrho, double R[9], double p[3]) {
  const ora_contact *k = &c->contacts[ci];
  int N = c->N;
  /* forcebasis (contacts.jl:16-23) */
  for (int i = 0; i < N; i++) {
    double th = i * (2 * M_PI / N);
    double b[3] = {mu * cos(th), mu * sin(th), 1.0};
    double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    for (int r = 0; r < 3; r++) B[r * N + i] = b[r] / nb;
  }
  *maxrho = maxnormalforce / (N * sqrt(mu * mu + 1)); /* contacts.jl:57 */
  /* toroot = transform_to_root(body) * z_up_transform(position, normal) (contacts.jl:61) */
  double Rn[9];
  rotation_between_z(normal, Rn);
  if (k->body >= 0) {
    const double *Rb = s->R[k->body];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        R[3 * a + b] = Rb[3 * a] * Rn[b] + Rb[3 * a + 1] * Rn[3 + b] + Rb[3 * a + 2] * Rn[6 + b];
    mv3(Rb, k->pos, p);
    for (int a = 0; a < 3; a++) p[a] += s->p[k->body][a];
  } else { memcpy(R, Rn, 72); memcpy(p, k->pos, 24); }
}

int ora_finalize(ora_ctrl *c, qpc_layout *out) {
  if (c->finalized) { if (out) fill_layout(c, out); return 0; }
  const ora_model *m = c->m;
  int nv = m->nv;
  /* initialize!: wrench balance rows are added last (momentum.jl:153-155,190) */
  c->wb_row0 = -1;
  if (c->floating >= 0) { c->wb_row0 = c->nrow; c->nrow += 6; c->neq += 6; }
  int n = c->nvar, mr = c->nrow;
  c->P = (double *)calloc((size_t)n * n, 8); c->q = (double *)calloc(n, 8);
  c->A = (double *)calloc((size_t)(mr > 0 ? mr : 1) * n, 8);
  c->l = (double *)calloc(mr > 0 ? mr : 1, 8); c->u = (double *)calloc(mr > 0 ? mr : 1, 8);
  c->Amask = (char *)calloc((size_t)(mr > 0 ? mr : 1) * n, 1); c->Pmask = (char *)calloc((size_t)n * n, 1);
  c->xsol = (double *)calloc(n, 8); c->ysol = (double *)calloc(mr > 0 ? mr : 1, 8);
  /* structural pattern as Parametron would emit it: dense mat-vec products keep structural
   * zeros (SURVEY Appendix A "stored nonzeros") */
  int N = c->N;
  for (int i = 0; i < n; i++) c->Pmask[i * n + i] = 1;
  for (int ti = 0; ti < c->nt; ti++) {
    ora_task *t = &c->tasks[ti];
    for (int r = 0; r < t->dim; r++) {
      char *row = c->Amask + (size_t)(t->row0 + r) * n;
      if (t->kind == QPC_TASK_JOINT_ACCELERATION) row[m->voff[t->joint] + r] = 1;
      else for (int k = 0; k < nv; k++) row[k] = 1;
      if (t->var0 >= 0) row[t->var0 + r] = 1;
    }
    if (t->var0 >= 0 && (t->wkind == QPC_WEIGHT_MATRIX || t->wkind == QPC_WEIGHT_MATRIX_PER_QP))
      for (int a = 0; a < t->dim; a++)
        for (int b = a; b < t->dim; b++) c->Pmask[(t->var0 + a) * n + t->var0 + b] = 1;
  }
  for (int ci = 0; ci < c->nc; ci++) {
    ora_contact *k = &c->contacts[ci];
    int rho = k->var0, f = rho + N, wa = f + 3, wl = wa + 3, r0 = k->row0;
    for (int r = 0; r < 3; r++) {
      char *row = c->Amask + (size_t)(r0 + r) * n;
      row[f + r] = 1;
      for (int i = 0; i < N; i++) row[rho + i] = 1;
    }
    for (int i = 0; i < N; i++) {
      c->Amask[(size_t)(r0 + 3 + i) * n + rho + i] = 1;
      c->Amask[(size_t)(r0 + 3 + N + i) * n + rho + i] = 1;
    }
    for (int r = 0; r < 3; r++) {
      char *row = c->Amask + (size_t)(r0 + 3 + 2 * N + r) * n;
      row[wl + r] = 1;
      for (int i = 0; i < 3; i++) row[f + i] = 1;
      row = c->Amask + (size_t)(r0 + 6 + 2 * N + r) * n;
      row[wa + r] = 1;
      for (int i = 0; i < 3; i++) row[wl + i] = 1;
    }
  }
  if (c->wb_row0 >= 0)
    for (int r = 0; r < 6; r++) {
      char *row = c->Amask + (size_t)(c->wb_row0 + r) * n;
      for (int k = 0; k < nv; k++) row[k] = 1;
      for (int ci = 0; ci < c->nc; ci++)
        for (int i = 0; i < 6; i++) row[c->contacts[ci].var0 + N + 3 + i] = 1;
    }
  /* CSC patterns */
  c->Pp = (int *)calloc(n + 1, 4); c->Ap = (int *)calloc(n + 1, 4);
  int pn = 0, an = 0;
  for (int j = 0; j < n; j++) {
    for (int i = 0; i <= j; i++) if (c->Pmask[i * n + j]) pn++;
    c->Pp[j + 1] = pn;
    for (int i = 0; i < mr; i++) if (c->Amask[(size_t)i * n + j]) an++;
    c->Ap[j + 1] = an;
  }
  c->Pi = (int *)calloc(pn > 0 ? pn : 1, 4); c->Ai = (int *)calloc(an > 0 ? an : 1, 4);
  c->Pxv = (double *)calloc(pn > 0 ? pn : 1, 8); c->Axv = (double *)calloc(an > 0 ? an : 1, 8);
  pn = an = 0;
  for (int j = 0; j < n; j++) {
    for (int i = 0; i <= j; i++) if (c->Pmask[i * n + j]) c->Pi[pn++] = i;
    for (int i = 0; i < mr; i++) if (c->Amask[(size_t)i * n + j]) c->Ai[an++] = i;
  }
  c->osqp = ora_osqp_setup(n, mr, c->Pp, c->Pi, c->Ap, c->Ai, &c->st);
  if (!c->osqp) return -1;
  c->finalized = 1;
  if (out) fill_layout(c, out);
  return 0;
}

/* evaluate every Parameter and rewrite the coefficients (what Parametron.solve! does before
 * MOI.optimize!) into the dense scratch matrices */
static void build_values(ora_ctrl *c, const double *q, const double *v, const double *desired,
                         const double *cweight, const double *cmaxf, const double *tw,
                         const double *surf) {
  const ora_model *m = c->m;
  ora_state *s = c->s;
  int n = c->nvar, mr = c->nrow, nv = m->nv, N = c->N;
  ora_state_update(m, q, v, s);
  memset(c->P, 0, sizeof(double) * n * n);
  memset(c->q, 0, sizeof(double) * n);
  memset(c->A, 0, sizeof(double) * (size_t)mr * n);
  /* objective: regularize! terms, P = 2 * weight */
  for (int k = 0; k < nv; k++) c->P[k * n + k] = 2.0 * (c->reg[k] + (c->reg_tw[k] > 0 ? tw[c->reg_tw[k] - 1] : 0.0));
  double J[6 * ORA_MAXV], bias[6];
  for (int ti = 0; ti < c->nt; ti++) {
    ora_task *t = &c->tasks[ti];
    ora_task_eval(c, s, ti, J, bias);
    for (int r = 0; r < t->dim; r++) {
      double *row = c->A + (size_t)(t->row0 + r) * n;
      double rhs = desired[t->des_off + r] - bias[r];
      if (t->var0 < 0) {
        /* err == 0  <=>  J v̇ = desired - bias */
        for (int k = 0; k < nv; k++) row[k] = J[r * nv + k];
        c->l[t->row0 + r] = c->u[t->row0 + r] = rhs;
      } else {
        /* e == J v̇ + bias - desired  <=>  e - J v̇ = bias - desired */
        for (int k = 0; k < nv; k++) row[k] = -J[r * nv + k];
        row[t->var0 + r] = 1.0;
        c->l[t->row0 + r] = c->u[t->row0 + r] = -rhs;
      }
    }
    if (t->var0 >= 0) {
      if (t->wkind == QPC_WEIGHT_SCALAR || t->wkind == QPC_WEIGHT_SCALAR_PER_QP) {
        double w = t->wkind == QPC_WEIGHT_SCALAR ? t->W[0] : tw[t->tw_off];
        for (int r = 0; r < t->dim; r++) c->P[(t->var0 + r) * n + t->var0 + r] = 2.0 * w;
      } else {
        const double *W = t->wkind == QPC_WEIGHT_MATRIX ? t->W : tw + t->tw_off;
        for (int a = 0; a < t->dim; a++)
          for (int b = 0; b < t->dim; b++)
            c->P[(t->var0 + a) * n + t->var0 + b] = W[a * t->dim + b] + W[b * t->dim + a];
      }
    }
  }
  double tot_ang_cols = 0; (void)tot_ang_cols;
  for (int ci = 0; ci < c->nc; ci++) {
    ora_contact *k = &c->contacts[ci];
    int rho = k->var0, f = rho + N, wa = f + 3, wl = wa + 3, r0 = k->row0;
    double B[3 * ORA_MAXBASIS], maxrho, R[9], p[3], nrm[3], mu = k->mu;
    memcpy(nrm, k->normal, 24);
    /* position / normal / mu may be Parameters (contacts.jl:39): per-QP record [nx ny nz mu][px py pz] */
    double pos_static[3];
    memcpy(pos_static, k->pos, 24);
    if (k->per_qp) {
      int so = k->surf_off;
      if (k->per_qp & 1) { memcpy(nrm, surf + so, 24); mu = surf[so + 3]; so += 4; }
      if (k->per_qp & 2) memcpy(k->pos, surf + so, 24);
    }
    ora_contact_eval(c, s, ci, nrm, mu, cmaxf[ci], B, &maxrho, R, p);
    memcpy(k->pos, pos_static, 24);
    /* objectiveterm: weight * (f . f) (contacts.jl:75-79) */
    for (int r = 0; r < 3; r++) c->P[(f + r) * n + f + r] = 2.0 * cweight[ci];
    /* force_local == basis * rho (contacts.jl:63) */
    for (int r = 0; r < 3; r++) {
      double *row = c->A + (size_t)(r0 + r) * n;
      row[f + r] = 1.0;
      for (int i = 0; i < N; i++) row[rho + i] = -B[r * N + i];
      c->l[r0 + r] = c->u[r0 + r] = 0.0;
    }
    /* rho >= 0 ; rho <= maxrho (contacts.jl:64-65) */
    for (int i = 0; i < N; i++) {
      c->A[(size_t)(r0 + 3 + i) * n + rho + i] = 1.0;
      c->l[r0 + 3 + i] = 0.0; c->u[r0 + 3 + i] = INF;
      c->A[(size_t)(r0 + 3 + N + i) * n + rho + i] = 1.0;
      c->l[r0 + 3 + N + i] = -INF; c->u[r0 + 3 + N + i] = maxrho;
    }
    /* linear(wrench_world) == R f (contacts.jl:66) */
    for (int r = 0; r < 3; r++) {
      double *row = c->A + (size_t)(r0 + 3 + 2 * N + r) * n;
      row[wl + r] = 1.0;
      for (int i = 0; i < 3; i++) row[f + i] = -R[3 * r + i];
      c->l[r0 + 3 + 2 * N + r] = c->u[r0 + 3 + 2 * N + r] = 0.0;
    }
    /* angular(wrench_world) == hat(p) * linear(wrench_world) (contacts.jl:67) */
    double hat[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
    for (int r = 0; r < 3; r++) {
      double *row = c->A + (size_t)(r0 + 6 + 2 * N + r) * n;
      row[wa + r] = 1.0;
      for (int i = 0; i < 3; i++) row[wl + i] = -hat[3 * r + i];
      c->l[r0 + 6 + 2 * N + r] = c->u[r0 + 6 + 2 * N + r] = 0.0;
    }
  }
  if (c->wb_row0 >= 0) {
    /* S' (A v̇ + Ȧv - Wg - sum wrench_world) == 0 (momentum.jl:162-190) */
    double fg[3] = {m->mtot * m->g[0], m->mtot * m->g[1], m->mtot * m->g[2]}, Wg[6];
    cross3(s->com, fg, Wg);
    Wg[3] = fg[0]; Wg[4] = fg[1]; Wg[5] = fg[2];
    int v0 = m->voff[c->floating];
    for (int r = 0; r < 6; r++) {
      const double *S = s->S[v0 + r];
      double *row = c->A + (size_t)(c->wb_row0 + r) * n;
      for (int k = 0; k < nv; k++) {
        double t = 0;
        for (int a = 0; a < 6; a++) t += S[a] * s->Amom[a][k];
        row[k] = t;
      }
      for (int ci = 0; ci < c->nc; ci++) {
        int wa = c->contacts[ci].var0 + N + 3;
        for (int a = 0; a < 6; a++) row[wa + a] = -S[a];
      }
      double rhs = 0;
      for (int a = 0; a < 6; a++) rhs += S[a] * (Wg[a] - s->Adv[a]);
      c->l[c->wb_row0 + r] = c->u[c->wb_row0 + r] = rhs;
    }
  }
}

int ora_build_qp(ora_ctrl *c, const double *q, const double *v, const double *desired,
                 const double *cweight, const double *cmaxf, const double *tw, const double *surf,
                 double *P, double *qv, double *A, double *l, double *u) {
  if (!c->finalized) return -1;
  build_values(c, q, v, desired, cweight, cmaxf, tw, surf);
  int n = c->nvar, mr = c->nrow;
  memcpy(P, c->P, sizeof(double) * n * n);
  memcpy(qv, c->q, sizeof(double) * n);
  memcpy(A, c->A, sizeof(double) * (size_t)mr * n);
  memcpy(l, c->l, sizeof(double) * mr);
  memcpy(u, c->u, sizeof(double) * mr);
  return 0;
}

int ora_tick(ora_ctrl *c, const double *q, const double *v, const double *desired,
             const double *cweight, const double *cmaxf, const double *tw, const double *surf,
             int warm, double *tau, double *vdot, double *cwrench, double *xfull,
             ora_tick_info *info) {
  if (!c->finalized) return -1;
  const ora_model *m = c->m;
  int n = c->nvar, mr = c->nrow, nv = m->nv, N = c->N;
  build_values(c, q, v, desired, cweight, cmaxf, tw, surf);
  int pn = 0, an = 0;
  for (int j = 0; j < n; j++) {
    for (int p = c->Pp[j]; p < c->Pp[j + 1]; p++) c->Pxv[pn++] = c->P[c->Pi[p] * n + j];
    for (int p = c->Ap[j]; p < c->Ap[j + 1]; p++) c->Axv[an++] = c->A[(size_t)c->Ai[p] * n + j];
  }
  ora_tick_info loc;
  if (!info) info = &loc;
  c->osqp->st = c->st;
  if (!warm) ora_osqp_cold_start(c->osqp);
  if (ora_osqp_update(c->osqp, c->Pxv, c->q, c->Axv, c->l, c->u) != 0) {
    memset(info, 0, sizeof(*info));
    info->status = QPC_STATUS_NUMERICAL_ERROR;
    return info->status;
  }
  (void)mr;
  int status = ora_osqp_solve(c->osqp, c->xsol, c->ysol, info);
  /* value(qpmodel, v̇[i]) (momentum.jl:62-64) */
  for (int k = 0; k < nv; k++) vdot[k] = c->xsol[k];
  /* contact wrenches per body (momentum.jl:65-72) */
  double wext[ORA_MAXB * 6];
  memset(wext, 0, sizeof(wext));
  for (int ci = 0; ci < c->nc; ci++) {
    int wa = c->contacts[ci].var0 + N + 3;
    for (int a = 0; a < 6; a++) {
      if (cwrench) cwrench[6 * ci + a] = c->xsol[wa + a];
      if (c->contacts[ci].body >= 0) wext[6 * c->contacts[ci].body + a] += c->xsol[wa + a];
    }
  }
  ora_inverse_dynamics(m, c->s, vdot, wext, tau); /* momentum.jl:75 */
  if (c->floating >= 0)                           /* zero_floating_joint_torques! momentum.jl:93-97 */
    for (int k = 0; k < m->nvj[c->floating]; k++) tau[m->voff[c->floating] + k] = 0.0;
  if (xfull) memcpy(xfull, c->xsol, sizeof(double) * n);
  return status;
}

int ora_tick_batch(ora_ctrl *c, int64_t B, const double *q, const double *v, const double *desired,
                   const double *cweight, const double *cmaxf, const double *tw, const double *surf,
                   int warm, double *tau, double *vdot, double *cwrench, int32_t *status,
                   int32_t *iters) {
  const ora_model *m = c->m;
  int bad = 0;
  double dummy = 0;
  for (int64_t b = 0; b < B; b++) {
    ora_tick_info info;
    int st = ora_tick(c, q + b * m->nq, v + b * m->nv, c->n_desired ? desired + b * c->n_desired : &dummy,
                      c->nc ? cweight + b * c->nc : &dummy, c->nc ? cmaxf + b * c->nc : &dummy,
                      c->n_tw ? tw + b * c->n_tw : NULL, c->n_surf ? surf + b * c->n_surf : NULL,
                      warm && b > 0, tau + b * m->nv, vdot + b * m->nv,
                      cwrench ? cwrench + b * 6 * c->nc : NULL, NULL, &info);
    if (status) status[b] = st;
    if (iters) iters[b] = info.iters;
    if (st != QPC_STATUS_OPTIMAL) bad++;
  }
  return bad;
}
