/*
 * oracle.h -- CPU restatement (plain C, fp64) of the QPControl.jl hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under qpcontrol.jl_amd/ (the product) may include, link,
 * import or call anything in oracle/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference is Julia whose arithmetic lives in un-vendored packages
 * (RigidBodyDynamics 2.2.0, Parametron 0.9.1, MathOptInterface 0.8.4, OSQP.jl 0.5.2 -> OSQP
 * 0.5.0 C library; pins: /root/reference/Manifest.toml:87-140, test/Manifest.toml:440-444).
 * None of them, and no `julia` binary, exist in this image, and the reference tree holds no
 * numeric golden vectors.  This restatement therefore follows the reference's call
 * sites (cited per function) plus the published algorithms of those packages, and is pinned by
 * the *property* tests the reference's own test-suite asserts (tests/test_oracle_*.py).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stdint.h>
#include "../include/qpc.h" /* shares qpc_model_desc and the enums with the product ABI */

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAXB 64   /* max moving bodies */
#define ORA_MAXV 96   /* max velocity dimension */
#define ORA_MAXQ 104
#define ORA_MAXTASK 96
#define ORA_MAXCONTACT 96
#define ORA_MAXBASIS 8

typedef struct ora_model {
  int nb, nq, nv;
  int parent[ORA_MAXB], jtype[ORA_MAXB];
  double axis[ORA_MAXB][3], Rf[ORA_MAXB][9], pf[ORA_MAXB][3];
  double mass[ORA_MAXB], com[ORA_MAXB][3], Icom[ORA_MAXB][6];
  int qoff[ORA_MAXB], voff[ORA_MAXB], nqj[ORA_MAXB], nvj[ORA_MAXB];
  int vbody[ORA_MAXV]; /* body (joint) owning velocity index k */
  double g[3];
  double mtot;
} ora_model;

/* spatial inertia in world frame about the world origin */
typedef struct ora_inertia {
  double m, h[3] /* m*c */, J[6] /* xx yy zz xy xz yz */;
} ora_inertia;

typedef struct ora_state {
  double q[ORA_MAXQ], v[ORA_MAXV];
  double R[ORA_MAXB][9], p[ORA_MAXB][3]; /* transform_to_root: x_w = R x_b + p */
  double S[ORA_MAXV][6];                 /* motion subspace columns in world frame */
  double T[ORA_MAXB][6];                 /* twist_wrt_world */
  double Ab[ORA_MAXB][6];                /* bias_acceleration (v̇=0, no gravity), world */
  ora_inertia Iw[ORA_MAXB], Ic[ORA_MAXB]; /* body and composite inertias, world */
  double com[3];
  double Amom[6][ORA_MAXV]; /* momentum_matrix, world frame about world origin */
  double Adv[6];            /* momentum_rate_bias */
} ora_state;

ora_model *ora_model_create(const qpc_model_desc *d);
void ora_model_destroy(ora_model *m);
int ora_model_nq(const ora_model *m);
int ora_model_nv(const ora_model *m);

ora_state *ora_state_create(void);
void ora_state_destroy(ora_state *s);
/* copyto!(state, x) + all cache sweeps (momentum.jl:57 and RBD caches) */
void ora_state_update(const ora_model *m, const double *q, const double *v, ora_state *s);

/* accessors used by the tests (flat outputs) */
void ora_get_transform(const ora_state *s, int body, double R[9], double p[3]);
void ora_get_twist(const ora_state *s, int body, double T[6]);
void ora_get_bias(const ora_state *s, int body, double A[6]);
void ora_get_com(const ora_state *s, double c[3]);
void ora_get_momentum_matrix(const ora_model *m, const ora_state *s, double *A /*6*nv row-major*/,
                             double Adv[6]);
/* geometric_jacobian!(jac, state, path(base->target), world->frame) : J is 6*nv row-major */
void ora_geometric_jacobian(const ora_model *m, const ora_state *s, int base, int target,
                            int frame, double *J);
/* mass matrix (nv*nv row-major), inverse dynamics with external wrenches per body in WORLD frame
 * (wext[nb*6] or NULL), forward dynamics */
void ora_mass_matrix(const ora_model *m, const ora_state *s, double *M);
void ora_inverse_dynamics(const ora_model *m, const ora_state *s, const double *vd,
                          const double *wext, double *tau);
int ora_forward_dynamics(const ora_model *m, const ora_state *s, const double *tau,
                         const double *wext, double *vd);
/* spatial accelerations of all bodies (world frame, gravity NOT included) for given v̇ */
void ora_spatial_accelerations(const ora_model *m, const ora_state *s, const double *vd,
                               double *acc /* nb*6 */);
/* q <- q (+) v*dt on the configuration manifold (quaternion exponential for floating joints) */
void ora_integrate(const ora_model *m, double *q, const double *v, double dt);
/* relative twist of `body` w.r.t. `base` expressed in `frame`; and transform of a spatial
 * acceleration (world coords, body wrt base) into `frame` including the velocity cross term */
void ora_relative_twist(const ora_model *m, const ora_state *s, int body, int base, int frame,
                        double T[6]);
void ora_transform_accel(const ora_model *m, const ora_state *s, const double Aw[6], int body,
                         int base, int frame, double out[6]);

/* PD laws of the StandingController tick (src/highlevel/standing.jl:58-89), ora_standing.c */
void ora_standing_com_law(const ora_model *m, const ora_state *s, const double *v, double kp, double kd,
                          const double comref[3], double out[3]);
void ora_standing_orientation_law(const ora_state *s, int body, double kp, double kd, double out[3]);
double ora_standing_joint_law(double kp, double kd, double q, double qref, double v);

/* ------------------------------------------------------------------------------------------
 * controller (reference layout QP)
 * ---------------------------------------------------------------------------------------- */
typedef struct ora_ctrl ora_ctrl;

typedef struct ora_osqp_settings {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, scaling, adaptive_rho, adaptive_rho_interval, check_termination, warm_start;
  double adaptive_rho_tolerance;
  int polish;        /* 0 = plain OSQP (what the reference runs); 1 = OSQP polish (delta 1e-6, 3 its);
                        2 = "exact" mode: polish with refinement to machine precision + KKT
                        certificate, re-running ADMM if the active-set guess is wrong */
} ora_osqp_settings;
void ora_osqp_default_settings(ora_osqp_settings *s); /* OSQP 0.5 defaults */
void ora_osqp_test_settings(ora_osqp_settings *s);     /* test/runtests.jl:35-43 */
void ora_osqp_notebook_settings(ora_osqp_settings *s); /* notebooks/Standing controller.ipynb:68-71 */

ora_ctrl *ora_ctrl_create(const ora_model *m, int n_basis, int floating_body);
void ora_ctrl_destroy(ora_ctrl *c);
int ora_add_task(ora_ctrl *c, int kind, int base, int target, int frame, const double *point,
                 int joint, int wkind, const double *weight);
int ora_regularize(ora_ctrl *c, int joint, double w);
int ora_regularize_per_qp(ora_ctrl *c, int joint); /* -> index of the weight in the per-tick task-weight record, or -1 */
int ora_add_contact(ora_ctrl *c, int body, const double pos[3], const double normal[3], double mu,
                    int per_qp_surface);
int ora_finalize(ora_ctrl *c, qpc_layout *out);
void ora_set_osqp_settings(ora_ctrl *c, const ora_osqp_settings *s);

/* J (dim*nv row-major) and bias such that  task_error = J*v̇ + bias - desired  (src/tasks.jl) */
int ora_task_eval(const ora_ctrl *c, const ora_state *s, int task, double *J, double *bias);
int ora_task_dim(const ora_ctrl *c, int task);
/* contact quantities (contacts.jl:53-67): basis B (3*N row-major), maxrho, R(9), p(3) */
void ora_contact_eval(const ora_ctrl *c, const ora_state *s, int contact, double normal[3],
                      double mu, double maxnormalforce, double *B, double *maxrho, double R[9],
                      double p[3]);

/* full reference-layout problem for one state: sizes from layout.ref_*; dense outputs
 * P (n*n), qv (n), A (m*n row-major), l (m), u (m).  Row order = @constraint call order. */
int ora_build_qp(ora_ctrl *c, const double *q, const double *v, const double *desired,
                 const double *cweight, const double *cmaxf, const double *task_weights,
                 const double *surfaces, double *P, double *qv, double *A, double *l, double *u);

typedef struct ora_tick_info {
  int status, iters, polish_ok, admm_restarts;
  double prim_res, dual_res;           /* OSQP residuals (unscaled) at exit */
  double kkt_stat, kkt_prim, kkt_comp; /* KKT certificate of the returned point */
  double obj;
} ora_tick_info;

/* one full tick (momentum.jl:41-81) for ONE state through the reference-layout QP + OSQP
 * restatement.  warm != 0 keeps the previous solution as OSQP warm start (the reference's
 * sequential behaviour); xfull (ref_n_var) may be NULL. */
int ora_tick(ora_ctrl *c, const double *q, const double *v, const double *desired,
             const double *cweight, const double *cmaxf, const double *task_weights,
             const double *surfaces, int warm, double *tau, double *vdot, double *cwrench,
             double *xfull, ora_tick_info *info);

/* batch loop over ora_tick (cold start per QP unless warm), returns #non-optimal */
int ora_tick_batch(ora_ctrl *c, int64_t B, const double *q, const double *v, const double *desired,
                   const double *cweight, const double *cmaxf, const double *task_weights,
                   const double *surfaces, int warm, double *tau, double *vdot, double *cwrench,
                   int32_t *status, int32_t *iters);

/* ------------------------------------------------------------------------------------------
 * generic dense-input QP solve in OSQP form (backs the L0 parity tests)
 * ---------------------------------------------------------------------------------------- */
/* timing of the OSQP restatement on k dense problems of one pattern (setup once; update + solve `reps` times each), ora_osqp.c */
int ora_qp_time(int n, int m, int k, const double *P, const double *q, const double *A, const double *l, const double *u,
                const ora_osqp_settings *st, int reps, int warm, double *us, int *iters, int *status);
int ora_qp_solve(int n, int m, const double *P, const double *q, const double *A, const double *l,
                 const double *u, const ora_osqp_settings *st, double *x, double *y,
                 ora_tick_info *info);

#ifdef __cplusplus
}
#endif
#endif
