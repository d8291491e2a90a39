// qpc_tick.h -- one MomentumBasedController tick for ONE robot state on ONE wavefront (64 lanes),
// every matrix of the QP resident in LDS.  This is the device code of the product; it is written
// as a sequence of PHASES (a block of per-lane code followed by a wave barrier) with all
// cross-lane communication through LDS, so that the very same source also compiles as a
// lane-serial CPU emulation with an LDS race detector (tests/emul/) -- the GPU box is minutes
// away, the emulator is not.
//
// The includer defines before including this file:
//   QPC_FN                    function qualifier (__device__ __forceinline__ / inline)
//   QPC_PHASE(lane) ... QPC_END_PHASE   per-lane block + barrier
//   QPC_SINCOS(x, s, c), QPC_SQRT(x), QPC_FABS(x), QPC_RCP(x) = 1/x, QPC_RSQRT(x) = 1/sqrt(x) (pivots: ~1 ulp)
// and instantiates qpc_tick<P> with P = LDS pointer type (real* on the device).
//
// Replaces, per reference function:
//   a2  copyto!(state,x) + RBD cache sweeps          -> phases "kinematics"
//   a3-a8 task_error of the 7 task types (src/tasks.jl) -> phases "assembly"
//   a9-a10 ContactPoint / objectiveterm (contacts.jl) -> phase "contacts"
//   a11 add_wrench_balance_constraint! (momentum.jl:162-193) -> wrench rows of the assembly
//   a12 addtask!/regularize! objective (momentum.jl:99-131)  -> Hv / W blocks
//   a13 Parametron.solve! -> OSQP (momentum.jl:58)    -> reduced range-space active-set solve
//   a14 checkstatus (momentum.jl:83-91)               -> status code
//   a15 solution extraction (momentum.jl:62-72)       -> vdot, contact wrenches
//   a16-a17 inverse_dynamics! + zero_floating_joint_torques! (momentum.jl:75-78) -> "rnea"
#pragma once
#include "../../include/qpc.h"
#include "qpc_program.h"

// inner-loop unrolling hint: full unroll when trip counts are compile-time constants (JIT),
// a modest factor in the generic kernel, nothing in the emulator
#ifndef QPC_UNROLL
#define QPC_UNROLL
#endif

// optional in-kernel section timers (diagnostic build only: -DQPC_STAMPS in qpc_runtime.hip)
#ifndef QPC_PROF_ACC
#define QPC_PROF_T0()
#define QPC_PROF_ACC(i)
#endif
enum { PS_KIN, PS_ASM, PS_SBUILD, PS_CHOL, PS_XINV, PS_REFAC, PS_RESID, PS_KKT, PS_UPDATE, PS_LOGIC, PS_FINAL, PS_RNEA,
       PS_K_INPUT, PS_K_JOINT, PS_K_WORLD, PS_K_SUBSP, PS_K_TWIST, PS_K_BIAS,
       PS_S_KR, PS_S_Y, PS_S_C22, PS_S_CHOL, PS_S_BACK, PS_S_TEST,
       PS_L_U0, PS_L_GUARD, PS_L_GICHECK, PS_L_GISTEP, PS_P_PROX, PS_P_BACK, PS_F_ACCEPT, PS_R_LOAD,
       PS_A_PRE, PS_A_ITEMS, PS_A_ROWS, PS_X_X, PS_X_MFMA, PS_R_SV, PS_R_BW, PS_P_QR, PS_P_RED, PS_P_POST, PS_COUNT };


// ------------------------------------------------------------------------------------------
// Arithmetic type of the tick.  `real` is double in the product's default build; the fp32 variant (BASELINE config 5: QPC_F32, built
// as its own hiprtc module with `typedef float real` and single-precision literals) runs the SAME source with the constants below at
// the fp32 rounding floor.  The includer defines `real` and `qpc_dtab` (the double tables in HBM, read as `real`).
//   QPC_HUGE / QPC_HUGE_TEST   "no candidate" sentinels of the min / arg-min reductions and their test
//   QPC_TINY                   guards of divisions by a vanishing scale
//   QPC_T_STOP                 a refinement correction at the rounding floor, relative to the iterate
//   QPC_T_CONFIRM              a confirming sweep's correction that certifies the polished iterate
//   QPC_T_PROX1                first proximal sweep: correction small enough for ONE sweep
//   QPC_T_INCONS               residual of the wrench rows that marks an inconsistent working set
//   QPC_T_QR                   smallest / largest |R_jj| below which an elimination of the presolve is refused
//   QPC_T_EXACT                smallest / largest Hessian diagonal above which the v̇ / slack blocks are used unregularised
//   QPC_EPS_MACH               unit round-off (Rotations.rotation_between's antiparallel test is 100 eps, contacts.jl:11)
#ifdef QPC_F32
#define QPC_HUGE 1e30
#define QPC_HUGE_TEST 1e29
#define QPC_TINY 1e-30
#define QPC_T_STOP 2e-6
#define QPC_T_CONFIRM 1e-4
#define QPC_T_PROX1 1e-2
#define QPC_T_INCONS 1e-3
#define QPC_T_QR 3e-2
#define QPC_T_EXACT 1e-4
#define QPC_EPS_MACH 1.1920929e-7
// The all-free block of a contact (N = 4) has the null direction n = (1, -1, 1, -1) of B'B, on which K = (H + eps I)^-1 carries 1 / eps.
// G n = 0 analytically, so in exact arithmetic that term drops out of Y = K G' and of rho = K G' lam_w; in fp32 it is 1 / eps times
// the rounding noise of G -- amplified by cond(S22) it flips signs of small rho's and doubles the working-set changes of the search
// (measured in the emulator: 6.3 -> 10.3 changes per QP at eps = 1e-6).  The fp32 tick therefore drops the term (the table stores the
// eigenvalue as an exact 0): the minimum-norm rho of a fully free contact, which is also what the eps form converges to.
#define QPC_K_NULL_TERM(lambda) ((lambda) == 0.0 ? 0.0 : 1.0)
// tolerance of the bound / multiplier test on the REFINED solution.  fp64: the acceptance tolerance (1e-7: far above the search's
// 1e-10, far below anything that matters).  fp32: the acceptance tolerance has to sit above the residual floor of an ill-conditioned
// QP (2e-3), and a working set that is wrong by that much moves tau by up to 5e-3 (measured on 32 768 config-5 states: 9 QPs beyond
// 1e-3) -- the bounds are tested at three times the search's own tolerance instead
#define QPC_VERIFY_TOL(D) ((real)3.0 * (real)(D).tol_bound)
#define QPC_ACCEPT_BAND 10.0 // primal residuals up to this multiple of tol_accept: NUMERICAL_ERROR instead of INFEASIBLE (see the acceptance test)
#else
#define QPC_K_NULL_TERM(lambda) 1.0
#define QPC_VERIFY_TOL(D) ((real)(D).tol_accept)
#define QPC_ACCEPT_BAND 1.0
#define QPC_HUGE 1e300
#define QPC_HUGE_TEST 1e299
#define QPC_TINY 1e-300
#define QPC_T_STOP 1e-13
#define QPC_T_CONFIRM 1e-9
#define QPC_T_PROX1 1e-4
#define QPC_T_INCONS 1e-6
#define QPC_T_QR 1e-4
#define QPC_T_EXACT 1e-6
#define QPC_EPS_MACH 2.220446049250313e-16
#endif

// QPC_PHASE_N(lane, n): a phase whose virtual lanes n.. have nothing to do (with two QPs per wavefront -- QPC_PACK2 -- a phase is run
// in trips of 32 lanes, and a compile-time bound saves the empty second trip; everywhere else it is QPC_PHASE)
#ifndef QPC_PHASE_N
#define QPC_PHASE_N(lane, n) QPC_PHASE(lane)
#endif
// two QPs per wavefront: the columns of T22 = S22^-1 do not ride along with the sweeps of the search (no idle lanes left for them):
// they are computed when somebody needs them (Solver::ensure_t22)
#ifdef QPC_PACK2
#define QPC_LAZY_T22 1
#else
#define QPC_LAZY_T22 0
#endif

namespace qpc {

// ------------------------------------------------------------------------------------------
// small dense helpers on registers
// ------------------------------------------------------------------------------------------
QPC_FN void cross3(const real *a, const real *b, real *c) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
QPC_FN void mv3(const real *R, const real *x, real *y) {
  real a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  real b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  real c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
QPC_FN void mtv3(const real *R, const real *x, real *y) {
  real a = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  real b = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  real c = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
QPC_FN void mm3(const real *A, const real *B, real *C) {
  real T[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = T[i];
}
// motion cross product ad(a) b
QPC_FN void crm(const real *a, const real *b, real *c) {
  real t1[3], t2[3], t3[3];
  cross3(a, b, t1); cross3(a, b + 3, t2); cross3(a + 3, b, t3);
  c[0] = t1[0]; c[1] = t1[1]; c[2] = t1[2];
  c[3] = t2[0] + t3[0]; c[4] = t2[1] + t3[1]; c[5] = t2[2] + t3[2];
}
// force cross product (w,v) x* (k,l)
QPC_FN void crf(const real *a, const real *f, real *c) {
  real t1[3], t2[3], t3[3];
  cross3(a, f, t1); cross3(a + 3, f + 3, t2); cross3(a, f + 3, t3);
  c[0] = t1[0] + t2[0]; c[1] = t1[1] + t2[1]; c[2] = t1[2] + t2[2];
  c[3] = t3[0]; c[4] = t3[1]; c[5] = t3[2];
}
// spatial inertia I = {m, h[3], J[6] = xx yy zz xy xz yz} times motion vector
QPC_FN void inertia_mul(const real *I, const real *t, real *o) {
  const real *h = I + 1, *J = I + 4, *w = t, *v = t + 3;
  real hv[3], hw[3];
  cross3(h, v, hv); cross3(h, w, hw);
  o[0] = J[0] * w[0] + J[3] * w[1] + J[4] * w[2] + hv[0];
  o[1] = J[3] * w[0] + J[1] * w[1] + J[5] * w[2] + hv[1];
  o[2] = J[4] * w[0] + J[5] * w[1] + J[2] * w[2] + hv[2];
  o[3] = I[0] * v[0] - hw[0]; o[4] = I[0] * v[1] - hw[1]; o[5] = I[0] * v[2] - hw[2];
}
// world -> frame (R,p) transform of a motion vector: w' = R^T w ; v' = R^T (v - p x w)
QPC_FN void motion_to_frame(const real *H, const real *in, real *out) {
  real pw[3], t[3];
  cross3(H + 9, in, pw);
  t[0] = in[3] - pw[0]; t[1] = in[4] - pw[1]; t[2] = in[5] - pw[2];
  mtv3(H, in, out); mtv3(H, t, out + 3);
}
template <class P> QPC_FN void ld(P p, int n, real *out) { for (int i = 0; i < n; i++) out[i] = p[i]; }
template <class P> QPC_FN void st(P p, int n, const real *in) { for (int i = 0; i < n; i++) p[i] = in[i]; }
template <class P> QPC_FN void ldH(P sm, const QpcDims &D, int body, real *H) {
  if (body < 0) { H[0] = 1; H[1] = 0; H[2] = 0; H[3] = 0; H[4] = 1; H[5] = 0; H[6] = 0; H[7] = 0; H[8] = 1; H[9] = 0; H[10] = 0; H[11] = 0; }
  else ld(sm + D.lds[L_H] + 12 * body, 12, H);
}
template <class P> QPC_FN void ld6(P sm, int off, int body, real *x) {
  if (body < 0) { for (int i = 0; i < 6; i++) x[i] = 0.0; }
  else ld(sm + off + 6 * body, 6, x);
}


// Strided LDS dot product with the loads issued in explicit batches of 8 operand pairs BEFORE the
// FMAs: hipcc keeps LDS loads in source order inside this (huge, fully inlined) kernel, so the
// naive `acc += a[i]*b[i]` loop pays one full LDS latency per term; batched, it pays one per 8.
#ifndef QPC_LAMBDA_INLINE
#define QPC_LAMBDA_INLINE
#endif
#ifndef QPC_UNROLL_FULL
#define QPC_UNROLL_FULL
#endif
// compile-time bound of the leading Schur block (the JIT defines it as the actual m1)
#ifndef QPC_M1_MAX
#define QPC_M1_MAX 64
#endif
// number of lower-triangular 16x16 tiles of the Schur complement (m <= 64 -> at most 10)
#ifndef QPC_NTILE_MAX
#define QPC_NTILE_MAX 10
#endif
// tiles of the X'[X | L21'] product: lower-triangular tiles of the m1 x m1 block plus one column of tiles for Z
#ifndef QPC_XTILE_MAX
#define QPC_XTILE_MAX 14
#endif
// compile-time bound of the tree depth (the JIT defines it as the actual depth)
#ifndef QPC_MAXD
#define QPC_MAXD QPC_GENERIC_MAXD
#endif
// full unroll only when trip counts are compile-time constants (JIT specialisation)
#ifndef QPC_UNROLL_JIT
#define QPC_UNROLL_JIT
#endif
// direct additions before the Goldfarb-Idnani phase (see phase B of the search); 0 disables
#ifndef QPC_DIRECT_ADDS
#define QPC_DIRECT_ADDS 2
#endif
// diagnostic: -DQPC_STOP_AFTER=<n> leaves the tick after section n (garbage outputs): instruction counters per section by
// differencing PMC runs (tools/pmc_probe.sh)
#ifdef QPC_STOP_AFTER
#define QPC_STOP(n) if (QPC_STOP_AFTER == (n)) return;
#else
#define QPC_STOP(n)
#endif
template <class PA, class PB> QPC_FN real dotp(PA a, int sa, PB b, int sb, int n) {
  real acc0 = 0.0, acc1 = 0.0;
  for (int i0 = 0; i0 < n; i0 += 8) {
    real x[8], y[8];
    QPC_UNROLL_FULL
    for (int u = 0; u < 8; u++) {
      const int i = i0 + u;
      const int ic = i < n ? i : 0;
      x[u] = a[ic * sa];
      y[u] = b[ic * sb];
    }
    QPC_UNROLL_FULL
    for (int u = 0; u < 8; u += 2) {
      acc0 += (i0 + u < n) ? x[u] * y[u] : 0.0;
      acc1 += (i0 + u + 1 < n) ? x[u + 1] * y[u + 1] : 0.0;
    }
  }
  return acc0 + acc1;
}

// uniform (all lanes identical) reductions over an LDS array, used OUTSIDE phases
template <class P> QPC_FN real umax(P sm, int off, int n) {
  real m = 0.0;
  for (int i = 0; i < n; i++) { real a = sm[off + i]; m = a > m ? a : m; }
  return m;
}

// in-place inverse of an SPD n x n matrix (row-major, leading dimension n, n <= NB) by ONE lane: Gauss-Jordan without
// pivoting in registers (identity padding up to the compile-time bound NB so that the sweep fully unrolls); the LDS
// copy is read once and written once
template <int NB, class P> QPC_FN void spd_inverse_small(P a, int n) {
  real r[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) r[i][j] = (i < n && j < n) ? (real)a[(i < n ? i : 0) * n + (j < n ? j : 0)] : (i == j ? 1.0 : 0.0);
#pragma unroll
  for (int k = 0; k < NB; k++) {
    const real piv = QPC_RCP(r[k][k]);
    r[k][k] = 1.0;
#pragma unroll
    for (int j = 0; j < NB; j++) r[k][j] *= piv;
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (i == k) continue;
      const real f = r[i][k];
      r[i][k] = 0.0;
#pragma unroll
      for (int j = 0; j < NB; j++) r[i][j] -= f * r[k][j];
    }
  }
#pragma unroll
  for (int i = 0; i < NB; i++)
#pragma unroll
    for (int j = 0; j < NB; j++)
      if (i < n && j < n) a[i * n + j] = r[i][j];
}
template <class P> QPC_FN void spd_inverse_inplace(P a, int n) {
  if (n <= 3) spd_inverse_small<3>(a, n); else spd_inverse_small<6>(a, n);
}


// (H_FF + eps I)^-1 of one contact's N x N rho block on its free set, in registers.
// NB >= N is a compile-time bound so that the Gauss-Jordan sweep fully unrolls.
template <int NB, class P> QPC_FN void kr_block(P sm, const QpcDims &D, int c, real eps) {
  const int N = D.N;
  real a[NB][NB];
  bool fr[NB];
#pragma unroll
  for (int i = 0; i < NB; i++) fr[i] = (i < N) && (sm[D.lds[L_WS] + c * N + (i < N ? i : 0)] == 0.0);
#pragma unroll
  for (int i = 0; i < NB; i++)
#pragma unroll
    for (int j = 0; j < NB; j++) {
      real h = (i < N && j < N) ? (real)sm[D.lds[L_HR] + c * QPC_CSH(N) + (i < N ? i : 0) * N + (j < N ? j : 0)] : 0.0;
      a[i][j] = (fr[i] && fr[j]) ? h + (i == j ? eps : 0.0) : (i == j ? 1.0 : 0.0);
    }
#pragma unroll
  for (int k = 0; k < NB; k++) {
    real piv = QPC_RCP(a[k][k]);
    a[k][k] = 1.0;
#pragma unroll
    for (int j = 0; j < NB; j++) a[k][j] *= piv;
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (i == k) continue;
      real f = a[i][k];
      a[i][k] = 0.0;
#pragma unroll
      for (int j = 0; j < NB; j++) a[i][j] -= f * a[k][j];
    }
  }
#pragma unroll
  for (int i = 0; i < NB; i++)
#pragma unroll
    for (int j = 0; j < NB; j++)
      if (i < N && j < N) sm[D.lds[L_KR] + c * QPC_CS(N) + i * N + j] = (fr[i] && fr[j]) ? a[i][j] : 0.0;
}

// x = S^-1 rhs for a symmetric 6x6 S: square-root-free LDL' in registers then forward/back substitution (with
// rhs = e_c this is column c of S^-1: 6 lanes, one column each, one phase).  A pivot
// below tau * sref[k] marks a redundant row: zero row/column in T (infinite-pivot rule).
// (LDL' instead of Cholesky: a reciprocal per pivot instead of a reciprocal square root -- 5 dependent instructions
// instead of 9 on the serial chain -- and no scaling of the off-diagonal entries by 1/L_kk.)
template <class P> QPC_FN void chol6_solve(P Sin, const real *sref, real tau, const real *rhs, real *x) {
  real L[6][6], U[6][6], invd[6], y[6]; // L unit lower triangular, U[i][k] = L[i][k] d_k
  QPC_UNROLL_FULL
  for (int i = 0; i < 6; i++)
    QPC_UNROLL_FULL
    for (int j = 0; j < 6; j++) { L[i][j] = 0.0; U[i][j] = (j <= i) ? (real)Sin[i * 6 + j] : 0.0; }
  QPC_UNROLL_FULL
  for (int k = 0; k < 6; k++) {
    real d = U[k][k];
    QPC_UNROLL_FULL
    for (int j = 0; j < k; j++) d -= U[k][j] * L[k][j];
    const bool ok = d > tau * (sref[k] > QPC_TINY ? sref[k] : QPC_TINY);
    invd[k] = ok ? QPC_RCP_SEARCH(d) : 0.0; // 1 / d_k
    QPC_UNROLL_FULL
    for (int i = k + 1; i < 6; i++) {
      real t = U[i][k];
      QPC_UNROLL_FULL
      for (int j = 0; j < k; j++) t -= U[i][j] * L[k][j];
      U[i][k] = t;
      L[i][k] = t * invd[k];
    }
  }
  QPC_UNROLL_FULL
  for (int r = 0; r < 6; r++) { // L y = rhs
    real t = rhs[r];
    QPC_UNROLL_FULL
    for (int j = 0; j < r; j++) t -= L[r][j] * y[j];
    y[r] = t;
  }
  QPC_UNROLL_FULL
  for (int r = 5; r >= 0; r--) { // L' x = D^-1 y
    real t = y[r] * invd[r];
    QPC_UNROLL_FULL
    for (int j = r + 1; j < 6; j++) t -= L[j][r] * x[j];
    x[r] = t;
  }
}

// ------------------------------------------------------------------------------------------
// per-QP global-memory records
// ------------------------------------------------------------------------------------------
struct TickIO {
  const double *q, *v, *desired, *cw, *cmax, *tw, *surf;
  double *tau, *vdot, *cwrench;
  int32_t *status, *iters;
  unsigned long long *prof; // PS_COUNT section cycle counters (diagnostic build) or nullptr
  unsigned long long *wset; // warm start (qpc_tick<P, true> only): working set of this QP's previous tick, in / out:
                            // [0] bit j = rho_j at its lower bound, [1] bit j = rho_j at its upper bound; nullptr / zeros = cold
};

// ------------------------------------------------------------------------------------------
// solver building blocks (each is a fixed sequence of phases; all lanes call them uniformly)
// ------------------------------------------------------------------------------------------
template <class P> struct Solver {
  const QpcDims &D;
  const int *IT;
  qpc_dtab DT;
  P sm;
  bool use_rho;
  QPC_LANE_SCOPE
  unsigned long long *PA, *PT; // section timers (diagnostic build)
  real last_gmax = 0.0, last_xmax = 0.0; // scales of the last base rho_solve (uniform)
  real last_dmax = 0.0;                   // max |rho - rho_prev| of the last proximal sweep (uniform)
  bool t22_stale = false;                   // QPC_LAZY_T22: L_S22I does not belong to the S22 in L_C22 yet (ensure_t22)
  bool c_valid = false;                     // L_T1 holds c = E H^-1 (-g) - d (clobbered by kkt_solve)
  bool confirmed = false;                   // eqp(from_current) stopped after ONE sweep whose correction was <= 1e-9 relative:
                                            // the residual arrays (of the iterate before that correction) bound the final ones
  // hot lookups staged in LDS (IT_HOT)
  QPC_FN int row_slack(int r) const { return (int)(real)sm[D.lds[L_HOT] + r]; }
  QPC_FN int slack_row(int s) const { return (int)(real)sm[D.lds[L_HOT] + D.m + s]; }
  QPC_FN int slack_woff(int s) const { return (int)(real)sm[D.lds[L_HOT] + D.m + D.ne + s]; }
  QPC_FN int slack_dim(int s) const { return (int)(real)sm[D.lds[L_HOT] + D.m + 2 * D.ne + s]; }
  QPC_FN int slack_s0(int s) const { return (int)(real)sm[D.lds[L_HOT] + D.m + 3 * D.ne + s]; }
  QPC_FN Solver(const QpcDims &d, const int *it, qpc_dtab dt, P s, unsigned long long *pa, unsigned long long *pt)
      : D(d), IT(it), DT(dt), sm(s), use_rho(d.nrho > 0), PA(pa), PT(pt) {}

  // sum_j G[a][j] * rho_fix[j] with rho_fix = hi on the upper-active set (batched like dotp)
  QPC_FN real dot_fix(int goff, int gstride, int j0, int n) const {
    real acc = 0.0;
    for (int i0 = 0; i0 < n; i0 += 8) {
      real g[8], w[8], h[8];
      QPC_UNROLL_FULL
      for (int u = 0; u < 8; u++) {
        const int i = i0 + u, ic = i < n ? i : 0;
        g[u] = sm[goff + ic * gstride];
        w[u] = sm[D.lds[L_WS] + j0 + ic];
        h[u] = sm[D.lds[L_HI] + j0 + ic];
      }
      QPC_UNROLL_FULL
      for (int u = 0; u < 8; u++) acc += (i0 + u < n && w[u] == 1.0) ? g[u] * h[u] : 0.0;
    }
    return acc;
  }

  // K blocks of the working set: (H_FF + eps I)^-1 per contact.  General form: Gauss-Jordan in registers, one lane per
  // contact (kr_block, ~170 dependent instructions per refactorization).  D.k_tab (N = 4, one static mu): H_c = 2 w_c B'B
  // with the same B'B for every contact, so the eigenvectors of every free set's block are static and the inverse is
  //   K_F = sum_k v_k v_k' / (2 w_c lambda_k + eps)      (DT_KTAB: per 4-bit free pattern lambda[4], V[4][4])
  // -- the same matrix to rounding (the null direction of the all-free block gets 1 / eps, as before), computed by one lane
  // per ray: its row of K from four reciprocals and a 4 x 4 product instead of the serial elimination.
  QPC_FN void kr_blocks() {
    const int N = D.N, nc = D.nc, nrho = D.nrho;
    const real eps = sm[D.lds[L_MISC] + MS_EPS];
    if (D.k_tab) {
      QPC_PHASE_N(lane, nrho) {
        const int j = lane < nrho ? lane : 0, c = j >> 2, jj = j & 3;
        int pat = 0;
        QPC_UNROLL_FULL
        for (int i = 0; i < 4; i++) pat |= (sm[D.lds[L_WS] + 4 * c + i] == 0.0) ? (1 << i) : 0;
        const int koff = D.k_classes > 1 ? (int)(real)sm[D.lds[L_MISC] + MS_KS + 8 + c] : 0; // this contact's table (its mu)
        qpc_dtab kt = DT + D.dt[DT_KTAB] + koff + pat * 20;
        const real w2 = sm[D.lds[L_MISC] + MS_KS + c];
        real t[4], k[4] = {0.0, 0.0, 0.0, 0.0};
        QPC_UNROLL_FULL
        for (int e = 0; e < 4; e++) t[e] = QPC_K_NULL_TERM(kt[e]) * kt[4 + 4 * jj + e] * QPC_RCP_SEARCH(w2 * kt[e] + eps); // v_k(jj) / (2 w lambda_k + eps)
        QPC_UNROLL_FULL
        for (int i = 0; i < 4; i++)
          QPC_UNROLL_FULL
          for (int e = 0; e < 4; e++) k[i] += t[e] * kt[4 + 4 * i + e];
        if (lane < nrho) {
          const int o = D.lds[L_KR] + QPC_CS(4) * c + 4 * jj;
          sm[o] = k[0]; sm[o + 1] = k[1]; sm[o + 2] = k[2]; sm[o + 3] = k[3];
        }
      } QPC_END_PHASE
      return;
    }
    QPC_PHASE(lane) {
      for (int c = lane; c < nc; c += 64) {
        if (N <= 4) kr_block<4>(sm, D, c, eps); else kr_block<8>(sm, D, c, eps);
      }
    } QPC_END_PHASE
  }

  // Kr (block inverse of Hrho+eps on the free set), S22 = B22 + G Kr G', T22 = S22^-1
  QPC_FN void refactor_rho() {
    const int N = D.N, nc = D.nc, nrho = D.nrho, m1 = D.m1;
    if (!use_rho && D.nw == 0) return;
    QPC_PROF_ACC(PS_LOGIC);
    if (use_rho) {
      kr_blocks();
      QPC_PHASE(lane) {
        for (int idx = lane; idx < nrho * 6; idx += 64) {
          int j = idx / 6, b = idx % 6, c = j / N, jj = j % N;
          sm[D.lds[L_Y] + b * QPC_GS(nrho) + j] = dotp(sm + D.lds[L_KR] + c * QPC_CS(N) + jj * N, 1, sm + D.lds[L_G] + b * QPC_GS(nrho) + c * N, 1, N);
        }
      } QPC_END_PHASE
    }
    QPC_PHASE(lane) {
      if (lane < 36) {
        int a = lane / 6, b = lane % 6;
        real acc = sm[D.lds[L_B22] + lane];
        if (use_rho) acc += dotp(sm + D.lds[L_G] + a * QPC_GS(nrho), 1, sm + D.lds[L_Y] + b * QPC_GS(nrho), 1, nrho);
        sm[D.lds[L_C22] + lane] = acc; // S22 = B22 + G Kr G'
      }
    } QPC_END_PHASE
    QPC_PHASE(lane) { // T22 = S22^-1 : lane c < 6 computes column c (Cholesky redone per lane, in registers)
      if (lane < 6) {
        real sref[6], x[6];
        for (int k = 0; k < 6; k++)
          sref[k] = sm[D.lds[L_SM] + (m1 + k) * D.ms + (m1 + k)] + (sm[D.lds[L_C22] + k * 6 + k] - sm[D.lds[L_B22] + k * 6 + k]);
        real ec[6];
        for (int r = 0; r < 6; r++) ec[r] = (r == lane) ? 1.0 : 0.0;
        chol6_solve(sm + D.lds[L_C22], sref, (real)D.tau_pivot, ec, x);
        for (int r = 0; r < 6; r++) sm[D.lds[L_S22I] + r * 6 + lane] = x[r];
      }
    } QPC_END_PHASE
    t22_stale = false;
    QPC_PROF_ACC(PS_REFAC);
  }

  // Row jj of K_c = (H_c,FF + eps I)^-1 in closed form (N = 4, one friction coefficient: D.k_mu2 = mu^2).  H_c = w2 B'B with
  // B'B = circ(1, a, o, a), a = 1 / (mu^2 + 1) (neighbouring rays), o = (1 - mu^2) / (mu^2 + 1) (opposite rays) -- contacts.jl:16-23
  // -- so every free set F of a contact is, seen from ray jj, one of: jj alone; jj + one neighbour; jj + the opposite ray; three
  // rays with jj at an end or in the middle; all four (circulant, eigenvalues e0 = d + 2 p + q, e1 = d - q (twice), e2 = eps:
  // B'B has the null vector (1, -1, 1, -1)).  d = w2 + eps, p = w2 a, q = w2 o; differences that cancel for small mu are
  // formed from the analytic constants (1 - a = mu^2 / (mu^2 + 1), ...).  r[k] = K[jj][(jj + k) & 3]; r = 0 when jj is pinned.
  // Same matrix as the eigen-decomposition table (DT_KTAB) to rounding, two reciprocals instead of four and no table reads.
  QPC_FN void k_row_closed(real w2, real eps, int pat, int jj, real *r) const {
    const real mu2 = (real)D.k_mu2, den = 1.0 / (mu2 + 1.0);
    const real a = den, o = (1.0 - mu2) * den, na = mu2 * den, no_ = 2.0 * mu2 * den, po = 2.0 * den;
    const real c3 = 2.0 * mu2 * den * den, c4 = mu2 * mu2 * den * den, e0c = 4.0 * den; // (1 + o) - 2 a^2 ; a^2 - o ; 1 + 2 a + o
    const int rot = ((pat >> jj) | (pat << (4 - jj))) & 15; // bit k: ray (jj + k) & 3 is free
    const bool f0 = rot & 1, f1 = rot & 2, f2 = rot & 4, f3 = rot & 8;
    const real d = w2 + eps, pp = w2 * a, qq = w2 * o;
    const real dmp = w2 * na + eps, dpp = w2 * (1.0 + a) + eps, dmq = w2 * no_ + eps, dpq = w2 * po + eps;
    const real D2p = dmp * dpp, D2q = dmq * dpq;               // d^2 - p^2, d^2 - q^2
    const real inner = w2 * w2 * c3 + eps * (w2 * (2.0 + o) + eps); // d (d + q) - 2 p^2
    const real num13 = w2 * w2 * c4 - eps * qq;                // p^2 - d q
    const int others = (f1 ? 1 : 0) + (f2 ? 1 : 0) + (f3 ? 1 : 0);
    // first reciprocal: of d | d^2 - p^2 | d^2 - q^2 | inner | e0 ; second: of d - q (three rays, four rays)
    const real x1 = others == 0 ? d : (others == 1 ? (f2 ? D2q : D2p) : (others == 2 ? inner : w2 * e0c + eps));
    const real r1 = QPC_RCP(x1), r2 = QPC_RCP(dmq);
    real k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0;
    if (others == 0) k0 = r1;
    else if (others == 1) {
      k0 = d * r1;
      if (f1) k1 = -pp * r1; else if (f3) k3 = -pp * r1; else k2 = -qq * r1;
    } else if (others == 2) {
      if (!f2) { k0 = dpq * r1; k1 = -pp * r1; k3 = k1; }        // jj between its two neighbours
      else {                                                       // jj at an end of the three
        k0 = D2p * r1 * r2; k2 = num13 * r1 * r2;
        if (f1) k1 = -pp * r1; else k3 = -pp * r1;
      }
    } else {
      const real ie0 = r1, ie1 = r2, ie2 = QPC_RCP(eps);
      k0 = 0.25 * (ie0 + 2.0 * ie1 + ie2); k1 = 0.25 * (ie0 - ie2); k2 = 0.25 * (ie0 - 2.0 * ie1 + ie2); k3 = k1;
    }
    r[0] = f0 ? k0 : 0.0; r[1] = f0 ? k1 : 0.0; r[2] = f0 ? k2 : 0.0; r[3] = f0 ? k3 : 0.0;
  }

  // ---- fused sweep of the working-set search (N = 4, K from the static table, no upper-active bound, nrho + 6 <= 64) ----
  // The sweep used to be eight phases (K | Y | S22 | LDL' by 7 lanes | G' lam_w | K q | multipliers | test), each an LDS
  // round trip of a few lanes' worth of work.  Fused: every ray lane builds ITS row of K and its six entries of
  // Y = (K G')' in one phase (ky_phase); after S22 (36 lanes, the one cross-lane sum of the sweep) EVERY ray lane runs
  // the 6 x 6 LDL' solve for lam_w itself -- the same instruction stream the 7 lanes ran, so it costs nothing extra --
  // and carries on in registers through G' lam_w, rho (quad gathers) and its multiplier: S22 -> lam_w -> rho never
  // touches LDS.  Lanes nrho .. nrho + 5 solve for the columns of T22 = S22^-1 in the same pass.
  // from_mask: the free rays come as a lane mask (bit j: rho_j free) from the ballot of the test that has just changed the
  // working set -- no LDS traffic between the test and this phase, which may then follow it across a SOFT phase end
  QPC_FN bool fused_ok() const {
#ifdef QPC_NO_FUSED_SWEEP
    return false;
#else
    return D.k_tab && D.N == 4 && D.nrho + 6 <= 64;
#endif
  }
  // QPC_LAZY_T22: T22 = S22^-1 of the current working set from the S22 the last fused sweep left in L_C22 (six lanes, one column each)
  QPC_FN void ensure_t22() {
    if (!QPC_LAZY_T22 || !t22_stale) return;
    const int m1 = D.m1;
    QPC_PHASE_N(lane, 32) {
      if (lane < 6) {
        real sref[6], x[6], ec[6];
        for (int k = 0; k < 6; k++)
          sref[k] = sm[D.lds[L_SM] + (m1 + k) * D.ms + (m1 + k)] + (sm[D.lds[L_C22] + k * 6 + k] - sm[D.lds[L_B22] + k * 6 + k]);
        for (int r = 0; r < 6; r++) ec[r] = (r == lane) ? 1.0 : 0.0;
        chol6_solve(sm + D.lds[L_C22], sref, (real)D.tau_pivot, ec, x);
        for (int r = 0; r < 6; r++) sm[D.lds[L_S22I] + r * 6 + lane] = x[r];
      }
    } QPC_END_PHASE
    t22_stale = false;
  }
  QPC_FN void ky_phase(bool from_mask = false, unsigned long long bfree = 0ull) {
    const int nrho = D.nrho;
    const real eps = sm[D.lds[L_MISC] + MS_EPS];
    QPC_PROF_ACC(PS_LOGIC);
    QPC_PHASE_N(lane, nrho) {
      const int j = lane < nrho ? lane : 0, c = j >> 2, jj = j & 3;
      int pat = 0;
      if (from_mask) pat = (int)((bfree >> (4 * c)) & 15ull);
      else {
        QPC_UNROLL_FULL
        for (int i = 0; i < 4; i++) pat |= (sm[D.lds[L_WS] + 4 * c + i] == 0.0) ? (1 << i) : 0;
      }
      const real w2 = sm[D.lds[L_MISC] + MS_KS + c];
      real k[4] = {0.0, 0.0, 0.0, 0.0};
      if (D.k_classes == 1 && (real)D.k_mu2 > 0.0) {
        real r[4];
        k_row_closed(w2, eps, pat, jj, r);
        QPC_UNROLL_FULL
        for (int i = 0; i < 4; i++) k[i] = r[(i - jj) & 3]; // K[jj][i] = r[(i - jj) & 3]
      } else {
        const int koff = D.k_classes > 1 ? (int)(real)sm[D.lds[L_MISC] + MS_KS + 8 + c] : 0; // this contact's table (its mu)
        qpc_dtab kt = DT + D.dt[DT_KTAB] + koff + pat * 20;
        real t[4];
        QPC_UNROLL_FULL
        for (int e = 0; e < 4; e++) t[e] = QPC_K_NULL_TERM(kt[e]) * kt[4 + 4 * jj + e] * QPC_RCP_SEARCH(w2 * kt[e] + eps); // v_k(jj) / (2 w lambda_k + eps)
        QPC_UNROLL_FULL
        for (int i = 0; i < 4; i++)
          QPC_UNROLL_FULL
          for (int e = 0; e < 4; e++) k[i] += t[e] * kt[4 + 4 * i + e];
      }
#if defined(QPC_EXP_DUMMY_DEP) && defined(QPC_JIT)
      { // calibration: QPC_EXP_DUMMY_DEP dependent multiply-adds ON the critical path of the phase (result feeds k[0] as + 0)
        real z = w2;
        QPC_UNROLL_FULL
        for (int u = 0; u < QPC_EXP_DUMMY_DEP; u++) z = z * 1.0000001 + eps;
        asm volatile("" : "+v"(z));
        k[0] += (z == -1.2345) ? 1.0 : 0.0;
      }
#endif
#if defined(QPC_EXP_DUMMY_IND) && defined(QPC_JIT)
      { // calibration: QPC_EXP_DUMMY_IND independent multiply-adds OFF the critical path (8 parallel chains, consumed by an empty asm)
        real z[8];
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) z[u] = w2 + (real)u;
        QPC_UNROLL_FULL
        for (int u = 0; u < QPC_EXP_DUMMY_IND; u++) z[u & 7] = z[u & 7] * 1.0000001 + eps;
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) asm volatile("" ::"v"(z[u]));
      }
#endif
      if (lane < nrho) {
        const int o = D.lds[L_KR] + QPC_CS(4) * c + 4 * jj;
        sm[o] = k[0]; sm[o + 1] = k[1]; sm[o + 2] = k[2]; sm[o + 3] = k[3];
        QPC_UNROLL_FULL
        for (int b = 0; b < 6; b++) { // Y[b][j] = sum_i K[jj][i] G[b][4 c + i]
          const int g = D.lds[L_G] + b * QPC_GS(nrho) + 4 * c;
          const real g0 = sm[g], g1 = sm[g + 1], g2 = sm[g + 2], g3 = sm[g + 3];
          sm[D.lds[L_Y] + b * QPC_GS(nrho) + j] = (k[0] * g0 + k[2] * g2) + (k[1] * g1 + k[3] * g3);
        }
      }
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_S_KR);
  }
  // ends SOFT: what follows may only read this lane's own LDS words (rho_j, u_j, ...) until the caller's QPC_PHASE_SYNC
  QPC_FN void fused_solve() {
    const int nrho = D.nrho, m1 = D.m1;
#if QPC_LAZY_T22
    QPC_PHASE_N(lane, 32) { // S22 = B22 + G Y' is symmetric: the 21 entries of its lower triangle, mirrored (one trip of a half-wave)
      if (lane < 21) {
        int a = 0, b = lane;
        while (b > a) { a++; b -= a; }
        const real v = sm[D.lds[L_B22] + a * 6 + b] + dotp(sm + D.lds[L_G] + a * QPC_GS(nrho), 1, sm + D.lds[L_Y] + b * QPC_GS(nrho), 1, nrho);
        sm[D.lds[L_C22] + a * 6 + b] = v;
        if (a != b) sm[D.lds[L_C22] + b * 6 + a] = v;
      }
    } QPC_END_PHASE
    t22_stale = true;
#else
    QPC_PHASE(lane) {
      if (lane < 36) {
        int a = lane / 6, b = lane % 6;
        sm[D.lds[L_C22] + lane] = sm[D.lds[L_B22] + lane] + dotp(sm + D.lds[L_G] + a * QPC_GS(nrho), 1, sm + D.lds[L_Y] + b * QPC_GS(nrho), 1, nrho);
      }
    } QPC_END_PHASE
#endif
    QPC_PROF_ACC(PS_S_C22);
    QPC_LANE_ARRAY(q4_, 10); // [0..3] (G' lam_w) of the quad, [4..7] rho of the quad, [8] own value in flight, [9] (G' lam_w)_j
    real rg = 0.0, rx = 0.0;
    QPC_PHASE_N(lane, QPC_LAZY_T22 ? nrho : nrho + 6) {
      real sref[6], x[6], rhs[6];
      for (int k = 0; k < 6; k++)
        sref[k] = sm[D.lds[L_SM] + (m1 + k) * D.ms + (m1 + k)] + (sm[D.lds[L_C22] + k * 6 + k] - sm[D.lds[L_B22] + k * 6 + k]);
      const int col = QPC_LAZY_T22 ? -1 : lane - nrho; // lanes nrho .. nrho + 5: column `col` of T22; every other lane: lam_w = S22^-1 u0
      for (int r = 0; r < 6; r++) rhs[r] = (col >= 0 && col < 6) ? ((r == col) ? 1.0 : 0.0) : (real)sm[D.lds[L_MISC] + MS_U0 + r];
      chol6_solve(sm + D.lds[L_C22], sref, (real)D.tau_pivot, rhs, x);
      if (lane == 0)
        for (int r = 0; r < 6; r++) sm[D.lds[L_LAM] + m1 + r] = x[r];
      if (col >= 0 && col < 6)
        for (int r = 0; r < 6; r++) sm[D.lds[L_S22I] + r * 6 + col] = x[r];
      real *q = QPC_LA(q4_, lane);
      const int j = lane < nrho ? lane : 0;
      real wj0 = 0.0, wj1 = 0.0; // (G' lam_w)_j with lam_w straight from the registers
      QPC_UNROLL_FULL
      for (int a = 0; a < 6; a += 2) {
        wj0 += sm[D.lds[L_G] + a * QPC_GS(nrho) + j] * x[a];
        wj1 += sm[D.lds[L_G] + (a + 1) * QPC_GS(nrho) + j] * x[a + 1];
      }
      q[9] = wj0 + wj1;
      q[8] = q[9];
    } QPC_END_PHASE_SOFT
    QPC_PROF_ACC(PS_S_CHOL);
    QPC_QUAD_GATHER(lane, QPC_LA(q4_, lane), QPC_LA(q4_, lane)[8])
    QPC_PHASE_N(lane, nrho) {
      real *q = QPC_LA(q4_, lane);
      real acc = 0.0;
      if (lane < nrho) { // (idle lanes must not touch K: lane 0 rewrites its row further down this soft-joined group)
        const int j = lane;
        for (int i = 0; i < 4; i++) acc += sm[D.lds[L_KR] + QPC_CS(4) * (j / 4) + 4 * (j % 4) + i] * q[i];
      }
      q[8] = acc;
    } QPC_END_PHASE_SOFT
    QPC_QUAD_GATHER(lane, QPC_LA(q4_, lane) + 4, QPC_LA(q4_, lane)[8])
    QPC_PHASE_N(lane, nrho) {
      const real *q = QPC_LA(q4_, lane);
      real g = 0.0, xm = 0.0;
      if (lane < nrho) {
        const int j = lane;
        const real rho_j = q[8];
        sm[D.lds[L_XR] + j] = rho_j;
        real gl = -q[9];
        for (int i = 0; i < 4; i++) gl += sm[D.lds[L_HR] + QPC_CSH(4) * (j / 4) + 4 * (j % 4) + i] * q[4 + i];
        const real wsj = sm[D.lds[L_WS] + j];
        sm[D.lds[L_U] + j] = (wsj == -1.0) ? gl : ((wsj == 1.0) ? -gl : 0.0);
        g = QPC_FABS(gl);
        xm = QPC_FABS(rho_j);
      }
      QPC_RED_MAX_PUT(0, rg, g); QPC_RED_MAX_PUT(1, rx, xm);
    } QPC_END_PHASE_SOFT
    last_gmax = QPC_RED_MAX_GET(0, rg); last_xmax = QPC_RED_MAX_GET(1, rx);
    QPC_PROF_ACC(PS_S_BACK);
  }

  // refactor_rho() followed by rho_solve(base = true).  Without upper-active bounds (r_r = 0, the
  // common case: rho_max is a large cap) the two fuse into 5 phases instead of 7: S22 is built
  // once and the same register Cholesky yields lam_w = S22^-1 u0 (lane 0) and the columns of
  // T22 = S22^-1 (lanes 1..6, needed by the GI directions and the final full solve).
  QPC_FN void refactor_solve(bool any_upper) {
    if (any_upper) { refactor_rho(); rho_solve(true, 0, 0.0, true); return; }
    const int N = D.N, nc = D.nc, nrho = D.nrho, m1 = D.m1;
#ifndef QPC_NO_FUSED_SWEEP
    if (fused_ok()) { ky_phase(); fused_solve(); QPC_PHASE_SYNC return; }
#endif
    QPC_PROF_ACC(PS_LOGIC);
    kr_blocks();
    QPC_PROF_ACC(PS_S_KR);
    QPC_PHASE(lane) {
      for (int idx = lane; idx < nrho * 6; idx += 64) {
        int j = idx / 6, b = idx % 6, c = j / N, jj = j % N;
        sm[D.lds[L_Y] + b * QPC_GS(nrho) + j] = dotp(sm + D.lds[L_KR] + c * QPC_CS(N) + jj * N, 1, sm + D.lds[L_G] + b * QPC_GS(nrho) + c * N, 1, N);
      }
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_S_Y);
    QPC_PHASE(lane) {
      if (lane < 36) {
        int a = lane / 6, b = lane % 6;
        sm[D.lds[L_C22] + lane] = sm[D.lds[L_B22] + lane] + dotp(sm + D.lds[L_G] + a * QPC_GS(nrho), 1, sm + D.lds[L_Y] + b * QPC_GS(nrho), 1, nrho);
      }
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_S_C22);
    QPC_PHASE(lane) {
      if (lane < 7) {
        real sref[6], x[6], rhs[6];
        for (int k = 0; k < 6; k++)
          sref[k] = sm[D.lds[L_SM] + (m1 + k) * D.ms + (m1 + k)] + (sm[D.lds[L_C22] + k * 6 + k] - sm[D.lds[L_B22] + k * 6 + k]);
        for (int r = 0; r < 6; r++) rhs[r] = lane == 0 ? (real)sm[D.lds[L_MISC] + MS_U0 + r] : ((r == lane - 1) ? 1.0 : 0.0);
        chol6_solve(sm + D.lds[L_C22], sref, (real)D.tau_pivot, rhs, x);
        for (int r = 0; r < 6; r++) {
          if (lane == 0) sm[D.lds[L_LAM] + m1 + r] = x[r];
          else sm[D.lds[L_S22I] + r * 6 + (lane - 1)] = x[r];
        }
      }
    } QPC_END_PHASE
    t22_stale = false;
    QPC_PROF_ACC(PS_S_CHOL);
    rho_back(true, false, PS_S_BACK);
  }

  // lam = S^-1 t (t in L_T1, result in `out`) through the block inverse:
  //   lam2 = T22 (t2 - Z' t1) ;  lam1 = S11I t1 - Z lam2     (S11I in the leading block of L_SM,
  //   Z = S11^-1 S12 in L_X2, T22 = (B22 + C22)^-1 in L_S22I) -- no serial chain.
  QPC_FN void solve_S(int out) {
    const int m1 = D.m1, nw = D.nw;
    ensure_t22();
    if (nw && m1 == 0) { // wrench rows only: lam_w = T22 t, one phase
      QPC_PHASE(lane) {
        if (lane < 6) sm[out + lane] = dotp(sm + D.lds[L_S22I] + lane * 6, 1, sm + D.lds[L_T1], 1, 6);
      } QPC_END_PHASE
      return;
    }
    if (nw) {
      QPC_PHASE(lane) { // u = t2 - Z' t1  (temporary in L_C22, free outside refactor_rho)
        if (lane < 6) sm[D.lds[L_C22] + lane] = sm[D.lds[L_T1] + m1 + lane] - dotp(sm + D.lds[L_X2] + lane, 6, sm + D.lds[L_T1], 1, m1);
      } QPC_END_PHASE
      QPC_PHASE(lane) {
        if (lane < 6) sm[out + m1 + lane] = dotp(sm + D.lds[L_S22I] + lane * 6, 1, sm + D.lds[L_C22], 1, 6);
      } QPC_END_PHASE
    }
    QPC_PHASE(lane) {
      for (int r = lane; r < m1; r += 64) {
        real acc = dotp(sm + D.lds[L_SM] + r * D.ms, 1, sm + D.lds[L_T1], 1, m1);
        if (nw) acc -= dotp(sm + D.lds[L_X2] + r * 6, 1, sm + out + m1, 1, 6);
        sm[out + r] = acc;
      }
    } QPC_END_PHASE
  }

  // [H+eps, E'; E, 0] [dx; dl] = [r1; r_c] on the free set. rhs in L_RV/L_RE/L_RR/L_RC,
  // result in L_DV (scaled: dx_v / sqh), L_DE, L_DR, L_DL.
  QPC_FN void kkt_solve() {
    const int N = D.N, nf = D.nf, ne = D.ne, nrho = D.nrho, m = D.m, m1 = D.m1;
    c_valid = false;
    QPC_PROF_ACC(PS_LOGIC);
    QPC_PHASE(lane) {
      if (use_rho)
        for (int j = lane; j < nrho; j += 64) {
          int c = j / N, jj = j % N;
          sm[D.lds[L_TR1] + j] = dotp(sm + D.lds[L_KR] + c * QPC_CS(N) + jj * N, 1, sm + D.lds[L_RR] + c * N, 1, N);
        }
      for (int s = lane; s < ne; s += 64) {
        int d = slack_dim(s), wr = slack_woff(s), s0 = slack_s0(s);
        sm[D.lds[L_TE1] + s] = dotp(sm + D.lds[L_WINV] + wr, 1, sm + D.lds[L_RE] + s0, 1, d);
      }
      for (int f = lane; f < nf; f += 64) sm[D.lds[L_DV] + f] = sm[D.lds[L_SQH] + f] * sm[D.lds[L_RV] + f];
    } QPC_END_PHASE
    QPC_PHASE(lane) {
      for (int r = lane; r < m; r += 64) {
        real acc = dotp(sm + D.lds[L_M] + r * D.nfs, 1, sm + D.lds[L_DV], 1, nf) - sm[D.lds[L_RC] + r];
        if (r < m1) {
          int sl = row_slack(r);
          if (sl >= 0) acc -= sm[D.lds[L_TE1] + sl];
        } else if (use_rho) {
          acc -= dotp(sm + D.lds[L_G] + (r - m1) * QPC_GS(nrho), 1, sm + D.lds[L_TR1], 1, nrho);
        }
        sm[D.lds[L_T1] + r] = acc;
      }
    } QPC_END_PHASE
    solve_S(D.lds[L_DL]);
    QPC_PHASE(lane) {
      for (int f = lane; f < nf; f += 64)
        sm[D.lds[L_DV] + f] = sm[D.lds[L_DV] + f] - dotp(sm + D.lds[L_M] + f, D.nfs, sm + D.lds[L_DL], 1, m);
      for (int s = lane; s < ne; s += 64) sm[D.lds[L_TE1] + s] = sm[D.lds[L_RE] + s] + sm[D.lds[L_DL] + slack_row(s)];
      if (use_rho)
        for (int j = lane; j < nrho; j += 64)
          sm[D.lds[L_TR2] + j] = sm[D.lds[L_RR] + j] + dotp(sm + D.lds[L_G] + j, QPC_GS(nrho), sm + D.lds[L_DL] + m1, 1, 6);
    } QPC_END_PHASE
    QPC_PHASE(lane) {
      for (int s = lane; s < ne; s += 64) {
        int d = slack_dim(s), wr = slack_woff(s), s0 = slack_s0(s);
        sm[D.lds[L_DE] + s] = dotp(sm + D.lds[L_WINV] + wr, 1, sm + D.lds[L_TE1] + s0, 1, d);
      }
      if (use_rho)
        for (int j = lane; j < nrho; j += 64) {
          int c = j / N, jj = j % N;
          sm[D.lds[L_DR] + j] = dotp(sm + D.lds[L_KR] + c * QPC_CS(N) + jj * N, 1, sm + D.lds[L_TR2] + c * N, 1, N);
        }
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_KKT);
  }

  // true (unregularised) KKT residuals of the EQP on the working set -> L_RV/L_RE/L_RR/L_RC;
  // Lagrangian gradient w.r.t. rho -> L_TR1
  QPC_FN void residuals() {
    const int N = D.N, nf = D.nf, ne = D.ne, nrho = D.nrho, m = D.m, m1 = D.m1;
    QPC_PROF_ACC(PS_LOGIC);
    QPC_PHASE(lane) {
      for (int r = lane; r < m; r += 64) {
        real acc = sm[D.lds[L_D] + r] - dotp(sm + D.lds[L_M] + r * D.nfs, 1, sm + D.lds[L_XV], 1, nf);
        if (r < m1) {
          int sl = row_slack(r);
          if (sl >= 0) acc += sm[D.lds[L_XE] + sl];
        } else if (use_rho) {
          acc += dotp(sm + D.lds[L_G] + (r - m1) * QPC_GS(nrho), 1, sm + D.lds[L_XR], 1, nrho);
        }
        sm[D.lds[L_RC] + r] = acc;
      }
      for (int f = lane; f < nf; f += 64) {
        real mtl = dotp(sm + D.lds[L_M] + f, D.nfs, sm + D.lds[L_LAM], 1, m);
        real sq = sm[D.lds[L_SQH] + f];
        sm[D.lds[L_RV] + f] = -(sm[D.lds[L_HV] + f] * sq * sm[D.lds[L_XV] + f] + sm[D.lds[L_GV] + f] + mtl / sq);
        sm[D.lds[L_DV] + f] = QPC_FABS(mtl / sq); // |E_v' lam|: scale of the stationarity test (L_DV is free here)
      }
      for (int s = lane; s < ne; s += 64) {
        int d = slack_dim(s), wr = slack_woff(s), s0 = slack_s0(s);
        real acc = dotp(sm + D.lds[L_W] + wr, 1, sm + D.lds[L_XE] + s0, 1, d);
        sm[D.lds[L_RE] + s] = -(acc - sm[D.lds[L_LAM] + slack_row(s)]);
      }
      if (use_rho)
        for (int j = lane; j < nrho; j += 64) {
          int c = j / N, jj = j % N;
          real gl = dotp(sm + D.lds[L_HR] + c * QPC_CSH(N) + jj * N, 1, sm + D.lds[L_XR] + c * N, 1, N) -
                      dotp(sm + D.lds[L_G] + j, QPC_GS(nrho), sm + D.lds[L_LAM] + m1, 1, 6);
          sm[D.lds[L_TR1] + j] = gl;
          sm[D.lds[L_WSP] + j] = gl; // kept for the acceptance test (L_TR1 is scratch of kkt_solve; L_WSP is dead after the search)
          sm[D.lds[L_RR] + j] = (sm[D.lds[L_WS] + j] == 0.0) ? -gl : 0.0;
        }
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_RESID);
  }

  // EQP of the current working set from scratch + `refine` iterative-refinement sweeps
  // from_current: keep the iterate (x, lam) and only run refinement sweeps on it
  QPC_FN int eqp(int refine, bool from_current = false) {
    const int nf = D.nf, ne = D.ne, nrho = D.nrho, m = D.m;
    if (!from_current) {
      QPC_PHASE(lane) {
        for (int f = lane; f < nf; f += 64) sm[D.lds[L_XV] + f] = 0.0;
        for (int s = lane; s < ne; s += 64) sm[D.lds[L_XE] + s] = 0.0;
        for (int r = lane; r < m; r += 64) sm[D.lds[L_LAM] + r] = 0.0;
        for (int j = lane; j < nrho; j += 64) sm[D.lds[L_XR] + j] = (sm[D.lds[L_WS] + j] == 1.0) ? (real)sm[D.lds[L_HI] + j] : 0.0;
      } QPC_END_PHASE
    }
    int it = from_current ? 1 : 0;
    real prev_dbig = QPC_HUGE;
    confirmed = false;
    for (;; it++) {
      if (it == 0) fresh_rhs(); else residuals();
      kkt_solve();
      real rbig = 0.0, rdbig = 0.0;
      QPC_PHASE(lane) {
        real big = 0.0, dbig = 0.0;
        for (int f = lane; f < nf; f += 64) {
          real dx = sm[D.lds[L_DV] + f], x = sm[D.lds[L_XV] + f] + dx;
          sm[D.lds[L_XV] + f] = x;
          real sq = sm[D.lds[L_SQH] + f];
          big = QPC_FABS(x * sq) > big ? QPC_FABS(x * sq) : big;
          dbig = QPC_FABS(dx * sq) > dbig ? QPC_FABS(dx * sq) : dbig;
        }
        for (int s = lane; s < ne; s += 64) {
          real dx = sm[D.lds[L_DE] + s], x = sm[D.lds[L_XE] + s] + dx;
          sm[D.lds[L_XE] + s] = x;
          big = QPC_FABS(x) > big ? QPC_FABS(x) : big;
          dbig = QPC_FABS(dx) > dbig ? QPC_FABS(dx) : dbig;
        }
        for (int j = lane; j < nrho; j += 64) {
          real dx = sm[D.lds[L_DR] + j], x = sm[D.lds[L_XR] + j] + dx;
          sm[D.lds[L_XR] + j] = x;
          big = QPC_FABS(x) > big ? QPC_FABS(x) : big;
          dbig = QPC_FABS(dx) > dbig ? QPC_FABS(dx) : dbig;
        }
        for (int r = lane; r < m; r += 64) sm[D.lds[L_LAM] + r] = sm[D.lds[L_LAM] + r] + sm[D.lds[L_DL] + r];
        if (refine > 0) { QPC_RED_MAX_PUT(0, rbig, big); QPC_RED_MAX_PUT(1, rdbig, dbig); }
      } QPC_END_PHASE
      QPC_PROF_ACC(PS_UPDATE);
      if (refine <= 0) break;
      rbig = QPC_RED_MAX_GET(0, rbig); rdbig = QPC_RED_MAX_GET(1, rdbig);
      // Stop when: converged; the correction sits on the rounding floor of this KKT system (small and
      // no longer shrinking: further sweeps only re-apply noise); the NEXT correction, predicted from
      // the observed contraction, is already below tolerance; or there is no progress at all.  A
      // slowly but steadily contracting sequence (Hessian eigenvalues comparable to eps, e.g.
      // friction coefficients near zero) is allowed up to 5x the nominal sweep count.
      if (it > 0) {
        const real tol = QPC_T_STOP * (1.0 + rbig);
        if (!(rdbig > tol)) break;
        // first correction of an already polished iterate: its size IS the error estimate of that
        // iterate, and the corrected one is better by the contraction factor (<= 1e-5)
        if (from_current && it == 1 && !(rdbig > QPC_T_CONFIRM * (1.0 + rbig))) { confirmed = true; break; }
        if (rdbig > 0.25 * prev_dbig && rdbig < QPC_T_CONFIRM * (1.0 + rbig)) break;
        if (rdbig > 0.95 * prev_dbig) break;
        if (it >= 2 && rdbig * (rdbig / prev_dbig) < tol) break;
        if (it >= refine && !(rdbig > 0.25 * prev_dbig)) break; // nominal budget spent and not in the slow regime
        if (it >= 5 * refine) break;
      }
      prev_dbig = rdbig;
    }
    return it;
  }

  // rho_F = K (r_r + G' lam_w) (+ rho_fix) and, in base mode, the bound multipliers and scales
  QPC_FN void rho_back(bool base, bool need_kr, int section = PS_KKT, bool prox = false) {
    const int N = D.N, nrho = D.nrho, m1 = D.m1;
    const int outr = base ? D.lds[L_XR] : D.lds[L_DR], outl = (base ? D.lds[L_LAM] : D.lds[L_DL]) + m1;
    real rg = 0.0, rx = 0.0, rd = 0.0;
    if (N == 4 && nrho <= 64) {
      // N = 4: the four rays of a contact sit in the four lanes of a quad, so each lane computes ITS (G' lam_w)_j and
      // ITS rho_j only and takes the other three from its quad with DPP quad_perm moves -- a quarter of the dot products
      // and LDS reads of the general form below, same operations in the same order (bit-identical results)
      QPC_LANE_ARRAY(q4_, 10); // [0..3] w of the quad, [4..7] rho of the quad, [8] own value in flight, [9] (G' lam_w)_j
      QPC_PHASE(lane) {
        real *q = QPC_LA(q4_, lane);
        const int j = lane < nrho ? lane : 0;
        real wj = dotp(sm + D.lds[L_G] + j, QPC_GS(nrho), sm + outl, 1, 6);
        q[9] = wj;
        if (need_kr) wj += sm[D.lds[L_RR] + j];
        q[8] = wj;
      } QPC_END_PHASE
      QPC_QUAD_GATHER(lane, QPC_LA(q4_, lane), QPC_LA(q4_, lane)[8])
      QPC_PHASE(lane) {
        real *q = QPC_LA(q4_, lane);
        const int j = lane < nrho ? lane : 0;
        real acc = 0.0;
        for (int i = 0; i < 4; i++) acc += sm[D.lds[L_KR] + QPC_CS(4) * (j / 4) + 4 * (j % 4) + i] * q[i]; // row jj of K_c: KR[c][jj][i] at 16 c + 4 jj + i
        if (base && sm[D.lds[L_WS] + j] == 1.0) acc += sm[D.lds[L_HI] + j];
        q[8] = acc;
      } QPC_END_PHASE
      QPC_QUAD_GATHER(lane, QPC_LA(q4_, lane) + 4, QPC_LA(q4_, lane)[8])
      QPC_PHASE(lane) {
        const real *q = QPC_LA(q4_, lane);
        real g = 0.0, xm = 0.0, dm = 0.0;
        if (lane < nrho) {
          const int j = lane;
          const real rho_j = q[8];
          if (prox) dm = QPC_FABS(rho_j - (real)sm[outr + j]);
          sm[outr + j] = rho_j;
          if (base) {
            real gl = -q[9];
            for (int i = 0; i < 4; i++) gl += sm[D.lds[L_HR] + QPC_CSH(4) * (j / 4) + 4 * (j % 4) + i] * q[4 + i];
            const real wsj = sm[D.lds[L_WS] + j];
            sm[D.lds[L_U] + j] = (wsj == -1.0) ? gl : ((wsj == 1.0) ? -gl : 0.0);
            g = QPC_FABS(gl);
            xm = QPC_FABS(rho_j);
          }
        }
        if (base) { QPC_RED_MAX_PUT(0, rg, g); QPC_RED_MAX_PUT(1, rx, xm); }
        if (prox) { QPC_RED_MAX_PUT(2, rd, dm); }
      } QPC_END_PHASE
      if (base) { last_gmax = QPC_RED_MAX_GET(0, rg); last_xmax = QPC_RED_MAX_GET(1, rx); }
      if (prox) last_dmax = QPC_RED_MAX_GET(2, rd);
      QPC_PROF_ACC(section);
      return;
    }
    QPC_PHASE(lane) { // rho_F = K (r_r + G' lam_w) (+ rho_fix); each lane rebuilds its contact's N rho's,
                      // and in base mode also its multiplier u_j = +-(H_rho rho - G' lam_w)_j
      real g = 0.0, xm = 0.0, dm = 0.0;
      for (int j = lane; j < nrho; j += 64) {
        int c = j / N, jj = j % N;
        real w[8], rho_c[8];
        for (int i = 0; i < N; i++) {
          w[i] = dotp(sm + D.lds[L_G] + c * N + i, QPC_GS(nrho), sm + outl, 1, 6);
          if (i == jj) g = w[i]; // (G' lam_w)_j, reused below
          if (need_kr) w[i] += sm[D.lds[L_RR] + c * N + i];
        }
        for (int i2 = 0; i2 < N; i2++) {
          real acc = 0.0;
          for (int i = 0; i < N; i++) acc += sm[D.lds[L_KR] + c * QPC_CS(N) + i2 * N + i] * w[i];
          if (base && sm[D.lds[L_WS] + c * N + i2] == 1.0) acc += sm[D.lds[L_HI] + c * N + i2];
          rho_c[i2] = acc;
        }
        if (prox) { const real dl_ = QPC_FABS(rho_c[jj] - (real)sm[outr + j]); dm = dl_ > dm ? dl_ : dm; }
        sm[outr + j] = rho_c[jj];
        if (base) {
          real gl = -g;
          for (int i = 0; i < N; i++) gl += sm[D.lds[L_HR] + c * QPC_CSH(N) + jj * N + i] * rho_c[i];
          real wsj = sm[D.lds[L_WS] + j];
          sm[D.lds[L_U] + j] = (wsj == -1.0) ? gl : ((wsj == 1.0) ? -gl : 0.0);
          g = QPC_FABS(gl);
          xm = QPC_FABS(rho_c[jj]);
        }
      }
      if (base) { QPC_RED_MAX_PUT(0, rg, g); QPC_RED_MAX_PUT(1, rx, xm); }
      if (prox) { QPC_RED_MAX_PUT(2, rd, dm); }
    } QPC_END_PHASE
    if (base) { last_gmax = QPC_RED_MAX_GET(0, rg); last_xmax = QPC_RED_MAX_GET(1, rx); }
    if (prox) last_dmax = QPC_RED_MAX_GET(2, rd);
    QPC_PROF_ACC(section);
  }

  // ---- wrench-space solves for the active-set SEARCH --------------------------------------
  // rho enters the equality block only through the 6 wrench-balance rows, so for a given working
  // set   lam_w = T22 (u0 - G (K r_r + rho_fix)),   rho_F = K (r_r + G' lam_w)   with the 6-vector
  // u0 = c2 - Z' c1 fixed for the whole tick (c = E H^-1 (-g) - d, no rho terms).  Task-row
  // multipliers, v̇ and slacks are NOT needed to decide the working set; they are recovered by the
  // refined full solve at the end.
  //   base = true : EQP of the working set (r_r = -(H_rho rho_fix)_F), result -> L_XR, L_LAM[m1..]
  //   base = false: Goldfarb-Idnani direction for r_r = nsign * e_p,   result -> L_DR, L_DL[m1..]
  QPC_FN void rho_solve(bool base, int p, real nsign, bool any_upper, bool prox = false, int section = PS_KKT) {
    const int N = D.N, nrho = D.nrho, m1 = D.m1;
    const int outr = base ? D.lds[L_XR] : D.lds[L_DR], outl = (base ? D.lds[L_LAM] : D.lds[L_DL]) + m1;
    QPC_PROF_ACC(PS_LOGIC);
    ensure_t22();
    const bool need_kr = !base || any_upper || prox;
    const real eps_prox = prox ? (real)sm[D.lds[L_MISC] + MS_EPS] : 0.0;
    if (need_kr && base && N == 4 && nrho <= 64) {
      // quad form (see rho_back): each lane its own r_r(j), the other three of the contact through DPP
      QPC_LANE_ARRAY(r4_, 5); // [0..3] r_r of the quad, [4] own
      QPC_PHASE(lane) {
        const int j = lane < nrho ? lane : 0;
        const real wsj = sm[D.lds[L_WS] + j];
        real rr = (wsj == 0.0) ? -dot_fix(D.lds[L_HR] + QPC_CSH(4) * (j / 4) + 4 * (j % 4), 1, 4 * (j / 4), 4) : 0.0;
        // proximal sweep: the eps of K = (H_FF + eps I)^-1 is compensated by eps * rho_prev
        if (prox && wsj == 0.0) rr += eps_prox * sm[D.lds[L_XR] + j];
        if (lane < nrho) sm[D.lds[L_RR] + j] = rr;
        QPC_LA(r4_, lane)[4] = rr;
      } QPC_END_PHASE
      QPC_QUAD_GATHER(lane, QPC_LA(r4_, lane), QPC_LA(r4_, lane)[4])
      QPC_PHASE(lane) {
        if (lane < nrho) {
          const int j = lane;
          const real *r4 = QPC_LA(r4_, lane);
          const real fix = (sm[D.lds[L_WS] + j] == 1.0) ? (real)sm[D.lds[L_HI] + j] : 0.0;
          real acc = 0.0;
          for (int i = 0; i < 4; i++) acc += sm[D.lds[L_KR] + QPC_CS(4) * (j / 4) + 4 * (j % 4) + i] * r4[i];
          sm[D.lds[L_TR1] + j] = acc + fix;
        }
      } QPC_END_PHASE
    } else if (need_kr) {
      QPC_PHASE(lane) { // TR1 = K r_r + rho_fix ; RR kept for the back-substitution
        for (int j = lane; j < nrho; j += 64) {
          int c = j / N, jj = j % N;
          real acc = 0.0, fix = 0.0;
          if (base) {
            fix = (sm[D.lds[L_WS] + j] == 1.0) ? (real)sm[D.lds[L_HI] + j] : 0.0;
            // r_r(i) = -(H rho_fix)_i on free i ;  acc = sum_i K[jj][i] r_r(i)
            for (int i = 0; i < N; i++) {
              real rr = (sm[D.lds[L_WS] + c * N + i] == 0.0) ? -dot_fix(D.lds[L_HR] + c * QPC_CSH(N) + i * N, 1, c * N, N) : 0.0;
              // proximal sweep: the eps of K = (H_FF + eps I)^-1 is compensated by eps * rho_prev
              if (prox && sm[D.lds[L_WS] + c * N + i] == 0.0) rr += eps_prox * sm[D.lds[L_XR] + c * N + i];
              acc += sm[D.lds[L_KR] + c * QPC_CS(N) + jj * N + i] * rr;
              if (i == jj) sm[D.lds[L_RR] + j] = rr;
            }
          } else {
            acc = (p / N == c) ? nsign * sm[D.lds[L_KR] + c * QPC_CS(N) + jj * N + (p % N)] : 0.0;
            sm[D.lds[L_RR] + j] = (j == p) ? nsign : 0.0;
          }
          sm[D.lds[L_TR1] + j] = acc + fix;
        }
      } QPC_END_PHASE
    }
    if (need_kr) {
      QPC_PHASE(lane) { // u = u0 - G (K r_r + rho_fix)   (temporary in L_C22[0..5])
        if (lane < 6) {
          real u = base ? (real)sm[D.lds[L_MISC] + MS_U0 + lane] : 0.0;
          u -= dotp(sm + D.lds[L_G] + lane * QPC_GS(nrho), 1, sm + D.lds[L_TR1], 1, nrho);
          sm[D.lds[L_C22] + lane] = u;
        }
      } QPC_END_PHASE
    }
    QPC_PHASE(lane) {
      if (lane < 6) sm[outl + lane] = dotp(sm + D.lds[L_S22I] + lane * 6, 1, sm + (need_kr ? D.lds[L_C22] : D.lds[L_MISC] + MS_U0), 1, 6);
    } QPC_END_PHASE
    rho_back(base, need_kr, section, prox);
    QPC_PROF_ACC(section);
  }

  // c = E H^-1 (-g) - d -> L_T1 (all rows), with -sqh .* g as a by-product in L_XV
  QPC_FN void compute_c() {
    const int nf = D.nf, m = D.m;
    QPC_PHASE(lane) {
      for (int f = lane; f < nf; f += 64) sm[D.lds[L_XV] + f] = -sm[D.lds[L_SQH] + f] * sm[D.lds[L_GV] + f];
    } QPC_END_PHASE
    QPC_PHASE(lane) {
      for (int r = lane; r < m; r += 64)
        sm[D.lds[L_T1] + r] = dotp(sm + D.lds[L_M] + r * D.nfs, 1, sm + D.lds[L_XV], 1, nf) - sm[D.lds[L_D] + r];
    } QPC_END_PHASE
    c_valid = true;
  }

  // Final solve of the EQP on the converged working set when the v̇ and slack blocks of the Hessian
  // are used UNREGULARISED (exact_ve): the only eps left is the one inside K = (H_rho,FF + eps I)^-1,
  // and rho enters the equality block through the 6 wrench rows alone.  So the eps bias is removed by
  // proximal sweeps in wrench space (each one a rho_solve with r_r += eps rho_prev: 4 short phases,
  // contraction eps / lambda_min(H_rho) ~ 1e-6 per sweep), and v̇, slacks and task-row multipliers
  // follow from ONE back-substitution:  lam_1 = S11^-1 c_1 - Z lam_w ,  x_v = H_v^-1 (-g - E_v' lam),
  // e = W^-1 lam_slack -- instead of 3-4 refinement sweeps of the full KKT system.  The caller checks
  // the true residuals afterwards and falls back to eqp(from_current) if they are not at rounding level.
  QPC_FN int polish_exact(bool any_upper) {
    const int nf = D.nf, ne = D.ne, m1 = D.m1, nw = D.nw, m = D.m;
    int k = 0;
    if (!c_valid) compute_c();
    if (use_rho) {
      real prev = QPC_HUGE;
      for (k = 1; k <= 10; k++) {
        rho_solve(true, 0, 0.0, any_upper, true, PS_P_PROX);
        const real tol = QPC_T_STOP * (1.0 + last_xmax);
        if (!(last_dmax > tol)) break;
        // the first sweep's correction IS the eps bias of the search's solution, i.e. the contraction factor times |rho|:
        // at <= 1e-4 the iterate is already 1e-8 accurate and the confirming sweep of the full system (contraction again)
        // finishes the job -- no second proximal sweep
#ifndef QPC_EXP_PROX2
        if (k == 1 && !(last_dmax > QPC_T_PROX1 * (1.0 + last_xmax))) break;
#endif
        if (k >= 2 && last_dmax * (last_dmax / prev) < tol) break;
        if (last_dmax > 0.95 * prev) break; // rounding floor / no progress
        prev = last_dmax;
      }
      if (m1 > 0) {
      QPC_PHASE(lane) { // lam_1 = S11^-1 c_1 - Z lam_w
        for (int r = lane; r < m1; r += 64) {
          real acc = dotp(sm + D.lds[L_SM] + r * D.ms, 1, sm + D.lds[L_T1], 1, m1);
          if (nw) acc -= dotp(sm + D.lds[L_X2] + r * 6, 1, sm + D.lds[L_LAM] + m1, 1, 6);
          sm[D.lds[L_LAM] + r] = acc;
        }
      } QPC_END_PHASE
      }
    } else {
      solve_S(D.lds[L_LAM]); // lam = S^-1 c (no rho: S22 = B22)
    }
    QPC_PHASE(lane) { // x_v (scaled) = -sqh g - M' lam ; e = W^-1 lam_slack
      for (int f = lane; f < nf; f += 64)
        sm[D.lds[L_XV] + f] = -sm[D.lds[L_SQH] + f] * sm[D.lds[L_GV] + f] - dotp(sm + D.lds[L_M] + f, D.nfs, sm + D.lds[L_LAM], 1, m);
      for (int s = lane; s < ne; s += 64) {
        int d = slack_dim(s), wr = slack_woff(s), s0 = slack_s0(s);
        real acc = 0.0;
        for (int i = 0; i < d; i++) acc += sm[D.lds[L_WINV] + wr + i] * sm[D.lds[L_LAM] + slack_row(s0 + i)];
        sm[D.lds[L_XE] + s] = acc;
      }
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_P_BACK);
    return k;
  }

  // EQP of the current working set solved from x = 0: the residual IS the right-hand side, so no
  // mat-vec pass is needed; one KKT solve, no refinement (used inside the active-set search)
  QPC_FN void fresh_rhs() {
    const int nf = D.nf, ne = D.ne, nrho = D.nrho, m = D.m, m1 = D.m1, N = D.N;
    QPC_PROF_ACC(PS_LOGIC);
    QPC_PHASE(lane) {
      for (int f = lane; f < nf; f += 64) sm[D.lds[L_RV] + f] = -sm[D.lds[L_GV] + f];
      for (int s = lane; s < ne; s += 64) sm[D.lds[L_RE] + s] = 0.0;
      for (int r = lane; r < m; r += 64) {
        real acc = sm[D.lds[L_D] + r];
        if (r >= m1 && use_rho) acc += dot_fix(D.lds[L_G] + (r - m1) * QPC_GS(nrho), 1, 0, nrho);
        sm[D.lds[L_RC] + r] = acc;
      }
      for (int j = lane; j < nrho; j += 64) {
        int c = j / N, jj = j % N;
        real acc = 0.0;
        if (sm[D.lds[L_WS] + j] == 0.0) acc = -dot_fix(D.lds[L_HR] + c * QPC_CSH(N) + jj * N, 1, c * N, N);
        sm[D.lds[L_RR] + j] = acc;
      }
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_RESID);
  }
  QPC_FN void eqp_fresh() {
    const int nf = D.nf, ne = D.ne, nrho = D.nrho, m = D.m;
    fresh_rhs();
    kkt_solve();
    QPC_PHASE(lane) {
      for (int f = lane; f < nf; f += 64) sm[D.lds[L_XV] + f] = sm[D.lds[L_DV] + f];
      for (int s = lane; s < ne; s += 64) sm[D.lds[L_XE] + s] = sm[D.lds[L_DE] + s];
      for (int r = lane; r < m; r += 64) sm[D.lds[L_LAM] + r] = sm[D.lds[L_DL] + r];
      for (int j = lane; j < nrho; j += 64)
        sm[D.lds[L_XR] + j] = sm[D.lds[L_DR] + j] + ((sm[D.lds[L_WS] + j] == 1.0) ? (real)sm[D.lds[L_HI] + j] : 0.0);
    } QPC_END_PHASE
    QPC_PROF_ACC(PS_UPDATE);
  }
};

// rotation_vector(R) (angle * axis), the error term of the orientation PD law (standing.jl:74-76).
// Branch on the angle like a robust log map: sin-based formula away from pi, diagonal-based near pi.
QPC_FN void rotation_vector(const real *R, real *rv) {
  const real ax = R[7] - R[5], ay = R[2] - R[6], az = R[3] - R[1]; // 2 sin(th) * axis
  const real s2 = QPC_SQRT(ax * ax + ay * ay + az * az);          // 2 sin(th)
  const real c2 = R[0] + R[4] + R[8] - 1.0;                      // 2 cos(th)
  const real th = QPC_ATAN2(s2, c2);
  if (s2 < 1e-12 && c2 > 0.0) { rv[0] = 0.5 * ax; rv[1] = 0.5 * ay; rv[2] = 0.5 * az; return; } // th -> 0
  if (s2 > 1e-6) { const real k = th / s2; rv[0] = k * ax; rv[1] = k * ay; rv[2] = k * az; return; }
  // th near pi: axis from the diagonal, signs from the off-diagonal sums
  const real cth = 0.5 * c2;
  real x = QPC_SQRT((R[0] - cth) / (1.0 - cth) > 0.0 ? (R[0] - cth) / (1.0 - cth) : 0.0);
  real y = QPC_SQRT((R[4] - cth) / (1.0 - cth) > 0.0 ? (R[4] - cth) / (1.0 - cth) : 0.0);
  real z = QPC_SQRT((R[8] - cth) / (1.0 - cth) > 0.0 ? (R[8] - cth) / (1.0 - cth) : 0.0);
  if (x >= y && x >= z) { if (R[1] + R[3] < 0.0) y = -y; if (R[2] + R[6] < 0.0) z = -z; if (ax < 0.0) { x = -x; y = -y; z = -z; } }
  else if (y >= z) { if (R[1] + R[3] < 0.0) x = -x; if (R[5] + R[7] < 0.0) z = -z; if (ay < 0.0) { x = -x; y = -y; z = -z; } }
  else { if (R[2] + R[6] < 0.0) x = -x; if (R[5] + R[7] < 0.0) y = -y; if (az < 0.0) { x = -x; y = -y; z = -z; } }
  rv[0] = th * x; rv[1] = th * y; rv[2] = th * z;
}

// desired entry i from the caller (setdesired!) or from a law that only needs (q, v):
// constant, joint-space PD kp (ref - q) - kd v (standing.jl:80-86).  COM / orientation laws need
// the kinematics and are filled in later (desired_late); until then the entry is 0.
QPC_FN real desired_early(const QpcDims &D, const int *IT, qpc_dtab DT, const TickIO &io, int i) {
  if (D.n_des_inputs == D.n_desired) return io.desired[i];
  const int *li = IT + D.it[IT_LAW] + 3 * i;
  qpc_dtab ld_ = DT + D.dt[DT_LAW] + 3 * i;
  const int kind = li[0];
  if (kind == QPC_LAW_INPUT) return io.desired[i];
  if (kind == QPC_LAW_CONSTANT) return ld_[2];
  if (kind == QPC_LAW_JOINT_PD) return ld_[0] * (ld_[2] - io.q[li[1]]) - ld_[1] * io.v[li[2]];
  if (((D.law_mask >> QPC_LAW_ORIENTATION_PD_ROOT) & 1) && kind == QPC_LAW_ORIENTATION_PD_ROOT) {
    // orientation PD of the floating body on the world (standing.jl:72-77): R = Rf R(quaternion), omega_body = v[v0 .. v0 + 2]
    const int body = li[1], a = li[2], nb = D.nb;
    const int q0 = IT[D.it[IT_QOFF] + body], v0 = IT[D.it[IT_VOFF] + body];
    const real w = io.q[q0], x = io.q[q0 + 1], y = io.q[q0 + 2], z = io.q[q0 + 3];
    const real Rj[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                          2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                          2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    real Rf[9], H[9], rv[3];
    for (int k = 0; k < 9; k++) Rf[k] = DT[D.dt[DT_RF] + k * nb + body];
    mm3(Rf, Rj, H);
    rotation_vector(H, rv);
    return -ld_[0] * rv[a] - ld_[1] * io.v[v0 + a];
  }
  return 0.0;
}


// SE3PDController (src/lowlevel/se3pdcontroller.jl:13-18) with its SE3Trajectory (src/trajectories/se3.jl:8-27) made of
// two Interpolated trajectories (src/trajectories/interpolated.jl:20-83), evaluated for ONE state:
//   rec = the task's trajectory record (include/qpc.h QPC_LAW_SE3_PD; rec[0] is entry 6 of the slice: t)
//   Hb / Ha = world pose of body / base (R row-major [0..8], p [9..11]), Tb / Ta = their world twists about the world origin
//   gains: ka, da (angular), kl, dl (linear), all expressed in the body frame (test/controller.jl:13)
// out[6] = Tdref + pd(gains, H, Href, T, Tref): desired spatial acceleration of body w.r.t. base in the body frame.
// The SE(3) PD law itself lives in the un-vendored RigidBodyDynamics.PDControl; restated as in api.SE3PDController
// (real-geodesic law: e = inv(Href) H, edot = T - Ad(inv(e)) Tref).
//   t = time in the segment's own clock, rec = the segment: rec[1] x0, rec[2] xf, rec[3..8] c, rec[9..16] quaternions, rec[17..22] p
//   (rec[0] is not read: for a single-segment record it is the slot of t, for a piece of a Piecewise record the word before the piece)
template <class PR>
QPC_FN void se3_pd_law(real t, PR rec, const real *Hb, const real *Ha, const real *Tb, const real *Ta, real ka, real da,
                       real kl, real dl, real *out) {
  const real x0 = rec[1], xf = rec[2];
  // theta = clamp((t - x0) / (xf - x0), 0, 1) and its derivative (interpolated.jl:29-40)
  const real dx = xf - x0;
  real th = (t - x0) / dx, dth = 1.0 / dx;
  if (th <= 0.0) { th = 0.0; dth = 0.0; } else if (th >= 1.0) { th = 1.0; dth = 0.0; }
  // alpha(theta) = sum c_i theta^i and two derivatives (PolyDerivEvaluator; identity = 0,1,0,...)
  real c[6];
  for (int i = 0; i < 6; i++) c[i] = rec[3 + i];
  const real al = c[0] + th * (c[1] + th * (c[2] + th * (c[3] + th * (c[4] + th * c[5]))));
  const real a1 = c[1] + th * (2.0 * c[2] + th * (3.0 * c[3] + th * (4.0 * c[4] + th * 5.0 * c[5])));
  const real a2 = 2.0 * c[2] + th * (6.0 * c[3] + th * (12.0 * c[4] + th * 20.0 * c[5]));
  const real ax = a1 * dth, axx = a2 * dth * dth; // chain rule: higher derivatives of theta vanish (interpolated.jl:47-51)
  // linear part: p = p0 + alpha (pf - p0) (interpolated.jl:60-65)
  real p[3], pd[3], pdd[3];
  for (int i = 0; i < 3; i++) { const real d = rec[20 + i] - rec[17 + i]; p[i] = rec[17 + i] + al * d; pd[i] = d * ax; pdd[i] = d * axx; }
  // angular part: y = y0 * Quat(AngleAxis(alpha * angle, axis)), dy/dalpha = axis * angle (interpolated.jl:72-83)
  const real w0 = rec[9], x0q = rec[10], y0q = rec[11], z0q = rec[12], w1 = rec[13], x1 = rec[14], y1 = rec[15], z1 = rec[16];
  // qd = inv(y0) * yf
  real dw = w0 * w1 + x0q * x1 + y0q * y1 + z0q * z1;
  real dv[3] = {w0 * x1 - x0q * w1 - y0q * z1 + z0q * y1, w0 * y1 + x0q * z1 - y0q * w1 - z0q * x1, w0 * z1 - x0q * y1 + y0q * x1 - z0q * w1};
  // no hemisphere flip: Rotations 0.13.0 (Manifest.toml:136-140) converts with theta = 2 atan(|v|, w), so a relative
  // quaternion with w < 0 gives an angle in (pi, 2 pi] about v / |v| -- the LONG arc, same end point
  const real sn = QPC_SQRT(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
  const real angle = 2.0 * QPC_ATAN2(sn, dw);
  real axis[3] = {1.0, 0.0, 0.0};
  if (sn > QPC_TINY) { axis[0] = dv[0] / sn; axis[1] = dv[1] / sn; axis[2] = dv[2] / sn; }
  real sh, ch;
  QPC_SINCOS(0.5 * al * angle, sh, ch);
  const real iw = ch, ix = sh * axis[0], iy = sh * axis[1], iz = sh * axis[2];
  const real qw = w0 * iw - x0q * ix - y0q * iy - z0q * iz, qx = w0 * ix + x0q * iw + y0q * iz - z0q * iy,
               qy = w0 * iy - x0q * iz + y0q * iw + z0q * ix, qz = w0 * iz + x0q * iy - y0q * ix + z0q * iw;
  const real Rr[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                        2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                        2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
  const real wbase[3] = {axis[0] * angle * ax, axis[1] * angle * ax, axis[2] * angle * ax};
  const real wdbase[3] = {axis[0] * angle * axx, axis[1] * angle * axx, axis[2] * angle * axx};
  // SE3Trajectory (se3.jl:8-27): everything re-expressed in the (desired) body frame
  real wr[3], nur[3], wdr[3], nudr[3], cr[3];
  mtv3(Rr, wbase, wr); mtv3(Rr, pd, nur); mtv3(Rr, wdbase, wdr); mtv3(Rr, pdd, nudr);
  cross3(wr, nur, cr);
  nudr[0] += cr[0]; nudr[1] += cr[1]; nudr[2] += cr[2];
  // state: H = relative_transform(body -> base), T = relative twist in the body frame (se3pdcontroller.jl:15-16)
  real R[9], Rat[9], dpw[3], pr[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rat[3 * i + j] = Ha[3 * j + i];
  mm3(Rat, Hb, R);
  dpw[0] = Hb[9] - Ha[9]; dpw[1] = Hb[10] - Ha[10]; dpw[2] = Hb[11] - Ha[11];
  mtv3(Ha, dpw, pr);
  real wrel[3] = {Tb[0] - Ta[0], Tb[1] - Ta[1], Tb[2] - Ta[2]}, vrel[3] = {Tb[3] - Ta[3], Tb[4] - Ta[4], Tb[5] - Ta[5]};
  real wxp[3], Tang[3], Tlin[3];
  cross3(wrel, Hb + 9, wxp);
  vrel[0] += wxp[0]; vrel[1] += wxp[1]; vrel[2] += wxp[2];
  mtv3(Hb, wrel, Tang); mtv3(Hb, vrel, Tlin);
  // e = inv(Href) * H ; Tref seen from the actual body frame: Ad(inv(e))
  real Re[9], Rrt[9], dpe[3] = {pr[0] - p[0], pr[1] - p[1], pr[2] - p[2]}, pe[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rrt[3 * i + j] = Rr[3 * j + i];
  mm3(Rrt, R, Re);
  mtv3(Rr, dpe, pe);
  real Trang[3], tl[3], Trlin[3];
  mtv3(Re, wr, Trang);
  cross3(wr, pe, tl);
  tl[0] += nur[0]; tl[1] += nur[1]; tl[2] += nur[2];
  mtv3(Re, tl, Trlin);
  real rv[3], Rtp[3];
  rotation_vector(Re, rv);
  mtv3(Re, pe, Rtp);
  for (int i = 0; i < 3; i++) {
    out[i] = wdr[i] - ka * rv[i] - da * (Tang[i] - Trang[i]);
    out[3 + i] = nudr[i] - kl * Rtp[i] - dl * (Tlin[i] - Trlin[i]);
  }
}

// Desired entries from laws that need the kinematics (standing.jl:60-77: centre-of-mass PD, pelvis orientation PD; SE3PDController
// on the device), lane = desired entry: runs after the momentum phase (it reads L_AMOM, MS_COM, L_H, L_T) and before the rows of the
// assembly read L_DES
template <class P> QPC_FN void desired_late(const QpcDims &D, const int *IT, qpc_dtab DT, P sm, int lane, const real *hlin) {
  const int MISC = D.lds[L_MISC];
  for (int i = lane; i < D.n_desired; i += 64) {
    const int *li = IT + D.it[IT_LAW] + 3 * i;
    qpc_dtab lw = DT + D.dt[DT_LAW] + 3 * i;
    // (D.law_mask: the JIT-specialised kernel drops the laws this controller does not use -- the SE(3) law alone is ~1.4 k
    //  instructions and its live values used to push the register allocation of the whole tick into scratch)
    if (((D.law_mask >> QPC_LAW_COM_PD) & 1) && li[0] == QPC_LAW_COM_PD) { // m c̈,  c̈ = -kp (c - c_ref) - kd h_lin / m
      const int a = li[1];
      const real hl = hlin[a]; // linear momentum of the robot: the sum of the bodies' own momenta (kinematics phase)
      const real mt = DT[D.dt[DT_GRAV] + 3];
      sm[D.lds[L_DES] + i] = mt * (-lw[0] * (sm[MISC + MS_COM + a] - lw[2])) - lw[1] * hl;
    } else if (((D.law_mask >> QPC_LAW_ORIENTATION_PD) & 1) && li[0] == QPC_LAW_ORIENTATION_PD) { // -kp rotation_vector(R) - kd R' omega (body frame)
      const int body = li[1], a = li[2];
      real H[12], rv[3];
      ldH(sm, D, body, H);
      rotation_vector(H, rv);
      real wb = 0.0;
      for (int r = 0; r < 3; r++) wb += H[3 * r + a] * sm[D.lds[L_T] + 6 * body + r];
      sm[D.lds[L_DES] + i] = -lw[0] * rv[a] - lw[1] * wb;
    } else if (((D.law_mask >> QPC_LAW_SE3_PD) & 1) && li[0] == QPC_LAW_SE3_PD) { // SE3PDController on the device (se3pdcontroller.jl:13-18), lane = component
      const int body = (li[1] & 255) - 1, base = ((li[1] >> 8) & 255) - 1, a = (li[1] >> 16) & 7, np = (li[1] >> 20) & 15, rec0 = li[2] + 6;
      real Hb[12], Ha[12], Tb[6], Ta[6], out[6];
      ldH(sm, D, body, Hb); ldH(sm, D, base, Ha);
      ld6(sm, D.lds[L_T], body, Tb); ld6(sm, D.lds[L_T], base, Ta);
      qpc_dtab lwa = DT + D.dt[DT_LAW] + 3 * li[2], lwl = lwa + 9; // gains of entries 0 (angular) and 3 (linear)
      if (np <= 1) se3_pd_law((real)sm[D.lds[L_DES] + rec0], sm + D.lds[L_DES] + rec0, Hb, Ha, Tb, Ta, lwa[0], lwa[1], lwl[0], lwl[1], out);
      else { // Piecewise (piecewise.jl:14-27): clamp to [first break, last break], the last piece whose break is <= t, its local clock
        const int br = D.lds[L_DES] + rec0 + 1; // breaks[0 .. np]
        real tc = sm[D.lds[L_DES] + rec0];
        const real b0 = sm[br], bn = sm[br + np];
        tc = tc < b0 ? b0 : (tc > bn ? bn : tc);
        int idx = 0;
        real bi = b0;
        for (int k = 1; k < np; k++) { const real bk = sm[br + k]; if (tc >= bk) { idx = k; bi = bk; } }
        se3_pd_law(tc - bi, sm + br + np + 22 * idx, Hb, Ha, Tb, Ta, lwa[0], lwa[1], lwl[0], lwl[1], out);
      }
      sm[D.lds[L_DES] + i] = out[a];
    }
  }
}

// ------------------------------------------------------------------------------------------
// the tick
// ------------------------------------------------------------------------------------------
// Static per-lane metadata (lane = body i and lane = velocity index k), read once per QP into a few
// registers (packed by the host compiler, IT_META): no table lookup sits on a dependent path of the
// tree sweeps, and the whole row costs 6 + MAXD/4 VGPRs.  (Keeping it live across the QPs of a
// workgroup was measured: the extra register pressure costs more than the reload.)
//   [0] jtype(2) | q0(7) << 2 | v0(7) << 9 | nvj(3) << 16 of body i
//   [1,2] subtree mask of body i for the composite inertia (zero when not needed)
//   [3] vbody(k)   [4,5] subtree mask of vbody(k)
//   [6 + s/4] byte s%4: s-th ancestor of body i (nearest first) + 1, 0 = none
#define QPC_META_WORDS (6 + (QPC_MAXD + 3) / 4)
struct TickMeta {
  QPC_LANE_INTS(v, QPC_META_WORDS);
};
#define MT_JT(mt) ((mt)[0] & 3)
#define MT_Q0(mt) (((mt)[0] >> 2) & 127)
#define MT_V0(mt) (((mt)[0] >> 9) & 127)
#define MT_NVJ(mt) (((mt)[0] >> 16) & 7)
#define MT_VBODY(mt) ((mt)[3] & 127)
#define MT_CMASK(mt) ((unsigned)(mt)[3] >> 8) /* contacts 0..23 attached to body `lane` */
#define MT_ANC(mt, s) ((int)(((unsigned)(mt)[6 + ((s) >> 2)] >> (8 * ((s) & 3))) & 255u) - 1)

// ------------------------------------------------------------------------------------------
// Null-space presolve (QpcDims::presolve, layout checked by qpc_compile.cpp): hard 6-row task blocks e whose rows read
//   A_e x_e + B_e x_b = b_e        (x_e: six free columns that only this block claims, x_b: the free columns that stay)
// determine x_e = P_e - Q_e x_b, [P_e | Q_e] = A_e^-1 [b_e | B_e].  Substituted into the rows that stay (M^ = M_b - M_e Q, d^ = d -
// M_e P) and into the objective 1/2 |x|^2 + g~'x of the scaled unknowns (H' = I + sum Q'Q, h = g~_b - sum Q'(P + g~_e)), and with
// y = C'x_b, H' = C C', the solver sees a problem of the SAME form (unit Hessian, linear term C^-1 h, rows M^ C^-T) in nf_s columns and
// m_s rows: DS.  Returns false -- and leaves everything as assembled -- when an elimination is ill conditioned (a leg near a
// singularity): that QP is solved in the full dimensions.
//   A_e^-1 by Householder QR in REGISTERS: block e lives in DPP row e (lanes 16 e ..), lane 16 e + c holds column c of
//   [A_e | b_e | B_e]; a reflection needs column k's lower part in every lane of the row (row_share broadcasts, no LDS, no pivoting),
//   the back-substitution the columns of R the same way.
template <class P> QPC_FN bool qpc_presolve(const QpcDims &D, const QpcDims &DS, P sm, unsigned long long *PA, unsigned long long *PT) {
  (void)PA; (void)PT;
  QPC_LANE_SCOPE
  const int n_e = D.presolve, nf_s = D.pre_nf, m_s = D.pre_m, nr = 6 * n_e, b0 = D.pre_b0;
  const int PZ = D.lds[L_PZ];
  QPC_LANE_ARRAY(qc_, 6); // this lane's column; after the back-substitution: the solution for this right-hand side
  QPC_LANE_ARRAY(qv_, 6); // what the row shares
  QPC_PHASE(lane) {
    real *col = QPC_LA(qc_, lane);
    const int e = lane >> 4, cc = lane & 15;
    const int c0 = e == 1 ? D.pre_c0[1] : (e == 2 ? D.pre_c0[2] : (e == 3 ? D.pre_c0[3] : D.pre_c0[0])); // first column of my block
    QPC_UNROLL_FULL
    for (int i = 0; i < 6; i++) {
      real v = 0.0;
      if (e < n_e && cc < 7 + nf_s) {
        const int r = 6 * e + i;
        v = cc == 6 ? (real)sm[D.lds[L_D] + r] : (real)sm[D.lds[L_M] + r * D.nfs + (cc < 6 ? c0 + cc : b0 + cc - 7)];
      }
      col[i] = v;
    }
  } QPC_END_PHASE_REG
  QPC_UNROLL_JIT
  for (int k = 0; k < 5; k++) { // H_k = I - v v' / (alpha (alpha - x_k)), v = x - alpha e_k, alpha = -sign(x_k) |x|
    QPC_ROW_SHARE(lane, QPC_LA(qv_, lane), 6 - k, QPC_LA(qc_, lane) + k, k)
    QPC_PHASE(lane) {
      real *col = QPC_LA(qc_, lane);
      const real *x = QPC_LA(qv_, lane);
      real nn = 0.0;
      QPC_UNROLL_FULL
      for (int i = 0; i < 6 - k; i++) nn += x[i] * x[i];
      const real nrm = nn > 0.0 ? nn * QPC_RSQRT(nn) : 0.0;
      const real alpha = x[0] > 0.0 ? -nrm : nrm;
      const real den = alpha * (alpha - x[0]);       // = v'v / 2 > 0 unless the column is zero
      const real tinv = den > 0.0 ? QPC_RCP(den) : 0.0;
      real w = (x[0] - alpha) * col[k];
      QPC_UNROLL_FULL
      for (int i = 1; i < 6 - k; i++) w += x[i] * col[k + i];
      w *= tinv;
      // (the reflected column itself comes out as alpha e_k up to rounding: w = 1 for it; what is left below its diagonal,
      // ~1e-16 of the entries, is never read again -- the back-substitution only uses the upper triangle)
      col[k] -= w * (x[0] - alpha);
      QPC_UNROLL_FULL
      for (int i = 1; i < 6 - k; i++) col[k + i] -= w * x[i];
    } QPC_END_PHASE_REG
  }
  unsigned long long bbad = 0ull;
  QPC_LANE_ARRAY(qd_, 2); // smallest / largest |R_jj| of my block
  QPC_UNROLL_JIT
  for (int j = 5; j >= 0; j--) { // R z = Q'rhs: z_j = y_j / R_jj, y_i -= R_ij z_j (column j of R from lane j of the row)
    QPC_ROW_SHARE(lane, QPC_LA(qv_, lane), j + 1, QPC_LA(qc_, lane), j)
    QPC_PHASE(lane) {
      real *col = QPC_LA(qc_, lane), *dd = QPC_LA(qd_, lane);
      const real *rj = QPC_LA(qv_, lane);
      if (j == 5) { dd[0] = QPC_HUGE; dd[1] = 0.0; }
      const real a = QPC_FABS(rj[j]);
      dd[0] = a < dd[0] ? a : dd[0];
      dd[1] = a > dd[1] ? a : dd[1];
      { // (only the right-hand-side lanes' results are used)
        const real z = a > 0.0 ? col[j] * QPC_RCP(rj[j]) : 0.0;
        col[j] = z;
        QPC_UNROLL_FULL
        for (int i = 0; i < j; i++) col[i] -= rj[i] * z;
      }
      if (j == 0) {
        const int e = lane >> 4, cc = (lane & 15) - 6;
        const bool mine = e < n_e && cc >= 0 && cc <= nf_s;
        // (measured: the torques lose ~1e-14 / ratio^2; 1e-4 keeps them at 1e-11).  The running min / max ignore a NaN (both
        // comparisons are false), so the last diagonal entry is tested by itself and a NaN anywhere shows up in the solution
        QPC_BALLOT_PUT(0, bbad, mine && (!(dd[0] > QPC_T_QR * dd[1]) || !(a == a) || !(col[0] == col[0]) || !(col[5] == col[5])));
        if (mine) { // columns of Z': P, then the columns of Q, then P + g~ of the eliminated unknowns
          for (int i = 0; i < 6; i++) sm[PZ + cc * nr + 6 * e + i] = col[i];
          if (cc == 0)
            for (int i = 0; i < 6; i++) {
              const int f = (e == 1 ? D.pre_c0[1] : (e == 2 ? D.pre_c0[2] : (e == 3 ? D.pre_c0[3] : D.pre_c0[0]))) + i;
              sm[PZ + (nf_s + 1) * nr + 6 * e + i] = col[i] + sm[D.lds[L_SQH] + f] * sm[D.lds[L_GV] + f];
            }
        }
      }
    } QPC_END_PHASE_REG
  }
  QPC_PHASE_SYNC
  QPC_BALLOT_GET(0, bbad);
  QPC_PROF_ACC(PS_P_QR);
  if (bbad) return false;
  QPC_PHASE(lane) { // M^, d^, H', h: one dot product of two nr-vectors per lane
    const int nh = nf_s * (nf_s + 1) / 2, n1 = m_s * nf_s, n2 = n1 + m_s, n3 = n2 + nh, n_red = n3 + nf_s;
    if (lane < n_red) {
      int left, right, out;
      real base, sgn = -1.0;
      bool isrow = false; // left operand: a row of M, whose eliminated columns sit block by block at pre_c0[e] (not one after the other)
      if (lane < n2) { // a row that stays: its part in the eliminated columns against a column of Q (M^) or P (d^)
        const int rr = lane < n1 ? lane / nf_s : lane - n1, b = lane < n1 ? lane % nf_s : -1;
        left = D.lds[L_M] + (nr + rr) * D.nfs;
        isrow = true;
        right = PZ + (1 + b) * nr;
        base = b >= 0 ? (real)sm[D.lds[L_M] + (nr + rr) * D.nfs + b0 + b] : (real)sm[D.lds[L_D] + nr + rr];
        out = D.lds[L_PA] + (b >= 0 ? rr * nf_s + b : m_s * nf_s + rr); // M^ (m_s x nf_s), then d^
      } else if (lane < n3) { // H'[a][b] = delta + <Q_a, Q_b>, packed lower
        int a = 0, b = lane - n2;
        while (b > a) { a++; b -= a; }
        left = PZ + (1 + a) * nr; right = PZ + (1 + b) * nr;
        base = a == b ? 1.0 : 0.0; sgn = 1.0;
        out = D.lds[L_PH] + lane - n2;
      } else { // h_b = g~_b - <Q_b, P + g~_e>
        const int b = lane - n3;
        left = PZ + (1 + b) * nr; right = PZ + (nf_s + 1) * nr;
        base = sm[D.lds[L_SQH] + b0 + b] * sm[D.lds[L_GV] + b0 + b];
        out = D.lds[L_PH] + nh + b;
      }
      real dot = 0.0;
      QPC_UNROLL_FULL
      for (int e = 0; e < 4; e++)
        if (e < n_e) dot += dotp(sm + left + (isrow ? D.pre_c0[e] : 6 * e), 1, sm + right + 6 * e, 1, 6);
      sm[out] = base + sgn * dot;
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_P_RED);
  const int F = D.pre_fold; // soft task rows that become part of the objective (0: they stay rows with slacks)
  if (F > 0) { // H' += M^_F' W2 M^_F, h -= M^_F' W2 d^_F  (W2: the slack Hessian blocks -- diagonal: weights are scalars here)
    QPC_PHASE(lane) {
      const int nh = nf_s * (nf_s + 1) / 2, MH = D.lds[L_PA], DH = D.lds[L_PA] + m_s * nf_s;
      if (lane < nh + nf_s) {
        int a = 0, b = lane;
        if (lane < nh) { while (b > a) { a++; b -= a; } } else { a = lane - nh; b = -1; }
        real acc = 0.0;
        QPC_UNROLL_FULL
        for (int s_ = 0; s_ < 6; s_++)
          if (s_ < F) acc += sm[MH + s_ * nf_s + a] * sm[D.lds[L_W] + D.pre_wd[s_]] * (b >= 0 ? (real)sm[MH + s_ * nf_s + b] : (real)sm[DH + s_]);
        const int o = D.lds[L_PH] + lane;
        sm[o] = sm[o] + (b >= 0 ? acc : -acc);
      }
    } QPC_END_PHASE
  }
  QPC_PHASE(lane) { // H' = C C' (every lane, in registers), C^-1; rows M^ C^-T, linear term C^-1 h
    real c[10], ci[10], dinv[4]; // packed lower triangles (nf_s <= 4), 1 / C_aa
    const int PH = D.lds[L_PH], nh = nf_s * (nf_s + 1) / 2;
    QPC_UNROLL_FULL
    for (int a = 0; a < 4; a++) {
      QPC_UNROLL_FULL
      for (int b = 0; b <= a; b++) {
        if (a < nf_s) {
          real acc = sm[PH + a * (a + 1) / 2 + b];
          QPC_UNROLL_FULL
          for (int j = 0; j < b; j++) acc -= c[a * (a + 1) / 2 + j] * c[b * (b + 1) / 2 + j];
          if (a == b) { dinv[a] = QPC_RSQRT(acc); c[a * (a + 1) / 2 + b] = acc * dinv[a]; } // H' >= I: no pivot can vanish
          else c[a * (a + 1) / 2 + b] = acc * dinv[b];
        }
      }
    }
    QPC_UNROLL_FULL
    for (int a = 0; a < 4; a++) {
      QPC_UNROLL_FULL
      for (int b = 0; b <= a; b++) { // row a of C^-1 by forward substitution on column b of the identity
        if (a < nf_s) {
          real acc = a == b ? 1.0 : 0.0;
          QPC_UNROLL_FULL
          for (int j = b; j < a; j++) acc -= c[a * (a + 1) / 2 + j] * ci[j * (j + 1) / 2 + b];
          ci[a * (a + 1) / 2 + b] = acc * dinv[a];
        }
      }
    }
    const int MH = D.lds[L_PA];
    // lane (rr, cc) of M^ C^-T and lane cc of C^-1 h are the same sum over j <= cc of (row of M^ | h) against row cc of C^-1
    const int m_f = m_s - F; // rows of the reduced problem: the rows that stay minus the folded ones
    if (lane >= 48 && lane < 48 + m_f) sm[DS.lds[L_D] + lane - 48] = sm[MH + m_s * nf_s + F + lane - 48];
    if (lane < m_f * nf_s + nf_s) {
      const bool isg = lane >= m_f * nf_s;
      const int rr = isg ? 0 : lane / nf_s, cc = isg ? lane - m_f * nf_s : lane % nf_s;
      const int src = isg ? PH + nh : MH + (F + rr) * nf_s;
      real acc = 0.0;
      QPC_UNROLL_FULL
      for (int j = 0; j < 4; j++) {
        real cij = 0.0; // C^-1[cc][j] picked with selects (cc is a lane value)
        QPC_UNROLL_FULL
        for (int a = 0; a < 4; a++)
          if (a < nf_s && j <= a) cij = a == cc ? ci[a * (a + 1) / 2 + j] : cij;
        if (j < nf_s) {
          acc += (j <= cc ? (real)sm[src + j] : 0.0) * cij;
          if (isg) sm[D.lds[L_PC] + cc * nf_s + j] = j <= cc ? cij : 0.0;
        }
      }
      if (isg) { sm[DS.lds[L_GV] + cc] = acc; sm[DS.lds[L_HV] + cc] = 1.0; sm[DS.lds[L_SQH] + cc] = 1.0; }
      else sm[DS.lds[L_M] + rr * DS.nfs + cc] = acc;
    }
  } QPC_END_PHASE
  return true;
}

// the eliminated and the whitened unknowns back from the solution y of the reduced problem: x_b = C^-T y, x_e = P_e - Q_e x_b.
// Returns a lane mask of the eliminated hard rows A_e x_e + B_e x_b = b_e that the restored solution violates by more than
// tol_accept (relative to 1 + |b|), or whose residual is not a number: the acceptance test of the reduced problem never sees these
// rows, and the conditioning test of the elimination (smallest / largest |R_jj| of an unpivoted QR) is a heuristic, not a bound
//  -- the caller downgrades the status.  The rows are read from the problem as assembled (L_M, L_D: untouched).
template <class P> QPC_FN unsigned long long qpc_postsolve(const QpcDims &D, const QpcDims &DS, P sm) {
  QPC_LANE_SCOPE
  const int n_e = D.presolve, nf_s = D.pre_nf, nr = 6 * n_e;
  QPC_PHASE(lane) {
    if (lane < nr + nf_s) {
      real xb[4] = {0.0, 0.0, 0.0, 0.0};
      QPC_UNROLL_FULL
      for (int b = 0; b < 4; b++) {
        real acc = 0.0;
        QPC_UNROLL_FULL
        for (int cc = b; cc < 4; cc++)
          if (cc < nf_s) acc += sm[D.lds[L_PC] + cc * nf_s + b] * sm[DS.lds[L_XV] + cc];
        xb[b] = acc;
      }
      const bool elim = lane < nr;
      real acc = elim ? (real)sm[D.lds[L_PZ] + lane] : 0.0;
      QPC_UNROLL_FULL
      for (int b = 0; b < 4; b++)
        if (b < nf_s) acc += elim ? -sm[D.lds[L_PZ] + (1 + b) * nr + lane] * xb[b] : (lane - nr == b ? xb[b] : 0.0);
      const int e = lane / 6, c0 = e == 1 ? D.pre_c0[1] : (e == 2 ? D.pre_c0[2] : (e == 3 ? D.pre_c0[3] : D.pre_c0[0]));
      sm[D.lds[L_XV] + (elim ? c0 + lane - 6 * e : D.pre_b0 + lane - nr)] = acc;
    }
  } QPC_END_PHASE
  unsigned long long bviol = 0ull, bnan = 0ull;
  QPC_PHASE(lane) {
    bool viol = false, isnan_ = false;
    if (lane < nr) {
      const real d = sm[D.lds[L_D] + lane];
      const real res = d - dotp(sm + D.lds[L_M] + lane * D.nfs, 1, sm + D.lds[L_XV], 1, D.nf);
      isnan_ = !(res == res);
      viol = !(QPC_FABS(res) <= (real)D.tol_accept * (1.0 + QPC_FABS(d)));
    }
    QPC_BALLOT_PUT(0, bviol, viol);
    QPC_BALLOT_PUT(1, bnan, isnan_);
  } QPC_END_PHASE
  QPC_BALLOT_GET(0, bviol);
  QPC_BALLOT_GET(1, bnan);
  return bnan != 0ull ? ~0ull : bviol; // all ones: not a number
}

// once per workgroup (before its first QP): hot solver lookups -> the persistent L_HOT region of
// the LDS slab (never overlaid, so it survives from QP to QP)
template <class P> QPC_FN void qpc_tick_prologue(const QpcDims &D, const QpcDims &DS, const int *IT, P sm) {
  QPC_LANE_SCOPE
  QPC_PHASE(lane) {
    for (int i = lane; i < D.m + 4 * D.ne; i += 64) sm[D.lds[L_HOT] + i] = (real)IT[D.it[IT_HOT] + i];
    if (D.presolve)
      for (int i = lane; i < DS.m + 4 * DS.ne; i += 64) sm[DS.lds[L_HOT] + i] = (real)IT[DS.it[IT_HOT] + i];
  } QPC_END_PHASE
}

// WARM = true is the warm-started variant (separately compiled: qpc_enable_warm_start): the active-set search starts
// from the working set the caller carries from tick to tick -- the counterpart of the reference's OSQP workspace, which
// persists between ticks (momentum.jl:58 calls solve! on the same model every tick) -- and hands the final one back.
// A search that fails from a warm set is repeated once from the cold start, so the answer never depends on the hint.
template <class P, bool WARM = false>
QPC_FN void qpc_tick(const QpcDims &D, const QpcDims &DS, const int *IT, qpc_dtab DT, P sm, const TickIO &io) {
  QPC_LANE_SCOPE
  const int nb = D.nb, nq = D.nq, nv = D.nv, nf = D.nf, nc = D.nc, N = D.N, nrho = D.nrho, ne = D.ne;
  const int m = D.m, m1 = D.m1, nw = D.nw, ntask = D.ntask;
  const int MISC = D.lds[L_MISC];
  const bool use_rho = nrho > 0;
  const int *parent = IT + D.it[IT_PARENT];
  const int *anc_ptr = IT + D.it[IT_ANC_PTR], *anc_idx = IT + D.it[IT_ANC_IDX];
  const int *desc_ptr = IT + D.it[IT_DESC_PTR], *desc_idx = IT + D.it[IT_DESC_IDX];
  (void)parent;
  unsigned long long prof_acc_[PS_COUNT], prof_t_ = 0;
  for (int i = 0; i < PS_COUNT; i++) prof_acc_[i] = 0;
  unsigned long long *PA = prof_acc_, *PT = &prof_t_;
  (void)PA; (void)PT;
  QPC_PROF_T0();

  // ======================================================================== inputs -> LDS
  TickMeta M;
  // What the inverse dynamics needs again after the solve but nothing in between does -- this lane's body inertia (10) and
  // solution-independent wrench W0 (6), this lane's motion subspace column (6) -- stays in the lane's REGISTERS across the
  // solver (QPC_KEEP_REGS).  It used to be parked in a per-workgroup slot in HBM (5.7 KB written and read back per QP); with
  // the lane index opaque per phase the kernel needs ~205 VGPRs without them, so the 44 registers are there (251 VGPRs,
  // no spills); HBM traffic per solve went from 13.9 KB to 1.9 KB, the algorithmic 1 968 B.
  QPC_LANE_ARRAY(keep_iw_, 16);
  QPC_LANE_ARRAY(keep_sk_, 6);
  real hscale = 0.0, hinvmax = 0.0, eps = 0.0, eps_ve = 0.0;
  unsigned long long bbad_in = 0ull, bneg_cap = 0ull, ben_ = 0ull;
  bool exact_ve = false, h_fast = false;
  const bool h_static_path = D.h_static;
  QPC_PHASE(lane) {
    int *mt = QPC_LI(M.v, lane);
    {
      const int aw = (D.maxd + 3) / 4, rowlen = 6 + aw;
      QPC_UNROLL_JIT
      for (int e = 0; e < QPC_META_WORDS; e++) mt[e] = e < rowlen ? IT[D.it[IT_META] + e * 64 + lane] : 0;
    }
    // every load of this phase is independent of the others (ONE round trip to the records, which may live in pinned
    // host memory behind PCIe: qpc_control_batch reads the caller's arrays in place); the pinned accelerations, which
    // need desired[IT_PDES_K[k]], are picked out of the LDS copy of the desired record one phase later
    for (int i = lane; i < nq; i += 64) sm[D.lds[L_Q] + i] = io.q[i];
    for (int i = lane; i < nv; i += 64) sm[D.lds[L_V] + i] = io.v[i];
    for (int i = lane; i < D.n_desired; i += 64) sm[D.lds[L_DES] + i] = desired_early(D, IT, DT, io, i);
    real hc = 0.0; // this lane's share of the regularisation scale: 2 |weight| of its enabled contacts
    bool neg_cap = false;
    bool bad_in = false; // a NaN contact weight or cap: a NaN cap would otherwise read as "disabled" (hi > 0 is false) and
                         // the QP would be solved as if the caller had meant that
    for (int i = lane; i < nc; i += 64) {
      const real cwi = io.cw[i], cmi = io.cmax[i];
      bad_in = bad_in || !(cwi == cwi) || !(cmi == cmi);
      neg_cap = neg_cap || cmi < 0.0; // 0 <= rho <= rho_max < 0 (contacts.jl:57) has no solution
      sm[D.lds[L_CW] + i] = cwi; sm[D.lds[L_CMAX] + i] = cmi;
      if (use_rho && cmi > 0.0 && 2.0 * QPC_FABS(cwi) > hc) hc = 2.0 * QPC_FABS(cwi);
      if (D.k_tab) {
        sm[MISC + MS_KS + i] = 2.0 * cwi; // scale of this contact's rho block of the Hessian (kr_blocks)
        if (D.k_classes > 1) sm[MISC + MS_KS + 8 + i] = (real)(D.k_class[i] * 320);
      }
    }
    for (int i = lane; i < D.n_tw; i += 64) sm[D.lds[L_TW] + i] = io.tw[i];
    for (int i = lane; i < D.n_surf; i += 64) sm[D.lds[L_SURF] + i] = io.surf[i];
    if (h_static_path) { QPC_RED_MAX_PUT(0, hscale, hc); }
    if (nc) { QPC_BALLOT_PUT(2, bbad_in, bad_in); QPC_BALLOT_PUT(3, bneg_cap, neg_cap); }
#ifdef QPC_F32
    if (nc) { QPC_BALLOT_PUT(4, ben_, lane < nc && io.cmax[lane < nc ? lane : 0] > 0.0); }
#endif
  } QPC_END_PHASE
  if (nc) { QPC_BALLOT_GET(2, bbad_in); QPC_BALLOT_GET(3, bneg_cap); }
#ifdef QPC_F32
  if (nc) { QPC_BALLOT_GET(4, ben_); }
#endif
  // Static Hessian blocks (D.h_static: no per-QP task weights): the regularisation scale is max(static diagonal,
  // contact weights of THIS QP), known right after the input phase -- the two phases that used to derive it from the LDS
  // copies of H (a max reduction over HV / W, then rsqrt / block inverses: 4 % of the tick) are only run in the general case
  if (h_static_path) {
    hscale = QPC_RED_MAX_GET(0, hscale);
    if ((real)D.h_static_max > hscale) hscale = (real)D.h_static_max;
    if (!(hscale > 0.0)) hscale = 1.0;
    eps = (real)D.eps_primal * hscale;
    exact_ve = D.ve_exact_ok && (QPC_T_EXACT * hscale <= (real)D.h_static_min);
    eps_ve = exact_ve ? 0.0 : eps;
    h_fast = exact_ve; // everything static is exact: HV, SQH, W, WINV come from the tables
  }

  QPC_PROF_ACC(PS_K_INPUT);
  // ======================================================================== kinematics (a2)
  QPC_PHASE(lane) { // joint transforms: x_parent = Rf (Rj x + pj) + pf
    for (int i = lane; i < nv; i += 64) { // pinned accelerations (hard JointAccelerationTask, src/tasks.jl:185-189)
      const int pd = IT[D.it[IT_PDES_K] + i];
      sm[D.lds[L_VD] + i] = pd >= 0 ? (real)sm[D.lds[L_DES] + pd] : 0.0;
    }
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
      const int jt = MT_JT(mt), q0 = MT_Q0(mt);
      real ax[3], Rf[9], pf[3]; // per-body constants, stored [component][body]
      for (int k = 0; k < 3; k++) { ax[k] = DT[D.dt[DT_AXIS] + k * nb + i]; pf[k] = DT[D.dt[DT_PF] + k * nb + i]; }
      for (int k = 0; k < 9; k++) Rf[k] = DT[D.dt[DT_RF] + k * nb + i];
      real Rj[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pj[3] = {0, 0, 0};
      if (jt == QPC_JOINT_REVOLUTE) {
        real s, c;
        QPC_SINCOS(sm[D.lds[L_Q] + q0], s, c);
        real C = 1 - c;
        Rj[0] = c + ax[0] * ax[0] * C; Rj[1] = ax[0] * ax[1] * C - ax[2] * s; Rj[2] = ax[0] * ax[2] * C + ax[1] * s;
        Rj[3] = ax[1] * ax[0] * C + ax[2] * s; Rj[4] = c + ax[1] * ax[1] * C; Rj[5] = ax[1] * ax[2] * C - ax[0] * s;
        Rj[6] = ax[2] * ax[0] * C - ax[1] * s; Rj[7] = ax[2] * ax[1] * C + ax[0] * s; Rj[8] = c + ax[2] * ax[2] * C;
      } else if (jt == QPC_JOINT_PRISMATIC) {
        real qq = sm[D.lds[L_Q] + q0];
        pj[0] = ax[0] * qq; pj[1] = ax[1] * qq; pj[2] = ax[2] * qq;
      } else if (jt == QPC_JOINT_QUAT_FLOATING) {
        real w = sm[D.lds[L_Q] + q0], x = sm[D.lds[L_Q] + q0 + 1], y = sm[D.lds[L_Q] + q0 + 2], z = sm[D.lds[L_Q] + q0 + 3];
        Rj[0] = 1 - 2 * (y * y + z * z); Rj[1] = 2 * (x * y - w * z); Rj[2] = 2 * (x * z + w * y);
        Rj[3] = 2 * (x * y + w * z); Rj[4] = 1 - 2 * (x * x + z * z); Rj[5] = 2 * (y * z - w * x);
        Rj[6] = 2 * (x * z - w * y); Rj[7] = 2 * (y * z + w * x); Rj[8] = 1 - 2 * (x * x + y * y);
        pj[0] = sm[D.lds[L_Q] + q0 + 4]; pj[1] = sm[D.lds[L_Q] + q0 + 5]; pj[2] = sm[D.lds[L_Q] + q0 + 6];
      }
      real H[12];
      mm3(Rf, Rj, H);
      mv3(Rf, pj, H + 9);
      H[9] += pf[0]; H[10] += pf[1]; H[11] += pf[2];
      st(sm + D.lds[L_RL] + 12 * i, 12, H);
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_K_JOINT);
  QPC_PHASE(lane) { // transform_to_root by walking the (register-resident) ancestor chain
    if (lane < nb) {
      const int *mt = QPC_LI(M.v, lane);
      real H[12], A[12], t[3];
      ld(sm + D.lds[L_RL] + 12 * lane, 12, H);
      QPC_UNROLL_JIT
      for (int s = 0; s < QPC_MAXD; s++) {
        const int an = MT_ANC(mt, s);
        if (an >= 0) {
          ld(sm + D.lds[L_RL] + 12 * an, 12, A);
          mv3(A, H + 9, t);
          mm3(A, H, H);
          H[9] = t[0] + A[9]; H[10] = t[1] + A[10]; H[11] = t[2] + A[11];
        }
      }
      st(sm + D.lds[L_H] + 12 * lane, 12, H);
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_K_WORLD);
  QPC_PHASE(lane) { // per body: motion subspace columns of its joint (world), joint twist, world inertia
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
      const int jt = MT_JT(mt), v0 = MT_V0(mt), n = MT_NVJ(mt);
      real H[12], tj[6] = {0, 0, 0, 0, 0, 0};
      ld(sm + D.lds[L_H] + 12 * i, 12, H);
      const real ax[3] = {DT[D.dt[DT_AXIS] + i], DT[D.dt[DT_AXIS] + nb + i], DT[D.dt[DT_AXIS] + 2 * nb + i]};
      for (int c = 0; c < n; c++) {
        real S[6] = {0, 0, 0, 0, 0, 0};
        if (jt == QPC_JOINT_REVOLUTE) { mv3(H, ax, S); cross3(H + 9, S, S + 3); }
        else if (jt == QPC_JOINT_PRISMATIC) { mv3(H, ax, S + 3); }
        else { // quaternion floating: columns of Ad(H)
          int cc = c % 3;
          real col[3] = {H[cc], H[3 + cc], H[6 + cc]};
          if (c < 3) { S[0] = col[0]; S[1] = col[1]; S[2] = col[2]; cross3(H + 9, col, S + 3); }
          else { S[3] = col[0]; S[4] = col[1]; S[5] = col[2]; }
        }
        st(sm + D.lds[L_S] + 6 * (v0 + c), 6, S);
        const real vv = sm[D.lds[L_V] + v0 + c];
        for (int a = 0; a < 6; a++) tj[a] += S[a] * vv;
      }
      st(sm + D.lds[L_TJ] + 6 * i, 6, tj);
      real cw[3];
      real I[6], cb[3];
      for (int k = 0; k < 6; k++) I[k] = DT[D.dt[DT_ICOM] + k * nb + i];
      for (int k = 0; k < 3; k++) cb[k] = DT[D.dt[DT_COM] + k * nb + i];
      mv3(H, cb, cw);
      cw[0] += H[9]; cw[1] += H[10]; cw[2] += H[11];
      real Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], Rt[9], Iw[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = H[3 * b + a];
      mm3(H, Ib, RI);
      mm3(RI, Rt, Iw);
      real mm = DT[D.dt[DT_MASS] + i], cc = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
      real o[10] = {mm, mm * cw[0], mm * cw[1], mm * cw[2],
                      Iw[0] + mm * (cc - cw[0] * cw[0]), Iw[4] + mm * (cc - cw[1] * cw[1]), Iw[8] + mm * (cc - cw[2] * cw[2]),
                      Iw[1] - mm * cw[0] * cw[1], Iw[2] - mm * cw[0] * cw[2], Iw[5] - mm * cw[1] * cw[2]};
      st(sm + D.lds[L_IW] + 10 * i, 10, o);
      for (int a = 0; a < 10; a++) QPC_LA(keep_iw_, lane)[a] = o[a]; // needed again by the RNEA only
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_K_SUBSP);
  QPC_PHASE(lane) { // twist_wrt_world (chain sum); bias increment T_i x (S_j v_j) + S_j v̇_j(pinned):
                    // the accelerations of the PINNED joints ride along with the velocity-product bias,
                    // so every J̇v / Ȧv below already contains J_pin v̇_pin (no separate pass)
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
      real T[6], tj[6], cr[6];
      ld(sm + D.lds[L_TJ] + 6 * i, 6, tj);
      for (int a = 0; a < 6; a++) T[a] = tj[a];
      QPC_UNROLL_JIT
      for (int s = 0; s < QPC_MAXD; s++) {
        const int an = MT_ANC(mt, s);
        if (an >= 0)
          for (int a = 0; a < 6; a++) T[a] += sm[D.lds[L_TJ] + 6 * an + a];
      }
      crm(T, tj, cr);
      const int v0 = MT_V0(mt), n = MT_NVJ(mt);
      for (int c = 0; c < n; c++) {
        const real vdp = sm[D.lds[L_VD] + v0 + c]; // zero for free joints at this point
        for (int a = 0; a < 6; a++) cr[a] += sm[D.lds[L_S] + 6 * (v0 + c) + a] * vdp;
      }
      st(sm + D.lds[L_T] + 6 * i, 6, T);
      st(sm + D.lds[L_CR] + 6 * i, 6, cr);
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_K_TWIST);
  real hlin[3] = {0.0, 0.0, 0.0}; // linear momentum of the whole robot (wave sum; only a centre-of-mass law reads it)
  const bool com_law = ((D.law_mask >> QPC_LAW_COM_PD) & 1) != 0;
  QPC_PHASE(lane) { // bias_acceleration (+ pinned), composite inertias (subtree mask), momentum-rate bias
    real hl_[3] = {0.0, 0.0, 0.0};
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
      real Ab[6], T[6], I[10], Ic[10], h[6], ia[6], cf[6];
      ld(sm + D.lds[L_CR] + 6 * i, 6, Ab);
      QPC_UNROLL_JIT
      for (int s = 0; s < QPC_MAXD; s++) {
        const int an = MT_ANC(mt, s);
        if (an >= 0)
          for (int a = 0; a < 6; a++) Ab[a] += sm[D.lds[L_CR] + 6 * an + a];
      }
      st(sm + D.lds[L_AB] + 6 * i, 6, Ab);
      for (int a = 0; a < 10; a++) Ic[a] = 0.0;
      unsigned long long msk = ((unsigned long long)(unsigned)mt[2] << 32) | (unsigned long long)(unsigned)mt[1];
      while (msk) {
        const int k = QPC_CTZ64(msk);
        msk &= msk - 1;
        for (int a = 0; a < 10; a++) Ic[a] += sm[D.lds[L_IW] + 10 * k + a];
      }
      if (i != D.crb_full_body) st(sm + D.lds[L_IC] + 10 * i, 10, Ic);
      ld(sm + D.lds[L_IW] + 10 * i, 10, I);
      ld(sm + D.lds[L_T] + 6 * i, 6, T);
      inertia_mul(I, T, h);
      for (int a = 0; a < 3; a++) hl_[a] = h[3 + a]; // this body's linear momentum (for a centre-of-mass law)
      inertia_mul(I, Ab, ia);
      crf(T, h, cf);
      for (int a = 0; a < 6; a++) ia[a] += cf[a];
      st(sm + D.lds[L_HB] + 6 * i, 6, ia); // I (Ab + A_pinned) + T x* I T
      // W0 = I (Ab - g) + T x* I T : everything of the body wrench that does not depend on the solution
      real mg[6] = {0, 0, 0, -DT[D.dt[DT_GRAV]], -DT[D.dt[DT_GRAV] + 1], -DT[D.dt[DT_GRAV] + 2]}, ig[6];
      inertia_mul(I, mg, ig);
      for (int a = 0; a < 6; a++) QPC_LA(keep_iw_, lane)[10 + a] = ia[a] + ig[a];
    }
    if (lane >= 51 && lane < 54) { // centre of mass (IW[1..3] = m c): last reader of L_IW, which L_AMOM overlays
      const int a = lane - 51;
      real acc = 0.0;
      for (int i0 = 0; i0 < nb; i0 += 8) {
        real x[8];
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) x[u] = sm[D.lds[L_IW] + 10 * (i0 + u < nb ? i0 + u : 0) + 1 + a];
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) acc += (i0 + u < nb) ? x[u] : 0.0;
      }
      sm[MISC + MS_COM + a] = acc / DT[D.dt[DT_GRAV] + 3];
    }
    if (lane >= 54 && D.crb_full_body >= 0) { // whole-robot composite inertia: component (lane-54), batched sum
      const int a = lane - 54;
      real acc = 0.0;
      for (int i0 = 0; i0 < nb; i0 += 8) {
        real x[8];
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) x[u] = sm[D.lds[L_IW] + 10 * (i0 + u < nb ? i0 + u : 0) + a];
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) acc += (i0 + u < nb) ? x[u] : 0.0;
      }
      sm[D.lds[L_IC] + 10 * D.crb_full_body + a] = acc;
    }
    if (com_law) { QPC_RED_SUM_PUT(0, hlin[0], hl_[0]); QPC_RED_SUM_PUT(1, hlin[1], hl_[1]); QPC_RED_SUM_PUT(2, hlin[2], hl_[2]); }
  } QPC_END_PHASE
  if (com_law) { hlin[0] = QPC_RED_SUM_GET(0, hlin[0]); hlin[1] = QPC_RED_SUM_GET(1, hlin[1]); hlin[2] = QPC_RED_SUM_GET(2, hlin[2]); }
  QPC_PROF_ACC(PS_K_BIAS);
  QPC_PHASE(lane) { // momentum matrix, CoM, momentum-rate bias, Hessian diagonal
    // static-Hessian path: its table reads are issued FIRST, so that they travel while the momentum matrix is computed
    // (in source order they sat behind three dependent waits at the end of the phase)
    real hv0_pre = 0.0, winv_pre = 0.0, grav_pre[4] = {0.0, 0.0, 0.0, 0.0};
    if (h_fast) {
      hv0_pre = DT[D.dt[DT_HV0] + (lane < nf ? lane : 0)];
      winv_pre = DT[D.dt[DT_WINV0] + (lane < D.wsize ? lane : 0)];
      for (int a = 0; a < 4; a++) grav_pre[a] = DT[D.dt[DT_GRAV] + a];
    }
    if (lane < nv) {
      const int k = lane;
      const int *mt = QPC_LI(M.v, lane);
      real Ic[10], S[6], h[6];
      ld(sm + D.lds[L_IC] + 10 * MT_VBODY(mt), 10, Ic);
      ld(sm + D.lds[L_S] + 6 * k, 6, S);
      for (int a = 0; a < 6; a++) QPC_LA(keep_sk_, lane)[a] = S[a]; // for tau_k = S_k . F (RNEA)
      inertia_mul(Ic, S, h);
      for (int a = 0; a < 6; a++) sm[D.lds[L_AMOM] + a * nv + k] = h[a];
    }
    if (lane >= 3 && lane < 9) {
      real acc = 0.0;
      for (int i0 = 0; i0 < nb; i0 += 8) { // loads in batches of 8 ahead of the adds, like the centre-of-mass sums above
        real x[8];
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) x[u] = sm[D.lds[L_HB] + 6 * (i0 + u < nb ? i0 + u : 0) + (lane - 3)];
        QPC_UNROLL_FULL
        for (int u = 0; u < 8; u++) acc += (i0 + u < nb) ? x[u] : 0.0;
      }
      sm[MISC + MS_ADV + (lane - 3)] = acc; // Ȧv + A_pin v̇_pin
    }
    for (int f = lane; f < nf; f += 64) { // Hv = 2 reg + folded soft joint tasks (a12)
      real hv = h_fast ? (f < 64 ? hv0_pre : DT[D.dt[DT_HV0] + f]) : DT[D.dt[DT_REG2] + IT[D.it[IT_FREE_K] + f]], gv = 0.0;
      for (int s = 0; s < D.nsj; s++)
        if (IT[D.it[IT_SJ_F] + s] == f) {
          int twi = IT[D.it[IT_SJ_TW] + s];
          real w = twi >= 0 ? sm[D.lds[L_TW] + twi] : DT[D.dt[DT_SJ_W] + s];
          if (!h_fast) hv += 2.0 * w;
          const int sd = IT[D.it[IT_SJ_DES] + s]; // -1: a per-QP regularisation weight (desired acceleration 0)
          if (sd >= 0) gv -= 2.0 * w * sm[D.lds[L_DES] + sd];
        }
      sm[D.lds[L_HV] + f] = hv;
      sm[D.lds[L_GV] + f] = gv;
      if (h_fast) sm[D.lds[L_SQH] + f] = QPC_RSQRT(hv);
    }
    if (h_fast) {
      for (int e = lane; e < D.wsize; e += 64) sm[D.lds[L_WINV] + e] = e < 64 ? winv_pre : DT[D.dt[DT_WINV0] + e];
      if (lane == 63) { sm[MISC + MS_EPS] = eps; sm[MISC + MS_HSCALE] = hscale; }
      if (lane >= 10 && lane < 16 && nw) { // gravity wrench about the world origin (momentum.jl:166-170); the centre of
                                           // mass is one phase old
        const int a = lane - 10;
        real fg[3], cxf[3], com[3] = {sm[MISC + MS_COM], sm[MISC + MS_COM + 1], sm[MISC + MS_COM + 2]};
        const real mt = grav_pre[3];
        fg[0] = mt * grav_pre[0]; fg[1] = mt * grav_pre[1]; fg[2] = mt * grav_pre[2];
        cross3(com, fg, cxf);
        sm[MISC + MS_WG + a] = a < 3 ? cxf[a] : fg[a - 3];
      }
    }
    for (int t = 0; t < ntask; t++) { // slack Hessian blocks 2*sym(W)
      int w0 = IT[D.it[IT_T_WOFF] + t];
      if (w0 < 0) continue;
      int d = IT[D.it[IT_T_DIM] + t], wk = IT[D.it[IT_T_WKIND] + t], tw0 = IT[D.it[IT_T_TW] + t];
      for (int e = lane; e < d * d; e += 64) {
        int a = e / d, b = e % d;
        real w;
        if (wk == QPC_WEIGHT_SCALAR_PER_QP) w = (a == b) ? 2.0 * sm[D.lds[L_TW] + tw0] : 0.0;
        else if (wk == QPC_WEIGHT_MATRIX_PER_QP) w = sm[D.lds[L_TW] + tw0 + a * d + b] + sm[D.lds[L_TW] + tw0 + b * d + a];
        else w = DT[D.dt[DT_T_W] + w0 + e];
        sm[D.lds[L_W] + w0 + e] = w;
      }
    }
  } QPC_END_PHASE

  QPC_PROF_ACC(PS_KIN);
  QPC_STOP(1)
  if (h_fast && D.n_des_late) { // static Hessian blocks: the laws get a phase of their own (the general path evaluates them in its
                                // regularisation-scale phase below)
    QPC_PHASE(lane) { desired_late(D, IT, DT, sm, lane, hlin); } QPC_END_PHASE
  }
  // regularisation scale: eps = eps_primal * max diagonal of the Hessian
  if (!h_fast) {
  QPC_PHASE(lane) {
    real h = 0.0, hm = QPC_HUGE; // largest diagonal of H; smallest diagonal of its v̇ and slack blocks
    for (int f = lane; f < nf; f += 64) { real a = sm[D.lds[L_HV] + f]; hm = a < hm ? a : hm; a = QPC_FABS(a); h = a > h ? a : h; }
    for (int t = lane; t < ntask; t += 64) {
      int w0 = IT[D.it[IT_T_WOFF] + t];
      if (w0 < 0) continue;
      int d = IT[D.it[IT_T_DIM] + t];
      for (int a = 0; a < d; a++) { real w = sm[D.lds[L_W] + w0 + a * d + a]; hm = w < hm ? w : hm; w = QPC_FABS(w); h = w > h ? w : h; }
    }
    QPC_RED_MAX_PUT(1, hinvmax, hm < QPC_HUGE ? (hm > QPC_TINY ? 1.0 / hm : QPC_HUGE) : 0.0); // (reductions are max over non-negative values)
    if (use_rho)
      for (int c = lane; c < nc; c += 64) {
        real w = (sm[D.lds[L_CMAX] + c] > 0.0) ? 2.0 * QPC_FABS(sm[D.lds[L_CW] + c]) : 0.0;
        h = w > h ? w : h;
      }
    if (D.n_des_late) desired_late(D, IT, DT, sm, lane, hlin); // desired laws that need the kinematics (standing.jl:60-77), lane = desired entry
    QPC_RED_MAX_PUT(0, hscale, h);
  } QPC_END_PHASE
  hscale = QPC_RED_MAX_GET(0, hscale);
  hinvmax = QPC_RED_MAX_GET(1, hinvmax);
  if (!(hscale > 0.0)) hscale = 1.0;
  eps = (real)D.eps_primal * hscale;
  // exact_ve: every v̇ and slack diagonal of the Hessian is comfortably positive (and no matrix-weighted
  // slack block): those blocks are then inverted WITHOUT the proximal eps, which confines the eps bias to
  // the (rank-deficient) rho blocks and enables the cheap wrench-space polish (Solver::polish_exact)
  exact_ve = D.ve_exact_ok && (hinvmax * QPC_T_EXACT * hscale <= 1.0); // min diagonal >= 1e-6 * max diagonal
  eps_ve = exact_ve ? 0.0 : eps;
  } // !h_fast

  // ======================================================================== assembly (a3-a12)
  if (!h_fast) {
  QPC_PHASE(lane) {
    if (lane == 0) { sm[MISC + MS_EPS] = eps; sm[MISC + MS_HSCALE] = hscale; }
    if (lane < 6 && nw) { // gravity wrench about the world origin (momentum.jl:166-170)
      real fg[3], cxf[3], com[3] = {sm[MISC + MS_COM], sm[MISC + MS_COM + 1], sm[MISC + MS_COM + 2]};
      real mt = DT[D.dt[DT_GRAV] + 3];
      fg[0] = mt * DT[D.dt[DT_GRAV]]; fg[1] = mt * DT[D.dt[DT_GRAV] + 1]; fg[2] = mt * DT[D.dt[DT_GRAV] + 2];
      cross3(com, fg, cxf);
      sm[MISC + MS_WG + lane] = lane < 3 ? cxf[lane] : fg[lane - 3];
    }
    for (int f = lane; f < nf; f += 64) sm[D.lds[L_SQH] + f] = QPC_RSQRT(sm[D.lds[L_HV] + f] + eps_ve);
    for (int t = lane; t < ntask; t += 64) { // (W + eps I)^-1 per soft generic task
      int w0 = IT[D.it[IT_T_WOFF] + t];
      if (w0 < 0) continue;
      int d = IT[D.it[IT_T_DIM] + t];
      for (int e = 0; e < d * d; e++) sm[D.lds[L_WINV] + w0 + e] = sm[D.lds[L_W] + w0 + e] + ((e / d == e % d) ? eps_ve : 0.0);
      spd_inverse_inplace(sm + D.lds[L_WINV] + w0, d);
    }
  } QPC_END_PHASE
  } // !h_fast

  QPC_PROF_ACC(PS_A_PRE);
  QPC_PHASE(lane) { // scaled constraint matrix M = E diag(sqh), right-hand side d
    for (int it_ = lane; it_ < D.n_items; it_ += 64) {
      // one item = one (row block, free column): the Jacobian column of a path task is transformed
      // into the task frame ONCE and fills all rows of the block (qpc_program.h IT_MCODE)
      const int c0 = IT[D.it[IT_MCODE] + 2 * it_], c1 = IT[D.it[IT_MCODE] + 2 * it_ + 1];
      const int kind = c0 & 7, sg = ((c0 >> 3) & 3) - 1, k = (c0 >> 5) & 127, aux = (c0 >> 12) & 127, t = (c0 >> 19) & 63;
      const int r0 = c1 & 127, dim = (c1 >> 7) & 15, f = (c1 >> 11) & 127;
      const real sq = sm[D.lds[L_SQH] + f];
      real val[6] = {0, 0, 0, 0, 0, 0};
      if (kind == 7) { // wrench balance rows (a11), S dropped (invertible)
        for (int a = 0; a < 6; a++) val[a] = sm[D.lds[L_AMOM] + a * nv + k];
      } else if (sg != 0) {
        if (kind == QPC_TASK_JOINT_ACCELERATION) {
          for (int a = 0; a < 6; a++) val[a] = (a == aux - 1) ? 1.0 : 0.0;
        } else if (kind == QPC_TASK_MOMENTUM_RATE || kind == QPC_TASK_LINEAR_MOMENTUM_RATE) {
          real lin[3] = {sm[D.lds[L_AMOM] + 3 * nv + k], sm[D.lds[L_AMOM] + 4 * nv + k], sm[D.lds[L_AMOM] + 5 * nv + k]};
          if (kind == QPC_TASK_LINEAR_MOMENTUM_RATE) { val[0] = lin[0]; val[1] = lin[1]; val[2] = lin[2]; }
          else { // centroidal: ang - com x lin (src/tasks.jl:208-213)
            real com[3] = {sm[MISC + MS_COM], sm[MISC + MS_COM + 1], sm[MISC + MS_COM + 2]}, cx[3];
            cross3(com, lin, cx);
            for (int a = 0; a < 3; a++) { val[a] = sm[D.lds[L_AMOM] + a * nv + k] - cx[a]; val[3 + a] = lin[a]; }
          }
        } else {
          const int frame = aux - 1;
          real H[12], S[6], J[6];
          ldH(sm, D, frame, H);
          ld(sm + D.lds[L_S] + 6 * k, 6, S);
          motion_to_frame(H, S, J);
          if (kind == QPC_TASK_SPATIAL_ACCELERATION) { for (int a = 0; a < 6; a++) val[a] = sg * J[a]; }
          else if (kind == QPC_TASK_ANGULAR_ACCELERATION) { for (int a = 0; a < 3; a++) val[a] = sg * J[a]; }
          else if (kind == QPC_TASK_LINEAR_ACCELERATION) { for (int a = 0; a < 3; a++) val[a] = sg * J[3 + a]; }
          else { // point jacobian J_lin - p^ J_ang in the base frame (src/tasks.jl:157-160)
            real Ht[12], pw[3], d3[3], pF[3], px[3];
            int target = IT[D.it[IT_T_TARGET] + t];
            ldH(sm, D, target, Ht);
            { const real pt_[3] = {DT[D.dt[DT_T_POINT] + 3 * t], DT[D.dt[DT_T_POINT] + 3 * t + 1], DT[D.dt[DT_T_POINT] + 3 * t + 2]}; mv3(Ht, pt_, pw); }
            d3[0] = pw[0] + Ht[9] - H[9]; d3[1] = pw[1] + Ht[10] - H[10]; d3[2] = pw[2] + Ht[11] - H[11];
            mtv3(H, d3, pF);
            cross3(pF, J, px);
            for (int a = 0; a < 3; a++) val[a] = sg * (J[3 + a] - px[a]);
          }
        }
      }
      for (int a = 0; a < 6; a++)
        if (a < dim) sm[D.lds[L_M] + (r0 + a) * D.nfs + f] = val[a] * sq;
    }
    QPC_PROF_ACC(PS_A_ITEMS);
    for (int r = lane; r < m; r += 64) {
      // packed row code: kind(3) | a(3) | frame+1(7) | base+1(7) | target+1(7)
      const int code = IT[D.it[IT_RCODE] + r], di = IT[D.it[IT_R_DES] + r], t = IT[D.it[IT_R_T] + r], pk = IT[D.it[IT_R_PINK] + r];
      const int kind = code & 7, a = (code >> 3) & 7, frame = ((code >> 6) & 127) - 1, base = ((code >> 13) & 127) - 1,
                target = ((code >> 20) & 127) - 1;
      real rhs;
      if (kind == 7) rhs = sm[MISC + MS_WG + a] - sm[MISC + MS_ADV + a];
      else {
        real des = sm[D.lds[L_DES] + di];
        if (kind == QPC_TASK_JOINT_ACCELERATION) {
          // generic joint rows (duplicate pin / matrix weight): a pinned acceleration moves to the rhs
          rhs = des - (pk >= 0 ? (real)sm[D.lds[L_VD] + pk] : 0.0);
        } else if (kind == QPC_TASK_MOMENTUM_RATE || kind == QPC_TASK_LINEAR_MOMENTUM_RATE) {
          int aa = kind == QPC_TASK_LINEAR_MOMENTUM_RATE ? a + 3 : a;
          real b = sm[MISC + MS_ADV + aa];
          if (aa < 3) {
            real com[3] = {sm[MISC + MS_COM], sm[MISC + MS_COM + 1], sm[MISC + MS_COM + 2]}, lin[3], cx[3];
            for (int i = 0; i < 3; i++) lin[i] = sm[MISC + MS_ADV + 3 + i];
            cross3(com, lin, cx);
            b -= cx[aa];
          }
          rhs = des - b;
        } else {
          // J̇v = transform(state, bias(target) - bias(base), frame) (src/tasks.jl:37-38); the bias
          // already carries the acceleration produced by the pinned joints
          real At[6], Abs[6], Tt[6], Tb[6], Tf[6], trel[6], cr[6], aw[6], aF[6], H[12];
          ld6(sm, D.lds[L_AB], target, At); ld6(sm, D.lds[L_AB], base, Abs);
          ld6(sm, D.lds[L_T], target, Tt); ld6(sm, D.lds[L_T], base, Tb); ld6(sm, D.lds[L_T], frame, Tf);
          for (int i = 0; i < 6; i++) trel[i] = Tt[i] - Tb[i];
          crm(Tf, trel, cr);
          for (int i = 0; i < 6; i++) aw[i] = At[i] - Abs[i] - cr[i];
          ldH(sm, D, frame, H);
          motion_to_frame(H, aw, aF);
          if (kind == QPC_TASK_SPATIAL_ACCELERATION || kind == QPC_TASK_ANGULAR_ACCELERATION) rhs = des - aF[a];
          else if (kind == QPC_TASK_LINEAR_ACCELERATION) rhs = des - aF[3 + a];
          else { // point: w x pdot + J̇v_ang x p + J̇v_lin (src/tasks.jl:166-170)
            real Ht[12], pw[3], d3[3], pF[3], TF[6], pd[3], t1[3], t2[3];
            ldH(sm, D, target, Ht);
            { const real pt_[3] = {DT[D.dt[DT_T_POINT] + 3 * t], DT[D.dt[DT_T_POINT] + 3 * t + 1], DT[D.dt[DT_T_POINT] + 3 * t + 2]}; mv3(Ht, pt_, pw); }
            d3[0] = pw[0] + Ht[9] - H[9]; d3[1] = pw[1] + Ht[10] - H[10]; d3[2] = pw[2] + Ht[11] - H[11];
            mtv3(H, d3, pF);
            motion_to_frame(H, trel, TF);
            cross3(TF, pF, pd);
            pd[0] += TF[3]; pd[1] += TF[4]; pd[2] += TF[5];
            cross3(TF, pd, t1);
            cross3(aF, pF, t2);
            rhs = des - (t1[a] + t2[a] + aF[3 + a]);
          }
        }
      }
      sm[D.lds[L_D] + r] = rhs;
    }
    QPC_PROF_ACC(PS_A_ROWS);
    // ---------------------------------------------------------------- contacts (a9, a10)
    if (use_rho)
      for (int j = lane; j < nrho; j += 64) {
        int c = j / N, i = j % N, body = IT[D.it[IT_C_BODY] + c];
        real nrm[3] = {DT[D.dt[DT_C_NORMAL] + 3 * c], DT[D.dt[DT_C_NORMAL] + 3 * c + 1], DT[D.dt[DT_C_NORMAL] + 3 * c + 2]};
        real mu = DT[D.dt[DT_C_MU] + c];
        const int pq = IT[D.it[IT_C_PERQP] + c]; // QPC_CONTACT_*_PER_QP bits (contacts.jl:39: Parameters)
        real pos[3] = {DT[D.dt[DT_C_POS] + 3 * c], DT[D.dt[DT_C_POS] + 3 * c + 1], DT[D.dt[DT_C_POS] + 3 * c + 2]};
        if (pq) {
          int so = IT[D.it[IT_C_SURF] + c];
          if (pq & 1) {
            nrm[0] = sm[D.lds[L_SURF] + so]; nrm[1] = sm[D.lds[L_SURF] + so + 1]; nrm[2] = sm[D.lds[L_SURF] + so + 2];
            mu = sm[D.lds[L_SURF] + so + 3];
            so += 4;
          }
          if (pq & 2) { pos[0] = sm[D.lds[L_SURF] + so]; pos[1] = sm[D.lds[L_SURF] + so + 1]; pos[2] = sm[D.lds[L_SURF] + so + 2]; }
        }
        // Rotations.rotation_between(z, n) (contacts.jl:11)
        real nn = QPC_SQRT(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        real w = nn + nrm[2], x, y, z = 0.0;
        if (QPC_FABS(w) < 100 * QPC_EPS_MACH) { x = 1; y = 0; } /* Rotations.perpendicular_vector((0,0,1)) = (1,0,0) */ else { x = -nrm[1]; y = nrm[0]; }
        real qn = QPC_RSQRT(w * w + x * x + y * y);
        w *= qn; x *= qn; y *= qn;
        real Rn[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
        real H[12], Rc[9], pc[3];
        ldH(sm, D, body, H);
        mm3(H, Rn, Rc);
        mv3(H, pos, pc);
        pc[0] += H[9]; pc[1] += H[10]; pc[2] += H[11];
        // forcebasis(mu, N) (contacts.jl:16-23)
        real sn, cs, inv = QPC_RSQRT(mu * mu + 1.0);
#ifdef QPC_BASIS_TABLE
        // JIT: cos / sin of the N basis directions are literals (computed by the host when the source is
        // generated); the lane's own pair is picked with selects instead of an fp64 sincos per use
        constexpr real kb_[16] = QPC_BASIS_TABLE;
        sn = 0.0; cs = 1.0;
        QPC_UNROLL_FULL
        for (int ii = 1; ii < 8; ii++)
          if (ii < N && ii == i) { cs = kb_[2 * ii]; sn = kb_[2 * ii + 1]; }
#else
        QPC_SINCOS(i * (6.283185307179586 / N), sn, cs);
#endif
        real bi[3] = {mu * cs * inv, mu * sn * inv, inv}, lin[3], ang[3];
        mv3(Rc, bi, lin);
        cross3(pc, lin, ang);
        for (int a = 0; a < 3; a++) { sm[D.lds[L_G] + a * QPC_GS(nrho) + j] = ang[a]; sm[D.lds[L_G] + (3 + a) * QPC_GS(nrho) + j] = lin[a]; }
        real cw = sm[D.lds[L_CW] + c];
#ifdef QPC_BASIS_TABLE
        QPC_UNROLL_FULL
        for (int i2 = 0; i2 < 8; i2++) {
          if (i2 < N) {
            const real c2 = kb_[2 * i2], s2 = kb_[2 * i2 + 1];
            real dot = (mu * cs * mu * c2 + mu * sn * mu * s2 + 1.0) * inv * inv;
            sm[D.lds[L_HR] + c * QPC_CSH(N) + i * N + i2] = 2.0 * cw * dot;
          }
        }
#else
        for (int i2 = 0; i2 < N; i2++) {
          real s2, c2;
          QPC_SINCOS(i2 * (6.283185307179586 / N), s2, c2);
          real dot = (mu * cs * mu * c2 + mu * sn * mu * s2 + 1.0) * inv * inv;
          sm[D.lds[L_HR] + c * QPC_CSH(N) + i * N + i2] = 2.0 * cw * dot;
        }
#endif
        real hi = sm[D.lds[L_CMAX] + c] * inv / N; // maxrho (contacts.jl:57)
        sm[D.lds[L_HI] + j] = hi;
        real ws0 = hi > 0.0 ? 0.0 : -1.0; // disable! (contacts.jl:72): rho pinned to 0
        if (WARM && io.wset && hi > 0.0 && j < 64) { // hint from the previous tick of this robot
          const unsigned long long wlo = io.wset[0], wup = io.wset[1];
          ws0 = ((wlo >> j) & 1ull) ? -1.0 : (((wup >> j) & 1ull) ? 1.0 : 0.0);
        }
        sm[D.lds[L_WS] + j] = ws0;
      }
  } QPC_END_PHASE

  QPC_PROF_ACC(PS_ASM);
  QPC_STOP(2)
  int status = QPC_STATUS_OPTIMAL, iters = 0;
  int refine = 0;
  // Everything from the Schur complement to the accepted solution is written against the dimensions of ITS argument, so that the
  // same code solves the problem as assembled (D) or a reduced one (fewer free columns and rows, other leading dimensions)
  auto solve_part = [&](const QpcDims &D) QPC_LAMBDA_INLINE {
  const int nf = D.nf, nrho = D.nrho, ne = D.ne, m = D.m, m1 = D.m1, nw = D.nw;
  const int MISC = D.lds[L_MISC];
  (void)nf; (void)ne; (void)nw;
  // ======================================================================== Schur complement S = M M' + W^-1 blocks
  // The one GEMM-shaped step of the tick (a12: the J'J small GEMM): lower-triangular 16x16 tiles of
  // S = M M' on the fp64 matrix core (v_mfma_f64_16x16x4_f64), operands straight from the LDS copy of
  // M (zero padded to multiples of 16 x 4), accumulators in registers, written once to LDS.
#ifdef QPC_NO_MFMA
  // (two QPs per wavefront -- QPC_PACK2 -- have no full wave for a matrix-core tile: the dot-product form, one entry of the lower
  //  triangle per lane and trip)
  QPC_PHASE(lane) {
    for (int e = lane; e < m * (m + 1) / 2; e += 64) {
      int r = 0, c = e;
      while (c > r) { r++; c -= r; }
      real v = dotp(sm + D.lds[L_M] + r * D.nfs, 1, sm + D.lds[L_M] + c * D.nfs, 1, nf);
      if (ne > 0 && r < m1) {
        const int slr = (int)(real)sm[D.lds[L_HOT] + r], slc = (int)(real)sm[D.lds[L_HOT] + c];
        if (slr >= 0 && slc >= 0) {
          const int s0r = (int)(real)sm[D.lds[L_HOT] + m + 3 * ne + slr], s0c = (int)(real)sm[D.lds[L_HOT] + m + 3 * ne + slc];
          if (s0r == s0c) v += sm[D.lds[L_WINV] + (int)(real)sm[D.lds[L_HOT] + m + ne + slr] + (slc - s0c)];
        }
      }
      sm[D.lds[L_SM] + r * D.ms + c] = v;
    }
  } QPC_END_PHASE
#else
  {
    const int nt = (m + 15) / 16, nk = (nf + 3) / 4;
    QPC_LANE_ARRAY(sacc_, 4 * QPC_NTILE_MAX);
    QPC_PHASE(lane) {
      real *acc = QPC_LA(sacc_, lane);
      for (int e = 0; e < 4 * QPC_NTILE_MAX; e++) acc[e] = 0.0;
    } QPC_END_PHASE
    {
      int tile = 0;
      QPC_UNROLL_JIT
      for (int ti = 0; ti < nt; ti++) {
        QPC_UNROLL_JIT
        for (int tj = 0; tj <= ti; tj++, tile++) {
          QPC_UNROLL_JIT
          for (int ks = 0; ks < nk; ks++) {
            QPC_MFMA_16x16x4(lane, QPC_LA(sacc_, lane) + 4 * tile,
                                 ((16 * ti + (lane & 15) < m && 4 * ks + (lane >> 4) < nf)
                                      ? (real)sm[D.lds[L_M] + (16 * ti + (lane & 15)) * D.nfs + 4 * ks + (lane >> 4)] : 0.0),
                                 ((16 * tj + (lane & 15) < m && 4 * ks + (lane >> 4) < nf)
                                      ? (real)sm[D.lds[L_M] + (16 * tj + (lane & 15)) * D.nfs + 4 * ks + (lane >> 4)] : 0.0))
          }
        }
      }
    }
    QPC_PHASE(lane) {
      const real *acc = QPC_LA(sacc_, lane);
      int tile = 0;
      QPC_UNROLL_JIT
      for (int ti = 0; ti < nt; ti++) {
        QPC_UNROLL_JIT
        for (int tj = 0; tj <= ti; tj++, tile++) {
          QPC_UNROLL_FULL
          for (int reg = 0; reg < 4; reg++) {
            const int r = 16 * ti + QPC_MFMA_CROW(lane, reg), c = 16 * tj + (lane & 15);
            if (r < m && c <= r) {
              real v = acc[4 * tile + reg];
              if (ne > 0 && r < m1) { // slack rows of the same soft task: + (W+eps)^-1 entry (LDS hot table)
                const int slr = (int)(real)sm[D.lds[L_HOT] + r], slc = (int)(real)sm[D.lds[L_HOT] + c];
                if (slr >= 0 && slc >= 0) {
                  const int s0r = (int)(real)sm[D.lds[L_HOT] + m + 3 * ne + slr], s0c = (int)(real)sm[D.lds[L_HOT] + m + 3 * ne + slc];
                  if (s0r == s0c) v += sm[D.lds[L_WINV] + (int)(real)sm[D.lds[L_HOT] + m + ne + slr] + (slc - s0c)];
                }
              }
              sm[D.lds[L_SM] + r * D.ms + c] = v;
            }
          }
        }
      }
    } QPC_END_PHASE
  }
#endif
  QPC_PROF_ACC(PS_SBUILD);
  QPC_STOP(3)
  // left-looking Cholesky of the leading m1 x m1 block (+ the L21 rows), infinite-pivot rule.
  // lane = row; the row's own factor entries stay in registers (constant indices once the k loop
  // is unrolled by the JIT specialisation), row k is broadcast from LDS.
  // X = L11^-1 rides along for free: the forward substitutions L x = e_c are the SAME recurrence applied to identity
  // rows, so lanes m .. m + m1 - 1 carry one column of X each through the k phases (more active lanes, not more
  // instructions) instead of a separate phase of m1 serial steps afterwards (it was 5 % of the tick).
#ifdef QPC_NO_XAUG
  const bool xaug = false;
#else
  const bool xaug = m + m1 <= 64;
#endif
  {
    QPC_LANE_ARRAY(lrow_, QPC_M1_MAX); // lane-private across the k phases
    QPC_UNROLL_JIT
    for (int k = 0; k < m1; k++) {
      QPC_PHASE(lane) {
        const bool idrow = xaug && lane >= m && lane < m + m1;
        if ((lane >= k && lane < m) || idrow) { // m <= 64 (checked by the host compiler)
          const int r = lane;
          real *lrow = QPC_LA(lrow_, lane);
          real accr = 0.0, acck = 0.0;
          for (int j0 = 0; j0 < k; j0 += 8) {
            real lk[8];
            QPC_UNROLL_FULL
            for (int u = 0; u < 8; u++) lk[u] = sm[D.lds[L_SM] + k * D.ms + (j0 + u < k ? j0 + u : 0)];
            QPC_UNROLL_FULL
            for (int u = 0; u < 8; u++)
              if (j0 + u < k) { accr += lk[u] * lrow[j0 + u]; acck += lk[u] * lk[u]; }
          }
          real skk = sm[D.lds[L_SM] + k * D.ms + k], dk = skk - acck;
          const real srk = idrow ? ((r - m) == k ? 1.0 : 0.0) : (real)sm[D.lds[L_SM] + (idrow ? k : r) * D.ms + k];
          bool ok = dk > (real)D.tau_pivot * (skk > QPC_TINY ? skk : QPC_TINY);
          const real inv = ok ? QPC_RSQRT(dk) : 0.0; // 1 / L_kk
          if (r == k) sm[D.lds[L_T1] + k] = inv; // 1/L_kk (0 = dropped row)
          else {
            const real v = (srk - accr) * inv;
            lrow[k] = v;
            if (idrow) sm[D.lds[L_X] + k * D.m1s + (r - m)] = v; // X[k][c], c = r - m (zero above the diagonal by itself)
            else sm[D.lds[L_SM] + r * D.ms + k] = v;
          }
        }
      } QPC_END_PHASE
    }
  }
  QPC_PROF_ACC(PS_CHOL);
  QPC_STOP(4)
  QPC_PHASE(lane) { // B22 = S22 - L21 L21' ; X = L11^-1 (lane = column, forward substitution in registers)
    if (lane < 36 && nw) {
      int a = lane / 6, b = lane % 6, hi_ = a > b ? a : b, lo_ = a > b ? b : a;
      sm[D.lds[L_B22] + lane] = sm[D.lds[L_SM] + (m1 + hi_) * D.ms + (m1 + lo_)] -
                                dotp(sm + D.lds[L_SM] + (m1 + a) * D.ms, 1, sm + D.lds[L_SM] + (m1 + b) * D.ms, 1, m1);
    }
    if (!xaug && lane < m1) { // (more than 64 rows in the augmented factorization: X by forward substitution)
      const int c = lane;
      real xc[QPC_M1_MAX];
      QPC_UNROLL_JIT
      for (int r = 0; r < m1; r++) {
        real t = (r == c) ? 1.0 : 0.0;
        for (int j0 = 0; j0 < r; j0 += 8) {
          real lr[8];
          QPC_UNROLL_FULL
          for (int u = 0; u < 8; u++) lr[u] = sm[D.lds[L_SM] + r * D.ms + (j0 + u < r ? j0 + u : 0)];
          QPC_UNROLL_FULL
          for (int u = 0; u < 8; u++)
            if (j0 + u < r) t -= lr[u] * xc[j0 + u];
        }
        real x = (r >= c) ? t * sm[D.lds[L_T1] + r] : 0.0;
        xc[r] = x;
        sm[D.lds[L_X] + r * D.m1s + c] = x;
      }
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_X_X);
  // S11I = X'X into the (now free) leading block of L_SM: the second GEMM-shaped step of the tick, on the
  // fp64 matrix core like S = M M' above -- lower-triangular 16x16 tiles, operands straight from the LDS
  // copy of X (lower triangular: tile row ti only needs k >= 16 ti), two LDS reads per lane and k-step
  // instead of the ~2 x 13 per entry of the dot-product form.  Z = X' L21' = S11^-1 S12 is one more column of
  // tiles of the same product, X' [X | L21'] (it used to be 6 m1 strided dot products in the write-out phase).
#ifdef QPC_NO_MFMA
  if (m1 > 0) { // dot-product form (see the Schur build): S11I[r][c] = sum_k X[k][r] X[k][c], Z[r][a] = sum_k X[k][r] L21[a][k]
    QPC_PHASE(lane) {
      for (int e = lane; e < m1 * (m1 + 1) / 2 + (nw ? m1 * 6 : 0); e += 64) {
        if (e < m1 * (m1 + 1) / 2) {
          int r = 0, c = e;
          while (c > r) { r++; c -= r; }
          real acc = 0.0;
          for (int k = r; k < m1; k++) acc += sm[D.lds[L_X] + k * D.m1s + r] * sm[D.lds[L_X] + k * D.m1s + c]; // (X is lower triangular: k >= r >= c)
          sm[D.lds[L_SM] + r * D.ms + c] = acc; // (the leading block of L_SM -- L11, dead by now; the Z entries read the L21 rows below it)
          if (c < r) sm[D.lds[L_SM] + c * D.ms + r] = acc;
        } else {
          const int z = e - m1 * (m1 + 1) / 2, r = z / 6, a = z % 6;
          real acc = 0.0;
          for (int k = r; k < m1; k++) acc += sm[D.lds[L_X] + k * D.m1s + r] * sm[D.lds[L_SM] + (m1 + a) * D.ms + k];
          sm[D.lds[L_X2] + r * 6 + a] = acc;
        }
      }
    } QPC_END_PHASE
  }
#else
  if (m1 > 0) { // (no leading block -- the reduced problem of the presolve with its soft rows folded --: nothing to invert)
    const int nt1 = (m1 + 15) / 16, nk1 = (m1 + 3) / 4, ntri = nt1 * (nt1 + 1) / 2;
    QPC_LANE_ARRAY(xacc_, 4 * QPC_XTILE_MAX);
    QPC_PHASE(lane) {
      real *acc = QPC_LA(xacc_, lane);
      for (int e = 0; e < 4 * QPC_XTILE_MAX; e++) acc[e] = 0.0;
    } QPC_END_PHASE
    {
      int tile = 0;
      QPC_UNROLL_JIT
      for (int ti = 0; ti < nt1; ti++) {
        QPC_UNROLL_JIT
        for (int tj = 0; tj <= ti; tj++, tile++) {
          QPC_UNROLL_JIT
          for (int ks = 4 * ti; ks < nk1; ks++) {
            QPC_MFMA_16x16x4(lane, QPC_LA(xacc_, lane) + 4 * tile,
                                 ((16 * ti + (lane & 15) < m1 && 4 * ks + (lane >> 4) < m1)
                                      ? (real)sm[D.lds[L_X] + (4 * ks + (lane >> 4)) * D.m1s + 16 * ti + (lane & 15)] : 0.0),
                                 ((16 * tj + (lane & 15) < m1 && 4 * ks + (lane >> 4) < m1)
                                      ? (real)sm[D.lds[L_X] + (4 * ks + (lane >> 4)) * D.m1s + 16 * tj + (lane & 15)] : 0.0))
          }
        }
      }
      if (nw) { // Z tiles: rows 16 ti .. of X' times the six columns L21'[k][a] = L21[a][k]
        QPC_UNROLL_JIT
        for (int ti = 0; ti < nt1; ti++) {
          QPC_UNROLL_JIT
          for (int ks = 4 * ti; ks < nk1; ks++) {
            QPC_MFMA_16x16x4(lane, QPC_LA(xacc_, lane) + 4 * (ntri + ti),
                                 ((16 * ti + (lane & 15) < m1 && 4 * ks + (lane >> 4) < m1)
                                      ? (real)sm[D.lds[L_X] + (4 * ks + (lane >> 4)) * D.m1s + 16 * ti + (lane & 15)] : 0.0),
                                 (((lane & 15) < 6 && 4 * ks + (lane >> 4) < m1)
                                      ? (real)sm[D.lds[L_SM] + (m1 + (lane & 15)) * D.ms + 4 * ks + (lane >> 4)] : 0.0))
          }
        }
      }
    }
    QPC_PROF_ACC(PS_X_MFMA);
    QPC_PHASE(lane) {
      const real *acc = QPC_LA(xacc_, lane);
      int tile = 0;
      QPC_UNROLL_JIT
      for (int ti = 0; ti < nt1; ti++) {
        QPC_UNROLL_JIT
        for (int tj = 0; tj <= ti; tj++, tile++) {
          QPC_UNROLL_FULL
          for (int reg = 0; reg < 4; reg++) {
            const int r = 16 * ti + QPC_MFMA_CROW(lane, reg), c = 16 * tj + (lane & 15);
            if (r < m1 && c <= r) {
              const real v = acc[4 * tile + reg];
              sm[D.lds[L_SM] + r * D.ms + c] = v;
              if (c < r) sm[D.lds[L_SM] + c * D.ms + r] = v;
            }
          }
        }
      }
      if (nw) {
        QPC_UNROLL_JIT
        for (int ti = 0; ti < nt1; ti++) {
          QPC_UNROLL_FULL
          for (int reg = 0; reg < 4; reg++) {
            const int r = 16 * ti + QPC_MFMA_CROW(lane, reg), a = lane & 15;
            if (r < m1 && a < 6) sm[D.lds[L_X2] + r * 6 + a] = acc[4 * (ntri + ti) + reg];
          }
        }
      }
    } QPC_END_PHASE
  }
#endif
  QPC_PROF_ACC(PS_XINV);
  QPC_STOP(5)
  // ======================================================================== active-set solve (a13)
  Solver<P> S(D, IT, DT, sm, PA, PT);
  bool warm_hint = false; // WARM: the search below starts from a non-trivial working set
  for (int attempt = 0; attempt < (WARM ? 2 : 1); attempt++) {
  if (WARM && attempt == 1) {
    // the warm-started search failed: once more from the cold start (all enabled rho free)
    if (!warm_hint || status == QPC_STATUS_OPTIMAL || status == QPC_STATUS_ALMOST_OPTIMAL) break;
    QPC_PHASE(lane) {
      for (int j = lane; j < nrho; j += 64) sm[D.lds[L_WS] + j] = sm[D.lds[L_HI] + j] > 0.0 ? 0.0 : -1.0;
    } QPC_END_PHASE
    status = QPC_STATUS_OPTIMAL;
    warm_hint = false;
  }
  if (use_rho) {
    // u0 = c2 - Z' c1 with c = E H^-1 (-g) - d : fixed for the whole tick (see Solver::rho_solve)
    S.compute_c();
    QPC_PHASE(lane) {
      if (lane < 6) sm[MISC + MS_U0 + lane] = sm[D.lds[L_T1] + m1 + lane] - dotp(sm + D.lds[L_X2] + lane, 6, sm + D.lds[L_T1], 1, m1);
    } QPC_END_PHASE
    bool any_upper = false;
    bool pd_converged = false; // the last primal-dual test found nothing to change: the working set is optimal at D.tol_bound
    if (WARM && attempt == 0) { // upper-active bounds / any hint at all in the starting set?
      unsigned long long bup0 = 0ull, bhint = 0ull;
      QPC_PHASE(lane) {
        real up = 0.0, hint = 0.0;
        for (int j = lane; j < nrho; j += 64) {
          const real w = sm[D.lds[L_WS] + j];
          if (w == 1.0) up = 1.0;
          if (w != 0.0 && sm[D.lds[L_HI] + j] > 0.0) hint = 1.0;
        }
        QPC_BALLOT_PUT(0, bup0, up != 0.0);
        QPC_BALLOT_PUT(1, bhint, hint != 0.0);
      } QPC_END_PHASE
      QPC_BALLOT_GET(0, bup0);
      QPC_BALLOT_GET(1, bhint);
      any_upper = bup0 != 0ull;
      warm_hint = bhint != 0ull;
    }
    QPC_PROF_ACC(PS_L_U0);
    S.refactor_solve(any_upper);
    // ---- phase A: primal-dual active-set sweeps (multi-change)
    const bool fsearch = S.fused_ok();
    bool soft_tail = false; // the last sweep ended on a SOFT phase end: a hard one is due before other lanes' words are read
    for (int pd = 0; pd < D.max_pdas; pd++) {
      const real gscale = 1.0 + S.last_gmax, tolb = (real)D.tol_bound * (1.0 + S.last_xmax); // U is current (rho_solve)
      real rchg = 0.0, rup = 0.0;
      unsigned long long bchg = 0ull, bup = 0ull, bfree = 0ull;
      QPC_PHASE(lane) {
        real changed = 0.0, up = 0.0;
        bool isfree = false; // this lane's ray is free in the NEW working set (fused sweeps: one ray per lane)
        for (int j = lane; j < nrho; j += 64) {
          real w = sm[D.lds[L_WS] + j], hi = sm[D.lds[L_HI] + j], x = sm[D.lds[L_XR] + j], nw_ = w;
          sm[D.lds[L_WSP] + j] = w; // the set this solution belongs to (fallback, see below)
          if (!(hi > 0.0)) continue; // disabled
          if (w != 0.0) { if (sm[D.lds[L_U] + j] < -(real)D.tol_bound * gscale) nw_ = 0.0; }
          else if (x < -tolb) nw_ = -1.0;
          else if (x > hi + tolb) nw_ = 1.0;
          if (nw_ != w) { sm[D.lds[L_WS] + j] = nw_; changed = 1.0; }
          if (nw_ == 1.0) up = 1.0;
          isfree = nw_ == 0.0;
        }
        QPC_BALLOT_PUT(0, bchg, changed != 0.0); // flags: one compare each instead of a 6-step max reduction
        QPC_BALLOT_PUT(1, bup, up != 0.0);
        if (fsearch) { QPC_BALLOT_PUT(2, bfree, isfree); }
      } QPC_END_PHASE_SOFT
      if (!fsearch) { QPC_PHASE_SYNC }
      QPC_BALLOT_GET(0, bchg);
      QPC_BALLOT_GET(1, bup);
      if (fsearch) { QPC_BALLOT_GET(2, bfree); }
      rchg = bchg ? 1.0 : 0.0;
      rup = bup ? 1.0 : 0.0;
      QPC_PROF_ACC(PS_S_TEST);
      if (!(rchg > 0.0)) { pd_converged = true; if (fsearch) soft_tail = true; break; } // (the test phase itself ended soft)
      any_upper = rup > 0.0;
      iters++;
      // fused sweeps: the test above, the K rows / Y of the new set (free pattern from the ballot, not from LDS) and the solve are
      // joined by SOFT phase ends -- only Y -> S22 and S22 -> lam_w cross lanes through LDS
      if (fsearch && !any_upper) { S.ky_phase(true, bfree); S.fused_solve(); soft_tail = true; }
      else { if (fsearch) { QPC_PHASE_SYNC } S.refactor_solve(any_upper); soft_tail = false; }
    }
    if (soft_tail) { QPC_PHASE_SYNC }
    // The multi-change sweeps can land on a working set with so many rho's pinned that the wrench
    // rows have no solution on it (the EQP is inconsistent although the QP is feasible); its
    // "solution" is meaningless and phase B, which assumes a valid point, must not start from it.
    // Measured: it happens to ~0.5 % of the QPs and only at the last sweep, so the working set of the
    // sweep before (L_WSP) is tried first; the all-free set -- consistent whenever the QP is feasible
    // at all, but ~30 Goldfarb-Idnani additions away from the optimum -- is the last resort.
    if (!any_upper) {
      for (int ga = 0; ga < 3; ga++) {
        real incons = 0.0; // residual of S22 lam_w = u0: rows the infinite-pivot rule dropped as dependent
                             // must still be satisfied (L_C22 still holds S22 of this working set: fused path)
        QPC_PHASE(lane) {
          real r = 0.0;
          if (lane < 6) {
            const real u = sm[MISC + MS_U0 + lane];
            r = QPC_FABS(u - dotp(sm + D.lds[L_C22] + lane * 6, 1, sm + D.lds[L_LAM] + m1, 1, 6)) / (1.0 + QPC_FABS(u));
          }
          QPC_RED_MAX_PUT(0, incons, r);
        } QPC_END_PHASE
        incons = QPC_RED_MAX_GET(0, incons);
        if (!(incons > QPC_T_INCONS) || ga == 2) break;
        QPC_PHASE(lane) {
          for (int j = lane; j < nrho; j += 64)
            sm[D.lds[L_WS] + j] = ga == 0 ? (real)sm[D.lds[L_WSP] + j] : (sm[D.lds[L_HI] + j] > 0.0 ? 0.0 : -1.0);
        } QPC_END_PHASE
        iters++;
        pd_converged = false; // another working set: phase B has to judge it
        S.refactor_solve(false);
      }
    }
    QPC_PROF_ACC(PS_L_GUARD);
    // ---- phase B: drops until dual feasible, then Goldfarb-Idnani additions (finite in exact
    //      arithmetic: no working set is visited twice).  A repeated working set therefore means the
    //      iteration is cycling on rounding noise (seen on infeasible single-contact QPs, which used to
    //      burn all max_active_set_iters): recognised from the bit masks of the last 8 working sets.
    // (skipped when the last primal-dual test found nothing to change: its conditions -- no multiplier below
    //  -tol_bound * gscale, no rho outside its bounds by more than tolb -- are exactly phase B's stopping test)
    if (!pd_converged) {
    const bool track_b = nrho <= 64; // one rho per lane: two ballots ARE the working set
    QPC_LANE_INTS(hist_, 4); // lane h < 8 remembers one of the last 8 working sets: lo / up masks as 2 x 2 ints (four VGPRs
                             // instead of 32 scalar registers of wave-uniform history)
    QPC_PHASE(lane) {
      int *hs = QPC_LI(hist_, lane);
      hs[0] = hs[1] = hs[2] = hs[3] = -1;
    } QPC_END_PHASE
    int hpos = 0;
    int direct_adds = 0;
    bool hist_cleared = false;
    (void)direct_adds; (void)hist_cleared;
    while (true) {
      if (iters > D.max_as_iters) { status = QPC_STATUS_ITERATION_LIMIT; break; }
      const real gscale = 1.0 + S.last_gmax, tolb = (real)D.tol_bound * (1.0 + S.last_xmax); // U is current (rho_solve)
      // most negative multiplier / most violated bound (lane-parallel + wave argmin)
      real ubad = 0.0, vneg = 0.0, rnan = 0.0;
      int jbad = -1, pviol = -1;
      unsigned long long lo_now = 0ull, up_now = 0ull, bnan = 0ull;
      QPC_PHASE(lane) {
        real bu = QPC_HUGE, bv = QPC_HUGE, bad = 0.0;
        int ju = 0x3fffffff, jv = 0x3fffffff;
        for (int j = lane; j < nrho; j += 64) {
          real w = sm[D.lds[L_WS] + j], hi = sm[D.lds[L_HI] + j], x = sm[D.lds[L_XR] + j];
          if (!(x == x) || QPC_FABS(x) > QPC_HUGE) bad = 1.0;
          if (!(hi > 0.0)) continue;
          if (w != 0.0) { real u = sm[D.lds[L_U] + j]; if (u < bu) { bu = u; ju = j; } }
          else { real vl = -x > x - hi ? -x : x - hi; if (-vl < bv) { bv = -vl; jv = j; } }
        }
        // (at most one rho per lane: the index of a candidate IS its lane, so the argmin is a minimum and a ballot)
        if (nrho <= 64) { QPC_RED_ARGMIN_LANE_PUT(0, ubad, jbad, bu, ju); QPC_RED_ARGMIN_LANE_PUT(1, vneg, pviol, bv, jv); }
        else { QPC_RED_ARGMIN_PUT(0, ubad, jbad, bu, ju); QPC_RED_ARGMIN_PUT(1, vneg, pviol, bv, jv); }
        QPC_BALLOT_PUT(2, bnan, bad != 0.0);
        const real wl_ = lane < nrho ? (real)sm[D.lds[L_WS] + (lane < nrho ? lane : 0)] : 0.0;
        QPC_BALLOT_PUT(3, lo_now, wl_ == -1.0);
        QPC_BALLOT_PUT(4, up_now, wl_ == 1.0);
      } QPC_END_PHASE
      QPC_RED_ARGMIN_GET(0, ubad, jbad);
      QPC_RED_ARGMIN_GET(1, vneg, pviol);
      QPC_BALLOT_GET(2, bnan);
      rnan = bnan ? 1.0 : 0.0;
      QPC_BALLOT_GET(3, lo_now);
      QPC_BALLOT_GET(4, up_now);
      if (track_b && iters >= 6) { // (ordinary QPs are done long before: the history only matters for the stragglers)
        unsigned long long bseen = 0ull;
        const int l0 = (int)(unsigned)lo_now, l1 = (int)(unsigned)(lo_now >> 32), u0_ = (int)(unsigned)up_now, u1 = (int)(unsigned)(up_now >> 32);
        QPC_PHASE(lane) {
          int *hs = QPC_LI(hist_, lane);
          QPC_BALLOT_PUT(5, bseen, lane < 8 && hs[0] == l0 && hs[1] == l1 && hs[2] == u0_ && hs[3] == u1);
          if (lane == hpos) { hs[0] = l0; hs[1] = l1; hs[2] = u0_; hs[3] = u1; }
        } QPC_END_PHASE
        QPC_BALLOT_GET(5, bseen);
        // (ordinary QPs are done long before iteration 10: a repeat only counts for the stragglers)
        if (iters >= 10 && bseen != 0ull) { status = QPC_STATUS_ITERATION_LIMIT; break; }
        hpos = (hpos + 1) & 7;
      }
      QPC_PROF_ACC(PS_L_GICHECK);
      const bool finite = !(rnan > 0.0);
      if (!(ubad < -(real)D.tol_bound * gscale)) jbad = -1;
      if (!(-vneg > tolb)) pviol = -1;
      if (!finite) { status = QPC_STATUS_NUMERICAL_ERROR; break; }
      if (jbad >= 0) {
        QPC_PHASE(lane) { if (lane == 0) sm[D.lds[L_WS] + jbad] = 0.0; } QPC_END_PHASE
        iters++;
        S.refactor_solve(any_upper); // dropping a bound cannot create an upper-active one
        continue;
      }
      if (pviol < 0) break; // optimal for this working set
      const int p = pviol;
      const real xp0 = sm[D.lds[L_XR] + p], hip = sm[D.lds[L_HI] + p];
      const real side = (-xp0 > xp0 - hip) ? -1.0 : 1.0, nsign = side < 0 ? 1.0 : -1.0;
#if QPC_DIRECT_ADDS > 0
      // A few DIRECT additions first: pin the most violated bound and re-solve the EQP (one sweep) instead of the
      // Goldfarb-Idnani step (direction solve + ratio test + partial / full step + sweep: about twice the work).  The
      // stopping test above is the same KKT test either way, so the answer cannot depend on the path; a direct addition
      // can make multipliers negative (they are dropped by the next pass of this loop) and could cycle, hence the budget --
      // afterwards, and for the stragglers, the finite Goldfarb-Idnani iteration takes over.  Measured (MI355X, config 3): budget
      // 0: 26.6, 2: 27.1, 4: 27.0, 8: 27.3 M solves/s, but from 4 on a few QPs in 10^4 end on the cycle guard instead of
      // their optimum; 2 keeps every status of configs 3 and 5 (one infeasible QP in 2 048 reports ITERATION_LIMIT instead).
      if (direct_adds < QPC_DIRECT_ADDS && side < 0.0 && !any_upper) {
        direct_adds++;
        QPC_PHASE(lane) { if (lane == 0) sm[D.lds[L_WS] + p] = side; } QPC_END_PHASE
        iters++;
        S.refactor_solve(false);
        // the pin can leave the wrench rows without a solution on the new set (same test as after phase A): the
        // Goldfarb-Idnani step would have seen the dependence and dropped a bound first, so undo and leave it to that
        real incons = 0.0;
        QPC_PHASE(lane) {
          real r = 0.0;
          if (lane < 6) {
            const real u = sm[MISC + MS_U0 + lane];
            r = QPC_FABS(u - dotp(sm + D.lds[L_C22] + lane * 6, 1, sm + D.lds[L_LAM] + m1, 1, 6)) / (1.0 + QPC_FABS(u));
          }
          QPC_RED_MAX_PUT(0, incons, r);
        } QPC_END_PHASE
        incons = QPC_RED_MAX_GET(0, incons);
        if (incons > QPC_T_INCONS) {
          direct_adds = QPC_DIRECT_ADDS;
          hist_cleared = true; // (the set before the pin comes back: not a cycle)
          QPC_PHASE(lane) {
            if (lane == 0) sm[D.lds[L_WS] + p] = 0.0;
            int *hs = QPC_LI(hist_, lane);
            hs[0] = hs[1] = hs[2] = hs[3] = -1;
          } QPC_END_PHASE
          iters++;
          S.refactor_solve(false);
        }
        continue;
      }
      if (!hist_cleared) { // sets seen during the direct additions may legitimately come back once: they are not cycles
        hist_cleared = true;
        QPC_PHASE(lane) {
          int *hs = QPC_LI(hist_, lane);
          hs[0] = hs[1] = hs[2] = hs[3] = -1;
        } QPC_END_PHASE
      }
#endif
      bool fail = false;
      while (true) {
        iters++;
        if (iters > D.max_as_iters) { fail = true; break; }
        S.rho_solve(false, p, nsign, true);
        real t1 = QPC_HUGE;
        int jdrop = -1;
        QPC_PHASE(lane) { // dual direction r_j = -s_j (H_eff z)_j on the working set; step ratios
          real best = QPC_HUGE;
          real bestj = -1.0;
          for (int j = lane; j < nrho; j += 64) {
            int c = j / N, jj = j % N;
            real w = sm[D.lds[L_WS] + j];
            real rj = 0.0;
            if (w != 0.0 && sm[D.lds[L_HI] + j] > 0.0) {
              real hz = eps * sm[D.lds[L_DR] + j];
              for (int i = 0; i < N; i++) hz += sm[D.lds[L_HR] + c * QPC_CSH(N) + jj * N + i] * sm[D.lds[L_DR] + c * N + i];
              for (int a = 0; a < 6; a++) hz -= sm[D.lds[L_G] + a * QPC_GS(nrho) + j] * sm[D.lds[L_DL] + m1 + a];
              rj = (w == -1.0) ? -hz : hz;
              if (rj > QPC_TINY) { real ratio = sm[D.lds[L_U] + j] / rj; if (ratio < best) { best = ratio; bestj = (real)j; } }
            }
            sm[D.lds[L_TR2] + j] = rj;
          }
          if (nrho <= 64) { QPC_RED_ARGMIN_LANE_PUT(0, t1, jdrop, best, bestj < 0.0 ? 0x3fffffff : (int)bestj); }
          else { QPC_RED_ARGMIN_PUT(0, t1, jdrop, best, bestj < 0.0 ? 0x3fffffff : (int)bestj); }
        } QPC_END_PHASE
        QPC_RED_ARGMIN_GET(0, t1, jdrop);
        if (!(t1 < QPC_HUGE_TEST)) jdrop = -1;
        const real xp = sm[D.lds[L_XR] + p];
        const real sp = nsign * xp - (side < 0 ? 0.0 : -hip); // < 0 : violation
        const real zn = nsign * sm[D.lds[L_DR] + p];
        const real t2 = zn > QPC_TINY ? -sp / zn : QPC_HUGE;
        const real tt = t1 < t2 ? t1 : t2;
        if (!(tt < QPC_HUGE_TEST)) {
          status = QPC_STATUS_INFEASIBLE; fail = true; break; }
        const bool full = t2 <= t1;
        QPC_PHASE(lane) { // the step moves rho and the wrench multipliers only (see rho_solve)
          if (lane < 6) sm[D.lds[L_LAM] + m1 + lane] = sm[D.lds[L_LAM] + m1 + lane] + tt * sm[D.lds[L_DL] + m1 + lane];
          for (int j = lane; j < nrho; j += 64) {
            sm[D.lds[L_XR] + j] = sm[D.lds[L_XR] + j] + tt * sm[D.lds[L_DR] + j];
            real u = sm[D.lds[L_U] + j] - tt * sm[D.lds[L_TR2] + j];
            if (!full && j == jdrop) { u = 0.0; sm[D.lds[L_WS] + j] = 0.0; }
            if (full && j == p) sm[D.lds[L_WS] + j] = side;
            sm[D.lds[L_U] + j] = u;
          }
        } QPC_END_PHASE
        QPC_PROF_ACC(PS_L_GISTEP);
        if (full) { any_upper = any_upper || side > 0.0; S.refactor_solve(any_upper); break; }
        S.refactor_rho();
      }
      if (fail) { if (status == QPC_STATUS_OPTIMAL) status = QPC_STATUS_ITERATION_LIMIT; break; }
    }
    } // !pd_converged
  } else {
    S.refactor_rho();
  }
  QPC_PROF_ACC(PS_LOGIC);
  QPC_STOP(6)
  // ---- final refinement on the converged working set + acceptance test (a14)
  refine = 0;
  if (status == QPC_STATUS_OPTIMAL) {
    for (int outer = 0;; outer++) {
      // exact_ve: wrench-space proximal sweeps + one back-substitution (Solver::polish_exact), then
      // refinement of the full KKT system from that iterate -- one sweep when its correction confirms
      // that the iterate was already accurate; otherwise: solve + refine the full system from scratch
#ifdef QPC_EXP_NOCONFIRM
      if (exact_ve) { refine = S.polish_exact(true); S.residuals(); S.confirmed = true; }
#else
      if (exact_ve) { refine = S.polish_exact(true); refine += S.eqp(D.max_refine, true) - 1; }
#endif
      else refine = S.eqp(D.max_refine);
      // acceptance needs the true KKT residuals of the final iterate.  When the single confirming sweep changed the
      // iterate by <= 1e-9 relative, the residuals it started from (still in L_RV / L_RE / L_RR / L_RC, gradient in
      // L_WSP) are those of an iterate 1e-9 away -- far inside the 1e-7 acceptance tolerance -- and are not recomputed
      if (!S.confirmed) S.residuals();
      real arp = 0.0, ars = 0.0, asp = 0.0, ass = 0.0, anan = 0.0, abv = 0.0, agm = 0.0, axm = 0.0;
      unsigned long long bbad = 0ull, bchanged = 0ull;
      QPC_PHASE(lane) {
        real rp = 0.0, rs = 0.0, sp = 0.0, ss = 0.0, bv = 0.0, gm = 0.0, xm = 0.0;
        for (int r = lane; r < m; r += 64) {
          real a = QPC_FABS(sm[D.lds[L_RC] + r]); rp = a > rp || !(a == a) ? a : rp;
          real b = QPC_FABS(sm[D.lds[L_D] + r]); sp = b > sp ? b : sp;
        }
        for (int f = lane; f < nf; f += 64) {
          real a = QPC_FABS(sm[D.lds[L_RV] + f]); rs = a > rs || !(a == a) ? a : rs;
          real b = QPC_FABS(sm[D.lds[L_GV] + f]); ss = b > ss ? b : ss;
          b = sm[D.lds[L_DV] + f]; ss = b > ss ? b : ss; // |E_v' lam| scale, left there by Solver::residuals
        }
        for (int s = lane; s < ne; s += 64) { real a = QPC_FABS(sm[D.lds[L_RE] + s]); rs = a > rs || !(a == a) ? a : rs; }
        for (int j = lane; j < nrho; j += 64) {
          real a = QPC_FABS(sm[D.lds[L_RR] + j]); rs = a > rs || !(a == a) ? a : rs;
          // bound conditions of the REFINED solution (the search decided on eps-regularised solves):
          // L_WSP = Lagrangian gradient w.r.t. rho (Solver::residuals)
          const real w = sm[D.lds[L_WS] + j], hi = sm[D.lds[L_HI] + j], x = sm[D.lds[L_XR] + j], gl = sm[D.lds[L_WSP] + j];
          if (hi > 0.0) {
            const real u = (w == -1.0) ? gl : ((w == 1.0) ? -gl : 0.0);
            sm[D.lds[L_U] + j] = u;
            gm = QPC_FABS(gl) > gm ? QPC_FABS(gl) : gm;
            xm = QPC_FABS(x) > xm ? QPC_FABS(x) : xm;
          }
        }
        real bad = (!(rp == rp) || !(rs == rs)) ? 1.0 : 0.0;
        QPC_RED_MAX_PUT(0, arp, rp == rp ? rp : 0.0); QPC_RED_MAX_PUT(1, ars, rs == rs ? rs : 0.0);
        QPC_RED_MAX_PUT(2, asp, sp); QPC_RED_MAX_PUT(3, ass, ss); QPC_BALLOT_PUT(4, bbad, bad != 0.0);
        QPC_RED_MAX_PUT(5, agm, gm); QPC_RED_MAX_PUT(6, axm, xm);
      } QPC_END_PHASE
      QPC_PROF_ACC(PS_F_ACCEPT);
      arp = QPC_RED_MAX_GET(0, arp); ars = QPC_RED_MAX_GET(1, ars); asp = QPC_RED_MAX_GET(2, asp);
      ass = QPC_RED_MAX_GET(3, ass); QPC_BALLOT_GET(4, bbad); anan = bbad ? 1.0 : 0.0;
      agm = QPC_RED_MAX_GET(5, agm); axm = QPC_RED_MAX_GET(6, axm);
      const bool nan = anan > 0.0;
      real r_prim = arp / (1.0 + asp), r_stat = ars / (1.0 + ass);
      if (nan || !(r_prim < QPC_HUGE) || !(r_stat < QPC_HUGE)) { status = QPC_STATUS_NUMERICAL_ERROR; break; }
      // equality rows inconsistent on the feasible set.  fp32: a residual within QPC_ACCEPT_BAND of the tolerance is what an
      // ill-conditioned but feasible QP leaves at that precision (a robot balanced on ONE contact point: cond(S22) > 1e5, pivots
      // dropped as noise) -- reported as NUMERICAL_ERROR ("not resolved"), not as a verdict on the problem; truly inconsistent rows
      // leave residuals of order one
      if (r_prim > (real)D.tol_accept) { status = r_prim > QPC_ACCEPT_BAND * (real)D.tol_accept ? QPC_STATUS_INFEASIBLE : QPC_STATUS_NUMERICAL_ERROR; break; }
      if (r_stat > (real)D.tol_accept) status = QPC_STATUS_ALMOST_OPTIMAL;

      if (!use_rho) break;
      // primal-dual test on the refined solution, at the acceptance tolerance: normally nothing moves
      // (one phase); if the eps-regularised search settled on a working set that is not optimal for
      // the true problem (near-degenerate friction cones), correct it and polish again
      S.last_gmax = agm; S.last_xmax = axm;
      QPC_PHASE(lane) {
        real changed = 0.0;
        const real tolu = QPC_VERIFY_TOL(D) * (1.0 + agm), tolb = QPC_VERIFY_TOL(D) * (1.0 + axm);
        for (int j = lane; j < nrho; j += 64) {
          real w = sm[D.lds[L_WS] + j], hi = sm[D.lds[L_HI] + j], x = sm[D.lds[L_XR] + j], nw_ = w;
          if (!(hi > 0.0)) continue;
          if (w != 0.0) { if (sm[D.lds[L_U] + j] < -tolu) nw_ = 0.0; }
          else if (x < -tolb) nw_ = -1.0;
          else if (x > hi + tolb) nw_ = 1.0;
          if (nw_ != w) { changed = 1.0; if (outer < 8) sm[D.lds[L_WS] + j] = nw_; }
        }
        QPC_BALLOT_PUT(0, bchanged, changed != 0.0);
      } QPC_END_PHASE
      QPC_BALLOT_GET(0, bchanged);
      abv = bchanged ? 1.0 : 0.0;
      if (!(abv > 0.0)) break;
      if (outer >= 8) { status = QPC_STATUS_ITERATION_LIMIT; break; }
      status = QPC_STATUS_OPTIMAL;
      iters++;
      S.refactor_rho();
    }
  }

  } // attempt (WARM: warm start, then cold start if that failed)
  }; // solve_part
  bool reduced = false; // this QP went through the null-space presolve
  if (D.presolve && exact_ve && qpc_presolve(D, DS, sm, PA, PT)) {
    QPC_PROF_ACC(PS_ASM);
    reduced = true;
    solve_part(DS);
    QPC_PROF_ACC(PS_FINAL);
    const unsigned long long bviol = qpc_postsolve(D, DS, sm);
    if (status == QPC_STATUS_OPTIMAL || status == QPC_STATUS_ALMOST_OPTIMAL) { // an eliminated foot-task row is off by > tol_accept / not a number
      if (bviol == ~0ull) status = QPC_STATUS_NUMERICAL_ERROR;
      else if (bviol != 0ull) status = QPC_STATUS_ALMOST_OPTIMAL;
    }
    QPC_PROF_ACC(PS_P_POST);
  } else {
    solve_part(D);
  }
  if (WARM && io.wset) { // the working set goes back to the caller (zeros = "start cold" after a failed solve)
    unsigned long long blo = 0ull, bupf = 0ull;
    QPC_PHASE(lane) {
      const real w = (lane < nrho) ? (real)sm[D.lds[L_WS] + (lane < nrho ? lane : 0)] : 0.0;
      QPC_BALLOT_PUT(0, blo, w == -1.0);
      QPC_BALLOT_PUT(1, bupf, w == 1.0);
    } QPC_END_PHASE
    QPC_BALLOT_GET(0, blo);
    QPC_BALLOT_GET(1, bupf);
    const bool good = status == QPC_STATUS_OPTIMAL || status == QPC_STATUS_ALMOST_OPTIMAL;
    QPC_PHASE(lane) {
      if (lane == 0) { io.wset[0] = good ? blo : 0ull; io.wset[1] = good ? bupf : 0ull; }
    } QPC_END_PHASE
  }

  QPC_PROF_ACC(PS_FINAL);
  QPC_STOP(7)
  // ======================================================================== extraction (a15) + RNEA (a16, a17)
  QPC_LANE_ARRAY(sk_, 6);  // this lane's motion subspace column, back from the spill slot
  QPC_LANE_ARRAY(iw_, 16); // this lane's body inertia (10) and W0 (6): one HBM round trip for all three
  QPC_PHASE(lane) {
    for (int f = lane; f < nf; f += 64) sm[D.lds[L_VD] + IT[D.it[IT_FREE_K] + f]] = sm[D.lds[L_XV] + f] * sm[D.lds[L_SQH] + f];
    for (int idx = lane; idx < nc * 6; idx += 64) { // value(wrench_world) = G_c rho_c
      int c = idx / 6, a = idx % 6;
      real acc = 0.0;
      if (use_rho) for (int i = 0; i < N; i++) acc += sm[D.lds[L_G] + a * QPC_GS(nrho) + c * N + i] * sm[D.lds[L_XR] + c * N + i];
      sm[D.lds[L_CWR] + idx] = acc;
    }
    if (lane < nb) {
      real *iw = QPC_LA(iw_, lane);
      for (int a = 0; a < 16; a++) iw[a] = QPC_LA(keep_iw_, lane)[a];
    }
    if (lane < nv) { // S_k v̇_k of the FREE velocity indices (the pinned ones already ride in Ab)
      real *sk = QPC_LA(sk_, lane);
      for (int a = 0; a < 6; a++) sk[a] = QPC_LA(keep_sk_, lane)[a];
      const int f = IT[D.it[IT_FREE_OF] + lane];
      const real vd = f >= 0 ? (real)sm[D.lds[L_XV] + f] * (real)sm[D.lds[L_SQH] + f] : 0.0;
      for (int a = 0; a < 6; a++) sm[D.lds[L_SVK] + 6 * lane + a] = sk[a] * vd;
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_R_LOAD);
  QPC_PHASE(lane) { // S_j v̇_j per body
    if (lane < nb) {
      const int *mt = QPC_LI(M.v, lane);
      real sv[6] = {0, 0, 0, 0, 0, 0};
      const int v0 = MT_V0(mt), n = MT_NVJ(mt);
      for (int c = 0; c < n; c++)
        for (int a = 0; a < 6; a++) sv[a] += sm[D.lds[L_SVK] + 6 * (v0 + c) + a];
      st(sm + D.lds[L_SV] + 6 * lane, 6, sv);
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_R_SV);
  QPC_LANE_ARRAY(bwp_, 6); // D.dfs: this body's wrench, then the sum of the wrenches of bodies 0 .. lane
  QPC_PHASE(lane) { // body wrench: I A_free + W0 - external (contact) wrench,  W0 = I (Ab - g) + T x* I T
    for (int a = 0; a < 6; a++) QPC_LA(bwp_, lane)[a] = 0.0;
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
      real A[6], I[10], ia[6];
      ld(sm + D.lds[L_SV] + 6 * i, 6, A);
      QPC_UNROLL_JIT
      for (int s = 0; s < QPC_MAXD; s++) {
        const int an = MT_ANC(mt, s);
        if (an >= 0)
          for (int a = 0; a < 6; a++) A[a] += sm[D.lds[L_SV] + 6 * an + a];
      }
      const real *iw = QPC_LA(iw_, lane);
      for (int a = 0; a < 10; a++) I[a] = iw[a];
      inertia_mul(I, A, ia);
      for (int a = 0; a < 6; a++) ia[a] += iw[10 + a];
      if (nc <= 24) { // contacts of this body from the lane's metadata word: no table look-ups on this path
        unsigned cm = MT_CMASK(mt);
        while (cm) {
          const int c = QPC_CTZ64(cm);
          cm &= cm - 1;
          for (int a = 0; a < 6; a++) ia[a] -= sm[D.lds[L_CWR] + 6 * c + a];
        }
      } else {
        for (int e = IT[D.it[IT_BC_PTR] + i]; e < IT[D.it[IT_BC_PTR] + i + 1]; e++) {
          int c = IT[D.it[IT_BC_IDX] + e];
          for (int a = 0; a < 6; a++) ia[a] -= sm[D.lds[L_CWR] + 6 * c + a];
        }
      }
      if (D.dfs) { for (int a = 0; a < 6; a++) QPC_LA(bwp_, lane)[a] = ia[a]; }
      else st(sm + D.lds[L_BW] + 6 * i, 6, ia);
    }
  } QPC_END_PHASE
  if (D.dfs) { // depth-first numbering: a subtree is an index range, its wrench the difference of two prefix sums
    QPC_WAVE_INCL_SCAN(lane, QPC_LA(bwp_, lane), 6)
    QPC_PHASE(lane) {
      if (lane < nb) st(sm + D.lds[L_BW] + 6 * lane, 6, QPC_LA(bwp_, lane));
    } QPC_END_PHASE
  }
  QPC_PROF_ACC(PS_R_BW);
  QPC_PHASE(lane) { // tau_k = S_k . (sum of subtree wrenches); floating joint torques exactly 0
    if (lane < nv) {
      const int k = lane;
      const int *mt = QPC_LI(M.v, lane);
      real tau = 0.0;
      if (MT_VBODY(mt) != D.floating) {
        real F[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long msk = ((unsigned long long)(unsigned)mt[5] << 32) | (unsigned long long)(unsigned)mt[4];
        if (D.dfs) { // subtree = bodies first .. last: prefix[last] - prefix[first - 1]
          const int first = QPC_CTZ64(msk), last = first + QPC_POPC64(msk) - 1;
          for (int a = 0; a < 6; a++) F[a] = sm[D.lds[L_BW] + 6 * last + a] - (first > 0 ? (real)sm[D.lds[L_BW] + 6 * (first > 0 ? first - 1 : 0) + a] : 0.0);
        } else
        while (msk) {
          const int b = QPC_CTZ64(msk);
          msk &= msk - 1;
          for (int a = 0; a < 6; a++) F[a] += sm[D.lds[L_BW] + 6 * b + a];
        }
        const real *sk = QPC_LA(sk_, lane);
        for (int a = 0; a < 6; a++) tau += sk[a] * F[a];
      }
      io.tau[k] = QPC_TO_DOUBLE(tau);
      io.vdot[k] = QPC_TO_DOUBLE((real)sm[D.lds[L_VD] + k]);
    }
    if (io.cwrench)
      for (int idx = lane; idx < nc * 6; idx += 64) io.cwrench[idx] = QPC_TO_DOUBLE((real)sm[D.lds[L_CWR] + idx]);
    if (lane == 0) {
      const bool good = status == QPC_STATUS_OPTIMAL || status == QPC_STATUS_ALMOST_OPTIMAL;
      int st_ = (good && bbad_in != 0ull) ? QPC_STATUS_NUMERICAL_ERROR : ((good && bneg_cap != 0ull) ? QPC_STATUS_INFEASIBLE : status);
#ifdef QPC_F32
      // fp32: a floating robot balanced on ONE enabled contact point is a QP whose force distribution fp32 cannot resolve (cond(S22) >
      // 1e5; measured on 32 768 config-5 states: accepted points up to 44 % off, p99.9 26 %): never reported as solved
      if (st_ <= QPC_STATUS_ALMOST_OPTIMAL && nw > 0 && nc <= 64 && QPC_POPC64(ben_) == 1) st_ = QPC_STATUS_NUMERICAL_ERROR;
#endif
      io.status[0] = st_;
      if (io.iters) io.iters[0] = iters + 1000 * refine + (reduced ? 1000000 : 0);
    }
  } QPC_END_PHASE
  QPC_PROF_ACC(PS_RNEA);
#ifdef QPC_STAMPS
  if (io.prof && (threadIdx.x & 63) == 0)
    for (int i = 0; i < PS_COUNT; i++) io.prof[i] = prof_acc_[i];
#endif
}

} // namespace qpc
