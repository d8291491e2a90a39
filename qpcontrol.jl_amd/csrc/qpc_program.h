// qpc_program.h -- the compiled "task program" shared by the host compiler (qpc_compile.cpp)
// and the device code (qpc_tick.h): a POD header of sizes + offsets into two flat tables (ints
// and doubles) that live in HBM, plus the per-QP LDS layout.
//
// The program is what the reference builds lazily as Parametron Parameters / LazyExpressions at
// addtask!/addcontact!/initialize! time (momentum.jl:99-193); here it is frozen at qpc_finalize.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

// integer tables (offsets into the int blob)
enum {
  IT_PARENT, IT_JTYPE, IT_QOFF, IT_VOFF, IT_NVJ,
  IT_ANC_PTR, IT_ANC_IDX,   // ancestors of body i, nearest first (CSR)
  IT_DESC_PTR, IT_DESC_IDX, // descendants of body i including i (CSR)
  IT_VBODY, IT_VCOMP,       // velocity index -> body, component within the joint
  IT_FREE_OF,               // velocity index -> free index or -1 (pinned)
  IT_FREE_K,                // free index -> velocity index
  IT_PIN_K, IT_PIN_DES,     // pinned: velocity index, index into the desired record
  IT_SJ_F, IT_SJ_DES, IT_SJ_TW, // folded soft joint tasks: free index, desired index, tw index or -1
  IT_T_KIND, IT_T_BASE, IT_T_TARGET, IT_T_FRAME, IT_T_DIM, IT_T_ROW0, IT_T_DES, IT_T_WKIND,
  IT_T_TW, IT_T_SLACK0, IT_T_WOFF, // generic tasks
  IT_T_SGN,                 // [ntask*nv] path sign of velocity index k for task t (-1,0,+1)
  IT_ROW_TASK, IT_ROW_COMP, // row -> generic task (or -1 for wrench rows), component
  IT_C_BODY, IT_C_PERQP, IT_C_SURF, // contacts
  IT_BC_PTR, IT_BC_IDX,     // contacts attached to body i (CSR)
  IT_SLACK_TASK, IT_SLACK_COMP, // slack index -> generic task, component
  IT_ANC_TAB,   // [nb * maxd] s-th ancestor of body i (nearest first), -1 padded; maxd = max(maxdepth,1)
  IT_CRB_MASK,  // [nb * 2] like IT_DESC_MASK but zero where the composite inertia is not needed (joint has no free
                //          velocity column) or comes from the whole-robot total (crb_full_body)
  IT_DESC_MASK, // [nb * 2] 64-bit subtree membership mask of body i (lo, hi), bit k = body k in subtree(i)
  IT_PDES_K,    // [nv] desired-record index that pins velocity index k, or -1 (free)
  IT_MCODE,     // [n_items * 2] packed assembly code of one (row block, free column) item of the constraint matrix:
                //   int0 = kind(3) | sg+1(2) | k(7) | frame+1 or joint component+1 (7) | generic task(6)
                //   int1 = first row(7) | rows(4) | free column f(7)       kind 7 = wrench-balance block
  IT_RCODE,     // [m] packed row code ; IT_R_DES [m] desired index ; IT_R_T [m] generic task ; IT_R_PINK [m]
  IT_R_DES, IT_R_T, IT_R_PINK,
  IT_META, // [(6 + (maxd + 3) / 4) * 64] packed per-lane metadata (qpc_tick.h TickMeta), word-major: [word][lane]
  IT_LAW, // [n_desired * 3] desired law of entry i: kind (QPC_LAW_*), arg0, arg1
          //   JOINT_PD: q index, v index;  COM_PD: component, -;  ORIENTATION_PD: body, component
  IT_SIM_TREE, // [(nb + 1) * 4] simulation step: per body {children mask lo, hi, height, depth}; last row {max height, max depth}
  IT_HOT, // solver lookups staged into LDS per QP: row_slack[m], slack_row[ne], slack_woff[ne], slack_dim[ne], slack_s0[ne]
  IT_HOT_S, // IT_HOT of the reduced problem of the presolve (rows renumbered)
  IT_COUNT
};

// double tables
enum {
  DT_AXIS, DT_RF, DT_PF, DT_MASS, DT_COM, DT_ICOM, // per-body constants, component-major: [component][body]
  DT_REG2, DT_GRAV /*3 + mtot*/,
  DT_SJ_W,   // static weight of folded soft joint tasks
  DT_T_POINT, DT_T_W, // per generic task: point[3]; static W (dim*dim, row-major, at IT_T_WOFF)
  DT_C_POS, DT_C_NORMAL, DT_C_MU,
  DT_LAW, // [n_desired * 3] kp, kd, ref of the desired law of entry i
  DT_HV0,   // [nf] static Hessian diagonal of the free accelerations: 2 reg + folded soft joint tasks with static weights
  DT_WINV0, // [wsize] inverses of the static slack Hessian blocks 2 sym(W) (no eps), valid when h_static
  DT_KTAB,  // [k_classes * 16 * 20] k_tab: eigenvalues [4] and eigenvectors [4][4] of B'B on each of the 16 free sets of a contact's four rays
  DT_COUNT
};

// LDS regions (offsets in doubles into the per-QP LDS slab)
enum {
  L_Q, L_V, L_DES, L_CW, L_CMAX, L_TW, L_SURF,
  L_HOT,  // m + 4*ne hot solver lookups (IT_HOT) as doubles
  L_RL,   // nb*12 local transforms
  L_H,    // nb*12 world transforms
  L_S,    // nv*6 motion subspaces (world)
  L_TJ, L_T, L_CR, L_AB, // nb*6 each
  L_IW, L_IC,            // nb*10 each
  L_HB,                  // nb*6
  L_MISC,                // 48 scalars
  L_AMOM,                // 6*nv
  L_ACCP,                // (ntask+1)*6 contribution of pinned accelerations
  L_M,                   // m * nfs   scaled constraint matrix  E * diag(sqrt(Hv^-1))
  L_D,                   // m rhs
  L_HV, L_SQH, L_GV,     // nf each
  L_W, L_WINV,           // sum dim^2 over soft generic tasks
  L_SM,                  // m * ms  Schur complement / its Cholesky factor (lower)
  L_X,                   // m1 * m1s explicit inverse of L11
  L_X2,                  // m1 * 6  Z = S11^-1 S12
  L_B22, L_C22, L_L22, L_S22I, // 36 each
  L_G,                   // 6 * nrho
  L_Y,                   // 6 x nrho, (K G')' with G's padded row stride QPC_GS(nrho)
  L_HR, L_KR,            // nc * N * N
  L_HI, L_WS, L_U,       // nrho
  L_XV, L_XE, L_XR, L_LAM, // iterate
  L_DV, L_DE, L_DR, L_DL, // step / direction
  L_RV, L_RE, L_RR, L_RC, // residuals / right-hand sides
  L_T1, L_T2, L_T3,      // m-sized temporaries
  L_TR1, L_TR2,          // nrho-sized temporaries
  L_TE1,                 // ne-sized temporary
  L_RED,                 // 4 x 64 reduction scratch
  L_VD,                  // nv solution accelerations
  L_SV, L_F,             // nb*6
  L_CWR,                 // nc*6 contact wrenches
  L_BW,                  // nb*6 body wrenches (RNEA)
  L_SVK,                 // nv*6 S_k * vdot_k (RNEA)
  L_WSP,                 // nrho working set before the last primal-dual update (fallback of the search)
  // ---- presolve (QpcDims::presolve; qpc_tick.h presolve / postsolve)
  L_PA,                  // m_s x nf_s: M^ before the whitening, overlays the solver scratch
  L_PZ,                  // (nf_s + 2) x 6 n_e: P, the columns of Q, P + g~ of the eliminated unknowns (persistent: the postsolve reads it)
  L_PH,                  // nf_s (nf_s + 1) / 2 + nf_s: H' = I + Q'Q (packed lower) and the linear term h
  L_PC,                  // nf_s x nf_s: C^-1 (lower), H' = C C'
  L_COUNT
};

// internal law code (not part of include/qpc.h): QPC_LAW_ORIENTATION_PD on the FLOATING body of a mechanism whose floating joint hangs on
// the world -- its orientation and body-frame angular velocity are the joint's own coordinates, so the law is evaluated in the input
// phase (before the kinematics) and the task can be a pin of the joint's angular accelerations (qpc_compile.cpp pass 1)
#define QPC_LAW_ORIENTATION_PD_ROOT 16

// Row stride of G (6 x nrho, L_G): nrho is typically a multiple of 32 doubles, which would put the six rows -- read
// side by side by the lanes (a, b) of the S22 build and by the wrench-row mat-vecs -- on the same LDS banks; two extra
// doubles per row shift each row by four banks and keep 16-byte alignment
#define QPC_GS(nrho) (((nrho) % 4 == 0) ? (nrho) + 2 : (nrho))

// Per-contact stride of the N x N blocks H_rho / K (L_HR, L_KR): 16 doubles = 128 B is exactly one sweep over the LDS
// banks, so the lanes that work one contact each (kr_block) would all hit the same banks; two doubles of padding
#define QPC_CS(N) ((((N) * (N)) % 16 == 0) ? (N) * (N) + 2 : (N) * (N))

#define QPC_CSH(N) QPC_CS(N) // H_rho blocks likewise

// slots inside L_MISC
enum {
  MS_COM = 0,      // 3
  MS_ADV = 3,      // 6
  MS_WG = 9,       // 6
  MS_EPS = 15, MS_HSCALE = 16,
  MS_U0 = 17,      // 6: wrench-space constant u0 = c2 - Z'c1 of the active-set search
  MS_KS = 24,      // 16: k_tab: 2 * contact weight per contact; with k_classes > 1 (nc <= 8) slots 8.. hold the contacts' table offsets
  MS_COUNT = 40
};

// tree-depth bound of the ahead-of-time generic kernel (the JIT-specialised kernel has no such
// bound).  NOTE: hipcc -O3 (ROCm 7.2) miscompiles the generic kernel with a bound of 32 (wrong
// kinematics on gfx950, correct at -O1 and in the emulator); 16 is validated by the GPU tests.
#define QPC_GENERIC_MAXD 16

struct QpcDims {
  int nb, nq, nv, nf, npin, nsj, ntask, nc, N, nrho, ne, m, m1, nw;
  int floating, fv0; // floating body and its first velocity index (or -1)
  int n_desired, n_tw, n_surf, maxdepth;
  int n_items; // (row block, free column) items of the constraint-matrix assembly (IT_MCODE)
  int ve_exact_ok; // no matrix-weighted slack block: the v̇ / slack blocks of H may be used unregularised (exact polish)
  int n_des_inputs, n_des_late; // desired entries read from the caller / computed after the kinematics (COM, orientation laws)
  int law_mask;      // bit k: some desired entry follows law kind k (QPC_LAW_*): the specialised kernel only carries the laws in use
  int maxd; // row length of IT_ANC_TAB
  int crb_full_body; // body whose subtree is the whole mechanism and owns free columns (its composite inertia = total), or -1
  int nfs, ms, m1s;  // padded leading dimensions (odd) of L_M, L_SM, L_X
  int wsize;         // doubles in L_W / L_WINV
  int h_static;      // no per-QP task weights: the v̇ / slack blocks of the Hessian are the same for every QP (DT_HV0, DT_WINV0)
  int k_tab;         // N = 4, static friction coefficients: the rho block of a contact is 2 w_c B'B with a B'B that only depends on
                     // mu, so (H_rho,FF + eps I)^-1 of every free set follows from a static eigen-decomposition (DT_KTAB)
  int k_classes;     // number of distinct mu (one 16 x 20 table each)
  int k_class[16];   // table of contact c
  int it[IT_COUNT];
  int dt[DT_COUNT];
  int lds[L_COUNT];
  int lds_doubles;   // per-QP LDS slab size
  int n_int, n_dbl;  // table sizes
  int dfs;           // the bodies are numbered in depth-first order (every subtree is a contiguous index range, at most 32 bodies):
                     // sums over a subtree are differences of a wave prefix sum (qpc_tick.h, inverse dynamics)
  // Null-space presolve of hard 6-row task blocks with six private free columns (the foot tasks of a standing controller: the six joint
  // accelerations of a leg are determined by the task and the base acceleration): v_L = P - Q v_b is substituted into the remaining rows
  // and the objective, the solver runs on nf_s columns / m_s rows (QpcProgram::dims_s), the postsolve restores the eliminated columns.
  // A QP whose elimination is ill-conditioned (a leg near a singularity) is solved as assembled, by the same kernel.
  int presolve;      // number of eliminated blocks n_e (0: none).  Block e is rows 6 e .. 6 e + 5 and free columns pre_c0[e] .. + 5
  int pre_nf, pre_m; // nf_s, m_s: the free columns pre_b0 .. pre_b0 + nf_s - 1 and the rows 6 n_e .. m - 1 stay
  int pre_c0[4], pre_b0;
  int pre_fold;      // rows 6 n_e .. 6 n_e + pre_fold - 1 that stay are all SOFT task rows (slack s = row - 6 n_e): with so few unknowns left
                     // their terms 1/2 (M^ x_b - d^)' W2 (M^ x_b - d^) go into H' and h before the whitening, and the solver sees
                     // neither these rows nor their slacks -- only the wrench rows (0: rows and slacks are kept)
  int pre_wd[6];     // ... where the weight of folded row s sits in L_W (the diagonal entry of its slack block: no matrix weights here)
  // solver settings (qpc_settings)
  int max_as_iters, max_refine, max_pdas;
  double eps_primal, tau_pivot, tol_bound, tol_accept;
  double h_static_max, h_static_min; // largest |diagonal| / smallest diagonal of the static v̇ and slack blocks (h_static)
  double k_mu2;      // k_tab with ONE friction coefficient: mu^2 (the K blocks then have closed forms, Solver::k_row_closed); else -1
};

// LDS slab of the simulation step (qpc_sim.h), offsets in doubles (fields in allocation order)
struct QpcSimLayout {
  int q, v, tau, rl, h, s, tj, t, cr, iw, cf, ia, pa, acc, x, total;
};
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define QPC_HOST_DEVICE __host__ __device__
#else
#define QPC_HOST_DEVICE
#endif
QPC_HOST_DEVICE inline QpcSimLayout qpc_sim_layout(const QpcDims &D) {
  QpcSimLayout L{};
  int o = 0;
#define QPC_SIM_TAKE(field, n) L.field = o; o += (n)
  QPC_SIM_TAKE(q, D.nq); QPC_SIM_TAKE(v, D.nv); QPC_SIM_TAKE(tau, D.nv);
  QPC_SIM_TAKE(rl, 12 * D.nb); // local and world transforms are dead once the ground reactions exist (the floating-joint
  QPC_SIM_TAKE(h, 12 * D.nb);  // lanes keep their own transform in registers): the articulated inertias overlay both
  QPC_SIM_TAKE(tj, 6 * D.nb); QPC_SIM_TAKE(t, 6 * D.nb);
  { const int ncf = 6 * (D.nc > 0 ? D.nc : 1); QPC_SIM_TAKE(cf, ncf > D.nv ? ncf : D.nv); }
  // what only the body's own lane reads again (world inertia, 1-DoF motion subspace column, c_i) lives in its registers
  L.s = L.cr = L.iw = -1;
  // articulated-body contributions to the parent: 21 (packed symmetric 6x6, over rl + h) + 6 (over tj) per body
  L.ia = L.rl;
  L.pa = L.tj;
  L.acc = L.t; // pA0_i is parked where T_i was, by the lane that read it
  L.x = L.cf;  // the contact wrenches are consumed before the first acceleration is written
#undef QPC_SIM_TAKE
  L.total = o;
  return L;
}
