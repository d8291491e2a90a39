// qpc_sim.h -- one simulation step for ONE robot state on ONE wavefront: forward
// dynamics under the torques of the controller tick, a compliant ground at the controller's contact
// points and a semi-implicit Euler update, so that the closed loop of the reference's acceptance test
// (notebooks/Standing controller.ipynb cells 9-11: RigidBodySim `Dynamics(mechanism, controller)` +
// soft contacts + `simulate`) runs per batch element without leaving the device.
//
// Written in the same phase vocabulary as qpc_tick.h (QPC_PHASE / QPC_END_PHASE, all cross-lane
// traffic through LDS), so the same source runs under the race-detecting CPU emulation of tests/emul.
//
//   kinematics            : transform_to_root, world motion subspaces, twists, bias accelerations,
//                           world inertias (the a2 sweeps of the tick, without pinned accelerations)
//   ground contact        : per contact point of the controller (addcontact!, momentum.jl:133-148) a
//                           normal spring/damper and viscous tangential friction capped by Coulomb's law
//   forward dynamics      : articulated-body algorithm in world coordinates, lane = body, one phase per tree
//                           level: leaves -> roots for the articulated inertias / bias forces (siblings in
//                           parallel, a parent gathers its children), roots -> leaves for the accelerations.
//                           A first version built the mass matrix (CRBA) and factorised it with a 36-step
//                           Cholesky + 36-step back substitution; the level-parallel form has ~2 x 12 steps.
//   integration           : v += dt v̇ ; q <- q (+) v dt on the configuration manifold
// Checked against the oracle's forward dynamics / integrate and the Python contact model of
// tests/sim.py (tests/test_device_sim.py).
#pragma once
#include "../../include/qpc.h"
#include "qpc_program.h"
#include "qpc_tick.h" // small dense helpers, TickMeta

namespace qpc {

typedef QpcSimLayout SimLayout;
QPC_FN SimLayout sim_layout(const QpcDims &D) { return qpc_sim_layout(D); }

struct SimIO {
  const double *q, *v, *tau; // this robot's records
  const int32_t *status;     // tick status of this robot or nullptr
  double *q_out, *v_out;     // may alias q, v
  int32_t *bad;              // failure counter of this robot or nullptr
};

// packed lower triangle of a symmetric 6x6: entry (r, c), r >= c
#define QPC_SYM6(r, c) ((r) * ((r) + 1) / 2 + (c))

// RPW = robots per wavefront.  1: lane = body (or velocity index, contact, ...) of ONE robot.  2 (mechanisms of at most 32
// bodies): lanes 0-31 work robot ios[0] on the LDS slab at sm0, lanes 32-63 robot ios[1] on the slab at sm0 + slab -- the
// step is bound by instruction issue (every level of the articulated-body passes runs the whole phase for two or three
// active lanes), and one instruction stream then advances two robots.  Inside a phase `lane` is the lane WITHIN the robot.
#define QPC_SIM_LOCAL \
  const int sub_ = (RPW == 2) ? (lane_ >> 5) : 0; \
  const int lane = (RPW == 2) ? (lane_ & 31) : lane_; \
  P sm = sm0 + sub_ * slab; \
  const SimIO io = (RPW == 2 && sub_) ? ios[RPW - 1] : ios[0]; /* selects on constant indices: no private-memory array */ \
  (void)sub_; (void)lane; (void)sm; (void)io;
template <class P, int RPW = 1>
QPC_FN void qpc_sim_step(const QpcDims &D, const int *IT, const double *DT, P sm0, const SimIO *ios, const qpc_sim_params &sp, int slab = 0) {
  QPC_LANE_SCOPE
  constexpr int LW = 64 / RPW; // lanes per robot
  const int nb = D.nb, nq = D.nq, nv = D.nv, nc = D.nc;
  const SimLayout L = sim_layout(D);
  const int maxh = IT[D.it[IT_SIM_TREE] + 4 * nb], maxdp = IT[D.it[IT_SIM_TREE] + 4 * nb + 1];

  // per-lane static metadata in registers, as in the tick (TickMeta): joint type / offsets / ancestor chain of
  // body `lane`; plus its children mask, height above the leaves and depth below the roots
  TickMeta M;
  QPC_LANE_INTS(tr_, 4);
  QPC_PHASE(lane_) { QPC_SIM_LOCAL
    int *mt = QPC_LI(M.v, lane_);
    {
      const int aw = (D.maxd + 3) / 4, rowlen = 6 + aw;
      QPC_UNROLL_JIT
      for (int e = 0; e < QPC_META_WORDS; e++) mt[e] = e < rowlen ? IT[D.it[IT_META] + e * 64 + lane] : 0;
    }
    int *tr = QPC_LI(tr_, lane_);
    for (int e = 0; e < 4; e++) tr[e] = lane < nb ? IT[D.it[IT_SIM_TREE] + 4 * lane + e] : (e < 2 ? 0 : -1);
    // a failed QP (checkstatus, momentum.jl:83-91) applies no torque for this step and is counted
    const bool failed = io.status && io.status[0] > QPC_STATUS_ALMOST_OPTIMAL;
    for (int i = lane; i < nq; i += LW) sm[L.q + i] = io.q[i];
    for (int i = lane; i < nv; i += LW) { sm[L.v + i] = io.v[i]; sm[L.tau + i] = failed ? 0.0 : io.tau[i]; }
  } QPC_END_PHASE

  QPC_PHASE(lane_) { QPC_SIM_LOCAL // joint transforms  x_parent = Rf (Rj x + pj) + pf
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane_);
      const int jt = MT_JT(mt), q0 = MT_Q0(mt);
      double ax[3], Rf[9], pf[3]; // per-body constants, stored [component][body]
      for (int k = 0; k < 3; k++) { ax[k] = DT[D.dt[DT_AXIS] + k * nb + i]; pf[k] = DT[D.dt[DT_PF] + k * nb + i]; }
      for (int k = 0; k < 9; k++) Rf[k] = DT[D.dt[DT_RF] + k * nb + i];
      double Rj[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pj[3] = {0, 0, 0};
      if (jt == QPC_JOINT_REVOLUTE) {
        double s, c;
        QPC_SINCOS(sm[L.q + q0], s, c);
        const double C = 1 - c;
        Rj[0] = c + ax[0] * ax[0] * C; Rj[1] = ax[0] * ax[1] * C - ax[2] * s; Rj[2] = ax[0] * ax[2] * C + ax[1] * s;
        Rj[3] = ax[1] * ax[0] * C + ax[2] * s; Rj[4] = c + ax[1] * ax[1] * C; Rj[5] = ax[1] * ax[2] * C - ax[0] * s;
        Rj[6] = ax[2] * ax[0] * C - ax[1] * s; Rj[7] = ax[2] * ax[1] * C + ax[0] * s; Rj[8] = c + ax[2] * ax[2] * C;
      } else if (jt == QPC_JOINT_PRISMATIC) {
        const double qq = sm[L.q + q0];
        pj[0] = ax[0] * qq; pj[1] = ax[1] * qq; pj[2] = ax[2] * qq;
      } else if (jt == QPC_JOINT_QUAT_FLOATING) {
        const double w = sm[L.q + q0], x = sm[L.q + q0 + 1], y = sm[L.q + q0 + 2], z = sm[L.q + q0 + 3];
        Rj[0] = 1 - 2 * (y * y + z * z); Rj[1] = 2 * (x * y - w * z); Rj[2] = 2 * (x * z + w * y);
        Rj[3] = 2 * (x * y + w * z); Rj[4] = 1 - 2 * (x * x + z * z); Rj[5] = 2 * (y * z - w * x);
        Rj[6] = 2 * (x * z - w * y); Rj[7] = 2 * (y * z + w * x); Rj[8] = 1 - 2 * (x * x + y * y);
        pj[0] = sm[L.q + q0 + 4]; pj[1] = sm[L.q + q0 + 5]; pj[2] = sm[L.q + q0 + 6];
      }
      double H[12];
      mm3(Rf, Rj, H);
      mv3(Rf, pj, H + 9);
      H[9] += pf[0]; H[10] += pf[1]; H[11] += pf[2];
      st(sm + L.rl + 12 * i, 12, H);
    }
  } QPC_END_PHASE

  QPC_LANE_ARRAY(hreg_, 12); // world transform of body `lane` (the floating-joint lanes need it after LDS has moved on)
  // what only the body's OWN lane reads again stays in its registers instead of LDS (10 KB per robot instead of 17.6: two
  // packed workgroups per SIMD): [0..9] world inertia, [10..15] motion subspace column of a 1-DoF joint, [16..21] c_i
  QPC_LANE_ARRAY(own_, 22);
  QPC_PHASE(lane_) { QPC_SIM_LOCAL // transform_to_root along the (register-resident) ancestor chain, nearest first
    if (lane < nb) {
      const int *mt = QPC_LI(M.v, lane_);
      double *H = QPC_LA(hreg_, lane_);
      double A[12], t[3];
      ld(sm + L.rl + 12 * lane, 12, H);
      QPC_UNROLL_JIT
      for (int s = 0; s < QPC_MAXD; s++) {
        const int an = MT_ANC(mt, s);
        if (an >= 0) {
          ld(sm + L.rl + 12 * an, 12, A);
          mv3(A, H + 9, t);
          mm3(A, H, H);
          H[9] = t[0] + A[9]; H[10] = t[1] + A[10]; H[11] = t[2] + A[11];
        }
      }
      st(sm + L.h + 12 * lane, 12, H);
    }
  } QPC_END_PHASE

  QPC_PHASE(lane_) { QPC_SIM_LOCAL // motion subspace columns (world), joint twist, world inertia
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane_);
      const int jt = MT_JT(mt), v0 = MT_V0(mt), n = MT_NVJ(mt);
      const double *H = QPC_LA(hreg_, lane_);
      double tj[6] = {0, 0, 0, 0, 0, 0};
      const double ax[3] = {DT[D.dt[DT_AXIS] + i], DT[D.dt[DT_AXIS] + nb + i], DT[D.dt[DT_AXIS] + 2 * nb + i]};
      for (int c = 0; c < n; c++) {
        double S[6] = {0, 0, 0, 0, 0, 0};
        if (jt == QPC_JOINT_REVOLUTE) { mv3(H, ax, S); cross3(H + 9, S, S + 3); }
        else if (jt == QPC_JOINT_PRISMATIC) { mv3(H, ax, S + 3); }
        else {
          const int cc = c % 3;
          double col[3] = {H[cc], H[3 + cc], H[6 + cc]};
          if (c < 3) { S[0] = col[0]; S[1] = col[1]; S[2] = col[2]; cross3(H + 9, col, S + 3); }
          else { S[3] = col[0]; S[4] = col[1]; S[5] = col[2]; }
        }
        if (n == 1) { for (int a = 0; a < 6; a++) QPC_LA(own_, lane_)[10 + a] = S[a]; }
        const double vv = sm[L.v + v0 + c];
        for (int a = 0; a < 6; a++) tj[a] += S[a] * vv;
      }
      st(sm + L.tj + 6 * i, 6, tj);
      double cw[3];
      double I[6], cb[3];
      for (int k = 0; k < 6; k++) I[k] = DT[D.dt[DT_ICOM] + k * nb + i];
      for (int k = 0; k < 3; k++) cb[k] = DT[D.dt[DT_COM] + k * nb + i];
      mv3(H, cb, cw);
      cw[0] += H[9]; cw[1] += H[10]; cw[2] += H[11];
      double Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], Rt[9], Iw[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = H[3 * b + a];
      mm3(H, Ib, RI);
      mm3(RI, Rt, Iw);
      const double mm = DT[D.dt[DT_MASS] + i], cc = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
      double o[10] = {mm, mm * cw[0], mm * cw[1], mm * cw[2],
                      Iw[0] + mm * (cc - cw[0] * cw[0]), Iw[4] + mm * (cc - cw[1] * cw[1]), Iw[8] + mm * (cc - cw[2] * cw[2]),
                      Iw[1] - mm * cw[0] * cw[1], Iw[2] - mm * cw[0] * cw[2], Iw[5] - mm * cw[1] * cw[2]};
      for (int a = 0; a < 10; a++) QPC_LA(own_, lane_)[a] = o[a];
    }
  } QPC_END_PHASE

  QPC_PHASE(lane_) { QPC_SIM_LOCAL // twist_wrt_world, velocity-product acceleration c_i = T_i x (S_i v_i)
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane_);
      double T[6], tj[6], cr[6];
      ld(sm + L.tj + 6 * i, 6, tj);
      for (int a = 0; a < 6; a++) T[a] = tj[a];
      QPC_UNROLL_JIT
      for (int s = 0; s < QPC_MAXD; s++) {
        const int an = MT_ANC(mt, s);
        if (an >= 0)
          for (int a = 0; a < 6; a++) T[a] += sm[L.tj + 6 * an + a];
      }
      crm(T, tj, cr);
      st(sm + L.t + 6 * i, 6, T);
      for (int a = 0; a < 6; a++) QPC_LA(own_, lane_)[16 + a] = cr[a];
    }
  } QPC_END_PHASE

  QPC_PHASE(lane_) { QPC_SIM_LOCAL // ground reaction per contact point
    for (int c = lane; c < nc; c += LW) {
      // compliant half space z <= ground_z: normal spring/damper (unilateral), viscous tangential
      // friction capped at mu * normal force; wrench about the world origin
      const int body = IT[D.it[IT_C_BODY] + c];
      double H[12], T[6], p[3], wxp[3], f[3] = {0, 0, 0}, mo[3];
      if (body < 0) { // a contact point of the WORLD body (qpc_add_contact accepts QPC_WORLD): it pushes on nothing that moves
        for (int a = 0; a < 6; a++) sm[L.cf + 6 * c + a] = 0.0;
        continue;
      }
      ld(sm + L.h + 12 * body, 12, H);
      ld(sm + L.t + 6 * body, 6, T);
      mv3(H, DT + D.dt[DT_C_POS] + 3 * c, p);
      p[0] += H[9]; p[1] += H[10]; p[2] += H[11];
      const double pen = sp.ground_z - p[2];
      if (pen > 0.0) {
        cross3(T, p, wxp);
        const double vel[3] = {T[3] + wxp[0], T[4] + wxp[1], T[5] + wxp[2]};
        double fn = sp.stiffness * pen - sp.damping * vel[2];
        fn = fn > 0.0 ? fn : 0.0;
        double ftx = -sp.tangential_damping * vel[0], fty = -sp.tangential_damping * vel[1];
        const double cap = DT[D.dt[DT_C_MU] + c] * fn, nrm = QPC_SQRT(ftx * ftx + fty * fty);
        if (nrm > cap) { const double sc = cap / (nrm > 1e-300 ? nrm : 1e-300); ftx *= sc; fty *= sc; }
        f[0] = ftx; f[1] = fty; f[2] = fn;
      }
      cross3(p, f, mo);
      for (int a = 0; a < 3; a++) { sm[L.cf + 6 * c + a] = mo[a]; sm[L.cf + 6 * c + 3 + a] = f[a]; }
    }
  } QPC_END_PHASE

  // ---- articulated-body pass, leaves -> roots, one phase per height.  In world coordinates nothing has to be
  // transformed between bodies: IA_i = I_i + sum_children Ia_c,  pA_i = T_i x* I_i T_i - f_ext,i + sum_children pa_c,
  //   1-DoF joint:  U = IA S, d = S.U, u = tau - S.pA,  Ia = IA - U U'/d,  pa = pA + Ia c_i + U u/d
  //   fixed joint:  Ia = IA, pa = pA + IA c_i
  //   floating joint (S invertible): the subtree decouples, Ia = 0, pa = S^-T tau (the joint "torque" as a world wrench)
  // The lane keeps U, 1/d, u (or IA, pA - S^-T tau for a floating joint) in registers for the second pass.
  QPC_LANE_ARRAY(ab_, 29); // [0..5] U, [6] 1/d, [7] u   |  floating: [0..20] IA packed, [21..26] S^-T tau - pA ; [28] failure
  QPC_PHASE(lane_) { QPC_SIM_LOCAL // bias forces pA0_i = T_i x* I_i T_i - f_ext,i for all bodies at once (parked in the acceleration slots)
    QPC_LA(ab_, lane_)[28] = 0.0;
    if (lane < nb) {
      const int i = lane;
      double T[6], I[10], h[6], pA[6];
      for (int a = 0; a < 10; a++) I[a] = QPC_LA(own_, lane_)[a];
      ld(sm + L.t + 6 * i, 6, T); // (L.acc overlays L.t: this lane reads its twist before it parks pA0 in the same slot)
      inertia_mul(I, T, h);
      crf(T, h, pA);
      for (int e = IT[D.it[IT_BC_PTR] + i]; e < IT[D.it[IT_BC_PTR] + i + 1]; e++) {
        const int cc = IT[D.it[IT_BC_IDX] + e];
        for (int a = 0; a < 6; a++) pA[a] -= sm[L.cf + 6 * cc + a];
      }
      st(sm + L.acc + 6 * i, 6, pA);
    }
  } QPC_END_PHASE
  for (int lvl = 0; lvl <= maxh; lvl++) {
    QPC_PHASE(lane_) { QPC_SIM_LOCAL
      const int *tr = QPC_LI(tr_, lane_);
      if (lane < nb && tr[2] == lvl) {
        const int i = lane;
        const int *mt = QPC_LI(M.v, lane_);
        const int jt = MT_JT(mt), v0 = MT_V0(mt);
        double *ab = QPC_LA(ab_, lane_);
        double IA[21], pA[6], I[10], c[6];
        for (int a = 0; a < 10; a++) I[a] = QPC_LA(own_, lane_)[a];
        ld(sm + L.acc + 6 * i, 6, pA);
        for (int a = 0; a < 6; a++) c[a] = QPC_LA(own_, lane_)[16 + a];
        { // spatial inertia {m, h, J} as a packed symmetric 6x6 in [angular; linear] order: [[J, h^], [h^', m 1]]
          const double m_ = I[0], hx = I[1], hy = I[2], hz = I[3];
          IA[QPC_SYM6(0, 0)] = I[4]; IA[QPC_SYM6(1, 0)] = I[7]; IA[QPC_SYM6(1, 1)] = I[5];
          IA[QPC_SYM6(2, 0)] = I[8]; IA[QPC_SYM6(2, 1)] = I[9]; IA[QPC_SYM6(2, 2)] = I[6];
          IA[QPC_SYM6(3, 0)] = 0.0; IA[QPC_SYM6(3, 1)] = hz; IA[QPC_SYM6(3, 2)] = -hy;
          IA[QPC_SYM6(4, 0)] = -hz; IA[QPC_SYM6(4, 1)] = 0.0; IA[QPC_SYM6(4, 2)] = hx;
          IA[QPC_SYM6(5, 0)] = hy; IA[QPC_SYM6(5, 1)] = -hx; IA[QPC_SYM6(5, 2)] = 0.0;
          IA[QPC_SYM6(3, 3)] = m_; IA[QPC_SYM6(4, 3)] = 0.0; IA[QPC_SYM6(4, 4)] = m_;
          IA[QPC_SYM6(5, 3)] = 0.0; IA[QPC_SYM6(5, 4)] = 0.0; IA[QPC_SYM6(5, 5)] = m_;
        }
        unsigned long long kids = ((unsigned long long)(unsigned)tr[1] << 32) | (unsigned long long)(unsigned)tr[0];
        while (kids) {
          const int k = QPC_CTZ64(kids);
          kids &= kids - 1;
          for (int a = 0; a < 21; a++) IA[a] += sm[L.ia + 21 * k + a];
          for (int a = 0; a < 6; a++) pA[a] += sm[L.pa + 6 * k + a];
        }
        double Ia[21], pa[6];
        if (jt == QPC_JOINT_QUAT_FLOATING) {
          // tau (body-frame wrench [n; f] dual to v = [omega_body; v_body]) as a world wrench: S^-T tau
          const double *H = QPC_LA(hreg_, lane_);
          double tb[6], fw[3], nw[3], pxf[3];
          for (int a = 0; a < 6; a++) tb[a] = sm[L.tau + v0 + a];
          mv3(H, tb, nw); mv3(H, tb + 3, fw);
          cross3(H + 9, fw, pxf);
          for (int a = 0; a < 21; a++) { ab[a] = IA[a]; Ia[a] = 0.0; }
          for (int a = 0; a < 3; a++) { pa[a] = nw[a] + pxf[a]; pa[3 + a] = fw[a]; }
          for (int a = 0; a < 6; a++) ab[21 + a] = pa[a] - pA[a];
        } else if (jt == QPC_JOINT_FIXED) {
          for (int a = 0; a < 21; a++) Ia[a] = IA[a];
          for (int r = 0; r < 6; r++) {
            double acc = pA[r];
            for (int k = 0; k < 6; k++) acc += IA[r >= k ? QPC_SYM6(r, k) : QPC_SYM6(k, r)] * c[k];
            pa[r] = acc;
          }
        } else {
          double S[6], U[6];
          for (int a = 0; a < 6; a++) S[a] = QPC_LA(own_, lane_)[10 + a];
          double d = 0.0, u = sm[L.tau + v0];
          for (int r = 0; r < 6; r++) {
            double acc = 0.0;
            for (int k = 0; k < 6; k++) acc += IA[r >= k ? QPC_SYM6(r, k) : QPC_SYM6(k, r)] * S[k];
            U[r] = acc;
            d += S[r] * acc;
            u -= S[r] * pA[r];
          }
          const bool ok = d > 0.0;
          if (!ok) ab[28] = 1.0; // not positive definite: the step is rejected below
          const double dinv = ok ? QPC_RCP(d) : 0.0;
          for (int r = 0; r < 6; r++)
            for (int k = 0; k <= r; k++) Ia[QPC_SYM6(r, k)] = IA[QPC_SYM6(r, k)] - U[r] * U[k] * dinv;
          for (int r = 0; r < 6; r++) {
            double acc = pA[r] + U[r] * (u * dinv);
            for (int k = 0; k < 6; k++) acc += Ia[r >= k ? QPC_SYM6(r, k) : QPC_SYM6(k, r)] * c[k];
            pa[r] = acc;
          }
          for (int a = 0; a < 6; a++) ab[a] = U[a];
          ab[6] = dinv; ab[7] = u;
        }
        st(sm + L.ia + 21 * i, 21, Ia);
        st(sm + L.pa + 6 * i, 6, pa);
      }
    } QPC_END_PHASE
  }

  // ---- acceleration pass, roots -> leaves, one phase per depth: a' = a_parent + c_i (a_world = -g),
  //   1-DoF: qdd = (u - U.a')/d, a = a' + S qdd;  fixed: a = a';  floating: a = IA^-1 (S^-T tau - pA), qdd = S^-1 (a - a')
  for (int lvl = 0; lvl <= maxdp; lvl++) {
    QPC_PHASE(lane_) { QPC_SIM_LOCAL
      const int *tr = QPC_LI(tr_, lane_);
      if (lane < nb && tr[3] == lvl) {
        const int i = lane;
        const int *mt = QPC_LI(M.v, lane_);
        const int jt = MT_JT(mt), v0 = MT_V0(mt), par = MT_ANC(mt, 0);
        const double *ab = QPC_LA(ab_, lane_);
        double ap[6], a[6];
        for (int a = 0; a < 6; a++) ap[a] = QPC_LA(own_, lane_)[16 + a];
        if (par >= 0) { for (int k = 0; k < 6; k++) ap[k] += sm[L.acc + 6 * par + k]; }
        else { for (int k = 0; k < 3; k++) ap[3 + k] -= DT[D.dt[DT_GRAV] + k]; }
        if (jt == QPC_JOINT_QUAT_FLOATING) {
          double S6[36], sref[6], rhs[6], d[6], qdd[6];
          for (int r = 0; r < 6; r++)
            for (int k = 0; k < 6; k++) S6[6 * r + k] = ab[r >= k ? QPC_SYM6(r, k) : QPC_SYM6(k, r)];
          for (int r = 0; r < 6; r++) { sref[r] = S6[7 * r]; rhs[r] = ab[21 + r]; }
          chol6_solve(S6, sref, 1e-14, rhs, a);
          for (int k = 0; k < 6; k++) d[k] = a[k] - ap[k];
          motion_to_frame(QPC_LA(hreg_, lane_), d, qdd);
          for (int k = 0; k < 6; k++) sm[L.x + v0 + k] = qdd[k];
        } else if (jt == QPC_JOINT_FIXED) {
          for (int k = 0; k < 6; k++) a[k] = ap[k];
        } else {
          double S[6], ua = ab[7];
          for (int a = 0; a < 6; a++) S[a] = QPC_LA(own_, lane_)[10 + a];
          for (int k = 0; k < 6; k++) ua -= ab[k] * ap[k];
          const double qdd = ua * ab[6];
          for (int k = 0; k < 6; k++) a[k] = ap[k] + S[k] * qdd;
          sm[L.x + v0] = qdd;
        }
        st(sm + L.acc + 6 * i, 6, a);
      }
    } QPC_END_PHASE
  }

  unsigned long long brej = 0ull; // lanes whose articulated inertia was not positive: their robot's step is rejected
  { // (a wave-wide ballot over the PHYSICAL lanes: the emulator's form of the macros addresses its scratch through `sm` and `lane`)
    P sm = sm0;
    (void)sm;
    QPC_PHASE(lane) {
      QPC_BALLOT_PUT(0, brej, QPC_LA(ab_, lane)[28] != 0.0);
    } QPC_END_PHASE
    QPC_BALLOT_GET(0, brej);
  }
#define QPC_SIM_REJECT const bool reject = (RPW == 2) ? ((brej >> (32 * sub_)) & 0xffffffffull) != 0ull : brej != 0ull;
  QPC_PHASE(lane_) { QPC_SIM_LOCAL // semi-implicit Euler: the new velocity moves the configuration
    QPC_SIM_REJECT
    const bool failed = io.status && io.status[0] > QPC_STATUS_ALMOST_OPTIMAL;
    for (int k = lane; k < nv; k += LW) {
      const double vn = reject ? (double)sm[L.v + k] : sm[L.v + k] + sp.dt * sm[L.x + k];
      sm[L.tau + k] = vn; // tau is dead: reuse as the new velocity
      io.v_out[k] = vn;
    }
    if (lane == 0 && io.bad && (failed || reject)) io.bad[0] = io.bad[0] + 1;
  } QPC_END_PHASE
  QPC_PHASE(lane_) { QPC_SIM_LOCAL // configuration update on the manifold (revolute / prismatic: q += v dt; floating joint:
                    // quat <- quat * exp(w_body dt / 2) renormalised, pos += R(quat) v_body dt)
    QPC_SIM_REJECT
    if (lane < nb) {
      const int *mt = QPC_LI(M.v, lane_);
      const int jt = MT_JT(mt), q0 = MT_Q0(mt), v0 = MT_V0(mt);
      const double dt = reject ? 0.0 : sp.dt;
      if (jt == QPC_JOINT_REVOLUTE || jt == QPC_JOINT_PRISMATIC) {
        io.q_out[q0] = sm[L.q + q0] + sm[L.tau + v0] * dt;
      } else if (jt == QPC_JOINT_QUAT_FLOATING) {
        const double w = sm[L.q + q0], x = sm[L.q + q0 + 1], y = sm[L.q + q0 + 2], z = sm[L.q + q0 + 3];
        const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                             2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                             2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
        double vi[6], dp[3], dq[4];
        for (int a = 0; a < 6; a++) vi[a] = sm[L.tau + v0 + a];
        mv3(R, vi + 3, dp);
        const double wn = QPC_SQRT(vi[0] * vi[0] + vi[1] * vi[1] + vi[2] * vi[2]), th = wn * dt;
        if (th < 1e-12) { dq[0] = 1; dq[1] = 0.5 * vi[0] * dt; dq[2] = 0.5 * vi[1] * dt; dq[3] = 0.5 * vi[2] * dt; }
        else {
          double sh, ch;
          QPC_SINCOS(0.5 * th, sh, ch);
          const double sc = sh / wn;
          dq[0] = ch; dq[1] = sc * vi[0]; dq[2] = sc * vi[1]; dq[3] = sc * vi[2];
        }
        const double nw = w * dq[0] - x * dq[1] - y * dq[2] - z * dq[3];
        const double nx = w * dq[1] + x * dq[0] + y * dq[3] - z * dq[2];
        const double ny = w * dq[2] - x * dq[3] + y * dq[0] + z * dq[1];
        const double nz = w * dq[3] + x * dq[2] - y * dq[1] + z * dq[0];
        const double nn = 1.0 / QPC_SQRT(nw * nw + nx * nx + ny * ny + nz * nz);
        io.q_out[q0] = nw * nn; io.q_out[q0 + 1] = nx * nn; io.q_out[q0 + 2] = ny * nn; io.q_out[q0 + 3] = nz * nn;
        for (int a = 0; a < 3; a++) io.q_out[q0 + 4 + a] = sm[L.q + q0 + 4 + a] + dp[a] * dt;
      }
    }
  } QPC_END_PHASE
}

} // namespace qpc
