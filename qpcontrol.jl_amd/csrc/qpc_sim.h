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
//   dynamics bias  c      : RNEA with v̇ = 0 and the contact wrenches as external wrenches
//   mass matrix    M      : composite rigid body algorithm (lower triangle, lane = row)
//   M v̇ = tau - c         : Cholesky with the right-hand side carried as an extra row, lane = row, the row's
//                           factor entries in registers (JIT-specialised like the tick)
//   integration           : v += dt v̇ ; q <- q (+) v dt on the configuration manifold
// Checked against the oracle's forward dynamics / integrate and the Python contact model of
// tests/sim.py (tests/test_device_sim.py).
#pragma once
#include "../../include/qpc.h"
#include "qpc_program.h"
#include "qpc_tick.h" // small dense helpers, TickMeta

namespace qpc {

// compile-time bound of nv + 1 (rows of the factorization incl. the right-hand side row): the JIT
// specialisation defines it as the actual value so that the row registers index statically
#ifndef QPC_SIM_ROWS_MAX
#define QPC_SIM_ROWS_MAX 64
#endif

typedef QpcSimLayout SimLayout;
QPC_FN SimLayout sim_layout(const QpcDims &D) { return qpc_sim_layout(D); }

struct SimIO {
  const double *q, *v, *tau; // this robot's records
  const int32_t *status;     // tick status of this robot or nullptr
  double *q_out, *v_out;     // may alias q, v
  int32_t *bad;              // failure counter of this robot or nullptr
};

// once per workgroup: velocity index -> body into the persistent head of the slab
template <class P> QPC_FN void qpc_sim_prologue(const QpcDims &D, const int *IT, P sm) {
  const SimLayout L = sim_layout(D);
  QPC_PHASE(lane) {
    for (int k = lane; k < D.nv; k += 64) sm[L.vb + k] = (double)IT[D.it[IT_VBODY] + k];
  } QPC_END_PHASE
}

template <class P>
QPC_FN void qpc_sim_step(const QpcDims &D, const int *IT, const double *DT, P sm, const SimIO &io, const qpc_sim_params &sp) {
  const int nb = D.nb, nq = D.nq, nv = D.nv, nc = D.nc;
  const SimLayout L = sim_layout(D);
  // a failed QP (checkstatus, momentum.jl:83-91) applies no torque for this step and is counted
  const bool failed = io.status && io.status[0] > QPC_STATUS_ALMOST_OPTIMAL;

  // per-lane static metadata in registers, as in the tick (TickMeta): joint type / offsets / ancestor
  // chain of body `lane`, subtree mask of the body of velocity index `lane`; plus the full subtree
  // mask of body `lane` for the composite inertias (the tick's copy is pruned to the free joints)
  TickMeta M;
  QPC_LANE_INTS(dm_, 2);
  QPC_PHASE(lane) {
    int *mt = QPC_LI(M.v, lane);
    {
      const int aw = (D.maxd + 3) / 4, rowlen = 6 + aw;
      QPC_UNROLL_JIT
      for (int e = 0; e < QPC_META_WORDS; e++) mt[e] = e < rowlen ? IT[D.it[IT_META] + lane * rowlen + e] : 0;
    }
    int *dm = QPC_LI(dm_, lane);
    dm[0] = lane < nb ? IT[D.it[IT_DESC_MASK] + 2 * lane] : 0;
    dm[1] = lane < nb ? IT[D.it[IT_DESC_MASK] + 2 * lane + 1] : 0;
    for (int i = lane; i < nq; i += 64) sm[L.q + i] = io.q[i];
    for (int i = lane; i < nv; i += 64) { sm[L.v + i] = io.v[i]; sm[L.tau + i] = failed ? 0.0 : io.tau[i]; }
    if (lane == 0) sm[L.flag] = 0.0;
  } QPC_END_PHASE

  QPC_PHASE(lane) { // joint transforms  x_parent = Rf (Rj x + pj) + pf
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
      const int jt = MT_JT(mt), q0 = MT_Q0(mt);
      const double *ax = DT + D.dt[DT_AXIS] + 3 * i, *Rf = DT + D.dt[DT_RF] + 9 * i, *pf = DT + D.dt[DT_PF] + 3 * i;
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

  QPC_PHASE(lane) { // transform_to_root along the (register-resident) ancestor chain, nearest first
    if (lane < nb) {
      const int *mt = QPC_LI(M.v, lane);
      double H[12], A[12], t[3];
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

  QPC_PHASE(lane) { // motion subspace columns (world), joint twist, world inertia
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
      const int jt = MT_JT(mt), v0 = MT_V0(mt), n = MT_NVJ(mt);
      double H[12], tj[6] = {0, 0, 0, 0, 0, 0};
      ld(sm + L.h + 12 * i, 12, H);
      const double *ax = DT + D.dt[DT_AXIS] + 3 * i;
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
        st(sm + L.s + 6 * (v0 + c), 6, S);
        const double vv = sm[L.v + v0 + c];
        for (int a = 0; a < 6; a++) tj[a] += S[a] * vv;
      }
      st(sm + L.tj + 6 * i, 6, tj);
      double cw[3];
      const double *I = DT + D.dt[DT_ICOM] + 6 * i;
      mv3(H, DT + D.dt[DT_COM] + 3 * i, cw);
      cw[0] += H[9]; cw[1] += H[10]; cw[2] += H[11];
      double Ib[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]}, RI[9], Rt[9], Iw[9];
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Rt[3 * a + b] = H[3 * b + a];
      mm3(H, Ib, RI);
      mm3(RI, Rt, Iw);
      const double mm = DT[D.dt[DT_MASS] + i], cc = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
      double o[10] = {mm, mm * cw[0], mm * cw[1], mm * cw[2],
                      Iw[0] + mm * (cc - cw[0] * cw[0]), Iw[4] + mm * (cc - cw[1] * cw[1]), Iw[8] + mm * (cc - cw[2] * cw[2]),
                      Iw[1] - mm * cw[0] * cw[1], Iw[2] - mm * cw[0] * cw[2], Iw[5] - mm * cw[1] * cw[2]};
      st(sm + L.iw + 10 * i, 10, o);
    }
  } QPC_END_PHASE

  QPC_PHASE(lane) { // twist_wrt_world, bias increment T_i x (S_i v_i)
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane);
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
      st(sm + L.cr + 6 * i, 6, cr);
    }
  } QPC_END_PHASE

  QPC_PHASE(lane) { // bias accelerations, composite inertias; ground reaction per contact point
    if (lane < nb) {
      const int i = lane;
      const int *mt = QPC_LI(M.v, lane), *dm = QPC_LI(dm_, lane);
      double Ab[6], Ic[10];
      ld(sm + L.cr + 6 * i, 6, Ab);
      QPC_UNROLL_JIT
      for (int s = 0; s < QPC_MAXD; s++) {
        const int an = MT_ANC(mt, s);
        if (an >= 0)
          for (int a = 0; a < 6; a++) Ab[a] += sm[L.cr + 6 * an + a];
      }
      st(sm + L.ab + 6 * i, 6, Ab);
      for (int a = 0; a < 10; a++) Ic[a] = 0.0;
      unsigned long long msk = ((unsigned long long)(unsigned)dm[1] << 32) | (unsigned long long)(unsigned)dm[0];
      while (msk) {
        const int k = QPC_CTZ64(msk);
        msk &= msk - 1;
        for (int a = 0; a < 10; a++) Ic[a] += sm[L.iw + 10 * k + a];
      }
      st(sm + L.ic + 10 * i, 10, Ic);
    }
    for (int c = lane; c < nc; c += 64) {
      // compliant half space z <= ground_z: normal spring/damper (unilateral), viscous tangential
      // friction capped at mu * normal force; wrench about the world origin
      const int body = IT[D.it[IT_C_BODY] + c];
      double H[12], T[6], p[3], wxp[3], f[3] = {0, 0, 0}, mo[3];
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

  QPC_LANE_ARRAY(sk_, 6); // motion subspace column of velocity index `lane`
  QPC_PHASE(lane) { // body wrenches of the v̇ = 0 inverse dynamics; h_k = Ic(vbody k) S_k
    if (lane < nb) {
      const int i = lane;
      double Ab[6], T[6], I[10], h[6], ia[6], cf[6];
      ld(sm + L.ab + 6 * i, 6, Ab);
      ld(sm + L.t + 6 * i, 6, T);
      ld(sm + L.iw + 10 * i, 10, I);
      for (int a = 0; a < 3; a++) Ab[3 + a] -= DT[D.dt[DT_GRAV] + a]; // root acceleration -g
      inertia_mul(I, Ab, ia);
      inertia_mul(I, T, h);
      crf(T, h, cf);
      for (int a = 0; a < 6; a++) ia[a] += cf[a];
      for (int e = IT[D.it[IT_BC_PTR] + i]; e < IT[D.it[IT_BC_PTR] + i + 1]; e++) {
        const int c = IT[D.it[IT_BC_IDX] + e];
        for (int a = 0; a < 6; a++) ia[a] -= sm[L.cf + 6 * c + a];
      }
      st(sm + L.f + 6 * i, 6, ia);
    }
    if (lane < nv) {
      const int *mt = QPC_LI(M.v, lane);
      double Ic[10], h[6];
      double *S = QPC_LA(sk_, lane);
      ld(sm + L.ic + 10 * MT_VBODY(mt), 10, Ic);
      ld(sm + L.s + 6 * lane, 6, S);
      inertia_mul(Ic, S, h);
      st(sm + L.hk + 6 * lane, 6, h);
    }
  } QPC_END_PHASE

  QPC_PHASE(lane) { // column j of the mass matrix (lower triangle: rows k >= j) and entry j of the rhs row
    if (lane < nv) {
      const int j = lane;
      const int *mt = QPC_LI(M.v, lane);
      const double *S = QPC_LA(sk_, lane);
      const unsigned long long sub = ((unsigned long long)(unsigned)mt[5] << 32) | (unsigned long long)(unsigned)mt[4];
      double F[6] = {0, 0, 0, 0, 0, 0};
      unsigned long long msk = sub;
      while (msk) {
        const int b = QPC_CTZ64(msk);
        msk &= msk - 1;
        for (int a = 0; a < 6; a++) F[a] += sm[L.f + 6 * b + a];
      }
      double cj = 0.0;
      for (int a = 0; a < 6; a++) cj += S[a] * F[a];
      sm[L.mm + nv * L.mstride + j] = sm[L.tau + j] - cj;
      // M[k][j] = S_j . Ic(vbody k) S_k when joint j supports the body of k (k in the subtree of vbody j)
      for (int k0 = 0; k0 < nv; k0 += 4) {
        double hb[4][6];
        int on[4];
        QPC_UNROLL_FULL
        for (int u = 0; u < 4; u++) {
          const int k = k0 + u < nv ? k0 + u : 0;
          on[u] = (int)((sub >> (int)(double)sm[L.vb + k]) & 1ull);
          QPC_UNROLL_FULL
          for (int a = 0; a < 6; a++) hb[u][a] = sm[L.hk + 6 * k + a];
        }
        QPC_UNROLL_FULL
        for (int u = 0; u < 4; u++) {
          const int k = k0 + u;
          if (k < nv && k >= j) {
            double acc = 0.0;
            QPC_UNROLL_FULL
            for (int a = 0; a < 6; a++) acc += S[a] * hb[u][a];
            sm[L.mm + k * L.mstride + j] = on[u] ? acc : 0.0;
          }
        }
      }
    }
  } QPC_END_PHASE

  // left-looking Cholesky of M with the right-hand side as row nv (lane = row, the row's own factor
  // entries in registers, row k broadcast from LDS -- the scheme of the tick's S11 factorization):
  // afterwards row nv holds y = L^-1 (tau - c) and dinv[k] = 1 / L_kk
  QPC_LANE_ARRAY(lrow_, QPC_SIM_ROWS_MAX);
  QPC_UNROLL_JIT
  for (int k = 0; k < nv; k++) {
    QPC_PHASE(lane) {
      if (lane >= k && lane <= nv) {
        const int r = lane;
        double *lrow = QPC_LA(lrow_, lane);
        double accr = 0.0, acck = 0.0;
        for (int j0 = 0; j0 < k; j0 += 8) {
          double lk[8];
          QPC_UNROLL_FULL
          for (int u = 0; u < 8; u++) lk[u] = sm[L.mm + k * L.mstride + (j0 + u < k ? j0 + u : 0)];
          QPC_UNROLL_FULL
          for (int u = 0; u < 8; u++)
            if (j0 + u < k) { accr += lk[u] * lrow[j0 + u]; acck += lk[u] * lk[u]; }
        }
        const double mkk = sm[L.mm + k * L.mstride + k], mrk = sm[L.mm + r * L.mstride + k], dk = mkk - acck;
        const bool ok = dk > 1e-14 * mkk;
        const double inv = ok ? QPC_RSQRT(dk) : 0.0; // 1 / L_kk
        if (r == k) { sm[L.dinv + k] = inv; if (!ok) sm[L.flag] = 1.0; } // not positive definite: the step is rejected
        else { const double v = (mrk - accr) * inv; lrow[k] = v; sm[L.mm + r * L.mstride + k] = v; }
      }
    } QPC_END_PHASE
  }
  // back substitution L' v̇ = y, one phase per unknown: lane j keeps y_j in a register and applies the
  // update of the previous unknown before lane k closes v̇_k
  {
    QPC_LANE_ARRAY(y_, 1);
    QPC_PHASE(lane) {
      if (lane < nv) QPC_LA(y_, lane)[0] = sm[L.mm + nv * L.mstride + lane];
    } QPC_END_PHASE
    QPC_UNROLL_JIT
    for (int k = nv - 1; k >= 0; k--) {
      QPC_PHASE(lane) {
        if (lane <= k) {
          double *y = QPC_LA(y_, lane);
          if (k + 1 < nv) y[0] -= sm[L.mm + (k + 1) * L.mstride + lane] * sm[L.x + k + 1];
          if (lane == k) sm[L.x + k] = y[0] * sm[L.dinv + k];
        }
      } QPC_END_PHASE
    }
  }

  const bool reject = sm[L.flag] != 0.0;
  QPC_PHASE(lane) { // semi-implicit Euler: the new velocity moves the configuration
    for (int k = lane; k < nv; k += 64) {
      const double vn = reject ? (double)sm[L.v + k] : sm[L.v + k] + sp.dt * sm[L.x + k];
      sm[L.tau + k] = vn; // tau is dead: reuse as the new velocity
      io.v_out[k] = vn;
    }
    if (lane == 0 && io.bad && (failed || reject)) io.bad[0] = io.bad[0] + 1;
  } QPC_END_PHASE
  QPC_PHASE(lane) { // configuration update on the manifold (revolute / prismatic: q += v dt; floating joint:
                    // quat <- quat * exp(w_body dt / 2) renormalised, pos += R(quat) v_body dt)
    if (lane < nb) {
      const int *mt = QPC_LI(M.v, lane);
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
