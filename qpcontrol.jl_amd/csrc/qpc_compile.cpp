// qpc_compile.cpp -- freezes a controller into the device "task program" (see qpc_program.h).
// Pure host C++.  What is decided here, once, is what Parametron re-derives every tick on the CPU
// path: which accelerations are pinned by hard JointAccelerationTasks
// (src/tasks.jl:185-189), which soft tasks fold into the diagonal of the Hessian, the row layout of
// the equality block, the kinematic-path sign tables of every Jacobian (geometric_jacobian!,
// src/tasks.jl:33-34) and the per-QP LDS layout.
#include "qpc_compile.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_err;
void qpc_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *qpc_last_error(void) { return g_err.c_str(); }

int qpc_task_dim(const qpc_model *m, int kind, int joint) {
  switch (kind) {
    case QPC_TASK_SPATIAL_ACCELERATION: case QPC_TASK_MOMENTUM_RATE: return 6;
    case QPC_TASK_ANGULAR_ACCELERATION: case QPC_TASK_LINEAR_ACCELERATION:
    case QPC_TASK_POINT_ACCELERATION: case QPC_TASK_LINEAR_MOMENTUM_RATE: return 3;
    case QPC_TASK_JOINT_ACCELERATION:
      if (joint < 0 || joint >= m->nb) return -1;
      return m->nvj[joint];
  }
  return -1;
}

namespace {
struct Builder {
  std::vector<int> it;
  std::vector<double> dt;
  QpcDims *D;
  void ibegin(int id) { D->it[id] = (int)it.size(); }
  void dbegin(int id) { D->dt[id] = (int)dt.size(); }
};
int odd(int x) { return x | 1; }
} // namespace

int qpc_compile_program(qpc_controller *c) {
  const qpc_model *m = c->model;
  QpcProgram &P = c->prog;
  QpcDims &D = P.dims;
  std::memset(&D, 0, sizeof(D));
  Builder B;
  B.D = &D;
  const int nb = m->nb, nv = m->nv, N = c->N;
  { // slices of the per-QP `desired` record: the task dimension, or a trajectory record for a task with the SE3 PD law
    int off = 0;
    for (auto &t : c->tasks) { t.des_off = off; off += qpc_task_desired_len(t); }
    c->n_desired = off;
  }
  D.nb = nb; D.nq = m->nq; D.nv = nv; D.N = N; D.nc = (int)c->contacts.size();
  D.floating = c->floating;
  D.fv0 = c->floating >= 0 ? m->voff[c->floating] : -1;
  D.n_desired = c->n_desired; D.n_tw = c->n_tw; D.n_surf = c->n_surf;
  D.ve_exact_ok = 1;
  for (auto &t : c->tasks)
    if (t.wkind == QPC_WEIGHT_MATRIX || t.wkind == QPC_WEIGHT_MATRIX_PER_QP) D.ve_exact_ok = 0; // eigenvalues not on the diagonal
  if (c->floating >= 0 && m->jtype[c->floating] != QPC_JOINT_QUAT_FLOATING) {
    qpc_set_error("floating_body must be the successor of a QuaternionFloating joint");
    return QPC_ERR_INVALID_ARGUMENT;
  }
  D.nw = c->floating >= 0 ? 6 : 0;
  std::vector<double> law_tab;
  // ---- pass 1: pins (hard JointAccelerationTask on not-yet-pinned velocity indices)
  std::vector<int> pin_des(nv, -1);
  std::vector<char> task_is_pin(c->tasks.size(), 0);
  for (size_t ti = 0; ti < c->tasks.size(); ti++) {
    const QpcTask &t = c->tasks[ti];
    if (t.kind != QPC_TASK_JOINT_ACCELERATION || t.wkind != QPC_WEIGHT_HARD) continue;
    bool fresh = true;
    for (int k = 0; k < t.dim; k++) fresh = fresh && pin_des[m->voff[t.joint] + k] < 0;
    if (!fresh) continue; // duplicate pin stays a generic (possibly inconsistent) row block
    for (int k = 0; k < t.dim; k++) pin_des[m->voff[t.joint] + k] = t.des_off + k;
    task_is_pin[ti] = 1;
  }
  // A hard Angular / Linear / SpatialAccelerationTask of the FLOATING body w.r.t. the world, expressed in that body's own frame (the
  // pelvis task of StandingController, standing.jl:43-44), is a pin too: the velocity coordinates of a QuaternionFloating joint ARE
  // the body-frame twist, so the task's Jacobian is the matching rows of [I6 0 ... 0] for every state and its bias vanishes (the
  // body's twist is its own joint twist: T x T = 0): vdot[fv0 + r] = desired_r.  For Atlas: three rows and three free columns fewer,
  // the leading Schur block 18 -> 15 fits ONE 16 x 16 matrix-core tile.  Only when the desired value is known before the kinematics
  // (caller input, a constant, or the orientation law of this very body on the world, which desired_early evaluates from q, v alone).
  if (c->floating >= 0 && !QPC_EXP_ENV("QPC_NO_BASE_PIN")) {
    for (size_t ti = 0; ti < c->tasks.size(); ti++) {
      const QpcTask &t = c->tasks[ti];
      const bool ang = t.kind == QPC_TASK_ANGULAR_ACCELERATION, lin = t.kind == QPC_TASK_LINEAR_ACCELERATION,
                 spa = t.kind == QPC_TASK_SPATIAL_ACCELERATION;
      if (!(ang || lin || spa) || t.wkind != QPC_WEIGHT_HARD || t.base != QPC_WORLD || t.target != c->floating || t.frame != t.target ||
          m->parent[t.target] >= 0) // (a floating joint hanging on a moving body: its ancestors' joints enter the Jacobian)
        continue;
      const bool root_law = ang && t.law == QPC_LAW_ORIENTATION_PD;
      if (!(t.law == QPC_LAW_INPUT || t.law == QPC_LAW_CONSTANT || root_law)) continue;
      const int v0 = D.fv0 + (lin ? 3 : 0); // rows of a spatial task: angular first, like the joint's velocity coordinates
      bool fresh = true;
      for (int k = 0; k < t.dim; k++) fresh = fresh && pin_des[v0 + k] < 0;
      if (!fresh) continue;
      for (int k = 0; k < t.dim; k++) pin_des[v0 + k] = t.des_off + k;
      task_is_pin[ti] = 1;
    }
  }
  // ---- pass 1.5: which hard 6-row SpatialAccelerationTasks can the null-space presolve eliminate (QpcDims::presolve; the layout is
  // set up further down)?  Exactly six free one-DoF velocity coordinates on the task's path outside the floating joint, claimed by no
  // other block.  The host owns the ORDER of the free columns and of the row blocks -- neither is visible at the ABI -- so it lists
  // the columns that stay first, then the six columns of every eliminated block side by side, and puts the eliminated tasks first
  // among the row blocks: the regular layout the device code indexes by arithmetic, whatever the model's joint numbering and the
  // order in which the tasks were registered (standing.jl:37-38 iterates a Dict: unspecified).
  std::vector<size_t> elim_task;
  std::vector<std::vector<int>> elim_k;
  if (!getenv("QPC_NO_PRESOLVE") && c->floating >= 0 && !c->contacts.empty() && D.ve_exact_ok) {
    std::vector<char> claimed(nv, 0);
    for (size_t ti = 0; ti < c->tasks.size() && elim_task.size() < 4; ti++) {
      const QpcTask &t = c->tasks[ti];
      if (task_is_pin[ti] || t.kind != QPC_TASK_SPATIAL_ACCELERATION || t.wkind != QPC_WEIGHT_HARD || t.dim != 6) continue;
      std::vector<char> mark(nb, 0);
      for (int b = t.target; b >= 0; b = m->parent[b]) mark[b] = 1;
      int lca = -1;
      for (int b = t.base; b >= 0; b = m->parent[b]) if (mark[b]) { lca = b; break; }
      std::vector<int> priv;
      bool ok = true;
      auto visit = [&](int b) {
        if (b == c->floating) return; // its free columns stay
        for (int k = 0; k < m->nvj[b]; k++) {
          const int kv = m->voff[b] + k;
          if (pin_des[kv] >= 0) continue; // pinned: a constant of the row
          if (m->nvj[b] != 1 || claimed[kv]) ok = false;
          priv.push_back(kv);
        }
      };
      for (int b = t.base; b >= 0 && b != lca; b = m->parent[b]) visit(b);
      for (int b = t.target; b >= 0 && b != lca; b = m->parent[b]) visit(b);
      if (!ok || priv.size() != 6) continue;
      std::sort(priv.begin(), priv.end());
      for (int kv : priv) claimed[kv] = 1;
      elim_task.push_back(ti);
      elim_k.push_back(priv);
    }
  }
  std::vector<int> free_of(nv, -1), free_k, pin_k, pin_d;
  {
    std::vector<char> in_block(nv, 0);
    for (auto &blk : elim_k) for (int kv : blk) in_block[kv] = 1;
    for (int k = 0; k < nv; k++) {
      if (pin_des[k] >= 0) { pin_k.push_back(k); pin_d.push_back(pin_des[k]); }
      else if (!in_block[k]) { free_of[k] = (int)free_k.size(); free_k.push_back(k); }
    }
    for (auto &blk : elim_k) for (int kv : blk) { free_of[kv] = (int)free_k.size(); free_k.push_back(kv); }
  }
  D.nf = (int)free_k.size(); D.npin = (int)pin_k.size();
  // ---- pass 2: folded soft joint tasks and generic tasks
  std::vector<int> sj_f, sj_des, sj_tw;
  std::vector<double> sj_w;
  std::vector<int> generic;
  for (size_t ti = 0; ti < c->tasks.size(); ti++) {
    const QpcTask &t = c->tasks[ti];
    if (task_is_pin[ti]) continue;
    if (t.kind == QPC_TASK_JOINT_ACCELERATION &&
        (t.wkind == QPC_WEIGHT_SCALAR || t.wkind == QPC_WEIGHT_SCALAR_PER_QP)) {
      for (int k = 0; k < t.dim; k++) {
        int f = free_of[m->voff[t.joint] + k];
        if (f < 0) continue; // soft task on a pinned acceleration is a constant
        sj_f.push_back(f); sj_des.push_back(t.des_off + k);
        sj_tw.push_back(t.wkind == QPC_WEIGHT_SCALAR_PER_QP ? t.tw_off : -1);
        sj_w.push_back(t.wkind == QPC_WEIGHT_SCALAR ? t.W[0] : 0.0);
      }
      continue;
    }
    generic.push_back((int)ti);
  }
  { // the eliminable tasks first (in block order), the others behind them in registration order
    std::vector<int> g2;
    for (size_t ti : elim_task) g2.push_back((int)ti);
    for (int ti : generic) if (std::find(elim_task.begin(), elim_task.end(), (size_t)ti) == elim_task.end()) g2.push_back(ti);
    generic.swap(g2);
  }
  // regularize!(c, joint, weight::Parameter) (qpc_regularize_per_qp): w |vdot_k|^2 is a folded soft joint task with desired 0
  // (IT_SJ_DES = -1) and a per-QP weight; on a pinned acceleration it is a constant of the objective
  for (int k = 0; k < nv; k++)
    if (c->reg_tw[k] >= 0 && free_of[k] >= 0) { sj_f.push_back(free_of[k]); sj_des.push_back(-1); sj_tw.push_back(c->reg_tw[k]); sj_w.push_back(0.0); }
  D.nsj = (int)sj_f.size(); D.ntask = (int)generic.size();
  // rows and slacks
  int m1 = 0, ne = 0, wsize = 0;
  std::vector<int> row0(generic.size()), slack0(generic.size(), -1), woff(generic.size(), -1);
  for (size_t g = 0; g < generic.size(); g++) {
    const QpcTask &t = c->tasks[generic[g]];
    row0[g] = m1; m1 += t.dim;
    if (t.wkind != QPC_WEIGHT_HARD) { slack0[g] = ne; ne += t.dim; woff[g] = wsize; wsize += t.dim * t.dim; }
  }
  D.m1 = m1; D.m = m1 + D.nw; D.ne = ne; D.wsize = wsize;
  D.nrho = (D.nw > 0) ? D.nc * N : 0; // without wrench balance the rho decouple and stay 0
  // leading dimensions of M and S: even (16-byte row starts, so a row loads as 16-byte pairs) and not a multiple of 8
  // doubles (rows read side by side by consecutive lanes would share banks with a period of 2 or 4 lanes)
  auto lead = [](int n) { int s = (n > 0 ? n : 1); s += s & 1; if (s % 8 == 0) s += 2; return s; };
  D.nfs = lead(D.nf); D.ms = lead(D.m); D.m1s = odd(m1 > 0 ? m1 : 1);

  // ---- integer tables
  B.ibegin(IT_PARENT); for (int i = 0; i < nb; i++) B.it.push_back(m->parent[i]);
  B.ibegin(IT_JTYPE); for (int i = 0; i < nb; i++) B.it.push_back(m->jtype[i]);
  B.ibegin(IT_QOFF); for (int i = 0; i < nb; i++) B.it.push_back(m->qoff[i]);
  B.ibegin(IT_VOFF); for (int i = 0; i < nb; i++) B.it.push_back(m->voff[i]);
  B.ibegin(IT_NVJ); for (int i = 0; i < nb; i++) B.it.push_back(m->nvj[i]);
  {
    std::vector<int> ptr(nb + 1, 0), idx;
    int maxdepth = 0;
    for (int i = 0; i < nb; i++) {
      int d = 0;
      for (int a = m->parent[i]; a >= 0; a = m->parent[a]) { idx.push_back(a); d++; }
      ptr[i + 1] = (int)idx.size();
      maxdepth = std::max(maxdepth, d);
    }
    D.maxdepth = maxdepth;
    B.ibegin(IT_ANC_PTR); B.it.insert(B.it.end(), ptr.begin(), ptr.end());
    B.ibegin(IT_ANC_IDX); B.it.insert(B.it.end(), idx.begin(), idx.end());
    std::vector<std::vector<int>> desc(nb);
    for (int i = 0; i < nb; i++)
      for (int a = i; a >= 0; a = m->parent[a]) desc[a].push_back(i);
    // opt-in (QPC_DFS_SCAN=1): +1.4 % throughput, but the difference of two prefix sums that both hold the wrenches of the heavy
    // bodies cancels ~1e-12 N m into the torque of a light distal link, which forward dynamics turns into 2e-9 rad/s^2 there
    // (tests/test_physical_consistency.py asserts 5e-10)
    D.dfs = (nb <= 32 && QPC_EXP_ENV("QPC_DFS_SCAN")) ? 1 : 0; // (desc[a] is ascending: contiguous iff last - first + 1 == size)
    for (int i = 0; i < nb; i++)
      if (desc[i].front() != i || desc[i].back() - i + 1 != (int)desc[i].size()) D.dfs = 0;
    std::vector<int> dptr(nb + 1, 0), didx;
    for (int i = 0; i < nb; i++) { didx.insert(didx.end(), desc[i].begin(), desc[i].end()); dptr[i + 1] = (int)didx.size(); }
    B.ibegin(IT_DESC_PTR); B.it.insert(B.it.end(), dptr.begin(), dptr.end());
    B.ibegin(IT_DESC_IDX); B.it.insert(B.it.end(), didx.begin(), didx.end());
  }
  B.ibegin(IT_VBODY);
  for (int i = 0; i < nb; i++) for (int k = 0; k < m->nvj[i]; k++) B.it.push_back(i);
  B.ibegin(IT_VCOMP);
  for (int i = 0; i < nb; i++) for (int k = 0; k < m->nvj[i]; k++) B.it.push_back(k);
  B.ibegin(IT_FREE_OF); B.it.insert(B.it.end(), free_of.begin(), free_of.end());
  B.ibegin(IT_FREE_K); B.it.insert(B.it.end(), free_k.begin(), free_k.end());
  B.ibegin(IT_PIN_K); B.it.insert(B.it.end(), pin_k.begin(), pin_k.end());
  B.ibegin(IT_PIN_DES); B.it.insert(B.it.end(), pin_d.begin(), pin_d.end());
  B.ibegin(IT_SJ_F); B.it.insert(B.it.end(), sj_f.begin(), sj_f.end());
  B.ibegin(IT_SJ_DES); B.it.insert(B.it.end(), sj_des.begin(), sj_des.end());
  B.ibegin(IT_SJ_TW); B.it.insert(B.it.end(), sj_tw.begin(), sj_tw.end());
  auto push_task_field = [&](int id, auto getter) {
    B.ibegin(id);
    for (size_t g = 0; g < generic.size(); g++) B.it.push_back(getter(g, c->tasks[generic[g]]));
  };
  push_task_field(IT_T_KIND, [](size_t, const QpcTask &t) { return t.kind; });
  push_task_field(IT_T_BASE, [](size_t, const QpcTask &t) { return t.base; });
  push_task_field(IT_T_TARGET, [](size_t, const QpcTask &t) { return t.target; });
  push_task_field(IT_T_FRAME, [](size_t, const QpcTask &t) { return t.frame; });
  push_task_field(IT_T_DIM, [](size_t, const QpcTask &t) { return t.dim; });
  push_task_field(IT_T_ROW0, [&](size_t g, const QpcTask &) { return row0[g]; });
  push_task_field(IT_T_DES, [](size_t, const QpcTask &t) { return t.des_off; });
  push_task_field(IT_T_WKIND, [](size_t, const QpcTask &t) { return t.wkind; });
  push_task_field(IT_T_TW, [](size_t, const QpcTask &t) { return t.tw_off; });
  push_task_field(IT_T_SLACK0, [&](size_t g, const QpcTask &) { return slack0[g]; });
  push_task_field(IT_T_WOFF, [&](size_t g, const QpcTask &) { return woff[g]; });
  // path sign tables
  B.ibegin(IT_T_SGN);
  for (size_t g = 0; g < generic.size(); g++) {
    const QpcTask &t = c->tasks[generic[g]];
    std::vector<int> sgn(nv, 0);
    if (t.kind == QPC_TASK_JOINT_ACCELERATION) {
      for (int k = 0; k < t.dim; k++) sgn[m->voff[t.joint] + k] = 1;
    } else if (t.kind == QPC_TASK_MOMENTUM_RATE || t.kind == QPC_TASK_LINEAR_MOMENTUM_RATE) {
      for (int k = 0; k < nv; k++) sgn[k] = 1;
    } else {
      std::vector<char> mark(nb, 0);
      for (int b = t.target; b >= 0; b = m->parent[b]) mark[b] = 1;
      int lca = -1;
      for (int b = t.base; b >= 0; b = m->parent[b]) if (mark[b]) { lca = b; break; }
      for (int b = t.base; b >= 0 && b != lca; b = m->parent[b])
        for (int k = 0; k < m->nvj[b]; k++) sgn[m->voff[b] + k] = -1;
      for (int b = t.target; b >= 0 && b != lca; b = m->parent[b])
        for (int k = 0; k < m->nvj[b]; k++) sgn[m->voff[b] + k] = 1;
    }
    B.it.insert(B.it.end(), sgn.begin(), sgn.end());
  }
  B.ibegin(IT_ROW_TASK);
  for (size_t g = 0; g < generic.size(); g++) for (int a = 0; a < c->tasks[generic[g]].dim; a++) B.it.push_back((int)g);
  for (int a = 0; a < D.nw; a++) B.it.push_back(-1);
  B.ibegin(IT_ROW_COMP);
  for (size_t g = 0; g < generic.size(); g++) for (int a = 0; a < c->tasks[generic[g]].dim; a++) B.it.push_back(a);
  for (int a = 0; a < D.nw; a++) B.it.push_back(a);
  B.ibegin(IT_C_BODY); for (auto &k : c->contacts) B.it.push_back(k.body);
  B.ibegin(IT_C_PERQP); for (auto &k : c->contacts) B.it.push_back(k.per_qp);
  B.ibegin(IT_C_SURF); for (auto &k : c->contacts) B.it.push_back(k.surf_off);
  {
    std::vector<int> ptr(nb + 1, 0), idx;
    for (int i = 0; i < nb; i++) {
      for (size_t ci = 0; ci < c->contacts.size(); ci++) if (c->contacts[ci].body == i) idx.push_back((int)ci);
      ptr[i + 1] = (int)idx.size();
    }
    B.ibegin(IT_BC_PTR); B.it.insert(B.it.end(), ptr.begin(), ptr.end());
    B.ibegin(IT_BC_IDX); B.it.insert(B.it.end(), idx.begin(), idx.end());
  }
  B.ibegin(IT_SLACK_TASK);
  for (size_t g = 0; g < generic.size(); g++) if (slack0[g] >= 0) for (int a = 0; a < c->tasks[generic[g]].dim; a++) B.it.push_back((int)g);
  B.ibegin(IT_SLACK_COMP);
  for (size_t g = 0; g < generic.size(); g++) if (slack0[g] >= 0) for (int a = 0; a < c->tasks[generic[g]].dim; a++) B.it.push_back(a);
  {
    const int maxd = D.maxdepth > 0 ? D.maxdepth : 1;
    D.maxd = maxd;
    std::vector<int> tab((size_t)nb * maxd, -1), mask(2 * (size_t)nb, 0), pdes(nv, -1), crb_keep;
    for (int i = 0; i < nb; i++) {
      int sidx = 0;
      for (int a = m->parent[i]; a >= 0; a = m->parent[a]) tab[(size_t)i * maxd + sidx++] = a;
      for (int a = i; a >= 0; a = m->parent[a]) {
        if (i < 32) mask[2 * a] |= (int)(1u << i); else mask[2 * a + 1] |= (int)(1u << (i - 32));
      }
    }
    for (size_t p = 0; p < pin_k.size(); p++) pdes[pin_k[p]] = pin_d[p];
    B.ibegin(IT_ANC_TAB); B.it.insert(B.it.end(), tab.begin(), tab.end());
    {
      std::vector<int> crb(mask);
      D.crb_full_body = -1;
      bool com_law = false;
      for (auto &t : c->tasks) com_law = com_law || t.law == QPC_LAW_COM_PD;
      for (int i = 0; i < nb; i++) {
        bool has_free = false;   // (a COM law takes the linear momentum from the bodies' own momenta, not from A v: no pinned columns)
        (void)com_law;
        for (int k = 0; k < m->nvj[i]; k++) has_free = has_free || free_of[m->voff[i] + k] >= 0;
        unsigned long long mk = ((unsigned long long)(unsigned)mask[2 * i + 1] << 32) | (unsigned)mask[2 * i];
        const unsigned long long all = nb >= 64 ? ~0ull : ((1ull << nb) - 1);
        if (!has_free) { crb[2 * i] = crb[2 * i + 1] = 0; }
        else if (mk == all && D.crb_full_body < 0) { D.crb_full_body = i; crb[2 * i] = crb[2 * i + 1] = 0; }
      }
      B.ibegin(IT_CRB_MASK); B.it.insert(B.it.end(), crb.begin(), crb.end());
      crb_keep = crb;
    }
    { // packed per-lane metadata: lane = body i (words 0-2, 6..) and lane = velocity index k (words 3-5)
      const int aw = (maxd + 3) / 4, rowlen = 6 + aw;
      std::vector<int> meta((size_t)64 * rowlen, 0);
      for (int l = 0; l < 64; l++) {
        int *row = &meta[(size_t)l * rowlen];
        if (l < nb) {
          row[0] = (m->jtype[l] & 3) | (m->qoff[l] << 2) | (m->voff[l] << 9) | (m->nvj[l] << 16);
          row[1] = crb_keep[2 * l]; row[2] = crb_keep[2 * l + 1];
          for (int sidx = 0; sidx < maxd; sidx++) row[6 + sidx / 4] |= ((tab[(size_t)l * maxd + sidx] + 1) & 255) << (8 * (sidx % 4));
        }
        if (l < nv) {
          int vb = -1;
          for (int b = 0; b < nb; b++) if (l >= m->voff[b] && l < m->voff[b] + m->nvj[b]) vb = b;
          row[3] = vb & 127;
          row[4] = mask[2 * vb]; row[5] = mask[2 * vb + 1];
        }
        if (l < nb) // bits 8..31 of word 3: contacts 0..23 attached to body l (the RNEA subtracts their wrenches)
          for (size_t ci = 0; ci < c->contacts.size() && ci < 24; ci++)
            if (c->contacts[ci].body == l) row[3] |= (int)(1u << (8 + ci));
      }
      // stored word-major ([word][lane]): one load instruction of the wave reads 64 consecutive ints
      std::vector<int> metat((size_t)64 * rowlen, 0);
      for (int l = 0; l < 64; l++)
        for (int e = 0; e < rowlen; e++) metat[(size_t)e * 64 + l] = meta[(size_t)l * rowlen + e];
      B.ibegin(IT_META); B.it.insert(B.it.end(), metat.begin(), metat.end());
    }
    B.ibegin(IT_DESC_MASK); B.it.insert(B.it.end(), mask.begin(), mask.end());
    { // simulation step (qpc_sim.h): per body lane {children mask lo, hi, height above the leaves, depth below the
      // roots}, then one extra row {max height, max depth, 0, 0}
      std::vector<int> sim(4 * (size_t)(nb + 1), 0), height(nb, 0);
      int maxh = 0, maxdp = 0;
      for (int i = nb - 1; i >= 0; i--) { // parent[i] < i: children are visited before their parents
        const int pa = m->parent[i];
        if (pa >= 0) {
          if (i < 32) sim[4 * pa] |= (int)(1u << i); else sim[4 * pa + 1] |= (int)(1u << (i - 32));
          height[pa] = std::max(height[pa], height[i] + 1);
        }
      }
      for (int i = 0; i < nb; i++) {
        int dp = 0;
        for (int a = m->parent[i]; a >= 0; a = m->parent[a]) dp++;
        sim[4 * i + 2] = height[i]; sim[4 * i + 3] = dp;
        maxh = std::max(maxh, height[i]); maxdp = std::max(maxdp, dp);
      }
      sim[4 * nb] = maxh; sim[4 * nb + 1] = maxdp;
      B.ibegin(IT_SIM_TREE); B.it.insert(B.it.end(), sim.begin(), sim.end());
    }
    B.ibegin(IT_PDES_K); B.it.insert(B.it.end(), pdes.begin(), pdes.end());
    // packed assembly codes: kind(3) | sg+1(2) | a(3) | k(7) | frame+1(7) | t(6)   (kind 7 = wrench row)
    std::vector<int> rcode(D.m, 0), rdes(D.m, -1), rt(D.m, -1), rpink(D.m, -1);
    struct Item { int cost, c0, c1; };
    std::vector<Item> items;
    for (int r = 0; r < D.m; r++) {
      int g = -1, a = r - m1;
      if (r < m1) { for (size_t gg = 0; gg < generic.size(); gg++) if (r >= row0[gg] && r < row0[gg] + c->tasks[generic[gg]].dim) g = (int)gg; a = r - row0[g]; }
      const QpcTask *t = g >= 0 ? &c->tasks[generic[g]] : nullptr;
      int kind = t ? t->kind : 7, frame = t ? t->frame : -1, base = t ? t->base : -1, target = t ? t->target : -1;
      rcode[r] = kind | (a << 3) | ((frame + 1) << 6) | ((base + 1) << 13) | ((target + 1) << 20);
      rdes[r] = t ? t->des_off + a : -1;
      rt[r] = g;
      // one assembly item per (row block, free column): emitted with the block's first row
      for (int f = 0; f < D.nf && a == 0; f++) {
        int k = free_k[f], sg = 0, aux = frame + 1, cost = 1;
        const int dim = t ? t->dim : 6;
        if (!t) sg = 1;
        else if (t->kind == QPC_TASK_JOINT_ACCELERATION) {
          const int comp = k - m->voff[t->joint];
          sg = (comp >= 0 && comp < t->dim) ? 1 : 0;
          aux = sg ? comp + 1 : 0; // the one row of the block that carries the 1
        }
        else if (t->kind == QPC_TASK_MOMENTUM_RATE || t->kind == QPC_TASK_LINEAR_MOMENTUM_RATE) { sg = 1; cost = 2; }
        else {
          std::vector<char> mark(nb, 0);
          for (int b = t->target; b >= 0; b = m->parent[b]) mark[b] = 1;
          int lca = -1;
          for (int b = t->base; b >= 0; b = m->parent[b]) if (mark[b]) { lca = b; break; }
          int kb = -1;
          for (int b = 0; b < nb; b++) if (k >= m->voff[b] && k < m->voff[b] + m->nvj[b]) kb = b;
          for (int b = t->base; b >= 0 && b != lca; b = m->parent[b]) if (b == kb) sg = -1;
          for (int b = t->target; b >= 0 && b != lca; b = m->parent[b]) if (b == kb) sg = 1;
          if (sg != 0) cost = t->kind == QPC_TASK_POINT_ACCELERATION ? 12 : 8; // one Jacobian column transform
        }
        Item it_;
        it_.cost = cost;
        it_.c0 = kind | ((sg + 1) << 3) | (k << 5) | (aux << 12) | ((g >= 0 ? g : 0) << 19);
        it_.c1 = r | (dim << 7) | (f << 11);
        items.push_back(it_);
      }
      if (t && t->kind == QPC_TASK_JOINT_ACCELERATION) {
        int k = m->voff[t->joint] + a;
        if (free_of[k] < 0) rpink[r] = k;
      }
    }
    // heaviest items first: lane l takes items l, l + 64, ... so every round is balanced
    std::stable_sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.cost > y.cost; });
    D.n_items = (int)items.size();
    B.ibegin(IT_MCODE);
    for (auto &it_ : items) { B.it.push_back(it_.c0); B.it.push_back(it_.c1); }
    B.it.push_back(0);
    B.ibegin(IT_RCODE); B.it.insert(B.it.end(), rcode.begin(), rcode.end());
    B.ibegin(IT_R_DES); B.it.insert(B.it.end(), rdes.begin(), rdes.end());
    B.ibegin(IT_R_T); B.it.insert(B.it.end(), rt.begin(), rt.end());
    B.ibegin(IT_R_PINK); B.it.insert(B.it.end(), rpink.begin(), rpink.end());
  }
  // ---- desired laws (qpc_task_set_law; standing.jl:58-89 evaluated on the device)
  {
    std::vector<int> law_i(3 * (size_t)c->n_desired, 0);
    std::vector<double> law_d(3 * (size_t)c->n_desired, 0.0);
    D.n_des_inputs = 0; D.n_des_late = 0; D.law_mask = 0;
    for (auto &t : c->tasks)
      for (int a = 0; a < t.dim; a++) {
        const int i = t.des_off + a;
        D.law_mask |= 1 << t.law; // (refined below for the root orientation law)
        law_i[3 * i] = t.law; law_d[3 * i] = t.kp; law_d[3 * i + 1] = t.kd;
        switch (t.law) {
          case QPC_LAW_INPUT: D.n_des_inputs++; break;
          case QPC_LAW_CONSTANT: law_d[3 * i + 2] = t.ref[a]; break;
          case QPC_LAW_JOINT_PD: law_i[3 * i + 1] = m->qoff[t.joint]; law_i[3 * i + 2] = m->voff[t.joint]; law_d[3 * i + 2] = t.ref[0]; break;
          case QPC_LAW_COM_PD: law_i[3 * i + 1] = a; law_d[3 * i + 2] = t.ref[a]; D.n_des_late++; break;
          case QPC_LAW_ORIENTATION_PD:
            law_i[3 * i + 1] = t.target; law_i[3 * i + 2] = a;
            if (t.target == c->floating && c->floating >= 0 && m->parent[t.target] < 0 && !QPC_EXP_ENV("QPC_NO_BASE_PIN")) {
              law_i[3 * i] = QPC_LAW_ORIENTATION_PD_ROOT; // from the joint's own coordinates, before the kinematics
              D.law_mask |= 1 << QPC_LAW_ORIENTATION_PD_ROOT;
            } else D.n_des_late++;
            break;
          case QPC_LAW_SE3_PD: // entry a: body / base / component packed, offset of the task's record; per-entry gains
            law_i[3 * i + 1] = (t.target + 1) | ((t.base + 1) << 8) | (a << 16) | ((t.ref.size() >= 3 ? (int)t.ref[2] : 1) << 20); // (bits 20..: pieces)
            law_i[3 * i + 2] = t.des_off;
            if (a >= 3) { law_d[3 * i] = t.ref[0]; law_d[3 * i + 1] = t.ref[1]; }
            D.n_des_late++;
            break;
        }
      }
    D.law_mask = 0;
    for (size_t i = 0; i < (size_t)c->n_desired; i++) D.law_mask |= 1 << law_i[3 * i];
    for (auto &t : c->tasks) // the rest of a trajectory record are plain per-QP inputs
      if (t.law == QPC_LAW_SE3_PD) D.n_des_inputs += qpc_task_desired_len(t) - t.dim;
    B.ibegin(IT_LAW); B.it.insert(B.it.end(), law_i.begin(), law_i.end());
    B.it.push_back(0);
    law_tab.swap(law_d);
  }
  B.ibegin(IT_HOT);
  {
    std::vector<int> row_slack(D.m, -1), srow, swoff, sdim, ss0;
    for (size_t g = 0; g < generic.size(); g++) {
      const QpcTask &t = c->tasks[generic[g]];
      if (slack0[g] < 0) continue;
      for (int a = 0; a < t.dim; a++) {
        row_slack[row0[g] + a] = slack0[g] + a;
        srow.push_back(row0[g] + a); swoff.push_back(woff[g] + a * t.dim); sdim.push_back(t.dim); ss0.push_back(slack0[g]);
      }
    }
    B.it.insert(B.it.end(), row_slack.begin(), row_slack.end());
    B.it.insert(B.it.end(), srow.begin(), srow.end());
    B.it.insert(B.it.end(), swoff.begin(), swoff.end());
    B.it.insert(B.it.end(), sdim.begin(), sdim.end());
    B.it.insert(B.it.end(), ss0.begin(), ss0.end());
  }
  B.it.push_back(0); // never empty

  // ---- double tables
  // per-body constants component-major ([component][body]): lane = body reads them, so one load instruction of the
  // wave covers nb consecutive doubles instead of nb records 24 / 72 / 48 bytes apart
  auto by_component = [&](const std::vector<double> &v, int dim) {
    for (int k = 0; k < dim; k++) for (int i = 0; i < nb; i++) B.dt.push_back(v[(size_t)i * dim + k]);
  };
  B.dbegin(DT_AXIS); by_component(m->axis, 3);
  B.dbegin(DT_RF); by_component(m->Rf, 9);
  B.dbegin(DT_PF); by_component(m->pf, 3);
  B.dbegin(DT_MASS); B.dt.insert(B.dt.end(), m->mass.begin(), m->mass.end());
  B.dbegin(DT_COM); by_component(m->com, 3);
  B.dbegin(DT_ICOM); by_component(m->icom, 6);
  B.dbegin(DT_REG2); for (int k = 0; k < nv; k++) B.dt.push_back(2.0 * c->reg[k]);
  B.dbegin(DT_GRAV); for (int k = 0; k < 3; k++) B.dt.push_back(m->gravity[k]);
  B.dt.push_back(m->mtot);
  B.dbegin(DT_SJ_W); B.dt.insert(B.dt.end(), sj_w.begin(), sj_w.end());
  B.dbegin(DT_T_POINT);
  for (size_t g = 0; g < generic.size(); g++) for (int k = 0; k < 3; k++) B.dt.push_back(c->tasks[generic[g]].point[k]);
  B.dbegin(DT_T_W);
  for (size_t g = 0; g < generic.size(); g++) {
    const QpcTask &t = c->tasks[generic[g]];
    if (slack0[g] < 0) continue;
    // symmetric Hessian block of the slack: 2*sym(W) (P = 2x written weights)
    for (int a = 0; a < t.dim; a++)
      for (int b = 0; b < t.dim; b++) {
        double w = 0.0;
        if (t.wkind == QPC_WEIGHT_SCALAR) w = (a == b) ? 2.0 * t.W[0] : 0.0;
        else if (t.wkind == QPC_WEIGHT_MATRIX) w = t.W[a * t.dim + b] + t.W[b * t.dim + a];
        B.dt.push_back(w);
      }
  }
  B.dbegin(DT_C_POS); for (auto &k : c->contacts) for (int a = 0; a < 3; a++) B.dt.push_back(k.pos[a]);
  B.dbegin(DT_C_NORMAL); for (auto &k : c->contacts) for (int a = 0; a < 3; a++) B.dt.push_back(k.normal[a]);
  B.dbegin(DT_C_MU); for (auto &k : c->contacts) B.dt.push_back(k.mu);
  B.dt.push_back(0.0);
  B.dbegin(DT_LAW); B.dt.insert(B.dt.end(), law_tab.begin(), law_tab.end());
  B.dt.push_back(0.0);
  // ---- static Hessian blocks (no per-QP task weights): what the tick would otherwise re-derive for every QP
  {
    D.h_static = (c->n_tw == 0) ? 1 : 0;
    double hmax = 0.0, hmin = 1e300;
    B.dbegin(DT_HV0);
    for (int f = 0; f < D.nf; f++) {
      double hv = 2.0 * c->reg[free_k[f]];
      for (size_t si = 0; si < sj_f.size(); si++)
        if (sj_f[si] == f && sj_tw[si] < 0) hv += 2.0 * sj_w[si];
      B.dt.push_back(hv);
      hmax = std::max(hmax, std::fabs(hv)); hmin = std::min(hmin, hv);
    }
    B.dt.push_back(0.0);
    B.dbegin(DT_WINV0);
    std::vector<double> winv((size_t)wsize + 1, 0.0);
    for (size_t g = 0; g < generic.size(); g++) {
      const QpcTask &t = c->tasks[generic[g]];
      if (slack0[g] < 0) continue;
      const int d = t.dim;
      std::vector<double> a((size_t)d * d, 0.0), inv((size_t)d * d, 0.0);
      for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
          double w = 0.0;
          if (t.wkind == QPC_WEIGHT_SCALAR) w = (i == j) ? 2.0 * t.W[0] : 0.0;
          else if (t.wkind == QPC_WEIGHT_MATRIX) w = t.W[i * d + j] + t.W[j * d + i];
          a[(size_t)i * d + j] = w;
          inv[(size_t)i * d + j] = (i == j) ? 1.0 : 0.0;
        }
      for (int i = 0; i < d; i++) { hmax = std::max(hmax, std::fabs(a[(size_t)i * d + i])); hmin = std::min(hmin, a[(size_t)i * d + i]); }
      bool ok = true;
      for (int k = 0; k < d && ok; k++) { // Gauss-Jordan without pivoting (SPD blocks), like the device code
        const double piv = a[(size_t)k * d + k];
        if (!(std::fabs(piv) > 1e-300)) { ok = false; break; }
        for (int j = 0; j < d; j++) { a[(size_t)k * d + j] /= piv; inv[(size_t)k * d + j] /= piv; }
        for (int i = 0; i < d; i++) {
          if (i == k) continue;
          const double f_ = a[(size_t)i * d + k];
          for (int j = 0; j < d; j++) { a[(size_t)i * d + j] -= f_ * a[(size_t)k * d + j]; inv[(size_t)i * d + j] -= f_ * inv[(size_t)k * d + j]; }
        }
      }
      if (!ok) D.h_static = 0; // singular static weight: leave it to the per-QP (regularised) path
      for (int e = 0; e < d * d; e++) winv[(size_t)woff[g] + e] = inv[(size_t)e];
    }
    B.dt.insert(B.dt.end(), winv.begin(), winv.end());
    D.h_static_max = hmax;
    D.h_static_min = hmin < 1e300 ? hmin : 0.0;
    if (D.nf == 0 && wsize == 0) D.h_static = 0;
  }

  // ---- k_tab: eigen-decompositions of the rho block of the Hessian on every free set of a contact's rays.  With N = 4 rays
  // the block is 2 w_c B'B, B'B_ij = (1 + mu^2 cos(theta_i - theta_j)) / (1 + mu^2) (contacts.jl:16-23): the same matrix for
  // every contact when they share mu, whatever the normal and the weight.  Per 4-bit free pattern: lambda[4], V[4][4]
  // (V[i][k] = component i of eigenvector k; pinned rays have zero rows, unused eigen slots zero columns and lambda = 1).
  {
    D.k_tab = 0;
    if (B.dt.size() & 1) B.dt.push_back(0.0); // 16-byte rows
    B.dbegin(DT_KTAB);
    bool ok = D.N == 4 && D.nrho > 0 && D.nrho <= 64 && D.nc <= 16 && !QPC_EXP_ENV("QPC_NO_KTAB");
    // one table per distinct friction coefficient; with several, the per-contact table index needs the upper 8 slots of MS_KS
    std::vector<double> mus;
    std::vector<int> cls;
    for (auto &k : c->contacts) {
      if ((k.per_qp & QPC_CONTACT_SURFACE_PER_QP) || !(k.mu >= 1e-3) || !(k.mu <= 1e3)) ok = false;
      size_t m = 0;
      while (m < mus.size() && mus[m] != k.mu) m++;
      if (m == mus.size()) mus.push_back(k.mu);
      cls.push_back((int)m);
    }
    if (mus.size() > 1 && D.nc > 8) ok = false;
    D.k_classes = 0;
    for (int i = 0; i < 16; i++) D.k_class[i] = 0;
    if (ok) {
      D.k_classes = (int)mus.size();
      for (int i = 0; i < D.nc; i++) D.k_class[i] = cls[(size_t)i];
    }
    for (size_t mi = 0; ok && mi < mus.size(); mi++) {
      const double mu = mus[mi];
      double G[4][4];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) G[i][j] = (1.0 + mu * mu * std::cos((i - j) * 1.5707963267948966)) / (1.0 + mu * mu);
      for (int pat = 0; pat < 16; pat++) {
        int idx[4], n = 0;
        for (int i = 0; i < 4; i++) if (pat & (1 << i)) idx[n++] = i;
        double A[4][4] = {{0}}, V[4][4] = {{0}};
        for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) A[i][j] = G[idx[i]][idx[j]]; V[i][i] = 1.0; }
        for (int sweep = 0; sweep < 60; sweep++) { // cyclic Jacobi: A -> diag, V = eigenvectors (columns)
          double offd = 0.0;
          for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) offd += A[i][j] * A[i][j];
          if (offd < 1e-60) break;
          for (int pp = 0; pp < n; pp++)
            for (int q = pp + 1; q < n; q++) {
              if (std::fabs(A[pp][q]) < 1e-300) continue;
              const double th = 0.5 * std::atan2(2.0 * A[pp][q], A[q][q] - A[pp][pp]), cs = std::cos(th), sn = std::sin(th);
              for (int k2 = 0; k2 < n; k2++) { const double a = A[k2][pp], b = A[k2][q]; A[k2][pp] = cs * a - sn * b; A[k2][q] = sn * a + cs * b; }
              for (int k2 = 0; k2 < n; k2++) { const double a = A[pp][k2], b = A[q][k2]; A[pp][k2] = cs * a - sn * b; A[q][k2] = sn * a + cs * b; }
              for (int k2 = 0; k2 < n; k2++) { const double a = V[k2][pp], b = V[k2][q]; V[k2][pp] = cs * a - sn * b; V[k2][q] = sn * a + cs * b; }
            }
        }
        double lmax = 0.0;
        for (int e = 0; e < n; e++) lmax = std::max(lmax, A[e][e]);
        for (int e = 0; e < 4; e++) { // B'B is PSD: the null direction of the all-free block gets exactly 0 (-> K = 1 / eps along it)
          double l = e < n ? A[e][e] : 1.0;
          if (e < n && l < 1e-12 * lmax) l = 0.0;
          B.dt.push_back(l);
        }
        double Vf[4][4] = {{0}};
        for (int i = 0; i < n; i++) for (int e = 0; e < n; e++) Vf[idx[i]][e] = V[i][e];
        for (int i = 0; i < 4; i++) for (int e = 0; e < 4; e++) B.dt.push_back(Vf[i][e]);
      }
      D.k_tab = 1;
    }
    // closed-form K rows (Solver::k_row_closed) instead of the table: opt-in.  Measured on MI355X: 27.8 against 28.8 M solves/s --
    // the table's four reciprocals run side by side and its loads hide behind them, the closed form is one longer dependent chain
    D.k_mu2 = (D.k_tab && D.k_classes == 1 && QPC_EXP_ENV("QPC_K_CLOSED")) ? mus[0] * mus[0] : -1.0;
    B.dt.push_back(0.0);
  }

  // ---- LDS layout with lifetime overlays (regions that are never live together share bytes).
  //   [ persistent | small assembly outputs | A | B ]
  //   A = { kinematics-only scratch: RL->IC, TJ->HB, CR }  U  { big assembly outputs: M, G, HR, HI, WS }
  //   B = { what the assembly still reads: H, inputs, T, S, AB, IW->AMOM }  U  { solver scratch }  U  { RNEA scratch }
  // What the inverse dynamics needs again after the solve (I_i, W0_i, S_k) is NOT kept in LDS: it is
  // lane-private and stays in the lane's registers (qpc_tick.h keep_iw_ / keep_sk_).
  // Every overlay below was validated by the emulator's race / uninitialised-read detector.
  const int nrho = D.nrho, mm = D.m;
  auto even = [](int x) { return x + (x & 1); };
  int off = 0;
  auto L = [&](int id, int n) { D.lds[id] = off; off = even(off + (n > 0 ? n : 0)); };
  // persistent for the whole tick
  L(L_VD, nv); L(L_MISC, MS_COUNT); L(L_HOT, D.m + 4 * ne);
  // small assembly outputs (written while the kinematics scratch is still live)
  L(L_D, mm); L(L_HV, D.nf); L(L_SQH, D.nf); L(L_GV, D.nf); L(L_W, wsize); L(L_WINV, wsize);
  // ---- region A
  const int a_base = off;
  L(L_RL, nb * 12); D.lds[L_IC] = D.lds[L_RL];                       // IC born after RL is dead
  L(L_TJ, nb * 6); D.lds[L_HB] = D.lds[L_TJ];                       // HB born after TJ is dead
  L(L_CR, nb * 6);
  const int a_kin_end = off;
  off = a_base;
  L(L_M, mm * D.nfs); L(L_G, 6 * QPC_GS(nrho)); L(L_HR, D.nc * QPC_CSH(N)); L(L_HI, nrho); L(L_WS, nrho);
  if (a_kin_end > off) off = a_kin_end;
  // ---- region B
  const int b_base = off;
  // -- early group (dead once the Schur complement build starts)
  L(L_H, nb * 12);
  L(L_Q, D.nq); L(L_V, nv); L(L_DES, D.n_desired); L(L_CW, D.nc); L(L_CMAX, D.nc); L(L_TW, D.n_tw); L(L_SURF, D.n_surf);
  D.lds[L_ACCP] = D.lds[L_Q]; // retired
  L(L_T, nb * 6); L(L_S, nv * 6); L(L_AB, nb * 6);
  { int iw = nb * 10, amom = 6 * nv; L(L_IW, iw > amom ? iw : amom); } // AMOM born after the last read of IW
  D.lds[L_AMOM] = D.lds[L_IW];
  D.lds[L_F] = D.lds[L_IW]; // retired (W0 lives in the spill slot)
  const int early_end = off;
  // -- solver group
  off = b_base;
  L(L_SM, mm * D.ms);
  D.lds[L_SV] = D.lds[L_SM];                                         // RNEA scratch born after the solve
  D.lds[L_BW] = D.lds[L_SV] + even(nb * 6);
  D.lds[L_CWR] = D.lds[L_BW] + even(nb * 6);
  D.lds[L_SVK] = D.lds[L_CWR] + even(D.nc * 6);
  { int late_end = D.lds[L_SVK] + even(nv * 6); if (late_end > off) off = late_end; }
  L(L_X2, m1 * 6); L(L_B22, 36); L(L_C22, 36); L(L_S22I, 36);
  D.lds[L_L22] = D.lds[L_C22]; // retired
  L(L_KR, D.nc * QPC_CS(N)); L(L_T1, mm); D.lds[L_T2] = D.lds[L_T1]; D.lds[L_T3] = D.lds[L_T1]; // T2/T3 retired
  L(L_XV, D.nf); L(L_XE, ne); L(L_XR, nrho); L(L_LAM, mm); L(L_U, nrho); L(L_WSP, nrho);
  const int xbase = off;
  // X (explicit L11^-1) is dead before the first iterate vector is written
  L(L_RV, D.nf); L(L_RE, ne); L(L_RR, nrho); L(L_RC, mm);
  L(L_DV, D.nf); L(L_DE, ne); L(L_DR, nrho); L(L_DL, mm);
  L(L_TR1, nrho); L(L_TR2, nrho); L(L_TE1, ne);
  const int vec_end = off;
  D.lds[L_X] = xbase;
  D.lds[L_Y] = xbase; // Y lives only inside refactor_rho, when the rhs/direction vectors are dead
  { int xe = xbase + even(m1 * D.m1s), ye = xbase + even(QPC_GS(nrho) * 6); off = vec_end; if (xe > off) off = xe; if (ye > off) off = ye; }
  D.lds[L_RED] = off; // unused on the device (wave reductions); the emulator has its own scratch
  if (early_end > off) off = early_end;
  // ---- null-space presolve (QpcDims::presolve): hard 6-row SpatialAccelerationTask blocks whose path holds exactly six free 1-DoF
  // columns outside the floating joint that no other eliminated block claims -- their rows are then [A_e (6 x 6, those columns) | B_e
  // (columns that stay)] and nothing else, and v_e = A_e^-1 (b_e - B_e v_b) for every state in which A_e is well conditioned.
  // The device code indexes by arithmetic, not by tables, so the layout has to be a regular one: the eliminated blocks are the first
  // row blocks, the six columns of a block are consecutive, the columns that stay are consecutive (a standing controller's foot tasks
  // registered first, as StandingController does, in either order, on a model that lists the joints of a leg one after the other).
  D.presolve = 0; D.pre_nf = D.nf; D.pre_m = D.m; D.pre_b0 = 0; D.pre_fold = 0;
  for (int k = 0; k < 4; k++) D.pre_c0[k] = 0;
  for (int k = 0; k < 6; k++) D.pre_wd[k] = 0;
  D.it[IT_HOT_S] = 0;
  D.lds[L_PA] = D.lds[L_PZ] = D.lds[L_PH] = D.lds[L_PC] = 0;
  QpcDims DS; // finished below (a copy of D with the solver's dimensions and array places)
  int s_lds[7] = {0, 0, 0, 0, 0, 0, 0}; // places of M, d, HV, SQH, GV, XV, HOT of the reduced problem
  if (!getenv("QPC_NO_PRESOLVE") && D.nw == 6 && D.nrho > 0 && D.ve_exact_ok && c->floating >= 0) {
    struct Elim { int g, row0; int cols[6]; };
    std::vector<Elim> el;
    std::vector<char> col_taken(D.nf, 0);
    for (size_t g = 0; g < generic.size() && el.size() < 4; g++) {
      const QpcTask &t = c->tasks[generic[g]];
      if (t.kind != QPC_TASK_SPATIAL_ACCELERATION || t.wkind != QPC_WEIGHT_HARD || t.dim != 6 || row0[g] != 6 * (int)el.size()) break;
      // path columns (as IT_T_SGN): joints between the lowest common ancestor of base and target and either of them
      std::vector<char> mark(nb, 0);
      for (int b = t.target; b >= 0; b = m->parent[b]) mark[b] = 1;
      int lca = -1;
      for (int b = t.base; b >= 0; b = m->parent[b]) if (mark[b]) { lca = b; break; }
      std::vector<int> priv;
      bool ok = true;
      auto visit = [&](int b) {
        if (b == c->floating) return; // its free columns stay
        for (int k = 0; k < m->nvj[b]; k++) {
          const int f = free_of[m->voff[b] + k];
          if (f < 0) continue;              // pinned: a constant of the row
          if (m->nvj[b] != 1 || col_taken[f]) ok = false;
          priv.push_back(f);
        }
      };
      for (int b = t.base; b >= 0 && b != lca; b = m->parent[b]) visit(b);
      for (int b = t.target; b >= 0 && b != lca; b = m->parent[b]) visit(b);
      if (!ok || priv.size() != 6) break;
      std::sort(priv.begin(), priv.end());
      if (priv[5] != priv[0] + 5) break;
      Elim e; e.g = (int)g; e.row0 = row0[g];
      for (int i = 0; i < 6; i++) { e.cols[i] = priv[i]; col_taken[priv[i]] = 1; }
      el.push_back(e);
    }
    const int n_e = (int)el.size(), nf_s = D.nf - 6 * n_e, m1_s = m1 - 6 * n_e, m_s = m1_s + D.nw;
    std::vector<int> bcols;
    for (int f = 0; f < D.nf; f++) if (!col_taken[f]) bcols.push_back(f);
    bool regular = n_e > 0 && nf_s >= 1 && nf_s <= 4 && n_e <= 4 && m_s * nf_s + m_s + nf_s * (nf_s + 1) / 2 + nf_s <= 64 &&
                   m_s * nf_s + nf_s <= 48 && m_s <= 16; // (lanes of the reduction phase; lanes of the whitening phase: M rows below 48, d from 48 on)
    for (int b = 1; b < (int)bcols.size(); b++) regular = regular && bcols[b] == bcols[0] + b;
    if (regular) {
      // the task rows that stay are all soft, their slack blocks are the first ones (slack s = row - 6 n_e), and no weight is a
      // per-QP matrix: fold them into the objective of the (few) unknowns that are left
      bool fold = m1_s > 0 && m1_s <= 6 && ne == m1_s && !QPC_EXP_ENV("QPC_NO_PRESOLVE_FOLD");
      for (size_t g = 0; g < generic.size(); g++)
        if (row0[g] >= 6 * n_e) fold = fold && slack0[g] == row0[g] - 6 * n_e;
      std::vector<int> hot_s;
      if (fold) hot_s.assign(D.nw, -1); // wrench rows only: no slacks
      else
      { // hot lookups of the reduced problem: the slack tables with the rows renumbered (row r -> r - 6 n_e)
        std::vector<int> row_slack(m_s, -1), srow, swoff, sdim, ss0;
        for (size_t g = 0; g < generic.size(); g++) {
          const QpcTask &t = c->tasks[generic[g]];
          if (slack0[g] < 0) continue;
          for (int a = 0; a < t.dim; a++) {
            row_slack[row0[g] + a - 6 * n_e] = slack0[g] + a;
            srow.push_back(row0[g] + a - 6 * n_e); swoff.push_back(woff[g] + a * t.dim); sdim.push_back(t.dim); ss0.push_back(slack0[g]);
          }
        }
        hot_s.insert(hot_s.end(), row_slack.begin(), row_slack.end());
        hot_s.insert(hot_s.end(), srow.begin(), srow.end());
        hot_s.insert(hot_s.end(), swoff.begin(), swoff.end());
        hot_s.insert(hot_s.end(), sdim.begin(), sdim.end());
        hot_s.insert(hot_s.end(), ss0.begin(), ss0.end());
      }
      auto lead2 = [](int n) { int s_ = (n > 0 ? n : 1); s_ += s_ & 1; if (s_ % 8 == 0) s_ += 2; return s_; };
      const int nfs_s = lead2(nf_s), nr = 6 * n_e;
      const int need = even(nr * (nf_s + 2)) + even(nf_s * (nf_s + 1) / 2 + nf_s) + even(nf_s * nf_s) + even(m_s * nfs_s) + even(m_s) +
                       4 * even(nf_s) + even(m_s + 4 * ne);
      auto per_cu = [](int doubles) { const int n = doubles > 0 ? (160 * 1024) / (doubles * 8) : 8; return n > 8 ? 8 : n; };
      if (per_cu(off + need) >= 1 && per_cu(off + need) >= per_cu(off) && b_base + even(m_s * nf_s) + even(m_s) <= off) {
        D.presolve = n_e; D.pre_nf = nf_s; D.pre_m = m_s; D.pre_b0 = bcols[0];
        for (int e = 0; e < n_e; e++) D.pre_c0[e] = el[e].cols[0];
        D.pre_fold = fold ? m1_s : 0;
        for (size_t g = 0; g < generic.size() && fold; g++)
          if (row0[g] >= 6 * n_e)
            for (int a = 0; a < c->tasks[generic[g]].dim; a++) D.pre_wd[row0[g] + a - 6 * n_e] = woff[g] + a * c->tasks[generic[g]].dim + a;
        D.lds[L_PA] = b_base; // region B is idle between the assembly and the solve
        L(L_PZ, nr * (nf_s + 2)); L(L_PH, nf_s * (nf_s + 1) / 2 + nf_s); L(L_PC, nf_s * nf_s);
        s_lds[0] = off; off = even(off + m_s * nfs_s);
        s_lds[1] = off; off = even(off + m_s);
        for (int k = 2; k < 6; k++) { s_lds[k] = off; off = even(off + nf_s); }
        s_lds[6] = off; off = even(off + m_s + 4 * ne);
        B.ibegin(IT_HOT_S); B.it.insert(B.it.end(), hot_s.begin(), hot_s.end()); B.it.push_back(0);
      }
    }
  }
  D.lds_doubles = off;
  if (const char *e = getenv("QPC_DUMP_LDS")) { // diagnostic: the overlay map (offsets in doubles)
    if (e[0] == '1') {
      fprintf(stderr, "[qpc] LDS map: total %d doubles; a_base %d a_kin_end %d b_base %d early_end %d xbase %d vec_end %d\n", off, a_base, a_kin_end, b_base, early_end, xbase, vec_end);
      for (int i = 0; i < L_COUNT; i++) fprintf(stderr, "[qpc]   lds[%d] = %d\n", i, D.lds[i]);
    }
  }
  D.n_int = (int)B.it.size(); D.n_dbl = (int)B.dt.size();
  // settings
  D.max_as_iters = c->settings.max_active_set_iters;
  D.max_refine = c->settings.max_refine_iters;
  // primal-dual sweeps before the finite phase B takes over.  Measured on the Atlas workload (mean
  // changes per QP / p99): 12 sweeps 6.92 / 36, 6: 6.74, 5: 6.41, 4: 6.08 / 18, 3: 6.82 -- QPs that have
  // not converged after 4 sweeps rarely do, and Goldfarb-Idnani from their working set is cheaper
  D.max_pdas = 4;
  if (const char *e = QPC_EXP_ENV("QPC_MAX_PDAS")) { const int v = atoi(e); if (v >= 0 && v <= 64) D.max_pdas = v; } // experiments
  D.eps_primal = c->settings.eps_primal_reg;
  D.tau_pivot = c->settings.eps_dual_reg;
  D.tol_bound = c->settings.tol_bound;
  D.tol_accept = 1e-7;
  if (c->precision == QPC_PRECISION_F32) {
    // fp32 tick (qpc_set_precision): every tolerance at the fp32 rounding floor (2^-24 = 6e-8).  The proximal eps has to stay
    // well above it relative to the curvature it regularises (the K blocks carry 1 / eps on the null direction of B'B, and
    // Y = K G' cancels it against G: eps / (2 w lambda) = 1e-3 costs three of the seven digits there), pivots below 1e-5 of
    // their diagonal are noise, and a multiplier or bound violation below 1e-4 of its scale cannot be told from rounding.
    D.eps_primal = 1e-6; D.tau_pivot = 1e-5; D.tol_bound = 1e-4; D.tol_accept = 2e-3;
    // A search that has not ended after 48 working-set changes (mean 5.5, p99 18, the longest of 32 768 fp64 searches 41) is cycling
    // on rounding noise with a period the 8-set history does not see (3 QPs in 10^4): stop it there -- one such QP running to the
    // 200 changes of the fp64 default holds the whole launch for 0.4 ms (measured: 1.15 ms per launch instead of 0.8)
    if (D.max_as_iters > 48) D.max_as_iters = 48;
    if (const char *e = QPC_EXP_ENV("QPC_F32_MAXIT")) D.max_as_iters = atoi(e);
    if (const char *e = QPC_EXP_ENV("QPC_F32_EPS")) D.eps_primal = atof(e); // (experiments: the tolerance set was found with these)
    if (const char *e = QPC_EXP_ENV("QPC_F32_TAU")) D.tau_pivot = atof(e);
    if (const char *e = QPC_EXP_ENV("QPC_F32_TOLB")) D.tol_bound = atof(e);
    if (const char *e = QPC_EXP_ENV("QPC_F32_TOLA")) D.tol_accept = atof(e);
  }
  DS = D;
  if (D.presolve) {
    auto lead2 = [](int n) { int s_ = (n > 0 ? n : 1); s_ += s_ & 1; if (s_ % 8 == 0) s_ += 2; return s_; };
    DS.nf = D.pre_nf; DS.m = D.pre_m - D.pre_fold; DS.m1 = DS.m - D.nw;
    if (D.pre_fold) { DS.ne = 0; DS.wsize = 0; }
    DS.nfs = lead2(DS.nf); DS.ms = lead2(DS.m); DS.m1s = odd(DS.m1 > 0 ? DS.m1 : 1);
    DS.lds[L_M] = s_lds[0]; DS.lds[L_D] = s_lds[1]; DS.lds[L_HV] = s_lds[2]; DS.lds[L_SQH] = s_lds[3]; DS.lds[L_GV] = s_lds[4];
    DS.lds[L_XV] = s_lds[5]; DS.lds[L_HOT] = s_lds[6];
    DS.it[IT_HOT] = D.it[IT_HOT_S];
  }
  P.dims_s = DS;
  P.itab.swap(B.it);
  P.dtab.swap(B.dt);

  // ---- public layout
  qpc_layout &Lo = c->layout;
  std::memset(&Lo, 0, sizeof(Lo));
  Lo.nq = m->nq; Lo.nv = nv; Lo.n_tasks = (int)c->tasks.size(); Lo.n_contacts = D.nc; Lo.n_basis = N;
  Lo.n_desired = c->n_desired; Lo.n_task_weights = c->n_tw; Lo.n_surface = c->n_surf;
  int nvar = nv, neq = 0;
  for (auto &t : c->tasks) { neq += t.dim; if (t.wkind != QPC_WEIGHT_HARD) nvar += t.dim; }
  nvar += D.nc * (N + 9); neq += D.nc * 9 + D.nw;
  Lo.ref_n_var = nvar; Lo.ref_n_eq = neq; Lo.ref_n_ineq = D.nc * 2 * N;
  Lo.n_free = D.nf; Lo.n_pinned = D.npin; Lo.n_slack = ne; Lo.n_rho = nrho; Lo.n_eq_rows = mm;
  Lo.lds_bytes_per_qp = D.lds_doubles * (c->precision == QPC_PRECISION_F32 ? 4 : 8);
  Lo.n_desired_inputs = D.n_des_inputs;
  if (nb > 64 || nv > 64 || D.ntask > 63 || D.maxdepth > 32) {
    qpc_set_error("limits of the one-wave-per-QP kernels exceeded (bodies <= 64, velocities <= 64, tree depth <= 32)");
    return QPC_ERR_UNSUPPORTED;
  }
  if (D.m > 64) {
    qpc_set_error("more than 64 equality rows in the reduced QP are not supported by the lane=row factorization");
    return QPC_ERR_UNSUPPORTED;
  }
  if (c->warm_start && D.nrho > 64) {
    qpc_set_error("warm start carries the working set as two 64-bit masks: at most 64 rho's");
    return QPC_ERR_UNSUPPORTED;
  }
  if (Lo.lds_bytes_per_qp > 160 * 1024) {
    qpc_set_error("problem does not fit the 160 KiB LDS of one CU");
    return QPC_ERR_UNSUPPORTED;
  }
  return QPC_OK;
}

void qpc_l1_build_tables(const qpc_controller *c, std::vector<int> &it, std::vector<double> &dt, int &n, int &m) {
  const qpc_model *md = c->model;
  const int nv = md->nv, ntask = (int)c->tasks.size(), nc = (int)c->contacts.size(), N = c->N;
  int ne = 0, wsize = 0;
  it.assign(7, 0);
  dt.assign(c->reg.begin(), c->reg.end());
  for (auto &t : c->tasks) {
    int s0 = -1, wo = 0;
    if (t.wkind != QPC_WEIGHT_HARD) { s0 = ne; ne += t.dim; }
    if (t.wkind == QPC_WEIGHT_SCALAR || t.wkind == QPC_WEIGHT_MATRIX) { wo = wsize; wsize += (int)t.W.size(); dt.insert(dt.end(), t.W.begin(), t.W.end()); }
    const int row[6] = {t.dim, t.wkind, t.des_off, s0, wo, t.tw_off};
    it.insert(it.end(), row, row + 6);
  }
  for (auto &k : c->contacts) it.push_back((k.per_qp & 1) ? k.surf_off : -1); // L1 reads mu only (the pose comes evaluated)
  it.push_back(wsize);
  for (auto &k : c->contacts) dt.push_back(k.mu);
  dt.push_back(0.0);
  it[0] = nv; it[1] = ntask; it[2] = nc; it[3] = N; it[4] = c->floating >= 0 ? 1 : 0; it[5] = ne; it[6] = c->n_desired;
  n = nv + ne + nc * N;
  m = c->n_desired + (c->floating >= 0 ? 6 : 0) + nc * N;
}
