// qpc_capi_build.cpp -- the construction half of the C ABI (include/qpc.h): model and controller
// handles, addtask!/regularize!/addcontact! mirrors.  Pure host C++, no HIP; the run half
// (qpc_finalize upload, qpc_control_batch*) lives in qpc_runtime.hip.
#include <cmath>
#include <cstring>
#include <new>

#include "qpc_compile.h"

extern "C" {

int32_t qpc_model_create(const qpc_model_desc *d, qpc_model **out) {
  if (!d || !out || d->nbodies <= 0) { qpc_set_error("qpc_model_create: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  qpc_model *m = new (std::nothrow) qpc_model();
  if (!m) return QPC_ERR_OUT_OF_MEMORY;
  m->nb = d->nbodies;
  int nq = 0, nv = 0;
  for (int i = 0; i < m->nb; i++) {
    if (d->parent[i] >= i || d->parent[i] < -1) {
      delete m; qpc_set_error("qpc_model_create: bodies must be topologically ordered (parent[i] < i)");
      return QPC_ERR_INVALID_ARGUMENT;
    }
    int nqj, nvj;
    switch (d->joint_type[i]) {
      case QPC_JOINT_REVOLUTE: case QPC_JOINT_PRISMATIC: nqj = 1; nvj = 1; break;
      case QPC_JOINT_QUAT_FLOATING: nqj = 7; nvj = 6; break;
      case QPC_JOINT_FIXED: nqj = 0; nvj = 0; break;
      default: delete m; qpc_set_error("qpc_model_create: unknown joint type"); return QPC_ERR_INVALID_ARGUMENT;
    }
    m->parent.push_back(d->parent[i]); m->jtype.push_back(d->joint_type[i]);
    m->qoff.push_back(nq); m->voff.push_back(nv); m->nqj.push_back(nqj); m->nvj.push_back(nvj);
    nq += nqj; nv += nvj;
    m->mass.push_back(d->mass[i]); m->mtot += d->mass[i];
  }
  m->nq = nq; m->nv = nv;
  m->axis.assign(d->joint_axis, d->joint_axis + 3 * m->nb);
  m->Rf.assign(d->joint_R, d->joint_R + 9 * m->nb);
  m->pf.assign(d->joint_p, d->joint_p + 3 * m->nb);
  m->com.assign(d->com, d->com + 3 * m->nb);
  m->icom.assign(d->inertia, d->inertia + 6 * m->nb);
  std::memcpy(m->gravity, d->gravity, sizeof(m->gravity));
  *out = m;
  return QPC_OK;
}
void qpc_model_destroy(qpc_model *m) { delete m; }
int32_t qpc_model_nq(const qpc_model *m) { return m ? m->nq : -1; }
int32_t qpc_model_nv(const qpc_model *m) { return m ? m->nv : -1; }
int32_t qpc_model_joint_ranges(const qpc_model *m, int32_t body, int32_t *q0, int32_t *nq, int32_t *v0, int32_t *nv) {
  if (!m || body < 0 || body >= m->nb) return QPC_ERR_INVALID_ARGUMENT;
  if (q0) *q0 = m->qoff[body];
  if (nq) *nq = m->nqj[body];
  if (v0) *v0 = m->voff[body];
  if (nv) *nv = m->nvj[body];
  return QPC_OK;
}

int32_t qpc_default_settings(qpc_settings *s) {
  if (!s) return QPC_ERR_INVALID_ARGUMENT;
  s->max_active_set_iters = 200; s->max_refine_iters = 6;
  s->eps_primal_reg = 1e-9; s->eps_dual_reg = 1e-11; s->tol_residual = 1e-10; s->tol_bound = 1e-10;
  return QPC_OK;
}

int32_t qpc_controller_create(const qpc_model *model, int32_t n_basis, int32_t floating_body, qpc_controller **out) {
  if (!model || !out || n_basis < 1 || n_basis > 8 || floating_body < -1 || floating_body >= model->nb) {
    qpc_set_error("qpc_controller_create: bad arguments (n_basis in 1..8)");
    return QPC_ERR_INVALID_ARGUMENT;
  }
  qpc_controller *c = new (std::nothrow) qpc_controller();
  if (!c) return QPC_ERR_OUT_OF_MEMORY;
  c->model = model; c->N = n_basis; c->floating = floating_body;
  c->reg.assign(model->nv, 0.0);
  c->reg_tw.assign(model->nv, -1);
  qpc_default_settings(&c->settings);
  std::memset(&c->layout, 0, sizeof(c->layout));
  *out = c;
  return QPC_OK;
}

int32_t qpc_set_settings(qpc_controller *c, const qpc_settings *s) {
  if (!c || !s) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("settings must be set before qpc_finalize"); return QPC_ERR_ALREADY_FINALIZED; }
  c->settings = *s;
  return QPC_OK;
}

int32_t qpc_add_task(qpc_controller *c, int32_t kind, int32_t base_body, int32_t target_body, int32_t frame_body,
                     const double *point, int32_t joint_body, int32_t weight_kind, const double *weight,
                     int32_t *out_task_id) {
  if (!c) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("tasks must be added before the first tick"); return QPC_ERR_ALREADY_FINALIZED; }
  const qpc_model *m = c->model;
  QpcTask t;
  std::memset(t.point, 0, sizeof(t.point));
  t.kind = kind; t.base = base_body; t.target = target_body; t.joint = joint_body; t.wkind = weight_kind;
  t.dim = qpc_task_dim(m, kind, joint_body);
  if (t.dim <= 0) { qpc_set_error("qpc_add_task: bad kind / joint"); return QPC_ERR_INVALID_ARGUMENT; }
  bool path_task = kind <= QPC_TASK_POINT_ACCELERATION;
  if (path_task && (base_body < -1 || base_body >= m->nb || target_body < -1 || target_body >= m->nb)) {
    qpc_set_error("qpc_add_task: body index out of range"); return QPC_ERR_INVALID_ARGUMENT;
  }
  t.frame = frame_body == QPC_FRAME_DEFAULT ? target_body : frame_body;
  if (kind == QPC_TASK_POINT_ACCELERATION) {
    t.frame = base_body; // src/tasks.jl:139-140,154
    if (target_body < 0) { qpc_set_error("PointAccelerationTask needs a moving target body"); return QPC_ERR_INVALID_ARGUMENT; }
  }
  if (path_task && (t.frame < -1 || t.frame >= m->nb)) { qpc_set_error("qpc_add_task: frame out of range"); return QPC_ERR_INVALID_ARGUMENT; }
  if (point) std::memcpy(t.point, point, sizeof(t.point));
  t.des_off = c->n_desired; t.tw_off = -1;
  switch (weight_kind) {
    case QPC_WEIGHT_HARD: break;
    case QPC_WEIGHT_SCALAR:
      if (!weight) { qpc_set_error("qpc_add_task: weight missing"); return QPC_ERR_INVALID_ARGUMENT; }
      t.W.assign(weight, weight + 1); break;
    case QPC_WEIGHT_MATRIX:
      if (!weight) { qpc_set_error("qpc_add_task: weight missing"); return QPC_ERR_INVALID_ARGUMENT; }
      t.W.assign(weight, weight + t.dim * t.dim); break;
    case QPC_WEIGHT_SCALAR_PER_QP: t.tw_off = c->n_tw; c->n_tw += 1; break;
    case QPC_WEIGHT_MATRIX_PER_QP: t.tw_off = c->n_tw; c->n_tw += t.dim * t.dim; break;
    default: qpc_set_error("qpc_add_task: bad weight kind"); return QPC_ERR_INVALID_ARGUMENT;
  }
  c->n_desired += t.dim;
  c->tasks.push_back(t);
  if (out_task_id) *out_task_id = (int32_t)c->tasks.size() - 1;
  return QPC_OK;
}

int32_t qpc_task_set_law(qpc_controller *c, int32_t task_id, int32_t law, double kp, double kd, const double *ref,
                         int32_t n_ref) {
  if (!c || task_id < 0 || task_id >= (int)c->tasks.size() || n_ref < 0 || (n_ref && !ref)) {
    qpc_set_error("qpc_task_set_law: bad task id / reference"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (c->finalized) { qpc_set_error("desired laws must be set before the first tick"); return QPC_ERR_ALREADY_FINALIZED; }
  QpcTask &t = c->tasks[task_id];
  const qpc_model *m = c->model;
  int need = 0;
  switch (law) {
    case QPC_LAW_INPUT: break;
    case QPC_LAW_CONSTANT: need = t.dim; break;
    case QPC_LAW_JOINT_PD:
      if (t.kind != QPC_TASK_JOINT_ACCELERATION || m->nvj[t.joint] != 1 || m->nqj[t.joint] != 1) {
        qpc_set_error("QPC_LAW_JOINT_PD needs a JointAccelerationTask of a 1-DoF joint"); return QPC_ERR_INVALID_ARGUMENT;
      }
      need = 1; break;
    case QPC_LAW_COM_PD:
      if (t.kind != QPC_TASK_LINEAR_MOMENTUM_RATE) {
        qpc_set_error("QPC_LAW_COM_PD needs a LinearMomentumRateTask"); return QPC_ERR_INVALID_ARGUMENT;
      }
      need = 3; break;
    case QPC_LAW_ORIENTATION_PD:
      if (t.kind != QPC_TASK_ANGULAR_ACCELERATION || t.base != QPC_WORLD || t.target < 0 || t.frame != t.target) {
        qpc_set_error("QPC_LAW_ORIENTATION_PD needs an AngularAccelerationTask world -> body expressed in the body frame");
        return QPC_ERR_INVALID_ARGUMENT;
      }
      break;
    case QPC_LAW_SE3_PD:
      if (t.kind != QPC_TASK_SPATIAL_ACCELERATION || t.target < 0 || t.frame != t.target) {
        qpc_set_error("QPC_LAW_SE3_PD needs a SpatialAccelerationTask expressed in its target body's frame");
        return QPC_ERR_INVALID_ARGUMENT;
      }
      need = n_ref == 3 ? 3 : 2; // linear gains (kp, kd) [+ number of pieces of a Piecewise trajectory]; the angular ones are the kp / kd arguments
      if (n_ref == 3 && !(ref[2] >= 2.0 && ref[2] <= (double)QPC_SE3_MAX_PIECES && ref[2] == (double)(int)ref[2])) {
        qpc_set_error("QPC_LAW_SE3_PD: the number of pieces (ref[2]) must be an integer in 2 .. QPC_SE3_MAX_PIECES"); return QPC_ERR_INVALID_ARGUMENT;
      }
      break;
    default: qpc_set_error("qpc_task_set_law: unknown law"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (n_ref != need) { qpc_set_error("qpc_task_set_law: wrong reference length for this law"); return QPC_ERR_INVALID_ARGUMENT; }
  t.law = law; t.kp = kp; t.kd = kd;
  t.ref.assign(ref, ref + n_ref);
  return QPC_OK;
}

int32_t qpc_task_desired_range(const qpc_controller *c, int32_t task_id, int32_t *offset, int32_t *length) {
  if (!c || task_id < 0 || task_id >= (int)c->tasks.size()) return QPC_ERR_INVALID_ARGUMENT;
  if (!c->finalized) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  const QpcTask &t = c->tasks[task_id];
  if (offset) *offset = t.des_off;
  if (length) *length = qpc_task_desired_len(t);
  return QPC_OK;
}

int32_t qpc_set_precision(qpc_controller *c, int32_t precision) {
  if (!c || (precision != QPC_PRECISION_F64 && precision != QPC_PRECISION_F32)) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("qpc_set_precision must be called before qpc_finalize"); return QPC_ERR_ALREADY_FINALIZED; }
  c->precision = precision;
  return QPC_OK;
}

int32_t qpc_set_qps_per_wavefront(qpc_controller *c, int32_t n) {
  if (!c || (n != 1 && n != 2)) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("qpc_set_qps_per_wavefront must be called before qpc_finalize"); return QPC_ERR_ALREADY_FINALIZED; }
  c->qps_per_wave = n;
  return QPC_OK;
}

int32_t qpc_regularize(qpc_controller *c, int32_t joint, double w) {
  if (!c || joint < 0 || joint >= c->model->nb) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("regularize! must be called before the first tick"); return QPC_ERR_ALREADY_FINALIZED; }
  for (int k = 0; k < c->model->nvj[joint]; k++) c->reg[c->model->voff[joint] + k] += w;
  return QPC_OK;
}

int32_t qpc_regularize_per_qp(qpc_controller *c, int32_t joint, int32_t *out_weight_index) {
  if (!c || joint < 0 || joint >= c->model->nb || c->model->nvj[joint] < 1) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("regularize! must be called before the first tick"); return QPC_ERR_ALREADY_FINALIZED; }
  if (c->reg_tw[c->model->voff[joint]] >= 0) { qpc_set_error("qpc_regularize_per_qp: this joint already has a per-QP weight"); return QPC_ERR_INVALID_ARGUMENT; }
  for (int k = 0; k < c->model->nvj[joint]; k++) c->reg_tw[c->model->voff[joint] + k] = c->n_tw;
  if (out_weight_index) *out_weight_index = c->n_tw;
  c->n_tw += 1;
  return QPC_OK;
}

int32_t qpc_add_contact(qpc_controller *c, int32_t body, const double position[3], const double normal[3], double mu,
                        int32_t per_qp_surface, int32_t *out_contact_id) {
  if (!c || !position || !normal || body < -1 || body >= c->model->nb) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("contacts must be added before the first tick"); return QPC_ERR_ALREADY_FINALIZED; }
  QpcContact k;
  if (per_qp_surface < 0 || per_qp_surface > 3) { qpc_set_error("qpc_add_contact: per_qp_surface is a bit set of QPC_CONTACT_*_PER_QP"); return QPC_ERR_INVALID_ARGUMENT; }
  k.body = body; k.per_qp = per_qp_surface; k.surf_off = -1; k.mu = mu;
  std::memcpy(k.pos, position, sizeof(k.pos));
  std::memcpy(k.normal, normal, sizeof(k.normal));
  if (k.per_qp) { k.surf_off = c->n_surf; c->n_surf += ((k.per_qp & 1) ? 4 : 0) + ((k.per_qp & 2) ? 3 : 0); }
  c->contacts.push_back(k);
  if (out_contact_id) *out_contact_id = (int32_t)c->contacts.size() - 1;
  return QPC_OK;
}

int32_t qpc_enable_warm_start(qpc_controller *c, int32_t enable) {
  if (!c) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { qpc_set_error("qpc_enable_warm_start must be called before qpc_finalize"); return QPC_ERR_ALREADY_FINALIZED; }
  c->warm_start = enable ? 1 : 0;
  return QPC_OK;
}

int32_t qpc_get_layout(const qpc_controller *c, qpc_layout *out) {
  if (!c || !out) return QPC_ERR_INVALID_ARGUMENT;
  if (!c->finalized) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  *out = c->layout;
  return QPC_OK;
}

int32_t qpc_contact_bodies(const qpc_controller *c, int32_t *n_bodies, int32_t *bodies, int32_t *body_of_contact) {
  if (!c || !n_bodies) return QPC_ERR_INVALID_ARGUMENT;
  std::vector<int> list;
  for (size_t k = 0; k < c->contacts.size(); k++) {
    size_t i = 0;
    while (i < list.size() && list[i] != c->contacts[k].body) i++;
    if (i == list.size()) list.push_back(c->contacts[k].body);
    if (body_of_contact) body_of_contact[k] = (int32_t)i;
  }
  *n_bodies = (int32_t)list.size();
  if (bodies) for (size_t i = 0; i < list.size(); i++) bodies[i] = list[i];
  return QPC_OK;
}
int32_t qpc_contact_wrench_per_body(const qpc_controller *c, int64_t B, const double *cw, double *bw) {
  if (!c || B < 0 || (B > 0 && (!cw || !bw))) return QPC_ERR_INVALID_ARGUMENT;
  const int nc = (int)c->contacts.size();
  int32_t nb = 0;
  std::vector<int32_t> of(nc > 0 ? nc : 1);
  qpc_contact_bodies(c, &nb, nullptr, of.data());
  for (int64_t b = 0; b < B; b++) {
    double *o = bw + b * 6 * nb;
    for (int i = 0; i < 6 * nb; i++) o[i] = 0.0;
    for (int k = 0; k < nc; k++) for (int e = 0; e < 6; e++) o[6 * of[k] + e] += cw[(b * nc + k) * 6 + e]; // (contact order: a fixed summation order)
  }
  return QPC_OK;
}

const char *qpc_status_string(int32_t s) {
  switch (s) {
    case QPC_STATUS_OPTIMAL: return "OPTIMAL";
    case QPC_STATUS_ALMOST_OPTIMAL: return "ALMOST_OPTIMAL";
    case QPC_STATUS_INFEASIBLE: return "INFEASIBLE";
    case QPC_STATUS_DUAL_INFEASIBLE: return "DUAL_INFEASIBLE";
    case QPC_STATUS_ITERATION_LIMIT: return "ITERATION_LIMIT";
    case QPC_STATUS_NUMERICAL_ERROR: return "NUMERICAL_ERROR";
  }
  return "UNKNOWN";
}
int32_t qpc_status_triple(int32_t s, const char **termination, const char **primal, const char **dual) {
  static const char *const T[6][3] = {
      {"OPTIMAL", "FEASIBLE_POINT", "FEASIBLE_POINT"},
      {"ALMOST_OPTIMAL", "UNKNOWN_RESULT_STATUS", "UNKNOWN_RESULT_STATUS"},
      {"INFEASIBLE", "NO_SOLUTION", "INFEASIBILITY_CERTIFICATE"},
      {"DUAL_INFEASIBLE", "INFEASIBILITY_CERTIFICATE", "NO_SOLUTION"},
      {"ITERATION_LIMIT", "UNKNOWN_RESULT_STATUS", "UNKNOWN_RESULT_STATUS"},
      {"NUMERICAL_ERROR", "UNKNOWN_RESULT_STATUS", "UNKNOWN_RESULT_STATUS"}};
  if (s < 0 || s > 5) return QPC_ERR_INVALID_ARGUMENT;
  if (termination) *termination = T[s][0];
  if (primal) *primal = T[s][1];
  if (dual) *dual = T[s][2];
  return QPC_OK;
}
const char *qpc_version(void) { return "qpcontrol_hip 0.1.0 (gfx950)"; }

} // extern "C"

// diagnostic: number of K tables of the compiled program (0: Gauss-Jordan form; tests/test_k_table.py)
extern "C" int32_t qpc_debug_k_tables(qpc_controller *c) {
  if (!c || !c->finalized) return -1;
  return c->prog.dims.k_tab ? c->prog.dims.k_classes : 0;
}

// diagnostic: number of row blocks the null-space presolve eliminates (0: the program has none; tests/test_presolve.py)
extern "C" int32_t qpc_debug_presolve_blocks(qpc_controller *c) {
  if (!c || !c->finalized) return -1;
  return c->prog.dims.presolve;
}
