// qpc_compile.h -- host-side model / controller objects and the program compiler (pure C++,
// no HIP): mirrors Mechanism + MomentumBasedController construction of the reference
// (momentum.jl:15-33,99-160) and freezes it into a QpcDims + flat tables (qpc_program.h).
#pragma once
#include "qpc_env.h"
#include <string>
#include <vector>

#include "../../include/qpc.h"
#include "qpc_program.h"

struct qpc_model {
  int nb = 0, nq = 0, nv = 0;
  std::vector<int> parent, jtype, qoff, voff, nqj, nvj;
  std::vector<double> axis, Rf, pf, mass, com, icom;
  double gravity[3] = {0, 0, 0};
  double mtot = 0;
};

struct QpcTask {
  int kind, base, target, frame, joint, wkind, dim, des_off, tw_off;
  double point[3];
  std::vector<double> W;
  int law = 0; // QPC_LAW_*
  double kp = 0, kd = 0;
  std::vector<double> ref;
};
struct QpcContact {
  int body, per_qp, surf_off;
  double pos[3], normal[3], mu;
};

struct QpcProgram {
  QpcDims dims;
  QpcDims dims_s; // dims.presolve: the reduced problem the solver runs on (same tables and slab; other nf, m1, m, leading
                  // dimensions, and its own places for M, d, the Hessian vectors, the iterate's v̇ part and the hot lookups)
  std::vector<int> itab;
  std::vector<double> dtab;
};

struct QpcRuntime; // device states, worker threads: owned by the run half (qpc_runtime.hip; tests/emul/emul.cpp)

struct qpc_controller {
  const qpc_model *model = nullptr;
  int N = 4, floating = -1;
  std::vector<QpcTask> tasks;
  std::vector<QpcContact> contacts;
  std::vector<double> reg;
  std::vector<int> reg_tw; // per velocity index: entry of task_weights that holds its per-QP regularisation weight, or -1
  int n_desired = 0, n_tw = 0, n_surf = 0;
  bool finalized = false;
  qpc_settings settings;
  QpcProgram prog;
  qpc_layout layout;
  QpcRuntime *rt = nullptr;
  int qps_per_wave = 1;                     // qpc_set_qps_per_wavefront: 2 = two QPs per wavefront (lanes 0-31 / 32-63), hiprtc kernel only
  int precision = 0;                        // qpc_set_precision: QPC_PRECISION_F64 / _F32 (arithmetic of the tick kernel)
  int warm_start = 0;                       // qpc_enable_warm_start: also build the warm-started kernel variant
  int pipe_slots = 0;                       // qpc_set_host_pipeline (0 / -1: defaults)
  long long pipe_chunk = 0, pipe_inline = -1;
};

// entries of a task's slice of the per-QP `desired` record: its dimension, or the trajectory record of a device-side SE(3) law
inline int qpc_task_desired_len(const QpcTask &t) {
  if (t.law != QPC_LAW_SE3_PD) return t.dim;
  const int np = t.ref.size() >= 3 ? (int)t.ref[2] : 1;
  return np >= 2 ? QPC_SE3_PW_RECORD(np) : QPC_SE3_RECORD;
}
void qpc_set_error(const std::string &msg);
int qpc_task_dim(const qpc_model *m, int kind, int joint);
// freeze the controller into prog + layout; returns QPC_OK or an error code
int qpc_compile_program(qpc_controller *c);
// static tables of the L1 path (layout documented in qpc_l1_dense.h); n / m = size of its dense QP
void qpc_l1_build_tables(const qpc_controller *c, std::vector<int> &it, std::vector<double> &dt, int &n, int &m);
