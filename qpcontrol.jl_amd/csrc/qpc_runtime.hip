// qpc_runtime.hip -- the run half of the C ABI (include/qpc.h): qpc_finalize (program upload),
// qpc_control_batch / qpc_control_batch_device (kernel launch), timing hooks.  gfx950 only.
//
// Execution model: one wavefront (64 lanes) per QP, one 64-thread workgroup per wavefront so that
// the phase barrier is a single-wave s_barrier; each workgroup owns one LDS slab
// (QpcDims::lds_doubles) and grid-strides over the batch.  Inputs/outputs are per-QP contiguous
// records, loaded/stored by consecutive lanes (coalesced); H, A, b and every factor live only in
// LDS and never touch HBM.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "qpc_compile.h"
#include "qpc_jit.h"

#include "qpc_device.h"
#ifdef QPC_STAMPS
#define QPC_PROF_T0() do { *PT = clock64(); } while (0)
#define QPC_PROF_ACC(i) do { unsigned long long n__ = clock64(); PA[i] += n__ - *PT; *PT = n__; } while (0)
#endif
#include "qpc_tick.h"
#include "qpc_sim.h"
#include "qpc_qp_dense.h"
#include "qpc_l1_dense.h"

// L0: one lane per dense QP (plumbing path, see qpc_qp_dense.h)
extern "C" __global__ void __launch_bounds__(64)
qpc_qp_dense_kernel(int n, int m, long long B, const double *P, const double *q, const double *A, const double *l,
                    const double *u, qpc_settings st, double *ws, long long ws_stride, double *x, double *y,
                    int32_t *status, int32_t *iters) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  qpc::qp_solve_dense(n, m, P + b * n * n, q + b * n, A + b * m * n, l + b * m, u + b * m, st, ws + b * ws_stride,
                      x + b * n, y + b * m, status + b, iters ? iters + b : nullptr);
}

// L1: one lane per QP, dense assembly from host-evaluated blocks + the L0 solver (qpc_l1_dense.h)
struct QpcL1Args {
  long long B;
  const int *it;
  const double *dt;
  const double *J, *bias, *desired, *Am, *mrhs, *cR, *cp, *cw, *cmax, *tw, *surf;
  double *ws, *vdot, *cwrench;
  int32_t *status, *iters;
  long long ws_stride;
  int nv, nd, nc, ntw, nsurf;
};
extern "C" __global__ void __launch_bounds__(64) qpc_l1_kernel(const QpcL1Args a, qpc_settings st) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  qpc::L1Prog pg{a.it, a.dt};
  qpc::l1_assemble_solve(pg, a.J + b * a.nd * a.nv, a.bias + b * a.nd, a.desired + b * a.nd,
                         a.Am ? a.Am + b * 6 * a.nv : nullptr, a.mrhs ? a.mrhs + b * 6 : nullptr, a.cR + b * 9 * a.nc,
                         a.cp + b * 3 * a.nc, a.cw + b * a.nc, a.cmax + b * a.nc, a.tw + b * a.ntw, a.surf + b * a.nsurf, st,
                         a.ws + b * a.ws_stride, a.vdot + b * a.nv, a.cwrench + b * 6 * a.nc, a.status + b, a.iters + b);
}

struct QpcBatchArgs {
  long long B;
  const double *q, *v, *desired, *cw, *cmax, *tw, *surf;
  double *tau, *vdot, *cwrench;
  int32_t *status, *iters;
  unsigned long long *prof;
  unsigned long long *ticket; // dynamic work distribution: a counter that is never reset; a launch draws exactly
                              // B + gridDim.x tickets (every workgroup draws one past the end), so the host knows
                              // where the next launch starts (ticket_base) without a memset in between
  unsigned long long ticket_base;
  double *scratch;            // QPC_SCRATCH_DOUBLES per resident workgroup (lane-private spill slots)
};

extern "C" __global__ void __launch_bounds__(64, 1)
qpc_tick_kernel(const QpcDims D, const int *__restrict__ IT, const double *__restrict__ DT, const QpcBatchArgs A) {
  extern __shared__ __attribute__((aligned(16))) double qpc_lds[];
  qpc::qpc_tick_prologue<double *>(D, IT, qpc_lds);
  // active-set iteration counts differ per QP (p99 ~ 3x mean): QPs are handed out through one
  // device-scope ticket counter instead of a static grid-stride, so no workgroup idles at the tail
  while (true) {
    unsigned long long t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(A.ticket, 1ull);
    const long long b = (long long)(__shfl(t, 0, 64) - A.ticket_base);
    if (b >= A.B) break;
    qpc::TickIO io;
    io.q = A.q + b * D.nq; io.v = A.v + b * D.nv;
    io.desired = A.desired + b * D.n_desired;
    io.cw = A.cw + b * D.nc; io.cmax = A.cmax + b * D.nc;
    io.tw = A.tw + b * D.n_tw; io.surf = A.surf + b * D.n_surf;
    io.tau = A.tau + b * D.nv; io.vdot = A.vdot + b * D.nv;
    io.cwrench = A.cwrench ? A.cwrench + b * 6 * D.nc : nullptr;
    io.status = A.status + b; io.iters = A.iters ? A.iters + b : nullptr;
    io.prof = A.prof ? A.prof + b * PS_COUNT : nullptr;
    io.scratch = A.scratch + (size_t)blockIdx.x * QPC_SCRATCH_DOUBLES;
    qpc::qpc_tick<double *>(D, IT, DT, qpc_lds, io);
    __syncthreads(); // the slab is reused by the next QP of this workgroup
  }
}

// f-2: one simulation step per robot (qpc_sim.h), one wavefront per robot, static grid-stride (the
// step has no data-dependent iteration count, so there is no tail to balance)
struct QpcSimArgs {
  long long B;
  double *q, *v;
  const double *tau;
  const int32_t *status;
  int32_t *bad;
  qpc_sim_params sp;
};
extern "C" __global__ void __launch_bounds__(64)
qpc_sim_kernel(const QpcDims D, const int *__restrict__ IT, const double *__restrict__ DT, const QpcSimArgs A) {
  extern __shared__ __attribute__((aligned(16))) double qpc_lds[];
  for (long long b = blockIdx.x; b < A.B; b += gridDim.x) {
    qpc::SimIO io;
    io.q = A.q + b * D.nq; io.v = A.v + b * D.nv; io.tau = A.tau + b * D.nv;
    io.status = A.status ? A.status + b : nullptr;
    io.q_out = A.q + b * D.nq; io.v_out = A.v + b * D.nv;
    io.bad = A.bad ? A.bad + b : nullptr;
    qpc::qpc_sim_step<double *>(D, IT, DT, qpc_lds, io, A.sp);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
struct QpcDeviceState {
  int device = -1;
  int *d_it = nullptr;
  double *d_dt = nullptr;
  int n_cu = 0, blocks_per_cu = 1;
  hipStream_t stream = nullptr; // internal stream of the host-pointer entry point
  // staging buffers of the host-pointer entry point
  void *stage = nullptr;
  size_t stage_bytes = 0;
  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<hipEvent_t> pool;
  unsigned long long *prof = nullptr; // PS_COUNT cycle counters per QP (diagnostic builds only)
  // JIT-specialised kernel (qpc_jit.cpp); null -> the generic ahead-of-time kernel is launched
  hipModule_t jit_module = nullptr;
  hipFunction_t jit_func = nullptr;
  hipFunction_t jit_sim_func = nullptr; // qpc_sim_kernel_jit of the same module
  unsigned long long *d_ticket = nullptr;
  unsigned long long ticket_next = 0; // value of *d_ticket once every launch enqueued so far has drained
  double *d_scratch = nullptr;
  int sim_blocks_per_cu = 1;
};

static int hip_fail(hipError_t e, const char *what) {
  qpc_set_error(std::string(what) + ": " + hipGetErrorString(e));
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? QPC_ERR_NO_DEVICE : QPC_ERR_HIP;
}
#define HIP_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_fail(e__, #call); } while (0)

static int launch(qpc_controller *c, const QpcBatchArgs &a, hipStream_t stream) {
  QpcDeviceState *S = c->dev;
  const QpcDims &D = c->prog.dims;
  long long grid = (long long)S->n_cu * S->blocks_per_cu;
  if (grid > a.B) grid = a.B;
  if (grid < 1) return QPC_OK;
  size_t lds = (size_t)D.lds_doubles * sizeof(double);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (S->timing) {
    auto get = [&](hipEvent_t &e) -> hipError_t {
      if (!S->pool.empty()) { e = S->pool.back(); S->pool.pop_back(); return hipSuccess; }
      return hipEventCreate(&e);
    };
    HIP_TRY(get(e0)); HIP_TRY(get(e1));
    HIP_TRY(hipEventRecord(e0, stream));
  }
  QpcBatchArgs aa = a;
  aa.ticket = S->d_ticket;
  aa.ticket_base = S->ticket_next;
  S->ticket_next += (unsigned long long)a.B + (unsigned long long)grid;
  aa.scratch = S->d_scratch;
  if (S->jit_func) {
    const int *it = S->d_it;
    const double *dt = S->d_dt;
    void *args[] = {(void *)&it, (void *)&dt, (void *)&aa};
    HIP_TRY(hipModuleLaunchKernel(S->jit_func, (unsigned)grid, 1, 1, 64, 1, 1, 0, stream, args, nullptr));
  } else {
    hipLaunchKernelGGL(qpc_tick_kernel, dim3((unsigned)grid), dim3(64), lds, stream, D, S->d_it, S->d_dt, aa);
    HIP_TRY(hipGetLastError());
  }
  if (S->timing) {
    HIP_TRY(hipEventRecord(e1, stream));
    S->events.emplace_back(e0, e1);
  }
  return QPC_OK;
}

static int launch_sim(qpc_controller *c, const QpcSimArgs &a, hipStream_t stream) {
  QpcDeviceState *S = c->dev;
  const QpcDims &D = c->prog.dims;
  long long grid = (long long)S->n_cu * S->sim_blocks_per_cu;
  if (grid > a.B) grid = a.B;
  if (grid < 1) return QPC_OK;
  if (S->jit_sim_func) {
    const int *it = S->d_it;
    const double *dt = S->d_dt;
    QpcSimArgs aa = a;
    void *args[] = {(void *)&it, (void *)&dt, (void *)&aa};
    HIP_TRY(hipModuleLaunchKernel(S->jit_sim_func, (unsigned)grid, 1, 1, 64, 1, 1, 0, stream, args, nullptr));
    return QPC_OK;
  }
  const size_t lds = (size_t)qpc_sim_layout(D).total * sizeof(double);
  hipLaunchKernelGGL(qpc_sim_kernel, dim3((unsigned)grid), dim3(64), lds, stream, D, S->d_it, S->d_dt, a);
  HIP_TRY(hipGetLastError());
  return QPC_OK;
}

extern "C" {

int32_t qpc_finalize(qpc_controller *c, int32_t device_id, qpc_layout *out) {
  if (!c) return QPC_ERR_INVALID_ARGUMENT;
  if (c->finalized) { if (out) *out = c->layout; return QPC_OK; }
  int rc = qpc_compile_program(c);
  if (rc != QPC_OK) return rc;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    qpc_set_error("qpc_finalize: no HIP device available (this library has no CPU fallback)");
    return QPC_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) { qpc_set_error("qpc_finalize: device_id out of range"); return QPC_ERR_NO_DEVICE; }
  HIP_TRY(hipSetDevice(device_id));
  QpcDeviceState *S = new QpcDeviceState();
  S->device = device_id;
  const QpcDims &D = c->prog.dims;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  S->n_cu = prop.multiProcessorCount;
  size_t lds = (size_t)D.lds_doubles * sizeof(double);
  if (lds > (size_t)prop.maxSharedMemoryPerMultiProcessor && lds > 160 * 1024) {
    delete S; qpc_set_error("qpc_finalize: LDS slab exceeds the device limit"); return QPC_ERR_UNSUPPORTED;
  }
  HIP_TRY(hipFuncSetAttribute((const void *)qpc_tick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 16) per_cu = 16; // 32 waves/CU cap; leave room for registers
  S->blocks_per_cu = per_cu;
  {
    const size_t slds = (size_t)qpc_sim_layout(D).total * sizeof(double);
    if (slds > 160 * 1024) { delete S; qpc_set_error("qpc_finalize: simulation LDS slab exceeds the device limit"); return QPC_ERR_UNSUPPORTED; }
    HIP_TRY(hipFuncSetAttribute((const void *)qpc_sim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
    int sper = (int)((160 * 1024) / (slds > 0 ? slds : 1));
    S->sim_blocks_per_cu = sper < 1 ? 1 : (sper > 8 ? 8 : sper);
  }
  HIP_TRY(hipMalloc(&S->d_it, sizeof(int) * c->prog.itab.size()));
  HIP_TRY(hipMalloc(&S->d_dt, sizeof(double) * c->prog.dtab.size()));
  HIP_TRY(hipMemcpy(S->d_it, c->prog.itab.data(), sizeof(int) * c->prog.itab.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S->d_dt, c->prog.dtab.data(), sizeof(double) * c->prog.dtab.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipStreamCreate(&S->stream));
  HIP_TRY(hipMalloc(&S->d_ticket, sizeof(unsigned long long)));
  HIP_TRY(hipMemset(S->d_ticket, 0, sizeof(unsigned long long)));
  HIP_TRY(hipMalloc(&S->d_scratch, sizeof(double) * QPC_SCRATCH_DOUBLES * (size_t)S->n_cu * S->blocks_per_cu));
  // JIT specialisation on this controller's program (QPC_DISABLE_JIT=1 keeps the generic kernel)
  const char *nojit = getenv("QPC_DISABLE_JIT");
  if (nojit && nojit[0] == '1' && D.maxdepth > QPC_GENERIC_MAXD) {
    qpc_set_error("QPC_DISABLE_JIT=1 but the tree is deeper than the generic kernel supports");
    return QPC_ERR_UNSUPPORTED;
  }
  if (!(nojit && nojit[0] == '1')) {
    std::vector<char> code;
    std::string log;
    std::string arch = prop.gcnArchName;
    int jrc = qpc_jit_compile(D, arch, code, log);
    if (jrc == 0 && hipModuleLoadData(&S->jit_module, code.data()) == hipSuccess &&
        hipModuleGetFunction(&S->jit_func, S->jit_module, "qpc_tick_kernel_jit") == hipSuccess) {
      // specialised kernel ready; the simulation step of the same module (optional)
      if (hipModuleGetFunction(&S->jit_sim_func, S->jit_module, "qpc_sim_kernel_jit") != hipSuccess) S->jit_sim_func = nullptr;
    } else {
      S->jit_func = nullptr; S->jit_sim_func = nullptr;
      const char *strict = getenv("QPC_REQUIRE_JIT");
      if (strict && strict[0] == '1') {
        qpc_set_error("JIT specialisation failed: " + log);
        return QPC_ERR_HIP;
      }
      if (D.maxdepth > QPC_GENERIC_MAXD) {
        qpc_set_error("JIT specialisation failed and the tree is deeper than the generic kernel supports: " + log);
        return QPC_ERR_UNSUPPORTED;
      }
      fprintf(stderr, "[qpc] JIT specialisation unavailable (%s); using the generic gfx950 kernel\n", log.c_str());
    }
  }
  c->dev = S;
  c->finalized = true;
  if (out) *out = c->layout;
  return QPC_OK;
}

void qpc_controller_destroy(qpc_controller *c) {
  if (!c) return;
  if (c->dev) {
    QpcDeviceState *S = c->dev;
    (void)hipSetDevice(S->device);
    if (S->stream) (void)hipStreamSynchronize(S->stream);
    for (auto &p : S->events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto &e : S->pool) (void)hipEventDestroy(e);
    if (S->d_it) (void)hipFree(S->d_it);
    if (S->d_dt) (void)hipFree(S->d_dt);
    if (S->stage) (void)hipFree(S->stage);
    if (S->d_ticket) (void)hipFree(S->d_ticket);
    if (S->d_scratch) (void)hipFree(S->d_scratch);
    if (S->stream) (void)hipStreamDestroy(S->stream);
    if (S->jit_module) (void)hipModuleUnload(S->jit_module);
    delete S;
  }
  delete c;
}

int32_t qpc_control_batch_device(qpc_controller *c, int64_t B, const double *q, const double *v,
                                 const double *desired, const double *cw, const double *cmax, const double *tw,
                                 const double *surf, double *tau, double *vdot, double *cwrench, int32_t *status,
                                 int32_t *iters, void *hip_stream) {
  if (!c || !c->finalized || !c->dev) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  const QpcDims &D = c->prog.dims;
  if (B < 0 || !q || !v || !tau || !vdot || !status || (D.n_des_inputs && !desired) || (D.nc && (!cw || !cmax)) ||
      (D.n_tw && !tw) || (D.n_surf && !surf)) {
    qpc_set_error("qpc_control_batch_device: missing array"); return QPC_ERR_INVALID_ARGUMENT;
  }
  QpcBatchArgs a;
  a.B = B; a.q = q; a.v = v; a.desired = desired; a.cw = cw; a.cmax = cmax; a.tw = tw; a.surf = surf;
  a.tau = tau; a.vdot = vdot; a.cwrench = cwrench; a.status = status; a.iters = iters;
  a.prof = c->dev->prof;
  HIP_TRY(hipSetDevice(c->dev->device));
  return launch(c, a, (hipStream_t)hip_stream);
}

int32_t qpc_control_batch(qpc_controller *c, int64_t B, const double *q, const double *v, const double *desired,
                          const double *cw, const double *cmax, const double *tw, const double *surf, double *tau,
                          double *vdot, double *cwrench, int32_t *status, int32_t *iters) {
  if (!c || !c->finalized || !c->dev) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  const QpcDims &D = c->prog.dims;
  QpcDeviceState *S = c->dev;
  if (B < 0 || !q || !v || !tau || !vdot || !status || (D.n_des_inputs && !desired) || (D.nc && (!cw || !cmax)) ||
      (D.n_tw && !tw) || (D.n_surf && !surf)) {
    qpc_set_error("qpc_control_batch: missing array"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return QPC_OK;
  HIP_TRY(hipSetDevice(S->device));
  // one staging slab: [q v desired cw cmax tw surf | tau vdot cwrench | status iters]
  size_t nin[7] = {(size_t)D.nq, (size_t)D.nv, desired ? (size_t)D.n_desired : 0, (size_t)D.nc, (size_t)D.nc, (size_t)D.n_tw, (size_t)D.n_surf};
  const double *hin[7] = {q, v, desired, cw, cmax, tw, surf};
  size_t nout[3] = {(size_t)D.nv, (size_t)D.nv, (size_t)6 * D.nc};
  double *hout[3] = {tau, vdot, cwrench};
  size_t total = 0;
  for (int i = 0; i < 7; i++) total += nin[i] * B * sizeof(double);
  for (int i = 0; i < 3; i++) total += nout[i] * B * sizeof(double);
  total += 2 * B * sizeof(int32_t) + 64;
  if (total > S->stage_bytes) {
    if (S->stage) HIP_TRY(hipFree(S->stage));
    S->stage = nullptr; S->stage_bytes = 0;
    HIP_TRY(hipMalloc(&S->stage, total));
    S->stage_bytes = total;
  }
  char *p = (char *)S->stage;
  const double *din[7];
  for (int i = 0; i < 7; i++) {
    din[i] = (const double *)p;
    size_t bytes = nin[i] * B * sizeof(double);
    if (bytes) HIP_TRY(hipMemcpyAsync(p, hin[i], bytes, hipMemcpyHostToDevice, S->stream));
    p += bytes;
  }
  double *dout[3];
  for (int i = 0; i < 3; i++) { dout[i] = (double *)p; p += nout[i] * B * sizeof(double); }
  int32_t *dstatus = (int32_t *)p; p += B * sizeof(int32_t);
  int32_t *diters = (int32_t *)p;
  QpcBatchArgs a;
  a.B = B; a.q = din[0]; a.v = din[1]; a.desired = din[2]; a.cw = din[3]; a.cmax = din[4]; a.tw = din[5]; a.surf = din[6];
  a.tau = dout[0]; a.vdot = dout[1]; a.cwrench = D.nc ? dout[2] : nullptr; a.status = dstatus; a.iters = diters;
  a.prof = S->prof;
  int rc = launch(c, a, S->stream);
  if (rc != QPC_OK) return rc;
  for (int i = 0; i < 3; i++) {
    size_t bytes = nout[i] * B * sizeof(double);
    if (bytes && hout[i]) HIP_TRY(hipMemcpyAsync(hout[i], dout[i], bytes, hipMemcpyDeviceToHost, S->stream));
  }
  HIP_TRY(hipMemcpyAsync(status, dstatus, B * sizeof(int32_t), hipMemcpyDeviceToHost, S->stream));
  if (iters) HIP_TRY(hipMemcpyAsync(iters, diters, B * sizeof(int32_t), hipMemcpyDeviceToHost, S->stream));
  HIP_TRY(hipStreamSynchronize(S->stream));
  return QPC_OK;
}

int32_t qpc_assemble_solve_batch(qpc_controller *c, int64_t B, const double *task_J, const double *task_bias,
                                 const double *desired, const double *momentum_A, const double *momentum_rhs,
                                 const double *contact_R, const double *contact_p, const double *cw, const double *cmax,
                                 const double *tw, const double *surf, double *vdot, double *cwrench, int32_t *status,
                                 int32_t *iters) {
  if (!c || !c->finalized || !c->dev) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  const int nv = c->model->nv, nd = c->n_desired, nc = (int)c->contacts.size(), fl = c->floating >= 0;
  if (B < 0 || !vdot || !status || (nd && (!task_J || !task_bias || !desired)) || (fl && (!momentum_A || !momentum_rhs)) ||
      (nc && (!contact_R || !contact_p || !cw || !cmax)) || (c->n_tw && !tw) || (c->n_surf && !surf)) {
    qpc_set_error("qpc_assemble_solve_batch: missing array"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return QPC_OK;
  if (c->N > 8) { qpc_set_error("qpc_assemble_solve_batch: friction basis larger than 8"); return QPC_ERR_UNSUPPORTED; }
  HIP_TRY(hipSetDevice(c->dev->device));
  std::vector<int> it;
  std::vector<double> dt;
  int n = 0, m = 0;
  qpc_l1_build_tables(c, it, dt, n, m);
  const long long wsd = qpc::l1_workspace_doubles(n, m);
  // one device slab: [tables | inputs | workspace | outputs]
  const size_t nin[11] = {(size_t)nd * nv, (size_t)nd, (size_t)nd, fl ? (size_t)6 * nv : 0, fl ? (size_t)6 : 0, (size_t)9 * nc,
                          (size_t)3 * nc, (size_t)nc, (size_t)nc, (size_t)c->n_tw, (size_t)c->n_surf};
  const double *hin[11] = {task_J, task_bias, desired, momentum_A, momentum_rhs, contact_R, contact_p, cw, cmax, tw, surf};
  size_t total = it.size() * sizeof(int) + 64 + dt.size() * sizeof(double);
  for (int i = 0; i < 11; i++) total += nin[i] * B * sizeof(double);
  total += ((size_t)wsd + nv + 6 * (size_t)(nc > 0 ? nc : 1)) * B * sizeof(double) + 2 * (size_t)B * sizeof(int32_t) + 256;
  char *buf = nullptr;
  if (hipMalloc(&buf, total) != hipSuccess) { qpc_set_error("qpc_assemble_solve_batch: out of device memory"); return QPC_ERR_OUT_OF_MEMORY; }
  int rc = QPC_OK;
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  char *p = buf;
  auto take = [&](size_t bytes) { char *r = p; p += (bytes + 63) & ~(size_t)63; return r; };
  QpcL1Args a;
  a.B = B; a.nv = nv; a.nd = nd; a.nc = nc; a.ntw = c->n_tw; a.nsurf = c->n_surf; a.ws_stride = wsd;
  double *ddt = (double *)take(dt.size() * sizeof(double));
  int *dit = (int *)take(it.size() * sizeof(int));
  chk(hipMemcpy(ddt, dt.data(), dt.size() * sizeof(double), hipMemcpyHostToDevice), "copy tables");
  chk(hipMemcpy(dit, it.data(), it.size() * sizeof(int), hipMemcpyHostToDevice), "copy tables");
  a.it = dit; a.dt = ddt;
  const double *din[11];
  for (int i = 0; i < 11; i++) {
    const size_t bytes = nin[i] * B * sizeof(double);
    din[i] = (const double *)take(bytes);
    if (bytes) chk(hipMemcpy((void *)din[i], hin[i], bytes, hipMemcpyHostToDevice), "copy inputs");
  }
  a.J = din[0]; a.bias = din[1]; a.desired = din[2]; a.Am = fl ? din[3] : nullptr; a.mrhs = fl ? din[4] : nullptr;
  a.cR = din[5]; a.cp = din[6]; a.cw = din[7]; a.cmax = din[8]; a.tw = din[9]; a.surf = din[10];
  a.ws = (double *)take((size_t)wsd * B * sizeof(double));
  a.vdot = (double *)take((size_t)nv * B * sizeof(double));
  a.cwrench = (double *)take((size_t)6 * (nc > 0 ? nc : 1) * B * sizeof(double));
  a.status = (int32_t *)take((size_t)B * sizeof(int32_t));
  a.iters = (int32_t *)take((size_t)B * sizeof(int32_t));
  if (rc == QPC_OK) {
    hipLaunchKernelGGL(qpc_l1_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, 0, a, c->settings);
    chk(hipGetLastError(), "launch qpc_l1_kernel");
    chk(hipDeviceSynchronize(), "qpc_l1_kernel");
    chk(hipMemcpy(vdot, a.vdot, (size_t)nv * B * sizeof(double), hipMemcpyDeviceToHost), "copy vdot");
    if (cwrench && nc) chk(hipMemcpy(cwrench, a.cwrench, (size_t)6 * nc * B * sizeof(double), hipMemcpyDeviceToHost), "copy wrenches");
    chk(hipMemcpy(status, a.status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost), "copy status");
    if (iters) chk(hipMemcpy(iters, a.iters, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost), "copy iters");
  }
  (void)hipFree(buf);
  return rc;
}

int32_t qpc_qp_solve_batch(int32_t n, int32_t m, int64_t B, const double *P, const double *q, const double *A,
                           const double *l, const double *u, const qpc_settings *settings, int32_t device_id,
                           double *x, double *y, int32_t *status, int32_t *iters) {
  if (n <= 0 || m < 0 || B < 0 || !P || !q || (m && (!A || !l || !u)) || !x || (m && !y) || !status) {
    qpc_set_error("qpc_qp_solve_batch: bad arguments"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return QPC_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    qpc_set_error("qpc_qp_solve_batch: no HIP device available (this library has no CPU fallback)");
    return QPC_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) { qpc_set_error("qpc_qp_solve_batch: device_id out of range"); return QPC_ERR_NO_DEVICE; }
  HIP_TRY(hipSetDevice(device_id));
  qpc_settings st;
  if (settings) st = *settings; else qpc_default_settings(&st);
  const long long wsd = qpc::qp_dense_workspace_doubles(n, m);
  const size_t nP = (size_t)B * n * n, nq = (size_t)B * n, nA = (size_t)B * m * n, nm = (size_t)B * m;
  const size_t total = (nP + nq + nA + 2 * nm + nq + nm + (size_t)B * wsd) * sizeof(double) + 2 * (size_t)B * sizeof(int32_t) + 64;
  char *buf = nullptr;
  if (hipMalloc(&buf, total) != hipSuccess) { qpc_set_error("qpc_qp_solve_batch: out of device memory"); return QPC_ERR_OUT_OF_MEMORY; }
  double *dP = (double *)buf, *dq = dP + nP, *dA = dq + nq, *dl = dA + nA, *du = dl + nm, *dx = du + nm, *dy = dx + nq,
         *dws = dy + nm;
  int32_t *dst = (int32_t *)(dws + (size_t)B * wsd), *dit = dst + B;
  int rc = QPC_OK;
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  chk(hipMemcpy(dP, P, nP * sizeof(double), hipMemcpyHostToDevice), "copy P");
  chk(hipMemcpy(dq, q, nq * sizeof(double), hipMemcpyHostToDevice), "copy q");
  if (m) {
    chk(hipMemcpy(dA, A, nA * sizeof(double), hipMemcpyHostToDevice), "copy A");
    chk(hipMemcpy(dl, l, nm * sizeof(double), hipMemcpyHostToDevice), "copy l");
    chk(hipMemcpy(du, u, nm * sizeof(double), hipMemcpyHostToDevice), "copy u");
  }
  if (rc == QPC_OK) {
    hipLaunchKernelGGL(qpc_qp_dense_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, 0, (int)n, (int)m, (long long)B, dP, dq,
                       dA, dl, du, st, dws, wsd, dx, dy, dst, dit);
    chk(hipGetLastError(), "launch qpc_qp_dense_kernel");
    chk(hipDeviceSynchronize(), "qpc_qp_dense_kernel");
    chk(hipMemcpy(x, dx, nq * sizeof(double), hipMemcpyDeviceToHost), "copy x");
    if (m) chk(hipMemcpy(y, dy, nm * sizeof(double), hipMemcpyDeviceToHost), "copy y");
    chk(hipMemcpy(status, dst, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost), "copy status");
    if (iters) chk(hipMemcpy(iters, dit, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost), "copy iters");
  }
  (void)hipFree(buf);
  return rc;
}

static int sim_check(qpc_controller *c, int64_t B, const void *q, const void *v, const void *tau, const qpc_sim_params *p, const char *who) {
  if (!c || !c->finalized || !c->dev) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  if (B < 0 || !q || !v || !tau || !p) { qpc_set_error(std::string(who) + ": missing array"); return QPC_ERR_INVALID_ARGUMENT; }
  if (!c->dev->jit_sim_func && c->prog.dims.maxdepth > QPC_GENERIC_MAXD) {
    qpc_set_error(std::string(who) + ": tree deeper than the generic kernel supports and no JIT kernel"); return QPC_ERR_UNSUPPORTED;
  }
  if (!(p->dt > 0.0) || !(p->stiffness >= 0.0) || !(p->damping >= 0.0) || !(p->tangential_damping >= 0.0)) {
    qpc_set_error(std::string(who) + ": dt must be positive and the ground gains non-negative"); return QPC_ERR_INVALID_ARGUMENT;
  }
  return QPC_OK;
}

int32_t qpc_sim_step_batch_device(qpc_controller *c, int64_t B, double *q, double *v, const double *tau, const int32_t *status,
                                  const qpc_sim_params *params, int32_t *bad, void *hip_stream) {
  int rc = sim_check(c, B, q, v, tau, params, "qpc_sim_step_batch_device");
  if (rc != QPC_OK) return rc;
  HIP_TRY(hipSetDevice(c->dev->device));
  QpcSimArgs a;
  a.B = B; a.q = q; a.v = v; a.tau = tau; a.status = status; a.bad = bad; a.sp = *params;
  return launch_sim(c, a, (hipStream_t)hip_stream);
}

int32_t qpc_rollout_batch_device(qpc_controller *c, int64_t B, int64_t n_steps, double *q, double *v, const double *desired,
                                 const double *cw, const double *cmax, const double *tw, const double *surf,
                                 const qpc_sim_params *params, double *tau, double *vdot, int32_t *status, int32_t *bad,
                                 void *hip_stream) {
  int rc = sim_check(c, B, q, v, tau, params, "qpc_rollout_batch_device");
  if (rc != QPC_OK) return rc;
  if (n_steps < 0 || !vdot || !status) { qpc_set_error("qpc_rollout_batch_device: missing array"); return QPC_ERR_INVALID_ARGUMENT; }
  // two kernels per step on one stream; the tick's launch is asynchronous and its ticket base is
  // tracked on the host, so n_steps steps are enqueued back to back without any synchronisation
  for (int64_t s = 0; s < n_steps; s++) {
    rc = qpc_control_batch_device(c, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, nullptr, status, nullptr, hip_stream);
    if (rc != QPC_OK) return rc;
    rc = qpc_sim_step_batch_device(c, B, q, v, tau, status, params, bad, hip_stream);
    if (rc != QPC_OK) return rc;
  }
  return QPC_OK;
}

// host-pointer variants: one device slab, synchronous.  tick = false: one simulation step with the caller's tau / status.
static int rollout_host(qpc_controller *c, int64_t B, int64_t n_steps, bool tick, double *q, double *v, const double *desired,
                        const double *cw, const double *cmax, const double *tw, const double *surf, const qpc_sim_params *params,
                        double *tau, int32_t *status, int32_t *bad) {
  const QpcDims &D = c->prog.dims;
  QpcDeviceState *S = c->dev;
  if (B == 0) return QPC_OK;
  HIP_TRY(hipSetDevice(S->device));
  enum { A_Q, A_V, A_TAU, A_VDOT, A_DES, A_CW, A_CMAX, A_TW, A_SURF, A_COUNT };
  const size_t dim[A_COUNT] = {(size_t)D.nq, (size_t)D.nv, (size_t)D.nv, (size_t)D.nv, desired ? (size_t)D.n_desired : 0,
                               (size_t)D.nc, (size_t)D.nc, (size_t)D.n_tw, (size_t)D.n_surf};
  const double *host_in[A_COUNT] = {q, v, tick ? nullptr : tau, nullptr, desired, cw, cmax, tw, surf};
  auto padded = [&](int i) { return ((dim[i] * B * sizeof(double)) + 63) & ~(size_t)63; };
  size_t total = 2 * (size_t)B * sizeof(int32_t) + 256;
  for (int i = 0; i < A_COUNT; i++) total += padded(i);
  char *buf = nullptr;
  if (hipMalloc(&buf, total) != hipSuccess) { qpc_set_error("rollout: out of device memory"); return QPC_ERR_OUT_OF_MEMORY; }
  int rc = QPC_OK;
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  char *p = buf;
  double *dev[A_COUNT];
  for (int i = 0; i < A_COUNT; i++) { dev[i] = (double *)p; p += padded(i); }
  int32_t *dstatus = (int32_t *)p, *dbad = dstatus + B;
  for (int i = 0; i < A_COUNT; i++)
    if (host_in[i] && dim[i])
      chk(hipMemcpyAsync(dev[i], host_in[i], dim[i] * B * sizeof(double), hipMemcpyHostToDevice, S->stream), "copy inputs");
  if (!tick && status) chk(hipMemcpyAsync(dstatus, status, B * sizeof(int32_t), hipMemcpyHostToDevice, S->stream), "copy status");
  if (bad) chk(hipMemcpyAsync(dbad, bad, B * sizeof(int32_t), hipMemcpyHostToDevice, S->stream), "copy bad");
  else chk(hipMemsetAsync(dbad, 0, B * sizeof(int32_t), S->stream), "clear bad");
  if (rc == QPC_OK) {
    if (tick)
      rc = qpc_rollout_batch_device(c, B, n_steps, dev[A_Q], dev[A_V], desired ? dev[A_DES] : nullptr, dev[A_CW], dev[A_CMAX],
                                    dev[A_TW], dev[A_SURF], params, dev[A_TAU], dev[A_VDOT], dstatus, dbad, S->stream);
    else
      rc = qpc_sim_step_batch_device(c, B, dev[A_Q], dev[A_V], dev[A_TAU], status ? dstatus : nullptr, params, dbad, S->stream);
  }
  if (rc == QPC_OK) {
    chk(hipMemcpyAsync(q, dev[A_Q], dim[A_Q] * B * sizeof(double), hipMemcpyDeviceToHost, S->stream), "copy q");
    chk(hipMemcpyAsync(v, dev[A_V], dim[A_V] * B * sizeof(double), hipMemcpyDeviceToHost, S->stream), "copy v");
    if (tick && tau) chk(hipMemcpyAsync(tau, dev[A_TAU], dim[A_TAU] * B * sizeof(double), hipMemcpyDeviceToHost, S->stream), "copy tau");
    if (tick && status) chk(hipMemcpyAsync(status, dstatus, B * sizeof(int32_t), hipMemcpyDeviceToHost, S->stream), "copy status");
    if (bad) chk(hipMemcpyAsync(bad, dbad, B * sizeof(int32_t), hipMemcpyDeviceToHost, S->stream), "copy bad");
  }
  chk(hipStreamSynchronize(S->stream), "rollout");
  (void)hipFree(buf);
  return rc;
}

int32_t qpc_sim_step_batch(qpc_controller *c, int64_t B, double *q, double *v, const double *tau, const int32_t *status,
                           const qpc_sim_params *params, int32_t *bad) {
  int rc = sim_check(c, B, q, v, tau, params, "qpc_sim_step_batch");
  if (rc != QPC_OK) return rc;
  return rollout_host(c, B, 1, false, q, v, nullptr, nullptr, nullptr, nullptr, nullptr, params, const_cast<double *>(tau),
                      const_cast<int32_t *>(status), bad);
}

int32_t qpc_rollout_batch(qpc_controller *c, int64_t B, int64_t n_steps, double *q, double *v, const double *desired,
                          const double *cw, const double *cmax, const double *tw, const double *surf,
                          const qpc_sim_params *params, double *tau, int32_t *status, int32_t *bad) {
  int rc = sim_check(c, B, q, v, q, params, "qpc_rollout_batch");
  if (rc != QPC_OK) return rc;
  const QpcDims &D = c->prog.dims;
  if (n_steps < 0 || (D.n_des_inputs && !desired) || (D.nc && (!cw || !cmax)) || (D.n_tw && !tw) || (D.n_surf && !surf)) {
    qpc_set_error("qpc_rollout_batch: missing array"); return QPC_ERR_INVALID_ARGUMENT;
  }
  return rollout_host(c, B, n_steps, true, q, v, desired, cw, cmax, tw, surf, params, tau, status, bad);
}

// Measured vector-fp64 FMA peak of the device (SURVEY.md 8d asks for a measured peak beside the datasheet's 78.6 TFLOP/s):
// every lane runs 16 independent v_fma_f64 chains, enough waves to fill every SIMD; flops = 2 per FMA.
extern "C" __global__ void __launch_bounds__(256) qpc_fp64_peak_kernel(double *out, int iters, double a, double b) {
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = a + 1e-3 * k + 1e-6 * threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x[k] = fma(x[k], a, b);
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) s += x[k];
  if (s == 123.456) out[0] = s; // never true: keeps the chains alive
}

int32_t qpc_debug_measure_fp64_peak(int32_t device_id, double *tflops) {
  if (!tflops) return QPC_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  double *out = nullptr;
  HIP_TRY(hipMalloc(&out, sizeof(double)));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  const int blocks = prop.multiProcessorCount * 8, iters = 20000;
  double best = 0.0;
  for (int rep = 0; rep < 4; rep++) {
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(qpc_fp64_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-7);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 2.0 * 16.0 * (double)iters * 256.0 * (double)blocks;
    if (rep > 0 && ms > 0.f && fl / (ms * 1e-3) > best) best = fl / (ms * 1e-3);
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
  *tflops = best / 1e12;
  return QPC_OK;
}

// diagnostic (QPC_STAMPS builds): device buffer of B*PS_COUNT uint64 section cycle counters
int32_t qpc_debug_set_profile_buffer(qpc_controller *c, void *device_buffer) {
  if (!c || !c->dev) return QPC_ERR_NOT_FINALIZED;
  c->dev->prof = (unsigned long long *)device_buffer;
  return QPC_OK;
}

int32_t qpc_debug_jit_active(qpc_controller *c) { return (c && c->dev && c->dev->jit_func) ? 1 : 0; }

int32_t qpc_timing_enable(qpc_controller *c, int32_t enable) {
  if (!c || !c->dev) return QPC_ERR_NOT_FINALIZED;
  c->dev->timing = enable != 0;
  return QPC_OK;
}

int32_t qpc_timing_read(qpc_controller *c, double *avg_ms, int64_t *n_launches) {
  if (!c || !c->dev) return QPC_ERR_NOT_FINALIZED;
  QpcDeviceState *S = c->dev;
  HIP_TRY(hipSetDevice(S->device));
  double sum = 0.0;
  long long n = 0;
  for (auto &p : S->events) {
    HIP_TRY(hipEventSynchronize(p.second));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
    sum += ms; n++;
    S->pool.push_back(p.first); S->pool.push_back(p.second);
  }
  S->events.clear();
  if (avg_ms) *avg_ms = n ? sum / n : 0.0;
  if (n_launches) *n_launches = n;
  return QPC_OK;
}

} // extern "C"
