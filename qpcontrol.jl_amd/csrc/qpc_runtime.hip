// qpc_runtime.hip -- the run half of the C ABI (include/qpc.h): qpc_finalize[_devices] (program upload to
// every device), qpc_control_batch[_async] + qpc_wait (host-pointer jobs: sharded over the devices, pipelined in
// chunks, qpc_pipeline.h), qpc_control_batch_device* (kernel launch on the caller's stream), L0 / L1 plumbing
// entries, closed-loop step / rollout, timing hooks.  gfx950 only.
//
// Execution model: one wavefront (64 lanes) per QP, one 64-thread workgroup per wavefront so that
// the phase barrier is a single-wave s_barrier; each workgroup owns one LDS slab
// (QpcDims::lds_doubles) and grid-strides over the batch.  Inputs/outputs are per-QP contiguous
// records, loaded/stored by consecutive lanes (coalesced); H, A, b and every factor live only in
// LDS and never touch HBM.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "qpc_compile.h"
#include "qpc_jit.h"
#include "qpc_pipeline.h"

#include "qpc_device.h"
#ifdef QPC_STAMPS
#define QPC_PROF_T0() do { *PT = clock64(); } while (0)
#define QPC_PROF_ACC(i) do { unsigned long long n__ = clock64(); PA[i] += n__ - *PT; *PT = n__; } while (0)
#endif
#include "qpc_tick.h"
#include "qpc_sim.h"
#include "qpc_qp_dense.h"
#include "qpc_l1_dense.h"

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
  unsigned long long *ticket; // dynamic work distribution: every workgroup draws QP indices from this counter until it
                              // draws one past the end, so a launch draws exactly B + gridDim.x tickets; the workgroup
                              // that draws the LAST of them re-arms the counter (stores 0) for the next launch.  No host
                              // bookkeeping, no memset between launches, safe under hipGraph replay.  Launches that share
                              // a counter must not overlap (one counter per pipeline slot, see QpcSlot)
  unsigned long long *wsets;  // warm-start working sets, 2 x uint64 per QP (in/out), or nullptr (cold start)
  const int32_t *order = nullptr; // ticket t is QP order[t] (a permutation of 0 .. B-1: expensive QPs first, qpc_ticket_order_device), or nullptr
};

extern "C" __global__ void __launch_bounds__(64, 1)
qpc_tick_kernel(const QpcDims D, const QpcDims DS, const int *__restrict__ IT, const double *__restrict__ DT, const QpcBatchArgs A) {
  extern __shared__ __attribute__((aligned(16))) double qpc_lds[];
  qpc::qpc_tick_prologue<double *>(D, DS, IT, qpc_lds);
  // active-set iteration counts differ per QP (p99 ~ 3x mean): QPs are handed out through one
  // device-scope ticket counter instead of a static grid-stride, so no workgroup idles at the tail
  while (true) {
    unsigned long long t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(A.ticket, 1ull);
    const long long tk = (long long)__shfl(t, 0, 64);
    if (tk >= A.B) {
      if ((threadIdx.x & 63) == 0 && tk == A.B + (long long)gridDim.x - 1) atomicExch(A.ticket, 0ull); // last draw of the launch
      break;
    }
    const long long b = A.order ? (long long)A.order[tk] : tk;
    qpc::TickIO io;
    io.q = A.q + b * D.nq; io.v = A.v + b * D.nv;
    io.desired = A.desired + b * D.n_desired;
    io.cw = A.cw + b * D.nc; io.cmax = A.cmax + b * D.nc;
    io.tw = A.tw + b * D.n_tw; io.surf = A.surf + b * D.n_surf;
    io.tau = A.tau + b * D.nv; io.vdot = A.vdot + b * D.nv;
    io.cwrench = A.cwrench ? A.cwrench + b * 6 * D.nc : nullptr;
    io.status = A.status + b; io.iters = A.iters ? A.iters + b : nullptr;
    io.prof = A.prof ? A.prof + b * PS_COUNT : nullptr;
    io.wset = A.wsets ? A.wsets + 2 * b : nullptr;
    qpc::qpc_tick<double *>(D, DS, IT, DT, qpc_lds, io);
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
    qpc::qpc_sim_step<double *>(D, IT, DT, qpc_lds, &io, A.sp);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// One pipeline slot of a device: everything two launches in flight must NOT share -- the ticket
// counter, the staging buffers -- plus the stream the slot's work is
// ordered on.  Worker slots [0, P) and the inline slot P own their stream; the user slot P + 1 (the
// *_device entry points) runs on whatever stream the caller passes and remembers the last one so that
// a change of stream is turned into an event dependency instead of a race on the counter.
struct QpcSlot {
  hipStream_t stream = nullptr; // own stream (null for the user slot)
  unsigned long long *d_ticket = nullptr;
  char *d_stage = nullptr; size_t d_stage_bytes = 0; // device staging slab of the host-pointer path
  char *h_stage = nullptr; size_t h_stage_bytes = 0; // pinned host staging slab (pageable caller arrays)
  hipEvent_t last_event = nullptr; // recorded after the slot's last launch (user slot only)
  hipStream_t last_stream = nullptr;
  bool has_last = false;
};

struct QpcDeviceState {
  int device = -1;
  int *d_it = nullptr;
  double *d_dt = nullptr;
  float *d_dtf = nullptr; // the same tables in single precision (the fp32 tick's module reads these)
  int n_cu = 0, blocks_per_cu = 1, sim_blocks_per_cu = 1;
  std::vector<QpcSlot> slots; // [0, P) workers, [P] inline host path, [P + 1] user streams
  // timing (bench.py / roofline): events on the launch stream
  bool timing = false;
  std::mutex timing_mu;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::vector<hipEvent_t> pool;
  unsigned long long *prof = nullptr; // PS_COUNT cycle counters per QP (diagnostic builds only)
  // JIT-specialised kernels (qpc_jit.cpp); null -> the generic ahead-of-time kernel is launched
  hipModule_t jit_module = nullptr;
  hipFunction_t jit_func = nullptr;
  hipFunction_t jit_sim_func = nullptr;  // qpc_sim_kernel_jit of the same module
  hipFunction_t jit_warm_func = nullptr; // qpc_tick_kernel_jit_warm (qpc_enable_warm_start)
  // persistent buffers: L1 workspace, rollout slab, warm-start working sets of the rollout
  char *l1_buf = nullptr; size_t l1_bytes = 0;
  char *roll_buf = nullptr; size_t roll_bytes = 0;
  unsigned long long *roll_ws = nullptr; size_t roll_ws_bytes = 0;
};

static int hip_fail(hipError_t e, const char *what) {
  qpc_set_error(std::string(what) + ": " + hipGetErrorString(e));
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? QPC_ERR_NO_DEVICE : QPC_ERR_HIP;
}
#define HIP_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_fail(e__, #call); } while (0)

class QpcHipBackend;
struct QpcRuntime {
  std::vector<QpcDeviceState *> devs;
  int slots = 6;            // worker slots per device
  QpcHipBackend *backend = nullptr;
  QpcHostPipeline *pipe = nullptr;
  bool zc_pending = false;  // zero-copy launches in flight on the inline-slot streams (qpc_control_batch_async with pinned arrays)
  bool zero_copy = true;    // QPC_ZERO_COPY=0: always stage through device buffers
  bool stage_zero_copy = true; // pageable arrays: the kernel reads / writes the chunk's pinned staging slab in place over PCIe
                               // (QPC_STAGE_ZEROCOPY=0: DMA the slab to / from device memory instead; measured 10.7 vs 13.2 M solves/s)
  int inline_slot() const { return slots; }
  int user_slot(int k = 0) const { return slots + 1 + k; } // QPC_USER_SLOTS launch slots of the *_device entry points
};

// dynamic-LDS limit of the generic kernels is a per-function attribute shared by every controller of the
// process: never lower it below what an earlier (still live) controller needs
static std::mutex g_attr_mu;
static size_t g_tick_lds_max = 0, g_sim_lds_max = 0;

static int launch(qpc_controller *c, QpcDeviceState *S, QpcSlot &sl, const QpcBatchArgs &a, hipStream_t stream, bool warm) {
  const QpcDims &D = c->prog.dims;
  const bool pack2 = c->qps_per_wave == 2 && !warm; // (the warm-started variant keeps one QP per wavefront)
  long long grid = (long long)S->n_cu * S->blocks_per_cu;
  if (grid > (a.B + (pack2 ? 1 : 0)) / (pack2 ? 2 : 1)) grid = (a.B + (pack2 ? 1 : 0)) / (pack2 ? 2 : 1);
  if (grid < 1) return QPC_OK;
  size_t lds = (size_t)D.lds_doubles * (c->precision == QPC_PRECISION_F32 ? sizeof(float) : sizeof(double));
  const bool user_stream = sl.stream == nullptr;
  bool capturing = false;
  if (user_stream && stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess) capturing = cs != hipStreamCaptureStatusNone;
    else (void)hipGetLastError();
  }
  if (user_stream && !capturing && sl.has_last && sl.last_stream != stream)
    HIP_TRY(hipStreamWaitEvent(stream, sl.last_event, 0)); // launches of one slot share its ticket counter: serialise
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (S->timing && !capturing) {
    std::lock_guard<std::mutex> lk(S->timing_mu);
    auto get = [&](hipEvent_t &e) -> hipError_t {
      if (!S->pool.empty()) { e = S->pool.back(); S->pool.pop_back(); return hipSuccess; }
      return hipEventCreate(&e);
    };
    HIP_TRY(get(e0)); HIP_TRY(get(e1));
    HIP_TRY(hipEventRecord(e0, stream));
  }
  QpcBatchArgs aa = a;
  aa.ticket = sl.d_ticket;
  hipFunction_t fn = warm ? S->jit_warm_func : S->jit_func;
  if (warm && !fn) { qpc_set_error("warm-start kernel not available (qpc_enable_warm_start before qpc_finalize; needs the hiprtc-specialised kernel)"); return QPC_ERR_UNSUPPORTED; }
  if (!warm) aa.wsets = nullptr;
  if (fn) {
    const int *it = S->d_it;
    const void *dt = c->precision == QPC_PRECISION_F32 ? (const void *)S->d_dtf : (const void *)S->d_dt;
    void *args[] = {(void *)&it, (void *)&dt, (void *)&aa};
    unsigned dyn = (qpc_jit_dynamic_lds() && c->precision != QPC_PRECISION_F32) ? (unsigned)(D.lds_doubles * sizeof(double)) : 0u; // experiment: QPC_EXP_DYN_LDS=1
    if (dyn) if (const char *e = QPC_EXP_ENV("QPC_EXP_LDS_PAD")) dyn += (unsigned)atoi(e); // ... QPC_EXP_LDS_PAD=bytes: occupancy at will
    HIP_TRY(hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 64, 1, 1, dyn, stream, args, nullptr));
  } else {
    hipLaunchKernelGGL(qpc_tick_kernel, dim3((unsigned)grid), dim3(64), lds, stream, D, c->prog.dims_s, S->d_it, S->d_dt, aa);
    HIP_TRY(hipGetLastError());
  }
  if (e0) {
    HIP_TRY(hipEventRecord(e1, stream));
    std::lock_guard<std::mutex> lk(S->timing_mu);
    S->events.emplace_back(e0, e1);
  }
  if (user_stream && !capturing) {
    HIP_TRY(hipEventRecord(sl.last_event, stream));
    sl.last_stream = stream; sl.has_last = true;
  }
  return QPC_OK;
}

static int launch_sim(qpc_controller *c, QpcDeviceState *S, const QpcSimArgs &a, hipStream_t stream) {
  const QpcDims &D = c->prog.dims;
  long long grid = (long long)S->n_cu * S->sim_blocks_per_cu;
  if (S->jit_sim_func) { // the specialised step may run two robots per wavefront (qpc_jit_sim_rpw)
    const int rpw = qpc_jit_sim_rpw(D);
    const size_t slab = ((size_t)qpc_sim_layout(D).total + 1) / 2 * 2 * sizeof(double) * rpw;
    long long per_cu = (long long)((160 * 1024) / (slab > 0 ? slab : 1));
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    grid = (long long)S->n_cu * per_cu;
    if (grid > (a.B + rpw - 1) / rpw) grid = (a.B + rpw - 1) / rpw;
  }
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

static void destroy_device_state(QpcDeviceState *S) {
  if (!S) return;
  (void)hipSetDevice(S->device);
  for (auto &sl : S->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.has_last) (void)hipEventSynchronize(sl.last_event);
  }
  for (auto &p : S->events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  for (auto &e : S->pool) (void)hipEventDestroy(e);
  for (auto &sl : S->slots) {
    if (sl.d_ticket) (void)hipFree(sl.d_ticket);
    if (sl.d_stage) (void)hipFree(sl.d_stage);
    if (sl.h_stage) (void)hipHostFree(sl.h_stage);
    if (sl.last_event) (void)hipEventDestroy(sl.last_event);
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  if (S->d_it) (void)hipFree(S->d_it);
  if (S->d_dt) (void)hipFree(S->d_dt);
  if (S->d_dtf) (void)hipFree(S->d_dtf);
  if (S->l1_buf) (void)hipFree(S->l1_buf);
  if (S->roll_buf) (void)hipFree(S->roll_buf);
  if (S->roll_ws) (void)hipFree(S->roll_ws);
  if (S->jit_module) (void)hipModuleUnload(S->jit_module);
  delete S;
}

// program upload + per-slot resources + kernel modules of ONE device.  `code` = the hiprtc code object of the
// controller (empty: generic kernel).  On any failure everything allocated so far is released by the caller
// through destroy_device_state (S is always left in a destructible state).
static int init_device_state(qpc_controller *c, QpcDeviceState *S, int device_id, int n_slots, const std::vector<char> &code,
                             const hipDeviceProp_t &prop) {
  const QpcDims &D = c->prog.dims;
  S->device = device_id;
  HIP_TRY(hipSetDevice(device_id));
  S->n_cu = prop.multiProcessorCount;
  const bool f32 = c->precision == QPC_PRECISION_F32, pack2 = c->qps_per_wave == 2;
  const size_t lds = (size_t)D.lds_doubles * (f32 ? sizeof(float) : sizeof(double)) * (pack2 ? 2 : 1);
  if (lds > 160 * 1024 / (pack2 ? 2 : 1) * (pack2 ? 2 : 1)) { qpc_set_error("qpc_finalize: LDS slab exceeds the device limit"); return QPC_ERR_UNSUPPORTED; }
  const size_t slds = (size_t)qpc_sim_layout(D).total * sizeof(double);
  if (slds > 160 * 1024) { qpc_set_error("qpc_finalize: simulation LDS slab exceeds the device limit"); return QPC_ERR_UNSUPPORTED; }
  {
    std::lock_guard<std::mutex> lk(g_attr_mu);
    if (!f32 && lds > g_tick_lds_max) g_tick_lds_max = lds; // (the generic kernel is fp64 only)
    if (slds > g_sim_lds_max) g_sim_lds_max = slds;
    HIP_TRY(hipFuncSetAttribute((const void *)qpc_tick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_tick_lds_max));
    HIP_TRY(hipFuncSetAttribute((const void *)qpc_sim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_sim_lds_max));
  }
  int per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
  if (per_cu < 1) per_cu = 1;
  if (per_cu > 16) per_cu = 16; // 32 waves/CU cap; leave room for registers
  if ((f32 || pack2) && per_cu > 4 * qpc_jit_tick_waves(f32, pack2)) per_cu = 4 * qpc_jit_tick_waves(f32, pack2); // the persistent grid must be resident at once
  if (const char *e = QPC_EXP_ENV("QPC_EXP_BLOCKS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= 16) per_cu = v; } // experiment
  S->blocks_per_cu = per_cu;
  int sper = (int)((160 * 1024) / (slds > 0 ? slds : 1));
  S->sim_blocks_per_cu = sper < 1 ? 1 : (sper > 8 ? 8 : sper);
  HIP_TRY(hipMalloc(&S->d_it, sizeof(int) * c->prog.itab.size()));
  HIP_TRY(hipMalloc(&S->d_dt, sizeof(double) * c->prog.dtab.size()));
  HIP_TRY(hipMemcpy(S->d_it, c->prog.itab.data(), sizeof(int) * c->prog.itab.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(S->d_dt, c->prog.dtab.data(), sizeof(double) * c->prog.dtab.size(), hipMemcpyHostToDevice));
  {
    std::vector<float> dtf(c->prog.dtab.begin(), c->prog.dtab.end());
    HIP_TRY(hipMalloc(&S->d_dtf, sizeof(float) * (dtf.size() + 1)));
    HIP_TRY(hipMemcpy(S->d_dtf, dtf.data(), sizeof(float) * dtf.size(), hipMemcpyHostToDevice));
  }
  S->slots.resize((size_t)n_slots + 1 + QPC_USER_SLOTS);
  for (size_t i = 0; i < S->slots.size(); i++) {
    QpcSlot &sl = S->slots[i];
    if (i <= (size_t)n_slots) HIP_TRY(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking)); // workers and the inline slot own a stream
    else HIP_TRY(hipEventCreateWithFlags(&sl.last_event, hipEventDisableTiming));                  // user slots run on the caller's
    HIP_TRY(hipMalloc(&sl.d_ticket, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(sl.d_ticket, 0, sizeof(unsigned long long)));
  }
  if (!code.empty()) {
    if (hipModuleLoadData(&S->jit_module, code.data()) == hipSuccess &&
        hipModuleGetFunction(&S->jit_func, S->jit_module, "qpc_tick_kernel_jit") == hipSuccess) {
      if (hipModuleGetFunction(&S->jit_sim_func, S->jit_module, "qpc_sim_kernel_jit") != hipSuccess) S->jit_sim_func = nullptr;
      if (!c->warm_start || hipModuleGetFunction(&S->jit_warm_func, S->jit_module, "qpc_tick_kernel_jit_warm") != hipSuccess) S->jit_warm_func = nullptr;
      (void)hipGetLastError();
    } else {
      (void)hipGetLastError();
      S->jit_func = nullptr; S->jit_sim_func = nullptr; S->jit_warm_func = nullptr;
      qpc_set_error("loading the hiprtc code object failed");
      return QPC_ERR_HIP;
    }
  }
  return QPC_OK;
}

// ------------------------------------------------------------------------------------------
// HIP backend of the host-pointer pipeline (qpc_pipeline.h): one chunk = stage in, tick kernel, stage out
static inline size_t pad64(size_t b) { return (b + 63) & ~(size_t)63; }

class QpcHipBackend : public QpcChunkBackend {
 public:
  explicit QpcHipBackend(qpc_controller *c) : c_(c) {}
  int n_devices() const override { return (int)c_->rt->devs.size(); }
  void bind_thread(int d) override { (void)hipSetDevice(c_->rt->devs[d]->device); }
  int run_chunk(int d, int slot, const QpcHostJob &job, int64_t lo, int64_t n, std::string &err) override {
    const int rc = run(d, slot, job, lo, n);
    if (rc != QPC_OK) err = qpc_last_error(); // thread-local of THIS worker thread -> handed to the waiting caller
    return rc;
  }

 private:
  int run(int d, int slot, const QpcHostJob &job, int64_t lo, int64_t n) {
    QpcDeviceState *S = c_->rt->devs[d];
    QpcSlot &sl = S->slots[slot];
    const QpcDims &D = c_->prog.dims;
    // staging slab: [inputs | outputs | status | iters | working sets], every array 64-byte aligned
    size_t in_off[QpcHostJob::N_IN], out_off[QpcHostJob::N_OUT], off = 0;
    bool all_in_pageable = true, any_in_pageable = false, any_out_pageable = false, all_out_pageable = true;
    for (int i = 0; i < QpcHostJob::N_IN; i++) {
      in_off[i] = off;
      const size_t bytes = job.in[i] ? job.in_dim[i] * (size_t)n * sizeof(double) : 0;
      off += pad64(bytes);
      if (bytes) { if (job.in_pinned[i]) all_in_pageable = false; else any_in_pageable = true; }
    }
    const size_t in_end = off;
    // the kernel always writes tau, vdot, status; cwrench and iters only when the caller wants them
    for (int i = 0; i < QpcHostJob::N_OUT; i++) {
      out_off[i] = off;
      const size_t bytes = (job.out[i] || i != QpcHostJob::OUT_CWRENCH) ? job.out_dim[i] * (size_t)n * sizeof(double) : 0;
      off += pad64(bytes);
      if (job.out[i] && job.out_dim[i]) { if (job.out_pinned[i]) all_out_pageable = false; else any_out_pageable = true; }
    }
    const size_t st_off = off; off += pad64((size_t)n * sizeof(int32_t));
    const size_t it_off = off; off += pad64(job.iters ? (size_t)n * sizeof(int32_t) : 0);
    const size_t ws_off = off; off += pad64(job.working_set ? 2 * (size_t)n * sizeof(uint64_t) : 0);
    const size_t out_end = off;
    if (job.status_pinned) all_out_pageable = false; else any_out_pageable = true;
    if (job.iters) { if (job.iters_pinned) all_out_pageable = false; else any_out_pageable = true; }
    if (job.working_set) { if (job.ws_pinned) { all_out_pageable = false; all_in_pageable = false; } else { any_out_pageable = true; any_in_pageable = true; } }
    if ((any_in_pageable || any_out_pageable) && off > sl.h_stage_bytes) {
      if (sl.h_stage) HIP_TRY(hipHostFree(sl.h_stage));
      sl.h_stage = nullptr; sl.h_stage_bytes = 0;
      HIP_TRY(hipHostMalloc(&sl.h_stage, off, hipHostMallocDefault));
      sl.h_stage_bytes = off;
    }
    hipStream_t st = sl.stream;
    if (c_->rt->stage_zero_copy && all_in_pageable && all_out_pageable) return run_on_staging(S, sl, job, lo, n, in_off, out_off, st_off, it_off, ws_off);
    if (off > sl.d_stage_bytes) { // (the device slab is only needed by the DMA form below)
      if (sl.d_stage) HIP_TRY(hipFree(sl.d_stage));
      sl.d_stage = nullptr; sl.d_stage_bytes = 0;
      HIP_TRY(hipMalloc(&sl.d_stage, off));
      sl.d_stage_bytes = off;
    }
    // ---- inputs
    for (int i = 0; i < QpcHostJob::N_IN; i++) {
      const size_t bytes = job.in[i] ? job.in_dim[i] * (size_t)n * sizeof(double) : 0;
      if (!bytes) continue;
      const double *src = job.in[i] + (size_t)lo * job.in_dim[i];
      if (job.in_pinned[i]) HIP_TRY(hipMemcpyAsync(sl.d_stage + in_off[i], src, bytes, hipMemcpyHostToDevice, st));
      else {
        std::memcpy(sl.h_stage + in_off[i], src, bytes);
        if (!all_in_pageable) HIP_TRY(hipMemcpyAsync(sl.d_stage + in_off[i], sl.h_stage + in_off[i], bytes, hipMemcpyHostToDevice, st));
      }
    }
    if (job.working_set) {
      const size_t bytes = 2 * (size_t)n * sizeof(uint64_t);
      const uint64_t *src = job.working_set + 2 * (size_t)lo;
      if (job.ws_pinned) HIP_TRY(hipMemcpyAsync(sl.d_stage + ws_off, src, bytes, hipMemcpyHostToDevice, st));
      else { std::memcpy(sl.h_stage + ws_off, src, bytes); HIP_TRY(hipMemcpyAsync(sl.d_stage + ws_off, sl.h_stage + ws_off, bytes, hipMemcpyHostToDevice, st)); }
    }
    if (all_in_pageable && in_end) // every input went through the pinned slab: ONE DMA for the whole input region
      HIP_TRY(hipMemcpyAsync(sl.d_stage, sl.h_stage, in_end, hipMemcpyHostToDevice, st));
    // ---- kernel
    auto din = [&](int i) { return job.in[i] ? (const double *)(sl.d_stage + in_off[i]) : nullptr; };
    QpcBatchArgs a;
    a.B = n;
    a.q = din(QpcHostJob::IN_Q); a.v = din(QpcHostJob::IN_V); a.desired = din(QpcHostJob::IN_DES);
    a.cw = din(QpcHostJob::IN_CW); a.cmax = din(QpcHostJob::IN_CMAX); a.tw = din(QpcHostJob::IN_TW); a.surf = din(QpcHostJob::IN_SURF);
    a.tau = (double *)(sl.d_stage + out_off[QpcHostJob::OUT_TAU]); a.vdot = (double *)(sl.d_stage + out_off[QpcHostJob::OUT_VDOT]);
    a.cwrench = (D.nc && job.out[QpcHostJob::OUT_CWRENCH]) ? (double *)(sl.d_stage + out_off[QpcHostJob::OUT_CWRENCH]) : nullptr;
    a.status = (int32_t *)(sl.d_stage + st_off);
    a.iters = job.iters ? (int32_t *)(sl.d_stage + it_off) : nullptr;
    a.wsets = job.working_set ? (unsigned long long *)(sl.d_stage + ws_off) : nullptr;
    a.prof = nullptr;
    int rc = launch(c_, S, sl, a, st, job.working_set != nullptr);
    if (rc != QPC_OK) return rc;
    // ---- outputs
    if (all_out_pageable) {
      HIP_TRY(hipMemcpyAsync(sl.h_stage + in_end, sl.d_stage + in_end, out_end - in_end, hipMemcpyDeviceToHost, st)); // ONE DMA
    } else {
      for (int i = 0; i < QpcHostJob::N_OUT; i++) {
        const size_t bytes = job.out_dim[i] * (size_t)n * sizeof(double);
        if (!job.out[i] || !bytes) continue;
        void *dst = job.out_pinned[i] ? (void *)(job.out[i] + (size_t)lo * job.out_dim[i]) : (void *)(sl.h_stage + out_off[i]);
        HIP_TRY(hipMemcpyAsync(dst, sl.d_stage + out_off[i], bytes, hipMemcpyDeviceToHost, st));
      }
      HIP_TRY(hipMemcpyAsync(job.status_pinned ? (void *)(job.status + lo) : (void *)(sl.h_stage + st_off), sl.d_stage + st_off,
                             (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      if (job.iters)
        HIP_TRY(hipMemcpyAsync(job.iters_pinned ? (void *)(job.iters + lo) : (void *)(sl.h_stage + it_off), sl.d_stage + it_off,
                               (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      if (job.working_set)
        HIP_TRY(hipMemcpyAsync(job.ws_pinned ? (void *)(job.working_set + 2 * (size_t)lo) : (void *)(sl.h_stage + ws_off), sl.d_stage + ws_off,
                               2 * (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (int i = 0; i < QpcHostJob::N_OUT; i++) {
      const size_t bytes = job.out_dim[i] * (size_t)n * sizeof(double);
      if (job.out[i] && bytes && !job.out_pinned[i]) std::memcpy(job.out[i] + (size_t)lo * job.out_dim[i], sl.h_stage + out_off[i], bytes);
    }
    if (!job.status_pinned) std::memcpy(job.status + lo, sl.h_stage + st_off, (size_t)n * sizeof(int32_t));
    if (job.iters && !job.iters_pinned) std::memcpy(job.iters + lo, sl.h_stage + it_off, (size_t)n * sizeof(int32_t));
    if (job.working_set && !job.ws_pinned) std::memcpy(job.working_set + 2 * (size_t)lo, sl.h_stage + ws_off, 2 * (size_t)n * sizeof(uint64_t));
    return QPC_OK;
  }
  // pageable caller arrays, no DMA at all: the kernel reads the chunk's inputs from the pinned staging slab over PCIe and
  // writes its outputs there (what is left is the host memcpy into / out of the slab, done by the slot's worker thread)
  int run_on_staging(QpcDeviceState *S, QpcSlot &sl, const QpcHostJob &job, int64_t lo, int64_t n, const size_t *in_off,
                     const size_t *out_off, size_t st_off, size_t it_off, size_t ws_off) {
    const QpcDims &D = c_->prog.dims;
    for (int i = 0; i < QpcHostJob::N_IN; i++) {
      const size_t bytes = job.in[i] ? job.in_dim[i] * (size_t)n * sizeof(double) : 0;
      if (bytes) std::memcpy(sl.h_stage + in_off[i], job.in[i] + (size_t)lo * job.in_dim[i], bytes);
    }
    if (job.working_set) std::memcpy(sl.h_stage + ws_off, job.working_set + 2 * (size_t)lo, 2 * (size_t)n * sizeof(uint64_t));
    auto hin = [&](int i) { return job.in[i] ? (const double *)(sl.h_stage + in_off[i]) : nullptr; };
    QpcBatchArgs a;
    a.B = n;
    a.q = hin(QpcHostJob::IN_Q); a.v = hin(QpcHostJob::IN_V); a.desired = hin(QpcHostJob::IN_DES);
    a.cw = hin(QpcHostJob::IN_CW); a.cmax = hin(QpcHostJob::IN_CMAX); a.tw = hin(QpcHostJob::IN_TW); a.surf = hin(QpcHostJob::IN_SURF);
    a.tau = (double *)(sl.h_stage + out_off[QpcHostJob::OUT_TAU]); a.vdot = (double *)(sl.h_stage + out_off[QpcHostJob::OUT_VDOT]);
    a.cwrench = (D.nc && job.out[QpcHostJob::OUT_CWRENCH]) ? (double *)(sl.h_stage + out_off[QpcHostJob::OUT_CWRENCH]) : nullptr;
    a.status = (int32_t *)(sl.h_stage + st_off);
    a.iters = job.iters ? (int32_t *)(sl.h_stage + it_off) : nullptr;
    a.wsets = job.working_set ? (unsigned long long *)(sl.h_stage + ws_off) : nullptr;
    a.prof = nullptr;
    int rc = launch(c_, S, sl, a, sl.stream, job.working_set != nullptr);
    if (rc != QPC_OK) return rc;
    HIP_TRY(hipStreamSynchronize(sl.stream));
    for (int i = 0; i < QpcHostJob::N_OUT; i++) {
      const size_t bytes = job.out_dim[i] * (size_t)n * sizeof(double);
      if (job.out[i] && bytes) std::memcpy(job.out[i] + (size_t)lo * job.out_dim[i], sl.h_stage + out_off[i], bytes);
    }
    std::memcpy(job.status + lo, sl.h_stage + st_off, (size_t)n * sizeof(int32_t));
    if (job.iters) std::memcpy(job.iters + lo, sl.h_stage + it_off, (size_t)n * sizeof(int32_t));
    if (job.working_set) std::memcpy(job.working_set + 2 * (size_t)lo, sl.h_stage + ws_off, 2 * (size_t)n * sizeof(uint64_t));
    return QPC_OK;
  }
  qpc_controller *c_;
};

// is `p` page-locked host memory HIP can DMA from / to directly (hipHostMalloc, hipHostRegister, torch pin_memory)?
static bool host_pinned(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  std::memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

static int fill_job(qpc_controller *c, QpcHostJob &job, int64_t B, const double *q, const double *v, const double *desired,
                    const double *cw, const double *cmax, const double *tw, const double *surf, double *tau, double *vdot,
                    double *cwrench, int32_t *status, int32_t *iters, uint64_t *wsets, const char *who) {
  if (!c || !c->finalized || !c->rt) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  const QpcDims &D = c->prog.dims;
  if (B < 0 || !q || !v || !tau || !vdot || !status || (D.n_des_inputs && !desired) || (D.nc && (!cw || !cmax)) ||
      (D.n_tw && !tw) || (D.n_surf && !surf)) {
    qpc_set_error(std::string(who) + ": missing array"); return QPC_ERR_INVALID_ARGUMENT;
  }
  job.B = B;
  const double *hin[QpcHostJob::N_IN] = {q, v, desired, D.nc ? cw : nullptr, D.nc ? cmax : nullptr, D.n_tw ? tw : nullptr, D.n_surf ? surf : nullptr};
  const size_t nin[QpcHostJob::N_IN] = {(size_t)D.nq, (size_t)D.nv, (size_t)D.n_desired, (size_t)D.nc, (size_t)D.nc, (size_t)D.n_tw, (size_t)D.n_surf};
  for (int i = 0; i < QpcHostJob::N_IN; i++) { job.in[i] = nin[i] ? hin[i] : nullptr; job.in_dim[i] = nin[i]; job.in_pinned[i] = job.in[i] && host_pinned(job.in[i]); }
  double *hout[QpcHostJob::N_OUT] = {tau, vdot, D.nc ? cwrench : nullptr};
  const size_t nout[QpcHostJob::N_OUT] = {(size_t)D.nv, (size_t)D.nv, (size_t)6 * D.nc};
  for (int i = 0; i < QpcHostJob::N_OUT; i++) { job.out[i] = hout[i]; job.out_dim[i] = nout[i]; job.out_pinned[i] = job.out[i] && host_pinned(job.out[i]); }
  job.status = status; job.iters = iters; job.working_set = wsets;
  job.status_pinned = host_pinned(status); job.iters_pinned = iters && host_pinned(iters); job.ws_pinned = wsets && host_pinned(wsets);
  return QPC_OK;
}

// Every array of the job is page-locked: no staging at all.  One launch per device on its shard, the kernel reads the
// caller's records over PCIe (one round trip per QP: qpc_tick.h input phase) and writes tau, vdot, wrenches, statuses
// straight into the caller's arrays.  Measured (MI355X, config 3, 32768 QPs): 85 % of the device-resident rate, against
// 50-60 % for chunked H2D || kernel || D2H.
static bool job_all_pinned(const QpcHostJob &j) {
  for (int i = 0; i < QpcHostJob::N_IN; i++) if (j.in[i] && j.in_dim[i] && !j.in_pinned[i]) return false;
  for (int i = 0; i < QpcHostJob::N_OUT; i++) if (j.out[i] && j.out_dim[i] && !j.out_pinned[i]) return false;
  if (!j.status_pinned || (j.iters && !j.iters_pinned) || (j.working_set && !j.ws_pinned)) return false;
  return true;
}
template <class T> static T *dev_view(T *host) { // device address of page-locked host memory (identity on ROCm; asked for anyway)
  if (!host) return nullptr;
  void *d = nullptr;
  if (hipHostGetDevicePointer(&d, (void *)host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return (T *)d;
}
static int zero_copy_launch(qpc_controller *c, const QpcHostJob &job) {
  QpcRuntime *rt = c->rt;
  const QpcDims &D = c->prog.dims;
  const int G = (int)rt->devs.size();
  const double *din[QpcHostJob::N_IN];
  double *dout[QpcHostJob::N_OUT];
  for (int i = 0; i < QpcHostJob::N_IN; i++) { din[i] = (job.in[i] && job.in_dim[i]) ? dev_view(job.in[i]) : nullptr; if (job.in[i] && job.in_dim[i] && !din[i]) return -1; }
  for (int i = 0; i < QpcHostJob::N_OUT; i++) { dout[i] = (job.out[i] && job.out_dim[i]) ? dev_view(job.out[i]) : nullptr; if (job.out[i] && job.out_dim[i] && !dout[i]) return -1; }
  int32_t *dstatus = dev_view(job.status), *diters = dev_view(job.iters);
  unsigned long long *dws = (unsigned long long *)dev_view(job.working_set);
  if (!dstatus || (job.iters && !diters) || (job.working_set && !dws)) return -1;
  for (int g = 0; g < G; g++) {
    int64_t lo, hi;
    qpc_shard_bounds(job.B, g, G, lo, hi);
    if (hi == lo) continue;
    QpcDeviceState *S = rt->devs[g];
    HIP_TRY(hipSetDevice(S->device));
    QpcSlot &sl = S->slots[rt->inline_slot()];
    auto in = [&](int i) { return din[i] ? din[i] + (size_t)lo * job.in_dim[i] : nullptr; };
    auto out = [&](int i) { return dout[i] ? dout[i] + (size_t)lo * job.out_dim[i] : nullptr; };
    QpcBatchArgs a;
    a.B = hi - lo;
    a.q = in(QpcHostJob::IN_Q); a.v = in(QpcHostJob::IN_V); a.desired = in(QpcHostJob::IN_DES); a.cw = in(QpcHostJob::IN_CW);
    a.cmax = in(QpcHostJob::IN_CMAX); a.tw = in(QpcHostJob::IN_TW); a.surf = in(QpcHostJob::IN_SURF);
    a.tau = out(QpcHostJob::OUT_TAU); a.vdot = out(QpcHostJob::OUT_VDOT); a.cwrench = D.nc ? out(QpcHostJob::OUT_CWRENCH) : nullptr;
    a.status = dstatus + lo; a.iters = diters ? diters + lo : nullptr; a.wsets = dws ? dws + 2 * lo : nullptr;
    a.prof = nullptr;
    rt->zc_pending = true;
    int rc = launch(c, S, sl, a, sl.stream, dws != nullptr);
    if (rc != QPC_OK) return rc;
  }
  HIP_TRY(hipSetDevice(rt->devs[0]->device));
  return QPC_OK;
}
static int zero_copy_wait(qpc_controller *c) {
  QpcRuntime *rt = c->rt;
  if (!rt->zc_pending) return QPC_OK;
  rt->zc_pending = false;
  int rc = QPC_OK;
  for (auto *S : rt->devs) {
    if (hipSetDevice(S->device) != hipSuccess) continue;
    hipError_t e = hipStreamSynchronize(S->slots[rt->inline_slot()].stream);
    if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, "qpc_wait: zero-copy tick");
  }
  (void)hipSetDevice(rt->devs[0]->device);
  return rc;
}

extern "C" {

int32_t qpc_finalize_devices(qpc_controller *c, int32_t n_devices, const int32_t *device_ids, qpc_layout *out) {
  if (!c || n_devices < 1 || !device_ids) { qpc_set_error("qpc_finalize_devices: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  if (c->finalized) { if (out) *out = c->layout; return QPC_OK; }
  int rc = qpc_compile_program(c);
  if (rc != QPC_OK) return rc;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    qpc_set_error("qpc_finalize: no HIP device available (this library has no CPU fallback)");
    return QPC_ERR_NO_DEVICE;
  }
  for (int i = 0; i < n_devices; i++) {
    if (device_ids[i] < 0 || device_ids[i] >= ndev) { qpc_set_error("qpc_finalize: device_id out of range"); return QPC_ERR_NO_DEVICE; }
    for (int j = 0; j < i; j++) if (device_ids[j] == device_ids[i]) { qpc_set_error("qpc_finalize_devices: duplicate device id"); return QPC_ERR_INVALID_ARGUMENT; }
  }
  const QpcDims &D = c->prog.dims;
  hipDeviceProp_t prop0;
  HIP_TRY(hipGetDeviceProperties(&prop0, device_ids[0]));
  // JIT specialisation on this controller's program, once for all devices (QPC_DISABLE_JIT=1 keeps the generic kernel)
  std::vector<char> code;
  std::string jit_disk_hit; // the code object came from this file of the disk cache
  const char *nojit = getenv("QPC_DISABLE_JIT");
  const bool jit_off = nojit && nojit[0] == '1';
  const bool f32 = c->precision == QPC_PRECISION_F32, pack2 = c->qps_per_wave == 2;
  if (pack2 && (jit_off || c->warm_start || D.presolve > 2 || (D.nrho > 0 && (D.N != 4 || D.nrho > 32)))) {
    qpc_set_error("qpc_set_qps_per_wavefront(2): needs the hiprtc-specialised kernel, no warm start, a four-ray contact basis with at most 32 rays "
                  "and at most two presolved foot blocks");
    return QPC_ERR_UNSUPPORTED;
  }
  if (jit_off && f32) {
    qpc_set_error("QPC_DISABLE_JIT=1 but the fp32 tick (qpc_set_precision) only exists as a hiprtc-specialised kernel");
    return QPC_ERR_UNSUPPORTED;
  }
  if (f32 && c->warm_start) { // (the header says so, nothing enforced it -- an untested fp32 warm kernel would have been compiled and launched)
    qpc_set_error("qpc_enable_warm_start with QPC_PRECISION_F32: warm starts are fp64-only (qpc_set_precision, include/qpc.h)");
    return QPC_ERR_UNSUPPORTED;
  }
  if (jit_off && D.maxdepth > QPC_GENERIC_MAXD) {
    qpc_set_error("QPC_DISABLE_JIT=1 but the tree is deeper than the generic kernel supports");
    return QPC_ERR_UNSUPPORTED;
  }
  if (!jit_off) {
    std::string log;
    const int jrc = qpc_jit_compile(D, c->prog.dims_s, prop0.gcnArchName, c->warm_start != 0, f32, pack2, code, log, &jit_disk_hit);
    if (jrc != 0) {
      code.clear();
      if (f32 || pack2) { qpc_set_error("JIT specialisation of the fp32 / packed tick failed (there is no ahead-of-time form of it): " + log); return QPC_ERR_HIP; }
      const char *strict = getenv("QPC_REQUIRE_JIT");
      if (strict && strict[0] == '1') { qpc_set_error("JIT specialisation failed: " + log); return QPC_ERR_HIP; }
      if (D.maxdepth > QPC_GENERIC_MAXD) {
        qpc_set_error("JIT specialisation failed and the tree is deeper than the generic kernel supports: " + log);
        return QPC_ERR_UNSUPPORTED;
      }
      fprintf(stderr, "[qpc] JIT specialisation unavailable (%s); using the generic gfx950 kernel\n", log.c_str());
    }
  }
  QpcRuntime *rt = new QpcRuntime();
  rt->slots = c->pipe_slots > 0 ? c->pipe_slots : 6; // measured 4: 13.0, 6: 15.6, 8: 15.6, 12: 14.8 M solves/s (pageable, 32768 QPs)
  if (const char *es = getenv("QPC_HOST_SLOTS")) { const int v = atoi(es); if (v >= 1 && v <= 8) rt->slots = v; }
  auto fail = [&](int code_) { for (auto *S : rt->devs) destroy_device_state(S); delete rt; return code_; };
  for (int i = 0; i < n_devices; i++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ids[i]) != hipSuccess) { qpc_set_error("hipGetDeviceProperties failed"); return fail(QPC_ERR_HIP); }
    if (std::string(prop.gcnArchName) != std::string(prop0.gcnArchName)) { qpc_set_error("qpc_finalize_devices: devices of different architectures"); return fail(QPC_ERR_UNSUPPORTED); }
    QpcDeviceState *S = new QpcDeviceState();
    rt->devs.push_back(S); // owned by rt from here on: released by fail() on any error below
    rc = init_device_state(c, S, device_ids[i], rt->slots, code, prop);
    if (rc != QPC_OK && rc == QPC_ERR_HIP && !code.empty() && !jit_disk_hit.empty() && !S->jit_func) {
      // a code object from the DISK cache that does not load (another ROCm, a damaged file): the file goes, the kernel is compiled
      // afresh, once -- instead of every start of the process falling to the generic kernel (or, fp32 / packed, failing) for good
      fprintf(stderr, "[qpc] cached code object %s unusable (%s); recompiling\n", jit_disk_hit.c_str(), qpc_last_error());
      qpc_jit_forget(jit_disk_hit);
      jit_disk_hit.clear();
      std::string log;
      for (auto *T : rt->devs) destroy_device_state(T);
      rt->devs.clear(); i = -1;
      if (qpc_jit_compile(D, c->prog.dims_s, prop0.gcnArchName, c->warm_start != 0, f32, pack2, code, log) != 0) code.clear();
      if (code.empty() && (f32 || pack2)) { qpc_set_error("JIT specialisation of the fp32 / packed tick failed: " + log); return fail(QPC_ERR_HIP); }
      continue;
    }
    if (rc != QPC_OK) {
      const char *strict = getenv("QPC_REQUIRE_JIT");
      if (!code.empty() && rc == QPC_ERR_HIP && !(strict && strict[0] == '1') && D.maxdepth <= QPC_GENERIC_MAXD && !S->jit_func && !f32 && !pack2) {
        // code object would not load: fall back to the generic gfx950 kernel on every device
        fprintf(stderr, "[qpc] JIT code object unusable (%s); using the generic gfx950 kernel\n", qpc_last_error());
        for (auto *T : rt->devs) destroy_device_state(T);
        rt->devs.clear(); code.clear(); i = -1;
        continue;
      }
      return fail(rc);
    }
  }
  HIP_TRY(hipSetDevice(device_ids[0]));
  c->rt = rt;
  rt->backend = new QpcHipBackend(c);
  int64_t chunk = c->pipe_chunk > 0 ? c->pipe_chunk : 4096, inl = c->pipe_inline >= 0 ? c->pipe_inline : 2048;
  if (const char *ec = getenv("QPC_HOST_CHUNK")) { const long long v = atoll(ec); if (v >= 1) chunk = v; }
  rt->pipe = new QpcHostPipeline(rt->backend, rt->slots, chunk, inl);
  if (const char *ez = getenv("QPC_ZERO_COPY")) rt->zero_copy = ez[0] != '0';
  if (const char *ez = QPC_EXP_ENV("QPC_STAGE_ZEROCOPY")) rt->stage_zero_copy = ez[0] != '0';
  c->finalized = true;
  if (out) *out = c->layout;
  return QPC_OK;
}

int32_t qpc_finalize(qpc_controller *c, int32_t device_id, qpc_layout *out) {
  return qpc_finalize_devices(c, 1, &device_id, out);
}

int32_t qpc_n_devices(const qpc_controller *c) { return (c && c->rt) ? (int32_t)c->rt->devs.size() : 0; }

int32_t qpc_set_host_pipeline(qpc_controller *c, int64_t chunk_qps, int32_t slots_per_device, int64_t inline_qps) {
  if (!c) return QPC_ERR_INVALID_ARGUMENT;
  if (slots_per_device > 0) {
    if (c->finalized) { qpc_set_error("qpc_set_host_pipeline: the number of slots is fixed by qpc_finalize"); return QPC_ERR_ALREADY_FINALIZED; }
    if (slots_per_device > 8) { qpc_set_error("qpc_set_host_pipeline: at most 8 slots per device (more persistent grids in flight starve each other)"); return QPC_ERR_INVALID_ARGUMENT; }
    c->pipe_slots = slots_per_device;
  }
  if (chunk_qps > 0) c->pipe_chunk = chunk_qps;
  if (inline_qps >= 0) c->pipe_inline = inline_qps;
  if (c->rt && c->rt->pipe) {
    std::string e;
    (void)c->rt->pipe->wait(e);
    if (chunk_qps > 0) c->rt->pipe->set_chunk_qps(chunk_qps);
    if (inline_qps >= 0) c->rt->pipe->set_inline_qps(inline_qps);
  }
  return QPC_OK;
}

void *qpc_host_alloc(uint64_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? (size_t)bytes : 1, hipHostMallocDefault | hipHostMallocPortable) != hipSuccess) {
    hip_fail(hipGetLastError(), "hipHostMalloc");
    return nullptr;
  }
  return p;
}
void qpc_host_free(void *p) { if (p) (void)hipHostFree(p); }
int32_t qpc_host_register(void *p, uint64_t bytes) {
  if (!p || !bytes) return QPC_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipHostRegister(p, (size_t)bytes, hipHostRegisterDefault | hipHostRegisterPortable));
  return QPC_OK;
}
int32_t qpc_host_unregister(void *p) {
  if (!p) return QPC_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipHostUnregister(p));
  return QPC_OK;
}

void qpc_controller_destroy(qpc_controller *c) {
  if (!c) return;
  if (c->rt) {
    QpcRuntime *rt = c->rt;
    delete rt->pipe; // joins the workers after the job in flight (if any)
    delete rt->backend;
    for (auto *S : rt->devs) destroy_device_state(S);
    delete rt;
  }
  delete c;
}

static int control_device(qpc_controller *c, int device_index, int64_t B, const double *q, const double *v, const double *desired,
                          const double *cw, const double *cmax, const double *tw, const double *surf, double *tau, double *vdot,
                          double *cwrench, int32_t *status, int32_t *iters, unsigned long long *wsets, void *hip_stream,
                          const char *who, int slot = 0, const int32_t *order = nullptr) {
  if (!c || !c->finalized || !c->rt) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  if (device_index < 0 || device_index >= (int)c->rt->devs.size()) { qpc_set_error(std::string(who) + ": device index out of range"); return QPC_ERR_INVALID_ARGUMENT; }
  if (slot < 0 || slot >= QPC_USER_SLOTS) { qpc_set_error(std::string(who) + ": launch slot out of range"); return QPC_ERR_INVALID_ARGUMENT; }
  const QpcDims &D = c->prog.dims;
  if (B < 0 || !q || !v || !tau || !vdot || !status || (D.n_des_inputs && !desired) || (D.nc && (!cw || !cmax)) ||
      (D.n_tw && !tw) || (D.n_surf && !surf)) {
    qpc_set_error(std::string(who) + ": missing array"); return QPC_ERR_INVALID_ARGUMENT;
  }
  QpcDeviceState *S = c->rt->devs[device_index];
  QpcBatchArgs a;
  a.B = B; a.q = q; a.v = v; a.desired = desired; a.cw = cw; a.cmax = cmax; a.tw = tw; a.surf = surf;
  a.tau = tau; a.vdot = vdot; a.cwrench = cwrench; a.status = status; a.iters = iters;
  a.prof = slot == 0 ? S->prof : nullptr; // the (diagnostic) section-profile buffer is indexed by QP: overlapping launches on other slots must not write it
  a.wsets = wsets;
  a.order = order;
  HIP_TRY(hipSetDevice(S->device));
  return launch(c, S, S->slots[c->rt->user_slot(slot)], a, (hipStream_t)hip_stream, wsets != nullptr);
}

int32_t qpc_control_batch_device(qpc_controller *c, int64_t B, const double *q, const double *v,
                                 const double *desired, const double *cw, const double *cmax, const double *tw,
                                 const double *surf, double *tau, double *vdot, double *cwrench, int32_t *status,
                                 int32_t *iters, void *hip_stream) {
  return control_device(c, 0, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr, hip_stream,
                        "qpc_control_batch_device");
}

int32_t qpc_control_batch_device_on(qpc_controller *c, int32_t device_index, int64_t B, const double *q, const double *v,
                                    const double *desired, const double *cw, const double *cmax, const double *tw,
                                    const double *surf, double *tau, double *vdot, double *cwrench, int32_t *status,
                                    int32_t *iters, void *hip_stream) {
  return control_device(c, device_index, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr,
                        hip_stream, "qpc_control_batch_device_on");
}

int32_t qpc_control_batch_device_slot(qpc_controller *c, int32_t device_index, int32_t slot, int64_t B, const double *q, const double *v,
                                      const double *desired, const double *cw, const double *cmax, const double *tw,
                                      const double *surf, double *tau, double *vdot, double *cwrench, int32_t *status,
                                      int32_t *iters, void *hip_stream) {
  return control_device(c, device_index, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr,
                        hip_stream, "qpc_control_batch_device_slot", slot);
}

int32_t qpc_control_batch_device_ordered(qpc_controller *c, int32_t device_index, int32_t slot, int64_t B, const double *q, const double *v,
                                         const double *desired, const double *cw, const double *cmax, const double *tw,
                                         const double *surf, double *tau, double *vdot, double *cwrench, int32_t *status,
                                         int32_t *iters, const int32_t *order, void *hip_stream) {
  return control_device(c, device_index, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr,
                        hip_stream, "qpc_control_batch_device_ordered", slot, order);
}

// order[0 .. B-1] = the QP indices sorted by descending key (counting sort on 256 buckets, one workgroup): key = cost[b] % modulus
// (modulus > 0) or cost[b], clamped to 0 .. 255
extern "C" __global__ void __launch_bounds__(1024) qpc_ticket_order_kernel(long long B, const int32_t *cost, int modulus, int32_t *order) {
  __shared__ unsigned hist[256], start[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  auto key = [&](long long b) { int k = cost[b]; if (modulus > 0) k %= modulus; return k < 0 ? 0 : (k > 255 ? 255 : k); };
  for (long long b = threadIdx.x; b < B; b += blockDim.x) atomicAdd(&hist[key(b)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) { unsigned o = 0; for (int k = 255; k >= 0; k--) { start[k] = o; o += hist[k]; } }
  __syncthreads();
  for (long long b = threadIdx.x; b < B; b += blockDim.x) order[atomicAdd(&start[key(b)], 1u)] = (int32_t)b;
}
int32_t qpc_ticket_order_device(qpc_controller *c, int32_t device_index, int64_t B, const int32_t *cost, int32_t modulus, int32_t *order,
                                void *hip_stream) {
  if (!c || !c->finalized || !c->rt) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  if (device_index < 0 || device_index >= (int)c->rt->devs.size() || B < 0 || !cost || !order) { qpc_set_error("qpc_ticket_order_device: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  if (B == 0) return QPC_OK;
  HIP_TRY(hipSetDevice(c->rt->devs[device_index]->device));
  hipLaunchKernelGGL(qpc_ticket_order_kernel, dim3(1), dim3(1024), 0, (hipStream_t)hip_stream, (long long)B, cost, (int)modulus, order);
  HIP_TRY(hipGetLastError());
  return QPC_OK;
}

// contactwrenches[BodyID(body)] (momentum.jl:64-72): the per-body sums of the per-contact wrenches; a thread per (QP, body, component),
// the contacts of a body added in contact order.  of[k] (<= 64 contacts: by value) = index of contact k's body
struct QpcBodyOf { int nc, nb; unsigned char of[64]; };
extern "C" __global__ void __launch_bounds__(256) qpc_body_wrench_kernel(long long B, QpcBodyOf M, const double *cw, double *bw) {
  const long long tot = B * 6 * M.nb;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / (6 * M.nb);
    const int r = (int)(e % (6 * M.nb)), body = r / 6, comp = r % 6;
    double s = 0.0;
    for (int k = 0; k < M.nc; k++) if (M.of[k] == body) s += cw[(b * M.nc + k) * 6 + comp];
    bw[e] = s;
  }
}
int32_t qpc_contact_wrench_per_body_device(qpc_controller *c, int32_t device_index, int64_t B, const double *cw, double *bw, void *hip_stream) {
  if (!c || !c->finalized || !c->rt) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  if (device_index < 0 || device_index >= (int)c->rt->devs.size() || B < 0 || (B > 0 && (!cw || !bw))) { qpc_set_error("qpc_contact_wrench_per_body_device: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  QpcBodyOf M;
  M.nc = (int)c->contacts.size();
  if (M.nc > 64) { qpc_set_error("qpc_contact_wrench_per_body_device: more than 64 contact points"); return QPC_ERR_UNSUPPORTED; }
  int32_t nb = 0, of[64];
  qpc_contact_bodies(c, &nb, nullptr, of);
  M.nb = nb;
  for (int k = 0; k < M.nc; k++) M.of[k] = (unsigned char)of[k];
  if (B == 0 || nb == 0) return QPC_OK;
  HIP_TRY(hipSetDevice(c->rt->devs[device_index]->device));
  const long long tot = B * 6 * nb;
  hipLaunchKernelGGL(qpc_body_wrench_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)hip_stream, (long long)B, M, cw, bw);
  HIP_TRY(hipGetLastError());
  return QPC_OK;
}

int32_t qpc_control_batch_device_warm(qpc_controller *c, int32_t device_index, int64_t B, const double *q, const double *v,
                                      const double *desired, const double *cw, const double *cmax, const double *tw,
                                      const double *surf, double *tau, double *vdot, double *cwrench, int32_t *status,
                                      int32_t *iters, uint64_t *working_set, void *hip_stream) {
  if (!working_set) { qpc_set_error("qpc_control_batch_device_warm: working_set missing"); return QPC_ERR_INVALID_ARGUMENT; }
  return control_device(c, device_index, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters,
                        (unsigned long long *)working_set, hip_stream, "qpc_control_batch_device_warm");
}

static int host_wait(qpc_controller *c) { // everything the host-pointer entry points may still have in flight
  std::string err;
  int rc = c->rt->pipe->wait(err);
  if (rc != QPC_OK) qpc_set_error(err);
  const int rz = zero_copy_wait(c);
  if (rc == QPC_OK) rc = rz;
  return rc;
}
// ... on behalf of a call that is NOT qpc_wait: the error of a job in flight is that job's, not the new call's (qpc.h) -- it is
// parked where qpc_last_async_error finds it instead of being lost, and announced once on stderr
static void host_wait_drop(qpc_controller *c) {
  const int rc = host_wait(c);
  if (rc != QPC_OK) c->rt->pipe->note_dropped(rc, qpc_last_error());
}

// run a filled job: zero-copy when every array is page-locked, else the staged pipeline
static int run_job(qpc_controller *c, const QpcHostJob &job, bool async) {
  // one job in flight per controller.  The error of an async job that was never waited for was THAT job's: it is
  // dropped here (qpc.h), the new job runs and reports its own outcome -- a caller must never get an error code for
  // a batch that was not even attempted
  host_wait_drop(c);
  int rc = QPC_OK;
  if (c->rt->zero_copy && job_all_pinned(job)) {
    rc = zero_copy_launch(c, job);
    if (rc == QPC_OK) return async ? QPC_OK : zero_copy_wait(c);
    if (rc > 0) { (void)zero_copy_wait(c); return rc; }
    // rc < 0: a device view of some array could not be obtained -> staged path
  }
  if (async) { c->rt->pipe->submit(job); return QPC_OK; }
  std::string err;
  rc = c->rt->pipe->run(job, err);
  if (rc != QPC_OK) qpc_set_error(err);
  return rc;
}

int32_t qpc_control_batch(qpc_controller *c, int64_t B, const double *q, const double *v, const double *desired,
                          const double *cw, const double *cmax, const double *tw, const double *surf, double *tau,
                          double *vdot, double *cwrench, int32_t *status, int32_t *iters) {
  QpcHostJob job;
  int rc = fill_job(c, job, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr, "qpc_control_batch");
  if (rc != QPC_OK) return rc;
  if (B == 0) { host_wait_drop(c); return QPC_OK; }
  return run_job(c, job, false);
}

int32_t qpc_control_batch_async(qpc_controller *c, int64_t B, const double *q, const double *v, const double *desired,
                                const double *cw, const double *cmax, const double *tw, const double *surf, double *tau,
                                double *vdot, double *cwrench, int32_t *status, int32_t *iters) {
  QpcHostJob job;
  int rc = fill_job(c, job, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr, "qpc_control_batch_async");
  if (rc != QPC_OK) return rc;
  if (B == 0) return QPC_OK;
  return run_job(c, job, true);
}

int32_t qpc_control_batch_warm(qpc_controller *c, int64_t B, const double *q, const double *v, const double *desired,
                               const double *cw, const double *cmax, const double *tw, const double *surf, double *tau,
                               double *vdot, double *cwrench, int32_t *status, int32_t *iters, uint64_t *working_set) {
  if (!working_set) { qpc_set_error("qpc_control_batch_warm: working_set missing"); return QPC_ERR_INVALID_ARGUMENT; }
  if (c && !c->warm_start) { qpc_set_error("qpc_control_batch_warm: call qpc_enable_warm_start before qpc_finalize"); return QPC_ERR_UNSUPPORTED; }
  QpcHostJob job;
  int rc = fill_job(c, job, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, working_set, "qpc_control_batch_warm");
  if (rc != QPC_OK) return rc;
  if (B == 0) { host_wait_drop(c); return QPC_OK; }
  return run_job(c, job, false);
}

int32_t qpc_wait(qpc_controller *c) {
  if (!c || !c->finalized || !c->rt) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  return host_wait(c);
}

int32_t qpc_last_async_error(qpc_controller *c, char *message, int32_t message_bytes) {
  if (!c || !c->finalized || !c->rt) return QPC_ERR_NOT_FINALIZED;
  std::string msg;
  const int rc = c->rt->pipe->take_dropped(msg);
  if (message && message_bytes > 0) snprintf(message, (size_t)message_bytes, "%s", msg.c_str());
  return rc;
}

int32_t qpc_assemble_solve_batch(qpc_controller *c, int64_t B, const double *task_J, const double *task_bias,
                                 const double *desired, const double *momentum_A, const double *momentum_rhs,
                                 const double *contact_R, const double *contact_p, const double *cw, const double *cmax,
                                 const double *tw, const double *surf, double *vdot, double *cwrench, int32_t *status,
                                 int32_t *iters) {
  if (!c || !c->finalized || !c->rt) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  const int nv = c->model->nv, nd = c->n_desired, nc = (int)c->contacts.size(), fl = c->floating >= 0;
  if (B < 0 || !vdot || !status || (nd && (!task_J || !task_bias || !desired)) || (fl && (!momentum_A || !momentum_rhs)) ||
      (nc && (!contact_R || !contact_p || !cw || !cmax)) || (c->n_tw && !tw) || (c->n_surf && !surf)) {
    qpc_set_error("qpc_assemble_solve_batch: missing array"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return QPC_OK;
  if (c->N > 8) { qpc_set_error("qpc_assemble_solve_batch: friction basis larger than 8"); return QPC_ERR_UNSUPPORTED; }
  for (auto &t : c->tasks)
    if (t.law == QPC_LAW_SE3_PD) { qpc_set_error("qpc_assemble_solve_batch: tasks with a trajectory-record law are not supported by the L1 path"); return QPC_ERR_UNSUPPORTED; }
  host_wait_drop(c); // the inline slot's stream is shared with small host-pointer jobs (a stale async error is that job's)
  int rc = QPC_OK;
  QpcDeviceState *S = c->rt->devs[0];
  hipStream_t st = S->slots[c->rt->inline_slot()].stream;
  HIP_TRY(hipSetDevice(S->device));
  std::vector<int> it;
  std::vector<double> dt;
  int n = 0, m = 0;
  qpc_l1_build_tables(c, it, dt, n, m);
  const long long wsd = qpc::l1_workspace_doubles(n, m);
  // one persistent device slab: [tables | inputs | workspace | outputs]; every sub-buffer is 64-byte aligned and the
  // total is computed with exactly the rounding take() applies below
  const size_t nin[11] = {(size_t)nd * nv, (size_t)nd, (size_t)nd, fl ? (size_t)6 * nv : 0, fl ? (size_t)6 : 0, (size_t)9 * nc,
                          (size_t)3 * nc, (size_t)nc, (size_t)nc, (size_t)c->n_tw, (size_t)c->n_surf};
  const double *hin[11] = {task_J, task_bias, desired, momentum_A, momentum_rhs, contact_R, contact_p, cw, cmax, tw, surf};
  const size_t ncw = 6 * (size_t)(nc > 0 ? nc : 1);
  size_t total = pad64(dt.size() * sizeof(double)) + pad64(it.size() * sizeof(int));
  for (int i = 0; i < 11; i++) total += pad64(nin[i] * (size_t)B * sizeof(double));
  total += pad64((size_t)wsd * B * sizeof(double)) + pad64((size_t)nv * B * sizeof(double)) + pad64(ncw * B * sizeof(double)) +
           2 * pad64((size_t)B * sizeof(int32_t));
  if (total > S->l1_bytes) {
    if (S->l1_buf) (void)hipFree(S->l1_buf);
    S->l1_buf = nullptr; S->l1_bytes = 0;
    if (hipMalloc(&S->l1_buf, total) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("qpc_assemble_solve_batch: out of device memory"); return QPC_ERR_OUT_OF_MEMORY; }
    S->l1_bytes = total;
  }
  char *buf = S->l1_buf;
  rc = QPC_OK;
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  char *p = buf;
  auto take = [&](size_t bytes) { char *r = p; p += pad64(bytes); return r; };
  QpcL1Args a;
  a.B = B; a.nv = nv; a.nd = nd; a.nc = nc; a.ntw = c->n_tw; a.nsurf = c->n_surf; a.ws_stride = wsd;
  double *ddt = (double *)take(dt.size() * sizeof(double));
  int *dit = (int *)take(it.size() * sizeof(int));
  chk(hipMemcpyAsync(ddt, dt.data(), dt.size() * sizeof(double), hipMemcpyHostToDevice, st), "copy tables");
  chk(hipMemcpyAsync(dit, it.data(), it.size() * sizeof(int), hipMemcpyHostToDevice, st), "copy tables");
  a.it = dit; a.dt = ddt;
  const double *din[11];
  for (int i = 0; i < 11; i++) {
    const size_t bytes = nin[i] * (size_t)B * sizeof(double);
    din[i] = (const double *)take(bytes);
    if (bytes) chk(hipMemcpyAsync((void *)din[i], hin[i], bytes, hipMemcpyHostToDevice, st), "copy inputs");
  }
  a.J = din[0]; a.bias = din[1]; a.desired = din[2]; a.Am = fl ? din[3] : nullptr; a.mrhs = fl ? din[4] : nullptr;
  a.cR = din[5]; a.cp = din[6]; a.cw = din[7]; a.cmax = din[8]; a.tw = din[9]; a.surf = din[10];
  a.ws = (double *)take((size_t)wsd * B * sizeof(double));
  a.vdot = (double *)take((size_t)nv * B * sizeof(double));
  a.cwrench = (double *)take(ncw * B * sizeof(double));
  a.status = (int32_t *)take((size_t)B * sizeof(int32_t));
  a.iters = (int32_t *)take((size_t)B * sizeof(int32_t));
  if ((size_t)(p - buf) > total) { qpc_set_error("qpc_assemble_solve_batch: internal slab sizing error"); return QPC_ERR_HIP; }
  if (rc == QPC_OK) {
    hipLaunchKernelGGL(qpc_l1_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, a, c->settings);
    chk(hipGetLastError(), "launch qpc_l1_kernel");
    chk(hipMemcpyAsync(vdot, a.vdot, (size_t)nv * B * sizeof(double), hipMemcpyDeviceToHost, st), "copy vdot");
    if (cwrench && nc) chk(hipMemcpyAsync(cwrench, a.cwrench, (size_t)6 * nc * B * sizeof(double), hipMemcpyDeviceToHost, st), "copy wrenches");
    chk(hipMemcpyAsync(status, a.status, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, st), "copy status");
    if (iters) chk(hipMemcpyAsync(iters, a.iters, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, st), "copy iters");
  }
  chk(hipStreamSynchronize(st), "qpc_l1_kernel"); // also on error: `it` / `dt` must outlive the async table copies
  return rc;
}

static int sim_check(qpc_controller *c, int64_t B, const void *q, const void *v, const void *tau, const qpc_sim_params *p, const char *who) {
  if (!c || !c->finalized || !c->rt) { qpc_set_error("controller not finalized"); return QPC_ERR_NOT_FINALIZED; }
  if (B < 0 || !q || !v || !tau || !p) { qpc_set_error(std::string(who) + ": missing array"); return QPC_ERR_INVALID_ARGUMENT; }
  if (c->precision == QPC_PRECISION_F32) { // (the fp32 module carries no simulation kernel: the fp32 tick would be paired with the generic fp64 step)
    qpc_set_error(std::string(who) + ": the closed-loop entry points are fp64-only (the controller was finalized with QPC_PRECISION_F32)"); return QPC_ERR_UNSUPPORTED;
  }
  if (!c->rt->devs[0]->jit_sim_func && c->prog.dims.maxdepth > QPC_GENERIC_MAXD) {
    qpc_set_error(std::string(who) + ": tree deeper than the generic kernel supports and no JIT kernel"); return QPC_ERR_UNSUPPORTED;
  }
  if (!(p->dt > 0.0) || !(p->stiffness >= 0.0) || !(p->damping >= 0.0) || !(p->tangential_damping >= 0.0)) {
    qpc_set_error(std::string(who) + ": dt must be positive and the ground gains non-negative"); return QPC_ERR_INVALID_ARGUMENT;
  }
  return QPC_OK;
}

static int sim_step_device(qpc_controller *c, int di, int64_t B, double *q, double *v, const double *tau, const int32_t *status,
                           const qpc_sim_params *params, int32_t *bad, hipStream_t stream) {
  QpcDeviceState *S = c->rt->devs[di];
  HIP_TRY(hipSetDevice(S->device));
  QpcSimArgs a;
  a.B = B; a.q = q; a.v = v; a.tau = tau; a.status = status; a.bad = bad; a.sp = *params;
  return launch_sim(c, S, a, stream);
}

int32_t qpc_sim_step_batch_device(qpc_controller *c, int64_t B, double *q, double *v, const double *tau, const int32_t *status,
                                  const qpc_sim_params *params, int32_t *bad, void *hip_stream) {
  int rc = sim_check(c, B, q, v, tau, params, "qpc_sim_step_batch_device");
  if (rc != QPC_OK) return rc;
  return sim_step_device(c, 0, B, q, v, tau, status, params, bad, (hipStream_t)hip_stream);
}

// n_steps x (tick -> step) on device `di`, stream `stream`; wsets != nullptr: warm-started ticks (working sets carried
// from step to step in `wsets`, 2 x uint64 per robot, zero-initialised by the caller = cold first tick)
static int rollout_device(qpc_controller *c, int di, int64_t B, int64_t n_steps, double *q, double *v, const double *desired,
                          const double *cw, const double *cmax, const double *tw, const double *surf, const qpc_sim_params *params,
                          double *tau, double *vdot, int32_t *status, int32_t *bad, unsigned long long *wsets, hipStream_t stream) {
  // two kernels per step on one stream, nothing else: the tick's work counter re-arms itself on the device
  for (int64_t s = 0; s < n_steps; s++) {
    int rc = control_device(c, di, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, nullptr, status, nullptr, wsets, stream,
                            "qpc_rollout_batch_device");
    if (rc != QPC_OK) return rc;
    rc = sim_step_device(c, di, B, q, v, tau, status, params, bad, stream);
    if (rc != QPC_OK) return rc;
  }
  return QPC_OK;
}

int32_t qpc_rollout_batch_device(qpc_controller *c, int64_t B, int64_t n_steps, double *q, double *v, const double *desired,
                                 const double *cw, const double *cmax, const double *tw, const double *surf,
                                 const qpc_sim_params *params, double *tau, double *vdot, int32_t *status, int32_t *bad,
                                 void *hip_stream) {
  int rc = sim_check(c, B, q, v, tau, params, "qpc_rollout_batch_device");
  if (rc != QPC_OK) return rc;
  if (n_steps < 0 || !vdot || !status) { qpc_set_error("qpc_rollout_batch_device: missing array"); return QPC_ERR_INVALID_ARGUMENT; }
  unsigned long long *wsets = nullptr;
  if (c->warm_start && c->rt->devs[0]->jit_warm_func && n_steps > 0) { // warm-started ticks: working sets in a persistent device buffer
    QpcDeviceState *S = c->rt->devs[0];
    HIP_TRY(hipSetDevice(S->device));
    const size_t need = 2 * (size_t)B * sizeof(unsigned long long);
    if (need > S->roll_ws_bytes) {
      if (S->roll_ws) HIP_TRY(hipFree(S->roll_ws));
      S->roll_ws = nullptr; S->roll_ws_bytes = 0;
      HIP_TRY(hipMalloc(&S->roll_ws, need));
      S->roll_ws_bytes = need;
    }
    HIP_TRY(hipMemsetAsync(S->roll_ws, 0, need, (hipStream_t)hip_stream));
    wsets = S->roll_ws;
  }
  return rollout_device(c, 0, B, n_steps, q, v, desired, cw, cmax, tw, surf, params, tau, vdot, status, bad, wsets, (hipStream_t)hip_stream);
}

// host-pointer variants: the batch is sharded over the controller's devices; per device one persistent slab, everything
// asynchronous on the device's inline-slot stream, one synchronisation per device at the end (the devices run
// concurrently).  tick = false: one simulation step with the caller's tau / status.
static int rollout_host(qpc_controller *c, int64_t B, int64_t n_steps, bool tick, double *q, double *v, const double *desired,
                        const double *cw, const double *cmax, const double *tw, const double *surf, const qpc_sim_params *params,
                        double *tau, int32_t *status, int32_t *bad) {
  const QpcDims &D = c->prog.dims;
  if (B == 0) return QPC_OK;
  host_wait_drop(c); // a stale error of an un-waited async job is that job's (qpc.h)
  int rc = QPC_OK;
  enum { A_Q, A_V, A_TAU, A_VDOT, A_DES, A_CW, A_CMAX, A_TW, A_SURF, A_COUNT };
  const size_t dim[A_COUNT] = {(size_t)D.nq, (size_t)D.nv, (size_t)D.nv, (size_t)D.nv, desired ? (size_t)D.n_desired : 0,
                               (size_t)D.nc, (size_t)D.nc, (size_t)D.n_tw, (size_t)D.n_surf};
  const double *host_in[A_COUNT] = {q, v, tick ? nullptr : tau, nullptr, desired, cw, cmax, tw, surf};
  const int G = (int)c->rt->devs.size();
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  struct Shard { int64_t lo, n; double *dev[A_COUNT]; int32_t *dstatus, *dbad; hipStream_t st; };
  std::vector<Shard> sh(G);
  const bool warm = tick && c->warm_start && c->rt->devs[0]->jit_warm_func;
  for (int g = 0; g < G && rc == QPC_OK; g++) {
    QpcDeviceState *S = c->rt->devs[g];
    int64_t lo, hi;
    qpc_shard_bounds(B, g, G, lo, hi);
    Shard &s = sh[g];
    s.lo = lo; s.n = hi - lo; s.st = S->slots[c->rt->inline_slot()].stream;
    if (s.n == 0) continue;
    chk(hipSetDevice(S->device), "hipSetDevice");
    auto padded = [&](int i) { return pad64(dim[i] * (size_t)s.n * sizeof(double)); };
    size_t total = 2 * pad64((size_t)s.n * sizeof(int32_t)) + pad64(warm ? 2 * (size_t)s.n * sizeof(unsigned long long) : 0);
    for (int i = 0; i < A_COUNT; i++) total += padded(i);
    if (total > S->roll_bytes) {
      if (S->roll_buf) (void)hipFree(S->roll_buf);
      S->roll_buf = nullptr; S->roll_bytes = 0;
      if (hipMalloc(&S->roll_buf, total) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("rollout: out of device memory"); rc = QPC_ERR_OUT_OF_MEMORY; break; }
      S->roll_bytes = total;
    }
    char *p = S->roll_buf;
    for (int i = 0; i < A_COUNT; i++) { s.dev[i] = (double *)p; p += padded(i); }
    s.dstatus = (int32_t *)p; p += pad64((size_t)s.n * sizeof(int32_t));
    s.dbad = (int32_t *)p; p += pad64((size_t)s.n * sizeof(int32_t));
    unsigned long long *dws = warm ? (unsigned long long *)p : nullptr;
    for (int i = 0; i < A_COUNT; i++)
      if (host_in[i] && dim[i])
        chk(hipMemcpyAsync(s.dev[i], host_in[i] + (size_t)lo * dim[i], dim[i] * (size_t)s.n * sizeof(double), hipMemcpyHostToDevice, s.st), "copy inputs");
    if (!tick && status) chk(hipMemcpyAsync(s.dstatus, status + lo, (size_t)s.n * sizeof(int32_t), hipMemcpyHostToDevice, s.st), "copy status");
    if (bad) chk(hipMemcpyAsync(s.dbad, bad + lo, (size_t)s.n * sizeof(int32_t), hipMemcpyHostToDevice, s.st), "copy bad");
    else chk(hipMemsetAsync(s.dbad, 0, (size_t)s.n * sizeof(int32_t), s.st), "clear bad");
    if (dws) chk(hipMemsetAsync(dws, 0, 2 * (size_t)s.n * sizeof(unsigned long long), s.st), "clear working sets");
    if (rc != QPC_OK) break;
    if (tick) {
      // the inline slot owns its stream and ticket counter: launch through it (not the user slot)
      for (int64_t k = 0; k < n_steps && rc == QPC_OK; k++) {
        QpcBatchArgs a;
        a.B = s.n; a.q = s.dev[A_Q]; a.v = s.dev[A_V]; a.desired = desired ? s.dev[A_DES] : nullptr; a.cw = s.dev[A_CW]; a.cmax = s.dev[A_CMAX];
        a.tw = s.dev[A_TW]; a.surf = s.dev[A_SURF]; a.tau = s.dev[A_TAU]; a.vdot = s.dev[A_VDOT]; a.cwrench = nullptr;
        a.status = s.dstatus; a.iters = nullptr; a.prof = nullptr; a.wsets = dws;
        rc = launch(c, S, S->slots[c->rt->inline_slot()], a, s.st, dws != nullptr);
        if (rc == QPC_OK) rc = sim_step_device(c, g, s.n, s.dev[A_Q], s.dev[A_V], s.dev[A_TAU], s.dstatus, params, s.dbad, s.st);
      }
    } else {
      rc = sim_step_device(c, g, s.n, s.dev[A_Q], s.dev[A_V], s.dev[A_TAU], status ? s.dstatus : nullptr, params, s.dbad, s.st);
    }
    if (rc == QPC_OK) {
      chk(hipMemcpyAsync(q + (size_t)lo * dim[A_Q], s.dev[A_Q], dim[A_Q] * (size_t)s.n * sizeof(double), hipMemcpyDeviceToHost, s.st), "copy q");
      chk(hipMemcpyAsync(v + (size_t)lo * dim[A_V], s.dev[A_V], dim[A_V] * (size_t)s.n * sizeof(double), hipMemcpyDeviceToHost, s.st), "copy v");
      if (tick && tau) chk(hipMemcpyAsync(tau + (size_t)lo * dim[A_TAU], s.dev[A_TAU], dim[A_TAU] * (size_t)s.n * sizeof(double), hipMemcpyDeviceToHost, s.st), "copy tau");
      if (tick && status) chk(hipMemcpyAsync(status + lo, s.dstatus, (size_t)s.n * sizeof(int32_t), hipMemcpyDeviceToHost, s.st), "copy status");
      if (bad) chk(hipMemcpyAsync(bad + lo, s.dbad, (size_t)s.n * sizeof(int32_t), hipMemcpyDeviceToHost, s.st), "copy bad");
    }
  }
  for (int g = 0; g < G; g++) {
    if (sh[g].n == 0 && rc == QPC_OK) continue;
    (void)hipSetDevice(c->rt->devs[g]->device);
    chk(hipStreamSynchronize(c->rt->devs[g]->slots[c->rt->inline_slot()].stream), "rollout");
  }
  (void)hipSetDevice(c->rt->devs[0]->device);
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

// diagnostic (QPC_STAMPS builds): device buffer of B*PS_COUNT uint64 section cycle counters (device 0, *_device entry)
int32_t qpc_debug_set_profile_buffer(qpc_controller *c, void *device_buffer) {
  if (!c || !c->rt) return QPC_ERR_NOT_FINALIZED;
  c->rt->devs[0]->prof = (unsigned long long *)device_buffer;
  return QPC_OK;
}

int32_t qpc_debug_jit_active(qpc_controller *c) { return (c && c->rt && c->rt->devs[0]->jit_func) ? 1 : 0; }
int32_t qpc_debug_warm_active(qpc_controller *c) { return (c && c->rt && c->rt->devs[0]->jit_warm_func) ? 1 : 0; }

int32_t qpc_timing_enable(qpc_controller *c, int32_t enable) {
  if (!c || !c->rt) return QPC_ERR_NOT_FINALIZED;
  for (auto *S : c->rt->devs) S->timing = enable != 0;
  return QPC_OK;
}

int32_t qpc_timing_read(qpc_controller *c, double *avg_ms, int64_t *n_launches) {
  if (!c || !c->rt) return QPC_ERR_NOT_FINALIZED;
  double sum = 0.0;
  long long n = 0;
  for (auto *S : c->rt->devs) {
    HIP_TRY(hipSetDevice(S->device));
    std::lock_guard<std::mutex> lk(S->timing_mu);
    for (auto &p : S->events) {
      HIP_TRY(hipEventSynchronize(p.second));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
      sum += ms; n++;
      S->pool.push_back(p.first); S->pool.push_back(p.second);
    }
    S->events.clear();
  }
  (void)hipSetDevice(c->rt->devs[0]->device);
  if (avg_ms) *avg_ms = n ? sum / n : 0.0;
  if (n_launches) *n_launches = n;
  return QPC_OK;
}

} // extern "C"
