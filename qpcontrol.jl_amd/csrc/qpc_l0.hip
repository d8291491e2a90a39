// qpc_l0.hip -- L0 of the C ABI (include/qpc.h): raw QPs in OSQP problem form, what backs the reference's own plug point -- the
// optimizer slot of `MomentumBasedController{N}(mechanism, optimizer)` (/root/reference/src/lowlevel/momentum.jl:15-16,27,58).
// The kernels (one lane per QP: qpc_qp_dense.h; a workgroup per QP, dense: qpc_qp_wg.h; a workgroup per QP along the elimination
// tree of a prepared problem: qpc_qp_sp.h) and the entry points qpc_qp_solve_batch[_device], qpc_qp_setup / qpc_qp_solve / qpc_qp_free.
// gfx950 only.  A translation unit of its own: these kernels are rebuilt without the tick.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "qpc_compile.h"
#include "qpc_env.h"
#include "qpc_device.h"
#include "qpc_qp_dense.h"
#include "qpc_qp_wg.h"
#include "qpc_qp_prep.h"
#include "qpc_qp_sym.h"
#include "qpc_qp_sp.h"

static int hip_fail(hipError_t e, const char *what) {
  qpc_set_error(std::string(what) + ": " + hipGetErrorString(e));
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? QPC_ERR_NO_DEVICE : QPC_ERR_HIP;
}
#define HIP_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return hip_fail(e__, #call); } while (0)
static inline size_t pad64(size_t b) { return (b + 63) & ~(size_t)63; }
#define QPC_L0_LEFT_CHUNK 2048 // QPs per launch of the dense forms behind qpc_qp_solve (their per-QP workspace and dense P, A are sized for this many)

// L0, first pass: one WORKGROUP per dense QP, working matrices in LDS (qpc_qp_wg.h): the problems Parametron builds for this
// controller (diagonal P, sparse A).  status[b] = QPC_WG_FALLBACK marks a QP it leaves to the one-lane kernel below.
extern "C" __global__ void __launch_bounds__(QPC_WG_THREADS)
qpc_qp_wg_kernel(int n, int m, long long B, const double *P, const double *q, const double *A, const double *l,
                 const double *u, qpc_settings st, double *ws, long long ws_stride, double *x, double *y,
                 int32_t *status, int32_t *iters, unsigned long long *ticket) {
  extern __shared__ __attribute__((aligned(16))) double qpc_wg_lds[];
  // QPs are handed out by a counter (zeroed by the host before the launch), not by block index: workgroups go to the eight XCDs
  // round-robin, so any cost pattern with a period of 8 in the batch (measured: the eight committed problems, cycled) would put all
  // the expensive QPs on ONE XCD -- a mixed batch of 2 048 took as long as 2 048 copies of its slowest member
  while (true) {
    if (threadIdx.x == 0) ((unsigned long long *)qpc_wg_lds)[15] = atomicAdd(ticket, 1ull);
    __syncthreads();
    const long long b = (long long)((unsigned long long *)qpc_wg_lds)[15];
    if (b >= B) break;
    const int s = qpc::qp_solve_wg(n, m, P + b * n * n, q + b * n, A + b * m * n, l + b * m, u + b * m, st,
                                   ws + b * ws_stride, qpc_wg_lds, QPC_WG_LDS_DOUBLES, x + b * n, y + b * m,
                                   iters ? iters + b : nullptr);
    if (threadIdx.x == 0) status[b] = s;
    __syncthreads();
#ifdef QPC_WG_PROF // phase stamps of the first QP (100 MHz ticks -> us): set-up, S build, elimination, sweeps, Goldfarb-Idnani, yE, refinement
    if (threadIdx.x == 0 && b == 0) {
      const unsigned long long *t = (const unsigned long long *)(ws + qpc::qp_dense_workspace_doubles(n, m) - 16);
      printf("wg_prof status %d us:", s);
      for (int k = 1; k <= 8; k++) printf(" %.1f", (double)(t[k] - t[k - 1]) * 0.01);
      printf(" total %.1f\n", (double)(t[8] - t[0]) * 0.01);
    }
#endif
  }
}

// L0 for a prepared problem (qpc_qp_setup): the pattern's tables instead of dense matrices; y0: warm-start multipliers or nullptr
extern "C" __global__ void __launch_bounds__(QPC_WG_THREADS)
qpc_qp_wg_prep_kernel(qpc::QpPrepView V, long long B, long long nnzP, const double *Pv, const double *q, const double *Av,
                      const double *l, const double *u, qpc_settings st, double *ws, long long ws_stride, const double *y0,
                      double *x, double *y, int32_t *status, int32_t *iters, unsigned long long *ticket, double *ykeep, int only_fallback) {
  extern __shared__ __attribute__((aligned(16))) double qpc_wg_lds[];
  while (true) { // (tickets: see qpc_qp_wg_kernel)
    if (threadIdx.x == 0) ((unsigned long long *)qpc_wg_lds)[15] = atomicAdd(ticket, 1ull);
    __syncthreads();
    const long long b = (long long)((unsigned long long *)qpc_wg_lds)[15];
    if (b >= B) {
      // a launch draws exactly B + gridDim.x tickets (every workgroup one past the end): whoever draws the last of them re-arms the
      // counter, so that the host neither clears it before a launch nor keeps track of it (the tick's scheme)
      if (threadIdx.x == 0 && b == B + (long long)gridDim.x - 1) atomicExch(ticket, 0ull);
      break;
    }
    if (only_fallback && status[b] != QPC_WG_FALLBACK) continue; // (the sparse form has solved it; uniform: every thread reads the same word)
#ifdef QPC_WG_PROF
    const long long c0__ = clock64(), w0__ = wall_clock64();
#endif
    const int s = qpc::qp_solve_wg_prep(V, Pv + b * nnzP, q + b * V.n, Av + b * V.nnzA, l + b * V.m, u + b * V.m, st,
                                        ws + b * ws_stride, qpc_wg_lds, QPC_WG_LDS_DOUBLES, y0 ? y0 + b * V.m : nullptr,
                                        x + b * V.n, y + b * V.m, iters ? iters + b : nullptr);
    if (threadIdx.x == 0) status[b] = s;
    __syncthreads();
    // the multipliers stay on the device for the next call's warm start (this QP's own slot: read at its start, written at its end)
    if (ykeep && s != QPC_WG_FALLBACK) for (int i = threadIdx.x; i < V.m; i += blockDim.x) ykeep[b * V.m + i] = y[b * V.m + i];
#ifdef QPC_WG_PROF
    if (threadIdx.x == 0 && b == 0) {
      const unsigned long long *t = (const unsigned long long *)(ws + qpc::qp_dense_workspace_doubles(V.n, V.m) - 16);
      printf("wg_prep_prof (shader clock %.0f MHz) status %d us:", (double)(clock64() - c0__) / ((double)(wall_clock64() - w0__) * 0.01), s);
      for (int k = 1; k <= 8; k++) printf(" %.1f", (double)(t[k] - t[k - 1]) * 0.01);
      printf(" total %.1f\n", (double)(t[8] - t[0]) * 0.01);
      if (blockIdx.x == 0) {
        printf("  elimination: big matrix panel+rows %.1f (cumulative incl. trailing %.1f) us; small triangles %.1f / %.1f us\n", (double)qpc::qpc_wg_acc[0] * 0.01,
               (double)qpc::qpc_wg_acc[1] * 0.01, (double)qpc::qpc_wg_acc[2] * 0.01, (double)qpc::qpc_wg_acc[3] * 0.01);
        printf("  wave 0 inside its tile loops %.1f us\n", (double)qpc::qpc_wg_acc[4] * 0.01);
        for (int k = 0; k < 8; k++) qpc::qpc_wg_acc[k] = 0;
      }
    }
#endif
  }
}
// L0 for a prepared problem with a PLAN (qpc_qp_sym.h, qpc_qp_sp.h): the sparse workgroup form.  The plan's tables go to LDS once per
// workgroup; QPs are drawn from the device counter like above.  A QP whose equality rows are not the plan's is left (QPC_WG_FALLBACK).
extern "C" __global__ void __launch_bounds__(QPC_SP_THREADS) __attribute__((amdgpu_waves_per_eu(1, 2))) // (the register-resident panels want the 256 registers of two wavefronts per SIMD)
qpc_qp_sp_kernel(const int *plan, int lds_doubles, long long B, long long nnzP, const double *Pv, const double *q, const double *Av,
                 const double *l, const double *u, qpc_settings st, const double *y0, double *x, double *y, int32_t *status,
                 int32_t *iters, unsigned long long *ticket, double *ykeep, unsigned long long *stamps) {
  extern __shared__ __attribute__((aligned(16))) double qpc_sp_lds[];
  __shared__ unsigned long long qpc_sp_ticket;
  const int n = plan[qpc::SP_N], m = plan[qpc::SP_M], nnzA = plan[qpc::SP_NNZA], nint = plan[qpc::SP_LDS_INTS];
  int *plan_lds = (int *)qpc_sp_lds;
  double *lds = qpc_sp_lds + (nint + 1) / 2;
  qpc::qp_sp_load_plan(plan, plan_lds);
  while (true) {
    if (threadIdx.x == 0) qpc_sp_ticket = atomicAdd(ticket, 1ull);
    __syncthreads();
    const long long b = (long long)qpc_sp_ticket;
    __syncthreads();
    if (b >= B) {
      if (threadIdx.x == 0 && b == B + (long long)gridDim.x - 1) atomicExch(ticket, 0ull); // (the launch's last ticket re-arms the counter)
      break;
    }
#ifdef QPC_SP_PROF
    const long long c0__ = clock64(), w0__ = wall_clock64();
#endif
    const int s = qpc::qp_solve_sp(plan, plan_lds, Pv + b * nnzP, q + b * n, Av + b * nnzA, l + b * m, u + b * m, st, lds,
                                   lds_doubles - (nint + 1) / 2, y0 ? y0 + b * m : nullptr, x + b * n, y + b * m,
                                   iters ? iters + b : nullptr, stamps);
    if (threadIdx.x == 0) status[b] = s;
    __syncthreads();
    if (ykeep && s != QPC_WG_FALLBACK) for (int i = threadIdx.x; i < m; i += blockDim.x) ykeep[b * m + i] = y[b * m + i];
#ifdef QPC_SP_PROF
    if (threadIdx.x == 0 && (b == 0 || (b == 1 && gridDim.x == 1)) && stamps) {
      // set-up (values, H^-1, xu) | G | fronts of level 0 | its urgent updates | fronts of level 1 beside the other updates | ... | iterations |
      // equality multipliers | refinement | output
      static const char *nm[13] = {"", "setup", "G", "F0", "U0u", "F1+U0", "U1u", "F2+U1", "rest", "iter", "yE", "refine", "out"};
      printf("sp_prof (shader clock %.0f MHz) status %d iters %d us:", (double)(clock64() - c0__) / ((double)(wall_clock64() - w0__) * 0.01), s, iters ? iters[b] : -1);
      for (int k = 1; k <= 12; k++) printf(" %s %.1f", nm[k], (double)(stamps[k] - stamps[k - 1]) * 0.01);
      printf(" | total %.1f\n", (double)(stamps[12] - stamps[0]) * 0.01);
      printf("   refinement: x %.1f, residuals %.1f, right-hand side %.1f, forward %.1f, working rows %.1f, back %.1f, update %.1f, residuals again %.1f us\n",
             (double)(stamps[15] - stamps[10]) * 0.01, (double)(stamps[16] - stamps[15]) * 0.01, (double)(stamps[17] - stamps[16]) * 0.01, (double)(stamps[18] - stamps[17]) * 0.01,
             (double)(stamps[19] - stamps[18]) * 0.01, (double)(stamps[20] - stamps[19]) * 0.01, (double)(stamps[21] - stamps[20]) * 0.01, (double)(stamps[22] - stamps[21]) * 0.01);
      printf("   working-set factorizations %llu (rows in all %llu): gather %.1f, panels %.1f, multipliers %.1f, A x %.1f us\n", qpc::qpc_sp_acc[8], qpc::qpc_sp_acc[9],
             (double)qpc::qpc_sp_acc[0] * 0.01, (double)qpc::qpc_sp_acc[1] * 0.01, (double)qpc::qpc_sp_acc[2] * 0.01, (double)qpc::qpc_sp_acc[3] * 0.01);
      for (int k = 0; k < 16; k++) qpc::qpc_sp_acc[k] = 0;
    }
#endif
  }
}
// the dense matrices of the prepared problem's QPs (for the dense path: a P with off-diagonal entries, or a QP the workgroup left)
// how many QPs of a batch a workgroup form has left to the next one (status == QPC_WG_FALLBACK): qpc_qp_solve_device decides on the host
// whether the next form runs at all
extern "C" __global__ void __launch_bounds__(256) qpc_qp_count_left_kernel(const int32_t *status, long long B, int *count) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long left = __ballot(b < B && status[b] == QPC_WG_FALLBACK);
  if ((threadIdx.x & 63) == 0 && left) atomicAdd(count, (int)__popcll(left));
}

extern "C" __global__ void qpc_qp_densify_kernel(int n, int m, long long B, long long nnzP, const int *Prow, const int *Pcol,
                                                 const double *Pv, long long nnzA, const int *Arow, const int *Acol,
                                                 const double *Av, double *P, double *A) {
  const long long per = nnzP + nnzA, tot = per * B;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / per, k = e % per;
    if (k < nnzP) {
      const double v = Pv[b * nnzP + k];
      P[b * n * n + (long long)Prow[k] * n + Pcol[k]] = v;
      P[b * n * n + (long long)Pcol[k] * n + Prow[k]] = v;
    } else {
      const long long ka = k - nnzP;
      A[b * m * n + (long long)Arow[ka] * n + Acol[ka]] = Av[b * nnzA + ka];
    }
  }
}

// L0: one lane per dense QP (the general method: any P, the Goldfarb-Idnani phase; see qpc_qp_dense.h).  only_fallback: just the
// QPs the workgroup kernel left (status QPC_WG_FALLBACK)
extern "C" __global__ void __launch_bounds__(64)
qpc_qp_dense_kernel(int n, int m, long long B, const double *P, const double *q, const double *A, const double *l,
                    const double *u, qpc_settings st, double *ws, long long ws_stride, double *x, double *y,
                    int32_t *status, int32_t *iters, int only_fallback) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (only_fallback && status[b] != QPC_WG_FALLBACK) return;
  qpc::qp_solve_dense(n, m, P + b * n * n, q + b * n, A + b * m * n, l + b * m, u + b * m, st, ws + b * ws_stride,
                      x + b * n, y + b * m, status + b, iters ? iters + b : nullptr);
}

extern "C" {

// the workgroup kernels of L0 ask for the whole LDS of a CU: the limit is a per-device function attribute, set once per device
static int qpc_wg_lds_attr(const void *kernel, int device_id, int lds_bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, hipError_t> done;
  std::lock_guard<std::mutex> lk(mu);
  auto key = std::make_pair(kernel, device_id);
  auto it = done.find(key);
  if (it == done.end()) it = done.emplace(key, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)).first;
  if (it->second != hipSuccess) return hip_fail(it->second, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  return QPC_OK;
}

// ---- L0: raw dense QPs.  Workspace per QP in HBM (qpc_qp_dense.h).
uint64_t qpc_qp_workspace_bytes(int32_t n, int32_t m, int64_t B) {
  if (n <= 0 || m < 0 || B < 0) return 0;
  return (uint64_t)qpc::qp_dense_workspace_doubles(n, m) * (uint64_t)B * sizeof(double);
}

int32_t qpc_qp_solve_batch_device(int32_t n, int32_t m, int64_t B, const double *P, const double *q, const double *A,
                                  const double *l, const double *u, const qpc_settings *settings, int32_t device_id,
                                  double *x, double *y, int32_t *status, int32_t *iters, void *workspace,
                                  uint64_t workspace_bytes, void *hip_stream) {
  if (n <= 0 || m < 0 || B < 0 || !P || !q || (m && (!A || !l || !u)) || !x || (m && !y) || !status || !workspace) {
    qpc_set_error("qpc_qp_solve_batch_device: bad arguments"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (workspace_bytes < qpc_qp_workspace_bytes(n, m, B)) { qpc_set_error("qpc_qp_solve_batch_device: workspace smaller than qpc_qp_workspace_bytes(n, m, B)"); return QPC_ERR_INVALID_ARGUMENT; }
  if (B == 0) return QPC_OK;
  HIP_TRY(hipSetDevice(device_id));
  qpc_settings st;
  if (settings) st = *settings; else qpc_default_settings(&st);
  // first the workgroup kernel (LDS-resident working set: the structured problems of this controller), then the one-lane kernel on
  // whatever it left; QPC_L0_ONE_LANE=1 (experiments build) keeps the old single pass
  int wg = m > 0 && n <= QPC_WG_DIM && m <= QPC_WG_DIM;
  if (const char *e = QPC_EXP_ENV("QPC_L0_ONE_LANE")) if (e[0] == '1') wg = 0;
  if (wg) {
    const int lds_bytes = qpc::qp_wg_lds_doubles(n, m) * (int)sizeof(double);
    if (int arc = qpc_wg_lds_attr((const void *)qpc_qp_wg_kernel, device_id, lds_bytes)) return arc;
    static std::mutex cu_mu;
    static std::map<int, int> cu_of; // (hipGetDeviceProperties costs tens of microseconds: once per device)
    int n_cu = 0;
    {
      std::lock_guard<std::mutex> lk(cu_mu);
      auto it = cu_of.find(device_id);
      if (it == cu_of.end()) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device_id));
        it = cu_of.emplace(device_id, prop.multiProcessorCount).first;
      }
      n_cu = it->second;
    }
    const long long grid = B < (long long)n_cu * 4 ? B : (long long)n_cu * 4;
    // the ticket counter: the last of the spare doubles at the end of the workspace
    unsigned long long *ticket = (unsigned long long *)((double *)workspace + (long long)qpc::qp_dense_workspace_doubles(n, m) * B - 1);
    HIP_TRY(hipMemsetAsync(ticket, 0, sizeof(unsigned long long), (hipStream_t)hip_stream));
    hipLaunchKernelGGL(qpc_qp_wg_kernel, dim3((unsigned)grid), dim3(QPC_WG_THREADS), (size_t)lds_bytes, (hipStream_t)hip_stream, (int)n, (int)m,
                       (long long)B, P, q, A, l, u, st, (double *)workspace, (long long)qpc::qp_dense_workspace_doubles(n, m), x, y,
                       status, iters, ticket);
    HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(qpc_qp_dense_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, (hipStream_t)hip_stream, (int)n, (int)m,
                     (long long)B, P, q, A, l, u, st, (double *)workspace, (long long)qpc::qp_dense_workspace_doubles(n, m), x, y,
                     status, iters, wg);
  HIP_TRY(hipGetLastError());
  return QPC_OK;
}

// persistent device slab + stream of the host-pointer L0 entry, one per device of the process
struct QpcL0Cache { char *buf = nullptr; size_t bytes = 0; hipStream_t stream = nullptr; char *h_stage = nullptr; };
static std::mutex g_l0_mu;
static std::map<int, QpcL0Cache> g_l0;

int32_t qpc_qp_solve_batch(int32_t n, int32_t m, int64_t B, const double *P, const double *q, const double *A,
                           const double *l, const double *u, const qpc_settings *settings, int32_t device_id,
                           double *x, double *y, int32_t *status, int32_t *iters) {
  if (n <= 0 || m < 0 || B < 0 || !P || !q || (m && (!A || !l || !u)) || !x || (m && !y) || !status) {
    qpc_set_error("qpc_qp_solve_batch: bad arguments"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return QPC_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    qpc_set_error("qpc_qp_solve_batch: no HIP device available (this library has no CPU fallback)");
    return QPC_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) { qpc_set_error("qpc_qp_solve_batch: device_id out of range"); return QPC_ERR_NO_DEVICE; }
  HIP_TRY(hipSetDevice(device_id));
  std::lock_guard<std::mutex> lk(g_l0_mu); // one L0 call per process at a time (plumbing path)
  QpcL0Cache &C = g_l0[device_id];
  if (!C.stream) HIP_TRY(hipStreamCreateWithFlags(&C.stream, hipStreamNonBlocking));
  const size_t nP = (size_t)B * n * n, nq = (size_t)B * n, nA = (size_t)B * m * n, nm = (size_t)B * m;
  const size_t wsb = (size_t)qpc_qp_workspace_bytes(n, m, B);
  const size_t total = pad64(nP * 8) + 2 * pad64(nq * 8) + pad64(nA * 8) + 3 * pad64(nm * 8) + pad64(wsb) + 2 * pad64((size_t)B * sizeof(int32_t));
  if (total > C.bytes) {
    if (C.buf) (void)hipFree(C.buf);
    C.buf = nullptr; C.bytes = 0;
    if (hipMalloc(&C.buf, total) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("qpc_qp_solve_batch: out of device memory"); return QPC_ERR_OUT_OF_MEMORY; }
    C.bytes = total;
  }
  char *p = C.buf;
  auto take = [&](size_t bytes) { char *r = p; p += pad64(bytes); return r; };
  // inputs side by side, outputs side by side: a SMALL batch (B = 1 is the reference's regime) moves in one copy each way through a
  // page-locked buffer instead of nine copies of pageable memory
  double *dP = (double *)take(nP * 8), *dq = (double *)take(nq * 8), *dA = (double *)take(nA * 8), *dl = (double *)take(nm * 8),
         *du = (double *)take(nm * 8);
  const size_t in_bytes = (size_t)(p - C.buf);
  char *const out0 = p;
  double *dx = (double *)take(nq * 8), *dy = (double *)take(nm * 8);
  int32_t *dst = (int32_t *)take((size_t)B * sizeof(int32_t)), *dit = (int32_t *)take((size_t)B * sizeof(int32_t));
  const size_t out_bytes = (size_t)(p - out0);
  double *dws = (double *)take(wsb);
  const size_t stage_cap = (size_t)2 << 20;
  if (in_bytes + out_bytes <= stage_cap && !C.h_stage) { if (hipHostMalloc(&C.h_stage, stage_cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); C.h_stage = nullptr; } }
  const bool staged = in_bytes + out_bytes <= stage_cap && C.h_stage != nullptr;
  int rc = QPC_OK;
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  if (staged) {
    char *h = C.h_stage;
    memcpy(h + ((char *)dP - C.buf), P, nP * 8);
    memcpy(h + ((char *)dq - C.buf), q, nq * 8);
    if (m) { memcpy(h + ((char *)dA - C.buf), A, nA * 8); memcpy(h + ((char *)dl - C.buf), l, nm * 8); memcpy(h + ((char *)du - C.buf), u, nm * 8); }
    chk(hipMemcpyAsync(C.buf, h, in_bytes, hipMemcpyHostToDevice, C.stream), "copy the inputs");
  } else {
    chk(hipMemcpyAsync(dP, P, nP * sizeof(double), hipMemcpyHostToDevice, C.stream), "copy P");
    chk(hipMemcpyAsync(dq, q, nq * sizeof(double), hipMemcpyHostToDevice, C.stream), "copy q");
    if (m) {
      chk(hipMemcpyAsync(dA, A, nA * sizeof(double), hipMemcpyHostToDevice, C.stream), "copy A");
      chk(hipMemcpyAsync(dl, l, nm * sizeof(double), hipMemcpyHostToDevice, C.stream), "copy l");
      chk(hipMemcpyAsync(du, u, nm * sizeof(double), hipMemcpyHostToDevice, C.stream), "copy u");
    }
  }
  if (rc == QPC_OK) rc = qpc_qp_solve_batch_device(n, m, B, dP, dq, dA, dl, du, settings, device_id, dx, dy, dst, dit, dws, wsb, C.stream);
  if (rc == QPC_OK && staged) {
    char *ho = C.h_stage + in_bytes;
    chk(hipMemcpyAsync(ho, out0, out_bytes, hipMemcpyDeviceToHost, C.stream), "copy the outputs");
    chk(hipStreamSynchronize(C.stream), "qpc_qp_dense_kernel");
    if (rc == QPC_OK) {
      memcpy(x, ho, nq * 8);
      if (m) memcpy(y, ho + ((char *)dy - out0), nm * 8);
      memcpy(status, ho + ((char *)dst - out0), (size_t)B * sizeof(int32_t));
      if (iters) memcpy(iters, ho + ((char *)dit - out0), (size_t)B * sizeof(int32_t));
    }
    return rc;
  }
  if (rc == QPC_OK) {
    chk(hipMemcpyAsync(x, dx, nq * sizeof(double), hipMemcpyDeviceToHost, C.stream), "copy x");
    if (m) chk(hipMemcpyAsync(y, dy, nm * sizeof(double), hipMemcpyDeviceToHost, C.stream), "copy y");
    chk(hipMemcpyAsync(status, dst, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, C.stream), "copy status");
    if (iters) chk(hipMemcpyAsync(iters, dit, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, C.stream), "copy iters");
  }
  chk(hipStreamSynchronize(C.stream), "qpc_qp_dense_kernel");
  return rc;
}

// ---- L0 with the optimizer's life cycle: pattern once (qpc_qp_setup), values every call (qpc_qp_solve), warm start from the previous
//      call's multipliers (kept on the device)
struct qpc_qp_problem {
  qpc::QpPrepHost H;
  int device = 0;
  hipStream_t stream = nullptr;
  int *d_ints = nullptr, *d_coo = nullptr; // the pattern's tables; the caller's COO pattern [Prow | Pcol | Arow | Acol]
  char *buf = nullptr;                     // per-call slab (values, outputs, workspace), grown on demand
  size_t bytes = 0;
  double *d_yprev = nullptr;               // multipliers of the previous call, B_prev x m
  int64_t B_prev = 0, cap_prev = 0;
  char *dense = nullptr;                   // dense P, A of the batch (only when the dense path is needed)
  size_t dense_bytes = 0;
  unsigned long long *d_ticket = nullptr;  // the launch's QP counter (zero between launches: the kernel re-arms it)
  char *h_stage = nullptr;                 // page-locked staging of a SMALL batch: one copy in, one copy out (B = 1 is the reference's regime)
  size_t h_stage_bytes = 0;
  int n_cu = 0;
  // the sparse workgroup form: the plan for one set of equality rows (made at the first solve, or by qpc_qp_analyze), on the device
  char *ws = nullptr;                      // workspace of the dense forms, grown on demand
  size_t ws_bytes = 0;
  qpc::QpSpPlanHost plan;
  bool plan_tried = false;
  int *d_plan = nullptr;
  size_t d_plan_ints = 0;
  int sp_lds_doubles = 0;                  // LDS the kernel is launched with (0: the plan does not fit)
  unsigned long long *d_stamps = nullptr;
  hipStream_t stream2 = nullptr;           // a LARGE batch goes in chunks over two streams: a chunk's copies beside the other's kernel
  unsigned long long *d_ticket2 = nullptr;
  int *d_left = nullptr, *h_left = nullptr; // qpc_qp_solve_device: how many QPs a workgroup form has left (device counter, page-locked copy)
  bool timing = false;                     // qpc_qp_enable_timing: events around the kernels of a solve
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_kernel_ms = 0.0;
};

// the rows of a QP that are equalities (l == u): the kernel's own test
static void qpc_eq_rows(int m, const double *l, const double *u, std::vector<uint8_t> &eq) {
  eq.resize((size_t)m);
  for (int i = 0; i < m; i++) eq[(size_t)i] = (l[i] > -1e19 && u[i] < 1e19 && u[i] - l[i] <= 1e-12 * (1.0 + fabs(l[i]))) ? 1 : 0;
}
// makes (or keeps) the plan for this set of equality rows and uploads it
static int qpc_qp_plan(qpc_qp_problem *p, const std::vector<uint8_t> &eq) {
  if (p->plan_tried && eq == p->plan.eq) return QPC_OK;
  p->plan_tried = true;
  p->sp_lds_doubles = 0;
  try {
    qpc::qp_sp_plan_build(p->H, eq.data(), p->plan);
  } catch (const std::bad_alloc &) {
    p->plan = qpc::QpSpPlanHost();
  }
  if (!p->plan.valid) { p->plan.eq = eq; return QPC_OK; } // (remembered: not tried again for the same rows)
  // LDS: the plan's tables and the working arrays, the working set's factor sized for as many inequality rows as fit
  // (the kernel also holds 8 bytes of STATIC LDS -- its ticket word -- which count against the same 160 KB: a plan that took the last
  //  byte made the dispatch fail with HSA_STATUS_ERROR_INVALID_ALLOCATION; found by tests/offline/fuzz_l0.py seed 7003 on the GPU)
  const int total = QPC_SP_LDS_CAP_DOUBLES;
  int kw = p->plan.mi;
  while (kw > 0 && qpc::qp_sp_lds_doubles(p->plan.ints.data(), kw) > total) kw--;
  const int need = qpc::qp_sp_lds_doubles(p->plan.ints.data(), kw);
  if (need > total || (kw < p->plan.mi && kw < 32)) return QPC_OK; // (does not fit: the dense workgroup form stays)
  if (p->plan.ints.size() > p->d_plan_ints) {
    if (p->d_plan) (void)hipFree(p->d_plan);
    p->d_plan = nullptr; p->d_plan_ints = 0;
    HIP_TRY(hipMalloc(&p->d_plan, p->plan.ints.size() * sizeof(int)));
    p->d_plan_ints = p->plan.ints.size();
  }
  HIP_TRY(hipMemcpyAsync(p->d_plan, p->plan.ints.data(), p->plan.ints.size() * sizeof(int), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->sp_lds_doubles = need;
  return QPC_OK;
}

int32_t qpc_qp_setup(int32_t n, int32_t m, int64_t nnz_P, const int32_t *P_rows, const int32_t *P_cols, int64_t nnz_A,
                     const int32_t *A_rows, const int32_t *A_cols, int32_t device_id, qpc_qp_problem **out) {
  if (!out || (nnz_P && (!P_rows || !P_cols)) || (nnz_A && (!A_rows || !A_cols))) { qpc_set_error("qpc_qp_setup: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  *out = nullptr;
  std::unique_ptr<qpc_qp_problem> p;
  try {
    p.reset(new qpc_qp_problem);
    const std::string err = qpc::qp_prep_build(p->H, n, m, nnz_P, P_rows, P_cols, nnz_A, A_rows, A_cols);
    if (!err.empty()) { qpc_set_error("qpc_qp_setup: " + err); return QPC_ERR_INVALID_ARGUMENT; }
  } catch (const std::bad_alloc &) { // (an exception must not cross the C boundary)
    qpc_set_error("qpc_qp_setup: out of host memory while analysing the pattern");
    return QPC_ERR_OUT_OF_MEMORY;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    qpc_set_error("qpc_qp_setup: no HIP device available (this library has no CPU fallback)");
    return QPC_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) { qpc_set_error("qpc_qp_setup: device_id out of range"); return QPC_ERR_NO_DEVICE; }
  p->device = device_id;
  auto setup = [&]() -> int { // (a failure half-way: qpc_qp_free below releases what was allocated)
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    p->n_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&p->d_ints, p->H.ints.size() * sizeof(int)));
    HIP_TRY(hipMemcpy(p->d_ints, p->H.ints.data(), p->H.ints.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<int> coo;
    coo.insert(coo.end(), p->H.Prow.begin(), p->H.Prow.end()); coo.insert(coo.end(), p->H.Pcol.begin(), p->H.Pcol.end());
    coo.insert(coo.end(), p->H.Arow.begin(), p->H.Arow.end()); coo.insert(coo.end(), p->H.Acol.begin(), p->H.Acol.end());
    coo.push_back(0);
    HIP_TRY(hipMalloc(&p->d_ticket, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(p->d_ticket, 0, sizeof(unsigned long long)));
    HIP_TRY(hipMalloc(&p->d_coo, coo.size() * sizeof(int)));
    HIP_TRY(hipMemcpy(p->d_coo, coo.data(), coo.size() * sizeof(int), hipMemcpyHostToDevice));
    return QPC_OK;
  };
  if (int rc = setup()) { qpc_qp_free(p.release()); return rc; }
  *out = p.release();
  return QPC_OK;
}

int32_t qpc_qp_analyze(qpc_qp_problem *p, const double *l, const double *u) {
  if (!p || (p->H.m > 0 && (!l || !u))) { qpc_set_error("qpc_qp_analyze: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  if (p->H.m <= 0) return QPC_OK;
  HIP_TRY(hipSetDevice(p->device));
  std::vector<uint8_t> eq;
  qpc_eq_rows(p->H.m, l, u, eq);
  return qpc_qp_plan(p, eq);
}

int32_t qpc_qp_enable_timing(qpc_qp_problem *p, int32_t on) {
  if (!p) { qpc_set_error("qpc_qp_enable_timing: null problem"); return QPC_ERR_INVALID_ARGUMENT; }
  HIP_TRY(hipSetDevice(p->device));
  if (on && !p->ev0) { HIP_TRY(hipEventCreate(&p->ev0)); HIP_TRY(hipEventCreate(&p->ev1)); }
  p->timing = on != 0;
  return QPC_OK;
}
int32_t qpc_qp_last_kernel_ms(const qpc_qp_problem *p, double *ms) {
  if (!p || !ms) { qpc_set_error("qpc_qp_last_kernel_ms: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  *ms = p->last_kernel_ms;
  return QPC_OK;
}

int32_t qpc_qp_plan_info(const qpc_qp_problem *p, int64_t *out, int32_t n_out) {
  if (!p || !out || n_out < 0) { qpc_set_error("qpc_qp_plan_info: bad arguments"); return QPC_ERR_INVALID_ARGUMENT; }
  const qpc::QpSpPlanHost &P = p->plan;
  const bool on = p->plan_tried && P.valid && p->sp_lds_doubles > 0;
  long long dense = 0;
  for (int j = 0; j < P.ne; j++) dense += (long long)(P.m + 1 - j) * (P.m + 2 - j) / 2;
  const int64_t v[12] = {on ? 1 : 0, P.ne, P.mi, P.nfront, P.nlevel, P.njob, P.nval, (int64_t)p->sp_lds_doubles * 8, P.maxw, P.maxr, P.flops, dense};
  for (int i = 0; i < 12 && i < n_out; i++) out[i] = v[i];
  return QPC_OK;
}

void qpc_qp_free(qpc_qp_problem *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->stream) { (void)hipStreamSynchronize(p->stream); (void)hipStreamDestroy(p->stream); }
  if (p->d_ints) (void)hipFree(p->d_ints);
  if (p->d_coo) (void)hipFree(p->d_coo);
  if (p->d_ticket) (void)hipFree(p->d_ticket);
  if (p->buf) (void)hipFree(p->buf);
  if (p->d_yprev) (void)hipFree(p->d_yprev);
  if (p->dense) (void)hipFree(p->dense);
  if (p->h_stage) (void)hipHostFree(p->h_stage);
  if (p->d_plan) (void)hipFree(p->d_plan);
  if (p->ws) (void)hipFree(p->ws);
  if (p->d_stamps) (void)hipFree(p->d_stamps);
  if (p->ev0) { (void)hipEventDestroy(p->ev0); (void)hipEventDestroy(p->ev1); }
  if (p->stream2) { (void)hipStreamSynchronize(p->stream2); (void)hipStreamDestroy(p->stream2); }
  if (p->d_ticket2) (void)hipFree(p->d_ticket2);
  if (p->d_left) (void)hipFree(p->d_left);
  if (p->h_left) (void)hipHostFree(p->h_left);
  delete p;
}

int32_t qpc_qp_solve(qpc_qp_problem *p, int64_t B, const double *P_vals, const double *q, const double *A_vals, const double *l,
                     const double *u, const qpc_settings *settings, int32_t warm_start, double *x, double *y, int32_t *status,
                     int32_t *iters) {
  if (!p) { qpc_set_error("qpc_qp_solve: null problem"); return QPC_ERR_INVALID_ARGUMENT; }
  const int n = p->H.n, m = p->H.m;
  const long long nnzP = p->H.nnzP, nnzA = p->H.nnzA;
  if (B < 0 || (nnzP && !P_vals) || !q || (m && ((nnzA && !A_vals) || !l || !u)) || !x || (m && !y) || !status) {
    qpc_set_error("qpc_qp_solve: bad arguments"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return QPC_OK;
#ifdef QPC_EXPERIMENTS // QPC_L0_HOST_PROF=1 (experiments library): where the host side of a small call spends its time (stderr, microseconds)
  static const bool host_prof = [] { const char *e = getenv("QPC_L0_HOST_PROF"); return e && e[0] == '1'; }();
  struct timespec hp0__;
  clock_gettime(CLOCK_MONOTONIC, &hp0__);
  double hp_last__ = 0.0;
  auto HP = [&](const char *what) {
    if (!host_prof) return;
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    const double us = (t.tv_sec - hp0__.tv_sec) * 1e6 + (t.tv_nsec - hp0__.tv_nsec) * 1e-3;
    fprintf(stderr, "l0_host_prof %-28s +%7.2f us (at %7.2f)\n", what, us - hp_last__, us);
    hp_last__ = us;
  };
#else
  auto HP = [](const char *) {};
#endif
  HIP_TRY(hipSetDevice(p->device));
  HP("hipSetDevice");
  qpc_settings st;
  if (settings) st = *settings; else qpc_default_settings(&st);
  const size_t nPv = (size_t)B * nnzP, nq = (size_t)B * n, nAv = (size_t)B * nnzA, nm = (size_t)B * m;
  // the dense forms (what the sparse form leaves) work through the batch in chunks of QPC_L0_LEFT_CHUNK QPs: their workspace (875 KB per
  // Atlas-sized QP) and the dense copies of P and A are sized for a chunk, not for the batch
  const int64_t lch = B < QPC_L0_LEFT_CHUNK ? B : (int64_t)QPC_L0_LEFT_CHUNK;
  const size_t wsb = (size_t)qpc_qp_workspace_bytes(n, m > 0 ? m : 1, lch);
  const size_t total = pad64(nPv * 8) + 2 * pad64(nq * 8) + pad64(nAv * 8) + 3 * pad64(nm * 8) + 2 * pad64((size_t)B * sizeof(int32_t));
  if (total > p->bytes) {
    if (p->buf) (void)hipFree(p->buf);
    p->buf = nullptr; p->bytes = 0;
    if (hipMalloc(&p->buf, total) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("qpc_qp_solve: out of device memory"); return QPC_ERR_OUT_OF_MEMORY; }
    p->bytes = total;
  }
  char *c = p->buf;
  auto take = [&](size_t bytes) { char *r = c; c += pad64(bytes); return r; };
  // inputs side by side, outputs side by side: a small batch moves in ONE copy each way
  double *dPv = (double *)take(nPv * 8), *dq = (double *)take(nq * 8), *dAv = (double *)take(nAv * 8), *dl = (double *)take(nm * 8),
         *du = (double *)take(nm * 8);
  const size_t in_bytes = (size_t)(c - p->buf);
  char *const out0 = c;
  double *dx = (double *)take(nq * 8), *dy = (double *)take(nm * 8);
  int32_t *dst = (int32_t *)take((size_t)B * sizeof(int32_t)), *dit = (int32_t *)take((size_t)B * sizeof(int32_t));
  const size_t out_bytes = (size_t)(c - out0);
  double *dws = nullptr; // the dense forms' per-QP workspace (875 KB per Atlas-sized QP): only when one of them runs
  auto want_ws = [&]() -> int {
    if (wsb > p->ws_bytes) {
      if (p->ws) (void)hipFree(p->ws);
      p->ws = nullptr; p->ws_bytes = 0;
      if (hipMalloc(&p->ws, wsb) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("qpc_qp_solve: out of device memory (workspace of the dense forms)"); return QPC_ERR_OUT_OF_MEMORY; }
      p->ws_bytes = wsb;
    }
    dws = (double *)p->ws;
    return QPC_OK;
  };
  int rc = QPC_OK;
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  const bool warm = warm_start && p->d_yprev && p->B_prev >= B && m > 0;
  const bool wg = m > 0 && p->H.wg_tables; // (diagonal P, n, m <= QPC_WG_DIM: qp_prep_build)
  // the sparse workgroup form first (qpc_qp_sp.h): its plan is made for the equality rows of the batch's first QP (kept while they stay)
  bool sp = false;
  if (m > 0 && n <= QPC_SP_DIM && m <= QPC_SP_DIM) {
    const char *e = QPC_EXP_ENV("QPC_L0_NO_SPARSE"); // (experiments library only: the product has no switches)
    if (!(e && e[0] == '1')) {
      std::vector<uint8_t> eq;
      qpc_eq_rows(m, l, u, eq);
      if (int prc = qpc_qp_plan(p, eq)) return prc;
      sp = p->plan.valid && p->sp_lds_doubles > 0;
    }
  }
  HP("plan check");
  const bool small = (wg || sp) && in_bytes + out_bytes <= ((size_t)1 << 20);
  if (small && in_bytes + out_bytes > p->h_stage_bytes) {
    if (p->h_stage) (void)hipHostFree(p->h_stage);
    p->h_stage = nullptr; p->h_stage_bytes = 0;
    if (hipHostMalloc(&p->h_stage, (size_t)1 << 20, hipHostMallocDefault) == hipSuccess) p->h_stage_bytes = (size_t)1 << 20; else (void)hipGetLastError();
  }
  const bool staged = small && p->h_stage != nullptr;
  // a LARGE batch through the sparse form moves in chunks over two streams: what bounds the call is the copy of the values over PCIe
  // (13 KB per Atlas problem: 106 MB for 8 192) -- the kernel of one chunk runs while the host stages the next one's inputs
  const int64_t chunk = 1024;
  bool piped = sp && !staged && B >= 4 * chunk && !p->timing;
  if (piped) {
    const char *e = QPC_EXP_ENV("QPC_L0_NO_CHUNKS");
    if (e && e[0] == '1') piped = false;
    if (piped && !p->stream2) {
      if (hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking) != hipSuccess || hipMalloc(&p->d_ticket2, sizeof(unsigned long long)) != hipSuccess ||
          hipMemset(p->d_ticket2, 0, sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); piped = false; }
    }
  }
  if (piped) { // (the copies are part of the chunk loop below)
  } else if (staged) { // the same layout in the page-locked buffer, one copy
    char *h = p->h_stage;
    if (nPv) memcpy(h + ((char *)dPv - p->buf), P_vals, nPv * 8);
    memcpy(h + ((char *)dq - p->buf), q, nq * 8);
    if (nAv) memcpy(h + ((char *)dAv - p->buf), A_vals, nAv * 8);
    memcpy(h + ((char *)dl - p->buf), l, nm * 8);
    memcpy(h + ((char *)du - p->buf), u, nm * 8);
    HP("stage inputs (memcpy)");
    chk(hipMemcpyAsync(p->buf, h, in_bytes, hipMemcpyHostToDevice, p->stream), "copy the inputs");
    HP("hipMemcpyAsync H2D");
  } else {
    if (nPv) chk(hipMemcpyAsync(dPv, P_vals, nPv * 8, hipMemcpyHostToDevice, p->stream), "copy P values");
    chk(hipMemcpyAsync(dq, q, nq * 8, hipMemcpyHostToDevice, p->stream), "copy q");
    if (m) {
      if (nAv) chk(hipMemcpyAsync(dAv, A_vals, nAv * 8, hipMemcpyHostToDevice, p->stream), "copy A values");
      chk(hipMemcpyAsync(dl, l, nm * 8, hipMemcpyHostToDevice, p->stream), "copy l");
      chk(hipMemcpyAsync(du, u, nm * 8, hipMemcpyHostToDevice, p->stream), "copy u");
    }
  }
  const long long ws_stride = (long long)qpc::qp_dense_workspace_doubles(n, m > 0 ? m : 1);
  std::vector<int32_t> hst((size_t)B, QPC_WG_FALLBACK);
  auto keep_y = [&]() { // the multipliers stay on the device for the next call's warm start
    if (!m) return;
    if (B > p->cap_prev) {
      if (p->d_yprev) (void)hipFree(p->d_yprev);
      p->d_yprev = nullptr; p->cap_prev = 0; p->B_prev = 0;
      if (hipMalloc(&p->d_yprev, nm * 8) == hipSuccess) p->cap_prev = B; else (void)hipGetLastError();
    }
    if (p->d_yprev) { chk(hipMemcpyAsync(p->d_yprev, dy, nm * 8, hipMemcpyDeviceToDevice, p->stream), "keep y"); p->B_prev = B; }
  };
  const double *y0 = warm ? p->d_yprev : (const double *)nullptr;
  if (rc == QPC_OK && (wg || sp) && B > p->cap_prev) { // the kernels leave every QP's multipliers here for the next call's warm start
    if (p->d_yprev) (void)hipFree(p->d_yprev);
    p->d_yprev = nullptr; p->cap_prev = 0; p->B_prev = 0;
    if (hipMalloc(&p->d_yprev, nm * 8) == hipSuccess) p->cap_prev = B; else (void)hipGetLastError();
  }
  // after a workgroup kernel: the statuses (a small batch: all the outputs, in one copy); returns true when every QP is solved and
  // the outputs have been handed over
  auto timed = [&]() { // after the stream has been synchronised
    if (!p->timing) return;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p->ev0, p->ev1) == hipSuccess) p->last_kernel_ms = (double)ms; else (void)hipGetLastError();
  };
  auto collect = [&](const char *who) -> bool {
    if (p->timing) chk(hipEventRecord(p->ev1, p->stream), "hipEventRecord");
    if (staged) {
      char *ho = p->h_stage + in_bytes;
      chk(hipMemcpyAsync(ho, out0, out_bytes, hipMemcpyDeviceToHost, p->stream), "copy the outputs");
      HP("hipMemcpyAsync D2H");
      chk(hipStreamSynchronize(p->stream), who);
      HP("hipStreamSynchronize");
      if (rc != QPC_OK) { (void)hipMemset(p->d_ticket, 0, sizeof(unsigned long long)); return false; }
      timed();
      if (p->d_yprev) p->B_prev = B;
      memcpy(hst.data(), ho + ((char *)dst - out0), (size_t)B * sizeof(int32_t));
      bool left = false;
      for (int64_t b = 0; b < B; b++) left = left || hst[(size_t)b] == QPC_WG_FALLBACK;
      if (left) return false;
      memcpy(x, ho, nq * 8);
      memcpy(y, ho + ((char *)dy - out0), nm * 8);
      if (iters) memcpy(iters, ho + ((char *)dit - out0), (size_t)B * sizeof(int32_t));
      for (int64_t b = 0; b < B; b++) status[b] = hst[(size_t)b];
      return true;
    }
    chk(hipMemcpyAsync(hst.data(), dst, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream), "copy status");
    chk(hipStreamSynchronize(p->stream), who);
    if (rc != QPC_OK) { (void)hipMemset(p->d_ticket, 0, sizeof(unsigned long long)); return false; }
    timed();
    if (p->d_yprev) p->B_prev = B;
    return false;
  };
  // (timing: the events bracket everything between the copies in and the copies out -- the kernels of the three forms)
  if (p->timing) chk(hipEventRecord(p->ev0, p->stream), "hipEventRecord");
  bool left = true;
  if (rc == QPC_OK && sp) {
    const int lds_bytes = p->sp_lds_doubles * (int)sizeof(double);
    if (int arc = qpc_wg_lds_attr((const void *)qpc_qp_sp_kernel, p->device, lds_bytes)) return arc;
    const long long per_cu = 160 * 1024 / lds_bytes > 0 ? 160 * 1024 / lds_bytes : 1;
    long long grid = B < (long long)p->n_cu * per_cu ? B : (long long)p->n_cu * per_cu;
#ifdef QPC_SP_PROF
    if (const char *eg = getenv("QPC_L0_GRID")) grid = atoi(eg) > 0 ? atoi(eg) : grid; // (profiling build: several QPs through ONE workgroup)
    if (!p->d_stamps) { if (hipMalloc(&p->d_stamps, 32 * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); p->d_stamps = nullptr; } }
#endif
    if (piped) {
      hipStream_t strm[2] = {p->stream, p->stream2};
      unsigned long long *tick[2] = {p->d_ticket, p->d_ticket2};
      const int64_t nchunk = (B + chunk - 1) / chunk;
      auto out_copy = [&](int64_t c) { // (a copy into pageable memory returns when it is done: issued one chunk late, it waits beside the next kernel)
        const int64_t b0 = c * chunk, nb = std::min(chunk, B - b0);
        hipStream_t sm = strm[c & 1];
        chk(hipMemcpyAsync(x + b0 * n, dx + b0 * n, (size_t)nb * n * 8, hipMemcpyDeviceToHost, sm), "copy x");
        chk(hipMemcpyAsync(y + b0 * m, dy + b0 * m, (size_t)nb * m * 8, hipMemcpyDeviceToHost, sm), "copy y");
        chk(hipMemcpyAsync(hst.data() + b0, dst + b0, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, sm), "copy status");
        if (iters) chk(hipMemcpyAsync(iters + b0, dit + b0, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, sm), "copy iters");
      };
      for (int64_t c = 0; c < nchunk && rc == QPC_OK; c++) {
        const int64_t b0 = c * chunk, nb = std::min(chunk, B - b0);
        hipStream_t sm = strm[c & 1];
        if (nnzP) chk(hipMemcpyAsync(dPv + b0 * nnzP, P_vals + b0 * nnzP, (size_t)nb * nnzP * 8, hipMemcpyHostToDevice, sm), "copy P values");
        chk(hipMemcpyAsync(dq + b0 * n, q + b0 * n, (size_t)nb * n * 8, hipMemcpyHostToDevice, sm), "copy q");
        if (nnzA) chk(hipMemcpyAsync(dAv + b0 * nnzA, A_vals + b0 * nnzA, (size_t)nb * nnzA * 8, hipMemcpyHostToDevice, sm), "copy A values");
        chk(hipMemcpyAsync(dl + b0 * m, l + b0 * m, (size_t)nb * m * 8, hipMemcpyHostToDevice, sm), "copy l");
        chk(hipMemcpyAsync(du + b0 * m, u + b0 * m, (size_t)nb * m * 8, hipMemcpyHostToDevice, sm), "copy u");
        const long long g = nb < grid ? nb : grid;
        hipLaunchKernelGGL(qpc_qp_sp_kernel, dim3((unsigned)g), dim3(QPC_SP_THREADS), (size_t)lds_bytes, sm, (const int *)p->d_plan, p->sp_lds_doubles,
                           (long long)nb, nnzP, dPv + b0 * nnzP, dq + b0 * n, dAv + b0 * nnzA, dl + b0 * m, du + b0 * m, st, y0 ? y0 + b0 * m : nullptr,
                           dx + b0 * n, dy + b0 * m, dst + b0, dit + b0, tick[c & 1], p->d_yprev ? p->d_yprev + b0 * m : nullptr, p->d_stamps);
        chk(hipGetLastError(), "qpc_qp_sp_kernel");
        if (c >= 1) out_copy(c - 1);
      }
      if (rc == QPC_OK) out_copy(nchunk - 1);
      chk(hipStreamSynchronize(p->stream), "qpc_qp_sp_kernel");
      chk(hipStreamSynchronize(p->stream2), "qpc_qp_sp_kernel");
      if (rc != QPC_OK) { (void)hipMemset(p->d_ticket, 0, sizeof(unsigned long long)); (void)hipMemset(p->d_ticket2, 0, sizeof(unsigned long long)); return rc; }
      if (p->d_yprev) p->B_prev = B;
      bool any_left = false;
      for (int64_t b = 0; b < B; b++) any_left = any_left || hst[(size_t)b] == QPC_WG_FALLBACK;
      if (!any_left) { for (int64_t b = 0; b < B; b++) status[b] = hst[(size_t)b]; return QPC_OK; }
    } else {
      HP("before launch");
      hipLaunchKernelGGL(qpc_qp_sp_kernel, dim3((unsigned)grid), dim3(QPC_SP_THREADS), (size_t)lds_bytes, p->stream, (const int *)p->d_plan,
                         p->sp_lds_doubles, (long long)B, nnzP, dPv, dq, dAv, dl, du, st, y0, dx, dy, dst, dit, p->d_ticket, p->d_yprev, p->d_stamps);
      chk(hipGetLastError(), "qpc_qp_sp_kernel");
      HP("launch");
      if (collect("qpc_qp_sp_kernel")) return QPC_OK;
    }
    if (rc != QPC_OK) return rc;
    left = false;
    for (int64_t b = 0; b < B; b++) left = left || hst[(size_t)b] == QPC_WG_FALLBACK;
  }
  auto chunk_has_left = [&](int64_t b0, int64_t nb) { // (hst: FALLBACK everywhere before any form has run)
    for (int64_t b = b0; b < b0 + nb; b++) if (hst[(size_t)b] == QPC_WG_FALLBACK) return true;
    return false;
  };
  if (rc == QPC_OK && wg && left) { // the dense workgroup form: everything, or what the sparse form left
    if (int wrc = want_ws()) return wrc;
    const int lds_bytes = QPC_WG_LDS_DOUBLES * (int)sizeof(double);
    if (int arc = qpc_wg_lds_attr((const void *)qpc_qp_wg_prep_kernel, p->device, lds_bytes)) return arc;
    for (int64_t b0 = 0; b0 < B && rc == QPC_OK; b0 += lch) {
      const int64_t nb = std::min(lch, B - b0);
      if (!chunk_has_left(b0, nb)) continue;
      const long long grid = nb < (long long)p->n_cu * 4 ? nb : (long long)p->n_cu * 4;
      // (no memset of the QP counter: the launch's last ticket re-arms it; the chunks follow each other on the stream and share the workspace)
      hipLaunchKernelGGL(qpc_qp_wg_prep_kernel, dim3((unsigned)grid), dim3(QPC_WG_THREADS), (size_t)lds_bytes, p->stream, p->H.view(p->d_ints),
                         (long long)nb, nnzP, dPv + b0 * nnzP, dq + b0 * n, dAv + b0 * nnzA, dl + b0 * m, du + b0 * m, st, dws, ws_stride,
                         y0 ? y0 + b0 * m : nullptr, dx + b0 * n, dy + b0 * m, dst + b0, dit + b0, p->d_ticket, p->d_yprev ? p->d_yprev + b0 * m : nullptr,
                         sp ? 1 : 0);
      chk(hipGetLastError(), "qpc_qp_wg_prep_kernel");
    }
    if (collect("qpc_qp_wg_prep_kernel")) return QPC_OK;
    if (rc != QPC_OK) return rc;
  }
  bool need_dense = false;
  for (int64_t b = 0; b < B && !need_dense; b++) need_dense = hst[(size_t)b] == QPC_WG_FALLBACK;
  if (rc == QPC_OK && need_dense) { // the dense path: scatter P, A and run the dense entry on what is left, chunk by chunk
    if (int wrc = want_ws()) return wrc;
    const size_t nP = (size_t)lch * n * n, nA = (size_t)lch * (m > 0 ? m : 1) * n;
    const size_t need = pad64(nP * 8) + pad64(nA * 8);
    if (need > p->dense_bytes) {
      if (p->dense) (void)hipFree(p->dense);
      p->dense = nullptr; p->dense_bytes = 0;
      if (hipMalloc(&p->dense, need) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("qpc_qp_solve: out of device memory (dense path)"); return QPC_ERR_OUT_OF_MEMORY; }
      p->dense_bytes = need;
    }
    double *dP = (double *)p->dense, *dA = (double *)(p->dense + pad64(nP * 8));
    const int *Prow = p->d_coo, *Pcol = Prow + nnzP, *Arow = Pcol + nnzP, *Acol = Arow + nnzA;
    for (int64_t b0 = 0; b0 < B && rc == QPC_OK; b0 += lch) {
      const int64_t nb = std::min(lch, B - b0);
      if (!chunk_has_left(b0, nb)) continue;
      chk(hipMemsetAsync(p->dense, 0, need, p->stream), "clear dense P, A");
      if (nnzP + nnzA > 0) {
        hipLaunchKernelGGL(qpc_qp_densify_kernel, dim3(1024), dim3(256), 0, p->stream, n, m, (long long)nb, nnzP, Prow, Pcol, dPv + b0 * nnzP, nnzA, Arow,
                           Acol, dAv + b0 * nnzA, dP, dA);
        chk(hipGetLastError(), "qpc_qp_densify_kernel");
      }
      if (!wg && !sp) { // everything through the dense entry (its own workgroup pass detects what it can take)
        if (rc == QPC_OK) rc = qpc_qp_solve_batch_device(n, m, nb, dP, dq + b0 * n, dA, dl + b0 * m, du + b0 * m, &st, p->device, dx + b0 * n, dy + b0 * m,
                                                         dst + b0, dit + b0, dws, wsb, p->stream);
      } else {   // only the QPs the workgroup forms left: the one-lane kernel
        hipLaunchKernelGGL(qpc_qp_dense_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, p->stream, n, m, (long long)nb, dP, dq + b0 * n, dA,
                           dl + b0 * m, du + b0 * m, st, dws, ws_stride, dx + b0 * n, dy + b0 * m, dst + b0, dit + b0, 1);
        chk(hipGetLastError(), "qpc_qp_dense_kernel");
      }
    }
    chk(hipMemcpyAsync(hst.data(), dst, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream), "copy status");
  }
  if (rc == QPC_OK) {
    chk(hipMemcpyAsync(x, dx, nq * 8, hipMemcpyDeviceToHost, p->stream), "copy x");
    if (m) chk(hipMemcpyAsync(y, dy, nm * 8, hipMemcpyDeviceToHost, p->stream), "copy y");
    if (iters) chk(hipMemcpyAsync(iters, dit, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, p->stream), "copy iters");
    if (need_dense) keep_y(); // (the workgroup kernel has kept its own; the QPs the dense path re-solved have new multipliers)
  }
  chk(hipStreamSynchronize(p->stream), "qpc_qp_solve");
  if (rc == QPC_OK) for (int64_t b = 0; b < B; b++) status[b] = hst[(size_t)b];
  return rc;
}

// ---- the prepared entry on DEVICE pointers: values, bounds and results stay in HBM (a caller that builds its QPs on the GPU; the rate
//      without the 13 KB per problem over PCIe).  The same three forms in the same order; between them the host learns how many QPs are
//      left (one 4-byte copy and one synchronisation of `hip_stream` per form that ran) -- normally the sparse form takes them all.
int32_t qpc_qp_solve_device(qpc_qp_problem *p, int64_t B, const double *dPv, const double *dq, const double *dAv, const double *dl,
                            const double *du, const qpc_settings *settings, int32_t warm_start, double *dx, double *dy, int32_t *dst,
                            int32_t *dit, void *hip_stream) {
  if (!p) { qpc_set_error("qpc_qp_solve_device: null problem"); return QPC_ERR_INVALID_ARGUMENT; }
  const int n = p->H.n, m = p->H.m;
  const long long nnzP = p->H.nnzP, nnzA = p->H.nnzA;
  if (B < 0 || (nnzP && !dPv) || !dq || (m && ((nnzA && !dAv) || !dl || !du)) || !dx || (m && !dy) || !dst) {
    qpc_set_error("qpc_qp_solve_device: bad arguments"); return QPC_ERR_INVALID_ARGUMENT;
  }
  if (B == 0) return QPC_OK;
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t sm = (hipStream_t)hip_stream;
  qpc_settings st;
  if (settings) st = *settings; else qpc_default_settings(&st);
  if (!p->d_left) {
    HIP_TRY(hipMalloc(&p->d_left, sizeof(int)));
    if (hipHostMalloc(&p->h_left, sizeof(int), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p->d_left); p->d_left = nullptr; qpc_set_error("qpc_qp_solve_device: out of page-locked memory"); return QPC_ERR_OUT_OF_MEMORY; }
  }
  const size_t nm = (size_t)B * m;
  const bool sp = m > 0 && p->plan_tried && p->plan.valid && p->sp_lds_doubles > 0; // (the plan: qpc_qp_analyze, or an earlier qpc_qp_solve)
  const bool wg = m > 0 && p->H.wg_tables;
  const bool warm = warm_start && p->d_yprev && p->B_prev >= B && m > 0;
  if ((wg || sp) && B > p->cap_prev) { // the kernels leave every QP's multipliers here for the next call's warm start
    HIP_TRY(hipStreamSynchronize(sm));
    if (p->d_yprev) (void)hipFree(p->d_yprev);
    p->d_yprev = nullptr; p->cap_prev = 0; p->B_prev = 0;
    if (hipMalloc(&p->d_yprev, nm * 8) == hipSuccess) p->cap_prev = B; else (void)hipGetLastError();
  }
  const double *y0 = warm ? p->d_yprev : (const double *)nullptr;
  int rc = QPC_OK;
  auto chk = [&](hipError_t e, const char *w) { if (e != hipSuccess && rc == QPC_OK) rc = hip_fail(e, w); };
  auto left_after = [&](const char *who) -> long long { // synchronises: how many QPs the form that just ran has left
    chk(hipMemsetAsync(p->d_left, 0, sizeof(int), sm), "clear the counter");
    hipLaunchKernelGGL(qpc_qp_count_left_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, sm, (const int32_t *)dst, (long long)B, p->d_left);
    chk(hipGetLastError(), "qpc_qp_count_left_kernel");
    chk(hipMemcpyAsync(p->h_left, p->d_left, sizeof(int), hipMemcpyDeviceToHost, sm), "copy the counter");
    chk(hipStreamSynchronize(sm), who);
    if (rc != QPC_OK) { (void)hipMemset(p->d_ticket, 0, sizeof(unsigned long long)); return -1; }
    return (long long)*p->h_left;
  };
  long long left = B;
  if (sp) {
    const int lds_bytes = p->sp_lds_doubles * (int)sizeof(double);
    if (int arc = qpc_wg_lds_attr((const void *)qpc_qp_sp_kernel, p->device, lds_bytes)) return arc;
    const long long per_cu = 160 * 1024 / lds_bytes > 0 ? 160 * 1024 / lds_bytes : 1;
    const long long grid = B < (long long)p->n_cu * per_cu ? B : (long long)p->n_cu * per_cu;
    hipLaunchKernelGGL(qpc_qp_sp_kernel, dim3((unsigned)grid), dim3(QPC_SP_THREADS), (size_t)lds_bytes, sm, (const int *)p->d_plan, p->sp_lds_doubles,
                       (long long)B, nnzP, dPv, dq, dAv, dl, du, st, y0, dx, dy, dst, dit, p->d_ticket, p->d_yprev, (unsigned long long *)nullptr);
    chk(hipGetLastError(), "qpc_qp_sp_kernel");
    left = left_after("qpc_qp_sp_kernel");
    if (rc != QPC_OK) return rc;
    if (p->d_yprev) p->B_prev = B;
    if (left == 0) return QPC_OK;
  }
  // what is left (other equality rows than the plan's; no plan at all): the dense forms, chunk by chunk (the host does not know WHICH QPs are
  // left, only how many: every chunk is launched, a workgroup skips a solved QP after one load)
  const int64_t lch = B < QPC_L0_LEFT_CHUNK ? B : (int64_t)QPC_L0_LEFT_CHUNK;
  const size_t wsb = (size_t)qpc_qp_workspace_bytes(n, m > 0 ? m : 1, lch);
  if (wsb > p->ws_bytes) {
    if (p->ws) (void)hipFree(p->ws);
    p->ws = nullptr; p->ws_bytes = 0;
    if (hipMalloc(&p->ws, wsb) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("qpc_qp_solve_device: out of device memory (workspace of the dense forms)"); return QPC_ERR_OUT_OF_MEMORY; }
    p->ws_bytes = wsb;
  }
  double *dws = (double *)p->ws;
  const long long ws_stride = (long long)qpc::qp_dense_workspace_doubles(n, m > 0 ? m : 1);
  if (wg) {
    const int lds_bytes = QPC_WG_LDS_DOUBLES * (int)sizeof(double);
    if (int arc = qpc_wg_lds_attr((const void *)qpc_qp_wg_prep_kernel, p->device, lds_bytes)) return arc;
    for (int64_t b0 = 0; b0 < B && rc == QPC_OK; b0 += lch) {
      const int64_t nb = std::min(lch, B - b0);
      const long long grid = nb < (long long)p->n_cu * 4 ? nb : (long long)p->n_cu * 4;
      hipLaunchKernelGGL(qpc_qp_wg_prep_kernel, dim3((unsigned)grid), dim3(QPC_WG_THREADS), (size_t)lds_bytes, sm, p->H.view(p->d_ints), (long long)nb, nnzP,
                         dPv + b0 * nnzP, dq + b0 * n, dAv + b0 * nnzA, dl + b0 * m, du + b0 * m, st, dws, ws_stride, y0 ? y0 + b0 * m : nullptr,
                         dx + b0 * n, dy + b0 * m, dst + b0, dit ? dit + b0 : nullptr, p->d_ticket, p->d_yprev ? p->d_yprev + b0 * m : nullptr, sp ? 1 : 0);
      chk(hipGetLastError(), "qpc_qp_wg_prep_kernel");
    }
    left = left_after("qpc_qp_wg_prep_kernel");
    if (rc != QPC_OK) return rc;
    if (p->d_yprev) p->B_prev = B;
    if (left == 0) return QPC_OK;
  }
  { // the dense path: scatter P, A and run the dense entry on what is left
    const size_t nP = (size_t)lch * n * n, nA = (size_t)lch * (m > 0 ? m : 1) * n;
    const size_t need = pad64(nP * 8) + pad64(nA * 8);
    if (need > p->dense_bytes) {
      if (p->dense) (void)hipFree(p->dense);
      p->dense = nullptr; p->dense_bytes = 0;
      if (hipMalloc(&p->dense, need) != hipSuccess) { (void)hipGetLastError(); qpc_set_error("qpc_qp_solve_device: out of device memory (dense path)"); return QPC_ERR_OUT_OF_MEMORY; }
      p->dense_bytes = need;
    }
    double *dP = (double *)p->dense, *dA = (double *)(p->dense + pad64(nP * 8));
    const int *Prow = p->d_coo, *Pcol = Prow + nnzP, *Arow = Pcol + nnzP, *Acol = Arow + nnzA;
    for (int64_t b0 = 0; b0 < B && rc == QPC_OK; b0 += lch) {
      const int64_t nb = std::min(lch, B - b0);
      chk(hipMemsetAsync(p->dense, 0, need, sm), "clear dense P, A");
      if (nnzP + nnzA > 0) {
        hipLaunchKernelGGL(qpc_qp_densify_kernel, dim3(1024), dim3(256), 0, sm, n, m, (long long)nb, nnzP, Prow, Pcol, dPv + b0 * nnzP, nnzA, Arow, Acol,
                           dAv + b0 * nnzA, dP, dA);
        chk(hipGetLastError(), "qpc_qp_densify_kernel");
      }
      if (!wg && !sp) {
        if (rc == QPC_OK) rc = qpc_qp_solve_batch_device(n, m, nb, dP, dq + b0 * n, dA, dl + b0 * m, du + b0 * m, &st, p->device, dx + b0 * n, dy + b0 * m,
                                                         dst + b0, dit ? dit + b0 : nullptr, dws, wsb, sm);
      } else {
        hipLaunchKernelGGL(qpc_qp_dense_kernel, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, sm, n, m, (long long)nb, dP, dq + b0 * n, dA, dl + b0 * m,
                           du + b0 * m, st, dws, ws_stride, dx + b0 * n, dy + b0 * m, dst + b0, dit ? dit + b0 : nullptr, 1);
        chk(hipGetLastError(), "qpc_qp_dense_kernel");
      }
    }
    if (m && rc == QPC_OK) { // the multipliers for the next call's warm start (the workgroup kernels keep their own)
      if (B > p->cap_prev) {
        chk(hipStreamSynchronize(sm), "qpc_qp_solve_device");
        if (p->d_yprev) (void)hipFree(p->d_yprev);
        p->d_yprev = nullptr; p->cap_prev = 0; p->B_prev = 0;
        if (hipMalloc(&p->d_yprev, nm * 8) == hipSuccess) p->cap_prev = B; else (void)hipGetLastError();
      }
      if (p->d_yprev) { chk(hipMemcpyAsync(p->d_yprev, dy, nm * 8, hipMemcpyDeviceToDevice, sm), "keep y"); p->B_prev = B; }
    }
    chk(hipStreamSynchronize(sm), "qpc_qp_solve_device");
  }
  return rc;
}

} // extern "C"
