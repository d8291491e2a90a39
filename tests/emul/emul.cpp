// emul.cpp -- TEST-ONLY lane-serial CPU emulation of the device code in
// qpcontrol.jl_amd/csrc/qpc_tick.h.  A "phase" runs its per-lane block for lanes 0..63 in turn;
// LDS is a host array behind a checking proxy that reports, per phase,
//   * read of a word another lane wrote in the same phase   (RAW race on the GPU)
//   * write of a word another lane read or wrote in the same phase (WAR / WAW race)
//   * read of a never-written word, writes outside a phase
// This is the "CPU build" for sanitizers and algorithm debugging; it is never part of the
// product library and never a fallback: it exports the same qpc_finalize / qpc_control_batch
// symbols from a separate test-only shared object (tests/_build/libqpc_emul.so).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../qpcontrol.jl_amd/csrc/qpc_compile.h"
#include "../../qpcontrol.jl_amd/csrc/qpc_pipeline.h"

// arithmetic type of the emulated tick: double, or the float class of the fp32 variant (libqpc_emul_f32.so: -DQPC_F32), the
// counterpart of the fp32 hiprtc module of qpc_jit.cpp
#include "../../qpcontrol.jl_amd/csrc/qpc_real.h"

struct EmuState {
  std::vector<real> mem;
  std::vector<int> wphase, wlane, rphase, rlane;
  int phase = 0, lane = -1, section = -1;
  long races = 0, uninit = 0, outside = 0;
  std::string first;
  void reset(int n) {
    mem.assign(n, real(std::nan("")));
    wphase.assign(n, -1); wlane.assign(n, -1); rphase.assign(n, -1); rlane.assign(n, -1);
    phase = 0; lane = -1;
  }
  void note(const char *what, int i) {
    if (first.empty()) { char b[160]; snprintf(b, sizeof b, "%s at lds[%d] phase %d (after section %d) lane %d (writer lane %d, reader lane %d)", what, i, phase, section, lane, wlane[i], rlane[i]); first = b; }
  }
  void on_read(int i) {
    if (wphase[i] < 0) { uninit++; note("uninitialised read", i); if (getenv("EMU_TRACE")) fprintf(stderr, "uninit read lds[%d] phase %d section %d lane %d\n", i, phase, section, lane); }
    if (lane >= 0) {
      if (wphase[i] == phase && wlane[i] != lane) { races++; note("RAW race", i); }
      if (rphase[i] == phase) { if (rlane[i] != lane) rlane[i] = -2; }
      else { rphase[i] = phase; rlane[i] = lane; }
    }
  }
  void on_write(int i) {
    if (lane < 0) { outside++; note("write outside a phase", i); return; }
    if (wphase[i] == phase && wlane[i] != lane) { races++; note("WAW race", i); }
    if (rphase[i] == phase && rlane[i] != lane) { races++; note("WAR race", i); }
    wphase[i] = phase; wlane[i] = lane;
  }
};
static thread_local EmuState *g_emu = nullptr;
static inline void emu_enter_lane(int l) { g_emu->lane = l; }

struct Ref {
  int i;
  operator real() const { g_emu->on_read(i); return g_emu->mem[i]; }
  Ref &operator=(real v) { g_emu->on_write(i); g_emu->mem[i] = v; return *this; }
  Ref &operator=(const Ref &o) { return *this = (real)o; }
};
struct Lds {
  int base;
  Ref operator[](int i) const { return Ref{base + i}; }
  Lds operator+(int o) const { return Lds{base + o}; }
};

#define QPC_PROF_T0()
#define QPC_PROF_ACC(i) (g_emu->section = (i)) // the last section marker passed: locates a report in the source
#define QPC_FN inline
#define QPC_PHASE(lane) for (int lane = 0; lane < 64; ++lane) { emu_enter_lane(lane);
#define QPC_END_PHASE } g_emu->lane = -1; g_emu->phase++;
// soft end: the lane loop ends (blocks still run one after the other) but the race detector's phase number stays, so a
// cross-lane RAW / WAR / WAW between soft-joined blocks is reported like one inside a single phase
#define QPC_END_PHASE_SOFT } g_emu->lane = -1;
#define QPC_END_PHASE_REG QPC_END_PHASE_SOFT
#define QPC_PHASE_SYNC g_emu->phase++;
#define QPC_LANE_SCOPE
#ifdef QPC_F32
typedef float emu_raw;
#else
typedef double emu_raw;
#endif
#define QPC_SINCOS(x, s, c) do { emu_raw x__ = (emu_raw)(real)(x); s = real(std::sin(x__)); c = real(std::cos(x__)); } while (0)
#define QPC_SQRT(x) real(std::sqrt((emu_raw)(real)(x)))
#define QPC_FABS(x) real(std::fabs((emu_raw)(real)(x)))
#define QPC_ATAN2(y, x) real(std::atan2((emu_raw)(real)(y), (emu_raw)(real)(x)))
#define QPC_RCP(x) real((emu_raw)1 / (emu_raw)(real)(x))
#define QPC_RCP_SEARCH(x) real((emu_raw)1 / (emu_raw)(real)(x))
#define QPC_RSQRT(x) real((emu_raw)1 / std::sqrt((emu_raw)(real)(x)))
#define QPC_LANE_ARRAY(name, n) real name[64][n]
#define QPC_LA(name, lane) name[lane]
#define QPC_LANE_INTS(name, n) int name[64][n]
#define QPC_LI(name, lane) name[lane]
#define QPC_CTZ64(x) __builtin_ctzll((unsigned long long)(x))
#define QPC_POPC64(x) __builtin_popcountll((unsigned long long)(x))
// lane-serial model of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (operand / accumulator lane maps of the guide: the f64
// instruction has its own C/D row map)
#ifdef QPC_F32
#define QPC_MFMA_CROW(lane, reg) (4 * ((lane) >> 4) + (reg))
#else
#define QPC_MFMA_CROW(lane, reg) (((lane) >> 4) + 4 * (reg))
#endif
static thread_local real emu_mfma_a[64], emu_mfma_b[64];
#define QPC_MFMA_16x16x4(lane, ACC, AEXPR, BEXPR)                                                    \
  QPC_PHASE(lane) { emu_mfma_a[lane] = (AEXPR); emu_mfma_b[lane] = (BEXPR); } QPC_END_PHASE               \
  QPC_PHASE(lane) {                                                                                       \
    real *acc__ = (ACC);                                                                                \
    for (int r__ = 0; r__ < 4; r__++) {                                                                   \
      const int row__ = QPC_MFMA_CROW(lane, r__), col__ = lane & 15;                                        \
      for (int k__ = 0; k__ < 4; k__++) acc__[r__] += emu_mfma_a[row__ + 16 * k__] * emu_mfma_b[col__ + 16 * k__]; \
    }                                                                                                     \
  } QPC_END_PHASE
// lane-serial model of the DPP quad broadcasts: publish, then gather from the quad
static thread_local real emu_quad[64];
#define QPC_QUAD_GATHER(lane, OUT4, XEXPR)                                                    \
  QPC_PHASE(lane) { emu_quad[lane] = (XEXPR); } QPC_END_PHASE_SOFT                              \
  QPC_PHASE(lane) { real *o__ = (OUT4); for (int i__ = 0; i__ < 4; i__++) o__[i__] = emu_quad[(lane & ~3) + i__]; } QPC_END_PHASE_SOFT
// lane-serial model of the DPP row_share broadcasts: publish, then gather from lane K of the row of 16
static thread_local real emu_row[64][8];
#define QPC_ROW_SHARE(lane, OUT, N, X, K)                                                                              \
  QPC_PHASE(lane) { const real *x__ = (X); for (int i__ = 0; i__ < (N); i__++) emu_row[lane][i__] = x__[i__]; } QPC_END_PHASE_SOFT \
  QPC_PHASE(lane) { real *o__ = (OUT); for (int i__ = 0; i__ < (N); i__++) o__[i__] = emu_row[(lane & ~15) + (K)][i__]; } QPC_END_PHASE_SOFT
// lane-serial model of the wave prefix sum (the device adds in tree order: results agree to rounding, not bit for bit)
#define QPC_WAVE_INCL_SCAN(lane, ARR, N)                                                                                \
  QPC_PHASE(lane) { const real *x__ = (ARR); for (int i__ = 0; i__ < (N); i__++) emu_row[lane][i__] = x__[i__]; } QPC_END_PHASE_SOFT \
  QPC_PHASE(lane) { real *o__ = (ARR); for (int i__ = 0; i__ < (N); i__++) { real s__ = 0.0; for (int l__ = 0; l__ <= lane; l__++) s__ += emu_row[l__][i__]; o__[i__] = s__; } } QPC_END_PHASE_SOFT
// reductions across a phase boundary: lane-serial form through the (emulator-only) scratch that
// follows the kernel's LDS slab
#define QPC_RED_MAX_PUT(slot, name, val) sm[D.lds_doubles + 128 * (slot) + lane] = (val)
// like the device (qpc_device.h wave_max_scale): maximum of non-negative values, low mantissa word dropped
#ifdef QPC_F32
static inline real emu_scale_trunc(real x) { return x < real(0) ? -x : x; } // (fp32: the device takes the maximum of the whole word)
#else
static inline double emu_scale_trunc(double x) {
  unsigned long long b;
  std::memcpy(&b, &x, 8);
  b &= 0x7fffffff00000000ull;
  std::memcpy(&x, &b, 8);
  return x;
}
#endif
#define QPC_RED_MAX_GET(slot, name) emu_scale_trunc(qpc::umax(sm, D.lds_doubles + 128 * (slot), 64))
#define QPC_RED_SUM_PUT(slot, name, val) sm[D.lds_doubles + 128 * (slot) + lane] = (val)
static inline real emu_sum64(const real *x) { real s = 0.0; for (int l = 0; l < 64; l++) s += x[l]; return s; }
#define QPC_RED_SUM_GET(slot, name) ([&]() { real x__[64]; for (int l__ = 0; l__ < 64; l__++) x__[l__] = sm[D.lds_doubles + 128 * (slot) + l__]; return emu_sum64(x__); }())
#define QPC_BALLOT_PUT(slot, name, pred) sm[D.lds_doubles + 128 * (slot) + lane] = (pred) ? 1.0 : 0.0
#define QPC_BALLOT_GET(slot, name) \
  do { name = 0ull; for (int l__ = 0; l__ < 64; l__++) if ((real)sm[D.lds_doubles + 128 * (slot) + l__] != 0.0) name |= 1ull << l__; } while (0)
#define QPC_RED_ARGMIN_PUT(slot, name, iname, val, idx) \
  do { sm[D.lds_doubles + 128 * (slot) + lane] = (val); sm[D.lds_doubles + 128 * (slot) + 64 + lane] = (real)(idx); } while (0)
#define QPC_RED_ARGMIN_LANE_PUT(slot, name, iname, val, idx) QPC_RED_ARGMIN_PUT(slot, name, iname, val, idx)
#define QPC_RED_ARGMIN_GET(slot, name, iname) \
  do { name = QPC_HUGE; iname = 0x3fffffff; \
       for (int l__ = 0; l__ < 64; l__++) { real v__ = sm[D.lds_doubles + 128 * (slot) + l__]; \
         int i__ = (int)(real)sm[D.lds_doubles + 128 * (slot) + 64 + l__]; \
         if (v__ < name || (v__ == name && i__ < iname)) { name = v__; iname = i__; } } } while (0)
// ancestor-chain bound of the tick: the library's limit (tree depth <= 32, checked by qpc_compile_program) -- the device gets the actual
// depth from the JIT; QPC_GENERIC_MAXD = 16 is a workaround for a hipcc miscompile of the ahead-of-time kernel only
#define QPC_MAXD 32
#include "../../qpcontrol.jl_amd/csrc/qpc_tick.h"
#ifndef QPC_F32 // (the simulation step and the dense L0 / L1 solvers are fp64 code: not part of the fp32 emulation)
#include "../../qpcontrol.jl_amd/csrc/qpc_sim.h"
#include "../../qpcontrol.jl_amd/csrc/qpc_qp_dense.h"
#include "../../qpcontrol.jl_amd/csrc/qpc_qp_wg.h"
#include "../../qpcontrol.jl_amd/csrc/qpc_qp_prep.h"
#include "../../qpcontrol.jl_amd/csrc/qpc_qp_sym.h"
#include "../../qpcontrol.jl_amd/csrc/qpc_qp_sp.h"
#include "../../qpcontrol.jl_amd/csrc/qpc_l1_dense.h"
#endif

static std::mutex g_rep_mu; // the host pipeline's worker threads report into these
static long g_races = 0, g_uninit = 0, g_outside = 0;
static std::string g_first;

// one emulated tick per QP of [lo, lo + n): what a chunk of the host pipeline does on a (fake) device
static void emu_ticks(qpc_controller *c, const QpcHostJob &job, int64_t lo, int64_t n) {
  const QpcDims &D = c->prog.dims;
  EmuState emu;
  g_emu = &emu;
  long races = 0, uninit = 0, outside = 0;
  std::string first;
  auto in = [&](int i, int64_t b) { return job.in[i] ? job.in[i] + b * (int64_t)job.in_dim[i] : nullptr; };
  for (int64_t b = lo; b < lo + n; b++) {
    emu.reset(D.lds_doubles + 128 * 8);
    qpc::TickIO io;
    io.q = in(QpcHostJob::IN_Q, b); io.v = in(QpcHostJob::IN_V, b); io.desired = in(QpcHostJob::IN_DES, b);
    io.cw = in(QpcHostJob::IN_CW, b); io.cmax = in(QpcHostJob::IN_CMAX, b);
    io.tw = in(QpcHostJob::IN_TW, b); io.surf = in(QpcHostJob::IN_SURF, b);
    io.tau = job.out[QpcHostJob::OUT_TAU] + b * D.nv; io.vdot = job.out[QpcHostJob::OUT_VDOT] + b * D.nv;
    io.cwrench = job.out[QpcHostJob::OUT_CWRENCH] ? job.out[QpcHostJob::OUT_CWRENCH] + b * 6 * D.nc : nullptr;
    io.status = job.status + b; io.iters = job.iters ? job.iters + b : nullptr; io.prof = nullptr;
    io.wset = job.working_set ? (unsigned long long *)job.working_set + 2 * b : nullptr;
    qpc::qpc_tick_prologue<Lds>(D, c->prog.dims_s, c->prog.itab.data(), Lds{0});
    if (job.working_set) qpc::qpc_tick<Lds, true>(D, c->prog.dims_s, c->prog.itab.data(), qpc_dtab{c->prog.dtab.data()}, Lds{0}, io);
    else qpc::qpc_tick<Lds, false>(D, c->prog.dims_s, c->prog.itab.data(), qpc_dtab{c->prog.dtab.data()}, Lds{0}, io);
    races += emu.races; uninit += emu.uninit; outside += emu.outside;
    if (first.empty()) first = emu.first;
  }
  g_emu = nullptr;
  std::lock_guard<std::mutex> lk(g_rep_mu);
  g_races += races; g_uninit += uninit; g_outside += outside;
  if (g_first.empty()) g_first = first;
}

// CPU backend of the host pipeline (qpc_pipeline.h) with any number of fake devices: the pipeline's shard / chunk
// arithmetic, worker hand-off and error propagation run exactly as in the product, the "device" is the emulator
class EmuBackend : public QpcChunkBackend {
 public:
  EmuBackend(qpc_controller *c, int ndev) : c_(c), ndev_(ndev) {}
  int n_devices() const override { return ndev_; }
  void bind_thread(int) override {}
  int run_chunk(int d, int slot, const QpcHostJob &job, int64_t lo, int64_t n, std::string &err) override {
    if (fail_at >= 0 && lo <= fail_at && fail_at < lo + n) { err = "injected failure"; return QPC_ERR_HIP; } // test hook
    {
      std::lock_guard<std::mutex> lk(mu);
      chunks.push_back({d, slot, lo, n});
    }
    emu_ticks(c_, job, lo, n);
    return QPC_OK;
  }
  struct Rec { int dev, slot; int64_t lo, n; };
  std::mutex mu;
  std::vector<Rec> chunks;
  int64_t fail_at = -1;

 private:
  qpc_controller *c_;
  int ndev_;
};
struct QpcRuntime {
  EmuBackend *backend = nullptr;
  QpcHostPipeline *pipe = nullptr;
};

static int fill_job(qpc_controller *c, QpcHostJob &job, int64_t B, const double *q, const double *v, const double *desired,
                    const double *cw, const double *cmax, const double *tw, const double *surf, double *tau, double *vdot,
                    double *cwrench, int32_t *status, int32_t *iters, uint64_t *wsets) {
  if (!c || !c->finalized || !c->rt) return QPC_ERR_NOT_FINALIZED;
  const QpcDims &D = c->prog.dims;
  job.B = B;
  const double *hin[QpcHostJob::N_IN] = {q, v, desired, cw, cmax, tw, surf};
  const size_t nin[QpcHostJob::N_IN] = {(size_t)D.nq, (size_t)D.nv, (size_t)D.n_desired, (size_t)D.nc, (size_t)D.nc, (size_t)D.n_tw, (size_t)D.n_surf};
  for (int i = 0; i < QpcHostJob::N_IN; i++) { job.in[i] = nin[i] ? hin[i] : nullptr; job.in_dim[i] = nin[i]; }
  double *hout[QpcHostJob::N_OUT] = {tau, vdot, D.nc ? cwrench : nullptr};
  const size_t nout[QpcHostJob::N_OUT] = {(size_t)D.nv, (size_t)D.nv, (size_t)6 * D.nc};
  for (int i = 0; i < QpcHostJob::N_OUT; i++) { job.out[i] = hout[i]; job.out_dim[i] = nout[i]; }
  job.status = status; job.iters = iters; job.working_set = wsets;
  return QPC_OK;
}

extern "C" {

int32_t qpc_finalize_devices(qpc_controller *c, int32_t n_devices, const int32_t *device_ids, qpc_layout *out) {
  if (!c || n_devices < 1 || !device_ids) return QPC_ERR_INVALID_ARGUMENT;
  if (!c->finalized) {
#ifdef QPC_F32
    if (c->precision != QPC_PRECISION_F32) { qpc_set_error("libqpc_emul_f32.so emulates the fp32 tick: call qpc_set_precision(ctrl, QPC_PRECISION_F32) first"); return QPC_ERR_UNSUPPORTED; }
#else
    if (c->precision != QPC_PRECISION_F64) { qpc_set_error("libqpc_emul.so emulates the fp64 tick (the fp32 variant is libqpc_emul_f32.so)"); return QPC_ERR_UNSUPPORTED; }
#endif
    int rc = qpc_compile_program(c);
    if (rc != QPC_OK) return rc;
    c->rt = new QpcRuntime();
    c->rt->backend = new EmuBackend(c, n_devices);
    c->rt->pipe = new QpcHostPipeline(c->rt->backend, c->pipe_slots > 0 ? c->pipe_slots : 3, c->pipe_chunk > 0 ? c->pipe_chunk : 8192,
                                      c->pipe_inline >= 0 ? c->pipe_inline : 2048);
    c->finalized = true;
  }
  if (out) *out = c->layout;
  return QPC_OK;
}
int32_t qpc_finalize(qpc_controller *c, int32_t device_id, qpc_layout *out) { return qpc_finalize_devices(c, 1, &device_id, out); }
int32_t qpc_n_devices(const qpc_controller *c) { return (c && c->rt) ? c->rt->backend->n_devices() : 0; }
void qpc_controller_destroy(qpc_controller *c) {
  if (!c) return;
  if (c->rt) { delete c->rt->pipe; delete c->rt->backend; delete c->rt; }
  delete c;
}
int32_t qpc_set_host_pipeline(qpc_controller *c, int64_t chunk_qps, int32_t slots_per_device, int64_t inline_qps) {
  if (!c) return QPC_ERR_INVALID_ARGUMENT;
  if (slots_per_device > 0) { if (c->finalized) return QPC_ERR_ALREADY_FINALIZED; c->pipe_slots = slots_per_device; }
  if (chunk_qps > 0) c->pipe_chunk = chunk_qps;
  if (inline_qps >= 0) c->pipe_inline = inline_qps;
  if (c->rt) {
    std::string e;
    (void)c->rt->pipe->wait(e);
    if (chunk_qps > 0) c->rt->pipe->set_chunk_qps(chunk_qps);
    if (inline_qps >= 0) c->rt->pipe->set_inline_qps(inline_qps);
  }
  return QPC_OK;
}

static int run_job(qpc_controller *c, const QpcHostJob &job) {
  if (job.B == 0) return QPC_OK;
  std::string err;
  const int rc = c->rt->pipe->run(job, err);
  if (rc != QPC_OK) qpc_set_error(err);
  return rc;
}

int32_t qpc_control_batch(qpc_controller *c, int64_t B, const double *q, const double *v, const double *desired,
                          const double *cw, const double *cmax, const double *tw, const double *surf, double *tau,
                          double *vdot, double *cwrench, int32_t *status, int32_t *iters) {
  QpcHostJob job;
  int rc = fill_job(c, job, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr);
  return rc != QPC_OK ? rc : run_job(c, job);
}
// the device-pointer entry of a multi-device controller, emulated: "device" memory is host memory, the shard is ticked on the caller's
// thread (bench.py's in-library leg runs through this on emulated devices)
int32_t qpc_control_batch_device_on(qpc_controller *c, int32_t device_index, int64_t B, const double *q, const double *v, const double *desired,
                                    const double *cw, const double *cmax, const double *tw, const double *surf, double *tau, double *vdot,
                                    double *cwrench, int32_t *status, int32_t *iters, void *) {
  if (!c || device_index < 0 || device_index >= qpc_n_devices(c)) { qpc_set_error("qpc_control_batch_device_on: device index out of range"); return QPC_ERR_INVALID_ARGUMENT; }
  QpcHostJob job;
  int rc = fill_job(c, job, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr);
  if (rc != QPC_OK) return rc;
  emu_ticks(c, job, 0, B);
  return QPC_OK;
}
int32_t qpc_contact_wrench_per_body_device(qpc_controller *c, int32_t device_index, int64_t B, const double *cw, double *bw, void *) {
  if (!c || device_index < 0 || device_index >= qpc_n_devices(c)) return QPC_ERR_INVALID_ARGUMENT;
  return qpc_contact_wrench_per_body(c, B, cw, bw);
}
int32_t qpc_control_batch_warm(qpc_controller *c, int64_t B, const double *q, const double *v, const double *desired,
                               const double *cw, const double *cmax, const double *tw, const double *surf, double *tau,
                               double *vdot, double *cwrench, int32_t *status, int32_t *iters, uint64_t *working_set) {
  if (!working_set) return QPC_ERR_INVALID_ARGUMENT;
  if (c && !c->warm_start) { qpc_set_error("qpc_control_batch_warm: call qpc_enable_warm_start before qpc_finalize"); return QPC_ERR_UNSUPPORTED; }
  QpcHostJob job;
  int rc = fill_job(c, job, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, working_set);
  return rc != QPC_OK ? rc : run_job(c, job);
}
int32_t qpc_control_batch_async(qpc_controller *c, int64_t B, const double *q, const double *v, const double *desired,
                                const double *cw, const double *cmax, const double *tw, const double *surf, double *tau,
                                double *vdot, double *cwrench, int32_t *status, int32_t *iters) {
  QpcHostJob job;
  int rc = fill_job(c, job, B, q, v, desired, cw, cmax, tw, surf, tau, vdot, cwrench, status, iters, nullptr);
  if (rc != QPC_OK) return rc;
  if (B > 0) c->rt->pipe->submit(job);
  return QPC_OK;
}
int32_t qpc_wait(qpc_controller *c) {
  if (!c || !c->finalized || !c->rt) return QPC_ERR_NOT_FINALIZED;
  std::string err;
  const int rc = c->rt->pipe->wait(err);
  if (rc != QPC_OK) qpc_set_error(err);
  return rc;
}
int32_t qpc_last_async_error(qpc_controller *c, char *message, int32_t message_bytes) {
  if (!c || !c->finalized || !c->rt) return QPC_ERR_NOT_FINALIZED;
  std::string msg;
  const int rc = c->rt->pipe->take_dropped(msg);
  if (message && message_bytes > 0) snprintf(message, (size_t)message_bytes, "%s", msg.c_str());
  return rc;
}
// test hooks of the fake-device backend: which (device, slot) ran which chunk; inject a failure into the chunk holding QP b
int32_t emu_chunk_log(qpc_controller *c, int64_t *out, int32_t max_rows) {
  std::lock_guard<std::mutex> lk(c->rt->backend->mu);
  int n = 0;
  for (auto &r : c->rt->backend->chunks) {
    if (n < max_rows) { out[4 * n] = r.dev; out[4 * n + 1] = r.slot; out[4 * n + 2] = r.lo; out[4 * n + 3] = r.n; }
    n++;
  }
  c->rt->backend->chunks.clear();
  return n;
}
void emu_fail_at(qpc_controller *c, int64_t b) { c->rt->backend->fail_at = b; }

#ifndef QPC_F32
// L1 (plain serial code per QP, like L0)
int32_t qpc_assemble_solve_batch(qpc_controller *c, int64_t B, const double *task_J, const double *task_bias,
                                 const double *desired, const double *momentum_A, const double *momentum_rhs,
                                 const double *contact_R, const double *contact_p, const double *cw, const double *cmax,
                                 const double *tw, const double *surf, double *vdot, double *cwrench, int32_t *status,
                                 int32_t *iters) {
  if (!c || !c->finalized) return QPC_ERR_NOT_FINALIZED;
  const int nv = c->model->nv, nd = c->n_desired, nc = (int)c->contacts.size(), fl = c->floating >= 0;
  std::vector<int> it;
  std::vector<double> dt;
  int n = 0, m = 0;
  qpc_l1_build_tables(c, it, dt, n, m);
  std::vector<double> ws(qpc::l1_workspace_doubles(n, m)), wr(6 * (nc > 0 ? nc : 1));
  qpc::L1Prog pg{it.data(), dt.data()};
  for (int64_t b = 0; b < B; b++) {
    int32_t itn = 0;
    qpc::l1_assemble_solve(pg, task_J + b * nd * nv, task_bias + b * nd, desired + b * nd, fl ? momentum_A + b * 6 * nv : nullptr,
                           fl ? momentum_rhs + b * 6 : nullptr, contact_R + b * 9 * nc, contact_p + b * 3 * nc, cw + b * nc,
                           cmax + b * nc, tw ? tw + b * c->n_tw : nullptr, surf ? surf + b * c->n_surf : nullptr, c->settings,
                           ws.data(), vdot + b * nv, wr.data(), status + b, &itn);
    if (cwrench) for (int i = 0; i < 6 * nc; i++) cwrench[b * 6 * nc + i] = wr[i];
    if (iters) iters[b] = itn;
  }
  return QPC_OK;
}

static long g_l0_wg = 0; // QPs the workgroup form solved (test hook: emu_l0_wg_count)
// L0 (qpc_qp_dense.h is plain serial code per QP: the emulation is the same function on the host)
int32_t qpc_qp_solve_batch(int32_t n, int32_t m, int64_t B, const double *P, const double *q, const double *A,
                           const double *l, const double *u, const qpc_settings *settings, int32_t device_id,
                           double *x, double *y, int32_t *status, int32_t *iters) {
  (void)device_id;
  qpc_settings st;
  if (settings) st = *settings; else qpc_default_settings(&st);
  // like the device: the workgroup form first (qpc_qp_wg.h, here one serial "thread"), the one-lane form on what it leaves
  std::vector<double> ws(qpc::qp_dense_workspace_doubles(n, m)), lds(qpc::qp_wg_lds_doubles(n, m));
  const char *one = getenv("QPC_L0_ONE_LANE");
  const char *cap = getenv("QPC_EMU_WG_LDS"); // test hook: a smaller "LDS" sends the big matrix of the workgroup form to its global workspace
  const int lds_cap = cap ? atoi(cap) : (int)lds.size();
  for (int64_t b = 0; b < B; b++) {
    int s = QPC_WG_FALLBACK;
    if (m > 0 && !(one && one[0] == '1'))
      s = qpc::qp_solve_wg(n, m, P + b * n * n, q + b * n, A + b * m * n, l + b * m, u + b * m, st,
                           ws.data(), lds.data(), lds_cap, x + b * n, y + b * m, iters ? iters + b : nullptr);
    g_l0_wg += s != QPC_WG_FALLBACK;
    if (s != QPC_WG_FALLBACK) status[b] = s;
    else qpc::qp_solve_dense(n, m, P + b * n * n, q + b * n, A + b * m * n, l + b * m, u + b * m, st, ws.data(), x + b * n,
                             y + b * m, status + b, iters ? iters + b : nullptr);
  }
  return QPC_OK;
}

// f-2: simulation step / rollout through the lane-serial emulation of qpc_sim.h
int32_t qpc_sim_step_batch(qpc_controller *c, int64_t B, double *q, double *v, const double *tau, const int32_t *status,
                           const qpc_sim_params *params, int32_t *bad) {
  if (!c || !c->finalized) return QPC_ERR_NOT_FINALIZED;
  if (!q || !v || !tau || !params || !(params->dt > 0.0)) return QPC_ERR_INVALID_ARGUMENT;
  const QpcDims &D = c->prog.dims;
  EmuState emu;
  g_emu = &emu;
  // like the device (qpc_jit.cpp qpc_jit_sim_rpw): two robots per wavefront when a robot fits half a wave
  const char *e1 = getenv("QPC_SIM_RPW");
  const int rpw = (D.nb <= 32 && D.nc <= 32 && !(e1 && e1[0] == '1')) ? 2 : 1;
  const int slab = (qpc_sim_layout(D).total > D.lds_doubles ? qpc_sim_layout(D).total : D.lds_doubles) + 128 * 8; // + reduction scratch
  for (int64_t b0 = 0; b0 < B; b0 += rpw) {
    emu.reset(slab * rpw);
    qpc::SimIO io[2];
    for (int r = 0; r < rpw; r++) {
      const int64_t b = b0 + r < B ? b0 + r : B - 1;
      io[r].q = q + b * D.nq; io[r].v = v + b * D.nv; io[r].tau = tau + b * D.nv; io[r].status = status ? status + b : nullptr;
      io[r].q_out = q + b * D.nq; io[r].v_out = v + b * D.nv; io[r].bad = (bad && b0 + r < B) ? bad + b : nullptr;
    }
    if (rpw == 2) qpc::qpc_sim_step<Lds, 2>(D, c->prog.itab.data(), c->prog.dtab.data(), Lds{0}, io, *params, slab);
    else qpc::qpc_sim_step<Lds, 1>(D, c->prog.itab.data(), c->prog.dtab.data(), Lds{0}, io, *params, 0);
    g_races += emu.races; g_uninit += emu.uninit; g_outside += emu.outside;
    if (g_first.empty()) g_first = emu.first;
  }
  g_emu = nullptr;
  return QPC_OK;
}

int32_t qpc_rollout_batch(qpc_controller *c, int64_t B, int64_t n_steps, double *q, double *v, const double *desired,
                          const double *cw, const double *cmax, const double *tw, const double *surf,
                          const qpc_sim_params *params, double *tau, int32_t *status, int32_t *bad) {
  if (!c || !c->finalized) return QPC_ERR_NOT_FINALIZED;
  const QpcDims &D = c->prog.dims;
  std::vector<double> t(B * D.nv), vd(B * D.nv);
  std::vector<int32_t> st(B);
  for (int64_t s = 0; s < n_steps; s++) {
    int rc = qpc_control_batch(c, B, q, v, desired, cw, cmax, tw, surf, t.data(), vd.data(), nullptr, st.data(), nullptr);
    if (rc != QPC_OK) return rc;
    rc = qpc_sim_step_batch(c, B, q, v, t.data(), st.data(), params, bad);
    if (rc != QPC_OK) return rc;
  }
  if (tau) memcpy(tau, t.data(), sizeof(double) * t.size());
  if (status) memcpy(status, st.data(), sizeof(int32_t) * st.size());
  return QPC_OK;
}

#else // QPC_F32: the simulation step and the dense L0 / L1 solvers are fp64 code, not part of the fp32 emulation
static int unsupported_f32(const char *what) { qpc_set_error(std::string(what) + ": fp64 only (this is the emulation of the fp32 tick)"); return QPC_ERR_UNSUPPORTED; }
int32_t qpc_assemble_solve_batch(qpc_controller *, int64_t, const double *, const double *, const double *, const double *, const double *,
                                 const double *, const double *, const double *, const double *, const double *, const double *, double *,
                                 double *, int32_t *, int32_t *) { return unsupported_f32("qpc_assemble_solve_batch"); }
int32_t qpc_qp_solve_batch(int32_t, int32_t, int64_t, const double *, const double *, const double *, const double *, const double *,
                           const qpc_settings *, int32_t, double *, double *, int32_t *, int32_t *) { return unsupported_f32("qpc_qp_solve_batch"); }
int32_t qpc_sim_step_batch(qpc_controller *, int64_t, double *, double *, const double *, const int32_t *, const qpc_sim_params *, int32_t *) {
  return unsupported_f32("qpc_sim_step_batch");
}
int32_t qpc_rollout_batch(qpc_controller *, int64_t, int64_t, double *, double *, const double *, const double *, const double *, const double *,
                          const double *, const qpc_sim_params *, double *, int32_t *, int32_t *) { return unsupported_f32("qpc_rollout_batch"); }
#endif // QPC_F32

#ifndef QPC_F32
// L0 with the optimizer's life cycle (qpc_qp_setup / qpc_qp_solve): the host tables of qpc_qp_prep.h, the same kernel source
struct qpc_qp_problem {
  qpc::QpPrepHost H;
  qpc::QpSpPlanHost plan; // the sparse workgroup solver's plan for the set of equality rows of the last batch's first QP
  bool plan_tried = false;
  std::vector<double> yprev;
  int64_t B_prev = 0;
};
static long g_l0_sp = 0; // QPs solved by the sparse workgroup form since the last emu_l0_sp_count()
static void emu_eq_rows(int m, const double *l, const double *u, std::vector<uint8_t> &eq) {
  eq.resize(m);
  for (int i = 0; i < m; i++) eq[i] = (l[i] > -1e19 && u[i] < 1e19 && u[i] - l[i] <= 1e-12 * (1.0 + std::fabs(l[i]))) ? 1 : 0;
}
int32_t qpc_qp_setup(int32_t n, int32_t m, int64_t nnz_P, const int32_t *P_rows, const int32_t *P_cols, int64_t nnz_A,
                     const int32_t *A_rows, const int32_t *A_cols, int32_t device_id, qpc_qp_problem **out) {
  (void)device_id;
  if (!out) return QPC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  qpc_qp_problem *p = new qpc_qp_problem;
  const std::string err = qpc::qp_prep_build(p->H, n, m, nnz_P, P_rows, P_cols, nnz_A, A_rows, A_cols);
  if (!err.empty()) { delete p; qpc_set_error("qpc_qp_setup: " + err); return QPC_ERR_INVALID_ARGUMENT; }
  *out = p;
  return QPC_OK;
}
void qpc_qp_free(qpc_qp_problem *p) { delete p; }
int32_t qpc_qp_analyze(qpc_qp_problem *p, const double *l, const double *u) {
  if (!p || (p->H.m > 0 && (!l || !u))) return QPC_ERR_INVALID_ARGUMENT;
  if (p->H.m <= 0) return QPC_OK;
  std::vector<uint8_t> eq;
  emu_eq_rows(p->H.m, l, u, eq);
  if (!p->plan_tried || eq != p->plan.eq) { qpc::qp_sp_plan_build(p->H, eq.data(), p->plan); p->plan_tried = true; if (!p->plan.valid) p->plan.eq = eq; }
  return QPC_OK;
}
// (the emulation's "device" memory is host memory: the device-pointer entry is the host entry; the stream is ignored)
int32_t qpc_qp_solve(qpc_qp_problem *p, int64_t B, const double *P_vals, const double *q, const double *A_vals, const double *l,
                     const double *u, const qpc_settings *settings, int32_t warm_start, double *x, double *y, int32_t *status,
                     int32_t *iters);
int32_t qpc_qp_solve_device(qpc_qp_problem *p, int64_t B, const double *P_vals, const double *q, const double *A_vals, const double *l,
                            const double *u, const qpc_settings *settings, int32_t warm_start, double *x, double *y, int32_t *status,
                            int32_t *iters, void *hip_stream) {
  (void)hip_stream;
  return qpc_qp_solve(p, B, P_vals, q, A_vals, l, u, settings, warm_start, x, y, status, iters);
}
// the plan's int array (diagnostics of the tests): returns its length, copies at most n ints
int emu_l0_plan_ints(qpc_qp_problem *p, int *out, int n) {
  if (!p || !p->plan_tried || !p->plan.valid) return 0;
  for (int i = 0; i < n && i < (int)p->plan.ints.size(); i++) out[i] = p->plan.ints[i];
  return (int)p->plan.ints.size();
}
int32_t qpc_qp_enable_timing(qpc_qp_problem *, int32_t) { return QPC_OK; }
int32_t qpc_qp_last_kernel_ms(const qpc_qp_problem *, double *ms) { if (ms) *ms = 0.0; return QPC_OK; }
int32_t qpc_qp_plan_info(const qpc_qp_problem *p, int64_t *out, int32_t n_out) {
  if (!p || !out || n_out < 0) return QPC_ERR_INVALID_ARGUMENT;
  const qpc::QpSpPlanHost &P = p->plan;
  bool on = p->plan_tried && P.valid;
  long long dense = 0;
  for (int j = 0; j < P.ne; j++) dense += (long long)(P.m + 1 - j) * (P.m + 2 - j) / 2;
  int kw = P.mi; // (like the runtime, qpc_l0.hip qpc_qp_plan: the working set's factor takes what is left of the 160 KB, and a plan that leaves
                 //  room for fewer than 32 working rows -- or none -- is not used)
  while (on && kw > 0 && qpc::qp_sp_lds_doubles(P.ints.data(), kw) > QPC_SP_LDS_CAP_DOUBLES) kw--;
  if (on && (qpc::qp_sp_lds_doubles(P.ints.data(), kw) > QPC_SP_LDS_CAP_DOUBLES || (kw < P.mi && kw < 32))) on = false;
  const int64_t v[12] = {on ? 1 : 0, P.ne, P.mi, P.nfront, P.nlevel, P.njob, P.nval, on ? (int64_t)qpc::qp_sp_lds_doubles(P.ints.data(), kw) * 8 : 0, P.maxw, P.maxr, P.flops, dense};
  for (int i = 0; i < 12 && i < n_out; i++) out[i] = v[i];
  return QPC_OK;
}
int32_t qpc_qp_solve(qpc_qp_problem *p, int64_t B, const double *P_vals, const double *q, const double *A_vals, const double *l,
                     const double *u, const qpc_settings *settings, int32_t warm_start, double *x, double *y, int32_t *status,
                     int32_t *iters) {
  if (!p || B < 0) return QPC_ERR_INVALID_ARGUMENT;
  qpc_settings st;
  if (settings) st = *settings; else qpc_default_settings(&st);
  const int n = p->H.n, m = p->H.m;
  const long long nnzP = p->H.nnzP, nnzA = p->H.nnzA;
  std::vector<double> ws(qpc::qp_dense_workspace_doubles(n, m > 0 ? m : 1)), lds(qpc::qp_wg_lds_doubles(n, m));
  const char *cap = getenv("QPC_EMU_WG_LDS");
  const int lds_cap = cap ? atoi(cap) : (int)lds.size();
  const bool warm = warm_start && p->B_prev >= B && m > 0;
  const qpc::QpPrepView V = p->H.view(p->H.ints.data());
  std::vector<double> ynew((size_t)B * m);
  // the sparse workgroup form (qpc_qp_sp.h) first: its plan is made for the equality rows of the batch's first QP
  const char *nosp = getenv("QPC_L0_NO_SPARSE");
  const bool try_sp = m > 0 && B > 0 && !(nosp && nosp[0] == '1');
  if (try_sp) {
    std::vector<uint8_t> eq;
    emu_eq_rows(m, l, u, eq);
    if (!p->plan_tried || eq != p->plan.eq) { qpc::qp_sp_plan_build(p->H, eq.data(), p->plan); p->plan_tried = true; if (!p->plan.valid) p->plan.eq = eq; }
  }
  const int wg_total = cap ? atoi(cap) : QPC_WG_LDS_DOUBLES; // doubles of "LDS" (the device's 160 KB)
  std::vector<double> sp_lds;
  int sp_cap = 0;
  bool sp_fits = try_sp && p->plan.valid;
  if (sp_fits && !cap) { // the runtime's rule (qpc_l0.hip qpc_qp_plan): room for at least 32 working rows, or for all of them
    int kw = p->plan.mi;
    while (kw > 0 && qpc::qp_sp_lds_doubles(p->plan.ints.data(), kw) > QPC_SP_LDS_CAP_DOUBLES) kw--;
    if (qpc::qp_sp_lds_doubles(p->plan.ints.data(), kw) > QPC_SP_LDS_CAP_DOUBLES || (kw < p->plan.mi && kw < 32)) sp_fits = false;
  }
  if (sp_fits) { sp_cap = (cap ? wg_total : QPC_SP_LDS_CAP_DOUBLES) - (p->plan.lds_ints + 1) / 2; // (the runtime's cap)
    if (sp_cap > 0) sp_lds.assign((size_t)sp_cap, 0.0); }
  for (int64_t b = 0; b < B; b++) {
    int s = QPC_WG_FALLBACK;
    if (sp_cap > 0) {
      s = qpc::qp_solve_sp(p->plan.ints.data(), p->plan.ints.data(), P_vals + b * nnzP, q + b * n, A_vals + b * nnzA, l + b * m, u + b * m, st,
                           sp_lds.data(), sp_cap, warm ? p->yprev.data() + b * m : nullptr, x + b * n, y + b * m, iters ? iters + b : nullptr,
                           nullptr);
      g_l0_sp += s != QPC_WG_FALLBACK;
    }
    if (s == QPC_WG_FALLBACK && m > 0 && p->H.wg_tables)
      s = qpc::qp_solve_wg_prep(V, P_vals + b * nnzP, q + b * n, A_vals + b * nnzA, l + b * m, u + b * m, st, ws.data(), lds.data(), lds_cap,
                                warm ? p->yprev.data() + b * m : nullptr, x + b * n, y + b * m, iters ? iters + b : nullptr);
    g_l0_wg += s != QPC_WG_FALLBACK;
    if (s != QPC_WG_FALLBACK) status[b] = s;
    else { // the dense path
      std::vector<double> P((size_t)n * n, 0.0), A((size_t)(m > 0 ? m : 1) * n, 0.0);
      for (long long k = 0; k < nnzP; k++) { P[(size_t)p->H.Prow[k] * n + p->H.Pcol[k]] = P_vals[b * nnzP + k]; P[(size_t)p->H.Pcol[k] * n + p->H.Prow[k]] = P_vals[b * nnzP + k]; }
      for (long long k = 0; k < nnzA; k++) A[(size_t)p->H.Arow[k] * n + p->H.Acol[k]] = A_vals[b * nnzA + k];
      int32_t s2 = 0;
      qpc_qp_solve_batch(n, m, 1, P.data(), q + b * n, A.data(), l + b * m, u + b * m, &st, 0, x + b * n, y + b * m, &s2, iters ? iters + b : nullptr);
      status[b] = s2;
    }
    for (int i = 0; i < m; i++) ynew[(size_t)b * m + i] = y[b * m + i];
  }
  p->yprev.swap(ynew);
  p->B_prev = B;
  return QPC_OK;
}
long emu_l0_wg_count() { const long r = g_l0_wg; g_l0_wg = 0; return r; }
long emu_l0_sp_count() { const long r = g_l0_sp; g_l0_sp = 0; return r; }
#endif
// race-detector report: counts since the last call
void emu_report(long *races, long *uninit, long *outside, char *first, int n) {
  std::lock_guard<std::mutex> lk(g_rep_mu);
  *races = g_races; *uninit = g_uninit; *outside = g_outside;
  snprintf(first, n, "%s", g_first.c_str());
  g_races = g_uninit = g_outside = 0; g_first.clear();
}
}

// self-test of the race detector: lane l reads the word lane l-1 writes in the same phase
extern "C" long emu_selftest_race() {
  EmuState emu;
  g_emu = &emu;
  emu.reset(128);
  Lds sm{0};
  QPC_PHASE(lane) { sm[lane] = 1.0; } QPC_END_PHASE
  QPC_PHASE(lane) {
    real x = sm[(lane + 63) % 64];
    sm[lane] = x + 1.0;
  } QPC_END_PHASE
  long r = emu.races;
  g_emu = nullptr;
  return r;
}
