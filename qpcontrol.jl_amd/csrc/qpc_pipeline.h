// qpc_pipeline.h -- host side of the host-pointer entry points (qpc_control_batch[_async], qpc_wait):
// what turns ONE call with host matrices -- the shape of the reference's callable
// (controller)(tau, t, x), momentum.jl:41-81, for a batch -- into a pipelined, multi-device job.
//
//   * the batch is split into contiguous shards, one per device of the controller (SURVEY.md 8e:
//     QP index range [g B/G, (g+1) B/G) -> device g; no collective, the QPs are independent);
//   * every shard is cut into chunks; each device has `slots` worker threads, each owning one
//     pipeline slot (stream, staging buffers, ticket counter) of that device.  A worker takes the next
//     chunk of its device, stages the inputs, launches the tick kernel, stages the outputs back and
//     takes the next one: with several slots per device the H2D copy of one chunk, the kernel of
//     another and the D2H copy of a third overlap, and the tail of one kernel is filled by the
//     workgroups of the next;
//   * the call returns as soon as the job is posted (async) or when every chunk is done (sync / wait).
//
// Backend-agnostic on purpose: the product's backend is HIP (qpc_runtime.hip); the test emulator
// (tests/emul/emul.cpp) plugs in a CPU backend with several fake devices, so that shard / chunk
// arithmetic, thread hand-off and error propagation are tested without a GPU.  Pure C++17, no HIP.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// one host-pointer job: the arguments of qpc_control_batch with the per-QP record sizes
struct QpcHostJob {
  int64_t B = 0;
  enum { IN_Q, IN_V, IN_DES, IN_CW, IN_CMAX, IN_TW, IN_SURF, N_IN };
  enum { OUT_TAU, OUT_VDOT, OUT_CWRENCH, N_OUT };
  const double *in[N_IN] = {};
  size_t in_dim[N_IN] = {}; // doubles per QP (0: array absent)
  double *out[N_OUT] = {};
  size_t out_dim[N_OUT] = {};
  int32_t *status = nullptr, *iters = nullptr;
  uint64_t *working_set = nullptr; // [2 * B] warm-start masks, in/out (warm-start controllers only)
  bool in_pinned[N_IN] = {}, out_pinned[N_OUT] = {}, status_pinned = false, iters_pinned = false, ws_pinned = false;
};

// what a device backend has to provide
class QpcChunkBackend {
 public:
  virtual ~QpcChunkBackend() {}
  virtual int n_devices() const = 0;
  // once per worker thread before its first chunk (e.g. hipSetDevice)
  virtual void bind_thread(int device_index) = 0;
  // QPs [lo, lo + n) of `job` on pipeline slot `slot` of device `device_index`: stage in, run, stage out; returns when
  // the outputs are in the caller's arrays.  Returns QPC_OK or an error code with a message in `err`.
  virtual int run_chunk(int device_index, int slot, const QpcHostJob &job, int64_t lo, int64_t n, std::string &err) = 0;
};

// contiguous shard [lo, hi) of `B` items owned by part `g` of `G` (sizes differ by at most one)
inline void qpc_shard_bounds(int64_t B, int g, int G, int64_t &lo, int64_t &hi) {
  const int64_t base = B / G, rem = B % G;
  lo = g * base + (g < rem ? g : rem);
  hi = lo + base + (g < rem ? 1 : 0);
}

class QpcHostPipeline {
 public:
  // slots worker threads per device; the extra slot index `slots` of device 0 is reserved for the
  // caller's thread (small synchronous jobs run inline: no thread hand-off on the latency path)
  QpcHostPipeline(QpcChunkBackend *backend, int slots, int64_t chunk_qps, int64_t inline_qps)
      : be_(backend), slots_(slots < 1 ? 1 : slots), chunk_(chunk_qps < 1 ? 1 : chunk_qps), inline_(inline_qps) {
    const int nd = be_->n_devices();
    next_.reset(new std::atomic<int64_t>[nd]);
    for (int d = 0; d < nd; d++) next_[d].store(0);
    for (int d = 0; d < nd; d++)
      for (int s = 0; s < slots_; s++) threads_.emplace_back([this, d, s] { worker(d, s); });
  }
  ~QpcHostPipeline() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      done_cv_.wait(lk, [this] { return active_ == 0; });
      quit_ = true;
    }
    work_cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
  int slots() const { return slots_; }
  int inline_slot() const { return slots_; }
  int64_t chunk_qps() const { return chunk_; }
  void set_chunk_qps(int64_t c) { std::lock_guard<std::mutex> lk(mu_); chunk_ = c < 1 ? 1 : c; }
  void set_inline_qps(int64_t n) { std::lock_guard<std::mutex> lk(mu_); inline_ = n; }

  // post a job (waits for the previous one first: one job in flight per controller).  The arrays of
  // `job` must stay valid until wait() returns.
  int submit(const QpcHostJob &job) {
    std::string e;
    int rc = wait(e);
    if (rc) note_dropped(rc, e); // the error of an earlier job was that job's: reported by its own wait, or -- never waited for -- parked
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = job;
      rc_ = 0; err_.clear();
      const int nd = be_->n_devices();
      for (int d = 0; d < nd; d++) next_[d].store(0);
      abort_.store(false);
      active_ = (int)threads_.size();
      generation_++;
    }
    work_cv_.notify_all();
    return 0;
  }
  // block until the posted job (if any) is complete; returns its first error
  int wait(std::string &err) {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return active_ == 0; });
    err = err_;
    const int rc = rc_;
    rc_ = 0; err_.clear();
    return rc;
  }
  // synchronous run; small jobs on the caller's thread
  int run(const QpcHostJob &job, std::string &err) {
    if (job.B <= inline_) {
      { const int rd = wait(err); if (rd) note_dropped(rd, err); } // an async job nobody waited for: its error was that job's (see submit), this job still runs
      err.clear();
      int rc = 0;
      be_->bind_thread(0);
      for (int64_t lo = 0; lo < job.B; lo += chunk_) {
        const int64_t n = job.B - lo < chunk_ ? job.B - lo : chunk_;
        rc = be_->run_chunk(0, slots_, job, lo, n, err);
        if (rc) return rc;
      }
      return 0;
    }
    submit(job);
    return wait(err);
  }

  // The error of an async job that nobody waited for is not the next call's error (that call reports its own batch), but it must
  // not vanish either: it is parked here until take_dropped reads it (qpc_last_async_error), and announced once on stderr.
  void note_dropped(int rc, const std::string &msg) {
    std::lock_guard<std::mutex> lk(drop_mu_);
    dropped_rc_ = rc; dropped_msg_ = msg;
    fprintf(stderr, "qpcontrol_hip: an asynchronous batch failed and nobody called qpc_wait for it (code %d: %s); its outputs are "
                    "not valid -- see qpc_last_async_error\n", rc, msg.c_str());
  }
  int take_dropped(std::string &msg) {
    std::lock_guard<std::mutex> lk(drop_mu_);
    const int rc = dropped_rc_;
    msg = dropped_msg_;
    dropped_rc_ = 0; dropped_msg_.clear();
    return rc;
  }

 private:
  std::mutex drop_mu_;
  int dropped_rc_ = 0;
  std::string dropped_msg_;
  void worker(int d, int s) {
    bool bound = false;
    uint64_t seen = 0;
    while (true) {
      QpcHostJob job;
      int64_t chunk;
      {
        std::unique_lock<std::mutex> lk(mu_);
        work_cv_.wait(lk, [&] { return quit_ || generation_ != seen; });
        if (quit_) return;
        seen = generation_;
        job = job_;
        chunk = chunk_;
      }
      if (!bound) { be_->bind_thread(d); bound = true; }
      int64_t lo, hi;
      qpc_shard_bounds(job.B, d, be_->n_devices(), lo, hi);
      int rc = 0;
      std::string err;
      while (true) {
        if (abort_.load()) break; // a chunk of this job failed somewhere: stop taking new ones
        const int64_t k = next_[d].fetch_add(1);
        if (k > (hi - lo) / chunk) break;
        const int64_t c0 = lo + k * chunk;
        if (c0 >= hi) break;
        const int64_t n = hi - c0 < chunk ? hi - c0 : chunk;
        rc = be_->run_chunk(d, s, job, c0, n, err);
        if (rc) { abort_.store(true); break; }
      }
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (rc && !rc_) { rc_ = rc; err_ = err; }
        if (--active_ == 0) done_cv_.notify_all();
      }
    }
  }

  QpcChunkBackend *be_;
  int slots_;
  int64_t chunk_, inline_;
  std::vector<std::thread> threads_;
  std::unique_ptr<std::atomic<int64_t>[]> next_;
  std::atomic<bool> abort_{false};
  std::mutex mu_;
  std::condition_variable work_cv_, done_cv_;
  QpcHostJob job_;
  uint64_t generation_ = 0;
  int active_ = 0;
  bool quit_ = false;
  int rc_ = 0;
  std::string err_;
};
