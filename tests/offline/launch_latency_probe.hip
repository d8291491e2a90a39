// launch_latency_probe.hip -- where the ~45 us between the L0 kernel's 72 us and the 117 us of a B = 1 qpc_qp_solve call go.
// Times, per call form, the median over 2 000 calls of a kernel that does ~nothing (one workgroup of 512 threads):
//   a  launch + hipStreamSynchronize
//   b  launch + the host spinning on a flag in page-locked memory the kernel stores to (system scope) -- no synchronize call
//   c  hipMemcpyAsync H2D 15 KB + launch + hipMemcpyAsync D2H 3 KB + hipStreamSynchronize            (the call as it is today)
//   d  launch of a kernel that READS 15 KB from page-locked host memory and WRITES 3 KB + flag to it, host spins  (no copies, no synchronize)
//   e  as d but hipStreamSynchronize instead of the spin
//   f  launch + polling hipStreamQuery until it is no longer hipErrorNotReady            (a host-side spin: no protocol with the kernel)
//   g  as c with the polling instead of hipStreamSynchronize
//   h  as c with hipEventRecord + hipEventSynchronize on a default (spinning) event
//   hipcc -O2 --offload-arch=gfx950 tests/offline/launch_latency_probe.hip -o /tmp/llp && /tmp/llp
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                                     \
  do {                                                                                                            \
    hipError_t e_ = (x);                                                                                          \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

__global__ void __launch_bounds__(512) k_nothing(double *out) {
  if (threadIdx.x == 0 && out) out[0] = 1.0;
}

__global__ void __launch_bounds__(512) k_flag(volatile unsigned long long *flag, unsigned long long v) {
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store((unsigned long long *)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// reads nin doubles from `in` (host or device), writes nout doubles to `out`, then the flag
__global__ void __launch_bounds__(512) k_io(const double *__restrict__ in, int nin, double *out, int nout, unsigned long long *flag, unsigned long long v) {
  __shared__ double s[4096];
  for (int i = threadIdx.x; i < nin; i += 512) s[i] = in[i];
  __syncthreads();
  for (int i = threadIdx.x; i < nout; i += 512) out[i] = s[i] + 1.0;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0 && flag) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double median(std::vector<double> &v) {
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  const int N = 2000, nin = 1888, nout = 330;  // 15.1 KB in, 2.6 KB out: the Atlas problem's values and its x, y
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *d_in, *d_out, *h_in, *h_out;
  unsigned long long *h_flag;
  CK(hipMalloc(&d_in, nin * 8));
  CK(hipMalloc(&d_out, 4096 * 8));
  CK(hipHostMalloc(&h_in, nin * 8, hipHostMallocDefault));
  CK(hipHostMalloc(&h_out, 4096 * 8, hipHostMallocDefault));
  CK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault));
  for (int i = 0; i < nin; i++) h_in[i] = i;
  *h_flag = 0;
  using clk = std::chrono::steady_clock;
  auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  std::vector<double> t(N);
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (int form = 0; form < 8; form++) {
    for (int it = -50; it < N; it++) {
      const unsigned long long v = (unsigned long long)(form * 100000 + it + 100);
      const auto t0 = clk::now();
      if (form == 0) {
        hipLaunchKernelGGL(k_nothing, dim3(1), dim3(512), 0, st, d_out);
        CK(hipStreamSynchronize(st));
      } else if (form == 1) {
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(512), 0, st, h_flag, v);
        while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != v) {}
      } else if (form == 2) {
        CK(hipMemcpyAsync(d_in, h_in, nin * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_io, dim3(1), dim3(512), 0, st, d_in, nin, d_out, nout, (unsigned long long *)nullptr, v);
        CK(hipMemcpyAsync(h_out, d_out, nout * 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
      } else if (form == 3) {
        hipLaunchKernelGGL(k_io, dim3(1), dim3(512), 0, st, h_in, nin, h_out, nout, h_flag, v);
        while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != v) {}
      } else if (form == 4) {
        hipLaunchKernelGGL(k_io, dim3(1), dim3(512), 0, st, h_in, nin, h_out, nout, h_flag, v);
        CK(hipStreamSynchronize(st));
      } else if (form == 5) {
        hipLaunchKernelGGL(k_nothing, dim3(1), dim3(512), 0, st, d_out);
        hipError_t e;
        while ((e = hipStreamQuery(st)) == hipErrorNotReady) {}
        CK(e);
      } else if (form == 6) {
        CK(hipMemcpyAsync(d_in, h_in, nin * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_io, dim3(1), dim3(512), 0, st, d_in, nin, d_out, nout, (unsigned long long *)nullptr, v);
        CK(hipMemcpyAsync(h_out, d_out, nout * 8, hipMemcpyDeviceToHost, st));
        hipError_t e;
        while ((e = hipStreamQuery(st)) == hipErrorNotReady) {}
        CK(e);
      } else {
        CK(hipMemcpyAsync(d_in, h_in, nin * 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_io, dim3(1), dim3(512), 0, st, d_in, nin, d_out, nout, (unsigned long long *)nullptr, v);
        CK(hipMemcpyAsync(h_out, d_out, nout * 8, hipMemcpyDeviceToHost, st));
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
      }
      const auto t1 = clk::now();
      if (it >= 0) t[it] = us(t0, t1);
    }
    CK(hipStreamSynchronize(st));
    std::vector<double> s = t;
    const double med = median(s);
    static const char *names[] = {"a launch + synchronize", "b launch + spin on a host flag", "c H2D 15 KB + launch + D2H 3 KB + synchronize",
                                  "d kernel reads / writes page-locked host memory, spin", "e as d, synchronize",
                                  "f launch + polling hipStreamQuery", "g as c, polling hipStreamQuery", "h as c, hipEventSynchronize on a default event"};
    printf("{\"form\": \"%s\", \"median_us\": %.2f, \"p10_us\": %.2f, \"p90_us\": %.2f}\n", names[form], med, s[N / 10], s[N * 9 / 10]);
  }
  return 0;
}
