// Which ways of raising a flag reach a kernel that is ALREADY running?  (design probe for streamed inputs)
//   a: hipMemcpyAsync H2D (pinned source) into fine-grained device memory, other stream
//   b: same into coarse-grained device memory
//   c: hipStreamWriteValue32 on another stream
//   d: the host writes a page-locked host word the kernel polls over PCIe
// Every spin loop is bounded (~0.5 s).  Build: hipcc --offload-arch=gfx950 -O2 -o flag_probe flag_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void waiter(const unsigned *flag, unsigned *out, int spins_max) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
    __builtin_amdgcn_s_sleep(64);
    if (++spins > spins_max) break;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)spins; out[1] = *flag; }
}
// a waiter shaped like the tick kernel: 256 VGPRs (two waves fill a SIMD's register file), 20 KB of LDS per workgroup --
// while 2048 of these are resident, NO other wave (e.g. of a runtime blit kernel) fits on the device
__global__ void __launch_bounds__(64, 2) waiter_fat(const unsigned *flag, unsigned *out, int spins_max) {
  __shared__ double lds[2550];
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  lds[threadIdx.x] = (double)threadIdx.x;
  __syncthreads();
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
    __builtin_amdgcn_s_sleep(64);
    if (++spins > spins_max) break;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (unsigned)spins; out[1] = *flag; out[2] = (unsigned)lds[5]; }
}
static void host_flag(void *p) { __atomic_store_n((unsigned *)p, 7u, __ATOMIC_RELEASE); }
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  unsigned *fine, *coarse, *hflag, *hsrc, *out;
  CK(hipExtMallocWithFlags((void **)&fine, 64, hipDeviceMallocFinegrained));
  CK(hipMalloc(&coarse, 64));
  CK(hipHostMalloc((void **)&hflag, 64, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&hsrc, 64, hipHostMallocDefault));
  CK(hipHostMalloc((void **)&out, 64, hipHostMallocDefault));
  hsrc[0] = 7u;
  const int spins_max = 300000;
  for (int mode = 0; mode < 4; mode++) {
    for (int grid : {1, 2048}) {
      unsigned *flag = mode == 0 ? fine : mode == 1 ? coarse : mode == 2 ? fine : hflag;
      if (mode == 3) hflag[0] = 0; else CK(hipMemset(flag, 0, 4));
      CK(hipDeviceSynchronize());
      out[0] = out[1] = 0;
      hipLaunchKernelGGL(waiter, dim3(grid), dim3(64), 0, s1, flag, out, spins_max);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
      auto t0 = std::chrono::steady_clock::now();
      if (mode == 0 || mode == 1) CK(hipMemcpyAsync(flag, hsrc, 4, hipMemcpyHostToDevice, s2));
      else if (mode == 2) CK(hipStreamWriteValue32(s2, flag, 7u, 0));
      else { __atomic_store_n(hflag, 7u, __ATOMIC_RELEASE); }
      CK(hipStreamSynchronize(s1));
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      CK(hipStreamSynchronize(s2));
      printf("mode %c grid %4d: kernel ended %.3f ms after the flag was posted, spins %u%s, flag seen %u\n", "abcd"[mode], grid, ms, out[0],
             out[0] > (unsigned)spins_max ? " (LIMIT)" : "", out[1]);
    }
  }
  // e: the full sequence of the streamed-input design: reset by memset on s2 -> event -> s1 waits -> waiter kernel on s1,
  //    then 5 x 2.4 MB pinned H2D copies + the 4-byte flag copy on s2
  {
    const size_t nb = 2400000;
    char *hbig, *dbig; hipEvent_t ev;
    CK(hipHostMalloc((void **)&hbig, nb * 5, hipHostMallocDefault));
    CK(hipMalloc(&dbig, nb * 5));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int rep = 0; rep < 3; rep++) {
      out[0] = out[1] = 0;
      CK(hipMemsetAsync(fine, 0, 4, s2));
      CK(hipEventRecord(ev, s2));
      CK(hipStreamWaitEvent(s1, ev, 0));
      hipLaunchKernelGGL(waiter, dim3(2048), dim3(64), 0, s1, fine, out, spins_max);
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 5; i++) CK(hipMemcpyAsync(dbig + i * nb, hbig + i * nb, nb, hipMemcpyHostToDevice, s2));
      CK(hipMemcpyAsync(fine, hsrc, 4, hipMemcpyHostToDevice, s2));
      double enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      CK(hipStreamSynchronize(s1));
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      CK(hipStreamSynchronize(s2));
      printf("mode e rep %d: enqueue %.3f ms, kernel ended after %.3f ms, spins %u%s, flag seen %u\n", rep, enq, ms, out[0],
             out[0] > (unsigned)spins_max ? " (LIMIT)" : "", out[1]);
    }
  }
  // f-h: the same sequence against the FAT waiter; the flag is raised by f: 4-byte H2D copy, g: hipStreamWriteValue32,
  //      h: hipLaunchHostFunc writing a page-locked host word
  {
    const size_t nb = 2400000;
    char *hbig, *dbig; hipEvent_t ev;
    CK(hipHostMalloc((void **)&hbig, nb * 5, hipHostMallocDefault));
    CK(hipMalloc(&dbig, nb * 5));
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int mode = 0; mode < 3; mode++)
      for (int rep = 0; rep < 2; rep++) {
        unsigned *flag = mode == 2 ? hflag : fine;
        out[0] = out[1] = 0;
        if (mode == 2) hflag[0] = 0; else CK(hipMemset(fine, 0, 4));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(waiter_fat, dim3(2048), dim3(64), 0, s1, flag, out, spins_max);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 5; i++) CK(hipMemcpyAsync(dbig + i * nb, hbig + i * nb, nb, hipMemcpyHostToDevice, s2));
        if (mode == 0) CK(hipMemcpyAsync(fine, hsrc, 4, hipMemcpyHostToDevice, s2));
        else if (mode == 1) CK(hipStreamWriteValue32(s2, fine, 7u, 0));
        else CK(hipLaunchHostFunc(s2, host_flag, hflag));
        CK(hipStreamSynchronize(s1));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipStreamSynchronize(s2));
        printf("mode %c rep %d (fat waiter): kernel ended after %.3f ms, spins %u%s, flag seen %u\n", "fgh"[mode], rep, ms, out[0],
               out[0] > (unsigned)spins_max ? " (LIMIT)" : "", out[1]);
      }
  }
  return 0;
}
