// microbenchmark (GPU box): the register-resident elimination of qpc_qp_sp.h by itself -- one wavefront, KR columns, timed with
// the 100 MHz wall clock inside the kernel.  hipcc --offload-arch=gfx950 -O3 -o /tmp/reg_probe tests/offline/reg_elim_probe.hip && /tmp/reg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double qpc_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
}
__device__ __forceinline__ double rl(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int KR>
__device__ __forceinline__ void elim(double (&a)[KR], double (&mul)[KR], bool first, double refv, double tau, double &dv) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < KR; j++) {
    if (first) {
      const double d = rl(a[j], j), rf = rl(refv, j);
      const bool ok = d > tau * (rf > 1e-300 ? rf : 1e-300);
      const double di = ok ? qpc_rcp(d) : 0.0;
      mul[j] = a[j] * di;
      dv = lane == j ? di : dv;
    }
#pragma unroll
    for (int c = j + 1; c < KR; c++) a[c] -= a[j] * rl(mul[j], c);
  }
}
template <int KR>
__global__ void __launch_bounds__(64) kern(double *V, int r, const double *ref, double tau, double *dinv, unsigned long long *t) {
  const int lane = threadIdx.x & 63;
  extern __shared__ double lds[];
  for (int e = lane; e < KR * r; e += 64) lds[e] = V[e];
  __syncthreads();
  for (int rep = 0; rep < 3; rep++) {
    const unsigned long long t0 = wall_clock64();
    double a[KR], mul[KR], dv = 0.0;
    const double refv = ref[lane];
#pragma unroll
    for (int c = 0; c < KR; c++) a[c] = lds[c * r + lane];
    elim<KR>(a, mul, true, refv, tau, dv);
#pragma unroll
    for (int c = 1; c < KR; c++) lds[c * r + lane] = a[c];
    dinv[lane] = dv;
    __syncthreads();
    if (lane == 0) t[rep] = wall_clock64() - t0;
  }
}
template <int KR> void run() {
  const int r = 64;
  double *V, *ref, *dinv;
  unsigned long long *t;
  hipMalloc(&V, KR * r * 8); hipMalloc(&ref, 64 * 8); hipMalloc(&dinv, 64 * 8); hipMalloc(&t, 64);
  double h[KR * r], hr[64];
  for (int c = 0; c < KR; c++) for (int i = 0; i < r; i++) h[c * r + i] = (i == c ? 4.0 + c : 0.01 * ((i * 7 + c * 3) % 11));
  for (int i = 0; i < 64; i++) hr[i] = 1.0;
  hipMemcpy(V, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(ref, hr, sizeof(hr), hipMemcpyHostToDevice);
  for (int k = 0; k < 2; k++) hipLaunchKernelGGL(kern<KR>, dim3(1), dim3(64), KR * r * 8, 0, V, r, ref, 1e-12, dinv, t);
  hipDeviceSynchronize();
  unsigned long long ht[3];
  hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost);
  printf("KR %2d: %.2f %.2f %.2f us (%d update blocks)\n", KR, ht[0] * 0.01, ht[1] * 0.01, ht[2] * 0.01, KR * (KR - 1) / 2);
}
int main() { run<8>(); run<16>(); run<24>(); run<32>(); return 0; }
