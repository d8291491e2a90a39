// fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for KNOWN byte counts in the tick kernel's access pattern (gfx950).
//
// MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B for 16 B/lane streaming reads (= half the bytes) and is uncalibrated
// for other widths -- "calibrate on a known byte count in your own access pattern".  The tick reads per-QP RECORDS: one wavefront per QP
// (drawn from a ticket counter, so neighbouring QPs land on arbitrary XCDs), 8 B per lane, lanes < record length, records of 37 / 36 / 36 /
// 8 / 8 doubles packed back to back (no 128-B alignment).  The kernels below read a buffer of known size in that pattern and in the
// aligned / streaming patterns beside it; tools/fetch_calib.sh runs them under `rocprofv3 --pmc FETCH_SIZE` and prints counted / known.
//
//   hipcc -O2 --offload-arch=gfx950 tools/fetch_calib.hip -o /tmp/fetch_calib && /tmp/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                                     \
  do {                                                                                                            \
    hipError_t e_ = (x);                                                                                          \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

// 16 B per lane, every lane, grid-stride: the guide's calibrated case (expected: counted = known / 2)
__global__ void calib_stream16(const double2 *__restrict__ in, long long n2, double *out) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
    const double2 t = in[i];
    acc += t.x + t.y;
  }
  if (acc == 1.2345e300) out[0] = acc;
}

// 8 B per lane, every lane, grid-stride
__global__ void calib_stream8(const double *__restrict__ in, long long n, double *out) {
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += in[i];
  if (acc == 1.2345e300) out[0] = acc;
}

// the tick's pattern: a wavefront per record of R doubles (8 B per lane, lanes < R; a second load for R > 64), records drawn from a ticket
// counter, packed back to back.  R is a template parameter so that every record length is a kernel of its own name in the counter file.
template <int R>
__global__ void calib_records(const double *__restrict__ in, long long nrec, unsigned long long *ticket, double *out) {
  const int lane = threadIdx.x;
  double acc = 0.0;
  for (;;) {
    unsigned long long tk = 0;
    if (lane == 0) tk = atomicAdd(ticket, 1ull);
    tk = __shfl(tk, 0);
    if ((long long)tk >= nrec) break;
    const double *rec = in + (long long)tk * R;
    for (int i = lane; i < R; i += 64) acc += rec[i];
  }
  if (acc == 1.2345e300) out[0] = acc;
}

template <int R>
static void run_records(const double *d, long long bytes, unsigned long long *ticket, double *out, int reps) {
  const long long nrec = bytes / (8ll * R);
  for (int r = 0; r < reps; r++) {
    CK(hipMemset(ticket, 0, 8));
    hipLaunchKernelGGL(calib_records<R>, dim3(256 * 8), dim3(64), 0, 0, d, nrec, ticket, out);
    CK(hipDeviceSynchronize());
  }
  printf("{\"kernel\": \"calib_records<%d>\", \"known_bytes\": %lld, \"record_bytes\": %d}\n", R, nrec * 8ll * R, 8 * R);
}

int main(int argc, char **argv) {
  const long long bytes = (argc > 1 ? atoll(argv[1]) : 768ll) << 20;  // 768 MiB: three times the memory-side cache, nothing survives from launch to launch
  const int reps = 4;
  double *d, *out;
  unsigned long long *ticket;
  CK(hipMalloc(&d, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&ticket, 8));
  CK(hipMemset(d, 0, bytes));
  CK(hipDeviceSynchronize());
  for (int r = 0; r < reps; r++) {
    hipLaunchKernelGGL(calib_stream16, dim3(256 * 8), dim3(256), 0, 0, (const double2 *)d, bytes / 16, out);
    CK(hipDeviceSynchronize());
  }
  printf("{\"kernel\": \"calib_stream16\", \"known_bytes\": %lld}\n", bytes);
  for (int r = 0; r < reps; r++) {
    hipLaunchKernelGGL(calib_stream8, dim3(256 * 8), dim3(256), 0, 0, d, bytes / 8, out);
    CK(hipDeviceSynchronize());
  }
  printf("{\"kernel\": \"calib_stream8\", \"known_bytes\": %lld}\n", bytes);
  run_records<64>(d, bytes, ticket, out, reps);   // 512 B, aligned, every lane: the record pattern without partial lines
  run_records<32>(d, bytes, ticket, out, reps);   // 256 B, aligned, half the lanes
  run_records<16>(d, bytes, ticket, out, reps);   // 128 B = one line
  run_records<8>(d, bytes, ticket, out, reps);    // 64 B = half a line: the contact weight / force limit records
  run_records<37>(d, bytes, ticket, out, reps);   // 296 B, unaligned: q
  run_records<36>(d, bytes, ticket, out, reps);   // 288 B, unaligned: v, desired
  run_records<125>(d, bytes, ticket, out, reps);  // 1 000 B: a QP's whole input if it were one record
  CK(hipFree(d));
  return 0;
}
