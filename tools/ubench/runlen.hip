// runlen.hip -- HBM read (and write) rate as a function of the CONTIGUOUS RUN LENGTH per request when many series
// stream concurrently.  Layout as the library's U array: `nser` series of `rows` x 64 bytes, series-major.  A wavefront
// instruction (64 lanes x 16 B = 1 KiB) covers 1024/run series with `run` contiguous bytes each; a wavefront walks its
// series front to back, `depth` instructions in flight.  hipcc --offload-arch=gfx950 -O3 runlen.hip -o runlen
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int RUN, int DEPTH, bool WRITE>
__global__ __launch_bounds__(64) void k(const double2 *__restrict__ src, double2 *__restrict__ dst, long rows, double *out) {
  constexpr int LPS = RUN / 16;        // lanes per series
  constexpr int SPI = 64 / LPS;        // series per instruction
  const int lane = threadIdx.x;
  const long ser = (long)blockIdx.x * SPI + lane / LPS;
  const long bytes_per_series = rows * 64;
  const double2 *p = src + (ser * bytes_per_series) / 16 + lane % LPS;
  double2 *q = dst + (ser * bytes_per_series) / 16 + lane % LPS;
  const long steps = bytes_per_series / RUN;
  double acc = 0.0;
  for (long s = 0; s < steps; s += DEPTH) {
    double2 v[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) v[i] = p[(s + i) * LPS];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      if (WRITE) q[(s + i) * LPS] = v[i];
      else acc += v[i].x + v[i].y;
    }
  }
  if (!WRITE && acc == 12345.678) out[0] = acc;
}

template <int RUN, bool WRITE>
void run(const double2 *src, double2 *dst, long nser, long rows, double *out) {
  constexpr int SPI = 64 / (RUN / 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RUN, 8, WRITE>), dim3(nser / SPI), dim3(64), 0, 0, src, dst, rows, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nser * rows * 64 * (WRITE ? 2 : 1);
  printf("%s run %4d B  series %6ld  waves %6ld : %7.3f ms  %6.2f TB/s\n", WRITE ? "copy" : "read", RUN, nser, nser / SPI, ms, bytes / ms / 1e9);
}

int main(int argc, char **argv) {
  const long rows = 4096;
  for (long nser : {8192L, 16384L, 65536L}) {
    const size_t bytes = (size_t)nser * rows * 64;
    double2 *src, *dst; double *out;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&out, 8);
    hipMemset(src, 1, bytes);
    run<64, false>(src, dst, nser, rows, out);
    run<128, false>(src, dst, nser, rows, out);
    run<256, false>(src, dst, nser, rows, out);
    run<512, false>(src, dst, nser, rows, out);
    run<1024, false>(src, dst, nser, rows, out);
    run<64, true>(src, dst, nser, rows, out);
    run<128, true>(src, dst, nser, rows, out);
    run<256, true>(src, dst, nser, rows, out);
    run<1024, true>(src, dst, nser, rows, out);
    hipFree(src); hipFree(dst); hipFree(out);
  }
  return 0;
}
