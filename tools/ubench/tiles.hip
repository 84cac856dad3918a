// tiles.hip -- HBM read rate of the one-lane-per-series access pattern: a wavefront owns 64 series (256 KiB apart) and
// fetches, per tile, RUN contiguous bytes of each (RUN/16 instructions of 1 KiB, each covering 1024/RUN series), keeps
// AHEAD tiles in flight and spends `work` dependent FMAs per tile.  Which run length / prefetch depth reaches the HBM
// rate with 4 wavefronts per CU?   hipcc --offload-arch=gfx950 -O3 tiles.hip -o tiles
#include <hip/hip_runtime.h>
#include <cstdio>

template <int RUN, int AHEAD>
__global__ __launch_bounds__(64) void k(const double2 *__restrict__ src, long rows, int work, double *out) {
  constexpr int LPS = RUN / 16, SPI = 64 / LPS, NI = LPS;
  const int lane = threadIdx.x;
  const long bps = rows * 64;  // bytes per series
  const double2 *p = src + ((long)blockIdx.x * 64 + lane / LPS) * (bps / 16) + lane % LPS;
  const long stride_i = (long)SPI * (bps / 16);  // next group of series
  const long tiles = bps / RUN;
  double2 buf[AHEAD][NI];
#pragma unroll
  for (int a = 0; a < AHEAD; ++a)
#pragma unroll
    for (int i = 0; i < NI; ++i) buf[a][i] = p[i * stride_i + (long)a * LPS];
  double acc = 1.0, x = 1.0000001;
  for (long t = 0; t < tiles; t += AHEAD) {
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < NI; ++i) s += buf[a][i].x + buf[a][i].y;
      long nt = t + a + AHEAD; nt = nt < tiles ? nt : tiles - 1;
#pragma unroll
      for (int i = 0; i < NI; ++i) buf[a][i] = p[i * stride_i + nt * LPS];
      for (int w = 0; w < work; ++w) acc = fma(acc, x, s);
    }
  }
  if (acc == 12345.678) out[0] = acc;
}

template <int RUN, int AHEAD>
void run(const double2 *src, long nser, long rows, int work, double *out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RUN, AHEAD>), dim3(nser / 64), dim3(64), 0, 0, src, rows, work, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("run %4d B ahead %d work %4d series %6ld waves %5ld : %7.3f ms  %5.2f TB/s\n", RUN, AHEAD, work, nser, nser / 64, ms,
         (double)nser * rows * 64 / ms / 1e9);
}

int main() {
  const long rows = 4096;
  for (long nser : {32768L, 65536L}) {
    const size_t bytes = (size_t)nser * rows * 64;
    double2 *src; double *out;
    (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 8);
    (void)hipMemset(src, 1, bytes);
    for (int work : {0, 100, 400}) {
      run<64, 1>(src, nser, rows, work, out);
      run<64, 2>(src, nser, rows, work, out);
      run<64, 4>(src, nser, rows, work, out);
      run<128, 1>(src, nser, rows, work, out);
      run<128, 2>(src, nser, rows, work, out);
      run<128, 4>(src, nser, rows, work, out);
      run<256, 1>(src, nser, rows, work, out);
      run<256, 2>(src, nser, rows, work, out);
      run<512, 1>(src, nser, rows, work, out);
      run<512, 2>(src, nser, rows, work, out);
    }
    (void)hipFree(src); (void)hipFree(out);
  }
  return 0;
}
