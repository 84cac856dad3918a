// vmem_rate.hip -- what one wave64 global LOAD / STORE instruction costs the CU, by shape (gfx950, one wavefront per SIMD):
// every wavefront streams its own region with NI independent instructions in flight.  Shapes per instruction:
//   0: 8 B per lane, lanes consecutive (512 B contiguous)          1: 8 B per lane, 8 runs of 64 B, 256 KB apart (a row of 8 series)
//   2: 16 B per lane, lanes consecutive (1 KB contiguous)          3: 16 B per lane, 8 runs of 128 B, 256 KB apart (a line of 8 series)
//   4: 4 B per lane consecutive (256 B)                            5: 8 lanes x the same 8 bytes, 8 addresses (a scalar of 8 series)
// hipcc --offload-arch=gfx950 -O3 vmem_rate.hip -o vmem_rate && ./vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SHAPE, bool STORE>
__global__ __launch_bounds__(64) void k(double *buf, size_t wave_stride, int iters, double *sink) {
  const int l = threadIdx.x;
  char *base = reinterpret_cast<char *>(buf) + (size_t)blockIdx.x * wave_stride;
  size_t off, step;
  if (SHAPE == 0) { off = l * 8; step = 512; }
  else if (SHAPE == 1) { off = (size_t)(l >> 3) * 262144 + (l & 7) * 8; step = 64; }
  else if (SHAPE == 2) { off = l * 16; step = 1024; }
  else if (SHAPE == 3) { off = (size_t)(l >> 3) * 262144 + (l & 7) * 16; step = 128; }
  else if (SHAPE == 4) { off = l * 4; step = 256; }
  else { off = (size_t)(l >> 3) * 262144; step = 8; }
  double acc = 0.0;
  for (int it = 0; it < iters; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      char *p = base + off + (size_t)(it + u) * step;
      if (SHAPE == 2 || SHAPE == 3) {
        if (STORE) *reinterpret_cast<double2 *>(p) = make_double2(1.0, 2.0);
        else { const double2 v = *reinterpret_cast<const double2 *>(p); acc += v.x + v.y; }
      } else if (SHAPE == 4) {
        if (STORE) *reinterpret_cast<float *>(p) = 1.0f;
        else acc += *reinterpret_cast<const float *>(p);
      } else {
        if (STORE) *reinterpret_cast<double *>(p) = 1.0;
        else acc += *reinterpret_cast<const double *>(p);
      }
    }
  }
  if (acc == 1.2345) sink[0] = acc;
}

template <int SHAPE, bool STORE>
void run(double *buf, size_t wave_stride, int iters, double *sink, const char *what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int waves = 1024;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, STORE>), dim3(waves), dim3(64), 0, 0, buf, wave_stride, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes_per = (SHAPE == 2 || SHAPE == 3) ? 1024 : (SHAPE == 4 ? 256 : (SHAPE == 5 ? 64 : 512));
  const double instr = (double)waves * iters;
  printf("%-6s %-58s %8.3f ms  %6.1f ns per instruction and CU  %6.2f TB/s\n", STORE ? "store" : "load", what, ms,
         ms * 1e6 / (instr / 256.0), instr * bytes_per / ms / 1e9);
}

int main() {
  const size_t wave_stride = 8ull * 262144;   // 2 MiB per wavefront: the 8 "series" of a wavefront 256 KiB apart
  double *buf, *sink;
  hipMalloc(&buf, 1024 * wave_stride); hipMalloc(&sink, 8);
  hipMemset(buf, 0, 1024 * wave_stride);
#define BOTH(S, IT, W) run<S, false>(buf, wave_stride, IT, sink, W); run<S, true>(buf, wave_stride, IT, sink, W);
  BOTH(0, 4096, "8 B per lane, 512 B contiguous")
  BOTH(1, 4096, "8 B per lane, 8 runs of 64 B (a row of 8 series)")
  BOTH(2, 2048, "16 B per lane, 1 KB contiguous")
  BOTH(3, 2048, "16 B per lane, 8 runs of 128 B (a line of 8 series)")
  BOTH(4, 4096, "4 B per lane, 256 B contiguous")
  BOTH(5, 4096, "8 lanes x the same 8 bytes, 8 addresses")
  return 0;
}
