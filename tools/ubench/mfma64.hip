// mfma64.hip -- v_mfma_f64_16x16x4_f64 on gfx950: operand layout check (asymmetric A, B) and issue cost.
//   A: lane l holds A[i = l & 15][k = l >> 4];  B: lane l holds B[k = l >> 4][j = l & 15];
//   C/D: 4 doubles per lane, D[row = (l >> 4) + 4 r][col = l & 15]   (cdna_hip_programming.md section 3)
// hipcc --offload-arch=gfx950 -O3 mfma64.hip -o mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_check(const double *A, const double *B, double *D) {  // A 16x4, B 4x16 row-major, D 16x16
  const int l = threadIdx.x;
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(64) void k_time(double *out, int iters, long long *cyc) {
  const int l = threadIdx.x;
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// MFMA with independent fp64 FMAs interleaved: do the two pipes overlap for one wavefront?
__global__ __launch_bounds__(64) void k_mix(double *out, int iters, int nfma, long long *cyc) {
  const int l = threadIdx.x;
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  double f[8];
  for (int i = 0; i < 8; ++i) f[i] = 1.0 + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      for (int q = 0; q < nfma; ++q) f[q & 7] = fma(f[q & 7], 1.0000001, 1e-9);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 64 + l] = s;
  if (l == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1.0 + i * 0.37 + k * 1.9;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 0.5 - k * 0.11 + j * 0.73 + k * j * 0.01;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD; long long *dc;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 8192 * 64 * sizeof(double)); hipMalloc(&dc, 8);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
  printf("layout check: max |D - A B| = %.3g\n", err);
  const int iters = 20000;
  long long c;
  hipLaunchKernelGGL((k_time<1>), dim3(1), dim3(64), 0, 0, dD, iters, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("dependent chain (1 accumulator), 1 wave : %.1f cycles / MFMA\n", (double)c / iters);
  hipLaunchKernelGGL((k_time<4>), dim3(1), dim3(64), 0, 0, dD, iters, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("4 independent accumulators, 1 wave      : %.1f cycles / MFMA\n", (double)c / iters / 4);
  hipLaunchKernelGGL((k_time<8>), dim3(1), dim3(64), 0, 0, dD, iters, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("8 independent accumulators, 1 wave      : %.1f cycles / MFMA\n", (double)c / iters / 8);
  hipLaunchKernelGGL((k_time<4>), dim3(1024), dim3(64), 0, 0, dD, iters, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("4 accumulators, 1024 waves (1 / SIMD)   : %.1f cycles / MFMA\n", (double)c / iters / 4);
  for (int nf : {0, 4, 8, 12, 16}) {
    hipLaunchKernelGGL(k_mix, dim3(1024), dim3(64), 0, 0, dD, iters, nf, dc); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("MFMA + %2d independent fp64 FMAs each     : %.1f cycles / MFMA\n", nf, (double)c / iters / 4);
  }
  // chip-level throughput by wall clock: 1, 2, 4 wavefronts per SIMD (is one wavefront enough to keep the pipe busy?)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {1024, 2048, 4096, 8192}) {
    hipLaunchKernelGGL((k_time<4>), dim3(waves), dim3(64), 0, 0, dD, iters, dc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_time<4>), dim3(waves), dim3(64), 0, 0, dD, iters, dc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)waves * iters * 4 * 2048.0;
    printf("%5d waves x 4 accumulators: %.3f ms, %.1f TFLOP/s fp64 matrix (%.1f ns per MFMA per SIMD)\n", waves, ms, flops / ms / 1e9,
           ms * 1e6 / ((double)waves / 1024.0 * iters * 4));
  }
  return 0;
}
