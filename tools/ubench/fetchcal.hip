// fetchcal.hip -- calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes this library uses:
// a known number of bytes is read (and written) with 16 B / lane and with 8 B / lane accesses, in kilobyte runs and in
// 64-byte runs spread over 16 series.  Run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE) and compare the counter
// with the byte count printed here.   hipcc --offload-arch=gfx950 -O3 fetchcal.hip -o fetchcal
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename VT, int RUN_LANES, bool WRITE>   // RUN_LANES lanes form one contiguous run
__global__ __launch_bounds__(64) void k_cal(const VT *__restrict__ src, VT *__restrict__ dst, long per_series, double *out) {
  constexpr int SPI = 64 / RUN_LANES;  // series per instruction
  const int lane = threadIdx.x;
  const long ser = (long)blockIdx.x * SPI + lane / RUN_LANES;
  const VT *p = src + ser * per_series + lane % RUN_LANES;
  VT *q = dst + ser * per_series + lane % RUN_LANES;
  const long steps = per_series / RUN_LANES;
  double acc = 0.0;
  for (long s = 0; s < steps; s += 4) {
    VT v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = p[(s + i) * RUN_LANES];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (WRITE) q[(s + i) * RUN_LANES] = v[i];
      else acc += ((const double *)&v[i])[0];
    }
  }
  if (!WRITE && acc == 12345.678) out[0] = acc;
}

template <typename VT, int RUN_LANES, bool WRITE>
void run(const char *name, const void *src, void *dst, long nser, long bytes_per_series, double *out) {
  constexpr int SPI = 64 / RUN_LANES;
  hipLaunchKernelGGL((k_cal<VT, RUN_LANES, WRITE>), dim3(nser / SPI), dim3(64), 0, 0, (const VT *)src, (VT *)dst,
                     bytes_per_series / (long)sizeof(VT), out);
  (void)hipDeviceSynchronize();
  printf("%-44s kernel reads %.3f GB%s\n", name, (double)nser * bytes_per_series / 1e9, WRITE ? " and writes as much" : "");
}

int main() {
  const long nser = 16384, bps = 4096L * 64;  // 4.29 GB: past the 256 MiB Infinity Cache
  void *src, *dst; double *out;
  (void)hipMalloc(&src, nser * bps); (void)hipMalloc(&dst, nser * bps); (void)hipMalloc(&out, 8);
  (void)hipMemset(src, 1, nser * bps);
  run<double2, 64, false>("read16_run1024  (k_cal<double2,64,false>)", src, dst, nser, bps, out);
  run<double2, 4, false>("read16_run64    (k_cal<double2,4,false>)", src, dst, nser, bps, out);
  run<double2, 8, false>("read16_run128   (k_cal<double2,8,false>)", src, dst, nser, bps, out);
  run<double, 64, false>("read8_run512    (k_cal<double,64,false>)", src, dst, nser, bps, out);
  run<double, 8, false>("read8_run64     (k_cal<double,8,false>)", src, dst, nser, bps, out);
  run<double2, 64, true>("copy16_run1024  (k_cal<double2,64,true>)", src, dst, nser, bps, out);
  run<double2, 4, true>("copy16_run64    (k_cal<double2,4,true>)", src, dst, nser, bps, out);
  run<double, 8, true>("copy8_run64     (k_cal<double,8,true>)", src, dst, nser, bps, out);
  return 0;
}
