// Accuracy of v_rcp_f64 alone, with one and with two Newton steps, against IEEE division (relative error in ulps of 2^-52).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *x, double *e0, double *e1, double *e2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i], ref = 1.0 / d;
  double r = __builtin_amdgcn_rcp(d);
  e0[i] = fabs(r - ref) / ref;
  double e = fma(-d, r, 1.0); r = fma(r, e, r);
  e1[i] = fabs(r - ref) / ref;
  e = fma(-d, r, 1.0); r = fma(r, e, r);
  e2[i] = fabs(r - ref) / ref;
}
int main() {
  const int n = 1 << 22;
  double *x, *e0, *e1, *e2;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&e0, n * 8); hipMallocManaged(&e1, n * 8); hipMallocManaged(&e2, n * 8);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = ldexp(1.0 + (double)(s >> 11) / 9007199254740992.0, (int)(s % 40) - 20); }
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, e0, e1, e2, n);
  hipDeviceSynchronize();
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) { m0 = fmax(m0, e0[i]); m1 = fmax(m1, e1[i]); m2 = fmax(m2, e2[i]); }
  printf("max rel err: rcp alone %.3e (%.2f ulp)  +1 Newton %.3e (%.2f ulp)  +2 Newton %.3e (%.2f ulp)\n", m0, m0 / 2.22e-16, m1, m1 / 2.22e-16, m2, m2 / 2.22e-16);
  return 0;
}
