// sincos_check.hip -- accuracy of sincos_cw (c2_loglik_helpers.hpp) against the host libm on a log-uniform + uniform grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "../../celerite2_amd/csrc/c2_loglik_helpers.hpp"
__global__ void k(const double *x, double *s, double *c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) c2::sincos_cw(x[i], s[i], c[i]);
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), s(n), c(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-1.0, 1.0), e(-6.0, 6.2);
  for (int i = 0; i < n; ++i) x[i] = (i & 1) ? u(g) * 1.5e6 : copysign(pow(10.0, e(g)), u(g));
  x[0] = 0.0; x[1] = M_PI / 2; x[2] = M_PI; x[3] = 1e-300; x[4] = 355.0; x[5] = 1.5e6; x[6] = 3e7;
  double *dx, *ds, *dc;
  (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&ds, n * 8); (void)hipMalloc(&dc, n * 8);
  (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
  (void)hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
  double es = 0, ec = 0; int is = 0, ic = 0;
  for (int i = 0; i < n; ++i) {
    const double a = fabs(s[i] - sin(x[i])), b = fabs(c[i] - cos(x[i]));
    if (a > es) { es = a; is = i; }
    if (b > ec) { ec = b; ic = i; }
  }
  printf("max |sin err| %.3g at x = %.17g; max |cos err| %.3g at x = %.17g (absolute; |sin|,|cos| <= 1)\n", es, x[is], ec, x[ic]);
  return 0;
}
