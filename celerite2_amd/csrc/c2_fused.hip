// c2_fused.hip -- COMPOSITE log-likelihood gradient: the literal op chain an autodiff frontend runs
// (factor_fwd -> solve_lower_fwd -> seeds -> solve_lower_rev -> factor_rev), materialising the S and F
// workspaces in HBM exactly like the reference.  Kept as an internal cross-check of the fused
// checkpoint/recompute kernels in c2_loglik.hip (tests compare the two) and as the A/B baseline in profiles/.
#include "c2_common.hpp"
#include "../../include/celerite2_amd.h"

extern "C" int c2_factor_rev_acc(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                 int64_t c_bs, const double *U, const double *d, const double *W, const double *S,
                                 const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                                 double *bV, int accumulate, c2_stream_t stream);

namespace c2 {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;

// Seeds of the reverse pass + the log-likelihood value, one 256-thread block per
// series:  bd = -1/(2d) + z^2/(2 d^2),  bz = -z/d  (d ll / d d, d ll / d z).
__global__ __launch_bounds__(256) void k_seeds(int64_t N, const double *__restrict__ d, const double *__restrict__ z,
                                               const int32_t *__restrict__ flag, double *__restrict__ bd,
                                               double *__restrict__ bz, double *__restrict__ ll) {
  const int64_t b = blockIdx.x;
  const double *db = d + b * N, *zb = z + b * N;
  double *bdb = bd + b * N, *bzb = bz + b * N;
  double acc = 0.0;
  for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {
    const double dn = db[n], zn = zb[n];
    const double inv = 1.0 / dn;
    acc += log(dn) + zn * zn * inv;
    bdb[n] = 0.5 * inv * (zn * zn * inv - 1.0);
    bzb[n] = -zn * inv;
  }
  __shared__ double sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ll[b] = flag[b] ? -INFINITY : -0.5 * (sm[0] + (double)N * kLog2Pi);
}

}  // namespace c2

using namespace c2;

namespace {
inline int launch_ok() { return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP; }
struct GradWork {
  double *d, *W, *S, *z, *F, *bd, *bz, *bW;
  size_t bytes;
};
inline GradWork carve(void *work, int64_t B, int64_t N, int64_t J) {
  GradWork w;
  double *base = (double *)work;
  size_t off = 0;
  auto take = [&](int64_t n) { double *p = base ? base + off : nullptr; off += (size_t)n; return p; };
  w.d = take(B * N);
  w.W = take(B * N * J);
  w.S = take(B * N * J * J);
  w.z = take(B * N);
  w.F = take(B * N * J);
  w.bd = take(B * N);
  w.bz = take(B * N);
  w.bW = take(B * N * J);
  w.bytes = off * sizeof(double);
  return w;
}
}  // namespace

extern "C" {

size_t c2_loglik_grad_composite_workspace_bytes(int64_t B, int64_t N, int64_t J) {
  if (B < 1 || N < 1 || J < 1) return 0;
  return carve(nullptr, B, N, J).bytes;
}

int c2_loglik_grad_composite(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                   const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                   double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                   size_t work_bytes, c2_stream_t stream) {
  if (B < 1 || N < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  if (!t || !c || !a || !U || !V || !y || !ll || !bt || !bc || !ba || !bU || !bV || !by || !flag || !work)
    return C2_ERR_INVALID;
  const GradWork w = carve(work, B, N, J);
  if (work_bytes < w.bytes) return C2_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  int e;
  // forward with workspaces (backprop.factor_fwd / solve_lower_fwd)
  if ((e = c2_factor(B, N, J, t, t_bs, c, c_bs, a, U, V, w.d, w.W, w.S, flag, stream))) return e;
  if ((e = c2_solve_lower(B, N, J, 1, t, t_bs, c, c_bs, U, w.W, y, w.z, w.F, stream))) return e;
  // value + cotangent seeds
  hipLaunchKernelGGL(k_seeds, dim3((unsigned)B), dim3(256), 0, s, N, w.d, w.z, flag, w.bd, w.bz, ll);
  if ((e = launch_ok())) return e;
  // reverse (backprop.solve_lower_rev, then factor_rev accumulating into bt, bc, bU)
  if ((e = c2_solve_lower_rev(B, N, J, 1, t, t_bs, c, c_bs, U, w.W, y, w.z, w.F, w.bz, bt, bc, bU, w.bW, by, stream)))
    return e;
  return c2_factor_rev_acc(B, N, J, t, t_bs, c, c_bs, U, w.d, w.W, w.S, w.bd, w.bW, bt, bc, ba, bU, bV, 1, stream);
}

}  // extern "C"
