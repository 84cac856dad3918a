// c2_fused.hip -- fused log-likelihood kernels (the north-star hot path).
//
// c2_loglik      : factor + solve_lower + the two reductions of the reference's
//                  callers (numpy.py:66-87,104-109; core.py:428) in ONE pass over
//                  (t, a, U, V, y); nothing but ll[b] and flag[b] is written.
// c2_loglik_grad : log-likelihood and its reverse-mode gradient w.r.t.
//                  (t, c, a, U, V, y).
#include "c2_common.hpp"
#include "../../include/celerite2_amd.h"

extern "C" int c2_factor_rev_acc(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                 int64_t c_bs, const double *U, const double *d, const double *W, const double *S,
                                 const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                                 double *bV, int accumulate, c2_stream_t stream);

namespace c2 {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;

// =============================================================================
// Fused forward log-likelihood.  Same recursion as k_factor + k_sweep<LOWER,SOLVE>
// (reference forward.hpp:105-134 + internal.hpp:135-145) with lane j owning
// column j of S and element j of the solve state F.
// =============================================================================
template <int G, int PFN>
__global__ __launch_bounds__(kWave) void k_loglik(int64_t B, int64_t N, int J, const double *__restrict__ t,
                                                  int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                  const double *__restrict__ a, const double *__restrict__ U,
                                                  const double *__restrict__ V, const double *__restrict__ y,
                                                  double *__restrict__ ll, int32_t *__restrict__ flag) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t b = g / G;
  const int j = (int)(g % G);
  const bool valid = b < B;
  if (!valid) b = B - 1;
  const bool act = j < J;
  const int jj = act ? j : 0;
  const double *tb = t + b * t_bs, *ab = a + b * N, *yb = y + b * N;
  const double *Ub = U + b * N * J + jj, *Vb = V + b * N * J + jj;
  const double cj = act ? c[b * c_bs + j] : 0.0;

  double Sc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) Sc[i] = 0.0;

  double dprev = ab[0];
  double w = act ? Vb[0] / dprev : 0.0;
  double zprev = yb[0];
  double Fj = 0.0;
  double tprev = tb[0];
  double logdet = log(dprev);
  double quad = zprev * zprev / dprev;

  double rt[PFN], ra[PFN], ry[PFN], ru[PFN], rv[PFN];
  auto load_row = [&](int r, int64_t n) {
    const int64_t nn = (n < N) ? n : N - 1;
    rt[r] = tb[nn]; ra[r] = ab[nn]; ry[r] = yb[nn];
    ru[r] = act ? Ub[nn * J] : 0.0; rv[r] = act ? Vb[nn * J] : 0.0;
  };
#pragma unroll
  for (int r = 0; r < PFN; ++r) load_row(r, 1 + r);

  int32_t fl = 0;
  bool alive = true;
  for (int64_t n0 = 1; n0 < N; n0 += PFN) {
#pragma unroll
    for (int r = 0; r < PFN; ++r) {
      const int64_t n = n0 + r;
      if (n < N && alive) {
        const double tn = rt[r], an = ra[r], yn = ry[r], u = ru[r], v = rv[r];
        load_row(r, n + PFN);
        const double p = exp(cj * (tprev - tn));
        tprev = tn;
        const double dw = dprev * w;
        double tau = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const double wi = gget<G>(w, i), pi = gget<G>(p, i), ui = gget<G>(u, i);
          const double s = (pi * p) * fma(dw, wi, Sc[i]);
          Sc[i] = s;
          tau = fma(ui, s, tau);
        }
        Fj = p * fma(w, zprev, Fj);
        const double dn = an - gsum<G>(tau * u);
        const double zn = yn - gsum<G>(u * Fj);
        if (dn <= 0.0) {
          fl = (int32_t)n;
          alive = false;
        } else {
          w = (v - tau) / dn;
          dprev = dn;
          zprev = zn;
          logdet += log(dn);
          quad = fma(zn * zn, 1.0 / dn, quad);
        }
      }
    }
    if (!alive) break;
  }
  if (valid && j == 0) {
    flag[b] = fl;
    ll[b] = fl ? -INFINITY : -0.5 * (logdet + (double)N * kLog2Pi) - 0.5 * quad;
  }
}

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

int c2_loglik(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
              const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
              c2_stream_t stream) {
  if (B < 1 || N < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  if (!t || !c || !a || !U || !V || !y || !ll || !flag) return C2_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const int G_ = group_size(J);
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
  switch (G_) {
    case 1: hipLaunchKernelGGL((k_loglik<1, 8>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, V, y, ll, flag); break;
    case 2: hipLaunchKernelGGL((k_loglik<2, 8>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, V, y, ll, flag); break;
    case 4: hipLaunchKernelGGL((k_loglik<4, 8>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, V, y, ll, flag); break;
    case 8: hipLaunchKernelGGL((k_loglik<8, 8>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, V, y, ll, flag); break;
    case 16: hipLaunchKernelGGL((k_loglik<16, 4>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, V, y, ll, flag); break;
    default: hipLaunchKernelGGL((k_loglik<32, 4>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, V, y, ll, flag); break;
  }
  return launch_ok();
}

size_t c2_loglik_grad_workspace_bytes(int64_t B, int64_t N, int64_t J) {
  if (B < 1 || N < 1 || J < 1) return 0;
  return carve(nullptr, B, N, J).bytes;
}

int c2_loglik_grad(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
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
