// c2_kron.hip -- the 2-D (multi-band) extension of the log-likelihood path, rank-1 band covariance:
//
//     K = T (x) alpha alpha^T + diag ,     T_{nn'} = k(|t_n - t_n'|)   (N epochs x M bands, interleaved n-major)
//
// EXTENSION, PARITY UNPINNED BY THE REFERENCE: the reference contains no 2-D code (there is no core2.hpp; only
// README.md:14-17 points at the paper).  SURVEY.md section 8a-2D gives the construction this file implements: with a
// rank-1 band covariance the interleaved series of length N*M obeys the SAME 1-D recursions (forward.hpp:105-134,
// internal.hpp:135-145) with U' = U (x) alpha, V' = V (x) alpha, a' = diag + alpha^2 k(0) and dt = 0 (p = 1) between
// the bands of one epoch.  Two methods, both behind c2_kron_loglik / c2_kron_loglik_grad:
//
//   C2_KRON_INTERLEAVED  the construction as stated: a device generator writes (t', a', U', V') of the N*M series,
//                        the fused 1-D kernels (c2_loglik.hip) run on it, a reducer folds the gradients back onto
//                        (t, a, U, V, alpha, diag).  192 MB per GP at N = 50000, M = 16, J = 6 -- the naive view.
//                        A CROSS-CHECK, not a production path: 941 ms per 32 series at the BASELINE shape (0.0008 of
//                        the roofline; its 400 000 chunks with runs of identical times stay row by row) against 2.85 ms
//                        for the collapsed method, which is the default.
//   C2_KRON_COLLAPSED    structure-aware and exact: the M bands of an epoch observe ONE latent value x(t_n) through
//                        y_m = alpha_m x + eps_m, eps_m ~ N(0, D_m), so they are equivalent to a single observation
//                            ytil_n = (sum_m alpha_m y_m / D_m) / A_n ,   variance 1 / A_n ,   A_n = sum_m alpha_m^2 / D_m
//                        plus a term that does not involve the GP:
//                            ll = ll_1D(t, a + 1/A, U, V; ytil)
//                                 + sum_n [ -(M-1)/2 log 2pi - 1/2 sum_m log D_m - 1/2 log A_n
//                                           - 1/2 (sum_m y_m^2 / D_m - b_n^2 / A_n) ] ,  b_n = sum_m alpha_m y_m / D_m.
//                        The sequential recursion runs over N epochs instead of N*M rows (16x fewer dependent steps at
//                        M = 16); the per-band work is two embarrassingly parallel passes.  Needs diag > 0.
// Both are checked against the dense Kronecker matrix (oracle/dense.py: kron_dense) at tiny sizes and against each
// other and the 1-D oracle on the interleaved series at larger ones (tests/test_gpu_kron.py).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "../../include/celerite2_amd.h"
#include "c2_dispatch.hpp"

extern "C" void c2_internal_set_error(const char *msg);
extern "C" int c2_internal_loglik_grad_rows(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                            int64_t c_bs, const double *a, const double *U, const double *V,
                                            const double *y, double *ll, double *bt, double *bc, double *ba, double *bU,
                                            double *bV, double *by, int32_t *flag, void *work, size_t work_bytes,
                                            c2_stream_t stream);

namespace c2k {

constexpr int kThreads = 256;
constexpr int kEpt = 4;                        // epochs per thread in the collapse kernels
constexpr int kEpb = kThreads * kEpt;          // epochs per block
constexpr double kLog2Pi = 1.8378770664093454835606594728112;

__device__ __forceinline__ double block_sum(double x, double *red) {  // red: kThreads / 64 doubles
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kThreads / 64; ++i) s += red[i];
  return s;
}

// ---- interleaved method: generator ------------------------------------------------------------------------------
// One thread per (b, i = n M + m, j):  U'[b,i,j] = alpha_m U[b,n,j], V' likewise; the j == 0 thread also writes
// t'[i] = t[n] and a'[b,i] = diag[b,n,m] + alpha_m^2 a[b,n]   (a = k(0): the 1-D diagonal WITHOUT white noise).
__global__ void k_kron_expand(int64_t B, int64_t N, int M, int J, const double *__restrict__ t, int64_t t_bs,
                              const double *__restrict__ a, const double *__restrict__ U,
                              const double *__restrict__ V, const double *__restrict__ alpha, int64_t alpha_bs,
                              const double *__restrict__ diag, double *__restrict__ t2, double *__restrict__ a2,
                              double *__restrict__ U2, double *__restrict__ V2) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * N * M * J) return;
  const int64_t row = g / J;  // b * N * M + i
  const int j = (int)(g - row * J);
  const int64_t bn = row / M;  // b * N + n
  const int m = (int)(row - bn * M);
  const int64_t b = bn / N, n = bn - b * N;
  const double al = alpha[b * alpha_bs + m];
  U2[g] = al * U[bn * J + j];
  V2[g] = al * V[bn * J + j];
  if (j == 0) {
    a2[row] = fma(al * al, a[bn], diag[row]);
    if (t_bs != 0 || b == 0) t2[(t_bs ? b * N * M : 0) + n * M + m] = t[b * t_bs + n];
  }
}

// ---- interleaved method: gradients of the N*M series folded back -------------------------------------------------
// Thread per epoch (b, n):  bt_n = sum_m bt'; ba_n = sum_m alpha_m^2 ba'; bdiag = ba'; bU_n = sum_m alpha_m bU'_{nm};
// balpha_m += 2 alpha_m a_n ba'_{nm} + sum_j (bU'_{nmj} U_nj + bV'_{nmj} V_nj)   -> per-block partials.
__global__ __launch_bounds__(kThreads) void k_kron_expand_rev(
    int64_t B, int64_t N, int M, int J, const double *__restrict__ a, const double *__restrict__ U,
    const double *__restrict__ V, const double *__restrict__ alpha, int64_t alpha_bs, const double *__restrict__ bt2,
    const double *__restrict__ ba2, const double *__restrict__ bU2, const double *__restrict__ bV2,
    double *__restrict__ bt, double *__restrict__ ba, double *__restrict__ bU, double *__restrict__ bV,
    double *__restrict__ bdiag, double *__restrict__ part /* (B, nch, M) */, int nch) {
  __shared__ double red[kThreads / 64];
  const int64_t b = blockIdx.y;
  const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const bool ok = n < N;
  const int64_t bn = b * N + (ok ? n : N - 1);
  const double *al = alpha + b * alpha_bs;
  double sbt = 0.0, sba = 0.0;
  for (int m = 0; m < M; ++m) {
    const double g = ba2[bn * M + m];
    sbt += bt2[bn * M + m];
    sba = fma(al[m] * al[m], g, sba);
    if (ok) bdiag[bn * M + m] = g;
  }
  if (ok) { bt[bn] = sbt; ba[bn] = sba; }
  for (int j = 0; j < J; ++j) {
    double su = 0.0, sv = 0.0;
    for (int m = 0; m < M; ++m) {
      su = fma(al[m], bU2[(bn * M + m) * J + j], su);
      sv = fma(al[m], bV2[(bn * M + m) * J + j], sv);
    }
    if (ok) { bU[bn * J + j] = su; bV[bn * J + j] = sv; }
  }
  const double an = a[bn];
  for (int m = 0; m < M; ++m) {
    double s = 2.0 * al[m] * an * ba2[bn * M + m];
    for (int j = 0; j < J; ++j) {
      s = fma(bU2[(bn * M + m) * J + j], U[bn * J + j], s);
      s = fma(bV2[(bn * M + m) * J + j], V[bn * J + j], s);
    }
    const double tot = block_sum(ok ? s : 0.0, red);
    if (threadIdx.x == 0) part[(b * nch + blockIdx.x) * M + m] = tot;
  }
}

// ---- collapsed method: the M bands of an epoch -> one effective observation --------------------------------------
struct Epoch {
  double A, yt, R, ld;  // sum alpha^2/D, ytil = (sum alpha y/D)/A, sum (y - alpha ytil)^2/D, sum log D
  bool bad;
};
// R is formed from the residuals r_m = y_m - alpha_m ytil, not as sum y^2/D - b^2/A: for data dominated by the
// common signal the two terms of the latter cancel to the noise level.
__device__ __forceinline__ Epoch epoch_sums(const double *__restrict__ al, const double *__restrict__ D,
                                            const double *__restrict__ y, int M) {
  Epoch e{0.0, 0.0, 0.0, 0.0, false};
  double bsum = 0.0;
  for (int m = 0; m < M; ++m) {
    const double d = D[m], ay = al[m] / d;
    e.bad = e.bad || !(d > 0.0);
    e.A = fma(al[m], ay, e.A);
    bsum = fma(ay, y[m], bsum);
    e.ld += log(d);
  }
  e.yt = bsum / e.A;
  for (int m = 0; m < M; ++m) {
    const double r = fma(-al[m], e.yt, y[m]);
    e.R = fma(r / D[m], r, e.R);
  }
  return e;
}

// a_eff = a + 1/A, y_eff = b/A, and the GP-free part of the log-likelihood as per-block partial sums.
__global__ __launch_bounds__(kThreads) void k_kron_collapse(int64_t B, int64_t N, int M, const double *__restrict__ a,
                                                            const double *__restrict__ alpha, int64_t alpha_bs,
                                                            const double *__restrict__ diag,
                                                            const double *__restrict__ y, double *__restrict__ a_eff,
                                                            double *__restrict__ y_eff, double *__restrict__ part,
                                                            int32_t *__restrict__ badflag, int nch) {
  __shared__ double red[kThreads / 64];
  const int64_t b = blockIdx.y;
  const double *al = alpha + b * alpha_bs;
  double corr = 0.0;
  int32_t bad = 0;
#pragma unroll
  for (int e = 0; e < kEpt; ++e) {
    const int64_t n = (int64_t)blockIdx.x * kEpb + e * kThreads + threadIdx.x;
    if (n < N) {
      const int64_t bn = b * N + n;
      const Epoch s = epoch_sums(al, diag + bn * M, y + bn * M, M);
      a_eff[bn] = a[bn] + 1.0 / s.A;
      y_eff[bn] = s.yt;
      corr += -0.5 * ((double)(M - 1) * kLog2Pi + s.ld + log(s.A) + s.R);
      if (s.bad || !(s.A > 0.0)) bad = 1;
    }
  }
  const double tot = block_sum(corr, red);
  if (threadIdx.x == 0) part[b * nch + blockIdx.x] = tot;
  if (bad) atomicOr(reinterpret_cast<int *>(badflag + b), 1);
}

// ll[b] = ll_1D[b] + sum of the partials (fixed order -> deterministic); a non-positive diag makes the series invalid.
__global__ void k_kron_finish(int64_t B, int nch, const double *__restrict__ part, const int32_t *__restrict__ badflag,
                              double *__restrict__ ll, int32_t *__restrict__ flag) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int k = 0; k < nch; ++k) s += part[b * nch + k];
  if (badflag[b]) {
    ll[b] = -INFINITY;
    if (flag[b] == 0) flag[b] = -1;  // not a pivot failure: a band variance <= 0 (collapsed method needs diag > 0)
  } else {
    ll[b] += s;  // -inf stays -inf for a failed factorisation
  }
}

// Chain rule back through (ytil, 1/A) and the GP-free term, in residual form (r_m = y_m - alpha_m ytil; ytil minimises
// R, so dR/dytil = 0).  g_s = d ll_1D / d a_eff (= ba of the 1-D pass), g_y = d ll_1D / d ytil:
//   by_nm    = g_y alpha_m / (D_m A) - r_m / D_m
//   balpha_m = sum_n [ g_y (r_m - alpha_m ytil) / (D_m A) - 2 g_s alpha_m / (D_m A^2) - alpha_m / (D_m A) + ytil r_m / D_m ]
//   bdiag_nm = [ -g_y alpha_m r_m / A + g_s alpha_m^2 / A^2 + alpha_m^2 / (2A) + r_m^2 / 2 ] / D_m^2 - 1 / (2 D_m)
__global__ __launch_bounds__(kThreads) void k_kron_collapse_rev(
    int64_t B, int64_t N, int M, const double *__restrict__ alpha, int64_t alpha_bs, const double *__restrict__ diag,
    const double *__restrict__ y, const double *__restrict__ g_s, const double *__restrict__ g_y,
    const int32_t *__restrict__ badflag, double *__restrict__ bdiag, double *__restrict__ by,
    double *__restrict__ part /* (B, nch, M) */, int nch) {
  __shared__ double red[kThreads / 64];
  const int64_t b = blockIdx.y;
  const double *al = alpha + b * alpha_bs;
  const bool invalid = badflag[b] != 0;  // a band variance <= 0: every gradient of the series is NaN
  double rA[kEpt], yt[kEpt], gs[kEpt], gy[kEpt];
  int64_t row[kEpt];
#pragma unroll
  for (int e = 0; e < kEpt; ++e) {
    const int64_t n = (int64_t)blockIdx.x * kEpb + e * kThreads + threadIdx.x;
    row[e] = -1;
    rA[e] = yt[e] = gs[e] = gy[e] = 0.0;
    if (n < N) {
      const int64_t bn = b * N + n;
      row[e] = bn;
      const Epoch s = epoch_sums(al, diag + bn * M, y + bn * M, M);
      rA[e] = 1.0 / s.A; yt[e] = s.yt; gs[e] = g_s[bn]; gy[e] = g_y[bn];
      if (invalid) gs[e] = gy[e] = __builtin_nan("");
    }
  }
  for (int m = 0; m < M; ++m) {
    double sum = 0.0;
    const double am = al[m];
#pragma unroll
    for (int e = 0; e < kEpt; ++e) {
      if (row[e] >= 0) {
        const int64_t i = row[e] * M + m;
        const double rd = 1.0 / diag[i], r = fma(-am, yt[e], y[i]);
        const double arA = am * rA[e];
        sum += (gy[e] * (r - am * yt[e]) * rA[e] - 2.0 * gs[e] * arA * rA[e] - arA + yt[e] * r) * rd;
        bdiag[i] = ((-gy[e] * arA * r + gs[e] * arA * arA + 0.5 * am * arA + 0.5 * r * r) * rd - 0.5) * rd;
        by[i] = (gy[e] * arA - r) * rd;
      }
    }
    const double tot = block_sum(sum, red);
    if (threadIdx.x == 0) part[(b * nch + blockIdx.x) * M + m] = tot;
  }
}

// ---- the same two kernels with lanes over the BANDS (M a power of two up to 32) -----------------------------------------
// A thread per (epoch, band): a wavefront reads 64 / M epochs x M bands as one dense run, the sums over the bands of an
// epoch are butterflies inside groups of M lanes.  Same grid, outputs and partial-sum layout as the kernels above (a
// block covers the same kEpb epochs of one series), which remain for every other M.  32 x 50000 x 16: collapse 1.66 ->
// 0.30 ms, collapse_rev 2.31 -> 0.15 ms.
template <int G>
__device__ __forceinline__ double band_sum(double x) {
#pragma unroll
  for (int o = 1; o < G; o <<= 1) x += __shfl_xor(x, o, 64);
  return x;
}
template <int G>
__global__ __launch_bounds__(kThreads) void k_kron_collapse_b(int64_t B, int64_t N, const double *__restrict__ a,
                                                              const double *__restrict__ alpha, int64_t alpha_bs,
                                                              const double *__restrict__ diag,
                                                              const double *__restrict__ y, double *__restrict__ a_eff,
                                                              double *__restrict__ y_eff, double *__restrict__ part,
                                                              int32_t *__restrict__ badflag, int nch) {
  constexpr int EPS = kThreads / G;   // epochs per slab of kThreads elements
  __shared__ double red[kThreads / 64];
  const int64_t b = blockIdx.y;
  const int m = threadIdx.x % G, eo = threadIdx.x / G;
  const double al = alpha[b * alpha_bs + m];
  double corr = 0.0;
  int32_t bad = 0;
  for (int e0 = 0; e0 < kEpb; e0 += EPS) {
    const int64_t n = (int64_t)blockIdx.x * kEpb + e0 + eo;
    const bool in = n < N;
    const int64_t bn = b * N + (in ? n : N - 1), i = bn * G + m;
    const double d = diag[i], yv = y[i], ay = al / d;
    const double A = band_sum<G>(al * ay), bs = band_sum<G>(ay * yv), ld = band_sum<G>(log(d));
    const double yt = bs / A, r = fma(-al, yt, yv);
    const double R = band_sum<G>(r / d * r);
    const bool ebad = band_sum<G>(d > 0.0 ? 0.0 : 1.0) != 0.0 || !(A > 0.0);
    if (in && m == 0) {
      a_eff[bn] = a[bn] + 1.0 / A;
      y_eff[bn] = yt;
      corr += -0.5 * ((double)(G - 1) * kLog2Pi + ld + log(A) + R);
      if (ebad) bad = 1;
    }
  }
  const double tot = block_sum(corr, red);
  if (threadIdx.x == 0) part[b * nch + blockIdx.x] = tot;
  if (bad) atomicOr(reinterpret_cast<int *>(badflag + b), 1);
}
template <int G>
__global__ __launch_bounds__(kThreads) void k_kron_collapse_rev_b(
    int64_t B, int64_t N, const double *__restrict__ alpha, int64_t alpha_bs, const double *__restrict__ diag,
    const double *__restrict__ y, const double *__restrict__ g_s, const double *__restrict__ g_y,
    const int32_t *__restrict__ badflag, double *__restrict__ bdiag, double *__restrict__ by,
    double *__restrict__ part /* (B, nch, M) */, int nch) {
  constexpr int EPS = kThreads / G;
  __shared__ double red[kThreads];
  const int64_t b = blockIdx.y;
  const int m = threadIdx.x % G, eo = threadIdx.x / G;
  const double am = alpha[b * alpha_bs + m];
  const bool invalid = badflag[b] != 0;  // a band variance <= 0: every gradient of the series is NaN
  double sum = 0.0;
  for (int e0 = 0; e0 < kEpb; e0 += EPS) {
    const int64_t n = (int64_t)blockIdx.x * kEpb + e0 + eo;
    const bool in = n < N;
    const int64_t bn = b * N + (in ? n : N - 1), i = bn * G + m;
    const double d = diag[i], yv = y[i], rd = 1.0 / d, ay = am * rd;
    const double A = band_sum<G>(am * ay), bs = band_sum<G>(ay * yv);
    const double rA = 1.0 / A, yt = bs * rA, r = fma(-am, yt, yv), arA = am * rA;
    double gs = g_s[bn], gy = g_y[bn];
    if (invalid) gs = gy = __builtin_nan("");
    if (in) {
      sum += (gy * (r - am * yt) * rA - 2.0 * gs * arA * rA - arA + yt * r) * rd;
      bdiag[i] = ((-gy * arA * r + gs * arA * arA + 0.5 * am * arA + 0.5 * r * r) * rd - 0.5) * rd;
      by[i] = (gy * arA - r) * rd;
    }
  }
  // the band's sum over the block's epochs: the EPS threads that share m, in a fixed order
  red[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x < G) {
    double tot = 0.0;
    for (int q = 0; q < EPS; ++q) tot += red[q * G + threadIdx.x];
    part[(b * nch + blockIdx.x) * G + threadIdx.x] = tot;
  }
}

// balpha[b, m] = sum over the per-block partials, in a fixed order.
__global__ void k_kron_balpha(int64_t B, int M, int nch, const double *__restrict__ part, double *__restrict__ balpha) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * M) return;
  const int64_t b = g / M;
  const int m = (int)(g - b * M);
  double s = 0.0;
  for (int k = 0; k < nch; ++k) s += part[(b * nch + k) * M + m];
  balpha[g] = s;
}

static bool use_banded() {
  return !(c2::opt::has(c2::opt::k_kron_banded) && c2::opt::ival(c2::opt::k_kron_banded) == 0);   // 0: the thread-per-epoch kernels for every M (A/B runs, tests)
}
static void launch_collapse(int64_t B, int64_t N, int64_t M, const double *a, const double *alpha, int64_t alpha_bs,
                            const double *diag, const double *y, double *a_eff, double *y_eff, double *part,
                            int32_t *badflag, int nch, hipStream_t s) {
  const dim3 grid((unsigned)nch, (unsigned)B);
#define C2K_C(G) hipLaunchKernelGGL((k_kron_collapse_b<G>), grid, dim3(kThreads), 0, s, B, N, a, alpha, alpha_bs, diag, y, a_eff, y_eff, part, badflag, nch)
  if (use_banded() && M == 2) C2K_C(2);
  else if (use_banded() && M == 4) C2K_C(4);
  else if (use_banded() && M == 8) C2K_C(8);
  else if (use_banded() && M == 16) C2K_C(16);
  else if (use_banded() && M == 32) C2K_C(32);
  else hipLaunchKernelGGL(k_kron_collapse, grid, dim3(kThreads), 0, s, B, N, (int)M, a, alpha, alpha_bs, diag, y, a_eff,
                          y_eff, part, badflag, nch);
#undef C2K_C
}
static void launch_collapse_rev(int64_t B, int64_t N, int64_t M, const double *alpha, int64_t alpha_bs,
                                const double *diag, const double *y, const double *g_s, const double *g_y,
                                const int32_t *badflag, double *bdiag, double *by, double *part, int nch, hipStream_t s) {
  const dim3 grid((unsigned)nch, (unsigned)B);
#define C2K_R(G) hipLaunchKernelGGL((k_kron_collapse_rev_b<G>), grid, dim3(kThreads), 0, s, B, N, alpha, alpha_bs, diag, y, g_s, g_y, badflag, bdiag, by, part, nch)
  if (use_banded() && M == 2) C2K_R(2);
  else if (use_banded() && M == 4) C2K_R(4);
  else if (use_banded() && M == 8) C2K_R(8);
  else if (use_banded() && M == 16) C2K_R(16);
  else if (use_banded() && M == 32) C2K_R(32);
  else hipLaunchKernelGGL(k_kron_collapse_rev, grid, dim3(kThreads), 0, s, B, N, (int)M, alpha, alpha_bs, diag, y, g_s, g_y,
                          badflag, bdiag, by, part, nch);
#undef C2K_R
}

inline int launch_ok() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return C2_OK;
  c2_internal_set_error(hipGetErrorString(e));
  return C2_ERR_HIP;
}
inline size_t al2(size_t n) { return (n + 1) & ~(size_t)1; }  // keep sub-arrays 16-byte aligned

struct Plan {  // carve-up of the caller's workspace, in doubles
  size_t a_eff, y_eff, part, badflag, g_y, one_d, t2, a2, U2, V2, bt2, ba2, bU2, bV2, total;
  int nch;
};
inline Plan plan(int64_t B, int64_t N, int64_t M, int64_t J, int method, int grad) {
  Plan p{};
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += al2(n); return at; };
  if (method == C2_KRON_COLLAPSED) {
    p.nch = (int)((N + kEpb - 1) / kEpb);
    p.a_eff = take((size_t)B * N);
    p.y_eff = take((size_t)B * N);
    p.part = take((size_t)B * p.nch * (grad ? (size_t)M : 1));
    p.badflag = take(((size_t)B + 1) / 2);
    if (grad) {
      p.g_y = take((size_t)B * N);
      p.one_d = take(c2_loglik_grad_workspace_bytes(B, N, J) / sizeof(double));
    }
  } else {
    const size_t R = (size_t)N * M;
    p.nch = (int)((N + kThreads - 1) / kThreads);
    p.t2 = take((size_t)B * R);
    p.a2 = take((size_t)B * R);
    p.U2 = take((size_t)B * R * J);
    p.V2 = take((size_t)B * R * J);
    if (grad) {
      p.bt2 = take((size_t)B * R);
      p.ba2 = take((size_t)B * R);
      p.bU2 = take((size_t)B * R * J);
      p.bV2 = take((size_t)B * R * J);
      p.part = take((size_t)B * p.nch * M);
      p.one_d = take(c2_loglik_grad_workspace_bytes(B, (int64_t)R, J) / sizeof(double));
    }
  }
  p.total = o;
  return p;
}

}  // namespace c2k

using namespace c2k;

extern "C" {

size_t c2_kron_loglik_workspace_bytes(int64_t B, int64_t N, int64_t M, int64_t J, int method, int grad) {
  if (B < 1 || N < 1 || M < 1 || J < 1 || J > C2_FAST_WIDTH) return 0;
  if (method != C2_KRON_COLLAPSED && method != C2_KRON_INTERLEAVED) return 0;
  return plan(B, N, M, J, method, grad).total * sizeof(double);
}

static int kron_check(int64_t B, int64_t N, int64_t M, int64_t J, int method) {
  if (B < 1 || N < 1 || M < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH) return C2_ERR_UNSUPPORTED;
  if (method != C2_KRON_COLLAPSED && method != C2_KRON_INTERLEAVED) return C2_ERR_INVALID;
  if (M > INT32_MAX || B > 65535) return C2_ERR_UNSUPPORTED;  // grid.y
  return C2_OK;
}

int c2_kron_loglik(int64_t B, int64_t N, int64_t M, int64_t J, const double *t, int64_t t_bs, const double *c,
                   int64_t c_bs, const double *a, const double *U, const double *V, const double *alpha,
                   int64_t alpha_bs, const double *diag, const double *y, double *ll, int32_t *flag, int method,
                   void *work, size_t work_bytes, c2_stream_t stream) {
  if (int e = kron_check(B, N, M, J, method)) return e;
  if (!t || !c || !a || !U || !V || !alpha || !diag || !y || !ll || !flag || !work) return C2_ERR_INVALID;
  const Plan p = plan(B, N, M, J, method, 0);
  if (work_bytes < p.total * sizeof(double)) return C2_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  double *w = (double *)work;
  if (method == C2_KRON_INTERLEAVED) {
    const int64_t total = B * N * M * J;
    hipLaunchKernelGGL(k_kron_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, B, N, (int)M, (int)J, t,
                       t_bs, a, U, V, alpha, alpha_bs, diag, w + p.t2, w + p.a2, w + p.U2, w + p.V2);
    if (int e = launch_ok()) return e;
    return c2_loglik(B, N * M, J, w + p.t2, t_bs ? N * M : 0, c, c_bs, w + p.a2, w + p.U2, w + p.V2, y, ll, flag,
                     stream);
  }
  int32_t *badflag = reinterpret_cast<int32_t *>(w + p.badflag);
  if (hipMemsetAsync(badflag, 0, sizeof(int32_t) * (size_t)B, s) != hipSuccess) return C2_ERR_HIP;
  launch_collapse(B, N, M, a, alpha, alpha_bs, diag, y, w + p.a_eff, w + p.y_eff, w + p.part, badflag, p.nch, s);
  if (int e = launch_ok()) return e;
  if (int e = c2_loglik(B, N, J, t, t_bs, c, c_bs, w + p.a_eff, U, V, w + p.y_eff, ll, flag, stream)) return e;
  hipLaunchKernelGGL(k_kron_finish, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, p.nch, w + p.part, badflag,
                     ll, flag);
  return launch_ok();
}

int c2_kron_loglik_grad(int64_t B, int64_t N, int64_t M, int64_t J, const double *t, int64_t t_bs, const double *c,
                        int64_t c_bs, const double *a, const double *U, const double *V, const double *alpha,
                        int64_t alpha_bs, const double *diag, const double *y, double *ll, double *bt, double *bc,
                        double *ba, double *bU, double *bV, double *balpha, double *bdiag, double *by, int32_t *flag,
                        int method, void *work, size_t work_bytes, c2_stream_t stream) {
  if (int e = kron_check(B, N, M, J, method)) return e;
  if (!t || !c || !a || !U || !V || !alpha || !diag || !y || !ll || !flag || !work || !bt || !bc || !ba || !bU ||
      !bV || !balpha || !bdiag || !by)
    return C2_ERR_INVALID;
  const Plan p = plan(B, N, M, J, method, 1);
  if (work_bytes < p.total * sizeof(double)) return C2_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  double *w = (double *)work;
  if (method == C2_KRON_INTERLEAVED) {
    const int64_t R = N * M, total = B * R * J;
    hipLaunchKernelGGL(k_kron_expand, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, B, N, (int)M, (int)J, t,
                       t_bs, a, U, V, alpha, alpha_bs, diag, w + p.t2, w + p.a2, w + p.U2, w + p.V2);
    if (int e = launch_ok()) return e;
    // by of the interleaved series IS by (B, N, M); bc is shared by both views
    if (int e = c2_internal_loglik_grad_rows(B, R, J, w + p.t2, t_bs ? R : 0, c, c_bs, w + p.a2, w + p.U2, w + p.V2, y, ll,
                               w + p.bt2, bc, w + p.ba2, w + p.bU2, w + p.bV2, by, flag, w + p.one_d,
                               c2_loglik_grad_workspace_bytes(B, R, J), stream))
      return e;
    hipLaunchKernelGGL(k_kron_expand_rev, dim3((unsigned)p.nch, (unsigned)B), dim3(kThreads), 0, s, B, N, (int)M,
                       (int)J, a, U, V, alpha, alpha_bs, w + p.bt2, w + p.ba2, w + p.bU2, w + p.bV2, bt, ba, bU, bV,
                       bdiag, w + p.part, p.nch);
    if (int e = launch_ok()) return e;
  } else {
    int32_t *badflag = reinterpret_cast<int32_t *>(w + p.badflag);
    if (hipMemsetAsync(badflag, 0, sizeof(int32_t) * (size_t)B, s) != hipSuccess) return C2_ERR_HIP;
    // the GP-free partial sums reuse the head of the (B, nch, M) partial array: they are consumed by k_kron_finish
    // before k_kron_collapse_rev overwrites it
    launch_collapse(B, N, M, a, alpha, alpha_bs, diag, y, w + p.a_eff, w + p.y_eff, w + p.part, badflag, p.nch, s);
    if (int e = launch_ok()) return e;
    // a_eff = a + 1/A, so d ll / d a = d ll / d a_eff: the 1-D pass writes the caller's ba directly
    if (int e = c2_loglik_grad(B, N, J, t, t_bs, c, c_bs, w + p.a_eff, U, V, w + p.y_eff, ll, bt, bc, ba, bU, bV,
                               w + p.g_y, flag, w + p.one_d, c2_loglik_grad_workspace_bytes(B, N, J), stream))
      return e;
    hipLaunchKernelGGL(k_kron_finish, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, p.nch, w + p.part,
                       badflag, ll, flag);
    if (int e = launch_ok()) return e;
    launch_collapse_rev(B, N, M, alpha, alpha_bs, diag, y, ba, w + p.g_y, badflag, bdiag, by, w + p.part, p.nch, s);
    if (int e = launch_ok()) return e;
  }
  hipLaunchKernelGGL(k_kron_balpha, dim3((unsigned)((B * M + 255) / 256)), dim3(256), 0, s, B, (int)M, p.nch,
                     w + p.part, balpha);
  return launch_ok();
}

}  // extern "C"
