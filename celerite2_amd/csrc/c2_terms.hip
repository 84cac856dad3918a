// c2_terms.hip -- log-likelihood (+ gradient) straight from the kernel's celerite COEFFICIENTS:
// what a sampler differentiates is not (a, U, V) but the term coefficients (ar, cr, ac, bc, cc, dc), the times and
// the white-noise diagonal.  The reference gets there by autodiff of its term code in jax / pymc
// (python/celerite2/jax/terms.py, pymc/terms.py) around get_celerite_matrices (driver.cpp:422-477, terms.py:117-177);
// here the whole chain stays on the device:
//     coefficients --k_matrices--> (c, a, U, V) --c2_loglik[_grad]--> ll, (bt, bc, ba, bU, bV, by)
//                  --k_terms_rev_rows / k_terms_rev_coef--> (bar, bcr, bac, bbc, bcc, bdc, bx, bdiag, by)
// with the reverse of get_celerite_matrices written out by hand (columns Jr + 2k, Jr + 2k + 1 of complex term k, with
// s = sin(dc x), co = cos(dc x):  V = (co, s),  U = (ac co + bc s, ac s - bc co),  a = diag + sum ar + sum ac):
//     bar_r = sum_n (ba_n + bU_n[r]) ,   bcr_r = bc[r] ,   bcc_k = bc[i0] + bc[i1]
//     bac_k = sum_n (ba_n + bU0 co + bU1 s) ,   bbc_k = sum_n (bU0 s - bU1 co)
//     g_nk  = -bU0 U1 + bU1 U0 - bV0 s + bV1 co      (cotangent of the phase dc x_n)
//     bdc_k = sum_n g_nk x_n ,   bx_n = bt_n + sum_k g_nk dc_k ,   bdiag_n = ba_n.
// This is the COMPOSED form: the matrices are materialised once in the caller-provided workspace (the fused kernels
// read them as they read a caller's); folding the generation into the recursion kernels themselves is the next step
// (DESIGN.md section 8).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

extern "C" void c2_internal_set_error(const char *msg);

namespace c2terms {

constexpr int kThreads = 256;

// c (B, J) = [cr, cc0, cc0, cc1, cc1, ...]   (terms.py:171-173)
__global__ void k_rates(int64_t B, int Jr, int Jc, const double *__restrict__ cr, const double *__restrict__ cc,
                        int coef_batched, double *__restrict__ c, const unsigned long long *__restrict__ gate) {
  if (c2::gate_closed(gate)) return;
  const int J = Jr + 2 * Jc;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * J) return;
  const int64_t b = g / J;
  const int j = (int)(g - b * J);
  c[g] = (j < Jr) ? cr[(coef_batched ? b * Jr : 0) + j] : cc[(coef_batched ? b * Jc : 0) + (j - Jr) / 2];
}

// One thread per (series, row): bx_n = bt_n + sum_k g_nk dc_k, bdiag_n = ba_n.
__global__ void k_terms_rev_rows(int64_t B, int64_t N, int Jr, int Jc, const double *__restrict__ ac,
                                 const double *__restrict__ bc, const double *__restrict__ dc, int coef_batched,
                                 const double *__restrict__ x, int64_t x_bs, const double *__restrict__ bt,
                                 const double *__restrict__ ba, const double *__restrict__ bU,
                                 const double *__restrict__ bV, double *__restrict__ bx, double *__restrict__ bdiag,
                                 const unsigned long long *__restrict__ gate) {
  if (c2::gate_closed(gate)) return;
  const int J = Jr + 2 * Jc;
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B * N) return;
  const int64_t b = row / N, n = row - b * N;
  const double xn = x[b * x_bs + n];
  const double *acb = ac + (coef_batched ? b * Jc : 0), *bcb = bc + (coef_batched ? b * Jc : 0),
               *dcb = dc + (coef_batched ? b * Jc : 0);
  double s = bt[row];
  for (int k = 0; k < Jc; ++k) {
    double sn, co;
    sincos(dcb[k] * xn, &sn, &co);
    const int i0 = Jr + 2 * k;
    const double u0 = acb[k] * co + bcb[k] * sn, u1 = acb[k] * sn - bcb[k] * co;
    const double g = -bU[row * J + i0] * u1 + bU[row * J + i0 + 1] * u0 - bV[row * J + i0] * sn + bV[row * J + i0 + 1] * co;
    s = fma(g, dcb[k], s);
  }
  bx[row] = s;
  bdiag[row] = ba[row];
}

__device__ __forceinline__ double block_sum(double v, double *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kThreads / 64; ++i) s += red[i];
  return s;
}

// One block per (series, term): the sums over the rows (fixed order -> deterministic).
__global__ __launch_bounds__(kThreads) void k_terms_rev_coef(
    int64_t B, int64_t N, int Jr, int Jc, const double *__restrict__ ac, const double *__restrict__ bc,
    const double *__restrict__ dc, int coef_batched, const double *__restrict__ x, int64_t x_bs,
    const double *__restrict__ bcv, const double *__restrict__ ba, const double *__restrict__ bU,
    const double *__restrict__ bV, double *__restrict__ bar, double *__restrict__ bcr, double *__restrict__ bac,
    double *__restrict__ bbc, double *__restrict__ bcc, double *__restrict__ bdc,
    const unsigned long long *__restrict__ gate) {
  __shared__ double red[kThreads / 64];
  if (c2::gate_closed(gate)) return;
  const int J = Jr + 2 * Jc;
  const int64_t b = blockIdx.x;
  const int q = blockIdx.y;  // term: real terms first
  const double *bab = ba + b * N, *bUb = bU + b * N * J, *bVb = bV + b * N * J;
  if (q < Jr) {
    double s = 0.0;
    for (int64_t n = threadIdx.x; n < N; n += kThreads) s += bab[n] + bUb[n * J + q];
    s = block_sum(s, red);
    if (threadIdx.x == 0) { bar[b * Jr + q] = s; bcr[b * Jr + q] = bcv[b * J + q]; }
    return;
  }
  const int k = q - Jr, i0 = Jr + 2 * k;
  const double a_ = ac[(coef_batched ? b * Jc : 0) + k], b_ = bc[(coef_batched ? b * Jc : 0) + k],
               d_ = dc[(coef_batched ? b * Jc : 0) + k];
  double sa = 0.0, sb = 0.0, sd = 0.0;
  for (int64_t n = threadIdx.x; n < N; n += kThreads) {
    const double xn = x[b * x_bs + n];
    double sn, co;
    sincos(d_ * xn, &sn, &co);
    const double u0 = a_ * co + b_ * sn, u1 = a_ * sn - b_ * co;
    const double g0 = bUb[n * J + i0], g1 = bUb[n * J + i0 + 1], h0 = bVb[n * J + i0], h1 = bVb[n * J + i0 + 1];
    sa += bab[n] + g0 * co + g1 * sn;
    sb += g0 * sn - g1 * co;
    sd = fma(-g0 * u1 + g1 * u0 - h0 * sn + h1 * co, xn, sd);
  }
  sa = block_sum(sa, red);
  sb = block_sum(sb, red);
  sd = block_sum(sd, red);
  if (threadIdx.x == 0) {
    bac[b * Jc + k] = sa; bbc[b * Jc + k] = sb; bdc[b * Jc + k] = sd;
    bcc[b * Jc + k] = bcv[b * J + i0] + bcv[b * J + i0 + 1];
  }
}

inline int launch_ok() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return C2_OK;
  c2_internal_set_error(hipGetErrorString(e));
  return C2_ERR_HIP;
}
inline size_t al2(size_t n) { return (n + 1) & ~(size_t)1; }

struct Plan {
  size_t c, a, U, V, bt, bc, ba, bU, bV, one_d, total;  // doubles
};
inline Plan plan(int64_t B, int64_t N, int64_t J, int grad) {
  Plan p{};
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += al2(n); return at; };
  p.c = take((size_t)B * J);
  p.a = take((size_t)B * N);
  p.U = take((size_t)B * N * J);
  p.V = take((size_t)B * N * J);
  if (grad) {
    p.bt = take((size_t)B * N);
    p.bc = take((size_t)B * J);
    p.ba = take((size_t)B * N);
    p.bU = take((size_t)B * N * J);
    p.bV = take((size_t)B * N * J);
    p.one_d = take(c2_loglik_grad_workspace_bytes(B, N, J) / sizeof(double));
  }
  p.total = o;
  return p;
}

inline int check(int64_t B, int64_t N, int64_t Jr, int64_t Jc) {
  if (B < 1 || N < 1 || Jr < 0 || Jc < 0 || Jr + 2 * Jc < 1) return C2_ERR_INVALID;
  if (Jr + 2 * Jc > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  return C2_OK;
}

}  // namespace c2terms

using namespace c2terms;

// One-lane-per-series kernels that generate U_n / V_n from the coefficients in the lane (c2_loglik_t.hip): no matrices in
// memory.  J == 8 only.  C2_TERMS_FUSED=1 forces them, =0 disables them; otherwise batches that fill the chip.
extern "C" {
int c2_internal_loglik_tt(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                          const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                          int64_t x_bs, const double *diag, const double *y, double *ll, int32_t *flag,
                          c2_stream_t stream);
int c2_internal_loglik_tt_grad(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                               const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                               int64_t x_bs, const double *diag, const double *y, double *ll, double *bar, double *bcr,
                               double *bac, double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                               int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream);
size_t c2_internal_loglik_t_record_doubles(int64_t B, int64_t N);
int c2_internal_matrices(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *ac,
                         const double *bc, const double *dc, int coef_batched, const double *x, int64_t x_bs,
                         const double *diag, double *a, double *U, double *V, const unsigned long long *gate,
                         c2_stream_t stream);
int c2_internal_loglik_grad_replay(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                   int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                   double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                   int32_t *flag, void *work, const unsigned long long *gate, c2_stream_t stream);
}
#ifndef C2_TERMS_FUSED_MIN_BATCH_FWD
#define C2_TERMS_FUSED_MIN_BATCH_FWD 16384
#endif
#ifndef C2_TERMS_FUSED_MIN_BATCH_GRAD
#define C2_TERMS_FUSED_MIN_BATCH_GRAD 16384
#endif
static bool use_fused(int64_t B, int64_t J, bool grad) {
  if (J != 8) return false;
  const char *e = getenv("C2_TERMS_FUSED");
  if (e) return atoi(e) != 0;
  return B >= (grad ? C2_TERMS_FUSED_MIN_BATCH_GRAD : C2_TERMS_FUSED_MIN_BATCH_FWD);
}

static int matrices(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *cr, const double *ac,
                    const double *bc, const double *cc, const double *dc, int coef_batched, const double *x,
                    int64_t x_bs, const double *diag, double *w, const Plan &p, const unsigned long long *gate,
                    hipStream_t s) {
  const int64_t J = Jr + 2 * Jc;
  hipLaunchKernelGGL(k_rates, dim3((unsigned)((B * J + 255) / 256)), dim3(256), 0, s, B, (int)Jr, (int)Jc, cr, cc,
                     coef_batched, w + p.c, gate);
  if (int e = launch_ok()) return e;
  return c2_internal_matrices(B, N, Jr, Jc, ar, ac, bc, dc, coef_batched, x, x_bs, diag, w + p.a, w + p.U, w + p.V, gate,
                              (c2_stream_t)s);
}

extern "C" {

// Either variant fits: [guard word, 16 bytes] [records of the fused kernels | plan of the composed chain]; the choice
// between them is made per call (batch size, C2_TERMS_FUSED), the size does not depend on it.
size_t c2_loglik_terms_workspace_bytes(int64_t B, int64_t N, int64_t Jr, int64_t Jc, int grad) {
  if (check(B, N, Jr, Jc)) return 0;
  const int64_t J = Jr + 2 * Jc;
  size_t n = plan(B, N, J, grad).total;
  if (grad && J == 8) {
    const size_t r = c2_internal_loglik_t_record_doubles(B, N);
    n = 2 + (r > n ? r : n);
  }
  return n * sizeof(double);
}

int c2_loglik_terms(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *cr, const double *ac,
                    const double *bc, const double *cc, const double *dc, int coef_batched, const double *x,
                    int64_t x_bs, const double *diag, const double *y, double *ll, int32_t *flag, void *work,
                    size_t work_bytes, c2_stream_t stream) {
  if (int e = check(B, N, Jr, Jc)) return e;
  if (!x || !diag || !y || !ll || !flag || !work || (Jr && (!ar || !cr)) || (Jc && (!ac || !bc || !cc || !dc)))
    return C2_ERR_INVALID;
  const int64_t J = Jr + 2 * Jc;
  if (work_bytes < c2_loglik_terms_workspace_bytes(B, N, Jr, Jc, 0)) return C2_ERR_INVALID;
  if (use_fused(B, J, false))
    return c2_internal_loglik_tt(B, N, Jc, coef_batched, ar, cr, ac, bc, cc, dc, x, x_bs, diag, y, ll, flag, stream);
  const Plan p = plan(B, N, J, 0);
  double *w = (double *)work;
  hipStream_t s = (hipStream_t)stream;
  if (int e = matrices(B, N, Jr, Jc, ar, cr, ac, bc, cc, dc, coef_batched, x, x_bs, diag, w, p, nullptr, s)) return e;
  return c2_loglik(B, N, J, x, x_bs, w + p.c, J, w + p.a, w + p.U, w + p.V, y, ll, flag, stream);
}

int c2_loglik_terms_grad(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *cr,
                         const double *ac, const double *bc, const double *cc, const double *dc, int coef_batched,
                         const double *x, int64_t x_bs, const double *diag, const double *y, double *ll, double *bar,
                         double *bcr, double *bac, double *bbc, double *bcc, double *bdc, double *bx, double *bdiag,
                         double *by, int32_t *flag, void *work, size_t work_bytes, c2_stream_t stream) {
  if (int e = check(B, N, Jr, Jc)) return e;
  if (!x || !diag || !y || !ll || !flag || !work || !bx || !bdiag || !by || (Jr && (!ar || !cr || !bar || !bcr)) ||
      (Jc && (!ac || !bc || !cc || !dc || !bac || !bbc || !bcc || !bdc)))
    return C2_ERR_INVALID;
  const int64_t J = Jr + 2 * Jc;
  if (work_bytes < c2_loglik_terms_workspace_bytes(B, N, Jr, Jc, 1)) return C2_ERR_INVALID;
  const Plan p = plan(B, N, J, 1);
  double *w = (double *)work;
  hipStream_t s = (hipStream_t)stream;
  const unsigned long long *gate = nullptr;
  if (use_fused(B, J, true)) {
    // Forward with records + reverse sweep, both forming the rows in the lane.  As in c2_loglik_grad, the forward pass
    // leaves its stability measure in `guard`; if it exceeds kBackwardGuard the reverse sweep returns at once and the
    // composed chain below -- every kernel of it behind the same word -- produces the gradients instead.
    unsigned long long *guard = (unsigned long long *)work;
    if (hipMemsetAsync(guard, 0, 16, s) != hipSuccess) return C2_ERR_HIP;
    w += 2;
    if (int e = c2_internal_loglik_tt_grad(B, N, Jc, coef_batched, ar, cr, ac, bc, cc, dc, x, x_bs, diag, y, ll, bar, bcr,
                                           bac, bbc, bcc, bdc, bx, bdiag, by, flag, w, guard, stream))
      return e;
    gate = guard;
  }
  if (int e = matrices(B, N, Jr, Jc, ar, cr, ac, bc, cc, dc, coef_batched, x, x_bs, diag, w, p, gate, s)) return e;
  if (gate) {
    if (int e = c2_internal_loglik_grad_replay(B, N, J, x, x_bs, w + p.c, J, w + p.a, w + p.U, w + p.V, y, ll, w + p.bt,
                                               w + p.bc, w + p.ba, w + p.bU, w + p.bV, by, flag, w + p.one_d, gate,
                                               stream))
      return e;
  } else if (int e = c2_loglik_grad(B, N, J, x, x_bs, w + p.c, J, w + p.a, w + p.U, w + p.V, y, ll, w + p.bt, w + p.bc,
                                    w + p.ba, w + p.bU, w + p.bV, by, flag, w + p.one_d,
                                    c2_loglik_grad_workspace_bytes(B, N, J), stream))
    return e;
  hipLaunchKernelGGL(k_terms_rev_rows, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, s, B, N, (int)Jr, (int)Jc,
                     ac, bc, dc, coef_batched, x, x_bs, (const double *)(w + p.bt), (const double *)(w + p.ba),
                     (const double *)(w + p.bU), (const double *)(w + p.bV), bx, bdiag, gate);
  if (int e = launch_ok()) return e;
  hipLaunchKernelGGL(k_terms_rev_coef, dim3((unsigned)B, (unsigned)(Jr + Jc)), dim3(kThreads), 0, s, B, N, (int)Jr,
                     (int)Jc, ac, bc, dc, coef_batched, x, x_bs, (const double *)(w + p.bc), (const double *)(w + p.ba),
                     (const double *)(w + p.bU), (const double *)(w + p.bV), bar, bcr, bac, bbc, bcc, bdc, gate);
  return launch_ok();
}

}  // extern "C"
