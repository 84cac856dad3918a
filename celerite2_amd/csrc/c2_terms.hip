// c2_terms.hip -- log-likelihood (+ gradient) straight from the kernel's celerite COEFFICIENTS:
// what a sampler differentiates is not (a, U, V) but the term coefficients (ar, cr, ac, bc, cc, dc), the times and
// the white-noise diagonal.  The reference gets there by autodiff of its term code in jax / pymc
// (python/celerite2/jax/terms.py, pymc/terms.py) around get_celerite_matrices (driver.cpp:422-477, terms.py:117-177);
// here the whole chain stays on the device:
//     coefficients --k_matrices--> (c, a, U, V) --c2_loglik[_grad]--> ll, (bt, bc, ba, bU, bV, by)
//                  --k_terms_rev--> (bar, bcr, bac, bbc, bcc, bdc, bx, bdiag, by)
// with the reverse of get_celerite_matrices written out by hand (columns Jr + 2k, Jr + 2k + 1 of complex term k, with
// s = sin(dc x), co = cos(dc x):  V = (co, s),  U = (ac co + bc s, ac s - bc co),  a = diag + sum ar + sum ac):
//     bar_r = sum_n (ba_n + bU_n[r]) ,   bcr_r = bc[r] ,   bcc_k = bc[i0] + bc[i1]
//     bac_k = sum_n (ba_n + bU0 co + bU1 s) ,   bbc_k = sum_n (bU0 s - bU1 co)
//     g_nk  = -bU0 U1 + bU1 U0 - bV0 s + bV1 co      (cotangent of the phase dc x_n)
//     bdc_k = sum_n g_nk x_n ,   bx_n = bt_n + sum_k g_nk dc_k ,   bdiag_n = ba_n.
// This file holds the COMPOSED form -- the matrices materialised once in the caller-provided workspace, the fused kernels
// reading them as they read a caller's -- and the dispatch between it and the kernels that form the rows in the lanes and
// fold this reverse into their reverse step (one lane per series: c2_loglik_t.hip; two: c2_loglik_k2.hip; four:
// c2_loglik_q4.hip; a group of J lanes: c2_loglik.hip).  The composed form serves small batches, widths other than 8, 4, 2 and
// the groups of 64 series a fused pair declines (backward guard, unsorted times, phases beyond the branch-free sincos): every
// kernel of it runs behind the gate words the fused pair left (profiles/r06_terms_lanes.md).
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
  const int J = Jr + 2 * Jc;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * J) return;
  const int64_t b = g / J;
  if (c2::gate_closed(gate, b)) return;
  const int j = (int)(g - b * J);
  c[g] = (j < Jr) ? cr[(coef_batched ? b * Jr : 0) + j] : cc[(coef_batched ? b * Jc : 0) + (j - Jr) / 2];
}

// Reverse of the matrix recipe (driver.cpp:456-474 transposed), ONE pass over (bt, ba, bU, bV, V) per series:
//     bx_n = bt_n + sum_k g_nk dc_k,  bdiag_n = ba_n,   g_nk = -bU0 U1 + bU1 U0 - bV0 sin + bV1 cos   (phase cotangent)
//     bar_r = sum_n (ba_n + bU_n[r]),  bac_k = sum_n (ba_n + bU0 cos + bU1 sin),  bbc_k = sum_n (bU0 sin - bU1 cos),
//     bdc_k = sum_n g_nk x_n,  bcr = bc[:Jr],  bcc_k = bc[Jr + 2k] + bc[Jr + 2k + 1].
// cos / sin of the phases are READ from V (the recipe stored them there), not recomputed.  One block per series; a
// thread owns one term and walks the rows (consecutive lanes = consecutive terms of a row: dense 64-byte-row reads), the
// terms of a row are summed across its lanes for bx, and the per-term sums are reduced over the block in a fixed order
// (deterministic).  (Two kernels -- one per (series, row), one block per (series, term) re-reading every row for its
// 16 bytes -- took 4.3 ms per 8192 x 4096 rows at J = 8 with the library sincos; this one 1.6 ms.)
__global__ __launch_bounds__(kThreads) void k_terms_rev(
    int64_t B, int64_t N, int Jr, int Jc, const double *__restrict__ ac, const double *__restrict__ bc,
    const double *__restrict__ dc, int coef_batched, const double *__restrict__ x, int64_t x_bs,
    const double *__restrict__ V, const double *__restrict__ bt, const double *__restrict__ bcv,
    const double *__restrict__ ba, const double *__restrict__ bU, const double *__restrict__ bV,
    double *__restrict__ bar, double *__restrict__ bcr, double *__restrict__ bac, double *__restrict__ bbc,
    double *__restrict__ bcc, double *__restrict__ bdc, double *__restrict__ bx, double *__restrict__ bdiag,
    const unsigned long long *__restrict__ gate, int nsplit, double *__restrict__ part) {
  // nsplit > 1 (a handful of long series): blockIdx.y takes a slice of the rows and leaves its sums in
  // part[series][slice][term][4]; k_terms_rev_finish adds the slices in order
  __shared__ double red[kThreads][4];
  if (c2::gate_none_closed(gate)) return;   // (the fallback of a fused path with nothing to do; block-uniform)
  const int Q = Jr + Jc, J = Jr + 2 * Jc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rpw = 64 / Q;                       // rows per wavefront and iteration
  const int q = lane % Q, r = lane / Q;
  const bool active = r < rpw;
  const int rows_it = (kThreads / 64) * rpw;    // rows per block iteration
  const int i0 = q < Jr ? q : Jr + 2 * (q - Jr);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    if (c2::gate_closed(gate, b)) continue;   // (block-uniform)
    const int64_t o = coef_batched ? b * Jc : 0;
    double a_ = 0.0, b_ = 0.0, d_ = 0.0;
    if (q >= Jr) { a_ = ac[o + q - Jr]; b_ = bc[o + q - Jr]; d_ = dc[o + q - Jr]; }
    const double *xb = x + b * x_bs;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, sba = 0.0;
    const int64_t per = ((N + nsplit - 1) / nsplit + rows_it - 1) / rows_it * rows_it;   // rows per slice
    const int64_t nlo = (int64_t)blockIdx.y * per, nhi = (nlo + per < N) ? nlo + per : N;
    for (int64_t nw = nlo + (int64_t)wave * rpw; nw < nhi; nw += rows_it) {   // wavefront-uniform bound: the shuffles below
      const int64_t n = nw + r;
      const bool valid = active && n < nhi;
      const int64_t row = b * N + (valid ? n : 0);
      double contrib = 0.0;
      if (valid) {
        if (q < Jr) {
          s0 += bU[row * J + i0];
        } else {
          const double cs = V[row * J + i0], sn = V[row * J + i0 + 1];
          const double g0 = bU[row * J + i0], g1 = bU[row * J + i0 + 1], h0 = bV[row * J + i0], h1 = bV[row * J + i0 + 1];
          const double u0 = a_ * cs + b_ * sn, u1 = a_ * sn - b_ * cs;
          const double g = -g0 * u1 + g1 * u0 - h0 * sn + h1 * cs;
          s0 += g0 * cs + g1 * sn;
          s1 += g0 * sn - g1 * cs;
          s2 = fma(g, xb[n], s2);
          contrib = g * d_;
        }
      }
      double tot = contrib;
      for (int sft = 1; sft < Q; ++sft) tot += __shfl_down(contrib, sft, 64);   // lane q == 0 collects its row
      if (valid && q == 0) {
        const double ban = ba[row];
        bx[row] = bt[row] + tot;
        bdiag[row] = ban;
        sba += ban;
      }
    }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2; red[threadIdx.x][3] = sba;
    __syncthreads();
    if ((int)threadIdx.x < Q) {   // thread t sums term t over every (wavefront, row slot), fixed order
      const int t = threadIdx.x;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, ab = 0.0;
      for (int w = 0; w < kThreads / 64; ++w)
        for (int rr = 0; rr < rpw; ++rr) {
          const int src = w * 64 + rr * Q;
          a0 += red[src + t][0]; a1 += red[src + t][1]; a2 += red[src + t][2]; ab += red[src][3];
        }
      if (nsplit > 1) {
        double *o = part + ((b * nsplit + blockIdx.y) * Q + t) * 4;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = ab;
      } else if (t < Jr) {
        bar[b * Jr + t] = ab + a0;
        bcr[b * Jr + t] = bcv[b * J + t];
      } else {
        const int k = t - Jr;
        bac[b * Jc + k] = ab + a0; bbc[b * Jc + k] = a1; bdc[b * Jc + k] = a2;
        bcc[b * Jc + k] = bcv[b * J + Jr + 2 * k] + bcv[b * J + Jr + 2 * k + 1];
      }
    }
    __syncthreads();
  }
}

// the slices of k_terms_rev added in order (one block per series, thread <-> term)
__global__ void k_terms_rev_finish(int Jr, int Jc, int nsplit, const double *__restrict__ part,
                                   const double *__restrict__ bcv, double *__restrict__ bar, double *__restrict__ bcr,
                                   double *__restrict__ bac, double *__restrict__ bbc, double *__restrict__ bcc,
                                   double *__restrict__ bdc, const unsigned long long *__restrict__ gate) {
  const int Q = Jr + Jc, J = Jr + 2 * Jc, t = threadIdx.x;
  const int64_t b = blockIdx.x;
  if (c2::gate_closed(gate, b)) return;
  if (t >= Q) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, ab = 0.0;
  for (int sp = 0; sp < nsplit; ++sp) {
    const double *o = part + ((b * nsplit + sp) * Q + t) * 4;
    a0 += o[0]; a1 += o[1]; a2 += o[2]; ab += o[3];
  }
  if (t < Jr) {
    bar[b * Jr + t] = ab + a0;
    bcr[b * Jr + t] = bcv[b * J + t];
  } else {
    const int k = t - Jr;
    bac[b * Jc + k] = ab + a0; bbc[b * Jc + k] = a1; bdc[b * Jc + k] = a2;
    bcc[b * Jc + k] = bcv[b * J + Jr + 2 * k] + bcv[b * J + Jr + 2 * k + 1];
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
  if (Jr + 2 * Jc > C2_FAST_WIDTH) return C2_ERR_UNSUPPORTED;
  return C2_OK;
}

}  // namespace c2terms

using namespace c2terms;

// One-lane-per-series kernels that generate U_n / V_n from the coefficients in the lane (c2_loglik_t.hip): no matrices in
// memory.  Widths 8, 4, 2.  C2_TERMS_FUSED=1 forces them, =0 disables them; otherwise batches that fill the chip.
// At width 8 the two-lane pair of c2_loglik_k2.hip has the same form (k_k2_tt_*) and takes the batches in between
// (C2_TERMS_TWO_LANES, profiles/r06_terms_lanes.md).
extern "C" {
#define C2_DECL_TT(J_)                                                                                                 \
  int c2_internal_loglik_tt##J_(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr, \
                                const double *ac, const double *bc, const double *cc, const double *dc,               \
                                const double *x, int64_t x_bs, const double *diag, const double *y, double *ll,       \
                                int32_t *flag, c2_stream_t stream);                                                   \
  int c2_internal_loglik_tt_grad##J_(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar,            \
                                     const double *cr, const double *ac, const double *bc, const double *cc,          \
                                     const double *dc, const double *x, int64_t x_bs, const double *diag,             \
                                     const double *y, double *ll, double *bar, double *bcr, double *bac, double *bbc, \
                                     double *bcc, double *bdc, double *bx, double *bdiag, double *by, int32_t *flag,  \
                                     double *rec, unsigned long long *guard, c2_stream_t stream);                     \
  size_t c2_internal_loglik_t_record_doubles##J_(int64_t B, int64_t N);
C2_DECL_TT(8)
C2_DECL_TT(4)
C2_DECL_TT(2)
#undef C2_DECL_TT
// the same with two lanes per series (c2_loglik_k2.hip, J == 8): batches that give the one-lane mapping half a chip
int c2_internal_loglik_k2_tt(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                             const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                             int64_t x_bs, const double *diag, const double *y, double *ll, int32_t *flag, c2_stream_t stream);
int c2_internal_loglik_k2_tt_grad(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                                  const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                                  int64_t x_bs, const double *diag, const double *y, double *ll, double *bar, double *bcr,
                                  double *bac, double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                                  int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream);
size_t c2_internal_loglik_k2_record_doubles(int64_t B, int64_t N);
// ... and with a group of J = 8, 4 or 2 lanes per series (c2_loglik.hip: k_loglik_fwd / k_loglik_rev<..., TT>): at most one
// wavefront per SIMD
size_t c2_internal_loglik_g8_tt_doubles(int64_t B, int64_t N, int64_t J);
int c2_internal_loglik_g8_tt_ok(int64_t B, int64_t N, int64_t J);
// ... and with four (c2_loglik_q4.hip: k_q4_fwd / k_q4_rev<..., TT>): the batches between the eight-lane and the two-lane range
size_t c2_internal_loglik_q4_record_doubles(int64_t B, int64_t N);
int c2_internal_loglik_q4_tt(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *ac, const double *bc,
                             const double *dc, const double *c, const double *x, int64_t x_bs, const double *diag, const double *y,
                             double *ll, int32_t *flag, unsigned long long *words, unsigned long long *guard, c2_stream_t stream);
size_t c2_internal_loglik_q4_span_words(int64_t B, int64_t N);
int c2_internal_loglik_q4_tt_grad(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *ac,
                                  const double *bc, const double *dc, const double *c, const double *x, int64_t x_bs,
                                  const double *diag, const double *y, double *ll, double *bar, double *bcr, double *bac,
                                  double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                                  int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream);
int c2_internal_loglik_g8_tt(int64_t B, int64_t N, int64_t J, int64_t Jc, int coef_batched, const double *ar, const double *ac,
                             const double *bc, const double *dc, const double *c, const double *x, int64_t x_bs,
                             const double *diag, const double *y, double *ll, int32_t *flag, unsigned long long *guard,
                             c2_stream_t stream);
int c2_internal_loglik_g8_gated(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
                                const unsigned long long *gate, c2_stream_t stream);
int c2_internal_loglik_g8_tt_grad(int64_t B, int64_t N, int64_t J, int64_t Jc, int coef_batched, const double *ar, const double *ac,
                                  const double *bc, const double *dc, const double *c, const double *x, int64_t x_bs,
                                  const double *diag, const double *y, double *ll, double *bar, double *bcr, double *bac,
                                  double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                                  int32_t *flag, double *work, unsigned long long *guard, c2_stream_t stream);
int c2_internal_matrices(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *ac,
                         const double *bc, const double *dc, int coef_batched, const double *x, int64_t x_bs,
                         const double *diag, double *a, double *U, double *V, const unsigned long long *gate,
                         c2_stream_t stream);
int c2_internal_loglik_grad_replay(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                   int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                   double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                   int32_t *flag, void *work, const unsigned long long *gate, c2_stream_t stream);
}
static bool fused_width(int64_t J) { return J == 8 || J == 4 || J == 2; }
// guard words in front of the fused kernels' records: the head + one per wavefront of 64 series, rounded to 16 bytes
static size_t fused_gate_words(int64_t B) { return (size_t)((c2::kGateHeadWords + (B + 63) / 64 + 1) & ~(int64_t)1); }
static size_t fused_record_doubles(int64_t B, int64_t N, int64_t J) {
  const size_t r3 = al2((size_t)B * J) + c2_internal_loglik_g8_tt_doubles(B, N, J);   // (the rates in front: Plan::c)
  if (J == 8) {   // any lane mapping
    const size_t r1 = c2_internal_loglik_t_record_doubles8(B, N), r2 = c2_internal_loglik_k2_record_doubles(B, N);
    const size_t r4 = al2((size_t)B * 8) + c2_internal_loglik_q4_record_doubles(B, N);
    size_t r = r1 > r2 ? r1 : r2;
    r = r > r3 ? r : r3;
    return r > r4 ? r : r4;
  }
  const size_t r1 = J == 4 ? c2_internal_loglik_t_record_doubles4(B, N) : c2_internal_loglik_t_record_doubles2(B, N);
  return r1 > r3 ? r1 : r3;
}
// Two lanes per series (J == 8) between the composed chain and the one-lane kernels: C2_TERMS_TWO_LANES=1 forces them, =0
// disables them; otherwise by batch size (a forced C2_TERMS_FUSED decides first)
static bool use_two_lanes(int64_t B, int64_t J, bool grad) {
  if (J != 8) return false;
  if (c2::opt::has(c2::opt::k_terms_two_lanes)) return c2::opt::ival(c2::opt::k_terms_two_lanes) != 0;
  if (c2::opt::has(c2::opt::k_terms_fused)) return false;
  return B >= c2::opt::ival(grad ? c2::opt::k_terms_two_lanes_min_batch_grad : c2::opt::k_terms_two_lanes_min_batch_fwd) &&
         B < c2::opt::ival(grad ? c2::opt::k_terms_two_lanes_max_batch_grad : c2::opt::k_terms_two_lanes_max_batch_fwd);
}
static bool use_fused(int64_t B, int64_t J, bool grad) {
  if (!fused_width(J)) return false;
  if (c2::opt::has(c2::opt::k_terms_fused)) return c2::opt::ival(c2::opt::k_terms_fused) != 0;
  return B >= c2::opt::ival(grad ? c2::opt::k_terms_fused_min_batch_grad : c2::opt::k_terms_fused_min_batch_fwd);
}

// Eight lanes per series (J == 8, gradient) below the two-lane range: C2_TERMS_EIGHT_LANES=1 forces, =0 disables; otherwise
// by batch size (a forced C2_TERMS_FUSED / C2_TERMS_TWO_LANES decides first)
// (the gradient's threshold is quoted in series at J = 8 -- eight series per wavefront -- and scales with the series per wavefront:
// the crossovers against the composed chain measured at widths 4 and 2 sit at the same number of wavefronts)
static bool use_eight_lanes(int64_t B, int64_t N, int64_t J, bool grad) {
  if (!fused_width(J)) return false;
  if (grad && !c2_internal_loglik_g8_tt_ok(B, N, J)) return false;
  if (c2::opt::has(c2::opt::k_terms_eight_lanes)) return c2::opt::ival(c2::opt::k_terms_eight_lanes) != 0;
  if (c2::opt::has(c2::opt::k_terms_fused) || c2::opt::has(c2::opt::k_terms_two_lanes)) return false;
  const int64_t B8 = B * J / 8;
  if (grad) {
    if (J == 2 && B >= c2::opt::ival(c2::opt::k_terms_group_max_batch_grad_j2)) return false;   // (one lane per series from there)
    return B8 >= c2::opt::ival(c2::opt::k_terms_eight_lanes_min_batch_grad);
  }
  if (J == 2) return false;   // width 2 forward: the composed chain, then one lane per series (the group form never leads)
  const int64_t lo = J == 8 ? c2::opt::ival(c2::opt::k_terms_eight_lanes_min_batch_fwd) : c2::opt::ival(c2::opt::k_terms_group_min_batch_fwd_j4);
  return B >= lo && B8 < c2::opt::ival(c2::opt::k_terms_eight_lanes_max_batch_fwd);
}

// Four lanes per series (J == 8, gradient): C2_TERMS_FOUR_LANES=1 forces, =0 disables; otherwise by batch size (any other forced
// mapping decides first)
static bool use_four_lanes(int64_t B, int64_t J, bool grad = true) {
  if (J != 8) return false;
  if (c2::opt::has(c2::opt::k_terms_four_lanes)) return c2::opt::ival(c2::opt::k_terms_four_lanes) != 0;
  if (c2::opt::has(c2::opt::k_terms_fused) || c2::opt::has(c2::opt::k_terms_two_lanes) || c2::opt::has(c2::opt::k_terms_eight_lanes))
    return false;
  if (!grad) return B >= c2::opt::ival(c2::opt::k_terms_four_lanes_min_batch_fwd) && B < c2::opt::ival(c2::opt::k_terms_four_lanes_max_batch_fwd);
  return B >= c2::opt::ival(c2::opt::k_terms_four_lanes_min_batch_grad) && B < c2::opt::ival(c2::opt::k_terms_four_lanes_max_batch_grad);
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
  if (grad && fused_width(J)) {
    const size_t r = fused_record_doubles(B, N, J);
    n = fused_gate_words(B) + (r > n ? r : n);
  } else if (!grad && fused_width(J)) {
    n += fused_gate_words(B);   // the group-mapping forward form: one word per group of 64 series in front of the plan
    if (J == 8) n += al2(c2_internal_loglik_q4_span_words(B, N));   // ... and the four-lane form's span words behind it
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
  const bool four_f = use_four_lanes(B, J, false), eight_f = !four_f && use_eight_lanes(B, N, J, false);
  if (four_f || eight_f) {
    // every lane forms its column of U, V (k_loglik_fwd<..., TT>); a group of 64 series with a phase beyond the branch-free
    // sincos is closed in the gate words and goes through matrices in memory, every kernel of that chain behind the same words
    unsigned long long *guard = (unsigned long long *)work;
    double *w = (double *)work + fused_gate_words(B);
    hipStream_t s = (hipStream_t)stream;
    const Plan p = plan(B, N, J, 0);
    hipLaunchKernelGGL(k_rates, dim3((unsigned)((B * J + 255) / 256)), dim3(256), 0, s, B, (int)Jr, (int)Jc, cr, cc, coef_batched,
                       w + p.c, (const unsigned long long *)nullptr);
    if (int e = launch_ok()) return e;
    if (four_f) {
      unsigned long long *words = (unsigned long long *)(w + p.total);   // (behind the plan: k_anchor_spans' scratch)
      if (int e = c2_internal_loglik_q4_tt(B, N, Jc, coef_batched, ar, ac, bc, dc, w + p.c, x, x_bs, diag, y, ll, flag, words, guard, stream))
        return e;
    } else if (int e = c2_internal_loglik_g8_tt(B, N, J, Jc, coef_batched, ar, ac, bc, dc, w + p.c, x, x_bs, diag, y, ll, flag, guard,
                                                stream))
      return e;
    const unsigned long long *gate = c2::gate_per_wave(guard + c2::kGateHeadWords);
    if (int e = matrices(B, N, Jr, Jc, ar, cr, ac, bc, cc, dc, coef_batched, x, x_bs, diag, w, p, gate, s)) return e;
    return c2_internal_loglik_g8_gated(B, N, J, x, x_bs, w + p.c, J, w + p.a, w + p.U, w + p.V, y, ll, flag, gate, stream);
  }
  if (use_two_lanes(B, J, false))
    return c2_internal_loglik_k2_tt(B, N, Jc, coef_batched, ar, cr, ac, bc, cc, dc, x, x_bs, diag, y, ll, flag, stream);
  if (use_fused(B, J, false))
    return (J == 8 ? c2_internal_loglik_tt8 : (J == 4 ? c2_internal_loglik_tt4 : c2_internal_loglik_tt2))(
        B, N, Jc, coef_batched, ar, cr, ac, bc, cc, dc, x, x_bs, diag, y, ll, flag, stream);
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
  const bool four = use_four_lanes(B, J);
  const bool eight = !four && use_eight_lanes(B, N, J, true);
  const bool two = !four && !eight && use_two_lanes(B, J, true);
  if (eight || four) {
    // the eight-lane pair with the rows formed in the lanes; its gate words (written by the launch) send a group of 64 series
    // it declines -- a span beyond the backward guard, a phase beyond the branch-free sincos -- to the composed chain below
    unsigned long long *guard = (unsigned long long *)work;
    w += fused_gate_words(B);
    hipLaunchKernelGGL(k_rates, dim3((unsigned)((B * J + 255) / 256)), dim3(256), 0, s, B, (int)Jr, (int)Jc, cr, cc, coef_batched,
                       w + p.c, (const unsigned long long *)nullptr);
    if (int e = launch_ok()) return e;
    if (int e = four ? c2_internal_loglik_q4_tt_grad(B, N, Jc, coef_batched, ar, ac, bc, dc, w + p.c, x, x_bs, diag, y, ll, bar, bcr,
                                                     bac, bbc, bcc, bdc, bx, bdiag, by, flag, w + al2((size_t)B * 8), guard, stream)
                     : c2_internal_loglik_g8_tt_grad(B, N, J, Jc, coef_batched, ar, ac, bc, dc, w + p.c, x, x_bs, diag, y, ll, bar,
                                                     bcr, bac, bbc, bcc, bdc, bx, bdiag, by, flag, w + al2((size_t)B * J), guard,
                                                     stream))
      return e;
    gate = c2::gate_per_wave(guard + c2::kGateHeadWords);
  } else if (two || use_fused(B, J, true)) {
    // Forward with records + reverse sweep, both forming the rows in the lane.  As in c2_loglik_grad, the forward pass
    // leaves its stability measure in `guard`; if it exceeds kBackwardGuard the reverse sweep returns at once and the
    // composed chain below -- every kernel of it behind the same word -- produces the gradients instead.
    unsigned long long *guard = (unsigned long long *)work;
    // (two lanes per series: the two wavefronts of a group of 64 series raise its word together -- every word starts at zero)
    if (hipMemsetAsync(guard, 0, 8 * (two ? fused_gate_words(B) : (size_t)c2::kGateHeadWords), s) != hipSuccess) return C2_ERR_HIP;
    w += fused_gate_words(B);
    auto fused = two ? c2_internal_loglik_k2_tt_grad
                     : (J == 8 ? c2_internal_loglik_tt_grad8 : (J == 4 ? c2_internal_loglik_tt_grad4 : c2_internal_loglik_tt_grad2));
    if (int e = fused(B, N, Jc, coef_batched, ar, cr, ac, bc, cc, dc, x, x_bs, diag, y, ll, bar, bcr, bac, bbc, bcc, bdc,
                      bx, bdiag, by, flag, w, guard, stream))
      return e;
    gate = c2::gate_per_wave(guard + c2::kGateHeadWords);   // per group of 64 series (c2_loglik_helpers.hpp)
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
  // a handful of long series: slices of the rows in parallel (partial sums in the 1-D workspace, free by now)
  int nsplit = 1;
  if (B < 64 && N >= 8192) {
    nsplit = (int)((N + 2047) / 2048);
    if (nsplit > 256) nsplit = 256;
  }
  double *part = w + p.one_d;
  // (behind a fused path -- gated -- a small grid striding over the series: most of its groups are closed to it)
  const int64_t trb = gate ? (B < 2048 ? B : 2048) : (B < 0x7fffffff ? B : 0x7fffffff);
  hipLaunchKernelGGL(k_terms_rev, dim3((unsigned)trb, (unsigned)nsplit), dim3(kThreads), 0, s,
                     B, N, (int)Jr, (int)Jc, ac, bc, dc, coef_batched, x, x_bs, (const double *)(w + p.V),
                     (const double *)(w + p.bt), (const double *)(w + p.bc), (const double *)(w + p.ba),
                     (const double *)(w + p.bU), (const double *)(w + p.bV), bar, bcr, bac, bbc, bcc, bdc, bx, bdiag, gate,
                     nsplit, part);
  if (nsplit > 1)
    hipLaunchKernelGGL(k_terms_rev_finish, dim3((unsigned)B), dim3(64), 0, s, (int)Jr, (int)Jc, nsplit,
                       (const double *)part, (const double *)(w + p.bc), bar, bcr, bac, bbc, bcc, bdc, gate);
  return launch_ok();
}

}  // extern "C"
