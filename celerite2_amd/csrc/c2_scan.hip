// c2_scan.hip -- time-parallel matmul_lower / matmul_upper for LONG series (BASELINE config 4:
// one series, N = 10^7, J = 16, nrhs = 32 `dot_tril` posterior draws).
//
// The matmul sweeps (reference internal.hpp:105-146 / 148-189 with is_solve = false) are LINEAR recurrences
// with a DIAGONAL transition:   G_{n+1} = p_n o G_n + V_n^T y_n ,  Z_n += U_n (p_n o G_n)      (lower)
// where G_n is the reference's workspace row n (F before the decay, internal.hpp:142).  A chunk [s, e] of
// rows therefore maps its incoming state affinely, G_{e+1} = E_c + D_c o G_s, with
//   E_c = the chunk's outgoing state for a zero incoming state,   D_c = prod p = exp(c (t_{s-1} - t_e)).
// Three passes turn the N-step chain into N/Lc independent chains:
//   1. k_mm_chunk<LOCAL>: every chunk computes E_c (zero carry-in), no output;
//   2. k_mm_carry       : one thread per (series, j, k) folds the chunk carries in sequence (n_chunk steps);
//   3. k_mm_chunk<FINAL>: every chunk replays its rows with the true carry-in and writes Z (and F).
// Only decays exp(-c |dt|) <= 1 are ever formed, so there is nothing to overflow (unlike the
// exp(+c (t - t0)) factors a matrix/MFMA formulation of the chunk would need).
// Each chunk only reads Y rows of its own range (the boundary term V_{s-1}^T y_{s-1} lives in the carry),
// so Y == Z in-place use (numpy.py:100-102 dot_tril) stays legal.
// The solve sweeps have a dense J x J transition (I - W^T U) per step and stay sequential.
#include <cstdint>
#include <type_traits>

#include "c2_common.hpp"
#include "c2_rscatter.hpp"
#include "../../include/celerite2_amd.h"

namespace c2 {

// unit = (series b, chunk ch).  A unit is walked by G x KG lanes: lane (kg, j) owns state elements
// G_n(j, k) for the KT columns k = kt0 + kg*KT .. +KT-1, so that t / U / V rows and the decay are fetched and
// computed once per unit for up to KG*KT right-hand sides (config 4: G = 16, KG = 4, KT = 8 -> one unit per
// wavefront, all 32 columns).  blockIdx.y covers nrhs in slabs of KG*KT columns.
template <int G, int KG, int KT, bool LOWER, bool FINAL, bool FAST>
__global__ __launch_bounds__(kWave) void k_mm_chunk(int64_t B, int64_t N, int J, int64_t nrhs, int64_t Lc,
                                                    int64_t nchunk, const double *__restrict__ t, int64_t t_bs,
                                                    const double *__restrict__ c, int64_t c_bs,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    const double *Y, double *Z, double *F, double *carry, int zero_z) {
  constexpr int UL = G * KG;  // lanes per unit
  constexpr int UPW = kWave / UL;  // units per wavefront (1 -> unit index, hence t addresses, are wave-uniform)
  const int lane = threadIdx.x;
  int64_t unit = (int64_t)blockIdx.x * UPW + lane / UL;
  const int j = lane % G;
  const int kg = (lane / G) % KG;
  const int64_t nunit = B * nchunk;
  const bool valid = unit < nunit;
  if (!valid) unit = nunit - 1;
  const int64_t b = unit / nchunk, ch = unit % nchunk;
  const bool act = j < J;
  const int jj = act ? j : 0;
  const int64_t k0 = ((int64_t)blockIdx.y * KG + kg) * KT;
  const int kn = (nrhs - k0 < KT) ? (int)((nrhs - k0 > 0) ? nrhs - k0 : 0) : KT;
  const int64_t k0c = (k0 < nrhs) ? k0 : 0;  // keep addresses in range for idle column groups
  const bool st = valid && act && kn > 0, st0 = valid && j == 0 && kn > 0;
  const double *tb = t + b * t_bs;
  const double *Ab = (LOWER ? V : U) + b * N * J + jj;  // row fed into the state
  const double *Bb = (LOWER ? U : V) + b * N * J + jj;  // row applied to the state
  const double *Yb = Y + b * N * nrhs + k0c;
  double *Zb = Z + b * N * nrhs + k0c;
  double *Fb = (FINAL && F) ? F + b * N * J * nrhs + jj + J * k0c : nullptr;
  double *cb = carry + ((b * nchunk + ch) * J + jj) * nrhs + k0c;
  const double cj = act ? c[b * c_bs + j] : 0.0;
  const int64_t s = ch * Lc, e = (s + Lc < N ? s + Lc : N) - 1;  // rows of this chunk
  const int64_t len = e - s + 1;

  double Gk[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) Gk[k] = (FINAL && act && k < kn) ? cb[k] : 0.0;

  // software prefetch of the next row (plenty of wavefronts per SIMD here; one row ahead is enough)
  auto row_of = [&](int64_t i) { return LOWER ? s + i : e - i; };
  int64_t n = row_of(0);
  double an = act ? Ab[n * J] : 0.0, bn = act ? Bb[n * J] : 0.0, tn = tb[n];
  double tm = LOWER ? (n >= 1 ? tb[n - 1] : tn) : (n <= N - 2 ? tb[n + 1] : tn);
  // FAST: the KT = 8 values of a column group are fetched by ONE instruction (lane j takes column j & 7; the
  // other lanes of the group duplicate) and broadcast to the group through LDS -- the CU's address unit, not
  // HBM, is what a per-lane 64-byte row read would saturate (profiles/r01_ubench_instruction_costs.md).
  __shared__ __attribute__((aligned(16))) double ybuf[kWave / 8 * 8 + 8];
  double yk[KT];
  double ypre = 0.0;  // FAST: this lane's column of the prefetched row
  const int yslot = (lane / G) * 8;  // 8 doubles per (unit, column group)
  auto fetch_y = [&](int64_t row) { return Yb[row * nrhs + (j & 7)]; };
  auto spread_y = [&](double (&dst)[KT], double mine) {
    ybuf[yslot + (j & 7)] = mine;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double2 *y2 = reinterpret_cast<const double2 *>(ybuf + yslot);
#pragma unroll
    for (int k = 0; k < KT / 2; ++k) { const double2 v2 = y2[k]; dst[2 * k] = v2.x; dst[2 * k + 1] = v2.y; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };
  auto load_y = [&](double (&dst)[KT], int64_t row) {
#pragma unroll
    for (int k = 0; k < KT; ++k) dst[k] = (k < kn) ? Yb[row * nrhs + k] : 0.0;
  };
  if constexpr (FAST) spread_y(yk, fetch_y(n));
  else load_y(yk, n);

  for (int64_t i = 0; i < len; ++i) {
    const int64_t nn = row_of(i + 1 < len ? i + 1 : i);
    const double an1 = act ? Ab[nn * J] : 0.0, bn1 = act ? Bb[nn * J] : 0.0, tn1 = tb[nn];
    double yk1[KT];
    if constexpr (FAST) ypre = fetch_y(nn);
    else load_y(yk1, nn);

    const double dt = LOWER ? tm - tn : tn - tm;  // 0 for the first row of the series (no step)
    const double p = exp(cj * dt);
    if constexpr (FINAL && FAST) {
      // all KT = 8 columns live, nrhs % 8 == 0: reduce-scatter the eight row results, one store per row
      double part[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (Fb && act) Fb[n * J * nrhs + J * k] = Gk[k];
        const double f = p * Gk[k];
        part[k] = bn * f;
        Gk[k] = fma(an, yk[k], f);
      }
      int kout;
      const double red = rscatter8<G>(part, j, kout);
      if (valid && (G == 8 || (j & 1) == 0)) {
        double *zp = Zb + n * nrhs + kout;
        *zp = (zero_z ? 0.0 : *zp) + red;
      }
    } else {
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (FINAL) {
          if (Fb && st && k < kn) Fb[n * J * nrhs + J * k] = Gk[k];  // workspace row n = G_n (internal.hpp:142)
          const double f = p * Gk[k];
          const double red = gsum<G>(bn * f);
          if (st0 && k < kn) {
            const double zold = zero_z ? 0.0 : Zb[n * nrhs + k];
            Zb[n * nrhs + k] = zold + red;
          }
          Gk[k] = fma(an, yk[k], f);
        } else {
          Gk[k] = fma(an, yk[k], p * Gk[k]);
        }
      }
    }
    tm = tn; tn = tn1; an = an1; bn = bn1; n = nn;
    if constexpr (FAST) spread_y(yk, ypre);
    else {
#pragma unroll
      for (int k = 0; k < KT; ++k) yk[k] = yk1[k];
    }
  }
  if (!FINAL && st)
    for (int k = 0; k < kn; ++k) cb[k] = Gk[k];  // E_c
}

// carry[b][ch] holds E_ch on entry and the true incoming state of chunk ch on exit.
template <bool LOWER>
__global__ void k_mm_carry(int64_t B, int64_t N, int J, int64_t nrhs, int64_t Lc, int64_t nchunk, const double *t,
                           int64_t t_bs, const double *c, int64_t c_bs, double *carry) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * J * nrhs) return;
  const int64_t b = g / (J * nrhs);
  const int j = (int)((g / nrhs) % J);
  const int64_t k = g % nrhs;
  const double *tb = t + b * t_bs;
  const double cj = c[b * c_bs + j];
  double *cb = carry + (b * nchunk * J + j) * nrhs + k;
  double gin = 0.0;
  for (int64_t q = 0; q < nchunk; ++q) {
    const int64_t ch = LOWER ? q : nchunk - 1 - q;  // processing order of the chunks
    const int64_t s = ch * Lc, e = (s + Lc < N ? s + Lc : N) - 1;
    double *slot = cb + ch * J * nrhs;
    const double Ec = *slot;
    *slot = gin;
    // D_c: product of the decays applied inside the chunk
    double D;
    if (LOWER) D = (s >= 1) ? exp(cj * (tb[s - 1] - tb[e])) : 0.0;     // chunk 0: G_0 = 0, D irrelevant
    else D = (e <= N - 2) ? exp(cj * (tb[s] - tb[e + 1])) : 0.0;
    gin = fma(D, gin, Ec);
  }
}

}  // namespace c2

using namespace c2;

namespace {
template <bool LOWER>
int run_chunked(int64_t B, int64_t N, int64_t J, int64_t nrhs, int64_t Lc, const double *t, int64_t t_bs,
                const double *c, int64_t c_bs, const double *U, const double *V, const double *Y, double *Z, double *F,
                int zero_z, hipStream_t s) {
  const int64_t nchunk = (N + Lc - 1) / Lc;
  const size_t bytes = sizeof(double) * (size_t)B * nchunk * J * nrhs;
  double *carry = nullptr;
  bool async = true;
  if (c2::temp_alloc((void **)&carry, bytes, s) != hipSuccess) {
    (void)hipGetLastError();
    async = false;
    if (hipMalloc((void **)&carry, bytes) != hipSuccess) return C2_ERR_HIP;
  }
  const int G_ = group_size(J);
  // lanes of a unit: G x KG, KG = 64/G column groups when there are enough right-hand sides to feed them
  auto launch = [&](auto gtag, auto kgtag, auto kttag, auto finaltag) {
    constexpr int G = decltype(gtag)::value, KG = decltype(kgtag)::value, KT = decltype(kttag)::value;
    constexpr bool FINAL_ = decltype(finaltag)::value;
    const int64_t slab = (int64_t)KG * KT;
    constexpr int UPW_ = kWave / (G * KG);
    const dim3 grid((unsigned)((B * nchunk + UPW_ - 1) / UPW_), (unsigned)((nrhs + slab - 1) / slab));
    // FAST: every column group full (nrhs a multiple of the slab), no padding lanes, 16-byte aligned rows
    const bool fast = (KT == 8) && (G == 8 || G == 16) && (J == G) && (nrhs % slab == 0) &&
                      (((uintptr_t)Y | (uintptr_t)Z) % 16 == 0);
    if constexpr (KT == 8 && (G == 8 || G == 16)) {
      if (fast) {
        hipLaunchKernelGGL((k_mm_chunk<G, KG, KT, LOWER, FINAL_, true>), grid, dim3(kWave), 0, s, B, N, (int)J, nrhs,
                           Lc, nchunk, t, t_bs, c, c_bs, U, V, Y, Z, F, carry, zero_z);
        return;
      }
    }
    hipLaunchKernelGGL((k_mm_chunk<G, KG, KT, LOWER, FINAL_, false>), grid, dim3(kWave), 0, s, B, N, (int)J, nrhs, Lc,
                       nchunk, t, t_bs, c, c_bs, U, V, Y, Z, F, carry, zero_z);
  };
  auto pass = [&](auto finaltag) {
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>; using I32 = std::integral_constant<int, 32>;
    const bool wide = nrhs >= 16;  // enough columns to spread over 64/G column groups of 8 (or 4)
    switch (G_) {
      case 1: launch(I1{}, I1{}, I4{}, finaltag); break;
      case 2: launch(I2{}, I1{}, I4{}, finaltag); break;
      case 4: if (wide) launch(I4{}, I8{}, I4{}, finaltag); else launch(I4{}, I1{}, I4{}, finaltag); break;
      case 8: if (wide) launch(I8{}, I4{}, I8{}, finaltag); else launch(I8{}, I1{}, I4{}, finaltag); break;
      case 16: if (wide) launch(I16{}, I4{}, I8{}, finaltag); else launch(I16{}, I1{}, I4{}, finaltag); break;
      default: if (wide) launch(I32{}, I2{}, I8{}, finaltag); else launch(I32{}, I1{}, I4{}, finaltag); break;
    }
  };
  pass(std::false_type{});
  const int64_t nt = B * J * nrhs;
  hipLaunchKernelGGL((k_mm_carry<LOWER>), dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, s, B, N, (int)J, nrhs, Lc,
                     nchunk, t, t_bs, c, c_bs, carry);
  pass(std::true_type{});
  int rc = (hipGetLastError() == hipSuccess) ? C2_OK : C2_ERR_HIP;
  if (async) {
    if (hipFreeAsync(carry, s) != hipSuccess) rc = C2_ERR_HIP;
  } else {
    (void)hipStreamSynchronize(s);
    (void)hipFree(carry);
  }
  return rc;
}
}  // namespace

// Internal entry used by c2_matmul_lower / c2_matmul_upper (c2_ops.hip) for long series.
extern "C" int c2_internal_matmul_chunked(int lower, int64_t B, int64_t N, int64_t J, int64_t nrhs, int64_t Lc,
                                          const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                          const double *U, const double *V, const double *Y, double *Z, double *F,
                                          int zero_z, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  return lower ? run_chunked<true>(B, N, J, nrhs, Lc, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, s)
               : run_chunked<false>(B, N, J, nrhs, Lc, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, s);
}
