// c2_sweep_small.hip -- solve_lower / solve_upper / matmul_lower / matmul_upper (reference internal.hpp:105-189) for TWO TO
// FIVE right-hand sides, lanes over J: the mapping and stream handling of the single-rhs kernel k_sweep1 (c2_sweep.hip), the
// per-rhs quantities as short arrays.  The reverse passes: c2_sweep_small_rev.hip.  (Files of their own: their instances
// are the longest compilations of the library.)
#include <cstdint>
#include <type_traits>

#include "c2_dispatch.hpp"
#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2 {

// -----------------------------------------------------------------------------------------------------------------
// The forward sweeps the same way: k_sweep1 with KT right-hand sides.  Per step: the two width-J rows (a register ring R
// rows ahead); t, y[k] (and z[k] when the product accumulates) arrive and z[k] leaves TRANSPOSED in time, one request per R
// steps and stream.  With the workspace (WF) lane j stores F[n, j + J k]: KT runs of 64 bytes per series and step.
// In-place Z == Y stays legal (rows are read blocks ahead of the row being written).
// -----------------------------------------------------------------------------------------------------------------
template <int G, int R, int KT, bool LOWER, bool SOLVE, bool PAD>
__global__ __launch_bounds__(kWave) void k_sweepT(int64_t B, int64_t N, int Jrt, const double *t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *U,
                                                  const double *V, const double *Y, double *Z, double *F, int zero_z) {
  constexpr int SPW = kWave / G, NV = (R + G - 1) / G, NIN = 1 + 2 * KT;
  __shared__ __attribute__((aligned(16))) double sin_[2][NIN][SPW][R];  // t, y[k], z-in[k] of two blocks
  __shared__ __attribute__((aligned(16))) double sout[KT][SPW][R];
  const int J = PAD ? Jrt : G;
  const Geo<G> L(B, J);
  const int j = L.j, grp = L.lane / G;
  const bool act = PAD ? L.act : true;
  const bool loadz = !SOLVE && !zero_z;  // matmul accumulates into the caller's Z (forward.hpp:228-239)
  const int64_t on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + L.jj;
  const double *tb = t + L.b0 * t_bs + (int64_t)L.sl * t_bs, *yb = Y + (L.b0 * N + on) * KT;
  double *zb = Z + (L.b0 * N + on) * KT;
  const double *Ab = (LOWER ? V : U) + L.b0 * N * J + oj;  // row fed into F
  const double *Bb = (LOWER ? U : V) + L.b0 * N * J + oj;  // row applied to F
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  double *Fb = F ? F + (L.b0 * N * J + (int64_t)L.sl * N * J) * KT + L.jj : nullptr;  // F[n, j + J k], written before the decay
  const bool stf = F && (PAD ? (L.valid && act) : true);
  auto rowof = [&](int64_t s) { return LOWER ? s : N - 1 - s; };

  // Step 0 is the first step of block 0 from a neutral state (no row before it: A = 0, x = 0, F = 0 at t_0): it yields
  // Z = Y (solve, forward.hpp:168,205) / Z = 0 (matmul called with zero_z) / Z unchanged (matmul accumulate) and the zero
  // workspace row (internal.hpp:127 / :170) like any other step -- so blocks cover positions [R b, R b + R) and the
  // transposed requests of t, y, z are whole aligned runs (profiles/r05_alignment.md).
  const int64_t r0 = rowof(0);
  double xprev[KT], Fs[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) { xprev[k] = 0.0; Fs[k] = 0.0; }
  double aprev = 0.0;
  double tprev = tb[r0];

  // transposed scalar streams: registers hold block b+2, LDS blocks b and b+1
  double vin[NIN][NV];
  auto vload = [&](int64_t sb) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t s = sb + m * G + j;
      s = (s < N) ? s : N - 1;
      const int64_t n = rowof(s);
      vin[0][m] = tb[n];
#pragma unroll
      for (int k = 0; k < KT; ++k) { vin[1 + k][m] = yb[n * KT + k]; vin[1 + KT + k][m] = loadz ? zb[n * KT + k] : 0.0; }
    }
  };
  auto vstage = [&](int q) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == R || idx < R) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) sin_[q][i][grp][idx] = vin[i][m];
      }
    }
  };
  vload(0); vstage(0);
  vload(R); vstage(1);
  vload(2 * R);

  double ra[R], rb[R];
  auto load_row = [&](int r, int64_t s) {
    s = (s < N) ? s : N - 1;
    const int64_t n = rowof(s);
    ra[r] = act ? Ab[n * J] : 0.0;
    rb[r] = act ? Bb[n * J] : 0.0;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(r, r);
  lds_order();

  auto block = [&](int64_t s0, int q, auto checked_tag) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t s = s0 + r;
      if (!CHECKED || s < N) {
        const double tn = sin_[q][0][grp][r];
        const double an = ra[r], bn = rb[r];
        load_row(r, s + R);
        const double p = exp_decay(cj * (LOWER ? tprev - tn : tn - tprev));
        tprev = tn;
        double red[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const double fpre = fma(aprev, xprev[k], Fs[k]);  // internal.hpp:140 (lower) / :183 (upper)
          if (stf) Fb[rowof(s) * J * KT + J * k] = fpre;      // saved before the decay (internal.hpp:142 / :185)
          const double f = p * fpre;                         // internal.hpp:143 / :186
          Fs[k] = f;
          red[k] = bn * f;
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) red[k] = gsum<G>(red[k]);
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const double yn = sin_[q][1 + k][grp][r], zin = sin_[q][1 + KT + k][grp][r];
          const double zn = SOLVE ? yn - red[k] : zin + red[k];  // internal.hpp:144 / :187
          sout[k][grp][r] = zn;
          xprev[k] = SOLVE ? zn : yn;
        }
        aprev = an;
      }
    }
    lds_order();
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if ((G * NV == R || idx < R) && (!CHECKED || s0 + idx < N)) {
#pragma unroll
        for (int k = 0; k < KT; ++k) zb[rowof(s0 + idx) * KT + k] = sout[k][grp][idx];
      }
    }
    vstage(q);
    vload(s0 + 3 * R);
    lds_order();
  };
  int64_t s0 = 0;
  int q = 0;
  for (; s0 + 2 * R <= N; s0 += R, q ^= 1) block(s0, q, std::false_type{});
  for (; s0 < N; s0 += R, q ^= 1) block(s0, q, std::true_type{});
}

}  // namespace c2

using namespace c2;

namespace {
// (templates on G: inside them `if constexpr` really discards the instances a width does not get)
template <int G, int KT, bool LO, bool SO>
void launch_kt(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z, hipStream_t s) {
  const dim3 grid((unsigned)((B * G + kWave - 1) / kWave));
  if (J == G)
    hipLaunchKernelGGL((k_sweepT<G, 8, KT, LO, SO, false>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z);
  else
    hipLaunchKernelGGL((k_sweepT<G, 8, KT, LO, SO, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z);
}
template <int G, bool LO, bool SO>
void launch_g(int kt, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z, hipStream_t s) {
#define C2_KA B, N, J, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, s
  if (kt == 2) launch_kt<G, 2, LO, SO>(C2_KA);
  else if (kt == 3) launch_kt<G, 3, LO, SO>(C2_KA);
  else if (kt == 4) launch_kt<G, 4, LO, SO>(C2_KA);
  else if constexpr (G == 8) launch_kt<G, 5, LO, SO>(C2_KA);   // five: compiled for widths 5 .. 8 only
#undef C2_KA
}
}  // namespace

// two to five right-hand sides (two or three with the workspace), lanes over J with transposed scalar streams (k_sweepT);
// C2_ERR_UNSUPPORTED otherwise
extern "C" int c2_internal_sweepT(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                  int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                  const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream) {
  // measured at B = 8192, N = 4096, J = 8 against the lanes-over-rhs kernel (in-process A/B): 2 / 3 / 4 / 5 right-hand
  // sides 1.94 / 1.87 / 1.87 / 1.97 -> 1.42 / 1.50 / 1.64 / 1.78 ms, level at 7; with the workspace 2.73 / 3.51 -> 2.42 / 3.20 ms
  // at 2 / 3 and behind from 4 (there the other kernel sends whole workspace rows through an LDS tile)
  if (nrhs < 2 || nrhs > (F ? 3 : 5) || J > 32 || N < 2) return C2_ERR_UNSUPPORTED;
  if (nrhs > 4 && group_size(J) != 8) return C2_ERR_UNSUPPORTED;
  if (opt::has(opt::k_sweept) && opt::ival(opt::k_sweept) == 0) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int kt = (int)nrhs;
#define C2_ARGS kt, B, N, J, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, s
#define C2_ST1(G)                                                                                        \
  do {                                                                                                   \
    if (lower) { if (solve) launch_g<G, true, true>(C2_ARGS); else launch_g<G, true, false>(C2_ARGS); }  \
    else { if (solve) launch_g<G, false, true>(C2_ARGS); else launch_g<G, false, false>(C2_ARGS); }      \
  } while (0)
  switch (group_size(J)) {
    case 1: C2_ST1(1); break;
    case 2: C2_ST1(2); break;
    case 4: C2_ST1(4); break;
    case 8: C2_ST1(8); break;
    case 16: C2_ST1(16); break;
    default: C2_ST1(32); break;
  }
#undef C2_ST1
#undef C2_ARGS
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

