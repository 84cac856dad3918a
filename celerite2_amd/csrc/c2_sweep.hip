// c2_sweep.hip -- the single-right-hand-side lower / upper sweeps (solve_lower, solve_upper, matmul_lower, matmul_upper,
// with or without the F workspace: what GaussianProcess.apply_inverse / dot_tril / sample call, numpy.py:95-108) in the
// formulation of the fused log-likelihood kernel instead of the first-cut one in c2_ops.hip:
//   * per-series scalar streams (t, y, z in; z out) move TRANSPOSED in time -- lane j of a group takes row n0 + j, one
//     global_load / global_store per R rows, staged through LDS -- instead of eight lanes sharing every 8-byte word;
//   * the width-J rows (the one fed into F, the one applied to it) ride a register ring R = 8 rows ahead;
//   * p = exp(c dt) is the 16-instruction exp_decay, the group reduction a 3-level DPP butterfly.
// Reference: internal::forward / backward, c++/include/celerite2/internal.hpp:105-146 / 148-189 (policy structs
// update_f :45-85, update_z :87-103); wrappers forward.hpp:156-170, 193-207, 228-239, 260-271.
// A sweep runs over steps s = 1 .. N-1; step s is row n = s (lower) or n = N-1-s (upper).
#include <type_traits>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2 {

template <int G, int R, bool LOWER, bool SOLVE, bool PAD>
__global__ __launch_bounds__(kWave) void k_sweep1(int64_t B, int64_t N, int Jrt, const double *t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *U,
                                                  const double *V, const double *Y, double *Z, double *F, int zero_z) {
  constexpr int SPW = kWave / G, NV = (R + G - 1) / G;
  __shared__ __attribute__((aligned(16))) double sin_[2][3][SPW][R];  // t, y, z-in of two blocks
  __shared__ __attribute__((aligned(16))) double sout[SPW][R];
  const int J = PAD ? Jrt : G;
  const Geo<G> L(B, J);
  const int lane = L.lane, j = L.j, grp = lane / G;
  const bool act = PAD ? L.act : true;
  const bool loadz = !SOLVE && !zero_z;  // matmul accumulates into the caller's Z (forward.hpp:228-239)
  const int64_t on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + L.jj;
  const double *tb = t + L.b0 * t_bs + (int64_t)L.sl * t_bs, *yb = Y + L.b0 * N + on;
  double *zb = Z + L.b0 * N + on;
  const double *Ab = (LOWER ? V : U) + L.b0 * N * J + oj;  // row fed into F
  const double *Bb = (LOWER ? U : V) + L.b0 * N * J + oj;  // row applied to F
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  double *Fb = F ? F + L.b0 * N * J + oj : nullptr;  // workspace F[n, j] (nrhs = 1), written before the decay
  const bool stf = F && (PAD ? (L.valid && act) : true);
  auto rowof = [&](int64_t s) { return LOWER ? s : N - 1 - s; };

  // step 0: Z = Y (solve, forward.hpp:168,205) / Z = 0 (matmul called with zero_z) / untouched (matmul accumulate)
  const int64_t r0 = rowof(0);
  double xprev = yb[r0];
  if (SOLVE) zb[r0] = xprev;
  else if (zero_z) zb[r0] = 0.0;
  double aprev = act ? Ab[r0 * J] : 0.0;
  double tprev = tb[r0];
  double Fs = 0.0;
  if (stf) Fb[r0 * J] = 0.0;  // internal.hpp:127 / :170

  // transposed scalar streams: registers hold block b+2, LDS blocks b and b+1
  double vt[NV], vy[NV], vz[NV];
  auto vload = [&](int64_t sb) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t s = sb + m * G + j;
      s = (s < N) ? s : N - 1;
      const int64_t n = rowof(s);
      vt[m] = tb[n]; vy[m] = yb[n];
      vz[m] = loadz ? zb[n] : 0.0;
    }
  };
  auto vstage = [&](int q) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == R || idx < R) {
        sin_[q][0][grp][idx] = vt[m]; sin_[q][1][grp][idx] = vy[m]; sin_[q][2][grp][idx] = vz[m];
      }
    }
  };
  vload(1); vstage(0);
  vload(1 + R); vstage(1);
  vload(1 + 2 * R);

  double ra[R], rb[R];
  auto load_row = [&](int r, int64_t s) {
    s = (s < N) ? s : N - 1;
    const int64_t n = rowof(s);
    ra[r] = act ? Ab[n * J] : 0.0;
    rb[r] = act ? Bb[n * J] : 0.0;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(r, 1 + r);
  lds_order();

  auto block = [&](int64_t s0, int q, auto checked_tag) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t s = s0 + r;
      if (!CHECKED || s < N) {
        const double tn = sin_[q][0][grp][r], yn = sin_[q][1][grp][r], zin = sin_[q][2][grp][r];
        const double an = ra[r], bn = rb[r];
        load_row(r, s + R);
        const double p = exp_decay(cj * (LOWER ? tprev - tn : tn - tprev));
        tprev = tn;
        const double fpre = fma(aprev, xprev, Fs);  // internal.hpp:140 (lower) / :183 (upper)
        if (stf) Fb[rowof(s) * J] = fpre;            // saved before the decay (internal.hpp:142 / :185)
        const double f = p * fpre;                   // internal.hpp:143 / :186
        Fs = f;
        const double red = gsum<G>(bn * f);
        const double zn = SOLVE ? yn - red : zin + red;  // internal.hpp:144 / :187
        sout[grp][r] = zn;
        xprev = SOLVE ? zn : yn;
        aprev = an;
      }
    }
    lds_order();
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if ((G * NV == R || idx < R) && (!CHECKED || s0 + idx < N)) zb[rowof(s0 + idx)] = sout[grp][idx];
    }
    vstage(q);
    vload(s0 + 3 * R);
    lds_order();
  };
  int64_t s0 = 1;
  int q = 0;
  for (; s0 + 2 * R <= N; s0 += R, q ^= 1) block(s0, q, std::false_type{});
  for (; s0 < N; s0 += R, q ^= 1) block(s0, q, std::true_type{});
}

}  // namespace c2

using namespace c2;

// lower != 0: solve_lower / matmul_lower, else the upper sweeps; solve != 0: Z = Y -/+ ..., else Z (+)= ...
extern "C" int c2_internal_sweep1(int lower, int solve, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                  const double *c, int64_t c_bs, const double *U, const double *V, const double *Y,
                                  double *Z, double *F, int zero_z, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int G_ = group_size(J);
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
#define C2_SW(G, LO, SO)                                                                                           \
  do {                                                                                                             \
    if (J == G)                                                                                                    \
      hipLaunchKernelGGL((k_sweep1<G, 8, LO, SO, false>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U, \
                         V, Y, Z, F, zero_z);                                                                         \
    else                                                                                                           \
      hipLaunchKernelGGL((k_sweep1<G, 8, LO, SO, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U,  \
                         V, Y, Z, F, zero_z);                                                                         \
  } while (0)
#define C2_SW_G(LO, SO)                    \
  switch (G_) {                            \
    case 1: C2_SW(1, LO, SO); break;       \
    case 2: C2_SW(2, LO, SO); break;       \
    case 4: C2_SW(4, LO, SO); break;       \
    case 8: C2_SW(8, LO, SO); break;       \
    case 16: C2_SW(16, LO, SO); break;     \
    default: C2_SW(32, LO, SO); break;     \
  }
  if (lower) {
    if (solve) { C2_SW_G(true, true); } else { C2_SW_G(true, false); }
  } else {
    if (solve) { C2_SW_G(false, true); } else { C2_SW_G(false, false); }
  }
#undef C2_SW_G
#undef C2_SW
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
