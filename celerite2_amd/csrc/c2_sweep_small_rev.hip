// c2_sweep_small_rev.hip -- reverse passes of solve_lower / solve_upper / matmul_lower / matmul_upper (reference
// internal.hpp:191-303 forward_rev / backward_rev; reverse.hpp:87-217) for TWO TO SEVEN right-hand sides, lanes over J: the
// mapping and stream handling of the single-rhs kernel k_sweep1_rev (c2_sweep.hip), the per-rhs quantities as short
// arrays.  The forward sweeps: c2_sweep_small.hip.
#include <cstdint>
#include <type_traits>

#include "c2_dispatch.hpp"
#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2 {

// -----------------------------------------------------------------------------------------------------------------
// KT = nrhs right-hand sides: k_sweep1_rev with the per-rhs quantities as short arrays.  The
// first-round kernel for these shapes (k_sweep_rev, c2_ops.hip) fetched every per-series scalar of a step -- two times,
// x_m[k], bZ_m[k] -- with eight lanes sharing 8 bytes and stored bt_n, bY_m[k] the same way: 13 loads and 6 stores a step
// at three right-hand sides, 8.7 ms per 8192 series of 4096 rows (0.26 of the roofline).  Here they move TRANSPOSED in
// time like in k_sweep1_rev -- lane j of a group takes position u0 + j, one request per R steps and stream, staged through
// LDS -- which leaves the width-J rows (B_n, A_m, the KT rows of the workspace; bB_n, bA_m) as the per-step requests.
// -----------------------------------------------------------------------------------------------------------------
template <int G, int R, int KT, bool LOWER, bool SOLVE, bool PAD>
__global__ __launch_bounds__(kWave) void k_sweepT_rev(int64_t B, int64_t N, int Jrt, const double *__restrict__ t,
                                                      int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                      const double *__restrict__ U, const double *__restrict__ V,
                                                      const double *__restrict__ Y, const double *__restrict__ Z,
                                                      const double *__restrict__ F, const double *__restrict__ bZ,
                                                      double *__restrict__ bt, double *__restrict__ bc,
                                                      double *__restrict__ bU, double *__restrict__ bV,
                                                      double *__restrict__ bY) {
  constexpr int SPW = kWave / G, NV = (R + G - 1) / G, NIN = 1 + 2 * KT, NOUT = 1 + KT;
  // Scalar tiles hold the ALIGNED positions R b .. R b + R - 1 of a block (see k_sweep1_rev, c2_sweep.hip): step u = R b + r reads
  // position u + 1 from entry r + 1 (entry 0 of the next block's tile for r = R - 1); bY of position u + 1 goes to entry r of the
  // block's buffer and leaves as the run R b .. R b + R - 1: the previous block's last entry and R - 1 of this one's.
  __shared__ __attribute__((aligned(16))) double sin_[2][NIN][SPW][R];   // t, x[k], bZ[k] of two blocks
  __shared__ __attribute__((aligned(16))) double sout[SPW][R];           // bt (position u)
  __shared__ __attribute__((aligned(16))) double soutY[2][KT][SPW][R];   // bY[k] (position u + 1) of this block and the one before
  const int J = PAD ? Jrt : G;
  const Geo<G> L(B, J);
  const int j = L.j, grp = L.lane / G;
  const bool act = PAD ? L.act : true;
  const bool st = PAD ? (L.valid && act) : true;
  const int64_t on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + L.jj;
  const double *tb = t + L.b0 * t_bs + (int64_t)L.sl * t_bs;
  const double *xb = (SOLVE ? Z : Y) + (L.b0 * N + on) * KT, *bzb = bZ + (L.b0 * N + on) * KT;
  double *btb = bt + L.b0 * N + on, *byb = bY + (L.b0 * N + on) * KT;
  const double *Ab = (LOWER ? V : U) + L.b0 * N * J + oj;  // row fed into F (index m)
  const double *Bb = (LOWER ? U : V) + L.b0 * N * J + oj;  // row applied to F (index n)
  double *bAb = (LOWER ? bV : bU) + L.b0 * N * J + oj, *bBb = (LOWER ? bU : bV) + L.b0 * N * J + oj;
  const double *Fb = F + (L.b0 * N * J + (int64_t)L.sl * N * J) * KT + L.jj;   // F[n, j + J k]
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  constexpr double sgn = SOLVE ? -1.0 : 1.0;
  auto rowof = [&](int64_t q) { return LOWER ? N - 1 - q : q; };

  const int64_t r0 = rowof(0);
  double bz[KT], bF[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    bz[k] = bzb[r0 * KT + k];
    soutY[1][k][grp][R - 1] = SOLVE ? bz[k] : 0.0;      // bY of position 0: reverse.hpp:112 (bY = bZ) / :178 (bY = 0); leaves with block 0
    bF[k] = 0.0;
  }
  if (st) bAb[r0 * J] = 0.0;       // never receives a contribution
  double tprev = tb[r0];
  double bcj = 0.0, carry = 0.0;

  double vin[NIN][NV];
  auto vload = [&](int64_t qb) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t q = qb + m * G + j;
      q = (q < N) ? q : N - 1;
      const int64_t n = rowof(q);
      vin[0][m] = tb[n];
#pragma unroll
      for (int k = 0; k < KT; ++k) { vin[1 + k][m] = xb[n * KT + k]; vin[1 + KT + k][m] = bzb[n * KT + k]; }
    }
  };
  auto vstage = [&](int s) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == R || idx < R) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) sin_[s][i][grp][idx] = vin[i][m];
      }
    }
  };
  vload(0); vstage(0);
  vload(R); vstage(1);
  vload(2 * R);

  double rb[R], rf[R][KT], ra[R];
  auto load_row = [&](int r, int64_t u) {  // B and F rows of position u, A row of position u+1
    const int64_t qn = (u < N) ? u : N - 1, qm = (u + 1 < N) ? u + 1 : N - 1;
    const int64_t n = rowof(qn), m = rowof(qm);
    rb[r] = act ? Bb[n * J] : 0.0;
    ra[r] = act ? Ab[m * J] : 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) rf[r][k] = act ? Fb[n * J * KT + J * k] : 0.0;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(r, r);
  lds_order();

  auto block = [&](int64_t u0, int s, auto checked_tag) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t u = u0 + r;
      if (!CHECKED || u + 1 < N) {
        const int64_t n = rowof(u), m = rowof(u + 1);
        const int e1 = (r + 1) % R;                     // entry of position u + 1 (r: unrolled) ...
        const int s1 = (r + 1 < R) ? s : (s ^ 1);       // ... in this block's tile or the next one's
        const double tm = sin_[s1][0][grp][e1];
        double xm[KT], bzm[KT], Fn[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) { xm[k] = sin_[s1][1 + k][grp][e1]; bzm[k] = sin_[s1][1 + KT + k][grp][e1]; Fn[k] = rf[r][k]; }
        const double bn = rb[r], am = ra[r];
        load_row(r, u + R);
        const double dt = tm - tprev;  // lower: t[m] - t[n]; upper: t[n] - t[m] with the roles of prev/next swapped
        const double dte = LOWER ? dt : -dt;
        tprev = tm;
        const double p = exp_decay(cj * dte);
        // reverse of update_z (internal.hpp:232-233 / 289-290)
        double val = 0.0, dotFbF = 0.0;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          val = fma(bz[k], p * Fn[k], val);
          bF[k] = fma(sgn * bn, bz[k], bF[k]);
          dotFbF = fma(Fn[k], bF[k], dotFbF);
        }
        if (st) bBb[n * J] = sgn * val;
        // reverse of the decay (internal.hpp:236-241 / 293-298)
        const double bp = dotFbF * p;
        bcj = fma(dte, bp, bcj);
        // update_f::reverse (internal.hpp:55-63 matmul, 76-84 solve)
        double bam = 0.0, f = cj * bp, g[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          bF[k] *= p;
          bam = fma(xm[k], bF[k], bam);
          g[k] = am * bF[k];
        }
        f = gsum<G>(f);
#pragma unroll
        for (int k = 0; k < KT; ++k) g[k] = gsum<G>(g[k]);
        sout[grp][r] = LOWER ? carry - f : f - carry;
        carry = f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const double out = SOLVE ? bzm[k] + g[k] : g[k];
          soutY[s][k][grp][r] = out;
          bz[k] = SOLVE ? out : bzm[k];
        }
        if (st) bAb[m * J] = bam;
      }
    }
    lds_order();
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == R || idx < R) {
        if (!CHECKED || u0 + idx + 1 < N) btb[rowof(u0 + idx)] = sout[grp][idx];
        if (!CHECKED || u0 + idx < N) {
#pragma unroll
          for (int k = 0; k < KT; ++k)
            byb[rowof(u0 + idx) * KT + k] = idx == 0 ? soutY[s ^ 1][k][grp][R - 1] : soutY[s][k][grp][idx - 1];
        }
      }
    }
    vstage(s);
    vload(u0 + 3 * R);
    lds_order();
  };
  int64_t u0 = 0;
  int s = 0;
  for (; u0 + 2 * R + 1 <= N; u0 += R, s ^= 1) block(u0, s, std::false_type{});
  for (; u0 + 1 < N; u0 += R, s ^= 1) block(u0, s, std::true_type{});

  if (u0 < N) {   // the last block's last entry: position N - 1 = R b opens a run of its own
#pragma unroll
    for (int k = 0; k < KT; ++k) byb[rowof(u0) * KT + k] = soutY[s ^ 1][k][grp][R - 1];
  }
  const int64_t rl = rowof(N - 1);
  btb[rl] = LOWER ? carry : -carry;
  if (st) {
    bBb[rl * J] = 0.0;  // bU.row(0) / bV.row(N-1) never touched
    bc[L.b * J + j] = bcj;
  }
}

}  // namespace c2

using namespace c2;

namespace {
// (templates on G: inside them `if constexpr` really discards the instances a width does not get)
template <int G, int KT, bool LO, bool SO>
void launch_kt(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V, const double *Y, const double *Z, const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bV, double *bY, hipStream_t s) {
  const dim3 grid((unsigned)((B * G + kWave - 1) / kWave));
  if (J == G)
    hipLaunchKernelGGL((k_sweepT_rev<G, 8, KT, LO, SO, false>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY);
  else
    hipLaunchKernelGGL((k_sweepT_rev<G, 8, KT, LO, SO, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY);
}
template <int G, bool LO, bool SO>
void launch_g(int kt, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V, const double *Y, const double *Z, const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bV, double *bY, hipStream_t s) {
#define C2_KA B, N, J, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY, s
  if (kt == 2) launch_kt<G, 2, LO, SO>(C2_KA);
  else if (kt == 3) launch_kt<G, 3, LO, SO>(C2_KA);
  else if (kt == 4) launch_kt<G, 4, LO, SO>(C2_KA);
  else if constexpr (G == 8) {   // five to seven: compiled for widths 5 .. 8 only
    if (kt == 5) launch_kt<G, 5, LO, SO>(C2_KA);
    else if (kt == 6) launch_kt<G, 6, LO, SO>(C2_KA);
    else launch_kt<G, 7, LO, SO>(C2_KA);
  }
#undef C2_KA
}
}  // namespace


// two to seven right-hand sides, lanes over J with transposed scalar streams (k_sweepT_rev); C2_ERR_UNSUPPORTED otherwise
extern "C" int c2_internal_sweepT_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                      double *bc, double *bU, double *bV, double *bY, c2_stream_t stream) {
  if (nrhs < 2 || nrhs > 7 || J > 32 || N < 2) return C2_ERR_UNSUPPORTED;
  if (nrhs > 4 && group_size(J) != 8) return C2_ERR_UNSUPPORTED;
  if (opt::has(opt::k_sweept_rev) && opt::ival(opt::k_sweept_rev) == 0) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int kt = (int)nrhs;
#define C2_ARGS kt, B, N, J, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY, s
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

