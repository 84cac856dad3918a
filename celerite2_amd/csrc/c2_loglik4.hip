// c2_loglik4.hip -- the fused forward log-likelihood kernel with TWO columns per lane (the gradient pair of this mapping is
// c2_loglik_q4.hip).
//
// Same algorithm as c2_loglik.hip, different lane mapping: a series of
// width J = 2*LG is walked by LG lanes, lane jl owning columns 2jl and 2jl+1 of the J x J state, so a wavefront
// carries 64/LG series (16 at J = 8 instead of 8).  Per series this halves everything that c2_loglik.hip
// replicates or pays per lane group -- the scalar chain (reductions, reciprocal), the cross-lane traffic (a
// quad_perm-only gather of 2 doubles from 3 partners instead of 7, two butterfly levels instead of three), the
// LDS gathers (one ds_read_b128 per partner) -- at the price of twice the per-series register/LDS footprint per
// wavefront.  It needs >= 1024 * 64/LG series to fill the chip (16384 at J = 8), so it is the large-batch
// variant; c2_loglik.hip stays the small-batch one.
//
// Conventions (XOR order over the lane index): slot q = 2k + e of a gathered vector is element 2(jl^k) + e;
// SX[m][q] = S(2(jl^k)+e, 2jl+m).
#include <type_traits>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

#ifndef C2_FWD4_R0
#define C2_FWD4_R0 8
#endif
#ifndef C2_FWD4_OCC
#define C2_FWD4_OCC 1
#endif

namespace c2 {

// XOR gather of a pair vector from an LDS slot (one double2 per lane): LG ds_read_b128.
template <int LG>
__device__ __forceinline__ void xgather2_lds(const double2 *slot, int lane, double (&out)[2 * LG]) {
#pragma unroll
  for (int k = 0; k < LG; ++k) {
    const double2 v = slot[lane ^ k];
    out[2 * k] = v.x;
    out[2 * k + 1] = v.y;
  }
}
// ... and on the VALU with DPP (the one chain-critical gather per step)
template <int LG>
__device__ __forceinline__ void xgather2_dpp(double x0, double x1, double (&out)[2 * LG]) {
  double a[LG], b[LG];
  xgather_dpp<LG>(x0, nullptr, 0, a);
  xgather_dpp<LG>(x1, nullptr, 0, b);
#pragma unroll
  for (int k = 0; k < LG; ++k) { out[2 * k] = a[k]; out[2 * k + 1] = b[k]; }
}

// The chain part of one forward step, two columns per lane.
template <int LG>
__device__ __forceinline__ void fwd_chain2(const double (&p)[2], const double (&u)[2], const double (&v)[2], double an,
                                           double yn, const double (&pX)[2 * LG], const double (&uX)[2 * LG],
                                           double (&SX)[2][2 * LG], double (&F)[2], double (&w)[2], double &d, double &z,
                                           double &rd) {
  constexpr int J = 2 * LG;
  double wX[J];
  xgather2_dpp<LG>(w[0], w[1], wX);
  const double dw0 = d * w[0], dw1 = d * w[1];
  double t0a = 0.0, t0b = 0.0, t1a = 0.0, t1b = 0.0;
#pragma unroll
  for (int q = 0; q < J; ++q) {
    const double s0 = (pX[q] * p[0]) * fma(dw0, wX[q], SX[0][q]);
    const double s1 = (pX[q] * p[1]) * fma(dw1, wX[q], SX[1][q]);
    SX[0][q] = s0;
    SX[1][q] = s1;
    if (q & 1) { t0b = fma(uX[q], s0, t0b); t1b = fma(uX[q], s1, t1b); }
    else { t0a = fma(uX[q], s0, t0a); t1a = fma(uX[q], s1, t1a); }
  }
  const double tau0 = t0a + t0b, tau1 = t1a + t1b;
  F[0] = p[0] * fma(w[0], z, F[0]);
  F[1] = p[1] * fma(w[1], z, F[1]);
  double rd_ = fma(tau0, u[0], tau1 * u[1]), rz_ = fma(u[0], F[0], u[1] * F[1]);
  gsum2<LG>(rd_, rz_);
  const double dn = an - rd_, zn = yn - rz_;
  rd = rcp_nr(dn);
  w[0] = (v[0] - tau0) * rd;
  w[1] = (v[1] - tau1) * rd;
  d = dn;
  z = zn;
}

// =============================================================================
// Forward pass, log-likelihood only.
// =============================================================================
template <int LG, int R>
__global__ __launch_bounds__(kWave, C2_FWD4_OCC) void k_loglik4_fwd(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                                          const double *__restrict__ c, int64_t c_bs,
                                                          const double *__restrict__ a, const double *__restrict__ U,
                                                          const double *__restrict__ V, const double *__restrict__ y,
                                                          double *__restrict__ ll, int32_t *__restrict__ flag) {
  constexpr int J = 2 * LG, SPW = kWave / LG, NV = (R + LG - 1) / LG;
  __shared__ __attribute__((aligned(16))) double2 xs2[2][kWave];
  __shared__ __attribute__((aligned(16))) double sin_[2][3][SPW][R];
  const Geo<LG> L(B, LG);
  const int lane = L.lane, jl = L.j, grp = lane / LG;
  const int64_t ot = (int64_t)L.sl * t_bs, on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + 2 * jl;
  const double *tb = t + L.b0 * t_bs + ot, *ab = a + L.b0 * N + on, *yb = y + L.b0 * N + on;
  const double2 *Ub = reinterpret_cast<const double2 *>(U + L.b0 * N * J + oj);  // row stride LG double2
  const double2 *Vb = reinterpret_cast<const double2 *>(V + L.b0 * N * J + oj);
  const double cj[2] = {c[L.b * c_bs + 2 * jl], c[L.b * c_bs + 2 * jl + 1]};
  double SX[2][J];
#pragma unroll
  for (int q = 0; q < J; ++q) { SX[0][q] = 0.0; SX[1][q] = 0.0; }
  // row 0 is the first step of block 0 from the neutral state "row -1" (S = 0, F = 0, W = 0, z = 0, d = 1 at t_0): blocks cover
  // rows [R b, R b + R) and the transposed requests of t, a, y are whole aligned runs (see k_loglik_fwd, c2_loglik.hip)
  double d = 1.0;
  double rd = 1.0;
  double w[2] = {0.0, 0.0};
  double z = 0.0;
  double F[2] = {0.0, 0.0};
  double prod = 1.0;
  int eacc = 0;
  double quad = 0.0;
  int32_t fl = 0;
  // transposed scalar streams (see c2_loglik.hip): registers hold block b+2, LDS blocks b and b+1
  double vt[NV], va[NV], vy[NV];
  auto vload = [&](int64_t nb) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t row = nb + m * LG + jl;
      row = (row < N) ? row : N - 1;
      vt[m] = tb[row]; va[m] = ab[row]; vy[m] = yb[row];
    }
  };
  auto vstage = [&](int q) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * LG + jl;
      if (LG * NV == R || idx < R) {
        sin_[q][0][grp][idx] = vt[m]; sin_[q][1][grp][idx] = va[m]; sin_[q][2][grp][idx] = vy[m];
      }
    }
  };
  vload(0); vstage(0);
  vload(R); vstage(1);
  vload(2 * R);

  double2 ru[R], rv[R];
  const double2 *up = Ub, *vp = Vb;  // row n0 of the current block
  auto load_row = [&](int r, int ahead, int64_t n, bool clamp) {
    int64_t o = ahead;
    if (clamp && n >= N) o -= n - (N - 1);
    ru[r] = up[o * LG]; rv[r] = vp[o * LG];
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(r, r, r, true);

  lds_order();
  double tnext = sin_[0][0][grp][0];
  double pc[2] = {1.0, 1.0};   // (the neutral state sits at t_0)
  double uc[2] = {ru[0].x, ru[0].y};
  double pXc[J], uXc[J];
  xs2[0][lane] = make_double2(pc[0], pc[1]);
  xs2[1][lane] = ru[0];
  lds_order();
  xgather2_lds<LG>(xs2[0], lane, pXc);
  xgather2_lds<LG>(xs2[1], lane, uXc);
  lds_order();

  auto block = [&](int64_t n0, int q, auto checked_tag) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = n0 + r;
      if (!CHECKED || n < N) {
        const double tn = tnext, an = sin_[q][1][grp][r], yn = sin_[q][2][grp][r];
        const double vv_[2] = {rv[r].x, rv[r].y};
        const double tn1 = (r + 1 < R) ? sin_[q][0][grp][r + 1] : sin_[q ^ 1][0][grp][0];
        const int rn = (r + 1) % R;
        const double pn1[2] = {exp_decay(cj[0] * (tn - tn1)), exp_decay(cj[1] * (tn - tn1))};
        const double2 un1 = ru[rn];
        xs2[0][lane] = make_double2(pn1[0], pn1[1]);
        xs2[1][lane] = un1;
        lds_order();
        double pXn[J], uXn[J];
        xgather2_lds<LG>(xs2[0], lane, pXn);
        xgather2_lds<LG>(xs2[1], lane, uXn);
        lds_order();
        fwd_chain2<LG>(pc, uc, vv_, an, yn, pXc, uXc, SX, F, w, d, z, rd);
        load_row(r, r + R, n + R, CHECKED);
        fl = ((fl == 0) & (d <= 0.0)) ? (int32_t)n : fl;
        prod *= d;
        quad = fma(z * z, rd, quad);
        if (r % 2 == 1 || r == R - 1) {  // renormalise every second row: safe for pivots in 1e-150 .. 1e150
          int e;
          prod = frexp(prod, &e);
          eacc += e;
        }
        tnext = tn1;
        pc[0] = pn1[0]; pc[1] = pn1[1]; uc[0] = un1.x; uc[1] = un1.y;
#pragma unroll
        for (int k = 0; k < J; ++k) { pXc[k] = pXn[k]; uXc[k] = uXn[k]; }
      }
    }
    lds_order();
    vstage(q);
    vload(n0 + 3 * R);
    lds_order();
  };
  int64_t n0 = 0;
  int q = 0;
  auto advance = [&]() { up += R * LG; vp += R * LG; q ^= 1; };
  for (; n0 + 2 * R <= N; n0 += R) { block(n0, q, std::false_type{}); advance(); }
  for (; n0 < N; n0 += R) { block(n0, q, std::true_type{}); advance(); }

  if (L.valid && jl == 0) {
    flag[L.b] = fl;
    int e;
    prod = frexp(prod, &e);
    const double logdet = log(prod) + (double)(eacc + e) * kLn2;
    ll[L.b] = fl ? -INFINITY : -0.5 * (logdet + (double)N * kLog2Pi) - 0.5 * quad;
  }
}

}  // namespace c2

using namespace c2;

namespace {
inline int launch_ok4() { return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP; }
}  // namespace

extern "C" {

int c2_internal_loglik4(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                        const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
                        c2_stream_t stream) {
  constexpr int LG = 4;
  const dim3 grid((unsigned)((B * LG + kWave - 1) / kWave));
  hipLaunchKernelGGL((k_loglik4_fwd<LG, C2_FWD4_R0>), grid, dim3(kWave), 0, (hipStream_t)stream, B, N, t, t_bs, c, c_bs, a, U, V, y, ll,
                     flag);
  return launch_ok4();
}

}  // extern "C"
