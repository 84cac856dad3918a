// c2_loglik.hip -- the north-star hot path: fused batched log-likelihood and its
// reverse-mode gradient, hand-written for gfx950.
//
//   c2_loglik      : factor (forward.hpp:105-134) + solve_lower (internal.hpp:135-145) + the two
//                    reductions of the reference's callers (numpy.py:84-87,104-109; core.py:428) in ONE
//                    pass over (t, a, U, V, y); only ll[b] and flag[b] are written.
//   c2_loglik_grad : the same forward pass, additionally dropping a small CHECKPOINT of the recursion
//                    state every C steps, followed by one reverse sweep that -- segment by segment, last
//                    to first -- recomputes the C forward steps from the checkpoint (keeping the C
//                    J x J states in registers) and runs solve_lower_rev (internal.hpp:225-245) fused
//                    with factor_rev (reverse.hpp:52-84) over them.  The reference's S (N,J,J) and
//                    F (N,J) workspaces (1312 B per step at J=8) are never materialised in HBM.
//
// Mapping (c2_common.hpp): G = 2^ceil(log2 J) lanes per series, 64/G series per wavefront, one
// wavefront per workgroup.  Lane j owns column j of S / M and element j of every width-J vector.
// Width-J vectors that every lane of a group needs (p, U_n, W_{n-1}, bV_n) are exchanged through a
// per-wave LDS slot: one ds_write_b64 by each lane, G/2 broadcast ds_read_b128 by every lane -- the LDS
// pipe is otherwise idle and this keeps the exchange off the VALU, which is the binding unit here.
// Scalar all-reduces (d_n, z_n, ...) are DPP butterflies (gsum).  With 8192 series per GPU at J=8 the
// launch is exactly one wavefront per SIMD, so HBM latency is hidden by an explicit register prefetch
// ring (R rows ahead), not by occupancy.
#include "c2_common.hpp"
#include "../../include/celerite2_amd.h"

extern "C" int c2_loglik_grad_composite(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                        const double *c, int64_t c_bs, const double *a, const double *U,
                                        const double *V, const double *y, double *ll, double *bt, double *bc,
                                        double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                                        size_t work_bytes, c2_stream_t stream);
extern "C" size_t c2_loglik_grad_composite_workspace_bytes(int64_t B, int64_t N, int64_t J);

namespace c2 {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kLn2 = 0.69314718055994530941723212145818;

// 1/d for a well-scaled positive d: v_rcp_f64 seed + two Newton steps (full fp64 accuracy; the
// denormal/overflow scaling of a general IEEE division is not needed for pivots of an SPD matrix).
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  return r;
}

// Broadcast read of the G doubles a group wrote into an LDS slot.
template <int G>
__device__ __forceinline__ void lds_get(const double *slot, int gbase, double (&out)[G]) {
  if constexpr (G == 1) {
    out[0] = slot[gbase];
  } else {
    const double2 *p = reinterpret_cast<const double2 *>(slot + gbase);
#pragma unroll
    for (int k = 0; k < G / 2; ++k) {
      const double2 v = p[k];
      out[2 * k] = v.x;
      out[2 * k + 1] = v.y;
    }
  }
}
__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Checkpoint record of one lane: S[0..G-1] (column j), F_j, w_j, d, z  -> G+4 doubles.
template <int G>
struct Ckpt {
  static constexpr int W = G + 4;
};

// One forward step (row n) of factor + solve_lower.  On entry S/F/w/d/z describe row n-1, on exit row n.
// Returns p_j; the half... (see callers for what is saved).
template <int G>
__device__ __forceinline__ double fwd_step(double cj, double dt, double an, double yn, double u, double v,
                                           double (&S)[G], double &F, double &w, double &d, double &z, double &rd,
                                           double *slotP, double *slotU, double *slotW, int lane, int gbase) {
  const double p = exp(cj * dt);
  slotP[lane] = p;
  slotU[lane] = u;
  slotW[lane] = w;  // W row n-1
  lds_order();
  double pi[G], ui[G], wi[G];
  lds_get<G>(slotP, gbase, pi);
  lds_get<G>(slotU, gbase, ui);
  lds_get<G>(slotW, gbase, wi);
  const double dw = d * w;
  double tau = 0.0;
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const double s = (pi[i] * p) * fma(dw, wi[i], S[i]);  // S = P (S + d w^T w) P   (forward.hpp:115-123)
    S[i] = s;
    tau = fma(ui[i], s, tau);                              // tau = U_n S            (forward.hpp:126)
  }
  F = p * fma(w, z, F);                                    // F = P (F + W_{n-1}^T z_{n-1})  (internal.hpp:140-143)
  const double dn = an - gsum<G>(tau * u);                 // forward.hpp:127
  const double zn = yn - gsum<G>(u * F);                   // internal.hpp:144
  rd = rcp_nr(dn);
  w = (v - tau) * rd;                                      // forward.hpp:131
  d = dn;
  z = zn;
  return p;
}

// =============================================================================
// Forward pass.  R = prefetch ring length (rows), C = checkpoint interval (R % C == 0).
// =============================================================================
template <int G, int R, int C, bool CKPT>
__global__ __launch_bounds__(kWave, 1) void k_loglik_fwd(int64_t B, int64_t N, int J, const double *__restrict__ t,
                                                         int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                         const double *__restrict__ a,
                                                         const double *__restrict__ U,
                                                         const double *__restrict__ V,
                                                         const double *__restrict__ y, double *__restrict__ ll,
                                                         int32_t *__restrict__ flag, double *__restrict__ ckpt,
                                                         int64_t nseg) {
  static_assert(R % C == 0, "ring length must be a multiple of the checkpoint interval");
  __shared__ __attribute__((aligned(16))) double xs[3][kWave];
  const int lane = threadIdx.x;
  const int gbase = lane & ~(G - 1);
  const int64_t g = (int64_t)blockIdx.x * kWave + lane;
  int64_t b = g / G;
  const int j = lane & (G - 1);
  const bool valid = b < B;
  if (!valid) b = B - 1;
  const bool act = j < J;
  const int jj = act ? j : 0;
  const double *tb = t + b * t_bs, *ab = a + b * N, *yb = y + b * N;
  const double *Ub = U + b * N * J + jj, *Vb = V + b * N * J + jj;
  const double cj = act ? c[b * c_bs + j] : 0.0;
  double *ck = CKPT ? ckpt + ((b * nseg) * G + j) * Ckpt<G>::W : nullptr;

  double S[G];
#pragma unroll
  for (int i = 0; i < G; ++i) S[i] = 0.0;
  double d = ab[0];
  double rd = 1.0 / d;
  double w = act ? Vb[0] * rd : 0.0;
  double z = yb[0];
  double F = 0.0;
  double tprev = tb[0];
  double prod = d;       // running product of pivots, renormalised with frexp -> log det
  int eacc = 0;
  double quad = z * z * rd;

  double rt[R], ra[R], ry[R], ru[R], rv[R];
  auto load_row = [&](int r, int64_t n) {
    const int64_t nn = (n < N) ? n : N - 1;
    rt[r] = tb[nn]; ra[r] = ab[nn]; ry[r] = yb[nn];
    ru[r] = act ? Ub[nn * J] : 0.0; rv[r] = act ? Vb[nn * J] : 0.0;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(r, 1 + r);

  int32_t fl = 0;
  bool alive = true;
  for (int64_t n0 = 1; n0 < N; n0 += R) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = n0 + r;
      if (n < N && alive) {
        if (CKPT && (r % C == 0) && valid) {  // state after row n-1 = checkpoint (n-1)/C
          double *q = ck + ((n - 1) / C) * (G * Ckpt<G>::W);
#pragma unroll
          for (int i = 0; i < G; ++i) q[i] = S[i];
          q[G] = F; q[G + 1] = w; q[G + 2] = d; q[G + 3] = z;
        }
        const double tn = rt[r], an = ra[r], yn = ry[r], u = ru[r], v = rv[r];
        load_row(r, n + R);
        fwd_step<G>(cj, tprev - tn, an, yn, u, v, S, F, w, d, z, rd, xs[0], xs[1], xs[2], lane, gbase);
        tprev = tn;
        if (d <= 0.0) {  // forward.hpp:128 (NaN passes, as in the reference)
          fl = (int32_t)n;
          alive = false;
        } else {
          prod *= d;
          quad = fma(z * z, rd, quad);
          if (r % 8 == 7) {
            int e;
            prod = frexp(prod, &e);
            eacc += e;
          }
        }
      }
    }
    if (!alive) break;
  }
  if (valid && j == 0) {
    flag[b] = fl;
    const double logdet = log(prod) + (double)eacc * kLn2;
    ll[b] = fl ? -INFINITY : -0.5 * (logdet + (double)N * kLog2Pi) - 0.5 * quad;
  }
}


// =============================================================================
// Reverse sweep with segment recomputation.  For segment k (rows n_lo = 1 + kC ... n_lo + C - 1), last
// segment first:
//   1. reload checkpoint k (state of row n_lo - 1) and the segment's input rows;
//   2. recompute the C forward steps; keep per step: the post-decay state column S_n(:,j) in registers,
//      p_n, U_n, W_{n-1} in LDS (the same slots the forward gathers use), F_n, 1/d_n, z_n, dt_n;
//   3. issue the loads of segment k-1 (they land while step 4 runs);
//   4. run the fused reverse steps n = n_lo + C - 1 ... n_lo.
// Reverse step n (reference internal.hpp:226-245 for the solve, reverse.hpp:58-81 for the factor), with
// bz_n, ba_n, bV_n the already complete cotangents of row n on entry:
//   bU_n  = -bz_n F_n  -  (bV_n + 2 ba_n U_n) S_n                 (S_n post-decay = workspace row * diag(p))
//   bF   += -U_n bz_n ;  M -= U_n^T y + y^T U_n  (y = bV_n + ba_n U_n;  M = bS + bS^T, see k_factor_rev)
//   bp    = F_n.bF + diag(S_n M) ;  bc += dt bp ;  f = sum c bp ;  bt_n = f_{n+1} - f_n
//   bF    = p bF ;  M = P M P ;  q = W_{n-1} M
//   G = W_{n-1}.bF ,  Q = q.W_{n-1}
//   bz_{n-1} = -z_{n-1}/d_{n-1} + G                                 (seed + V_{n-1} bF)
//   bV_{n-1} = z_{n-1} bF / d_{n-1} + q                             (bW_{n-1}/d_{n-1} + W_{n-1}(bS+bS^T))
//   ba_{n-1} = bd_{n-1} - Q/2 - z_{n-1} G / d_{n-1}                 (bd + w bS w^T - W_{n-1}.bV_{n-1})
// with the seeds bd = (z^2/d - 1)/(2d), the derivative of the log-likelihood w.r.t. d (and -z/d w.r.t. z).
// =============================================================================
template <int G, int C>
__global__ __launch_bounds__(kWave, 1) void k_loglik_rev(int64_t B, int64_t N, int J, const double *__restrict__ t,
                                                         int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                         const double *__restrict__ a,
                                                         const double *__restrict__ U,
                                                         const double *__restrict__ V,
                                                         const double *__restrict__ y,
                                                         const double *__restrict__ ckpt, int64_t nseg,
                                                         const int32_t *__restrict__ flag, double *__restrict__ bt,
                                                         double *__restrict__ bc, double *__restrict__ ba,
                                                         double *__restrict__ bU, double *__restrict__ bV,
                                                         double *__restrict__ by) {
  __shared__ __attribute__((aligned(16))) double vP[C][kWave], vU[C][kWave], vW[C][kWave], xB[kWave];
  const int lane = threadIdx.x;
  const int gbase = lane & ~(G - 1);
  const int64_t g = (int64_t)blockIdx.x * kWave + lane;
  int64_t b = g / G;
  const int j = lane & (G - 1);
  const bool valid = b < B;
  if (!valid) b = B - 1;
  const bool act = j < J;
  const int jj = act ? j : 0;
  const bool st = valid && act, st0 = valid && j == 0;
  const double *tb = t + b * t_bs, *ab = a + b * N, *yb = y + b * N;
  const double *Ub = U + b * N * J + jj, *Vb = V + b * N * J + jj;
  const double *ck = ckpt + ((b * nseg) * G + j) * Ckpt<G>::W;
  double *btb = bt + b * N, *bab = ba + b * N, *byb = by + b * N;
  double *bUb = bU + b * N * J + jj, *bVb = bV + b * N * J + jj;
  const double cj = act ? c[b * c_bs + j] : 0.0;
  if (flag[b] != 0) return;  // failed factorisation: gradient undefined (uniform inside a group)

  double M[G];
#pragma unroll
  for (int i = 0; i < G; ++i) M[i] = 0.0;
  double bF = 0.0, carry = 0.0, bcj = 0.0;
  double bVn = 0.0, ban = 0.0, bzn = 0.0;

  // "current segment" inputs / checkpoint (loaded one segment ahead)
  double it[C + 1], ia[C], iy[C], iu[C], iv[C];
  double cS[G], cF, cw, cd, cz;
  auto load_segment = [&](int64_t k) {
    const int64_t n_lo = 1 + k * C;
    it[0] = tb[n_lo - 1];
#pragma unroll
    for (int r = 0; r < C; ++r) {
      const int64_t n = (n_lo + r < N) ? n_lo + r : N - 1;
      it[r + 1] = tb[n]; ia[r] = ab[n]; iy[r] = yb[n];
      iu[r] = act ? Ub[n * J] : 0.0; iv[r] = act ? Vb[n * J] : 0.0;
    }
    const double *q = ck + k * (G * Ckpt<G>::W);
#pragma unroll
    for (int i = 0; i < G; ++i) cS[i] = q[i];
    cF = q[G]; cw = q[G + 1]; cd = q[G + 2]; cz = q[G + 3];
  };

  if (nseg > 0) load_segment(nseg - 1);
  else {  // N == 1: no steps, only the seeds of row 0
    cd = ab[0]; cz = yb[0];
  }

  for (int64_t k = nseg - 1; k >= 0; --k) {
    const int64_t n_lo = 1 + k * C;
    const int cnt = (N - n_lo < C) ? (int)(N - n_lo) : C;

    // ---- recompute the forward steps of this segment ------------------------------------------------
    double S[G];
#pragma unroll
    for (int i = 0; i < G; ++i) S[i] = cS[i];
    double F = cF, w = cw, d = cd, z = cz, rd = rcp_nr(cd);
    double Sf[C][G], Fp[C], pv[C], dtv[C], rdv[C + 1], zv[C + 1];
    rdv[0] = rd; zv[0] = z;
#pragma unroll
    for (int r = 0; r < C; ++r) {
      if (r < cnt) {
        const double dt = it[r] - it[r + 1];
        const double p = fwd_step<G>(cj, dt, ia[r], iy[r], iu[r], iv[r], S, F, w, d, z, rd, vP[r], vU[r], vW[r],
                                     lane, gbase);
#pragma unroll
        for (int i = 0; i < G; ++i) Sf[r][i] = S[i];
        Fp[r] = F; pv[r] = p; dtv[r] = dt; rdv[r + 1] = rd; zv[r + 1] = z;
      }
    }
    if (k == nseg - 1) {  // cotangents of the last row: pure seeds
      const double rdl = rdv[cnt], zl = zv[cnt];
      ban = 0.5 * rdl * (zl * zl * rdl - 1.0);
      bzn = -zl * rdl;
      bVn = 0.0;
      if (st0) byb[N - 1] = bzn;
    }
    // ---- prefetch the next (earlier) segment while the reverse steps run --------------------------------
    if (k > 0) load_segment(k - 1);

    // ---- fused reverse steps ----------------------------------------------------------------------------
#pragma unroll
    for (int r = C - 1; r >= 0; --r) {
      if (r < cnt) {
        const int64_t n = n_lo + r;
        const double p = pv[r], dt = dtv[r], Fpn = Fp[r];
        const double rdm = rdv[r], zm = zv[r];
        const double u = vU[r][lane], wm = vW[r][lane];
        if (st0) bab[n] = ban;
        if (st) bVb[n * J] = bVn;
        xB[lane] = bVn;
        lds_order();
        double bVi[G], ui[G], pi[G], wi[G];
        lds_get<G>(xB, gbase, bVi);
        lds_get<G>(vU[r], gbase, ui);
        lds_get<G>(vP[r], gbase, pi);
        lds_get<G>(vW[r], gbase, wi);
        // solve_lower_rev part
        const double bU1 = -bzn * Fpn;
        bF = fma(-u, bzn, bF);
        const double bp_s = Fpn * bF;
        bF *= p;
        // factor_rev part
        const double yv = fma(ban, u, bVn);
        double xs = 0.0, bpf = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const double yi = fma(ban, ui[i], bVi[i]);
          const double xi = fma(ban, ui[i], yi);
          xs = fma(xi, Sf[r][i], xs);
          M[i] -= fma(ui[i], yv, yi * u);
          bpf = fma(Sf[r][i], M[i], bpf);
        }
        if (st) bUb[n * J] = bU1 - xs;
        const double bp = bp_s + bpf;
        bcj = fma(dt, bp, bcj);
        const double f = gsum<G>(cj * bp);
        if (st0) btb[n] = carry - f;
        carry = f;
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          M[i] *= pi[i] * p;
          q = fma(wi[i], M[i], q);
        }
        const double Gs = gsum<G>(wm * bF);
        const double Q = gsum<G>(q * wm);
        const double zr = zm * rdm;
        bzn = Gs - zr;
        if (st0) byb[n - 1] = bzn;
        bVn = fma(zr, bF, q);
        ban = 0.5 * rdm * (zm * zr - 1.0) - 0.5 * Q - zr * Gs;
      }
    }
  }
  if (nseg == 0) {  // N == 1
    const double rd0 = 1.0 / cd;
    ban = 0.5 * rd0 * (cz * cz * rd0 - 1.0);
    bzn = -cz * rd0;
    if (st0) byb[0] = bzn;
  }
  // row 0 (reverse.hpp:83-84)
  if (st0) { bab[0] = ban; btb[0] = carry; }
  if (st) { bVb[0] = bVn; bUb[0] = 0.0; bc[b * J + j] = bcj; }
}

}  // namespace c2

using namespace c2;

namespace {
inline int launch_ok() { return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP; }

template <bool CKPT>
int launch_fwd(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
               const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
               double *ckpt, int64_t nseg, hipStream_t s) {
  const int G_ = group_size(J);
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
#define C2_FWD(G, R, C)                                                                                             \
  hipLaunchKernelGGL((k_loglik_fwd<G, R, C, CKPT>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, \
                     V, y, ll, flag, ckpt, nseg)
  switch (G_) {
    case 1: C2_FWD(1, 16, 8); break;
    case 2: C2_FWD(2, 16, 8); break;
    case 4: C2_FWD(4, 16, 8); break;
    case 8: C2_FWD(8, 16, 8); break;
    case 16: C2_FWD(16, 8, 4); break;
    default: C2_FWD(32, 4, 2); break;
  }
#undef C2_FWD
  return launch_ok();
}
}  // namespace

extern "C" {

int c2_loglik(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
              const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
              c2_stream_t stream) {
  if (B < 1 || N < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  if (!t || !c || !a || !U || !V || !y || !ll || !flag) return C2_ERR_INVALID;
  return launch_fwd<false>(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, nullptr, 0, (hipStream_t)stream);
}

// Checkpoint interval per group size (must match launch_fwd / launch_rev below).
static inline int ckpt_interval(int G_) { return G_ <= 8 ? 8 : (G_ == 16 ? 4 : 2); }

size_t c2_loglik_grad_workspace_bytes(int64_t B, int64_t N, int64_t J) {
  if (B < 1 || N < 1 || J < 1 || J > C2_MAX_WIDTH) return 0;
  const int G_ = group_size(J), C_ = ckpt_interval(G_);
  const int64_t nseg = (N - 1 + C_ - 1) / C_;
  const size_t bytes = (size_t)B * (size_t)nseg * G_ * (G_ + 4) * sizeof(double);
  return bytes ? bytes : 8;
}

int c2_loglik_grad(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                   const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                   double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                   size_t work_bytes, c2_stream_t stream) {
  if (B < 1 || N < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  if (!t || !c || !a || !U || !V || !y || !ll || !bt || !bc || !ba || !bU || !bV || !by || !flag || !work)
    return C2_ERR_INVALID;
  if (work_bytes < c2_loglik_grad_workspace_bytes(B, N, J)) return C2_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const int G_ = group_size(J), C_ = ckpt_interval(G_);
  const int64_t nseg = (N - 1 + C_ - 1) / C_;
  double *ckpt = (double *)work;
  if (int e = launch_fwd<true>(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, ckpt, nseg, s)) return e;
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
#define C2_REV(G, C)                                                                                              \
  hipLaunchKernelGGL((k_loglik_rev<G, C>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, a, U, V, y,   \
                     (const double *)ckpt, nseg, (const int32_t *)flag, bt, bc, ba, bU, bV, by)
  switch (G_) {
    case 1: C2_REV(1, 8); break;
    case 2: C2_REV(2, 8); break;
    case 4: C2_REV(4, 8); break;
    case 8: C2_REV(8, 8); break;
    case 16: C2_REV(16, 4); break;
    default: C2_REV(32, 2); break;
  }
#undef C2_REV
  return launch_ok();
}

}  // extern "C"
