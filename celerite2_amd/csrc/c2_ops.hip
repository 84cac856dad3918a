// c2_ops.hip -- op-level gfx950 kernels behind the c2_* device entry points.
//
// One kernel per reference recursion (reference file:line in each header
// comment), all on the sub-wave mapping described in c2_common.hpp: G lanes per
// series, lane j owns column j of the J x J state / row j of the J x nrhs state.
//
// These kernels are the faithful, aliasing-tolerant implementations of the
// reference's per-op API (in-place d==a, W==V, Z==Y allowed).  Input rows are
// software-prefetched PF steps ahead through a register ring so that the serial
// recursion never waits on HBM latency; a row is always LOADED before the same
// row of an aliased output is STORED (program order, no __restrict__ on
// aliasable pairs), which is the device equivalent of the reference's `tmp`
// previous-row trick (internal.hpp:134-141).
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "c2_common.hpp"
#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2 {

constexpr int PF = 4;  // prefetch distance (steps) of the register ring
#ifndef C2_GM_TWO_PHASE_MAX_NRHS
#define C2_GM_TWO_PHASE_MAX_NRHS 2
#endif

struct Lane {
  int64_t b;   // series index (clamped to B-1 for padding lanes)
  int j;       // index inside the group
  bool valid;  // series < B
};
template <int G>
__device__ __forceinline__ Lane lane_of(int64_t B) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  Lane L;
  L.b = g / G;
  L.j = (int)(g % G);
  L.valid = L.b < B;
  if (!L.valid) L.b = B - 1;
  return L;
}

// Group all-gather in NATURAL order through a per-wave LDS slot: every lane publishes its value, then reads the G
// values of its group with G/2 broadcast ds_read_b128 (no ds_bpermute, no VALU).
template <int G>
__device__ __forceinline__ void lds_allgather(double *slot, int lane, double x, double (&out)[G]) {
  if constexpr (G == 1) {
    out[0] = x;
  } else {
    slot[lane] = x;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double2 *p = reinterpret_cast<const double2 *>(slot + (lane & ~(G - 1)));
#pragma unroll
    for (int k = 0; k < G / 2; ++k) {
      const double2 v = p[k];
      out[2 * k] = v.x;
      out[2 * k + 1] = v.y;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

// Coalesced movement of the S workspace (B,N,J,J).  Lane (s, j) owns column j of its series' J x J row, i.e.
// J contiguous doubles at offset j*J -- addressed directly, one instruction touches 64 separate 8-byte words.
// Instead the wavefront's rows are transposed through an LDS tile so that every global instruction moves
// 16 bytes per lane in 2J*8-byte contiguous runs per series (J == G only; padded widths keep the direct path).
template <int G>
__device__ __forceinline__ void s_row_store(double *tile, int lane, const double (&col)[G], double *row) {
  const int s = lane / G, j = lane % G;
  double2 *tw = reinterpret_cast<double2 *>(tile + s * G * G + j * G);
#pragma unroll
  for (int k = 0; k < G / 2; ++k) tw[k] = make_double2(col[2 * k], col[2 * k + 1]);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
  for (int q = 0; q < G / 2; ++q) {
    const int e = q * 2 * G + 2 * j;
    *reinterpret_cast<double2 *>(row + e) = *reinterpret_cast<const double2 *>(tile + s * G * G + e);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
template <int G>
__device__ __forceinline__ void s_row_fetch(double2 (&raw)[G / 2 > 0 ? G / 2 : 1], int lane, const double *row) {
  const int j = lane % G;
#pragma unroll
  for (int q = 0; q < G / 2; ++q) raw[q] = *reinterpret_cast<const double2 *>(row + q * 2 * G + 2 * j);
}
template <int G>
__device__ __forceinline__ void s_row_unpack(double *tile, int lane, const double2 (&raw)[G / 2 > 0 ? G / 2 : 1],
                                             double (&col)[G]) {
  const int s = lane / G, j = lane % G;
#pragma unroll
  for (int q = 0; q < G / 2; ++q) *reinterpret_cast<double2 *>(tile + s * G * G + q * 2 * G + 2 * j) = raw[q];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const double2 *tr = reinterpret_cast<const double2 *>(tile + s * G * G + j * G);
#pragma unroll
  for (int k = 0; k < G / 2; ++k) { const double2 v = tr[k]; col[2 * k] = v.x; col[2 * k + 1] = v.y; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// =============================================================================
// factor -- reference forward.hpp:69-135.
//   d_0 = a_0, W_0 = V_0/d_0;  per n: S += d w^T w; S = diag(p) S [-> workspace];
//   S = S diag(p); tau = U_n S; d_n = a_n - tau.U_n; W_n = (V_n - tau)/d_n.
// Lane j holds column j of S (Sc[i] = S(i,j)).
// =============================================================================
template <int G>
__global__ __launch_bounds__(kWave) void k_factor(int64_t B, int64_t N, int J, const double *t, int64_t t_bs,
                                                  const double *c, int64_t c_bs, const double *a, const double *U,
                                                  const double *V, double *d, double *W, double *S, int32_t *flag) {
  __shared__ __attribute__((aligned(16))) double gs[3][kWave];
  __shared__ __attribute__((aligned(16))) double stile[G >= 2 ? kWave * G : 2];
  const int lane = threadIdx.x;
  const Lane L = lane_of<G>(B);
  const int j = L.j;
  const bool act = j < J;
  const bool st = L.valid && act, st0 = L.valid && j == 0;
  const int jj = act ? j : 0;
  const bool tiled = (G >= 2) && (J == G) && S && (((uintptr_t)S) % 16 == 0);  // coalesced S rows
  double *Srow = S ? S + L.b * N * J * J : nullptr;
  const double *tb = t + L.b * t_bs, *ab = a + L.b * N;
  const double *Ub = U + L.b * N * J + jj, *Vb = V + L.b * N * J + jj;
  double *db = d + L.b * N, *Wb = W + L.b * N * J + jj;
  double *Sb = S ? S + L.b * N * J * J + (int64_t)jj * J : nullptr;
  const double cj = act ? c[L.b * c_bs + j] : 0.0;

  double Sc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) Sc[i] = 0.0;

  double dprev = ab[0];
  double w = act ? Vb[0] / dprev : 0.0;
  double tprev = tb[0];

  // prefetch ring for rows 1..PF
  double rt[PF], ra[PF], ru[PF], rv[PF];
#pragma unroll
  for (int r = 0; r < PF; ++r) {
    const int64_t nn = (1 + r < N) ? 1 + r : N - 1;
    rt[r] = tb[nn]; ra[r] = ab[nn];
    ru[r] = act ? Ub[nn * J] : 0.0; rv[r] = act ? Vb[nn * J] : 0.0;
  }
  if (st0) db[0] = dprev;
  if (st) Wb[0] = w;
  if (Sb && st)
    for (int i = 0; i < J; ++i) Sb[i] = 0.0;  // S.row(0).setZero() (forward.hpp:92)

  int32_t fl = 0;
  bool alive = true;
  for (int64_t n0 = 1; n0 < N; n0 += PF) {
#pragma unroll
    for (int r = 0; r < PF; ++r) {
      const int64_t n = n0 + r;
      if (n < N && alive) {
        const double tn = rt[r], an = ra[r], u = ru[r], v = rv[r];
        {  // refill this ring slot with row n+PF (before any store to row n)
          const int64_t nn = (n + PF < N) ? n + PF : N - 1;
          rt[r] = tb[nn]; ra[r] = ab[nn];
          ru[r] = act ? Ub[nn * J] : 0.0; rv[r] = act ? Vb[nn * J] : 0.0;
        }
        const double p = exp(cj * (tprev - tn));
        tprev = tn;
        const double dw = dprev * w;
        double tau = 0.0;
        double wA[G], pA[G], uA[G], sh[G];
        lds_allgather<G>(gs[0], lane, w, wA);
        lds_allgather<G>(gs[1], lane, p, pA);
        lds_allgather<G>(gs[2], lane, u, uA);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const double wi = wA[i], pi = pA[i], ui = uA[i];
          double s = pi * fma(dw, wi, Sc[i]);
          sh[i] = s;  // S[n, i + J*j], half-scaled (forward.hpp:120)
          if (Sb && !tiled && st && i < J) Sb[n * J * J + i] = s;
          s *= p;
          Sc[i] = s;
          tau = fma(ui, s, tau);
        }
        if constexpr (G >= 2) {
          if (tiled) s_row_store<G>(stile, lane, sh, Srow + n * J * J);
        }
        const double dn = an - gsum<G>(tau * u);
        if (st0) db[n] = dn;
        if (dn <= 0.0) {  // forward.hpp:128 (NaN passes, as in the reference)
          fl = (int32_t)n;
          alive = false;
        } else {
          w = (v - tau) / dn;
          if (st) Wb[n * J] = w;
          dprev = dn;
        }
      }
    }
    if (!alive) break;
  }
  if (st0) flag[L.b] = fl;
}

// -----------------------------------------------------------------------------
// The S workspace of core::factor (forward.hpp:92-120) as a second, chain-free pass once d and W exist (they come
// from the tuned fused kernel): S_ws[n] = diag(p_n)(T + d_{n-1} w_{n-1}^T w_{n-1}), T <- S_ws[n] diag(p_n) has no
// reduction, no division and no dependence on d_n, so it runs at the speed of its stores.  Lane j owns column j in
// natural order; rows leave through the LDS tile of s_row_store as dense 16-byte-per-lane runs.  J == G only.
// Rows after a failed pivot stay untouched, like the reference's early return (forward.hpp:128).
// -----------------------------------------------------------------------------
// LN (G = 8, W 16-byte aligned): the grid and the pivots arrive as transposed tiles of eight rows (lane j <-> row n0 + j: one
// request per eight rows instead of one per row in which eight lanes share 8 bytes), the W rows as halves of aligned
// 128-byte lines through a four-row LDS tile -- 1.4 requests per row next to the four stores instead of three (the pass is
// bound by the memory instructions a single wavefront keeps in flight, profiles/r03_sweep_rev_lines.md).
template <int G, bool LN = false>
__global__ __launch_bounds__(kWave) void k_s_replay(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                                    const double *__restrict__ c, int64_t c_bs,
                                                    const double *__restrict__ d, const double *__restrict__ W,
                                                    const int32_t *__restrict__ flag, double *__restrict__ S) {
  constexpr int J = G, R = 8, SPW = kWave / G;
  static_assert(!LN || G == 8, "lines: eight lanes per series");
  __shared__ __attribute__((aligned(16))) double gs[2][kWave];
  __shared__ __attribute__((aligned(16))) double stile[kWave * G];
  __shared__ __attribute__((aligned(16))) double tWl[LN ? 4 : 1][kWave];          // W rows (row & 3)
  __shared__ __attribute__((aligned(16))) double scT[LN ? 2 : 1][2][SPW][R];     // [block parity][t_n / d_{n-1}][series][row]
  const int lane = threadIdx.x;
  const Lane L = lane_of<G>(B);
  const int j = L.j, grp = lane / G;
  const double *tb = t + L.b * t_bs, *db = d + L.b * N, *Wb = W + L.b * N * J + j;
  double *Srow = S + L.b * N * J * J;
  const double cj = c[L.b * c_bs + j];
  const int32_t fl = flag[L.b];
  const int64_t nlast = fl ? (int64_t)fl : N - 1;  // the reference saves row n before it tests d_n
  // the series of a wavefront run to the longest of their ranges (they differ only after a failed pivot)
  int64_t nmax = nlast;
  for (int o = G; o < kWave; o *= 2) {
    const int64_t other = __shfl_xor((long long)nmax, o);
    nmax = other > nmax ? other : nmax;
  }
  double Sc[G];
#pragma unroll
  for (int i = 0; i < G; ++i) Sc[i] = 0.0;
  if (L.valid) {
#pragma unroll
    for (int i = 0; i < G; ++i) Srow[j * J + i] = 0.0;  // S.row(0).setZero() (forward.hpp:92)
  }
  double rt[LN ? 1 : R], rdm[LN ? 1 : R], rw[LN ? 1 : R];
  auto load_row = [&](int r, int64_t n) {
    n = (n < N) ? n : N - 1;
    n = (n >= 1) ? n : 1;
    rt[r] = tb[n]; rdm[r] = db[n - 1]; rw[r] = Wb[(n - 1) * J];
  };
  // LN: scalar tiles (registers hold block b + 2, LDS blocks b and b + 1) and a ring of four W lines (slot = line mod 4)
  double vt = 0.0, vd = 0.0;
  double2 lw[LN ? R / 2 : 1];
  const int hrow = j >> 2, hcol = 2 * (j & 3);
  const double *Wl = W + L.b * N * J + hcol;
  auto vload = [&](int64_t nb) {
    int64_t n = nb + j;
    n = (n < N) ? n : N - 1;
    n = (n >= 1) ? n : 1;
    vt = tb[n]; vd = db[n - 1];
  };
  auto vstage = [&](int q) { scT[q][0][grp][j] = vt; scT[q][1][grp][j] = vd; };
  auto wline = [&](int64_t l) -> double2 {
    int64_t row = 2 * l + hrow;
    row = (row < N) ? row : N - 1;
    return *reinterpret_cast<const double2 *>(Wl + row * J);
  };
  if (N > 1) {
    if constexpr (LN) {
      vload(1); vstage(0);
      vload(1 + R); vstage(1);
      vload(1 + 2 * R);
#pragma unroll
      for (int q = 0; q < R / 2; ++q) lw[q] = wline(q);
      lds_order();
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) load_row(r, 1 + r);
    }
  }
  double tprev = tb[0];
  int bq = 0;   // parity of the current block of eight rows (LN)
  auto step = [&](int r, int64_t n, auto guarded_tag) {
    constexpr bool GUARDED = decltype(guarded_tag)::value;
    double tn, dw, w;
    double wA[G], pA[G], sh[G];
    if constexpr (LN) {
      // n = n0 + r with n0 = 1 (mod 8): row n - 1 of W opens the line (n - 1) / 2 at even r (slot r / 2 of the ring)
      if (r % 2 == 0) {   // (r: unrolled)
        const int64_t l = (n - 1) >> 1;
        *reinterpret_cast<double2 *>(&tWl[(r + hrow) & 3][grp * G + hcol]) = lw[(r / 2) % (R / 2)];
        lw[(r / 2) % (R / 2)] = wline(l + R / 2);
        lds_order();
      }
      tn = scT[bq][0][grp][r];
      const double dm = scT[bq][1][grp][r];
#pragma unroll
      for (int i = 0; i < G; i += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(&tWl[r & 3][grp * G + i]);
        wA[i] = v.x; wA[i + 1] = v.y;
      }
      w = tWl[r & 3][lane];
      dw = dm * w;
    } else {
      tn = rt[r]; dw = rdm[r] * rw[r]; w = rw[r];
      load_row(r, n + R);
    }
    const double p = exp_decay(cj * (tprev - tn));
    tprev = tn;
    if constexpr (!LN) lds_allgather<G>(gs[0], lane, w, wA);
    lds_allgather<G>(gs[1], lane, p, pA);
#pragma unroll
    for (int i = 0; i < G; ++i) {
      sh[i] = pA[i] * fma(dw, wA[i], Sc[i]);  // diag(p) (S + d w^T w)   (forward.hpp:115-116)
      Sc[i] = sh[i] * p;                      // S diag(p)               (forward.hpp:123)
    }
    if (!GUARDED || (L.valid && n <= nlast)) s_row_store<G>(stile, lane, sh, Srow + n * J * J);  // forward.hpp:120
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if constexpr (LN) {
      if (r == R - 1) {   // end of the block: stage block b + 2's scalars, fetch block b + 3's
        lds_order();
        vstage(bq);
        vload(n + 1 + 2 * R);
        bq ^= 1;
        lds_order();
      }
    }
  };
  // Full blocks of a full wavefront run without a single branch: with row guards in the body the compiler can no
  // longer count the stores behind a prefetched load and waits for ALL of them (vmcnt(0)) once per row.
  int64_t nmin = nlast;
  for (int o = G; o < kWave; o *= 2) {
    const int64_t other = __shfl_xor((long long)nmin, o);
    nmin = other < nmin ? other : nmin;
  }
  const bool wave_full = __all(L.valid);
  int64_t n0 = 1;
  if (wave_full) {
    for (; n0 + R - 1 <= nmin; n0 += R) {
#pragma unroll
      for (int r = 0; r < R; ++r) step(r, n0 + r, std::false_type{});
    }
  }
  for (; n0 <= nmax; n0 += R) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (n0 + r <= nmax) step(r, n0 + r, std::true_type{});
    }
  }
}

// =============================================================================
// Shared sweeps -- reference internal.hpp:105-146 (forward) / 148-189 (backward).
//   LOWER: n = 1..N-1,  F += V_{n-1}^T x_{n-1}; [F -> workspace row n]; F = diag(p) F; Z_n -/+= U_n F
//   UPPER: n = N-2..0,  F += U_{n+1}^T x_{n+1}; [F -> workspace row n]; F = diag(p) F; Z_n -/+= V_n F
//   x = Z (solve) or Y (matmul).  Lane j holds F(j, k0..k0+KT-1); blockIdx.y = rhs tile.
// =============================================================================
template <int G, int KT, bool LOWER, bool SOLVE>
__global__ __launch_bounds__(kWave) void k_sweep(int64_t B, int64_t N, int J, int64_t nrhs, const double *t,
                                                 int64_t t_bs, const double *c, int64_t c_bs, const double *U,
                                                 const double *V, const double *Y, double *Z, double *F,
                                                 int zero_z) {
  const Lane L = lane_of<G>(B);
  const int j = L.j;
  const bool act = j < J;
  const bool st = L.valid && act, st0 = L.valid && j == 0;
  const int jj = act ? j : 0;
  const int64_t k0 = (int64_t)blockIdx.y * KT;
  const int kn = (nrhs - k0 < KT) ? (int)(nrhs - k0) : KT;
  const double *tb = t + L.b * t_bs;
  const double *Ab = (LOWER ? V : U) + L.b * N * J + jj;  // row fed into F
  const double *Bb = (LOWER ? U : V) + L.b * N * J + jj;  // row applied to F
  const double *Yb = Y + L.b * N * nrhs + k0;
  double *Zb = Z + L.b * N * nrhs + k0;
  double *Fb = F ? F + L.b * N * J * nrhs + jj + J * k0 : nullptr;
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  const bool loadz = !SOLVE && !zero_z;

  const int64_t r0 = LOWER ? 0 : N - 1;
  double Fk[KT], xprev[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    Fk[k] = 0.0;
    const double y0 = (k < kn) ? Yb[r0 * nrhs + k] : 0.0;
    xprev[k] = y0;
  }
  double am = act ? Ab[r0 * J] : 0.0;
  double tprev = tb[r0];

  double rt[PF], rb[PF], ra[PF], ry[PF][KT], rz[PF][KT];
  auto load_row = [&](int r, int64_t s) {
    const int64_t sc = (s < N) ? s : N - 1;
    const int64_t n = LOWER ? sc : N - 1 - sc;
    rt[r] = tb[n];
    rb[r] = act ? Bb[n * J] : 0.0;
    ra[r] = act ? Ab[n * J] : 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      ry[r][k] = (k < kn) ? Yb[n * nrhs + k] : 0.0;
      rz[r][k] = (loadz && k < kn) ? Zb[n * nrhs + k] : 0.0;
    }
  };
#pragma unroll
  for (int r = 0; r < PF; ++r) load_row(r, 1 + r);

  // first row: solve -> Z = Y (forward.hpp:168,205); matmul(zero_z) -> Z = 0
  if (st0) {
    for (int k = 0; k < kn; ++k) {
      if (SOLVE) Zb[r0 * nrhs + k] = xprev[k];
      else if (zero_z) Zb[r0 * nrhs + k] = 0.0;
    }
  }
  if (Fb && st)
    for (int k = 0; k < kn; ++k) Fb[r0 * J * nrhs + J * k] = 0.0;  // internal.hpp:127 / 170

  auto step = [&](int r, int64_t s, auto plain_tag) {
    // PLAIN: a full rhs tile on a wavefront without padding lanes -- no store predicates at all (the G lanes of a
    // group hold the same z and store it to the same address, which is harmless), hence no branches in the body.
    constexpr bool PLAIN = decltype(plain_tag)::value;
    const int64_t n = LOWER ? s : N - 1 - s;
    const double tn = rt[r], bn = rb[r], an = ra[r];
    double yk[KT], zk[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) { yk[k] = ry[r][k]; zk[k] = rz[r][k]; }
    load_row(r, s + PF);
    const double p = exp_decay(cj * (LOWER ? tprev - tn : tn - tprev));
    tprev = tn;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      double f = fma(am, xprev[k], Fk[k]);
      if (PLAIN) { if (Fb) Fb[n * J * nrhs + J * k] = f; }
      else if (Fb && st && k < kn) Fb[n * J * nrhs + J * k] = f;  // F[n, j + J*k] before the decay (internal.hpp:142)
      f *= p;
      Fk[k] = f;
      const double red = gsum<G>(bn * f);
      const double znew = SOLVE ? yk[k] - red : zk[k] + red;
      if (PLAIN || (st0 && k < kn)) Zb[n * nrhs + k] = znew;
      xprev[k] = SOLVE ? znew : yk[k];
    }
    am = an;
  };
  // Full blocks of such wavefronts run branch-free: behind a row guard the compiler cannot count the stores issued
  // after a prefetched load and drains them all (s_waitcnt vmcnt(0)) once per row.
  const bool plain = __all(st && kn == KT);
  int64_t s0 = 1;
  if (plain) {
    for (; s0 + PF <= N; s0 += PF) {
#pragma unroll
      for (int r = 0; r < PF; ++r) step(r, s0 + r, std::true_type{});
    }
  }
  for (; s0 < N; s0 += PF) {
#pragma unroll
    for (int r = 0; r < PF; ++r) {
      if (s0 + r < N) step(r, s0 + r, std::false_type{});
    }
  }
}

// =============================================================================
// Reverse of the shared sweeps -- reference internal.hpp:191-246 (forward_rev),
// 248-303 (backward_rev), wrapped as in reverse.hpp:87-217 (outputs zeroed, for
// solves bY = bZ and the running cotangent lives in bY).
// Lane j holds bF(j, k-tile).  rhs tiles are processed sequentially inside the
// kernel; bt, bc and the two low-rank cotangents accumulate across tiles.
// =============================================================================
template <int G, int KT, bool LOWER, bool SOLVE>
__global__ __launch_bounds__(kWave) void k_sweep_rev(int64_t B, int64_t N, int J, int64_t nrhs,
                                                     const double *__restrict__ t, int64_t t_bs,
                                                     const double *__restrict__ c, int64_t c_bs,
                                                     const double *__restrict__ U, const double *__restrict__ V,
                                                     const double *__restrict__ Y, const double *__restrict__ Z,
                                                     const double *__restrict__ F, const double *bZ, double *bt,
                                                     double *bc, double *bU, double *bV, double *bY) {
  const Lane L = lane_of<G>(B);
  const int j = L.j;
  const bool act = j < J;
  const bool st = L.valid && act, st0 = L.valid && j == 0;
  const int jj = act ? j : 0;
  const double *tb = t + L.b * t_bs;
  const double *Ab = (LOWER ? V : U) + L.b * N * J + jj;
  const double *Bb = (LOWER ? U : V) + L.b * N * J + jj;
  double *bAb = (LOWER ? bV : bU) + L.b * N * J + jj;
  double *bBb = (LOWER ? bU : bV) + L.b * N * J + jj;
  const double *Xb = (SOLVE ? Z : Y) + L.b * N * nrhs;
  const double *bZb = bZ + L.b * N * nrhs;
  double *bYb = bY + L.b * N * nrhs;
  double *btb = bt + L.b * N;
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  const double sgn = SOLVE ? -1.0 : 1.0;

  for (int64_t k0 = 0; k0 < nrhs; k0 += KT) {
    const int kn = (nrhs - k0 < KT) ? (int)(nrhs - k0) : KT;
    const bool acc = k0 > 0;
    const double *Fb = F + L.b * N * J * nrhs + jj + J * k0;
    double bF[KT], bz[KT];
    const int64_t nfirst = LOWER ? N - 1 : 0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      bF[k] = 0.0;
      bz[k] = (k < kn) ? bZb[nfirst * nrhs + k0 + k] : 0.0;
    }
    if (st0)
      for (int k = 0; k < kn; ++k) bYb[nfirst * nrhs + k0 + k] = SOLVE ? bz[k] : 0.0;
    if (st && !acc) bAb[nfirst * J] = 0.0;  // never receives a contribution
    double bcj = 0.0, carry = 0.0;

    // ring: everything row-indexed that step (n, m) needs
    double rtn[PF], rtm[PF], rbn[PF], ram[PF], rF[PF][KT], rX[PF][KT], rbz[PF][KT];
    auto load_row = [&](int r, int64_t s) {  // s = N-1 .. 1 (clamped)
      const int64_t sc = (s >= 1) ? s : 1;
      const int64_t n = LOWER ? sc : N - 1 - sc;
      const int64_t m = LOWER ? n - 1 : n + 1;
      rtn[r] = tb[n]; rtm[r] = tb[m];
      rbn[r] = act ? Bb[n * J] : 0.0;
      ram[r] = act ? Ab[m * J] : 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const bool ok = k < kn;
        rF[r][k] = (ok && act) ? Fb[n * J * nrhs + J * k] : 0.0;
        rX[r][k] = ok ? Xb[m * nrhs + k0 + k] : 0.0;
        rbz[r][k] = ok ? bZb[m * nrhs + k0 + k] : 0.0;
      }
    };
    if (N > 1) {
#pragma unroll
      for (int r = 0; r < PF; ++r) load_row(r, N - 1 - r);
    }

    auto step = [&](int r, int64_t s, auto plain_tag) {
      // PLAIN: full rhs tile, no padding lanes -- no store predicates (group scalars are stored by all G lanes of the
      // group to the same address with the same value), hence no branches and exact vmcnt waits
      constexpr bool PLAIN = decltype(plain_tag)::value;
      const int64_t n = LOWER ? s : N - 1 - s;
      const int64_t m = LOWER ? n - 1 : n + 1;
      const double dt = LOWER ? rtm[r] - rtn[r] : rtn[r] - rtm[r];
      const double bn = rbn[r], am = ram[r];
      double Fn[KT], Xm[KT], bzm[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) { Fn[k] = rF[r][k]; Xm[k] = rX[r][k]; bzm[k] = rbz[r][k]; }
      load_row(r, s - PF);

      const double p = exp_decay(cj * dt);
      // reverse of update_z (internal.hpp:232-233 / 289-290)
      double val = 0.0, dotFbF = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        val = fma(bz[k], p * Fn[k], val);
        bF[k] = fma(sgn * bn, bz[k], bF[k]);
        dotFbF = fma(Fn[k], bF[k], dotFbF);
      }
      if (PLAIN || st) bBb[n * J] = acc ? bBb[n * J] + sgn * val : sgn * val;
      // reverse of the decay (internal.hpp:236-241 / 293-298)
      const double bp = dotFbF * p;
      bcj = fma(dt, bp, bcj);
      const double f = gsum<G>(cj * bp);
      if (PLAIN || st0) {
        const double v = LOWER ? carry - f : f - carry;
        btb[n] = acc ? btb[n] + v : v;
      }
      carry = f;
      // update_f::reverse (internal.hpp:55-63 matmul, 76-84 solve)
      double bam = 0.0;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        bF[k] *= p;
        bam = fma(Xm[k], bF[k], bam);
        const double g = gsum<G>(am * bF[k]);
        const double out = SOLVE ? bzm[k] + g : g;
        if (PLAIN || (st0 && k < kn)) bYb[m * nrhs + k0 + k] = out;
        bz[k] = SOLVE ? out : bzm[k];
      }
      if (PLAIN || st) bAb[m * J] = acc ? bAb[m * J] + bam : bam;
    };
    const bool plain = __all(st && kn == KT);
    int64_t s0 = N - 1;
    if (plain) {
      for (; s0 - PF + 1 >= 1; s0 -= PF) {
#pragma unroll
        for (int r = 0; r < PF; ++r) step(r, s0 - r, std::true_type{});
      }
    }
    for (; s0 >= 1; s0 -= PF) {
#pragma unroll
      for (int r = 0; r < PF; ++r) {
        if (s0 - r >= 1) step(r, s0 - r, std::false_type{});
      }
    }
    const int64_t mlast = LOWER ? 0 : N - 1;
    if (st0) {
      const double v = LOWER ? carry : -carry;
      btb[mlast] = acc ? btb[mlast] + v : v;
    }
    if (st && !acc) bBb[mlast * J] = 0.0;  // bU.row(0) / bV.row(N-1) never touched
    if (st) {
      double *bcb = bc + L.b * J + j;
      *bcb = acc ? *bcb + bcj : bcj;
    }
  }
}

// =============================================================================
// factor_rev -- reference reverse.hpp:10-85.
// Only the symmetric part of the reference's bS ever reaches an output (bp uses
// bS(k,i)+bS(i,k); ba uses w bS w^T; bV uses bS+bS^T; the P bS P update keeps
// the symmetric/antisymmetric split), so the kernel carries M = bS + bS^T with
// lane j owning column j, next to column j of the workspace S_n.  `accumulate`
// adds into bt, bc, bU (used by the fused log-likelihood gradient).
// =============================================================================
template <int G>
__global__ __launch_bounds__(kWave) void k_factor_rev(int64_t B, int64_t N, int J, const double *__restrict__ t,
                                                      int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                      const double *__restrict__ U, const double *__restrict__ d,
                                                      const double *__restrict__ W, const double *__restrict__ S,
                                                      const double *__restrict__ bd, const double *__restrict__ bW,
                                                      double *bt, double *bc, double *ba, double *bU, double *bV,
                                                      int accumulate) {
  __shared__ __attribute__((aligned(16))) double gs[5][kWave];
  __shared__ __attribute__((aligned(16))) double stile[G >= 2 ? kWave * G : 2];
  const int lane = threadIdx.x;
  const Lane L = lane_of<G>(B);
  const int j = L.j;
  const bool act = j < J;
  const bool st = L.valid && act, st0 = L.valid && j == 0;
  const int jj = act ? j : 0;
  // (reading S through the LDS tile was measured slower here -- 13.9 vs 9.7 ms: with a 2-row ring the extra LDS
  //  round trip lands on the critical path -- so the reverse kernel keeps the direct column loads)
  const bool tiled = false;
  const double *Srow = S + L.b * N * J * J;
  const double *tb = t + L.b * t_bs, *db = d + L.b * N, *bdb = bd + L.b * N;
  const double *Ub = U + L.b * N * J + jj, *Wb = W + L.b * N * J + jj, *bWb = bW + L.b * N * J + jj;
  const double *Sb = S + L.b * N * J * J + (int64_t)jj * J;
  double *btb = bt + L.b * N, *bab = ba + L.b * N;
  double *bUb = bU + L.b * N * J + jj, *bVb = bV + L.b * N * J + jj;
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  const bool acc = accumulate != 0;

  double M[G];
#pragma unroll
  for (int i = 0; i < G; ++i) M[i] = 0.0;

  double ban = bdb[N - 1];
  double bVn = act ? bWb[(N - 1) * J] / db[N - 1] : 0.0;
  double bcj = 0.0, carry = 0.0;

  constexpr int PFR = 2;
  double rtn[PFR], rtm[PFR], ru[PFR], rwn[PFR], rwm[PFR], rdm[PFR], rbdm[PFR], rbWm[PFR], rS[PFR][G];
  double2 rawS[PFR][G / 2 > 0 ? G / 2 : 1];
  auto load_row = [&](int r, int64_t n) {
    const int64_t nc = (n >= 1) ? n : 1;
    rtn[r] = tb[nc]; rtm[r] = tb[nc - 1];
    ru[r] = act ? Ub[nc * J] : 0.0;
    rwn[r] = act ? Wb[nc * J] : 0.0;
    rwm[r] = act ? Wb[(nc - 1) * J] : 0.0;
    rdm[r] = db[nc - 1]; rbdm[r] = bdb[nc - 1];
    rbWm[r] = act ? bWb[(nc - 1) * J] : 0.0;
    if (tiled) {
      if constexpr (G >= 2) s_row_fetch<G>(rawS[r], lane, Srow + nc * J * J);
    } else {
#pragma unroll
      for (int i = 0; i < G; ++i) rS[r][i] = (act && i < J) ? Sb[nc * J * J + i] : 0.0;
    }
  };
  if (N > 1) {
#pragma unroll
    for (int r = 0; r < PFR; ++r) load_row(r, N - 1 - r);
  }

  for (int64_t n0 = N - 1; n0 >= 1; n0 -= PFR) {
#pragma unroll
    for (int r = 0; r < PFR; ++r) {
      const int64_t n = n0 - r;
      if (n >= 1) {
        const double dt = rtm[r] - rtn[r];
        const double u = ru[r], wn = rwn[r], wm = rwm[r], dm = rdm[r], bdm = rbdm[r], bWm = rbWm[r];
        double Sc[G];
        if (tiled) {
          if constexpr (G >= 2) s_row_unpack<G>(stile, lane, rawS[r], Sc);
        } else {
#pragma unroll
          for (int i = 0; i < G; ++i) Sc[i] = rS[r][i];
        }
        load_row(r, n - PFR);

        const double p = exp(cj * dt);
        // Step 6 (reverse.hpp:65-67)
        ban -= gsum<G>(wn * bVn);
        if (st0) bab[n] = ban;
        if (st) bVb[n * J] = bVn;
        const double y = fma(ban, u, bVn);   // bV + ba U
        const double x = fma(ban, u, y);     // bV + 2 ba U
        double xs = 0.0, bpacc = 0.0;
        double xA[G], uA[G], yA[G], pA[G], wA[G];
        lds_allgather<G>(gs[0], lane, x, xA);
        lds_allgather<G>(gs[1], lane, u, uA);
        lds_allgather<G>(gs[2], lane, y, yA);
        lds_allgather<G>(gs[3], lane, p, pA);
        lds_allgather<G>(gs[4], lane, wm, wA);
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const double xi = xA[i], ui = uA[i], yi = yA[i];
          xs = fma(xi, Sc[i], xs);
          M[i] -= fma(ui, y, yi * u);          // M -= U^T y + y^T U
          bpacc = fma(Sc[i], M[i], bpacc);     // diag(bS Sn + Sn^T bS)_j = sum_i Sn(i,j) M(i,j)
        }
        const double bun = -p * xs;
        if (st) bUb[n * J] = acc ? bUb[n * J] + bun : bun;
        // Step 4 (reverse.hpp:70-74)
        const double bp = bpacc * p;
        bcj = fma(dt, bp, bcj);
        const double f = gsum<G>(cj * bp);
        if (st0) {
          const double v = carry - f;
          btb[n] = acc ? btb[n] + v : v;
        }
        carry = f;
        // Step 3 (reverse.hpp:77-80)
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const double pi = pA[i], wi = wA[i];
          M[i] *= pi * p;
          q = fma(wi, M[i], q);               // (w M)_j
        }
        ban = fma(0.5, gsum<G>(q * wm), bdm);  // ba_{n-1} = bd_{n-1} + w bS w^T = bd + (w M w^T)/2
        bVn = bWm / dm + q;                    // bV_{n-1} = bW_{n-1}/d_{n-1} + w (bS + bS^T)
      }
    }
  }
  // reverse.hpp:83-84
  ban -= gsum<G>(bVn * (act ? Wb[0] : 0.0));
  if (st0) {
    bab[0] = ban;
    btb[0] = acc ? btb[0] + carry : carry;
  }
  if (st) {
    bVb[0] = bVn;
    if (!acc) bUb[0] = 0.0;
    double *bcb = bc + L.b * J + j;
    *bcb = acc ? *bcb + bcj : bcj;
  }
}

// =============================================================================
// general_matmul_lower / upper -- reference forward.hpp:285-332 / 346-392.
// Two-pointer merge of the sorted grids t1 (N) and t2 (M); the merge indices are
// uniform inside a group (they depend on the series only).
// =============================================================================
template <int G, int KT, bool LOWER>
__global__ __launch_bounds__(kWave) void k_general(int64_t B, int64_t N, int64_t M, int J, int64_t nrhs,
                                                   const double *t1, int64_t t1_bs, const double *t2, int64_t t2_bs,
                                                   const double *c, int64_t c_bs, const double *U, const double *V,
                                                   const double *Y, double *Z, double *F) {
  const Lane L = lane_of<G>(B);
  const int j = L.j;
  const bool act = j < J;
  const bool st = L.valid && act, st0 = L.valid && j == 0;
  const int jj = act ? j : 0;
  const int64_t k0 = (int64_t)blockIdx.y * KT;
  const int kn = (nrhs - k0 < KT) ? (int)(nrhs - k0) : KT;
  const double *t1b = t1 + L.b * t1_bs, *t2b = t2 + L.b * t2_bs;
  const double *Ub = U + L.b * N * J + jj, *Vb = V + L.b * M * J + jj;
  const double *Yb = Y + L.b * M * nrhs + k0;
  double *Zb = Z + L.b * N * nrhs + k0;
  double *Fb = F ? F + L.b * M * J * nrhs + (int64_t)jj * nrhs + k0 : nullptr;  // row-major F[m, j*nrhs + k]
  const double cj = act ? c[L.b * c_bs + j] : 0.0;

  double Fm[KT];
  const int64_t m0 = LOWER ? 0 : M - 1;
  {
    const double v0 = act ? Vb[m0 * J] : 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) Fm[k] = (k < kn) ? v0 * Yb[m0 * nrhs + k] : 0.0;
  }
  if (Fb && st) {
    // F.row(0).setZero() in both variants (forward.hpp:297, 358); the lower variant then
    // stores Fm into row 0 (forward.hpp:313), the upper variant stores nothing for row M-1.
    for (int k = 0; k < kn; ++k) Fb[k] = LOWER ? Fm[k] : 0.0;
  }
  auto absorb = [&](int64_t m, double dt) {
    const double p = exp(cj * dt);
    const double vm = act ? Vb[m * J] : 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const double yk = (k < kn) ? Yb[m * nrhs + k] : 0.0;
      Fm[k] = fma(vm, yk, p * Fm[k]);
      if (Fb && st && k < kn) Fb[m * J * nrhs + k] = Fm[k];
    }
  };
  auto emit = [&](int64_t n, double dt) {
    const double p = exp(cj * dt);
    const double up = (act ? Ub[n * J] : 0.0) * p;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const double red = gsum<G>(up * Fm[k]);
      if (st0 && k < kn) Zb[n * nrhs + k] += red;
    }
  };
  if (LOWER) {
    double tn = t2b[0];
    int64_t n = 0, m = 1;
    for (; n < N; ++n)
      if (t1b[n] >= tn) break;
    for (; n < N; ++n) {
      tn = t1b[n];
      while (m < M && t2b[m] <= tn) {
        absorb(m, t2b[m - 1] - t2b[m]);
        m++;
      }
      emit(n, t2b[m - 1] - tn);
    }
  } else {
    double tn = t2b[M - 1];
    int64_t n = N - 1, m = M - 2;
    for (; n >= 0; --n)
      if (t1b[n] < tn) break;
    for (; n >= 0; --n) {
      tn = t1b[n];
      while (m >= 0 && t2b[m] > tn) {
        absorb(m, t2b[m] - t2b[m + 1]);
        m--;
      }
      emit(n, tn - t2b[m + 1]);
    }
  }
}

// -----------------------------------------------------------------------------
// general_matmul_* in two data-parallel phases.  The merge of forward.hpp:316-330 only decides WHICH state row an
// output row reads: the state itself, F_m = p_m o F_{m-1} + V_m^T Y_m (lower; mirrored for upper), depends on the
// t2 grid alone.  Phase 1 (k_gm_state) is a plain sweep over m that writes every visited F_m (the `F` output of the
// backprop variant, or a stream-ordered temporary); phase 2 (k_gm_emit) is embarrassingly parallel over the output
// rows: m(n) by binary search in t2, then Z_n += (U_n o exp(c dt)) F_{m(n)}.
// -----------------------------------------------------------------------------
// first index with t2[idx] > x (t2 sorted, length M)
__device__ __forceinline__ int64_t upper_bound_t(const double *t2, int64_t M, double x) {
  int64_t lo = 0, hi = M;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (t2[mid] <= x) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// The same search done by the G lanes of a group together: each round probes G interior points of [lo, hi), the
// group's ballot tells between which two probes the answer lies ((G+1)-ary search: 5 rounds instead of 13 dependent
// loads at M = 4096, G = 8).  x must be uniform inside the group; the result is uniform too.
template <int G>
__device__ __forceinline__ int64_t group_upper_bound_t(const double *t2, int64_t M, double x, int lane) {
  if constexpr (G == 1) {
    return upper_bound_t(t2, M, x);
  } else {
    const int jg = lane & (G - 1), shift = lane & ~(G - 1);
    int64_t lo = 0, hi = M;
    while (lo < hi) {
      const int64_t width = hi - lo;
      const int64_t pos = lo + ((int64_t)(jg + 1) * width) / (G + 1);
      const bool le = t2[pos] <= x;
      const unsigned long long mask = __ballot(le);
      const int cnt = __popcll((mask >> shift) & ((1ull << G) - 1ull));  // probes are sorted: the first cnt are <= x
      const int64_t below = lo + ((int64_t)cnt * width) / (G + 1);       // probe cnt-1
      const int64_t above = lo + ((int64_t)(cnt + 1) * width) / (G + 1); // probe cnt
      const int64_t nlo = (cnt > 0) ? below + 1 : lo;
      const int64_t nhi = (cnt < G) ? above : hi;
      lo = nlo;
      hi = nhi;
    }
    return lo;
  }
}

template <int G, int KT, bool LOWER>
__global__ __launch_bounds__(kWave) void k_gm_state(int64_t B, int64_t N, int64_t M, int J, int64_t nrhs,
                                                    const double *__restrict__ t1, int64_t t1_bs,
                                                    const double *__restrict__ t2, int64_t t2_bs,
                                                    const double *__restrict__ c, int64_t c_bs,
                                                    const double *__restrict__ V, const double *__restrict__ Y,
                                                    double *__restrict__ F) {
  constexpr int PFG = 8;
  const Lane L = lane_of<G>(B);
  const int j = L.j;
  const bool act = j < J;
  const bool st = L.valid && act;
  const int jj = act ? j : 0;
  const int64_t k0 = (int64_t)blockIdx.y * KT;
  const int kn = (nrhs - k0 < KT) ? (int)(nrhs - k0) : KT;
  const double *t1b = t1 + L.b * t1_bs, *t2b = t2 + L.b * t2_bs;
  const double *Vb = V + L.b * M * J + jj;
  const double *Yb = Y + L.b * M * nrhs + k0;
  double *Fb = F + L.b * M * J * nrhs + (int64_t)jj * nrhs + k0;  // row-major F[m, j*nrhs + k] (forward.hpp:300)
  const double cj = act ? c[L.b * c_bs + j] : 0.0;

  // rows the reference's merge visits: lower 0..m_hi, upper m_lo..M-2 (plus the zeroed row 0)
  int64_t nsteps;  // number of absorb steps after the initial row
  if (LOWER) {
    const int64_t ub = upper_bound_t(t2b, M, t1b[N - 1]);  // rows with t2 <= t1[N-1]
    nsteps = (ub >= 1) ? ub - 1 : 0;
  } else {
    const int64_t ub = upper_bound_t(t2b, M, t1b[0]);  // first row with t2 > t1[0]
    nsteps = (ub <= M - 2) ? (M - 1 - ub) : 0;
  }
  const int64_t m0 = LOWER ? 0 : M - 1;
  const bool vec2 = (nrhs % 2 == 0) && kn == KT;
  double Fm[KT];
  {
    const double v0 = act ? Vb[m0 * J] : 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) Fm[k] = (k < kn) ? v0 * Yb[m0 * nrhs + k] : 0.0;
  }
  if (st) {
    // F.row(0).setZero() in both variants (forward.hpp:297, 358); the lower variant then stores Fm into row 0
    // (forward.hpp:313), the upper variant never stores row M-1.
    for (int k = 0; k < kn; ++k) Fb[k] = LOWER ? Fm[k] : 0.0;
  }
  double tprev = t2b[m0];
  double rt[PFG], rv[PFG], ry[PFG][KT];
  auto load_row = [&](int r, int64_t s) {  // s = 1 .. nsteps (clamped)
    const int64_t sc = (s <= nsteps) ? s : (nsteps > 0 ? nsteps : 0);
    const int64_t m = LOWER ? sc : M - 1 - sc;
    rt[r] = t2b[m];
    rv[r] = act ? Vb[m * J] : 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) ry[r][k] = (k < kn) ? Yb[m * nrhs + k] : 0.0;
  };
#pragma unroll
  for (int r = 0; r < PFG; ++r) load_row(r, 1 + r);
  auto step = [&](int r, int64_t s, auto plain_tag) {
    constexpr bool PLAIN = decltype(plain_tag)::value;  // every lane stores its full tile: no store predicate
    const int64_t m = LOWER ? s : M - 1 - s;
    const double tm = rt[r], vm = rv[r];
    double yk[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) yk[k] = ry[r][k];
    load_row(r, s + PFG);
    const double p = exp_decay(cj * (LOWER ? tprev - tm : tm - tprev));
    tprev = tm;
#pragma unroll
    for (int k = 0; k < KT; ++k) Fm[k] = fma(vm, yk[k], p * Fm[k]);
    if (PLAIN || st) {
      double *Fr = Fb + m * J * nrhs;
      if (KT % 2 == 0 && vec2) {  // 16-byte stores: the lane's KT entries are contiguous in the row-major state
#pragma unroll
        for (int k = 0; k < KT; k += 2) reinterpret_cast<double2 *>(Fr)[k / 2] = make_double2(Fm[k], Fm[k + 1]);
      } else {
#pragma unroll
        for (int k = 0; k < KT; ++k)
          if (PLAIN || k < kn) Fr[k] = Fm[k];
      }
    }
  };
  // Full blocks of a wavefront whose lanes all store run branch-free: behind a row guard the compiler cannot count
  // the stores issued after a prefetched load and drains them all (s_waitcnt vmcnt(0)) once per row.
  int64_t nmin = nsteps;
  for (int o = G; o < kWave; o *= 2) {
    const int64_t other = __shfl_xor((long long)nmin, o);
    nmin = other < nmin ? other : nmin;
  }
  const bool plain = __all(st && kn == KT);
  int64_t s0 = 1;
  if (plain) {
    for (; s0 + PFG - 1 <= nmin; s0 += PFG) {
#pragma unroll
      for (int r = 0; r < PFG; ++r) step(r, s0 + r, std::true_type{});
    }
  }
  for (; s0 <= nsteps; s0 += PFG) {
#pragma unroll
    for (int r = 0; r < PFG; ++r) {
      if (s0 + r <= nsteps) step(r, s0 + r, std::false_type{});
    }
  }
}

// One group of G lanes per output row (b, n); all right-hand sides in a loop.
template <int G, bool LOWER>
__global__ __launch_bounds__(kWave) void k_gm_emit(int64_t B, int64_t N, int64_t M, int J, int64_t nrhs,
                                                   const double *__restrict__ t1, int64_t t1_bs,
                                                   const double *__restrict__ t2, int64_t t2_bs,
                                                   const double *__restrict__ c, int64_t c_bs,
                                                   const double *__restrict__ U, const double *__restrict__ V,
                                                   const double *__restrict__ Y, const double *__restrict__ F,
                                                   double *__restrict__ Z) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = g / G;
  const int j = (int)(g % G);
  const bool valid = row < B * N;
  if (!valid) row = B * N - 1;
  const int64_t b = row / N, n = row % N;
  const bool act = j < J;
  const int jj = act ? j : 0;
  const double *t2b = t2 + b * t2_bs;
  const double tn = t1[b * t1_bs + n];
  const int64_t ub = group_upper_bound_t<G>(t2b, M, tn, (int)(threadIdx.x & (kWave - 1)));
  const int64_t m = LOWER ? ub - 1 : ub;       // state row this output reads
  const bool hit = LOWER ? (m >= 0) : (m <= M - 1);  // uniform inside a group
  const int64_t mc = hit ? m : 0;
  const double dt = LOWER ? t2b[mc] - tn : tn - t2b[mc];
  const double cj = act ? c[b * c_bs + j] : 0.0;
  const double up = (act ? U[(b * N + n) * J + jj] : 0.0) * exp(cj * dt);
  // the upper variant never stores its initial row M-1 (forward.hpp:358-375): rebuild it from V, Y
  const bool rebuild = !LOWER && mc == M - 1;
  const double vlast = (rebuild && act) ? V[(b * M + mc) * J + jj] : 0.0;
  const double *Fr = F + ((b * M + mc) * J + jj) * nrhs;
  const double *Yr = Y + (b * M + mc) * nrhs;
  double *Zr = Z + (b * N + n) * nrhs;
  int64_t k = 0;
  if (nrhs % 2 == 0) {  // 16-byte loads of the state row
    for (; k < nrhs; k += 2) {
      double2 f = make_double2(0.0, 0.0);
      if (rebuild) f = make_double2(vlast * Yr[k], vlast * Yr[k + 1]);
      else if (act) f = *reinterpret_cast<const double2 *>(Fr + k);
      double r0 = up * f.x, r1 = up * f.y;
      gsum2<G>(r0, r1);
      if (valid && hit && j == 0) {
        double2 z = *reinterpret_cast<double2 *>(Zr + k);
        z.x += r0; z.y += r1;
        *reinterpret_cast<double2 *>(Zr + k) = z;
      }
    }
  }
  for (; k < nrhs; ++k) {
    const double f = rebuild ? vlast * Yr[k] : (act ? Fr[k] : 0.0);
    const double red = gsum<G>(up * f);
    if (valid && hit && j == 0) Zr[k] += red;
  }
}

// =============================================================================
// get_celerite_matrices -- reference python/celerite2/driver.cpp:422-477.
// A thread owns one term and walks kMatRows rows; consecutive threads write consecutive columns of the same row (a
// complex term its cos/sin column pair as one 16-byte store when the pair is aligned), so a wavefront's stores are dense
// runs and every sincos is evaluated once.  1.15-1.2 ms per 8192 x 4096 rows at J = 8 (0.53-0.56 of the roofline; the
// one-(row, term)-per-thread version with the library sincos inlined ran 1.8 ms at 3 wavefronts per SIMD).
// =============================================================================
constexpr int kMatRows = 8;   // rows per thread of k_matrices
// x sorted (a precondition of the recursions): the largest |x| of a series sits at one of its ends
__device__ __forceinline__ bool matrices_big_phase(double dc_, const double *xb, int64_t N) {
  return !(fabs(dc_) * fmax(fabs(xb[0]), fabs(xb[N - 1])) < kSincosFastMax);
}
__global__ __launch_bounds__(256) void k_matrices(int64_t B, int64_t N, int Jr, int Jc, const double *__restrict__ ar,
                                                  const double *__restrict__ ac, const double *__restrict__ bc,
                                                  const double *__restrict__ dc, int coef_batched,
                                                  const double *__restrict__ x, int64_t x_bs,
                                                  const double *__restrict__ diag, double *__restrict__ a,
                                                  double *__restrict__ U, double *__restrict__ V,
                                                  const unsigned long long *__restrict__ gate) {
  // A thread owns ONE term and walks kMatRows rows of one series: its coefficients are loaded once and the row
  // iterations are independent, so their x loads / sincos / stores overlap (one (row, term) pair per thread spent
  // 74 % of its 7500 cycles waiting for dependent loads).  Consecutive threads hold consecutive terms of a row, so a
  // wavefront's stores are dense runs (a complex term its cos / sin column pair as one 16-byte store when aligned).
  if (gate_none_closed(gate)) return;   // (a fallback launch with nothing to do)
  const int Q = Jr + Jc, J = Jr + 2 * Jc;
  const int rpi = 256 / Q;                      // rows per block iteration
  const int q = (int)threadIdx.x % Q, r = (int)threadIdx.x / Q;
  if (r >= rpi) return;
  const int64_t n0 = (int64_t)blockIdx.x * rpi * kMatRows + r;
  for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
    if (gate_closed(gate, b)) continue;
    const double *arb = ar + (coef_batched ? b * Jr : 0);
    const double *acb = ac + (coef_batched ? b * Jc : 0), *bcb = bc + (coef_batched ? b * Jc : 0),
                 *dcb = dc + (coef_batched ? b * Jc : 0);
    double asum = 0.0, c0 = 0.0, c1 = 0.0, c2_ = 0.0;
    bool big = false;
    if (q == 0) {  // a = diag + sum(ar) + sum(ac), in the reference's summation order (driver.cpp:456-458)
      for (int i = 0; i < Jr; ++i) asum += arb[i];
      for (int i = 0; i < Jc; ++i) asum += acb[i];
    }
    if (q < Jr) c0 = arb[q];
    else {
      c0 = acb[q - Jr]; c1 = bcb[q - Jr]; c2_ = dcb[q - Jr];
      big = matrices_big_phase(c2_, x + b * x_bs, N);   // k_matrices_big writes this term's columns
    }
    const int ind = q < Jr ? q : Jr + 2 * (q - Jr);
#pragma unroll 1   // one row in flight per thread: 86 registers, 5 wavefronts per SIMD (unrolled by 2: 106 / 4, 15 % slower)
    for (int k = 0; k < kMatRows; ++k) {
      const int64_t n = n0 + (int64_t)k * rpi;
      if (n >= N) break;
      const int64_t row = b * N + n;
      if (q == 0) a[row] = diag[row] + asum;
      if (big) continue;
      double *Un = U + row * J + ind, *Vn = V + row * J + ind;
      if (q < Jr) {
        *Vn = 1.0;
        *Un = c0;
        continue;
      }
      double sn, cs;
      const double ph = c2_ * x[b * x_bs + n];
      sincos_cw_fast(ph, sn, cs);
      // (the range test above looks at the ENDS of the grid.  A row whose phase leaves the range of the branch-free
      // reduction although the ends do not -- an UNSORTED x, which driver.cpp:460-474 accepts -- is marked here and
      // rewritten by k_matrices_big, launched right behind this kernel, with the library's sincos)
      if (!(fabs(ph) < kSincosFastMax)) sn = cs = __builtin_nan("");
      const double u0 = c0 * cs + c1 * sn, u1 = c0 * sn - c1 * cs;
      if ((Jr & 1) == 0) {  // J even and ind even: the pair is 16-byte aligned
        *reinterpret_cast<double2 *>(Vn) = make_double2(cs, sn);
        *reinterpret_cast<double2 *>(Un) = make_double2(u0, u1);
      } else {
        Vn[0] = cs; Vn[1] = sn;
        Un[0] = u0; Un[1] = u1;
      }
    }
  }
}

// What k_matrices left out: phases beyond the range of the branch-free sincos (raw Julian dates times a fast frequency) --
// whole terms (decided from the ends of the grid, the columns k_matrices skipped) and single rows of an unsorted grid
// (the reference has no sortedness precondition here, driver.cpp:460-474).  One thread per ROW; it reads x and returns
// in the common case, so the library's large-argument reduction (and its 160 registers) stays out of the kernel that
// does the work: 0.27 GB of reads on top of its 2.4 GB at 8192 x 4096 x 8.
template <bool GATED>
__global__ __launch_bounds__(256) void k_matrices_big(int64_t B, int64_t N, int Jr, int Jc, const double *__restrict__ ac,
                                                      const double *__restrict__ bc, const double *__restrict__ dc,
                                                      int coef_batched, const double *__restrict__ x, int64_t x_bs,
                                                      double *__restrict__ U, double *__restrict__ V,
                                                      const unsigned long long *__restrict__ gate) {
  if (GATED && gate_none_closed(gate)) return;   // (a fallback launch with nothing to do)
  // (GATED: grid-stride -- a gated launch comes with a small grid, most of its groups are closed to it; otherwise one row per
  // thread and no loop: the plain form is 1.8 x slower with the loop around it)
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < B * N; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = g / N, n = g - b * N;
    if (GATED && gate_closed(gate, b)) continue;
    const int J = Jr + 2 * Jc;
    const int64_t o = coef_batched ? b * Jc : 0;
    const double *xb = x + b * x_bs;
    const double xn = xb[n], xm = fmax(fabs(xb[0]), fabs(xb[N - 1]));
    for (int i = 0; i < Jc; ++i) {
      const double d_ = dc[o + i], ph = d_ * xn;
      if ((fabs(ph) < kSincosFastMax) && (fabs(d_) * xm < kSincosFastMax)) continue;   // k_matrices wrote this pair
      double sn, cs;
      sincos(ph, &sn, &cs);
      const double a_ = ac[o + i], b_ = bc[o + i];
      double *Un = U + g * J + Jr + 2 * i, *Vn = V + g * J + Jr + 2 * i;
      Vn[0] = cs; Vn[1] = sn;
      Un[0] = a_ * cs + b_ * sn; Un[1] = a_ * sn - b_ * cs;
    }
    if (!GATED) break;
  }
}

// K[b, n, m] = k(t1[n] - t2[m])  (terms.py:58-79): a thread per output, the terms in registers' reach through L1 (a few
// doubles per series); branch-free sincos inside its range, the library's beyond
__global__ __launch_bounds__(256) void k_kernel_values(int64_t B, int64_t N, int64_t M, int Jr, int Jc,
                                                       const double *__restrict__ ar, const double *__restrict__ cr,
                                                       const double *__restrict__ ac, const double *__restrict__ bc,
                                                       const double *__restrict__ cc, const double *__restrict__ dc,
                                                       int coef_batched, const double *__restrict__ t1, int64_t t1_bs,
                                                       const double *__restrict__ t2, int64_t t2_bs, double *__restrict__ K) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * N * M) return;
  const int64_t b = g / (N * M), r = g - b * N * M, n = r / M, m = r - n * M;
  const double tau = fabs(t1[b * t1_bs + n] - t2[b * t2_bs + m]);
  const int64_t orr = coef_batched ? b * Jr : 0, oc = coef_batched ? b * Jc : 0;
  double k = 0.0;
  for (int i = 0; i < Jr; ++i) k = fma(ar[orr + i], exp(-cr[orr + i] * tau), k);
  for (int i = 0; i < Jc; ++i) {
    const double ph = dc[oc + i] * tau;
    double sn, cs;
    if (fabs(ph) < kSincosFastMax) sincos_cw_fast(ph, sn, cs);   // (dc may be negative; NaN takes the library path)
    else sincos(ph, &sn, &cs);
    k = fma(exp(-cc[oc + i] * tau), fma(ac[oc + i], cs, bc[oc + i] * sn), k);
  }
  K[g] = k;
}

// The same on TILES (round 6): a workgroup fills 32 rows x 64 columns.  cos / sin of a complex term's phase is taken once per
// tile ROW and once per COLUMN relative to the tile's first row time t0 -- phi_n = dc (t1[n] - t0), psi_m = dc (t2[m] - t0) --
// and combined per entry by the angle-addition formulas: cos(dc tau) = cos phi cos psi + sin phi sin psi, sin(dc |tau|) =
// sgn(tau) (sin phi cos psi - cos phi sin psi), tau = t1[n] - t2[m].  Per entry and term that leaves ONE exponential (the
// 16-instruction decay kernel) and a handful of multiply-adds instead of an exponential and a sincos (~100 instructions):
// 64 x 4096 x 256, four terms: 1.04 -> ~0.3 ms.  Phases are differences against t0, so their rounding is that of dc tau itself.
// Second session of round 6: the EXPONENTIAL is factored the same way.  A wavefront owns 8 rows x 64 columns; with tmin / tmax the
// smallest / largest of its 8 row times, a column x <= tmin sees exp(-c (t_r - x)) = exp(-c (t_r - tmin)) exp(-c (tmin - x)) and a
// column x >= tmax sees exp(-c (x - t_r)) = exp(-c (tmax - t_r)) exp(-c (x - tmax)) -- every factor an exponential of a NON-POSITIVE
// argument (nothing overflows whatever the gap), the row factors one table per tile (LDS), the column factor one exponential per
// lane and term instead of eight.  Columns that fall strictly inside the 8 rows' range (one row group of a tile column in ~8: a
// wavefront-uniform test) take the exponential per entry as before, as do growing terms (c < 0).  Per entry and term: one multiply.
constexpr int kKvRows = 32, kKvCols = 64;
__host__ __device__ inline size_t kv_lds_doubles(int64_t Jr, int64_t Jc) {
  return (size_t)2 * Jc * kKvRows + kKvRows + (size_t)2 * (Jr + Jc) * kKvRows + (size_t)4 * Jc * kKvRows + 8;
}
__global__ __launch_bounds__(256, 4) void k_kernel_values_tile(int64_t N, int64_t M, int Jr, int Jc, const double *__restrict__ ar,
                                                            const double *__restrict__ cr, const double *__restrict__ ac,
                                                            const double *__restrict__ bc, const double *__restrict__ cc,
                                                            const double *__restrict__ dc, int coef_batched,
                                                            const double *__restrict__ t1, int64_t t1_bs,
                                                            const double *__restrict__ t2, int64_t t2_bs, double *__restrict__ K) {
  extern __shared__ double kv_lds[];   // [Jc][32] cos phi, [Jc][32] sin phi, [32] t1 rows, [Jr + Jc][32] row factors up / down, [Jc][32] x 4 their products with cos / sin phi, tmin, tmax
  const int JT = Jr + Jc;
  double *cs_n = kv_lds, *sn_n = cs_n + (size_t)Jc * kKvRows, *tn_s = sn_n + (size_t)Jc * kKvRows;
  double *e_up = tn_s + kKvRows, *e_dn = e_up + (size_t)JT * kKvRows;
  double *ec_up = e_dn + (size_t)JT * kKvRows, *es_up = ec_up + (size_t)Jc * kKvRows, *ec_dn = es_up + (size_t)Jc * kKvRows,
         *es_dn = ec_dn + (size_t)Jc * kKvRows, *tmn = es_dn + (size_t)Jc * kKvRows, *tmx = tmn + 4;
  const int64_t b = blockIdx.z, n0 = (int64_t)blockIdx.y * kKvRows, m = (int64_t)blockIdx.x * kKvCols + (threadIdx.x & 63);
  const int rg = threadIdx.x >> 6;   // rows rg * 8 .. rg * 8 + 7 of the tile
  const int64_t orr = coef_batched ? b * Jr : 0, oc = coef_batched ? b * Jc : 0;
  const double *t1b = t1 + b * t1_bs, *t2b = t2 + b * t2_bs;
  const double t0 = t1b[n0];
  if (threadIdx.x < kKvRows) tn_s[threadIdx.x] = t1b[n0 + threadIdx.x < N ? n0 + threadIdx.x : N - 1];
  for (int q = threadIdx.x; q < Jc * kKvRows; q += 256) {
    const int i = q / kKvRows, r = q - i * kKvRows;
    const double ph = dc[oc + i] * (t1b[n0 + r < N ? n0 + r : N - 1] - t0);
    double sn, cs;
    sincos_cw(ph, sn, cs);
    cs_n[q] = cs; sn_n[q] = sn;
  }
  __syncthreads();
  if (threadIdx.x < 4) {   // smallest / largest time of each group of 8 rows (the rows need not be sorted)
    double lo = tn_s[threadIdx.x * 8], hi = lo;
    for (int r = 1; r < 8; ++r) { const double v = tn_s[threadIdx.x * 8 + r]; lo = fmin(lo, v); hi = fmax(hi, v); }
    tmn[threadIdx.x] = lo; tmx[threadIdx.x] = hi;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < JT * kKvRows; q += 256) {
    const int i = q / kKvRows, r = q - i * kKvRows;
    const double c_ = i < Jr ? cr[orr + i] : cc[oc + i - Jr];
    const double tr = tn_s[r];
    const double eu = exp_decay(-c_ * (tr - tmn[r >> 3])), ed = exp_decay(-c_ * (tmx[r >> 3] - tr));   // (c < 0: never read)
    e_up[q] = eu; e_dn[q] = ed;
    if (i >= Jr) {   // a complex term: the row's share of e (a cos + b sin), see the fast path below
      const int qc = (i - Jr) * kKvRows + r;
      ec_up[qc] = eu * cs_n[qc]; es_up[qc] = eu * sn_n[qc];
      ec_dn[qc] = ed * cs_n[qc]; es_dn[qc] = ed * sn_n[qc];
    }
  }
  __syncthreads();
  const bool live = m < M;
  const double xm = t2b[live ? m : M - 1], dx = xm - t0;
  const double tlo = tmn[rg], thi = tmx[rg];
  const bool below = xm <= tlo, above = xm >= thi;
  const bool direct = __any(!(below || above));          // some column of this wavefront lies inside the rows' range (or is NaN)
  const double far = below ? tlo - xm : xm - thi;        // >= 0 where it is used
  const double *e_row = (below ? e_up : e_dn) + rg * 8;  // + term * 32 + r
  double k[8], tau[8], sg[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const double d = tn_s[rg * 8 + r] - xm;
    tau[r] = fabs(d); sg[r] = d < 0.0 ? -1.0 : 1.0; k[r] = 0.0;
  }
  // e[r] = exp(-c tau[r]) of term ti (rate c_)
  auto decay8 = [&](const double c_, const int ti, double (&e)[8]) __attribute__((always_inline)) {
    if (c_ >= 0.0 && !direct) {          // (uniform) factored: row table x one exponential per lane
      const double fm = exp_decay(-c_ * far);
#pragma unroll
      for (int r = 0; r < 8; ++r) e[r] = e_row[ti * kKvRows + r] * fm;
    } else if (c_ >= 0.0) {              // a decaying term -- the 16-instruction kernel
#pragma unroll
      for (int r = 0; r < 8; ++r) e[r] = exp_decay(-c_ * tau[r]);
    } else {                             // a growing one takes the library's exp
#pragma unroll
      for (int r = 0; r < 8; ++r) e[r] = exp(-c_ * tau[r]);
    }
  };
  for (int i = 0; i < Jr; ++i) {
    const double a_ = ar[orr + i];
    double e[8];
    decay8(cr[orr + i], i, e);
#pragma unroll
    for (int r = 0; r < 8; ++r) k[r] = fma(a_, e[r], k[r]);
  }
  // Fast path of a complex term (a column outside its wavefront's rows: the sign of tau is one for all eight): with phi_r, psi the
  // phases of row and column against t0,  a cos(dc tau) + b sin(dc |tau|) = cos phi_r (a cos psi - b s sin psi) + sin phi_r (a sin psi
  // + b s cos psi),  s = sgn(tau) -- two per-COLUMN numbers; times the column's exponential they meet the row's E cos phi, E sin phi
  // from the tables: TWO multiply-adds and two LDS reads per entry and term (nine and three before).  The kernel is bound by its VALU
  // instructions (893 per wavefront, four wavefronts per SIMD 88 % VALU-active: rocprofv3 --pmc): 348 -> 298 us at 64 x 4096 x 256 with
  // the registers capped at 128 (uncapped the two paths take 173: 412 us).  A workgroup that loops over the column tiles with ONE set of
  // row tables (a quarter of the instructions) spills under that cap: 389 us, not taken.
  const double sgc = below ? 1.0 : -1.0;
  const double *ec_row = (below ? ec_up : ec_dn) + rg * 8, *es_row = (below ? es_up : es_dn) + rg * 8;
  for (int i = 0; i < Jc; ++i) {
    const double a_ = ac[oc + i], b_ = bc[oc + i];
    double sm, cm;
    sincos_cw(dc[oc + i] * dx, sm, cm);
    if (cc[oc + i] >= 0.0 && !direct) {   // (uniform)
      const double fm = exp_decay(-cc[oc + i] * far), bs = b_ * sgc;
      const double Ac = fm * fma(a_, cm, -bs * sm), As = fm * fma(a_, sm, bs * cm);
#pragma unroll
      for (int r = 0; r < 8; ++r) k[r] = fma(ec_row[i * kKvRows + r], Ac, fma(es_row[i * kKvRows + r], As, k[r]));
      continue;
    }
    const double *cr_ = cs_n + (size_t)i * kKvRows + rg * 8, *sr_ = sn_n + (size_t)i * kKvRows + rg * 8;
    double e[8];
    decay8(cc[oc + i], Jr + i, e);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const double cn = cr_[r], sn = sr_[r];
      const double cosv = fma(cn, cm, sn * sm), sinv = sg[r] * fma(sn, cm, -cn * sm);
      k[r] = fma(e[r], fma(a_, cosv, b_ * sinv), k[r]);
    }
  }
  if (live) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t n = n0 + rg * 8 + r;
      if (n < N) K[(b * N + n) * M + m] = k[r];
    }
  }
}

// out[b, m] = sum_n Z[b, n, m]^2 / d[b, n]: the quadratic form of the predictive variance from the LOWER solve alone,
// diag(Kxs K^-1 Kxs^T)_m = sum_n (L^-1 Kxs^T)_nm^2 / d_n (K = L D L^T; core.py:134-140 reaches the same number through
// apply_inverse = both solves and a second pass over the two N x M arrays).  One pass over Z: grid (ceil(M / 64), B), the sixteen
// wavefronts of a workgroup take interleaved rows.
__global__ __launch_bounds__(1024) void k_colsumsq_over_d(int64_t N, int64_t M, const double *__restrict__ Z,
                                                          const double *__restrict__ d, double *__restrict__ out) {
  __shared__ double part[16][64];
  const int64_t b = blockIdx.y, m = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int w = threadIdx.x >> 6;
  const double *zb = Z + b * N * M + (m < M ? m : M - 1), *db = d + b * N;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // (four rows in flight per wavefront: the loop is bound by load latency)
  int64_t n = w;
  for (; n + 48 < N; n += 64) {
    const double z0 = zb[n * M], z1 = zb[(n + 16) * M], z2 = zb[(n + 32) * M], z3 = zb[(n + 48) * M];
    const double d0 = db[n], d1 = db[n + 16], d2 = db[n + 32], d3 = db[n + 48];
    a0 = fma(z0, z0 / d0, a0); a1 = fma(z1, z1 / d1, a1); a2 = fma(z2, z2 / d2, a2); a3 = fma(z3, z3 / d3, a3);
  }
  for (; n < N; n += 16) { const double z = zb[n * M]; a0 = fma(z, z / db[n], a0); }
  part[w][threadIdx.x & 63] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (w == 0 && m < M) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += part[q][threadIdx.x];
    out[b * M + m] = s;
  }
}

// Z = Y * sqrt(d)[:, None]   (numpy.py:101)
__global__ void k_scale_sqrt(int64_t total, int64_t nrhs, const double *d, const double *Y, double *Z) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < total) Z[g] = Y[g] * sqrt(d[g / nrhs]);
}

}  // namespace c2

// =============================================================================
// C-ABI launchers
// =============================================================================
using namespace c2;

namespace {
thread_local char g_err[256] = "";
inline int hip_check(hipError_t e) {
  if (e == hipSuccess) return C2_OK;
  snprintf(g_err, sizeof(g_err), "%s", hipGetErrorString(e));
  return C2_ERR_HIP;
}
inline int check_launch() { return hip_check(hipGetLastError()); }
inline dim3 grid_for(int64_t B, int G, int64_t ytiles = 1) {
  const int64_t lanes = B * G;
  return dim3((unsigned)((lanes + kWave - 1) / kWave), (unsigned)ytiles, 1);
}
inline int check_dims(int64_t B, int64_t N, int64_t J) {
  if (B < 1 || N < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  return C2_OK;
}
}  // namespace

#define C2_DISPATCH_G(G_, ...)                                  \
  switch (G_) {                                                 \
    case 1: { constexpr int G = 1; __VA_ARGS__; } break;        \
    case 2: { constexpr int G = 2; __VA_ARGS__; } break;        \
    case 4: { constexpr int G = 4; __VA_ARGS__; } break;        \
    case 8: { constexpr int G = 8; __VA_ARGS__; } break;        \
    case 16: { constexpr int G = 16; __VA_ARGS__; } break;      \
    default: { constexpr int G = 32; __VA_ARGS__; } break;      \
  }

// wide models (C2_FAST_WIDTH < J <= C2_MAX_WIDTH): a workgroup per series, the state in LDS (c2_wide.hip)
extern "C" int c2_wide_factor(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                              const double *a, const double *U, const double *V, double *d, double *W, double *S,
                              int32_t *flag, c2_stream_t stream);
extern "C" int c2_wide_sweep(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                             int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V, const double *Y,
                             double *Z, double *F, int zero_z, c2_stream_t stream);
extern "C" int c2_wide_sweep_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                 int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                 const double *Y, const double *Z, const double *F, const double *bZ, double *bt, double *bc,
                                 double *bU, double *bV, double *bY, c2_stream_t stream);
extern "C" int c2_wide_factor_rev(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                  int64_t c_bs, const double *U, const double *d, const double *W, const double *S,
                                  const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                                  double *bV, int accumulate, c2_stream_t stream);
extern "C" int c2_wide_general(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1,
                               int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c, int64_t c_bs,
                               const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z,
                               c2_stream_t stream);

extern "C" int c2_internal_matmul_chunked(int lower, int64_t B, int64_t N, int64_t J, int64_t nrhs, int64_t Lc,
                                          const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                          const double *U, const double *V, const double *Y, double *Z, double *F,
                                          int zero_z, c2_stream_t stream);

extern "C" int c2_internal_sweep1(int lower, int solve, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                  const double *c, int64_t c_bs, const double *U, const double *V, const double *Y,
                                  double *Z, double *F, int zero_z, c2_stream_t stream);

extern "C" int c2_internal_sweepT(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                  int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                  const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream);
extern "C" int c2_internal_sweepT_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                      double *bc, double *bU, double *bV, double *bY, c2_stream_t stream);
extern "C" int c2_internal_sweep1_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                      double *bc, double *bU, double *bV, double *bY, c2_stream_t stream);

extern "C" int c2_internal_sweepK(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                  int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                  const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream);

extern "C" int c2_internal_matmul_lower_mfma(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                                             const double *c, int64_t c_bs, const double *U, const double *V,
                                             const double *d, const double *Y, double *Z, int zero_z,
                                             c2_stream_t stream);
static bool use_mfma() {
  return !(opt::has(opt::k_mfma) && opt::ival(opt::k_mfma) == 0);
}

extern "C" int c2_internal_use_timepar_solve(int64_t B, int64_t N, int64_t J);
extern "C" size_t c2_internal_timepar_solve_doubles(int64_t B, int64_t N, int64_t J);
extern "C" int c2_internal_solve_timepar(int lower, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                         const double *c, int64_t c_bs, const double *U, const double *W,
                                         const double *Y, double *Z, double *scratch, c2_stream_t stream);
extern "C" int c2_internal_tpg_short_chunks(int64_t B, int64_t N);
#define C2_DECL_SC(R_)                                                                                               \
  extern "C" size_t c2_internal_solve_chunks_doubles##R_(int64_t B, int64_t N, int64_t J);                          \
  extern "C" int c2_internal_solve_chunks##R_(int lower, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, \
                                              const double *c, int64_t c_bs, const double *U, const double *W,      \
                                              const double *Y, double *Z, double *scratch, c2_stream_t stream,      \
                                              int64_t nrhs, double *F);
C2_DECL_SC(64)
C2_DECL_SC(32)
C2_DECL_SC(16)
#undef C2_DECL_SC
// shortest series the long-series forms of the solves with F / several right-hand sides and of the reverse sweeps take
// (C2_LONG_MIN_ROWS overrides; measured below)
static int64_t long_min_rows() {
  const int64_t v = opt::ival(opt::k_long_min_rows);
  return v >= 128 ? v : 512;
}
// shapes the chunked products of c2_scan.hip take: too few (series x rhs-tile) chains to fill the chip with the row-by-row
// kernel, series long enough to cut (C2_SCAN_MIN_ROWS overrides the 1024)
static bool matmul_chunked_shape(int64_t B, int64_t N, int64_t J, int64_t nrhs) {
  const int64_t chains = B * ((nrhs + 3) / 4) * group_size(J);  // lanes kept busy by the sequential kernel
  const int64_t min_rows = opt::ival(opt::k_scan_min_rows) >= 256 ? opt::ival(opt::k_scan_min_rows) : 1024;
  if (chains >= (int64_t)kWave * 2048) return false;
  // below 16384 rows the chunked form (~0.5 ns per row and series) must beat a walk of ~0.15 us per row whatever the
  // batch: one series of 4096 rows 0.60 -> 0.08 ms, 64 x 4096 0.63 -> 0.15, 256 x 4096 0.62 -> 0.48, 1024 x 2048 0.36 -> 1.01
  return N >= 16384 || (N >= min_rows && B <= 128 && N >= 8 * B);
}
// shapes the chunk-map solves of c2_timepar_grad.hip take (tools/bench_ops.py, J = 8, ms row by row -> chunk maps):
// 1 x 1024 0.13 -> 0.05, 1 x 4096 + F 0.60 -> 0.06, 64 x 1024 0.13 -> 0.05, 512 x 2048 + F 0.32 -> 0.14, 256 x 4096 + F
// 0.63 -> 0.15; 8 right-hand sides: 1 x 1024 0.32 -> 0.27, 1 x 8192 2.56 -> 0.45 (1 x 512: 0.16 -> 0.24, not taken);
// 2048 x 1024 (32768 chunks) + F 0.20 -> 0.25, not taken; 64 x 4096 with 8 right-hand sides 1.29 -> 0.40; 512 x 4096 with 8:
// 1.45 -> 1.9, not taken -- the rule below is that cost model
extern "C" int c2_internal_sweep_cols(int lower, int solve, int64_t B, int64_t N, int64_t Jw, int64_t nrhs, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, double *Z, int64_t *B8, c2_stream_t stream);
extern "C" int c2_internal_sweep_cols_rev(int lower, int solve, int64_t B, int64_t N, int64_t Jw, int64_t nrhs, const double *t,
                                          int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                          const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                          double *bc, double *bU, double *bV, double *bY, int64_t *B8, c2_stream_t stream);
extern "C" size_t c2_internal_solve_cols_doubles(int64_t B, int64_t N, int64_t J, int64_t nrhs);
extern "C" int c2_internal_solve_cols(int lower, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                                      const double *c, int64_t c_bs, const double *U, const double *W, const double *Y,
                                      double *Z, double *scratch, c2_stream_t stream);
static bool solve_chunks_shape(int64_t B, int64_t N, int64_t J, int64_t nrhs) {
  if (J > 8 || nrhs > 64 || N < long_min_rows()) return false;
  const int64_t k64 = B * ((N + 63) / 64);   // chunks of one launch: the right-hand sides run one after the other
  if (k64 > 32768) return false;
  // measured (ms; the constants live in c2_dispatch.hpp): a column over k64 chunks 0.05 + 6e-6 k64 (64 x 4096: 0.07,
  // 512 x 4096: 0.24); row by row N (0.15 + 0.025 nrhs) us whatever the batch (4096 rows: 0.63 with one right-hand side,
  // 1.29 with 8)
  const double chunked_ms = (double)nrhs * (opt::val(opt::k_solve_chunk_col_ms) + opt::val(opt::k_solve_chunk_ms) * (double)k64);
  double rows_ms = 1e-3 * (double)N * (opt::val(opt::k_solve_row_us) + opt::val(opt::k_solve_row_rhs_us) * (double)nrhs);
  if (nrhs > 8) {
    // nine and more right-hand sides run with lanes over the right-hand sides (k_sweepK: 16 / 32 / 64 lanes per series,
    // tiles of 64 columns side by side in the grid): a row costs one wavefront ~0.26 us whatever nrhs, and the launch as
    // long as one wavefront needs while the chip holds all of them (profiles/r04_large_nrhs.md: 1 x 4096 with 256 or 1024
    // right-hand sides 1.05 ms, 64 x 4096 with 256: 1.14 ms -- the linear-in-nrhs extrapolation above said 27 / 105 ms and
    // sent 64 columns through the chunk maps one by one: 2.6 ms)
    int64_t KL = 16;
    while (KL < 64 && KL < nrhs) KL *= 2;
    const double waves = (double)((B + 64 / KL - 1) / (64 / KL)) * (double)((nrhs + KL - 1) / KL);
    rows_ms = 1e-3 * (double)N * 0.26 * (waves > 2048.0 ? waves / 2048.0 : 1.0);
  }
  if (N >= 16384) return k64 * nrhs <= 32768 || chunked_ms < rows_ms;
  return chunked_ms < rows_ms;
}
// Many right-hand sides on a small batch: the chunk maps with lanes over the columns (c2_solve_cols.hip) against the row-by-row
// kernel with lanes over the right-hand sides.  Measured (profiles/r04_large_nrhs.md): a row costs a wavefront of k_sweepK
// ~0.26 us and the launch as long as ONE wavefront needs while the chip holds all of them; the chunked form walks 2 x Lc
// rows per wavefront (~0.3 us a step) behind a chain of ~0.3 us per chunk and three launches.
static bool solve_cols_shape(int64_t B, int64_t N, int64_t J, int64_t nrhs) {
  if (opt::has(opt::k_solve_cols)) return opt::ival(opt::k_solve_cols) != 0 && J <= 16 && N >= 2 && B <= 65535;
  if (J > 16 || nrhs < 16 || N < opt::ival(opt::k_solve_cols_min_rows) || B > 65535) return false;
  const double wide = J > 8 ? 1.25 : 1.0;
  int64_t KL = 16;
  while (KL < 64 && KL < nrhs) KL *= 2;
  const double tiles = (double)((nrhs + 63) / 64);
  const double waves = (double)((B + 64 / KL - 1) / (64 / KL)) * (double)((nrhs + KL - 1) / KL);
  // row by row: one wavefront's N steps of 0.26 us; a launch of w x 1024 wavefronts takes (1 + 0.33 w) of that while the chip
  // holds it (w <= 1) and 1.08 w beyond (round 6, tools/cols_probe.py on 60 shapes: 64 wavefronts 1.07 ms, 512: 1.28, 1024:
  // 1.40 - 1.44, 2048: 2.30 - 2.38, 4096: 4.5; the first model had the chip hold 2048 at the price of one)
  const double w = waves / 1024.0;
  const double rows_ms = 1e-3 * (double)N * 0.26 * wide * (1.0 + 0.33 * w > 1.08 * w ? 1.0 + 0.33 * w : 1.08 * w);
  // chunk maps over the columns: the launches and their temporary + a price per wavefront-walk of 64 rows.  Re-fitted twice in round 6
  // (tools/cols_probe.py, 60 shapes): after the first walk lost its 442 registers 0.09 ms + 2.5e-5 per walk (round 5: 0.10 + 3.4e-5), with
  // the walks on precomputed row records 0.08 + 2.05e-5, width 16 0.20 + 2.9e-5; 64 x 4096 x 512: 0.84 against 1.28 ms row by row,
  // 256 x 4096 x 128: 1.01 against 1.29 -- both were missed by the first model
  int64_t Lc = 64;
  while (Lc < 1024 && (double)B * (double)((N + Lc - 1) / Lc) * tiles > 8192.0) Lc *= 2;   // (the plan of c2_solve_cols.hip)
  const double K = (double)((N + Lc - 1) / Lc), cw = (double)B * K * (tiles + 1.0) * (double)Lc / 64.0;
  const double cols_ms = J > 8 ? 0.20 + 2.9e-5 * cw : 0.08 + 2.05e-5 * cw;
  return cols_ms < rows_ms;
}
static bool solve_chunks_enabled() {
  return !(opt::has(opt::k_timepar) && opt::ival(opt::k_timepar) == 0);   // the switch of the time-parallel solves: 0 keeps them row by row
}
template <bool LOWER, bool SOLVE>
static int launch_sweep(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                        int64_t c_bs, const double *U, const double *V, const double *Y, double *Z, double *F,
                        int zero_z, c2_stream_t stream) {
  if (int e = check_dims(B, N, J)) return e;
  if (nrhs < 1 || !t || !c || !U || !V || !Y || !Z) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH)
    return c2_wide_sweep(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
  hipStream_t s = (hipStream_t)stream;
  if (!SOLVE) {
    // Long series with too few (series x rhs-tile) chains to fill the chip: the matmul recurrence is linear with
    // a diagonal transition, so it is cut into time chunks that run in parallel (c2_scan.hip).
    if (matmul_chunked_shape(B, N, J, nrhs)) {
      // J = 16 with 16 / 32 / 64 right-hand sides and no workspace: the blocks of 16 rows are dense fp64 contractions
      // on the matrix cores (c2_mfma.hip).  C2_MFMA=0 keeps the VALU path (A/B runs).
      if (LOWER && !F && N >= 16384 && use_mfma()) {
        const int e = c2_internal_matmul_lower_mfma(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, nullptr, Y, Z, zero_z, stream);
        if (e != C2_ERR_UNSUPPORTED) return e;
      }
      // chunk length: the walks of the chunks shrink with Lc, the fold of their carries (one step per chunk) grows with
      // N / Lc -- the power of two next to 0.5 sqrt(N), longer once ~2048 units per rhs slab are in flight (one series, J = 8: 20000 rows 0.79 -> 0.16 ms at Lc = 128, 1e5 rows 0.80 -> 0.31
      // at 256, 1e6 rows 1.28 -> 1.18 at 512; C2_SCAN_MIN_CHUNK overrides)
      int64_t Lc = 64;
      while (Lc < 1024 && 4 * Lc * Lc < N) Lc *= 2;
      if (opt::has(opt::k_scan_min_chunk) && opt::ival(opt::k_scan_min_chunk) >= 64) Lc = opt::ival(opt::k_scan_min_chunk);
      while (Lc < 16384 && B * ((N + 2 * Lc - 1) / (2 * Lc)) >= 2048) Lc *= 2;
      return c2_internal_matmul_chunked(LOWER ? 1 : 0, B, N, J, nrhs, Lc, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z,
                                        stream);
    }
  }
  if (SOLVE && !F && solve_cols_shape(B, N, J, nrhs)) {
    // many right-hand sides on a small batch (apply_inverse on an N x M matrix, core.py:56-60): chunk maps with lanes over
    // the columns, every column in the same three launches (c2_solve_cols.hip).  Scratch is a stream-ordered temporary;
    // not inside graph captures.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capturing);
    if (capturing == hipStreamCaptureStatusNone) {
      const size_t nd = c2_internal_solve_cols_doubles(B, N, J, nrhs);
      void *tmp = nullptr;
      if (nd > 0 && c2::temp_alloc(&tmp, nd * sizeof(double), s) == hipSuccess) {
        int rc = c2_internal_solve_cols(LOWER ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, (double *)tmp, stream);
        if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
        if (rc != C2_ERR_UNSUPPORTED) return rc;
      }
      (void)hipGetLastError();
    }
  }
  if (SOLVE && solve_chunks_shape(B, N, J, nrhs) && solve_chunks_enabled()) {
    // a small batch of series of 512 rows and more: chunk maps, from 8192 rows with the chain over the chunks in two levels
    // (c2_timepar_grad.hip; every width up to 8), right-hand side by right-hand side, the workspace written on the way if asked for (one series of 1e5
    // rows, 8 right-hand sides: 31 ms row by row).  Scratch is a stream-ordered temporary; not inside graph captures.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capturing);
    if (capturing == hipStreamCaptureStatusNone) {
      const int sh = c2_internal_tpg_short_chunks(B, N);
      const size_t nd = sh == 2 ? c2_internal_solve_chunks_doubles16(B, N, J)
                                : (sh == 1 ? c2_internal_solve_chunks_doubles32(B, N, J) : c2_internal_solve_chunks_doubles64(B, N, J));
      void *tmp = nullptr;
      if (nd > 0 && c2::temp_alloc(&tmp, nd * sizeof(double), s) == hipSuccess) {
        int rc = (sh == 2 ? c2_internal_solve_chunks16 : (sh == 1 ? c2_internal_solve_chunks32 : c2_internal_solve_chunks64))(LOWER ? 1 : 0, B, N, J, t, t_bs, c, c_bs, U, V,
                                                                               Y, Z, (double *)tmp, stream, nrhs, F);
        if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
        return rc;
      }
      (void)hipGetLastError();
    }
  }
  if (SOLVE && nrhs == 1 && !F && c2_internal_use_timepar_solve(B, N, J)) {
    // a small batch of long series and one right-hand side: the solve recursion is affine in its state -- chunks of 64
    // rows in parallel, chained, applied (c2_timepar.hip).  Scratch (series longer than 4096 rows only) is a
    // stream-ordered temporary; not inside graph captures.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capturing);
    if (capturing == hipStreamCaptureStatusNone) {
      const size_t nd = c2_internal_timepar_solve_doubles(B, N, J);
      void *tmp = nullptr;
      if (nd > 0 && c2::temp_alloc(&tmp, nd * sizeof(double), s) == hipSuccess) {
        int rc = c2_internal_solve_timepar(LOWER ? 1 : 0, B, N, J, t, t_bs, c, c_bs, U, V, Y, Z, (double *)tmp, stream);
        if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
        return rc;
      }
      (void)hipGetLastError();
    }
  }
  if (nrhs == 1)  // a vector: the tuned single-rhs kernel (c2_sweep.hip)
    return c2_internal_sweep1(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
  if (nrhs <= 7) {  // two to seven: lanes over J, per-series scalars transposed in time (c2_sweep_small.hip)
    const int e = c2_internal_sweepT(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  if (!F && (SOLVE || zero_z) && nrhs >= 9 && nrhs <= 32 && J == 8) {
    // nine to 32 right-hand sides at J = 8 on whole wavefronts of eight series: eight lanes per series, several columns per
    // lane, Y / Z in groups of four rows (c2_sweep_cols.hip); the B % 8 series left over on the kernels below
    int64_t B8 = 0;
    const int e = c2_internal_sweep_cols(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, &B8, stream);
    if (e != C2_ERR_UNSUPPORTED) {
      if (e != C2_OK || B8 == B) return e;
      const int64_t o = B8 * N;
      t += B8 * t_bs; c += B8 * c_bs;
      U += o * J; V += o * J; Y += o * nrhs; Z += o * nrhs;
      B -= B8;
    }
  }
  if (nrhs >= 3) {  // lanes over the right-hand sides (c2_sweep.hip) when the shape fits
    const int e = c2_internal_sweepK(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z,
                                     stream);
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  if (nrhs == 2) {  // a full tile of two: the branch-free body
    constexpr int KT = 2;
    C2_DISPATCH_G(group_size(J),
                  hipLaunchKernelGGL((k_sweep<G, KT, LOWER, SOLVE>), grid_for(B, G, 1), dim3(kWave), 0, s, B, N, (int)J,
                                     nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z));
    return check_launch();
  }
  constexpr int KT = 4;
  C2_DISPATCH_G(group_size(J),
                hipLaunchKernelGGL((k_sweep<G, KT, LOWER, SOLVE>), grid_for(B, G, (nrhs + KT - 1) / KT), dim3(kWave), 0,
                                   s, B, N, (int)J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z));
  return check_launch();
}

extern "C" int c2_internal_generalK(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1,
                                    int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c, int64_t c_bs,
                                    const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z,
                                    c2_stream_t stream);
extern "C" int c2_internal_general_tile(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs,
                                        const double *t1, int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c,
                                        int64_t c_bs, const double *U, const double *V, const double *Y, double *Z,
                                        double *F, double *scratch, c2_stream_t stream);
extern "C" size_t c2_internal_general_tile_doubles(int64_t B, int64_t M, int64_t J, int64_t nrhs);
extern "C" int64_t c2_internal_general_chunks_plan(int64_t B, int64_t M, int64_t nrhs);
extern "C" size_t c2_internal_general_chunks_doubles(int64_t B, int64_t M, int64_t J, int64_t nrhs, int64_t Lc);
extern "C" int c2_internal_general_chunks(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, int64_t Lc,
                                          const double *t1, int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c,
                                          int64_t c_bs, const double *U, const double *V, const double *Y, double *Z,
                                          double *scratch, c2_stream_t stream);
static bool use_general_tile() {
  return !(opt::has(opt::k_general_tile) && opt::ival(opt::k_general_tile) == 0);   // 0: the kernels below (A/B runs, tests of every path)
}
static bool use_generalK() {
  return !(opt::has(opt::k_generalk) && opt::ival(opt::k_generalk) == 0);   // 0: the first-round kernels (A/B runs, tests of both paths)
}
template <bool LOWER>
static int launch_general(int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, int64_t t1_bs,
                          const double *t2, int64_t t2_bs, const double *c, int64_t c_bs, const double *U,
                          const double *V, const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream) {
  if (int e = check_dims(B, N, J)) return e;
  if (M < 1 || nrhs < 1 || !t1 || !t2 || !c || !U || !V || !Y || !Z) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH)
    return c2_wide_general(LOWER ? 1 : 0, B, N, M, J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
  hipStream_t s = (hipStream_t)stream;
  if (zero_z) {
    if (int e = hip_check(hipMemsetAsync(Z, 0, sizeof(double) * B * N * nrhs, s))) return e;
  }
  // a wavefront per series, lanes over 64 consecutive rows of either grid (c2_general_tile.hip)
  // (three right-hand sides on a large batch: lanes over the right-hand sides are ahead, 3.45 against 3.79 ms at B = 8192,
  // N = M = 4096, J = 8, in-process A/B; small batches keep the tiles, which cut long series into chunks)
  // many right-hand sides on a small batch (the products of the predictive covariance, core.py:142-150): lanes over the right-hand
  // sides, the t2 grid cut into chunks that run in parallel (c2_general.hip, CH).  Scratch is a stream-ordered temporary; not inside
  // graph captures.
  if (!F && nrhs >= 33 && M >= 2 && use_generalK()) {
    int64_t Lc = c2_internal_general_chunks_plan(B, M, nrhs);
    if (opt::has(opt::k_general_rhs_chunks)) Lc = opt::ival(opt::k_general_rhs_chunks) >= 8 ? opt::ival(opt::k_general_rhs_chunks) : 0;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capturing);
    if (Lc > 0 && capturing == hipStreamCaptureStatusNone) {
      const size_t nd = c2_internal_general_chunks_doubles(B, M, J, nrhs, Lc);
      void *tmp = nullptr;
      if (nd > 0 && c2::temp_alloc(&tmp, nd * sizeof(double), s) == hipSuccess) {
        int rc = c2_internal_general_chunks(LOWER ? 1 : 0, B, N, M, J, nrhs, Lc, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, (double *)tmp,
                                            stream);
        if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
        if (rc != C2_ERR_UNSUPPORTED) return rc;
      }
      (void)hipGetLastError();
    }
  }
  if (use_general_tile() && !(nrhs == 3 && B >= 512 && use_generalK())) {
    // small batches of long series are cut into chunks along time: a stream-ordered temporary for the chunk maps (not
    // inside a graph capture, and one wavefront per series if the allocation fails)
    void *tmp = nullptr;
    const size_t nd = c2_internal_general_tile_doubles(B, M, J, nrhs);
    if (nd > 0) {
      hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
      (void)hipStreamIsCapturing(s, &capturing);
      if (capturing != hipStreamCaptureStatusNone || c2::temp_alloc(&tmp, nd * sizeof(double), s) != hipSuccess) {
        (void)hipGetLastError();
        tmp = nullptr;
      }
    }
    int e = c2_internal_general_tile(LOWER ? 1 : 0, B, N, M, J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F,
                                     (double *)tmp, stream);
    if (tmp && hipFreeAsync(tmp, s) != hipSuccess && e == C2_OK) e = C2_ERR_HIP;
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  // three or more right-hand sides: lanes over the right-hand sides, one merge event per iteration (c2_general.hip)
  if (nrhs >= 3 && use_generalK()) {
    const int e = c2_internal_generalK(LOWER ? 1 : 0, B, N, M, J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F, 0,
                                       stream);
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  constexpr int KT = 4;
  // Two data-parallel phases (state sweep over t2, then one group per output row) need every state row in memory:
  // the caller's F when it asks for it, otherwise a stream-ordered temporary.  Above kGeneralTempMax bytes of
  // temporary the sequential merge kernel runs instead (no scratch memory at all).
  constexpr size_t kGeneralTempMax = (size_t)32 << 30;
  const size_t fbytes = sizeof(double) * (size_t)B * M * J * nrhs;
  double *Fw = F;
  if (!Fw) {
    // without a caller workspace the state rows are pure overhead (2 x 8 J nrhs bytes per row): two phases pay off
    // for one or two right-hand sides (prediction); beyond that they only draw level with the sequential merge
    // (22.5 against 24 ms at nrhs = 8, measured with -DC2_GM_TWO_PHASE_MAX_NRHS=64), which needs no scratch memory
    if (nrhs > C2_GM_TWO_PHASE_MAX_NRHS || fbytes > kGeneralTempMax || c2::temp_alloc((void **)&Fw, fbytes, s) != hipSuccess) {
      (void)hipGetLastError();
      C2_DISPATCH_G(group_size(J),
                    hipLaunchKernelGGL((k_general<G, KT, LOWER>), grid_for(B, G, (nrhs + KT - 1) / KT), dim3(kWave), 0,
                                       s, B, N, M, (int)J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F));
      return check_launch();
    }
  }
  C2_DISPATCH_G(group_size(J), {
    if (nrhs == 1)
      hipLaunchKernelGGL((k_gm_state<G, 1, LOWER>), grid_for(B, G), dim3(kWave), 0, s, B, N, M, (int)J, nrhs, t1, t1_bs,
                         t2, t2_bs, c, c_bs, V, Y, Fw);
    else
      hipLaunchKernelGGL((k_gm_state<G, KT, LOWER>), grid_for(B, G, (nrhs + KT - 1) / KT), dim3(kWave), 0, s, B, N, M,
                         (int)J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, V, Y, Fw);
    hipLaunchKernelGGL((k_gm_emit<G, LOWER>), grid_for(B * N, G), dim3(kWave), 0, s, B, N, M, (int)J, nrhs, t1, t1_bs,
                       t2, t2_bs, c, c_bs, U, V, Y, (const double *)Fw, Z);
  });
  int rc = check_launch();
  if (!F) {
    const int rf = hip_check(hipFreeAsync(Fw, s));
    if (rc == C2_OK) rc = rf;
  }
  return rc;
}
extern "C" int c2_internal_sweepK_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                      double *bc, double *bU, double *bV, double *bY, c2_stream_t stream);
extern "C" int c2_internal_sweep_rev_long(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs,
                                          const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *U,
                                          const double *V, const double *Y, const double *Z, const double *F,
                                          const double *bZ, double *bt, double *bc, double *bU, double *bV, double *bY,
                                          c2_stream_t stream);
template <bool LOWER, bool SOLVE>
static int launch_sweep_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                            const double *c, int64_t c_bs, const double *U, const double *V, const double *Y,
                            const double *Z, const double *F, const double *bZ, double *bt, double *bc, double *bU,
                            double *bV, double *bY, c2_stream_t stream) {
  if (int e = check_dims(B, N, J)) return e;
  if (nrhs < 1 || !t || !c || !U || !V || !Y || !Z || !F || !bZ || !bt || !bc || !bU || !bV || !bY)
    return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH)
    return c2_wide_sweep_rev(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV,
                             bY, stream);
  hipStream_t s = (hipStream_t)stream;
  {   // a small batch of LONG series: the opposite sweep (parallel along time for these shapes) + a pass local to the rows
    const bool rl_set = opt::has(opt::k_rev_long);   // 0: keep the row-by-row kernels (A/B runs); 1 forces the form
    const bool rl_off = rl_set && opt::ival(opt::k_rev_long) == 0, rl_on = rl_set && opt::ival(opt::k_rev_long) == 1;
    // the per-row pass costs ~0.5 ns per row and series, the row-by-row kernel ~0.17 us per row whatever the batch:
    // 64 x 1024 0.18 -> 0.07 ms, 256 x 4096 0.87 -> 0.58, 512 x 1024 0.20 -> 0.33 (not taken)
    // (whatever form the opposite sweep takes below 16384 rows: row by row it still costs a third of the reverse kernel --
    // one series of 1024 rows, 8 right-hand sides: 1.01 -> 0.36 ms)
    // the row-by-row kernels cost the same per row whatever the batch (until it fills the chip), this form costs per row
    // AND series: 512 x 4096 solve 0.88 -> 0.47 ms (product 0.78 -> 0.91: not taken), 8 right-hand sides 5.4 -> 2.8;
    // 1024 x 4096 with 8: 5.5 -> 3.8 (with one 0.92 -> 1.25: not taken); 2048 x 1024: level at best, not taken
    const int64_t bmax = nrhs >= 4 ? 1024 : (SOLVE ? 512 : 128);
    const bool small = N >= (SOLVE ? long_min_rows() : 1024) && B <= bmax && (N >= 2048 || N >= 8 * B);
    const bool fits = SOLVE ? (solve_chunks_shape(B, N, J, nrhs) && solve_chunks_enabled())
                            : matmul_chunked_shape(B, N, J, nrhs);
    const bool on = !rl_off && B <= 0xffff && (rl_on || (N >= 16384 ? fits : small));
    if (on) {
      const int e = c2_internal_sweep_rev_long(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F,
                                               bZ, bt, bc, bU, bV, bY, stream);
      if (e != C2_ERR_UNSUPPORTED) return e;
    }
  }
  if (nrhs == 1)  // a vector: the tuned single-rhs kernel (c2_sweep.hip)
    return c2_internal_sweep1_rev(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU,
                                  bV, bY, stream);
  if (nrhs <= 7) {   // two to seven: lanes over J, per-series scalars transposed in time (c2_sweep.hip)
    const int e = c2_internal_sweepT_rev(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt,
                                         bc, bU, bV, bY, stream);
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  if (nrhs >= 9 && nrhs <= 32 && J == 8) {
    // nine to 32 right-hand sides at J = 8 on whole wavefronts of eight series: eight lanes per series, two to four columns per lane
    // (c2_sweep_cols.hip); the B % 8 series left over on the kernels below
    int64_t B8 = 0;
    const int e = c2_internal_sweep_cols_rev(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ,
                                             bt, bc, bU, bV, bY, &B8, stream);
    if (e != C2_ERR_UNSUPPORTED) {
      if (e != C2_OK || B8 == B) return e;
      const int64_t o = B8 * N;
      t += B8 * t_bs; c += B8 * c_bs;
      U += o * J; V += o * J; Y += o * nrhs; Z += o * nrhs; F += o * J * nrhs; bZ += o * nrhs;
      bt += o; bc += B8 * J; bU += o * J; bV += o * J; bY += o * nrhs;
      B -= B8;
    }
  }
  {  // several right-hand sides: lanes over them (c2_sweep_rev.hip) where the shape fits; C2_SWEEPK_REV=0 for A/B runs
    if (!(opt::has(opt::k_sweepk_rev) && opt::ival(opt::k_sweepk_rev) == 0)) {
      const int e = c2_internal_sweepK_rev(LOWER ? 1 : 0, SOLVE ? 1 : 0, B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ,
                                           bt, bc, bU, bV, bY, stream);
      if (e != C2_ERR_UNSUPPORTED) return e;
    }
  }
  C2_DISPATCH_G(group_size(J),
                hipLaunchKernelGGL((k_sweep_rev<G, 4, LOWER, SOLVE>), grid_for(B, G), dim3(kWave), 0, s, B, N, (int)J,
                                   nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY));
  return check_launch();
}

// kappa[b] = max_n a_n / d_n of a factored slice (one workgroup per series); a failed factorisation (flag != 0): +inf.
__global__ __launch_bounds__(256) void k_condition(int64_t N, const double *__restrict__ a, const double *__restrict__ d,
                                                   const int32_t *__restrict__ flag, double *__restrict__ kappa) {
  __shared__ double red[4];
  const int64_t b = blockIdx.x;
  double m = 0.0;
  for (int64_t n = threadIdx.x; n < N; n += 256) {
    const double r = a[b * N + n] / d[b * N + n];
    m = r > m ? r : m;
  }
  for (int o = 32; o >= 1; o >>= 1) {
    const double v = __shfl_xor(m, o, kWave);
    m = v > m ? v : m;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) m = red[w] > m ? red[w] : m;
    kappa[b] = flag[b] != 0 ? INFINITY : m;
  }
}

extern "C" {

const char *c2_version(void) { return "celerite2_amd 0.1.0 (gfx950)"; }
const char *c2_last_error(void) { return g_err; }
void c2_internal_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
int c2_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int c2_internal_factor_fused(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                             int64_t c_bs, const double *a, const double *U, const double *V, double *d, double *W,
                             int32_t *flag, int allow_timepar, c2_stream_t stream);

extern "C" int c2_internal_factor_states_timepar(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                                 const double *c, int64_t c_bs, const double *a, const double *U,
                                                 const double *V, double *d, double *W, double *S, int32_t *flag,
                                                 c2_stream_t stream);
int c2_factor(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
              const double *a, const double *U, const double *V, double *d, double *W, double *S, int32_t *flag,
              c2_stream_t stream) {
  if (int e = check_dims(B, N, J)) return e;
  if (!t || !c || !a || !U || !V || !d || !W || !flag) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH) return c2_wide_factor(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, S, flag, stream);
  if (!S)  // no workspace requested: the tuned forward kernel of the fused log-likelihood doubles as factor
    return c2_internal_factor_fused(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, /*time-parallel allowed*/ 1, stream);
  hipStream_t s = (hipStream_t)stream;
  {   // a small batch of long series: Newton iterations for d, W, the S rows by chunks
    const int e = c2_internal_factor_states_timepar(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, S, flag, stream);
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  if ((J == 2 || J == 4 || J == 8 || J == 16) && ((uintptr_t)S) % 16 == 0) {
    // d, W, flag from the tuned fused kernel, then the S rows by a chain-free replay (store-bound)
    // (with S the row-by-row kernel: the reverse-mode chain that asks for S is checked element by element at 1e-12)
    if (int e = c2_internal_factor_fused(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, 0, stream)) return e;
    C2_DISPATCH_G(group_size(J), {
      if constexpr (G == 8) {
        if (((uintptr_t)W) % 16 == 0 && !(opt::has(opt::k_s_replay_lines) && opt::ival(opt::k_s_replay_lines) == 0))
          hipLaunchKernelGGL((k_s_replay<G, true>), grid_for(B, G), dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs,
                             (const double *)d, (const double *)W, (const int32_t *)flag, S);
        else
          hipLaunchKernelGGL((k_s_replay<G, false>), grid_for(B, G), dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs,
                             (const double *)d, (const double *)W, (const int32_t *)flag, S);
      } else if constexpr (G >= 2 && G <= 16)
        hipLaunchKernelGGL((k_s_replay<G, false>), grid_for(B, G), dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs,
                           (const double *)d, (const double *)W, (const int32_t *)flag, S);
    });
    return check_launch();
  }
  C2_DISPATCH_G(group_size(J), hipLaunchKernelGGL(k_factor<G>, grid_for(B, G), dim3(kWave), 0, s, B, N, (int)J, t,
                                                  t_bs, c, c_bs, a, U, V, d, W, S, flag));
  return check_launch();
}

int c2_solve_lower(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                   int64_t c_bs, const double *U, const double *W, const double *Y, double *Z, double *F,
                   c2_stream_t stream) {
  return launch_sweep<true, true>(B, N, J, nrhs, t, t_bs, c, c_bs, U, W, Y, Z, F, 0, stream);
}
int c2_solve_upper(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                   int64_t c_bs, const double *U, const double *W, const double *Y, double *Z, double *F,
                   c2_stream_t stream) {
  return launch_sweep<false, true>(B, N, J, nrhs, t, t_bs, c, c_bs, U, W, Y, Z, F, 0, stream);
}
int c2_matmul_lower(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                    int64_t c_bs, const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z,
                    c2_stream_t stream) {
  return launch_sweep<true, false>(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
}
int c2_matmul_upper(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                    int64_t c_bs, const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z,
                    c2_stream_t stream) {
  return launch_sweep<false, false>(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
}

int c2_general_matmul_lower(int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, int64_t t1_bs,
                            const double *t2, int64_t t2_bs, const double *c, int64_t c_bs, const double *U,
                            const double *V, const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream) {
  return launch_general<true>(B, N, M, J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
}
int c2_general_matmul_upper(int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, int64_t t1_bs,
                            const double *t2, int64_t t2_bs, const double *c, int64_t c_bs, const double *U,
                            const double *V, const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream) {
  return launch_general<false>(B, N, M, J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F, zero_z, stream);
}

// Internal: factor_rev with optional accumulation into bt/bc/bU (used by c2_loglik_grad).
int c2_internal_factor_rev_replay(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                  int64_t c_bs, const double *U, const double *d, const double *W, const double *S,
                                  const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                                  double *bV, const unsigned long long *gate, c2_stream_t stream);

int c2_factor_rev_acc(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                      const double *U, const double *d, const double *W, const double *S, const double *bd,
                      const double *bW, double *bt, double *bc, double *ba, double *bU, double *bV, int accumulate,
                      c2_stream_t stream) {
  if (int e = check_dims(B, N, J)) return e;
  if (!t || !c || !U || !d || !W || !S || !bd || !bW || !bt || !bc || !ba || !bU || !bV) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH)
    return c2_wide_factor_rev(B, N, J, t, t_bs, c, c_bs, U, d, W, S, bd, bW, bt, bc, ba, bU, bV, accumulate, stream);
  hipStream_t s = (hipStream_t)stream;
  C2_DISPATCH_G(group_size(J), hipLaunchKernelGGL(k_factor_rev<G>, grid_for(B, G), dim3(kWave), 0, s, B, N, (int)J, t,
                                                  t_bs, c, c_bs, U, d, W, S, bd, bW, bt, bc, ba, bU, bV, accumulate));
  return check_launch();
}

extern "C" int c2_internal_factor_rev_long(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                           int64_t c_bs, const double *U, const double *d, const double *W, const double *S,
                                           const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                                           double *bV, c2_stream_t stream);
int c2_factor_rev(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                  const double *a, const double *U, const double *V, const double *d, const double *W,
                  const double *S, const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                  double *bV, c2_stream_t stream) {
  (void)a; (void)V;  // unused by the reference as well (reverse.hpp:29-30)
  if (int e = check_dims(B, N, J)) return e;
  if (!t || !c || !U || !d || !W || !S || !bd || !bW || !bt || !bc || !ba || !bU || !bV) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH)
    return c2_wide_factor_rev(B, N, J, t, t_bs, c, c_bs, U, d, W, S, bd, bW, bt, bc, ba, bU, bV, 0, stream);
  {   // a small batch of long series: parallel along time
    const int e = c2_internal_factor_rev_long(B, N, J, t, t_bs, c, c_bs, U, d, W, S, bd, bW, bt, bc, ba, bU, bV, stream);
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  // the segment-replay kernel of the fused gradient, with the caller's S rows as checkpoints (c2_loglik.hip)
  return c2_internal_factor_rev_replay(B, N, J, t, t_bs, c, c_bs, U, d, W, S, bd, bW, bt, bc, ba, bU, bV, nullptr, stream);
}

int c2_solve_lower_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                       int64_t c_bs, const double *U, const double *W, const double *Y, const double *Z,
                       const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bW, double *bY,
                       c2_stream_t stream) {
  return launch_sweep_rev<true, true>(B, N, J, nrhs, t, t_bs, c, c_bs, U, W, Y, Z, F, bZ, bt, bc, bU, bW, bY, stream);
}
int c2_solve_upper_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                       int64_t c_bs, const double *U, const double *W, const double *Y, const double *Z,
                       const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bW, double *bY,
                       c2_stream_t stream) {
  return launch_sweep_rev<false, true>(B, N, J, nrhs, t, t_bs, c, c_bs, U, W, Y, Z, F, bZ, bt, bc, bU, bW, bY, stream);
}
int c2_matmul_lower_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                        int64_t c_bs, const double *U, const double *V, const double *Y, const double *Z,
                        const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bV, double *bY,
                        c2_stream_t stream) {
  return launch_sweep_rev<true, false>(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY, stream);
}
int c2_matmul_upper_rev(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                        int64_t c_bs, const double *U, const double *V, const double *Y, const double *Z,
                        const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bV, double *bY,
                        c2_stream_t stream) {
  return launch_sweep_rev<false, false>(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY,
                                        stream);
}

// `gate`: see gate_closed (c2_loglik_helpers.hpp); nullptr from the public entry point.
int c2_internal_matrices(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *ac,
                         const double *bc, const double *dc, int coef_batched, const double *x, int64_t x_bs,
                         const double *diag, double *a, double *U, double *V, const unsigned long long *gate,
                         c2_stream_t stream) {
  if (B < 1 || N < 1 || Jr < 0 || Jc < 0 || Jr + 2 * Jc < 1) return C2_ERR_INVALID;
  if (!x || !diag || !a || !U || !V || (Jr && !ar) || (Jc && (!ac || !bc || !dc))) return C2_ERR_INVALID;
  const int64_t rows_per_block = (int64_t)(256 / (Jr + Jc)) * kMatRows;
  // a gated launch is the fallback of a fused path: most (normally all) of its groups of 64 series are closed to it, so it comes
  // with a small grid whose blocks stride over the series (and return at once when no group fell back) instead of one block per
  // series and row block that looks at the gate and leaves (65536 series: 0.44 + 0.26 ms of empty blocks per call)
  const int64_t by = gate ? (B < 512 ? B : 512) : (B < 65535 ? B : 65535);
  hipLaunchKernelGGL(k_matrices, dim3((unsigned)((N + rows_per_block - 1) / rows_per_block), (unsigned)by), dim3(256), 0, (hipStream_t)stream, B, N,
                     (int)Jr, (int)Jc, ar, ac, bc, dc, coef_batched, x, x_bs, diag, a, U, V, gate);
  if (int e = check_launch()) return e;
  if (Jc > 0) {
    int64_t nb = (B * N + 255) / 256;
    if (gate && nb > 4096) nb = 4096;
    bool stride = gate != nullptr;
    if (nb > 0x7fffffffLL) { nb = 0x7fffffffLL; stride = true; }   // (more rows than a grid has threads: the striding form)
    if (stride)
      hipLaunchKernelGGL(k_matrices_big<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, B, N, (int)Jr, (int)Jc, ac, bc,
                         dc, coef_batched, x, x_bs, U, V, gate);
    else
      hipLaunchKernelGGL(k_matrices_big<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, B, N, (int)Jr, (int)Jc, ac, bc,
                         dc, coef_batched, x, x_bs, U, V, gate);
  }
  return check_launch();
}
int c2_get_celerite_matrices(int64_t B, int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *ac,
                             const double *bc, const double *dc, int coef_batched, const double *x, int64_t x_bs,
                             const double *diag, double *a, double *U, double *V, c2_stream_t stream) {
  return c2_internal_matrices(B, N, Jr, Jc, ar, ac, bc, dc, coef_batched, x, x_bs, diag, a, U, V, nullptr, stream);
}

int c2_kernel_values(int64_t B, int64_t N, int64_t M, int64_t Jr, int64_t Jc, const double *ar, const double *cr,
                     const double *ac, const double *bc, const double *cc, const double *dc, int coef_batched,
                     const double *t1, int64_t t1_bs, const double *t2, int64_t t2_bs, double *K, c2_stream_t stream) {
  if (B < 1 || N < 1 || M < 1 || Jr < 0 || Jc < 0 || Jr + Jc < 1) return C2_ERR_INVALID;
  if (!t1 || !t2 || !K || (Jr && (!ar || !cr)) || (Jc && (!ac || !bc || !cc || !dc))) return C2_ERR_INVALID;
  const int64_t total = B * N * M;
  const size_t lds = sizeof(double) * kv_lds_doubles(Jr, Jc);
  // tiles of 32 x 64 entries (one sincos per tile row and column instead of one per entry) when there is enough of a grid
  // to amortise them and the row phases fit LDS; C2_KERNEL_VALUES_TILE=0: the thread-per-entry kernel
  if (N >= 8 && M >= 8 && B <= 65535 && (N + kKvRows - 1) / kKvRows <= 65535 && lds <= 48 * 1024 &&
      !(opt::has(opt::k_kernel_values_tile) && opt::ival(opt::k_kernel_values_tile) == 0)) {
    const dim3 grid((unsigned)((M + kKvCols - 1) / kKvCols), (unsigned)((N + kKvRows - 1) / kKvRows), (unsigned)B);
    hipLaunchKernelGGL(k_kernel_values_tile, grid, dim3(256), lds, (hipStream_t)stream, N, M, (int)Jr, (int)Jc, ar, cr, ac, bc, cc,
                       dc, coef_batched, t1, t1_bs, t2, t2_bs, K);
    return check_launch();
  }
  if ((total + 255) / 256 > 0x7fffffffLL) return C2_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_kernel_values, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, N, M,
                     (int)Jr, (int)Jc, ar, cr, ac, bc, cc, dc, coef_batched, t1, t1_bs, t2, t2_bs, K);
  return check_launch();
}

int c2_colsumsq_over_d(int64_t B, int64_t N, int64_t M, const double *Z, const double *d, double *out, c2_stream_t stream) {
  if (B < 1 || N < 1 || M < 1 || !Z || !d || !out) return C2_ERR_INVALID;
  if (B > 65535) return C2_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_colsumsq_over_d, dim3((unsigned)((M + 63) / 64), (unsigned)B), dim3(1024), 0, (hipStream_t)stream, N, M, Z, d,
                     out);
  return check_launch();
}

int c2_dot_tril(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs, const double *c,
                int64_t c_bs, const double *U, const double *W, const double *d, const double *Y, double *Z,
                c2_stream_t stream) {
  if (int e = check_dims(B, N, J)) return e;
  if (nrhs < 1 || !t || !c || !U || !W || !d || !Y || !Z) return C2_ERR_INVALID;
  // long series, J = 16: scaling, unit-lower product and accumulation in one matrix-core pass (c2_mfma.hip)
  if (N >= 16384 && B * ((nrhs + 3) / 4) * group_size(J) < (int64_t)kWave * 2048 && use_mfma()) {
    const int e = c2_internal_matmul_lower_mfma(B, N, J, nrhs, t, t_bs, c, c_bs, U, W, d, Y, Z, 0, stream);
    if (e != C2_ERR_UNSUPPORTED) return e;
  }
  const int64_t total = B * N * nrhs;
  hipLaunchKernelGGL(k_scale_sqrt, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total,
                     nrhs, d, Y, Z);
  if (int e = check_launch()) return e;
  return launch_sweep<true, false>(B, N, J, nrhs, t, t_bs, c, c_bs, U, W, Z, Z, nullptr, 0, stream);
}

// kappa = max_n a_n / d_n per series (include/celerite2_amd.h): `factor` on slices of the batch into library temporaries
// (d, W of at most 4096 series at a time), then one reduction per series.
int c2_condition(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                 const double *a, const double *U, const double *V, double *kappa, int32_t *flag, c2_stream_t stream) {
  if (int e = check_dims(B, N, J)) return e;
  if (!t || !c || !a || !U || !V || !kappa || !flag) return C2_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  const int64_t slice = B < 4096 ? B : 4096;
  double *tmp = nullptr;
  if (temp_alloc((void **)&tmp, sizeof(double) * (size_t)slice * (size_t)N * (size_t)(J + 1), s) != hipSuccess) {
    (void)hipGetLastError();
    return C2_ERR_HIP;
  }
  double *d = tmp, *W = tmp + (size_t)slice * (size_t)N;
  int rc = C2_OK;
  for (int64_t b0 = 0; b0 < B && rc == C2_OK; b0 += slice) {
    const int64_t nb = B - b0 < slice ? B - b0 : slice;
    rc = c2_factor(nb, N, J, t + b0 * t_bs, t_bs, c + b0 * c_bs, c_bs, a + b0 * N, U + b0 * N * J, V + b0 * N * J, d, W,
                   nullptr, flag + b0, stream);
    if (rc != C2_OK) break;
    hipLaunchKernelGGL(k_condition, dim3((unsigned)nb), dim3(256), 0, s, N, a + b0 * N, (const double *)d,
                       (const int32_t *)(flag + b0), kappa + b0);
    rc = check_launch();
  }
  const int rf = hip_check(hipFreeAsync(tmp, s));
  return rc == C2_OK ? rf : rc;
}

}  // extern "C"
