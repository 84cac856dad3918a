// c2_timepar_grad.hip -- the fused log-likelihood GRADIENT parallel along time, for small batches of long series
// (widths 1 .. 8).  Row by row such a batch is pure latency: ~1.2 us per row for the forward + reverse pair with a
// handful of wavefronts on the chip.
//
// Forward quantities.  d, W (c2_factor; forward.hpp:105-134) and z = L^-1 y (c2_solve_lower; internal.hpp:135-145) of
// every row come from the existing kernels -- themselves time-parallel where they can be (c2_timepar.hip).  With d, W, z
// known the states entering row n obey LINEAR recurrences with diagonal transitions,
//     S_{n+1} = P (S_n + d_n w_n^T w_n) P ,   F_{n+1} = P (F_n + w_n z_n) ,   P = diag exp(-c (t_{n+1} - t_n)),
// so chunks of 64 rows sum their own contributions in parallel (k_local) and a scan over the chunks (k_starts) gives the
// state every chunk starts from.
//
// Reverse sweep (the fused step of c2_loglik.hip / reverse.hpp:52-84 + internal.hpp:225-245).  The adjoints (bS, bF) of
// the states obey, with A_n = P (I - w_n u_n^T),
//     bF_n = A_n^T bF_{n+1} + (z_n / d_n) u_n^T ,
//     bS_n = A_n^T bS_{n+1} A_n - (z_n / d_n) sym(bF_n u_n^T) + (1/2)(1/d_n - z_n^2/d_n^2) u_n^T u_n ,
// i.e. a chunk acts on the adjoint it receives at its end as an AFFINE map
//     bF_start = Phi^T bF_end + gF ,   bS_start = Phi^T bS_end Phi + sum_k bF_end[k] C_k + gS .
// k_maps finds (Phi, C_k, gS, gF) of every chunk by J + 1 sweeps over its rows (bF_end = e_k with bS_end = 0; zero end
// adjoint with the sources), one lane per chunk and sweep; k_chain walks the chunks of a series backwards applying the maps
// (lanes <-> the J x J entries); k_final gives every chunk its true end adjoint and runs the one sweep that writes the
// gradients.  That sweep needs the states S_n, F_n entering each row in reverse order: the chunk replays them forward
// into a lane-major scratch record first (no backward recursion, hence no stability condition).
//
// git show cd74ef8:tools/proto/tpg.py is the same construction in numpy, checked against the sequential oracle.
//
// Also here, because they share the chunk machinery:
//  * `factor` by NEWTON iterations on the chunk start states (k_newton_*; prototype: git show cd74ef8:tools/proto/factor_newton.py) -- what the
//    forward quantities above come from at every width but 4 / 2, and on long series there as well;
//  * z = L^-1 y by affine chunk maps at any even width (k_solve_*), and the forward-only log-likelihood composed from
//    the two (c2_internal_loglik_wide);
//  * every chain over the chunks in TWO LEVELS from kTwoLevelMin chunks per series: blocks of kBlock chunks compose
//    their prefix maps in parallel, one wavefront per series walks the blocks, the prefix maps are applied in parallel
//    (k_newton_block / _blocks / _apply, k_solve_block / _blocks / _starts, k_starts_blocks / _apply,
//    adjoint_chain_two_level).
#include <cstdint>
#include <cstdlib>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

extern "C" int c2_internal_factor_fused_ws(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                           int64_t c_bs, const double *a, const double *U, const double *V, double *d,
                                           double *W, int32_t *flag, int allow_timepar, double *scratch,
                                           c2_stream_t stream);
extern "C" size_t c2_internal_factor_scratch_doubles(int64_t B, int64_t N, int64_t J);
extern "C" double *c2_internal_get_debug_sink();   // diagnostics (c2_loglik.hip)

// 1: the dispatch's choice of factor kernels (widths 4 / 2 below 32768 rows: the composed maps of c2_timepar.hip, verified
// to 5e-11 only -- which the gradient inherits: 1.5e-10 / 6.5e-11 of the largest gradient entry on two WELL-conditioned
// draws of the round-2 stress run, tests/test_gpu_fuzz.py::test_time_parallel_gradient_known_hard_draws); 2 (default
// since round 3): Newton iterations at every width (1e-13 or better, conditioning-aware; 0.30 -> 0.48 ms for one series
// of 4096 rows at J = 2)
#ifndef C2TG_FACTOR_MODE
#define C2TG_FACTOR_MODE 2
#endif

// The file is compiled once per chunk length: C2TG_ROWS = 64 (default) and 32 (c2_timepar_grad32.hip includes it) -- a
// handful of series spreads over twice as many lanes with half the walk per lane, at the price of chains twice as long.
#ifndef C2TG_ROWS
#define C2TG_ROWS 64
#endif
#define C2TG_CAT2(a, b) a##b
#define C2TG_CAT(a, b) C2TG_CAT2(a, b)
#define C2TG_NAME(stem) C2TG_CAT(stem, C2TG_ROWS)   // c2_internal_loglik_grad_timepar -> ..._timepar64
#define c2tg C2TG_CAT(c2tg_r, C2TG_ROWS)            // one namespace per chunk length

namespace c2tg {
using namespace c2;

constexpr int kRows = C2TG_ROWS;   // rows per chunk
constexpr int64_t kTwoLevelMin = 128;   // chunks per series from which the sequential chains over the chunks run in two levels
constexpr int kBlock = 32;              // chunks per block of the upper level

template <int J>
struct Dim {
  static constexpr int NS = J * (J + 1) / 2;   // packed symmetric J x J
  static constexpr int NST = NS + J;           // a state: S (packed), F
  static constexpr int MAPR = (J + 1) * NST;   // a chunk map: sweeps e_0 .. e_{J-1} (C_k, Phi^T e_k), then (gS, gF)
};
__host__ __device__ constexpr int sidx(int J, int i, int j) { return i * J - i * (i - 1) / 2 + (j - i); }
__host__ __device__ constexpr int sym(int J, int i, int j) { return i <= j ? sidx(J, i, j) : sidx(J, j, i); }

struct Geo {   // lane <-> chunk
  int64_t g, b, k, lo;
  int len;       // rows of the chunk (0: no chunk behind this lane)
  int64_t wave;
  int lane;
};
__device__ __forceinline__ Geo chunk_of(int64_t B, int64_t N, int64_t K) {
  Geo G;
  G.lane = threadIdx.x;
  G.wave = blockIdx.x;
  G.g = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const bool valid = G.g < B * K;
  const int64_t gc = valid ? G.g : B * K - 1;
  G.b = gc / K;
  G.k = gc - G.b * K;
  G.lo = G.k * kRows;
  const int64_t left = N - G.lo;
  G.len = valid ? (int)(left < kRows ? left : kRows) : 0;
  return G;
}

template <int J>
__device__ __forceinline__ void load_row(const double *p, double (&x)[J]) {
  if constexpr (J % 2 == 0) {   // rows of an even width are 16-byte aligned runs
#pragma unroll
    for (int j = 0; j < J; j += 2) {
      const double2 v = *reinterpret_cast<const double2 *>(p + j);
      x[j] = v.x; x[j + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < J; ++j) x[j] = p[j];
  }
}
template <int J>
__device__ __forceinline__ void store_row(double *p, const double (&x)[J]) {
  if constexpr (J % 2 == 0) {
#pragma unroll
    for (int j = 0; j < J; j += 2) *reinterpret_cast<double2 *>(p + j) = make_double2(x[j], x[j + 1]);
  } else {
#pragma unroll
    for (int j = 0; j < J; ++j) p[j] = x[j];
  }
}

// The inputs of one row of a chunk, fetched ONE ITERATION AHEAD of their use (a chunk is walked by a single lane: with
// one wavefront per SIMD nothing else hides the latency of the loads).  r is clamped into the chunk, so lanes with
// shorter chunks fetch valid rows they do not use.
template <int J, bool WITH_U, bool WITH_V>
struct RowIn {
  double u[WITH_U ? J : 1], v[WITH_V ? J : 1], w[J], d, z, dt;   // dt: gap to the next row (0 behind the last row)
};
// REV: the chunks walk the series from its far end (position s <-> row N-1-s; solve_upper)
template <int J, bool WITH_U, bool WITH_V, bool REV = false>
__device__ __forceinline__ void fetch_row(RowIn<J, WITH_U, WITH_V> &R, const Geo &G, int64_t N, int r, const double *tb,
                                          const double *Ub, const double *Vb, const double *Wb, const double *db,
                                          const double *zb, int64_t zs = 1) {   // zs: stride of d / z (a column of Y)
  const int rc = r < G.len ? (r < 0 ? 0 : r) : (G.len > 0 ? G.len - 1 : 0);
  const int64_t pos = G.lo + rc, n = REV ? N - 1 - pos : pos;
  if constexpr (WITH_U) load_row<J>(Ub + n * J, R.u);
  if constexpr (WITH_V) load_row<J>(Vb + n * J, R.v);
  load_row<J>(Wb + n * J, R.w);
  R.d = db[n * zs]; R.z = zb[n * zs];
  const int64_t nn = REV ? (n > 0 ? n - 1 : n) : (n + 1 < N ? n + 1 : n);   // the next row of the walk
  const double t0 = tb[n], t1 = tb[nn];
  R.dt = REV ? t0 - t1 : t1 - t0;
}

// state + its own row:  S += d w^T w,  F += w z
template <int J>
__device__ __forceinline__ void absorb(double (&S)[Dim<J>::NS], double (&F)[J], const double (&w)[J], double d, double z) {
#pragma unroll
  for (int i = 0; i < J; ++i) {
    const double dw = d * w[i];
#pragma unroll
    for (int j = i; j < J; ++j) S[sidx(J, i, j)] = fma(dw, w[j], S[sidx(J, i, j)]);
    F[i] = fma(w[i], z, F[i]);
  }
}
template <int J>
__device__ __forceinline__ void decay(double (&S)[Dim<J>::NS], double (&F)[J], const double (&p)[J]) {
#pragma unroll
  for (int i = 0; i < J; ++i) {
#pragma unroll
    for (int j = i; j < J; ++j) S[sidx(J, i, j)] *= p[i] * p[j];
    F[i] *= p[i];
  }
}

// ---- forward: contributions of the chunks, start states ---------------------------------------------------------------
// loc[g]: state entering the first row of chunk g+1 if chunk g had started from zero; llp[g]: sum (log d + z^2 / d)
template <int J>
__global__ __launch_bounds__(kWave) void k_local(int64_t B, int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                 const double *__restrict__ c, int64_t c_bs, const double *__restrict__ d,
                                                 const double *__restrict__ W, const double *__restrict__ z,
                                                 double *__restrict__ loc, double *__restrict__ llp) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST;
  const Geo G = chunk_of(B, N, K);
  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[G.b * c_bs + j];
  const double *tb = t + G.b * t_bs, *db = d + G.b * N, *zb = z + G.b * N, *Wb = W + G.b * N * J;
  double S[NS], F[J];
#pragma unroll
  for (int e = 0; e < NS; ++e) S[e] = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j) F[j] = 0.0;
  double acc = 0.0;
  RowIn<J, false, false> cur, nxt;
  fetch_row<J, false, false>(cur, G, N, 0, tb, nullptr, nullptr, Wb, db, zb);
#pragma unroll 1
  for (int r = 0; r < kRows; ++r) {
    fetch_row<J, false, false>(nxt, G, N, r + 1, tb, nullptr, nullptr, Wb, db, zb);
    if (r < G.len) {
      double p[J];
      acc += log(cur.d) + cur.z * cur.z / cur.d;
      absorb<J>(S, F, cur.w, cur.d, cur.z);
      if (G.lo + r + 1 < N) {
#pragma unroll
        for (int j = 0; j < J; ++j) p[j] = exp_decay(-cj[j] * cur.dt);
        decay<J>(S, F, p);
      }
    }
    cur = nxt;
  }
  if (G.len > 0) {
    double *o = loc + G.g * NST;
#pragma unroll
    for (int e = 0; e < NS; ++e) o[e] = S[e];
#pragma unroll
    for (int j = 0; j < J; ++j) o[NS + j] = F[j];
    llp[G.g] = acc;
  }
}

// One wavefront per series, lane <-> entry of the state: start[g] = state entering the first row of chunk g.  Also the
// log-likelihood (numpy.py:66-109): -1/2 sum (log d + z^2/d) - N/2 log 2 pi, -inf for a failed factorisation.
template <int J>
__global__ __launch_bounds__(kWave) void k_starts(int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *__restrict__ loc,
                                                  const double *__restrict__ llp, const int32_t *__restrict__ flag,
                                                  double *__restrict__ start, double *__restrict__ ll) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const double *tb = t + b * t_bs;
  // rate of the lane's entry: c_i + c_j for S(i, j), c_i for F_i
  double rate = 0.0;
  if (lane < NST) {
    if (lane >= NS) rate = c[b * c_bs + (lane - NS)];
    else {
      int i = 0, rem = lane;
      while (rem >= J - i) { rem -= J - i; ++i; }
      rate = c[b * c_bs + i] + c[b * c_bs + i + rem];
    }
  }
  double cur = 0.0;
  constexpr int PF = 8;   // chunks whose inputs are in flight together (the recurrence itself is one fma per chunk)
  for (int64_t k0 = 0; k0 < K; k0 += PF) {
    double lv[PF], Tv[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int64_t k = k0 + q < K ? k0 + q : K - 1;
      lv[q] = lane < NST ? loc[(b * K + k) * NST + lane] : 0.0;
      Tv[q] = k + 1 < K ? tb[(k + 1) * kRows] - tb[k * kRows] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      if (k0 + q < K) {
        if (lane < NST) start[(b * K + k0 + q) * NST + lane] = cur;
        // loc is already decayed to the next chunk's first row; the carried state decays over the whole chunk
        cur = fma(exp_decay(-rate * Tv[q]), cur, lv[q]);
      }
    }
  }
  double s = 0.0;
  for (int64_t k = lane; k < K; k += kWave) s += llp[b * K + k];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
  if (lane == 0) ll[b] = flag[b] != 0 ? -__builtin_huge_val() : -0.5 * s - 0.5 * (double)N * kLog2Pi;
}

// The same scan for long series (K >= kTwoLevelMin chunks), lane <-> chunk: the recurrence start_{k+1} = D_k o start_k +
// loc_k is elementwise affine, so 64 chunks are scanned across the lanes ((D_a, f_a) o (D_b, f_b) = (D_a D_b, D_b f_a +
// f_b)), every block of 64 chunks from zero (k_starts_blocks: also the block's total), and a second kernel gives each
// block the state it starts from -- the totals of the blocks before it, combined in order -- and applies it
// (k_starts_apply).  pre: [series][chunk][2 NST] = inclusive (D, f) of the chunk inside its block.
template <int J>
__global__ __launch_bounds__(kWave) void k_starts_blocks(int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                         const double *__restrict__ c, int64_t c_bs,
                                                         const double *__restrict__ loc, double *__restrict__ pre) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST;
  const int lane = threadIdx.x;
  const int64_t NB = (K + kWave - 1) / kWave;
  const int64_t b = blockIdx.x / NB, nb = blockIdx.x % NB;
  const int64_t k = nb * kWave + lane;
  const bool valid = k < K;
  const int64_t kc = valid ? k : K - 1;
  const double *tb = t + b * t_bs;
  // decay of the carried state over chunk k (identity behind the last chunk), contribution of chunk k
  const double T = (valid && k + 1 < K) ? tb[(k + 1) * kRows] - tb[k * kRows] : 0.0;
  double e[J], D[NST], f[NST];
#pragma unroll
  for (int j = 0; j < J; ++j) e[j] = exp_decay(-c[b * c_bs + j] * T);
#pragma unroll
  for (int i = 0; i < J; ++i) {
#pragma unroll
    for (int j = i; j < J; ++j) D[sidx(J, i, j)] = e[i] * e[j];
    D[NS + i] = e[i];
  }
  const double *l = loc + (b * K + kc) * NST;
#pragma unroll
  for (int q = 0; q < NST; ++q) f[q] = valid ? l[q] : 0.0;   // (NST is odd at J = 2, 6: no 16-byte pieces)
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const bool has = lane >= d;
#pragma unroll
    for (int q = 0; q < NST; ++q) {
      const double Dp = __shfl_up(D[q], d, kWave), fp = __shfl_up(f[q], d, kWave);
      if (has) { f[q] = fma(D[q], fp, f[q]); D[q] *= Dp; }
    }
  }
  if (valid) {
    double *o = pre + (b * K + k) * (2 * NST);
#pragma unroll
    for (int q = 0; q < NST; ++q) { o[q] = D[q]; o[NST + q] = f[q]; }
  }
}
template <int J>
__global__ __launch_bounds__(kWave) void k_starts_apply(int64_t N, int64_t K, const double *__restrict__ pre,
                                                        const double *__restrict__ llp,
                                                        const int32_t *__restrict__ flag, double *__restrict__ start,
                                                        double *__restrict__ ll) {
  constexpr int NST = Dim<J>::NST;
  static_assert(NST <= kWave, "an entry per lane");
  const int lane = threadIdx.x;
  const int64_t NB = (K + kWave - 1) / kWave;
  const int64_t b = blockIdx.x / NB, nb = blockIdx.x % NB;
  // state entering this block: the totals (last chunk) of the blocks before it, in order; lane <-> entry
  double carry = 0.0;
  if (lane < NST) {
    for (int64_t q = 0; q < nb; ++q) {
      const double *o = pre + (b * K + q * kWave + kWave - 1) * (2 * NST);
      carry = fma(o[lane], carry, o[NST + lane]);
    }
  }
  // start of chunk k0 + r: carry for r = 0, else D_{r-1} carry + f_{r-1}
  const int64_t k0 = nb * kWave;
  const int64_t cnt = K - k0 < kWave ? K - k0 : kWave;
  if (lane < NST) {
    start[(b * K + k0) * NST + lane] = carry;
    for (int64_t r = 1; r < cnt; ++r) {
      const double *o = pre + (b * K + k0 + r - 1) * (2 * NST);
      start[(b * K + k0 + r) * NST + lane] = fma(o[lane], carry, o[NST + lane]);
    }
  }
  if (nb == 0) {
    double s = 0.0;
    for (int64_t k = lane; k < K; k += kWave) s += llp[b * K + k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
    if (lane == 0) ll[b] = flag[b] != 0 ? -__builtin_huge_val() : -0.5 * s - 0.5 * (double)N * kLog2Pi;
  }
}

// ---- verification words ---------------------------------------------------------------------------------------------------
// Nothing above or below is approximated, but every chain over the chunks sums in another order than the row-by-row
// recursion, so -- like the forward pass of c2_timepar.hip -- the result is VERIFIED on the device rather than trusted: the
// sweep that writes the gradients (k_final) walks every chunk with the ordinary recursions from the start state / end adjoint
// the chains gave it, so comparing what a chunk arrives at with what its neighbour was given bounds the distance to the
// sequential recursion (which is these same walks with the neighbours agreeing exactly).  vw[0]: the gate, written by
// k_verify_combine: worst mismatch / (kVerifyTol / 2), i.e. above kBackwardGuard = 2 <=> the row-by-row kernels launched
// behind it run (gate_closed); vw[1]: S at the chunk boundaries, relative to sqrt(S_ii S_jj); vw[2]: the adjoints (bS, bF),
// weighted with the state they pair with; vw[3]: kappa = max a_n / d_n (informational); vw[4]: F at the chunk boundaries in
// units of sqrt(S_jj); vw[5]: the solve's state at the chunk boundaries (k_solve_apply), in units of the terms of z.
constexpr double kVerifyTol = 2e-12;   // (largest mismatch over 8528 stress draws: 9e-14, profiles/r03_timepar_verification.md)
constexpr int kVerifyWords = 8;
constexpr int kNewtonMax = 8;                       // Newton iterations of the factor at most (below)
constexpr int kNewtonHdr = 3 * (kNewtonMax + 2);   // its words[0 .. kNewtonMax + 1]: the iterations' updates in units of half their tolerance; then their kappas; then the updates themselves
__device__ __forceinline__ void publish_max(unsigned long long *w, double v) {   // (every lane of the wavefront calls it)
  v = (v == v) ? v : __builtin_huge_val();   // a NaN opens the gate
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  if (threadIdx.x == 0 && v > 0.0) atomicMax(w, (unsigned long long)__double_as_longlong(v));
}
// cond_limit > 0 (option timepar_cond_limit): a conditioning kappa beyond it opens the gate as well
__global__ void k_verify_combine(unsigned long long *vw, double cond_limit) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double es = __longlong_as_double((long long)vw[1]), eb = __longlong_as_double((long long)vw[2]);
  const double ef = __longlong_as_double((long long)vw[4]), ez = __longlong_as_double((long long)vw[5]);
  const double kap = __longlong_as_double((long long)vw[3]);
  double worst = fmax(fmax(es, eb), fmax(ef, ez)) / (0.5 * kVerifyTol);
  if (cond_limit > 0.0 && kap > cond_limit) worst = fmax(worst, 2.0 * kap / cond_limit + 1.0);
  vw[0] = (unsigned long long)__double_as_longlong(worst);
}

// ---- the adjoint step ---------------------------------------------------------------------------------------------------
// (bS, bF): adjoint of the state entering row n+1 on entry, of the state entering row n on exit.  p: decay between the
// two rows (1 behind the last row).  SRC: with the sources of the log-likelihood (d ll/d z = -z/d, d ll/d d).
struct RowOut { double bz, bd; };
// EXT: external sources instead (factor_rev, reverse.hpp:26-85): bde = adjoint of d_n, bwe = adjoint of w_n handed in.
template <int J, bool SRC, bool OUT, bool EXT = false>
__device__ __forceinline__ RowOut adjoint_row(double (&bS)[Dim<J>::NS], double (&bF)[J], const double (&p)[J],
                                              const double (&u)[J], const double (&w)[J], double d, double z, double rd,
                                              double (&g2)[J], double (&btau)[J], double bde = 0.0,
                                              const double *bwe = nullptr) {
#pragma unroll
  for (int i = 0; i < J; ++i) {
#pragma unroll
    for (int j = i; j < J; ++j) bS[sidx(J, i, j)] *= p[i] * p[j];
    bF[i] *= p[i];
  }
  double g[J];
#pragma unroll
  for (int i = 0; i < J; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) s = fma(bS[sym(J, i, j)], w[j], s);
    g[i] = s;
  }
  double wg = 0.0, wbF = 0.0;
#pragma unroll
  for (int i = 0; i < J; ++i) { wg = fma(w[i], g[i], wg); wbF = fma(w[i], bF[i], wbF); }
  const double zd = z * rd;
  RowOut o;
  o.bz = wbF - (SRC ? zd : 0.0);
  o.bd = -wg - zd * wbF - (SRC ? 0.5 * (rd - zd * zd) : 0.0);
  if constexpr (EXT) {
    double wb = 0.0;
#pragma unroll
    for (int i = 0; i < J; ++i) wb = fma(w[i], bwe[i], wb);
    o.bd += bde - rd * wb;
  }
  double hb[J];   // btau / 2
#pragma unroll
  for (int i = 0; i < J; ++i) {
    double bv = fma(2.0, g[i], zd * bF[i]);          // bV_n = bw / d
    if constexpr (EXT) bv = fma(rd, bwe[i], bv);
    if (OUT) g2[i] = bv;
    btau[i] = -bv - o.bd * u[i];
    hb[i] = 0.5 * btau[i];
    bF[i] = fma(-o.bz, u[i], bF[i]);
  }
#pragma unroll
  for (int i = 0; i < J; ++i) {
#pragma unroll
    for (int j = i; j < J; ++j)
      bS[sidx(J, i, j)] = fma(hb[i], u[j], fma(u[i], hb[j], bS[sidx(J, i, j)]));   // += sym(btau u^T)
  }
  return o;
}

// ---- adjoint maps of the chunks ---------------------------------------------------------------------------------------
// One lane per (chunk, sweep): blockIdx.y is the sweep, so a small batch still spreads over J + 1 times as many
// wavefronts as it has chunks / 64 (one series of 4096 rows: 9 wavefronts instead of 1 walking 9 sweeps in turn).
// map[g][s * NST ...]: result (bS, bF) of sweep s (s < J: from bF_end = e_s; s = J: from zero, with the sources).
// EXT: the sources of sweep J are the adjoints bde (of d), bwe (of W) handed in, and no z couples the two states.
template <int J, bool EXT = false>
__global__ __launch_bounds__(kWave) void k_maps(int64_t B, int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                const double *__restrict__ c, int64_t c_bs, const double *__restrict__ U,
                                                const double *__restrict__ d, const double *__restrict__ W,
                                                const double *__restrict__ z, double *__restrict__ map,
                                                const double *__restrict__ bde = nullptr,
                                                const double *__restrict__ bwe = nullptr) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST, MAPR = Dim<J>::MAPR;
  const Geo G = chunk_of(B, N, K);
  const int sweep = blockIdx.y;
  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[G.b * c_bs + j];
  const double *tb = t + G.b * t_bs, *db = d + G.b * N, *zb = z + G.b * N;
  const double *Wb = W + G.b * N * J, *Ub = U + G.b * N * J;
  double bS[NS], bF[J];
#pragma unroll
  for (int e = 0; e < NS; ++e) bS[e] = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j) bF[j] = sweep == j ? 1.0 : 0.0;
  RowIn<J, true, false> cur, nxt;
  fetch_row<J, true, false>(cur, G, N, kRows - 1, tb, Ub, nullptr, Wb, db, zb);
#pragma unroll 1
  for (int r = kRows - 1; r >= 0; --r) {
    fetch_row<J, true, false>(nxt, G, N, r - 1, tb, Ub, nullptr, Wb, db, zb);
    if (r < G.len) {
      double p[J], g2[J], btau[J];
      const double rd = 1.0 / cur.d;
#pragma unroll
      for (int j = 0; j < J; ++j) p[j] = exp_decay(-cj[j] * cur.dt);
      if constexpr (EXT) {
        if (sweep < J) (void)adjoint_row<J, false, false>(bS, bF, p, cur.u, cur.w, cur.d, 0.0, rd, g2, btau);
        else {
          const int64_t n = G.b * N + G.lo + r;
          double be[J];
          load_row<J>(bwe + n * J, be);
          (void)adjoint_row<J, false, false, true>(bS, bF, p, cur.u, cur.w, cur.d, 0.0, rd, g2, btau, bde[n], be);
        }
      } else {
        if (sweep < J) (void)adjoint_row<J, false, false>(bS, bF, p, cur.u, cur.w, cur.d, cur.z, rd, g2, btau);
        else (void)adjoint_row<J, true, false>(bS, bF, p, cur.u, cur.w, cur.d, cur.z, rd, g2, btau);
      }
    }
    cur = nxt;
  }
  if (G.len > 0) {
    double *o = map + G.g * MAPR + (int64_t)sweep * NST;
#pragma unroll
    for (int e = 0; e < NS; ++e) o[e] = bS[e];
#pragma unroll
    for (int j = 0; j < J; ++j) o[NS + j] = bF[j];
  }
}

// The same with SP sweeps per lane sharing the rows and the decay factors of a pass (one lane per chunk): once the chunks
// alone fill the chip, reading every row J + 1 times costs more than the idle lanes did (32 x 50000 at J = 6: 0.48
// against 0.98 ms).
template <int J, int SP>
__global__ __launch_bounds__(kWave) void k_maps_shared(int64_t B, int64_t N, int64_t K, const double *__restrict__ t,
                                                       int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                       const double *__restrict__ U, const double *__restrict__ d,
                                                       const double *__restrict__ W, const double *__restrict__ z,
                                                       double *__restrict__ map) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST, MAPR = Dim<J>::MAPR;
  const Geo G = chunk_of(B, N, K);
  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[G.b * c_bs + j];
  const double *tb = t + G.b * t_bs, *db = d + G.b * N, *zb = z + G.b * N;
  const double *Wb = W + G.b * N * J, *Ub = U + G.b * N * J;
#pragma unroll 1
  for (int s0 = 0; s0 <= J; s0 += SP) {
    double bS[SP][NS], bF[SP][J];
#pragma unroll
    for (int q = 0; q < SP; ++q) {
#pragma unroll
      for (int e = 0; e < NS; ++e) bS[q][e] = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) bF[q][j] = (s0 + q == j) ? 1.0 : 0.0;
    }
    RowIn<J, true, false> cur, nxt;
    fetch_row<J, true, false>(cur, G, N, kRows - 1, tb, Ub, nullptr, Wb, db, zb);
#pragma unroll 1
    for (int r = kRows - 1; r >= 0; --r) {
      fetch_row<J, true, false>(nxt, G, N, r - 1, tb, Ub, nullptr, Wb, db, zb);
      if (r < G.len) {
        double p[J], g2[J], btau[J];
        const double rd = 1.0 / cur.d;
#pragma unroll
        for (int j = 0; j < J; ++j) p[j] = exp_decay(-cj[j] * cur.dt);
#pragma unroll
        for (int q = 0; q < SP; ++q) {
          if (s0 + q < J) (void)adjoint_row<J, false, false>(bS[q], bF[q], p, cur.u, cur.w, cur.d, cur.z, rd, g2, btau);
          else if (s0 + q == J) (void)adjoint_row<J, true, false>(bS[q], bF[q], p, cur.u, cur.w, cur.d, cur.z, rd, g2, btau);
        }
      }
      cur = nxt;
    }
    if (G.len > 0) {
#pragma unroll
      for (int q = 0; q < SP; ++q) {
        if (s0 + q <= J) {
          double *o = map + G.g * MAPR + (int64_t)(s0 + q) * NST;
#pragma unroll
          for (int e = 0; e < NS; ++e) o[e] = bS[q][e];
#pragma unroll
          for (int j = 0; j < J; ++j) o[NS + j] = bF[q][j];
        }
      }
    }
  }
}

// ---- chain over the chunks of a series, backwards ---------------------------------------------------------------------
// One wavefront per series; lane (i, j) = entry of the J x J adjoint.  ends[g] = adjoint of the state entering the row
// behind chunk g (zero behind the last one).
template <int J>
__global__ __launch_bounds__(kWave) void k_chain(int64_t K, const double *__restrict__ map, double *__restrict__ ends) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST, MAPR = Dim<J>::MAPR;
  __shared__ double bSm[J][J + 1], Xm[J][J + 1], Pm[J][J + 1], bFv[J];
  const int lane = threadIdx.x;
  const bool act = lane < J * J;
  const int i = act ? lane / J : 0, j = act ? lane % J : 0;
  const int64_t b = blockIdx.x;
  const int sij = sym(J, i, j);
  if (act) bSm[i][j] = 0.0;
  if (lane < J) bFv[lane] = 0.0;
  lds_order();
  double phit, ck[J], gs, gf;
  auto fetch = [&](int64_t k) {
    const double *m = map + (b * K + k) * MAPR;
    phit = m[(int64_t)j * NST + NS + i];   // Phi^T(i, j): entry i of the bF result of sweep j
#pragma unroll
    for (int q = 0; q < J; ++q) ck[q] = m[(int64_t)q * NST + sij];
    gs = m[(int64_t)J * NST + sij];
    gf = m[(int64_t)J * NST + NS + i];
  };
  fetch(K - 1);
  for (int64_t k = K - 1; k >= 0; --k) {
    const double mphit = phit, mgs = gs, mgf = gf;
    double mck[J];
#pragma unroll
    for (int q = 0; q < J; ++q) mck[q] = ck[q];
    if (k > 0) fetch(k - 1);
    double *e = ends + (b * K + k) * NST;
    const double cur = bSm[i][j];
    if (act && i <= j) e[sij] = cur;
    if (act && j == 0) e[NS + i] = bFv[i];
    Pm[i][j] = mphit;   // inactive lanes rewrite entry (0, 0) with the same value
    lds_order();
    double x = 0.0;      // (bS Phi)(i, j) = sum_l bS(i, l) Phi^T(j, l)
#pragma unroll
    for (int l = 0; l < J; ++l) x = fma(bSm[i][l], Pm[j][l], x);
    Xm[i][j] = x;
    double nf = mgf, cf = mgs;
#pragma unroll
    for (int q = 0; q < J; ++q) { nf = fma(Pm[i][q], bFv[q], nf); cf = fma(bFv[q], mck[q], cf); }
    lds_order();
    double o = cf;       // (Phi^T bS Phi)(i, j) = sum_q Phi^T(i, q) X(q, j)
#pragma unroll
    for (int q = 0; q < J; ++q) o = fma(Pm[i][q], Xm[q][j], o);
    lds_order();
    if (act) bSm[i][j] = o;
    if (act && j == 0) bFv[i] = nf;
    lds_order();
  }
}

// ---- the sweep that writes the gradients --------------------------------------------------------------------------------
// sf: lane-major record of the states entering the rows of a chunk: sf[((wave * 64 + r) * NST + e) * 64 + lane].
// EXT: factor_rev -- sources bde, bwe as in k_maps, no F state, no by.
template <int J, bool EXT = false>
__global__ __launch_bounds__(kWave) void k_final(int64_t B, int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                 const double *__restrict__ c, int64_t c_bs, const double *__restrict__ U,
                                                 const double *__restrict__ V, const double *__restrict__ d,
                                                 const double *__restrict__ W, const double *__restrict__ z,
                                                 const double *__restrict__ start, const double *__restrict__ ends,
                                                 const int32_t *__restrict__ flag, double *__restrict__ sf,
                                                 double *__restrict__ dT, double *__restrict__ bcp,
                                                 double *__restrict__ ba, double *__restrict__ bU,
                                                 double *__restrict__ bV, double *__restrict__ by,
                                                 unsigned long long *__restrict__ vw,
                                                 const double *__restrict__ bde = nullptr,
                                                 const double *__restrict__ bwe = nullptr) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST;
  const Geo G = chunk_of(B, N, K);
  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[G.b * c_bs + j];
  const double *tb = t + G.b * t_bs, *db = d + G.b * N, *zb = z + G.b * N;
  const double *Wb = W + G.b * N * J, *Ub = U + G.b * N * J, *Vb = EXT ? Ub : V + G.b * N * J;   // (EXT: V unused, reverse.hpp:29-30)
  double *bab = ba + G.b * N, *byb = EXT ? nullptr : by + G.b * N, *bUb = bU + G.b * N * J, *bVb = bV + G.b * N * J, *dTb = dT + G.b * N;
  double *sfw = sf + (size_t)G.wave * kRows * NST * kWave + G.lane;
  const bool failed = flag[G.b] != 0;
  const double nan = __builtin_nan("");

  // forward replay: the states entering the rows, then the one entering the row behind the chunk
  double Sn[NS], Fn[J], smx[J], fmx[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { smx[j] = 0.0; fmx[j] = 0.0; }
  {
    const double *s = start + G.g * NST;
#pragma unroll
    for (int e = 0; e < NS; ++e) Sn[e] = G.len > 0 ? s[e] : 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) Fn[j] = (G.len > 0 && !EXT) ? s[NS + j] : 0.0;
  }
  {
    RowIn<J, false, false> cur, nxt;
    fetch_row<J, false, false>(cur, G, N, 0, tb, nullptr, nullptr, Wb, db, zb);
#pragma unroll 1
    for (int r = 0; r < kRows; ++r) {
      fetch_row<J, false, false>(nxt, G, N, r + 1, tb, nullptr, nullptr, Wb, db, zb);
      if (r < G.len) {
        double *o = sfw + (size_t)r * NST * kWave;
#pragma unroll
        for (int e = 0; e < NS; ++e) o[(size_t)e * kWave] = Sn[e];
#pragma unroll
        for (int j = 0; j < J; ++j) o[(size_t)(NS + j) * kWave] = Fn[j];
        double p[J];
        absorb<J>(Sn, Fn, cur.w, cur.d, EXT ? 0.0 : cur.z);
#pragma unroll
        for (int j = 0; j < J; ++j) {   // the units of the verification below: the largest S_jj, |F_j| of the chunk's rows
          smx[j] = fmax(smx[j], Sn[sidx(J, j, j)]);
          fmx[j] = fmax(fmx[j], fabs(Fn[j]));
        }
        if (G.lo + r + 1 < N) {
#pragma unroll
          for (int j = 0; j < J; ++j) p[j] = exp_decay(-cj[j] * cur.dt);
          decay<J>(Sn, Fn, p);
        }
      }
      cur = nxt;
    }
  }
  // Sn, Fn: state entering row lo + len (meaningless behind the last row of the series, where the adjoint is zero)
  {   // ... which the scan over the chunks gave to the next chunk as its start state
    double es = 0.0, ef = 0.0;
    if (G.len > 0 && G.k + 1 < K && !failed) {
      // (in units of the chunk's own rows, not of the boundary: behind a gap in time the state there is ~0)
      const double *s = start + (G.g + 1) * NST;
      double sd[J];
#pragma unroll
      for (int i = 0; i < J; ++i) sd[i] = sqrt(smx[i]);
#pragma unroll
      for (int i = 0; i < J; ++i) {
#pragma unroll
        for (int j = i; j < J; ++j)
          es = fmax(es, fabs(Sn[sidx(J, i, j)] - s[sidx(J, i, j)]) / fmax(sd[i] * sd[j], 1e-300));
        if constexpr (!EXT) ef = fmax(ef, fabs(Fn[i] - s[NS + i]) / fmax(fmax(fmx[i], sd[i]), 1e-300));
      }
    }
    publish_max(vw + 1, es);
    publish_max(vw + 4, ef);
  }
  double kap = 0.0, bsm = 0.0, bfm = 0.0;   // largest a_n / d_n, |bS_ii| S_ii, bF_i^2 S_ii over the chunk's rows
  double bS[NS], bF[J], bcj[J];
  {
    const double *e = ends + G.g * NST;
#pragma unroll
    for (int q = 0; q < NS; ++q) bS[q] = G.len > 0 ? e[q] : 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) { bF[j] = (G.len > 0 && !EXT) ? e[NS + j] : 0.0; bcj[j] = 0.0; }
  }
  RowIn<J, true, true> cur, nxt;
  fetch_row<J, true, true>(cur, G, N, kRows - 1, tb, Ub, Vb, Wb, db, zb);
#pragma unroll 1
  for (int r = kRows - 1; r >= 0; --r) {
    fetch_row<J, true, true>(nxt, G, N, r - 1, tb, Ub, Vb, Wb, db, zb);
    if (r < G.len) {
      const int64_t n = G.lo + r;
      double p[J], g2[J], btau[J];
      // the state entering row n: requested now, needed behind the adjoint step
      double Sm[NS], Fm[J];
      const double *o = sfw + (size_t)r * NST * kWave;
#pragma unroll
      for (int e = 0; e < NS; ++e) Sm[e] = o[(size_t)e * kWave];
#pragma unroll
      for (int j = 0; j < J; ++j) Fm[j] = o[(size_t)(NS + j) * kWave];
      const double dn = cur.d, zn = EXT ? 0.0 : cur.z, rd = 1.0 / dn, dt = cur.dt;
      // the decay between rows n and n+1 (internal.hpp:238-243 / reverse.hpp:70-76): p_i bp_i = 2 sum_j bS'(i,j)
      // S_{n+1}(i,j) + bF'_i F_{n+1,i} with the states ENTERING row n+1 (dt = 0 behind the last row)
      double bdt = 0.0;
#pragma unroll
      for (int i = 0; i < J; ++i) {
        double s = bF[i] * Fn[i];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) s = fma(2.0 * bS[sym(J, i, jj)], Sn[sym(J, i, jj)], s);
        bcj[i] = fma(-dt, s, bcj[i]);
        bdt = fma(-cj[i], s, bdt);
        p[i] = exp_decay(-cj[i] * dt);
      }
      if (n + 1 >= N) bdt = 0.0;
      RowOut ro;
      if constexpr (EXT) {
        double be[J];
        load_row<J>(bwe + (G.b * N + n) * J, be);
        ro = adjoint_row<J, false, true, true>(bS, bF, p, cur.u, cur.w, dn, zn, rd, g2, btau, bde[G.b * N + n], be);
      } else {
        ro = adjoint_row<J, true, true>(bS, bF, p, cur.u, cur.w, dn, zn, rd, g2, btau);
      }
#pragma unroll
      for (int e = 0; e < NS; ++e) Sn[e] = Sm[e];
#pragma unroll
      for (int j = 0; j < J; ++j) Fn[j] = Fm[j];
      // bU_n = -bz F_n - bd tau_n + S_n btau,  tau_n = v_n - d_n w_n
      double bu[J], ut = 0.0;   // ut = u_n tau_n = a_n - d_n
#pragma unroll
      for (int i = 0; i < J; ++i) {
        double s;
        if constexpr (EXT) {   // tau_n = S_n u_n from the replayed state
          double tau = 0.0;
#pragma unroll
          for (int jj = 0; jj < J; ++jj) tau = fma(Sn[sym(J, i, jj)], cur.u[jj], tau);
          s = -ro.bd * tau;
          ut = fma(cur.u[i], tau, ut);
        } else {
          const double tau = fma(-dn, cur.w[i], cur.v[i]);
          s = fma(-ro.bz, Fn[i], -ro.bd * tau);
          ut = fma(cur.u[i], tau, ut);
        }
#pragma unroll
        for (int jj = 0; jj < J; ++jj) s = fma(Sn[sym(J, i, jj)], btau[jj], s);
        bu[i] = failed ? nan : s;
        g2[i] = failed ? nan : g2[i];
      }
      kap = fmax(kap, 1.0 + fabs(ut) * rd);
#pragma unroll
      for (int i = 0; i < J; ++i) {   // (bS, bF: adjoint of the state entering row n; Sn: that state)
        bsm = fmax(bsm, fabs(bS[sidx(J, i, i)]) * Sn[sidx(J, i, i)]);
        if constexpr (!EXT) bfm = fmax(bfm, bF[i] * bF[i] * Sn[sidx(J, i, i)]);
      }
      store_row<J>(bUb + n * J, bu);
      store_row<J>(bVb + n * J, g2);
      bab[n] = failed ? nan : ro.bd;
      if constexpr (!EXT) byb[n] = failed ? nan : ro.bz;
      dTb[n] = bdt;
    }
    cur = nxt;
  }
  if (G.len > 0) {
#pragma unroll
    for (int j = 0; j < J; ++j) bcp[G.g * J + j] = bcj[j];
  }
  {   // (bS, bF): adjoint of the state entering row lo -- what the chain over the chunks gave to the chunk in front of this
      // one as its end adjoint; weighted with the state it pairs with (Sn is back at the state entering row lo)
    double eb = 0.0;
    if (G.len > 0 && G.k > 0 && !failed) {
      // in units of the largest weighted adjoint of the chunk's rows (at one boundary an adjoint may pass through zero)
      const double *e = ends + (G.g - 1) * NST;
      double sd[J], num = 0.0, numf = 0.0;
#pragma unroll
      for (int i = 0; i < J; ++i) sd[i] = sqrt(fabs(Sn[sidx(J, i, i)]));
#pragma unroll
      for (int i = 0; i < J; ++i) {
#pragma unroll
        for (int j = i; j < J; ++j)
          num = fmax(num, fabs(bS[sidx(J, i, j)] - e[sidx(J, i, j)]) * (sd[i] * sd[j]));
        if constexpr (!EXT) numf = fmax(numf, fabs(bF[i] - e[NS + i]) * sd[i]);
      }
      eb = num > 0.0 ? num / fmax(bsm, 1e-300) : 0.0;
      if constexpr (!EXT) eb = fmax(eb, numf > 0.0 ? numf / fmax(sqrt(bfm), 1e-300) : 0.0);
    }
    publish_max(vw + 2, eb);
    publish_max(vw + 3, failed ? 0.0 : kap);
  }
}

// bt_n = dT_{n-1} - dT_n (dT_n: gradient w.r.t. the gap t_{n+1} - t_n)
__global__ void k_finish_t(int64_t B, int64_t N, const double *__restrict__ dT, const int32_t *__restrict__ flag,
                           double *__restrict__ bt) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g < B * N) {
    const int64_t b = g / N, n = g - b * N;
    const double v = (n > 0 ? dT[g - 1] : 0.0) - dT[g];
    bt[g] = flag[b] != 0 ? __builtin_nan("") : v;
  }
}
// bc = sum of the chunks' parts: one wavefront per (series, j), lanes strided over the chunks, a fixed tree
template <int J>
__global__ __launch_bounds__(kWave) void k_finish_c(int64_t K, const double *__restrict__ bcp,
                                                    const int32_t *__restrict__ flag, double *__restrict__ bc) {
  const int64_t b = blockIdx.x / J;
  const int j = (int)(blockIdx.x % J);
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < K; k += kWave) s += bcp[(b * K + k) * J + j];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
  if (threadIdx.x == 0) bc[b * J + j] = flag[b] != 0 ? __builtin_nan("") : s;
}

// ---- z = L^-1 y (solve_lower, one right-hand side) for widths the tiled kernels of c2_timepar.hip do not cover (6) ------
// The recursion is affine in its state with the SAME propagator as above: F_{n+1} = P (I - w_n u_n^T) F_n + P w_n y_n,
// z_n = y_n - u_n F_n (internal.hpp:135-145).  k_solve_maps: (Phi_k, g_k) of every chunk; k_solve_chain: the states the
// chunks start from; k_solve_apply: z of every row.
// ys: stride of y (1, or nrhs for one column of a row-major Y); PHI = false: g_k only (further columns of the same series)
template <int J, bool REV = false, bool PHI = true>
__global__ __launch_bounds__(kWave) void k_solve_maps(int64_t B, int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                      const double *__restrict__ c, int64_t c_bs, const double *__restrict__ U,
                                                      const double *__restrict__ W, const double *__restrict__ y, int64_t ys,
                                                      double *__restrict__ Phi, double *__restrict__ gk) {
  const Geo G = chunk_of(B, N, K);
  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[G.b * c_bs + j];
  const double *tb = t + G.b * t_bs, *yb = y + G.b * N * ys, *Ub = U + G.b * N * J, *Wb = W + G.b * N * J;
  double M[J][J], F[J];
#pragma unroll
  for (int i = 0; i < J; ++i) {
    F[i] = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) M[i][j] = i == j ? 1.0 : 0.0;
  }
  RowIn<J, true, false> cur, nxt;   // d <- y (unused), z <- y
  fetch_row<J, true, false, REV>(cur, G, N, 0, tb, Ub, nullptr, Wb, yb, yb, ys);
#pragma unroll 1
  for (int r = 0; r < kRows; ++r) {
    fetch_row<J, true, false, REV>(nxt, G, N, r + 1, tb, Ub, nullptr, Wb, yb, yb, ys);
    if (r < G.len) {
      double p[J], uf = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) { p[j] = exp_decay(-cj[j] * cur.dt); uf = fma(cur.u[j], F[j], uf); }
      const double zn = cur.z - uf;
#pragma unroll
      for (int i = 0; i < J; ++i) F[i] = fma(cur.w[i], zn, F[i]) * p[i];
      if constexpr (PHI) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          double um = 0.0;
#pragma unroll
          for (int l = 0; l < J; ++l) um = fma(cur.u[l], M[l][j], um);
#pragma unroll
          for (int i = 0; i < J; ++i) M[i][j] = fma(-cur.w[i], um, M[i][j]) * p[i];
        }
      }
    }
    cur = nxt;
  }
  if (G.len > 0) {
#pragma unroll
    for (int i = 0; i < J; ++i) {
      gk[G.g * J + i] = F[i];
      if constexpr (PHI) {
#pragma unroll
        for (int j = 0; j < J; ++j) Phi[G.g * (J * J) + i * J + j] = M[i][j];
      }
    }
  }
}
// one wavefront per series, lane i < J <-> F_i
template <int J>
__global__ __launch_bounds__(kWave) void k_solve_chain(int64_t K, const double *__restrict__ Phi,
                                                       const double *__restrict__ gk, double *__restrict__ Fst) {
  const int lane = threadIdx.x, i = lane < J ? lane : 0;
  const int64_t b = blockIdx.x;
  double F = 0.0;
  constexpr int PF = 4;
  for (int64_t k0 = 0; k0 < K; k0 += PF) {
    double ph[PF][J], gv[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int64_t g = b * K + (k0 + q < K ? k0 + q : K - 1);
      gv[q] = gk[g * J + i];
#pragma unroll
      for (int j = 0; j < J; ++j) ph[q][j] = Phi[g * (J * J) + i * J + j];
    }
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      if (k0 + q < K) {
        if (lane < J) Fst[(b * K + k0 + q) * J + lane] = F;
        double nf = gv[q];
#pragma unroll
        for (int j = 0; j < J; ++j) nf = fma(ph[q][j], __shfl(F, j, kWave), nf);
        F = nf;
      }
    }
  }
}
// k_solve_chain in two levels for long series (see k_newton_block): F_{k+1} = Phi_k F_k + g_k composes over a block of
// chunks as F_{k+1} = Psi_k F_block + gamma_k with Psi_k = Phi_k Psi_{k-1}, gamma_k = Phi_k gamma_{k-1} + g_k.
template <int J>
__global__ __launch_bounds__(kWave) void k_solve_block(int64_t K, int64_t NB, const double *__restrict__ Phi,
                                                       const double *__restrict__ gk, double *__restrict__ Psi,
                                                       double *__restrict__ Gam) {
  __shared__ double Pm[J][J + 1], Sm[J][J + 1], Gv[J];
  const int lane = threadIdx.x;
  const bool act = lane < J * J;
  const int i = act ? lane / J : 0, j = act ? lane % J : 0;
  const int64_t b = blockIdx.x / NB, nb = blockIdx.x % NB;
  const int64_t k0 = nb * kBlock, k1 = (k0 + kBlock < K) ? k0 + kBlock : K;   // chunks k0 .. k1-1
  if (act) Sm[i][j] = i == j ? 1.0 : 0.0;
  if (lane < J) Gv[lane] = 0.0;
  lds_order();
  double ph, gv;
  auto fetch = [&](int64_t k) {
    ph = Phi[(b * K + k) * (J * J) + i * J + j];
    gv = gk[(b * K + k) * J + i];
  };
  fetch(k0);
  for (int64_t k = k0; k < k1; ++k) {
    const double mph = ph, mgv = gv;
    if (k + 1 < k1) fetch(k + 1);
    Pm[i][j] = mph;
    lds_order();
    double ps = 0.0, gn = mgv;   // (Phi Psi)(i, j), (Phi gamma + g)(i)
#pragma unroll
    for (int l = 0; l < J; ++l) { ps = fma(Pm[i][l], Sm[l][j], ps); gn = fma(Pm[i][l], Gv[l], gn); }
    lds_order();
    if (act) {
      Sm[i][j] = ps;
      Psi[(b * K + k) * (J * J) + i * J + j] = ps;
      if (j == 0) { Gv[i] = gn; Gam[(b * K + k) * J + i] = gn; }
    }
    lds_order();
  }
}
// one wavefront per series over the blocks; Fb[series][block][J] = state entering the block
template <int J>
__global__ __launch_bounds__(kWave) void k_solve_blocks(int64_t K, int64_t NB, const double *__restrict__ Psi,
                                                        const double *__restrict__ Gam, double *__restrict__ Fb) {
  const int lane = threadIdx.x, i = lane < J ? lane : 0;
  const int64_t b = blockIdx.x;
  double F = 0.0;
  for (int64_t nb = 0; nb < NB; ++nb) {
    const int64_t kl = ((nb + 1) * kBlock < K ? (nb + 1) * kBlock : K) - 1;
    if (lane < J) Fb[(b * NB + nb) * J + lane] = F;
    double nf = Gam[(b * K + kl) * J + i];
#pragma unroll
    for (int j = 0; j < J; ++j) nf = fma(Psi[(b * K + kl) * (J * J) + i * J + j], __shfl(F, j, kWave), nf);
    F = nf;
  }
}
// Fst[chunk] = state entering the chunk: the block's for its first chunk, Psi_{k-1} F_block + gamma_{k-1} behind it
template <int J>
__global__ void k_solve_starts(int64_t B, int64_t K, int64_t NB, const double *__restrict__ Psi,
                               const double *__restrict__ Gam, const double *__restrict__ Fb, double *__restrict__ Fst) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * K * J) return;
  const int i = (int)(g % J);
  const int64_t bk = g / J, b = bk / K, k = bk - b * K, nb = k / kBlock;
  const double *F = Fb + (b * NB + nb) * J;
  double v;
  if (k % kBlock == 0) v = F[i];
  else {
    v = Gam[(bk - 1) * J + i];
#pragma unroll
    for (int j = 0; j < J; ++j) v = fma(Psi[(bk - 1) * (J * J) + i * J + j], F[j], v);
  }
  Fst[g] = v;
}
template <int J, bool REV = false>
__global__ __launch_bounds__(kWave) void k_solve_apply(int64_t B, int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                       const double *__restrict__ c, int64_t c_bs, const double *__restrict__ U,
                                                       const double *__restrict__ W, const double *__restrict__ y,
                                                       int64_t ys, const double *__restrict__ Fst, double *__restrict__ z,
                                                       double *__restrict__ Fw, int64_t fs,
                                                       unsigned long long *__restrict__ vw,
                                                       const int32_t *__restrict__ flag) {
  // Fw (optional): this right-hand side's J entries of the reference's workspace rows, Fw[row fs + j] -- the state of the row
  // before its decay, in both directions (internal.hpp:140-141, 179-180: update_workspace precedes Fn = p Fn)
  const Geo G = chunk_of(B, N, K);
  double cj[J], F[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { cj[j] = c[G.b * c_bs + j]; F[j] = G.len > 0 ? Fst[G.g * J + j] : 0.0; }
  const double *tb = t + G.b * t_bs, *yb = y + G.b * N * ys, *Ub = U + G.b * N * J, *Wb = W + G.b * N * J;
  double *zb = z + G.b * N * ys;
  double *fb = Fw ? Fw + G.b * N * fs : nullptr;
  RowIn<J, true, false> cur, nxt;
  double ul[J], yl = 0.0;   // |u|, |y| of the chunk's last row: the units of the verification below
#pragma unroll
  for (int j = 0; j < J; ++j) ul[j] = 0.0;
  fetch_row<J, true, false, REV>(cur, G, N, 0, tb, Ub, nullptr, Wb, yb, yb, ys);
  if (fb && G.len > 0 && G.lo == 0) {   // the first row of the walk carries no state
#pragma unroll
    for (int i = 0; i < J; ++i) fb[(REV ? N - 1 : 0) * fs + i] = 0.0;
  }
#pragma unroll 1
  for (int r = 0; r < kRows; ++r) {
    fetch_row<J, true, false, REV>(nxt, G, N, r + 1, tb, Ub, nullptr, Wb, yb, yb, ys);
    if (r < G.len) {
      double uf = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) uf = fma(cur.u[j], F[j], uf);
      const double zn = cur.z - uf;
      const int64_t pos = G.lo + r, n = REV ? N - 1 - pos : pos;
      zb[n * ys] = zn;
      if (vw && r == G.len - 1) {
        yl = fabs(cur.z);
#pragma unroll
        for (int j = 0; j < J; ++j) ul[j] = fabs(cur.u[j]);
      }
#pragma unroll
      for (int i = 0; i < J; ++i) F[i] = fma(cur.w[i], zn, F[i]);
      if (fb && pos + 1 < N) {   // the state of the next row of the walk, before its decay
        const int64_t nn = REV ? n - 1 : n + 1;
#pragma unroll
        for (int i = 0; i < J; ++i) fb[nn * fs + i] = F[i];
      }
#pragma unroll
      for (int i = 0; i < J; ++i) F[i] *= exp_decay(-cj[i] * cur.dt);
    }
    cur = nxt;
  }
  if (vw) {   // F: the state entering the next chunk's first row -- what the chain gave that chunk to start from?
    double ez = 0.0;
    if (G.len > 0 && G.k + 1 < K && !(flag && flag[G.b] != 0)) {   // (a failed factorisation: W is not a solve's W)
      double num = 0.0, den = yl;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const double fn = Fst[(G.g + 1) * J + j];
        num = fma(ul[j], fabs(F[j] - fn), num);
        den = fma(ul[j], fabs(fn), den);
      }
      ez = num > 0.0 ? num / fmax(den, 1e-300) : 0.0;
    }
    publish_max(vw, ez);
  }
}

// ---- forward-only log-likelihood from d and z (factor by Newton iterations + chunk-map solve) ------------------------------
__global__ __launch_bounds__(kWave) void k_ll_chunks(int64_t B, int64_t N, int64_t K, const double *__restrict__ d,
                                                     const double *__restrict__ z, double *__restrict__ llp) {
  const Geo G = chunk_of(B, N, K);
  const double *db = d + G.b * N + G.lo, *zb = z + G.b * N + G.lo;
  double acc = 0.0;
  for (int r = 0; r < G.len; ++r) acc += log(db[r]) + zb[r] * zb[r] / db[r];
  if (G.len > 0) llp[G.g] = acc;
}
__global__ __launch_bounds__(kWave) void k_ll_series(int64_t N, int64_t K, const double *__restrict__ llp,
                                                     const int32_t *__restrict__ flag, double *__restrict__ ll) {
  const int64_t b = blockIdx.x;
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < K; k += kWave) s += llp[b * K + k];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, kWave);
  if (threadIdx.x == 0) ll[b] = flag[b] != 0 ? -__builtin_huge_val() : -0.5 * s - 0.5 * (double)N * kLog2Pi;
}

// z = L^-1 y by chunk maps (k_solve_*); scratch: B K (2 J^2 + 3 J) + B (K / kBlock + 1) J doubles
template <int J, bool REV = false>
static void solve_chunks(int64_t B, int64_t N, int64_t K, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                         const double *U, const double *W, const double *y, double *z, double *scratch, hipStream_t s,
                         int64_t ys = 1, bool with_phi = true, double *Fw = nullptr, int64_t fs = 0,
                         unsigned long long *vw = nullptr, const int32_t *flag = nullptr) {
  // ys > 1: one column of a row-major Y / Z; with_phi = false: Phi (and its block products) of an earlier call with the
  // same series are still in the scratch
  const size_t BK = (size_t)B * K;
  double *Phi = scratch, *gk = Phi + BK * J * J, *Fst = gk + BK * J;
  const dim3 cgrid((unsigned)((B * K + kWave - 1) / kWave));
  if (with_phi)
    hipLaunchKernelGGL((k_solve_maps<J, REV, true>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, U, W, y, ys, Phi, gk);
  else
    hipLaunchKernelGGL((k_solve_maps<J, REV, false>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, U, W, y, ys, Phi, gk);
  if (K >= kTwoLevelMin) {
    const int64_t NB = (K + kBlock - 1) / kBlock;
    double *Psi = Fst + BK * J, *Gam = Psi + BK * J * J, *Fb = Gam + BK * J;
    hipLaunchKernelGGL((k_solve_block<J>), dim3((unsigned)(B * NB)), dim3(kWave), 0, s, K, NB, (const double *)Phi,
                       (const double *)gk, Psi, Gam);
    hipLaunchKernelGGL((k_solve_blocks<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, NB, (const double *)Psi,
                       (const double *)Gam, Fb);
    hipLaunchKernelGGL((k_solve_starts<J>), dim3((unsigned)((B * K * J + 255) / 256)), dim3(256), 0, s, B, K, NB,
                       (const double *)Psi, (const double *)Gam, (const double *)Fb, Fst);
  } else {
    hipLaunchKernelGGL((k_solve_chain<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, (const double *)Phi,
                       (const double *)gk, Fst);
  }
  hipLaunchKernelGGL((k_solve_apply<J, REV>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, U, W, y, ys,
                     (const double *)Fst, z, Fw, fs, vw, flag);
}

struct Layout {
  size_t vw, d, W, z, loc, start, ends, map, sf, dT, bcp, llp, fs, total;
};
template <int J>
static Layout layout(int64_t B, int64_t N) {
  constexpr size_t NST = Dim<J>::NST, MAPR = Dim<J>::MAPR;
  const size_t K = (size_t)((N + kRows - 1) / kRows), BN = (size_t)B * N, BK = (size_t)B * K;
  const size_t waves = (BK + kWave - 1) / kWave;
  Layout L;
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += (n + 1) & ~(size_t)1; return at; };   // 16-byte aligned pieces
  L.vw = take(kVerifyWords);   // (first: the caller gates its row-by-row kernels on word 0 of the workspace)
  L.d = take(BN); L.W = take(BN * J); L.z = take(BN);
  L.loc = take(BK * NST); L.start = take(BK * NST); L.ends = take(BK * NST);
  L.map = take(BK * MAPR);
  L.sf = take(waves * kRows * NST * kWave);
  L.dT = take(BN); L.bcp = take(BK * J); L.llp = take(BK);
  L.fs = take(c2_internal_factor_scratch_doubles(B, N, J));   // room for the factor's own time-parallel form: no allocation
  L.total = o;
  return L;
}

// the adjoint chain in two levels (long series); defined behind the Newton kernels it shares
template <int J>
static void adjoint_chain_two_level(int64_t B, int64_t K, const double *map, double *ends, double *scratch, hipStream_t s);

template <int J>
static int run(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *a, const double *U,
               const double *V, const double *y, double *ll, double *bt, double *bc, double *ba, double *bU,
               double *bV, double *by, int32_t *flag, double *work, hipStream_t s) {
  const Layout L = layout<J>(B, N);
  const int64_t K = (N + kRows - 1) / kRows;
  double *d = work + L.d, *W = work + L.W, *z = work + L.z;
  unsigned long long *vw = reinterpret_cast<unsigned long long *>(work + L.vw);
  if (hipMemsetAsync(vw, 0, kVerifyWords * sizeof(double), s) != hipSuccess) return C2_ERR_HIP;
  if (int e = c2_internal_factor_fused_ws(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, C2TG_FACTOR_MODE, work + L.fs,
                                          (c2_stream_t)s))
    return e;
  if (double *sink = c2_internal_get_debug_sink())   // (diagnostics: the factor's iteration words, behind the 8 verification words)
    (void)hipMemcpyAsync(sink + kVerifyWords, work + L.fs, kNewtonHdr * sizeof(double), hipMemcpyDeviceToDevice, s);
  const dim3 cgrid((unsigned)((B * K + kWave - 1) / kWave));
  // z by the chunk maps (allocation-free, so the whole call can be captured in a graph; scratch: the record of the states,
  // which k_final fills afterwards)
  solve_chunks<J>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, work + L.sf, s, 1, true, nullptr, 0, vw + 5, flag);
  hipLaunchKernelGGL((k_local<J>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, (const double *)d, (const double *)W,
                     (const double *)z, work + L.loc, work + L.llp);
  if (K >= kTwoLevelMin) {   // (the prefix maps live in the region of the adjoint maps, written later)
    const int64_t NBs = (K + kWave - 1) / kWave;
    hipLaunchKernelGGL((k_starts_blocks<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, N, K, t, t_bs, c, c_bs,
                       (const double *)(work + L.loc), work + L.map);
    hipLaunchKernelGGL((k_starts_apply<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, N, K,
                       (const double *)(work + L.map), (const double *)(work + L.llp), (const int32_t *)flag,
                       work + L.start, ll);
  } else
  hipLaunchKernelGGL((k_starts<J>), dim3((unsigned)B), dim3(kWave), 0, s, N, K, t, t_bs, c, c_bs,
                     (const double *)(work + L.loc),
                     (const double *)(work + L.llp), (const int32_t *)flag, work + L.start, ll);
  // the chunks alone put a wavefront on every eighth SIMD: share the rows between the sweeps (width 8, three sweeps of
  // 44 doubles a lane, spills into the accumulation registers and only pays when the chip is full: 256 x 4096 1.8 vs 2.4 ms)
  if (cgrid.x >= (J == 8 ? 1024u : 128u)) {
    constexpr int SP = J >= 6 ? 3 : (J + 1);
    hipLaunchKernelGGL((k_maps_shared<J, SP>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, U, (const double *)d,
                       (const double *)W, (const double *)z, work + L.map);
  } else
  hipLaunchKernelGGL((k_maps<J>), dim3(cgrid.x, (unsigned)(J + 1)), dim3(kWave), 0, s, B, N, K, t, t_bs, c,
                     c_bs, U, (const double *)d, (const double *)W, (const double *)z, work + L.map);
  if (K >= kTwoLevelMin)   // (scratch: the record of the states, which k_final fills afterwards)
    adjoint_chain_two_level<J>(B, K, work + L.map, work + L.ends, work + L.sf, s);
  else
    hipLaunchKernelGGL((k_chain<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, (const double *)(work + L.map),
                       work + L.ends);
  hipLaunchKernelGGL((k_final<J>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, U, V, (const double *)d, (const double *)W,
                     (const double *)z, (const double *)(work + L.start), (const double *)(work + L.ends),
                     (const int32_t *)flag, work + L.sf, work + L.dT, work + L.bcp, ba, bU, bV, by, vw);
  hipLaunchKernelGGL(k_verify_combine, dim3(1), dim3(1), 0, s, vw, opt::val(opt::k_timepar_cond_limit));
  hipLaunchKernelGGL(k_finish_t, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, s, B, N,
                     (const double *)(work + L.dT), (const int32_t *)flag, bt);
  hipLaunchKernelGGL((k_finish_c<J>), dim3((unsigned)(B * J)), dim3(kWave), 0, s, K, (const double *)(work + L.bcp),
                     (const int32_t *)flag, bc);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// ---- factor by Newton iterations on the chunk start states (widths 1 .. 8) --------------------------------------------------
// Unknowns: X_k, the state entering chunk k (X_0 = 0).  Equations: X_{k+1} = f_k(X_k), f_k = the factor recursion
// (forward.hpp:105-134) over the 64 rows of chunk k.  The Jacobian of f_k is a congruence, dX -> Phi_k dX Phi_k^T with
// Phi_k = prod_n P_{n+1} (I - w_n u_n^T) (the closed-loop propagator of the underlying Kalman filter), so one Newton
// iteration is
//   (1) k_newton_pass: every chunk of every series, one lane each, walks its rows from the current X_k: d, W of its rows,
//       its end state E_k = f_k(X_k) and Phi_k (a J x J product carried along);
//   (2) k_newton_chain: one wavefront per series walks the chunks, delta_{k+1} = Phi_k delta_k Phi_k^T + E_k - X_{k+1},
//       X_{k+1} += delta_{k+1}, and writes the largest relative update into the iteration's device word.
// Convergence is quadratic (the numpy prototype: updates 1, 4e-2, 3e-4, 2e-8, 2e-15 on the bench series); the
// update of iteration p measures the error of the X that pass p used, so the iterations stop -- each is launched behind
// the previous word as its gate -- once it is below kNewtonTol, with d, W of that pass final.  If the last iteration still
// moves, or a pass met a d that is not positive and finite, the caller's row-by-row kernel runs behind the last word.
constexpr double kNewtonTol = 1e-13;
// ... relative to sqrt(X_ii X_jj).  An error rho of X moves d_n = a_n - u_n X u_n^T by ~rho (a_n - d_n), i.e. by rho * kappa_n
// RELATIVE to d_n with kappa_n = a_n / d_n (1600 on the two series of the 6030-seed stress run of round 2 that missed 1e-10:
// the update that ended their iterations was just below 1e-13).  Every pass therefore measures kappa = max a_n / d_n into
// the iteration's second word and the chains stop at  min(kNewtonTol, kNewtonCondTol / kappa)  instead: d, W good to
// kNewtonCondTol whatever the conditioning, or -- if rounding keeps the updates above that -- the row-by-row kernel.
constexpr double kNewtonCondTol = 1e-11;
__device__ __forceinline__ double newton_tol(const unsigned long long *kapw) {
  const double kap = __longlong_as_double((long long)*kapw);
  return fmin(kNewtonTol, kNewtonCondTol / fmax(kap, 1.0));
}
// ... and rounding does keep them above it once kappa passes a few hundred: the updates of a converged iteration sit at
// 4e-15 (kappa = 90), 7e-15 (340), 1.1e-14 (1300), 2.6e-14 (5400) of the state -- the floor of float64, not an error that another
// iteration removes -- while kNewtonCondTol / kappa asks for 7.7e-15 at kappa = 1300: round 3's rule sent every such batch
// through all eight iterations AND the row-by-row kernel (the 1-D problem inside BASELINE configs[4], white noise 1 / A =
// 0.0125: 2.4 -> 22 ms).  Convergence is quadratic, so an iteration whose PREDECESSOR's update was
// already below kNewtonBasin started from a state good to ~0.1 kNewtonBasin^2: its own update is that floor, and it stands if it is below
// kNewtonFloorTol (anything larger is not rounding).  d, W then carry kappa x floor -- what the row-by-row recursion carries
// too (no float64 order beats eps kappa).
constexpr double kNewtonBasin = 3e-7, kNewtonFloorTol = 1e-11;   // (updates go 1e-3 -> 1.2e-7 -> floor on the bench series: (3e-7)^2 x 0.1 = 1e-14)
// half the tolerance iteration p is held to: kapw = its conditioning word; kapw + (kNewtonMax + 2) = its absolute update,
// kapw + (kNewtonMax + 2) - 1 the previous iteration's (0 for the first: the memset)
__device__ __forceinline__ double newton_half_tol(const unsigned long long *gate, const unsigned long long *kapw) {
  double tol = newton_tol(kapw);
  if (gate) {   // (iteration p >= 2)
    const double prev = __longlong_as_double((long long)kapw[(kNewtonMax + 2) - 1]);
    if (prev > 0.0 && prev <= kNewtonBasin) tol = fmax(tol, kNewtonFloorTol);
  }
  return 0.5 * tol;
}
// SCAN (width 8, round 6): X holds the EXACT start states (run8_states): no propagator, no end states stored -- the end state
// of the chunk is compared here with the start state its successor was given, the mismatch (relative to sqrt(S_ii S_jj), S_ii
// the largest of the chunk's start state, its end state and the successor's start state: behind a gap in time the state
// itself is ~0) goes to `word` in units of half of kE8CheckTol.
constexpr double kE8CheckTol = 1e-12;   // rounding leaves 1e-15 .. 3e-14 (tools/e8_words.py); anything beyond is not rounding
template <int J, bool SCAN = false>
__global__ __launch_bounds__(kWave) void k_newton_pass(int64_t B, int64_t N, int64_t K, const double *__restrict__ t,
                                                       int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                       const double *__restrict__ a, const double *__restrict__ U,
                                                       const double *__restrict__ V, double *__restrict__ d,
                                                       double *__restrict__ W, int32_t *__restrict__ flag,
                                                       const double *__restrict__ X, double *__restrict__ E,
                                                       double *__restrict__ Phi,
                                                       const unsigned long long *__restrict__ gate,
                                                       unsigned long long *__restrict__ word,
                                                       unsigned long long *__restrict__ kapw) {
  constexpr int NS = Dim<J>::NS;
  if (gate_closed(gate)) return;
  const Geo G = chunk_of(B, N, K);
  double cj[J];
  double kap = 0.0;   // max a_n / d_n over the rows of the chunk
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[G.b * c_bs + j];
  const double *tb = t + G.b * t_bs, *ab = a + G.b * N, *Ub = U + G.b * N * J, *Vb = V + G.b * N * J;
  double *db = d + G.b * N, *Wb = W + G.b * N * J;
  double S[NS], M[J][J];
#pragma unroll
  for (int e = 0; e < NS; ++e) S[e] = G.len > 0 ? X[G.g * NS + e] : 0.0;
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) M[i][j] = i == j ? 1.0 : 0.0;
  bool bad = false;
  double un[J], vn[J], an, dtn;   // row inputs one iteration ahead
  auto fetch = [&](int r, double (&u)[J], double (&v)[J], double &av, double &dt) {
    const int rc = r < G.len ? r : (G.len > 0 ? G.len - 1 : 0);
    const int64_t n = G.lo + rc;
    load_row<J>(Ub + n * J, u);
    load_row<J>(Vb + n * J, v);
    av = ab[n];
    dt = tb[n + 1 < N ? n + 1 : n] - tb[n];
  };
  fetch(0, un, vn, an, dtn);
#pragma unroll 1
  for (int r = 0; r < kRows; ++r) {
    double u[J], v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { u[j] = un[j]; v[j] = vn[j]; }
    const double av = an, dt = dtn;
    fetch(r + 1, un, vn, an, dtn);
    if (r < G.len) {
      const int64_t n = G.lo + r;
      double tau[J], w[J], p[J];
      double dn = av;
#pragma unroll
      for (int i = 0; i < J; ++i) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < J; ++j) sum = fma(S[sym(J, i, j)], u[j], sum);
        tau[i] = sum;
        dn = fma(-u[i], sum, dn);
      }
      bad = bad || !(dn > 0.0) || !(dn < __builtin_huge_val());
      const double rd = 1.0 / dn;
      kap = fmax(kap, fabs(av * rd));
#pragma unroll
      for (int i = 0; i < J; ++i) { w[i] = (v[i] - tau[i]) * rd; p[i] = exp_decay(-cj[i] * dt); }
      db[n] = dn;
      store_row<J>(Wb + n * J, w);
#pragma unroll
      for (int i = 0; i < J; ++i) {
        const double dwp = dn * w[i] * p[i];
#pragma unroll
        for (int j = i; j < J; ++j)
          S[sidx(J, i, j)] = fma(dwp, w[j] * p[j], S[sidx(J, i, j)] * (p[i] * p[j]));   // (S + d w_i w_j) p_i p_j
      }
      // M <- P (I - w u^T) M
      if constexpr (!SCAN) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
          double um = 0.0;
#pragma unroll
          for (int l = 0; l < J; ++l) um = fma(u[l], M[l][j], um);
#pragma unroll
          for (int i = 0; i < J; ++i) M[i][j] = fma(-w[i], um, M[i][j]) * p[i];
        }
      }
    }
  }
  if constexpr (SCAN) {
    double worst = 0.0;
    if (G.len > 0 && G.k + 1 < K) {
      const double *x0 = X + G.g * NS, *x1 = X + (G.g + 1) * NS;
      double dg[J];
#pragma unroll
      for (int i = 0; i < J; ++i)
        dg[i] = fmax(fmax(fabs(x1[sidx(J, i, i)]), fabs(S[sidx(J, i, i)])), fabs(x0[sidx(J, i, i)]));
#pragma unroll
      for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = i; j < J; ++j) {
          const double r = fabs(S[sidx(J, i, j)] - x1[sidx(J, i, j)]) / fmax(sqrt(dg[i] * dg[j]), 1e-300);
          worst = (r == r) ? fmax(worst, r) : __builtin_huge_val();
        }
    }
    if (G.len > 0 && G.k == 0) flag[G.b] = 0;
    publish_max(word, worst * (2.0 / kE8CheckTol));
  } else if (G.len > 0) {
#pragma unroll
    for (int e = 0; e < NS; ++e) E[G.g * NS + e] = S[e];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = 0; j < J; ++j) Phi[G.g * (J * J) + i * J + j] = M[i][j];
    if (G.k == 0) flag[G.b] = 0;
  }
  if (__any(bad) && threadIdx.x == 0)
    atomicMax(word, (unsigned long long)__double_as_longlong(__builtin_huge_val()));
  if (!(kap < __builtin_huge_val())) kap = 0.0;   // (a bad pivot already opened the word above)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) kap = fmax(kap, __shfl_xor(kap, o, kWave));
  if (threadIdx.x == 0 && kap > 0.0) atomicMax(kapw, (unsigned long long)__double_as_longlong(kap));
}

// One wavefront per series; lane (i, j) = entry of the J x J update.
template <int J>
__global__ __launch_bounds__(kWave) void k_newton_chain(int64_t K, double *__restrict__ X, const double *__restrict__ E,
                                                        const double *__restrict__ Phi,
                                                        const unsigned long long *__restrict__ gate,
                                                        unsigned long long *__restrict__ word,
                                                        const unsigned long long *__restrict__ kapw) {
  constexpr int NS = Dim<J>::NS;
  if (gate_closed(gate)) return;
  __shared__ double Dm[J][J + 1], Ym[J][J + 1], Pm[J][J + 1], Xd[J];
  const int lane = threadIdx.x;
  const bool act = lane < J * J;
  const int i = act ? lane / J : 0, j = act ? lane % J : 0;
  const int64_t b = blockIdx.x;
  const int sij = sym(J, i, j), sii = sidx(J, i, i);
  if (act) Dm[i][j] = 0.0;
  lds_order();
  double worst = 0.0, xdm = 0.0;
  double ph, ev, xv, xd;
  auto fetch = [&](int64_t k) {   // chunk k: Phi_k, E_k; X_{k+1} and its diagonal entry X_{k+1}(i, i)
    const int64_t g = b * K + k;
    ph = Phi[g * (J * J) + i * J + j];
    ev = E[g * NS + sij];
    xv = X[(g + 1) * NS + sij];
    xd = E[g * NS + sii];
  };
  if (K >= 2) fetch(0);
  for (int64_t k = 0; k + 1 < K; ++k) {
    const double mph = ph, mev = ev, mxv = xv, mxd = xd;
    if (k + 2 < K) fetch(k + 1);
    Pm[i][j] = mph;
    xdm = fmax(xdm, mxd);   // (running maximum: behind a gap in time the state itself is ~0)
    if (act && j == 0) Xd[i] = xdm;
    lds_order();
    double y = 0.0;      // (delta Phi^T)(i, j)
#pragma unroll
    for (int l = 0; l < J; ++l) y = fma(Dm[i][l], Pm[j][l], y);
    Ym[i][j] = y;
    lds_order();
    double nx = mev;     // E + Phi delta Phi^T
#pragma unroll
    for (int q = 0; q < J; ++q) nx = fma(Pm[i][q], Ym[q][j], nx);
    const double dl = nx - mxv;
    const double scale = sqrt(fabs(Xd[i] * Xd[j]));   // |X(i, j)| <= sqrt(X(i, i) X(j, j)) for the positive semidefinite state
    const double rel = fabs(dl) / fmax(scale, 1e-300);
    worst = fmax(worst, act ? rel : 0.0);
    if (!(rel == rel)) worst = __builtin_huge_val();
    lds_order();
    if (act) Dm[i][j] = dl;
    if (act && i <= j) X[(b * K + k + 1) * NS + sij] = nx;
    lds_order();
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, kWave));
  if (lane == 0 && worst > 0.0)
    atomicMax(const_cast<unsigned long long *>(kapw) + (kNewtonMax + 2), (unsigned long long)__double_as_longlong(worst));
  worst /= newton_half_tol(gate, kapw);   // > 2 <=> above the tolerance
  if (lane == 0 && worst > 0.0) atomicMax(word, (unsigned long long)__double_as_longlong(worst));
}

// ---- the same chain in two levels, for long series (K >= kTwoLevelMin chunks) --------------------------------------------
// The update recurrence delta_{k+1} = Phi_k delta_k Phi_k^T + r_k (r_k = E_k - X_{k+1}) composes: over the steps of a
// block of kBlock chunks, delta_{k+1} = Psi_k delta_block Psi_k^T + rho_k with Psi_k = Phi_k Psi_{k-1},
// rho_k = Phi_k rho_{k-1} Phi_k^T + r_k.  (1) k_newton_block: one wavefront per block walks its steps and stores the
// prefix maps (Psi_k, rho_k); (2) k_newton_blocks: one wavefront per series walks the BLOCKS (K / kBlock steps
// instead of K); (3) k_newton_apply: one wavefront per block applies the prefix maps to the block's start update,
// X_{k+1} += delta_{k+1}, and measures the updates.
template <int J>
__global__ __launch_bounds__(kWave) void k_newton_block(int64_t K, int64_t NB, const double *__restrict__ X,
                                                        const double *__restrict__ E, const double *__restrict__ Phi,
                                                        double *__restrict__ Psi, double *__restrict__ Rho,
                                                        const unsigned long long *__restrict__ gate,
                                                        const double *__restrict__ Rsrc = nullptr) {
  // Rsrc (J x J per step): the sources r_k given directly instead of E_k - X_{k+1} (the adjoint chain)
  constexpr int NS = Dim<J>::NS;
  if (gate_closed(gate)) return;
  __shared__ double Dm[J][J + 1], Ym[J][J + 1], Pm[J][J + 1], Sm[J][J + 1];
  const int lane = threadIdx.x;
  const bool act = lane < J * J;
  const int i = act ? lane / J : 0, j = act ? lane % J : 0;
  const int64_t b = blockIdx.x / NB, nb = blockIdx.x % NB;
  const int64_t k0 = nb * kBlock, k1 = (k0 + kBlock < K - 1) ? k0 + kBlock : K - 1;   // steps k0 .. k1-1
  const int sij = sym(J, i, j);
  if (act) { Dm[i][j] = 0.0; Sm[i][j] = i == j ? 1.0 : 0.0; }
  lds_order();
  double ph, rv;
  auto fetch = [&](int64_t k) {
    const int64_t g = b * K + k;
    ph = Phi[g * (J * J) + i * J + j];
    rv = Rsrc ? Rsrc[g * (J * J) + i * J + j] : E[g * NS + sij] - X[(g + 1) * NS + sij];
  };
  fetch(k0);
  for (int64_t k = k0; k < k1; ++k) {
    const double mph = ph, mrv = rv;
    if (k + 1 < k1) fetch(k + 1);
    Pm[i][j] = mph;
    lds_order();
    double y = 0.0, ps = 0.0;   // (rho Phi^T)(i, j), (Phi Psi)(i, j)
#pragma unroll
    for (int l = 0; l < J; ++l) { y = fma(Dm[i][l], Pm[j][l], y); ps = fma(Pm[i][l], Sm[l][j], ps); }
    Ym[i][j] = y;
    lds_order();
    double rn = mrv;            // Phi rho Phi^T + r
#pragma unroll
    for (int q = 0; q < J; ++q) rn = fma(Pm[i][q], Ym[q][j], rn);
    lds_order();
    if (act) {
      Dm[i][j] = rn; Sm[i][j] = ps;
      Psi[(b * K + k) * (J * J) + i * J + j] = ps;
      Rho[(b * K + k) * (J * J) + i * J + j] = rn;
    }
    lds_order();
  }
}
template <int J>
__global__ __launch_bounds__(kWave) void k_newton_blocks(int64_t K, int64_t NB, const double *__restrict__ Psi,
                                                         const double *__restrict__ Rho, double *__restrict__ Dstart,
                                                         const unsigned long long *__restrict__ gate) {
  if (gate_closed(gate)) return;
  __shared__ double Dm[J][J + 1], Ym[J][J + 1], Pm[J][J + 1];
  const int lane = threadIdx.x;
  const bool act = lane < J * J;
  const int i = act ? lane / J : 0, j = act ? lane % J : 0;
  const int64_t b = blockIdx.x;
  if (act) Dm[i][j] = 0.0;
  lds_order();
  double ph, rv;
  auto fetch = [&](int64_t nb) {   // the maps of the block's last step
    const int64_t kl = ((nb + 1) * kBlock < K - 1 ? (nb + 1) * kBlock : K - 1) - 1;
    ph = Psi[(b * K + kl) * (J * J) + i * J + j];
    rv = Rho[(b * K + kl) * (J * J) + i * J + j];
  };
  fetch(0);
  for (int64_t nb = 0; nb < NB; ++nb) {
    const double mph = ph, mrv = rv;
    if (nb + 1 < NB) fetch(nb + 1);
    if (act) Dstart[(b * NB + nb) * (J * J) + i * J + j] = Dm[i][j];
    Pm[i][j] = mph;
    lds_order();
    double y = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) y = fma(Dm[i][l], Pm[j][l], y);
    Ym[i][j] = y;
    lds_order();
    double dn = mrv;
#pragma unroll
    for (int q = 0; q < J; ++q) dn = fma(Pm[i][q], Ym[q][j], dn);
    lds_order();
    if (act) Dm[i][j] = dn;
    lds_order();
  }
}
template <int J>
__global__ __launch_bounds__(kWave) void k_newton_apply(int64_t K, int64_t NB, double *__restrict__ X,
                                                        const double *__restrict__ E, const double *__restrict__ Psi,
                                                        const double *__restrict__ Rho, const double *__restrict__ Dstart,
                                                        const unsigned long long *__restrict__ gate,
                                                        unsigned long long *__restrict__ word,
                                                        const unsigned long long *__restrict__ kapw) {
  constexpr int NS = Dim<J>::NS;
  if (gate_closed(gate)) return;
  __shared__ double Dm[J][J + 1], Ym[J][J + 1], Pm[J][J + 1], Xd[J];
  const int lane = threadIdx.x;
  const bool act = lane < J * J;
  const int i = act ? lane / J : 0, j = act ? lane % J : 0;
  const int64_t b = blockIdx.x / NB, nb = blockIdx.x % NB;
  const int64_t k0 = nb * kBlock, k1 = (k0 + kBlock < K - 1) ? k0 + kBlock : K - 1;
  const int sij = sym(J, i, j), sii = sidx(J, i, i);
  if (act) Dm[i][j] = Dstart[(b * NB + nb) * (J * J) + i * J + j];
  lds_order();
  double worst = 0.0, xdm = 0.0, ph, rv, xv, xd;
  auto fetch = [&](int64_t k) {
    const int64_t g = b * K + k;
    ph = Psi[g * (J * J) + i * J + j];
    rv = Rho[g * (J * J) + i * J + j];
    xv = X[(g + 1) * NS + sij];
    xd = E[g * NS + sii];
  };
  // (the scale of the updates -- a running maximum of the diagonal -- starts from the chunks in front of the block)
  if (k0 > 0) xdm = fmax(xdm, E[(b * K + k0 - 1) * NS + sii]);
  if (k0 >= kBlock) xdm = fmax(xdm, E[(b * K + k0 - kBlock) * NS + sii]);
  fetch(k0);
  for (int64_t k = k0; k < k1; ++k) {
    const double mph = ph, mrv = rv, mxv = xv, mxd = xd;
    if (k + 1 < k1) fetch(k + 1);
    lds_order();   // the previous step's readers of Pm, Ym, Xd are done
    Pm[i][j] = mph;
    xdm = fmax(xdm, mxd);
    if (act && j == 0) Xd[i] = xdm;
    lds_order();
    double y = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) y = fma(Dm[i][l], Pm[j][l], y);
    Ym[i][j] = y;
    lds_order();
    double dl = mrv;   // delta_{k+1} = Psi delta_block Psi^T + rho
#pragma unroll
    for (int q = 0; q < J; ++q) dl = fma(Pm[i][q], Ym[q][j], dl);
    const double scale = sqrt(fabs(Xd[i] * Xd[j]));
    const double rel = fabs(dl) / fmax(scale, 1e-300);
    worst = fmax(worst, act ? rel : 0.0);
    if (!(rel == rel)) worst = __builtin_huge_val();
    if (act && i <= j) X[(b * K + k + 1) * NS + sij] = mxv + dl;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, kWave));
  if (lane == 0 && worst > 0.0)
    atomicMax(const_cast<unsigned long long *>(kapw) + (kNewtonMax + 2), (unsigned long long)__double_as_longlong(worst));
  worst /= newton_half_tol(gate, kapw);
  if (lane == 0 && worst > 0.0) atomicMax(word, (unsigned long long)__double_as_longlong(worst));
}

// ---- the adjoint chain in two levels ---------------------------------------------------------------------------------------
// bF_start = Phi^T bF_end + gF does not involve bS, so it is chained first (the affine two-level chain of the solve, on
// the chunks in REVERSE order: step k' = K-1-k); with bF at every chunk end known, the coupling sum_q bF_end[q] C_q + gS
// is a plain source and bS_start = Phi^T bS_end Phi + source has the form of the Newton chain.
template <int J>
__global__ void k_adj_gather(int64_t B, int64_t K, const double *__restrict__ map, double *__restrict__ PhiA,
                             double *__restrict__ gA) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST, MAPR = Dim<J>::MAPR;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * K * J * J) return;
  const int ij = (int)(g % (J * J)), i = ij / J, j = ij % J;
  const int64_t bk = g / (J * J), b = bk / K, k = bk - b * K, kr = b * K + (K - 1 - k);
  const double *m = map + bk * MAPR;
  PhiA[kr * (J * J) + ij] = m[(int64_t)j * NST + NS + i];   // Phi^T(i, j)
  if (j == 0) gA[kr * J + i] = m[(int64_t)J * NST + NS + i];
}
template <int J>
__global__ void k_adj_source(int64_t B, int64_t K, const double *__restrict__ map, const double *__restrict__ Fst,
                             double *__restrict__ R) {
  constexpr int NST = Dim<J>::NST, MAPR = Dim<J>::MAPR;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= B * K * J * J) return;
  const int ij = (int)(g % (J * J)), i = ij / J, j = ij % J, sij = sym(J, i, j);
  const int64_t bkr = g / (J * J), b = bkr / K, kr = bkr - b * K, bk = b * K + (K - 1 - kr);
  const double *m = map + bk * MAPR, *bF = Fst + bkr * J;
  double r = m[(int64_t)J * NST + sij];
#pragma unroll
  for (int q = 0; q < J; ++q) r = fma(bF[q], m[(int64_t)q * NST + sij], r);
  R[g] = r;
}
// ends[chunk] = (bS, bF) behind the chunk: the state entering step k' = K-1-chunk
template <int J>
__global__ __launch_bounds__(kWave) void k_adj_apply(int64_t K, int64_t NB, const double *__restrict__ Psi,
                                                     const double *__restrict__ Rho, const double *__restrict__ Dstart,
                                                     const double *__restrict__ Fst, double *__restrict__ ends) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST;
  __shared__ double Dm[J][J + 1], Ym[J][J + 1], Pm[J][J + 1];
  const int lane = threadIdx.x;
  const bool act = lane < J * J;
  const int i = act ? lane / J : 0, j = act ? lane % J : 0;
  const int64_t b = blockIdx.x / NB, nb = blockIdx.x % NB;
  const int64_t k0 = nb * kBlock, k1 = (k0 + kBlock < K - 1) ? k0 + kBlock : K - 1;
  const int sij = sym(J, i, j);
  if (act) Dm[i][j] = Dstart[(b * NB + nb) * (J * J) + i * J + j];
  lds_order();
  if (nb == 0) {   // behind the last chunk: nothing
    if (act && i <= j) ends[(b * K + K - 1) * NST + sij] = 0.0;
    if (act && j == 0) ends[(b * K + K - 1) * NST + NS + i] = 0.0;
  }
  double ph, rv, bf;
  auto fetch = [&](int64_t k) {
    ph = Psi[(b * K + k) * (J * J) + i * J + j];
    rv = Rho[(b * K + k) * (J * J) + i * J + j];
    bf = Fst[(b * K + k + 1) * J + i];
  };
  fetch(k0);
  for (int64_t k = k0; k < k1; ++k) {
    const double mph = ph, mrv = rv, mbf = bf;
    if (k + 1 < k1) fetch(k + 1);
    lds_order();
    Pm[i][j] = mph;
    lds_order();
    double y = 0.0;
#pragma unroll
    for (int l = 0; l < J; ++l) y = fma(Dm[i][l], Pm[j][l], y);
    Ym[i][j] = y;
    lds_order();
    double dl = mrv;
#pragma unroll
    for (int q = 0; q < J; ++q) dl = fma(Pm[i][q], Ym[q][j], dl);
    double *e = ends + (b * K + (K - 2 - k)) * NST;   // state entering step k + 1 = behind chunk K - 2 - k
    if (act && i <= j) e[sij] = dl;
    if (act && j == 0) e[NS + i] = mbf;
  }
}
template <int J>
static void adjoint_chain_two_level(int64_t B, int64_t K, const double *map, double *ends, double *scratch, hipStream_t s) {
  const size_t BK = (size_t)B * K;
  const int64_t NBa = (K + kBlock - 1) / kBlock, NBs = (K - 1 + kBlock - 1) / kBlock;
  double *PhiA = scratch, *gA = PhiA + BK * J * J, *Fst = gA + BK * J, *PsiF = Fst + BK * J, *GamF = PsiF + BK * J * J,
         *Fb = GamF + BK * J, *R = Fb + (size_t)B * NBa * J, *Psi = R + BK * J * J, *Rho = Psi + BK * J * J,
         *Dstart = Rho + BK * J * J;
  const unsigned nel = (unsigned)((BK * J * J + 255) / 256);
  hipLaunchKernelGGL((k_adj_gather<J>), dim3(nel), dim3(256), 0, s, B, K, map, PhiA, gA);
  // bF behind every chunk (reverse order): the affine chain of the solve
  hipLaunchKernelGGL((k_solve_block<J>), dim3((unsigned)(B * NBa)), dim3(kWave), 0, s, K, NBa, (const double *)PhiA,
                     (const double *)gA, PsiF, GamF);
  hipLaunchKernelGGL((k_solve_blocks<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, NBa, (const double *)PsiF,
                     (const double *)GamF, Fb);
  hipLaunchKernelGGL((k_solve_starts<J>), dim3((unsigned)((BK * J + 255) / 256)), dim3(256), 0, s, B, K, NBa,
                     (const double *)PsiF, (const double *)GamF, (const double *)Fb, Fst);
  // bS: a congruence with a known source
  hipLaunchKernelGGL((k_adj_source<J>), dim3(nel), dim3(256), 0, s, B, K, map, (const double *)Fst, R);
  hipLaunchKernelGGL((k_newton_block<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, K, NBs, (const double *)nullptr,
                     (const double *)nullptr, (const double *)PhiA, Psi, Rho, (const unsigned long long *)nullptr,
                     (const double *)R);
  hipLaunchKernelGGL((k_newton_blocks<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, NBs, (const double *)Psi,
                     (const double *)Rho, Dstart, (const unsigned long long *)nullptr);
  hipLaunchKernelGGL((k_adj_apply<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, K, NBs, (const double *)Psi,
                     (const double *)Rho, (const double *)Dstart, (const double *)Fst, ends);
}

// ---- the S workspace of factor (forward.hpp:106-123) from d, W, parallel along time -------------------------------------
// Given d, W the state obeys a linear recurrence with a diagonal transition, S' = P (S + d w^T w) P: the chunk start
// states come from k_local + k_starts (the replay of the gradient pass, z := d unused), and one lane per chunk writes its
// rows.  Row n of the workspace is the half-decayed state the reference saves between its two scalings (forward.hpp:115-
// 123): p_i (S + d w^T w)_{ij} at the flat index i + j J, row 0 zero.
template <int J>
__global__ __launch_bounds__(kWave) void k_s_rows(int64_t B, int64_t N, int64_t K, const double *__restrict__ t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *__restrict__ d,
                                                  const double *__restrict__ W, const double *__restrict__ start,
                                                  double *__restrict__ Sw) {
  constexpr int NS = Dim<J>::NS, NST = Dim<J>::NST;
  const Geo G = chunk_of(B, N, K);
  double cj[J], S[NS];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[G.b * c_bs + j];
#pragma unroll
  for (int e = 0; e < NS; ++e) S[e] = G.len > 0 ? start[G.g * NST + e] : 0.0;
  const double *tb = t + G.b * t_bs, *db = d + G.b * N, *Wb = W + G.b * N * J;
  double *sb = Sw + G.b * N * (int64_t)(J * J);
  if (G.len > 0 && G.lo == 0) {
    double zero[J];
#pragma unroll
    for (int j = 0; j < J; ++j) zero[j] = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) store_row<J>(sb + j * J, zero);
  }
  RowIn<J, false, false> cur, nxt;
  fetch_row<J, false, false>(cur, G, N, 0, tb, nullptr, nullptr, Wb, db, db);
#pragma unroll 1
  for (int r = 0; r < kRows; ++r) {
    fetch_row<J, false, false>(nxt, G, N, r + 1, tb, nullptr, nullptr, Wb, db, db);
    const int64_t n = G.lo + r;
    if (r < G.len && n + 1 < N) {
      double p[J];
#pragma unroll
      for (int j = 0; j < J; ++j) p[j] = exp_decay(-cj[j] * cur.dt);
#pragma unroll
      for (int i = 0; i < J; ++i) {
        const double dw = cur.d * cur.w[i];
#pragma unroll
        for (int j = i; j < J; ++j) S[sidx(J, i, j)] = fma(dw, cur.w[j], S[sidx(J, i, j)]);
      }
      double *o = sb + (n + 1) * (int64_t)(J * J);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        double col[J];
#pragma unroll
        for (int i = 0; i < J; ++i) col[i] = p[i] * S[sym(J, i, j)];
        store_row<J>(o + j * J, col);
      }
#pragma unroll
      for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = i; j < J; ++j) S[sidx(J, i, j)] *= p[i] * p[j];
    }
    cur = nxt;
  }
}
template <int J>
static void run_s_rows(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *d,
                       const double *W, const int32_t *flag, double *Sw, double *scratch, hipStream_t s) {
  constexpr size_t NST = Dim<J>::NST;
  const int64_t K = (N + kRows - 1) / kRows;
  const size_t BK = (size_t)B * K;
  double *loc = scratch, *start = loc + BK * NST, *llp = start + BK * NST, *ll = llp + BK, *pre = ll + B;
  const dim3 cgrid((unsigned)((B * K + kWave - 1) / kWave));
  hipLaunchKernelGGL((k_local<J>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, d, W, d, loc, llp);
  if (K >= kTwoLevelMin) {
    const int64_t NBs = (K + kWave - 1) / kWave;
    hipLaunchKernelGGL((k_starts_blocks<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, N, K, t, t_bs, c, c_bs,
                       (const double *)loc, pre);
    hipLaunchKernelGGL((k_starts_apply<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, N, K, (const double *)pre,
                       (const double *)llp, flag, start, ll);
  } else {
    hipLaunchKernelGGL((k_starts<J>), dim3((unsigned)B), dim3(kWave), 0, s, N, K, t, t_bs, c, c_bs, (const double *)loc,
                       (const double *)llp, flag, start, ll);
  }
  hipLaunchKernelGGL((k_s_rows<J>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, d, W, (const double *)start, Sw);
}
// ---- factor_rev (reverse.hpp:26-85) parallel along time ------------------------------------------------------------------
// The reverse pass of the gradient above with the adjoints of d and W handed in (EXT) instead of the log-likelihood's
// sources: the states of the rows replayed from the chunk start states (k_local + k_starts on d, W), the chunk maps
// (k_maps<J, true>: Phi_k from the homogeneous sweeps, the response to the sources from sweep J), the chain, k_final.
// Scratch: the gradient's layout (its d, W, z pieces unused) + one flag word per series.
template <int J>
static int run_factor_rev(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *U,
                          const double *V, const double *d, const double *W, const double *bde, const double *bwe, double *bt,
                          double *bc, double *ba, double *bU, double *bV, double *work, hipStream_t s) {
  const Layout L = layout<J>(B, N);
  const int64_t K = (N + kRows - 1) / kRows;
  int32_t *flag = reinterpret_cast<int32_t *>(work + L.total);
  double *ll = work + L.z;   // (unused pieces of the layout: the log-likelihood k_starts writes, the F junk)
  unsigned long long *vw = reinterpret_cast<unsigned long long *>(work + L.vw);
  if (hipMemsetAsync(flag, 0, (size_t)B * sizeof(int32_t), s) != hipSuccess) return C2_ERR_HIP;
  if (hipMemsetAsync(vw, 0, kVerifyWords * sizeof(double), s) != hipSuccess) return C2_ERR_HIP;
  const dim3 cgrid((unsigned)((B * K + kWave - 1) / kWave));
  hipLaunchKernelGGL((k_local<J>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, d, W, d, work + L.loc, work + L.llp);
  if (K >= kTwoLevelMin) {
    const int64_t NBs = (K + kWave - 1) / kWave;
    hipLaunchKernelGGL((k_starts_blocks<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, N, K, t, t_bs, c, c_bs,
                       (const double *)(work + L.loc), work + L.map);
    hipLaunchKernelGGL((k_starts_apply<J>), dim3((unsigned)(B * NBs)), dim3(kWave), 0, s, N, K,
                       (const double *)(work + L.map), (const double *)(work + L.llp), (const int32_t *)flag,
                       work + L.start, ll);
  } else {
    hipLaunchKernelGGL((k_starts<J>), dim3((unsigned)B), dim3(kWave), 0, s, N, K, t, t_bs, c, c_bs,
                       (const double *)(work + L.loc), (const double *)(work + L.llp), (const int32_t *)flag,
                       work + L.start, ll);
  }
  hipLaunchKernelGGL((k_maps<J, true>), dim3(cgrid.x, (unsigned)(J + 1)), dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, U, d,
                     W, d, work + L.map, bde, bwe);
  if (K >= kTwoLevelMin)
    adjoint_chain_two_level<J>(B, K, work + L.map, work + L.ends, work + L.sf, s);
  else
    hipLaunchKernelGGL((k_chain<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, (const double *)(work + L.map),
                       work + L.ends);
  hipLaunchKernelGGL((k_final<J, true>), cgrid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, U, V, d, W, d,
                     (const double *)(work + L.start), (const double *)(work + L.ends), (const int32_t *)flag,
                     work + L.sf, work + L.dT, work + L.bcp, ba, bU, bV, (double *)nullptr, vw, bde, bwe);
  hipLaunchKernelGGL(k_verify_combine, dim3(1), dim3(1), 0, s, vw, opt::val(opt::k_timepar_cond_limit));
  hipLaunchKernelGGL(k_finish_t, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, s, B, N,
                     (const double *)(work + L.dT), (const int32_t *)flag, bt);
  hipLaunchKernelGGL((k_finish_c<J>), dim3((unsigned)(B * J)), dim3(kWave), 0, s, K, (const double *)(work + L.bcp),
                     (const int32_t *)flag, bc);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
}  // namespace c2tg

using namespace c2tg;

// doubles of workspace of the time-parallel gradient (0: width not covered)
extern "C" size_t C2TG_NAME(c2_internal_timepar_grad_doubles)(int64_t B, int64_t N, int64_t J) {
  switch (J) {
    case 1: return layout<1>(B, N).total;
    case 2: return layout<2>(B, N).total;
    case 3: return layout<3>(B, N).total;
    case 4: return layout<4>(B, N).total;
    case 5: return layout<5>(B, N).total;
    case 6: return layout<6>(B, N).total;
    case 7: return layout<7>(B, N).total;
    case 8: return layout<8>(B, N).total;
    default: return 0;
  }
}

// Same arguments and outputs as c2_loglik_grad (t, c per series or shared by the batch: strides 0).
extern "C" int C2TG_NAME(c2_internal_loglik_grad_timepar)(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                               const double *c, int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                               double *ll, double *bt, double *bc, double *ba, double *bU, double *bV,
                                               double *by, int32_t *flag, double *work, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (J) {
    case 1: return run<1>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    case 3: return run<3>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    case 5: return run<5>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    case 7: return run<7>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    case 2: return run<2>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    case 4: return run<4>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    case 6: return run<6>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    case 8: return run<8>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, s);
    default: return C2_ERR_UNSUPPORTED;
  }
}

// factor (d, W) by Newton iterations on the chunk start states (widths 1 .. 8).  work: c2_internal_factor_iter_doubles;
// its first kNewtonHdr words are the iterations' words (updates, then conditioning) -- the caller launches its row-by-row
// kernel behind `*last_word`.
extern "C" size_t c2_internal_e8_states_doubles(int64_t B, int64_t N, int64_t R);
extern "C" int c2_internal_e8_states(int64_t B, int64_t N, int64_t R, const double *t, int64_t t_bs, const double *c,
                                     int64_t c_bs, const double *a, const double *U, const double *V, double *X, double *work,
                                     unsigned long long *guard, c2_stream_t stream);
extern "C" size_t C2TG_NAME(c2_internal_factor_iter_doubles)(int64_t B, int64_t N, int64_t J) {
  if (J < 1 || J > 8) return 0;
  const size_t K = (size_t)((N + kRows - 1) / kRows);
  size_t n = (size_t)kNewtonHdr + (size_t)B * K * (size_t)(2 * (J * (J + 1) / 2) + J * J);
  size_t extra = 0;
  if ((int64_t)K >= kTwoLevelMin) extra = (size_t)B * K * (size_t)(2 * J * J) + (size_t)B * ((K + kBlock - 1) / kBlock + 1) * (size_t)(J * J);
  if (J == 8) {   // the scan of the chunk elements overlays the two-level chains' arrays (one or the other runs)
    const size_t sc = c2_internal_e8_states_doubles(B, N, kRows);
    extra = sc > extra ? sc : extra;
  }
  return n + extra;
}
// Width 8 (round 6): the chunk start states EXACT from the scanned chunk elements (c2_timepar.hip: run8_states -- the
// elements' tree as an up-sweep, a down-sweep of `apply`s), then ONE pass for d, W.  The pass's end state of every chunk is
// compared with the start state its successor was given (k_newton_pass<J, true>: the residual a Newton iteration would start from) and
// the mismatch, in units of half of kE8CheckTol, goes into the word the caller's row-by-row kernel is gated on.
// mismatch of a chunk boundary relative to sqrt(X_ii X_jj): rounding leaves 1e-15 .. 3e-14 (kappa up to 5e3, see the floor of
// the Newton updates above); anything beyond 1e-12 is not rounding
static bool use_e8_states(int64_t B, int64_t N) {
  if (opt::has(opt::k_factor_scan8)) return opt::ival(opt::k_factor_scan8) != 0;
  return B <= 65535;
}
template <int J>
static int run_factor_iter(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                           const double *a, const double *U, const double *V, double *d, double *W, int32_t *flag,
                           double *work, const unsigned long long **last_word, hipStream_t s) {
  constexpr int NS = Dim<J>::NS;
  const int64_t K = (N + kRows - 1) / kRows;
  const int P = K < kNewtonMax ? (int)K : kNewtonMax;   // iteration p makes the first p chunks exact
  unsigned long long *words = (unsigned long long *)work;
  const size_t BK = (size_t)B * K;
  double *X = work + kNewtonHdr, *E = X + BK * NS, *Phi = E + BK * NS;
  if (hipMemsetAsync(work, 0, ((size_t)kNewtonHdr + BK * NS) * sizeof(double), s) != hipSuccess) return C2_ERR_HIP;
  const dim3 grid((unsigned)((B * K + kWave - 1) / kWave));
  if constexpr (J == 8) {
    if (K >= 2 && use_e8_states(B, N)) {
      double *scan = Phi + BK * J * J;   // behind the pass's arrays (c2_internal_factor_iter_doubles)
      const int e = c2_internal_e8_states(B, N, kRows, t, t_bs, c, c_bs, a, U, V, X, scan, words + 1, (c2_stream_t)s);
      if (e == C2_OK) {
        hipLaunchKernelGGL((k_newton_pass<J, true>), grid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, a, U, V, d, W, flag,
                           (const double *)X, E, Phi, (const unsigned long long *)nullptr, words + 1, words + (kNewtonMax + 2) + 1);
        *last_word = words + 1;
        return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
      }
      if (e != C2_ERR_UNSUPPORTED) return e;
    }
  }
  for (int p = 1; p <= P; ++p) {
    const unsigned long long *gate = p >= 2 ? words + p - 1 : nullptr;
    hipLaunchKernelGGL((k_newton_pass<J>), grid, dim3(kWave), 0, s, B, N, K, t, t_bs, c, c_bs, a, U, V, d, W, flag,
                       (const double *)X, E, Phi, gate, words + p, words + (kNewtonMax + 2) + p);
    if (K >= kTwoLevelMin) {
      const int64_t NB = (K - 1 + kBlock - 1) / kBlock;
      double *Psi = Phi + BK * J * J, *Rho = Psi + BK * J * J, *Dstart = Rho + BK * J * J;
      hipLaunchKernelGGL((k_newton_block<J>), dim3((unsigned)(B * NB)), dim3(kWave), 0, s, K, NB, (const double *)X,
                         (const double *)E, (const double *)Phi, Psi, Rho, gate);
      hipLaunchKernelGGL((k_newton_blocks<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, NB, (const double *)Psi,
                         (const double *)Rho, Dstart, gate);
      hipLaunchKernelGGL((k_newton_apply<J>), dim3((unsigned)(B * NB)), dim3(kWave), 0, s, K, NB, X, (const double *)E,
                         (const double *)Psi, (const double *)Rho, (const double *)Dstart, gate, words + p,
                         (const unsigned long long *)(words + (kNewtonMax + 2) + p));
    } else {
      hipLaunchKernelGGL((k_newton_chain<J>), dim3((unsigned)B), dim3(kWave), 0, s, K, X, (const double *)E,
                         (const double *)Phi, gate, words + p, (const unsigned long long *)(words + (kNewtonMax + 2) + p));
    }
  }
  *last_word = words + P;
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
extern "C" int C2TG_NAME(c2_internal_factor_iter)(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                       int64_t c_bs, const double *a, const double *U, const double *V, double *d,
                                       double *W, int32_t *flag, double *work, const unsigned long long **last_word,
                                       c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (J == 1) return run_factor_iter<1>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  if (J == 3) return run_factor_iter<3>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  if (J == 5) return run_factor_iter<5>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  if (J == 7) return run_factor_iter<7>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  if (J == 2) return run_factor_iter<2>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  if (J == 4) return run_factor_iter<4>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  if (J == 6) return run_factor_iter<6>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  if (J == 8) return run_factor_iter<8>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, s);
  return C2_ERR_UNSUPPORTED;
}

// Forward-only log-likelihood for small batches of long series (widths 1 .. 8): d, W by c2_factor (Newton iterations on the
// chunk start states where the dispatch takes them), z by the time-parallel solve, a reduction.
extern "C" size_t C2TG_NAME(c2_internal_loglik_wide_doubles)(int64_t B, int64_t N, int64_t J) {
  if (J < 1 || J > 8) return 0;
  const size_t K = (size_t)((N + kRows - 1) / kRows), BN = (size_t)B * N, BK = (size_t)B * K;
  return BN * (2 + J) + BK * (1 + 2 * (size_t)J * J + 3 * J) + (size_t)B * (K / kBlock + 2) * J + 8;
}
extern "C" int C2TG_NAME(c2_internal_loglik_wide)(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                       int64_t c_bs, const double *a,
                                       const double *U, const double *V, const double *y, double *ll, int32_t *flag,
                                       double *work, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (J < 1 || J > 8) return C2_ERR_UNSUPPORTED;
  const int64_t K = (N + kRows - 1) / kRows;
  const size_t BN = (size_t)B * N, BK = (size_t)B * K;
  double *d = work, *z = d + BN, *W = z + BN, *llp = W + BN * J, *Phi = llp + BK, *gk = Phi + BK * J * J, *Fst = gk + BK * J;
  if (int e = c2_factor(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, nullptr, flag, stream)) return e;
  const dim3 cgrid((unsigned)((B * K + kWave - 1) / kWave));
  if (J == 1) solve_chunks<1>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  else if (J == 2) solve_chunks<2>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  else if (J == 4) solve_chunks<4>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  else if (J == 3) solve_chunks<3>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  else if (J == 5) solve_chunks<5>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  else if (J == 6) solve_chunks<6>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  else if (J == 7) solve_chunks<7>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  else solve_chunks<8>(B, N, K, t, t_bs, c, c_bs, U, W, y, z, Phi, s);
  hipLaunchKernelGGL(k_ll_chunks, cgrid, dim3(kWave), 0, s, B, N, K, (const double *)d, (const double *)z, llp);
  hipLaunchKernelGGL(k_ll_series, dim3((unsigned)B), dim3(kWave), 0, s, N, K, (const double *)llp,
                     (const int32_t *)flag, ll);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

extern "C" size_t C2TG_NAME(c2_internal_factor_rev_timepar_doubles)(int64_t B, int64_t N, int64_t J) {
  size_t n = 0;
  switch (J) {
    case 1: n = layout<1>(B, N).total; break;
    case 2: n = layout<2>(B, N).total; break;
    case 3: n = layout<3>(B, N).total; break;
    case 4: n = layout<4>(B, N).total; break;
    case 5: n = layout<5>(B, N).total; break;
    case 6: n = layout<6>(B, N).total; break;
    case 7: n = layout<7>(B, N).total; break;
    case 8: n = layout<8>(B, N).total; break;
    default: return 0;
  }
  return n + (size_t)(B + 1) / 2 + 2;
}
extern "C" int C2TG_NAME(c2_internal_factor_rev_timepar)(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                                         const double *c, int64_t c_bs, const double *U, const double *V,
                                                         const double *d, const double *W, const double *bd,
                                                         const double *bW, double *bt, double *bc, double *ba, double *bU,
                                                         double *bV, double *work, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
#define C2TG_FR(J_) \
  case J_: return run_factor_rev<J_>(B, N, t, t_bs, c, c_bs, U, V, d, W, bd, bW, bt, bc, ba, bU, bV, work, s);
  switch (J) {
    C2TG_FR(1) C2TG_FR(2) C2TG_FR(3) C2TG_FR(4) C2TG_FR(5) C2TG_FR(6) C2TG_FR(7) C2TG_FR(8)
    default: return C2_ERR_UNSUPPORTED;
  }
#undef C2TG_FR
}
extern "C" size_t C2TG_NAME(c2_internal_s_rows_doubles)(int64_t B, int64_t N, int64_t J) {
  if (J < 1 || J > 8) return 0;
  const size_t K = (size_t)((N + kRows - 1) / kRows), BK = (size_t)B * K, NST = (size_t)(J * (J + 1) / 2 + J);
  return BK * (2 * NST + 1) + ((int64_t)K >= kTwoLevelMin ? BK * 2 * NST : 0) + (size_t)B + 8;
}
extern "C" int C2TG_NAME(c2_internal_s_rows)(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                             int64_t c_bs, const double *d, const double *W, const int32_t *flag, double *Sw,
                                             double *scratch, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (J) {
    case 1: run_s_rows<1>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    case 2: run_s_rows<2>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    case 3: run_s_rows<3>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    case 4: run_s_rows<4>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    case 5: run_s_rows<5>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    case 6: run_s_rows<6>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    case 7: run_s_rows<7>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    case 8: run_s_rows<8>(B, N, t, t_bs, c, c_bs, d, W, flag, Sw, scratch, s); break;
    default: return C2_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// solve_lower / solve_upper (every right-hand side, optional workspace) by chunk maps (long series of a small batch: the chains in two
// levels).  lower: z_n = y_n - U_n F_n, F += W_n z_n; upper: the same walk from the far end with the roles of U and W
// exchanged (internal.hpp:148-189).  Z may alias Y.  scratch: c2_internal_solve_chunks_doubles.
extern "C" size_t C2TG_NAME(c2_internal_solve_chunks_doubles)(int64_t B, int64_t N, int64_t J) {
  if (J < 1 || J > 8) return 0;
  const size_t K = (size_t)((N + kRows - 1) / kRows), BK = (size_t)B * K;
  return BK * (2 * (size_t)J * J + 3 * J) + (size_t)B * (K / kBlock + 2) * J + 8;
}
extern "C" int C2TG_NAME(c2_internal_solve_chunks)(int lower, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                                   const double *c, int64_t c_bs, const double *U, const double *W,
                                                   const double *Y, double *Z, double *scratch, c2_stream_t stream,
                                                   int64_t nrhs, double *F) {
  // nrhs columns of row-major Y, Z (and of the optional workspace F[n, k J + j]: the
  // reference's J x nrhs state is column-major, internal.hpp:127-128), one after the other: the chunk
  // propagators Phi do not depend on the column and are computed with the first one
  hipStream_t s = (hipStream_t)stream;
  const int64_t K = (N + kRows - 1) / kRows;
#define C2TG_SOLVE(J_)                                                                                                   \
  case J_:                                                                                                               \
    for (int64_t k = 0; k < nrhs; ++k) {                                                                                 \
      if (lower) solve_chunks<J_, false>(B, N, K, t, t_bs, c, c_bs, U, W, Y + k, Z + k, scratch, s, nrhs, k == 0,       \
                                         F ? F + k * J_ : nullptr, (int64_t)J_ * nrhs);                                 \
      else solve_chunks<J_, true>(B, N, K, t, t_bs, c, c_bs, W, U, Y + k, Z + k, scratch, s, nrhs, k == 0,              \
                                  F ? F + k * J_ : nullptr, (int64_t)J_ * nrhs);                                        \
    }                                                                                                                    \
    break;
  switch (J) {
    C2TG_SOLVE(1) C2TG_SOLVE(2) C2TG_SOLVE(3) C2TG_SOLVE(4) C2TG_SOLVE(5) C2TG_SOLVE(6) C2TG_SOLVE(7) C2TG_SOLVE(8)
    default: return C2_ERR_UNSUPPORTED;
  }
#undef C2TG_SOLVE
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
