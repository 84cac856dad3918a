// c2_general_tile.hip -- general_matmul_lower / general_matmul_upper (prediction products at new coordinates;
// reference forward.hpp:285-332, 346-392) with ONE WAVEFRONT PER SERIES and lanes over 64 consecutive ROWS.
//
// The reference walks a two-pointer merge of the sorted grids t1 (N output rows) and t2 (M rows feeding the state):
//     lower:  for each n:  while (t2[m] <= t1[n]) { F = p o F + V_m^T Y_m ; m++ }    p = exp(-c (t2[m] - t2[m-1]))
//                          Z_n += (U_n o exp(-c (t1[n] - t2[m-1]))) F
// (upper: the mirror image, from the far end).  The merge only decides WHICH state row an output reads; the state
// recursion itself runs over the t2 grid alone and is linear with a diagonal transition.  So a wavefront takes the t2
// rows 64 at a time -- lane i loads row s0 + i as one dense 64-byte run, the 64 states of the tile come out of an
// inclusive scan over the lanes with the operator (p_a, f_a) o (p_b, f_b) = (p_a p_b, p_b f_a + f_b) (products of decay
// factors only: nothing that can overflow), and go to LDS next to the tile's times.  The outputs whose state row lies
// in the tile are a contiguous run of the t1 grid: they are taken 64 at a time as well, lane i finds its state row by
// a binary search over the 64 times in LDS, reads that row and writes Z.  Both streams are consumed strictly in
// order, tile by tile, so each is fetched one tile ahead and nobody ever waits for a row whose address depends on the
// outcome of a comparison -- what bounded the event-per-iteration kernels (c2_general.hip: 0.9 us per event) and the
// two-phase form (c2_ops.hip: a dependent binary search in global memory per output row).
//
// Walk coordinates: position s = 0 .. M-1 along the t2 stream, q = 0 .. N-1 along t1, walk time tau = t (lower) or -t
// (upper: the walk starts at the far end).  "row s feeds output q"  <=>  tau2[s] <= tau1[q] (lower; forward.hpp:318),
// tau2[s] < tau1[q] (upper; forward.hpp:378 absorbs while t2[m] > t1[n]).
//
// Workspace semantics as the reference: F[m, j * nrhs + k] (row-major), row m written when row m is absorbed, rows the
// merge never reaches left untouched, array row 0 = V_0^T Y_0 (lower) / 0 unless absorbed (upper), the row the upper
// walk starts from (M-1) never written.
#include <cstdint>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2gt {
using namespace c2;

template <int JM, int KT>
struct RowTile {       // lane i: one row of either stream
  double tau;          // walk time (+inf beyond the end of the stream)
  double w[JM];        // V row (t2 stream) / U row (t1 stream), zero-padded to JM
  double x[KT];        // Y row / Z row (this block's right-hand sides)
  double bound;        // t2 stream: walk time of the first row of the NEXT tile (+inf if there is none); uniform
};

// Inclusive scan over the 64 lanes of (p, f) under (p_a, f_a) o (p_b, f_b) = (p_a p_b, p_b f_a + f_b), a the EARLIER
// element.  Steps 1, 2, 4, 8 inside the rows of 16 lanes (DPP row_shr: a lane without a source reads 0), then the totals
// of the rows: rows 1 and 3 take lane 15 of the row before, rows 2 and 3 take lane 31.
template <int JM, int KT>
__device__ __forceinline__ void wave_scan(double (&p)[JM], double (&f)[JM * KT], int lane) {
  constexpr int NS = JM * KT;
  auto scan_step = [&](auto shift, bool has) {
    double pp[JM], fp[NS];
#pragma unroll
    for (int j = 0; j < JM; ++j) pp[j] = shift(p[j]);
#pragma unroll
    for (int e = 0; e < NS; ++e) fp[e] = shift(f[e]);
    if (has) {
#pragma unroll
      for (int e = 0; e < NS; ++e) f[e] = fma(p[e / KT], fp[e], f[e]);
#pragma unroll
      for (int j = 0; j < JM; ++j) p[j] *= pp[j];
    }
  };
  const int lr = lane & 15;
  scan_step([](double x) { return dpp_mov<0x111>(x); }, lr >= 1);   // row_shr:1
  scan_step([](double x) { return dpp_mov<0x112>(x); }, lr >= 2);
  scan_step([](double x) { return dpp_mov<0x114>(x); }, lr >= 4);
  scan_step([](double x) { return dpp_mov<0x118>(x); }, lr >= 8);
  scan_step([&](double x) { return __shfl(x, (lane & 48) - 1, kWave); }, (lane & 16) != 0);
  scan_step([](double x) { return __shfl(x, 31, kWave); }, lane >= 32);
}

// (p, f) of the rows of one tile before the scan: row s decays the state by exp(-c (tau_s - tau_{s-1})) and adds V_s^T Y_s
template <int JM, int KT>
__device__ __forceinline__ void tile_elements(const RowTile<JM, KT> &S, double tauc, bool svalid, const double (&cj)[JM],
                                              int lane, double (&p)[JM], double (&f)[JM * KT]) {
  double tprev = __shfl_up(S.tau, 1, kWave);
  if (lane == 0) tprev = tauc;
  const double dt = svalid ? S.tau - tprev : 0.0;
#pragma unroll
  for (int j = 0; j < JM; ++j) {
    p[j] = svalid ? exp_decay(-cj[j] * dt) : 1.0;
#pragma unroll
    for (int k = 0; k < KT; ++k) f[j * KT + k] = S.w[j] * S.x[k];   // zero on rows beyond the stream
  }
}

// lane i <- row pos0 + i of a stream (walk order), as dense runs where the row length allows
template <int JM, int KT, bool LOWER>
__device__ __forceinline__ void load_rows(RowTile<JM, KT> &R, const double *tb, const double *Wb, const double *Xb,
                                          int64_t len, int64_t pos0, int J, int64_t nrhs, int kn, bool dense, bool xpair,
                                          int lane) {
  const int64_t pos = pos0 + lane;
  const bool valid = pos < len;
  const int64_t pc = valid ? pos : len - 1;
  const int64_t r = LOWER ? pc : len - 1 - pc;
  const double tv = tb[r];
  R.tau = valid ? (LOWER ? tv : -tv) : __builtin_huge_val();
  const double *wr = Wb + r * J;
  if (dense) {
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      const double2 v = *reinterpret_cast<const double2 *>(wr + j);
      R.w[j] = v.x; R.w[j + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < JM; ++j) R.w[j] = j < J ? wr[j] : 0.0;
  }
  const double *xr = Xb + r * nrhs;
  if (xpair) {
#pragma unroll
    for (int k = 0; k < KT; k += 2) {
      const double2 v = *reinterpret_cast<const double2 *>(xr + k);
      R.x[k] = v.x; R.x[k + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) R.x[k] = k < kn ? xr[k] : 0.0;
  }
  if (!valid) {
#pragma unroll
    for (int j = 0; j < JM; ++j) R.w[j] = 0.0;
  }
}

template <int JM, int KT, bool LOWER, bool WF>
__global__ __launch_bounds__(kWave) void k_general_tile(int64_t B, int64_t N, int64_t M, int J, int64_t nrhs,
                                                        const double *__restrict__ t1, int64_t t1_bs,
                                                        const double *__restrict__ t2, int64_t t2_bs,
                                                        const double *__restrict__ c, int64_t c_bs,
                                                        const double *__restrict__ U, const double *__restrict__ V,
                                                        const double *__restrict__ Y, double *Z, double *F,
                                                        int C, int64_t rpc, const double *__restrict__ carry) {
  // C > 1: the series is cut into C chunks of rpc rows of the t2 grid (a multiple of 128), one wavefront each; the
  // state a chunk starts from comes from `carry` (k_tile_chunk_maps / k_tile_chunk_carries), its outputs are those whose
  // state row lies in the chunk.
  constexpr int NS = JM * KT;       // state entries per row
  constexpr int LS = NS + 2;        // LDS row stride (doubles): 16-byte aligned, off the 64-byte bank period
  // a window of TWO tiles (ring: row s lives at s mod 128), so that 64 consecutive outputs whose state rows straddle a
  // tile boundary -- the usual case when the grids are about equally dense -- are served in one pass, not two
  // (one tile for the widest states, 32 entries per row, measured no better: registers bound the occupancy there)
  constexpr int kWin = 2 * kWave;
  __shared__ __attribute__((aligned(16))) double Ft[kWin][LS];
  __shared__ double T2[kWin];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x / C;
  const int ch = (int)(blockIdx.x % C);
  const int64_t s_lo = (int64_t)ch * rpc, s_hi = (ch == C - 1) ? M : s_lo + rpc;
  const int64_t k0 = (int64_t)blockIdx.y * KT;
  const int kn = (nrhs - k0 < KT) ? (int)(nrhs - k0) : KT;
  const double *t1b = t1 + b * t1_bs, *t2b = t2 + b * t2_bs;
  const double *Ub = U + b * N * J, *Vb = V + b * M * J;
  const double *Yb = Y + b * M * nrhs + k0;
  double *Zb = Z + b * N * nrhs + k0;
  double *Fb = WF ? F + b * M * (int64_t)J * nrhs + k0 : nullptr;
  const double inf = __builtin_huge_val();
  double cj[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) cj[j] = j < J ? c[b * c_bs + j] : 0.0;
  const bool dense = (J == JM);     // rows are 16-byte aligned runs of JM doubles
  const bool xpair = (KT % 2 == 0) && kn == KT && (nrhs % 2 == 0);

  // does a t2 row at walk time a feed an output at walk time o?
  auto feeds = [&](double a, double o) { return LOWER ? a <= o : a < o; };
  auto rowN = [&](int64_t q) { return LOWER ? q : N - 1 - q; };
  auto rowM = [&](int64_t s) { return LOWER ? s : M - 1 - s; };
  auto load_state_tile = [&](RowTile<JM, KT> &R, int64_t s0) {
    load_rows<JM, KT, LOWER>(R, t2b, Vb, Yb, M, s0, J, nrhs, kn, dense, xpair, lane);
    const int64_t sn = s0 + kWave;
    const double tv = t2b[rowM(sn < M ? sn : M - 1)];
    R.bound = sn < M ? (LOWER ? tv : -tv) : inf;
  };
  auto load_out_tile = [&](RowTile<JM, KT> &R, int64_t q0) {
    load_rows<JM, KT, LOWER>(R, t1b, Ub, (const double *)Zb, N, q0, J, nrhs, kn, dense, xpair, lane);
  };

  // the first output of this chunk: the first one fed by the chunk's first row (64-ary search by the wavefront)
  int64_t q_lo = 0;
  if (ch > 0) {
    const double ts = LOWER ? t2b[s_lo] : -t2b[M - 1 - s_lo];
    int64_t lo = 0, hi = N;
    while (lo < hi) {
      const int64_t width = hi - lo;
      const int64_t pos = lo + ((int64_t)lane * width) / kWave;      // probes lo = pos_0 <= pos_1 <= ... < hi
      const double tq = LOWER ? t1b[pos] : -t1b[N - 1 - pos];
      const int nfalse = __popcll(__ballot(!feeds(ts, tq)));         // the predicate is monotone along the probes
      if (nfalse == 0) { hi = lo; break; }
      const int64_t plast = __shfl((long long)pos, nfalse - 1, kWave);   // last probe not fed
      const int64_t pnext = nfalse < kWave ? __shfl((long long)pos, nfalse, kWave) : hi;
      lo = plast + 1;
      hi = pnext > lo ? pnext : lo;
    }
    q_lo = lo;
    if (q_lo >= N) return;   // every output is served by earlier chunks: nothing absorbed here
  }

  RowTile<JM, KT> Sc, Sn, Oc, On;
  load_state_tile(Sc, s_lo);
  load_out_tile(Oc, q_lo);
  load_state_tile(Sn, s_lo + kWave);
  load_out_tile(On, q_lo + kWave);

  // the state carried into a tile is the last row of the previous one, still in LDS when it is needed; before the first
  // tile: zero / the chunk's carry.  The rows before the first tile "feed" every output (time -inf) -- no output of
  // this chunk has its state row there.
#pragma unroll
  for (int e = 0; e < NS; e += 2) *reinterpret_cast<double2 *>(&Ft[kWin - kWave + lane][e]) = make_double2(0.0, 0.0);
  T2[kWin - kWave + lane] = -inf;
  if (ch > 0 && lane < NS)
    Ft[kWin - 1][lane] = carry[((b * gridDim.y + blockIdx.y) * C + ch) * NS + lane];
  lds_order();
  // walk time of the carried state (the first row of the series decays by exp(0))
  double tauc = ch > 0 ? (LOWER ? t2b[s_lo - 1] : -t2b[M - s_lo]) : __shfl(Sc.tau, 0, kWave);
  int64_t s0 = s_lo, q = q_lo, q0 = q_lo;  // tile origins; q = outputs consumed so far
  int64_t s_end = -1;             // state row of the last output, once known
  bool finished = false;

  while (!finished && s0 < s_hi) {
    // ---- the 64 states of the tile: inclusive scan over the lanes -------------------------------------------------
    const bool svalid = s0 + lane < M;
    double p[JM], f[NS];
    tile_elements<JM, KT>(Sc, tauc, svalid, cj, lane, p, f);
    wave_scan<JM, KT>(p, f, lane);
    {   // F_s = P_s o F_carry + f_s, s = s0 + lane
      double Fc[NS];
#pragma unroll
      for (int e = 0; e < NS; e += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(&Ft[(s0 - 1) & (kWin - 1)][e]);
        Fc[e] = v.x; Fc[e + 1] = v.y;
      }
#pragma unroll
      for (int e = 0; e < NS; ++e) f[e] = fma(p[e / KT], Fc[e], f[e]);
    }
    lds_order();   // the readers of the tile before the previous one are done
    const int slot = (int)(s0 & (kWin - 1));
#pragma unroll
    for (int e = 0; e < NS; e += 2) *reinterpret_cast<double2 *>(&Ft[slot + lane][e]) = make_double2(f[e], f[e + 1]);
    T2[slot + lane] = Sc.tau;
    lds_order();
    const double bound = Sc.bound;
    const int woff = (slot + kWave) & (kWin - 1);   // ring position of the window's first row, s0 - (kWin - 64)

    // ---- outputs ----------------------------------------------------------------------------------------------------
    // serve the lanes in `mine` from the window; returns the state row (walk position) of lane `last`
    auto serve = [&](bool mine, int last) {
      int pos = 0;                              // number of window rows feeding this output
#pragma unroll
      for (int step = kWin / 2; step >= 1; step >>= 1)
        if (feeds(T2[(pos + step - 1 + woff) & (kWin - 1)], Oc.tau)) pos += step;
      if (feeds(T2[(pos + woff) & (kWin - 1)], Oc.tau)) pos += 1;   // reaches kWin only on lanes that are not `mine`
      const int64_t srow = s0 - (kWin - kWave) + pos - 1;   // < 0: an output ahead of the first t2 row (forward.hpp:303-306)
      if (mine && srow >= 0) {
        const int ri = (pos - 1 + woff) & (kWin - 1);
        const double dq = Oc.tau - T2[ri];
        double z[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) z[k] = Oc.x[k];
#pragma unroll
        for (int j = 0; j < JM; ++j) {
          const double w = Oc.w[j] * exp_decay(-cj[j] * dq);
#pragma unroll
          for (int k = 0; k < KT; ++k) z[k] = fma(w, Ft[ri][j * KT + k], z[k]);
        }
        double *zr = Zb + rowN(q0 + lane) * nrhs;
        if (xpair) {
#pragma unroll
          for (int k = 0; k < KT; k += 2) *reinterpret_cast<double2 *>(zr + k) = make_double2(z[k], z[k + 1]);
        } else {
#pragma unroll
          for (int k = 0; k < KT; ++k)
            if (k < kn) zr[k] = z[k];
        }
      }
      return (int64_t)__shfl((long long)srow, last, kWave);
    };
    while (true) {
      // every output left in the tile has its state row in the window or beyond it (those further back were served
      // before their rows left the window).  If the tile's LAST output does not reach beyond, serve the whole rest.
      const int done = (int)(q - q0);          // lanes of the output tile already served
      const int nv = (N - q0 < kWave) ? (int)(N - q0) : kWave;
      const bool all_in = !feeds(bound, __shfl(Oc.tau, nv - 1, kWave));
      if (!all_in) break;
      const int64_t sl = serve(lane >= done && lane < nv, nv - 1);
      q = q0 + nv;
      if (q == N) {                             // the last output fixes the last row the merge absorbs
        finished = true;
        s_end = sl;
        break;
      }
      Oc = On;
      q0 += kWave;
      load_out_tile(On, q0 + kWave);
    }
    if (!finished) {
      // outputs whose state row lies in the tile of the window that is overwritten next (the older one of two: those
      // not fed by this tile's first row; the only one: those not fed by the next tile's first row)
      const int done = (int)(q - q0);
      const bool mine = lane >= done && q0 + lane < N && !feeds(kWin > kWave ? T2[slot] : bound, Oc.tau);
      const int cnt = __popcll(__ballot(mine));
      if (cnt > 0) {
        (void)serve(mine, 0);
        q += cnt;
      }
    }
    if (!finished && s0 + kWave >= s_hi && s_hi < M) {
      // last tile of a chunk: the outputs left whose state row lies in it (a prefix of the current output tile)
      const int done = (int)(q - q0);
      const bool mine = lane >= done && q0 + lane < N && !feeds(bound, Oc.tau);
      if (__ballot(mine) != 0ull) (void)serve(mine, 0);
    }
    if (!finished) s_end = s0 + kWave - 1;      // outputs remain: they absorb every row of this tile

    if (WF) {   // rows the merge absorbed (row 0 of the walk: stored by the lower variant only)
      const int64_t s = s0 + lane;
      if (svalid && (s == 0 ? LOWER : s <= s_end)) {
        double *fr = Fb + rowM(s) * (int64_t)J * nrhs;
        if (dense && nrhs == KT) {   // the whole row in one run, laid out as the registers are
#pragma unroll
          for (int e = 0; e < NS; e += 2) *reinterpret_cast<double2 *>(fr + e) = make_double2(f[e], f[e + 1]);
        } else
#pragma unroll
        for (int j = 0; j < JM; ++j) {
          if (j < J) {
            if (xpair) {
#pragma unroll
              for (int k = 0; k < KT; k += 2)
                *reinterpret_cast<double2 *>(fr + (int64_t)j * nrhs + k) = make_double2(f[j * KT + k], f[j * KT + k + 1]);
            } else {
#pragma unroll
              for (int k = 0; k < KT; ++k)
                if (k < kn) fr[(int64_t)j * nrhs + k] = f[j * KT + k];
            }
          }
        }
      }
    }
    // ---- next tile (the carry stays in LDS) -----------------------------------------------------------------------
    tauc = T2[slot + kWave - 1];
    s0 += kWave;
    Sc = Sn;
    load_state_tile(Sn, s0 + kWave);
  }
  // F.row(0).setZero() (forward.hpp:358) unless the walk absorbed the row stored there (its last one, M >= 2)
  if (WF && !LOWER && ch == 0 && lane == 0 && !(M >= 2 && feeds(-t2b[0], -t1b[0]))) {
    for (int j = 0; j < J; ++j)
      for (int k = 0; k < kn; ++k) Fb[(int64_t)j * nrhs + k] = 0.0;
  }
}

// ---- chunks (small batches of long series) ----------------------------------------------------------------------------
// The state recursion is linear: a chunk of rows acts on the state it starts from as F -> P o F + f.  One wavefront per
// chunk composes (P, f) of its rows (the same tile scan, without outputs); one wavefront per series then scans the
// chunks, which gives every chunk the state it starts from.  maps: [series][rhs tile][chunk][JM + NS].
template <int JM, int KT, bool LOWER>
__global__ __launch_bounds__(kWave) void k_tile_chunk_maps(int64_t B, int64_t M, int J, int64_t nrhs,
                                                           const double *__restrict__ t2, int64_t t2_bs,
                                                           const double *__restrict__ c, int64_t c_bs,
                                                           const double *__restrict__ V, const double *__restrict__ Y,
                                                           int C, int64_t rpc, double *__restrict__ maps) {
  constexpr int NS = JM * KT;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x / C;
  const int ch = (int)(blockIdx.x % C);
  if (ch == C - 1) return;   // nobody starts behind the last chunk
  const int64_t s_lo = (int64_t)ch * rpc, s_hi = s_lo + rpc;   // < M: only the last chunk is ragged
  const int64_t k0 = (int64_t)blockIdx.y * KT;
  const int kn = (nrhs - k0 < KT) ? (int)(nrhs - k0) : KT;
  const double *t2b = t2 + b * t2_bs, *Vb = V + b * M * J, *Yb = Y + b * M * nrhs + k0;
  double cj[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) cj[j] = j < J ? c[b * c_bs + j] : 0.0;
  const bool dense = (J == JM), xpair = (KT % 2 == 0) && kn == KT && (nrhs % 2 == 0);
  double P[JM], Fa[NS];
#pragma unroll
  for (int j = 0; j < JM; ++j) P[j] = 1.0;
#pragma unroll
  for (int e = 0; e < NS; ++e) Fa[e] = 0.0;
  double tauc = LOWER ? t2b[s_lo > 0 ? s_lo - 1 : 0] : -t2b[s_lo > 0 ? M - s_lo : M - 1];
  RowTile<JM, KT> Sc, Sn;
  load_rows<JM, KT, LOWER>(Sc, t2b, Vb, Yb, M, s_lo, J, nrhs, kn, dense, xpair, lane);
  for (int64_t s0 = s_lo; s0 < s_hi; s0 += kWave) {
    load_rows<JM, KT, LOWER>(Sn, t2b, Vb, Yb, M, s0 + kWave, J, nrhs, kn, dense, xpair, lane);
    double p[JM], f[NS];
    tile_elements<JM, KT>(Sc, tauc, true, cj, lane, p, f);
    wave_scan<JM, KT>(p, f, lane);
#pragma unroll
    for (int e = 0; e < NS; ++e) Fa[e] = fma(__shfl(p[e / KT], kWave - 1, kWave), Fa[e], __shfl(f[e], kWave - 1, kWave));
#pragma unroll
    for (int j = 0; j < JM; ++j) P[j] *= __shfl(p[j], kWave - 1, kWave);
    tauc = __shfl(Sc.tau, kWave - 1, kWave);
    Sc = Sn;
  }
  double *out = maps + ((b * gridDim.y + blockIdx.y) * C + ch) * (JM + NS);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < JM; ++j) out[j] = P[j];
#pragma unroll
    for (int e = 0; e < NS; ++e) out[JM + e] = Fa[e];
  }
}

// lane <-> chunk (C <= 64): carry[series][rhs tile][chunk][NS] = state before the chunk's first row
template <int JM, int KT>
__global__ __launch_bounds__(kWave) void k_tile_chunk_carries(int C, const double *__restrict__ maps,
                                                              double *__restrict__ carry) {
  constexpr int NS = JM * KT;
  const int lane = threadIdx.x;
  const int64_t sy = (int64_t)blockIdx.x * gridDim.y + blockIdx.y;
  const double *in = maps + (sy * C + lane) * (JM + NS);
  double p[JM], f[NS];
  const bool have = lane < C - 1;
#pragma unroll
  for (int j = 0; j < JM; ++j) p[j] = have ? in[j] : 1.0;
#pragma unroll
  for (int e = 0; e < NS; ++e) f[e] = have ? in[JM + e] : 0.0;
  wave_scan<JM, KT>(p, f, lane);
  if (have) {
    double *out = carry + (sy * C + lane + 1) * NS;
#pragma unroll
    for (int e = 0; e < NS; ++e) out[e] = f[e];
  }
}

}  // namespace c2gt

using namespace c2gt;

// Chunks pay when the batch alone leaves most of the chip idle and the series are long enough to cut.
static int tile_chunks(int64_t B, int64_t M) {
  if (opt::has(opt::k_general_chunks) && opt::ival(opt::k_general_chunks) == 0) return 1;   // never; otherwise the automatic choice
  if (B >= 512 || M < 2048) return 1;
  int64_t C = (2048 + B - 1) / B;                // ~2 wavefronts per SIMD-quarter of the chip in flight
  if (C > 64) C = 64;
  if (C > M / 512) C = M / 512;                  // at least 4 passes of 128 rows per chunk
  return C < 2 ? 1 : (int)C;
}
// doubles of stream-ordered scratch the chunked form needs (0: not chunked)
extern "C" size_t c2_internal_general_tile_doubles(int64_t B, int64_t M, int64_t J, int64_t nrhs) {
  if (J > 16) return 0;
  const int JM = J <= 4 ? 4 : (J <= 8 ? 8 : 16);
  if (nrhs > (JM == 16 ? 2 : opt::ival(opt::k_general_tile_max_rhs)) && B >= 512) return 0;
  const int KT = nrhs == 1 ? 1 : ((nrhs == 2 || JM == 16) ? 2 : 4);
  const int C = tile_chunks(B, M);
  if (C < 2) return 0;
  const size_t ytiles = (size_t)((nrhs + KT - 1) / KT);
  return (size_t)B * ytiles * C * (JM + 2 * JM * KT);
}

// lower != 0: general_matmul_lower, else upper; Z is accumulated into (the caller zeroes it when asked to).  `scratch`
// (c2_internal_general_tile_doubles; may be null: one wavefront per series then) enables the chunked form.  Returns
// C2_ERR_UNSUPPORTED for shapes the mapping does not cover.
extern "C" int c2_internal_general_tile(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs,
                                        const double *t1, int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c,
                                        int64_t c_bs, const double *U, const double *V, const double *Y, double *Z,
                                        double *F, double *scratch, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (J > 16 || B > 0x3fffffLL) return C2_ERR_UNSUPPORTED;   // wider rows: four row tiles no longer fit the registers
  const int JM = J <= 4 ? 4 : (J <= 8 ? 8 : 16);
  // one pass over the rows per tile of KT right-hand sides: beyond one tile the lanes-over-right-hand-sides kernel
  // (c2_general.hip) does less redundant work (nrhs = 8: 7.2 ms either way, 19.6 against 11.5 ms with the F rows) --
  // unless the batch is small: that kernel walks each series row by row (one series of 1e5 rows, nrhs = 8: 99 ms)
  if (nrhs > (JM == 16 ? 2 : opt::ival(opt::k_general_tile_max_rhs)) && B >= 512) return C2_ERR_UNSUPPORTED;
  const int KT = nrhs == 1 ? 1 : ((nrhs == 2 || JM == 16) ? 2 : 4);
  const int64_t ytiles = (nrhs + KT - 1) / KT;
  int C = scratch ? tile_chunks(B, M) : 1;
  int64_t rpc = M;
  if (C > 1) {
    rpc = ((M + C - 1) / C + 127) / 128 * 128;
    C = (int)((M + rpc - 1) / rpc);
  }
  const int NS = JM * KT;
  double *maps = scratch, *carry = scratch ? scratch + (size_t)B * ytiles * C * (JM + NS) : nullptr;
  const dim3 grid((unsigned)(B * C), (unsigned)ytiles);
#define C2_GT2(JM_, KT_, LO, WF_)                                                                                   \
  hipLaunchKernelGGL((k_general_tile<JM_, KT_, LO, WF_>), grid, dim3(kWave), 0, s, B, N, M, (int)J, nrhs, t1, t1_bs, \
                     t2, t2_bs, c, c_bs, U, V, Y, Z, F, C, rpc, (const double *)carry)
#define C2_GT1(JM_, KT_)                                                                                              \
  do {                                                                                                                \
    if (C > 1) {                                                                                                      \
      if (lower) hipLaunchKernelGGL((k_tile_chunk_maps<JM_, KT_, true>), grid, dim3(kWave), 0, s, B, M, (int)J, nrhs, t2, t2_bs, c, c_bs, V, Y, C, rpc, maps);  \
      else hipLaunchKernelGGL((k_tile_chunk_maps<JM_, KT_, false>), grid, dim3(kWave), 0, s, B, M, (int)J, nrhs, t2, t2_bs, c, c_bs, V, Y, C, rpc, maps);      \
      hipLaunchKernelGGL((k_tile_chunk_carries<JM_, KT_>), dim3((unsigned)B, (unsigned)ytiles), dim3(kWave), 0, s, C, (const double *)maps, carry);          \
    }                                                                                                                 \
    if (lower) { if (F) C2_GT2(JM_, KT_, true, true); else C2_GT2(JM_, KT_, true, false); }                           \
    else       { if (F) C2_GT2(JM_, KT_, false, true); else C2_GT2(JM_, KT_, false, false); }                         \
  } while (0)
#define C2_GT(JM_)                                  \
  do {                                              \
    if (KT == 1) C2_GT1(JM_, 1);                    \
    else if (KT == 2) C2_GT1(JM_, 2);               \
    else C2_GT1(JM_, 4);                            \
  } while (0)
  switch (JM) {
    case 4: C2_GT(4); break;
    case 8: C2_GT(8); break;
    default:
      if (KT == 1) C2_GT1(16, 1);
      else C2_GT1(16, 2);
      break;
  }
#undef C2_GT
#undef C2_GT1
#undef C2_GT2
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
