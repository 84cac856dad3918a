// c2_loglik_k2.hip -- fused log-likelihood + reverse-mode gradient with TWO LANES PER SERIES (J = 8), for batches that
// give the one-lane mapping of c2_loglik_t.hip only half a chip (24576 ... 49151 series: the 2-GPU shard of BASELINE
// configs[2] is 32768).
//
// One lane per series costs a wavefront of 64 series ~1070 VALU instructions per reverse step, ~610 of them the pass over the
// 36 packed elements of the J x J states S and M; below 65536 series SIMDs stand empty and the step time is that instruction
// count.  Here a series is walked by a PAIR of lanes (32 series per wavefront, 1024 wavefronts at 32768 series).  Both lanes
// hold every width-8 vector and every recursion scalar (same values, same order of operations: bit-identical), and split the
// packed pass.  To give the two lanes ONE instruction stream on different data, lane h = 1 keeps all its vectors ROTATED by
// four (local index i <-> global (i + 4h) mod 8; a row is simply read from its tile with the 16-byte pieces in rotated order),
// and both own the same LOCAL set of 20 index pairs:
//     {(i, j): i <= j < 4}  u  {(i, 4 + m): i < m}  u  {(i, 4 + i)}        (10 + 6 + 4)
// Under the rotation the first 16 of lane 1 are exactly the pairs lane 0 does not have; the four pairs (i, 4 + i) map onto
// themselves, are kept by BOTH lanes and enter every sum over the elements with weight 1/2.  Sums over the elements
// (tau = u S; xs = x S, diag(S M), q = w M in the reverse step) are completed by ONE exchange with the partner lane per
// vector -- own[j] + partner[(j + 4) mod 8], a DPP quad permute -- and dot products of width-8 vectors are formed as
// (own half over local 0..3) + (the partner's half), which is the same two numbers added in either lane.  Where a lane only
// ever uses a half of a vector (the solve state F and its cotangent, the half rows of the outputs, the accumulators of bc)
// only that half is kept; w (forward) and bV (reverse), needed whole by the pass over the elements, are computed by halves
// and completed from the partner, the decay factors and their inverses likewise.
//
// Everything else is the one-lane design: rows through LDS transposes in aligned 128-byte lines, lane-major records of W,
// (d, z), t and a checkpoint every 32 rows written by the forward pass, the reverse sweep running the recursion BACKWARD from
// the checkpoints (reference steps: forward.hpp:105-134, internal.hpp:135-145, 225-245, reverse.hpp:52-84; the fused reverse
// step is derived in c2_loglik.hip).  With 20 instead of 36 elements per lane S lives in ordinary registers, M in LDS.
// Gaps in time are re-anchored as there: an extra checkpoint in front of a gap the backward recursion could not cross (up to
// twice the regular number per wavefront); a wavefront that runs out of them marks its group of 64 series and the replay
// kernels take those.  Which rows carry a checkpoint is the sign of their d record; the slots are consumed in reverse order of
// writing, so the reverse sweep needs their number and no list.
// Not here (the dispatch keeps the other mappings for them): widths other than 8.
//
// Coefficient-level form (TM != 0; SURVEY.md section 8f-1, driver.cpp:456-474): the rows U_n, V_n are formed in the lanes from
// (ar, cr, ac, bc, cc, dc) and x_n and no width-8 array is read or written.  A lane's local 0..3 are the global columns
// 4h .. 4h+3, i.e. two SLOTS of two columns each; J = Jr + 2 Jc = 8 makes Jr even, so a slot is either one complex term
// (cos / sin columns) or a pair of real terms, and a real pair is evaluated by the SAME instructions as a complex term with
// dc = 0 (cos = 1, sin = 0 exactly), ac = ar_0, bc = -ar_1: U = (ar_0, ar_1) bit for bit, V's second column selected to 1.
// One sincos per slot and step; the partner's half of U arrives by the same DPP move as its half of W.  The reverse sweep
// folds the reverse of the recipe (k_terms_rev of c2_terms.hip) into its step: per slot three running sums (bac, bbc, bdc --
// for a real pair the first two are sum bU_0 and -sum bU_1), the sum of ba, and bx_n = bt_n + sum_k g_nk dc_k.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/celerite2_amd.h"
#include "c2_loglik_helpers.hpp"

namespace c2k2 {
using namespace c2;

// Streaming hints (C2K2_NT; see C2T_NT in c2_loglik_t.hip): every byte of this pair is touched once.  With the scalar gradients leaving as
// whole lines (STR = 16 below) nothing has to SURVIVE in L2 here, so the hint is worth less than in the one-lane pair.
#ifndef C2K2_NT
#define C2K2_NT 0
#endif
typedef double k2d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2s(const double2 *p) {
#if C2K2_NT
  const k2d2v v = __builtin_nontemporal_load(reinterpret_cast<const k2d2v *>(p));
  return make_double2(v.x, v.y);
#else
  return *p;
#endif
}
__device__ __forceinline__ double ld1s(const double *p) {
#if C2K2_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st2s(double2 *p, double2 v) {
#if C2K2_NT
  k2d2v w; w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<k2d2v *>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st1s(double *p, double v) {
#if C2K2_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

constexpr int J = 8, NL = 20, SPW = 32, C = 32, ST = 8;
constexpr int RSTR = 2 * J;       // LDS stride (doubles) of a series in a two-row tile: one 128-byte line, its eight 16-byte
                                  // pieces XOR-swizzled by the series (swz) -- see row_read
constexpr int SSTR = ST + 1;      // ... in an eight-row scalar tile: 72 B
constexpr int CKD = NL + 4;       // doubles a lane keeps per checkpoint: its 20 elements of S and its four of F
constexpr double kGuard = kBackwardGuard;
// the local element set (see above); elements 16 .. 19 carry weight 1/2 in sums over the elements
__device__ constexpr int LI[NL] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3, 0, 0, 0, 1, 1, 2, 0, 1, 2, 3};
__device__ constexpr int LJ[NL] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3, 5, 6, 7, 6, 7, 7, 4, 5, 6, 7};
constexpr int kHalfFrom = 16;

struct Rec {
  size_t w, dz, t, ck, cnt, total;   // offsets / total in doubles
  int64_t nreg, nck;                 // regular checkpoints per wavefront (rows C, 2C, ..., and the last row); slots
};
__host__ __device__ inline int64_t n_ckpt(int64_t N) { return N >= 2 ? (N - 2) / C + 1 : 0; }
__host__ inline Rec rec_layout(int64_t B, int64_t N) {
  const size_t waves = ((size_t)B + SPW - 1) / SPW;
  Rec r;
  r.nreg = n_ckpt(N);
  r.nck = 3 * r.nreg;
  r.w = 0;                                           // [wave][n][4 pieces][32] double2
  r.dz = r.w + waves * (size_t)N * J * SPW;          // [wave][n][32] double2
  r.t = r.dz + waves * (size_t)N * 2 * SPW;          // [wave][n][32]
  r.ck = r.t + waves * (size_t)N * SPW;              // [wave][slot][CKD][64]
  r.cnt = r.ck + waves * (size_t)r.nck * CKD * kWave;   // [wave]: checkpoints written (an integer in a double's place)
  r.total = r.cnt + ((waves + 1) & ~(size_t)1);
  return r;
}

__device__ __forceinline__ double swap_pair(double x) { return dpp_mov<kDppXor1>(x); }   // the partner lane's value
// own[j] + partner[j + 4], j = 0..3: the sum over ALL elements from the two lanes' partial sums, for this lane's local 0..3
// (which are the partner's local 4..7)
__device__ __forceinline__ void pair_total4(const double (&own)[J], double (&tot)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) tot[j] = own[j] + swap_pair(own[j + 4]);
}
// a dot product of two width-8 vectors from its half over this lane's local 0..3: (own half) + (partner's half), the same two
// numbers in either lane
__device__ __forceinline__ double dot_halves(const double (&x)[4], const double *y) {
  double a = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) a = fma(x[i], y[i], a);
  return a + swap_pair(a);
}

// ---- tile movers (32 series per wavefront) -------------------------------------------------------------------------------
// two-row tile of a width-8 array: one instruction moves 8 series x 128 bytes (lane l: series 8 i + l / 8, piece l % 8)
// (`last`: the last series of the wavefront that exists -- fetches of the others read it instead, flushes skip them)
template <bool FULL>
__device__ __forceinline__ int clamp_series(int sr, int last) { return (FULL || sr < last) ? sr : last; }
template <bool FULL>
__device__ __forceinline__ void row_fetch(const double *__restrict__ base, int64_t N, int64_t n0, int lane, int last,
                                          double (&st)[8]) {
  int64_t r = n0 + (lane & 7) / 4;
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
  const int64_t off = r * J + 2 * (lane & 3);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double2 v = ld2s(reinterpret_cast<const double2 *>(base + (int64_t)clamp_series<FULL>(8 * i + lane / 8, last) * N * J + off));
    st[2 * i] = v.x; st[2 * i + 1] = v.y;
  }
}
// Swizzle of a series' line: piece w = 4 r + p (row r, 16-byte piece p) lives at slot w ^ swz(series).  A ds_read_b128
// is served in groups of 16 lanes -- eight series x the two lanes of a pair, which read pieces p and p ^ 2 of the same row --
// over 16 slots of 16 bytes (64 banks): with 128-byte lines the eight series of a group are four at slot offset 0 and four
// at 8, and bits 1, 2 of the series index, which enumerate either four, flip the piece's low bit and the row: every pair
// lands on its own two slots.  (Stride 144 B, the one-lane kernels' padding: SQ_LDS_BANK_CONFLICT was 70 % of the LDS cycles.)
__device__ __forceinline__ int swz(int series) { return ((series >> 1) & 1) | (((series >> 2) & 1) << 2); }
__device__ __forceinline__ void row_stage(double *tile, int lane, const double (&st)[8]) {
  const int w = (lane & 7) ^ swz(lane >> 3);   // (series 8 i + lane / 8: the swizzle does not depend on i)
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<double2 *>(tile + (8 * i + lane / 8) * RSTR + 2 * w) = make_double2(st[2 * i], st[2 * i + 1]);
}
// this lane's row r of its series, in LOCAL order (pieces rotated by two for the odd lane)
__device__ __forceinline__ void row_read(const double *tile, int sl, int h, int r, double (&x)[J]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double2 v = *reinterpret_cast<const double2 *>(tile + sl * RSTR + 2 * ((4 * r + ((q + 2 * h) & 3)) ^ swz(sl)));
    x[2 * q] = v.x; x[2 * q + 1] = v.y;
  }
}
// local elements 0..3 (= global 4h .. 4h+3) of a width-8 row into the tile: the two lanes of a pair write the row together
__device__ __forceinline__ void row_write_half(double *tile, int sl, int h, int r, const double *x) {
  *reinterpret_cast<double2 *>(tile + sl * RSTR + 2 * ((4 * r + 2 * h) ^ swz(sl))) = make_double2(x[0], x[1]);
  *reinterpret_cast<double2 *>(tile + sl * RSTR + 2 * ((4 * r + 2 * h + 1) ^ swz(sl))) = make_double2(x[2], x[3]);
}
// tile -> memory: rows n0, n0 + 1 of every series (rows outside [0, N-1] skipped)
template <bool FULL>
__device__ __forceinline__ void row_flush(double *__restrict__ base, int64_t N, int64_t n0, const double *tile, int lane,
                                          int last) {
  const int64_t r = n0 + (lane & 7) / 4;
  double2 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = *reinterpret_cast<const double2 *>(tile + (8 * i + lane / 8) * RSTR + 2 * ((lane & 7) ^ swz(lane >> 3)));
  if (r >= 0 && r < N) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (FULL || 8 * i + lane / 8 <= last)
        st2s(reinterpret_cast<double2 *>(base + (int64_t)(8 * i + lane / 8) * N * J + r * J + 2 * (lane & 3)), v[i]);
  }
}
// per-series scalar streams in 16-row tiles: one instruction moves 4 series x 128 bytes
template <bool FULL>
__device__ __forceinline__ void sc_fetch16(const double *__restrict__ base, int64_t sN, int64_t N, int64_t n16, int lane,
                                           int last, double (&st)[8]) {
  int64_t r = n16 + (lane & 15);
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
#pragma unroll
  for (int i = 0; i < 8; ++i) st[i] = ld1s(&base[(int64_t)clamp_series<FULL>(4 * i + lane / 16, last) * sN + r]);
}
__device__ __forceinline__ void sc_stage16(double *tile, int lane, int half, const double (&st)[8]) {
  if (((lane & 15) >> 3) == half) {
#pragma unroll
    for (int i = 0; i < 8; ++i) tile[(4 * i + lane / 16) * SSTR + (lane & 7)] = st[i];
  }
}
// eight-row scalar tile -> memory: one instruction moves 8 series x 64 bytes
template <bool FULL>
__device__ __forceinline__ void sc_flush(double *__restrict__ base, int64_t N, int64_t n0, const double *tile, int lane,
                                         int last) {
  const int64_t r = n0 + (lane & 7);
  double v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = tile[(8 * i + lane / 8) * SSTR + (lane & 7)];
  if (r >= 0 && r < N) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (FULL || 8 * i + lane / 8 <= last) base[(int64_t)(8 * i + lane / 8) * N + r] = v[i];
  }
}

// p_j = exp(c_j dt) in local order: each lane evaluates its local 0..3 and takes 4..7 -- the partner's 0..3, the same
// numbers it would have computed -- from the partner.  PAIRED: c_{2k} == c_{2k+1} for every series of the wavefront.
template <bool PAIRED>
__device__ __forceinline__ void decay(const double (&c)[J], double dt, double (&p)[J]) {
  if constexpr (PAIRED) {
    const double a = exp_decay(c[0] * dt), b = exp_decay(c[2] * dt);
    const double a2 = swap_pair(a), b2 = swap_pair(b);
    p[0] = p[1] = a; p[2] = p[3] = b; p[4] = p[5] = a2; p[6] = p[7] = b2;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = exp_decay(c[j] * dt);
#pragma unroll
    for (int j = 0; j < 4; ++j) p[4 + j] = swap_pair(p[j]);
  }
}
// 1 / p_j the same way
template <bool PAIRED>
__device__ __forceinline__ void inverses(const double (&p)[J], double (&ip)[J]) {
  if constexpr (PAIRED) {
    const double a = rcp_nr(p[0]), b = rcp_nr(p[2]);
    const double a2 = swap_pair(a), b2 = swap_pair(b);
    ip[0] = ip[1] = a; ip[2] = ip[3] = b; ip[4] = ip[5] = a2; ip[6] = ip[7] = b2;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) ip[j] = rcp_nr(p[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ip[4 + j] = swap_pair(ip[j]);
  }
}

// ---- rows generated from the celerite coefficients (driver.cpp:456-474; c2_loglik_t.hip has the one-lane form) -------------
struct TermsArgs {
  const double *ar, *cr, *ac, *bc, *cc, *dc;
  int batched;   // coefficients per series (1) or shared by the batch (0)
  int Jc;        // complex terms; Jr = 8 - 2 Jc real ones in front of them
};
struct TermsGrads {
  double *bar, *bcr, *bac, *bbc, *bcc, *bdc;
};
// TM: 0 rows from the caller's arrays; 1 coefficient-level, every phase of the wavefront inside the range of the branch-free
// sincos; 2 coefficient-level with the library's sincos (raw Julian dates times a fast frequency)
struct SlotCoef {
  double A[2], Bq[2], D[2], A0;
  bool re[2];   // the slot is a pair of real terms
  int g0;       // first global column of the lane (4 h)
  __device__ __forceinline__ void load(const TermsArgs &T, int64_t b, int h, double (&cj)[J]) {
    const int JC = T.Jc, JR = J - 2 * JC;
    const int64_t br = T.batched ? b * JR : 0, bk = T.batched ? b * JC : 0;
    g0 = 4 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int g = g0 + 2 * s;
      re[s] = g < JR;
      if (re[s]) {
        A[s] = T.ar[br + g]; Bq[s] = -T.ar[br + g + 1]; D[s] = 0.0;
        cj[2 * s] = T.cr[br + g]; cj[2 * s + 1] = T.cr[br + g + 1];
      } else {
        const int k = (g - JR) >> 1;
        A[s] = T.ac[bk + k]; Bq[s] = T.bc[bk + k]; D[s] = T.dc[bk + k];
        cj[2 * s] = cj[2 * s + 1] = T.cc[bk + k];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) cj[4 + j] = swap_pair(cj[j]);
    double sum = 0.0;   // driver.cpp:456-458: the sum of ar, then of ac
    for (int r = 0; r < JR; ++r) sum += T.ar[br + r];
    for (int k = 0; k < JC; ++k) sum += T.ac[bk + k];
    A0 = sum;
  }
  // x sorted: the largest |x| of a series sits at one of its ends
  __device__ __forceinline__ bool phases_fast(double x_first, double x_last) const {
    const double xm = fmax(fabs(x_first), fabs(x_last));
    return (fabs(D[0]) * xm < kSincosFastMax) && (fabs(D[1]) * xm < kSincosFastMax);
  }
  template <int TM>
  static __device__ __forceinline__ void sc(double ph, double &sn, double &cs) {
    if constexpr (TM == 1) {
      sincos_cw_fast(ph, sn, cs);
      // (the range test looks at the ENDS of the grid; a row of an UNSORTED grid beyond the range must not pass for a result)
      if (!(fabs(ph) < kSincosFastMax)) sn = cs = __builtin_nan("");
    } else sincos_cw(ph, sn, cs);   // (a real pair's phase 0 stays on the branch-free path: cos = 1, sin = 0 exactly)
  }
  // U_n whole (local order), sin / cos of the lane's two slots
  template <int TM>
  __device__ __forceinline__ void rows_sc(double xn, double (&u)[J], double (&sn)[2], double (&cs)[2]) const {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      sc<TM>(D[s] * xn, sn[s], cs[s]);
      u[2 * s] = fma(A[s], cs[s], Bq[s] * sn[s]);
      u[2 * s + 1] = fma(A[s], sn[s], -(Bq[s] * cs[s]));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) u[4 + j] = swap_pair(u[j]);
  }
  // ... and the lane's half of V_n
  template <int TM>
  __device__ __forceinline__ void rows(double xn, double (&u)[J], double (&v)[J]) const {
    double sn[2], cs[2];
    rows_sc<TM>(xn, u, sn, cs);
#pragma unroll
    for (int s = 0; s < 2; ++s) { v[2 * s] = cs[s]; v[2 * s + 1] = re[s] ? 1.0 : sn[s]; }
  }
};

// =====================================================================================================================
// Forward pass.  REC: also W rows, (d, z), t, checkpoints and the stability measure of the backward recursion.
// =====================================================================================================================
template <bool REC, bool PAIRED, bool FULL, int TM = 0>
__device__ __forceinline__ void fwd_body(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                         const double *__restrict__ c, int64_t c_bs, const double *__restrict__ a,
                                         const double *__restrict__ U, const double *__restrict__ V,
                                         const double *__restrict__ y, double *__restrict__ ll, int32_t *__restrict__ flag,
                                         double *__restrict__ rec, Rec R, unsigned long long *__restrict__ guard, double *lds,
                                         const TermsArgs T = TermsArgs{}) {
  constexpr bool TERMS = TM != 0;   // `a` is then the white-noise diagonal; c, U, V are not read
  const int lane = threadIdx.x, sl = lane >> 1, h = lane & 1;
  const int64_t b0 = (int64_t)blockIdx.x * SPW;
  const int last = (int)((B - b0 < SPW ? B - b0 : SPW) - 1);   // lanes beyond the batch walk a copy of its last series
  const bool real = sl <= last;
  const int64_t b = b0 + (real ? sl : last);
  double *tU = lds, *tV = tU + SPW * RSTR, *tT = tV + SPW * RSTR, *tA = tT + SPW * SSTR, *tY = tA + SPW * SSTR;
  const double *Ub = U + b0 * N * J, *Vb = V + b0 * N * J, *ab = a + b0 * N, *yb = y + b0 * N;
  const double *tb = t + (t_bs ? b0 * N : 0);
  const int64_t tN = t_bs ? N : 0;
  double cj[J], cmax = 0.0;
  SlotCoef tc;
  if constexpr (TERMS) tc.load(T, b, h, cj);
  else {
#pragma unroll
    for (int j = 0; j < J; ++j) cj[j] = c[b * c_bs + ((j + 4 * h) & 7)];
  }
#pragma unroll
  for (int j = 0; j < J; ++j) cmax = fmax(cmax, cj[j]);
  double2 *recW = REC ? reinterpret_cast<double2 *>(rec + R.w + (size_t)blockIdx.x * N * J * SPW) : nullptr;
  double2 *recDZ = REC ? reinterpret_cast<double2 *>(rec + R.dz + (size_t)blockIdx.x * N * 2 * SPW) : nullptr;
  double *recT = REC ? rec + R.t + (size_t)blockIdx.x * N * SPW : nullptr;
  double *recCK = REC ? rec + R.ck + (size_t)blockIdx.x * R.nck * CKD * kWave : nullptr;

  double S[NL];
#pragma unroll
  for (int e = 0; e < NL; ++e) S[e] = 0.0;
  double F[4], w[J];   // (F: this lane's local 0..3 only -- every use is a half of a dot product or a checkpoint)
  double tprev = t[b * t_bs];
  double d = a[b * N], z = y[b * N];
  if constexpr (TERMS) d += tc.A0;
  double rd = 1.0 / d;
  if constexpr (TERMS) {
    double u0[J], v0[J];
    tc.template rows<TM>(tprev, u0, v0);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = v0[j] * rd;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[4 + j] = swap_pair(w[j]);
  } else {
#pragma unroll
    for (int j = 0; j < J; ++j) w[j] = V[b * N * J + ((j + 4 * h) & 7)] * rd;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) F[j] = 0.0;
  double prod = d, quad = z * z * rd;
  int eacc = 0;
  int32_t fl = 0;
  int slot = 0, nextra = 0;   // wavefront-uniform: checkpoint slots written, extras among them
  int64_t lastck = 0;
  double tseg = tprev, gmax = 0.0;
  auto write_ckpt = [&](int64_t row) __attribute__((always_inline)) {   // the state as it stands = state after `row`
    double *ck = recCK + (size_t)slot * CKD * kWave;
#pragma unroll
    for (int e = 0; e < NL; ++e) st1s(&ck[e * kWave + lane], S[e]);
#pragma unroll
    for (int j = 0; j < 4; ++j) st1s(&ck[(NL + j) * kWave + lane], F[j]);
    ++slot;
    lastck = row;
  };
  auto write_rec = [&](int64_t n, bool seg_end) __attribute__((always_inline)) {
    st2s(&recW[((size_t)n * 4 + 2 * h) * SPW + sl], make_double2(w[0], w[1]));        // local 0..3 = global 4h .. 4h+3
    st2s(&recW[((size_t)n * 4 + 2 * h + 1) * SPW + sl], make_double2(w[2], w[3]));
    if (h == 0) {
      st2s(&recDZ[(size_t)n * SPW + sl], make_double2(seg_end ? -fabs(d) : fabs(d), z));   // the sign of d: this row carries a checkpoint
      st1s(&recT[(size_t)n * SPW + sl], tprev);
    }
    if (seg_end) write_ckpt(n);   // (wavefront-uniform)
  };
  if (REC) write_rec(0, false);

  double su[8], sv[8], st_[8], sa_[8], sy_[8];
  if constexpr (!TERMS) { row_fetch<FULL>(Ub, N, 0, lane, last, su); row_fetch<FULL>(Vb, N, 0, lane, last, sv); }
  sc_fetch16<FULL>(tb, tN, N, 0, lane, last, st_); sc_fetch16<FULL>(ab, N, N, 0, lane, last, sa_); sc_fetch16<FULL>(yb, N, N, 0, lane, last, sy_);
  for (int64_t n0 = 0; n0 < N; n0 += ST) {
    const int half = (int)((n0 >> 3) & 1);
    lds_order();
    sc_stage16(tT, lane, half, st_); sc_stage16(tA, lane, half, sa_); sc_stage16(tY, lane, half, sy_);
    if (half == 1) {
      sc_fetch16<FULL>(tb, tN, N, n0 + ST, lane, last, st_); sc_fetch16<FULL>(ab, N, N, n0 + ST, lane, last, sa_);
      sc_fetch16<FULL>(yb, N, N, n0 + ST, lane, last, sy_);
    }
#pragma unroll
    for (int rt = 0; rt < ST / 2; ++rt) {
      const int64_t nt = n0 + rt * 2;
      if (nt < N) {
        if constexpr (!TERMS) {
          lds_order();
          row_stage(tU, lane, su); row_stage(tV, lane, sv);
          row_fetch<FULL>(Ub, N, nt + 2, lane, last, su); row_fetch<FULL>(Vb, N, nt + 2, lane, last, sv);
          lds_order();
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int64_t n = nt + r;
          if (n < N && n > 0) {
            const int rs = rt * 2 + r;
            const double tn = tT[sl * SSTR + rs], yn = tY[sl * SSTR + rs];
            const double an = TERMS ? tA[sl * SSTR + rs] + tc.A0 : tA[sl * SSTR + rs];
            double u[J], v[J], p[J];
            const double dt = tprev - tn;
            if (REC) {
              // the backward recursion reaches row n-1 by inverting the decay of row n unless the state of row n-1 is on
              // record: re-anchor there if some series of the wavefront could not afford that (wavefront-uniform)
              if (lastck == n - 1) tseg = tn;   // the decay into the row behind a checkpoint is never inverted
              if (__any(cmax * (tn - tseg) > kGuard) && nextra < (int)(R.nck - R.nreg)) {
                write_ckpt(n - 1);
                if (h == 0) recDZ[(size_t)(n - 1) * SPW + sl] = make_double2(-fabs(d), z);   // (d, z still of row n-1)
                ++nextra;
                tseg = tn;
              }
              gmax = fmax(gmax, cmax * (tn - tseg));
            }
            tprev = tn;
            if constexpr (TERMS) tc.template rows<TM>(tn, u, v);   // (v: the lane's local 0..3 only)
            else { row_read(tU, sl, h, r, u); row_read(tV, sl, h, r, v); }
            decay<PAIRED>(cj, dt, p);
            // S = P (S + d w^T w) P (forward.hpp:115-123); tau = U_n S (forward.hpp:126): own elements, then the pair's total
            double dw[J], tau[J], taut[4];
#pragma unroll
            for (int i = 0; i < J; ++i) { dw[i] = d * w[i]; tau[i] = 0.0; }
#pragma unroll
            for (int e = 0; e < NL; ++e) {
              const int i = LI[e], j2 = LJ[e];
              const double s = (p[i] * p[j2]) * fma(dw[i], w[j2], S[e]);
              S[e] = s;
              const double se = e >= kHalfFrom ? 0.5 * s : s;
              tau[j2] = fma(u[i], se, tau[j2]);
              if (j2 != i) tau[i] = fma(u[j2], se, tau[i]);
            }
            pair_total4(tau, taut);
            // F = P (F + W_{n-1}^T z_{n-1})   (internal.hpp:140-143)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) F[j2] = p[j2] * fma(w[j2], z, F[j2]);
            d = an - dot_halves(taut, u);   // forward.hpp:127
            z = yn - dot_halves(F, u);      // internal.hpp:144
            rd = rcp_nr(d);
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) w[j2] = (v[j2] - taut[j2]) * rd;   // forward.hpp:131
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) w[4 + j2] = swap_pair(w[j2]);      // (the partner's local 0..3)
            fl = ((fl == 0) & (d <= 0.0)) ? (int32_t)n : fl;                // forward.hpp:128
            prod *= d;
            quad = fma(z * z, rd, quad);
            if (r & 1) { int e2; prod = frexp(prod, &e2); eacc += e2; }
            if (REC) write_rec(n, (n % C == 0) || (n == N - 1));
          }
        }
      }
    }
  }
  if (h == 0 && real) {
    int e2;
    prod = frexp(prod, &e2);
    const double logdet = log(prod) + (double)(eacc + e2) * kLn2;
    flag[b] = fl;
    ll[b] = fl ? -INFINITY : -0.5 * (logdet + (double)N * kLog2Pi) - 0.5 * quad;
  }
  if (REC) {
    if (lane == 0) rec[R.cnt + blockIdx.x] = __longlong_as_double((long long)slot);
    // the stability measure of this wavefront; two wavefronts share the guard word of their group of 64 series (zeroed by
    // the launcher): beyond kGuard the reverse sweeps of both return at once and the replay kernels take the group
    double g = (gmax == gmax) ? gmax : INFINITY;
#pragma unroll
    for (int sft = 1; sft < kWave; sft <<= 1) g = fmax(g, __shfl_xor(g, sft, kWave));
    if (lane == 0) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(g);
      atomicMax(guard + kGateHeadWords + (b0 >> 6), bits);
      atomicMax(guard, bits);
      if (g > kGuard) atomicAdd(guard + 1, 1ull);
    }
  }
}

constexpr int kFwdLds = (2 * SPW * RSTR + 3 * SPW * SSTR) * 8;

template <bool REC>
__global__ __launch_bounds__(kWave, 1) void k_k2_fwd(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                                     const double *__restrict__ c, int64_t c_bs,
                                                     const double *__restrict__ a, const double *__restrict__ U,
                                                     const double *__restrict__ V, const double *__restrict__ y,
                                                     double *__restrict__ ll, int32_t *__restrict__ flag,
                                                     double *__restrict__ rec, Rec R, unsigned long long *__restrict__ guard) {
  __shared__ __attribute__((aligned(16))) double lds[kFwdLds / 8];
  int64_t bb = (int64_t)blockIdx.x * SPW + (threadIdx.x >> 1);
  bb = bb < B ? bb : B - 1;
  bool paired = true;
#pragma unroll
  for (int k = 0; k < J / 2; ++k) paired = paired && (c[bb * c_bs + 2 * k] == c[bb * c_bs + 2 * k + 1]);
  const bool full = B - (int64_t)blockIdx.x * SPW >= SPW;
#define C2K2_FWD(P_, F_) fwd_body<REC, P_, F_>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, rec, R, guard, lds)
  if (__all(paired)) { if (full) C2K2_FWD(true, true); else C2K2_FWD(true, false); }
  else { if (full) C2K2_FWD(false, true); else C2K2_FWD(false, false); }
#undef C2K2_FWD
}

// =====================================================================================================================
// Reverse sweep: the fused step of c2_loglik_t.hip (solve_lower_rev internal.hpp:225-245 + factor_rev reverse.hpp:52-84 +
// the backward recursion of S and F) on the local element set.  Entering step n: bz, ba, bV = complete cotangents of row n;
// S, F = state of row n; M = bS + bS^T on the local elements; bF; carry = f_{n+1}.
// =====================================================================================================================
// C2K2_MLDS: the cotangent state M lives in LDS (pairs of elements, lane-major: conflict-free 16-byte accesses) instead of
// registers -- the reverse step has ~340 registers' worth of live values at its peak and every value beyond 256 costs
// a move to and from the accumulation registers per step (16384 series: 10.1 against 10.9 ms)
#ifndef C2K2_MLDS
#define C2K2_MLDS 1
#endif
#ifndef C2K2_SLDS
#define C2K2_SLDS 0   // the forward state S of the backward recursion likewise: measured, no (16384 series: 12.2 against 10.1 ms)
#endif
// The scalar gradients (ba, by, bt) leave the reverse sweep as SIXTEEN-row tiles -- whole aligned 128-byte lines per series (second
// session of round 6).  With eight-row tiles the two 64-byte halves of a line were written eight steps (~30 us) apart: the line has left
// L2 by then and the memory side merges each half on its own (a read-modify-write); tools/ubench/replay_traffic.hip, which reproduces
// the one-lane pair's times from its memory operations alone, puts that at 11 - 15 % of the reverse sweep (profiles/r06_halflines.md).
// C2K2_STR=8: the earlier tiles (A/B builds).
#ifndef C2K2_STR
#define C2K2_STR 16
#endif
constexpr int STR = C2K2_STR, SSTRR = STR + 1;   // rows / LDS stride (doubles) of a series in a scalar tile of the reverse sweep
constexpr int kRevLds = (2 * SPW * RSTR + 3 * SPW * SSTRR + (C2K2_MLDS ? NL * kWave : 0) + (C2K2_SLDS ? NL * kWave : 0)) * 8;

// STR-row scalar tile of the reverse sweep -> memory: one instruction moves (64 / STR) series x (8 STR) bytes
template <bool FULL>
__device__ __forceinline__ void sc_flush_r(double *__restrict__ base, int64_t N, int64_t n0, const double *tile, int lane,
                                           int last) {
  constexpr int SPI = kWave / STR, NI = SPW / SPI;   // series per instruction, instructions per tile
  const int64_t r = n0 + (lane & (STR - 1));
  double v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) v[i] = tile[(SPI * i + lane / STR) * SSTRR + (lane & (STR - 1))];
  if (r >= 0 && r < N) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (FULL || SPI * i + lane / STR <= last) base[(int64_t)(SPI * i + lane / STR) * N + r] = v[i];
  }
}

template <bool PAIRED, bool FULL, int TM = 0>
__device__ __forceinline__ void rev_body(int64_t B, int64_t N, const double *__restrict__ c, int64_t c_bs,
                                         const double *__restrict__ U, const int32_t *__restrict__ flag,
                                         const double *__restrict__ rec, Rec R, double *__restrict__ bt,
                                         double *__restrict__ bc, double *__restrict__ ba, double *__restrict__ bU,
                                         double *__restrict__ bV, double *__restrict__ by, double *lds,
                                         const TermsArgs T = TermsArgs{}, const TermsGrads G = TermsGrads{}) {
  constexpr bool TERMS = TM != 0;   // bt, ba, by are then bx, bdiag, by; c, U, bc, bU, bV are not touched
  const int lane = threadIdx.x, sl = lane >> 1, h = lane & 1;
  const int64_t b0 = (int64_t)blockIdx.x * SPW;
  const int last = (int)((B - b0 < SPW ? B - b0 : SPW) - 1);
  const bool real = sl <= last;
  const int64_t b = b0 + (real ? sl : last);
  double *tU = lds, *tBV = tU + SPW * RSTR, *tBA = tBV + SPW * RSTR, *tBY = tBA + SPW * SSTRR, *tBT = tBY + SPW * SSTRR;
  const double *Ub = U + b0 * N * J;
  double *bUb = bU + b0 * N * J, *bVb = bV + b0 * N * J, *bab = ba + b0 * N, *byb = by + b0 * N, *btb = bt + b0 * N;
  const double2 *recW = reinterpret_cast<const double2 *>(rec + R.w + (size_t)blockIdx.x * N * J * SPW);
  const double2 *recDZ = reinterpret_cast<const double2 *>(rec + R.dz + (size_t)blockIdx.x * N * 2 * SPW);
  const double *recT = rec + R.t + (size_t)blockIdx.x * N * SPW;
  const double *recCK = rec + R.ck + (size_t)blockIdx.x * R.nck * CKD * kWave;
  const bool failed = flag[b] != 0;   // NaN gradients for a failed factorisation
  const double nan = __builtin_nan("");
  double cj[J], bcj[4];
  SlotCoef tc;
  if constexpr (TERMS) tc.load(T, b, h, cj);
  else {
#pragma unroll
    for (int j = 0; j < J; ++j) cj[j] = c[b * c_bs + ((j + 4 * h) & 7)];
  }
  // coefficient-level form: running sums of the lane's two slots -- [s][0] bac (real pair: sum bU_0), [1] bbc (-sum bU_1),
  // [2] bdc -- and the sum of ba
  double acc[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}, sba = 0.0;

  // F, bF, the accumulators of bc: this lane's local 0..3 only (every use is a half of a dot product, a half row of an output
  // or a checkpoint); bV is needed whole by the pass over the elements and is completed from the partner
  double F[4], bF[4], bVn[J];
#if C2K2_SLDS
  double2 *Sl = reinterpret_cast<double2 *>(tBT + SPW * SSTRR + (C2K2_MLDS ? NL * kWave : 0)) + lane;
#else
  double S[NL];
#endif
#if C2K2_MLDS
  double2 *Ml = reinterpret_cast<double2 *>(tBT + SPW * SSTRR) + lane;   // elements 2q, 2q + 1 at Ml[q * kWave]
#pragma unroll
  for (int q = 0; q < NL / 2; ++q) Ml[q * kWave] = make_double2(0.0, 0.0);
#else
  double M[NL];
#pragma unroll
  for (int e = 0; e < NL; ++e) M[e] = 0.0;
#endif
#if !C2K2_SLDS
#pragma unroll
  for (int e = 0; e < NL; ++e) S[e] = 0.0;
#endif
#pragma unroll
  for (int j = 0; j < 4; ++j) { F[j] = 0.0; bF[j] = 0.0; bcj[j] = 0.0; }
#pragma unroll
  for (int j = 0; j < J; ++j) bVn[j] = failed ? nan : 0.0;
  double carry = 0.0;
  const int64_t nf = N - 1;
  const double2 dzl = recDZ[(size_t)nf * SPW + sl];
  const double rdl = 1.0 / fabs(dzl.x);
  double ban = 0.5 * rdl * (dzl.y * dzl.y * rdl - 1.0), bzn = -dzl.y * rdl;   // seeds of the last row
  if (failed) { ban = nan; bzn = nan; }
  double tcur = recT[(size_t)nf * SPW + sl];

  // slots are consumed in reverse order of writing
  int slot = __builtin_amdgcn_readfirstlane((int)__double_as_longlong(rec[R.cnt + blockIdx.x]));
  auto load_ckpt = [&]() {   // the recorded state of a checkpointed row replaces the recursed one
    --slot;
    const double *ck = recCK + (size_t)slot * CKD * kWave;
#if C2K2_SLDS
#pragma unroll
    for (int q = 0; q < NL / 2; ++q) Sl[q * kWave] = make_double2(ck[(2 * q) * kWave + lane], ck[(2 * q + 1) * kWave + lane]);
#else
#pragma unroll
    for (int e = 0; e < NL; ++e) S[e] = ld1s(&ck[e * kWave + lane]);
#endif
#pragma unroll
    for (int j = 0; j < 4; ++j) F[j] = ld1s(&ck[(NL + j) * kWave + lane]);
  };
  auto w_fetch = [&](int64_t row, double (&wv)[J]) {
    row = row < 0 ? 0 : row;
#ifdef C2K2_EXP_CACHED_RECORDS
    row &= 3;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 v = ld2s(&recW[((size_t)row * 4 + ((q + 2 * h) & 3)) * SPW + sl]);
      wv[2 * q] = v.x; wv[2 * q + 1] = v.y;
    }
  };
#ifdef C2K2_EXP_CACHED_RECORDS   // (timing experiment: every record load hits the cache -- what is left is arithmetic and stores)
  auto dz_fetch = [&](int64_t row) { return recDZ[(size_t)((row < 0 ? 0 : row) & 3) * SPW + sl]; };
  auto t_fetch = [&](int64_t row) { return recT[(size_t)((row < 0 ? 0 : row) & 3) * SPW + sl]; };
#else
  auto dz_fetch = [&](int64_t row) { return ld2s(&recDZ[(size_t)(row < 0 ? 0 : row) * SPW + sl]); };
  auto t_fetch = [&](int64_t row) { return ld1s(&recT[(size_t)(row < 0 ? 0 : row) * SPW + sl]); };
#endif

  if (N >= 2) {
    // ---- prologue: bV, ba, by of the last row are pure seeds ---------------------------------------------------------------
    const int ph0 = (int)(nf & 1);
    if constexpr (!TERMS) row_write_half(tBV, sl, h, ph0, bVn);
    tBA[sl * SSTRR + (nf & (STR - 1))] = ban;
    tBY[sl * SSTRR + (nf & (STR - 1))] = bzn;
    lds_order();
    if constexpr (!TERMS) {
      if (ph0 == 0) row_flush<FULL>(bVb, N, nf, tBV, lane, last);   // (an even last row: its pair partner lies beyond the series: only row nf)
    }
    if ((nf & (STR - 1)) == 0) {                      // the last row alone at the bottom of its scalar tile
      sc_flush_r<FULL>(bab, N, nf, tBA, lane, last);
      sc_flush_r<FULL>(byb, N, nf, tBY, lane, last);
    }
    double su[8], wa[J];
    double2 dza;
    double ta;
    if constexpr (!TERMS) {
      row_fetch<FULL>(Ub, N, nf - ph0, lane, last, su);
      if (ph0 != 1) {   // the first step is not a staging one: its tile goes into LDS here
        lds_order();
        row_stage(tU, lane, su);
        lds_order();
        row_fetch<FULL>(Ub, N, nf - ph0 - 2, lane, last, su);
      }
    }
    w_fetch(nf - 1, wa);
    dza = dz_fetch(nf - 1);
    ta = t_fetch(nf - 1);
    load_ckpt();   // the last row always carries one
    lds_order();

    auto step = [&](const int64_t n, auto phase_tag) __attribute__((always_inline)) {
      constexpr int PH = decltype(phase_tag)::value;   // n & 1
      __builtin_amdgcn_sched_barrier(0);
      double wb[J];
      if constexpr (PH == 1 && !TERMS) {   // the step that puts its pair of U rows into LDS and requests the pair below
        lds_order();
        row_stage(tU, lane, su);
        row_fetch<FULL>(Ub, N, n - 3, lane, last, su);
      }
      w_fetch(n - 2, wb);
      const double2 dzb = dz_fetch(n - 2);
      const double tb2 = t_fetch(n - 2);
      if constexpr (PH == 1 && !TERMS) lds_order();
      const int rs = (int)((n - 1) & (STR - 1));
      double u[J], p[J], ip[J];
      double sn2[2], cs2[2], gv[2];
      const double xn = tcur, ba_in = ban;
      if constexpr (TERMS) {
        tc.template rows_sc<TM>(xn, u, sn2, cs2);
#pragma unroll
        for (int s = 0; s < 2; ++s) gv[s] = fma(bVn[2 * s + 1], cs2[s], -(bVn[2 * s] * sn2[s]));   // -bV0 sin + bV1 cos of row n
      } else {
        row_read(tU, sl, h, PH, u);
      }
      const double tm = ta, dt = tm - tcur;
      tcur = tm;
      decay<PAIRED>(cj, dt, p);
      // solve_lower_rev part (internal.hpp:232-245)
      double bp0[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bF[j] = fma(-u[j], bzn, bF[j]);
        bp0[j] = F[j] * bF[j];
        bF[j] *= p[j];
      }
      // factor_rev part (reverse.hpp:65-80) and the backward recursion of S, one pass over the local elements
      const double dm = fabs(dza.x), rdm = rcp_nr(dm), zm = dza.y;
      const bool ckm = __builtin_amdgcn_readfirstlane(__double2hiint(dza.x)) < 0;   // row n-1 carries a checkpoint
      double x[J], xs[J], q[J], bps[J];
      const double ba2 = 2.0 * ban;
#pragma unroll
      for (int j = 0; j < J; ++j) { x[j] = fma(ba2, u[j], bVn[j]); xs[j] = 0.0; q[j] = 0.0; bps[j] = 0.0; }
      inverses<PAIRED>(p, ip);
#if C2K2_MLDS
      double2 mpair;
#endif
#if C2K2_SLDS
      double2 spair;
#endif
#pragma unroll
      for (int e = 0; e < NL; ++e) {
        const int i = LI[e], j2 = LJ[e];
        const double wgt = e >= kHalfFrom ? 0.5 : 1.0;
#if C2K2_SLDS
        if ((e & 1) == 0) spair = Sl[(e / 2) * kWave];
        const double sv = (e & 1) ? spair.y : spair.x;
#else
        const double sv = S[e];
#endif
        const double svw = e >= kHalfFrom ? 0.5 * sv : sv;
#if C2K2_MLDS
        if ((e & 1) == 0) mpair = Ml[(e / 2) * kWave];
        double m = (e & 1) ? mpair.y : mpair.x;
#else
        double m = M[e];
#endif
        xs[j2] = fma(x[i], svw, xs[j2]);
        if (j2 != i) xs[i] = fma(x[j2], svw, xs[i]);
        m = fma(-u[i], bVn[j2], m);
        m = fma(-x[i], u[j2], m);
        bps[j2] = fma(svw, m, bps[j2]);
        if (j2 != i) bps[i] = fma(svw, m, bps[i]);
        m *= p[i] * p[j2];
#if C2K2_MLDS
        if (e & 1) { mpair.y = m; Ml[(e / 2) * kWave] = mpair; } else mpair.x = m;
#else
        M[e] = m;
#endif
        const double mw = wgt * m;
        q[j2] = fma(wa[i], mw, q[j2]);
        if (j2 != i) q[i] = fma(wa[j2], mw, q[i]);
        const double snew = fma(-(dm * wa[i]), wa[j2], sv * (ip[i] * ip[j2]));
#if C2K2_SLDS
        if (e & 1) { spair.y = snew; Sl[(e / 2) * kWave] = spair; } else spair.x = snew;
#else
        S[e] = snew;
#endif
      }
      // from here on every vector is this lane's local 0..3; scalars are (own half) + (partner's half)
      double xst[4], qt[4], bp[4];
      pair_total4(xs, xst);
      pair_total4(q, qt);
      pair_total4(bps, bp);
#pragma unroll
      for (int j = 0; j < 4; ++j) bp[j] += bp0[j];
      double gsum = 0.0;
      {
        double o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = fma(-bzn, F[j], -xst[j]);   // bU_n = -bz_n F_n - x S_n
        if constexpr (TERMS) {   // the reverse of the recipe for row n (c2_terms.hip: k_terms_rev), this lane's two slots
          double gs = 0.0;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const double b0_ = o[2 * s], b1_ = o[2 * s + 1];
            acc[s][0] = fma(b0_, cs2[s], fma(b1_, sn2[s], acc[s][0]));
            acc[s][1] = fma(b0_, sn2[s], fma(-b1_, cs2[s], acc[s][1]));
            const double g = fma(-b0_, u[2 * s + 1], fma(b1_, u[2 * s], gv[s]));   // cotangent of the phase dc x_n
            acc[s][2] = fma(g, xn, acc[s][2]);
            gs = fma(g, tc.D[s], gs);
          }
          gsum = gs + swap_pair(gs);
          sba += ba_in;
        } else {
          row_write_half(tU, sl, h, PH, o);                             // bU_n takes the place of U_n in the tile
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) bcj[j] = fma(dt, bp[j], bcj[j]);
      const double f = dot_halves(bp, cj);
      const double btn = carry - f + gsum;   // coefficient-level form: bx_n
      carry = f;
#pragma unroll
      for (int j = 0; j < 4; ++j) F[j] = fma(-wa[j], zm, F[j] * ip[j]);   // F_{n-1} = P^-1 F_n - w_{n-1} z_{n-1}
      const double Gs = dot_halves(bF, wa), Q = dot_halves(qt, wa);
      const double zr = zm * rdm;
      bzn = Gs - zr;
#pragma unroll
      for (int j = 0; j < 4; ++j) bVn[j] = fma(zr, bF[j], qt[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) bVn[4 + j] = swap_pair(bVn[j]);   // (the partner's local 0..3)
      ban = 0.5 * rdm * (zm * zr - 1.0) - 0.5 * Q - zr * Gs;
      if (h == 0) {
        tBA[sl * SSTRR + rs] = ban;
        tBY[sl * SSTRR + rs] = bzn;
        tBT[sl * SSTRR + (int)(n & (STR - 1))] = btn;
      }
      if constexpr (!TERMS) row_write_half(tBV, sl, h, (PH + 1) & 1, bVn);   // bV_{n-1}
      lds_order();
      // width-8 outputs leave as aligned pairs of rows: bU at the end of the even step (rows n, n + 1), bV at the end of the odd
      // step (rows n - 1, n)
      if constexpr (!TERMS) {
        if constexpr (PH == 0) row_flush<FULL>(bUb, N, n, tU, lane, last);
        else row_flush<FULL>(bVb, N, n - 1, tBV, lane, last);
      }
      if ((n & (STR - 1)) == 0) sc_flush_r<FULL>(btb, N, n, tBT, lane, last);
      if (rs == 0) {
        sc_flush_r<FULL>(bab, N, n - 1, tBA, lane, last);
        sc_flush_r<FULL>(byb, N, n - 1, tBY, lane, last);
      }
      if (ckm && n >= 2) load_ckpt();   // the state of row n-1 is on record: it replaces the recursed one
#pragma unroll
      for (int j = 0; j < J; ++j) wa[j] = wb[j];
      dza = dzb;
      ta = tb2;
    };
    for (int64_t n = nf | 1; n >= 1; n -= 2) {
      if (n <= nf) step(n, std::integral_constant<int, 1>{});
      if (n >= 2) step(n - 1, std::integral_constant<int, 0>{});
    }
    // row 0: bU_0 = 0 (reverse.hpp:83), bt_0 = f_1; ba_0 / by_0 / bV_0 left with the last step
    {
      double zero[J];
#pragma unroll
      for (int j = 0; j < J; ++j) zero[j] = failed ? nan : 0.0;
      lds_order();
      if constexpr (TERMS) {   // row 0: bU_0 = 0; bV_0 and ba_0 are complete
        double u0[J], sn2[2], cs2[2], gs = 0.0;
        tc.template rows_sc<TM>(tcur, u0, sn2, cs2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const double g = fma(bVn[2 * s + 1], cs2[s], -(bVn[2 * s] * sn2[s]));
          acc[s][2] = fma(g, tcur, acc[s][2]);
          gs = fma(g, tc.D[s], gs);
        }
        carry += gs + swap_pair(gs);
        sba += ban;
      } else {
        row_write_half(tU, sl, h, 0, zero);   // bU_1 is waiting in row 1 of the tile
      }
      if (h == 0) tBT[sl * SSTRR] = carry;
      lds_order();
      if constexpr (!TERMS) row_flush<FULL>(bUb, N, 0, tU, lane, last);
      sc_flush_r<FULL>(btb, N, 0, tBT, lane, last);
    }
  } else {   // N == 1: seeds only
    if (h == 0 && real) {
      bab[(int64_t)sl * N] = ban;
      byb[(int64_t)sl * N] = bzn;
      btb[(int64_t)sl * N] = failed ? nan : 0.0;
      if constexpr (!TERMS) {
#pragma unroll
        for (int j = 0; j < J; ++j) { bUb[(int64_t)sl * N * J + j] = failed ? nan : 0.0; bVb[(int64_t)sl * N * J + j] = failed ? nan : 0.0; }
      }
    }
    if constexpr (TERMS) {   // the only row: sum ba = ba_0, every other sum is empty
      sba = ban;
      if (failed) {
#pragma unroll
        for (int s = 0; s < 2; ++s) { acc[s][0] = acc[s][1] = acc[s][2] = nan; }
#pragma unroll
        for (int j = 0; j < 4; ++j) bcj[j] = nan;
      }
    }
  }
  if (real) {
    if constexpr (TERMS) {
      const int JC = T.Jc, JR = J - 2 * JC;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int g = tc.g0 + 2 * s;
        if (tc.re[s]) {
          G.bar[b * JR + g] = sba + acc[s][0]; G.bar[b * JR + g + 1] = sba - acc[s][1];
          G.bcr[b * JR + g] = bcj[2 * s]; G.bcr[b * JR + g + 1] = bcj[2 * s + 1];
        } else {
          const int k = (g - JR) >> 1;
          G.bac[b * JC + k] = sba + acc[s][0];
          G.bbc[b * JC + k] = acc[s][1];
          G.bdc[b * JC + k] = acc[s][2];
          G.bcc[b * JC + k] = bcj[2 * s] + bcj[2 * s + 1];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bc[b * J + 4 * h + j] = bcj[j];   // local 0..3 = global 4h .. 4h+3
    }
  }
}

__global__ __launch_bounds__(kWave, 1) void k_k2_rev(int64_t B, int64_t N, const double *__restrict__ c, int64_t c_bs,
                                                     const double *__restrict__ U, const int32_t *__restrict__ flag,
                                                     const double *__restrict__ rec, Rec R,
                                                     const unsigned long long *__restrict__ guard, double *__restrict__ bt,
                                                     double *__restrict__ bc, double *__restrict__ ba,
                                                     double *__restrict__ bU, double *__restrict__ bV,
                                                     double *__restrict__ by) {
  __shared__ __attribute__((aligned(16))) double lds[kRevLds / 8];
  const int64_t b0 = (int64_t)blockIdx.x * SPW;
  if (__longlong_as_double((long long)guard[kGateHeadWords + (b0 >> 6)]) > kGuard) return;   // the replay kernels take this group
  int64_t bb = b0 + (threadIdx.x >> 1);
  bb = bb < B ? bb : B - 1;
  bool paired = true;
#pragma unroll
  for (int k = 0; k < J / 2; ++k) paired = paired && (c[bb * c_bs + 2 * k] == c[bb * c_bs + 2 * k + 1]);
  const bool full = B - b0 >= SPW;
#define C2K2_REV(P_, F_) rev_body<P_, F_>(B, N, c, c_bs, U, flag, rec, R, bt, bc, ba, bU, bV, by, lds)
  if (__all(paired)) { if (full) C2K2_REV(true, true); else C2K2_REV(true, false); }
  else { if (full) C2K2_REV(false, true); else C2K2_REV(false, false); }
#undef C2K2_REV
}

// ---- coefficient-level kernels: the same bodies, rows formed in the lanes ---------------------------------------------------
// wavefront-uniform: paired rates (every complex term is; a real pair only with equal rates) and phases inside the range of
// the branch-free sincos
__device__ __forceinline__ void terms_traits(int64_t B, int64_t N, const double *__restrict__ x, int64_t x_bs, const TermsArgs &T,
                                             bool &paired, bool &fast) {
  int64_t bb = (int64_t)blockIdx.x * SPW + (threadIdx.x >> 1);
  bb = bb < B ? bb : B - 1;
  SlotCoef tc;
  double cj[J];
  tc.load(T, bb, (int)(threadIdx.x & 1), cj);
  paired = __all(cj[0] == cj[1] && cj[2] == cj[3]);
  fast = __all(tc.phases_fast(x[bb * x_bs], x[bb * x_bs + N - 1]));
}

template <bool REC>
__global__ __launch_bounds__(kWave, 1) void k_k2_tt_fwd(int64_t B, int64_t N, const double *__restrict__ x, int64_t x_bs,
                                                        TermsArgs T, const double *__restrict__ diag,
                                                        const double *__restrict__ y, double *__restrict__ ll,
                                                        int32_t *__restrict__ flag, double *__restrict__ rec, Rec R,
                                                        unsigned long long *__restrict__ guard) {
  __shared__ __attribute__((aligned(16))) double lds[kFwdLds / 8];
  bool paired, fast;
  terms_traits(B, N, x, x_bs, T, paired, fast);
#define C2K2_TFWD(P_, M_) fwd_body<REC, P_, false, M_>(B, N, x, x_bs, nullptr, 0, diag, nullptr, nullptr, y, ll, flag, rec, R, guard, lds, T)
  if (paired) { if (fast) C2K2_TFWD(true, 1); else C2K2_TFWD(true, 2); }
  else { if (fast) C2K2_TFWD(false, 1); else C2K2_TFWD(false, 2); }
#undef C2K2_TFWD
}

__global__ __launch_bounds__(kWave, 1) void k_k2_tt_rev(int64_t B, int64_t N, const double *__restrict__ x, int64_t x_bs,
                                                        TermsArgs T, const int32_t *__restrict__ flag,
                                                        const double *__restrict__ rec, Rec R,
                                                        const unsigned long long *__restrict__ guard, TermsGrads G,
                                                        double *__restrict__ bx, double *__restrict__ bdiag,
                                                        double *__restrict__ by) {
  __shared__ __attribute__((aligned(16))) double lds[kRevLds / 8];
  const int64_t b0 = (int64_t)blockIdx.x * SPW;
  if (__longlong_as_double((long long)guard[kGateHeadWords + (b0 >> 6)]) > kGuard) return;   // the composed chain takes this group
  bool paired, fast;
  terms_traits(B, N, x, x_bs, T, paired, fast);
#define C2K2_TREV(P_, M_) rev_body<P_, false, M_>(B, N, nullptr, 0, nullptr, flag, rec, R, bx, nullptr, bdiag, nullptr, nullptr, by, lds, T, G)
  if (paired) { if (fast) C2K2_TREV(true, 1); else C2K2_TREV(true, 2); }
  else { if (fast) C2K2_TREV(false, 1); else C2K2_TREV(false, 2); }
#undef C2K2_TREV
}

}  // namespace c2k2

using namespace c2k2;

extern "C" {

int c2_internal_loglik_k2_ok(int64_t B, int64_t N, int64_t J) { return J == 8 && B >= 1 && N >= 1; }

int c2_internal_loglik_k2(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *a,
                          const double *U, const double *V, const double *y, double *ll, int32_t *flag, c2_stream_t stream) {
  Rec R{};
  hipLaunchKernelGGL((k_k2_fwd<false>), dim3((unsigned)((B + SPW - 1) / SPW)), dim3(kWave), 0, (hipStream_t)stream, B, N, t, t_bs, c, c_bs,
                     a, U, V, y, ll, flag, (double *)nullptr, R, (unsigned long long *)nullptr);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

size_t c2_internal_loglik_k2_record_doubles(int64_t B, int64_t N) { return rec_layout(B, N).total; }

// Forward with records + reverse sweep.  `guard`: kGateHeadWords + ceil(B / 64) device words, ALL zeroed by the caller on the same
// stream (the two wavefronts of a group of 64 series raise its word together).
int c2_internal_loglik_k2_grad(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                               const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                               double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, double *rec,
                               unsigned long long *guard, c2_stream_t stream) {
  const dim3 grid((unsigned)((B + SPW - 1) / SPW));
  const Rec R = rec_layout(B, N);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((k_k2_fwd<true>), grid, dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, rec, R, guard);
  if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;
  hipLaunchKernelGGL(k_k2_rev, grid, dim3(kWave), 0, s, B, N, c, c_bs, U, (const int32_t *)flag, (const double *)rec, R,
                     (const unsigned long long *)guard, bt, bc, ba, bU, bV, by);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// Coefficient-level forms (J = Jr + 2 Jc == 8; same arguments as c2_internal_loglik_tt8 / _tt_grad8 of c2_loglik_t.hip).
int c2_internal_loglik_k2_tt(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                             const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                             int64_t x_bs, const double *diag, const double *y, double *ll, int32_t *flag, c2_stream_t stream) {
  if (Jc < 0 || Jc > 4) return C2_ERR_UNSUPPORTED;
  const TermsArgs T{ar, cr, ac, bc, cc, dc, coef_batched, (int)Jc};
  Rec R{};
  hipLaunchKernelGGL((k_k2_tt_fwd<false>), dim3((unsigned)((B + SPW - 1) / SPW)), dim3(kWave), 0, (hipStream_t)stream, B, N, x, x_bs,
                     T, diag, y, ll, flag, (double *)nullptr, R, (unsigned long long *)nullptr);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// `guard`: kGateHeadWords + ceil(B / 64) words, ALL zeroed by the caller on the same stream (see c2_internal_loglik_k2_grad).
int c2_internal_loglik_k2_tt_grad(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                                  const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                                  int64_t x_bs, const double *diag, const double *y, double *ll, double *bar, double *bcr,
                                  double *bac, double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                                  int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream) {
  if (Jc < 0 || Jc > 4) return C2_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((B + SPW - 1) / SPW));
  const TermsArgs T{ar, cr, ac, bc, cc, dc, coef_batched, (int)Jc};
  const TermsGrads G{bar, bcr, bac, bbc, bcc, bdc};
  const Rec R = rec_layout(B, N);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((k_k2_tt_fwd<true>), grid, dim3(kWave), 0, s, B, N, x, x_bs, T, diag, y, ll, flag, rec, R, guard);
  if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;
  hipLaunchKernelGGL(k_k2_tt_rev, grid, dim3(kWave), 0, s, B, N, x, x_bs, T, (const int32_t *)flag, (const double *)rec, R,
                     (const unsigned long long *)guard, G, bx, bdiag, by);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

}  // extern "C"
