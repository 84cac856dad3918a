// c2_loglik_q4.hip -- fused log-likelihood + gradient with FOUR lanes per series (two columns of the J x J state per lane),
// J = 8: the mapping between the eight-lane pair (c2_loglik.hip) and the two-lane pair (c2_loglik_k2.hip).  16 series per
// wavefront: 8193 ... 16384 series put one wavefront on every SIMD where the eight-lane pair needs two.  Per series a step
// costs 0.65 of the eight-lane step: the scalar chain (reciprocal, reductions, seeds) is replicated over four lanes instead
// of eight, a gathered vector is 12 DPP moves per 16 series instead of 14 per 8.
//
// Both kernels work in a SCALED FRAME between anchors 32 rows apart (see k_loglik_rev<..., SC> in c2_loglik.hip for the
// reverse sweep's derivation).  Forward (forward.hpp:105-134 + internal.hpp:135-145), with h_n = exp(c (t_n - t_ref)), t_ref
// the time of the anchor row below:
//     S^_n = H_n S_n H_n = S^_{n-1} + d_{n-1} w~ w~^T          (w~ = h_{n-1} W_{n-1}; no decay factors)
//     tau_n = U_n S_n = (u- S^_n) / h_n                        (u- = U_n / h_n)
//     F~_n = h_n F_n = F~_{n-1} + w~ z_{n-1},   z_n = y_n - u- . F~_n,   d_n = a_n - (u- S^_n) . u-
// and at every anchor row the state is turned back into S, F (one gather and 32 multiplications per 32 rows), recorded as
// the checkpoint the reverse sweep re-anchors at, and becomes the reference of the next frame.  The frame needs
// c_j (t_n - t_ref) within the guard (2.0) over every anchor interval -- the same bound the backward recursion of the reverse
// sweep needs -- so a group of 64 series with a longer span (gaps in time) is not run here at all: `gate` holds one word per
// group (k_q4_gate below, from k_anchor_spans' words), both kernels return at once for a closed group and the eight-lane
// replay pair, launched behind them on the same words, takes it.
//
// Records (private to the pair, lane-major: every access one contiguous run per wavefront): W rows (16 B per lane and
// row), (d, z) pairs (transposed through LDS, series-major), one plain checkpoint (S columns of the lane + F: 18 doubles
// per lane) per anchor and after the last row.
#include <type_traits>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

#ifndef C2_Q4_LN_R
#define C2_Q4_LN_R 16   // rows per block of the forward kernel's LN instance (8: scalar requests as 64-byte runs; A/B builds)
#endif

namespace c2 {
namespace q4 {

#ifndef C2Q4_PAIRLINES
#define C2Q4_PAIRLINES 1
#endif
constexpr int LG = 4, J = 8, SPW = kWave / LG, C = 8, A = 4;   // C rows per segment, A segments between two anchors
constexpr int kCkD2 = 9;                                       // double2 per lane and checkpoint: SX[2][8] + F[2]

__device__ __forceinline__ bool open_group(const unsigned long long *gate, int64_t b0) {
  return __longlong_as_double((long long)gate[b0 >> 6]) <= kBackwardGuard;   // (NaN / +inf: closed)
}

// gathered pair vector in XOR order over the lane index: slot 2k + e = element 2 (jl ^ k) + e
__device__ __forceinline__ void xg2(double x0, double x1, double (&out)[J]) {
  out[0] = x0; out[1] = x1;
  out[2] = dpp_mov<kDppXor1>(x0); out[3] = dpp_mov<kDppXor1>(x1);
  out[4] = dpp_mov<kDppXor2>(x0); out[5] = dpp_mov<kDppXor2>(x1);
  out[6] = dpp_mov<kDppXor3>(x0); out[7] = dpp_mov<kDppXor3>(x1);
}
__device__ __forceinline__ void xg2_lds(const double2 *slot, int lane, double (&out)[J]) {
#pragma unroll
  for (int k = 0; k < LG; ++k) {
    const double2 v = slot[lane ^ k];
    out[2 * k] = v.x;
    out[2 * k + 1] = v.y;
  }
}
__device__ __forceinline__ void ck_store(double2 *rec, int lane, const double (&SX)[2][J], const double (&F)[2]) {
#pragma unroll
  for (int q = 0; q < J; ++q) rec[q * kWave + lane] = make_double2(SX[0][q], SX[1][q]);
  rec[J * kWave + lane] = make_double2(F[0], F[1]);
}
__device__ __forceinline__ void ck_load(const double2 *rec, int lane, double (&SX)[2][J], double (&F)[2]) {
#pragma unroll
  for (int q = 0; q < J; ++q) { const double2 v = rec[q * kWave + lane]; SX[0][q] = v.x; SX[1][q] = v.y; }
  const double2 f = rec[J * kWave + lane];
  F[0] = f.x; F[1] = f.y;
}
__device__ __forceinline__ void apark(double x, int &lo, int &hi) {
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(__double2loint(x)));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(__double2hiint(x)));
}
__device__ __forceinline__ double afetch(int lo, int hi) {
  int l, h;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
  return __hiloint2double(h, l);
}

// one word per group of 64 series: the largest of its wavefronts' span words (k_anchor_spans, words[2 w]); head[0] the largest
// of the launch, head[1] the number of closed groups (diagnostics, as the one-lane path keeps them)
__global__ __launch_bounds__(256) void k_q4_gate(int64_t nwaves, const unsigned long long *__restrict__ words,
                                                 unsigned long long *__restrict__ head, unsigned long long *__restrict__ gate) {
  const int64_t ngroups = (nwaves + 3) / 4;
  double big = 0.0;
  unsigned long long closed = 0;
  for (int64_t g = threadIdx.x; g < ngroups; g += blockDim.x) {
    double m = 0.0;
    for (int64_t w = 4 * g; w < 4 * g + 4 && w < nwaves; ++w) {
      const double x = __longlong_as_double((long long)words[2 * w]);
      m = (x > m || x != x) ? x : m;
      if (x != x) break;
    }
    if (m != m) m = __builtin_inf();
    gate[g] = (unsigned long long)__double_as_longlong(m);
    big = fmax(big, m);
    closed += !(m <= kBackwardGuard);
  }
  __shared__ double sb[256];
  __shared__ unsigned long long sc[256];
  sb[threadIdx.x] = big; sc[threadIdx.x] = closed;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)blockDim.x; ++i) { big = fmax(big, sb[i]); closed += sc[i]; }
    head[0] = (unsigned long long)__double_as_longlong(big);
    head[1] = closed;
  }
}

// ---- coefficient-level form (TT; SURVEY.md section 8f-1, driver.cpp:456-474) -----------------------------------------------------
// A lane's two columns are ONE slot: a complex term (cos / sin columns) or a pair of real terms (J = Jr + 2 Jc = 8 makes Jr
// even), evaluated by the same instructions -- a real pair is a complex term with dc = 0 (cos = 1, sin = 0 exactly), ac = ar_0,
// bc = -ar_1, and V's second column selected to 1.  One branch-free sincos per lane and row; no U / V rows are read, no bU / bV
// rows written: the reverse sweep folds the reverse of the recipe (c2_terms.hip: k_terms_rev) into its step -- three running
// sums per lane, the sum of ba, bx_n = bt_n + sum_k g_nk dc_k (one more group sum per row).  The rates come as the (B, 8) array
// of k_rates.  A group of 64 series with a phase beyond the branch-free range is closed in the gate (k_q4_gate_tt) like one
// beyond the guard, and the caller's composed chain takes it.
struct TermsArgsQ {
  const double *ar, *ac, *bc, *dc;
  int batched, Jc;
};
struct TermsGradsQ {
  double *bar, *bcr, *bac, *bbc, *bcc, *bdc;
};
struct SlotTerm {
  double A, Bq, D, A0;
  bool re;
  __device__ __forceinline__ void load(const TermsArgsQ &T, int64_t b, int jl) {
    const int JC = T.Jc, JR = J - 2 * JC, g = 2 * jl;
    const int64_t br = T.batched ? b * JR : 0, bk = T.batched ? b * JC : 0;
    re = g < JR;
    if (re) {
      A = T.ar[br + g]; Bq = -T.ar[br + g + 1]; D = 0.0;
    } else {
      const int k = (g - JR) >> 1;
      A = T.ac[bk + k]; Bq = T.bc[bk + k]; D = T.dc[bk + k];
    }
    double sum = 0.0;   // driver.cpp:456-458: the sum of ar, then of ac
    for (int r = 0; r < JR; ++r) sum += T.ar[br + r];
    for (int k = 0; k < JC; ++k) sum += T.ac[bk + k];
    A0 = sum;
  }
  // the lane's two columns of U_n (and sin, cos of the slot)
  __device__ __forceinline__ void usc(double x, double (&u)[2], double &sn, double &cs) const {
    const double ph = D * x;
    sincos_cw_fast(ph, sn, cs);
    if (!(fabs(ph) < kSincosFastMax)) sn = cs = __builtin_nan("");   // (an unsorted grid: the gate saw only its ends -- see LaneTerm::uv)
    u[0] = fma(A, cs, Bq * sn);
    u[1] = fma(A, sn, -(Bq * cs));
  }
  // ... and of V_n
  __device__ __forceinline__ void uv(double x, double (&u)[2], double (&v)[2]) const {
    double sn, cs;
    usc(x, u, sn, cs);
    v[0] = cs; v[1] = re ? 1.0 : sn;
  }
};
// k_q4_gate with the phases: a group is also closed (+inf) when dc x of one of its series leaves the range of the branch-free
// sincos (x sorted: the largest |x| of a series sits at one of its ends).  One wavefront per group of 64 series (four wavefronts
// of the pair), a lane per series; head[0], head[1] zeroed by the launcher on the same stream.
__global__ __launch_bounds__(kWave) void k_q4_gate_tt(int64_t B, int64_t N, int64_t nwaves, const unsigned long long *__restrict__ words,
                                                      TermsArgsQ T, const double *__restrict__ x, int64_t x_bs,
                                                      unsigned long long *__restrict__ head, unsigned long long *__restrict__ gate) {
  const int64_t g = blockIdx.x, b = 64 * g + threadIdx.x;
  bool fast = true;
  if (b < B) {
    const double xm = fmax(fabs(x[b * x_bs]), fabs(x[b * x_bs + N - 1]));
    for (int k = 0; k < T.Jc; ++k) fast = fast && (fabs(T.dc[(T.batched ? b * T.Jc : 0) + k]) * xm < kSincosFastMax);
  }
  double m = 0.0;
  if (threadIdx.x < 4 && 4 * g + threadIdx.x < nwaves) {
    m = __longlong_as_double((long long)words[2 * (4 * g + threadIdx.x)]);
    if (m != m) m = __builtin_inf();
  }
#pragma unroll
  for (int sft = 1; sft < kWave; sft <<= 1) m = fmax(m, __shfl_xor(m, sft, kWave));
  if (!__all(fast)) m = __builtin_inf();
  if (threadIdx.x == 0) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);   // (m >= 0: the bit patterns order like the numbers)
    gate[g] = bits;
    atomicMax(head, bits);
    if (!(m <= kBackwardGuard)) atomicAdd(head + 1, 1ull);
  }
}

// =============================================================================
// Forward pass with records, scaled frame.  One step ahead of the chain: ih_{n+1} = exp(-c (t_{n+1} - t_ref)), its
// reciprocal, u-_{n+1} = U_{n+1} ih_{n+1} and its gather through LDS (off the chain).
// =============================================================================
// LN (N even): the rows of U and V arrive as whole aligned 128-byte lines -- rows (2P, 2P+1) of a series share one.  The four
// lanes of a series would request 64 bytes of sixteen different lines per instruction (what this mapping queues on at the
// CU's address unit once every SIMD has its wavefront: profiles/r05_four_lanes.md); instead one instruction fetches the
// pair of eight series (lane l: series (l >> 3) + 8 m of the wavefront, 16-byte piece l & 7), a ring of four pairs in
// registers runs eight rows ahead, and a per-wave LDS tile hands every lane its own two columns one pair ahead of their use.
// R = rows per block of the transposed scalar streams (and of the row ring without LN): 8, or 16 with LN -- then a scalar
// request is one whole 128-byte line per series (four series per instruction) instead of a 64-byte run, and t, a, y enter
// the chip once instead of twice (the second half of a line does not survive eight rows in the cache).
template <bool LN, int R, bool TT = false, bool REC = true>
__global__ __launch_bounds__(kWave, 1) void k_q4_fwd(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                                     const double *__restrict__ c, int64_t c_bs, const double *__restrict__ a,
                                                     const double *__restrict__ U, const double *__restrict__ V,
                                                     const double *__restrict__ y, double *__restrict__ ll,
                                                     int32_t *__restrict__ flag, double2 *__restrict__ ckpt, int64_t nslot,
                                                     double2 *__restrict__ Wrec, double2 *__restrict__ DZst,
                                                     const unsigned long long *__restrict__ gate,
                                                     TermsArgsQ TQ = TermsArgsQ{}) {
  // TT: the coefficient-level form -- `a` is the white-noise diagonal, U / V are not read.  REC = false: log-likelihood only (no
  // W rows, (d, z) pairs, checkpoints)
  static_assert(!TT || !LN, "coefficient-level form: no rows to stage");
  if (!open_group(gate, (int64_t)blockIdx.x * SPW)) return;
  constexpr int NV = R / LG;
  __shared__ __attribute__((aligned(16))) double2 xs2[kWave];
  __shared__ __attribute__((aligned(16))) double sin_[2][3][SPW][R];
  __shared__ __attribute__((aligned(16))) double2 sout[SPW][R];
  const Geo<LG> L(B, LG);
  const int lane = L.lane, jl = L.j, grp = lane / LG;
  const int64_t ot = (int64_t)L.sl * t_bs, on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + 2 * jl;
  const double *tb = t + L.b0 * t_bs + ot, *ab = a + L.b0 * N + on, *yb = y + L.b0 * N + on;
  const double2 *Ub = reinterpret_cast<const double2 *>(U + L.b0 * N * J + oj);   // row stride LG double2
  const double2 *Vb = reinterpret_cast<const double2 *>(V + L.b0 * N * J + oj);
  const double cj[2] = {c[L.b * c_bs + 2 * jl], c[L.b * c_bs + 2 * jl + 1]};
  // complex terms come as pairs of equal rates (terms.py:171-173): when BOTH columns of every lane of the wavefront share
  // their rate, one exponential and one reciprocal per row serve the two (uniform branch; any other model takes both)
  const bool ceq = __all(cj[0] == cj[1]);
  SlotTerm stm;
  if constexpr (TT) stm.load(TQ, L.b, jl);
  double2 *ckw = ckpt + (size_t)blockIdx.x * nslot * (kCkD2 * kWave);
  double2 *wrp = Wrec + (size_t)blockIdx.x * N * kWave + lane;
  double2 *dzst = DZst + L.b0 * N + on;

  double SX[2][J];
#pragma unroll
  for (int q = 0; q < J; ++q) { SX[0][q] = 0.0; SX[1][q] = 0.0; }
  // Row 0 is the first step of block 0 from the neutral state "row -1" (S = 0, F = 0, W = 0, z = 0, d = 1 at t_0: the step
  // yields d_0 = a_0, W_0 = V_0 / d_0, z_0 = y_0, forward.hpp:107-108), so blocks cover rows [R b, R b + R) and every
  // transposed request of t, a, y / store of (d, z) is a whole aligned run (see k_loglik_fwd, profiles/r05_alignment.md)
  double d = 1.0;
  double rd = 1.0;
  double w[2] = {0.0, 0.0};
  double z = 0.0;
  double F[2] = {0.0, 0.0};
  double prod = 1.0;
  int eacc = 0;
  double quad = 0.0;
  int32_t fl = 0;

  // Per-series scalar streams move TRANSPOSED, eight lanes per series: one instruction fetches eight consecutive rows of
  // eight series (64-byte runs; lane l: series (l >> 3) + 8 m of the wavefront, row l & 7), two cover the sixteen.
  static_assert(R == 8 || R == 16, "blocks of 8 or 16 rows");
  constexpr int SPI = kWave / R;        // series per scalar instruction (R lanes per series: rows of a block)
  const int srow = lane & (R - 1);
  int ssl[NV];
  const double *tb8[NV], *ab8[NV], *yb8[NV];
  double2 *dz8[NV];
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    ssl[m] = lane / R + SPI * m;
    const int64_t bs = (L.b0 + ssl[m] < B) ? L.b0 + ssl[m] : B - 1;
    tb8[m] = t + bs * t_bs; ab8[m] = a + bs * N; yb8[m] = y + bs * N;
    dz8[m] = DZst + bs * N;
  }
  // (LN) row pairs: eight lanes per series (piece lane & 7 of the 128 bytes), two instructions for the sixteen series
  const double2 *Ul8[2], *Vl8[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int64_t bs = (L.b0 + (lane >> 3) + 8 * m < B) ? L.b0 + (lane >> 3) + 8 * m : B - 1;
    Ul8[m] = reinterpret_cast<const double2 *>(U + bs * N * J) + (lane & 7);
    Vl8[m] = reinterpret_cast<const double2 *>(V + bs * N * J) + (lane & 7);
  }
  double vt[NV], va[NV], vy[NV];
  auto vload = [&](int64_t nb) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t row = nb + srow;
      row = (row < N) ? row : N - 1;
      vt[m] = tb8[m][row]; va[m] = ab8[m][row]; vy[m] = yb8[m][row];
    }
  };
  auto vstage = [&](int q) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      sin_[q][0][ssl[m]][srow] = vt[m]; sin_[q][1][ssl[m]][srow] = va[m]; sin_[q][2][ssl[m]][srow] = vy[m];
    }
  };
  vload(0); vstage(0);
  vload(R); vstage(1);
  vload(2 * R);

  double2 ru[LN ? 1 : R], rv[LN ? 1 : R];
  const double2 *up = Ub, *vp = Vb;   // row n0 of the current block
  auto load_row = [&](int r, int ahead, int64_t n, bool clamp) {
    if constexpr (!LN && !TT) {
      int64_t o = ahead;
      if (clamp && n >= N) o -= n - (N - 1);
      ru[r] = up[o * LG]; rv[r] = vp[o * LG];
    }
  };
  // LN: ring slot P % 4 holds the pieces of pair P (two instructions: series slots 0 - 7 and 8 - 15 of the wavefront)
  __shared__ __attribute__((aligned(16))) double2 ltile[LN ? 2 : 1][LN ? 2 * kWave : 1];   // [U | V][series][piece]
  const int64_t plast = N / 2 - 1;
  double qux[LN ? 4 : 1][2], quy[LN ? 4 : 1][2], qvx[LN ? 4 : 1][2], qvy[LN ? 4 : 1][2];
  double cu[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, cv[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // [row of the pair][column]: current pair
  double nu[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, nv[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // ... the next one
  auto pair_load = [&](int slot, int64_t P) {
    const int64_t Pc = P < plast ? P : plast;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const double2 a2 = Ul8[m][Pc * 8], b2 = Vl8[m][Pc * 8];
      qux[slot][m] = a2.x; quy[slot][m] = a2.y; qvx[slot][m] = b2.x; qvy[slot][m] = b2.y;
    }
  };
  auto pair_stage = [&](int slot) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      ltile[0][m * kWave + lane] = make_double2(qux[slot][m], quy[slot][m]);
      ltile[LN ? 1 : 0][m * kWave + lane] = make_double2(qvx[slot][m], qvy[slot][m]);
    }
  };
  auto pair_own = [&]() {   // the lane's two columns of the staged pair's two rows
    const double2 a0 = ltile[0][grp * 8 + jl], a1 = ltile[0][grp * 8 + 4 + jl];
    const double2 b0 = ltile[LN ? 1 : 0][grp * 8 + jl], b1 = ltile[LN ? 1 : 0][grp * 8 + 4 + jl];
    nu[0][0] = a0.x; nu[0][1] = a0.y; nu[1][0] = a1.x; nu[1][1] = a1.y;
    nv[0][0] = b0.x; nv[0][1] = b0.y; nv[1][0] = b1.x; nv[1][1] = b1.y;
  };
  if constexpr (LN) {
    const double2 u0 = Ub[0], u1 = Ub[LG], v0 = Vb[0], w1 = Vb[LG];
    cu[0][0] = u0.x; cu[0][1] = u0.y; cu[1][0] = u1.x; cu[1][1] = u1.y;
    cv[0][0] = v0.x; cv[0][1] = v0.y; cv[1][0] = w1.x; cv[1][1] = w1.y;   // pair 0 (rows 0 and 1: the first two steps)
    pair_load(1, 1); pair_load(2, 2); pair_load(3, 3); pair_load(0, 4);   // (the first step stages pair 1 and requests pair 5)
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) load_row(r, r, r, true);
  }

  lds_order();
  double tref = tb[0];                    // reference time of the current frame (row 0, then every anchor row)
  double tnext = sin_[0][0][grp][0];
  // row n = 0, prepared: ih_n = h_n = 1 (the reference itself), u-_n (own pair) and its gather
  double ihc[2] = {1.0, 1.0}, hc[2] = {1.0, 1.0};
  double uc[2] = {TT ? 0.0 : (LN ? cu[0][0] : ru[0].x), TT ? 0.0 : (LN ? cu[0][1] : ru[0].y)};
  double vcur[2] = {0.0, 0.0};   // (TT) the lane's columns of V of the current row
  if constexpr (TT) stm.uv(tnext, uc, vcur);
  double hp[2] = {1.0, 1.0};              // h of the row the chain starts from (row 0: the reference itself)
  double uXc[J];
  xs2[lane] = make_double2(uc[0], uc[1]);
  lds_order();
  xg2_lds(xs2, lane, uXc);
  lds_order();

  auto block = [&](int64_t n0, int q, bool anchor_block, auto checked_tag) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = n0 + r;
      if (!CHECKED || n < N) {
        const double tn = tnext, yn = sin_[q][2][grp][r];
        const double an = TT ? sin_[q][1][grp][r] + stm.A0 : sin_[q][1][grp][r];
        double vv_[2], ur[2];   // V_n and U_{n+1} of the lane
        const int rn = (r + 1) % R;
        if constexpr (LN) {
          // blocks start at multiples of eight: r even <=> n even <=> the first row of pair P = n / 2
          if (r % 2 == 0) {
            vv_[0] = cv[0][0]; vv_[1] = cv[0][1]; ur[0] = cu[1][0]; ur[1] = cu[1][1];
            lds_order();
            pair_stage((r / 2 + 1) % 4);                            // pair P + 1 -> tile (read back at the end of this step)
            pair_load((r / 2 + 1) % 4, n / 2 + 5);                  // its slot: pair P + 5, eight rows ahead
          } else {
            vv_[0] = cv[1][0]; vv_[1] = cv[1][1]; ur[0] = nu[0][0]; ur[1] = nu[0][1];
          }
        } else if constexpr (!TT) {
          vv_[0] = rv[LN ? 0 : r].x; vv_[1] = rv[LN ? 0 : r].y; ur[0] = ru[LN ? 0 : rn].x; ur[1] = ru[LN ? 0 : rn].y;
        }
        const double tn1 = (r + 1 < R) ? sin_[q][0][grp][r + 1] : sin_[q ^ 1][0][grp][0];
        if constexpr (TT) {   // row n + 1 formed here (beyond the last row: the clamped time, unused)
          vv_[0] = vcur[0]; vv_[1] = vcur[1];
          stm.uv(tn1, ur, vcur);
        }
        // (a) the next row's frame factors and u- -> LDS -> gather.  Behind an anchor row the next row lives in the new frame.
        const double trn = (r == 0 && anchor_block) ? tn : tref;
        double ihn[2], hn[2];
        ihn[0] = exp_decay(cj[0] * (trn - tn1)); hn[0] = rcp_nr(ihn[0]);
        if (ceq) { ihn[1] = ihn[0]; hn[1] = hn[0]; }
        else { ihn[1] = exp_decay(cj[1] * (trn - tn1)); hn[1] = rcp_nr(ihn[1]); }
        const double un[2] = {ur[0] * ihn[0], ur[1] * ihn[1]};
        xs2[lane] = make_double2(un[0], un[1]);
        lds_order();
        double uXn[J];
        xg2_lds(xs2, lane, uXn);
        lds_order();
        // (b) the chain of row n
        const double wt[2] = {hp[0] * w[0], hp[1] * w[1]};
        double wX[J];
        xg2(wt[0], wt[1], wX);
        const double dw0 = d * wt[0], dw1 = d * wt[1];
        double t0a = 0.0, t0b = 0.0, t1a = 0.0, t1b = 0.0;
#pragma unroll
        for (int qq = 0; qq < J; ++qq) {
          const double s0 = fma(dw0, wX[qq], SX[0][qq]);     // S^ += d w~^T w~          (forward.hpp:115-123, scaled)
          const double s1 = fma(dw1, wX[qq], SX[1][qq]);
          SX[0][qq] = s0;
          SX[1][qq] = s1;
          if (qq & 1) { t0b = fma(uXc[qq], s0, t0b); t1b = fma(uXc[qq], s1, t1b); }
          else { t0a = fma(uXc[qq], s0, t0a); t1a = fma(uXc[qq], s1, t1a); }
        }
        const double th0 = t0a + t0b, th1 = t1a + t1b;       // u- S^ = h tau                      (forward.hpp:126)
        F[0] = fma(wt[0], z, F[0]);                          // F~ += w~ z_{n-1}                   (internal.hpp:140-143)
        F[1] = fma(wt[1], z, F[1]);
        double rd_ = fma(th0, uc[0], th1 * uc[1]), rz_ = fma(uc[0], F[0], uc[1] * F[1]);
        gsum2<LG>(rd_, rz_);
        const double dn = an - rd_, zn = yn - rz_;           // forward.hpp:127, internal.hpp:144
        rd = rcp_nr(dn);
        w[0] = fma(-ihc[0], th0, vv_[0]) * rd;               // W_n = (V_n - tau_n) / d_n          (forward.hpp:131)
        w[1] = fma(-ihc[1], th1, vv_[1]) * rd;
        d = dn;
        z = zn;
        if constexpr (REC) {
          wrp[(size_t)n * kWave] = make_double2(w[0], w[1]);
          sout[grp][r] = make_double2(d, z);
        }
        load_row(r, r + R, n + R, CHECKED);
        fl = ((fl == 0) & (d <= 0.0)) ? (int32_t)n : fl;     // forward.hpp:128 (no early exit: outputs are flagged)
        prod *= d;
        quad = fma(z * z, rd, quad);
        if (r % 2 == 1 || r == R - 1) {
          int e;
          prod = frexp(prod, &e);
          eacc += e;
        }
        if (r == 0 && anchor_block) {
          // anchor row: back to the plain state S = H^-1 S^ H^-1, F = F~ / h -- the checkpoint the reverse sweep re-anchors
          // at (state after row n = 32 i -> slot i - 1) and the reference of the next frame
          double iX[J];
          xg2(ihc[0], ihc[1], iX);
#pragma unroll
          for (int qq = 0; qq < J; ++qq) { SX[0][qq] *= iX[qq] * ihc[0]; SX[1][qq] *= iX[qq] * ihc[1]; }
          F[0] *= ihc[0]; F[1] *= ihc[1];
          if constexpr (REC) ck_store(ckw + (size_t)(n / (A * C) - 1) * (kCkD2 * kWave), lane, SX, F);
          tref = tn;
          hp[0] = 1.0; hp[1] = 1.0;
        } else {
          hp[0] = hc[0]; hp[1] = hc[1];
        }
        tnext = tn1;
        ihc[0] = ihn[0]; ihc[1] = ihn[1]; hc[0] = hn[0]; hc[1] = hn[1]; uc[0] = un[0]; uc[1] = un[1];
        if (LN && r % 2 == 1) {
#pragma unroll
          for (int e = 0; e < 2; ++e) { cu[e][0] = nu[e][0]; cu[e][1] = nu[e][1]; cv[e][0] = nv[e][0]; cv[e][1] = nv[e][1]; }
        }
        if (LN && r % 2 == 0) {   // own columns of pair P + 1: used from the next step on
          lds_order();
          pair_own();
        }
#pragma unroll
        for (int k = 0; k < J; ++k) uXc[k] = uXn[k];
      }
    }
    lds_order();
#pragma unroll
    for (int m = 0; m < NV; ++m) {   // (slots beyond the batch hold copies of its last series: same values, same addresses)
      if constexpr (REC) { if (!CHECKED || n0 + srow < N) dz8[m][n0 + srow] = sout[ssl[m]][srow]; }
    }
    vstage(q);
    vload(n0 + 3 * R);
    lds_order();
  };
  int64_t n0 = 0, blk = 0;
  int q = 0;
  auto advance = [&]() { up += R * LG; vp += R * LG; q ^= 1; ++blk; };
  constexpr int BA = A * C / R;   // blocks between two anchors (32 rows); an anchor row 32 i is the FIRST row of its block
  for (; n0 + 2 * R <= N; n0 += R) { block(n0, q, blk % BA == 0 && blk > 0, std::false_type{}); advance(); }   // every row load in range
  for (; n0 < N; n0 += R) { block(n0, q, blk % BA == 0 && blk > 0, std::true_type{}); advance(); }

  {  // the state after the last row, plain (hp = h of the last row in the current frame; 1 right behind an anchor)
    const double il[2] = {rcp_nr(hp[0]), rcp_nr(hp[1])};
    double iX[J];
    xg2(il[0], il[1], iX);
#pragma unroll
    for (int qq = 0; qq < J; ++qq) { SX[0][qq] *= iX[qq] * il[0]; SX[1][qq] *= iX[qq] * il[1]; }
    F[0] *= il[0]; F[1] *= il[1];
    if constexpr (REC) ck_store(ckw + (size_t)(nslot - 1) * (kCkD2 * kWave), lane, SX, F);
  }
  if (L.valid && jl == 0) {
    flag[L.b] = fl;
    int e;
    prod = frexp(prod, &e);
    const double logdet = log(prod) + (double)(eacc + e) * kLn2;
    ll[L.b] = fl ? -INFINITY : -0.5 * (logdet + (double)N * kLog2Pi) - 0.5 * quad;
  }
}

// =============================================================================
// Reverse sweep: backward recursion + adjoint recursion in the scaled frame of the anchor ABOVE (g_n = exp(-c (t_ref - t_n))
// <= 1), segments of C rows, last first.  Step n (internal.hpp:225-245 fused with reverse.hpp:52-84; see c2_loglik.hip):
//   u- = U_n / g_n, w~ = W_{n-1} g_{n-1}, x- = bV- + 2 ba u-      (the three gathered vectors)
//   bU_n = -(bz F~ + x- S^) / g_n;  bF- -= u- bz;  M^ -= u-^T bV- + x-^T u-;  bp = F~ bF- + diag(S^ M^)
//   q^ = w~ M^;  G = w~ . bF-;  Q = q^ . w~;  bz_{n-1} = G - z/d;  bV-_{n-1} = (z/d) bF- + q^;  ba_{n-1} = ...
//   S^_{n-1} = S^_n - d_{n-1} w~^T w~;  F~_{n-1} = F~_n - w~ z_{n-1}
// =============================================================================
// LN (N even): rows of U, bU, bV as whole 128-byte lines through LDS tiles (see k_q4_fwd): four line pairs of U per segment
// (rows 8k .. 8k+7, two instructions each; row 8k is handed down to the segment below); a lane's bU / bV columns go into a
// two-row tile and a completed pair leaves, during the step after, as two stores of eight whole lines each.
template <bool LN, bool TT = false>
__global__ __launch_bounds__(kWave, 1) void k_q4_rev(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                                     const double *__restrict__ c, int64_t c_bs, const double *__restrict__ U,
                                                     const double2 *__restrict__ Wrec, const double2 *__restrict__ DZst,
                                                     const double2 *__restrict__ ckpt, int64_t nslot, int64_t nseg,
                                                     const int32_t *__restrict__ flag, double *__restrict__ bt,
                                                     double *__restrict__ bc, double *__restrict__ ba, double *__restrict__ bU,
                                                     double *__restrict__ bV, double *__restrict__ by,
                                                     const unsigned long long *__restrict__ gate,
                                                     TermsArgsQ TQ = TermsArgsQ{}, TermsGradsQ GQ = TermsGradsQ{}) {
  // TT (coefficient-level form): bt, ba, by are bx, bdiag, by; U, bU, bV, bc are not touched; GQ takes the coefficient gradients
  static_assert(!TT || !LN, "coefficient-level form: no rows to stage");
  if (!open_group(gate, (int64_t)blockIdx.x * SPW)) return;
  constexpr int NV = C / LG;
  __shared__ __attribute__((aligned(16))) double rowT[C + 1][SPW], rowD[C + 1][SPW], rowR[C + 1][SPW], rowZ[C + 1][SPW];
  // scalar outputs leave as whole aligned runs of C rows: ba_{n-1}, by_{n-1} (complete at the end of step n) as rows C k ..
  // C k + C - 1 of segment k; bt_n (complete at step n) one segment late -- the segment's top row with the C - 1 rows the
  // segment above left in the other buffer (see k_loglik_rev)
  __shared__ __attribute__((aligned(16))) double oBA[SPW][C], oBT[2][SPW][C], oBY[SPW][C];
  const Geo<LG> L(B, LG);
  const int lane = L.lane, jl = L.j, grp = lane / LG;
  const int64_t ot = (int64_t)L.sl * t_bs, on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + 2 * jl;
  const double *tb = t + L.b0 * t_bs + ot;
  const double2 *Ub = reinterpret_cast<const double2 *>(U + L.b0 * N * J + oj);
  const double2 *wrp = Wrec + (size_t)blockIdx.x * N * kWave + lane;
  const double2 *dzb = DZst + L.b0 * N + on;
  const double2 *ckw = ckpt + (size_t)blockIdx.x * nslot * (kCkD2 * kWave);
  double *btb = bt + L.b0 * N + on, *bab = ba + L.b0 * N + on, *byb = by + L.b0 * N + on;
  double2 *bUb = reinterpret_cast<double2 *>(bU + L.b0 * N * J + oj);
  double2 *bVb = reinterpret_cast<double2 *>(bV + L.b0 * N * J + oj);
  const double cj[2] = {c[L.b * c_bs + 2 * jl], c[L.b * c_bs + 2 * jl + 1]};
  // (lanes beyond the batch walk a copy of its last series: identical values to identical addresses)
  // Failed factorisation: NaN gradients for this series, never stale memory (see k_loglik_rev).  Its lanes stay in the
  // sweep all the same -- they fetch and flush scalar rows for OTHER series of the wavefront (eight lanes per series
  // below) -- with every store that belongs to the failed series switched off.
  const bool ceq = __all(cj[0] == cj[1]);   // (see k_q4_fwd)
  const bool alive = flag[L.b] == 0;
  if (!alive) {
    const double nan = __builtin_nan("");
    for (int64_t n = jl; n < N; n += LG) { btb[n] = nan; bab[n] = nan; byb[n] = nan; }
    if constexpr (!TT) {
      for (int64_t n = 0; n < N; ++n) { bUb[n * LG] = make_double2(nan, nan); bVb[n * LG] = make_double2(nan, nan); }
      bc[L.b * J + 2 * jl] = nan; bc[L.b * J + 2 * jl + 1] = nan;
    }
  }
  SlotTerm stm;
  if constexpr (TT) stm.load(TQ, L.b, jl);
  // (TT) running sums of the lane's slot: [0] bac (a real pair: sum bU_0), [1] bbc (-sum bU_1), [2] bdc; the sum of ba; sin / cos
  // of every row of the segment parked next to u- and w~
  double acc[3] = {0.0, 0.0, 0.0}, sba = 0.0;
  int scAlo[TT ? C : 1][2], scAhi[TT ? C : 1][2];

  double MX[2][J];
#pragma unroll
  for (int q = 0; q < J; ++q) { MX[0][q] = 0.0; MX[1][q] = 0.0; }
  double bF[2] = {0.0, 0.0}, bcj[2] = {0.0, 0.0}, bVn[2] = {0.0, 0.0};
  double carry = 0.0, ban = 0.0, bzn = 0.0;

  // scalar streams transposed, eight lanes per series (see k_q4_fwd)
  const int srow = lane & 7;
  int ssl[NV];
  bool ok8[NV];
  const double *tb8[NV];
  const double2 *dzb8[NV];
  double *bab8[NV], *btb8[NV], *byb8[NV];
  const double2 *Ul8[NV];
  double2 *bUl8[NV], *bVl8[NV];
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    ssl[m] = (lane >> 3) + 8 * m;
    const int64_t bs = (L.b0 + ssl[m] < B) ? L.b0 + ssl[m] : B - 1;
    ok8[m] = flag[bs] == 0;
    tb8[m] = t + bs * t_bs; dzb8[m] = DZst + bs * N;
    bab8[m] = ba + bs * N; btb8[m] = bt + bs * N; byb8[m] = by + bs * N;
    Ul8[m] = reinterpret_cast<const double2 *>(U + bs * N * J) + srow;
    bUl8[m] = reinterpret_cast<double2 *>(bU + bs * N * J) + srow;
    bVl8[m] = reinterpret_cast<double2 *>(bV + bs * N * J) + srow;
  }
  const int64_t plast = N / 2 - 1;
  __shared__ __attribute__((aligned(16))) double2 utile[LN ? 4 : 1][LN ? 2 * kWave : 1];        // [pair][series][piece]
  __shared__ __attribute__((aligned(16))) double2 otile[LN ? 2 : 1][2][LN ? 2 * kWave : 1];     // [pair parity][bU | bV][series][piece]
  double qux[LN ? 4 : 1][NV], quy[LN ? 4 : 1][NV], ucar[2] = {0.0, 0.0};
  double vt[NV], vd[NV], vz[NV];
  double iu[C][2], iw[C][2];   // plain doubles: arrays of double2 end up in scratch
  double cS[2][J], cF[2];
  auto load_segment = [&](int64_t k) {
    const int64_t n_lo = 1 + k * C;
    const bool full = n_lo + C <= N;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t row = n_lo - 1 + srow;
      row = (row < N) ? row : N - 1;
      vt[m] = tb8[m][row];
      const double2 dz = dzb8[m][row];
      vd[m] = dz.x; vz[m] = dz.y;
    }
    if constexpr (LN) {   // pairs 4k .. 4k+3 = rows 8k .. 8k+7 (beyond the last pair: that one again, unused)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t P = 4 * k + i, Pc = P < plast ? P : plast;
#pragma unroll
        for (int m = 0; m < NV; ++m) { const double2 u2 = Ul8[m][Pc * 8]; qux[i][m] = u2.x; quy[i][m] = u2.y; }
      }
    }
#pragma unroll
    for (int r = 0; r < C; ++r) {
      const int64_t n = (full || n_lo + r < N) ? n_lo + r : N - 1;
      if constexpr (!LN && !TT) { const double2 u2 = Ub[n * LG]; iu[r][0] = u2.x; iu[r][1] = u2.y; }
      const double2 w2 = wrp[(size_t)(n - 1) * kWave];
      iw[r][0] = w2.x; iw[r][1] = w2.y;
    }
    // the state at the END of the segment where that is an anchor: every A-th segment boundary, the last row
    if (k == nseg - 1) ck_load(ckw + (size_t)(nslot - 1) * (kCkD2 * kWave), lane, cS, cF);
    else if ((k + 1) % A == 0) ck_load(ckw + (size_t)((k + 1) / A - 1) * (kCkD2 * kWave), lane, cS, cF);
  };

  double carT = tb[N - 1];
  double2 carDZ = dzb[N - 1];
  double carR = rcp_nr(carDZ.x);
  if (nseg > 0) load_segment(nseg - 1);

  int uAlo[C][2], uAhi[C][2], wAlo[C][2], wAhi[C][2];   // u-, w~ of the segment's rows wait in accumulation registers
  double SX[2][J], F[2] = {0.0, 0.0};
#pragma unroll
  for (int q = 0; q < J; ++q) { SX[0][q] = 0.0; SX[1][q] = 0.0; }
  double tref = 0.0, gtop[2] = {1.0, 1.0}, igtop[2] = {1.0, 1.0};
  double hA[NV], hY[NV], hT[NV];   // the held upper halves (see the flush at the end of a segment)
  bool hAok[NV], hTok[NV];
#pragma unroll
  for (int m = 0; m < NV; ++m) { hA[m] = hY[m] = hT[m] = 0.0; hAok[m] = hTok[m] = false; }
  int bq = 0;   // buffer of oBT the current segment writes
  for (int64_t k = nseg - 1; k >= 0; --k) {
    const int64_t n_lo = 1 + k * C;
    const int cnt = (N - n_lo < C) ? (int)(N - n_lo) : C;
    // ---- phase A: scalar rows to LDS; change of frame at an anchor; frame factors of the segment's rows -------
    if constexpr (LN) {   // the segment's U lines -> tiles (read back as the lanes' own columns behind the exponentials)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < NV; ++m) utile[i][m * kWave + lane] = make_double2(qux[i][m], quy[i][m]);
    }
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      rowT[srow][ssl[m]] = vt[m]; rowD[srow][ssl[m]] = vd[m]; rowR[srow][ssl[m]] = rcp_nr(vd[m]); rowZ[srow][ssl[m]] = vz[m];
    }
    lds_order();
    rowT[cnt][grp] = carT; rowD[cnt][grp] = carDZ.x; rowR[cnt][grp] = carR; rowZ[cnt][grp] = carDZ.y;
    lds_order();
    const bool anchor = (k + 1) % A == 0 || k == nseg - 1;   // (uniform)
    if (anchor) {
      double gX[J];
      xg2(gtop[0], gtop[1], gX);
#pragma unroll
      for (int q = 0; q < J; ++q) { MX[0][q] *= gX[q] * gtop[0]; MX[1][q] *= gX[q] * gtop[1]; }
      bF[0] *= gtop[0]; bF[1] *= gtop[1];
      bVn[0] *= gtop[0]; bVn[1] *= gtop[1];
      tref = rowT[cnt][grp];
      gtop[0] = gtop[1] = 1.0; igtop[0] = igtop[1] = 1.0;
#pragma unroll
      for (int q = 0; q < J; ++q) { SX[0][q] = cS[0][q]; SX[1][q] = cS[1][q]; }
      F[0] = cF[0]; F[1] = cF[1];
    }
    double gv[C][2], igv[C][2];   // g, 1 / g of rows n_lo-1 .. n_lo+C-2 (entry r = row n_lo-1+r)
#pragma unroll
    for (int r = 0; r < C; ++r) {
      // (rows of a short last segment beyond its end repeat the last row -- clamped loads -- so g = 1 there)
      const double tm = rowT[r][grp];
      gv[r][0] = exp_decay(cj[0] * (tm - tref));
      igv[r][0] = rcp_nr(gv[r][0]);
      if (ceq) { gv[r][1] = gv[r][0]; igv[r][1] = igv[r][0]; }
      else { gv[r][1] = exp_decay(cj[1] * (tm - tref)); igv[r][1] = rcp_nr(gv[r][1]); }
    }
    if constexpr (LN) {   // rows 8k+1 .. 8k+7 from this segment's lines, row 8k+8 handed down by the segment above
      lds_order();
      iu[C - 1][0] = ucar[0]; iu[C - 1][1] = ucar[1];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const double2 a0 = utile[i][grp * 8 + jl], a1 = utile[i][grp * 8 + 4 + jl];
        if (i == 0) { ucar[0] = a0.x; ucar[1] = a0.y; }
        else { iu[2 * i - 1][0] = a0.x; iu[2 * i - 1][1] = a0.y; }
        iu[2 * i][0] = a1.x; iu[2 * i][1] = a1.y;
      }
    }
    if constexpr (TT) {   // the lane's columns of U_n of rows n_lo .. n_lo + C - 1 (entry r + 1 of rowT)
#pragma unroll
      for (int r = 0; r < C; ++r) {
        double sn, cs;
        stm.usc(rowT[r + 1][grp], iu[r], sn, cs);
        apark(sn, scAlo[r][0], scAhi[r][0]);
        apark(cs, scAlo[r][1], scAhi[r][1]);
      }
    }
#pragma unroll
    for (int r = 0; r < C; ++r) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const double ign = (r == C - 1) ? igtop[m] : igv[r + 1 < C ? r + 1 : 0][m];
        apark(iu[r][m] * ign, uAlo[r][m], uAhi[r][m]);
        apark(iw[r][m] * gv[r][m], wAlo[r][m], wAhi[r][m]);
      }
    }
    if (k == nseg - 1) {   // cotangents of the last row: pure seeds
      const double rdl = rowR[cnt][grp], zl = rowZ[cnt][grp];
      ban = 0.5 * rdl * (zl * zl * rdl - 1.0);
      bzn = -zl * rdl;
      bVn[0] = 0.0; bVn[1] = 0.0;
      if (alive) { byb[N - 1] = bzn; bab[N - 1] = ban; }
    }
    carT = rowT[0][grp]; carDZ = make_double2(rowD[0][grp], rowZ[0][grp]); carR = rowR[0][grp];
    lds_order();

    // ---- phase C: fused reverse + backward-recursion steps; the next (earlier) segment is fetched half way through ----
#pragma unroll
    for (int r = C - 1; r >= 0; --r) {
      if (r == C / 2 - 1) {
        if (k > 0) load_segment(k - 1);
      }
      if (r < cnt) {
        const int64_t n = n_lo + r;
        const double rdm = rowR[r][grp], zm = rowZ[r][grp], dt = rowT[r][grp] - rowT[r + 1][grp];
        double u[2], wm[2], xv[2], gn[2], ign[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          u[m] = afetch(uAlo[r][m], uAhi[r][m]);
          wm[m] = afetch(wAlo[r][m], wAhi[r][m]);
          gn[m] = (r == C - 1) ? gtop[m] : gv[r + 1 < C ? r + 1 : 0][m];
          ign[m] = (r == C - 1) ? igtop[m] : igv[r + 1 < C ? r + 1 : 0][m];
          xv[m] = fma(2.0 * ban, u[m], bVn[m]);
        }
        double uX[J], wX[J], xX[J];
        xg2(u[0], u[1], uX);
        xg2(wm[0], wm[1], wX);
        // (LN) rows n -- odd first, then even -- of a pair fill its tile; blocks start at odd rows: r odd <=> n even
        double2 *ob = &otile[LN ? ((r + 1) / 2) & 1 : 0][0][grp * 8 + ((r & 1) ? 0 : 4) + jl];
        if constexpr (LN) {
          if (r % 2 == 0 && r + 1 < cnt) {   // the pair completed by the step before (row n + 1, even) leaves now
            lds_order();
            const int64_t P = (n + 1) / 2;
#pragma unroll
            for (int m = 0; m < NV; ++m) {
              if (ok8[m]) {
                bUl8[m][P * 8] = otile[((r + 2) / 2) & 1][0][m * kWave + lane];
                bVl8[m][P * 8] = otile[((r + 2) / 2) & 1][1][m * kWave + lane];
              }
            }
          }
        } else if constexpr (!TT) {
          if (alive) bVb[n * LG] = make_double2(bVn[0] * gn[0], bVn[1] * gn[1]);
        }
        const double bVo[2] = {bVn[0] * gn[0], bVn[1] * gn[1]};
        const double ba_in = ban;   // ba_n
        xg2(xv[0], xv[1], xX);
        double bpt[2], qv[2], bUo[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const double bU1 = -bzn * F[m];                 // internal.hpp:232
          bF[m] = fma(-u[m], bzn, bF[m]);                 // internal.hpp:233
          const double bp_s = F[m] * bF[m];               // internal.hpp:236
          double xs0 = 0.0, xs1 = 0.0, bp0 = 0.0, bp1 = 0.0, q0 = 0.0, q1 = 0.0;
#pragma unroll
          for (int q = 0; q < J; ++q) {
            double mm = fma(-uX[q], bVn[m], MX[m][q]);    // reverse.hpp:67-68 on M = bS + bS^T, scaled
            mm = fma(-xX[q], u[m], mm);
            MX[m][q] = mm;
            if (q & 1) { xs1 = fma(xX[q], SX[m][q], xs1); bp1 = fma(SX[m][q], mm, bp1); q1 = fma(wX[q], mm, q1); }
            else { xs0 = fma(xX[q], SX[m][q], xs0); bp0 = fma(SX[m][q], mm, bp0); q0 = fma(wX[q], mm, q0); }
          }
          bUo[m] = ign[m] * (bU1 - (xs0 + xs1));          // reverse.hpp:66 + internal.hpp:232
          bpt[m] = bp_s + (bp0 + bp1);
          bcj[m] = fma(dt, bpt[m], bcj[m]);
          qv[m] = q0 + q1;
        }
        double gsq = 0.0;   // (TT) sum_k g_nk dc_k of row n: what bx_n has on top of bt_n
        if constexpr (LN) { ob[0] = make_double2(bUo[0], bUo[1]); ob[2 * kWave] = make_double2(bVo[0], bVo[1]); }
        else if constexpr (TT) {   // the reverse of the recipe for row n, the lane's slot
          const double sn = afetch(scAlo[r][0], scAhi[r][0]), cs = afetch(scAlo[r][1], scAhi[r][1]);
          const double u0 = u[0] * gn[0], u1 = u[1] * gn[1];   // U_n (u is u- = U_n / g_n)
          acc[0] = fma(bUo[0], cs, fma(bUo[1], sn, acc[0]));
          acc[1] = fma(bUo[0], sn, fma(-bUo[1], cs, acc[1]));
          const double g = fma(-bUo[0], u1, fma(bUo[1], u0, fma(-bVo[0], sn, bVo[1] * cs)));   // cotangent of the phase dc x_n
          acc[2] = fma(g, rowT[r + 1][grp], acc[2]);
          gsq = gsum<LG>(g * stm.D);
          sba += ba_in;
        }
        else if (alive) bUb[n * LG] = make_double2(bUo[0], bUo[1]);
        double f = fma(cj[0], bpt[0], cj[1] * bpt[1]), Gs = fma(wm[0], bF[0], wm[1] * bF[1]),
               Q = fma(qv[0], wm[0], qv[1] * wm[1]);
        gsum3<LG>(f, Gs, Q);
        oBT[bq][grp][r] = TT ? carry - f + gsq : carry - f;   // (TT: bx_n)
        carry = f;
        const double zr = zm * rdm;
        bzn = Gs - zr;
        oBY[grp][r] = bzn;
        bVn[0] = fma(zr, bF[0], qv[0]);
        bVn[1] = fma(zr, bF[1], qv[1]);
        ban = 0.5 * rdm * (zm * zr - 1.0) - 0.5 * Q - zr * Gs;
        oBA[grp][r] = ban;                                   // ba_{n-1}
        const double dw0 = rowD[r][grp] * wm[0], dw1 = rowD[r][grp] * wm[1];   // the state of row n-1
#pragma unroll
        for (int q = 0; q < J; ++q) { SX[0][q] = fma(-dw0, wX[q], SX[0][q]); SX[1][q] = fma(-dw1, wX[q], SX[1][q]); }
        F[0] = fma(-wm[0], zm, F[0]);
        F[1] = fma(-wm[1], zm, F[1]);
      }
    }
    gtop[0] = gv[0][0]; gtop[1] = gv[0][1]; igtop[0] = igv[0][0]; igtop[1] = igv[0][1];
    lds_order();
    // The scalar gradients of a segment are an aligned run of 8 rows = HALF a 128-byte line per series.  Written when they are ready,
    // the two halves of a line reach memory eight steps apart -- the line has left L2 in between and each half is merged on the
    // memory side on its own (a read-modify-write: 11 - 15 % of the one-lane reverse sweep, profiles/r06_halflines.md).  So the
    // UPPER half (rows 16 i + 8 ..) waits in a register per lane and stream and leaves with the lower half, back to back
    // (C2Q4_PAIRLINES=0: as they come).
#pragma unroll
    for (int m = 0; m < NV; ++m) {   // (slots beyond the batch hold copies of its last series: same values, same addresses)
      if (ok8[m]) {
        const double vA = oBA[ssl[m]][srow], vY = oBY[ssl[m]][srow];
        const double vT = srow == 0 ? oBT[bq][ssl[m]][C - 1] : oBT[bq ^ 1][ssl[m]][srow - 1];
        const bool tv = n_lo + C - 1 + srow < N;
#if C2Q4_PAIRLINES
        if (k & 1) {                       // rows 8 k ..: the upper half of their lines
          hA[m] = vA; hY[m] = vY; hAok[m] = srow < cnt;
        } else {
          if (srow < cnt) bab8[m][n_lo - 1 + srow] = vA;
          if (hAok[m]) bab8[m][n_lo - 1 + C + srow] = hA[m];
          if (srow < cnt) byb8[m][n_lo - 1 + srow] = vY;
          if (hAok[m]) byb8[m][n_lo - 1 + C + srow] = hY[m];
          hAok[m] = false;
        }
        if ((k + 1) & 1) {                 // bt of rows 8 (k + 1) ..
          hT[m] = vT; hTok[m] = tv;
        } else {
          if (tv) btb8[m][n_lo + C - 1 + srow] = vT;
          if (hTok[m]) btb8[m][n_lo + 2 * C - 1 + srow] = hT[m];
          hTok[m] = false;
        }
#else
        if (srow < cnt) {
          bab8[m][n_lo - 1 + srow] = vA;
          byb8[m][n_lo - 1 + srow] = vY;
        }
        if (tv) btb8[m][n_lo + C - 1 + srow] = vT;
#endif
      }
    }
    bq ^= 1;
    lds_order();
  }
#pragma unroll
  for (int m = 0; m < NV; ++m) {   // the rows of bt the first segment left behind (1 .. C - 1; row 0 below) + their line's upper half
    if (ok8[m] && srow >= 1 && srow < N) btb8[m][srow] = oBT[bq ^ 1][ssl[m]][srow - 1];
#if C2Q4_PAIRLINES
    if (ok8[m] && hTok[m]) btb8[m][C + srow] = hT[m];
#endif
  }
  if (nseg == 0) {   // N == 1
    const double rd0 = 1.0 / carDZ.x, cz = carDZ.y;
    ban = 0.5 * rd0 * (cz * cz * rd0 - 1.0);
    bzn = -cz * rd0;
    if (alive) byb[0] = bzn;
  }
  if constexpr (TT) {   // row 0: bU_0 = 0; bV_0 and ba_0 are complete (carT = t_0)
    double u0[2], sn, cs;
    stm.usc(carT, u0, sn, cs);
    const double g = fma(-(bVn[0] * gtop[0]), sn, (bVn[1] * gtop[1]) * cs);
    acc[2] = fma(g, carT, acc[2]);
    carry += gsum<LG>(g * stm.D);
    sba += ban;
    const int JC = TQ.Jc, JR = J - 2 * JC, g0 = 2 * jl;
    const double nan = __builtin_nan("");
    if (stm.re) {
      GQ.bar[L.b * JR + g0] = alive ? sba + acc[0] : nan; GQ.bar[L.b * JR + g0 + 1] = alive ? sba - acc[1] : nan;
      GQ.bcr[L.b * JR + g0] = alive ? bcj[0] : nan; GQ.bcr[L.b * JR + g0 + 1] = alive ? bcj[1] : nan;
    } else {
      const int64_t o = L.b * JC + ((g0 - JR) >> 1);
      GQ.bac[o] = alive ? sba + acc[0] : nan; GQ.bbc[o] = alive ? acc[1] : nan;
      GQ.bdc[o] = alive ? acc[2] : nan; GQ.bcc[o] = alive ? bcj[0] + bcj[1] : nan;
    }
  }
  if (alive) {
    bab[0] = ban; btb[0] = carry;                            // row 0 (reverse.hpp:83-84)
    if constexpr (!TT) reinterpret_cast<double2 *>(bc + L.b * J)[jl] = make_double2(bcj[0], bcj[1]);
  }
  if constexpr (LN) {   // row 0 completes pair 0 (row 1 is in the tile of even pairs since the last step)
    otile[0][0][grp * 8 + jl] = make_double2(0.0, 0.0);
    otile[0][1][grp * 8 + jl] = make_double2(bVn[0] * gtop[0], bVn[1] * gtop[1]);
    lds_order();
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      if (ok8[m]) { bUl8[m][0] = otile[0][0][m * kWave + lane]; bVl8[m][0] = otile[0][1][m * kWave + lane]; }
    }
  } else if constexpr (TT) {
  } else if (alive) {
    bVb[0] = make_double2(bVn[0] * gtop[0], bVn[1] * gtop[1]);
    bUb[0] = make_double2(0.0, 0.0);
  }
}

struct Layout {
  int64_t nseg, nslot;
  size_t waves, ck, w, dz, words, total;   // doubles
};
inline Layout layout(int64_t B, int64_t N) {
  Layout l;
  l.nseg = (N - 1 + C - 1) / C;
  l.nslot = l.nseg / A + 1;
  l.waves = ((size_t)B * LG + kWave - 1) / kWave;
  l.ck = l.waves * (size_t)l.nslot * (2 * kCkD2 * kWave);
  l.w = l.waves * (size_t)N * (2 * kWave);
  l.dz = (size_t)B * N * 2;
  l.words = 2 * l.waves;
  l.total = l.ck + l.w + l.dz + l.words;
  return l;
}

}  // namespace q4

}  // namespace c2

using namespace c2;

// (c2_loglik.hip) one workgroup per wavefront of `spw` series: words[2 w] = c_max x the longest span between anchors four
// segments of C rows apart, words[2 w + 1] the same over single segments; +inf for unsorted / NaN times
extern "C" int c2_internal_anchor_spans(int64_t B, int64_t N, int64_t J, int C, int spw, const double *t, int64_t t_bs,
                                        const double *c, int64_t c_bs, unsigned long long *words, c2_stream_t stream);

extern "C" {

// records of the pair, in doubles (the caller adds the gate words in front and overlays the fallback's workspace)
size_t c2_internal_loglik_q4_record_doubles(int64_t B, int64_t N) { return q4::layout(B, N).total; }

// `guard`: kGateHeadWords + ceil(B / 64) device words (written here).  `rec`: c2_internal_loglik_q4_record_doubles(B, N) doubles.
int c2_internal_loglik_q4_grad(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                               const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                               double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, double *rec,
                               unsigned long long *guard, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const q4::Layout l = q4::layout(B, N);
  double2 *ck = reinterpret_cast<double2 *>(rec);
  double2 *W = reinterpret_cast<double2 *>(rec + l.ck);
  double2 *DZ = reinterpret_cast<double2 *>(rec + l.ck + l.w);
  unsigned long long *words = reinterpret_cast<unsigned long long *>(rec + l.ck + l.w + l.dz);
  unsigned long long *gate = guard + kGateHeadWords;
  const dim3 grid((unsigned)l.waves);
  static_assert(q4::A == 4, "k_anchor_spans measures spans of four segments");
  if (int e = c2_internal_anchor_spans(B, N, q4::J, q4::C, q4::SPW, t, t_bs, c, c_bs, words, stream)) return e;
  hipLaunchKernelGGL(q4::k_q4_gate, dim3(1), dim3(256), 0, s, (int64_t)l.waves, (const unsigned long long *)words, guard, gate);
  // rows of U, V, bU, bV as whole 128-byte lines (the LN instances): an even number of rows; C2_LOGLIK_Q4_LINES=0: row by row
  // (they pay from ~11000 series, where the wavefronts of a CU queue at its address unit; below, their LDS instructions only cost)
  const bool ln = N >= 2 && N % 2 == 0 &&
                  (opt::has(opt::k_loglik_q4_lines) ? opt::ival(opt::k_loglik_q4_lines) != 0 : B >= opt::ival(opt::k_q4_lines_min_batch));
#define C2_Q4_FWD_ARGS grid, dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, ck, l.nslot, W, DZ, (const unsigned long long *)gate
#define C2_Q4_REV_ARGS grid, dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs, U, (const double2 *)W, (const double2 *)DZ, (const double2 *)ck, \
                       l.nslot, l.nseg, (const int32_t *)flag, bt, bc, ba, bU, bV, by, (const unsigned long long *)gate
  if (ln) {
    hipLaunchKernelGGL((q4::k_q4_fwd<true, C2_Q4_LN_R>), C2_Q4_FWD_ARGS);
    hipLaunchKernelGGL(q4::k_q4_rev<true>, C2_Q4_REV_ARGS);
  } else {
    hipLaunchKernelGGL((q4::k_q4_fwd<false, 8>), C2_Q4_FWD_ARGS);
    hipLaunchKernelGGL(q4::k_q4_rev<false>, C2_Q4_REV_ARGS);
  }
#undef C2_Q4_FWD_ARGS
#undef C2_Q4_REV_ARGS
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// Coefficient-level forward only: the forward kernel without its records.  `guard`: kGateHeadWords + ceil(B / 64) words, written
// here (a word is 0 or +inf: the scaled frame of this kernel needs the span test as well, so `words` -- 2 per wavefront -- is
// scratch for k_anchor_spans).
int c2_internal_loglik_q4_tt(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *ac, const double *bc,
                             const double *dc, const double *c, const double *x, int64_t x_bs, const double *diag, const double *y,
                             double *ll, int32_t *flag, unsigned long long *words, unsigned long long *guard, c2_stream_t stream) {
  if (Jc < 0 || Jc > 4 || B < 1 || N < 1) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const q4::Layout l = q4::layout(B, N);
  unsigned long long *gate = guard + kGateHeadWords;
  const dim3 grid((unsigned)l.waves);
  const q4::TermsArgsQ T{ar, ac, bc, dc, coef_batched, (int)Jc};
  if (int e = c2_internal_anchor_spans(B, N, q4::J, q4::C, q4::SPW, x, x_bs, c, 8, words, stream)) return e;
  if (hipMemsetAsync(guard, 0, 8 * kGateHeadWords, s) != hipSuccess) return C2_ERR_HIP;
  hipLaunchKernelGGL(q4::k_q4_gate_tt, dim3((unsigned)((l.waves + 3) / 4)), dim3(kWave), 0, s, B, N, (int64_t)l.waves,
                     (const unsigned long long *)words, T, x, x_bs, guard, gate);
  hipLaunchKernelGGL((q4::k_q4_fwd<false, 16, true, false>), grid, dim3(kWave), 0, s, B, N, x, x_bs, c, (int64_t)8, diag,
                     (const double *)nullptr, (const double *)nullptr, y, ll, flag, (double2 *)nullptr, l.nslot, (double2 *)nullptr,
                     (double2 *)nullptr, (const unsigned long long *)gate, T);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
size_t c2_internal_loglik_q4_span_words(int64_t B, int64_t N) { return q4::layout(B, N).words; }

// Coefficient-level log-likelihood + gradient (J = Jr + 2 Jc = 8) on this pair, the rows formed in the lanes.  `c`: the rates
// (B, 8) (c2_terms.hip: k_rates); `rec`, `guard` as above -- a group of 64 series the pair declines (a span beyond the guard,
// unsorted times, a phase beyond the branch-free sincos) is left to the caller's composed chain, gated on the same words.
int c2_internal_loglik_q4_tt_grad(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *ac,
                                  const double *bc, const double *dc, const double *c, const double *x, int64_t x_bs,
                                  const double *diag, const double *y, double *ll, double *bar, double *bcr, double *bac,
                                  double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                                  int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream) {
  if (Jc < 0 || Jc > 4 || B < 1 || N < 1) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const q4::Layout l = q4::layout(B, N);
  double2 *ck = reinterpret_cast<double2 *>(rec);
  double2 *W = reinterpret_cast<double2 *>(rec + l.ck);
  double2 *DZ = reinterpret_cast<double2 *>(rec + l.ck + l.w);
  unsigned long long *words = reinterpret_cast<unsigned long long *>(rec + l.ck + l.w + l.dz);
  unsigned long long *gate = guard + kGateHeadWords;
  const dim3 grid((unsigned)l.waves);
  const q4::TermsArgsQ T{ar, ac, bc, dc, coef_batched, (int)Jc};
  const q4::TermsGradsQ G{bar, bcr, bac, bbc, bcc, bdc};
  if (int e = c2_internal_anchor_spans(B, N, q4::J, q4::C, q4::SPW, x, x_bs, c, 8, words, stream)) return e;
  if (hipMemsetAsync(guard, 0, 8 * kGateHeadWords, s) != hipSuccess) return C2_ERR_HIP;
  hipLaunchKernelGGL(q4::k_q4_gate_tt, dim3((unsigned)((l.waves + 3) / 4)), dim3(kWave), 0, s, B, N, (int64_t)l.waves,
                     (const unsigned long long *)words, T, x, x_bs, guard, gate);
  hipLaunchKernelGGL((q4::k_q4_fwd<false, 16, true>), grid, dim3(kWave), 0, s, B, N, x, x_bs, c, (int64_t)8, diag,
                     (const double *)nullptr, (const double *)nullptr, y, ll, flag, ck, l.nslot, W, DZ,
                     (const unsigned long long *)gate, T);
  hipLaunchKernelGGL((q4::k_q4_rev<false, true>), grid, dim3(kWave), 0, s, B, N, x, x_bs, c, (int64_t)8, (const double *)nullptr,
                     (const double2 *)W, (const double2 *)DZ, (const double2 *)ck, l.nslot, l.nseg, (const int32_t *)flag, bx,
                     (double *)nullptr, bdiag, (double *)nullptr, (double *)nullptr, by, (const unsigned long long *)gate, T, G);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

}  // extern "C"
