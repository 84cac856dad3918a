// c2_sweep_rev.hip -- reverse-mode passes of solve_lower / solve_upper / matmul_lower / matmul_upper (reference
// internal.hpp:191-303 forward_rev / backward_rev; reverse.hpp:87-217) for SEVERAL right-hand sides, lanes <-> right-hand
// sides: the mapping of k_sweepK (c2_sweep.hip) applied to the backward sweep.
//
// A series is walked by KL lanes (KL = 8 or 16 >= nrhs), lane k owning column k of the workspace row F_n (J x nrhs,
// F[n, j + J k]: the J entries of a column are contiguous, so a lane reads ITS column with 16-byte loads and the lanes
// of a series cover the 8 J nrhs bytes of a row as one dense run) and of the adjoint state bF.  Per step
//     bB_n  = s sum_k bZ_n[k] (p o F_n)[:, k]        bF += s B_n^T bZ_n          (s = -1 solve, +1 matmul)
//     bp    = p o sum_k F_n[:, k] bF[:, k]           bc += dt bp ;  phi = sum_j c_j bp_j ;  bt_n -= phi, bt_m += phi
//     bF    = p o bF ;   bA_m = sum_k X_m[k] bF[:, k] ;   bX_m[k] += A_m . bF[:, k]      (X = Z for solves, Y for products)
// the three sums over k are the only cross-lane traffic: 8 (or 16) partial values per lane each, combined by a
// reduce-scatter (c2_rscatter.hpp) that leaves element j of the result in lane j -- where it is stored from, as one dense
// run per row.  B = U, A = V for the lower sweeps and the other way round for the upper ones; row m = n -/+ 1.
// The first-round kernel (k_sweep_rev, lanes over J, c2_ops.hip) stays for shapes this mapping does not cover.
#include <hip/hip_runtime.h>

#include <cstdint>

#include <type_traits>

#include "c2_dispatch.hpp"
#include "c2_loglik_helpers.hpp"
#include "c2_rscatter.hpp"
#include "../../include/celerite2_amd.h"

namespace c2r {
using namespace c2;

// Section timing of the multi-rhs reverse step of k_sweepK_rev (diagnostic builds only: tools/build_variant.sh prof
// c2_sweep_rev.hip -DC2R_PROF).
#ifdef C2R_PROF
__device__ unsigned long long c2r_prof[8];
#define C2R_TICK(k)                                                    \
  do {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long now_ = __builtin_readcyclecounter();      \
    prof_[k] += now_ - tick_;                                          \
    tick_ = now_;                                                      \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)
#else
#define C2R_TICK(k) do {} while (0)
#endif

template <int KL, int JM, bool LOWER, bool SOLVE>
__global__ __launch_bounds__(kWave) void k_sweepK_rev(int64_t B, int64_t N, int J, int64_t nrhs,
                                                      const double *__restrict__ t, int64_t t_bs,
                                                      const double *__restrict__ c, int64_t c_bs,
                                                      const double *__restrict__ U, const double *__restrict__ V,
                                                      const double *__restrict__ Y, const double *__restrict__ Z,
                                                      const double *__restrict__ F, const double *__restrict__ bZ,
                                                      double *__restrict__ bt, double *__restrict__ bc,
                                                      double *__restrict__ bU, double *__restrict__ bV,
                                                      double *__restrict__ bY) {
  static_assert((KL == 8 || KL == 16) && (JM == 8 || JM == 16) && JM <= KL, "shapes of the reduce-scatter");
  constexpr int SPW = kWave / KL, NH = JM / 8;
  // rows requested ahead (a ring row is JM + 6 register pairs; nrhs = 16 at J = 8: 16.2 -> 13.9 ms from two to four)
  constexpr int R = JM == 8 ? 4 : 2;
  __shared__ __attribute__((aligned(16))) double rowbuf[2][SPW][3][KL];  // p_n, B_n, A_m of two consecutive steps
  const int lane = threadIdx.x, sl = lane / KL, k = lane % KL;
  int64_t b = (int64_t)blockIdx.x * SPW + sl;
  const bool vb = b < B;
  if (!vb) b = B - 1;
  const bool vk = k < nrhs;
  const int kc = vk ? k : (int)nrhs - 1;
  const bool actj = k < J;
  const int jk = actj ? k : 0;
  // element of the width-J results this lane ends up with after a reduce-scatter: k (KL = 8) or k >> 1 (KL = 16, both
  // lanes of a pair hold it; the even one owns it), plus 8 h for the second half of a width-16 vector
  const int eidx = (KL == 8) ? k : (k >> 1);
  const bool eown = (KL == 8) ? true : ((k & 1) == 0);
  const double *tb = t + b * t_bs;
  const double *Bb = (LOWER ? U : V) + b * N * J, *Ab = (LOWER ? V : U) + b * N * J;
  double *bBb = (LOWER ? bU : bV) + b * N * J, *bAb = (LOWER ? bV : bU) + b * N * J;
  const double *Xb = (SOLVE ? Z : Y) + b * N * nrhs + kc;
  const double *bZb = bZ + b * N * nrhs + kc;
  double *bYb = bY + b * N * nrhs + kc;
  const double *Fb = F + b * N * (int64_t)J * nrhs + (int64_t)J * kc;
  double *btb = bt + b * N;
  const double cj = actj ? c[b * c_bs + k] : 0.0;
  double ce[NH], bce[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) { ce[h] = (8 * h + eidx < J) ? c[b * c_bs + 8 * h + eidx] : 0.0; bce[h] = 0.0; }
  const double sgn = SOLVE ? -1.0 : 1.0;
  auto rowof = [&](int64_t s) { return LOWER ? s : N - 1 - s; };      // row n of step s
  auto rowm = [&](int64_t s) { return LOWER ? s - 1 : N - s; };       // row m = n -/+ 1 of step s

  double bF[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) bF[j] = 0.0;
  // cotangent bZ of the row about to be processed: solves carry the updated value, products the given one (fetched one
  // step ahead either way)
  const int64_t nlast = rowof(N - 1);
  double bzrun = vk ? bZb[nlast * nrhs] : 0.0;
  if (vb && vk) bYb[nlast * nrhs] = SOLVE ? bzrun : 0.0;  // the far end receives nothing from the sweep
  double carry = 0.0;                                    // phi of the previous (later) step

  // prefetch ring: the rows of step s are requested R steps ahead
  double rF[R][JM], rbz[R], rx[R], rbn[R], ram[R], rtn[R], rtm[R];
  auto load_step = [&](int r, int64_t s) {
    s = s >= 1 ? s : 1;
    const int64_t n = rowof(s), m = rowm(s);
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      if (j + 1 < J) {
        const double2 v = *reinterpret_cast<const double2 *>(Fb + n * (int64_t)J * nrhs + j);
        rF[r][j] = v.x; rF[r][j + 1] = v.y;
      } else {
        rF[r][j] = (j < J) ? Fb[n * (int64_t)J * nrhs + j] : 0.0;
        rF[r][j + 1] = 0.0;
      }
    }
    rbz[r] = bZb[m * nrhs];   // incoming cotangent of row m (the row of the NEXT step)
    rx[r] = Xb[m * nrhs];
    rbn[r] = actj ? Bb[n * J + jk] : 0.0;
    ram[r] = actj ? Ab[m * J + jk] : 0.0;
    rtn[r] = tb[n]; rtm[r] = tb[m];
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_step(r, N - 1 - r);

  int q = 0;
#ifdef C2R_PROF
  unsigned long long prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tick_ = __builtin_readcyclecounter();
#endif
  for (int64_t s0 = N - 1; s0 >= 1; s0 -= R) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t s = s0 - r;
      if (s >= 1) {
        const int64_t n = rowof(s), m = rowm(s);
        double Fn[JM];
#pragma unroll
        for (int j = 0; j < JM; ++j) Fn[j] = vk ? rF[r][j] : 0.0;
        const double bzin_m = vk ? rbz[r] : 0.0, xm = vk ? rx[r] : 0.0, bn = rbn[r], am = ram[r];
        const double dt = LOWER ? rtm[r] - rtn[r] : rtn[r] - rtm[r];  // internal.hpp:227 / 284
        C2R_TICK(0);
        load_step(r, s - R);
        C2R_TICK(5);
        const double p = exp_decay(cj * dt);
        rowbuf[q][sl][0][k] = p; rowbuf[q][sl][1][k] = bn; rowbuf[q][sl][2][k] = am;
        lds_order();
        C2R_TICK(1);
        const double bzn = bzrun;
        double pbB[JM], pbp[JM], pbA[JM], acc = 0.0;
#pragma unroll
        for (int j = 0; j < JM; j += 2) {
          const double2 p2 = *reinterpret_cast<const double2 *>(&rowbuf[q][sl][0][j]);
          const double2 b2 = *reinterpret_cast<const double2 *>(&rowbuf[q][sl][1][j]);
          const double2 a2 = *reinterpret_cast<const double2 *>(&rowbuf[q][sl][2][j]);
          const double pv[2] = {p2.x, p2.y}, bv[2] = {b2.x, b2.y}, av[2] = {a2.x, a2.y};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int jj = j + u;
            pbB[jj] = bzn * (pv[u] * Fn[jj]);               // internal.hpp:232 / 289
            bF[jj] = fma(sgn * bv[u], bzn, bF[jj]);          // internal.hpp:233 / 290
            pbp[jj] = Fn[jj] * bF[jj];                       // internal.hpp:236 / 293
            bF[jj] *= pv[u];                                 // internal.hpp:241 / 298
            pbA[jj] = xm * bF[jj];                           // update_f::reverse (internal.hpp:59 / 80)
            acc = fma(av[u], bF[jj], acc);                   // ... and the cotangent of row m (internal.hpp:60 / 81)
          }
        }
        q ^= 1;
        C2R_TICK(2);
        // cotangent of row m: solves fold it into the running bZ, products write it out
        if (SOLVE) {
          bzrun = bzin_m + acc;
          if (vb && vk) bYb[m * nrhs] = bzrun;
        } else {
          bzrun = bzin_m;  // row m is the row of the next step
          if (vb && vk) bYb[m * nrhs] = acc;
        }
        C2R_TICK(3);
        // the three sums over the right-hand sides
        double phi = 0.0;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          double v8[8];
          int ko;
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = pbB[8 * h + i];
          const double rB = sgn * rscatter8<KL>(v8, k, ko);
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = pbp[8 * h + i];
          const double rp = rscatter8<KL>(v8, k, ko);
#pragma unroll
          for (int i = 0; i < 8; ++i) v8[i] = pbA[8 * h + i];
          const double rA = rscatter8<KL>(v8, k, ko);
          const int e = 8 * h + eidx;
          // this lane's p_e: element e of the decay vector of THIS step (still in the other half of the double buffer)
          const double pe = rowbuf[q ^ 1][sl][0][e < KL ? e : 0];
          const double bpe = rp * pe;
          if (vb && eown && e < J) {
            bBb[n * J + e] = rB;
            bAb[m * J + e] = rA;
          }
          bce[h] = fma(dt, (e < J) ? bpe : 0.0, bce[h]);
          phi = fma(ce[h], (eown && e < J) ? bpe : 0.0, phi);
        }
        phi = gsum<KL>(phi);
        // LOWER: bt[n] -= phi, bt[m] += phi -> row n is complete now (it got +phi of the previous step)
        // UPPER: bt[m] -= phi, bt[n] += phi -> row n is complete now (it got -phi of the previous step)
        if (vb && k == 0) btb[n] = LOWER ? carry - phi : phi - carry;
        carry = phi;
        C2R_TICK(4);
      }
    }
  }
#ifdef C2R_PROF
  if (lane == 0 && blockIdx.x % 97 == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&c2r_prof[i], prof_[i]);
#endif
  // the near end: row of step 0 gets no bB, the far end no bA; bt of the near end is what the last step left
  if (vb) {
    const int64_t n0 = rowof(0);
    if (k == 0) btb[n0] = LOWER ? carry : -carry;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int e = 8 * h + eidx;
      if (eown && e < J) {
        bBb[n0 * J + e] = 0.0;
        bAb[nlast * J + e] = 0.0;
        bc[b * J + e] = bce[h];
      }
    }
  }
}


// ---- nrhs = J = 8 on full wavefronts: every width-8 row moves as half of an aligned 128-byte LINE ---------------------------
// k_sweepK_rev above asks for every row of U, W, X, bZ on its own and stores every row of bU, bW, bY on its own: 64-byte
// requests, fifteen memory instructions per step.  With one wavefront per SIMD the step is then bound by how fast the CU
// accepts REQUESTS, not by bytes or arithmetic (section timers of a -DC2R_PROF build: of 4700 cycles per step 2000 go into issuing
// the nine loads and 1900 into the sections that hold the six stores; taking two 64-byte loads out of the step saves 16 % of
// it, two stores 18 %, three of the four 128-byte workspace requests only 7 %).  Here the rows of a series pair up into the
// aligned lines (2 l, 2 l + 1) they share in memory: one 16-byte piece per lane, one request per line and array every two
// steps, staged through LDS rings of four rows (inputs, requested two lines ahead) and tiles of two rows (outputs); the
// workspace row (512 bytes a series) is requested as a dense run four steps ahead and turned into columns through LDS.
// Per two steps: 8 + 5 loads and 4 stores instead of 18 and 12.  B a multiple of 8, N >= 8; the first / last steps, where a
// line may reach beyond the series, take the same step with guarded, synchronous moves.
constexpr int kLS = 34;   // LDS stride (doubles) of a series in a ring of four 8-double rows: 272 B, conflict-free b128
constexpr int kOS = 18;   // ... in a tile of two rows: 144 B

template <bool LOWER, bool SOLVE>
__global__ __launch_bounds__(kWave) void k_sweep8_rev_lines(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                                            const double *__restrict__ c, int64_t c_bs,
                                                            const double *__restrict__ U, const double *__restrict__ V,
                                                            const double *__restrict__ Y, const double *__restrict__ Z,
                                                            const double *__restrict__ F, const double *__restrict__ bZ,
                                                            double *__restrict__ bt, double *__restrict__ bc,
                                                            double *__restrict__ bU, double *__restrict__ bV,
                                                            double *__restrict__ bY) {
  constexpr int J = 8, SPW = 8, RF = 4;
  __shared__ __attribute__((aligned(16))) double Bq[SPW * kLS], Aq[SPW * kLS], Xq[SPW * kLS], Zq[SPW * kLS];  // [sl][row & 3][8]
  __shared__ __attribute__((aligned(16))) double tq[SPW][4];
  __shared__ __attribute__((aligned(16))) double pq[SPW][J];
  __shared__ __attribute__((aligned(16))) double ftile[SPW * J * J];
  __shared__ __attribute__((aligned(16))) double oB[SPW * kOS], oA[SPW * kOS], oY[SPW * kOS];   // [sl][row & 1][8]
  __shared__ __attribute__((aligned(16))) double oT[SPW][2];
  const int lane = threadIdx.x, sl = lane >> 3, k = lane & 7;
  const int64_t b = (int64_t)blockIdx.x * SPW + sl;
  const int hrow = k >> 2, col = 2 * (k & 3);   // this lane's 16-byte piece of a line: row 2 l + hrow, columns col, col + 1
  constexpr int dir = LOWER ? -1 : 1;           // the sweep visits rows n, n + dir, ...; m = n + dir
  constexpr int kFirstParity = LOWER ? 1 : 0;   // parity of the row of a line the sweep meets first
  const double *tb = t + b * t_bs;
  const double *Bb = (LOWER ? U : V) + b * N * J, *Ab = (LOWER ? V : U) + b * N * J;
  double *bBb = (LOWER ? bU : bV) + b * N * J, *bAb = (LOWER ? bV : bU) + b * N * J;
  const double *Xb = (SOLVE ? Z : Y) + b * N * J, *Zb = bZ + b * N * J;
  double *bYb = bY + b * N * J, *btb = bt + b * N;
  const double *Fd = F + b * N * (int64_t)(J * J) + 2 * k;
  const double cj = c[b * c_bs + k];
  double bce = 0.0;
  constexpr double sgn = SOLVE ? -1.0 : 1.0;
  const int64_t lmax = (N - 1) >> 1;

  struct Line { double2 b, a, x, z; double tt; };
  // (values in, values out: handed around by reference the rings end up in scratch)
  auto req_line = [&](int64_t l) -> Line {
    Line R;
    l = l < 0 ? 0 : (l > lmax ? lmax : l);
    int64_t row = 2 * l + hrow; row = row < N ? row : N - 1;
    int64_t trow = 2 * l + (k & 1); trow = trow < N ? trow : N - 1;
    R.b = *reinterpret_cast<const double2 *>(Bb + row * J + col);
    R.a = *reinterpret_cast<const double2 *>(Ab + row * J + col);
    R.x = *reinterpret_cast<const double2 *>(Xb + row * J + col);
    R.z = *reinterpret_cast<const double2 *>(Zb + row * J + col);
    R.tt = tb[trow];
    return R;
  };
  auto put_line = [&](int64_t l, const Line R) {
    const int o = sl * kLS + ((2 * (int)(l & 1) + hrow) * J) + col;   // slot (row & 3) of the ring
    *reinterpret_cast<double2 *>(&Bq[o]) = R.b;
    *reinterpret_cast<double2 *>(&Aq[o]) = R.a;
    *reinterpret_cast<double2 *>(&Xq[o]) = R.x;
    *reinterpret_cast<double2 *>(&Zq[o]) = R.z;
    tq[sl][2 * (int)(l & 1) + (k & 1)] = R.tt;   // (lanes of equal k & 1 write the same value)
  };
  struct FRow { double2 q0, q1, q2, q3; };   // (named members: as an array the ring ends up in scratch)
  auto req_F = [&](int64_t n) -> FRow {
    FRow f;
    n = n < 0 ? 0 : (n < N ? n : N - 1);
    const double *a = Fd + n * (int64_t)(J * J);
    f.q0 = *reinterpret_cast<const double2 *>(a);
    f.q1 = *reinterpret_cast<const double2 *>(a + 16);
    f.q2 = *reinterpret_cast<const double2 *>(a + 32);
    f.q3 = *reinterpret_cast<const double2 *>(a + 48);
    return f;
  };
  // a finished line of an output tile: 16 bytes per lane, 128 per series.  GUARD: rows beyond either end stay unwritten
  auto flush = [&](const double *tile, double *base, int64_t l, auto guard_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value;
    const int64_t row = 2 * l + hrow;
    const double2 v = *reinterpret_cast<const double2 *>(&tile[sl * kOS + hrow * J + col]);
    if (!GUARD || (row >= 0 && row < N)) *reinterpret_cast<double2 *>(base + row * J + col) = v;
  };
  auto flush_t = [&](int64_t l, auto guard_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value;
    const double2 v = *reinterpret_cast<const double2 *>(&oT[sl][0]);
    // (two 8-byte stores by lanes 0 and 1 of the series: with an odd number of rows every other series starts 8 bytes off
    // a 16-byte boundary)
    if (k < 2 && (!GUARD || (2 * l + k >= 0 && 2 * l + k < N))) btb[2 * l + k] = k ? v.y : v.x;
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;

  double bF[J];
#pragma unroll
  for (int j = 0; j < J; ++j) bF[j] = 0.0;
  const int64_t nfar = LOWER ? N - 1 : 0, nnear = LOWER ? 0 : N - 1;
  double bzrun = Zb[nfar * J + k];
  double carry = 0.0;
  // the far end receives nothing from the sweep: bA = 0, bY = its own cotangent (solves) or 0 (products)
  oA[sl * kOS + (int)(nfar & 1) * J + k] = 0.0;
  oY[sl * kOS + (int)(nfar & 1) * J + k] = SOLVE ? bzrun : 0.0;
  lds_order();
  if ((nfar & 1) != kFirstParity) {   // ... and no step completes its line
    flush(oA, bAb, nfar >> 1, Yes{});
    flush(oY, bYb, nfar >> 1, Yes{});
  }

  // One step (row n, m = n + dir).  EDGE: lines and workspace row moved here, synchronously, with guards; otherwise the
  // rings are kept by the caller and PH says which lines the step completes (0: n is the first row of its line -- m the
  // second of the same line: bA, bY leave; 1: n the second: bB, bt leave).
  auto step = [&](const int64_t n, auto edge_tag, auto ph_tag, FRow fr) __attribute__((always_inline)) -> FRow {
    constexpr bool EDGE = decltype(edge_tag)::value;
    constexpr int PH = decltype(ph_tag)::value;
    const int64_t m = n + dir;
    if constexpr (EDGE) {
      put_line(n >> 1, req_line(n >> 1));
      put_line(m >> 1, req_line(m >> 1));   // (the same line again where m shares it)
      fr = req_F(n);
    }
    {
      double *w = &ftile[sl * J * J + 2 * k];
      *reinterpret_cast<double2 *>(w) = fr.q0;
      *reinterpret_cast<double2 *>(w + 16) = fr.q1;
      *reinterpret_cast<double2 *>(w + 32) = fr.q2;
      *reinterpret_cast<double2 *>(w + 48) = fr.q3;
    }
    FRow fnext = fr;
    if constexpr (!EDGE) fnext = req_F(n + RF * dir);   // (the slot is free once its row is on its way to LDS)
    lds_order();
    double Fn[J];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 v = *reinterpret_cast<const double2 *>(&ftile[(sl * J + k) * J + 2 * q]);
      Fn[2 * q] = v.x; Fn[2 * q + 1] = v.y;
    }
    const double tn = tq[sl][n & 3], tm = tq[sl][m & 3];
    const double dt = LOWER ? tm - tn : tn - tm;   // internal.hpp:227 / 284
    const double p = exp_decay(cj * dt);
    pq[sl][k] = p;
    const double xm = Xq[sl * kLS + (int)(m & 3) * J + k];
    const double bzring = Zq[sl * kLS + (int)((SOLVE ? m : n) & 3) * J + k];
    lds_order();
    const double bzn = SOLVE ? bzrun : bzring;
    double pbB[J], pbp[J], pbA[J], acc = 0.0;
#pragma unroll
    for (int j = 0; j < J; j += 2) {
      const double2 p2 = *reinterpret_cast<const double2 *>(&pq[sl][j]);
      const double2 b2 = *reinterpret_cast<const double2 *>(&Bq[sl * kLS + (int)(n & 3) * J + j]);
      const double2 a2 = *reinterpret_cast<const double2 *>(&Aq[sl * kLS + (int)(m & 3) * J + j]);
      const double pv[2] = {p2.x, p2.y}, bv[2] = {b2.x, b2.y}, av[2] = {a2.x, a2.y};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int jj = j + u;
        pbB[jj] = bzn * (pv[u] * Fn[jj]);               // internal.hpp:232 / 289
        bF[jj] = fma(sgn * bv[u], bzn, bF[jj]);          // internal.hpp:233 / 290
        pbp[jj] = Fn[jj] * bF[jj];                       // internal.hpp:236 / 293
        bF[jj] *= pv[u];                                 // internal.hpp:241 / 298
        pbA[jj] = xm * bF[jj];                           // update_f::reverse (internal.hpp:59 / 80)
        acc = fma(av[u], bF[jj], acc);                   // ... and the cotangent of row m (internal.hpp:60 / 81)
      }
    }
    // cotangent of row m: solves fold it into the running bZ, products write it out
    if (SOLVE) bzrun = bzring + acc;
    oY[sl * kOS + (int)(m & 1) * J + k] = SOLVE ? bzrun : acc;
    // the three sums over the right-hand sides: element k of each ends in lane k
    int ko;
    const double rB = sgn * rscatter8<8>(pbB, k, ko);
    const double rp = rscatter8<8>(pbp, k, ko);
    const double rA = rscatter8<8>(pbA, k, ko);
    const double bpe = rp * p;
    oB[sl * kOS + (int)(n & 1) * J + k] = rB;
    oA[sl * kOS + (int)(m & 1) * J + k] = rA;
    bce = fma(dt, bpe, bce);
    const double phi = gsum<8>(cj * bpe);
    // LOWER: bt[n] -= phi, bt[m] += phi -> row n is complete now (it got +phi of the previous step); UPPER: mirrored
    oT[sl][n & 1] = LOWER ? carry - phi : phi - carry;
    carry = phi;
    lds_order();
    if constexpr (EDGE) {
      if ((n & 1) != kFirstParity) { flush(oB, bBb, n >> 1, Yes{}); flush_t(n >> 1, Yes{}); }
      if ((m & 1) != kFirstParity) { flush(oA, bAb, m >> 1, Yes{}); flush(oY, bYb, m >> 1, Yes{}); }
    } else if constexpr (PH == 0) {
      flush(oA, bAb, m >> 1, No{}); flush(oY, bYb, m >> 1, No{});
    } else {
      flush(oB, bBb, n >> 1, No{}); flush_t(n >> 1, No{});
    }
    return fnext;
  };

  int64_t n = nfar, left = N - 1;   // steps left; the step at row n has m = n + dir
  FRow fr0, fr1, fr2, fr3;
  fr0 = fr1 = fr2 = fr3 = FRow{};
  if ((n & 1) != kFirstParity && left > 0) { step(n, Yes{}, std::integral_constant<int, 0>{}, fr0); n += dir; --left; }
  if (left >= 4) {
    // rings: line of n in LDS, the two lines behind it and the workspace rows of the next four steps requested
    Line lr0, lr1;
    put_line(n >> 1, req_line(n >> 1));
    lr0 = req_line((n >> 1) + dir);
    lr1 = req_line((n >> 1) + 2 * dir);
    static_assert(RF == 4, "four steps per iteration");
    fr0 = req_F(n); fr1 = req_F(n + dir); fr2 = req_F(n + 2 * dir); fr3 = req_F(n + 3 * dir);
    for (; left >= 4; left -= 4, n += 4 * dir) {
      const int64_t lc = n >> 1;
      put_line(lc + dir, lr0);
      lr0 = req_line(lc + 3 * dir);
      fr0 = step(n, No{}, std::integral_constant<int, 0>{}, fr0);
      fr1 = step(n + dir, No{}, std::integral_constant<int, 1>{}, fr1);
      put_line(lc + 2 * dir, lr1);
      lr1 = req_line(lc + 4 * dir);
      fr2 = step(n + 2 * dir, No{}, std::integral_constant<int, 0>{}, fr2);
      fr3 = step(n + 3 * dir, No{}, std::integral_constant<int, 1>{}, fr3);
    }
  }
  for (; left > 0; --left, n += dir) step(n, Yes{}, std::integral_constant<int, 0>{}, fr0);

  // the near end: no bB, bt = what the last step left; its lines leave now if no step completed them
  oB[sl * kOS + (int)(nnear & 1) * J + k] = 0.0;
  oT[sl][nnear & 1] = LOWER ? carry : -carry;
  lds_order();
  flush(oB, bBb, nnear >> 1, Yes{});
  flush_t(nnear >> 1, Yes{});
  flush(oA, bAb, nnear >> 1, Yes{});
  flush(oY, bYb, nnear >> 1, Yes{});
  bc[b * J + k] = bce;
}

}  // namespace c2r

using namespace c2r;

#ifdef C2R_PROF
// cycles per section summed over the sampled wavefronts (every 97th) since the last call; resets the counters
extern "C" void c2_internal_sweep_rev_prof_read(unsigned long long *out8) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(c2r::c2r_prof), sizeof(unsigned long long) * 8);
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(c2r::c2r_prof), z, sizeof(z));
}
#endif

extern "C" int c2_internal_sweepK_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                      double *bc, double *bU, double *bV, double *bY, c2_stream_t stream) {
  // five or more right-hand sides: with fewer, half the lanes of a series idle and the first-round kernel (lanes over
  // J) is as fast (nrhs = 3: 8.6 vs 9.0 ms at B = 8192, N = 4096, J = 8; nrhs = 8: 21.4 vs 8.9 ms)
  if (nrhs < 5 || nrhs > 16 || J > 16 || N < 2) return C2_ERR_UNSUPPORTED;
  if ((J & 1) || ((uintptr_t)F) % 16 != 0) return C2_ERR_UNSUPPORTED;  // 16-byte loads of the workspace columns
  const int JM = J <= 8 ? 8 : 16;
  const int KL = (nrhs <= 8 && JM == 8) ? 8 : 16;
  hipStream_t s = (hipStream_t)stream;
  // nrhs = J = 8: the line-pairing kernel on the whole wavefronts of the batch (every pointer it moves 16-byte pieces of
  // must allow that), the row-by-row kernel below on the B % 8 series left over
  if (J == 8 && nrhs == 8 && B >= 8 && N >= 8 && !(c2::opt::has(c2::opt::k_sweep_rev_lines) && c2::opt::ival(c2::opt::k_sweep_rev_lines) == 0) &&
      (((uintptr_t)U | (uintptr_t)V | (uintptr_t)Y | (uintptr_t)Z | (uintptr_t)F | (uintptr_t)bZ | (uintptr_t)bU | (uintptr_t)bV |
        (uintptr_t)bY) % 16) == 0) {
    const int64_t B8 = B / 8 * 8;
    const dim3 g8((unsigned)(B8 / 8));
#define C2_SL(LO, SO)                                                                                                  \
  hipLaunchKernelGGL((k_sweep8_rev_lines<LO, SO>), g8, dim3(kWave), 0, s, B8, N, t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, \
                     bc, bU, bV, bY)
    if (lower) { if (solve) C2_SL(true, true); else C2_SL(true, false); }
    else { if (solve) C2_SL(false, true); else C2_SL(false, false); }
#undef C2_SL
    if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;
    if (B8 == B) return C2_OK;
    const int64_t o = B8 * N;   // rows of the series already done
    t += B8 * t_bs; c += B8 * c_bs;
    U += o * J; V += o * J; Y += o * nrhs; Z += o * nrhs; F += o * J * nrhs; bZ += o * nrhs;
    bt += o; bc += B8 * J; bU += o * J; bV += o * J; bY += o * nrhs;
    B -= B8;
  }
  const dim3 grid((unsigned)((B + (kWave / KL) - 1) / (kWave / KL)));
#define C2_SKR1(KL_, JM_, LO, SO)                                                                                      \
  hipLaunchKernelGGL((k_sweepK_rev<KL_, JM_, LO, SO>), grid, dim3(kWave), 0, s, B, N, (int)J, nrhs, t, t_bs, c, c_bs, U, V, \
                     Y, Z, F, bZ, bt, bc, bU, bV, bY)
#define C2_SKR(KL_, JM_)                                            \
  do {                                                              \
    if (lower) {                                                    \
      if (solve) C2_SKR1(KL_, JM_, true, true);                     \
      else C2_SKR1(KL_, JM_, true, false);                          \
    } else {                                                        \
      if (solve) C2_SKR1(KL_, JM_, false, true);                    \
      else C2_SKR1(KL_, JM_, false, false);                         \
    }                                                               \
  } while (0)
  if (KL == 8) C2_SKR(8, 8);
  else if (JM == 8) C2_SKR(16, 8);
  else C2_SKR(16, 16);
#undef C2_SKR
#undef C2_SKR1
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// ---- the four reverse sweeps on a small batch of LONG series --------------------------------------------------------------
// Row by row a series of 1e5 rows takes 16-18 ms per right-hand side.  The adjoint state of the forward sweep IS the state
// of the opposite sweep applied to bZ (the reverse of solve_lower is a solve_upper, internal.hpp:191-303 read against
// 148-189): with H_n the adjoint of the state entering step n BEFORE its decay,
//     bY = sweep'(bZ)  and  H_n = s F'_m          (F': workspace of that sweep, m = n -/+ 1, s = -1 solve / +1 product)
// and everything else is local to a row:
//     bB_n = s p o sum_k X_n[k] F_n[:, k]         (X = bY for solves, bZ for products)
//     bA_m = p o sum_k Q_m[k] H_n[:, k]           (Q = Z for solves, Y for products)
//     bp   = p o sum_k F_n[:, k] H_n[:, k] ;  bc += dt bp ;  phi = sum_j c_j bp_j ;  bt_n -/+= phi, bt_m +/-= phi.
// So: one opposite sweep WITH its workspace -- parallel along time for these shapes (chunk maps of c2_timepar_grad.hip for
// the solves, c2_scan.hip for the products) -- and one pass with a thread per row (k_rev_rows) plus a reduction for bc.
namespace c2rl {
using namespace c2;
constexpr int kRowsPerBlock = 256;

// thread <-> (row, j): JL = 1 .. 32 lanes per row (the power of two >= J), so that the lanes of a row read the J
// contiguous entries of a workspace column together; kRowsPerBlock / JL rows per block.
template <int JL, bool LOWER, bool SOLVE>
__global__ __launch_bounds__(kRowsPerBlock) void k_rev_rows(int64_t N, int J, int64_t nrhs, const double *__restrict__ t,
                                                            int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                            const double *__restrict__ X, const double *__restrict__ Q,
                                                            const double *__restrict__ F, const double *__restrict__ Fa,
                                                            double *__restrict__ bt, double *__restrict__ outA,
                                                            double *__restrict__ outB, double *__restrict__ part) {
  constexpr int RB = kRowsPerBlock / JL;   // rows per block
  __shared__ double red[kRowsPerBlock / kWave][JL];
  const int j = threadIdx.x % JL, rl = threadIdx.x / JL;
  const int64_t b = blockIdx.y, r = (int64_t)blockIdx.x * RB + rl;
  const bool row = r < N, act = row && j < J;
  const bool va = row && (LOWER ? r >= 1 : r + 1 < N);        // the step AT this row
  const bool vb = row && (LOWER ? r + 1 < N : r >= 1);        // the step whose partner row this is
  const int64_t ma = LOWER ? r - 1 : r + 1, nb = LOWER ? r + 1 : r - 1;
  const double *tb = t + b * t_bs;
  const double s = SOLVE ? -1.0 : 1.0;
  const double dta = va ? -fabs(tb[r] - tb[ma]) : 0.0, dtb = vb ? -fabs(tb[nb] - tb[r]) : 0.0;
  const int64_t JK = (int64_t)J * nrhs;
  const int jc = j < J ? j : 0;
  const double *Fr = F + (b * N + (row ? r : 0)) * JK + jc, *Fn = F + (b * N + (vb ? nb : 0)) * JK + jc;
  const double *Hr = Fa + (b * N + (va ? ma : 0)) * JK + jc;      // s H of the step at this row
  const double *Hb = Fa + (b * N + (row ? r : 0)) * JK + jc;      // s H of the step whose partner this row is
  const double *Xr = X + (b * N + (row ? r : 0)) * nrhs, *Qr = Q + (b * N + (row ? r : 0)) * nrhs;
  const double cj = c[b * c_bs + jc];
  double oa = 0.0, ob = 0.0, bpa = 0.0, bpb = 0.0;
  if (va && act) {
    const double p = exp_decay(cj * dta);
    double acc = 0.0, bp = 0.0;
    for (int64_t k = 0; k < nrhs; ++k) {
      const double f = Fr[J * k];
      acc = fma(Xr[k], f, acc);
      bp = fma(f, Hr[J * k], bp);
    }
    oa = s * p * acc;
    bpa = s * p * bp;
  }
  if (vb && act) {
    const double p = exp_decay(cj * dtb);
    double acc = 0.0, bp = 0.0;
    for (int64_t k = 0; k < nrhs; ++k) {
      const double h = Hb[J * k];
      acc = fma(Qr[k], h, acc);
      bp = fma(Fn[J * k], h, bp);
    }
    ob = s * p * acc;
    bpb = s * p * bp;
  }
  if (act) {
    outA[(b * N + r) * J + j] = oa;
    outB[(b * N + r) * J + j] = ob;
  }
  // phi of the two steps: sums over the lanes of the row
  double fa = act ? cj * bpa : 0.0, fb = act ? cj * bpb : 0.0;
#pragma unroll
  for (int o = JL / 2; o >= 1; o >>= 1) {
    fa += __shfl_xor(fa, o, kWave);
    fb += __shfl_xor(fb, o, kWave);
  }
  if (row && j == 0) bt[b * N + r] = LOWER ? fb - fa : fa - fb;
  // bc_j: the block's rows in a fixed tree (lanes with the same j, then the wavefronts)
  double v = act ? dta * bpa : 0.0;
#pragma unroll
  for (int o = 32; o >= JL; o >>= 1) v += __shfl_xor(v, o, kWave);
  if ((int)(threadIdx.x % kWave) < JL) red[threadIdx.x / kWave][j] = v;
  __syncthreads();
  if (threadIdx.x < (unsigned)J) {
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < kRowsPerBlock / kWave; ++w) sum += red[w][threadIdx.x];
    part[((int64_t)b * gridDim.x + blockIdx.x) * J + threadIdx.x] = sum;
  }
}
// one wavefront per (series, j): lanes strided over the blocks, a fixed tree
__global__ __launch_bounds__(kWave) void k_rev_bc(int J, int64_t nblk, const double *__restrict__ part,
                                                  double *__restrict__ bc) {
  const int64_t b = blockIdx.x / J;
  const int j = (int)(blockIdx.x % J);
  double sum = 0.0;
  for (int64_t q = threadIdx.x; q < nblk; q += kWave) sum += part[(b * nblk + q) * J + j];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, kWave);
  if (threadIdx.x == 0) bc[blockIdx.x] = sum;
}
}  // namespace c2rl

// lower / solve select the op (solve_lower_rev, solve_upper_rev, matmul_lower_rev, matmul_upper_rev); the caller has
// checked that the opposite sweep takes its time-parallel form for this shape.  Not inside graph captures (temporaries).
extern "C" int c2_internal_sweep_rev_long(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs,
                                          const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *U,
                                          const double *V, const double *Y, const double *Z, const double *F,
                                          const double *bZ, double *bt, double *bc, double *bU, double *bV, double *bY,
                                          c2_stream_t stream) {
  using namespace c2rl;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &capturing);
  if (capturing != hipStreamCaptureStatusNone || (!solve && bY == bZ)) return C2_ERR_UNSUPPORTED;
  if (J > 32) return C2_ERR_UNSUPPORTED;
  int JL = 1;
  while (JL < J) JL *= 2;
  const int64_t nblk = (N + kRowsPerBlock / JL - 1) / (kRowsPerBlock / JL);
  const size_t nws = (size_t)B * N * J * nrhs, npart = (size_t)B * nblk * J;
  void *tmp = nullptr;
  if (c2::temp_alloc(&tmp, (nws + npart) * sizeof(double), s) != hipSuccess) {
    (void)hipGetLastError();
    return C2_ERR_UNSUPPORTED;
  }
  double *Fa = (double *)tmp, *part = Fa + nws;
  int rc;
  if (solve) rc = lower ? c2_solve_upper(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, bZ, bY, Fa, stream)
                        : c2_solve_lower(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, bZ, bY, Fa, stream);
  else rc = lower ? c2_matmul_upper(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, bZ, bY, Fa, 1, stream)
                  : c2_matmul_lower(B, N, J, nrhs, t, t_bs, c, c_bs, U, V, bZ, bY, Fa, 1, stream);
  if (rc == C2_OK) {
    const dim3 grid((unsigned)nblk, (unsigned)B);
    const double *X = solve ? bY : bZ, *Q = solve ? Z : Y;
    double *outA = lower ? bU : bV, *outB = lower ? bV : bU;
#define C2_RL(JL_, LO, SO)                                                                                              \
  hipLaunchKernelGGL((k_rev_rows<JL_, LO, SO>), grid, dim3(kRowsPerBlock), 0, s, N, (int)J, nrhs, t, t_bs, c, c_bs, X, Q, \
                     F, (const double *)Fa, bt, outA, outB, part)
#define C2_RLJ(JL_)                                                             \
  do {                                                                          \
    if (lower) { if (solve) C2_RL(JL_, true, true); else C2_RL(JL_, true, false); }   \
    else       { if (solve) C2_RL(JL_, false, true); else C2_RL(JL_, false, false); } \
  } while (0)
    switch (JL) {
      case 1: C2_RLJ(1); break;
      case 2: C2_RLJ(2); break;
      case 4: C2_RLJ(4); break;
      case 8: C2_RLJ(8); break;
      case 16: C2_RLJ(16); break;
      default: C2_RLJ(32); break;
    }
#undef C2_RLJ
#undef C2_RL
    hipLaunchKernelGGL(k_rev_bc, dim3((unsigned)(B * J)), dim3(kWave), 0, s, (int)J, nblk, (const double *)part, bc);
    if (hipGetLastError() != hipSuccess) rc = C2_ERR_HIP;
  }
  if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
  return rc;
}
