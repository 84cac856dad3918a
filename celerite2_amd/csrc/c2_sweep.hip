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
#include <cstdint>
#include <type_traits>

#include "c2_dispatch.hpp"
#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

#ifndef C2_SWEEP1_PAIRLINES
#define C2_SWEEP1_PAIRLINES 1
#endif
namespace c2 {

// LN >= 0 (G = J = 8, U and V 16-byte aligned): the two width-8 rows of a step are requested as halves of the aligned
// 128-byte lines (rows 2 l, 2 l + 1) they share -- one 16-byte piece per lane, one request per line and array every two
// steps, four lines ahead -- and reach their lanes through LDS tiles of four rows.  The sweep is bound by the number of
// memory instructions a single wavefront keeps in flight (40 VALU instructions in 830 cycles per step, 69 % of them waiting;
// profiles/r03_sweep_rev_lines.md).  LN = parity of the position r inside a block of eight steps at which the sweep enters
// a new line: 1 for the lower sweeps and for the upper ones on an even number of rows, 0 for the upper ones on an odd one.
template <int G, int R, bool LOWER, bool SOLVE, bool PAD, int LN = -1>
__global__ __launch_bounds__(kWave) void k_sweep1(int64_t B, int64_t N, int Jrt, const double *t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *U,
                                                  const double *V, const double *Y, double *Z, double *F, int zero_z) {
  constexpr int SPW = kWave / G, NV = (R + G - 1) / G;
  constexpr bool LINES = LN >= 0;
  static_assert(!LINES || (G == 8 && !PAD && R == 8), "lines: eight lanes per series, blocks of eight steps");
  __shared__ __attribute__((aligned(16))) double tAB[LINES ? 2 : 1][LINES ? 4 : 1][kWave];   // [A / B][row & 3][lane]
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

  // Step 0 is the first step of block 0 from a neutral state (no row before it: A = 0, x = 0, F = 0 at t_0): it yields
  // Z = Y (solve, forward.hpp:168,205) / Z = 0 (matmul called with zero_z) / Z unchanged (matmul accumulate) and the zero
  // workspace row (internal.hpp:127 / :170) like any other step -- so blocks cover positions [R b, R b + R) and the
  // transposed requests of t, y, z are whole aligned runs (profiles/r05_alignment.md).
  constexpr bool HOLDZ = C2_SWEEP1_PAIRLINES && G == 8 && R == 8 && NV == 1 && !PAD;
  double hZ = 0.0;
  int64_t hZrow = 0;
  bool hZok = false;
  const int64_t r0 = rowof(0);
  double xprev = 0.0;
  double aprev = 0.0;
  double tprev = tb[r0];
  double Fs = 0.0;

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
  vload(0); vstage(0);
  vload(R); vstage(1);
  vload(2 * R);

  double ra[LINES ? 1 : R], rb[LINES ? 1 : R];
  auto load_row = [&](int r, int64_t s) {
    s = (s < N) ? s : N - 1;
    const int64_t n = rowof(s);
    ra[r] = act ? Ab[n * J] : 0.0;
    rb[r] = act ? Bb[n * J] : 0.0;
  };
  // LINES: ring of R / 2 lines per array in the order the sweep enters them (line k of the sweep = l0 +/- k; slot k mod 4);
  // lane j holds piece j of a line: row 2 l + (j >> 2), columns 2 (j & 3), 2 (j & 3) + 1
  double2 la[LINES ? R / 2 : 1], lb[LINES ? R / 2 : 1];
  const int hrow = j >> 2, hcol = 2 * (j & 3);
  const double *Al = (LOWER ? V : U) + (L.b0 + L.sl) * N * J + hcol, *Bl = (LOWER ? U : V) + (L.b0 + L.sl) * N * J + hcol;
  const int64_t l0 = rowof(0) >> 1, lmax = (N - 1) >> 1;
  auto line_load = [&](const double *base, int64_t k) -> double2 {
    int64_t l = LOWER ? l0 + k : l0 - k;
    l = l < 0 ? 0 : (l > lmax ? lmax : l);
    int64_t row = 2 * l + hrow;
    row = row < N ? row : N - 1;
    return *reinterpret_cast<const double2 *>(base + row * J);
  };
  auto line_stage = [&](int which, int64_t k, double2 v) {   // rows 2 l, 2 l + 1 of the tile (slots row & 3)
    const int64_t l = LOWER ? l0 + k : l0 - k;
    *reinterpret_cast<double2 *>(&tAB[which][(2 * (int)(l & 1) + hrow)][grp * G + hcol]) = v;
  };
  int64_t kline = 0;   // lines entered so far
  if constexpr (LINES) {
#pragma unroll
    for (int q = 0; q < R / 2; ++q) { la[q] = line_load(Al, q); lb[q] = line_load(Bl, q); }
    if constexpr (LN == 1) {   // step 0 (an upper sweep on an odd number of rows) is the second row, in sweep order, of its line
      line_stage(0, 0, la[0]); line_stage(1, 0, lb[0]);
      la[0] = line_load(Al, R / 2); lb[0] = line_load(Bl, R / 2);
      kline = 1;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) load_row(r, r);
  }
  lds_order();

  auto block = [&](int64_t s0, int q, auto checked_tag) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t s = s0 + r;
      if (!CHECKED || s < N) {
        const double tn = sin_[q][0][grp][r], yn = sin_[q][1][grp][r], zin = sin_[q][2][grp][r];
        double an, bn;
        if constexpr (LINES) {
          if ((r & 1) == LN) {   // (r: unrolled) the sweep enters a line: slot (lines entered so far) mod 4 of the ring
            const int sl_ = (LN == 1 ? 1 + r / 2 : r / 2) % (R / 2);
            line_stage(0, kline, la[sl_]); line_stage(1, kline, lb[sl_]);
            la[sl_] = line_load(Al, kline + R / 2); lb[sl_] = line_load(Bl, kline + R / 2);
            ++kline;
            lds_order();
          }
          const int64_t n = rowof(s);
          an = tAB[0][n & 3][lane]; bn = tAB[1][n & 3][lane];
        } else {
          an = ra[r]; bn = rb[r];
          load_row(r, s + R);
        }
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
    if constexpr (HOLDZ) {
      // (a block's run of Z is HALF a 128-byte line per series: the half the sweep reaches first waits in a register and leaves
      // with the other, back to back -- profiles/r06_halflines.md.  In-place Z == Y stays legal: the rows a held store covers were
      // read two blocks ago.  C2_SWEEP1_PAIRLINES=0: as they come.)
      const int64_t row = rowof(s0 + j);
      const bool ok = !CHECKED || s0 + j < N;
      const double v = sout[grp][j];
      const int64_t rs = LOWER ? s0 : N - R - s0;            // lowest row of the run
      const bool first = LOWER ? ((rs >> 3) & 1) == 0 : ((rs >> 3) & 1) != 0;   // (uniform)
      if (first) {
        hZ = v; hZrow = row; hZok = ok;
      } else {
        if (ok) zb[row] = v;
        if (hZok) zb[hZrow] = hZ;
        hZok = false;
      }
    } else {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if ((G * NV == R || idx < R) && (!CHECKED || s0 + idx < N)) zb[rowof(s0 + idx)] = sout[grp][idx];
    }
    }
    vstage(q);
    vload(s0 + 3 * R);
    lds_order();
  };
  int64_t s0 = 0;
  int q = 0;
  for (; s0 + 2 * R <= N; s0 += R, q ^= 1) block(s0, q, std::false_type{});
  for (; s0 < N; s0 += R, q ^= 1) block(s0, q, std::true_type{});
  if constexpr (HOLDZ) { if (hZok) zb[hZrow] = hZ; }   // a half line whose partner never came
}

// -----------------------------------------------------------------------------------------------------------------
// Reverse of the single-rhs sweeps: internal::forward_rev / backward_rev (internal.hpp:191-246 / 248-303) behind
// solve_lower_rev, solve_upper_rev, matmul_lower_rev, matmul_upper_rev (reverse.hpp:87-217), nrhs = 1.
// Positions q = 0 .. N-1 run against the forward sweep: row(q) = N-1-q (lower) or q (upper).  Step u = 0 .. N-2
// handles n = row(u) (its U/V row, its workspace row F_n, outputs bt_n and the row gradient bB_n) and m = row(u+1)
// (its V/U row, x_m, bZ_m; outputs bY_m and the row gradient bA_m).  Same stream handling as k_sweep1.
// -----------------------------------------------------------------------------------------------------------------
// LN >= 0 (G = J = 8, 16-byte aligned row arrays): the five width-8 rows of a step -- B_n, F_n, A_m in; bB_n, bA_m out -- move
// as halves of the aligned 128-byte lines (rows 2 l, 2 l + 1) they share: one 16-byte piece per lane, one request per line and
// array every two steps, the inputs four lines ahead through four-row LDS tiles, the outputs through two-row tiles (without
// its two row stores the sweep runs 39 % faster: it is the number of memory instructions in flight that bounds it,
// profiles/r03_sweep_rev_lines.md).  LN = parity of the position r inside a block of eight at which the sweep enters a
// new line with its row n: 0 for the upper sweeps, N mod 2 for the lower ones; row m enters its lines at the other parity.
template <int G, int R, bool LOWER, bool SOLVE, bool PAD, int LN = -1>
__global__ __launch_bounds__(kWave) void k_sweep1_rev(int64_t B, int64_t N, int Jrt, const double *__restrict__ t,
                                                      int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                      const double *__restrict__ U, const double *__restrict__ V,
                                                      const double *__restrict__ Y, const double *__restrict__ Z,
                                                      const double *__restrict__ F, const double *__restrict__ bZ,
                                                      double *__restrict__ bt, double *__restrict__ bc,
                                                      double *__restrict__ bU, double *__restrict__ bV,
                                                      double *__restrict__ bY) {
  constexpr int SPW = kWave / G, NV = (R + G - 1) / G;
  constexpr bool LINES = LN >= 0;
  static_assert(!LINES || (G == 8 && !PAD && R == 8), "lines: eight lanes per series, blocks of eight steps");
  // Scalar tiles hold the ALIGNED positions R b .. R b + R - 1 of a block (whole 64-byte runs per series; requested at
  // R b + 1 .. R b + R every run straddled two sectors: +13 % bytes fetched, profiles/r05_alignment.md): step u = R b + r reads
  // t, x, bZ of position u + 1 from entry r + 1 (entry 0 of the NEXT block's tile for r = R - 1) and leaves bY of position u + 1
  // in entry r of the block's buffer; the run R b .. R b + R - 1 of bY is the previous block's last entry and R - 1 of this one's.
  __shared__ __attribute__((aligned(16))) double sin_[2][3][SPW][R];  // t, x, bZ of two blocks
  __shared__ __attribute__((aligned(16))) double sout[SPW][R];        // bt (position u)
  __shared__ __attribute__((aligned(16))) double soutY[2][SPW][R];    // bY (position u + 1) of this block and the one before
  __shared__ __attribute__((aligned(16))) double tin[LINES ? 3 : 1][LINES ? 4 : 1][kWave];   // [B / F / A][row & 3][lane]
  __shared__ __attribute__((aligned(16))) double tout[LINES ? 2 : 1][LINES ? 2 : 1][kWave];  // [bB / bA][row & 1][lane]
  const int J = PAD ? Jrt : G;
  const Geo<G> L(B, J);
  const int j = L.j, grp = L.lane / G, lane = L.lane;
  const bool act = PAD ? L.act : true;
  const bool st = PAD ? (L.valid && act) : true;
  const int64_t on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + L.jj;
  const double *tb = t + L.b0 * t_bs + (int64_t)L.sl * t_bs;
  const double *xb = (SOLVE ? Z : Y) + L.b0 * N + on, *bzb = bZ + L.b0 * N + on;
  double *btb = bt + L.b0 * N + on, *byb = bY + L.b0 * N + on;
  const double *Ab = (LOWER ? V : U) + L.b0 * N * J + oj;  // row fed into F (index m)
  const double *Bb = (LOWER ? U : V) + L.b0 * N * J + oj;  // row applied to F (index n)
  double *bAb = (LOWER ? bV : bU) + L.b0 * N * J + oj, *bBb = (LOWER ? bU : bV) + L.b0 * N * J + oj;
  const double *Fb = F + L.b0 * N * J + oj;
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  constexpr double sgn = SOLVE ? -1.0 : 1.0;
  auto rowof = [&](int64_t q) { return LOWER ? N - 1 - q : q; };

  constexpr bool HOLD = C2_SWEEP1_PAIRLINES && G == 8 && R == 8 && NV == 1 && !PAD;
  double hT = 0.0, hY = 0.0;
  int64_t hldrow = 0;
  bool hTok = false, hYok = false;
  const int64_t r0 = rowof(0);
  double bz = bzb[r0];
  soutY[1][grp][R - 1] = SOLVE ? bz : 0.0;      // bY of position 0: reverse.hpp:112 (bY = bZ) / :178 (bY = 0); leaves with block 0
  if constexpr (!LINES) { if (st) bAb[r0 * J] = 0.0; }       // never receives a contribution
  double tprev = tb[r0];
  double bF = 0.0, bcj = 0.0, carry = 0.0;

  // ---- LINES: rings of four lines per input array in the order the sweep enters them, output tiles -------------------
  constexpr int dirl = LOWER ? -1 : 1;
  constexpr int PN = LINES ? LN : 0, PM = 1 - PN;          // parity of r at which row n / row m enters a line
  constexpr int preN = PN == 1 ? 1 : 0, preM = PM == 1 ? 1 : 0;   // lines the prologue stages (r = 0 is then a second row)
  const int hrow = j >> 2, hcol = 2 * (j & 3);
  const int64_t lmax = (N - 1) >> 1;
  const int64_t sbase = (L.b0 + L.sl) * N * J + hcol;
  const double *Bl = (LOWER ? U : V) + sbase, *Fl = F + sbase, *Al = (LOWER ? V : U) + sbase;
  double *bBl = (LOWER ? bU : bV) + sbase, *bAl = (LOWER ? bV : bU) + sbase;
  const int64_t l0n = r0 >> 1, l0m = rowof(N > 1 ? 1 : 0) >> 1;
  auto line_load = [&](const double *base, int64_t l) -> double2 {
    l = l < 0 ? 0 : (l > lmax ? lmax : l);
    int64_t row = 2 * l + hrow;
    row = row < N ? row : N - 1;
    return *reinterpret_cast<const double2 *>(base + row * J);
  };
  auto line_stage = [&](int which, int64_t l, double2 v) {
    *reinterpret_cast<double2 *>(&tin[which][2 * (int)(l & 1) + hrow][grp * G + hcol]) = v;
  };
  auto line_flush = [&](int which, double *base, int64_t l, auto guard_tag) {   // a finished line of bB / bA
    constexpr bool GUARD = decltype(guard_tag)::value;
    const int64_t row = 2 * l + hrow;
    const double2 v = *reinterpret_cast<const double2 *>(&tout[which][hrow][grp * G + hcol]);
    if (!GUARD || (row >= 0 && row < N)) *reinterpret_cast<double2 *>(base + row * J) = v;
  };
  double2 lb[LINES ? R / 2 : 1], lf[LINES ? R / 2 : 1], la[LINES ? R / 2 : 1];
  int64_t kn = 0, km = 0;   // lines entered so far by row n / row m
  if constexpr (LINES) {
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
      lb[q] = line_load(Bl, l0n + dirl * q); lf[q] = line_load(Fl, l0n + dirl * q); la[q] = line_load(Al, l0m + dirl * q);
    }
    if constexpr (preN == 1) {
      line_stage(0, l0n, lb[0]); line_stage(1, l0n, lf[0]);
      lb[0] = line_load(Bl, l0n + dirl * (R / 2)); lf[0] = line_load(Fl, l0n + dirl * (R / 2));
      kn = 1;
    }
    if constexpr (preM == 1) {
      line_stage(2, l0m, la[0]);
      la[0] = line_load(Al, l0m + dirl * (R / 2));
      km = 1;
    }
    tout[1][r0 & 1][lane] = 0.0;   // bA of the first row: never receives a contribution
    lds_order();
    if constexpr (PN == 1) line_flush(1, bAl, r0 >> 1, std::true_type{});   // ... and no step completes its line
  }

  double vt[NV], vx[NV], vbz[NV];
  auto vload = [&](int64_t qb) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t q = qb + m * G + j;
      q = (q < N) ? q : N - 1;
      const int64_t n = rowof(q);
      vt[m] = tb[n]; vx[m] = xb[n]; vbz[m] = bzb[n];
    }
  };
  auto vstage = [&](int s) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == R || idx < R) {
        sin_[s][0][grp][idx] = vt[m]; sin_[s][1][grp][idx] = vx[m]; sin_[s][2][grp][idx] = vbz[m];
      }
    }
  };
  vload(0); vstage(0);
  vload(R); vstage(1);
  vload(2 * R);

  double rb[LINES ? 1 : R], rf[LINES ? 1 : R], ra[LINES ? 1 : R];
  auto load_row = [&](int r, int64_t u) {  // B and F rows of position u, A row of position u+1
    const int64_t qn = (u < N) ? u : N - 1, qm = (u + 1 < N) ? u + 1 : N - 1;
    const int64_t n = rowof(qn), m = rowof(qm);
    rb[r] = act ? Bb[n * J] : 0.0;
    rf[r] = act ? Fb[n * J] : 0.0;
    ra[r] = act ? Ab[m * J] : 0.0;
  };
  if constexpr (!LINES) {
#pragma unroll
    for (int r = 0; r < R; ++r) load_row(r, r);
  }
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
        const double tm = sin_[s1][0][grp][e1], xm = sin_[s1][1][grp][e1], bzm = sin_[s1][2][grp][e1];
        double bn, Fn, am;
        if constexpr (LINES) {
          if ((r & 1) == PN) {   // (r: unrolled) row n enters a line of B and F: slot (lines entered so far) mod 4
            const int sl_ = (preN + (r - PN) / 2) % (R / 2);
            const int64_t l = l0n + dirl * kn;
            line_stage(0, l, lb[sl_]); line_stage(1, l, lf[sl_]);
            lb[sl_] = line_load(Bl, l + dirl * (R / 2)); lf[sl_] = line_load(Fl, l + dirl * (R / 2));
            ++kn;
          } else {               // row m enters a line of A
            const int sl_ = (preM + (r - PM) / 2) % (R / 2);
            const int64_t l = l0m + dirl * km;
            line_stage(2, l, la[sl_]);
            la[sl_] = line_load(Al, l + dirl * (R / 2));
            ++km;
          }
          lds_order();
          bn = tin[0][n & 3][lane]; Fn = tin[1][n & 3][lane]; am = tin[2][m & 3][lane];
        } else {
          bn = rb[r]; Fn = rf[r]; am = ra[r];
          load_row(r, u + R);
        }
        const double dt = tm - tprev;  // lower: t[m] - t[n]; upper: t[n] - t[m] with the roles of prev/next swapped
        const double dte = LOWER ? dt : -dt;
        tprev = tm;
        const double p = exp_decay(cj * dte);
        // reverse of update_z (internal.hpp:232-233 / 289-290)
        const double val = bz * (p * Fn);
        bF = fma(sgn * bn, bz, bF);
        const double dotFbF = Fn * bF;
        if constexpr (LINES) tout[0][n & 1][lane] = sgn * val;
        else if (st) bBb[n * J] = sgn * val;
        // reverse of the decay (internal.hpp:236-241 / 293-298)
        const double bp = dotFbF * p;
        bcj = fma(dte, bp, bcj);
        bF *= p;
        // update_f::reverse (internal.hpp:55-63 matmul, 76-84 solve)
        const double bam = xm * bF;
        double f = cj * bp, g = am * bF;
        gsum2<G>(f, g);
        sout[grp][r] = LOWER ? carry - f : f - carry;
        carry = f;
        const double out = SOLVE ? bzm + g : g;
        soutY[s][grp][r] = out;
        bz = SOLVE ? out : bzm;
        if constexpr (LINES) {
          tout[1][m & 1][lane] = bam;
          lds_order();
          // the line this step completes: bA's where row m is the second row of its line, bB's where row n is
          if ((r & 1) == PN) line_flush(1, bAl, m >> 1, checked_tag);
          else line_flush(0, bBl, n >> 1, checked_tag);
        } else if (st) {
          bAb[m * J] = bam;
        }
      }
    }
    lds_order();
    if constexpr (HOLD) {
      // (a block's run of bt / bY is HALF a 128-byte line per series; written when ready, the two halves of a line reach memory
      // eight steps apart and are merged on the memory side one by one -- profiles/r06_halflines.md.  The half the sweep reaches
      // first waits in a register per stream and leaves with the other, back to back.  C2_SWEEP1_PAIRLINES=0: as they come.)
      const int64_t row = rowof(u0 + j);
      const bool okT = !CHECKED || u0 + j + 1 < N, okY = !CHECKED || u0 + j < N;
      const double vT = sout[grp][j], vY = j == 0 ? soutY[s ^ 1][grp][R - 1] : soutY[s][grp][j - 1];
      const int64_t rs = LOWER ? N - R - u0 : u0;            // lowest row of the run
      const bool first = LOWER ? ((rs >> 3) & 1) != 0 : ((rs >> 3) & 1) == 0;   // (uniform) the first half of its line the sweep meets
      if (first) {
        hT = vT; hY = vY; hldrow = row; hTok = okT; hYok = okY;
      } else {
        if (okT) btb[row] = vT;
        if (hTok) btb[hldrow] = hT;
        if (okY) byb[row] = vY;
        if (hYok) byb[hldrow] = hY;
        hTok = hYok = false;
      }
    } else {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == R || idx < R) {
        if (!CHECKED || u0 + idx + 1 < N) btb[rowof(u0 + idx)] = sout[grp][idx];
        if (!CHECKED || u0 + idx < N) byb[rowof(u0 + idx)] = idx == 0 ? soutY[s ^ 1][grp][R - 1] : soutY[s][grp][idx - 1];
      }
    }
    }
    vstage(s);
    vload(u0 + 3 * R);
    lds_order();
  };
  int64_t u0 = 0;
  int s = 0;
  if constexpr (LINES) {   // (the first block may complete a line whose other row lies beyond the end of the series: guarded)
    if (u0 + 1 < N) { block(u0, s, std::true_type{}); u0 += R; s ^= 1; }
  }
  for (; u0 + 2 * R + 1 <= N; u0 += R, s ^= 1) block(u0, s, std::false_type{});
  for (; u0 + 1 < N; u0 += R, s ^= 1) block(u0, s, std::true_type{});

  if constexpr (HOLD) {   // a half line still waiting for a partner that never came
    if (hTok) btb[hldrow] = hT;
    if (hYok) byb[hldrow] = hY;
  }
  if (u0 < N) byb[rowof(u0)] = soutY[s ^ 1][grp][R - 1];   // the last block's last entry: position N - 1 = R b opens a run of its own
  const int64_t rl = rowof(N - 1);
  btb[rl] = LOWER ? carry : -carry;
  if constexpr (LINES) {   // bB of the last row is zero; the lines no step completed leave now
    tout[0][rl & 1][lane] = 0.0;
    lds_order();
    line_flush(0, bBl, rl >> 1, std::true_type{});
    line_flush(1, bAl, rl >> 1, std::true_type{});
    bc[L.b * J + j] = bcj;
  } else if (st) {
    bBb[rl * J] = 0.0;  // bU.row(0) / bV.row(N-1) never touched
    bc[L.b * J + j] = bcj;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// Several right-hand sides: lanes <-> right-hand sides.  A series is walked by KL lanes (KL = nrhs rounded up to a
// power of two, at most 64; more right-hand sides -> tiles of 64 in blockIdx.y), lane k owning column k of the
// J x nrhs state F: z_n[k] = y_n[k] -/+ sum_j B_n[j] F[j][k] needs NO cross-lane reduction and the Y / Z rows move
// as dense runs.  What every lane needs per step are the three width-J vectors p_n, A_{n-1}, B_n: lane j (< J)
// loads / computes element j, they are staged in LDS and read back as broadcasts (ds_read_b128, 3J/2 per step).
// Same semantics as k_sweep (internal.hpp:105-189): in-place Z == Y is legal (rows are read R steps ahead of the
// row being written), matmul accumulates into Z unless zero_z.
// -----------------------------------------------------------------------------------------------------------------
// WF: also write the F workspace.  Shapes with nrhs <= KL (one tile of right-hand sides) and J == JM only (the launcher
// checks): a row of F is then nrhs x J doubles, lane k holding the J consecutive entries F[n, j + J k]; they leave through an LDS tile as dense
// 16-byte-per-lane runs (stored straight from the lanes they would be 16-byte pieces of 64 separate lines).
template <int KL, int JM, int R, bool LOWER, bool SOLVE, bool WF>
__global__ __launch_bounds__(kWave) void k_sweepK(int64_t B, int64_t N, int J, int64_t nrhs, const double *t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *U,
                                                  const double *V, const double *Y, double *Z, double *F, int zero_z) {
  static_assert(JM <= KL, "the lanes of a series also carry its width-J vectors");
  constexpr int SPW = kWave / KL;
  __shared__ __attribute__((aligned(16))) double rowbuf[2][SPW][3][KL];  // p_n, A_{n-1}, B_n of two consecutive steps
  __shared__ __attribute__((aligned(16))) double ftile[WF ? SPW * KL * JM : 2];
  const int lane = threadIdx.x, sl = lane / KL, k = lane % KL;
  int64_t b = (int64_t)blockIdx.x * SPW + sl;
  const bool vb = b < B;
  if (!vb) b = B - 1;
  int64_t kk = (int64_t)blockIdx.y * KL + k;
  const bool vk = vb && kk < nrhs;
  if (kk >= nrhs) kk = nrhs - 1;
  const bool actj = k < J;  // this lane also carries element k of the width-J vectors
  const int jk = actj ? k : 0;
  const bool loadz = !SOLVE && !zero_z;
  const double *tb = t + b * t_bs;
  const double *Ab = (LOWER ? V : U) + b * N * J + jk;  // row fed into F
  const double *Bb = (LOWER ? U : V) + b * N * J + jk;  // row applied to F
  const double *yb = Y + b * N * nrhs + kk;
  double *zb = Z + b * N * nrhs + kk;
  const double cj = actj ? c[b * c_bs + k] : 0.0;
  auto rowof = [&](int64_t s) { return LOWER ? s : N - 1 - s; };
  const int RL = (int)nrhs * JM;   // doubles in a workspace row (WF: nrhs <= KL, J == JM, one rhs tile)
  double *Frow = WF ? F + b * N * (int64_t)RL : nullptr;

  const int64_t r0 = rowof(0);
  double xprev = yb[r0 * nrhs];
  if (vk) {
    if (SOLVE) zb[r0 * nrhs] = xprev;        // forward.hpp:168, 205
    else if (zero_z) zb[r0 * nrhs] = 0.0;
  }
  if (WF && vb) {  // internal.hpp:127 / :170
#pragma unroll
    for (int qq = 0; qq < JM / 2; ++qq)
      if (qq * 2 * KL + 2 * k < RL) *reinterpret_cast<double2 *>(Frow + r0 * (int64_t)RL + qq * 2 * KL + 2 * k) = make_double2(0.0, 0.0);
  }
  double aprev = actj ? Ab[r0 * J] : 0.0;
  double tprev = tb[r0];
  double Fj[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) Fj[j] = 0.0;

  double rt[R], ra[R], rb[R], ry[R], rz[R];
  auto load_row = [&](int r, int64_t s) {
    s = (s < N) ? s : N - 1;
    const int64_t n = rowof(s);
    rt[r] = tb[n];
    ra[r] = actj ? Ab[n * J] : 0.0;
    rb[r] = actj ? Bb[n * J] : 0.0;
    ry[r] = yb[n * nrhs];
    rz[r] = loadz ? zb[n * nrhs] : 0.0;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(r, 1 + r);

  int q = 0;
  // One step; full blocks of R steps run without a condition (behind a row guard or a lane predicate the compiler cannot
  // count the memory operations in flight and waits for all of them).  Lanes beyond nrhs and the lanes of a series beyond
  // the batch are clamped copies of the last column / series: they store the same values to the same addresses.
  auto step = [&](const int r, const int64_t s) __attribute__((always_inline)) {
      {
        const int64_t n = rowof(s);
        const double tn = rt[r], an = ra[r], bn = rb[r], yn = ry[r], zin = rz[r];
        load_row(r, s + R);
        const double p = exp_decay(cj * (LOWER ? tprev - tn : tn - tprev));
        tprev = tn;
        rowbuf[q][sl][0][k] = p; rowbuf[q][sl][1][k] = aprev; rowbuf[q][sl][2][k] = bn;
        lds_order();
        double red = 0.0;
#pragma unroll
        for (int j = 0; j < JM; j += 2) {
          double2 p2, a2, b2;
          if constexpr (JM >= 2) {
            p2 = *reinterpret_cast<const double2 *>(&rowbuf[q][sl][0][j]);
            a2 = *reinterpret_cast<const double2 *>(&rowbuf[q][sl][1][j]);
            b2 = *reinterpret_cast<const double2 *>(&rowbuf[q][sl][2][j]);
          } else {
            p2 = make_double2(rowbuf[q][sl][0][0], 0.0); a2 = make_double2(rowbuf[q][sl][1][0], 0.0);
            b2 = make_double2(rowbuf[q][sl][2][0], 0.0);
          }
          const double f0 = fma(a2.x, xprev, Fj[j]);   // internal.hpp:140 / :183
          Fj[j] = p2.x * f0;                            // internal.hpp:143 / :186
          red = fma(b2.x, Fj[j], red);
          double f1 = 0.0;
          if (j + 1 < JM) {
            f1 = fma(a2.y, xprev, Fj[j + 1]);
            Fj[j + 1] = p2.y * f1;
            red = fma(b2.y, Fj[j + 1], red);
          }
          if constexpr (WF)  // saved before the decay (internal.hpp:142 / :185)
            *reinterpret_cast<double2 *>(&ftile[(sl * KL + k) * JM + j]) = make_double2(f0, f1);
        }
        if constexpr (WF) {
          lds_order();
          {
#pragma unroll
            for (int qq = 0; qq < JM / 2; ++qq) {
              const int e = qq * 2 * KL + 2 * k;   // (the row is the first nrhs columns of the tile: k-major like the workspace)
              if (e < RL)
                *reinterpret_cast<double2 *>(Frow + n * (int64_t)RL + e) =
                    *reinterpret_cast<const double2 *>(&ftile[sl * KL * JM + e]);
            }
          }
          lds_order();
        }
        const double zn = SOLVE ? yn - red : zin + red;  // internal.hpp:144 / :187
        zb[n * nrhs] = zn;
        xprev = SOLVE ? zn : yn;
        aprev = an;
        q ^= 1;
      }
  };
  int64_t s0 = 1;
  for (; s0 + R <= N; s0 += R) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, s0 + r);
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (s0 + r < N) step(r, s0 + r);
}


// ---- nrhs = J = 8 on full wavefronts: every width-8 row moves as half of an aligned 128-byte LINE -------------------------
// The forward counterpart of k_sweep8_rev_lines (c2_sweep_rev.hip, which has the measurements): k_sweepK asks for t, the two
// width-J rows and the Y row (and the Z row when it accumulates) of every step on their own and stores every Z row on its
// own -- five or six 64-byte requests a step, and it is the number of memory instructions a single wavefront keeps in
// flight that bounds the step.  Here the rows of a series pair up into the aligned lines (2 l, 2 l + 1) they share: one
// 16-byte piece per lane, one request per line and array every two steps, through LDS rings of four rows (inputs, two
// lines ahead) and a tile of two rows (Z).  The workspace row (WF) leaves as before, as a dense run through an LDS tile.
// In-place Z == Y stays legal: a line of Z is written after both of its rows were read, two lines behind the requests.
constexpr int kLS8 = 34;   // LDS stride (doubles) of a series in a ring of four 8-double rows: 272 B, conflict-free b128
constexpr int kOS8 = 18;   // ... in a tile of two rows: 144 B

template <bool LOWER, bool SOLVE, bool WF, bool LOADZ>
__global__ __launch_bounds__(kWave) void k_sweep8_lines(int64_t B, int64_t N, const double *t, int64_t t_bs,
                                                        const double *__restrict__ c, int64_t c_bs, const double *U,
                                                        const double *V, const double *Y, double *Z, double *F, int zero_z) {
  constexpr int J = 8, SPW = 8;
  __shared__ __attribute__((aligned(16))) double Aq[SPW * kLS8], Bq[SPW * kLS8], Yq[SPW * kLS8], Zq[LOADZ ? SPW * kLS8 : 2];
  __shared__ __attribute__((aligned(16))) double tq[SPW][4];
  __shared__ __attribute__((aligned(16))) double pq[SPW][J];
  __shared__ __attribute__((aligned(16))) double ftile[WF ? SPW * J * J : 2];
  __shared__ __attribute__((aligned(16))) double oZ[SPW * kOS8];
  const int lane = threadIdx.x, sl = lane >> 3, k = lane & 7;
  const int64_t b = (int64_t)blockIdx.x * SPW + sl;
  const int hrow = k >> 2, col = 2 * (k & 3);
  constexpr int dir = LOWER ? 1 : -1;           // the sweep visits rows n, n + dir, ...; the row before n is m = n - dir
  constexpr int kFirstParity = LOWER ? 0 : 1;   // parity of the row of a line the sweep meets first
  const double *tb = t + b * t_bs;
  const double *Ab = (LOWER ? V : U) + b * N * J, *Bb = (LOWER ? U : V) + b * N * J;   // fed into F / applied to F
  const double *Yb = Y + b * N * J;
  double *Zb = Z + b * N * J;
  double *Fd = WF ? F + b * N * (int64_t)(J * J) + 2 * k : nullptr;
  const double cj = c[b * c_bs + k];
  const int64_t lmax = (N - 1) >> 1;

  struct Line { double2 a, b, y, z; double tt; };
  // (values in, values out: handed around by reference the rings end up in scratch)
  auto req_line = [&](int64_t l) -> Line {
    Line R;
    l = l < 0 ? 0 : (l > lmax ? lmax : l);
    int64_t row = 2 * l + hrow; row = row < N ? row : N - 1;
    int64_t trow = 2 * l + (k & 1); trow = trow < N ? trow : N - 1;
    R.a = *reinterpret_cast<const double2 *>(Ab + row * J + col);
    R.b = *reinterpret_cast<const double2 *>(Bb + row * J + col);
    R.y = *reinterpret_cast<const double2 *>(Yb + row * J + col);
    if constexpr (LOADZ) R.z = *reinterpret_cast<const double2 *>(Zb + row * J + col);
    else R.z = make_double2(0.0, 0.0);
    R.tt = tb[trow];
    return R;
  };
  auto put_line = [&](int64_t l, const Line R) {
    const int o = sl * kLS8 + ((2 * (int)(l & 1) + hrow) * J) + col;   // slot (row & 3) of the ring
    *reinterpret_cast<double2 *>(&Aq[o]) = R.a;
    *reinterpret_cast<double2 *>(&Bq[o]) = R.b;
    *reinterpret_cast<double2 *>(&Yq[o]) = R.y;
    if constexpr (LOADZ) *reinterpret_cast<double2 *>(&Zq[o]) = R.z;
    tq[sl][2 * (int)(l & 1) + (k & 1)] = R.tt;   // (lanes of equal k & 1 write the same value)
  };
  auto flush = [&](int64_t l, auto guard_tag) {   // a finished line of Z: 16 bytes per lane, 128 per series
    constexpr bool GUARD = decltype(guard_tag)::value;
    const int64_t row = 2 * l + hrow;
    const double2 v = *reinterpret_cast<const double2 *>(&oZ[sl * kOS8 + hrow * J + col]);
    if (!GUARD || (row >= 0 && row < N)) *reinterpret_cast<double2 *>(Zb + row * J + col) = v;
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;

  // step 0: Z = Y (solve, forward.hpp:168,205) / Z = 0 (matmul called with zero_z) / untouched (matmul accumulate)
  const int64_t r0 = LOWER ? 0 : N - 1;
  double xprev = Yb[r0 * J + k];
  oZ[sl * kOS8 + (int)(r0 & 1) * J + k] = SOLVE ? xprev : (LOADZ ? Zb[r0 * J + k] : 0.0);
  if constexpr (WF) {  // internal.hpp:127 / :170
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<double2 *>(Fd + r0 * (int64_t)(J * J) + 16 * q) = make_double2(0.0, 0.0);
  }
  lds_order();
  if ((r0 & 1) != kFirstParity) flush(r0 >> 1, Yes{});   // no step completes the line of this row
  double Fj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) Fj[j] = 0.0;

  // One step (row n, the row before it m = n - dir).  EDGE: lines moved here, synchronously, with guards; otherwise the
  // rings are kept by the caller and PH says whether the step completes its line of Z (1: n is the second row of its line).
  // (Forming the decay vector a step ahead, off the chain F -> z -> F, was tried: no faster.)
  auto step = [&](const int64_t n, auto edge_tag, auto ph_tag) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    constexpr int PH = decltype(ph_tag)::value;
    const int64_t m = n - dir;
    if constexpr (EDGE) {
      put_line(m >> 1, req_line(m >> 1));
      put_line(n >> 1, req_line(n >> 1));   // (the same line again where m shares it)
      lds_order();
    }
    const double tn = tq[sl][n & 3], tm = tq[sl][m & 3];
    const double p = exp_decay(cj * (LOWER ? tm - tn : tn - tm));
    pq[sl][k] = p;
    const double yn = Yq[sl * kLS8 + (int)(n & 3) * J + k];
    double zin = 0.0;
    if constexpr (LOADZ) zin = Zq[sl * kLS8 + (int)(n & 3) * J + k];
    lds_order();
    double red = 0.0;
#pragma unroll
    for (int j = 0; j < J; j += 2) {
      const double2 p2 = *reinterpret_cast<const double2 *>(&pq[sl][j]);
      const double2 a2 = *reinterpret_cast<const double2 *>(&Aq[sl * kLS8 + (int)(m & 3) * J + j]);
      const double2 b2 = *reinterpret_cast<const double2 *>(&Bq[sl * kLS8 + (int)(n & 3) * J + j]);
      const double f0 = fma(a2.x, xprev, Fj[j]);       // internal.hpp:140 / :183
      Fj[j] = p2.x * f0;                                // internal.hpp:143 / :186
      red = fma(b2.x, Fj[j], red);
      const double f1 = fma(a2.y, xprev, Fj[j + 1]);
      Fj[j + 1] = p2.y * f1;
      red = fma(b2.y, Fj[j + 1], red);
      if constexpr (WF)  // saved before the decay (internal.hpp:142 / :185)
        *reinterpret_cast<double2 *>(&ftile[(sl * J + k) * J + j]) = make_double2(f0, f1);
    }
    const double zn = SOLVE ? yn - red : zin + red;  // internal.hpp:144 / :187
    oZ[sl * kOS8 + (int)(n & 1) * J + k] = zn;
    xprev = SOLVE ? zn : yn;
    lds_order();
    if constexpr (WF) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<double2 *>(Fd + n * (int64_t)(J * J) + 16 * q) =
            *reinterpret_cast<const double2 *>(&ftile[sl * J * J + 16 * q + 2 * k]);
    }
    if constexpr (EDGE) {
      if ((n & 1) != kFirstParity) flush(n >> 1, Yes{});
    } else if constexpr (PH == 1) {
      flush(n >> 1, No{});
    }
  };

  int64_t n = r0 + dir, left = N - 1;
  if (left > 0 && (n & 1) != kFirstParity) { step(n, Yes{}, std::integral_constant<int, 0>{}); n += dir; --left; }
  if (left >= 4) {
    // rings: the line of the row before n in LDS, the line of n and the one behind it requested
    put_line((n - dir) >> 1, req_line((n - dir) >> 1));
    Line lr0 = req_line(n >> 1), lr1 = req_line((n >> 1) + dir);
    for (; left >= 4; left -= 4, n += 4 * dir) {
      const int64_t lc = n >> 1;
      put_line(lc, lr0);
      lr0 = req_line(lc + 2 * dir);
      lds_order();
      step(n, No{}, std::integral_constant<int, 0>{});
      step(n + dir, No{}, std::integral_constant<int, 1>{});
      put_line(lc + dir, lr1);
      lr1 = req_line(lc + 3 * dir);
      lds_order();
      step(n + 2 * dir, No{}, std::integral_constant<int, 0>{});
      step(n + 3 * dir, No{}, std::integral_constant<int, 1>{});
    }
  }
  for (; left > 0; --left, n += dir) step(n, Yes{}, std::integral_constant<int, 0>{});
  // the line of the last row leaves now if no step completed it
  lds_order();
  flush((n - dir) >> 1, Yes{});
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
  const bool lines_ok = N >= 3 && (((uintptr_t)U | (uintptr_t)V) % 16) == 0 &&
                        !(opt::has(opt::k_sweep1_lines) && opt::ival(opt::k_sweep1_lines) == 0);
#define C2_SW(G, LO, SO)                                                                                           \
  do {                                                                                                             \
    if (J == G && G == 8 && lines_ok) {                                                                            \
      if (LO || N % 2 == 0)   /* the sweep enters a line at even positions (position 0 = row 0 / row N - 1) */          \
        hipLaunchKernelGGL((k_sweep1<G, 8, LO, SO, false, (G == 8 ? 0 : -1)>), grid, dim3(kWave), 0, s, B, N, (int)J, t, \
                           t_bs, c, c_bs, U, V, Y, Z, F, zero_z);                                                   \
      else                                                                                                         \
        hipLaunchKernelGGL((k_sweep1<G, 8, LO, SO, false, (G == 8 ? 1 : -1)>), grid, dim3(kWave), 0, s, B, N, (int)J, t, \
                           t_bs, c, c_bs, U, V, Y, Z, F, zero_z);                                                   \
    } else if (J == G)                                                                                             \
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

extern "C" int c2_internal_sweep1_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                      double *bc, double *bU, double *bV, double *bY, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int G_ = group_size(J);
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
  if (J == 8 && N >= 3 && (((uintptr_t)U | (uintptr_t)V | (uintptr_t)F | (uintptr_t)bU | (uintptr_t)bV) % 16) == 0 &&
      !(opt::has(opt::k_sweep1_rev_lines) && opt::ival(opt::k_sweep1_rev_lines) == 0)) {
    // rows by lines; the lower sweeps walk down from row N - 1: which rows open a line depends on the parity of N
#define C2_SWL(LO, SO, LN_)                                                                                          \
  hipLaunchKernelGGL((k_sweep1_rev<8, 8, LO, SO, false, LN_>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U, \
                     V, Y, Z, F, bZ, bt, bc, bU, bV, bY)
    if (lower) {
      if (N & 1) { if (solve) C2_SWL(true, true, 1); else C2_SWL(true, false, 1); }
      else { if (solve) C2_SWL(true, true, 0); else C2_SWL(true, false, 0); }
    } else {
      if (solve) C2_SWL(false, true, 0); else C2_SWL(false, false, 0);
    }
#undef C2_SWL
    return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
  }
#define C2_SWR(G, LO, SO)                                                                                             \
  do {                                                                                                                \
    if (J == G)                                                                                                       \
      hipLaunchKernelGGL((k_sweep1_rev<G, 8, LO, SO, false>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, \
                         U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY);                                                      \
    else                                                                                                              \
      hipLaunchKernelGGL((k_sweep1_rev<G, 8, LO, SO, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs,  \
                         U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY);                                                      \
  } while (0)
#define C2_SWR_G(LO, SO)                    \
  switch (G_) {                             \
    case 1: C2_SWR(1, LO, SO); break;       \
    case 2: C2_SWR(2, LO, SO); break;       \
    case 4: C2_SWR(4, LO, SO); break;       \
    case 8: C2_SWR(8, LO, SO); break;       \
    case 16: C2_SWR(16, LO, SO); break;     \
    default: C2_SWR(32, LO, SO); break;     \
  }
  if (lower) {
    if (solve) { C2_SWR_G(true, true); } else { C2_SWR_G(true, false); }
  } else {
    if (solve) { C2_SWR_G(false, true); } else { C2_SWR_G(false, false); }
  }
#undef C2_SWR_G
#undef C2_SWR
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// Multi-rhs sweeps with lanes over the right-hand sides.  Returns C2_ERR_UNSUPPORTED when the shape does not fit the
// mapping (J wider than the lanes of a series); the caller then takes the generic kernel.
extern "C" int c2_internal_sweepK(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t,
                                  int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                  const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  int KL = 8;
  while (KL < 64 && KL < nrhs) KL *= 2;
  const int JM = J <= 8 ? 8 : (J <= 16 ? 16 : 32);
  if (JM > KL) return C2_ERR_UNSUPPORTED;
  // the F workspace goes through the LDS tile: whole rows only
  if (F && !(nrhs <= KL && J == JM && ((uintptr_t)F) % 16 == 0)) return C2_ERR_UNSUPPORTED;
  // nrhs = J = 8: by lines on the whole wavefronts of the batch (every pointer 16-byte pieces are moved through must allow
  // that), the row-by-row kernel below on the B % 8 series left over
  if (J == 8 && nrhs == 8 && B >= 8 && N >= 8 &&
      !(opt::has(opt::k_sweepk_lines) && opt::ival(opt::k_sweepk_lines) == 0) &&
      (((uintptr_t)U | (uintptr_t)V | (uintptr_t)Y | (uintptr_t)Z | (uintptr_t)F) % 16) == 0) {
    const int64_t B8 = B / 8 * 8;
    const dim3 g8((unsigned)(B8 / 8));
    const bool loadz = !solve && !zero_z;
#define C2_S8(LO, SO, WF_, LZ)                                                                                      \
  hipLaunchKernelGGL((k_sweep8_lines<LO, SO, WF_, LZ>), g8, dim3(kWave), 0, s, B8, N, t, t_bs, c, c_bs, U, V, Y, Z, F, \
                     zero_z)
#define C2_S8F(LO, SO, LZ) do { if (F) C2_S8(LO, SO, true, LZ); else C2_S8(LO, SO, false, LZ); } while (0)
    if (lower) {
      if (solve) C2_S8F(true, true, false);
      else if (loadz) C2_S8F(true, false, true);
      else C2_S8F(true, false, false);
    } else {
      if (solve) C2_S8F(false, true, false);
      else if (loadz) C2_S8F(false, false, true);
      else C2_S8F(false, false, false);
    }
#undef C2_S8F
#undef C2_S8
    if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;
    if (B8 == B) return C2_OK;
    const int64_t o = B8 * N;   // rows of the series already done
    t += B8 * t_bs; c += B8 * c_bs;
    U += o * J; V += o * J; Y += o * nrhs; Z += o * nrhs;
    if (F) F += o * J * nrhs;
    B -= B8;
  }
  const dim3 grid((unsigned)((B + (kWave / KL) - 1) / (kWave / KL)), (unsigned)((nrhs + KL - 1) / KL));
#define C2_SK1(KL_, JM_, LO, SO)                                                                                     \
  do {                                                                                                               \
    if (F) hipLaunchKernelGGL((k_sweepK<KL_, JM_, 8, LO, SO, true>), grid, dim3(kWave), 0, s, B, N, (int)J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z);  \
    else hipLaunchKernelGGL((k_sweepK<KL_, JM_, 8, LO, SO, false>), grid, dim3(kWave), 0, s, B, N, (int)J, nrhs, t, t_bs, c, c_bs, U, V, Y, Z, F, zero_z); \
  } while (0)
#define C2_SK(KL_, JM_)                                                                                              \
  do {                                                                                                               \
    if (lower) {                                                                                                     \
      if (solve) C2_SK1(KL_, JM_, true, true); else C2_SK1(KL_, JM_, true, false);                                  \
    } else {                                                                                                         \
      if (solve) C2_SK1(KL_, JM_, false, true); else C2_SK1(KL_, JM_, false, false);                                \
    }                                                                                                                \
  } while (0)
  switch (KL * 100 + JM) {
    case 808: C2_SK(8, 8); break;
    case 1608: C2_SK(16, 8); break;
    case 1616: C2_SK(16, 16); break;
    case 3208: C2_SK(32, 8); break;
    case 3216: C2_SK(32, 16); break;
    case 3232: C2_SK(32, 32); break;
    case 6408: C2_SK(64, 8); break;
    case 6416: C2_SK(64, 16); break;
    default: C2_SK(64, 32); break;
  }
#undef C2_SK
#undef C2_SK1
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
