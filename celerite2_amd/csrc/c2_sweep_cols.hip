// c2_sweep_cols.hip -- forward sweeps (solve_lower / solve_upper / matmul_lower / matmul_upper without the workspace,
// internal.hpp:105-189) with 9 ... 32 right-hand sides at J = 8 on batches that fill the chip.
//
// k_sweepK (c2_sweep.hip) gives such a call 16 or 32 lanes per series: four or two series per wavefront, rows of 72 - 256
// bytes fetched one by one -- 3.84 ms at nine right-hand sides where eight take 1.76 (B = 8192, N = 4096:
// profiles/r03_nrhs_scan.md), because it is the NUMBER of memory instructions a wavefront keeps in flight that bounds the
// step, and a request that serves four series costs a slot like one that serves eight.  Here a series keeps its EIGHT
// lanes and lane k owns NC columns (k, k + 8, k + 16, ...; NC = 2, 3, 4), so a wavefront still carries eight series and
//   * the width-8 rows move as halves of aligned 128-byte lines (as in k_sweep8_lines);
//   * Y and Z move in groups of FOUR ROWS per series -- 4 nrhs doubles, one dense run of 288 ... 1024 bytes -- as 16-byte
//     (even nrhs) or 8-byte pieces, through an LDS ring of two groups (Y, requested two groups ahead) and a tile of one (Z);
// per four steps and eight series: 4 + 2 + nrhs / 4 + nrhs / 4 requests instead of 5 per step and four series.
// Every row is a regular step (the first one with F = 0 and x = 0: z = y, forward.hpp:168 / 205), groups that lie inside
// the series run without a guard.  Z may alias Y (a group of Z is written after its rows of Y were read, two groups
// behind the requests).  Not here: the workspace F, products that accumulate into Z, J != 8 -- they keep k_sweepK.
#include <cstdint>
#include <type_traits>

#include "../../include/celerite2_amd.h"
#include "c2_dispatch.hpp"
#include "c2_loglik_helpers.hpp"
#include "c2_rscatter.hpp"

namespace c2sc {
using namespace c2;

constexpr int J = 8, SPW = 8;
constexpr int kAS = 8 * J + 8;   // LDS stride (doubles) of a series in the ring of 8 width-8 rows: 576 B

template <int NC, bool LOWER, bool SOLVE, bool V2>
__global__ __launch_bounds__(kWave) void k_sweepC(int64_t B, int64_t N, int nrhs, const double *t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *U,
                                                  const double *V, const double *Y, double *Z) {
  constexpr int NCP = 8 * NC;             // columns of the LDS rows
  constexpr int YS = 8 * NCP + 8;         // stride of a series in the Y ring (8 rows): = 8 mod 32 doubles, conflict-free
  constexpr int ZS = 4 * NCP + 8;         // ... in the Z tile (4 rows)
  constexpr int NP = V2 ? 2 * NC : 4 * NC;   // pieces of 16 / 8 bytes a lane moves per group and array
  __shared__ __attribute__((aligned(16))) double Aq[SPW * kAS], Bq[SPW * kAS], Yq[SPW * YS], oZ[SPW * ZS];
  __shared__ __attribute__((aligned(16))) double tq[SPW][8], pq[SPW][4][J];   // the decay vectors of a group's four rows
  const int lane = threadIdx.x, sl = lane >> 3, k = lane & 7;
  const int64_t b = (int64_t)blockIdx.x * SPW + sl;
  const int hrow = k >> 2, col2 = 2 * (k & 3);
  constexpr int dir = LOWER ? 1 : -1;
  const double *tb = t + b * t_bs;
  const double *Ab = (LOWER ? V : U) + b * N * J, *Bb = (LOWER ? U : V) + b * N * J;   // fed into F / applied to F
  const double *Yb = Y + b * N * (int64_t)nrhs;
  double *Zb = Z + b * N * (int64_t)nrhs;
  const double cj = c[b * c_bs + k];
  const int64_t G = (N + 3) >> 2;          // groups of four rows
  const int64_t nelem = N * (int64_t)nrhs; // doubles of a series of Y

  // piece i of this lane inside a group: element e = (V2 ? 2 : 1) (k + 8 i) of the group's 4 nrhs doubles -> (row, column)
  int poff[NP];   // its offset in a 4-row LDS tile of NCP columns (-1: beyond the group)
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = (V2 ? 2 : 1) * (k + 8 * i);
    poff[i] = e < 4 * nrhs ? (e / nrhs) * NCP + (e % nrhs) : -1;
  }
  // two groups in flight, in two sets of plain registers chosen at compile time (a struct of arrays handed through the
  // lambdas lands in scratch: 760 scratch instructions and 4.7 ms at 16 right-hand sides)
  constexpr int NY = V2 ? 2 * NP : NP;
  double qa[2][4], qb[2][4], qt[2], qy[2][NY];
  auto req = [&](auto set_tag, int64_t g) {   // clamped: rows beyond the series read its last valid elements (never used)
    constexpr int S = decltype(set_tag)::value;
    g = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2) {
      int64_t row = 4 * g + 2 * l2 + hrow; row = row < N ? row : N - 1;
      const double2 va = *reinterpret_cast<const double2 *>(Ab + row * J + col2);
      const double2 vb = *reinterpret_cast<const double2 *>(Bb + row * J + col2);
      qa[S][2 * l2] = va.x; qa[S][2 * l2 + 1] = va.y; qb[S][2 * l2] = vb.x; qb[S][2 * l2 + 1] = vb.y;
    }
    int64_t trow = 4 * g + (k & 3); trow = trow < N ? trow : N - 1;
    qt[S] = tb[trow];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int64_t e = 4 * g * (int64_t)nrhs + (V2 ? 2 : 1) * (k + 8 * i);
      if constexpr (V2) {
        e = e < nelem - 1 ? e : nelem - 2;
        const double2 v = *reinterpret_cast<const double2 *>(Yb + e);
        qy[S][2 * i] = v.x; qy[S][2 * i + 1] = v.y;
      } else {
        e = e < nelem ? e : nelem - 1;
        qy[S][i] = Yb[e];
      }
    }
  };
  auto put = [&](auto set_tag, int64_t g) {   // into slot g & 1 of the rings
    constexpr int S = decltype(set_tag)::value;
    const int s4 = 4 * (int)(g & 1);
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2) {
      const int o = sl * kAS + (s4 + 2 * l2 + hrow) * J + col2;
      *reinterpret_cast<double2 *>(&Aq[o]) = make_double2(qa[S][2 * l2], qa[S][2 * l2 + 1]);
      *reinterpret_cast<double2 *>(&Bq[o]) = make_double2(qb[S][2 * l2], qb[S][2 * l2 + 1]);
    }
    tq[sl][s4 + (k & 3)] = qt[S];   // (lanes of equal k & 3 write the same value)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (poff[i] >= 0) {
        const int o = sl * YS + s4 * NCP + poff[i];
        if constexpr (V2) *reinterpret_cast<double2 *>(&Yq[o]) = make_double2(qy[S][2 * i], qy[S][2 * i + 1]);
        else Yq[o] = qy[S][i];
      }
    }
  };
  auto flush = [&](int64_t g, auto guard_tag) {   // the finished group of Z: the same pieces, LDS -> memory
    constexpr bool GUARD = decltype(guard_tag)::value;
    double v[V2 ? 2 * NP : NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int o = sl * ZS + (poff[i] >= 0 ? poff[i] : 0);
      if constexpr (V2) { const double2 w = *reinterpret_cast<const double2 *>(&oZ[o]); v[2 * i] = w.x; v[2 * i + 1] = w.y; }
      else v[i] = oZ[o];
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t e = 4 * g * (int64_t)nrhs + (V2 ? 2 : 1) * (k + 8 * i);
      if (poff[i] >= 0 && (!GUARD || e < nelem)) {
        if constexpr (V2) *reinterpret_cast<double2 *>(Zb + e) = make_double2(v[2 * i], v[2 * i + 1]);
        else Zb[e] = v[i];
      }
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;

  double Fj[J][NC], xprev[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    xprev[q] = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) Fj[j][q] = 0.0;
  }
  // one step: row n (ring slot n & 7), the row before it m = n - dir (the first row: m = n, p = 1, x = 0, F = 0)
  // the decay vectors of the four rows of group g, formed together off the chain F -> z -> F (four independent exponentials
  // per lane, ONE fence per group: a step that forms its own vector pays two LDS round trips and the exponential's
  // latency on every row -- 4.8 against 3.35 ms at 16 right-hand sides on the kernel this one replaces)
  auto decays = [&](const int64_t g, const bool first) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = 4 * g + r;
      const int sn = (int)(n & 7);
      const bool start = first && r == (LOWER ? 0 : 3);            // the first row of the sweep: p = 1 (F = 0 anyway)
      const int sm = start ? sn : (int)((n - dir) & 7);
      const double tn = tq[sl][sn], tm = tq[sl][sm];
      pq[sl][r][k] = exp_decay(cj * (LOWER ? tm - tn : tn - tm));
    }
  };
  auto step = [&](const int64_t n, const bool first) __attribute__((always_inline)) {
    const int sn = (int)(n & 7), sm = first ? sn : (int)((n - dir) & 7);
    double yn[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) yn[q] = Yq[sl * YS + sn * NCP + k + 8 * q];
    double red[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) red[q] = 0.0;
#pragma unroll
    for (int j = 0; j < J; j += 2) {
      const double2 p2 = *reinterpret_cast<const double2 *>(&pq[sl][n & 3][j]);
      const double2 a2 = *reinterpret_cast<const double2 *>(&Aq[sl * kAS + sm * J + j]);
      const double2 b2 = *reinterpret_cast<const double2 *>(&Bq[sl * kAS + sn * J + j]);
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        Fj[j][q] = p2.x * fma(a2.x, xprev[q], Fj[j][q]);           // internal.hpp:140, 143 / 183, 186
        red[q] = fma(b2.x, Fj[j][q], red[q]);
        Fj[j + 1][q] = p2.y * fma(a2.y, xprev[q], Fj[j + 1][q]);
        red[q] = fma(b2.y, Fj[j + 1][q], red[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const double zn = SOLVE ? yn[q] - red[q] : red[q];             // internal.hpp:144 / 187 (products: Z was zeroed)
      oZ[sl * ZS + (int)(n & 3) * NCP + k + 8 * q] = zn;
      xprev[q] = SOLVE ? zn : yn[q];
    }
  };

  // groups in the order of the sweep (two per trip: one register set each); rows of a group in the order of the sweep
  const int64_t g0 = LOWER ? 0 : G - 1;
  bool first = true;
  auto group = [&](auto set_tag, const int64_t g) __attribute__((always_inline)) {
    put(set_tag, g);
    req(set_tag, g + 2 * dir);
    lds_order();
    const bool full = 4 * g + 3 < N;   // (wavefront-uniform)
    if (full) {
      decays(g, first);
      lds_order();
#pragma unroll
      for (int r = 0; r < 4; ++r) { step(4 * g + (LOWER ? r : 3 - r), first); first = false; }
      lds_order();
      flush(g, No{});
    } else {
      // the partial group at the end of the series (the upper sweeps meet it first): its first row in sweep order is N - 1
      const int64_t nfirst = LOWER ? 4 * g : N - 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = 4 * g + r;
        if (n < N) {
          const int sn = (int)(n & 7);
          const bool start = first && n == nfirst;
          const int sm = start ? sn : (int)((n - dir) & 7);
          pq[sl][r][k] = exp_decay(cj * (LOWER ? tq[sl][sm] - tq[sl][sn] : tq[sl][sn] - tq[sl][sm]));
        }
      }
      lds_order();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = 4 * g + (LOWER ? r : 3 - r);
        if (n < N) { step(n, first); first = false; }
      }
      lds_order();
      flush(g, Yes{});
    }
    lds_order();
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  req(S0{}, g0);
  req(S1{}, g0 + dir);
  for (int64_t gi = 0; gi < G; gi += 2) {
    group(S0{}, g0 + dir * gi);
    if (gi + 1 < G) group(S1{}, g0 + dir * (gi + 1));
  }
}


// =====================================================================================================================
// The reverse sweeps (internal.hpp:191-303) of the same shapes: the workspace F of the forward sweep (J nrhs doubles a row:
// the bulk of the traffic), X = Z (solves) / Y (products) and the cotangent bZ in; bt, bc, bB, bA, bY out.  Eight lanes
// per series, NC columns per lane as above; the three sums over the right-hand sides are summed over a lane's columns
// first and reduce-scattered over the eight lanes (c2_rscatter.hpp) as in k_sweep8_rev_lines.
//
// The step at row n pairs it with m = n + dir, the NEXT row of the sweep: arrays indexed by n (B rows, F, bZ of a product;
// bB, bt out) live in 4-row tiles of the current group, arrays indexed by m (A rows, X, bZ of a solve) in tiles of FIVE
// rows -- the current group and, in slot 4, the first row of the next one, written from the registers that hold that group
// (requested two groups ahead) before they are put as a whole.  bA / bY rows (indexed by m) complete one step early: their
// group leaves after its third step, the group of bB / bt after the fourth.
// =====================================================================================================================
constexpr int kB4 = 34;   // LDS stride (doubles) of a series in a tile of four width-8 rows: 272 B, conflict-free b128
constexpr int kA5 = 42;   // ... of five width-8 rows: 336 B

template <int NC, bool LOWER, bool SOLVE, bool V2>
__global__ __launch_bounds__(kWave) void k_sweepC_rev(int64_t B, int64_t N, int nrhs, int ld, int c0, int acc, const double *__restrict__ t,
                                                      int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                      const double *__restrict__ U, const double *__restrict__ V,
                                                      const double *__restrict__ Y, const double *__restrict__ Z,
                                                      const double *__restrict__ F, const double *__restrict__ bZ,
                                                      double *__restrict__ bt, double *__restrict__ bc,
                                                      double *__restrict__ bU, double *__restrict__ bV,
                                                      double *__restrict__ bY) {
  constexpr int NCP = 8 * NC;
  constexpr int XS = 5 * NCP + 8;            // five rows of NCP columns: = 8 or 24 mod 32 doubles, conflict-free b64
  constexpr int ZS = 4 * NCP + 8;            // the bY tile (four rows)
  constexpr int FS = J * NCP + 8;            // a workspace row [column][j] per series
  constexpr int PPR = V2 ? NC / 2 : NC;      // pieces of 16 / 8 bytes per lane and ROW of an nrhs-wide array
  static_assert(!V2 || NC % 2 == 0, "16-byte pieces: an even number of columns per lane");
  constexpr int NP = 4 * PPR;                // ... per group of four rows
  constexpr int NY = V2 ? 2 * NP : NP;
  constexpr int NF = 4 * NC;                 // 16-byte pieces of a workspace row per lane
  constexpr int RF = 2;                      // workspace rows in flight (four: 64 more registers, the kernel spills)
  __shared__ __attribute__((aligned(16))) double Aq[SPW * kA5], Bq[SPW * kB4], Xq[SPW * XS], Zq[SPW * XS];
  __shared__ __attribute__((aligned(16))) double ftile[SPW * FS], oY[SPW * ZS], oA[SPW * kB4], oB[SPW * kB4];
  __shared__ __attribute__((aligned(16))) double tq[SPW][8], pq[SPW][4][J], oT[SPW][4];
  const int lane = threadIdx.x, sl = lane >> 3, k = lane & 7;
  const int64_t b = (int64_t)blockIdx.x * SPW + sl;
  const int hrow = k >> 2, col2 = 2 * (k & 3);
  constexpr int dir = LOWER ? -1 : 1;           // the sweep visits rows n, n + dir, ...; m = n + dir
  constexpr int r_first = LOWER ? 3 : 0;        // position inside its group of the row the sweep meets first
  constexpr int r_last = LOWER ? 0 : 3;
  const double *tb = t + b * t_bs;
  const double *Bb = (LOWER ? U : V) + b * N * J, *Ab = (LOWER ? V : U) + b * N * J;
  double *bBb = (LOWER ? bU : bV) + b * N * J, *bAb = (LOWER ? bV : bU) + b * N * J;
  // nrhs columns c0 .. c0 + nrhs - 1 of arrays whose rows hold ld of them (a slice of a wider right-hand side: the second
  // slice ADDS its sums over the right-hand sides -- bt, bc, bB, bA -- to what the first one wrote: acc)
  const double *Xb = (SOLVE ? Z : Y) + b * N * (int64_t)ld + c0, *Zb = bZ + b * N * (int64_t)ld + c0;
  double *bYb = bY + b * N * (int64_t)ld + c0, *btb = bt + b * N;
  const double *Fb = F + b * N * (int64_t)(J * ld) + (int64_t)c0 * J;
  const double cj = c[b * c_bs + k];
  constexpr double sgn = SOLVE ? -1.0 : 1.0;
  const int64_t G = (N + 3) >> 2;
  const int frow = J * nrhs;                    // doubles of this slice of a workspace row
  const int64_t fld = (int64_t)J * ld;          // ... and the pitch of the rows
  bool vq[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) vq[q] = k + 8 * q < nrhs;

  // piece i of a group: row i / PPR of the group, columns (V2 ? 2 : 1) (k + 8 (i % PPR)) ...
  int pcol[PPR];   // first column of this lane's piece i2 of a row (-1: beyond the slice)
#pragma unroll
  for (int i2 = 0; i2 < PPR; ++i2) {
    const int cc = (V2 ? 2 : 1) * (k + 8 * i2);
    pcol[i2] = cc < nrhs ? cc : -1;
  }
  // ---- a group of four rows of every input: ONE register set, requested a group ahead (two sets do not fit the register
  // file next to the workspace rows); the row the m-indexed tiles keep of the group after it travels on its own -----------
  double qa[1][4], qb[1][4], qt[1], qx[1][NY], qz[1][NY];
  double2 ca; double ct, cx[NC], cz[NC];
  auto req_carry = [&](int64_t g) {   // the first row (in sweep order) of group g
    g = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
    int64_t row = 4 * g + r_first; row = row < N ? row : N - 1;
    ca = *reinterpret_cast<const double2 *>(Ab + row * J + col2);
    ct = tb[row];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int64_t e = row * (int64_t)ld + (vq[q] ? k + 8 * q : 0);
      cx[q] = Xb[e]; cz[q] = Zb[e];
    }
  };
  auto put_carry = [&]() {            // into slot 4 of the m-indexed tiles
    *reinterpret_cast<double2 *>(&Aq[sl * kA5 + 4 * J + col2]) = ca;   // (lanes k and k + 4 write the same piece)
    tq[sl][4] = ct;
#pragma unroll
    for (int q = 0; q < NC; ++q) { Xq[sl * XS + 4 * NCP + k + 8 * q] = cx[q]; Zq[sl * XS + 4 * NCP + k + 8 * q] = cz[q]; }
  };
  auto req = [&](auto set_tag, int64_t g) {
    constexpr int S = decltype(set_tag)::value;
    g = g < 0 ? 0 : (g > G - 1 ? G - 1 : g);
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2) {
      int64_t row = 4 * g + 2 * l2 + hrow; row = row < N ? row : N - 1;
      const double2 va = *reinterpret_cast<const double2 *>(Ab + row * J + col2);
      const double2 vb = *reinterpret_cast<const double2 *>(Bb + row * J + col2);
      qa[S][2 * l2] = va.x; qa[S][2 * l2 + 1] = va.y; qb[S][2 * l2] = vb.x; qb[S][2 * l2 + 1] = vb.y;
    }
    int64_t trow = 4 * g + (k & 3); trow = trow < N ? trow : N - 1;
    qt[S] = tb[trow];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int64_t row = 4 * g + i / PPR; row = row < N ? row : N - 1;
      const int64_t e = row * (int64_t)ld + (pcol[i % PPR] >= 0 ? pcol[i % PPR] : 0);
      if constexpr (V2) {
        const double2 vx = *reinterpret_cast<const double2 *>(Xb + e), vz = *reinterpret_cast<const double2 *>(Zb + e);
        qx[S][2 * i] = vx.x; qx[S][2 * i + 1] = vx.y; qz[S][2 * i] = vz.x; qz[S][2 * i + 1] = vz.y;
      } else {
        qx[S][i] = Xb[e]; qz[S][i] = Zb[e];
      }
    }
  };
  auto put = [&](auto set_tag) {   // the whole group into slots 0 .. 3 of every tile
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2) {
      const int r = 2 * l2 + hrow;
      *reinterpret_cast<double2 *>(&Aq[sl * kA5 + r * J + col2]) = make_double2(qa[S][2 * l2], qa[S][2 * l2 + 1]);
      *reinterpret_cast<double2 *>(&Bq[sl * kB4 + r * J + col2]) = make_double2(qb[S][2 * l2], qb[S][2 * l2 + 1]);
    }
    tq[sl][k & 3] = qt[S];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (pcol[i % PPR] >= 0) {
        const int o = sl * XS + (i / PPR) * NCP + pcol[i % PPR];
        if constexpr (V2) {
          *reinterpret_cast<double2 *>(&Xq[o]) = make_double2(qx[S][2 * i], qx[S][2 * i + 1]);
          *reinterpret_cast<double2 *>(&Zq[o]) = make_double2(qz[S][2 * i], qz[S][2 * i + 1]);
        } else {
          Xq[o] = qx[S][i]; Zq[o] = qz[S][i];
        }
      }
    }
  };
  // ---- workspace rows: RF register sets, one per position of the row inside its group (mod RF) ---------------------------
  double fq[RF][2 * NF];
  auto req_F = [&](auto set_tag, int64_t n) {
    constexpr int S = decltype(set_tag)::value;
    n = n < 0 ? 0 : (n < N ? n : N - 1);
    const double *a = Fb + n * fld;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      int e = 2 * (k + 8 * i); e = e < frow ? e : frow - 2;
      const double2 v = *reinterpret_cast<const double2 *>(a + e);
      fq[S][2 * i] = v.x; fq[S][2 * i + 1] = v.y;
    }
  };
  auto put_F = [&](auto set_tag) {
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < NF; ++i)
      if (2 * (k + 8 * i) < frow)
        *reinterpret_cast<double2 *>(&ftile[sl * FS + 2 * (k + 8 * i)]) = make_double2(fq[S][2 * i], fq[S][2 * i + 1]);
  };
  // ---- output tiles -> memory ------------------------------------------------------------------------------------------------
  auto flush_lines = [&](const double *tile, double *base, int64_t g, auto guard_tag) {   // four width-8 rows: two lines
    constexpr bool GUARD = decltype(guard_tag)::value;
    double2 v[2];
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2) v[l2] = *reinterpret_cast<const double2 *>(&tile[sl * kB4 + (2 * l2 + hrow) * J + col2]);
    if (acc) {   // (wavefront-uniform) the second slice of a wide right-hand side: add to the first slice's sums
#pragma unroll
      for (int l2 = 0; l2 < 2; ++l2) {
        const int64_t row = 4 * g + 2 * l2 + hrow;
        if (!GUARD || row < N) {
          const double2 w = *reinterpret_cast<const double2 *>(base + row * J + col2);
          v[l2].x += w.x; v[l2].y += w.y;
        }
      }
    }
#pragma unroll
    for (int l2 = 0; l2 < 2; ++l2) {
      const int64_t row = 4 * g + 2 * l2 + hrow;
      if (!GUARD || row < N) *reinterpret_cast<double2 *>(base + row * J + col2) = v[l2];
    }
  };
  auto flush_Y = [&](int64_t g, auto guard_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value;
    double v[NY];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int o = sl * ZS + (i / PPR) * NCP + (pcol[i % PPR] >= 0 ? pcol[i % PPR] : 0);
      if constexpr (V2) { const double2 w = *reinterpret_cast<const double2 *>(&oY[o]); v[2 * i] = w.x; v[2 * i + 1] = w.y; }
      else v[i] = oY[o];
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t row = 4 * g + i / PPR;
      if (pcol[i % PPR] >= 0 && (!GUARD || row < N)) {
        const int64_t e = row * (int64_t)ld + pcol[i % PPR];
        if constexpr (V2) *reinterpret_cast<double2 *>(bYb + e) = make_double2(v[2 * i], v[2 * i + 1]);
        else bYb[e] = v[i];
      }
    }
  };
  auto flush_t = [&](int64_t g, auto guard_tag) {
    constexpr bool GUARD = decltype(guard_tag)::value;
    double v = oT[sl][k & 3];
    if (k < 4 && (!GUARD || 4 * g + k < N)) {
      if (acc) v += btb[4 * g + k];
      btb[4 * g + k] = v;
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  double bF[J][NC], bzrun[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
#pragma unroll
    for (int j = 0; j < J; ++j) bF[j][q] = 0.0;
  }
  const int64_t nfar = LOWER ? N - 1 : 0, nnear = LOWER ? 0 : N - 1;
#pragma unroll
  for (int q = 0; q < NC; ++q) bzrun[q] = vq[q] ? Zb[nfar * (int64_t)ld + k + 8 * q] : 0.0;
  double carry = 0.0, bce = 0.0;
  // the far end receives nothing from the sweep: bA = 0, bY = its own cotangent (solves) or 0 (products)
  oA[sl * kB4 + (int)(nfar & 3) * J + k] = 0.0;
#pragma unroll
  for (int q = 0; q < NC; ++q) oY[sl * ZS + (int)(nfar & 3) * NCP + k + 8 * q] = SOLVE ? bzrun[q] : 0.0;

  if ((nfar & 3) == r_last) {   // the far end alone in its group (a lower sweep with N % 4 == 1): no step completes that group
    lds_order();
    flush_lines(oA, bAb, nfar >> 2, Yes{});
    flush_Y(nfar >> 2, Yes{});
    lds_order();
  }

  double pk[4], dtk[4];   // this lane's element of the decay vectors of the current group's rows, and their time steps
  auto decays = [&](const int64_t g) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rm = r + dir;                                     // position of the paired row m = n + dir
      const double tn = tq[sl][r], tm = tq[sl][(rm >= 0 && rm < 4) ? rm : 4];
      dtk[r] = LOWER ? tm - tn : tn - tm;                         // internal.hpp:227 / 284
      pk[r] = exp_decay(cj * dtk[r]);
      pq[sl][r][k] = pk[r];
    }
  };
  // One step: row n = 4 g + r (r known at compile time), m = n + dir.  The workspace row of n sits in register set r % RF.
  auto step = [&](const int64_t g, auto r_tag) __attribute__((always_inline)) {
    constexpr int r = decltype(r_tag)::value;
    constexpr int rm = r + dir;
    constexpr int sm = (rm >= 0 && rm < 4) ? rm : 4;              // slot of row m in the five-row tiles
    constexpr int om = (rm + 4) & 3;                              // slot of row m in the four-row output tiles
    constexpr int fs = (LOWER ? 3 - r : r) % RF;
    const int64_t n = 4 * g + r;
    put_F(std::integral_constant<int, fs>{});
    req_F(std::integral_constant<int, fs>{}, n + RF * dir);      // (the set is free once its row is on its way to LDS)
    lds_order();
    double xm[NC], bzn[NC], bzring[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const double xv = Xq[sl * XS + sm * NCP + k + 8 * q];
      const double zv = Zq[sl * XS + (SOLVE ? sm : r) * NCP + k + 8 * q];
      xm[q] = vq[q] ? xv : 0.0;
      bzring[q] = vq[q] ? zv : 0.0;
      bzn[q] = SOLVE ? bzrun[q] : bzring[q];
    }
    double pbB[J], pbp[J], pbA[J], acc[NC];
#pragma unroll
    for (int q = 0; q < NC; ++q) acc[q] = 0.0;
#pragma unroll
    for (int j = 0; j < J; j += 2) {
      const double2 p2 = *reinterpret_cast<const double2 *>(&pq[sl][r][j]);
      const double2 b2 = *reinterpret_cast<const double2 *>(&Bq[sl * kB4 + r * J + j]);
      const double2 a2 = *reinterpret_cast<const double2 *>(&Aq[sl * kA5 + sm * J + j]);
      const double pv[2] = {p2.x, p2.y}, bv[2] = {b2.x, b2.y}, av[2] = {a2.x, a2.y};
      double Fn[2][NC];   // the workspace entries (j, j + 1) of this lane's columns (invalid columns: zero)
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const double2 v = *reinterpret_cast<const double2 *>(&ftile[sl * FS + (k + 8 * q) * J + j]);
        Fn[0][q] = vq[q] ? v.x : 0.0; Fn[1][q] = vq[q] ? v.y : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int jj = j + u;
        double sB = 0.0, sp = 0.0, sA = 0.0;
#pragma unroll
        for (int q = 0; q < NC; ++q) {
          sB = fma(bzn[q], pv[u] * Fn[u][q], sB);                // internal.hpp:232 / 289
          bF[jj][q] = fma(sgn * bv[u], bzn[q], bF[jj][q]);       // internal.hpp:233 / 290
          sp = fma(Fn[u][q], bF[jj][q], sp);                     // internal.hpp:236 / 293
          bF[jj][q] *= pv[u];                                    // internal.hpp:241 / 298
          sA = fma(xm[q], bF[jj][q], sA);                        // update_f::reverse (internal.hpp:59 / 80)
          acc[q] = fma(av[u], bF[jj][q], acc[q]);                // ... and the cotangent of row m (internal.hpp:60 / 81)
        }
        pbB[jj] = sB; pbp[jj] = sp; pbA[jj] = sA;
      }
    }
#pragma unroll
    for (int q = 0; q < NC; ++q) {   // cotangent of row m: solves fold it into the running bZ, products write it out
      if (SOLVE) bzrun[q] = bzring[q] + acc[q];
      oY[sl * ZS + om * NCP + k + 8 * q] = SOLVE ? bzrun[q] : acc[q];
    }
    int ko;
    const double rB = sgn * rscatter8<8>(pbB, k, ko);
    const double rp = rscatter8<8>(pbp, k, ko);
    const double rA = rscatter8<8>(pbA, k, ko);
    const double bpe = rp * pk[r];
    oB[sl * kB4 + r * J + k] = rB;
    oA[sl * kB4 + om * J + k] = rA;
    bce = fma(dtk[r], bpe, bce);
    const double phi = gsum<8>(cj * bpe);
    // LOWER: bt[n] -= phi, bt[m] += phi -> row n is complete now (it got +phi of the previous step); UPPER: mirrored
    oT[sl][r] = LOWER ? carry - phi : phi - carry;
    carry = phi;
    lds_order();
  };

  // ---- the sweep ---------------------------------------------------------------------------------------------------------------
  const int64_t g0 = LOWER ? G - 1 : 0;
  req(S0{}, g0);
  put(S0{});                                // (the only synchronous wait)
  req(S0{}, g0 + dir);
  req_carry(g0 + dir);
  // the workspace rows of the first RF steps.  The register set of a row is its position in sweep order inside its group,
  // mod RF (known at compile time in every step); the far end of a lower sweep starts inside its group when N % 4 != 0
  {
    const int p0 = (int)(LOWER ? 3 - (nfar & 3) : (nfar & 3)) % RF;
    auto first_rows = [&](auto i_tag) __attribute__((always_inline)) {
      constexpr int i = decltype(i_tag)::value;
      const int64_t n = nfar + i * dir;
      const int set = (p0 + i) % RF;
      if (set == 0) req_F(std::integral_constant<int, 0>{}, n);
      if (RF > 1 && set == 1) req_F(std::integral_constant<int, 1 % RF>{}, n);
      if constexpr (RF == 4) {
        if (set == 2) req_F(std::integral_constant<int, 2>{}, n);
        if (set == 3) req_F(std::integral_constant<int, 3>{}, n);
      }
    };
    first_rows(std::integral_constant<int, 0>{});
    first_rows(std::integral_constant<int, 1>{});
    if constexpr (RF == 4) { first_rows(std::integral_constant<int, 2>{}); first_rows(std::integral_constant<int, 3>{}); }
  }
  auto group = [&](const int64_t g) __attribute__((always_inline)) {
    put_carry();                            // the first row of the next group into slot 4
    req_carry(g + 2 * dir);
    lds_order();
    decays(g);
    lds_order();
    // interior group: four rows, each with a step, none of them an end of the series
    // (wavefront-uniform; an upper sweep whose last step of the group pairs with the near end -- alone in its group -- takes
    // the guarded path, which sends that group's bA / bY out with the step)
    const bool interior = 4 * g + 3 < N && (LOWER ? g >= 1 : 4 * g + 3 <= N - 3);
    if (interior) {
      if constexpr (LOWER) {
        step(g, std::integral_constant<int, 3>{}); step(g, std::integral_constant<int, 2>{});
        step(g, std::integral_constant<int, 1>{});
        flush_lines(oA, bAb, g, No{}); flush_Y(g, No{});
        step(g, std::integral_constant<int, 0>{});
        flush_lines(oB, bBb, g, No{}); flush_t(g, No{});
      } else {
        step(g, std::integral_constant<int, 0>{}); step(g, std::integral_constant<int, 1>{});
        step(g, std::integral_constant<int, 2>{});
        flush_lines(oA, bAb, g, No{}); flush_Y(g, No{});
        step(g, std::integral_constant<int, 3>{});
        flush_lines(oB, bBb, g, No{}); flush_t(g, No{});
      }
    } else {
      auto edge = [&](auto r_tag) __attribute__((always_inline)) {
        constexpr int r = decltype(r_tag)::value;
        const int64_t n = 4 * g + r, m = n + dir;
        if (n < N && n != nnear) {
          step(g, r_tag);
          if ((m & 3) == r_last || m == nnear) { flush_lines(oA, bAb, m >> 2, Yes{}); flush_Y(m >> 2, Yes{}); }
          if (r == r_last) { flush_lines(oB, bBb, g, Yes{}); flush_t(g, Yes{}); }
          lds_order();
        }
      };
      if constexpr (LOWER) {
        edge(std::integral_constant<int, 3>{}); edge(std::integral_constant<int, 2>{});
        edge(std::integral_constant<int, 1>{}); edge(std::integral_constant<int, 0>{});
      } else {
        edge(std::integral_constant<int, 0>{}); edge(std::integral_constant<int, 1>{});
        edge(std::integral_constant<int, 2>{}); edge(std::integral_constant<int, 3>{});
      }
    }
    lds_order();
    put(S0{});                              // the next group takes the tiles
    req(S0{}, g + 2 * dir);
    lds_order();
  };
  for (int64_t gi = 0; gi < G; ++gi) group(g0 + dir * gi);
  // the near end: no bB, bt = what the last step left
  oB[sl * kB4 + (int)(nnear & 3) * J + k] = 0.0;
  oT[sl][nnear & 3] = LOWER ? carry : -carry;
  lds_order();
  flush_lines(oB, bBb, nnear >> 2, Yes{});
  flush_t(nnear >> 2, Yes{});
  bc[b * J + k] = acc ? bc[b * J + k] + bce : bce;
}

}  // namespace c2sc

using namespace c2sc;

// Returns C2_ERR_UNSUPPORTED when the shape does not fit (the caller keeps its other kernels).  B8 = the series served
// here (whole wavefronts of eight); the caller runs the B - B8 left over on k_sweepK.
extern "C" int c2_internal_sweep_cols(int lower, int solve, int64_t B, int64_t N, int64_t Jw, int64_t nrhs, const double *t,
                                      int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                      const double *Y, double *Z, int64_t *B8, c2_stream_t stream) {
  *B8 = 0;
  if (Jw != 8 || nrhs < 9 || nrhs > 32 || B < 8 || N < 8) return C2_ERR_UNSUPPORTED;
  if ((((uintptr_t)U | (uintptr_t)V) % 16) != 0 || (((uintptr_t)Y | (uintptr_t)Z) % 8) != 0) return C2_ERR_UNSUPPORTED;
  if (c2::opt::has(c2::opt::k_sweep_cols) && c2::opt::ival(c2::opt::k_sweep_cols) == 0) return C2_ERR_UNSUPPORTED;
  const bool v2 = (nrhs % 2 == 0) && (((uintptr_t)Y | (uintptr_t)Z) % 16) == 0;
  const int NC = (int)((nrhs + 7) / 8);
  const int64_t nb = B / 8;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)nb);
#define C2_SC(NC_, LO, SO, V2_)                                                                                       \
  hipLaunchKernelGGL((k_sweepC<NC_, LO, SO, V2_>), grid, dim3(kWave), 0, s, nb * 8, N, (int)nrhs, t, t_bs, c, c_bs, U, V, Y, Z)
#define C2_SC_V(NC_, LO, SO) do { if (v2) C2_SC(NC_, LO, SO, true); else C2_SC(NC_, LO, SO, false); } while (0)
#define C2_SC_D(NC_)                                                                       \
  do {                                                                                     \
    if (lower) { if (solve) C2_SC_V(NC_, true, true); else C2_SC_V(NC_, true, false); }    \
    else { if (solve) C2_SC_V(NC_, false, true); else C2_SC_V(NC_, false, false); }        \
  } while (0)
  switch (NC) {
    case 2: C2_SC_D(2); break;
    case 3: C2_SC_D(3); break;
    default: C2_SC_D(4); break;
  }
#undef C2_SC_D
#undef C2_SC_V
#undef C2_SC
  if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;
  *B8 = nb * 8;
  return C2_OK;
}

// The reverse sweeps of the same shapes (9 .. 16 right-hand sides at J = 8; with three and four columns per lane the tiles
// no longer leave a CU its four wavefronts: not instantiated).  B8: the series served here.
extern "C" int c2_internal_sweep_cols_rev(int lower, int solve, int64_t B, int64_t N, int64_t Jw, int64_t nrhs, const double *t,
                                          int64_t t_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                                          const double *Y, const double *Z, const double *F, const double *bZ, double *bt,
                                          double *bc, double *bU, double *bV, double *bY, int64_t *B8, c2_stream_t stream) {
  *B8 = 0;
  if (Jw != 8 || nrhs < 9 || nrhs > 32 || B < 8 || N < 8) return C2_ERR_UNSUPPORTED;
  if ((((uintptr_t)U | (uintptr_t)V | (uintptr_t)F | (uintptr_t)bU | (uintptr_t)bV) % 16) != 0) return C2_ERR_UNSUPPORTED;
  if (c2::opt::has(c2::opt::k_sweep_cols) && c2::opt::ival(c2::opt::k_sweep_cols) == 0) return C2_ERR_UNSUPPORTED;
  const int64_t nb = B / 8;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)nb);
  // up to 16 right-hand sides in one launch; 17 .. 32 as two slices of the columns, the second adding its sums over the
  // right-hand sides to the first one's (three and four columns per lane do not fit the register file: 50 - 160 ms)
  const bool al16 = (((uintptr_t)Y | (uintptr_t)Z | (uintptr_t)bZ | (uintptr_t)bY) % 16) == 0 && nrhs % 2 == 0;
  // (16 + the rest: two slices of about the same width measured no better -- 24 as 12 + 12: 26.8 against 23.3 ms -- a slice
  // that does not start on a 128-byte line costs more than its columns)
  for (int c0 = 0; c0 < (int)nrhs; c0 += 16) {
    const int w = (int)nrhs - c0 < 16 ? (int)nrhs - c0 : 16;
    const bool v2 = al16 && w % 2 == 0;
    const int acc = c0 > 0 ? 1 : 0;
    const bool narrow = w <= 8 && !(c2::opt::has(c2::opt::k_sweep_cols) && c2::opt::ival(c2::opt::k_sweep_cols) == 2);
#define C2_SCR(NC_, LO, SO, V2_)                                                                                        \
  hipLaunchKernelGGL((k_sweepC_rev<NC_, LO, SO, V2_>), grid, dim3(kWave), 0, s, nb * 8, N, w, (int)nrhs, c0, acc, t, t_bs, c, \
                     c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY)
    // a slice of at most eight columns (17 .. 24 right-hand sides: the second slice) runs ONE column per lane
#define C2_SCR_V(LO, SO) do { if (narrow) C2_SCR(1, LO, SO, false); else if (v2) C2_SCR(2, LO, SO, true); else C2_SCR(2, LO, SO, false); } while (0)
    if (lower) { if (solve) C2_SCR_V(true, true); else C2_SCR_V(true, false); }
    else { if (solve) C2_SCR_V(false, true); else C2_SCR_V(false, false); }
#undef C2_SCR_V
#undef C2_SCR
  }
  if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;
  *B8 = nb * 8;
  return C2_OK;
}
