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
