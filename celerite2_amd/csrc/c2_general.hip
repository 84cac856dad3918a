// c2_general.hip -- general_matmul_lower / general_matmul_upper (prediction products at new coordinates; reference
// forward.hpp:285-332, 346-392) for SEVERAL right-hand sides: lanes <-> right-hand sides.
//
// The reference walks a two-pointer merge of the sorted grids t1 (N output rows) and t2 (M rows feeding the state):
//     lower:  for each n:  while (t2[m] <= t1[n]) { F = p o F + V_m^T Y_m ; m++ }    p = exp(-c (t2[m] - t2[m-1]))
//                          Z_n += (U_n o exp(-c (t1[n] - t2[m-1]))) F
// (upper: the mirror image, from the far end).  Here a series is walked by KL lanes (KL = nrhs rounded up to a power of
// two, at least the padded width), lane k owning column k of the J x nrhs state, so neither step needs a cross-lane
// reduction and the Y / Z rows move as dense runs -- the mapping of k_sweepK (c2_sweep.hip).  The merge itself is run
// ONE EVENT PER ITERATION with both event types predicated: every series of the wavefront either absorbs its next t2
// row or emits its next t1 row, so series whose grids interleave differently do not serialise each other; the decay
// vector is computed by lanes 0..J-1 and broadcast through LDS.  Eight rows of BOTH streams are resident in an LDS ring per
// series; the row eight positions down the moving stream is requested at the top of every event and written into the
// ring three events later (the merge decides only at run time which stream advances).
//
// Workspace semantics as the reference: F[m, j * nrhs + k] (row-major), row m written when row m is absorbed, rows the
// merge never reaches left untouched, row 0 = V_0^T Y_0 (lower) / 0 (upper), row M-1 never written by the upper variant.
#include <cstdint>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2g {
using namespace c2;

template <int KL, int JM, bool LOWER, bool WF>
__global__ __launch_bounds__(kWave) void k_generalK(int64_t B, int64_t N, int64_t M, int J, int64_t nrhs,
                                                    const double *__restrict__ t1, int64_t t1_bs,
                                                    const double *__restrict__ t2, int64_t t2_bs,
                                                    const double *__restrict__ c, int64_t c_bs,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    const double *__restrict__ Y, double *Z, double *F, int zero_z) {
  static_assert(JM <= KL, "the lanes of a series also carry its width-J vectors");
  constexpr int SPW = kWave / KL;
  // decay vector, event row; two doubles of padding per vector: a series' pair is 16 (KL + 2) bytes from the next one's, so
  // the b128 broadcasts of the SPW series of a wavefront fall into distinct banks (unpadded, KL = 8: 128-byte stride, four
  // series per bank group -- 13 % of the kernel's cycles were LDS bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT)
  __shared__ __attribute__((aligned(16))) double rowbuf[SPW][2][KL + 2];   // [0]: the decay vector of the event, [1]: prologue
  const int lane = threadIdx.x, sl = lane / KL, k = lane % KL;
  int64_t b = (int64_t)blockIdx.x * SPW + sl;
  const bool vb = b < B;
  if (!vb) b = B - 1;
  int64_t kk = (int64_t)blockIdx.y * KL + k;
  const bool vk = vb && kk < nrhs;
  if (kk >= nrhs) kk = nrhs - 1;
  const bool actj = k < J;
  const int jk = actj ? k : 0;
  const double *t1b = t1 + b * t1_bs, *t2b = t2 + b * t2_bs;
  const double *Ub = U + b * N * J + jk, *Vb = V + b * M * J + jk;
  const double *Yb = Y + b * M * nrhs + kk;
  double *Zb = Z + b * N * nrhs + kk;
  double *Fb = WF ? F + b * M * (int64_t)J * nrhs + kk : nullptr;
  const double cj = actj ? c[b * c_bs + k] : 0.0;
  // positions run 0, 1, 2, ... along the direction of the walk; row(pos) is the array index
  auto rowN = [&](int64_t s) { return LOWER ? s : N - 1 - s; };
  auto rowM = [&](int64_t s) { return LOWER ? s : M - 1 - s; };

  // state after the first t2 row (forward.hpp:297-300 / 358-361)
  double Fj[JM];
  {
    const double y0 = Yb[rowM(0) * nrhs];
    rowbuf[sl][1][k] = actj ? Vb[rowM(0) * J] : 0.0;
    lds_order();
#pragma unroll
    for (int j = 0; j < JM; ++j) Fj[j] = rowbuf[sl][1][j] * y0;
    lds_order();
    if (WF && vk) {
      for (int j = 0; j < J; ++j) Fb[(LOWER ? 0 : 0) * J * nrhs + (int64_t)j * nrhs] = LOWER ? Fj[j] : 0.0;  // row 0
    }
  }
  const double tfirst = t2b[rowM(0)];
  // outputs on the near side of the first t2 row get nothing (forward.hpp:303-306 / 364-367); the F rows stay
  // untouched if there is no output at all beyond it
  int64_t n = 0;
  while (n < N && (LOWER ? t1b[rowN(n)] < tfirst : t1b[rowN(n)] >= tfirst)) ++n;
  int64_t m = 1;
  double tlast = tfirst;
  // the next row of either stream (clamped at the end of its grid)
  auto clampN = [&](int64_t s) { return rowN(s < N ? s : N - 1); };
  auto clampM = [&](int64_t s) { return rowM(s < M ? s : M - 1); };
  // RD rows of either stream live in an LDS ring per series (slot = position mod RD: time, the row of V / U, the values of Y
  // / Z).  Which stream moves is decided at the TOP of an event (it only takes the two current times): the row RD
  // positions down the moving stream is requested there, into the slot the event's own row leaves, lands in registers and
  // is written into the ring five events later -- so nobody waits for it (the first version asked for a row at the end of
  // an event and selected it into place at once: one L2 / HBM latency per event, 7.0 ms per 8192 series of 4096 + 4096 rows
  // with 8 right-hand sides; the second kept three rows of either stream in registers and shifted them by selects, plus
  // touch loads eight rows down: 178 VALU and 5.8 memory instructions per event, 4.5 ms).  Every event issues the same
  // three loads, so the compiler counts them (vmcnt) instead of draining the queue; the event's row and the current times
  // are read from the ring by position, no register ring to shift.
  constexpr int RD = 8, NSLOT = 2 * RD + 1;            // (slot 2 RD: where the request of a finished series goes)
  // doubles per series: [T: NSLOT (+1)][R: NSLOT x KL][X: NSLOT x KL], padded to 8 (mod 32): the 64-byte rows of the two
  // series a quarter-wavefront reads fall into different banks (series 2336 bytes apart: 18 quad-cycles of bank conflicts
  // per event, rocprofv3 SQ_LDS_BANK_CONFLICT)
  constexpr int RS0 = (NSLOT + (NSLOT & 1)) + 2 * NSLOT * KL;
  constexpr int RS = RS0 + ((8 - RS0 % 32) + 32) % 32;
  __shared__ __attribute__((aligned(16))) double ring[SPW * RS];
  __shared__ __attribute__((aligned(16))) double ftile[WF ? SPW * JM * KL : 2];
  // (uniform) one tile of right-hand sides, full width, an even number of doubles per row
  const bool dense_f = WF && nrhs <= KL && gridDim.y == 1 && J == JM && ((J * nrhs) % 2) == 0 && (((uintptr_t)F) % 16) == 0;
  double *rg = ring + sl * RS;
  double *rgR = rg + NSLOT + (NSLOT & 1), *rgX = rgR + NSLOT * KL;   // (16-byte aligned rows)
  static_assert(RS0 <= RS && RS % 32 == 8, "ring layout");
  for (int q = 0; q < RD; ++q) {
    const int64_t rm = clampM(m + q), rn = clampN(n + q);
    const int sm = (int)((m + q) & (RD - 1)), sn = RD + (int)((n + q) & (RD - 1));
    rg[sm] = t2b[rm]; rgR[sm * KL + k] = actj ? Vb[rm * J] : 0.0; rgX[sm * KL + k] = Yb[rm * nrhs];
    rg[sn] = t1b[rn]; rgR[sn * KL + k] = actj ? Ub[rn * J] : 0.0; rgX[sn * KL + k] = Zb[rn * nrhs];
  }
  lds_order();
  double tm = rg[(int)(m & (RD - 1))], tn = rg[RD + (int)(n & (RD - 1))];
  struct Pend { double t, r, x; int slot; };
  Pend p0{0.0, 0.0, 0.0, 2 * RD}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0;
  auto event = [&](Pend &issue, Pend &resolve) __attribute__((always_inline)) {
    const bool live = n < N;
    // lower: absorb while t2[m] <= t1[n];  upper (walking down): absorb while t2[m] > t1[n]
    const bool absorb = live && m < M && (LOWER ? tm <= tn : tm > tn);
    const bool emit = live && !absorb;
    const int so = absorb ? (int)(m & (RD - 1)) : RD + (int)(n & (RD - 1));   // the event's row in the ring
    {   // the request of this event: the row RD positions down the moving stream
      const int64_t rm = clampM(m + RD), rn = clampN(n + RD);
      const double *pt = absorb ? t2b + rm : t1b + rn;
      const double *pr = absorb ? Vb + rm * J : Ub + rn * J;
      const double *px = absorb ? Yb + rm * nrhs : (const double *)Zb + rn * nrhs;
      issue.t = *pt; issue.r = actj ? *pr : 0.0; issue.x = *px;
      issue.slot = live ? so : 2 * RD;
    }
    const double xe = rgX[so * KL + k];   // y_m / z_n
    const double tev = absorb ? tm : tn;
    const double p = exp_decay(cj * (LOWER ? tlast - tev : tev - tlast));
    rowbuf[sl][0][k] = p;
    lds_order();
    double red = 0.0;
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      double2 p2, r2;
      if constexpr (JM >= 2) {
        p2 = *reinterpret_cast<const double2 *>(&rowbuf[sl][0][j]);
        r2 = *reinterpret_cast<const double2 *>(&rgR[so * KL + j]);
      } else {
        p2 = make_double2(rowbuf[sl][0][0], 0.0); r2 = make_double2(rgR[so * KL], 0.0);
      }
      const double f0 = p2.x * Fj[j];
      red = fma(r2.x, f0, red);                       // emit: (U_n o p) . F          (forward.hpp:329 / 389)
      Fj[j] = absorb ? fma(r2.x, xe, f0) : Fj[j];      // absorb: F = p o F + V_m^T y_m (forward.hpp:320-323 / 380-383)
      if (j + 1 < JM) {
        const double f1 = p2.y * Fj[j + 1];
        red = fma(r2.y, f1, red);
        Fj[j + 1] = absorb ? fma(r2.y, xe, f1) : Fj[j + 1];
      }
    }
    if constexpr (WF) {
      if (dense_f) {   // the row (J x nrhs doubles, [j][k]) leaves as 16-byte pieces through an LDS tile
        const int nr = (int)nrhs;
        if (k < nr) {
#pragma unroll
          for (int j = 0; j < JM; ++j) ftile[sl * JM * KL + j * nr + k] = Fj[j];
        }
        lds_order();
        if (absorb && vb) {
          double *fr = F + (b * M + rowM(m)) * (int64_t)(JM * nr);
#pragma unroll
          for (int q = 0; q < JM / 2; ++q)
            if (2 * (q * KL + k) < JM * nr)
              *reinterpret_cast<double2 *>(fr + 2 * (q * KL + k)) = *reinterpret_cast<const double2 *>(&ftile[sl * JM * KL + 2 * (q * KL + k)]);
        }
      } else if (absorb && vk) {
        const int64_t mr = rowM(m);
        for (int j = 0; j < J; ++j) Fb[mr * J * nrhs + (int64_t)j * nrhs] = Fj[j];
      }
    }
    if (emit && vk) Zb[rowN(n) * nrhs] = xe + red;
    // the row requested five events ago goes into its slot (the row that slot held was consumed by the event that
    // requested it); then the stream this event came from moves up and the current times are read again
    rg[resolve.slot] = resolve.t;
    rgR[resolve.slot * KL + k] = resolve.r;
    rgX[resolve.slot * KL + k] = resolve.x;
    tlast = absorb ? tm : tlast;
    m += absorb ? 1 : 0;
    n += emit ? 1 : 0;
    lds_order();
    tm = rg[(int)(m & (RD - 1))]; tn = rg[RD + (int)(n & (RD - 1))];
  };
  while (__any(n < N)) {
    event(p0, p1);
    event(p1, p2);   // (an event of a finished series is a no-op: nothing absorbed, nothing emitted, its request parked)
    event(p2, p3);
    event(p3, p4);
    event(p4, p5);
    event(p5, p0);
  }
}

}  // namespace c2g

using namespace c2g;

// lower != 0: general_matmul_lower, else upper.  Returns C2_ERR_UNSUPPORTED for shapes the mapping does not cover.
extern "C" int c2_internal_generalK(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1,
                                    int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c, int64_t c_bs,
                                    const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z,
                                    c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int JM = J <= 8 ? 8 : (J <= 16 ? 16 : 32);
  int KL = 8;
  while (KL < 64 && KL < nrhs) KL *= 2;
  if (KL < JM) KL = JM;
  const dim3 grid((unsigned)((B + (kWave / KL) - 1) / (kWave / KL)), (unsigned)((nrhs + KL - 1) / KL));
#define C2_GK1(KL_, JM_, LO)                                                                                            \
  do {                                                                                                                  \
    if (F) hipLaunchKernelGGL((k_generalK<KL_, JM_, LO, true>), grid, dim3(kWave), 0, s, B, N, M, (int)J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F, zero_z);  \
    else hipLaunchKernelGGL((k_generalK<KL_, JM_, LO, false>), grid, dim3(kWave), 0, s, B, N, M, (int)J, nrhs, t1, t1_bs, t2, t2_bs, c, c_bs, U, V, Y, Z, F, zero_z); \
  } while (0)
#define C2_GK(KL_, JM_)                                   \
  do {                                                    \
    if (lower) C2_GK1(KL_, JM_, true);                    \
    else C2_GK1(KL_, JM_, false);                         \
  } while (0)
  switch (KL * 100 + JM) {
    case 808: C2_GK(8, 8); break;
    case 1608: C2_GK(16, 8); break;
    case 1616: C2_GK(16, 16); break;
    case 3208: C2_GK(32, 8); break;
    case 3216: C2_GK(32, 16); break;
    case 3232: C2_GK(32, 32); break;
    case 6408: C2_GK(64, 8); break;
    case 6416: C2_GK(64, 16); break;
    default: C2_GK(64, 32); break;
  }
#undef C2_GK
#undef C2_GK1
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}
