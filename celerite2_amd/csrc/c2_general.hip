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
// vector and the row that goes with the event (V_m or U_n) are computed / held by lanes 0..J-1 and broadcast through
// LDS.  The next row of BOTH streams is always in registers, and a touch load eight rows further down each stream keeps
// the lines coming (the merge decides only at run time which stream advances).
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
  __shared__ __attribute__((aligned(16))) double rowbuf[SPW][2][KL + 2];
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
  // Three rows of either stream in registers: the current one and the two behind it.  Which stream moves is decided at
  // the TOP of an event (it only takes the two current times), so the row three positions down the moving stream is
  // requested there -- into one of two pending sets, alternating -- and lands in the stream's last slot at the END of the
  // NEXT event: nearly two events (~0.5 us) between a request and its first use, where the first version asked for a row
  // at the end of an event and selected it into place at once (one full L2 / HBM latency per event: 0.9 us, 7.0 ms per
  // 8192 series of 4096 + 4096 rows with 8 right-hand sides).  Every event issues the same five loads, so the compiler
  // can count them (vmcnt) instead of draining the queue.
  double tn = t1b[clampN(n)], un = actj ? Ub[clampN(n) * J] : 0.0, zn = Zb[clampN(n) * nrhs];
  double tm = t2b[clampM(m)], vm = actj ? Vb[clampM(m) * J] : 0.0, ym = Yb[clampM(m) * nrhs];
  double tn2 = t1b[clampN(n + 1)], un2 = actj ? Ub[clampN(n + 1) * J] : 0.0, zn2 = Zb[clampN(n + 1) * nrhs];
  double tm2 = t2b[clampM(m + 1)], vm2 = actj ? Vb[clampM(m + 1) * J] : 0.0, ym2 = Yb[clampM(m + 1) * nrhs];
  double tn3 = t1b[clampN(n + 2)], un3 = actj ? Ub[clampN(n + 2) * J] : 0.0, zn3 = Zb[clampN(n + 2) * nrhs];
  double tm3 = t2b[clampM(m + 2)], vm3 = actj ? Vb[clampM(m + 2) * J] : 0.0, ym3 = Yb[clampM(m + 2) * nrhs];
  struct Pend { double t, r, x; int tag; };   // tag: 0 nothing, 1 a t2 row, 2 a t1 row
  Pend pa{0.0, 0.0, 0.0, 0}, pb{0.0, 0.0, 0.0, 0};

  // (the touch loads that keep the lines coming, eight rows down the moving stream, are only there for their side effect;
  // so that nobody waits for them their values are summed into `sink` FOUR events after they were requested -- the loop
  // is unrolled four times over four pairs of registers)
  double sink = 0.0;
  struct Touch { double r, x; };
  Touch th0{0.0, 0.0}, th1{0.0, 0.0}, th2{0.0, 0.0}, th3{0.0, 0.0};
  auto event = [&](Pend &issue, Pend &resolve, Touch &tch) __attribute__((always_inline)) {
    const bool live = n < N;
    // lower: absorb while t2[m] <= t1[n];  upper (walking down): absorb while t2[m] > t1[n]
    const bool absorb = live && m < M && (LOWER ? tm <= tn : tm > tn);
    const bool emit = live && !absorb;
    {   // the requests of this event: the row three positions down the moving stream, a touch eight rows down
      const int64_t rm = clampM(m + 3), rn = clampN(n + 3);
      const double *pt = absorb ? t2b + rm : t1b + rn;
      const double *pr = absorb ? Vb + rm * J : Ub + rn * J;
      const double *px = absorb ? Yb + rm * nrhs : (const double *)Zb + rn * nrhs;
      issue.t = *pt; issue.r = actj ? *pr : 0.0; issue.x = *px;
      issue.tag = absorb ? 1 : (emit ? 2 : 0);
      sink += tch.r + tch.x;
      const int64_t fm = clampM(m + 8), fn = clampN(n + 8);
      const double *qr = absorb ? Vb + fm * J : Ub + fn * J;
      const double *qx = absorb ? Yb + fm * nrhs : (const double *)Zb + fn * nrhs;
      tch.r = *qr; tch.x = *qx;
    }
    const double tev = absorb ? tm : tn;
    const double p = exp_decay(cj * (LOWER ? tlast - tev : tev - tlast));
    rowbuf[sl][0][k] = p;
    rowbuf[sl][1][k] = absorb ? vm : un;
    lds_order();
    double red = 0.0;
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      double2 p2, r2;
      if constexpr (JM >= 2) {
        p2 = *reinterpret_cast<const double2 *>(&rowbuf[sl][0][j]);
        r2 = *reinterpret_cast<const double2 *>(&rowbuf[sl][1][j]);
      } else {
        p2 = make_double2(rowbuf[sl][0][0], 0.0); r2 = make_double2(rowbuf[sl][1][0], 0.0);
      }
      const double f0 = p2.x * Fj[j];
      red = fma(r2.x, f0, red);                       // emit: (U_n o p) . F          (forward.hpp:329 / 389)
      Fj[j] = absorb ? fma(r2.x, ym, f0) : Fj[j];      // absorb: F = p o F + V_m^T y_m (forward.hpp:320-323 / 380-383)
      if (j + 1 < JM) {
        const double f1 = p2.y * Fj[j + 1];
        red = fma(r2.y, f1, red);
        Fj[j + 1] = absorb ? fma(r2.y, ym, f1) : Fj[j + 1];
      }
    }
    lds_order();
    if (WF && absorb && vk) {
      const int64_t mr = rowM(m);
      for (int j = 0; j < J; ++j) Fb[mr * J * nrhs + (int64_t)j * nrhs] = Fj[j];
    }
    if (emit && vk) Zb[rowN(n) * nrhs] = zn + red;
    // (1) the row requested by the PREVIOUS event arrives in the last slot of its stream ...
    const bool rm_ = resolve.tag == 1, rn_ = resolve.tag == 2;
    tm3 = rm_ ? resolve.t : tm3; vm3 = rm_ ? resolve.r : vm3; ym3 = rm_ ? resolve.x : ym3;
    tn3 = rn_ ? resolve.t : tn3; un3 = rn_ ? resolve.r : un3; zn3 = rn_ ? resolve.x : zn3;
    // (2) ... then the stream this event came from moves up (its last slot is stale until this event's request lands)
    tlast = absorb ? tm : tlast;
    m += absorb ? 1 : 0;
    n += emit ? 1 : 0;
    tm = absorb ? tm2 : tm; vm = absorb ? vm2 : vm; ym = absorb ? ym2 : ym;
    tm2 = absorb ? tm3 : tm2; vm2 = absorb ? vm3 : vm2; ym2 = absorb ? ym3 : ym2;
    tn = emit ? tn2 : tn; un = emit ? un2 : un; zn = emit ? zn2 : zn;
    tn2 = emit ? tn3 : tn2; un2 = emit ? un3 : un2; zn2 = emit ? zn3 : zn2;
  };
  while (__any(n < N)) {
    event(pa, pb, th0);
    event(pb, pa, th1);   // (an event of a finished series is a no-op: nothing absorbed, nothing emitted, rows clamped)
    event(pa, pb, th2);
    event(pb, pa, th3);
  }
  if (sink == 1.2345678e300) Zb[0] = sink;  // keeps the touch loads alive; never true for finite data
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
