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
// ring five events later (the merge decides only at run time which stream advances).  The loop is software-pipelined
// (round 6): the front half of event e + 1 (decision, request, decay vector) is issued with the back half of event e
// (the multiply-adds on the state, the stores) -- profiles/r06_general.md.
//
// Workspace semantics as the reference: F[m, j * nrhs + k] (row-major), row m written when row m is absorbed, rows the
// merge never reaches left untouched, row 0 = V_0^T Y_0 (lower) / 0 (upper), row M-1 never written by the upper variant.
#include <cstdint>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2g {
using namespace c2;

// CH (round 6): a small batch with many right-hand sides is a handful of wavefronts on a chain of N + M events (64 series x 256
// columns: 256 wavefronts, 2.9 ms).  The state's propagator is DIAGONAL -- pure decay -- so the t2 grid is cut into chunks of Lc
// rows (blockIdx.z) and the same kernel runs twice: CH = 1 absorbs a chunk's rows from a zero state (no outputs) and leaves
// the sum in Gbuf; k_general_chain turns the sums into the state every chunk starts from (F_start(k + 1) = exp(-c dt_k) o
// F_start(k) + G_k); CH = 2 runs the event loop of chunk k from that state over the outputs whose last absorbed row lies in
// the chunk.  Which outputs those are is decided with the SAME comparisons the merge itself makes (a binary search on the
// absorb predicate), so ties fall where the sequential merge puts them.  No workspace F in this form.
// Gbuf: [series][chunk][JM][ncolp] (columns fastest).
template <int KL, int JM, bool LOWER, bool WF, int CH = 0>
__global__ __launch_bounds__(kWave) void k_generalK(int64_t B, int64_t N, int64_t M, int J, int64_t nrhs,
                                                    const double *__restrict__ t1, int64_t t1_bs,
                                                    const double *__restrict__ t2, int64_t t2_bs,
                                                    const double *__restrict__ c, int64_t c_bs,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    const double *__restrict__ Y, double *Z, double *F, int zero_z,
                                                    int64_t Lc = 0, int64_t K = 1, double *__restrict__ Gbuf = nullptr,
                                                    int64_t ncolp = 0) {
  static_assert(JM <= KL, "the lanes of a series also carry its width-J vectors");
  static_assert(CH == 0 || !WF, "no workspace in the chunked form");
  constexpr int SPW = kWave / KL;
  // decay vector, event row; two doubles of padding per vector: a series' pair is 16 (KL + 2) bytes from the next one's, so
  // the b128 broadcasts of the SPW series of a wavefront fall into distinct banks (unpadded, KL = 8: 128-byte stride, four
  // series per bank group -- 13 % of the kernel's cycles were LDS bank conflicts, rocprofv3 SQ_LDS_BANK_CONFLICT)
  __shared__ __attribute__((aligned(16))) double rowbuf[SPW][2][KL + 2];   // the decay vectors of two consecutive events (double buffer); [1]: also the prologue's row
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

  // chunk kc absorbs the positions [a0, a1) of the t2 grid (position 0 is the state the walk starts from)
  const int64_t kc = CH ? (int64_t)blockIdx.z : 0;
  const int64_t a0 = CH ? 1 + kc * Lc : 1, a1 = CH ? ((a0 + Lc < M) ? a0 + Lc : M) : M;
  double *gb = CH ? Gbuf + ((b * K + kc) * JM) * ncolp + (int64_t)blockIdx.y * KL + k : nullptr;   // element j at gb[j * ncolp]
  // state after the first t2 row (forward.hpp:297-300 / 358-361)
  double Fj[JM];
  if (CH == 0 || kc == 0) {
    const double y0 = Yb[rowM(0) * nrhs];
    rowbuf[sl][1][k] = actj ? Vb[rowM(0) * J] : 0.0;
    lds_order();
#pragma unroll
    for (int j = 0; j < JM; ++j) Fj[j] = rowbuf[sl][1][j] * y0;
    lds_order();
    if (WF && vk) {
      for (int j = 0; j < J; ++j) Fb[(LOWER ? 0 : 0) * J * nrhs + (int64_t)j * nrhs] = LOWER ? Fj[j] : 0.0;  // row 0
    }
  } else if (CH == 1) {
#pragma unroll
    for (int j = 0; j < JM; ++j) Fj[j] = 0.0;
  } else {
#pragma unroll
    for (int j = 0; j < JM; ++j) Fj[j] = gb[j * ncolp];   // the state after position a0 - 1 (k_general_chain)
  }
  const double tfirst = t2b[rowM(a0 - 1)];
  // lower: a row is absorbed in front of an output while t2[m] <= t1[n]; upper (walking down): while t2[m] > t1[n]
  auto absorbed = [&](double tm_, double tn_) { return LOWER ? tm_ <= tn_ : tm_ > tn_; };
  // the first output behind position pm of the t2 grid: outputs on the near side of it get nothing from it (forward.hpp:303-306 /
  // 364-367).  The predicate is monotone along the walk: a binary search (the first version walked the outputs one by one)
  auto first_output_behind = [&](int64_t pm) {
    const double tp = t2b[rowM(pm)];
    int64_t lo = 0, hi = N;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (absorbed(tp, t1b[rowN(mid)])) hi = mid; else lo = mid + 1;
    }
    return lo;
  };
  int64_t n = CH == 1 ? 0 : first_output_behind(a0 - 1);
  // the outputs of this chunk: those whose last absorbed row lies in it (the last row of a chunk opens the next chunk's)
  const int64_t n_end = (CH == 2 && kc + 1 < K) ? first_output_behind(a1 - 1) : N;
  int64_t m = a0;
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
  // The event loop is SOFTWARE-PIPELINED (round 6): what a series does next depends on its two grids only, so the FRONT half
  // of event e + 1 -- decide, request the row RD positions down, take the event's row (x, the width-J row) out of the ring,
  // the decay vector into rowbuf[q ^ 1] -- is issued together with the BACK half of event e -- the multiply-adds on the
  // state with the decay vector of rowbuf[q], the stores.  Neither half waits for the other inside an iteration (one fence
  // per iteration), so the exponential's chain of sixteen dependent operations runs under the state update of the event
  // before it instead of in front of its own: with ONE wavefront per SIMD (8192 series) nothing else hides it.  The only
  // chain left from event to event is compare -> select on the current and next times of both grids, which are kept in
  // registers one position ahead (tm1, tn1: positions m + 1, n + 1 are always in the ring -- a slot is refilled five
  // iterations after its row was taken, eight positions ahead).
  // (positions, lengths and row lengths as 32-bit integers from here on -- the launcher refuses grids of 2^31 rows: a 64-bit
  // product per address and 64-bit compares / selects per clamp were a fifth of the loop's instructions, and the loop is
  // bound by its instruction count: one wavefront per SIMD at 8192 series x 8 right-hand sides)
  int mi = (int)m, ni = (int)n;
  const int Ni = (int)N, Mi = (int)M, nr = (int)nrhs, a1i = (int)a1, nEnd = (int)n_end;   // (CH = 2: the chunk's outputs end at n_end; Ni, Mi also map positions to rows)
  double tm = rg[mi & (RD - 1)], tn = rg[RD + (ni & (RD - 1))];
  double tm1 = rg[(mi + 1) & (RD - 1)], tn1 = rg[RD + ((ni + 1) & (RD - 1))];
  struct Pend { double t, r, x; int slot; };
  struct Ev { bool absorb, emit; double xe; int row; double r[JM]; };
  Pend p0{0.0, 0.0, 0.0, 2 * RD}, p1 = p0, p2 = p0, p3 = p0, p4 = p0, p5 = p0;
  int q = 0;   // rowbuf[sl][q]: the decay vector of the event whose back half runs in this iteration
  auto front = [&](Pend &issue, Ev &ev) __attribute__((always_inline)) {
    // lower: absorb while t2[m] <= t1[n];  upper (walking down): absorb while t2[m] > t1[n];  CH = 1: the chunk's rows, no outputs
    const bool live = CH == 1 ? mi < a1i : ni < nEnd;
    const bool absorb = CH == 1 ? live : (live && mi < Mi && (LOWER ? tm <= tn : tm > tn));
    const bool emit = CH == 1 ? false : (live && !absorb);
    const int pos = absorb ? mi : ni;                        // position of the event's row along its stream
    const int len1 = (absorb ? Mi : Ni) - 1;
    const int so = (pos & (RD - 1)) + (absorb ? 0 : RD);     // the event's row in the ring
    {   // the request of this event: the row RD positions down the moving stream (clamped at the end of its grid)
      const int sreq = pos + RD < len1 ? pos + RD : len1;
      const int rreq = LOWER ? sreq : len1 - sreq;
      const double *pt = (absorb ? t2b : t1b) + rreq;
      const double *pr = (absorb ? Vb : Ub) + (int64_t)rreq * J;
      const double *px = (absorb ? Yb : (const double *)Zb) + (int64_t)rreq * nr;
      issue.t = *pt;
      const double rv = *pr;          // (every lane loads: lanes beyond the width read element 0 of the row and drop it)
      issue.r = actj ? rv : 0.0;
      issue.x = *px;
      issue.slot = live ? so : 2 * RD;
    }
    ev.absorb = absorb; ev.emit = emit;
    ev.row = LOWER ? pos : len1 - pos;
    ev.xe = rgX[so * KL + k];   // y_m / z_n
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      if constexpr (JM >= 2) {
        const double2 r2 = *reinterpret_cast<const double2 *>(&rgR[so * KL + j]);
        ev.r[j] = r2.x; ev.r[j + 1] = r2.y;
      } else {
        ev.r[0] = rgR[so * KL];
      }
    }
    const double tev = absorb ? tm : tn;
    rowbuf[sl][q ^ 1][k] = exp_decay(cj * (LOWER ? tlast - tev : tev - tlast));
    tlast = absorb ? tm : tlast;
    tm = absorb ? tm1 : tm;
    tn = emit ? tn1 : tn;
    mi += absorb ? 1 : 0;
    ni += emit ? 1 : 0;
    tm1 = rg[(mi + 1) & (RD - 1)]; tn1 = rg[RD + ((ni + 1) & (RD - 1))];
  };
  // The back half in two pieces: its decay vector is REQUESTED from LDS before the front half of the next event is issued
  // (LDS answers in order: behind the front half's reads of the ring the multiply-adds would wait for those as well) and
  // used after it.
  auto back_load = [&](double (&pv)[JM]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      if constexpr (JM >= 2) {
        const double2 p2 = *reinterpret_cast<const double2 *>(&rowbuf[sl][q][j]);
        pv[j] = p2.x; pv[j + 1] = p2.y;
      } else {
        pv[0] = rowbuf[sl][q][0];
      }
    }
  };
  auto back = [&](const Ev &ev, const double (&pv)[JM], Pend &resolve) __attribute__((always_inline)) {
    // (a scheduling barrier here -- so that the front half's instructions stay in front and cover the LDS latency of pv -- was
    // measured: 3.50 against 3.43 ms, the scheduler's own interleaving is the better one)
    double f0[JM];
#pragma unroll
    for (int j = 0; j < JM; ++j) f0[j] = pv[j] * Fj[j];
    // (the two kinds of event as predicated regions, not selects: a select of a double is two instructions, and there were
    // JM of them per event)
    if (ev.emit) {        // (U_n o p) . F   (forward.hpp:329 / 389); the state stays at the time of the last absorbed row
      double red = 0.0;   // (one chain in the order of the columns, as every version of this kernel summed it)
#pragma unroll
      for (int j = 0; j < JM; ++j) red = fma(ev.r[j], f0[j], red);
      if (vk) Zb[(int64_t)ev.row * nr] = ev.xe + red;
    }
    if (ev.absorb) {      // F = p o F + V_m^T y_m   (forward.hpp:320-323 / 380-383)
      // (written into the state's own registers: as plain C++ the compiler accumulates into f0 and copies it over, JM moves)
#pragma unroll
      for (int j = 0; j < JM; ++j) asm volatile("v_fma_f64 %0, %1, %2, %3" : "+v"(Fj[j]) : "v"(ev.r[j]), "v"(ev.xe), "v"(f0[j]));
    }
    if constexpr (WF) {
      if (dense_f) {   // the row (J x nrhs doubles, [j][k]) leaves as 16-byte pieces through an LDS tile
        if (k < nr) {
#pragma unroll
          for (int j = 0; j < JM; ++j) ftile[sl * JM * KL + j * nr + k] = Fj[j];
        }
        lds_order();
        if (ev.absorb && vb) {
          double *fr = F + (b * M + ev.row) * (int64_t)(JM * nr);
#pragma unroll
          for (int qq = 0; qq < JM / 2; ++qq)
            if (2 * (qq * KL + k) < JM * nr)
              *reinterpret_cast<double2 *>(fr + 2 * (qq * KL + k)) = *reinterpret_cast<const double2 *>(&ftile[sl * JM * KL + 2 * (qq * KL + k)]);
        }
      } else if (ev.absorb && vk) {
        for (int j = 0; j < J; ++j) Fb[(int64_t)ev.row * J * nrhs + (int64_t)j * nrhs] = Fj[j];
      }
    }
    // the row requested five iterations ago goes into its slot (the row that slot held was taken out by the front half that
    // requested it)
    rg[resolve.slot] = resolve.t;
    rgR[resolve.slot * KL + k] = resolve.r;
    rgX[resolve.slot * KL + k] = resolve.x;
    lds_order();
    q ^= 1;
  };
  Ev ea, eb;
  ea.absorb = false; ea.emit = false; ea.xe = 0.0; ea.row = 0;
#pragma unroll
  for (int j = 0; j < JM; ++j) ea.r[j] = 0.0;
  rowbuf[sl][0][k] = 1.0;   // (the decay vector of the empty event in front of the first one)
  lds_order();
  bool more = __any(CH == 1 ? mi < a1i : ni < nEnd);
  double pv[JM];
  while (more) {
    back_load(pv); front(p0, eb); back(ea, pv, p1);
    back_load(pv); front(p1, ea); back(eb, pv, p2);   // (an event of a finished series is a no-op: nothing absorbed, nothing emitted, its request parked)
    back_load(pv); front(p2, eb); back(ea, pv, p3);
    back_load(pv); front(p3, ea); back(eb, pv, p4);
    back_load(pv); front(p4, eb); back(ea, pv, p5);
    back_load(pv); front(p5, ea); back(eb, pv, p0);
    more = __any(CH == 1 ? mi < a1i : ni < nEnd);
  }
  {   // the back half of the last event
    Pend none{0.0, 0.0, 0.0, 2 * RD};
    back_load(pv);
    back(ea, pv, none);
  }
  if constexpr (CH == 1) {   // the chunk's sum, at the time of its last row
    if (vb) {
#pragma unroll
      for (int j = 0; j < JM; ++j) gb[j * ncolp] = Fj[j];
    }
  }
}

// Gbuf[chunk] <- the state the chunk starts from (it held the chunk's sum G_k): F_start(k + 1) = exp(-c |t_end(k) - t_end(k - 1)|) o
// F_start(k) + G_k, t_end(k) the time of the chunk's last row (t_end(-1): the first row of the grid).  A thread per (series, column).
template <int JM, bool LOWER>
__global__ __launch_bounds__(kWave) void k_general_chain(int64_t M, int J, int64_t Lc, int64_t K, const double *__restrict__ t2,
                                                         int64_t t2_bs, const double *__restrict__ c, int64_t c_bs,
                                                         double *__restrict__ Gbuf, int64_t ncolp) {
  const int64_t b = blockIdx.y, col = (int64_t)blockIdx.x * kWave + threadIdx.x;   // (ncolp is a multiple of 64)
  const double *t2b = t2 + b * t2_bs;
  auto rowM = [&](int64_t s) { return LOWER ? s : M - 1 - s; };
  double S[JM], cj[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) { S[j] = 0.0; cj[j] = j < J ? c[b * c_bs + j] : 0.0; }
  double tprev = t2b[rowM(0)];
  for (int64_t kc = 0; kc < K; ++kc) {
    const int64_t a0 = 1 + kc * Lc, a1 = (a0 + Lc < M) ? a0 + Lc : M;
    double *gb = Gbuf + ((b * K + kc) * JM) * ncolp + col;
    const double tend = t2b[rowM(a1 - 1)];
    const double dt = LOWER ? tprev - tend : tend - tprev;   // (the negative difference inside, as in the event loop)
    tprev = tend;
#pragma unroll
    for (int j = 0; j < JM; ++j) {
      const double g = gb[j * ncolp];
      if (kc > 0) gb[j * ncolp] = S[j];
      S[j] = fma(exp_decay(cj[j] * dt), S[j], g);
    }
  }
}

}  // namespace c2g

using namespace c2g;

// Many right-hand sides on a small batch, cut along time (CH above).  scratch: c2_internal_general_chunks_doubles doubles; Lc from
// c2_internal_general_chunks_plan.  Z accumulates as in the plain form; no workspace.
extern "C" int64_t c2_internal_general_chunks_plan(int64_t B, int64_t M, int64_t nrhs) {
  if (nrhs < 33 || M < 1024 || B > 65535) return 0;            // (64 lanes per series and tile)
  const int64_t waves = B * ((nrhs + 63) / 64);
  if (waves >= 1024) return 0;                                   // the chip is full without chunks
  int64_t K = 4096 / waves;                                      // ~ four wavefronts per SIMD
  if (K > (M - 1) / 128) K = (M - 1) / 128;                      // chunks of at least 128 rows
  if (K < 2) return 0;
  int64_t Lc = (M - 1 + K - 1) / K;
  return (Lc + 7) & ~(int64_t)7;
}
extern "C" size_t c2_internal_general_chunks_doubles(int64_t B, int64_t M, int64_t J, int64_t nrhs, int64_t Lc) {
  if (Lc < 1 || J > 32) return 0;
  const int64_t K = (M - 1 + Lc - 1) / Lc, JM = J <= 8 ? 8 : (J <= 16 ? 16 : 32), ncolp = ((nrhs + 63) / 64) * 64;
  return (size_t)(B * K * JM * ncolp);
}
extern "C" int c2_internal_general_chunks(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, int64_t Lc,
                                          const double *t1, int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c,
                                          int64_t c_bs, const double *U, const double *V, const double *Y, double *Z,
                                          double *scratch, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N > 2147483600 || M > 2147483600 || nrhs > 2147483600 || Lc < 1 || M < 2 || J > 32 || B > 65535) return C2_ERR_UNSUPPORTED;
  const int64_t K = (M - 1 + Lc - 1) / Lc, ntile = (nrhs + 63) / 64, ncolp = ntile * 64;
  if (K > 65535) return C2_ERR_UNSUPPORTED;
  const int JM = J <= 8 ? 8 : (J <= 16 ? 16 : 32);
  const dim3 grid((unsigned)B, (unsigned)ntile, (unsigned)K), gc((unsigned)ntile, (unsigned)B);
#define C2_GC(JM_, LO)                                                                                                         \
  do {                                                                                                                         \
    hipLaunchKernelGGL((k_generalK<64, JM_, LO, false, 1>), grid, dim3(kWave), 0, s, B, N, M, (int)J, nrhs, t1, t1_bs, t2, t2_bs, c, \
                       c_bs, U, V, Y, Z, (double *)nullptr, 0, Lc, K, scratch, ncolp);                                         \
    hipLaunchKernelGGL((k_general_chain<JM_, LO>), gc, dim3(kWave), 0, s, M, (int)J, Lc, K, t2, t2_bs, c, c_bs, scratch, ncolp);    \
    hipLaunchKernelGGL((k_generalK<64, JM_, LO, false, 2>), grid, dim3(kWave), 0, s, B, N, M, (int)J, nrhs, t1, t1_bs, t2, t2_bs, c, \
                       c_bs, U, V, Y, Z, (double *)nullptr, 0, Lc, K, scratch, ncolp);                                         \
  } while (0)
  if (JM == 8) { if (lower) C2_GC(8, true); else C2_GC(8, false); }
  else if (JM == 16) { if (lower) C2_GC(16, true); else C2_GC(16, false); }
  else { if (lower) C2_GC(32, true); else C2_GC(32, false); }
#undef C2_GC
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// lower != 0: general_matmul_lower, else upper.  Returns C2_ERR_UNSUPPORTED for shapes the mapping does not cover.
extern "C" int c2_internal_generalK(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1,
                                    int64_t t1_bs, const double *t2, int64_t t2_bs, const double *c, int64_t c_bs,
                                    const double *U, const double *V, const double *Y, double *Z, double *F, int zero_z,
                                    c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (N > 2147483600 || M > 2147483600 || nrhs > 2147483600) return C2_ERR_UNSUPPORTED;   // 32-bit positions in the event loop
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
