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
#include <cstdint>

#include "c2_loglik_helpers.hpp"
#include "c2_rscatter.hpp"
#include "../../include/celerite2_amd.h"

namespace c2r {
using namespace c2;

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
  constexpr int R = 2;  // rows requested ahead
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
        load_step(r, s - R);
        const double p = exp_decay(cj * dt);
        rowbuf[q][sl][0][k] = p; rowbuf[q][sl][1][k] = bn; rowbuf[q][sl][2][k] = am;
        lds_order();
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
        // cotangent of row m: solves fold it into the running bZ, products write it out
        if (SOLVE) {
          bzrun = bzin_m + acc;
          if (vb && vk) bYb[m * nrhs] = bzrun;
        } else {
          bzrun = bzin_m;  // row m is the row of the next step
          if (vb && vk) bYb[m * nrhs] = acc;
        }
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
      }
    }
  }
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

}  // namespace c2r

using namespace c2r;

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
  if (hipMallocAsync(&tmp, (nws + npart) * sizeof(double), s) != hipSuccess) {
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
