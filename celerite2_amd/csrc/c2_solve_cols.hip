// c2_solve_cols.hip -- solve_lower / solve_upper with MANY right-hand sides on a SMALL batch: apply_inverse on an N x M
// matrix (python/celerite2/core.py:56-60: Kinv_KxsT, the predictive variance / covariance of core.py:134-150; SURVEY.md
// section 8f-4), the reference's own use case being ONE light curve.
//
// Row by row (k_sweepK, c2_sweep.hip) such a call is a handful of wavefronts walking N dependent steps of ~0.26 us: one
// series of 4096 rows takes 1.05 ms with 64 or with 1024 right-hand sides (profiles/r04_large_nrhs.md).  The recursion
// (internal.hpp:135-145 / 177-188), written for the state G_n = F_n + A_n z_n (after row n is absorbed, before the decay
// into the next row),
//     F = P_n G_{n-1} ,   z_n = y_n - B_n . F ,   G_n = F + A_n z_n          (lower: A = W, B = U; upper: A = U, B = W)
// is AFFINE in G with a propagator (I - A_n B_n^T) P_n that does not depend on the right-hand side, so a chunk of rows maps
// the state it receives as G_end = Phi G_start + g, Phi (J x J) shared by every column, g one J-vector per column:
//   1. k_cols_walk<MODE 0>: every (series, chunk, tile of 64 columns) one wavefront, LANES OVER THE COLUMNS, zero start
//      state -> g; one more tile per (series, chunk) walks J "virtual columns" with y = 0 and start states e_j -> the
//      columns of Phi (the same code: linearity);
//   2. k_cols_chain: a thread per (series, column) applies the chunk maps one after the other -> the state every chunk
//      starts from (in place of g);
//   3. k_cols_walk<MODE 1>: the same walk from the true start states, writing z.
// Nothing is inverted and the maps are contractions (they carry decay factors <= 1), so there is nothing to verify; the
// result differs from the row-by-row kernel by rounding only.  Z may alias Y (pass 1 has read every row before pass 3
// writes).  No workspace F here: calls that ask for it keep their row-by-row / column-by-column forms.
#include <cstdint>
#include <type_traits>

#include "../../include/celerite2_amd.h"
#include "c2_common.hpp"
#include "c2_loglik_helpers.hpp"

namespace c2cols {
using namespace c2;

// step s of the walk <-> row n: lower 0 .. N-1, upper N-1 .. 0
template <bool LOWER>
__device__ __forceinline__ int64_t rowof(int64_t s, int64_t N) { return LOWER ? s : N - 1 - s; }

// Row records (round 6).  What a step needs of its row -- the decay vector p_n, the row fed into the state, the row applied to it -- is
// the same for every column, every tile of columns and both walks, and the walks are bound by instruction issue (~75 per step, 16 of
// them the exponential that all 64 lanes evaluate for the J that count).  k_cols_rows writes it ONCE per call as a record of 3 JM
// doubles per row, zero-padded from J to JM: [series][row][p | A | B]; a walk's step is then ONE load by 3 JM lanes, one LDS write and
// the broadcast reads -- no exponential, no t.  (The same records read through uniform addresses -- scalar loads straight into the
// multiply-adds' scalar operand, no LDS at all, 28 VALU instructions per step -- measured SLOWER: two scalar-cache round trips per
// step that eight wavefronts per SIMD do not cover, second walk 261 -> 341 us.)
template <int JM, bool LOWER>
__global__ __launch_bounds__(256) void k_cols_rows(int64_t N, int J, const double *__restrict__ t, int64_t t_bs,
                                                  const double *__restrict__ c, int64_t c_bs, const double *__restrict__ U,
                                                  const double *__restrict__ W, double *__restrict__ rec) {
  const int64_t b = blockIdx.y, idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = idx / JM;
  const int j = (int)(idx % JM);
  if (n >= N) return;
  const double *tb = t + b * t_bs;
  const double tn = tb[n];
  // the row the walk comes from: lower n - 1, upper n + 1; the first row of a walk has dt = 0, p = 1 exactly
  const double tp = LOWER ? (n > 0 ? tb[n - 1] : tn) : (n < N - 1 ? tb[n + 1] : tn);
  double *r = rec + (b * N + n) * (3 * JM);
  const bool actj = j < J;
  // exp(-c |dt|) with the negative difference inside (internal.hpp:139, 182)
  r[j] = actj ? exp_decay(c[b * c_bs + j] * (LOWER ? tp - tn : tn - tp)) : 0.0;
  r[JM + j] = actj ? (LOWER ? W : U)[(b * N + n) * J + j] : 0.0;       // fed into the state
  r[2 * JM + j] = actj ? (LOWER ? U : W)[(b * N + n) * J + j] : 0.0;   // applied to the state
}

// Gbuf: [series][chunk][JM][ncolp] (columns fastest: coalesced for the lanes of a walk and the threads of the chain)
// Phi : [series][chunk][JM][JM]    (Phi(i, j) at i * JM + j)
template <int JM, bool LOWER, int MODE>
__global__ __launch_bounds__(kWave) void k_cols_walk(int64_t N, int64_t nrhs, int64_t Lc, int64_t K, int ntile,
                                                     const double *__restrict__ rec, const double *Y, double *Z,
                                                     double *__restrict__ Phi, double *__restrict__ Gbuf, int64_t ncolp) {
  __shared__ __attribute__((aligned(16))) double rowbuf[2][3 * JM];   // [p | A | B] of two consecutive steps
  const int lane = threadIdx.x;
  const int64_t k = blockIdx.x, b = blockIdx.z;
  const int tile = blockIdx.y;
  const bool phi_tile = MODE == 0 && tile == ntile;   // the J virtual columns that give Phi
  int64_t col = (int64_t)tile * kWave + lane;
  const bool vcol = !phi_tile && col < nrhs;
  if (col >= nrhs) col = nrhs - 1;                    // clamped copies of the last column (their results are not kept)
  const bool rl = lane < 3 * JM;                      // this lane also moves element `lane` of the rows' records
  const double *rb = rec + b * N * (3 * JM) + (rl ? lane : 0);
  const double *yb = Y + b * N * nrhs + col;
  double *zb = Z + b * N * nrhs + col;
  const int64_t s_lo = k * Lc, s_hi = (s_lo + Lc < N) ? s_lo + Lc : N;
  double *gb = Gbuf + ((b * K + k) * JM) * ncolp + (int64_t)tile * kWave + lane;   // element j at gb[j * ncolp]

  double G[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) {
    if (MODE == 1) G[j] = gb[j * ncolp];
    else G[j] = (phi_tile && lane == j) ? 1.0 : 0.0;
  }
  // A walk is short (Lc rows) and there are many of them: what a step costs is the latency of its loads, so the rows are
  // requested R steps ahead into a register ring (one step ahead: 0.75 us a step, profiles/r04_large_nrhs.md)
  constexpr int R = 8;
  double rr[R], ry[R];
  auto load_row = [&](int r, int64_t s) {
    const int64_t nn = rowof<LOWER>(s < s_hi ? s : s_hi - 1, N);
    rr[r] = rb[nn * (3 * JM)];
    const double yv = yb[nn * nrhs];   // (loaded by the virtual columns too -- a valid, clamped column -- and dropped: no branch in the ring)
    ry[r] = phi_tile ? 0.0 : yv;
  };
#pragma unroll
  for (int r = 0; r < R; ++r) load_row(r, s_lo + r);
  int q = 0;
  auto step = [&](const int r, const int64_t s) __attribute__((always_inline)) {
    const int64_t n = rowof<LOWER>(s, N);
    const double rv = rr[r], yn = ry[r];
    load_row(r, s + R);
    if (rl) rowbuf[q][lane] = rv;
    lds_order();
    double red0 = 0.0, red1 = 0.0;   // (two partial sums: half the dependent chain)
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      const double2 p2 = *reinterpret_cast<const double2 *>(&rowbuf[q][j]);
      const double2 b2 = *reinterpret_cast<const double2 *>(&rowbuf[q][2 * JM + j]);
      G[j] *= p2.x;                  // F = P G: internal.hpp:143 / 186
      G[j + 1] *= p2.y;
      red0 = fma(b2.x, G[j], red0);
      red1 = fma(b2.y, G[j + 1], red1);
    }
    const double zn = yn - (red0 + red1);      // internal.hpp:144 / 187
    if (MODE == 1 && vcol) zb[n * nrhs] = zn;
#pragma unroll
    for (int j = 0; j < JM; j += 2) {
      const double2 a2 = *reinterpret_cast<const double2 *>(&rowbuf[q][JM + j]);
      G[j] = fma(a2.x, zn, G[j]);    // internal.hpp:140 / 183 (the state the next row decays)
      G[j + 1] = fma(a2.y, zn, G[j + 1]);
    }
    q ^= 1;
    // (MODE 0 has no store in its step, and without one the optimiser lets the state's chain trail the R unrolled steps' loads and LDS
    // reads: the first version took 442 registers there -- ONE wavefront per SIMD, the first walk 2.2 x the second (tools/kernel_regs.py
    // lists such kernels).  An empty asm that 'uses and redefines' the state pins every step where it is written.)
#pragma unroll
    for (int j = 0; j < JM; ++j) asm volatile("" : "+v"(G[j]));
  };
  int64_t s0 = s_lo;
  for (; s0 + R <= s_hi; s0 += R) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, s0 + r);
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (s0 + r < s_hi) step(r, s0 + r);
  if (MODE == 0) {
    if (phi_tile) {
      if (lane < JM) {
        double *ph = Phi + (b * K + k) * (JM * JM);
#pragma unroll
        for (int i = 0; i < JM; ++i) ph[i * JM + lane] = G[i];   // column `lane` of Phi (rows and columns beyond J: zero)
      }
    } else {
#pragma unroll
      for (int j = 0; j < JM; ++j) gb[j * ncolp] = G[j];
    }
  }
}

// Gbuf[chunk] <- the state the chunk starts from (it held g of the chunk): G_{k+1} = Phi_k G_k + g_k, G_0 = 0.
// A thread per (series, column) walks the K chunks one after the other, so what an iteration costs is what it WAITS for.
// Round 6: the chunk maps of the series are staged in LDS KP chunks at a time (they are the same for every column: the first
// version read them through the scalar cache, three round trips per chunk: ~1.3 us a chunk, 43 - 66 us for the 32 - 64
// chunks of a 4096-row series) and g is requested PF chunks ahead into a register ring.
template <int JM>
__global__ __launch_bounds__(kWave) void k_cols_chain(int64_t K, const double *__restrict__ Phi, double *Gbuf, int64_t ncolp) {
  constexpr int KP = JM <= 8 ? 64 : 16;   // 32 KB of maps per piece
  constexpr int PF = 4;
  __shared__ __attribute__((aligned(16))) double ph[KP * JM * JM];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y, col = (int64_t)blockIdx.x * kWave + lane;   // (ncolp is a multiple of 64)
  double G[JM], gq[PF][JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) G[j] = 0.0;
  auto gload = [&](auto slot_tag, int64_t k) __attribute__((always_inline)) {
    constexpr int Q = decltype(slot_tag)::value;
    const int64_t kc = k < K ? k : K - 1;
#pragma unroll
    for (int i = 0; i < JM; ++i) gq[Q][i] = Gbuf[((b * K + kc) * JM + i) * ncolp + col];
  };
  gload(std::integral_constant<int, 0>{}, 0); gload(std::integral_constant<int, 1>{}, 1);
  gload(std::integral_constant<int, 2>{}, 2); gload(std::integral_constant<int, 3>{}, 3);
  auto chunk = [&](auto slot_tag, int64_t k, int kl) __attribute__((always_inline)) {
    constexpr int Q = decltype(slot_tag)::value;
    double *gb = Gbuf + ((b * K + k) * JM) * ncolp + col;
#pragma unroll
    for (int i = 0; i < JM; ++i) gb[i * ncolp] = G[i];   // (g of this chunk was read PF chunks ago)
    const double *pm = ph + kl * (JM * JM);
    double gn[JM];
#pragma unroll
    for (int i = 0; i < JM; ++i) {
      double v = gq[Q][i];
#pragma unroll
      for (int j = 0; j < JM; j += 2) {
        const double2 p2 = *reinterpret_cast<const double2 *>(pm + i * JM + j);
        v = fma(p2.x, G[j], v);
        v = fma(p2.y, G[j + 1], v);
      }
      gn[i] = v;
    }
#pragma unroll
    for (int i = 0; i < JM; ++i) G[i] = gn[i];
    gload(slot_tag, k + PF);
  };
  for (int64_t k0 = 0; k0 < K; k0 += KP) {
    const int kn = (int)((K - k0) < KP ? (K - k0) : KP);
    lds_order();
    for (int idx = lane; idx < kn * JM * JM; idx += kWave) ph[idx] = Phi[(b * K + k0) * (JM * JM) + idx];
    lds_order();
    for (int kl = 0; kl < kn; kl += PF) {   // (K, KP multiples of nothing in particular: the tail is guarded)
      chunk(std::integral_constant<int, 0>{}, k0 + kl, kl);
      if (kl + 1 < kn) chunk(std::integral_constant<int, 1>{}, k0 + kl + 1, kl + 1);
      if (kl + 2 < kn) chunk(std::integral_constant<int, 2>{}, k0 + kl + 2, kl + 2);
      if (kl + 3 < kn) chunk(std::integral_constant<int, 3>{}, k0 + kl + 3, kl + 3);
    }
  }
}

// The first version of the chain: the maps through the scalar cache, g one chunk ahead.  Kept for width 16, where a chunk's map
// is 2 KB and the LDS version (16 chunks a piece, 128 b128 reads per chunk) measured 2.5 x slower (336 against ~130 us at 64 x 4096 x 64).
template <int JM>
__global__ __launch_bounds__(kWave) void k_cols_chain_s(int64_t K, const double *__restrict__ Phi, double *__restrict__ Gbuf,
                                                      int64_t ncolp) {
  const int64_t b = blockIdx.y, col = (int64_t)blockIdx.x * kWave + threadIdx.x;   // (ncolp is a multiple of 64)
  double G[JM], g[JM], gn[JM];
#pragma unroll
  for (int j = 0; j < JM; ++j) { G[j] = 0.0; g[j] = Gbuf[((b * K) * JM + j) * ncolp + col]; }
  for (int64_t k = 0; k < K; ++k) {
    double *gb = Gbuf + ((b * K + k) * JM) * ncolp + col;
    const double *ph = Phi + (b * K + k) * (JM * JM);
    const int64_t kn = k + 1 < K ? k + 1 : k;   // the next chunk's g is requested before this chunk's products
#pragma unroll
    for (int i = 0; i < JM; ++i) gn[i] = Gbuf[((b * K + kn) * JM + i) * ncolp + col];
#pragma unroll
    for (int i = 0; i < JM; ++i) gb[i * ncolp] = G[i];
#pragma unroll
    for (int i = 0; i < JM; ++i) {
      double v = g[i];
#pragma unroll
      for (int j = 0; j < JM; ++j) v = fma(ph[i * JM + j], G[j], v);
      g[i] = v;
    }
#pragma unroll
    for (int i = 0; i < JM; ++i) { G[i] = g[i]; g[i] = gn[i]; }
  }
}

struct Plan {
  int JM;
  int64_t Lc, K, ncolp;
  int ntile;
  size_t phi, gbuf, rec, total;   // doubles
};
static Plan plan(int64_t B, int64_t N, int64_t J, int64_t nrhs) {
  Plan p;
  p.JM = J <= 8 ? 8 : 16;
  p.ntile = (int)((nrhs + kWave - 1) / kWave);
  p.ncolp = (int64_t)p.ntile * kWave;
  // chunks of 64 rows while the launch stays within a few wavefronts per SIMD; longer ones beyond (the walks are what
  // costs: one step ~0.3 us; the chain costs ~0.2 us a chunk)
  p.Lc = 64;
  while (p.Lc < 1024 && B * ((N + p.Lc - 1) / p.Lc) * p.ntile > 8192) p.Lc *= 2;
  p.K = (N + p.Lc - 1) / p.Lc;
  p.phi = 0;
  p.gbuf = (size_t)B * p.K * p.JM * p.JM;
  p.rec = p.gbuf + (size_t)B * p.K * p.JM * p.ncolp;
  p.total = p.rec + (size_t)B * N * 3 * p.JM;
  return p;
}

}  // namespace c2cols

using namespace c2cols;

extern "C" {

// scratch of c2_internal_solve_cols, in doubles (0: shape not supported)
size_t c2_internal_solve_cols_doubles(int64_t B, int64_t N, int64_t J, int64_t nrhs) {
  if (B < 1 || N < 2 || J < 1 || J > 16 || nrhs < 1) return 0;
  return plan(B, N, J, nrhs).total;
}

// solve_lower (lower != 0) / solve_upper with nrhs columns of row-major Y, Z (B, N, nrhs); W is the factor's W.  Z may
// alias Y.  scratch: c2_internal_solve_cols_doubles(B, N, J, nrhs) doubles.
int c2_internal_solve_cols(int lower, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                           const double *c, int64_t c_bs, const double *U, const double *W, const double *Y, double *Z,
                           double *scratch, c2_stream_t stream) {
  if (B < 1 || N < 2 || J < 1 || J > 16 || nrhs < 1 || B > 65535) return C2_ERR_UNSUPPORTED;
  const Plan p = plan(B, N, J, nrhs);
  hipStream_t s = (hipStream_t)stream;
  double *Phi = scratch + p.phi, *Gbuf = scratch + p.gbuf;
  const dim3 g0((unsigned)p.K, (unsigned)(p.ntile + 1), (unsigned)B), g1((unsigned)p.K, (unsigned)p.ntile, (unsigned)B);
  const dim3 gc((unsigned)p.ntile, (unsigned)B);
  double *rec = scratch + p.rec;
  const dim3 gr((unsigned)((N * p.JM + 255) / 256), (unsigned)B);
#define C2_COLS(JM_, LO)                                                                                              \
  do {                                                                                                                \
    hipLaunchKernelGGL((k_cols_rows<JM_, LO>), gr, dim3(256), 0, s, N, (int)J, t, t_bs, c, c_bs, U, W, rec);          \
    hipLaunchKernelGGL((k_cols_walk<JM_, LO, 0>), g0, dim3(kWave), 0, s, N, nrhs, p.Lc, p.K, p.ntile, (const double *)rec, \
                       Y, Z, Phi, Gbuf, p.ncolp);                                                                      \
    if (JM_ <= 8) hipLaunchKernelGGL((k_cols_chain<JM_>), gc, dim3(kWave), 0, s, p.K, (const double *)Phi, Gbuf, p.ncolp); \
    else hipLaunchKernelGGL((k_cols_chain_s<JM_>), gc, dim3(kWave), 0, s, p.K, (const double *)Phi, Gbuf, p.ncolp);      \
    hipLaunchKernelGGL((k_cols_walk<JM_, LO, 1>), g1, dim3(kWave), 0, s, N, nrhs, p.Lc, p.K, p.ntile, (const double *)rec, \
                       Y, Z, Phi, Gbuf, p.ncolp);                                                                      \
  } while (0)
  if (p.JM == 8) { if (lower) C2_COLS(8, true); else C2_COLS(8, false); }
  else { if (lower) C2_COLS(16, true); else C2_COLS(16, false); }
#undef C2_COLS
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

}  // extern "C"
