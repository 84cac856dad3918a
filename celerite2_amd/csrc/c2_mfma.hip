// c2_mfma.hip -- dot_tril / matmul_lower for ONE LONG series as dense fp64 contractions on the matrix cores
// (BASELINE configs[3]: N = 10^7, J = 16, nrhs = 32; the one shape north_star reserves MFMA for).
//
// matmul_lower (reference internal.hpp:135-145 with is_solve = false) is a linear recurrence with a diagonal transition,
//     F_n = p_n o (F_{n-1} + V_{n-1}^T y_{n-1}) ,   z_n += U_n F_n ,   p_n = exp(-c (t_n - t_{n-1})),
// so inside a block of T = 16 rows, in the frame of the block's first row (t_ref = t_{n0}):
//     Ut_n = U_n o exp(-c (t_n - t_ref))  (<= 1) ,   Vt_m = V_m o exp(+c (t_m - t_ref))  (>= 1, bounded by the block span)
//     Z_blk += strict_tril(Ut Vt^T) Y_blk + Ut H          H    = state of the rows before the block, at t_ref
//     H'     = exp(-c (t_ref' - t_ref)) o (H + Vt^T Y_blk)  H' = the same for the next block
// -- four products of 16 x 16 (x 16) tiles per 16 columns of right-hand sides: 28 v_mfma_f64_16x16x4_f64 per block at
// nrhs = 32.  The operand layouts chain without any data movement: the accumulator of P^T = Vt Ut^T (lane l: rows
// (l >> 4) + 4 r, column l & 15) IS the A operand of P Y chunk r, and the state accumulator H (same layout) IS the B
// operand of Ut H chunk r (tools/ubench/mfma64.hip checks the maps).  dot_tril (numpy.py:100-102) scales the rows of Y by
// sqrt(d) on load and starts the Z accumulators from them, so the separate scaling pass of the VALU path disappears.
//
// Time parallelism as in c2_scan.hip: the series is cut into chunks, (1) every chunk computes its outgoing state for a
// zero incoming one, (2) a short sequential pass folds the chunk states, (3) every chunk replays with its true
// incoming state and writes Z.  A block whose span would make exp(+c (t_m - t_ref)) overflow the dynamic range kept
// for it (c_max * span > kMaxGrow) is walked row by row on the VALU instead, in the same registers.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2m {
using namespace c2;

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int T = 16;              // rows per block = J
constexpr int J = 16;
constexpr double kMaxGrow = 300.0;  // exp(+300) ~ 2e130: products with O(1) data stay far inside the double range

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// carry[b][chunk][NT * 4][64]: dump of the state accumulators, lane-major
template <int NT, bool FINAL, bool SCALE>
__global__ __launch_bounds__(kWave) void k_mm_mfma(int64_t B, int64_t N, int64_t nrhs, int64_t Lc, int64_t nchunk,
                                                   const double *__restrict__ t, int64_t t_bs,
                                                   const double *__restrict__ c, int64_t c_bs,
                                                   const double *__restrict__ U, const double *__restrict__ V,
                                                   const double *__restrict__ dsc, const double *Y, double *Z,
                                                   double *__restrict__ carry, double *__restrict__ Dc, int zero_z) {
  const int l = threadIdx.x, i = l & 15, kq = l >> 4;
  const int64_t unit = blockIdx.x;
  const int64_t b = unit / nchunk, ch = unit % nchunk;
  const double *tb = t + b * t_bs, *Ub = U + b * N * J, *Vb = V + b * N * J, *db = SCALE ? dsc + b * N : nullptr;
  const double *Yb = Y + b * N * nrhs;
  double *Zb = Z + b * N * nrhs;
  double *cb = carry + ((size_t)(b * nchunk + ch) * NT * 4) * kWave + l;
  const int64_t s = ch * Lc, e = (s + Lc < N ? s + Lc : N);  // rows [s, e)
  // rates: c_A[q] = c[4 q + kq] goes with the A-layout of U / V (row i, column 4 q + kq); c_i = c[i] with the
  // transposed layout of V (row 4 r + kq, column i); the state rows are j = kq + 4 r, i.e. c_A again
  double cA[4], cmax = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) cA[q] = c[b * c_bs + 4 * q + kq];
  for (int j = 0; j < J; ++j) cmax = fmax(cmax, c[b * c_bs + j]);
  // Values that only depend on (state row j, block) or on (row of the block) are computed ONCE per quad of lanes and
  // broadcast inside the quad by DPP (lanes 4a .. 4a+3 of a 16-lane row share kq): lane (i, kq) owns j = 4 (i & 3) + kq
  // resp. row 4 (i & 3) + kq -- one exponential / square root per lane instead of four identical ones.
  const int qo = i & 3;
  const double cown = c[b * c_bs + 4 * qo + kq];
  // exp(+c_j (t_n - t_ref)) is needed in two layouts: row i, columns 4 q + kq (A operands) and rows 4 r + kq, column i (the
  // transposed V of the state update).  The 256 values exist once per block: computed in the first layout, transposed
  // through LDS (4 ds_write_b64 + 4 ds_read_b64 instead of four more exponentials and reciprocals per lane).
  __shared__ double Ex[T][J + 1];

  d4 H[NT];
#pragma unroll
  for (int h = 0; h < NT; ++h) {
    if (FINAL) { for (int r = 0; r < 4; ++r) H[h][r] = cb[(size_t)(h * 4 + r) * kWave]; }
    else H[h] = d4{0.0, 0.0, 0.0, 0.0};
  }

  // raw operands of one block, fetched ONE BLOCK AHEAD of their use (software pipeline: the loads of block k+1 are in
  // flight while block k runs its exponentials and matrix products)
  struct Raw {
    double u[4], v[4], vT[4], y[NT][4], dd, ti, tref, tnext, tlast;
  };
  auto fetch = [&](int64_t n0, Raw &R) {
    if (n0 >= N) n0 = N - 1;  // past the end: a harmless in-range block (never used)
    const int64_t ri = (n0 + i < N) ? n0 + i : N - 1;
    const int64_t nnext = (n0 + T < N) ? n0 + T : N - 1, nlast = (n0 + T <= N) ? n0 + T - 1 : N - 1;
    R.tref = tb[n0]; R.tnext = tb[nnext]; R.tlast = tb[nlast]; R.ti = tb[ri];
#pragma unroll
    for (int q = 0; q < 4; ++q) { R.u[q] = Ub[ri * J + 4 * q + kq]; R.v[q] = Vb[ri * J + 4 * q + kq]; }
    {
      const int64_t rown = (n0 + 4 * qo + kq < N) ? n0 + 4 * qo + kq : N - 1;
      R.dd = SCALE ? db[rown] : 1.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = (n0 + 4 * r + kq < N) ? n0 + 4 * r + kq : N - 1;
      R.vT[r] = Vb[row * J + i];
#pragma unroll
      for (int h = 0; h < NT; ++h) R.y[h][r] = Yb[row * nrhs + 16 * h + i];
    }
  };
  Raw cur, nxt;
  fetch(s, cur);
  for (int64_t n0 = s; n0 < e; n0 += T) {
    fetch(n0 + T < e ? n0 + T : n0, nxt);
    const double tref = cur.tref, tnext = cur.tnext, tlast = cur.tlast, ti = cur.ti;
    const bool oki = n0 + i < N;
    int64_t rr[4];
    double sc[4];
    bool okr[4];
    const double sown = SCALE ? sqrt(cur.dd) : 1.0;   // row 4 (i & 3) + kq of the block
    const double sq[4] = {dpp_mov<0x00>(sown), dpp_mov<0x55>(sown), dpp_mov<0xAA>(sown), dpp_mov<0xFF>(sown)};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      okr[r] = n0 + 4 * r + kq < N;
      rr[r] = okr[r] ? n0 + 4 * r + kq : N - 1;
      sc[r] = okr[r] ? sq[r] : 0.0;
    }
    // right-hand sides in B layout (row 4 r + kq, column 16 h + i), scaled; they are also the start of Z
    d4 Yv[NT];
#pragma unroll
    for (int h = 0; h < NT; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) Yv[h][r] = sc[r] * cur.y[h][r];

    if (cmax * (tlast - tref) <= kMaxGrow) {
      // ---- matrix-core path --------------------------------------------------------------------------------------
      double Ut[4], Vt[4], VtT[4];
      lds_order();   // (the previous block's readers of Ex are done)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double eA = exp_decay(-cA[q] * (ti - tref));     // exp(-c (t_n - t_ref)), row i, column 4 q + kq
        const double eI = rcp_nr(eA);
        Ut[q] = oki ? cur.u[q] * eA : 0.0;
        Vt[q] = oki ? cur.v[q] * eI : 0.0;
        Ex[i][4 * q + kq] = eI;
      }
      // frame change of the state rows j = kq + 4 q: one exponential per lane (its own j), the quad's four by DPP
      const double down = exp_decay(-cown * (tnext - tref));
      const double dec[4] = {dpp_mov<0x00>(down), dpp_mov<0x55>(down), dpp_mov<0xAA>(down), dpp_mov<0xFF>(down)};
      lds_order();
#pragma unroll
      for (int r = 0; r < 4; ++r) VtT[r] = okr[r] ? cur.vT[r] * Ex[4 * r + kq][i] : 0.0;
      // The matrix cores take ~106 cycles per v_mfma_f64_16x16x4 at best and ~196 from one MFMA to the next on the SAME
      // accumulator (tools/ubench/mfma64.hip): every product below runs as two or more independent chains.
      // P^T = Vt Ut^T : rows m = kq + 4 r, column n = i; keep n > m
      d4 PtA = mfma(Vt[0], Ut[0], d4{0.0, 0.0, 0.0, 0.0}), PtB = mfma(Vt[2], Ut[2], d4{0.0, 0.0, 0.0, 0.0});
      d4 ZA[NT];   // Ut H (reads the state BEFORE its update below)
      if (FINAL) {
#pragma unroll
        for (int h = 0; h < NT; ++h) ZA[h] = mfma(Ut[0], H[h][0], d4{0.0, 0.0, 0.0, 0.0});
      }
      PtA = mfma(Vt[1], Ut[1], PtA);
      PtB = mfma(Vt[3], Ut[3], PtB);
      if (FINAL) {
#pragma unroll
        for (int q = 1; q < 4; ++q)
#pragma unroll
          for (int h = 0; h < NT; ++h) ZA[h] = mfma(Ut[q], H[h][q], ZA[h]);
      }
      d4 Pt;
#pragma unroll
      for (int r = 0; r < 4; ++r) Pt[r] = (i > kq + 4 * r) ? PtA[r] + PtB[r] : 0.0;
      // H + Vt^T Y and strict_tril(Ut Vt^T) Y interleaved: 2 NT independent accumulators
      d4 ZB[NT];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < NT; ++h) {
          if (FINAL) ZB[h] = mfma(Pt[r], Yv[h][r], r == 0 ? d4{0.0, 0.0, 0.0, 0.0} : ZB[h]);
          H[h] = mfma(VtT[r], Yv[h][r], H[h]);
        }
      if (FINAL) {
#pragma unroll
        for (int h = 0; h < NT; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // rows of the accumulator are n = kq + 4 r: start from the caller's Z (matmul) or from Y sqrt(d) (dot_tril)
            const double z0 = SCALE ? Yv[h][r] : ((zero_z || !okr[r]) ? 0.0 : Zb[rr[r] * nrhs + 16 * h + i]);
            if (okr[r]) Zb[rr[r] * nrhs + 16 * h + i] = z0 + (ZA[h][r] + ZB[h][r]);
          }
      }
#pragma unroll
      for (int h = 0; h < NT; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) H[h][r] *= dec[r];
    } else {
      // ---- a block with a long gap: row by row on the VALU, same registers (state rows kq + 4 r, column 16 h + i) -
      double tprev = tref;
      for (int m = 0; m < T && n0 + m < N; ++m) {
        const int64_t n = n0 + m;
        const double tn = tb[n];
        const double scn = SCALE ? sqrt(db[n]) : 1.0;
        double un[4], vn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double p = exp_decay(-cA[r] * (tn - tprev));
          un[r] = Ub[n * J + kq + 4 * r];
          vn[r] = Vb[n * J + kq + 4 * r];
#pragma unroll
          for (int h = 0; h < NT; ++h) H[h][r] *= p;
        }
#pragma unroll
        for (int h = 0; h < NT; ++h) {
          const double yn = scn * Yb[n * nrhs + 16 * h + i];
          double part = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) part = fma(un[r], H[h][r], part);
          part += __shfl_xor(part, 16, kWave);
          part += __shfl_xor(part, 32, kWave);
          if (FINAL && kq == 0) {
            const double z0 = SCALE ? yn : (zero_z ? 0.0 : Zb[n * nrhs + 16 * h + i]);
            Zb[n * nrhs + 16 * h + i] = z0 + part;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) H[h][r] = fma(vn[r], yn, H[h][r]);
        }
        tprev = tn;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double p = exp_decay(-cA[r] * (tnext - tprev));
#pragma unroll
        for (int h = 0; h < NT; ++h) H[h][r] *= p;
      }
    }
    cur = nxt;
  }
  if (!FINAL) {
#pragma unroll
    for (int h = 0; h < NT; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) cb[(size_t)(h * 4 + r) * kWave] = H[h][r];
    // D_ch = exp(-c (t_{s(ch+1)} - t_{s(ch)})): the decay of an incoming state across the whole chunk
    if (l < J) Dc[(b * nchunk + ch) * J + l] = exp_decay(-c[b * c_bs + l] * (tb[e < N ? e : N - 1] - tb[s]));
  }
}

// Fold of the chunk states:  Hin[ch + 1] = D_ch o Hin[ch] + E_ch  (affine maps compose associatively), as a three-phase
// scan with coalesced accesses: the chunks are cut into R ranges; (1) every (range, state register) wavefront composes
// its chunks into one map, (2) one wavefront per state register walks the R range maps, (3) every (range, register)
// wavefront walks its chunks again from the range's true incoming state and replaces E_ch by Hin[ch] in place.
// carry[b][ch][hr][64] (lane-major dumps of the state accumulators), Dc[b][ch][16].
constexpr int kScanRanges = 64;
template <int PHASE>
__global__ __launch_bounds__(kWave) void k_mm_mfma_scan(int64_t nchunk, int NT4, int64_t m, double *__restrict__ carry,
                                                        const double *__restrict__ Dc, double *__restrict__ rs) {
  const int l = threadIdx.x, hr = blockIdx.y;
  const int64_t b = blockIdx.z, r = blockIdx.x;
  const int j = (l >> 4) + 4 * (hr & 3);
  const size_t cs = (size_t)NT4 * kWave;
  double *cb = carry + ((size_t)b * nchunk * NT4 + hr) * kWave + l;
  const double *Db = Dc + (size_t)b * nchunk * J + j;
  double *rsb = rs + (((size_t)b * kScanRanges) * NT4 + hr) * 3 * kWave + l;  // [range][hr][3][64]: Dtot, Etot, Hin
  const size_t rstr = (size_t)NT4 * 3 * kWave;
  if (PHASE == 2) {
    double h = 0.0;
    for (int q = 0; q < kScanRanges; ++q) {
      const double Dt = rsb[q * rstr], Et = rsb[q * rstr + kWave];
      rsb[q * rstr + 2 * kWave] = h;
      h = fma(Dt, h, Et);
    }
    return;
  }
  const int64_t q0 = r * m, q1 = (q0 + m < nchunk) ? q0 + m : nchunk;
  if (PHASE == 1) {
    double Dt = 1.0, Et = 0.0;
    for (int64_t q = q0; q < q1; ++q) {
      const double D = Db[q * J], E = cb[q * cs];
      Et = fma(D, Et, E);
      Dt *= D;
    }
    rsb[r * rstr] = Dt; rsb[r * rstr + kWave] = Et;
  } else {
    double h = rsb[r * rstr + 2 * kWave];
    for (int64_t q = q0; q < q1; ++q) {
      const double D = Db[q * J], E = cb[q * cs];
      cb[q * cs] = h;
      h = fma(D, h, E);
    }
  }
}

template <int NT, bool SCALE>
int run(int64_t B, int64_t N, int64_t nrhs, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
        const double *U, const double *V, const double *d, const double *Y, double *Z, int zero_z, hipStream_t s) {
  // chunks: enough wavefronts to fill the chip a few times over, whole blocks of 16 rows
  int64_t want = (4096 + B - 1) / B;
  int64_t Lc = ((N + want - 1) / want + T - 1) / T * T;
  if (Lc < 4 * T) Lc = 4 * T;
  const int64_t nchunk = (N + Lc - 1) / Lc;
  const size_t n_carry = (size_t)B * nchunk * NT * 4 * kWave, n_dc = (size_t)B * nchunk * J,
               n_rs = (size_t)B * kScanRanges * NT * 4 * 3 * kWave;
  const size_t bytes = sizeof(double) * (n_carry + n_dc + n_rs);
  double *carry = nullptr;
  bool async = true;
  if (c2::temp_alloc((void **)&carry, bytes, s) != hipSuccess) {
    (void)hipGetLastError();
    async = false;
    if (hipMalloc((void **)&carry, bytes) != hipSuccess) return C2_ERR_HIP;
  }
  const dim3 grid((unsigned)(B * nchunk));
  double *Dc = carry + n_carry, *rs = Dc + n_dc;
  hipLaunchKernelGGL((k_mm_mfma<NT, false, SCALE>), grid, dim3(kWave), 0, s, B, N, nrhs, Lc, nchunk, t, t_bs, c, c_bs, U,
                     V, d, Y, Z, carry, Dc, zero_z);
  const int64_t m = (nchunk + kScanRanges - 1) / kScanRanges;
  const dim3 g13(kScanRanges, NT * 4, (unsigned)B), g2(1, NT * 4, (unsigned)B);
  hipLaunchKernelGGL((k_mm_mfma_scan<1>), g13, dim3(kWave), 0, s, nchunk, NT * 4, m, carry, (const double *)Dc, rs);
  hipLaunchKernelGGL((k_mm_mfma_scan<2>), g2, dim3(kWave), 0, s, nchunk, NT * 4, m, carry, (const double *)Dc, rs);
  hipLaunchKernelGGL((k_mm_mfma_scan<3>), g13, dim3(kWave), 0, s, nchunk, NT * 4, m, carry, (const double *)Dc, rs);
  hipLaunchKernelGGL((k_mm_mfma<NT, true, SCALE>), grid, dim3(kWave), 0, s, B, N, nrhs, Lc, nchunk, t, t_bs, c, c_bs, U,
                     V, d, Y, Z, carry, Dc, zero_z);
  int rc = (hipGetLastError() == hipSuccess) ? C2_OK : C2_ERR_HIP;
  if (async) {
    if (hipFreeAsync(carry, s) != hipSuccess) rc = C2_ERR_HIP;
  } else {
    (void)hipStreamSynchronize(s);
    (void)hipFree(carry);
  }
  return rc;
}

}  // namespace c2m

// Matrix-core path of matmul_lower (d == NULL: Z (+)= tril(U V^T) Y) and dot_tril (d != NULL: Z = (I + tril(U V^T))
// (sqrt(d) o Y)) for J == 16 and nrhs in {16, 32, 64}.  Returns C2_ERR_UNSUPPORTED for other shapes: the caller then
// takes the VALU path.  Y == Z is allowed (a block reads all its rows of Y before it writes them).
extern "C" int c2_internal_matmul_lower_mfma(int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                                             const double *c, int64_t c_bs, const double *U, const double *V,
                                             const double *d, const double *Y, double *Z, int zero_z,
                                             c2_stream_t stream) {
  if (J != 16 || (nrhs != 16 && nrhs != 32 && nrhs != 64)) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  using namespace c2m;
#define C2M_RUN(NT)                                                                                        \
  return d ? run<NT, true>(B, N, nrhs, t, t_bs, c, c_bs, U, V, d, Y, Z, zero_z, s)                         \
           : run<NT, false>(B, N, nrhs, t, t_bs, c, c_bs, U, V, d, Y, Z, zero_z, s)
  switch (nrhs) {
    case 16: C2M_RUN(1);
    case 32: C2M_RUN(2);
    default: C2M_RUN(4);
  }
#undef C2M_RUN
}
