// c2_loglik_t.hip -- fused log-likelihood (+ reverse-mode gradient) with ONE LANE PER SERIES, for batches large
// enough to fill the chip that way (>= 32768 series: 512+ wavefronts).  J == 8.
//
// Why a second mapping.  The group-of-8-lanes kernels (c2_loglik.hip) spend more than half of their issued VALU work
// on moving width-J vectors between the lanes of a group (DPP gathers, butterflies, scalars replicated 8x): ~300
// wave-instructions per series-group step in the reverse sweep, ~140 of them arithmetic.  With the whole J x J state of
// a series in ONE lane's registers (S and M symmetric-packed: 36 doubles each) every contraction is in-lane, nothing is
// gathered, reduced or replicated: ~5.5 wave-instructions per series-step forward, ~11 reverse, against ~19 / ~37.
// The price is (1) memory access: a lane streams its own series, so rows are moved between HBM and the lanes through
// LDS transposes (coalesced 128-byte runs on the HBM side, one row per lane on the register side), and (2) state: a
// lane cannot keep C replayed J x J states, so the reverse sweep does NOT replay -- it runs the recursion BACKWARD,
//     S_{n-1} = P_n^-1 S_n P_n^-1 - d_{n-1} w_{n-1}^T w_{n-1} ,   F_{n-1} = P_n^-1 F_n - w_{n-1} z_{n-1}
// (the inverse of forward.hpp:115-123 / internal.hpp:140-143), re-anchored every C rows on a checkpoint the forward
// pass wrote.  Errors of the backward recursion grow like exp(2 c (t_end - t_start)) inside a segment, so this path is
// taken only when max_j c_j * (segment span) <= kGuard for every segment of every series (checked on the device by the
// forward pass, stream-ordered: no host round trip); otherwise the replay kernels of c2_loglik.hip run.  W_n is
// recorded by the forward pass (the backward recursion cannot re-derive it), in a lane-major private layout.
//
// Reference steps: factor forward.hpp:105-134, solve_lower internal.hpp:135-145, solve_lower_rev internal.hpp:225-245,
// factor_rev reverse.hpp:52-84; the fused reverse step is the one derived in c2_loglik.hip.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

namespace c2t {
using namespace c2;

constexpr int J = 8;
constexpr int NS = J * (J + 1) / 2;  // packed symmetric J x J
constexpr int C = 16;                // checkpoint interval (rows)
constexpr double kGuard = 2.0;       // largest allowed max_j c_j * (t_end - t_start) of a segment
constexpr int RT = 2;                // rows per tile of the width-J streams (128-byte runs in HBM)
constexpr int ST = 8;                // rows per tile of the per-series scalar streams (64-byte runs)
constexpr int RSTR = RT * J + 2;     // LDS stride (doubles) of a series in a row tile: 144 B, conflict-free b128
constexpr int SSTR = ST + 1;         // LDS stride (doubles) of a series in a scalar tile: 72 B, conflict-free b64

__host__ __device__ constexpr int sidx(int i, int j) {  // packed index of S(i,j), i <= j
  return i * J - i * (i - 1) / 2 + (j - i);
}
__host__ __device__ constexpr int sym(int i, int j) { return i <= j ? sidx(i, j) : sidx(j, i); }

// ---- records private to the fwd/rev pair, lane-major: every access is one contiguous run per wavefront ---------------
//   W   : [wave][n][J/2][64] double2      (row n of W, two columns per 16-byte piece)
//   DZ  : [wave][n][64] double2           ((d_n, z_n))
//   CK  : [wave][k][NS + J][64] double    (state after row n_k: S packed, F)
struct Rec {
  size_t w, dz, ck, total;  // offsets / total in doubles
  int64_t nck;              // checkpoints per series
};
__host__ __device__ inline int64_t n_ckpt(int64_t N) { return (N - 1) / C + 1; }  // rows C, 2C, ... and row N-1
__host__ inline Rec rec_layout(int64_t B, int64_t N) {
  const size_t waves = ((size_t)B + kWave - 1) / kWave;
  Rec r;
  r.nck = n_ckpt(N);
  r.w = 0;
  r.dz = r.w + waves * (size_t)N * J * kWave;
  r.ck = r.dz + waves * (size_t)N * 2 * kWave;
  r.total = r.ck + waves * (size_t)r.nck * (NS + J) * kWave;
  return r;
}

// ---- tile movers -------------------------------------------------------------------------------------------------------
// A wavefront owns series b0 .. b0+63 (clamped to B-1).  Row tile: RT rows of J doubles of every series.  One global
// instruction moves 8 series x 128 bytes (lane l: series 8 i + l / 8, 16-byte piece l % 8).
struct RowIO {
  int sl[8];       // clamped series (within the wavefront) this lane serves in instruction i
  int piece;       // l % 8
  __device__ __forceinline__ RowIO(int lane, int last) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int s = 8 * i + lane / 8;
      sl[i] = s < last ? s : last;
    }
    piece = lane & 7;
  }
};

// global -> registers: rows n0, n0+1 (clamped to [0, N-1]) of every series
__device__ __forceinline__ void row_fetch(const double *__restrict__ base, int64_t N, int64_t n0, const RowIO &io,
                                          double2 (&st)[8]) {
  int64_t r = n0 + (io.piece >> 2);
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
  const int64_t off = r * J + 2 * (io.piece & 3);
#pragma unroll
  for (int i = 0; i < 8; ++i) st[i] = *reinterpret_cast<const double2 *>(base + (int64_t)io.sl[i] * N * J + off);
}
// registers -> LDS tile
__device__ __forceinline__ void row_stage(double *tile, int lane, const double2 (&st)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
    *reinterpret_cast<double2 *>(tile + (8 * i + lane / 8) * RSTR + 2 * (lane & 7)) = st[i];
}
// LDS tile -> this lane's row r
__device__ __forceinline__ void row_read(const double *tile, int lane, int r, double (&x)[J]) {
#pragma unroll
  for (int q = 0; q < J / 2; ++q) {
    const double2 v = *reinterpret_cast<const double2 *>(tile + lane * RSTR + r * J + 2 * q);
    x[2 * q] = v.x; x[2 * q + 1] = v.y;
  }
}
__device__ __forceinline__ void row_write(double *tile, int lane, int r, const double (&x)[J]) {
#pragma unroll
  for (int q = 0; q < J / 2; ++q)
    *reinterpret_cast<double2 *>(tile + lane * RSTR + r * J + 2 * q) = make_double2(x[2 * q], x[2 * q + 1]);
}
// LDS tile -> global: rows n0, n0+1 of every series; rows outside [lo, hi] and series beyond `last` are skipped
__device__ __forceinline__ void row_flush(double *__restrict__ base, int64_t N, int64_t n0, int64_t lo, int64_t hi,
                                          const double *tile, int lane, int last) {
  const int piece = lane & 7;
  const int64_t r = n0 + (piece >> 2);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int s = 8 * i + lane / 8;
    const double2 v = *reinterpret_cast<const double2 *>(tile + s * RSTR + 2 * piece);
    if (s <= last && r >= lo && r <= hi)
      *reinterpret_cast<double2 *>(base + (int64_t)s * N * J + r * J + 2 * (piece & 3)) = v;
  }
}

// Scalar tile: ST rows of one double of every series; one instruction moves 8 series x 64 bytes.
__device__ __forceinline__ void sc_fetch(const double *__restrict__ base, int64_t N, int64_t n0, const RowIO &io,
                                         double (&st)[8]) {
  int64_t r = n0 + io.piece;
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
#pragma unroll
  for (int i = 0; i < 8; ++i) st[i] = base[(int64_t)io.sl[i] * N + r];
}
__device__ __forceinline__ void sc_stage(double *tile, int lane, const double (&st)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) tile[(8 * i + lane / 8) * SSTR + (lane & 7)] = st[i];
}
__device__ __forceinline__ void sc_flush(double *__restrict__ base, int64_t N, int64_t n0, int64_t lo, int64_t hi,
                                         const double *tile, int lane, int last) {
  const int64_t r = n0 + (lane & 7);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int s = 8 * i + lane / 8;
    const double v = tile[s * SSTR + (lane & 7)];
    if (s <= last && r >= lo && r <= hi) base[(int64_t)s * N + r] = v;
  }
}

// p_j = exp(c_j dt).  PAIRED: c_{2k} == c_{2k+1} for every series of the wavefront (complex terms, terms.py:171-173):
// one exponential per pair -- bit-identical to evaluating both.
template <bool PAIRED>
__device__ __forceinline__ void decay(const double (&c)[J], double dt, double (&p)[J]) {
  if constexpr (PAIRED) {
#pragma unroll
    for (int k = 0; k < J / 2; ++k) p[2 * k] = p[2 * k + 1] = exp_decay(c[2 * k] * dt);
  } else {
#pragma unroll
    for (int j = 0; j < J; ++j) p[j] = exp_decay(c[j] * dt);
  }
}

// =============================================================================================================
// Forward pass.  REC = false: log-likelihood only.  REC = true: also W rows, (d, z) pairs, checkpoints and the
// stability guard of the backward recursion.
// =============================================================================================================
template <bool REC, bool PAIRED>
__device__ __forceinline__ void fwd_body(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                         const double *__restrict__ c, int64_t c_bs, const double *__restrict__ a,
                                         const double *__restrict__ U, const double *__restrict__ V,
                                         const double *__restrict__ y, double *__restrict__ ll,
                                         int32_t *__restrict__ flag, double *__restrict__ rec, Rec R,
                                         unsigned long long *__restrict__ guard, double *lds) {
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kWave;
  const int last = (int)((B - 1 - b0) < (kWave - 1) ? (B - 1 - b0) : (kWave - 1));
  const int sl = lane < last ? lane : last;
  const int64_t b = b0 + sl;
  const RowIO io(lane, last);
  double *tU = lds, *tV = tU + kWave * RSTR, *tT = tV + kWave * RSTR, *tA = tT + kWave * SSTR, *tY = tA + kWave * SSTR;
  const double *Ub = U + b0 * N * J, *Vb = V + b0 * N * J, *ab = a + b0 * N, *yb = y + b0 * N;
  // shared t: every series reads the same grid (stride 0 between series)
  const double *tb = t + (t_bs ? b0 * N : 0);
  const int64_t tN = t_bs ? N : 0;  // series stride of t

  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[b * c_bs + j];
  double cmax = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j) cmax = fmax(cmax, cj[j]);

  // records of this wavefront
  double2 *recW = REC ? reinterpret_cast<double2 *>(rec + R.w + (size_t)blockIdx.x * N * J * kWave) : nullptr;
  double2 *recDZ = REC ? reinterpret_cast<double2 *>(rec + R.dz + (size_t)blockIdx.x * N * 2 * kWave) : nullptr;
  double *recCK = REC ? rec + R.ck + (size_t)blockIdx.x * R.nck * (NS + J) * kWave : nullptr;

  // ---- row 0 --------------------------------------------------------------------------------------------------
  double S[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) S[k] = 0.0;
  double F[J], w[J];
  double d = a[b * N], z = y[b * N];
  double rd = 1.0 / d;
  double tprev = t[b * t_bs];
#pragma unroll
  for (int j = 0; j < J; ++j) { F[j] = 0.0; w[j] = V[b * N * J + j] * rd; }
  double prod = d, quad = z * z * rd;
  int eacc = 0;
  int32_t fl = 0;
  double tseg = tprev, gmax = 0.0;  // guard: c_max * span of the current segment
  if (REC) {
#pragma unroll
    for (int q = 0; q < J / 2; ++q) recW[q * kWave + lane] = make_double2(w[2 * q], w[2 * q + 1]);
    recDZ[lane] = make_double2(d, z);
  }

  // ---- prologue: tiles of rows 1.. --------------------------------------------------------------------------------
  double2 su[8], sv[8];
  double st_[8], sa_[8], sy_[8];
  row_fetch(Ub, N, 1, io, su); row_fetch(Vb, N, 1, io, sv);
  {  // scalar fetch with the t stride
    int64_t r = 1 + io.piece; r = r > N - 1 ? N - 1 : r;
#pragma unroll
    for (int i = 0; i < 8; ++i) st_[i] = tb[(int64_t)io.sl[i] * tN + r];
  }
  sc_fetch(ab, N, 1, io, sa_); sc_fetch(yb, N, 1, io, sy_);

  for (int64_t n0 = 1; n0 < N; n0 += ST) {
    // scalar tile of rows n0 .. n0+ST-1
    lds_order();
    sc_stage(tT, lane, st_); sc_stage(tA, lane, sa_); sc_stage(tY, lane, sy_);
    {
      int64_t r = n0 + ST + io.piece; r = r > N - 1 ? N - 1 : r;
#pragma unroll
      for (int i = 0; i < 8; ++i) st_[i] = tb[(int64_t)io.sl[i] * tN + r];
    }
    sc_fetch(ab, N, n0 + ST, io, sa_); sc_fetch(yb, N, n0 + ST, io, sy_);
#pragma unroll
    for (int rt = 0; rt < ST / RT; ++rt) {
      const int64_t nt = n0 + rt * RT;
      if (nt < N) {
        lds_order();
        row_stage(tU, lane, su); row_stage(tV, lane, sv);
        row_fetch(Ub, N, nt + RT, io, su); row_fetch(Vb, N, nt + RT, io, sv);
        lds_order();
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const int64_t n = nt + r;
          if (n < N) {
            const int rs = rt * RT + r;
            const double tn = tT[lane * SSTR + rs], an = tA[lane * SSTR + rs], yn = tY[lane * SSTR + rs];
            double u[J], v[J], p[J];
            row_read(tU, lane, r, u); row_read(tV, lane, r, v);
            const double dt = tprev - tn;
            tprev = tn;
            decay<PAIRED>(cj, dt, p);
            // S = P (S + d w^T w) P   (forward.hpp:115-123);  tau = U_n S  (forward.hpp:126)
            double dw[J], tau[J];
#pragma unroll
            for (int i = 0; i < J; ++i) { dw[i] = d * w[i]; tau[i] = 0.0; }
#pragma unroll
            for (int i = 0; i < J; ++i) {
#pragma unroll
              for (int j2 = i; j2 < J; ++j2) {
                const double s = (p[i] * p[j2]) * fma(dw[i], w[j2], S[sidx(i, j2)]);
                S[sidx(i, j2)] = s;
                tau[j2] = fma(u[i], s, tau[j2]);
                if (j2 != i) tau[i] = fma(u[j2], s, tau[i]);
              }
            }
            // F = P (F + W_{n-1}^T z_{n-1})   (internal.hpp:140-143)
            double rdn = 0.0, rzn = 0.0;
#pragma unroll
            for (int j2 = 0; j2 < J; ++j2) {
              F[j2] = p[j2] * fma(w[j2], z, F[j2]);
              rdn = fma(tau[j2], u[j2], rdn);
              rzn = fma(u[j2], F[j2], rzn);
            }
            d = an - rdn;   // forward.hpp:127
            z = yn - rzn;   // internal.hpp:144
            rd = rcp_nr(d);
#pragma unroll
            for (int j2 = 0; j2 < J; ++j2) w[j2] = (v[j2] - tau[j2]) * rd;  // forward.hpp:131
            fl = ((fl == 0) & (d <= 0.0)) ? (int32_t)n : fl;              // forward.hpp:128
            prod *= d;
            quad = fma(z * z, rd, quad);
            if (r & 1) { int e; prod = frexp(prod, &e); eacc += e; }
            if (REC) {
#pragma unroll
              for (int q = 0; q < J / 2; ++q) recW[((size_t)n * (J / 2) + q) * kWave + lane] = make_double2(w[2 * q], w[2 * q + 1]);
              recDZ[(size_t)n * kWave + lane] = make_double2(d, z);
              const bool seg_end = (n % C == 0) || (n == N - 1);
              if (seg_end) {  // uniform over the wavefront
                double *ck = recCK + (size_t)((n % C == 0) ? n / C - 1 : R.nck - 1) * (NS + J) * kWave;
#pragma unroll
                for (int k = 0; k < NS; ++k) ck[k * kWave + lane] = S[k];
#pragma unroll
                for (int j2 = 0; j2 < J; ++j2) ck[(NS + j2) * kWave + lane] = F[j2];
                gmax = fmax(gmax, cmax * (tn - tseg));
                tseg = tn;
              }
            }
          }
        }
      }
    }
  }
  if (lane <= last) {
    int e;
    prod = frexp(prod, &e);
    const double logdet = log(prod) + (double)(eacc + e) * kLn2;
    flag[b] = fl;
    ll[b] = fl ? -INFINITY : -0.5 * (logdet + (double)N * kLog2Pi) - 0.5 * quad;
    if (REC) {
      // NaN-aware: a NaN span must disable the fast path (the replay kernels propagate it like the reference)
      const double g = (gmax == gmax) ? gmax : INFINITY;
      atomicMax(guard, (unsigned long long)__double_as_longlong(g));  // g >= 0: the bit pattern is monotone
    }
  }
}

constexpr int kFwdLds = (2 * kWave * RSTR + 3 * kWave * SSTR) * 8;

template <bool REC>
__global__ __launch_bounds__(kWave, 1) void k_loglik_t_fwd(int64_t B, int64_t N, const double *__restrict__ t,
                                                           int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                           const double *__restrict__ a, const double *__restrict__ U,
                                                           const double *__restrict__ V, const double *__restrict__ y,
                                                           double *__restrict__ ll, int32_t *__restrict__ flag,
                                                           double *__restrict__ rec, Rec R,
                                                           unsigned long long *__restrict__ guard) {
  __shared__ __attribute__((aligned(16))) double lds[kFwdLds / 8];
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kWave;
  const int64_t bb = (b0 + lane) < B ? (b0 + lane) : (B - 1);
  bool paired = true;
#pragma unroll
  for (int k = 0; k < J / 2; ++k) paired = paired && (c[bb * c_bs + 2 * k] == c[bb * c_bs + 2 * k + 1]);
  if (__all(paired))
    fwd_body<REC, true>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, rec, R, guard, lds);
  else
    fwd_body<REC, false>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, rec, R, guard, lds);
}

}  // namespace c2t

using namespace c2t;

extern "C" {

// Forward-only log-likelihood, one lane per series (J == 8).  Internal: dispatched by c2_loglik for large batches.
int c2_internal_loglik_t(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                         const double *a, const double *U, const double *V, const double *y, double *ll,
                         int32_t *flag, c2_stream_t stream) {
  const dim3 grid((unsigned)((B + kWave - 1) / kWave));
  Rec R{};
  hipLaunchKernelGGL((k_loglik_t_fwd<false>), grid, dim3(kWave), 0, (hipStream_t)stream, B, N, t, t_bs, c, c_bs, a, U,
                     V, y, ll, flag, (double *)nullptr, R, (unsigned long long *)nullptr);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

}  // extern "C"
