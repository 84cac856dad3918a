// c2_timepar.hip -- the forward recursions parallel along TIME, for batches too small to fill the chip row by row (widths
// J = 2, 4; the single-rhs solves also at 8).
//
// The group kernels of c2_loglik.hip walk a series row by row: with a few hundred series the chip idles (BASELINE
// configs[1]: 1024 series x 4 lanes = 64 wavefronts on 1024 SIMDs, 212 ns per dependent step).  Here a series is cut into
// CHUNKS of 64 rows, a lane walks a chunk, a wavefront owns 64 consecutive chunks of one series:
//  * `factor` (forward.hpp:105-134) is a Kalman filter in disguise -- with T = S_n (the post-decay state row n sees),
//        T' = P (T + g^T g / delta) P,   g = v - u T,  delta = a - u T u^T          (forward.hpp:115-131, one row on)
//    is the covariance update of a filter whose "explained covariance" is T -- and `solve_lower` (internal.hpp:135-145),
//    F' = P ((I - w u) F + w y), is its mean update.  What a span of rows does to ANY state it is entered with is therefore
//    captured by a small ELEMENT (two J x J matrices, two symmetric ones, two vectors, two scalars) computed by running the
//    span from the zero state, and the elements of consecutive spans combine associatively and STABLY (scattering / information
//    form: only symmetric positive definite J x J systems; see "Chunk ELEMENTS" below; numpy prototype and the derivation:
//    docs/rounds/r05.md).
//  * Forward log-likelihood (k_tp_onepass [+ k_tp_join]): ONE pass over the rows -- every chunk's element, a tree over the 64
//    lanes, sum log d and sum z^2 / d read off the total.  BASELINE configs[1]: 0.110 ms (0.42 of the HBM roofline; the
//    composite linear-fractional maps of rounds 2 - 4, chained and verified: 0.262 ms), agreement with the row-by-row
//    kernels 5e-15.
//  * `factor` (k_tp_states [+ k_tp_carry + k_tp_long_starts], k_tp_factor): the matrices of the elements, an inclusive scan,
//    the G of a prefix = the exact start state of the next chunk; every chunk then runs the ordinary recursion from its
//    start state and writes d, W.  The end state of a chunk is compared with the start state of its successor (a cheap
//    check of the whole chain); a failed factorisation, or a mismatch, sends the batch to the row-by-row kernel gated behind.
//  * single-rhs solves (k_tps_*, second half of this file): contracting affine maps per chunk.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"


namespace c2tp {
using namespace c2;

constexpr int kRows = 64;        // rows per chunk (the solves; the longest chunk of the element kernels)
constexpr int kThreads = 64;
// rows per chunk of the element kernels: a wavefront walks its 64 chunks in lock step, so a short series is cut into shorter
// chunks -- 1000 rows: 63 lanes x 16 rows instead of 16 lanes x 64
__host__ __device__ inline int chunk_rows(int64_t N) { return N <= 1024 ? 16 : (N <= 2048 ? 32 : 64); }
// chunk-start mismatch (relative to |S|) of `factor` that counts as 1 on the guard word; the row-by-row kernel recomputes the
// batch beyond 2, i.e. 5e-11 (the scanned start states agree with the sequential recursion to ~1e-15)
constexpr double kTol = 2.5e-11;

__host__ __device__ constexpr int nsym(int J) { return J * (J + 1) / 2; }
__host__ __device__ constexpr int sidx(int J, int i, int j) { return i * J - i * (i - 1) / 2 + (j - i); }  // i <= j

// ---- data movement.  A wavefront owns 64 consecutive chunks (lane <-> chunk).  A lane streaming its own chunk would touch
// 64 different lines per load instruction and the lines it reuses over the next rows do not survive in L1 / L2 (measured:
// 10 x the bytes, 0.86 ms at configs[1]).  So rows move as in c2_loglik_t.hip: one instruction loads 8 chunks x 128 bytes
// (lane l: chunk 8 i + l / 8, 16-byte piece l % 8), the pieces go through an LDS tile with a conflict-free stride, and a
// lane reads its own chunk's rows from there.  Row tiles hold 16 / J rows (one 128-byte line per chunk), scalar tiles 8 rows.
template <int J>
struct Geo {
  static constexpr int RT = 16 / J;        // rows per row tile
  static constexpr int PPR = J / 2;        // 16-byte pieces per row
  static constexpr int RSTR = 18;          // LDS stride (doubles) of a chunk in a row tile: 144 B
  static constexpr int SSTR = 9;           // ... in a scalar tile: 72 B
};
// A wavefront owns chunks k0 .. k0+63 of ONE series (grid: x over groups of 64 chunks, y over series), so the chunk a lane
// helps to load in instruction i -- chunk k0 + 8 i + lane / 8 -- is an affine function of the lane: nothing to keep.
struct Chunks {
  int64_t sbase, tbase;   // first row of the series in the (B, N) arrays / in the time grid (0 when shared)
  int64_t N, K, k0;
  int R;                  // rows per chunk (chunk_rows: 64; 32 / 16 for short series)
  __device__ __forceinline__ int64_t chunk(int i, int lane) const {
    const int64_t k = k0 + 8 * i + lane / 8;
    return k < K ? k : K - 1;
  }
  __device__ __forceinline__ int len(int64_t k) const {
    const int64_t s = k * R;
    return (int)((s + R < N ? s + R : N) - s);
  }
};
// rows r0 .. r0+RT-1 (local, clamped into the chunk) of the 64 chunks: global -> registers (a tile ahead), registers -> LDS
template <int J>
__device__ __forceinline__ void fetch_row_tile(const double *__restrict__ base, const Chunks &c, int r0, int lane, double (&v)[16]) {
  using Gm = Geo<J>;
  const int q = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t k = c.chunk(i, lane);
    const int ln = c.len(k);
    int r = r0 + q / Gm::PPR;
    r = r < ln ? r : ln - 1;
    const double2 w = *reinterpret_cast<const double2 *>(base + (c.sbase + k * c.R + r) * J + 2 * (q % Gm::PPR));
    v[2 * i] = w.x; v[2 * i + 1] = w.y;
  }
}
template <int J>
__device__ __forceinline__ void stage_row_tile(double *tile, int lane, const double (&v)[16]) {
  const int q = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    *reinterpret_cast<double2 *>(tile + (8 * i + lane / 8) * Geo<J>::RSTR + 2 * q) = make_double2(v[2 * i], v[2 * i + 1]);
}
// rows r0+shift .. r0+shift+7 of a per-row scalar (clamped to the series), shift = 1 for "the next row's t"
template <bool TIME>
__device__ __forceinline__ void fetch_scalar_tile(const double *__restrict__ base, const Chunks &c, int r0, int shift, int lane,
                                                  double (&v)[8]) {
  const int q = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t r = c.chunk(i, lane) * c.R + r0 + q + shift;
    r = r < c.N - 1 ? r : c.N - 1;
    v[i] = base[(TIME ? c.tbase : c.sbase) + r];
  }
}
__device__ __forceinline__ void stage_scalar_tile(double *tile, int lane, const double (&v)[8]) {
  const int q = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) tile[(8 * i + lane / 8) * 9 + q] = v[i];
}
// the same stream requested SIXTEEN rows at a time -- a whole 128-byte line per chunk (a 64-byte request leaves the other half
// of the line to be fetched again eight rows later, when L2 has often dropped it: 1.35 x the algorithmic bytes measured) -- and
// staged in halves into the same 8-row tile: lane l holds rows r0 + (l % 16) of chunks 4 i + l / 16
template <bool TIME>
__device__ __forceinline__ void fetch_scalar_rows16(const double *__restrict__ base, const Chunks &c, int r0, int shift, int lane,
                                                    double (&v)[16]) {
  const int q = lane & 15;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int64_t k = c.k0 + 4 * i + lane / 16;
    k = k < c.K ? k : c.K - 1;
    int64_t r = k * c.R + r0 + q + shift;
    r = r < c.N - 1 ? r : c.N - 1;
    v[i] = base[(TIME ? c.tbase : c.sbase) + r];
  }
}
__device__ __forceinline__ void stage_scalar_half(double *tile, int lane, const double (&v)[16], int half) {
  const int q = lane & 15;
  if ((q >> 3) == half) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[(4 * i + lane / 16) * 9 + (q & 7)] = v[i];
  }
}

// S' = X Y^-1 for J x J blocks by Gauss-Jordan with partial pivoting on Y^T (rows of the augmented [Y^T | X^T]), fully
// unrolled.
// LDS tile -> global, the mirror images (rows beyond the chunk are skipped)
template <int J>
__device__ __forceinline__ void flush_row_tile(double *__restrict__ base, const Chunks &c, int r0, int lane, const double *tile) {
  using Gm = Geo<J>;
  const int q = lane & 7;
  double2 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const double2 *>(tile + (8 * i + lane / 8) * Gm::RSTR + 2 * q);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t k = c.k0 + 8 * i + lane / 8;
    const int r = r0 + q / Gm::PPR;
    if (k < c.K && r < c.len(k))
      *reinterpret_cast<double2 *>(base + (c.sbase + k * c.R + r) * J + 2 * (q % Gm::PPR)) = v[i];
  }
}
__device__ __forceinline__ void flush_scalar_tile(double *__restrict__ base, const Chunks &c, int r0, int lane, const double *tile) {
  const int q = lane & 7;
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = tile[(8 * i + lane / 8) * 9 + q];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t k = c.k0 + 8 * i + lane / 8;
    if (k < c.K && r0 + q < c.len(k)) base[c.sbase + k * c.R + r0 + q] = v[i];
  }
}

// relative mismatch between the end state a chunk computed and the start state its successor was given
template <int J>
__device__ __forceinline__ double start_mismatch(const double (&Send)[nsym(J)], const double *__restrict__ starts, int64_t G,
                                                 int64_t gnext) {
  double dmax = 0.0, smax = 0.0;
#pragma unroll
  for (int q = 0; q < nsym(J); ++q) {
    const double se = Send[q], sn = starts[(int64_t)q * G + gnext];
    dmax = fmax(dmax, fabs(se - sn));
    smax = fmax(smax, fabs(se));
    if (!(se == se) || !(sn == sn)) dmax = INFINITY;
  }
  return dmax / fmax(smax, 1e-300);
}

// =============================================================================================================
// Chunk ELEMENTS in scattering form.
//
// Run from the ZERO state over a span of rows, the recursions yield the span's element
//     A = prod P (I - w^T u)   (transition of F),   G = the end state T,   Q = sum r^T r / d  (r = u A_before),
//     g = the end state F,   h = sum z r / d,   q0 = sum z^2 / d,   prod d
// and entered with a state (T, F) instead, the same span gives (Gt = T (I - Q T)^-1, rho = h - Q F)
//     T' = G + A Gt A^T,   F' = g + A (I - T Q)^-1 (F - T h),   sum log d = log prod d + log det(I - Q T),
//     sum z^2 / d = q0 - 2 h.F + F^T Q F + rho^T Gt rho .
// Two consecutive spans therefore COMBINE into the element of their union, with nothing to invert but the symmetric
// positive definite  Ks = I - L^T Q2 L,  G1 = L L^T:
//     Gt = L Ks^-1 L^T  (= G1 (I - Q2 G1)^-1),   M = I + Gt Q2,   rho = h2 - Q2 g1
//     A = A2 M A1,   G = G2 + A2 Gt A2^T,   Q = Q1 + A1^T Q2 M A1,   g = g2 + A2 (g1 - Gt rho),
//     h = h1 + A1^T (rho + Q2 Gt rho),   q0 = q0_1 + q0_2 - g1.(h2 + rho) + rho.Gt rho,   prod d = prod1 prod2 det Ks
// Ks has the ratios d_n(T) / d_n(0) between its extreme eigenvalues: the combination is as well-conditioned as the
// factorisation itself, whatever the length of the spans -- they combine in a tree (or a scan) and nothing is verified or
// repeated.  Ks positive definite  <=>  every d_n of the later span stays positive when it is entered with G1 (the
// eigenvalues fall monotonically along the span): a failed Cholesky pivot, or d_n <= 0 from the zero state (an upper bound of
// the true d_n), marks the series failed, and the row-by-row kernel behind the gate reports the reference's flag.
template <int J>
struct Elem {
  double A[J][J], G[nsym(J)], Q[nsym(J)], g[J], h[J], q0, prod;
  int ex;   // prod d = prod * 2^ex
};
template <int J>
__device__ __forceinline__ void elem_identity(Elem<J> &e) {
#pragma unroll
  for (int i = 0; i < J; ++i) {
    e.g[i] = 0.0; e.h[i] = 0.0;
#pragma unroll
    for (int j = 0; j < J; ++j) e.A[i][j] = i == j ? 1.0 : 0.0;
  }
#pragma unroll
  for (int q = 0; q < nsym(J); ++q) { e.G[q] = 0.0; e.Q[q] = 0.0; }
  e.q0 = 0.0; e.prod = 1.0; e.ex = 0;
}
// VEC = false everywhere below: the matrices only (A, G, Q and the product of the pivots -- what `factor` needs)
template <int J, bool VEC = true>
__device__ __forceinline__ void elem_from_lane(const Elem<J> &e, int src, Elem<J> &o) {
#pragma unroll
  for (int i = 0; i < J; ++i) {
    if (VEC) { o.g[i] = __shfl(e.g[i], src, 64); o.h[i] = __shfl(e.h[i], src, 64); }
#pragma unroll
    for (int j = 0; j < J; ++j) o.A[i][j] = __shfl(e.A[i][j], src, 64);
  }
#pragma unroll
  for (int q = 0; q < nsym(J); ++q) { o.G[q] = __shfl(e.G[q], src, 64); o.Q[q] = __shfl(e.Q[q], src, 64); }
  if (VEC) o.q0 = __shfl(e.q0, src, 64);
  o.prod = __shfl(e.prod, src, 64); o.ex = __shfl(e.ex, src, 64);
}
__device__ __forceinline__ constexpr int sym(int J, int i, int j) { return i <= j ? sidx(J, i, j) : sidx(J, j, i); }

// Gt = T (I - Q T)^-1 = L Ks^-1 L^T for symmetric positive semi-definite T = L L^T (L: J columns, not triangular), Q;
// det = det Ks = det(I - Q T); returns false when Ks is not positive definite
template <int J>
__device__ __forceinline__ bool posterior(const double (&T)[nsym(J)], const double (&Q)[nsym(J)], double (&Gt)[nsym(J)],
                                          double &det) {
  // T = L L^T by the outer-product Cholesky with DIAGONAL PIVOTING: column k of L comes from the largest remaining diagonal
  // entry, and the factorisation stops at the rounding level of the largest one.  T is often numerically rank-deficient (the
  // explained covariance after a few rows that look in nearly the same direction): without pivoting its noise pivots are
  // divided into noise columns and L L^T misses T by 1e-7 (measured: 16-row chunks at width 8, log-likelihood off by 1e-9).
  double L[J][J], S[nsym(J)];
  double tol = 0.0;
#pragma unroll
  for (int q = 0; q < nsym(J); ++q) S[q] = T[q];
#pragma unroll
  for (int k = 0; k < J; ++k) {
    double sp = S[sidx(J, 0, 0)];
    int mi = 0;
#pragma unroll
    for (int i = 1; i < J; ++i) {
      const bool gt = S[sidx(J, i, i)] > sp;
      sp = gt ? S[sidx(J, i, i)] : sp;
      mi = gt ? i : mi;
    }
    if (k == 0) tol = 1e-15 * sp;
    const bool ok = sp > tol;
    const double inv = ok ? rcp_nr(sqrt(sp)) : 0.0;
    double l[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double v = S[sym(J, i, 0)];
#pragma unroll
      for (int jj = 1; jj < J; ++jj) v = mi == jj ? S[sym(J, i, jj)] : v;
      l[i] = v * inv;
      L[i][k] = l[i];
    }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int jj = i; jj < J; ++jj) S[sidx(J, i, jj)] = fma(-l[i], l[jj], S[sidx(J, i, jj)]);
  }
  // X = Q L,  Ks = I - L^T X (lower triangle)
  double X[J][J], Ks[J][J];
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(Q[sym(J, i, k)], L[k][j], v);
      X[i][j] = v;
    }
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double v = i == j ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(-L[k][i], X[k][j], v);
      Ks[i][j] = v;
    }
  // Ks = C C^T in place, det Ks = the product of the pivots
  double ic[J];
  bool good = true;
  det = 1.0;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    double sp = Ks[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) sp = fma(-Ks[j][k], Ks[j][k], sp);
    good = good && sp > 0.0;
    det *= sp;
    const double cjj = sqrt(sp);
    ic[j] = rcp_nr(cjj);
    Ks[j][j] = cjj;
#pragma unroll
    for (int i = j + 1; i < J; ++i) {
      double v = Ks[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = fma(-Ks[i][k], Ks[j][k], v);
      Ks[i][j] = v * ic[j];
    }
  }
  // Y = C^-1 L^T,  Gt = Y^T Y
  double Y[J][J];
#pragma unroll
  for (int m = 0; m < J; ++m)
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double v = L[m][i];
#pragma unroll
      for (int k = 0; k < i; ++k) v = fma(-Ks[i][k], Y[k][m], v);
      Y[i][m] = v * ic[i];
    }
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = i; j < J; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(Y[k][i], Y[k][j], v);
      Gt[sidx(J, i, j)] = v;
    }
  return good;
}

// e1 (the earlier span) <- e1 followed by e2
template <int J, bool VEC = true>
__device__ __forceinline__ void elem_combine(Elem<J> &e1, const Elem<J> &e2) {
  double Gt[nsym(J)], det;
  const bool good = posterior<J>(e1.G, e2.Q, Gt, det);
  if (VEC) {
    double rho[J], Gr[J], tv[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double v = e2.h[i];
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(-e2.Q[sym(J, i, k)], e1.g[k], v);
      rho[i] = v;
    }
    double q0 = e1.q0 + e2.q0;
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(Gt[sym(J, i, k)], rho[k], v);
      Gr[i] = v;
      q0 = fma(-e1.g[i], e2.h[i] + rho[i], q0);
    }
#pragma unroll
    for (int i = 0; i < J; ++i) {
      q0 = fma(rho[i], Gr[i], q0);
      double v = rho[i];
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(e2.Q[sym(J, i, k)], Gr[k], v);
      tv[i] = v;                       // rho + Q2 Gt rho
    }
    double gn[J], hn[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double v = e2.g[i], hv = e1.h[i];
#pragma unroll
      for (int k = 0; k < J; ++k) {
        v = fma(e2.A[i][k], e1.g[k] - Gr[k], v);
        hv = fma(e1.A[k][i], tv[k], hv);
      }
      gn[i] = v; hn[i] = hv;
    }
#pragma unroll
    for (int i = 0; i < J; ++i) { e1.g[i] = gn[i]; e1.h[i] = hn[i]; }
    e1.q0 = q0;
  }
  // QA = Q2 A1, MA = A1 + Gt QA (= M A1), Q = Q1 + A1^T (Q2 MA), G = G2 + (A2 Gt) A2^T, A = A2 MA
  double MA[J][J];
  {
    double QA[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) v = fma(e2.Q[sym(J, i, k)], e1.A[k][j], v);
        QA[i][j] = v;
      }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        double v = e1.A[i][j];
#pragma unroll
        for (int k = 0; k < J; ++k) v = fma(Gt[sym(J, i, k)], QA[k][j], v);
        MA[i][j] = v;
      }
  }
  {
    double QMA[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) v = fma(e2.Q[sym(J, i, k)], MA[k][j], v);
        QMA[i][j] = v;
      }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = i; j < J; ++j) {
        double v = e1.Q[sidx(J, i, j)];
#pragma unroll
        for (int k = 0; k < J; ++k) v = fma(e1.A[k][i], QMA[k][j], v);
        e1.Q[sidx(J, i, j)] = v;
      }
  }
  {
    double AG[J][J];
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = 0; j < J; ++j) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < J; ++k) v = fma(e2.A[i][k], Gt[sym(J, k, j)], v);
        AG[i][j] = v;
      }
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = i; j < J; ++j) {
        double v = e2.G[sidx(J, i, j)];
#pragma unroll
        for (int k = 0; k < J; ++k) v = fma(AG[i][k], e2.A[j][k], v);
        e1.G[sidx(J, i, j)] = v;
      }
  }
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(e2.A[i][k], MA[k][j], v);
      e1.A[i][j] = v;
    }
  int ex;
  e1.prod = frexp(e1.prod * e2.prod * det, &ex);
  e1.ex += e2.ex + ex;
  if (!good) e1.prod = __longlong_as_double(0x7ff8000000000000ll);
}
// the state T a span entered with T leaves behind: Tn = G + A Gt A^T (false: not positive definite)
template <int J>
__device__ __forceinline__ bool elem_apply(const Elem<J> &e, const double (&T)[nsym(J)], double (&Tn)[nsym(J)]) {
  double Gt[nsym(J)], det, AG[J][J];
  const bool good = posterior<J>(T, e.Q, Gt, det);
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(e.A[i][k], Gt[sym(J, k, j)], v);
      AG[i][j] = v;
    }
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = i; j < J; ++j) {
      double v = e.G[sidx(J, i, j)];
#pragma unroll
      for (int k = 0; k < J; ++k) v = fma(AG[i][k], e.A[j][k], v);
      Tn[sidx(J, i, j)] = v;
    }
  return good;
}
// the element of the first `count` lanes' spans, in lane 0 (lane l: spans l .. l + 2^level - 1 as far as they are complete)
template <int J>
__device__ __forceinline__ void elem_tree(Elem<J> &e, int lane, int count = 64) {
#pragma unroll 1
  for (int off = 1; off < count; off <<= 1) {
    Elem<J> o;
    elem_from_lane<J>(e, (lane + off) & 63, o);
    elem_combine<J>(e, o);
  }
}
// inclusive scan (matrices only): lane l ends up with the element of spans 0 .. l
template <int J>
__device__ __forceinline__ void elem_scan(Elem<J> &e, int lane, int count = 64) {
#pragma unroll 1
  for (int off = 1; off < count; off <<= 1) {
    Elem<J> o;
    elem_from_lane<J, false>(e, lane >= off ? lane - off : lane, o);
    elem_combine<J, false>(o, e);
    if (lane >= off) {
#pragma unroll
      for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) e.A[i][j] = o.A[i][j];
#pragma unroll
      for (int q = 0; q < nsym(J); ++q) { e.G[q] = o.G[q]; e.Q[q] = o.Q[q]; }
      e.prod = o.prod; e.ex = o.ex;
    }
  }
}
// matrices of an element in memory: [entry][at], stride `W` between entries
template <int J>
struct ElemIO {
  static constexpr int N_ = J * J + 2 * nsym(J) + 2 * J + 3;   // A, G, Q, g, h, q0, prod, ex
  static constexpr int NM = J * J + 2 * nsym(J) + 1;           // A, G, Q, prod (NaN: failed)
  static constexpr int REC = (N_ + 3) & ~3;                    // a contiguous record per chunk (k_tp_onepass<J, 2>)
};
template <int J>
__device__ __forceinline__ void elem_store_matrices(const Elem<J> &e, double *__restrict__ base, int64_t W, int64_t at) {
  int q = 0;
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) base[(int64_t)(q++) * W + at] = e.A[i][j];
#pragma unroll
  for (int i = 0; i < nsym(J); ++i) base[(int64_t)(q++) * W + at] = e.G[i];
#pragma unroll
  for (int i = 0; i < nsym(J); ++i) base[(int64_t)(q++) * W + at] = e.Q[i];
  base[(int64_t)(q++) * W + at] = e.prod;
}
template <int J>
__device__ __forceinline__ void elem_load_matrices(Elem<J> &e, const double *__restrict__ base, int64_t W, int64_t at) {
  int q = 0;
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) e.A[i][j] = base[(int64_t)(q++) * W + at];
#pragma unroll
  for (int i = 0; i < nsym(J); ++i) e.G[i] = base[(int64_t)(q++) * W + at];
#pragma unroll
  for (int i = 0; i < nsym(J); ++i) e.Q[i] = base[(int64_t)(q++) * W + at];
  e.prod = base[(int64_t)(q++) * W + at];
  e.ex = 0;
}

// rows of the chunks from the zero state -> elements -> the tree.  ONE wavefront per 64 chunks of a series; MODE 0: the
// series has at most 64 chunks and the wavefront finishes it (ll, flag, gate word); MODE 1: its element goes to `elems`
// ([entry][series * wavefronts + wavefront]) for k_tp_join; MODE 2: no tree, every chunk's element to `elems` as a record.
template <int J, int MODE>
__global__ __launch_bounds__(kThreads) void k_tp_onepass(int64_t B, int64_t N, int64_t K, int R, const double *__restrict__ t,
                                                         int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                         const double *__restrict__ a, const double *__restrict__ U,
                                                         const double *__restrict__ V, const double *__restrict__ yv,
                                                         double *__restrict__ elems, double *__restrict__ ll,
                                                         int32_t *__restrict__ flag, unsigned long long *__restrict__ guard) {
  using Gm = Geo<J>;
  constexpr int NS = nsym(J);
  __shared__ __attribute__((aligned(16))) double lds[2 * 64 * Gm::RSTR + 3 * 64 * Gm::SSTR];
  double *tU = lds, *tV = tU + 64 * Gm::RSTR, *tA = tV + 64 * Gm::RSTR, *tT = tA + 64 * Gm::SSTR, *tY = tT + 64 * Gm::SSTR;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y;
  const Chunks ch{b * N, b * t_bs, N, K, (int64_t)blockIdx.x * kThreads, R};
  int64_t k = ch.k0 + lane;
  const bool inr = k < K;
  if (!inr) k = K - 1;
  const int64_t s = k * R;
  const int len = ch.len(k);
  Elem<J> e;
  elem_identity<J>(e);
  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[b * c_bs + j];
  bool ceq = J % 2 == 0;
#pragma unroll
  for (int j = 0; j + 1 < J; j += 2) ceq = ceq && cj[j] == cj[j + 1];
  double prod = 1.0, failed = 0.0;
  int eacc = 0;
  double tn = t[b * t_bs + s];
  double vu[16], vv[16], va[16], vy[16], vt[16];   // tiles requested one tile ahead (scalars: 16 rows, staged in halves)
  fetch_scalar_rows16<false>(a, ch, 0, 0, lane, va);
  fetch_scalar_rows16<false>(yv, ch, 0, 0, lane, vy);
  fetch_scalar_rows16<true>(t, ch, 0, 1, lane, vt);   // t of the NEXT row
  fetch_row_tile<J>(U, ch, 0, lane, vu);
  fetch_row_tile<J>(V, ch, 0, lane, vv);
  for (int r0 = 0; r0 < R; r0 += 8) {
    lds_order();
    stage_scalar_half(tA, lane, va, (r0 >> 3) & 1);
    stage_scalar_half(tY, lane, vy, (r0 >> 3) & 1);
    stage_scalar_half(tT, lane, vt, (r0 >> 3) & 1);
    if ((r0 & 8) && r0 + 8 < R) {
      fetch_scalar_rows16<false>(a, ch, r0 + 8, 0, lane, va);
      fetch_scalar_rows16<false>(yv, ch, r0 + 8, 0, lane, vy);
      fetch_scalar_rows16<true>(t, ch, r0 + 8, 1, lane, vt);
    }
#pragma unroll 1
    for (int rt = 0; rt < 8; rt += Gm::RT) {
      lds_order();
      stage_row_tile<J>(tU, lane, vu);
      stage_row_tile<J>(tV, lane, vv);
      if (r0 + rt + Gm::RT < R) {
        fetch_row_tile<J>(U, ch, r0 + rt + Gm::RT, lane, vu);
        fetch_row_tile<J>(V, ch, r0 + rt + Gm::RT, lane, vv);
      }
      lds_order();
#pragma unroll
      for (int r = 0; r < Gm::RT; ++r) {
        const int i0 = r0 + rt + r;          // row of the chunk
        const int64_t n = s + i0;
        if (i0 < len) {
          double u[J], v[J], tau[J], rr[J], w[J];
#pragma unroll
          for (int j = 0; j < J; ++j) {
            u[j] = tU[lane * Gm::RSTR + r * J + j]; v[j] = tV[lane * Gm::RSTR + r * J + j];
            tau[j] = 0.0; rr[j] = 0.0;
          }
          // tau = u T (forward.hpp:126), rr = u A, z = y - u.g (internal.hpp:140)
#pragma unroll
          for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = i; j < J; ++j) {
              const double sv = e.G[sidx(J, i, j)];
              tau[j] = fma(u[i], sv, tau[j]);
              if (j != i) tau[i] = fma(u[j], sv, tau[i]);
            }
          double d = tA[lane * Gm::SSTR + rt + r], z0 = tY[lane * Gm::SSTR + rt + r];
#pragma unroll
          for (int j = 0; j < J; ++j) {
            d = fma(-tau[j], u[j], d);          // forward.hpp:127
            z0 = fma(-u[j], e.g[j], z0);
#pragma unroll
            for (int i = 0; i < J; ++i) rr[j] = fma(u[i], e.A[i][j], rr[j]);
          }
          const double rd = rcp_nr(d);
          failed = (failed == 0.0 && n > 0 && !(d > 0.0)) ? (double)n : failed;   // forward.hpp:128 (first row)
#pragma unroll
          for (int j = 0; j < J; ++j) w[j] = (v[j] - tau[j]) * rd;   // forward.hpp:131
          prod *= d;
          if (i0 & 1) { int ex; prod = frexp(prod, &ex); eacc += ex; }
          const double z0d = z0 * rd;
          e.q0 = fma(z0, z0d, e.q0);
#pragma unroll
          for (int i = 0; i < J; ++i) {
            e.h[i] = fma(z0d, rr[i], e.h[i]);
            const double rid = rr[i] * rd;
#pragma unroll
            for (int j = i; j < J; ++j) e.Q[sidx(J, i, j)] = fma(rid, rr[j], e.Q[sidx(J, i, j)]);
          }
          if (n + 1 < N) {   // on to row n + 1 (forward.hpp:115-123, internal.hpp:140-143)
            const double tn1 = tT[lane * Gm::SSTR + rt + r];
            double p[J];
            if (ceq) {   // (complex terms: the rates come in pairs)
#pragma unroll
              for (int j = 0; j + 1 < J; j += 2) { p[j] = exp_decay(cj[j] * (tn - tn1)); p[j + 1] = p[j]; }
            } else {
#pragma unroll
              for (int j = 0; j < J; ++j) p[j] = exp_decay(cj[j] * (tn - tn1));
            }
            tn = tn1;
#pragma unroll
            for (int i = 0; i < J; ++i) {
              const double dwi = d * w[i];
#pragma unroll
              for (int j = i; j < J; ++j) e.G[sidx(J, i, j)] = (p[i] * p[j]) * fma(dwi, w[j], e.G[sidx(J, i, j)]);
#pragma unroll
              for (int j = 0; j < J; ++j) e.A[i][j] = p[i] * fma(-w[i], rr[j], e.A[i][j]);
              e.g[i] = p[i] * fma(w[i], z0, e.g[i]);
            }
          }
        }
      }
    }
  }
  {
    int ex;
    e.prod = frexp(prod, &ex);
    e.ex = eacc + ex;
  }
  if (!inr) { elem_identity<J>(e); failed = 0.0; }   // (lanes beyond the series ran its last chunk for the loads' sake)
  if constexpr (MODE != 2) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const double f2 = __shfl_xor(failed, o, 64);
      failed = failed == 0.0 ? f2 : (f2 == 0.0 ? failed : fmin(failed, f2));
    }
  }
  if constexpr (MODE == 2) {   // every chunk's element as a record (width 8: combined by k_e8_tree)
    if (inr) {
      double *rec = elems + (size_t)(b * K + k) * ElemIO<J>::REC;
      int q = 0;
#pragma unroll
      for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) rec[q++] = e.A[i][j];
#pragma unroll
      for (int i = 0; i < NS; ++i) rec[q++] = e.G[i];
#pragma unroll
      for (int i = 0; i < NS; ++i) rec[q++] = e.Q[i];
#pragma unroll
      for (int i = 0; i < J; ++i) rec[q++] = e.g[i];
#pragma unroll
      for (int i = 0; i < J; ++i) rec[q++] = e.h[i];
      rec[q++] = e.q0;
      rec[q++] = failed != 0.0 ? __longlong_as_double(0x7ff8000000000000ll) : e.prod;
      rec[q++] = (double)e.ex;
    }
    return;
  }
  elem_tree<J>(e, lane, (int)((K - ch.k0) < kThreads ? (K - ch.k0) : kThreads));
  if constexpr (MODE == 0) {
    const double logdet = log(e.prod) + (double)e.ex * 0.693147180559945309417;
    const bool bad = failed != 0.0 || !(logdet == logdet) || !(e.q0 == e.q0);
    if (lane == 0) {
      ll[b] = -0.5 * (logdet + e.q0 + (double)N * 1.83787706640934548356);   // numpy.py:84-109
      flag[b] = 0;
      if (bad) {   // left to the row-by-row kernel behind the gate (it reports the reference's flag and -inf)
        atomicMax(guard, (unsigned long long)__double_as_longlong(INFINITY));
        atomicMax(guard + 1, (unsigned long long)__double_as_longlong(INFINITY));
      }
    }
  } else if (lane == 0) {
    const int64_t W = (int64_t)gridDim.x * B, at = b * gridDim.x + blockIdx.x;
    int q = 0;
#pragma unroll
    for (int i = 0; i < J; ++i)
#pragma unroll
      for (int j = 0; j < J; ++j) elems[(int64_t)(q++) * W + at] = e.A[i][j];
#pragma unroll
    for (int i = 0; i < NS; ++i) elems[(int64_t)(q++) * W + at] = e.G[i];
#pragma unroll
    for (int i = 0; i < NS; ++i) elems[(int64_t)(q++) * W + at] = e.Q[i];
#pragma unroll
    for (int i = 0; i < J; ++i) elems[(int64_t)(q++) * W + at] = e.g[i];
#pragma unroll
    for (int i = 0; i < J; ++i) elems[(int64_t)(q++) * W + at] = e.h[i];
    elems[(int64_t)(q++) * W + at] = e.q0;
    elems[(int64_t)(q++) * W + at] = failed != 0.0 ? __longlong_as_double(0x7ff8000000000000ll) : e.prod;
    elems[(int64_t)(q++) * W + at] = (double)e.ex;
  }
}
// the elements of a long series' wavefronts (more than 64 chunks) joined: one wavefront per series, 64 elements a round
template <int J>
__global__ __launch_bounds__(kThreads) void k_tp_join(int64_t B, int64_t N, int64_t Wn, const double *__restrict__ elems,
                                                      double *__restrict__ ll, int32_t *__restrict__ flag,
                                                      unsigned long long *__restrict__ guard) {
  constexpr int NS = nsym(J);
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x, W = Wn * B;
  Elem<J> carry;
  elem_identity<J>(carry);
  for (int64_t base = 0; base < Wn; base += kThreads) {
    Elem<J> e;
    elem_identity<J>(e);
    if (base + lane < Wn) {
      const int64_t at = b * Wn + base + lane;
      int q = 0;
#pragma unroll
      for (int i = 0; i < J; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) e.A[i][j] = elems[(int64_t)(q++) * W + at];
#pragma unroll
      for (int i = 0; i < NS; ++i) e.G[i] = elems[(int64_t)(q++) * W + at];
#pragma unroll
      for (int i = 0; i < NS; ++i) e.Q[i] = elems[(int64_t)(q++) * W + at];
#pragma unroll
      for (int i = 0; i < J; ++i) e.g[i] = elems[(int64_t)(q++) * W + at];
#pragma unroll
      for (int i = 0; i < J; ++i) e.h[i] = elems[(int64_t)(q++) * W + at];
      e.q0 = elems[(int64_t)(q++) * W + at];
      e.prod = elems[(int64_t)(q++) * W + at];
      e.ex = (int)elems[(int64_t)(q++) * W + at];
    }
    elem_tree<J>(e, lane, (int)((Wn - base) < kThreads ? (Wn - base) : kThreads));
    Elem<J> first;
    elem_from_lane<J>(e, 0, first);
    elem_combine<J>(carry, first);
  }
  const double logdet = log(carry.prod) + (double)carry.ex * 0.693147180559945309417;
  const bool bad = !(logdet == logdet) || !(carry.q0 == carry.q0);
  if (lane == 0) {
    ll[b] = -0.5 * (logdet + carry.q0 + (double)N * 1.83787706640934548356);
    flag[b] = 0;
    if (bad) {
      atomicMax(guard, (unsigned long long)__double_as_longlong(INFINITY));
      atomicMax(guard + 1, (unsigned long long)__double_as_longlong(INFINITY));
    }
  }
}

// =============================================================================================================
// Width 8: the elements of the chunks (k_tp_onepass<8, 2>: a record per chunk) combined by WORKGROUPS.  An element of width 8
// (155 doubles) does not fit a lane next to the temporaries of a combination, so here a WAVEFRONT combines one pair at a time
// with lane (i, j) <-> entry (i, j) of the 8 x 8 matrices, everything through its private LDS block: the Cholesky factor of G1,
// the symmetric positive definite Ks = I - L^T Q2 L inverted in place by Gauss-Jordan (its pivots are the Cholesky pivots
// squared: all positive <=> the later span's factorisation stays positive), eleven 8 x 8 products, a handful of vectors.
// A workgroup of 16 wavefronts reduces up to `span` (64 for a handful of series -- more workgroups, shorter levels --, else 512)
// consecutive elements of one series level by level (the levels' results in a global scratch block); longer series take
// another launch over the workgroups' results.
#ifndef C2_E8_WAVES
#define C2_E8_WAVES 16
#endif
constexpr int kE8Waves = C2_E8_WAVES;
struct E8Lds {
  double A1[64], G1[64], Q1[64], A2[64], G2[64], Q2[64], L[64], X[64], Ks[64], Z[64], Gt[64], MA[64], T1[64];
  double g1[8], h1[8], g2[8], h2[8], rho[8], Gr[8], tv[8], red[8];
};
constexpr int kE8A = 0, kE8G = 64, kE8Q = 100, kE8g = 136, kE8h = 144, kE8q0 = 152, kE8prod = 153, kE8ex = 154;

// ro <- r1 followed by r2 (records of ElemIO<8>::REC doubles in global memory); all 64 lanes of one wavefront
__device__ __forceinline__ void e8_combine(const double *__restrict__ r1, const double *__restrict__ r2, double *__restrict__ ro,
                                           E8Lds &m, int lane) {
  const int i = lane >> 3, j = lane & 7, sij = sym(8, i, j);
  auto mm = [&](const double *X, int xi, int xk, const double *Y, int yk, int yj) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v = fma(X[i * xi + k * xk], Y[k * yk + j * yj], v);
    return v;
  };
  lds_order();   // (the previous combination's reads are done)
  m.A1[lane] = r1[kE8A + lane]; m.G1[lane] = r1[kE8G + sij]; m.Q1[lane] = r1[kE8Q + sij];
  m.A2[lane] = r2[kE8A + lane]; m.G2[lane] = r2[kE8G + sij]; m.Q2[lane] = r2[kE8Q + sij];
  if (lane < 8) { m.g1[lane] = r1[kE8g + lane]; m.h1[lane] = r1[kE8h + lane]; m.g2[lane] = r2[kE8g + lane]; m.h2[lane] = r2[kE8h + lane]; }
  const double q01 = r1[kE8q0], q02 = r2[kE8q0], prod1 = r1[kE8prod], prod2 = r2[kE8prod];
  const int ex1 = (int)r1[kE8ex], ex2 = (int)r2[kE8ex];
  // G1 = L L^T by the outer-product Cholesky with diagonal pivoting (see `posterior`): S = the Schur complement (in m.Z)
  m.Z[lane] = m.G1[lane];
  double tol = 0.0;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    lds_order();
    double sp = m.Z[0];
    int mi = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const double dq = m.Z[q * 9];
      const bool gt = dq > sp;
      sp = gt ? dq : sp;
      mi = gt ? q : mi;
    }
    if (k == 0) tol = 1e-15 * sp;
    const bool ok = sp > tol;
    const double inv = ok ? 1.0 / sqrt(sp) : 0.0;
    const double li = m.Z[i * 8 + mi] * inv, lj = m.Z[j * 8 + mi] * inv, own = m.Z[lane];
    lds_order();
    m.Z[lane] = fma(-li, lj, own);
    if (j == k) m.L[lane] = li;
  }
  lds_order();
  m.X[lane] = mm(m.Q2, 8, 1, m.L, 8, 1);                                  // X = Q2 L
  lds_order();
  m.Ks[lane] = (i == j ? 1.0 : 0.0) - mm(m.L, 1, 8, m.X, 8, 1);           // Ks = I - L^T X
  // Ks <- Ks^-1 in place (Gauss-Jordan, no pivoting: symmetric positive definite); det Ks = the product of the pivots
  double det = 1.0;
  bool good = true;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {
    lds_order();
    const double p = m.Ks[k * 9], rkj = m.Ks[k * 8 + j], cik = m.Ks[i * 8 + k], own = m.Ks[lane];
    good = good && p > 0.0;
    det *= p;
    const double ip = 1.0 / p;
    lds_order();
    m.Ks[lane] = (i == k && j == k) ? ip : (i == k ? rkj * ip : (j == k ? -cik * ip : fma(-cik * ip, rkj, own)));
  }
  lds_order();
  m.Z[lane] = mm(m.Ks, 8, 1, m.L, 1, 8);                                  // Z = Ks^-1 L^T
  lds_order();
  m.Gt[lane] = mm(m.L, 8, 1, m.Z, 8, 1);                                  // Gt = L Z  (= G1 (I - Q2 G1)^-1)
  lds_order();
  // vectors (lanes 0 .. 7, one entry each)
  double q0 = 0.0;
  if (lane < 8) {
    double v = m.h2[lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) v = fma(-m.Q2[lane * 8 + k], m.g1[k], v);
    m.rho[lane] = v;
  }
  lds_order();
  if (lane < 8) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v = fma(m.Gt[lane * 8 + k], m.rho[k], v);
    m.Gr[lane] = v;
    m.red[lane] = fma(m.rho[lane], v, -m.g1[lane] * (m.h2[lane] + m.rho[lane]));
  }
  lds_order();
  double gn = 0.0, hn = 0.0;
  if (lane < 8) {
    double v = m.rho[lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) v = fma(m.Q2[lane * 8 + k], m.Gr[k], v);
    m.tv[lane] = v;                                                       // rho + Q2 Gt rho
    gn = m.g2[lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) gn = fma(m.A2[lane * 8 + k], m.g1[k] - m.Gr[k], gn);
  }
  lds_order();
  if (lane < 8) {
    hn = m.h1[lane];
#pragma unroll
    for (int k = 0; k < 8; ++k) hn = fma(m.A1[k * 8 + lane], m.tv[k], hn);
  }
  q0 = q01 + q02;
#pragma unroll
  for (int k = 0; k < 8; ++k) q0 += m.red[k];
  // matrices
  m.T1[lane] = mm(m.Q2, 8, 1, m.A1, 8, 1);                                // Q2 A1
  lds_order();
  m.MA[lane] = m.A1[lane] + mm(m.Gt, 8, 1, m.T1, 8, 1);                   // M A1
  lds_order();
  const double An = mm(m.A2, 8, 1, m.MA, 8, 1);                           // A = A2 M A1
  m.T1[lane] = mm(m.Q2, 8, 1, m.MA, 8, 1);                                // Q2 M A1
  m.X[lane] = mm(m.A2, 8, 1, m.Gt, 8, 1);                                 // A2 Gt
  lds_order();
  const double Qn = m.Q1[lane] + mm(m.A1, 1, 8, m.T1, 8, 1);              // Q = Q1 + A1^T Q2 M A1
  const double Gn = m.G2[lane] + mm(m.X, 8, 1, m.A2, 1, 8);               // G = G2 + A2 Gt A2^T
  ro[kE8A + lane] = An;
  if (i <= j) { ro[kE8G + sij] = Gn; ro[kE8Q + sij] = Qn; }
  if (lane < 8) { ro[kE8g + lane] = gn; ro[kE8h + lane] = hn; }
  if (lane == 0) {
    int ex;
    const double pr = frexp(prod1 * prod2 * det, &ex);
    ro[kE8q0] = q0;
    ro[kE8prod] = good ? pr : __longlong_as_double(0x7ff8000000000000ll);
    ro[kE8ex] = (double)(ex1 + ex2 + ex);
  }
}
// ---- width 8: the chunk pass with the element SPREAD OVER THE EIGHT LANES OF A GROUP (round 6).  k_tp_onepass<8, 2> keeps
// an element (155 doubles) in ONE lane: ~550 instructions per row with eight exponentials, 3.6 us per row.  Here lane j of a
// group owns COLUMN j of A, G and Q in XOR order (slot k = row j ^ k: c2_loglik_helpers.hpp), so that u T, u A are eight
// multiply-adds on the lane's own registers, d one butterfly sum, and the rank-one updates need the gathered w, r = u A and
// decay vectors (three DPP gathers); every lane computes ONE exponential.  A wavefront walks eight chunks; the records it
// writes are those of k_tp_onepass<8, 2> (ElemIO<8>), VEC = false: matrices only (g, h, q0 zero -- `factor`).
template <bool VEC>
__global__ __launch_bounds__(64) void k_e8_chunks(int64_t B, int64_t N, int64_t K, int R, const double *__restrict__ t,
                                                  int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                  const double *__restrict__ a, const double *__restrict__ U,
                                                  const double *__restrict__ V, const double *__restrict__ yv,
                                                  double *__restrict__ elems) {
  const int lane = threadIdx.x, j = lane & 7;
  int64_t g = (int64_t)blockIdx.x * 8 + (lane >> 3);   // chunk of this group, all series flattened
  const bool inr = g < B * K;
  if (!inr) g = B * K - 1;
  const int64_t b = g / K, k = g - b * K, s = k * R;
  const int len = (int)((s + R < N ? s + R : N) - s);
  const double cj = c[b * c_bs + j];
  const double *tb = t + b * t_bs, *ab = a + b * N, *Ub = U + b * N * 8, *Vb = V + b * N * 8, *yb = VEC ? yv + b * N : nullptr;
  double AX[8], GX[8], QX[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { AX[q] = q == 0 ? 1.0 : 0.0; GX[q] = 0.0; QX[q] = 0.0; }
  double gj = 0.0, hj = 0.0, q0 = 0.0, prod = 1.0;
  int eacc = 0;
  bool failed = false;
  // the next row's inputs, one row ahead
  double uX[8], vj, an, yn = 0.0, tn = tb[s], tn1;
  auto fetch = [&](int r, double (&u)[8], double &v, double &av, double &y, double &t1) {
    const int64_t n = s + (r < len ? r : len - 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = Ub[n * 8 + (j ^ q)];
    v = Vb[n * 8 + j];
    av = ab[n];
    if (VEC) y = yb[n];
    t1 = tb[n + 1 < N ? n + 1 : n];
  };
  fetch(0, uX, vj, an, yn, tn1);
#pragma unroll 1
  for (int r = 0; r < len; ++r) {
    double u[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = uX[q];
    const double v = vj, av = an, y = yn, t1 = tn1;
    const int64_t n = s + r;
    fetch(r + 1, uX, vj, an, yn, tn1);
    double tau = 0.0, rr = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { tau = fma(u[q], GX[q], tau); rr = fma(u[q], AX[q], rr); }   // (u T)_j, (u A)_j
    double dsum = tau * u[0], zsum = VEC ? u[0] * gj : 0.0;
    if (VEC) gsum2<8>(dsum, zsum); else dsum = gsum<8>(dsum);
    const double d = av - dsum;                              // forward.hpp:127
    const double rd = rcp_nr(d);
    failed = failed || (n > 0 && !(d > 0.0));                // forward.hpp:128
    const double w = (v - tau) * rd;                         // forward.hpp:131
    prod *= d;
    if (r & 1) { int ex; prod = frexp(prod, &ex); eacc += ex; }
    double rX[8];
    xgather_dpp<8>(rr, nullptr, lane, rX);
    const double rjd = rr * rd;
#pragma unroll
    for (int q = 0; q < 8; ++q) QX[q] = fma(rX[q], rjd, QX[q]);   // Q += r^T r / d
    double z0 = 0.0;
    if (VEC) {
      z0 = y - zsum;                                         // internal.hpp:144
      const double z0d = z0 * rd;
      q0 = fma(z0, z0d, q0);
      hj = fma(z0d, rr, hj);
    }
    if (n + 1 < N) {   // on to row n + 1 (forward.hpp:115-123, internal.hpp:140-143)
      const double p = exp_decay(cj * (tn - t1));
      double wX[8], pX[8];
      xgather_dpp<8>(w, nullptr, lane, wX);
      xgather_dpp<8>(p, nullptr, lane, pX);
      const double dw = d * w;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        GX[q] = (pX[q] * p) * fma(wX[q], dw, GX[q]);         // P (T + d w^T w) P
        AX[q] = pX[q] * fma(-wX[q], rr, AX[q]);              // P (I - w^T u) A
      }
      if (VEC) gj = p * fma(w, z0, gj);
    }
    tn = t1;
  }
  if (!inr) return;
  double *rec = elems + (size_t)g * ElemIO<8>::REC;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = j ^ q;
    rec[kE8A + i * 8 + j] = AX[q];
    if (i <= j) { rec[kE8G + sidx(8, i, j)] = GX[q]; rec[kE8Q + sidx(8, i, j)] = QX[q]; }
  }
  rec[kE8g + j] = gj; rec[kE8h + j] = hj;
  if (j == 0) {
    int ex;
    const double pr = frexp(prod, &ex);
    rec[kE8q0] = q0;
    rec[kE8prod] = failed ? __longlong_as_double(0x7ff8000000000000ll) : pr;
    rec[kE8ex] = (double)(eacc + ex);
  }
}

// grid (ceil(Kin / span), B): workgroup x of series b reduces elements span x .. of `in` ([series][Kin] records) to ONE: written
// to `out` ([series][gridDim.x] records) or -- when it is the only workgroup of its series -- turned into ll (numpy.py:84-109).
// `scr`: e8_level_records(span) records per workgroup: the levels in between, back to back, and the workgroup's own final
// record behind them.  (The levels of a span that is not a power of two need MORE than span records -- 25 -> 13 + 7 + 4 + 2:
// rounds 5's `span` records per workgroup overflowed into the next workgroup's region / past the buffer for K = 9, 17, 25, ...)
__host__ __device__ inline int e8_level_records(int span) {
  int total = 1;   // the final record
  for (int n = span; n > 1; n = (n + 1) >> 1) total += (n + 1) >> 1;
  return total;
}
__global__ __launch_bounds__(kE8Waves * 64) void k_e8_tree(int64_t N, int64_t Kin, int span, const double *__restrict__ in,
                                                            double *__restrict__ out, double *__restrict__ scr,
                                                            double *__restrict__ ll, int32_t *__restrict__ flag,
                                                            unsigned long long *__restrict__ guard) {
  constexpr int REC = ElemIO<8>::REC;
  __shared__ E8Lds lds[kE8Waves];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = blockIdx.y, first = (int64_t)blockIdx.x * span;
  int n = (int)((Kin - first) < span ? (Kin - first) : span);
  const double *src = in + (size_t)(b * Kin + first) * REC;
  const int stride = e8_level_records(span);
  double *lvl = scr + ((size_t)b * gridDim.x + blockIdx.x) * (size_t)stride * REC;   // levels: ceil(span / 2), ceil(span / 4), ... records, back to back
  const bool last = gridDim.x == 1;
  double *dst_final = last ? lvl + (size_t)(stride - 1) * REC : out + (size_t)(b * gridDim.x + blockIdx.x) * REC;
  if (n == 1) {   // nothing to combine: hand the element on
    if (w == 0) for (int q = lane; q < REC; q += 64) dst_final[q] = src[q];
  }
  while (n > 1) {
    const int pairs = n >> 1, nn = (n + 1) >> 1;
    double *dst = nn == 1 ? dst_final : lvl;
    for (int p = w; p < pairs; p += kE8Waves)
      e8_combine(src + (size_t)(2 * p) * REC, src + (size_t)(2 * p + 1) * REC, dst + (size_t)p * REC, lds[w], lane);
    if ((n & 1) && w == kE8Waves - 1)
      for (int q = lane; q < REC; q += 64) dst[(size_t)pairs * REC + q] = src[(size_t)(n - 1) * REC + q];
    __threadfence_block();   // (producers and consumers of a level are wavefronts of this workgroup: no device-scope release)
    __syncthreads();
    src = dst; lvl += (size_t)nn * REC; n = nn;
  }
  if (last) {
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
      const double prod = dst_final[kE8prod], q0 = dst_final[kE8q0], ex = dst_final[kE8ex];
      const double logdet = log(prod) + ex * 0.693147180559945309417;
      if (ll) {   // (nullptr: the chunk-start states of `factor` -- matrices only, the vectors of the records are junk)
        ll[b] = -0.5 * (logdet + q0 + (double)N * 1.83787706640934548356);
        flag[b] = 0;
      }
      if (!(logdet == logdet) || (ll && !(q0 == q0))) {   // left to the row-by-row kernel behind the gate
        atomicMax(guard, (unsigned long long)__double_as_longlong(INFINITY));
        atomicMax(guard + 1, (unsigned long long)__double_as_longlong(INFINITY));
      }
    }
  }
}
// rows per chunk at width 8: as short as keeps the chunk wavefronts within one round of the chip
__host__ inline int chunk_rows8(int64_t B, int64_t N) {
  for (int R = 16; R < 64; R *= 2)
    if (B * ((N + R - 1) / R) <= 65536) return R;
  return 64;
}
// chunk pass at width 8 with the element spread over the lanes of a group (k_e8_chunks) or in one lane (k_tp_onepass<8, 2>)
static bool e8_group_chunks() { return !(opt::has(opt::k_e8_group_chunks) && opt::ival(opt::k_e8_group_chunks) == 0); }
struct E8Plan {
  int R, span;
  int64_t K, blocks;
  size_t rec0, rec1, scr, total;   // offsets (doubles): chunk elements, the workgroups' results (ping / pong), level scratch
};
inline E8Plan e8_plan(int64_t B, int64_t N) {
  E8Plan p;
  p.R = chunk_rows8(B, N);
  p.K = (N + p.R - 1) / p.R;
  p.span = B * ((p.K + 63) / 64) <= 256 ? 64 : 512;   // a handful of series: a workgroup per 64 elements (one per CU at most)
  if ((int64_t)p.span > p.K) p.span = p.K > 1 ? (int)p.K : 1;
  constexpr size_t REC = ElemIO<8>::REC;
  p.blocks = (p.K + p.span - 1) / p.span;
  p.rec0 = 0;
  p.rec1 = p.rec0 + (size_t)B * (size_t)p.K * REC;
  p.scr = p.rec1 + 2 * (size_t)B * (size_t)p.blocks * REC;
  p.total = p.scr + (size_t)B * (size_t)p.blocks * (size_t)e8_level_records(p.span) * REC;   // (later launches: fewer blocks, spans no longer)
  return p;
}
inline int run8(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *a,
                const double *U, const double *V, const double *y, double *ll, int32_t *flag, double *work,
                unsigned long long *guard, hipStream_t s) {
  const E8Plan p = e8_plan(B, N);
  constexpr size_t REC = ElemIO<8>::REC;
  if (hipMemsetAsync(guard, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return C2_ERR_HIP;
  const dim3 gc((unsigned)((p.K + kThreads - 1) / kThreads), (unsigned)B);
  if (e8_group_chunks())
    hipLaunchKernelGGL((k_e8_chunks<true>), dim3((unsigned)((B * p.K + 7) / 8)), dim3(64), 0, s, B, N, p.K, p.R, t, t_bs, c, c_bs, a, U,
                       V, y, work + p.rec0);
  else
    hipLaunchKernelGGL((k_tp_onepass<8, 2>), gc, dim3(kThreads), 0, s, B, N, p.K, p.R, t, t_bs, c, c_bs, a, U, V, y, work + p.rec0, ll,
                       flag, guard);
  const double *in = work + p.rec0;
  int64_t Kin = p.K;
  double *pong[2] = {work + p.rec1, work + p.rec1 + (size_t)B * (size_t)p.blocks * REC};
  for (int it = 0;; ++it) {   // (every launch needs at most the first one's blocks x span records of level scratch)
    const int span = Kin < (int64_t)p.span ? (int)Kin : p.span;
    const int64_t blocks = (Kin + span - 1) / span;
    hipLaunchKernelGGL(k_e8_tree, dim3((unsigned)blocks, (unsigned)B), dim3(kE8Waves * 64), 0, s, N, Kin, span, in, pong[it & 1],
                       work + p.scr, ll, flag, guard);
    if (blocks == 1) break;
    in = pong[it & 1];
    Kin = blocks;
  }
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// ---- width 8: the EXACT chunk-start states (what `factor`, `factor + S` and the factor stage of the time-parallel gradient
// start their rows from -- five to eight Newton iterations on those states before round 6).  The tree above is the up-sweep
// of a scan; its levels stay in the scratch, and a DOWN-sweep hands every node the state its span is entered with:
//     start(left child) = start(node),     start(right child) = apply(element(left child), start(node))
// with `apply` the element formula T' = G + A Gt A^T, Gt = T (I - Q T)^-1 = L Ks^-1 L^T (T = L L^T pivoted, Ks symmetric
// positive definite: as well-conditioned as the factorisation, exact to rounding whatever the span).  One wavefront per
// application, lane (i, j) <-> matrix entry, as e8_combine; states are packed upper triangles (36 doubles, sidx order).
constexpr int kE8NS = 36;
// To <- the state the span of record `r` leaves behind when entered with Tin (nullptr: the zero state).  false: Ks not
// positive definite (a pivot of the span is not positive when entered with Tin).
__device__ __forceinline__ bool e8_apply(const double *__restrict__ r, const double *__restrict__ Tin, double *__restrict__ To,
                                         E8Lds &m, int lane) {
  const int i = lane >> 3, j = lane & 7, sij = sym(8, i, j);
  auto mm = [&](const double *X, int xi, int xk, const double *Y, int yk, int yj) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) v = fma(X[i * xi + k * xk], Y[k * yk + j * yj], v);
    return v;
  };
  lds_order();
  m.A2[lane] = r[kE8A + lane]; m.G2[lane] = r[kE8G + sij]; m.Q2[lane] = r[kE8Q + sij];
  m.Z[lane] = Tin ? Tin[sij] : 0.0;
  double tol = 0.0;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {   // T = L L^T, outer-product Cholesky with diagonal pivoting (see `posterior`)
    lds_order();
    double sp = m.Z[0];
    int mi = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const double dq = m.Z[q * 9];
      const bool gt = dq > sp;
      sp = gt ? dq : sp;
      mi = gt ? q : mi;
    }
    if (k == 0) tol = 1e-15 * sp;
    const bool ok = sp > tol;
    const double inv = ok ? 1.0 / sqrt(sp) : 0.0;
    const double li = m.Z[i * 8 + mi] * inv, lj = m.Z[j * 8 + mi] * inv, own = m.Z[lane];
    lds_order();
    m.Z[lane] = fma(-li, lj, own);
    if (j == k) m.L[lane] = li;
  }
  lds_order();
  m.X[lane] = mm(m.Q2, 8, 1, m.L, 8, 1);                                  // X = Q L
  lds_order();
  m.Ks[lane] = (i == j ? 1.0 : 0.0) - mm(m.L, 1, 8, m.X, 8, 1);           // Ks = I - L^T X
  bool good = true;
#pragma unroll 1
  for (int k = 0; k < 8; ++k) {   // Ks <- Ks^-1 (Gauss-Jordan; the pivots are positive <=> the span's pivots are)
    lds_order();
    const double p = m.Ks[k * 9], rkj = m.Ks[k * 8 + j], cik = m.Ks[i * 8 + k], own = m.Ks[lane];
    good = good && p > 0.0;
    const double ip = 1.0 / p;
    lds_order();
    m.Ks[lane] = (i == k && j == k) ? ip : (i == k ? rkj * ip : (j == k ? -cik * ip : fma(-cik * ip, rkj, own)));
  }
  lds_order();
  m.Z[lane] = mm(m.Ks, 8, 1, m.L, 1, 8);                                  // Z = Ks^-1 L^T
  lds_order();
  m.Gt[lane] = mm(m.L, 8, 1, m.Z, 8, 1);                                  // Gt = L Z
  lds_order();
  m.X[lane] = mm(m.A2, 8, 1, m.Gt, 8, 1);                                 // A Gt
  lds_order();
  const double Tn = m.G2[lane] + mm(m.X, 8, 1, m.A2, 1, 8);               // T' = G + A Gt A^T
  if (i <= j) To[sij] = good ? Tn : __longlong_as_double(0x7ff8000000000000ll);
  return good;
}
// The down-sweep of ONE launch of k_e8_tree (same grid, same `in`, `span`, `scr`).  Tin: the start state of every workgroup's
// span ([series][gridDim.x] packed states; nullptr: zero -- the launch whose single workgroup covered the series); Tout: the
// start state of every input element ([series][Kin]); tscr: e8_level_records(span) packed states per workgroup.
__global__ __launch_bounds__(kE8Waves * 64) void k_e8_down(int64_t Kin, int span, const double *__restrict__ in,
                                                            const double *__restrict__ scr, const double *__restrict__ Tin,
                                                            double *__restrict__ Tout, double *__restrict__ tscr,
                                                            unsigned long long *__restrict__ guard) {
  constexpr int REC = ElemIO<8>::REC;
  __shared__ E8Lds lds[kE8Waves];
  __shared__ int n_of[12], off_of[12];   // nodes of level l (0: the inputs), offset of level l >= 1 in the level scratch
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = blockIdx.y, first = (int64_t)blockIdx.x * span, wg = b * gridDim.x + blockIdx.x;
  const int n0 = (int)((Kin - first) < span ? (Kin - first) : span);
  const int stride = e8_level_records(span);
  const double *src0 = in + (size_t)(b * Kin + first) * REC, *lvl = scr + (size_t)wg * (size_t)stride * REC;
  double *st = tscr + (size_t)wg * (size_t)stride * kE8NS, *out0 = Tout + (size_t)(b * Kin + first) * kE8NS;
  const double *tin = Tin ? Tin + (size_t)wg * kE8NS : nullptr;
  int L = 0;
  if (threadIdx.x == 0) {
    int n = n0, off = 0, l = 0;
    n_of[0] = n0; off_of[0] = 0;
    while (n > 1) { const int nn = (n + 1) >> 1; ++l; n_of[l] = nn; off_of[l] = off; off += nn; n = nn; }
    n_of[11] = l;
  }
  __syncthreads();
  L = n_of[11];
  bool good = true;
  if (L == 0) {   // one element: its start state is the workgroup's
    if (w == 0 && lane < kE8NS) out0[lane] = tin ? tin[lane] : 0.0;
    return;
  }
  // level l + 1 -> level l, from the top (one node, state tin) down to the inputs
  for (int l = L - 1; l >= 0; --l) {
    const int nl = n_of[l], np = n_of[l + 1];
    const double *el = l == 0 ? src0 : lvl + (size_t)off_of[l] * REC;              // elements of level l
    double *sl = l == 0 ? out0 : st + (size_t)off_of[l] * kE8NS;                   // their start states
    const double *sp = (l + 1 == L) ? tin : st + (size_t)off_of[l + 1] * kE8NS;    // the parents' (top: the workgroup's)
    for (int p = w; p < np; p += kE8Waves) {
      const double *ps = sp ? sp + (size_t)((l + 1 == L) ? 0 : p) * kE8NS : nullptr;
      if (lane < kE8NS) sl[(size_t)(2 * p) * kE8NS + lane] = ps ? ps[lane] : 0.0;
      if (2 * p + 1 < nl)
        good = e8_apply(el + (size_t)(2 * p) * REC, ps, sl + (size_t)(2 * p + 1) * kE8NS, lds[w], lane) && good;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (!good && lane == 0) atomicMax(guard, (unsigned long long)__double_as_longlong(INFINITY));
}
struct E8Launch { int64_t Kin, blocks; int span; size_t in, out, scr, tscr, T; };   // offsets in doubles
struct E8StatesPlan {
  int R, n;
  int64_t K;
  E8Launch l[8];
  size_t rec0, total;
};
// chunks of R rows (the caller's: the chunk length of the kernels that consume the states)
inline E8StatesPlan e8_states_plan(int64_t B, int64_t N, int R) {
  constexpr size_t REC = ElemIO<8>::REC;
  E8StatesPlan p;
  p.R = R; p.K = (N + R - 1) / R; p.n = 0;
  p.rec0 = 0;
  size_t at = (size_t)B * (size_t)p.K * REC;
  int64_t Kin = p.K;
  size_t in = p.rec0;
  for (;;) {
    E8Launch &q = p.l[p.n];
    q.Kin = Kin;
    q.span = B * ((Kin + 63) / 64) <= 256 ? 64 : 512;
    if ((int64_t)q.span > Kin) q.span = Kin > 1 ? (int)Kin : 1;
    q.blocks = (Kin + q.span - 1) / q.span;
    const size_t lv = (size_t)B * (size_t)q.blocks * (size_t)e8_level_records(q.span);
    q.in = in;
    q.out = at; at += (size_t)B * (size_t)q.blocks * REC;
    q.scr = at; at += lv * REC;
    q.tscr = at; at += lv * kE8NS;
    q.T = at; at += p.n == 0 ? 0 : (size_t)B * (size_t)Kin * kE8NS;   // (launch 0 writes the caller's array)
    ++p.n;
    if (q.blocks == 1 || p.n == 8) break;
    in = q.out; Kin = q.blocks;
  }
  p.total = at;
  return p;
}
// X[b][k][36] <- the state chunk k of series b is entered with (k = 0: zero).  `guard`: raised (+inf) when a pivot is not
// positive / something is not finite -- the caller's row-by-row kernel behind it.
inline int run8_states(int64_t B, int64_t N, int R, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                       const double *a, const double *U, const double *V, double *X, double *work, unsigned long long *guard,
                       hipStream_t s) {
  const E8StatesPlan p = e8_states_plan(B, N, R);
  if (p.l[p.n - 1].blocks != 1 || B > 65535) return C2_ERR_UNSUPPORTED;
  const dim3 gc((unsigned)((p.K + kThreads - 1) / kThreads), (unsigned)B);
  if (e8_group_chunks())
    hipLaunchKernelGGL((k_e8_chunks<false>), dim3((unsigned)((B * p.K + 7) / 8)), dim3(64), 0, s, B, N, p.K, R, t, t_bs, c, c_bs, a,
                       U, V, (const double *)nullptr, work + p.rec0);
  else   // (the records' vectors are computed from `a` in place of y: finite junk nobody reads)
    hipLaunchKernelGGL((k_tp_onepass<8, 2>), gc, dim3(kThreads), 0, s, B, N, p.K, R, t, t_bs, c, c_bs, a, U, V, a, work + p.rec0,
                       (double *)nullptr, (int32_t *)nullptr, guard);
  for (int i = 0; i < p.n; ++i) {
    const E8Launch &q = p.l[i];
    hipLaunchKernelGGL(k_e8_tree, dim3((unsigned)q.blocks, (unsigned)B), dim3(kE8Waves * 64), 0, s, N, q.Kin, q.span,
                       (const double *)(work + q.in), work + q.out, work + q.scr, (double *)nullptr, (int32_t *)nullptr, guard);
  }
  for (int i = p.n - 1; i >= 0; --i) {
    const E8Launch &q = p.l[i];
    const double *Tin = i == p.n - 1 ? nullptr : work + p.l[i + 1].T;
    hipLaunchKernelGGL(k_e8_down, dim3((unsigned)q.blocks, (unsigned)B), dim3(kE8Waves * 64), 0, s, q.Kin, q.span,
                       (const double *)(work + q.in), (const double *)(work + q.scr), Tin, i == 0 ? X : work + q.T,
                       work + q.tscr, guard);
  }
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// ---- `factor`: chunk-start states.  The matrices of the chunk elements (no right-hand side) and an inclusive scan inside the
// wavefront.  SINGLE (a series of at most 64 chunks): the start state of chunk k + 1 is the G of the prefix 0 .. k, written
// to `starts`; otherwise the prefixes go to `pref` and k_tp_carry / k_tp_long_starts finish the job.
template <int J, bool SINGLE>
__global__ __launch_bounds__(kThreads) void k_tp_states(int64_t B, int64_t N, int64_t K, int R, const double *__restrict__ t,
                                                        int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                        const double *__restrict__ a, const double *__restrict__ U,
                                                        const double *__restrict__ V, double *__restrict__ starts,
                                                        double *__restrict__ pref, unsigned long long *__restrict__ guard) {
  using Gm = Geo<J>;
  constexpr int NS = nsym(J);
  __shared__ __attribute__((aligned(16))) double lds[2 * 64 * Gm::RSTR + 2 * 64 * Gm::SSTR];
  double *tU = lds, *tV = tU + 64 * Gm::RSTR, *tA = tV + 64 * Gm::RSTR, *tT = tA + 64 * Gm::SSTR;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y, G = B * K;
  const Chunks ch{b * N, b * t_bs, N, K, (int64_t)blockIdx.x * kThreads, R};
  int64_t k = ch.k0 + lane;
  const bool inr = k < K;
  if (!inr) k = K - 1;
  const int64_t g = b * K + k;
  const int64_t s = k * R;
  const int len = ch.len(k);
  Elem<J> e;
  elem_identity<J>(e);
  double cj[J];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[b * c_bs + j];
  bool ceq = J % 2 == 0;
#pragma unroll
  for (int j = 0; j + 1 < J; j += 2) ceq = ceq && cj[j] == cj[j + 1];
  bool failed = false;
  double tn = t[b * t_bs + s];
  double vu[16], vv[16], va[16], vt[16];   // tiles requested one tile ahead (scalars: 16 rows, staged in halves)
  fetch_scalar_rows16<false>(a, ch, 0, 0, lane, va);
  fetch_scalar_rows16<true>(t, ch, 0, 1, lane, vt);   // t of the NEXT row
  fetch_row_tile<J>(U, ch, 0, lane, vu);
  fetch_row_tile<J>(V, ch, 0, lane, vv);
  for (int r0 = 0; r0 < R; r0 += 8) {
    lds_order();
    stage_scalar_half(tA, lane, va, (r0 >> 3) & 1);
    stage_scalar_half(tT, lane, vt, (r0 >> 3) & 1);
    if ((r0 & 8) && r0 + 8 < R) {
      fetch_scalar_rows16<false>(a, ch, r0 + 8, 0, lane, va);
      fetch_scalar_rows16<true>(t, ch, r0 + 8, 1, lane, vt);
    }
#pragma unroll 1
    for (int rt = 0; rt < 8; rt += Gm::RT) {
      lds_order();
      stage_row_tile<J>(tU, lane, vu);
      stage_row_tile<J>(tV, lane, vv);
      if (r0 + rt + Gm::RT < R) {
        fetch_row_tile<J>(U, ch, r0 + rt + Gm::RT, lane, vu);
        fetch_row_tile<J>(V, ch, r0 + rt + Gm::RT, lane, vv);
      }
      lds_order();
#pragma unroll
      for (int r = 0; r < Gm::RT; ++r) {
        const int i0 = r0 + rt + r;
        const int64_t n = s + i0;
        if (i0 < len) {
          double u[J], v[J], tau[J], rr[J], w[J];
#pragma unroll
          for (int j = 0; j < J; ++j) {
            u[j] = tU[lane * Gm::RSTR + r * J + j]; v[j] = tV[lane * Gm::RSTR + r * J + j];
            tau[j] = 0.0; rr[j] = 0.0;
          }
#pragma unroll
          for (int i = 0; i < J; ++i)
#pragma unroll
            for (int j = i; j < J; ++j) {
              const double sv = e.G[sidx(J, i, j)];
              tau[j] = fma(u[i], sv, tau[j]);
              if (j != i) tau[i] = fma(u[j], sv, tau[i]);
            }
          double d = tA[lane * Gm::SSTR + rt + r];
#pragma unroll
          for (int j = 0; j < J; ++j) {
            d = fma(-tau[j], u[j], d);          // forward.hpp:127
#pragma unroll
            for (int i = 0; i < J; ++i) rr[j] = fma(u[i], e.A[i][j], rr[j]);
          }
          const double rd = rcp_nr(d);
          failed = failed || (n > 0 && !(d > 0.0));   // forward.hpp:128 (the first row's d is not checked there)
#pragma unroll
          for (int j = 0; j < J; ++j) w[j] = (v[j] - tau[j]) * rd;   // forward.hpp:131
#pragma unroll
          for (int i = 0; i < J; ++i) {
            const double rid = rr[i] * rd;
#pragma unroll
            for (int j = i; j < J; ++j) e.Q[sidx(J, i, j)] = fma(rid, rr[j], e.Q[sidx(J, i, j)]);
          }
          if (n + 1 < N) {   // on to row n + 1 (forward.hpp:115-123)
            const double tn1 = tT[lane * Gm::SSTR + rt + r];
            double p[J];
            if (ceq) {   // (complex terms: the rates come in pairs)
#pragma unroll
              for (int j = 0; j + 1 < J; j += 2) { p[j] = exp_decay(cj[j] * (tn - tn1)); p[j + 1] = p[j]; }
            } else {
#pragma unroll
              for (int j = 0; j < J; ++j) p[j] = exp_decay(cj[j] * (tn - tn1));
            }
            tn = tn1;
#pragma unroll
            for (int i = 0; i < J; ++i) {
              const double dwi = d * w[i];
#pragma unroll
              for (int j = i; j < J; ++j) e.G[sidx(J, i, j)] = (p[i] * p[j]) * fma(dwi, w[j], e.G[sidx(J, i, j)]);
#pragma unroll
              for (int j = 0; j < J; ++j) e.A[i][j] = p[i] * fma(-w[i], rr[j], e.A[i][j]);
            }
          }
        }
      }
    }
  }
  if (!inr) { elem_identity<J>(e); failed = false; }   // (lanes beyond the series ran its last chunk for the loads' sake)
  elem_scan<J>(e, lane, (int)((K - ch.k0) < kThreads ? (K - ch.k0) : kThreads));
  if (failed || !(e.prod == e.prod)) atomicMax(guard, (unsigned long long)__double_as_longlong(INFINITY));
  if constexpr (SINGLE) {
    if (lane < NS) starts[(int64_t)lane * G + b * K] = 0.0;   // chunk 0 starts from nothing (forward.hpp:107-113)
    if (inr && k + 1 < K) {
#pragma unroll
      for (int q = 0; q < NS; ++q) starts[(int64_t)q * G + g + 1] = e.G[q];
    }
  } else if (inr) {
    elem_store_matrices<J>(e, pref, G, g);
  }
}
// long series (more than 64 chunks): the state every WAVEFRONT of k_tp_states starts from.  One wavefront per series, lane <->
// wavefront of the series, 64 a round: the totals (the prefix of a wavefront's last chunk) scanned, applied to the state
// the round starts from.
template <int J>
__global__ __launch_bounds__(kThreads) void k_tp_carry(int64_t B, int64_t K, int64_t Wn, const double *__restrict__ pref,
                                                       double *__restrict__ tin, unsigned long long *__restrict__ guard) {
  constexpr int NS = nsym(J);
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x, G = B * K, W = B * Wn;
  double T[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) T[q] = 0.0;
  if (lane < NS) tin[(int64_t)lane * W + b * Wn] = 0.0;
  bool bad = false;
  for (int64_t base = 0; base < Wn; base += kThreads) {
    const int64_t w = base + lane;
    Elem<J> e;
    elem_identity<J>(e);
    if (w < Wn) {
      const int64_t last = ((w + 1) * kThreads < K ? (w + 1) * kThreads : K) - 1;
      elem_load_matrices<J>(e, pref, G, b * K + last);
    }
    elem_scan<J>(e, lane, (int)((Wn - base) < kThreads ? (Wn - base) : kThreads));
    double Tn[NS];
    const bool good = elem_apply<J>(e, T, Tn);
    if (w < Wn) bad = bad || !good || !(e.prod == e.prod);
    if (w + 1 < Wn) {
#pragma unroll
      for (int q = 0; q < NS; ++q) tin[(int64_t)q * W + b * Wn + w + 1] = Tn[q];
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) T[q] = __shfl(Tn[q], 63, 64);
  }
  if (bad) atomicMax(guard, (unsigned long long)__double_as_longlong(INFINITY));
}
// ... and the start state of every chunk: its wavefront's, pushed through the prefix of the chunks before it
template <int J>
__global__ __launch_bounds__(kThreads) void k_tp_long_starts(int64_t B, int64_t K, int64_t Wn, const double *__restrict__ pref,
                                                             const double *__restrict__ tin, double *__restrict__ starts,
                                                             unsigned long long *__restrict__ guard) {
  constexpr int NS = nsym(J);
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y, G = B * K, W = B * Wn;
  const int64_t k = (int64_t)blockIdx.x * kThreads + lane;
  if (k >= K) return;
  double T[NS], Tn[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) { T[q] = tin[(int64_t)q * W + b * Wn + blockIdx.x]; Tn[q] = T[q]; }
  if (lane > 0) {
    Elem<J> e;
    elem_load_matrices<J>(e, pref, G, b * K + k - 1);
    if (!elem_apply<J>(e, T, Tn)) atomicMax(guard, (unsigned long long)__double_as_longlong(INFINITY));
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) starts[(int64_t)q * G + b * K + k] = Tn[q];
}

// ---- `factor` itself (forward.hpp:69-135: d, W, flag): every chunk runs the recursion from its start state and writes its rows.
// d == a / W == V in place are NOT supported here (the caller keeps those on the row-by-row kernel: its fallback would read
// what this kernel overwrote).
template <int J>
__global__ __launch_bounds__(kThreads) void k_tp_factor(int64_t B, int64_t N, int64_t K, int R, const double *__restrict__ t,
                                                        int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                        const double *__restrict__ a, const double *__restrict__ U,
                                                        const double *__restrict__ V, const double *__restrict__ starts,
                                                        double *__restrict__ d_out,
                                                        double *__restrict__ W_out, int32_t *__restrict__ flag,
                                                        unsigned long long *__restrict__ guard) {
  using Gm = Geo<J>;
  constexpr int NS = nsym(J);
  __shared__ __attribute__((aligned(16))) double lds[3 * 64 * Gm::RSTR + 3 * 64 * Gm::SSTR];
  double *tU = lds, *tV = tU + 64 * Gm::RSTR, *tW = tV + 64 * Gm::RSTR, *tA = tW + 64 * Gm::RSTR, *tT = tA + 64 * Gm::SSTR,
         *tD = tT + 64 * Gm::SSTR;
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y, G = B * K;
  const Chunks ch{b * N, b * t_bs, N, K, (int64_t)blockIdx.x * kThreads, R};
  int64_t k = ch.k0 + lane;
  const bool inr = k < K;
  if (!inr) k = K - 1;
  const int64_t g = b * K + k;
  const int64_t s = k * R;
  const int len = ch.len(k);
  double cj[J], S[NS];
#pragma unroll
  for (int j = 0; j < J; ++j) cj[j] = c[b * c_bs + j];
#pragma unroll
  for (int q = 0; q < NS; ++q) S[q] = starts[(int64_t)q * G + g];
  if (blockIdx.x == 0 && lane == 0) flag[b] = 0;
  double failed = 0.0;
  double tn = t[b * t_bs + s];
  double vu[16], vv[16], va[8], vt[8];   // tiles requested one tile ahead
  fetch_scalar_tile<false>(a, ch, 0, 0, lane, va);
  fetch_scalar_tile<true>(t, ch, 0, 1, lane, vt);   // t of the NEXT row
  fetch_row_tile<J>(U, ch, 0, lane, vu);
  fetch_row_tile<J>(V, ch, 0, lane, vv);
  for (int r0 = 0; r0 < R; r0 += 8) {
    lds_order();
    stage_scalar_tile(tA, lane, va);
    stage_scalar_tile(tT, lane, vt);
    if (r0 + 8 < R) {
      fetch_scalar_tile<false>(a, ch, r0 + 8, 0, lane, va);
      fetch_scalar_tile<true>(t, ch, r0 + 8, 1, lane, vt);
    }
#pragma unroll 1
    for (int rt = 0; rt < 8; rt += Gm::RT) {
      lds_order();
      stage_row_tile<J>(tU, lane, vu);
      stage_row_tile<J>(tV, lane, vv);
      if (r0 + rt + Gm::RT < R) {
        fetch_row_tile<J>(U, ch, r0 + rt + Gm::RT, lane, vu);
        fetch_row_tile<J>(V, ch, r0 + rt + Gm::RT, lane, vv);
      }
      lds_order();
#pragma unroll
      for (int r = 0; r < Gm::RT; ++r) {
        const int i0 = r0 + rt + r;
        const int64_t n = s + i0;
        double u[J], v[J], tau[J], w[J];
#pragma unroll
        for (int j = 0; j < J; ++j) { u[j] = tU[lane * Gm::RSTR + r * J + j]; v[j] = tV[lane * Gm::RSTR + r * J + j]; tau[j] = 0.0; }
#pragma unroll
        for (int i = 0; i < J; ++i)
#pragma unroll
          for (int j = i; j < J; ++j) {
            const double sv = S[sidx(J, i, j)];
            tau[j] = fma(u[i], sv, tau[j]);
            if (j != i) tau[i] = fma(u[j], sv, tau[i]);
          }
        double d = tA[lane * Gm::SSTR + rt + r];
#pragma unroll
        for (int j = 0; j < J; ++j) d = fma(-tau[j], u[j], d);          // forward.hpp:127
        const double rd = rcp_nr(d);
        if (i0 < len) failed = (failed == 0.0 && n > 0 && !(d > 0.0)) ? (double)n : failed;   // forward.hpp:128
#pragma unroll
        for (int j = 0; j < J; ++j) {   // forward.hpp:131
          w[j] = (v[j] - tau[j]) * rd;
          tW[lane * Gm::RSTR + r * J + j] = w[j];
        }
        tD[lane * Gm::SSTR + rt + r] = d;
        if (i0 < len && n + 1 < N) {   // on to row n + 1 (forward.hpp:115-123)
          const double tn1 = tT[lane * Gm::SSTR + rt + r];
          double p[J];
#pragma unroll
          for (int j = 0; j < J; ++j) p[j] = exp_decay(cj[j] * (tn - tn1));
          tn = tn1;
#pragma unroll
          for (int i = 0; i < J; ++i) {
            const double dwi = d * w[i];
#pragma unroll
            for (int j = i; j < J; ++j) S[sidx(J, i, j)] = (p[i] * p[j]) * fma(dwi, w[j], S[sidx(J, i, j)]);
          }
        }
      }
      lds_order();
      flush_row_tile<J>(W_out, ch, r0 + rt, lane, tW);
    }
    lds_order();
    flush_scalar_tile(d_out, ch, r0, lane, tD);
  }
  // verification: this chunk's end state against the start state its successor was given; failures -> the row-by-row kernel
  double worst = 0.0;
  if (inr && k + 1 < K) worst = start_mismatch<J>(S, starts, G, g + 1);
  if (inr && failed != 0.0) worst = INFINITY;
  if (!(worst == worst)) worst = INFINITY;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) worst = fmax(worst, __shfl_xor(worst, o, 64));
  const double gval = worst / kTol;
  if (lane == 0 && gval > 0.0) atomicMax(guard, (unsigned long long)__double_as_longlong(gval));
}


template <int J>
int run(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *a,
        const double *U, const double *V, const double *y, double *ll, int32_t *flag, double *work,
        unsigned long long *guard, hipStream_t s) {
  const int R = chunk_rows(N);
  const int64_t K = (N + R - 1) / R;
  const dim3 gc((unsigned)((K + kThreads - 1) / kThreads), (unsigned)B), gs((unsigned)B);   // lane <-> chunk
  if (hipMemsetAsync(guard, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return C2_ERR_HIP;
  if (K <= kThreads) {
    hipLaunchKernelGGL((k_tp_onepass<J, 0>), gc, dim3(kThreads), 0, s, B, N, K, R, t, t_bs, c, c_bs, a, U, V, y, work, ll, flag,
                       guard);
  } else {
    hipLaunchKernelGGL((k_tp_onepass<J, 1>), gc, dim3(kThreads), 0, s, B, N, K, R, t, t_bs, c, c_bs, a, U, V, y, work, ll, flag,
                       guard);
    hipLaunchKernelGGL((k_tp_join<J>), gs, dim3(kThreads), 0, s, B, N, (int64_t)gc.x, (const double *)work, ll, flag, guard);
  }
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// scratch of `factor` (doubles): the chunk-start states; for series of more than 64 chunks also the prefix elements and the
// wavefronts' start states
template <int J>
size_t factor_doubles(int64_t B, int64_t N) {
  const size_t R = (size_t)chunk_rows(N), K = ((size_t)N + R - 1) / R, G = (size_t)B * K, Wn = (K + kThreads - 1) / kThreads;
  return (size_t)nsym(J) * G + (Wn > 1 ? (size_t)ElemIO<J>::NM * G + (size_t)nsym(J) * (size_t)B * Wn : 0);
}
template <int J>
int run_factor(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *a,
               const double *U, const double *V, double *d, double *W, int32_t *flag, double *work,
               unsigned long long *guard, hipStream_t s) {
  const int R = chunk_rows(N);
  const int64_t K = (N + R - 1) / R, G = B * K, Wn = (K + kThreads - 1) / kThreads;
  double *starts = work, *pref = starts + (size_t)nsym(J) * G, *tin = pref + (size_t)ElemIO<J>::NM * G;
  const dim3 gc((unsigned)Wn, (unsigned)B), gs((unsigned)B);
  if (hipMemsetAsync(guard, 0, 2 * sizeof(unsigned long long), s) != hipSuccess) return C2_ERR_HIP;
  if (Wn <= 1) {
    hipLaunchKernelGGL((k_tp_states<J, true>), gc, dim3(kThreads), 0, s, B, N, K, R, t, t_bs, c, c_bs, a, U, V, starts,
                       (double *)nullptr, guard);
  } else {
    hipLaunchKernelGGL((k_tp_states<J, false>), gc, dim3(kThreads), 0, s, B, N, K, R, t, t_bs, c, c_bs, a, U, V, starts, pref,
                       guard);
    hipLaunchKernelGGL((k_tp_carry<J>), gs, dim3(kThreads), 0, s, B, K, Wn, (const double *)pref, tin, guard);
    hipLaunchKernelGGL((k_tp_long_starts<J>), gc, dim3(kThreads), 0, s, B, K, Wn, (const double *)pref, (const double *)tin,
                       starts, guard);
  }
  hipLaunchKernelGGL((k_tp_factor<J>), gc, dim3(kThreads), 0, s, B, N, K, R, t, t_bs, c, c_bs, a, U, V, (const double *)starts, d,
                     W, flag, guard);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}


// =============================================================================================================
// solve_lower / solve_upper (forward.hpp:156-207 over internal.hpp:105-189) with ONE right-hand side, parallel along
// time.  In virtual order s = 0 .. N-1 (lower: row s; upper: row N-1-s) with A the row fed into the state and B the row
// applied to it (lower: A = W, B = U; upper: A = U, B = W):
//     z_s = y_s - B_s . F_s ,     F_{s+1} = P_{s+1} (F_s + A_s z_s) = P_{s+1} ((I - A_s B_s^T) F_s + A_s y_s)
// -- affine in F, with contracting maps: a chunk yields (G, g) with F_end = G F_start + g, the chain over the chunks is
// J^2 flops a step, and a second pass applies the recursion from the true start states.  Nothing to verify (no
// ill-conditioned composition: the maps are applied, never inverted).  Z may alias Y.
template <bool LOWER>
struct SweepIO {
  int64_t sbase, tbase, N, K, k0;
  __device__ __forceinline__ int64_t chunk(int i, int lane) const {
    const int64_t k = k0 + 8 * i + lane / 8;
    return k < K ? k : K - 1;
  }
  __device__ __forceinline__ int len(int64_t k) const {
    const int64_t s = k * kRows;
    return (int)((s + kRows < N ? s + kRows : N) - s);
  }
  __device__ __forceinline__ int64_t row(int64_t s) const { return LOWER ? s : N - 1 - s; }   // virtual -> actual
};
template <int J, bool LOWER>
__device__ __forceinline__ void sw_fetch_rows(const double *__restrict__ base, const SweepIO<LOWER> &c, int r0, int lane,
                                              double (&v)[16]) {
  using Gm = Geo<J>;
  const int q = lane & 7;
  const int r = LOWER ? q / Gm::PPR : Gm::RT - 1 - q / Gm::PPR;   // local row of the tile this piece belongs to
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t k = c.chunk(i, lane);
    const int ln = c.len(k);
    int rr = r0 + r;
    rr = rr < ln ? rr : ln - 1;
    const double2 w = *reinterpret_cast<const double2 *>(base + (c.sbase + c.row(k * kRows + rr)) * J + 2 * (q % Gm::PPR));
    v[2 * i] = w.x; v[2 * i + 1] = w.y;
  }
}
template <int J, bool LOWER>
__device__ __forceinline__ void sw_stage_rows(double *tile, int lane, const double (&v)[16]) {
  using Gm = Geo<J>;
  const int q = lane & 7;
  const int r = LOWER ? q / Gm::PPR : Gm::RT - 1 - q / Gm::PPR;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    *reinterpret_cast<double2 *>(tile + (8 * i + lane / 8) * Gm::RSTR + r * J + 2 * (q % Gm::PPR)) =
        make_double2(v[2 * i], v[2 * i + 1]);
}
template <bool LOWER, bool TIME>
__device__ __forceinline__ void sw_fetch_scalars(const double *__restrict__ base, const SweepIO<LOWER> &c, int r0, int shift,
                                                 int lane, double (&v)[8]) {
  const int q = lane & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int64_t sv = c.chunk(i, lane) * kRows + r0 + q + shift;
    sv = sv < c.N - 1 ? sv : c.N - 1;
    v[i] = base[(TIME ? c.tbase : c.sbase) + c.row(sv)];
  }
}
template <bool LOWER>
__device__ __forceinline__ void sw_flush_scalars(double *__restrict__ base, const SweepIO<LOWER> &c, int r0, int lane,
                                                 const double *tile) {
  const int q = lane & 7;
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = tile[(8 * i + lane / 8) * 9 + q];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t k = c.k0 + 8 * i + lane / 8;
    if (k < c.K && r0 + q < c.len(k)) base[c.sbase + c.row(k * kRows + r0 + q)] = v[i];
  }
}

// One pass over the rows of this lane's chunk.  APPLY = false: the chunk's affine map (G, g).  APPLY = true: z from the
// start state F, written through an LDS tile.
template <int J, bool LOWER, bool APPLY>
__device__ __forceinline__ void sweep_pass(const SweepIO<LOWER> &io, int lane, int len, double tstart, const double (&cj)[J],
                                           const double *__restrict__ t, const double *__restrict__ A,
                                           const double *__restrict__ Bm, const double *__restrict__ Y, double *Z,
                                           double *lds, double (&G)[J][J], double (&g)[J], double (&F)[J]) {
  using Gm = Geo<J>;
  double *tA = lds, *tB = tA + 64 * Gm::RSTR, *tY = tB + 64 * Gm::RSTR, *tT = tY + 64 * Gm::SSTR, *tZ = tT + 64 * Gm::SSTR;
  double tn = tstart;
  double va[16], vb[16], vy[8], vt[8];
  sw_fetch_scalars<LOWER, false>(Y, io, 0, 0, lane, vy);
  sw_fetch_scalars<LOWER, true>(t, io, 0, 1, lane, vt);   // t of the NEXT (virtual) row
  sw_fetch_rows<J, LOWER>(A, io, 0, lane, va);
  sw_fetch_rows<J, LOWER>(Bm, io, 0, lane, vb);
  for (int r0 = 0; r0 < kRows; r0 += 8) {
    lds_order();
    stage_scalar_tile(tY, lane, vy);
    stage_scalar_tile(tT, lane, vt);
    if (r0 + 8 < kRows) {
      sw_fetch_scalars<LOWER, false>(Y, io, r0 + 8, 0, lane, vy);
      sw_fetch_scalars<LOWER, true>(t, io, r0 + 8, 1, lane, vt);
    }
#pragma unroll 1
    for (int rt = 0; rt < 8; rt += Gm::RT) {
      lds_order();
      sw_stage_rows<J, LOWER>(tA, lane, va);
      sw_stage_rows<J, LOWER>(tB, lane, vb);
      if (r0 + rt + Gm::RT < kRows) {
        sw_fetch_rows<J, LOWER>(A, io, r0 + rt + Gm::RT, lane, va);
        sw_fetch_rows<J, LOWER>(Bm, io, r0 + rt + Gm::RT, lane, vb);
      }
      lds_order();
#pragma unroll
      for (int r = 0; r < Gm::RT; ++r) {
        const int i0 = r0 + rt + r;
        if (i0 < len) {
          double av[J], bv[J];
#pragma unroll
          for (int j = 0; j < J; ++j) { av[j] = tA[lane * Gm::RSTR + r * J + j]; bv[j] = tB[lane * Gm::RSTR + r * J + j]; }
          const double tn1 = tT[lane * Gm::SSTR + rt + r];
          double p[J];
#pragma unroll
          for (int j = 0; j < J; ++j) p[j] = exp_decay(-cj[j] * fabs(tn1 - tn));   // internal.hpp:139 / 182
          tn = tn1;
          const double yn = tY[lane * Gm::SSTR + rt + r];
          if constexpr (APPLY) {
            double z = yn;
#pragma unroll
            for (int j = 0; j < J; ++j) z = fma(-bv[j], F[j], z);      // update_z (internal.hpp:144 / 187)
            tZ[lane * Gm::SSTR + rt + r] = z;
#pragma unroll
            for (int j = 0; j < J; ++j) F[j] = p[j] * fma(av[j], z, F[j]);   // update_f, decay (internal.hpp:140-143)
          } else {
            double bG[J], z0 = yn;
#pragma unroll
            for (int j = 0; j < J; ++j) {
              z0 = fma(-bv[j], g[j], z0);
              double sj = 0.0;
#pragma unroll
              for (int i = 0; i < J; ++i) sj = fma(bv[i], G[i][j], sj);
              bG[j] = sj;
            }
#pragma unroll
            for (int i = 0; i < J; ++i) {
#pragma unroll
              for (int j = 0; j < J; ++j) G[i][j] = p[i] * fma(-av[i], bG[j], G[i][j]);
              g[i] = p[i] * fma(av[i], z0, g[i]);
            }
          }
        }
      }
    }
    if constexpr (APPLY) {
      lds_order();
      sw_flush_scalars<LOWER>(Z, io, r0, lane, tZ);
    }
  }
}
// chain over the chunks of a wavefront: F_start of every lane's chunk from the uniform F at entry; F at exit = after them
template <int J>
__device__ __forceinline__ void chain_affine(const double (&G)[J][J], const double (&g)[J], int steps, int lane, double (&F)[J],
                                             double (&mine)[J]) {
#pragma unroll
  for (int j = 0; j < J; ++j) mine[j] = F[j];   // lane 0's chunk starts from the state at entry
  for (int turn = 0; turn < steps; ++turn) {
    double Fn[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
      double gi = g[i];
#pragma unroll
      for (int j = 0; j < J; ++j) gi = fma(G[i][j], F[j], gi);
      Fn[i] = gi;
    }
#pragma unroll
    for (int i = 0; i < J; ++i) {
      F[i] = __shfl(Fn[i], turn, 64);
      mine[i] = lane == turn + 1 ? F[i] : mine[i];
    }
  }
}
constexpr int sweep_lds_doubles(int J) { return 2 * 64 * 18 + 3 * 64 * 9; }

// K <= 64: everything in one kernel (maps, chain in the wavefront, apply)
template <int J, bool LOWER>
__global__ __launch_bounds__(kThreads) void k_tps_fused(int64_t B, int64_t N, int64_t K, const double *__restrict__ t,
                                                        int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                        const double *__restrict__ A, const double *__restrict__ Bm,
                                                        const double *__restrict__ Y, double *Z) {
  __shared__ __attribute__((aligned(16))) double lds[sweep_lds_doubles(J)];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y;
  const SweepIO<LOWER> io{b * N, b * t_bs, N, K, 0};
  int64_t k = lane;
  const bool inr = k < K;
  if (!inr) k = K - 1;
  const int len = io.len(k);
  double cj[J], G[J][J], g[J], F[J], mine[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { cj[j] = c[b * c_bs + j]; g[j] = 0.0; F[j] = 0.0; }
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) G[i][j] = i == j ? 1.0 : 0.0;
  const double tstart = t[b * t_bs + io.row(k * kRows)];
  sweep_pass<J, LOWER, false>(io, lane, len, tstart, cj, t, A, Bm, Y, Z, lds, G, g, F);
  chain_affine<J>(G, g, (int)K - 1, lane, F, mine);
  sweep_pass<J, LOWER, true>(io, lane, inr ? len : 0, tstart, cj, t, A, Bm, Y, Z, lds, G, g, mine);
}
// K > 64: maps -> scratch, chain (one wavefront per series), apply
template <int J, bool LOWER>
__global__ __launch_bounds__(kThreads) void k_tps_maps(int64_t B, int64_t N, int64_t K, const double *__restrict__ t,
                                                       int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                       const double *__restrict__ A, const double *__restrict__ Bm,
                                                       const double *__restrict__ Y, double *__restrict__ scratch) {
  __shared__ __attribute__((aligned(16))) double lds[sweep_lds_doubles(J)];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y, Gn = B * K;
  const SweepIO<LOWER> io{b * N, b * t_bs, N, K, (int64_t)blockIdx.x * kThreads};
  int64_t k = io.k0 + lane;
  const bool inr = k < K;
  if (!inr) k = K - 1;
  double cj[J], G[J][J], g[J], F[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { cj[j] = c[b * c_bs + j]; g[j] = 0.0; F[j] = 0.0; }
#pragma unroll
  for (int i = 0; i < J; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) G[i][j] = i == j ? 1.0 : 0.0;
  sweep_pass<J, LOWER, false>(io, lane, io.len(k), t[b * t_bs + io.row(k * kRows)], cj, t, A, Bm, Y, nullptr, lds, G, g, F);
  if (!inr) return;
  const int64_t gi = b * K + k;
#pragma unroll
  for (int i = 0; i < J; ++i) {
    scratch[(int64_t)(J * J + i) * Gn + gi] = g[i];
#pragma unroll
    for (int j = 0; j < J; ++j) scratch[(int64_t)(i * J + j) * Gn + gi] = G[i][j];
  }
}
template <int J>
__global__ __launch_bounds__(kThreads) void k_tps_chain(int64_t B, int64_t K, double *__restrict__ scratch) {
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x, Gn = B * K;
  double F[J];
#pragma unroll
  for (int j = 0; j < J; ++j) F[j] = 0.0;
  for (int64_t base = 0; base < K; base += kThreads) {
    const int64_t kk = base + lane;
    const bool have = kk < K;
    const int64_t gi = b * K + (have ? kk : K - 1);
    double G[J][J], g[J], mine[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
      g[i] = scratch[(int64_t)(J * J + i) * Gn + gi];
#pragma unroll
      for (int j = 0; j < J; ++j) G[i][j] = scratch[(int64_t)(i * J + j) * Gn + gi];
    }
    const int steps = (int)((K - base) < kThreads ? (K - base) : kThreads);
    chain_affine<J>(G, g, steps, lane, F, mine);   // F at exit: the state after this group's last chunk
    if (have) {
#pragma unroll
      for (int j = 0; j < J; ++j) scratch[(int64_t)(J * J + J + j) * Gn + gi] = mine[j];
    }
  }
}
template <int J, bool LOWER>
__global__ __launch_bounds__(kThreads) void k_tps_apply(int64_t B, int64_t N, int64_t K, const double *__restrict__ t,
                                                        int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                        const double *__restrict__ A, const double *__restrict__ Bm,
                                                        const double *__restrict__ Y, double *Z,
                                                        const double *__restrict__ scratch) {
  __shared__ __attribute__((aligned(16))) double lds[sweep_lds_doubles(J)];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y, Gn = B * K;
  const SweepIO<LOWER> io{b * N, b * t_bs, N, K, (int64_t)blockIdx.x * kThreads};
  int64_t k = io.k0 + lane;
  const bool inr = k < K;
  if (!inr) k = K - 1;
  double cj[J], G[J][J], g[J], F[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { cj[j] = c[b * c_bs + j]; g[j] = 0.0; F[j] = scratch[(int64_t)(J * J + J + j) * Gn + b * K + k]; }
  sweep_pass<J, LOWER, true>(io, lane, inr ? io.len(k) : 0, t[b * t_bs + io.row(k * kRows)], cj, t, A, Bm, Y, Z, lds, G, g, F);
}

template <int J, bool LOWER>
int run_solve(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs, const double *A,
              const double *Bm, const double *Y, double *Z, double *scratch, hipStream_t s) {
  const int64_t K = (N + kRows - 1) / kRows;
  const dim3 gc((unsigned)((K + kThreads - 1) / kThreads), (unsigned)B);
  if (K <= kThreads) {
    hipLaunchKernelGGL((k_tps_fused<J, LOWER>), dim3(1, (unsigned)B), dim3(kThreads), 0, s, B, N, K, t, t_bs, c, c_bs, A, Bm,
                       Y, Z);
  } else {
    hipLaunchKernelGGL((k_tps_maps<J, LOWER>), gc, dim3(kThreads), 0, s, B, N, K, t, t_bs, c, c_bs, A, Bm, Y, scratch);
    hipLaunchKernelGGL((k_tps_chain<J>), dim3((unsigned)B), dim3(kThreads), 0, s, B, K, scratch);
    hipLaunchKernelGGL((k_tps_apply<J, LOWER>), gc, dim3(kThreads), 0, s, B, N, K, t, t_bs, c, c_bs, A, Bm, Y, Z,
                       (const double *)scratch);
  }
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

}  // namespace c2tp

extern "C" {

// doubles of scratch the time-parallel `factor` needs (0: shape not covered)
size_t c2_internal_timepar_doubles(int64_t B, int64_t N, int64_t J) {
  return J == 4 ? c2tp::factor_doubles<4>(B, N) : (J == 2 ? c2tp::factor_doubles<2>(B, N) : 0);
}
// ... and the forward log-likelihood: one element per wavefront of a series longer than 4096 rows
size_t c2_internal_loglik_timepar_doubles(int64_t B, int64_t N, int64_t J) {
  if (J == 8) return c2tp::e8_plan(B, N).total;
  if (J != 4 && J != 2) return 0;
  const size_t R = (size_t)c2tp::chunk_rows(N), K = ((size_t)N + R - 1) / R, gx = (K + c2tp::kThreads - 1) / c2tp::kThreads;
  return gx <= 1 ? 2 : (size_t)(J == 4 ? c2tp::ElemIO<4>::N_ : c2tp::ElemIO<2>::N_) * (size_t)B * gx;
}
// Width 8: exact chunk-start states for chunks of R rows (run8_states); scratch in doubles; X: (B, K, 36) packed upper triangles.
size_t c2_internal_e8_states_doubles(int64_t B, int64_t N, int64_t R) {
  return c2tp::e8_states_plan(B, N, (int)R).total;
}
int c2_internal_e8_states(int64_t B, int64_t N, int64_t R, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                          const double *a, const double *U, const double *V, double *X, double *work,
                          unsigned long long *guard, c2_stream_t stream) {
  if (R != 16 && R != 32 && R != 64) return C2_ERR_UNSUPPORTED;
  return c2tp::run8_states(B, N, (int)R, t, t_bs, c, c_bs, a, U, V, X, work, guard, (hipStream_t)stream);
}
// Forward log-likelihood, time-parallel.  `guard` (two device words, zeroed here): a failed factorisation (or a NaN) raises
// both; the caller launches the ordinary kernel behind it with `guard + 1` as its gate.
int c2_internal_loglik_timepar(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                               int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                               double *ll, int32_t *flag, double *work, unsigned long long *guard, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (J == 8) return c2tp::run8(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, work, guard, s);
  if (J == 4) return c2tp::run<4>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, work, guard, s);
  if (J == 2) return c2tp::run<2>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, work, guard, s);
  return C2_ERR_UNSUPPORTED;
}

// `factor` (d, W, flag; no S workspace, not in place), time-parallel; same scratch and guard protocol.
int c2_internal_factor_timepar(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                               int64_t c_bs, const double *a, const double *U, const double *V, double *d, double *W,
                               int32_t *flag, double *work, unsigned long long *guard, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (J == 4) return c2tp::run_factor<4>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, guard, s);
  if (J == 2) return c2tp::run_factor<2>(B, N, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, guard, s);
  return C2_ERR_UNSUPPORTED;
}

// solve_lower (lower != 0) / solve_upper with one right-hand side, time-parallel (widths 2, 4, 8).  scratch: the doubles
// c2_internal_timepar_solve_doubles returns (unused when N <= 4096).  Z may alias Y.
size_t c2_internal_timepar_solve_doubles(int64_t B, int64_t N, int64_t J) {
  if (J != 2 && J != 4 && J != 8) return 0;
  const size_t K = (size_t)((N + c2tp::kRows - 1) / c2tp::kRows);
  return K <= (size_t)c2tp::kThreads ? 2 : (size_t)(J * J + 2 * J) * (size_t)B * K;
}
int c2_internal_solve_timepar(int lower, int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                              int64_t c_bs, const double *U, const double *W, const double *Y, double *Z,
                              double *scratch, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  // lower: A (fed into the state) = W, B (applied to it) = U; upper: the other way round (internal.hpp:105-189)
#define C2_TPS(J_)                                                                                              \
  return lower ? c2tp::run_solve<J_, true>(B, N, t, t_bs, c, c_bs, W, U, Y, Z, scratch, s)                      \
               : c2tp::run_solve<J_, false>(B, N, t, t_bs, c, c_bs, U, W, Y, Z, scratch, s)
  if (J == 2) { C2_TPS(2); }
  if (J == 4) { C2_TPS(4); }
  if (J == 8) { C2_TPS(8); }
#undef C2_TPS
  return C2_ERR_UNSUPPORTED;
}

}  // extern "C"
