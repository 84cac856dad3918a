// c2_wide.hip -- every recursion of the library for WIDE models, 32 < J <= C2_MAX_WIDTH (128).
//
// The reference's dynamic-size path takes any width (python/celerite2/driver.hpp:98-99: Eigen::Dynamic when no fixed size
// matches); the tuned kernels of this library stop at J = 32 (one lane per column of the J x J state, at most 32 lanes per
// series).  Beyond that the state no longer fits a wavefront's registers, so these kernels keep it in LDS: ONE WORKGROUP of
// 256 threads per series, the J x J factor state / its adjoint (up to 128 KB) and the J x nrhs sweep states in shared
// memory, the loops of the reference's expressions spread over the threads with a barrier where the reference has a data
// dependence.  Same operation order inside every dot product as the CPU oracle (oracle/c2_oracle.cpp: sums over i, over j
// run sequentially in one thread), block-level tree sums only for the scalar reductions (d_n, the time-gradient factor).
// These are completeness kernels -- a wide model costs O(J^2) per row whatever is done -- not tuned ones: they exist so that
// nothing the reference accepts is refused.  Covered: factor (+S), solve_* / matmul_* (+F, any nrhs), general_matmul_*
// (+F), factor_rev, the four sweep reverses; the fused log-likelihood (+gradient) is composed from them (c2_loglik.hip).
//
//   factor            forward.hpp:69-135          sweeps           internal.hpp:105-189
//   general_matmul_*  forward.hpp:285-392         sweep reverses   internal.hpp:191-303
//   factor_rev        reverse.hpp:10-85
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/celerite2_amd.h"

namespace c2w {

constexpr int kThreads = 256;
constexpr int KT = 16;   // right-hand sides per workgroup of the forward sweeps / products

// sum over the workgroup (every thread gets it); red: 8 doubles of LDS.  Two barriers.
__device__ __forceinline__ double block_sum(double x, double *red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();   // (earlier readers of red are done)
  if ((threadIdx.x & 63) == 0) red[w] = x;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// factor (forward.hpp:69-135).  LDS: S[J*J] (S(i,j) at i + J*j, the reference's column-major Sn), w[J], p[J], u[J], red[8].
// d may alias a, W may alias V (row n is read before row n is written).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_factor(int64_t N, int J, const double *__restrict__ t, int64_t t_bs,
                                                     const double *__restrict__ c, int64_t c_bs, const double *a,
                                                     const double *__restrict__ U, const double *V, double *d, double *W,
                                                     double *__restrict__ Sws, int32_t *__restrict__ flag) {
  extern __shared__ double sm[];
  double *S = sm, *w = S + J * J, *p = w + J, *u = p + J, *red = u + J;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const double *tb = t + b * t_bs, *cb = c + b * c_bs, *ab = a + b * N, *Ub = U + b * N * J, *Vb = V + b * N * J;
  double *db = d + b * N, *Wb = W + b * N * J, *Sb = Sws ? Sws + b * N * J * J : nullptr;
  for (int e = tid; e < J * J; e += kThreads) {
    S[e] = 0.0;
    if (Sb) Sb[e] = 0.0;   // S.row(0).setZero()
  }
  double dprev = ab[0];
  __syncthreads();   // (every thread has read a[0] before d[0] may overwrite it)
  if (tid < J) {
    const double w0 = Vb[tid] / dprev;
    Wb[tid] = w0;
    w[tid] = w0;
  }
  if (tid == 0) db[0] = dprev;
  int32_t fl = 0;
  for (int64_t n = 1; n < N; ++n) {
    const double an = ab[n], dt = tb[n - 1] - tb[n];
    double vn = 0.0;
    if (tid < J) {
      p[tid] = exp(cb[tid] * dt);
      u[tid] = Ub[n * J + tid];
      vn = Vb[n * J + tid];
    }
    __syncthreads();
    for (int e = tid; e < J * J; e += kThreads) {
      const int i = e % J, j = e / J;
      double s = S[e] + (dprev * w[i]) * w[j];   // forward.hpp:115
      s = p[i] * s;                              // :116
      if (Sb) Sb[n * J * J + e] = s;             // :120 the half-scaled state
      S[e] = s * p[j];                           // :123
    }
    __syncthreads();
    double tau = 0.0;
    if (tid < J)
      for (int i = 0; i < J; ++i) tau += u[i] * S[i + J * tid];   // :126
    const double dn = an - block_sum(tid < J ? tau * u[tid] : 0.0, red);   // :127
    if (tid == 0) db[n] = dn;
    if (dn <= 0.0) { fl = (int32_t)n; break; }   // :128 (uniform: every thread holds the same dn)
    if (tid < J) {
      const double wn = (vn - tau) / dn;         // :131
      Wb[n * J + tid] = wn;
      w[tid] = wn;
    }
    dprev = dn;
    __syncthreads();
  }
  if (tid == 0) flag[b] = fl;
}

// ---------------------------------------------------------------------------------------------------------------------
// solve_lower / solve_upper / matmul_lower / matmul_upper (internal.hpp:105-189), KT right-hand sides per workgroup.
// LDS: Fn[J*KT] (Fn(j,k) at j + J*k), p[J], A[J], Bv[J], x[KT].  Z may alias Y.
// ---------------------------------------------------------------------------------------------------------------------
template <bool LOWER, bool SOLVE>
__global__ __launch_bounds__(kThreads) void k_sweep(int64_t N, int J, int64_t nrhs, const double *__restrict__ t,
                                                    int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                    const double *__restrict__ U, const double *__restrict__ V,
                                                    const double *Y, double *Z, double *__restrict__ Fws, int zero_z) {
  extern __shared__ double sm[];
  double *Fn = sm, *p = Fn + J * KT, *A = p + J, *Bv = A + J, *x = Bv + J;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x, k0 = (int64_t)blockIdx.y * KT;
  const int kc = (int)((nrhs - k0) < KT ? (nrhs - k0) : KT);
  const double *tb = t + b * t_bs, *cb = c + b * c_bs, *Ub = U + b * N * J, *Vb = V + b * N * J;
  const double *Yb = Y + b * N * nrhs + k0;
  double *Zb = Z + b * N * nrhs + k0, *Fb = Fws ? Fws + b * N * J * nrhs : nullptr;
  const int64_t first = LOWER ? 0 : N - 1;
  for (int e = tid; e < J * kc; e += kThreads) {
    Fn[e] = 0.0;
    if (Fb) Fb[first * J * nrhs + (e % J) + J * (k0 + e / J)] = 0.0;   // internal.hpp:127 / 170
  }
  double prev = 0.0, zprev = 0.0;   // thread kk < kc: Y / Z of the previously visited row, column k0 + kk
  if (tid < kc) {
    prev = Yb[first * nrhs + tid];
    if (SOLVE) { zprev = prev; Zb[first * nrhs + tid] = prev; }   // Z = Y first (forward.hpp:168, 205)
    else if (zero_z) Zb[first * nrhs + tid] = 0.0;                // the *_fwd variants: Z.setZero() (backprop.cpp matmul_*_fwd)
  }
  __syncthreads();
  for (int64_t s = 1; s < N; ++s) {
    const int64_t n = LOWER ? s : N - 1 - s, m = LOWER ? n - 1 : n + 1;
    const double dt = LOWER ? (tb[m] - tb[n]) : (tb[n] - tb[m]);
    if (tid < J) {
      p[tid] = exp(cb[tid] * dt);
      A[tid] = (LOWER ? Vb : Ub)[m * J + tid];
      Bv[tid] = (LOWER ? Ub : Vb)[n * J + tid];
    }
    double yn = 0.0, z0 = 0.0;
    if (tid < kc) {
      x[tid] = SOLVE ? zprev : prev;
      yn = Yb[n * nrhs + tid];                                   // (read before Z[n] is written: Z may alias Y)
      z0 = SOLVE ? yn : (zero_z ? 0.0 : Zb[n * nrhs + tid]);
    }
    __syncthreads();
    for (int e = tid; e < J * kc; e += kThreads) {
      const int j = e % J, kk = e / J;
      const double f = Fn[e] + A[j] * x[kk];                      // update_f (internal.hpp:45-85)
      if (Fb) Fb[n * J * nrhs + j + J * (k0 + kk)] = f;           // internal.hpp:142 / 185
      Fn[e] = f * p[j];                                           // :143 / 186
    }
    __syncthreads();
    if (tid < kc) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += Bv[j] * Fn[j + J * tid];
      const double zn = SOLVE ? z0 - acc : z0 + acc;              // :144 / 187
      Zb[n * nrhs + tid] = zn;
      prev = yn;
      zprev = zn;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Reverse of the sweeps (internal.hpp:191-303), all right-hand sides in one workgroup (the width-J outputs sum over them).
// LDS: bF[J*nrhs], p[J], bp[J], Bv[J], A[J], bz[nrhs], xr[nrhs], red[8].  Outputs are written once, complete: row n of
// bB at step n, row m of bA / bX at step n, bt through a carry (every row receives one +factor and one -factor).
// ---------------------------------------------------------------------------------------------------------------------
template <bool LOWER, bool SOLVE>
__global__ __launch_bounds__(kThreads) void k_sweep_rev(int64_t N, int J, int64_t nrhs, const double *__restrict__ t,
                                                        int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                        const double *__restrict__ U, const double *__restrict__ V,
                                                        const double *__restrict__ Y, const double *__restrict__ Z,
                                                        const double *__restrict__ F, const double *__restrict__ bZ,
                                                        double *__restrict__ bt, double *__restrict__ bc,
                                                        double *__restrict__ bU, double *__restrict__ bV,
                                                        double *__restrict__ bY) {
  extern __shared__ double sm[];
  const int K = (int)nrhs;
  double *bF = sm, *p = bF + J * K, *bp = p + J, *Bv = bp + J, *A = Bv + J, *bz = A + J, *xr = bz + K, *red = xr + K;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const double *tb = t + b * t_bs, *cb = c + b * c_bs, *Ub = U + b * N * J, *Vb = V + b * N * J;
  const double *Xb = (SOLVE ? Z : Y) + b * N * nrhs, *Fb = F + b * N * J * nrhs, *bZb = bZ + b * N * nrhs;
  double *btb = bt + b * N, *bUb = bU + b * N * J, *bVb = bV + b * N * J, *bYb = bY + b * N * nrhs;
  double *bBb = LOWER ? bUb : bVb, *bAb = LOWER ? bVb : bUb;
  for (int e = tid; e < J * K; e += kThreads) bF[e] = 0.0;
  const int64_t nfirst = LOWER ? N - 1 : 0;            // the row the reverse starts from
  for (int k = tid; k < K; k += kThreads) {
    const double v = bZb[nfirst * nrhs + k];
    bz[k] = v;
    bYb[nfirst * nrhs + k] = SOLVE ? v : 0.0;          // solve: bY = bZ there (never an m); matmul: bY row never touched
  }
  if (tid < J) {
    bAb[nfirst * J + tid] = 0.0;                       // bV[N-1] (lower) / bU[0] (upper) receive nothing
    if (N == 1) bBb[tid] = 0.0;
  }
  double bcj = 0.0, carry = 0.0;
  __syncthreads();
  for (int64_t s = N - 1; s >= 1; --s) {
    const int64_t n = LOWER ? s : N - 1 - s, m = LOWER ? n - 1 : n + 1;
    const double dt = LOWER ? (tb[m] - tb[n]) : (tb[n] - tb[m]);
    const double *Fn = Fb + n * J * nrhs;
    if (tid < J) {
      p[tid] = exp(cb[tid] * dt);
      Bv[tid] = (LOWER ? Ub : Vb)[n * J + tid];
      A[tid] = (LOWER ? Vb : Ub)[m * J + tid];
    }
    for (int k = tid; k < K; k += kThreads) xr[k] = Xb[m * nrhs + k];
    __syncthreads();
    // reverse of update_z (internal.hpp:232-233 / 289-290): bB_n, bF
    if (tid < J) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += bz[k] * (p[tid] * Fn[tid + J * k]);
      bBb[n * J + tid] = SOLVE ? -acc : acc;
    }
    for (int e = tid; e < J * K; e += kThreads) {
      const double v = Bv[e % J] * bz[e / J];
      bF[e] = SOLVE ? bF[e] - v : bF[e] + v;
    }
    __syncthreads();
    // reverse of the decay (internal.hpp:236-241 / 293-298)
    double bpj = 0.0;
    if (tid < J) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += Fn[tid + J * k] * bF[tid + J * k];
      bpj = acc * p[tid];
      bcj += dt * bpj;
    }
    const double factor = block_sum(tid < J ? cb[tid] * bpj : 0.0, red);
    if (tid == 0) btb[n] = LOWER ? carry - factor : factor - carry;   // lower: (+f' of the step above) - f; upper: mirrored
    carry = factor;
    for (int e = tid; e < J * K; e += kThreads) bF[e] *= p[e % J];
    __syncthreads();
    // update_f::reverse (internal.hpp:55-63 matmul, 76-84 solve): bA_m, bX_m
    if (tid < J) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) acc += xr[k] * bF[tid + J * k];
      bAb[m * J + tid] = acc;
    }
    for (int k = tid; k < K; k += kThreads) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += A[j] * bF[j + J * k];
      const double in = bZb[m * nrhs + k];
      const double out = SOLVE ? in + acc : acc;       // solve: bZ_m (= bY_m) += ...; matmul: bY_m = ...
      bYb[m * nrhs + k] = out;
      bz[k] = SOLVE ? out : in;                        // the cotangent of Z_m that the next step reads
    }
    __syncthreads();
  }
  // the last visited m (row 0 lower, N-1 upper): its bB gets nothing, its bt the carry
  const int64_t nlast = LOWER ? 0 : N - 1;
  if (tid < J) {
    if (N > 1) bBb[nlast * J + tid] = 0.0;
    bc[b * J + tid] = bcj;
  }
  if (tid == 0) btb[nlast] = LOWER ? carry : -carry;
}

// ---------------------------------------------------------------------------------------------------------------------
// factor_rev (reverse.hpp:10-85).  LDS: bS[J*J] (bS(i,j) at i + J*j), p, bp, x, yv, un, wm, bvn, red.  S is read from the
// caller's workspace (J*J doubles per row).  accumulate != 0: bt, bc, bU are ADDED to (the composite gradient chain).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_factor_rev(int64_t N, int J, const double *__restrict__ t, int64_t t_bs,
                                                         const double *__restrict__ c, int64_t c_bs,
                                                         const double *__restrict__ U, const double *__restrict__ d,
                                                         const double *__restrict__ W, const double *__restrict__ S,
                                                         const double *__restrict__ bd, const double *__restrict__ bW,
                                                         double *bt, double *bc, double *__restrict__ ba, double *bU,
                                                         double *__restrict__ bV, int accumulate) {
  extern __shared__ double sm[];
  double *bS = sm, *p = bS + J * J, *x = p + J, *yv = x + J, *un = yv + J, *wm = un + J, *bvn = wm + J, *red = bvn + J;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const double *tb = t + b * t_bs, *cb = c + b * c_bs, *Ub = U + b * N * J, *db = d + b * N, *Wb = W + b * N * J;
  const double *Sb = S + b * N * J * J, *bdb = bd + b * N, *bWb = bW + b * N * J;
  double *btb = bt + b * N, *bab = ba + b * N, *bUb = bU + b * N * J, *bVb = bV + b * N * J;
  for (int e = tid; e < J * J; e += kThreads) bS[e] = 0.0;
  if (tid < J) bvn[tid] = bWb[(N - 1) * J + tid] / db[N - 1];   // reverse.hpp:56-57
  double ba_cur = bdb[N - 1];                                    // :55
  double bcj = 0.0, carry = 0.0;
  __syncthreads();
  for (int64_t n = N - 1; n > 0; --n) {
    const double dt = tb[n - 1] - tb[n];
    const double *Sn = Sb + n * J * J;
    double wn = 0.0;
    if (tid < J) {
      p[tid] = exp(cb[tid] * dt);
      un[tid] = Ub[n * J + tid];
      wm[tid] = Wb[(n - 1) * J + tid];
      wn = Wb[n * J + tid];
    }
    // step 6 (reverse.hpp:65-67)
    const double ban = ba_cur - block_sum(tid < J ? wn * bvn[tid] : 0.0, red);
    if (tid == 0) bab[n] = ban;
    if (tid < J) {
      x[tid] = bvn[tid] + 2.0 * ban * un[tid];
      yv[tid] = bvn[tid] + ban * un[tid];
      bVb[n * J + tid] = bvn[tid];                               // row n of bV is complete
    }
    __syncthreads();
    if (tid < J) {
      double acc = 0.0;
      for (int i = 0; i < J; ++i) acc += x[i] * Sn[i + J * tid];
      const double v = -acc * p[tid];
      bUb[n * J + tid] = accumulate ? bUb[n * J + tid] + v : v;
    }
    for (int e = tid; e < J * J; e += kThreads) bS[e] -= un[e % J] * yv[e / J];
    __syncthreads();
    // step 4 (reverse.hpp:70-74): bp = diag(bS Sn + Sn^T bS) o p
    double bpk = 0.0;
    if (tid < J) {
      double acc = 0.0;
      for (int i = 0; i < J; ++i) acc += bS[tid + J * i] * Sn[i + J * tid];
      for (int i = 0; i < J; ++i) acc += Sn[i + J * tid] * bS[i + J * tid];
      bpk = acc * p[tid];
      bcj += dt * bpk;
    }
    const double factor = block_sum(tid < J ? cb[tid] * bpk : 0.0, red);
    if (tid == 0) {
      const double v = carry - factor;
      btb[n] = accumulate ? btb[n] + v : v;
    }
    carry = factor;
    // step 3 (reverse.hpp:77-80)
    for (int e = tid; e < J * J; e += kThreads) bS[e] = p[e % J] * bS[e] * p[e / J];
    __syncthreads();
    double bswt = 0.0, accv = 0.0;
    if (tid < J) {
      for (int j = 0; j < J; ++j) bswt += bS[tid + J * j] * wm[j];
      for (int i = 0; i < J; ++i) accv += wm[i] * (bS[i + J * tid] + bS[tid + J * i]);
    }
    const double q = block_sum(tid < J ? wm[tid] * bswt : 0.0, red);
    ba_cur = bdb[n - 1] + q;
    if (tid < J) bvn[tid] = bWb[(n - 1) * J + tid] / db[n - 1] + accv;
    __syncthreads();
  }
  // row 0 (reverse.hpp:83-84)
  const double dot0 = block_sum(tid < J ? bvn[tid] * Wb[tid] : 0.0, red);
  if (tid < J) {
    bVb[tid] = bvn[tid];
    if (!accumulate) bUb[tid] = 0.0;
    bc[b * J + tid] = accumulate ? bc[b * J + tid] + bcj : bcj;
  }
  if (tid == 0) {
    bab[0] = ba_cur - dot0;
    btb[0] = accumulate ? btb[0] + carry : carry;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// general_matmul_lower / upper (forward.hpp:285-392): the two-pointer merge, walked by the whole workgroup (one series:
// the control flow is uniform).  LDS: Fm[J*KT] (Fm(j,k) at j*KT + k), p[J], vr[J], ur[J].  F workspace row-major
// F[m, j*nrhs + k]; rows the merge never reaches stay untouched.
// ---------------------------------------------------------------------------------------------------------------------
template <bool LOWER>
__global__ __launch_bounds__(kThreads) void k_general(int64_t N, int64_t M, int J, int64_t nrhs, const double *__restrict__ t1,
                                                      int64_t t1_bs, const double *__restrict__ t2, int64_t t2_bs,
                                                      const double *__restrict__ c, int64_t c_bs,
                                                      const double *__restrict__ U, const double *__restrict__ V,
                                                      const double *__restrict__ Y, double *Z, double *__restrict__ Fws,
                                                      int zero_z) {
  extern __shared__ double sm[];
  double *Fm = sm, *p = Fm + J * KT, *vr = p + J, *ur = vr + J;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x, k0 = (int64_t)blockIdx.y * KT;
  const int kc = (int)((nrhs - k0) < KT ? (nrhs - k0) : KT);
  const double *t1b = t1 + b * t1_bs, *t2b = t2 + b * t2_bs, *cb = c + b * c_bs;
  const double *Ub = U + b * N * J, *Vb = V + b * M * J, *Yb = Y + b * M * nrhs + k0;
  double *Zb = Z + b * N * nrhs + k0, *Fb = Fws ? Fws + b * M * J * nrhs : nullptr;
  const int64_t mfirst = LOWER ? 0 : M - 1;
  if (tid < J) vr[tid] = Vb[mfirst * J + tid];
  __syncthreads();
  for (int e = tid; e < J * kc; e += kThreads) {
    const int j = e / kc, kk = e % kc;
    const double f = vr[j] * Yb[mfirst * nrhs + kk];
    Fm[j * KT + kk] = f;
    if (Fb) Fb[j * nrhs + k0 + kk] = LOWER ? f : 0.0;   // forward.hpp:311-314 (row 0 = V_0^T Y_0) / 358 (row 0 zeroed)
  }
  double tn = t2b[mfirst];
  int64_t n, m;
  if (LOWER) {
    m = 1;
    for (n = 0; n < N; ++n)
      if (t1b[n] >= tn) break;
  } else {
    m = M - 2;
    for (n = N - 1; n >= 0; --n)
      if (t1b[n] < tn) break;
  }
  // outputs on the near side of the first t2 row receive nothing (zero_z: they are still defined)
  if (zero_z && tid < kc) {
    if (LOWER) for (int64_t q = 0; q < n; ++q) Zb[q * nrhs + tid] = 0.0;
    else for (int64_t q = N - 1; q > n; --q) Zb[q * nrhs + tid] = 0.0;
  }
  __syncthreads();
  for (; LOWER ? n < N : n >= 0; LOWER ? ++n : --n) {
    tn = t1b[n];
    while (LOWER ? (m < M && t2b[m] <= tn) : (m >= 0 && t2b[m] > tn)) {
      const double dt = LOWER ? (t2b[m - 1] - t2b[m]) : (t2b[m] - t2b[m + 1]);
      if (tid < J) {
        p[tid] = exp(cb[tid] * dt);
        vr[tid] = Vb[m * J + tid];
      }
      __syncthreads();
      for (int e = tid; e < J * kc; e += kThreads) {
        const int j = e / kc, kk = e % kc;
        double f = p[j] * Fm[j * KT + kk];
        f += vr[j] * Yb[m * nrhs + kk];
        Fm[j * KT + kk] = f;
        if (Fb) Fb[m * J * nrhs + j * nrhs + k0 + kk] = f;
      }
      __syncthreads();
      LOWER ? ++m : --m;
    }
    const double dt = LOWER ? (t2b[m - 1] - tn) : (tn - t2b[m + 1]);
    if (tid < J) {
      p[tid] = exp(cb[tid] * dt);
      ur[tid] = Ub[n * J + tid];
    }
    __syncthreads();
    if (tid < kc) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += (ur[j] * p[j]) * Fm[j * KT + tid];
      Zb[n * nrhs + tid] = (zero_z ? 0.0 : Zb[n * nrhs + tid]) + acc;
    }
    __syncthreads();
  }
}

// a failed factorisation: -inf was written by the reduction; all six gradients become NaN (as the tuned kernels do)
__global__ void k_nan_failed(int64_t B, int64_t N, int J, const int32_t *__restrict__ flag, double *bt, double *bc, double *ba,
                             double *bU, double *bV, double *by) {
  const int64_t b = blockIdx.x;
  if (flag[b] == 0) return;
  const double nan = __builtin_nan("");
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) { bt[b * N + i] = nan; ba[b * N + i] = nan; by[b * N + i] = nan; }
  for (int64_t i = threadIdx.x; i < N * J; i += blockDim.x) { bU[b * N * J + i] = nan; bV[b * N * J + i] = nan; }
  for (int i = threadIdx.x; i < J; i += blockDim.x) bc[b * J + i] = nan;
}

// ll[b] = -1/2 (sum log d + N log 2 pi) - 1/2 sum z^2 / d   (numpy.py:104-109, core.py:428)
__global__ __launch_bounds__(kThreads) void k_ll(int64_t N, const double *__restrict__ d, const double *__restrict__ z,
                                                 const int32_t *__restrict__ flag, double *__restrict__ ll) {
  __shared__ double red[8];
  const int64_t b = blockIdx.x;
  double acc = 0.0;
  for (int64_t n = threadIdx.x; n < N; n += kThreads) {
    const double dn = d[b * N + n], zn = z[b * N + n];
    acc += log(dn) + zn * zn / dn;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) ll[b] = flag[b] ? -INFINITY : -0.5 * (s + (double)N * 1.8378770664093454835606594728112);
}

inline int ok() { return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP; }
inline bool fits(size_t doubles) { return doubles * sizeof(double) <= 160 * 1024; }
template <class K>
inline int set_lds(K kern, size_t bytes) {
  if (bytes <= 64 * 1024) return C2_OK;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess
             ? C2_OK : C2_ERR_HIP;
}

}  // namespace c2w

using namespace c2w;

extern "C" {

int c2_wide_factor(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                   const double *a, const double *U, const double *V, double *d, double *W, double *S, int32_t *flag,
                   c2_stream_t stream) {
  const size_t nd = (size_t)J * J + 3 * J + 8;
  if (!fits(nd)) return C2_ERR_UNSUPPORTED;
  if (int e = set_lds(k_factor, nd * 8)) return e;
  hipLaunchKernelGGL(k_factor, dim3((unsigned)B), dim3(kThreads), nd * 8, (hipStream_t)stream, N, (int)J, t, t_bs, c, c_bs, a, U,
                     V, d, W, S, flag);
  return ok();
}

int c2_wide_sweep(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                  const double *c, int64_t c_bs, const double *U, const double *V, const double *Y, double *Z, double *F,
                  int zero_z, c2_stream_t stream) {
  const size_t nd = (size_t)J * KT + 3 * J + KT;
  const dim3 grid((unsigned)B, (unsigned)((nrhs + KT - 1) / KT));
#define C2W_SW(LO, SO)                                                                                                  \
  do {                                                                                                                  \
    if (int e = set_lds(k_sweep<LO, SO>, nd * 8)) return e;                                                             \
    hipLaunchKernelGGL((k_sweep<LO, SO>), grid, dim3(kThreads), nd * 8, (hipStream_t)stream, N, (int)J, nrhs, t, t_bs, c, c_bs, \
                       U, V, Y, Z, F, zero_z);                                                                          \
  } while (0)
  if (lower && solve) C2W_SW(true, true);
  else if (lower) C2W_SW(true, false);
  else if (solve) C2W_SW(false, true);
  else C2W_SW(false, false);
#undef C2W_SW
  return ok();
}

int c2_wide_sweep_rev(int lower, int solve, int64_t B, int64_t N, int64_t J, int64_t nrhs, const double *t, int64_t t_bs,
                      const double *c, int64_t c_bs, const double *U, const double *V, const double *Y, const double *Z,
                      const double *F, const double *bZ, double *bt, double *bc, double *bU, double *bV, double *bY,
                      c2_stream_t stream) {
  const size_t nd = (size_t)J * nrhs + 4 * J + 2 * nrhs + 8;
  if (!fits(nd)) return C2_ERR_UNSUPPORTED;   // (J * nrhs beyond ~19000: not covered)
#define C2W_SR(LO, SO)                                                                                                  \
  do {                                                                                                                  \
    if (int e = set_lds(k_sweep_rev<LO, SO>, nd * 8)) return e;                                                         \
    hipLaunchKernelGGL((k_sweep_rev<LO, SO>), dim3((unsigned)B), dim3(kThreads), nd * 8, (hipStream_t)stream, N, (int)J, nrhs, \
                       t, t_bs, c, c_bs, U, V, Y, Z, F, bZ, bt, bc, bU, bV, bY);                                        \
  } while (0)
  if (lower && solve) C2W_SR(true, true);
  else if (lower) C2W_SR(true, false);
  else if (solve) C2W_SR(false, true);
  else C2W_SR(false, false);
#undef C2W_SR
  return ok();
}

int c2_wide_factor_rev(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                       const double *U, const double *d, const double *W, const double *S, const double *bd,
                       const double *bW, double *bt, double *bc, double *ba, double *bU, double *bV, int accumulate,
                       c2_stream_t stream) {
  const size_t nd = (size_t)J * J + 6 * J + 8;
  if (!fits(nd)) return C2_ERR_UNSUPPORTED;
  if (int e = set_lds(k_factor_rev, nd * 8)) return e;
  hipLaunchKernelGGL(k_factor_rev, dim3((unsigned)B), dim3(kThreads), nd * 8, (hipStream_t)stream, N, (int)J, t, t_bs, c, c_bs,
                     U, d, W, S, bd, bW, bt, bc, ba, bU, bV, accumulate);
  return ok();
}

int c2_wide_general(int lower, int64_t B, int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, int64_t t1_bs,
                    const double *t2, int64_t t2_bs, const double *c, int64_t c_bs, const double *U, const double *V,
                    const double *Y, double *Z, double *F, int zero_z, c2_stream_t stream) {
  const size_t nd = (size_t)J * KT + 3 * J;
  const dim3 grid((unsigned)B, (unsigned)((nrhs + KT - 1) / KT));
  if (lower) {
    if (int e = set_lds(k_general<true>, nd * 8)) return e;
    hipLaunchKernelGGL((k_general<true>), grid, dim3(kThreads), nd * 8, (hipStream_t)stream, N, M, (int)J, nrhs, t1, t1_bs, t2,
                       t2_bs, c, c_bs, U, V, Y, Z, F, zero_z);
  } else {
    if (int e = set_lds(k_general<false>, nd * 8)) return e;
    hipLaunchKernelGGL((k_general<false>), grid, dim3(kThreads), nd * 8, (hipStream_t)stream, N, M, (int)J, nrhs, t1, t1_bs, t2,
                       t2_bs, c, c_bs, U, V, Y, Z, F, zero_z);
  }
  return ok();
}

// fused log-likelihood of a wide model: factor + solve_lower + reduction; work: B N (J + 2) doubles
size_t c2_wide_loglik_doubles(int64_t B, int64_t N, int64_t J) { return (size_t)B * N * (J + 2); }
int c2_wide_loglik(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                   const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
                   double *work, c2_stream_t stream) {
  double *d = work, *z = d + B * N, *W = z + B * N;
  if (int e = c2_wide_factor(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, nullptr, flag, stream)) return e;
  if (int e = c2_wide_sweep(1, 1, B, N, J, 1, t, t_bs, c, c_bs, U, W, y, z, nullptr, 0, stream)) return e;
  hipLaunchKernelGGL(k_ll, dim3((unsigned)B), dim3(kThreads), 0, (hipStream_t)stream, N, (const double *)d, (const double *)z,
                     (const int32_t *)flag, ll);
  return ok();
}

int c2_wide_nan_failed(int64_t B, int64_t N, int64_t J, const int32_t *flag, double *bt, double *bc, double *ba, double *bU,
                       double *bV, double *by, c2_stream_t stream) {
  hipLaunchKernelGGL(k_nan_failed, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, B, N, (int)J, flag, bt, bc, ba, bU, bV,
                     by);
  return ok();
}

}  // extern "C"
