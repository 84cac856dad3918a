// =============================================================================
// oracle/c2_oracle.cpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (plain C++17, no Eigen, no dependencies) of celerite2's O(N)
// semiseparable recursions.  Only tests/, __graft_entry__.smoke() and the
// `cpu_baseline` leg of bench.py may load this library, and only as the checker
// / the timed CPU column -- never as something the product path calls.
//
// Why a restatement: the reference's own hot path is written in Eigen
// expression templates and `c++/vendor/eigen` is an empty, un-vendored git
// submodule (reference `.gitmodules:1-3`), so the reference cannot be compiled
// in this image (`oracle/_ref` is therefore NOT buildable -- see oracle/README
// and DESIGN.md).  Eigen only supplies expression evaluation; the algorithm is
// fully spelled out in the reference headers and is followed here loop for
// loop, in the same order of operations:
//
//   factor                 <- c++/include/celerite2/forward.hpp:69-135
//   forward / backward     <- c++/include/celerite2/internal.hpp:105-146, 148-189
//   solve_* / matmul_*     <- forward.hpp:156-170, 193-207, 228-239, 260-271
//   general_matmul_*       <- forward.hpp:285-332, 346-392
//   factor_rev             <- c++/include/celerite2/reverse.hpp:10-85
//   forward_rev/backward_rev <- internal.hpp:191-246, 248-303
//   *_rev wrappers         <- reverse.hpp:87-217
//   get_celerite_matrices  <- python/celerite2/driver.cpp:422-477
//   log-likelihood assembly<- python/celerite2/numpy.py:66-87,104-109,
//                             python/celerite2/core.py:407-428
//
// Parity pinning: this file is validated (tests/test_oracle.py) against dense
// linear algebra (numpy Cholesky / triangular products of the kernel matrix,
// the exact oracle the reference's own tests use: c++/test/test_factor.cpp:16-38,
// python/test/test_driver.py:26-135) and against finite differences for every
// *_rev (c++/test/helpers.hpp:230-244), on the reference's deterministic input
// recipes (c++/test/helpers.hpp:14-62, python/celerite2/testing.py:10-49), and
// against the committed golden fixtures under tests/golden/.
//
// Workspace layouts (reference-compatible):
//   S[n, i + J*j] = Sn(i,j)      (Sn column-major, forward.hpp:100-103)
//   F[n, j + J*k] = Fn(j,k)      (Fn column-major, internal.hpp:120-121)
//   general_matmul F[m, j*nrhs + k] (row-major, forward.hpp:300,313)
// =============================================================================
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int kMaxJ = 128;   // widest model the restatement takes on its stack (the device path: C2_MAX_WIDTH)

// JT > 0: compile-time width (the reference's FIXED_SIZE_MAP idea,
// python/celerite2/driver.hpp:27-34); JT == 0: run-time width.
template <int JT>
struct Width {
  int dyn;
  explicit Width(int J) : dyn(J) {}
  inline int operator()() const { return JT ? JT : dyn; }
};

// ----------------------------------------------------------------------------
// factor  (forward.hpp:69-135)
// ----------------------------------------------------------------------------
template <int JT>
int64_t factor_impl(int64_t N, int Jdyn, const double *t, const double *c, const double *a, const double *U,
                    const double *V, double *d, double *W, double *S) {
  const Width<JT> Jw(Jdyn);
  const int J = Jw();
  double Sn[(JT ? JT * JT : kMaxJ * kMaxJ)];
  double p[(JT ? JT : kMaxJ)], tmp[(JT ? JT : kMaxJ)], wprev[(JT ? JT : kMaxJ)];
  for (int k = 0; k < J * J; ++k) Sn[k] = 0.0;
  if (S) for (int k = 0; k < J * J; ++k) S[k] = 0.0;  // S.row(0).setZero()  (forward.hpp:92)

  // First row (forward.hpp:106-108)
  d[0] = a[0];
  for (int j = 0; j < J; ++j) W[j] = V[j] / d[0];

  for (int64_t n = 1; n < N; ++n) {
    const double dt = t[n - 1] - t[n];
    for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * dt);  // forward.hpp:112
    const double dprev = d[n - 1];
    for (int j = 0; j < J; ++j) wprev[j] = W[(n - 1) * J + j];

    // Sn += d(n-1) * W.row(n-1)^T * W.row(n-1)   (forward.hpp:115)
    for (int j = 0; j < J; ++j)
      for (int i = 0; i < J; ++i) Sn[i + J * j] += (dprev * wprev[i]) * wprev[j];
    // Sn = diag(p) * Sn                            (forward.hpp:116)
    for (int j = 0; j < J; ++j)
      for (int i = 0; i < J; ++i) Sn[i + J * j] = p[i] * Sn[i + J * j];
    // save the half-scaled state                   (forward.hpp:120)
    if (S) std::memcpy(S + n * J * J, Sn, sizeof(double) * J * J);
    // Sn *= diag(p)                                (forward.hpp:123)
    for (int j = 0; j < J; ++j)
      for (int i = 0; i < J; ++i) Sn[i + J * j] *= p[j];

    // tmp = U.row(n) * Sn ; d(n) = a(n) - tmp * U.row(n)^T   (forward.hpp:126-127)
    const double *Un = U + n * J;
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      for (int i = 0; i < J; ++i) acc += Un[i] * Sn[i + J * j];
      tmp[j] = acc;
    }
    double dot = 0.0;
    for (int j = 0; j < J; ++j) dot += tmp[j] * Un[j];
    const double dn = a[n] - dot;
    d[n] = dn;
    if (dn <= 0.0) return n;  // forward.hpp:128

    // W.row(n) = (V.row(n) - tmp) / d(n)             (forward.hpp:131)
    for (int j = 0; j < J; ++j) W[n * J + j] = (V[n * J + j] - tmp[j]) / dn;
  }
  return 0;
}

// ----------------------------------------------------------------------------
// Shared sweeps (internal.hpp:105-146 forward, 148-189 backward).
// LOWER=true  : n = 1..N-1,  F += V[n-1]^T x[n-1], Z[n] -/+= U[n] F
// LOWER=false : n = N-2..0,  F += U[n+1]^T x[n+1], Z[n] -/+= V[n] F
// ----------------------------------------------------------------------------
template <int JT, bool LOWER, bool SOLVE>
void sweep_impl(int64_t N, int Jdyn, int64_t nrhs, const double *t, const double *c, const double *U,
                const double *V, const double *Y, double *Z, double *F) {
  const Width<JT> Jw(Jdyn);
  const int J = Jw();
  std::vector<double> Fn_(static_cast<size_t>(J) * nrhs, 0.0), tmp_(nrhs);
  double *Fn = Fn_.data(), *tmp = tmp_.data();
  double p[(JT ? JT : kMaxJ)];
  if (N <= 0) return;
  const int64_t first = LOWER ? 0 : N - 1;
  if (F) for (int64_t k = 0; k < J * nrhs; ++k) F[first * J * nrhs + k] = 0.0;  // internal.hpp:127 / 170
  for (int64_t k = 0; k < nrhs; ++k) tmp[k] = Y[first * nrhs + k];                  // internal.hpp:134 / 174

  for (int64_t s = 1; s < N; ++s) {
    const int64_t n = LOWER ? s : N - 1 - s;      // current row
    const int64_t m = LOWER ? n - 1 : n + 1;      // previously visited row
    const double dt = LOWER ? (t[m] - t[n]) : (t[n] - t[m]);  // internal.hpp:139 / 182
    for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * dt);
    const double *Am = (LOWER ? V : U) + m * J;   // row fed into F
    const double *Bn = (LOWER ? U : V) + n * J;   // row applied to F
    // update_f<is_solve>::apply  (internal.hpp:45-85)
    for (int64_t k = 0; k < nrhs; ++k) {
      const double x = SOLVE ? Z[m * nrhs + k] : tmp[k];
      for (int j = 0; j < J; ++j) Fn[j + J * k] += Am[j] * x;
    }
    for (int64_t k = 0; k < nrhs; ++k) tmp[k] = Y[n * nrhs + k];  // internal.hpp:141 / 184
    if (F) std::memcpy(F + n * J * nrhs, Fn, sizeof(double) * J * nrhs);  // internal.hpp:142 / 185
    for (int64_t k = 0; k < nrhs; ++k)
      for (int j = 0; j < J; ++j) Fn[j + J * k] *= p[j];            // internal.hpp:143 / 186
    // update_z<is_solve>::apply(B.row(n) * Fn, Z.row(n))  (internal.hpp:144 / 187)
    for (int64_t k = 0; k < nrhs; ++k) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += Bn[j] * Fn[j + J * k];
      if (SOLVE) Z[n * nrhs + k] -= acc; else Z[n * nrhs + k] += acc;
    }
  }
}

// ----------------------------------------------------------------------------
// Reverse of the shared sweeps (internal.hpp:191-246 forward_rev, 248-303
// backward_rev).  bZ is read AND (for solves) updated in place; the wrappers
// below arrange the aliasing the reference uses (reverse.hpp:112-117).
// ----------------------------------------------------------------------------
template <int JT, bool LOWER, bool SOLVE>
void sweep_rev_impl(int64_t N, int Jdyn, int64_t nrhs, const double *t, const double *c, const double *U,
                    const double *V, const double *Y, const double *Z, const double *F, double *bZ,
                    double *bt, double *bc, double *bU, double *bV, double *bY) {
  const Width<JT> Jw(Jdyn);
  const int J = Jw();
  std::vector<double> bF_(static_cast<size_t>(J) * nrhs, 0.0);
  double *bF = bF_.data();
  double p[(JT ? JT : kMaxJ)], bp[(JT ? JT : kMaxJ)];

  // Iterate over the forward sweep's steps in reverse order.
  for (int64_t s = N - 1; s >= 1; --s) {
    const int64_t n = LOWER ? s : N - 1 - s;
    const int64_t m = LOWER ? n - 1 : n + 1;
    const double dt = LOWER ? (t[m] - t[n]) : (t[n] - t[m]);  // internal.hpp:227 / 284
    for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * dt);
    const double *Fn = F + n * J * nrhs;
    const double *Bn = (LOWER ? U : V) + n * J;
    double *bBn = (LOWER ? bU : bV) + n * J;
    const double *Am = (LOWER ? V : U) + m * J;
    double *bAm = (LOWER ? bV : bU) + m * J;

    // Reverse of update_z (internal.hpp:232-233 / 289-290)
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      for (int64_t k = 0; k < nrhs; ++k) acc += bZ[n * nrhs + k] * (p[j] * Fn[j + J * k]);
      if (SOLVE) bBn[j] -= acc; else bBn[j] += acc;
    }
    for (int64_t k = 0; k < nrhs; ++k)
      for (int j = 0; j < J; ++j) {
        const double v = Bn[j] * bZ[n * nrhs + k];
        if (SOLVE) bF[j + J * k] -= v; else bF[j + J * k] += v;
      }

    // Reverse of the decay (internal.hpp:236-241 / 293-298)
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      for (int64_t k = 0; k < nrhs; ++k) acc += Fn[j + J * k] * bF[j + J * k];
      bp[j] = acc * p[j];
    }
    for (int j = 0; j < J; ++j) bc[j] += dt * bp[j];
    double factor = 0.0;
    for (int j = 0; j < J; ++j) factor += c[j] * bp[j];
    if (LOWER) { bt[n] -= factor; bt[m] += factor; }
    else       { bt[m] -= factor; bt[n] += factor; }
    for (int64_t k = 0; k < nrhs; ++k)
      for (int j = 0; j < J; ++j) bF[j + J * k] *= p[j];

    // update_f<is_solve>::reverse (internal.hpp:55-63 matmul, 76-84 solve)
    const double *X = (SOLVE ? Z : Y) + m * nrhs;
    double *bX = (SOLVE ? bZ : bY) + m * nrhs;
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      for (int64_t k = 0; k < nrhs; ++k) acc += X[k] * bF[j + J * k];
      bAm[j] += acc;
    }
    for (int64_t k = 0; k < nrhs; ++k) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += Am[j] * bF[j + J * k];
      bX[k] += acc;
    }
  }
}

// ----------------------------------------------------------------------------
// factor_rev (reverse.hpp:10-85)
// ----------------------------------------------------------------------------
template <int JT>
void factor_rev_impl(int64_t N, int Jdyn, const double *t, const double *c, const double *U, const double *d,
                     const double *W, const double *S, const double *bd, const double *bW, double *bt, double *bc,
                     double *ba, double *bU, double *bV) {
  const Width<JT> Jw(Jdyn);
  const int J = Jw();
  double bS[(JT ? JT * JT : kMaxJ * kMaxJ)];     // default (col-major) Inner, element (i,j) at i + J*j
  double p[(JT ? JT : kMaxJ)], bp[(JT ? JT : kMaxJ)], x[(JT ? JT : kMaxJ)], bSWT[(JT ? JT : kMaxJ)];
  for (int k = 0; k < J * J; ++k) bS[k] = 0.0;
  for (int64_t n = 0; n < N; ++n) bt[n] = 0.0;
  for (int j = 0; j < J; ++j) bc[j] = 0.0;
  for (int64_t n = 0; n < N; ++n) ba[n] = bd[n];                                  // reverse.hpp:55
  for (int64_t n = 0; n < N; ++n)
    for (int j = 0; j < J; ++j) bV[n * J + j] = bW[n * J + j] / d[n];             // reverse.hpp:56-57

  for (int64_t n = N - 1; n > 0; --n) {
    const double dt = t[n - 1] - t[n];
    for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * dt);
    const double *Sn = S + n * J * J;   // Sn(i,j) at i + J*j
    const double *Un = U + n * J, *Wn = W + n * J, *Wm = W + (n - 1) * J;
    double *bVn = bV + n * J, *bVm = bV + (n - 1) * J, *bUn = bU + n * J;

    // Step 6 (reverse.hpp:65-67)
    double dot = 0.0;
    for (int j = 0; j < J; ++j) dot += Wn[j] * bVn[j];
    ba[n] -= dot;
    for (int i = 0; i < J; ++i) x[i] = bVn[i] + 2.0 * ba[n] * Un[i];
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      for (int i = 0; i < J; ++i) acc += x[i] * Sn[i + J * j];
      bUn[j] = -acc * p[j];
    }
    for (int j = 0; j < J; ++j) {
      const double yj = bVn[j] + ba[n] * Un[j];
      for (int i = 0; i < J; ++i) bS[i + J * j] -= Un[i] * yj;
    }

    // Step 4 (reverse.hpp:70-74): bp = diag(bS*Sn + Sn^T*bS) .* p
    for (int k = 0; k < J; ++k) {
      double acc = 0.0;
      for (int i = 0; i < J; ++i) acc += bS[k + J * i] * Sn[i + J * k];
      for (int i = 0; i < J; ++i) acc += Sn[i + J * k] * bS[i + J * k];
      bp[k] = acc * p[k];
    }
    for (int j = 0; j < J; ++j) bc[j] += dt * bp[j];
    double factor = 0.0;
    for (int j = 0; j < J; ++j) factor += c[j] * bp[j];
    bt[n] -= factor;
    bt[n - 1] += factor;

    // Step 3 (reverse.hpp:77-80)
    for (int j = 0; j < J; ++j)
      for (int i = 0; i < J; ++i) bS[i + J * j] = p[i] * bS[i + J * j] * p[j];
    for (int i = 0; i < J; ++i) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += bS[i + J * j] * Wm[j];
      bSWT[i] = acc;
    }
    double q = 0.0;
    for (int i = 0; i < J; ++i) q += Wm[i] * bSWT[i];
    ba[n - 1] += q;
    for (int j = 0; j < J; ++j) {
      double acc = 0.0;
      for (int i = 0; i < J; ++i) acc += Wm[i] * (bS[i + J * j] + bS[j + J * i]);
      bVm[j] += acc;
    }
  }
  for (int j = 0; j < J; ++j) bU[j] = 0.0;   // reverse.hpp:83
  double dot0 = 0.0;
  for (int j = 0; j < J; ++j) dot0 += bV[j] * W[j];
  ba[0] -= dot0;                             // reverse.hpp:84
}

// ----------------------------------------------------------------------------
// general_matmul_lower / upper (forward.hpp:285-332, 346-392).  Fm row-major.
// ----------------------------------------------------------------------------
void general_lower_impl(int64_t N, int64_t M, int J, int64_t nrhs, const double *t1, const double *t2,
                        const double *c, const double *U, const double *V, const double *Y, double *Z, double *F) {
  std::vector<double> Fm_(static_cast<size_t>(J) * nrhs), p_(J);
  double *Fm = Fm_.data(), *p = p_.data();
  if (F) for (int64_t k = 0; k < J * nrhs; ++k) F[k] = 0.0;  // F.row(0).setZero() (then overwritten, :313)
  for (int j = 0; j < J; ++j)
    for (int64_t k = 0; k < nrhs; ++k) Fm[j * nrhs + k] = V[j] * Y[k];
  if (F) std::memcpy(F, Fm, sizeof(double) * J * nrhs);
  double tn = t2[0];
  int64_t n, m = 1;
  for (n = 0; n < N; ++n)
    if (t1[n] >= tn) break;
  for (; n < N; ++n) {
    tn = t1[n];
    while (m < M && t2[m] <= tn) {
      for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * (t2[m - 1] - t2[m]));
      for (int j = 0; j < J; ++j)
        for (int64_t k = 0; k < nrhs; ++k) Fm[j * nrhs + k] = p[j] * Fm[j * nrhs + k];
      for (int j = 0; j < J; ++j)
        for (int64_t k = 0; k < nrhs; ++k) Fm[j * nrhs + k] += V[m * J + j] * Y[m * nrhs + k];
      if (F) std::memcpy(F + m * J * nrhs, Fm, sizeof(double) * J * nrhs);
      m++;
    }
    for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * (t2[m - 1] - tn));
    for (int64_t k = 0; k < nrhs; ++k) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += (U[n * J + j] * p[j]) * Fm[j * nrhs + k];
      Z[n * nrhs + k] += acc;
    }
  }
}

void general_upper_impl(int64_t N, int64_t M, int J, int64_t nrhs, const double *t1, const double *t2,
                        const double *c, const double *U, const double *V, const double *Y, double *Z, double *F) {
  std::vector<double> Fm_(static_cast<size_t>(J) * nrhs), p_(J);
  double *Fm = Fm_.data(), *p = p_.data();
  if (F) for (int64_t k = 0; k < J * nrhs; ++k) F[k] = 0.0;  // F.row(0).setZero()  (forward.hpp:358)
  for (int j = 0; j < J; ++j)
    for (int64_t k = 0; k < nrhs; ++k) Fm[j * nrhs + k] = V[(M - 1) * J + j] * Y[(M - 1) * nrhs + k];
  double tn = t2[M - 1];
  int64_t n, m = M - 2;
  for (n = N - 1; n >= 0; --n)
    if (t1[n] < tn) break;
  for (; n >= 0; --n) {
    tn = t1[n];
    while (m >= 0 && t2[m] > tn) {
      for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * (t2[m] - t2[m + 1]));
      for (int j = 0; j < J; ++j)
        for (int64_t k = 0; k < nrhs; ++k) Fm[j * nrhs + k] = p[j] * Fm[j * nrhs + k];
      for (int j = 0; j < J; ++j)
        for (int64_t k = 0; k < nrhs; ++k) Fm[j * nrhs + k] += V[m * J + j] * Y[m * nrhs + k];
      if (F) std::memcpy(F + m * J * nrhs, Fm, sizeof(double) * J * nrhs);
      m--;
    }
    for (int j = 0; j < J; ++j) p[j] = std::exp(c[j] * (tn - t2[m + 1]));
    for (int64_t k = 0; k < nrhs; ++k) {
      double acc = 0.0;
      for (int j = 0; j < J; ++j) acc += (U[n * J + j] * p[j]) * Fm[j * nrhs + k];
      Z[n * nrhs + k] += acc;
    }
  }
}

// J dispatch -- compile-time widths for the common sizes, run-time otherwise
// (the reference does this with UNWRAP_CASES, python/celerite2/driver.hpp:27-101).
#define C2O_DISPATCH_J(J, CALL)                \
  switch (J) {                                 \
    case 1: { constexpr int JT = 1; CALL; } break;  \
    case 2: { constexpr int JT = 2; CALL; } break;  \
    case 3: { constexpr int JT = 3; CALL; } break;  \
    case 4: { constexpr int JT = 4; CALL; } break;  \
    case 6: { constexpr int JT = 6; CALL; } break;  \
    case 8: { constexpr int JT = 8; CALL; } break;  \
    case 16: { constexpr int JT = 16; CALL; } break; \
    default: { constexpr int JT = 0; CALL; } break; \
  }

}  // namespace

extern "C" {

int c2o_max_width(void) { return kMaxJ; }

int64_t c2o_factor(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
                   const double *V, double *d, double *W, double *S /* nullable */) {
  int64_t flag = 0;
  C2O_DISPATCH_J(J, flag = factor_impl<JT>(N, (int)J, t, c, a, U, V, d, W, S));
  return flag;
}

// solve_lower / solve_upper: Z = Y first (forward.hpp:168, 205); F nullable.
void c2o_solve_lower(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                     const double *W, const double *Y, double *Z, double *F) {
  if (Z != Y) std::memmove(Z, Y, sizeof(double) * N * nrhs);
  C2O_DISPATCH_J(J, (sweep_impl<JT, true, true>(N, (int)J, nrhs, t, c, U, W, Y, Z, F)));
}
void c2o_solve_upper(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                     const double *W, const double *Y, double *Z, double *F) {
  if (Z != Y) std::memmove(Z, Y, sizeof(double) * N * nrhs);
  C2O_DISPATCH_J(J, (sweep_impl<JT, false, true>(N, (int)J, nrhs, t, c, U, W, Y, Z, F)));
}
// matmul_lower / matmul_upper: accumulate into caller's Z (forward.hpp:228-239, 260-271).
void c2o_matmul_lower(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                      const double *V, const double *Y, double *Z, double *F) {
  C2O_DISPATCH_J(J, (sweep_impl<JT, true, false>(N, (int)J, nrhs, t, c, U, V, Y, Z, F)));
}
void c2o_matmul_upper(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                      const double *V, const double *Y, double *Z, double *F) {
  C2O_DISPATCH_J(J, (sweep_impl<JT, false, false>(N, (int)J, nrhs, t, c, U, V, Y, Z, F)));
}

void c2o_general_matmul_lower(int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, const double *t2,
                              const double *c, const double *U, const double *V, const double *Y, double *Z,
                              double *F) {
  general_lower_impl(N, M, (int)J, nrhs, t1, t2, c, U, V, Y, Z, F);
}
void c2o_general_matmul_upper(int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, const double *t2,
                              const double *c, const double *U, const double *V, const double *Y, double *Z,
                              double *F) {
  general_upper_impl(N, M, (int)J, nrhs, t1, t2, c, U, V, Y, Z, F);
}

void c2o_factor_rev(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
                    const double *V, const double *d, const double *W, const double *S, const double *bd,
                    const double *bW, double *bt, double *bc, double *ba, double *bU, double *bV) {
  (void)a; (void)V;  // unused by the reference too (reverse.hpp:29-30)
  C2O_DISPATCH_J(J, (factor_rev_impl<JT>(N, (int)J, t, c, U, d, W, S, bd, bW, bt, bc, ba, bU, bV)));
}

static void zero(double *x, int64_t n) { for (int64_t i = 0; i < n; ++i) x[i] = 0.0; }

// solve_lower_rev (reverse.hpp:87-118): bY = bZ, then bY doubles as the running bZ.
void c2o_solve_lower_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                         const double *W, const double *Y, const double *Z, const double *F, const double *bZ,
                         double *bt, double *bc, double *bU, double *bW, double *bY) {
  zero(bt, N); zero(bc, J); zero(bU, N * J); zero(bW, N * J);
  std::memmove(bY, bZ, sizeof(double) * N * nrhs);
  C2O_DISPATCH_J(J, (sweep_rev_impl<JT, true, true>(N, (int)J, nrhs, t, c, U, W, Y, Z, F, bY, bt, bc, bU, bW, bY)));
}
void c2o_solve_upper_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                         const double *W, const double *Y, const double *Z, const double *F, const double *bZ,
                         double *bt, double *bc, double *bU, double *bW, double *bY) {
  zero(bt, N); zero(bc, J); zero(bU, N * J); zero(bW, N * J);
  std::memmove(bY, bZ, sizeof(double) * N * nrhs);
  C2O_DISPATCH_J(J, (sweep_rev_impl<JT, false, true>(N, (int)J, nrhs, t, c, U, W, Y, Z, F, bY, bt, bc, bU, bW, bY)));
}
// matmul_*_rev (reverse.hpp:153-217): bY zeroed, bZ is a pure input.
void c2o_matmul_lower_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                          const double *V, const double *Y, const double *Z, const double *F, const double *bZ,
                          double *bt, double *bc, double *bU, double *bV, double *bY) {
  zero(bt, N); zero(bc, J); zero(bU, N * J); zero(bV, N * J); zero(bY, N * nrhs);
  C2O_DISPATCH_J(J, (sweep_rev_impl<JT, true, false>(N, (int)J, nrhs, t, c, U, V, Y, Z, F, const_cast<double *>(bZ),
                                                      bt, bc, bU, bV, bY)));
}
void c2o_matmul_upper_rev(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,
                          const double *V, const double *Y, const double *Z, const double *F, const double *bZ,
                          double *bt, double *bc, double *bU, double *bV, double *bY) {
  zero(bt, N); zero(bc, J); zero(bU, N * J); zero(bV, N * J); zero(bY, N * nrhs);
  C2O_DISPATCH_J(J, (sweep_rev_impl<JT, false, false>(N, (int)J, nrhs, t, c, U, V, Y, Z, F, const_cast<double *>(bZ),
                                                       bt, bc, bU, bV, bY)));
}

// get_celerite_matrices (python/celerite2/driver.cpp:422-477)
void c2o_get_celerite_matrices(int64_t Jr, int64_t Jc, int64_t N, const double *ar, const double *ac,
                               const double *bc, const double *dc, const double *x, const double *diag, double *a,
                               double *U, double *V) {
  const int64_t J = Jr + 2 * Jc;
  double sum = 0.0;
  for (int64_t j = 0; j < Jr; ++j) sum += ar[j];
  for (int64_t j = 0; j < Jc; ++j) sum += ac[j];
  for (int64_t n = 0; n < N; ++n) {
    a[n] = diag[n] + sum;
    for (int64_t j = 0; j < Jr; ++j) { V[n * J + j] = 1.0; U[n * J + j] = ar[j]; }
    for (int64_t j = 0, ind = Jr; j < Jc; ++j, ind += 2) {
      const double arg = dc[j] * x[n];
      const double cs = V[n * J + ind] = std::cos(arg);
      const double sn = V[n * J + ind + 1] = std::sin(arg);
      U[n * J + ind] = ac[j] * cs + bc[j] * sn;
      U[n * J + ind + 1] = ac[j] * sn - bc[j] * cs;
    }
  }
}

// ----------------------------------------------------------------------------
// Log-likelihood assembly as the reference's numpy backend does it
// (python/celerite2/numpy.py:66-87 _do_compute, :104-109 _do_norm,
//  python/celerite2/core.py:428):  ll = -0.5*(sum log d + N log 2pi) - 0.5*sum z^2/d
// scratch: d (N), W (N*J), z (N).  Returns the factor flag (0 = ok).
// ----------------------------------------------------------------------------
int64_t c2o_loglik(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
                   const double *V, const double *y, double *ll, double *d, double *W, double *z) {
  const int64_t flag = c2o_factor(N, J, t, c, a, U, V, d, W, nullptr);
  if (flag) { *ll = -INFINITY; return flag; }   // quiet=True behaviour (numpy.py:78-82)
  c2o_solve_lower(N, J, 1, t, c, U, W, y, z, nullptr);
  double log_det = 0.0, norm = 0.0;
  for (int64_t n = 0; n < N; ++n) log_det += std::log(d[n]);
  for (int64_t n = 0; n < N; ++n) norm += z[n] * z[n] / d[n];
  *ll = -0.5 * (log_det + N * std::log(2.0 * M_PI)) - 0.5 * norm;
  return 0;
}

// Log-likelihood + reverse-mode gradient w.r.t. (t, c, a, U, V, y), chaining the
// reference ops exactly as an autodiff frontend would (pymc/ops.py:131-141):
//   seeds  bd = -0.5/d + 0.5 z^2/d^2 ,  bz = -z/d
//   (bt1,bc1,bU1,bW,by) = solve_lower_rev(..., bz)
//   (bt2,bc2,ba,bU2,bV) = factor_rev(..., bd, bW)
// work must hold N*(J*J + 5*J + 6) + J doubles.
int64_t c2o_loglik_grad(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
                        const double *V, const double *y, double *ll, double *bt, double *bc, double *ba,
                        double *bU, double *bV, double *by, double *work) {
  double *d = work, *W = d + N, *S = W + N * J, *z = S + N * J * J, *F = z + N, *bd = F + N * J, *bz = bd + N,
         *bW = bz + N, *bt2 = bW + N * J, *bU2 = bt2 + N, *bc2 = bU2 + N * J;
  const int64_t flag = c2o_factor(N, J, t, c, a, U, V, d, W, S);
  if (flag) { *ll = -INFINITY; return flag; }
  c2o_solve_lower(N, J, 1, t, c, U, W, y, z, F);
  double log_det = 0.0, norm = 0.0;
  for (int64_t n = 0; n < N; ++n) log_det += std::log(d[n]);
  for (int64_t n = 0; n < N; ++n) norm += z[n] * z[n] / d[n];
  *ll = -0.5 * (log_det + N * std::log(2.0 * M_PI)) - 0.5 * norm;
  for (int64_t n = 0; n < N; ++n) {
    bd[n] = -0.5 / d[n] + 0.5 * z[n] * z[n] / (d[n] * d[n]);
    bz[n] = -z[n] / d[n];
  }
  c2o_solve_lower_rev(N, J, 1, t, c, U, W, y, z, F, bz, bt, bc, bU, bW, by);
  c2o_factor_rev(N, J, t, c, a, U, V, d, W, S, bd, bW, bt2, bc2, ba, bU2, bV);
  for (int64_t n = 0; n < N; ++n) bt[n] += bt2[n];
  for (int64_t j = 0; j < J; ++j) bc[j] += bc2[j];
  for (int64_t k = 0; k < N * J; ++k) bU[k] += bU2[k];
  return 0;
}

int64_t c2o_loglik_grad_work_size(int64_t N, int64_t J) { return N * (J * J + 5 * J + 6) + J; }

// Batched drivers (contiguous (B,N[,J]) arrays; t and c have explicit batch
// strides so they can be shared, stride 0).  Threads over the batch only --
// each series is still the reference's single-threaded recursion.
void c2o_loglik_batched(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                        int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                        double *ll, int32_t *flag, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    std::vector<double> work(static_cast<size_t>(N) * (J + 2));
#pragma omp for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
      double *d = work.data(), *W = d + N, *z = W + N * J;
      flag[b] = (int32_t)c2o_loglik(N, J, t + b * t_bs, c + b * c_bs, a + b * N, U + b * N * J, V + b * N * J,
                                    y + b * N, ll + b, d, W, z);
    }
  }
}

void c2o_loglik_grad_batched(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                             int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                             double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                             int32_t *flag, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    std::vector<double> work(static_cast<size_t>(c2o_loglik_grad_work_size(N, J)));
#pragma omp for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
      flag[b] = (int32_t)c2o_loglik_grad(N, J, t + b * t_bs, c + b * c_bs, a + b * N, U + b * N * J, V + b * N * J,
                                         y + b * N, ll + b, bt + b * N, bc + b * J, ba + b * N, bU + b * N * J,
                                         bV + b * N * J, by + b * N, work.data());
    }
  }
}

int c2o_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
