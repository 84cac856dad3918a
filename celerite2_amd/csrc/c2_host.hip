// c2_host.hip -- c2h_* host entry points: the reference's per-op API on HOST
// arrays (B == 1), staged through device memory and run by the same gfx950
// kernels as the batched device API.  These are what the `driver` / `backprop`
// pybind11 modules bind (reference python/celerite2/driver.cpp, backprop.cpp).
//
// Aliasing is preserved end to end: two host arguments with the same address
// share ONE device buffer, so d == a, W == V, Z == Y exercise the kernels'
// in-place behaviour exactly as the reference's in-place calls do.  Every host
// array (outputs included) is uploaded first, so elements a kernel does not
// write (rows after a failed factorisation, workspace rows the merge never
// visits, accumulated Z) keep their caller-provided contents.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../../include/celerite2_amd.h"

extern "C" void c2_internal_set_error(const char *msg);

namespace {

struct Staging {
  struct Entry {
    const void *host;
    void *dev;
    size_t bytes;
    bool out;
  };
  std::vector<Entry> entries;
  int err = C2_OK;

  ~Staging() {
    for (auto &e : entries) (void)hipFree(e.dev);
  }
  bool fail(hipError_t e) {
    if (e == hipSuccess) return false;
    c2_internal_set_error(hipGetErrorString(e));
    err = C2_ERR_HIP;
    return true;
  }
  // Register a host array; returns the device pointer (shared when the host pointer repeats).
  double *map(const double *host, int64_t count, bool is_output) {
    if (err || host == nullptr) return nullptr;
    const size_t bytes = sizeof(double) * (size_t)count;
    for (auto &e : entries)
      if (e.host == host) {
        e.out = e.out || is_output;
        if (bytes > e.bytes) err = C2_ERR_INVALID;  // overlapping-but-different views are not supported
        return (double *)e.dev;
      }
    void *dev = nullptr;
    if (fail(hipMalloc(&dev, bytes ? bytes : 8))) return nullptr;
    entries.push_back({host, dev, bytes, is_output});
    if (fail(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice))) return nullptr;
    return (double *)dev;
  }
  const double *in(const double *h, int64_t n) { return map(h, n, false); }
  double *out(double *h, int64_t n) { return map(h, n, true); }
  int finish(int rc) {
    if (err) return err;
    if (rc != C2_OK) return rc;
    if (fail(hipDeviceSynchronize())) return err;
    for (auto &e : entries)
      if (e.out && fail(hipMemcpy(const_cast<void *>(e.host), e.dev, e.bytes, hipMemcpyDeviceToHost))) return err;
    return C2_OK;
  }
};

inline bool bad(int64_t N, int64_t J) { return N < 1 || J < 1; }

}  // namespace

extern "C" {

int c2h_factor(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
               const double *V, double *d, double *W, double *S, int64_t *flag) {
  if (bad(N, J) || !flag) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  Staging st;
  const double *t_ = st.in(t, N), *c_ = st.in(c, J), *a_ = st.in(a, N), *U_ = st.in(U, N * J), *V_ = st.in(V, N * J);
  double *d_ = st.out(d, N), *W_ = st.out(W, N * J), *S_ = S ? st.out(S, N * J * J) : nullptr;
  int32_t *flag_ = nullptr;
  if (st.fail(hipMalloc((void **)&flag_, sizeof(int32_t)))) return st.err;
  int rc = st.err ? st.err : c2_factor(1, N, J, t_, 0, c_, 0, a_, U_, V_, d_, W_, S_, flag_, nullptr);
  rc = st.finish(rc);
  int32_t f = 0;
  if (rc == C2_OK && hipMemcpy(&f, flag_, sizeof(f), hipMemcpyDeviceToHost) != hipSuccess) rc = C2_ERR_HIP;
  (void)hipFree(flag_);
  *flag = f;
  return rc;
}

#define C2H_SWEEP(NAME, CALL)                                                                                       \
  int c2h_##NAME(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,            \
                 const double *W, const double *Y, double *Z, double *F) {                                         \
    if (bad(N, J) || nrhs < 1) return C2_ERR_INVALID;                                                               \
    if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;                                                                \
    Staging st;                                                                                                     \
    const double *t_ = st.in(t, N), *c_ = st.in(c, J), *U_ = st.in(U, N * J), *W_ = st.in(W, N * J),               \
                 *Y_ = st.in(Y, N * nrhs);                                                                          \
    double *Z_ = st.out(Z, N * nrhs), *F_ = F ? st.out(F, N * J * nrhs) : nullptr;                                 \
    return st.finish(st.err ? st.err : CALL);                                                                       \
  }
C2H_SWEEP(solve_lower, c2_solve_lower(1, N, J, nrhs, t_, 0, c_, 0, U_, W_, Y_, Z_, F_, nullptr))
C2H_SWEEP(solve_upper, c2_solve_upper(1, N, J, nrhs, t_, 0, c_, 0, U_, W_, Y_, Z_, F_, nullptr))
#undef C2H_SWEEP

#define C2H_MATMUL(NAME)                                                                                            \
  int c2h_##NAME(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,            \
                 const double *V, const double *Y, double *Z, double *F, int zero_z) {                             \
    if (bad(N, J) || nrhs < 1) return C2_ERR_INVALID;                                                               \
    if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;                                                                \
    Staging st;                                                                                                     \
    const double *t_ = st.in(t, N), *c_ = st.in(c, J), *U_ = st.in(U, N * J), *V_ = st.in(V, N * J),               \
                 *Y_ = st.in(Y, N * nrhs);                                                                          \
    double *Z_ = st.out(Z, N * nrhs), *F_ = F ? st.out(F, N * J * nrhs) : nullptr;                                 \
    return st.finish(st.err ? st.err : c2_##NAME(1, N, J, nrhs, t_, 0, c_, 0, U_, V_, Y_, Z_, F_, zero_z, nullptr)); \
  }
C2H_MATMUL(matmul_lower)
C2H_MATMUL(matmul_upper)
#undef C2H_MATMUL

#define C2H_GENERAL(NAME)                                                                                           \
  int c2h_##NAME(int64_t N, int64_t M, int64_t J, int64_t nrhs, const double *t1, const double *t2,               \
                 const double *c, const double *U, const double *V, const double *Y, double *Z, double *F,         \
                 int zero_z) {                                                                                      \
    if (bad(N, J) || M < 1 || nrhs < 1) return C2_ERR_INVALID;                                                      \
    if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;                                                                \
    Staging st;                                                                                                     \
    const double *t1_ = st.in(t1, N), *t2_ = st.in(t2, M), *c_ = st.in(c, J), *U_ = st.in(U, N * J),               \
                 *V_ = st.in(V, M * J), *Y_ = st.in(Y, M * nrhs);                                                   \
    double *Z_ = st.out(Z, N * nrhs), *F_ = F ? st.out(F, M * J * nrhs) : nullptr;                                 \
    return st.finish(st.err ? st.err                                                                                \
                            : c2_##NAME(1, N, M, J, nrhs, t1_, 0, t2_, 0, c_, 0, U_, V_, Y_, Z_, F_, zero_z, nullptr)); \
  }
C2H_GENERAL(general_matmul_lower)
C2H_GENERAL(general_matmul_upper)
#undef C2H_GENERAL

int c2h_factor_rev(int64_t N, int64_t J, const double *t, const double *c, const double *a, const double *U,
                   const double *V, const double *d, const double *W, const double *S, const double *bd,
                   const double *bW, double *bt, double *bc, double *ba, double *bU, double *bV) {
  if (bad(N, J)) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  Staging st;
  const double *t_ = st.in(t, N), *c_ = st.in(c, J), *a_ = st.in(a, N), *U_ = st.in(U, N * J), *V_ = st.in(V, N * J),
               *d_ = st.in(d, N), *W_ = st.in(W, N * J), *S_ = st.in(S, N * J * J), *bd_ = st.in(bd, N),
               *bW_ = st.in(bW, N * J);
  double *bt_ = st.out(bt, N), *bc_ = st.out(bc, J), *ba_ = st.out(ba, N), *bU_ = st.out(bU, N * J),
         *bV_ = st.out(bV, N * J);
  return st.finish(st.err ? st.err
                          : c2_factor_rev(1, N, J, t_, 0, c_, 0, a_, U_, V_, d_, W_, S_, bd_, bW_, bt_, bc_, ba_, bU_,
                                          bV_, nullptr));
}

#define C2H_SWEEP_REV(NAME)                                                                                         \
  int c2h_##NAME(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,            \
                 const double *W, const double *Y, const double *Z, const double *F, const double *bZ, double *bt, \
                 double *bc, double *bU, double *bW, double *bY) {                                                 \
    if (bad(N, J) || nrhs < 1) return C2_ERR_INVALID;                                                               \
    if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;                                                                \
    Staging st;                                                                                                     \
    const double *t_ = st.in(t, N), *c_ = st.in(c, J), *U_ = st.in(U, N * J), *W_ = st.in(W, N * J),               \
                 *Y_ = st.in(Y, N * nrhs), *Z_ = st.in(Z, N * nrhs), *F_ = st.in(F, N * J * nrhs),                 \
                 *bZ_ = st.in(bZ, N * nrhs);                                                                        \
    double *bt_ = st.out(bt, N), *bc_ = st.out(bc, J), *bU_ = st.out(bU, N * J), *bW_ = st.out(bW, N * J),         \
           *bY_ = st.out(bY, N * nrhs);                                                                             \
    return st.finish(st.err ? st.err                                                                                \
                            : c2_##NAME(1, N, J, nrhs, t_, 0, c_, 0, U_, W_, Y_, Z_, F_, bZ_, bt_, bc_, bU_, bW_,   \
                                        bY_, nullptr));                                                             \
  }
C2H_SWEEP_REV(solve_lower_rev)
C2H_SWEEP_REV(solve_upper_rev)
C2H_SWEEP_REV(matmul_lower_rev)
C2H_SWEEP_REV(matmul_upper_rev)
#undef C2H_SWEEP_REV

int c2h_get_celerite_matrices(int64_t N, int64_t Jr, int64_t Jc, const double *ar, const double *ac,
                              const double *bc, const double *dc, const double *x, const double *diag, double *a,
                              double *U, double *V) {
  if (N < 1 || Jr < 0 || Jc < 0 || Jr + 2 * Jc < 1) return C2_ERR_INVALID;
  const int64_t J = Jr + 2 * Jc;
  Staging st;
  const double *ar_ = Jr ? st.in(ar, Jr) : nullptr, *ac_ = Jc ? st.in(ac, Jc) : nullptr,
               *bc_ = Jc ? st.in(bc, Jc) : nullptr, *dc_ = Jc ? st.in(dc, Jc) : nullptr, *x_ = st.in(x, N),
               *diag_ = st.in(diag, N);
  double *a_ = st.out(a, N), *U_ = st.out(U, N * J), *V_ = st.out(V, N * J);
  return st.finish(st.err ? st.err
                          : c2_get_celerite_matrices(1, N, Jr, Jc, ar_, ac_, bc_, dc_, 0, x_, 0, diag_, a_, U_, V_,
                                                     nullptr));
}

}  // extern "C"
