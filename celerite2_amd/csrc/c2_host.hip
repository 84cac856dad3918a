// c2_host.hip -- c2h_* host entry points: the reference's per-op API on HOST
// arrays (B == 1), staged through device memory and run by the same gfx950
// kernels as the batched device API.  These are what the `driver` / `backprop`
// pybind11 modules bind (reference python/celerite2/driver.cpp, backprop.cpp).
//
// Staging (round 5): one grow-only DEVICE ARENA, one PINNED bounce buffer and one
// stream per calling thread.  A call packs every argument into the bounce buffer,
// uploads them with ONE host-to-device copy, runs its kernels on the thread's stream,
// downloads the outputs with ONE device-to-host copy and waits on that stream only
// (no hipMalloc / hipFree / device-wide synchronisation per argument or per call:
// tools/host_call_cost.py).  Threads do not share staging state, so the entry points
// are re-entrant like the reference's.
//
// Aliasing is preserved end to end: two host arguments with the same address
// share ONE device buffer, so d == a, W == V, Z == Y exercise the kernels'
// in-place behaviour exactly as the reference's in-place calls do.  Outputs whose
// unwritten elements must keep the caller's contents (rows after a failed
// factorisation, workspace rows the merge never visits, accumulated Z) are
// uploaded first; outputs the kernels overwrite completely (the S / F
// workspaces, the gradients of the *_rev ops) are not (round 6).  The arena
// gives memory beyond 256 MiB back after the call; c2h_release_thread_cache()
// gives back everything the calling thread holds.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/celerite2_amd.h"

extern "C" void c2_internal_set_error(const char *msg);

namespace {

// Per-thread staging resources, grown on demand and kept for the life of the thread.
struct Arena {
  char *dev = nullptr, *pin = nullptr;
  size_t dev_cap = 0, pin_cap = 0;
  hipStream_t stream = nullptr;
  int device = -1;
  ~Arena() { release(); }
  void release() {
    if (dev) (void)hipFree(dev);
    if (pin) (void)hipHostFree(pin);
    if (stream) (void)hipStreamDestroy(stream);
    dev = pin = nullptr; dev_cap = pin_cap = 0; stream = nullptr;
  }
  // One large call (S at N = 1e6: 0.5 GB) must not stay allocated for the life of the thread: beyond kKeepMax the arena is
  // given back once the call has finished (the next call grows it again); c2h_release_thread_cache() gives back everything.
  static constexpr size_t kKeepMax = 256u << 20;
  void trim() {
    if (dev_cap > kKeepMax) { (void)hipFree(dev); dev = nullptr; dev_cap = 0; }
    if (pin_cap > kKeepMax) { (void)hipHostFree(pin); pin = nullptr; pin_cap = 0; }
  }
  hipError_t reserve(size_t bytes, bool want_pin) {
    int cur = 0;
    if (hipError_t e = hipGetDevice(&cur)) return e;
    if (cur != device) { release(); device = cur; }   // (the caller switched devices: start over on this one)
    if (!stream)
      if (hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) return e;
    if (bytes > dev_cap) {
      if (dev) (void)hipFree(dev);
      dev = nullptr; dev_cap = 0;
      const size_t cap = bytes + bytes / 2 + (1u << 16);
      if (hipError_t e = hipMalloc((void **)&dev, cap)) return e;
      dev_cap = cap;
    }
    if (want_pin && bytes > pin_cap) {
      if (pin) (void)hipHostFree(pin);
      pin = nullptr; pin_cap = 0;
      const size_t cap = bytes + bytes / 2 + (1u << 16);
      if (hipError_t e = hipHostMalloc((void **)&pin, cap, hipHostMallocDefault)) return e;
      pin_cap = cap;
    }
    return hipSuccess;
  }
};
thread_local Arena g_arena;

constexpr size_t kAlign = 256;              // every staged array starts on a 256-byte boundary (the kernels' 16-byte checks)
constexpr size_t kBounceMax = 64u << 20;    // beyond this, arguments go up one by one straight from the caller's memory

struct Staging {
  struct Entry {
    const void *host;
    size_t bytes, off;
    bool out;
    bool fresh;   // a pure output the kernels overwrite completely: laid out behind everything else, never uploaded
  };
  std::vector<Entry> entries;
  // inputs are packed in front, in/out arrays (uploaded AND downloaded) behind them, pure outputs last:
  // the upload is [0, in_bytes + io_bytes), the download [in_bytes, in_bytes + out_bytes)
  size_t in_bytes = 0, io_bytes = 0, out_bytes = 0;
  int err = C2_OK;
  bool committed = false;

  bool fail(hipError_t e) {
    if (e == hipSuccess) return false;
    c2_internal_set_error(hipGetErrorString(e));
    err = C2_ERR_HIP;
    return true;
  }
  // Register a host array; returns its handle (shared when the host pointer repeats).  -1: absent.
  int map(const void *host, size_t bytes, bool is_output, bool fresh = false) {
    if (host == nullptr) return -1;
    for (size_t i = 0; i < entries.size(); ++i)
      if (entries[i].host == host) {
        entries[i].out = entries[i].out || is_output;
        entries[i].fresh = entries[i].fresh && fresh;        // (aliased with an input or an in/out array: it is read)
        if (bytes > entries[i].bytes) err = C2_ERR_INVALID;  // overlapping-but-different views are not supported
        return (int)i;
      }
    entries.push_back({host, bytes, 0, is_output, is_output && fresh});
    return (int)entries.size() - 1;
  }
  int in(const double *h, int64_t n) { return map(h, sizeof(double) * (size_t)n, false); }
  // an output whose unwritten elements must keep the caller's contents (rows behind a failed pivot, an accumulated Z, workspace
  // rows a merge never visits): uploaded first
  int out(double *h, int64_t n) { return map(h, sizeof(double) * (size_t)n, true); }
  // an output every element of which the kernels write (the S / F workspaces of factor and the sweeps, the gradients of the
  // *_rev ops: reverse.hpp "fully overwritten"): NOT uploaded -- ADVICE r05: S at N = 4096, J = 8 is 2 MB the call used to push
  // over PCIe for nothing, and reading the caller's uninitialised buffers
  int fresh(double *h, int64_t n) { return map(h, sizeof(double) * (size_t)n, true, true); }
  static size_t up(size_t b) { return (b + kAlign - 1) / kAlign * kAlign; }
  // Sizes are known: lay the arrays out, grow the arena, ONE upload.
  bool commit() {
    committed = true;
    if (err) return false;
    for (auto &e : entries) if (!e.out) { e.off = in_bytes; in_bytes += up(e.bytes ? e.bytes : 8); }
    for (auto &e : entries) if (e.out && !e.fresh) { e.off = in_bytes + io_bytes; io_bytes += up(e.bytes ? e.bytes : 8); }
    out_bytes = io_bytes;
    for (auto &e : entries) if (e.out && e.fresh) { e.off = in_bytes + out_bytes; out_bytes += up(e.bytes ? e.bytes : 8); }
    const size_t total = in_bytes + out_bytes, upload = in_bytes + io_bytes;
    const bool bounce = total <= kBounceMax;
    if (fail(g_arena.reserve(total, bounce))) return false;
    if (bounce) {
      for (auto &e : entries) if (!e.fresh) std::memcpy(g_arena.pin + e.off, e.host, e.bytes);
      if (upload && fail(hipMemcpyAsync(g_arena.dev, g_arena.pin, upload, hipMemcpyHostToDevice, g_arena.stream))) return false;
    } else {
      for (auto &e : entries)
        if (!e.fresh && fail(hipMemcpyAsync(g_arena.dev + e.off, e.host, e.bytes, hipMemcpyHostToDevice, g_arena.stream))) return false;
    }
    return true;
  }
  double *p(int h) const { return h < 0 ? nullptr : reinterpret_cast<double *>(g_arena.dev + entries[(size_t)h].off); }
  c2_stream_t stream() const { return (c2_stream_t)g_arena.stream; }
  // ONE download of the output block, a wait on this thread's stream, results back into the caller's arrays.
  int finish(int rc) {
    const int r = finish_(rc);
    g_arena.trim();
    return r;
  }
  int finish_(int rc) {
    if (err) return err;
    if (rc != C2_OK) { (void)hipStreamSynchronize(g_arena.stream); return rc; }
    const bool bounce = in_bytes + out_bytes <= kBounceMax;
    if (bounce) {
      if (out_bytes && fail(hipMemcpyAsync(g_arena.pin + in_bytes, g_arena.dev + in_bytes, out_bytes, hipMemcpyDeviceToHost, g_arena.stream)))
        return err;
      if (fail(hipStreamSynchronize(g_arena.stream))) return err;
      for (auto &e : entries)
        if (e.out) std::memcpy(const_cast<void *>(e.host), g_arena.pin + e.off, e.bytes);
    } else {
      for (auto &e : entries)
        if (e.out && fail(hipMemcpyAsync(const_cast<void *>(e.host), g_arena.dev + e.off, e.bytes, hipMemcpyDeviceToHost, g_arena.stream)))
          return err;
      if (fail(hipStreamSynchronize(g_arena.stream))) return err;
    }
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
  const int t_ = st.in(t, N), c_ = st.in(c, J), a_ = st.in(a, N), U_ = st.in(U, N * J), V_ = st.in(V, N * J);
  const int d_ = st.out(d, N), W_ = st.out(W, N * J), S_ = S ? st.fresh(S, N * J * J) : -1;
  int32_t f[2] = {0, 0};   // the pivot flag travels with the outputs
  const int f_ = st.map(f, sizeof(f), true);
  int rc = !st.commit() ? st.err
                        : c2_factor(1, N, J, st.p(t_), 0, st.p(c_), 0, st.p(a_), st.p(U_), st.p(V_), st.p(d_), st.p(W_),
                                    st.p(S_), reinterpret_cast<int32_t *>(st.p(f_)), st.stream());
  rc = st.finish(rc);
  *flag = f[0];
  return rc;
}

#define C2H_SWEEP(NAME, CALL)                                                                                       \
  int c2h_##NAME(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,            \
                 const double *W, const double *Y, double *Z, double *F) {                                         \
    if (bad(N, J) || nrhs < 1) return C2_ERR_INVALID;                                                               \
    if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;                                                                \
    Staging st;                                                                                                     \
    const int t_ = st.in(t, N), c_ = st.in(c, J), U_ = st.in(U, N * J), W_ = st.in(W, N * J), Y_ = st.in(Y, N * nrhs); \
    const int Z_ = st.out(Z, N * nrhs), F_ = F ? st.fresh(F, N * J * nrhs) : -1;                                   \
    return st.finish(!st.commit() ? st.err : CALL);                                                                 \
  }
C2H_SWEEP(solve_lower, c2_solve_lower(1, N, J, nrhs, st.p(t_), 0, st.p(c_), 0, st.p(U_), st.p(W_), st.p(Y_), st.p(Z_), st.p(F_), st.stream()))
C2H_SWEEP(solve_upper, c2_solve_upper(1, N, J, nrhs, st.p(t_), 0, st.p(c_), 0, st.p(U_), st.p(W_), st.p(Y_), st.p(Z_), st.p(F_), st.stream()))
#undef C2H_SWEEP

#define C2H_MATMUL(NAME)                                                                                            \
  int c2h_##NAME(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,            \
                 const double *V, const double *Y, double *Z, double *F, int zero_z) {                             \
    if (bad(N, J) || nrhs < 1) return C2_ERR_INVALID;                                                               \
    if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;                                                                \
    Staging st;                                                                                                     \
    const int t_ = st.in(t, N), c_ = st.in(c, J), U_ = st.in(U, N * J), V_ = st.in(V, N * J), Y_ = st.in(Y, N * nrhs); \
    const int Z_ = st.out(Z, N * nrhs), F_ = F ? st.fresh(F, N * J * nrhs) : -1;                                   \
    return st.finish(!st.commit() ? st.err                                                                          \
                                  : c2_##NAME(1, N, J, nrhs, st.p(t_), 0, st.p(c_), 0, st.p(U_), st.p(V_), st.p(Y_), \
                                              st.p(Z_), st.p(F_), zero_z, st.stream()));                            \
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
    const int t1_ = st.in(t1, N), t2_ = st.in(t2, M), c_ = st.in(c, J), U_ = st.in(U, N * J), V_ = st.in(V, M * J), \
              Y_ = st.in(Y, M * nrhs);                                                                              \
    const int Z_ = st.out(Z, N * nrhs), F_ = F ? st.out(F, M * J * nrhs) : -1;                                     \
    return st.finish(!st.commit() ? st.err                                                                          \
                                  : c2_##NAME(1, N, M, J, nrhs, st.p(t1_), 0, st.p(t2_), 0, st.p(c_), 0, st.p(U_),  \
                                              st.p(V_), st.p(Y_), st.p(Z_), st.p(F_), zero_z, st.stream()));        \
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
  const int t_ = st.in(t, N), c_ = st.in(c, J), a_ = st.in(a, N), U_ = st.in(U, N * J), V_ = st.in(V, N * J),
            d_ = st.in(d, N), W_ = st.in(W, N * J), S_ = st.in(S, N * J * J), bd_ = st.in(bd, N), bW_ = st.in(bW, N * J);
  const int bt_ = st.fresh(bt, N), bc_ = st.fresh(bc, J), ba_ = st.fresh(ba, N), bU_ = st.fresh(bU, N * J), bV_ = st.fresh(bV, N * J);
  return st.finish(!st.commit() ? st.err
                                : c2_factor_rev(1, N, J, st.p(t_), 0, st.p(c_), 0, st.p(a_), st.p(U_), st.p(V_), st.p(d_),
                                                st.p(W_), st.p(S_), st.p(bd_), st.p(bW_), st.p(bt_), st.p(bc_), st.p(ba_),
                                                st.p(bU_), st.p(bV_), st.stream()));
}

#define C2H_SWEEP_REV(NAME)                                                                                         \
  int c2h_##NAME(int64_t N, int64_t J, int64_t nrhs, const double *t, const double *c, const double *U,            \
                 const double *W, const double *Y, const double *Z, const double *F, const double *bZ, double *bt, \
                 double *bc, double *bU, double *bW, double *bY) {                                                 \
    if (bad(N, J) || nrhs < 1) return C2_ERR_INVALID;                                                               \
    if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;                                                                \
    Staging st;                                                                                                     \
    const int t_ = st.in(t, N), c_ = st.in(c, J), U_ = st.in(U, N * J), W_ = st.in(W, N * J), Y_ = st.in(Y, N * nrhs), \
              Z_ = st.in(Z, N * nrhs), F_ = st.in(F, N * J * nrhs), bZ_ = st.in(bZ, N * nrhs);                      \
    const int bt_ = st.fresh(bt, N), bc_ = st.fresh(bc, J), bU_ = st.fresh(bU, N * J), bW_ = st.fresh(bW, N * J),  \
              bY_ = st.fresh(bY, N * nrhs);                                                                           \
    return st.finish(!st.commit() ? st.err                                                                          \
                                  : c2_##NAME(1, N, J, nrhs, st.p(t_), 0, st.p(c_), 0, st.p(U_), st.p(W_), st.p(Y_), \
                                              st.p(Z_), st.p(F_), st.p(bZ_), st.p(bt_), st.p(bc_), st.p(bU_),       \
                                              st.p(bW_), st.p(bY_), st.stream()));                                  \
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
  const int ar_ = Jr ? st.in(ar, Jr) : -1, ac_ = Jc ? st.in(ac, Jc) : -1, bc_ = Jc ? st.in(bc, Jc) : -1,
            dc_ = Jc ? st.in(dc, Jc) : -1, x_ = st.in(x, N), diag_ = st.in(diag, N);
  const int a_ = st.fresh(a, N), U_ = st.fresh(U, N * J), V_ = st.fresh(V, N * J);
  return st.finish(!st.commit() ? st.err
                                : c2_get_celerite_matrices(1, N, Jr, Jc, st.p(ar_), st.p(ac_), st.p(bc_), st.p(dc_), 0,
                                                           st.p(x_), 0, st.p(diag_), st.p(a_), st.p(U_), st.p(V_),
                                                           st.stream()));
}

// Give back the calling thread's staging resources (device arena, pinned bounce buffer, stream); the next c2h_* call of the
// thread allocates them again.
void c2h_release_thread_cache(void) { g_arena.release(); }

}  // extern "C"
