// c2_common.hpp -- device-side helpers shared by the gfx950 kernels.
//
// Execution mapping used by every recursion kernel in this library
// ("sub-wave packing"): a series is walked by a GROUP of G consecutive lanes
// of one 64-lane wavefront, G = J rounded up to a power of two (1..32), so a
// wavefront carries 64/G independent series.  Lane j of a group owns index j of
// the width-J objects: column j of the J x J factor state S, row j of the
// J x nrhs sweep state F.  Quantities that are scalars of the recursion (d_n,
// z_n, ...) are held redundantly by all G lanes.  Cross-lane traffic is limited
// to group all-reduces (DPP butterflies) and group broadcasts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

namespace c2 {

constexpr int kWave = 64;

// ---- DPP lane permutations on a 64-bit value (two 32-bit DPP moves) --------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // mov_dpp with bound_ctrl: no tied "old" operand, so no extra v_mov to initialise the destination.
  // Every permutation used here stays inside a DPP row, so no lane ever reads an invalid source.
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // row_half_mirror: lane i <- 7-i  (within 8)
constexpr int kDppMirror = 0x140;     // row_mirror:      lane i <- 15-i (within 16)

// All-reduce (sum) over each aligned group of G lanes; every lane of the group
// receives the sum.  Levels 1,2 use quad permutes, 4 and 8 the mirror patterns
// (legal because after the earlier levels all lanes of the smaller block already
// hold the same partial sum), level 16 goes through ds_bpermute.
template <int G>
__device__ __forceinline__ double gsum(double x) {
  if constexpr (G >= 2) x += dpp_mov<kDppXor1>(x);
  if constexpr (G >= 4) x += dpp_mov<kDppXor2>(x);
  if constexpr (G >= 8) x += dpp_mov<kDppHalfMirror>(x);
  if constexpr (G >= 16) x += dpp_mov<kDppMirror>(x);
  if constexpr (G >= 32) x += __shfl_xor(x, 16, kWave);
  return x;
}

// Two / three independent all-reduces with their butterfly levels interleaved, so that the DPP moves of one
// value fill the wait states (VALU write -> DPP read) and the add latency of the others.
template <int G>
__device__ __forceinline__ void gsum2(double &a, double &b) {
  if constexpr (G >= 2) { const double ta = dpp_mov<kDppXor1>(a), tb = dpp_mov<kDppXor1>(b); a += ta; b += tb; }
  if constexpr (G >= 4) { const double ta = dpp_mov<kDppXor2>(a), tb = dpp_mov<kDppXor2>(b); a += ta; b += tb; }
  if constexpr (G >= 8) { const double ta = dpp_mov<kDppHalfMirror>(a), tb = dpp_mov<kDppHalfMirror>(b); a += ta; b += tb; }
  if constexpr (G >= 16) { const double ta = dpp_mov<kDppMirror>(a), tb = dpp_mov<kDppMirror>(b); a += ta; b += tb; }
  if constexpr (G >= 32) { a += __shfl_xor(a, 16, kWave); b += __shfl_xor(b, 16, kWave); }
}
template <int G>
__device__ __forceinline__ void gsum3(double &a, double &b, double &c) {
  if constexpr (G >= 2) { const double ta = dpp_mov<kDppXor1>(a), tb = dpp_mov<kDppXor1>(b), tc = dpp_mov<kDppXor1>(c); a += ta; b += tb; c += tc; }
  if constexpr (G >= 4) { const double ta = dpp_mov<kDppXor2>(a), tb = dpp_mov<kDppXor2>(b), tc = dpp_mov<kDppXor2>(c); a += ta; b += tb; c += tc; }
  if constexpr (G >= 8) { const double ta = dpp_mov<kDppHalfMirror>(a), tb = dpp_mov<kDppHalfMirror>(b), tc = dpp_mov<kDppHalfMirror>(c); a += ta; b += tb; c += tc; }
  if constexpr (G >= 16) { const double ta = dpp_mov<kDppMirror>(a), tb = dpp_mov<kDppMirror>(b), tc = dpp_mov<kDppMirror>(c); a += ta; b += tb; c += tc; }
  if constexpr (G >= 32) { a += __shfl_xor(a, 16, kWave); b += __shfl_xor(b, 16, kWave); c += __shfl_xor(c, 16, kWave); }
}

// Value of x held by lane i of the caller's group.
template <int G>
__device__ __forceinline__ double gget(double x, int i) {
  if constexpr (G == 1) return x;
  return __shfl(x, i, G);
}

// Stream-ordered temporaries of the library.  They come from a memory pool the LIBRARY owns (one per device, created on
// first use, release threshold 1 GiB so that up to that much stays cached between calls: the default pool hands its memory
// back to the driver at every synchronisation and each call would pay a fresh allocation -- 0.2 ms for the 17 MB of a
// 64 x 4096 x 64 many-rhs solve that itself takes 0.4 ms).  The device's DEFAULT pool, which the process shares with
// everybody else, is left as it is; if the pool cannot be created the temporaries fall back to it, uncached.  Thread-safe
// (std::call_once per device); hipFreeAsync releases either kind.
inline hipError_t temp_alloc(void **p, size_t bytes, hipStream_t s) {
  constexpr int kMaxDev = 64;
  static std::once_flag once[kMaxDev];
  static hipMemPool_t pools[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return hipMallocAsync(p, bytes, s);
  std::call_once(once[dev], [dev] {
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool) {
      uint64_t keep = 1ull << 30;
      (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
      pools[dev] = pool;
    }
    (void)hipGetLastError();
  });
  if (pools[dev]) return hipMallocFromPoolAsync(p, bytes, pools[dev], s);
  return hipMallocAsync(p, bytes, s);
}

// Group-size dispatch: G = next power of two >= J.
inline int group_size(int64_t J) {
  int G = 1;
  while (G < J) G <<= 1;
  return G;
}

}  // namespace c2
