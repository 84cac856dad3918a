// c2_rscatter.hpp -- reduce-scatter of 8 per-lane partial sums over a group of 8 or 16 lanes (shared by the chunked scan
// of c2_scan.hip and the multi-rhs reverse sweeps of c2_sweep_rev.hip).
#pragma once
#include "c2_common.hpp"

namespace c2 {

// ---- reduce-scatter of 8 per-lane partial sums over a group of G = 8 or 16 lanes ---------------------------
// An all-reduce of each of the 8 values costs 8 x log2(G) butterfly levels of (2 DPP + 1 add).  Here every level
// halves the number of live values instead: a lane keeps the half selected by one bit of its index and trades the
// other half with its partner, so that after log2(8) levels each lane holds ONE fully reduced value (index
// `kout`, a function of the lane bits) -- 8+4+2(+1) exchanges instead of 8 x log2(G), and the eight results of a
// row end up in eight different lanes, ready for a single contiguous store.
constexpr int kRowShl4 = 0x104, kRowShr4 = 0x114, kRowRor8 = 0x128;
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_upd(double old, double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, 0xf, BANK, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, 0xf, BANK, false);
  return __hiloint2double(hi, lo);
}
// value held by lane (l ^ 4): lanes with bit 2 clear read `a` from l+4, lanes with bit 2 set read `b` from l-4
__device__ __forceinline__ double xor4_recv(double a, double b) {
  const double t = dpp_upd<kRowShl4, 0x5>(0.0, a);
  return dpp_upd<kRowShr4, 0xA>(t, b);
}
template <int G>
__device__ __forceinline__ double rscatter8(const double (&v)[8], int j, int &kout) {
  static_assert(G == 8 || G == 16, "reduce-scatter is specialised for 8- and 16-lane groups");
  double r[4];
  if constexpr (G == 16) {  // level xor 8 (row_ror:8): 8 -> 4 values
    const bool hi = (j & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double keep = hi ? v[i + 4] : v[i], send = hi ? v[i] : v[i + 4];
      r[i] = keep + dpp_mov<kRowRor8>(send);
    }
    const bool b2 = (j & 4) != 0;  // level xor 4: 4 -> 2
    double q[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) q[i] = (b2 ? r[i + 2] : r[i]) + xor4_recv(r[i], r[i + 2]);
    const bool b1 = (j & 2) != 0;  // level xor 2: 2 -> 1
    double sres = (b1 ? q[1] : q[0]) + dpp_mov<kDppXor2>(b1 ? q[0] : q[1]);
    sres += dpp_mov<kDppXor1>(sres);  // level xor 1: finish (both lanes of the pair hold the total)
    kout = (hi ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
    return sres;
  } else {
    const bool b2 = (j & 4) != 0;  // level xor 4: 8 -> 4
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (b2 ? v[i + 4] : v[i]) + xor4_recv(v[i], v[i + 4]);
    const bool b1 = (j & 2) != 0;  // level xor 2: 4 -> 2
    double q[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) q[i] = (b1 ? r[i + 2] : r[i]) + dpp_mov<kDppXor2>(b1 ? r[i] : r[i + 2]);
    const bool b0 = (j & 1) != 0;  // level xor 1: 2 -> 1
    const double sres = (b0 ? q[1] : q[0]) + dpp_mov<kDppXor1>(b0 ? q[0] : q[1]);
    kout = (b2 ? 4 : 0) + (b1 ? 2 : 0) + (b0 ? 1 : 0);
    return sres;
  }
}

}  // namespace c2
