// c2_loglik_helpers.hpp -- device helpers shared by the fused log-likelihood kernels (c2_loglik.hip: one column per
// lane; c2_loglik4.hip: two columns per lane).
#pragma once
#include <cstdint>
#include "c2_dispatch.hpp"
#include "c2_common.hpp"

namespace c2 {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
constexpr double kLn2 = 0.69314718055994530941723212145818;
// One-lane-per-series gradient path (c2_loglik_t.hip): the backward recursion is used only while
// max_j c_j * (t_end - t_start) <= kBackwardGuard on every checkpoint segment (error growth <= e^{2 * guard}).
constexpr double kBackwardGuard = 2.0;
// Kernels of a fallback chain take the guard word as `gate` and run only if the fast path declined (nullptr: always run).
// Two forms, told apart by the lowest bit of the pointer (the words are 8-byte aligned):
//   plain pointer : ONE word decides for the whole launch (the verification words of the time-parallel forms);
//   pointer | 1   : one word per group of 64 consecutive series -- the one-lane-per-series kernels decide per WAVEFRONT
//                   (a wavefront that runs out of re-anchoring checkpoints costs its own 64 series the fallback, not the
//                   batch): the part of a fallback kernel that works on series b runs iff word[b / 64] is beyond the guard.
constexpr int kGateHeadWords = 2;   // [0] largest guard of the launch (diagnostic), [1] wavefronts that fell back; then the words
__host__ __device__ inline const unsigned long long *gate_per_wave(const unsigned long long *words) {
  return reinterpret_cast<const unsigned long long *>(reinterpret_cast<uintptr_t>(words) | 1u);
}
__device__ __forceinline__ bool gate_closed(const unsigned long long *gate, int64_t b = 0) {
  if (!gate) return false;
  const uintptr_t g = reinterpret_cast<uintptr_t>(gate);
  const unsigned long long w = (g & 1u) ? reinterpret_cast<const unsigned long long *>(g & ~(uintptr_t)1)[b >> 6] : *gate;
  return !(__longlong_as_double((long long)w) > kBackwardGuard);
}

// Per-group form only: did NO group of the launch fall back?  (head[1] -- the word right in front of the groups' words -- counts
// them.)  Lets a fallback kernel with a large grid return before it looks at its series.
__device__ __forceinline__ bool gate_none_closed(const unsigned long long *gate) {
  const uintptr_t g = reinterpret_cast<uintptr_t>(gate);
  if (!gate || !(g & 1u)) return false;
  return reinterpret_cast<const unsigned long long *>(g & ~(uintptr_t)1)[-1] == 0ull;
}

// 1/d for a well-scaled positive d: v_rcp_f64 seed + two Newton steps (full fp64 accuracy; the
// denormal/overflow scaling of a general IEEE division is not needed for pivots of an SPD matrix).
__device__ __forceinline__ double rcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  return r;
}

// exp(x) for the decay factors p = exp(c (t_{n-1} - t_n)), x <= 0 in every valid call.
// k = rint(x log2 e), r = x - k ln2 (two-term Cody-Waite), degree-11 near-minimax polynomial on
// |r| <= ln2/2 (Chebyshev fit, exact-arithmetic error 4e-18, 1 ulp in double Horner), result scaled by
// v_ldexp_f64 (which also flushes the underflow range to 0).  16 VALU instructions.
// q*r + c with the constant forced into an SGPR pair (VOP3 v_fma_f64 takes one scalar operand).  Left to
// itself hipcc keeps the eleven coefficients in VGPRs and emits v_mov_b64 + v_fmac_f64 per Horner step.
__device__ __forceinline__ double fma_sconst(double q, double r, double c) {
  double o;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(q), "v"(r), "s"(c));
  return o;
}
__device__ __forceinline__ double exp_decay(double x) {
  x = (x < -1000.0) ? -1000.0 : x;  // clamp the underflow range; a NaN in t or c propagates (fmax would drop it)
  const double k = rint(x * 1.4426950408889634074);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double q = fma_sconst(2.51100492048186583e-08, r, 2.76326547225277896e-07);
  q = fma_sconst(q, r, 2.75572408872298695e-06);
  q = fma_sconst(q, r, 2.48014854415613131e-05);
  q = fma_sconst(q, r, 1.98412698900764028e-04);
  q = fma_sconst(q, r, 1.38888889523528631e-03);
  q = fma_sconst(q, r, 8.33333333331958900e-03);
  q = fma_sconst(q, r, 4.16666666664879531e-02);
  q = fma_sconst(q, r, 1.66666666666666796e-01);
  q = fma_sconst(q, r, 5.00000000000001887e-01);
  q = fma(q, r, 1.0);
  q = fma(q, r, 1.0);
  return ldexp(q, (int)k);
}

// sin and cos of the phase of a complex term, dc * x (driver.cpp:466-467), for the kernels that generate U and V rows
// on the fly.  Cody-Waite reduction by pi/2 in three fused steps (pi/2 = hi + mid + lo to ~160 bits; the first product
// is exact inside the fma), then the fdlibm kernel polynomials on |r| <= pi/4 (sin: degree 13, cos: degree 14; < 1 ulp).
// ~32 VALU instructions against ~100 for the library call; arguments beyond 2^20 quarter turns take the library path.
constexpr double kSincosFastMax = 1.6e6;
// |x| < kSincosFastMax only (the caller guarantees it; no branch, no library code in the caller's loop)
__device__ __forceinline__ void sincos_cw_fast(double x, double &sn, double &cs) {
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(k, -1.57079632679489655800e+00, x);
  r = fma(k, -6.12323399573676603587e-17, r);
  r = fma(k, 1.49738490485916983e-33, r);
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(ps, z, 2.75573137070700676789e-06);
  ps = fma(ps, z, -1.98412698298579493134e-04);
  ps = fma(ps, z, 8.33333333332248946124e-03);
  ps = fma(ps, z, -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(pc, z, -2.75573143513906633035e-07);
  pc = fma(pc, z, 2.48015872894767294178e-05);
  pc = fma(pc, z, -1.38888888888741095749e-03);
  pc = fma(pc, z, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k & 3;
  const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
  sn = (q & 2) ? -a : a;
  cs = ((q + 1) & 2) ? -b : b;
}
__device__ __forceinline__ void sincos_cw(double x, double &sn, double &cs) {
  if (!(fabs(x) < kSincosFastMax)) {  // also NaN / inf
    sincos(x, &sn, &cs);
    return;
  }
  sincos_cw_fast(x, sn, cs);
}

__device__ __forceinline__ void lds_order() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// -----------------------------------------------------------------------------------------------
// XOR-ordered group gathers.  Every width-J object a lane keeps in registers is stored in XOR order:
// slot k of lane j holds element (j ^ k), so that a gathered vector x becomes xX[k] = x[j ^ k] and all
// contractions sum_i a_i B(i,j) read sum_k aX[k] BX[k].  Two ways to gather:
//   xgather_dpp : on the VALU with DPP lane permutes (7 permutes of a double at G=8) -- low latency, used
//                 for the ONE vector per step that sits on the recursion's critical path;
//   xgather_lds : through LDS (lane j reads slot[(lane ^ k)], G conflict-free ds_read_b64) -- free for
//                 the VALU, used for vectors known ahead of time (p, U_n, saved W_{n-1}).
// -----------------------------------------------------------------------------------------------
constexpr int kDppXor3 = 0x1B;  // quad_perm [3,2,1,0]

template <int G>
__device__ __forceinline__ void xgather_lds(const double *slot, int lane, double (&out)[G]) {
#pragma unroll
  for (int k = 0; k < G; ++k) out[k] = slot[lane ^ k];
}

template <int G>
__device__ __forceinline__ void xgather_dpp(double x, double *xslot, int lane, double (&out)[G]) {
  out[0] = x;
  if constexpr (G == 2) {
    out[1] = dpp_mov<kDppXor1>(x);
  } else if constexpr (G == 4 || G == 8 || G == 16) {
    out[1] = dpp_mov<kDppXor1>(x);
    out[2] = dpp_mov<kDppXor2>(x);
    out[3] = dpp_mov<kDppXor3>(x);
    if constexpr (G >= 8) {
      const double y = dpp_mov<kDppHalfMirror>(x);  // lane l <- l ^ 7
      out[7] = y;
      out[6] = dpp_mov<kDppXor1>(y);
      out[5] = dpp_mov<kDppXor2>(y);
      out[4] = dpp_mov<kDppXor3>(y);
    }
    if constexpr (G == 16) {
      const double z = dpp_mov<kDppMirror>(x);      // lane l <- l ^ 15
      out[15] = z;
      out[14] = dpp_mov<kDppXor1>(z);
      out[13] = dpp_mov<kDppXor2>(z);
      out[12] = dpp_mov<kDppXor3>(z);
      const double zy = dpp_mov<kDppHalfMirror>(z);  // lane l <- l ^ 8
      out[8] = zy;
      out[9] = dpp_mov<kDppXor1>(zy);
      out[10] = dpp_mov<kDppXor2>(zy);
      out[11] = dpp_mov<kDppXor3>(zy);
    }
  } else if constexpr (G == 32) {  // crosses DPP rows: go through LDS
    xslot[lane] = x;
    lds_order();
    xgather_lds<G>(xslot, lane, out);
  }
}

// Lane geometry shared by the forward and reverse kernels: wave-uniform bases + small per-lane offsets,
// so that global addresses are (SGPR base) + (VGPR offset) + (immediate) and cost no VALU per load.
template <int G>
struct Geo {
  int lane, j, jj;
  bool valid, act;
  int64_t b0;     // first series of this wavefront (uniform)
  int64_t b;      // this lane's series (clamped)
  int sl;         // series index inside the wavefront (clamped)
  __device__ __forceinline__ Geo(int64_t B, int J) {
    constexpr int SPW = kWave / G;
    lane = threadIdx.x;
    j = lane & (G - 1);
    b0 = (int64_t)blockIdx.x * SPW;
    sl = lane / G;
    const int64_t maxsl = B - 1 - b0;
    valid = sl <= maxsl;
    if (!valid) sl = (int)maxsl;
    b = b0 + sl;
    act = j < J;
    jj = act ? j : 0;
  }
};

}  // namespace c2
