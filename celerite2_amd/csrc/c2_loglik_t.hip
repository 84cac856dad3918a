// c2_loglik_t.hip -- fused log-likelihood (+ reverse-mode gradient) with ONE LANE PER SERIES, for batches large
// enough to fill the chip that way (>= 24576 series).  Widths J = 8, 4, 2 (one compilation each: C2T_J) and 6 (computed
// as 8 with two empty columns: C2T_JS).
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

// The file is compiled once per width: C2T_J = 8 (default), 4, 2 (c2_loglik_t4.hip / c2_loglik_t2.hip include it).  A
// tile of the width-J streams is always ONE 128-byte line per series: RT = 16 / J rows.
// C2T_JS < C2T_J (c2_loglik_t6.hip: 6 in 8): the arrays in memory have rows of JS doubles, the lane computes on rows of
// J with the missing columns zero (U = V = c = 0: they decouple, P = 1 keeps them zero) -- only the tile movers and the
// loads of c know.  Rows of 48 bytes do not tile a 128-byte line, so the runs are no longer aligned lines.
#ifndef C2T_J
#define C2T_J 8
#endif
#ifndef C2T_JS
#define C2T_JS C2T_J
#endif
#define C2T_CAT2(a, b) a##b
#define C2T_CAT(a, b) C2T_CAT2(a, b)
#define C2T_NAME(stem) C2T_CAT(stem, C2T_JS)  // c2_internal_loglik_t -> c2_internal_loglik_t8
#define c2t C2T_CAT(c2t_j, C2T_JS)            // one namespace per width

namespace c2t {
using namespace c2;

constexpr int J = C2T_J;
static_assert(J == 8 || J == 4 || J == 2, "128-byte tiles of whole rows");
constexpr int NS = J * (J + 1) / 2;  // packed symmetric J x J
constexpr int PPR = J / 2;           // 16-byte pieces per row
constexpr int JS = C2T_JS;           // row length of the arrays in memory
static_assert(JS <= J && JS % 2 == 0, "whole 16-byte pieces");
constexpr int PPS = JS / 2;          // pieces of a row that exist in memory
// offset (doubles) of piece p of a tile of RT rows inside a series, and whether the piece exists
__device__ __forceinline__ constexpr int piece_off(int p) { return (p / PPR) * JS + 2 * (p % PPR); }
__device__ __forceinline__ constexpr bool piece_in(int p) { return JS == J || (p % PPR) < PPS; }
#ifndef C2T_C
#define C2T_C 32
#endif
#ifndef C2T_M2
#define C2T_M2 1   // the rank-2 update of M in two fma per element (0: three, x in the registers of bV)
#endif
constexpr int C = C2T_C;             // checkpoint interval (rows)
constexpr double kGuard = kBackwardGuard;  // largest allowed max_j c_j * (t_end - t_start) of a segment
constexpr int RT = 16 / J;           // rows per tile of the width-J streams: one aligned 128-byte line per series
constexpr int LPS = RT * J / 2;      // lanes that fetch one series' run (16 bytes each)
constexpr int NI = LPS;              // global instructions per row tile (64 / LPS series each)
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
//   CK  : [wave][slot][NS + J][64] double (state after row n_slot: S packed, F), slots in the order written
//   CKR : [wave][n] int32                 (slot of the checkpoint of row n, meaningful where the row carries one -- which is
//                                          the sign of the row's d in DZ)
// C2T_TM = 1: the per-row records (W, DZ, T) are TIME-major across the wavefronts of the launch -- [n][wave][..][64] -- so
// that the rows the resident wavefronts touch at any moment (they walk the series in near lockstep) form ONE contiguous
// stretch of memory, spread evenly over every HBM stack / channel whatever the allocator's placement; 0: [wave][n][..][64]
// (each wavefront its own stretch, 16 MB apart at the bench shape).
#ifndef C2T_TM
#define C2T_TM 0
#endif
#ifndef C2T_RECFIRST
#define C2T_RECFIRST 0
#endif
struct Rec {
  size_t w, dz, ck, ckr, t, total;  // offsets / total in doubles
  int64_t nck;                      // checkpoint slots per wavefront (capacity)
  int64_t nreg;                     // ... of which the regular ones (rows C, 2C, ..., N-1)
};
// Regular checkpoints = state after rows C, 2C, ... and after the last row N-1: ceil((N-1)/C) of them.  EXTRA checkpoints
// re-anchor the backward recursion wherever it would otherwise have to invert a decay it cannot (a gap in time: a night,
// a season): before row n is absorbed, if for ANY series of the wavefront max_j c_j * (t_n - t of the row behind its last
// checkpoint) exceeds kGuard, the state of row n-1 is recorded -- for the whole wavefront, so the record stays lane-major
// and the reverse sweep turns at wavefront-uniform rows.  After a long gap the decays are ~0 and the state is all but
// reset, so nothing is lost by not walking across it.  Up to 2 nreg extras per wavefront (shared grids and batches with a
// few gappy series need a handful; 64 series with three gaps each at rows of their own 192); a wavefront that runs out
// leaves its excess in ITS guard word: its reverse sweep returns at once and the replay kernels, gated per group of 64
// series (gate_closed, c2_loglik_helpers.hpp), take those 64 series -- not the batch.
__host__ __device__ inline int64_t n_ckpt(int64_t N) { return N >= 2 ? (N - 2) / C + 1 : 0; }
__host__ inline Rec rec_layout(int64_t B, int64_t N) {
  const size_t waves = ((size_t)B + kWave - 1) / kWave;
  Rec r;
  r.nreg = n_ckpt(N);
  r.nck = 3 * r.nreg;   // twice as many extras as regular ones: every series of a wavefront its own three gaps at N = 4096
  r.w = 0;
  r.dz = r.w + waves * (size_t)N * J * kWave;
  r.ck = r.dz + waves * (size_t)N * 2 * kWave;
  r.ckr = r.ck + waves * (size_t)r.nck * (NS + J) * kWave;
  r.t = r.ckr + waves * (((size_t)N + 1) / 2);       // one int32 per row, 8-byte aligned per wavefront
  r.total = r.t + waves * (size_t)N * kWave;
  return r;
}

// ---- tile movers -------------------------------------------------------------------------------------------------------
// A wavefront owns series b0 .. b0+63 (clamped to B-1).  Row tile: RT rows of J doubles of every series.  One global
// instruction moves 8 series x 128 bytes (lane l: series 8 i + l / 8, 16-byte piece l % 8).
// FULL: the wavefront is known to own 64 valid series -- no clamp, so the addresses of a tile are one per-lane vector
// plus uniform multiples of the series stride (with the clamp the compiler keeps one address vector per instruction
// alive across the sweep, 32+ registers that end up in scratch).
template <bool FULL>
struct RowIOT {
  int lane, last;
  int piece;       // l % 8 (scalar tiles)
  int rpiece;      // l % LPS (row tiles)
  __device__ __forceinline__ RowIOT(int lane_, int last_) : lane(lane_), last(last_), piece(lane_ & 7), rpiece(lane_ % LPS) {}
  // clamped series (within the wavefront) this lane serves in instruction i of a scalar / row tile -- recomputed at
  // every use (two VALU instructions) rather than kept in registers
  __device__ __forceinline__ int sl(int i) const { const int s = 8 * i + lane / 8; return (FULL || s < last) ? s : last; }
  __device__ __forceinline__ int rl(int i) const { const int s = (kWave / LPS) * i + lane / LPS; return (FULL || s < last) ? s : last; }
};
using RowIO = RowIOT<false>;

// Streaming hints (C2T_NT).  Every byte of this pair is touched ONCE -- inputs, records, gradients -- so nothing it streams needs to
// stay in L2; what DOES need to stay is the half-written lines of the scalar gradients: ba, by, bt leave the reverse sweep as 64-byte
// runs, and the other half of each 128-byte line follows eight steps (~30 us, ~20 MB of traffic through a 4 MB L2) later.  Evicted in
// between, each half is merged on the memory side on its own (read-modify-write).  tools/ubench/replay_traffic.hip -- the pair's memory
// operations without its arithmetic, which reproduces the kernels' times -- puts that at 11 - 15 % of the reverse sweep, and shows
// that the non-temporal hint on every OTHER stream is as good as whole-line stores (for which this kernel has no LDS left: 16-row
// scalar tiles need 13 KB more per wavefront): 15.8 -> 13.8 ms.  Level 1: the records and the width-J gradients (round 2's flag, then
// measured 'no gain' -- with the inputs still displacing the half lines).  Level 2 (default since round 6): the API rows and scalars
// the passes read as well.  Bench step, six alternating fresh processes on one box: 30.06 (28.8 - 31.1) -> 27.75 ms (27.2 - 28.9)
// (profiles/r06_halflines.md).  0: plain accesses (A/B builds).
#ifndef C2T_NT
#define C2T_NT 2
#endif
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2_in(const double2 *p) {
#if C2T_NT >= 2
  const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
  return make_double2(v.x, v.y);
#else
  return *p;
#endif
}
__device__ __forceinline__ double ld1_in(const double *p) {
#if C2T_NT >= 2
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
// global -> registers: rows n0 .. n0+RT-1 (clamped to [0, N-1]) of every series
// (staging registers are plain doubles: arrays of double2 end up in scratch)
template <class IO>
__device__ __forceinline__ void row_fetch(const double *__restrict__ base, int64_t N, int64_t n0, const IO &io,
                                          double (&st)[2 * NI]) {
  int64_t r = n0 + io.rpiece / PPR;
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
  const int pc = piece_in(io.rpiece) ? io.rpiece % PPR : 0;   // a missing piece reads piece 0 and is zeroed
  const int64_t off = r * JS + 2 * pc;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const double2 v = ld2_in(reinterpret_cast<const double2 *>(base + (int64_t)io.rl(i) * N * JS + off));
    st[2 * i] = piece_in(io.rpiece) ? v.x : 0.0; st[2 * i + 1] = piece_in(io.rpiece) ? v.y : 0.0;
  }
}
// registers -> LDS tile
__device__ __forceinline__ void row_stage(double *tile, int lane, const double (&st)[2 * NI]) {
#pragma unroll
  for (int i = 0; i < NI; ++i)
    *reinterpret_cast<double2 *>(tile + ((kWave / LPS) * i + lane / LPS) * RSTR + 2 * (lane % LPS)) =
        make_double2(st[2 * i], st[2 * i + 1]);
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
// LDS tile -> global: rows n0 .. n0+RT-1 of every series; rows outside [lo, hi] and series beyond `last` are skipped
__device__ __forceinline__ void row_flush(double *__restrict__ base, int64_t N, int64_t n0, int64_t lo, int64_t hi,
                                          const double *tile, int lane, int last) {
  const int piece = lane % LPS;
  const int64_t r = n0 + piece / PPR;
  if (last == kWave - 1 && n0 >= lo && n0 + RT - 1 <= hi) {  // uniform: full wavefront, whole tile in range
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int s = (kWave / LPS) * i + lane / LPS;
      if (piece_in(piece))
        *reinterpret_cast<double2 *>(base + (int64_t)s * N * JS + r * JS + 2 * (piece % PPR)) =
            *reinterpret_cast<const double2 *>(tile + s * RSTR + 2 * piece);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int s = (kWave / LPS) * i + lane / LPS;
    const double2 v = *reinterpret_cast<const double2 *>(tile + s * RSTR + 2 * piece);
    if (s <= last && r >= lo && r <= hi && piece_in(piece))
      *reinterpret_cast<double2 *>(base + (int64_t)s * N * JS + r * JS + 2 * (piece % PPR)) = v;
  }
}

// The same for a full wavefront and a tile that lies inside the series: no tests, no branches.
__device__ __forceinline__ void row_flush_full(double *__restrict__ base, int64_t N, int64_t n0, const double *tile,
                                               int lane) {
  const int piece = lane % LPS;
  const int64_t off = (n0 + piece / PPR) * JS + 2 * (piece % PPR);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int s = (kWave / LPS) * i + lane / LPS;
    if (piece_in(piece))
      *reinterpret_cast<double2 *>(base + (int64_t)s * N * JS + off) =
          *reinterpret_cast<const double2 *>(tile + s * RSTR + 2 * piece);
  }
}

// Scalar tile: ST rows of one double of every series; one instruction moves 8 series x 64 bytes.
__device__ __forceinline__ void sc_fetch(const double *__restrict__ base, int64_t N, int64_t n0, const RowIO &io,
                                         double (&st)[8]) {
  int64_t r = n0 + io.piece;
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
#pragma unroll
  for (int i = 0; i < 8; ++i) st[i] = base[(int64_t)io.sl(i) * N + r];
}
__device__ __forceinline__ void sc_stage(double *tile, int lane, const double (&st)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) tile[(8 * i + lane / 8) * SSTR + (lane & 7)] = st[i];
}
// The same streams as 16-ROW tiles (the forward pass): one instruction moves 4 series x 128 bytes -- whole aligned lines,
// fetched once (two 64-byte halves requested eight rows apart cost the line twice: it has left L2 by then, +24 B per
// series-step, 6.4 GB per step at the bench shape).  The tile stays in 16 registers per stream and reaches the 8-row LDS
// tile one half at a time.  sN: series stride of the array (0: shared by the batch).
__device__ __forceinline__ void sc_fetch16(const double *__restrict__ base, int64_t sN, int64_t N, int64_t n16, int lane,
                                           int last, double (&st)[16]) {
  int64_t r = n16 + (lane & 15);
  r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int s = 4 * i + lane / 16;
    st[i] = ld1_in(&base[(int64_t)(s < last ? s : last) * sN + r]);
  }
}
__device__ __forceinline__ void sc_stage16(double *tile, int lane, int half, const double (&st)[16]) {
  if (((lane & 15) >> 3) == half) {
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[(4 * i + lane / 16) * SSTR + (lane & 7)] = st[i];
  }
}
__device__ __forceinline__ void sc_flush(double *__restrict__ base, int64_t N, int64_t n0, int64_t lo, int64_t hi,
                                         const double *tile, int lane, int last) {
  const int64_t r = n0 + (lane & 7);
  if (last == kWave - 1 && n0 >= lo && n0 + ST - 1 <= hi) {  // uniform: full wavefront, whole tile in range
    double v[8];  // all reads first, then all stores (one reused register would serialise eight LDS latencies)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = tile[(8 * i + lane / 8) * SSTR + (lane & 7)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) base[(int64_t)(8 * i + lane / 8) * N + r] = v[i];
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int s = 8 * i + lane / 8;
    const double v = tile[s * SSTR + (lane & 7)];
    if (s <= last && r >= lo && r <= hi) base[(int64_t)s * N + r] = v;
  }
}

// A double parked in a pair of accumulation registers (see c2_loglik.hip): VALU instructions cannot read AGPRs, but
// the 256 a lane owns at one wavefront per SIMD are otherwise idle -- the reverse sweep keeps the forward state S
// there (72 of them) and pays two v_accvgpr moves per access, leaving the arithmetic registers to M and the vectors.
// `volatile`: the moves keep their program order, so the scheduler cannot hoist all 144 reads of a step to its top
// (which is what an ILP-driven schedule does with them, and what made the first version of this kernel spill).
__device__ __forceinline__ void apark(double x, int &lo, int &hi) {
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(__double2loint(x)));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(__double2hiint(x)));
}
// A global load straight into an AGPR (no arithmetic register in flight): dst = *(int *)(sbase + voff + IMM).  The
// compiler does not see it as a load -- the consumer waits with await_aloads().  (Its own s_waitcnt's stay correct: the
// counter is in order, an extra outstanding load only makes them wait longer.)
template <int IMM>
__device__ __forceinline__ void aload(int &dst, const double *sbase, unsigned voff) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3 nt" : "=a"(dst) : "v"(voff), "s"(sbase), "n"(IMM));
}
__device__ __forceinline__ void await_aloads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double afetch(int lo, int hi) {
  int l, h;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
  return __hiloint2double(h, l);
}

#ifndef C2T_MLDS
#define C2T_MLDS 0   // half of M in LDS for the coefficient-level sweep: measured, 3 - 5 % SLOWER (20.4 against 19.8 ms at 65536 series)
#endif
// the records (written once by the forward pass, read once by the reverse sweep) and the width-J gradients: C2T_NT >= 1
__device__ __forceinline__ double2 ld2_stream(const double2 *p) {
#if C2T_NT
  const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
  return make_double2(v.x, v.y);
#else
  return *p;
#endif
}
__device__ __forceinline__ double ld1_stream(const double *p) {
#if C2T_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st2_stream(double2 *p, double2 v) {
#if C2T_NT
  d2v w; w.x = v.x; w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st1_stream(double *p, double v) {
#if C2T_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
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

// Section timing of the reverse step (diagnostic builds only: tools/build_variant.sh prof c2_loglik_t.hip -DC2T_PROF).
#ifdef C2T_PROF
__device__ unsigned long long c2t_prof[8];
#define C2T_TICK(k)                                                    \
  do {                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long now_ = __builtin_readcyclecounter();      \
    prof_[k] += now_ - tick_;                                          \
    tick_ = now_;                                                      \
    __builtin_amdgcn_sched_barrier(0);                                 \
  } while (0)
#else
#define C2T_TICK(k) do {} while (0)
#endif

// ---- rows generated from the celerite coefficients (driver.cpp:456-474) ----------------------------------------------------
// JC >= 0 turns a kernel into its coefficient-level form: the series has JR = J - 2 JC real and
// JC complex terms, U_n and V_n are formed in the lane from (ar, ac, bc, dc) and x_n (one sincos per complex term), the
// `a` argument is the white-noise diagonal (a_n = diag_n + sum ar + sum ac), the rates are c = [cr, cc0, cc0, ...].
// No U / V rows are read: 24 bytes of input per step instead of 152.  JC = -1: rows come from the caller's arrays.
struct TermsArgs {
  const double *ar, *cr, *ac, *bc, *cc, *dc;
  int batched;  // coefficients per series (1) or shared by the batch (0)
};
template <int JC>
struct TermCoef {
  static constexpr int JR = J - 2 * (JC > 0 ? JC : 0);
  double ar[JR > 0 ? JR : 1], ac[JC > 0 ? JC : 1], bc[JC > 0 ? JC : 1], dc[JC > 0 ? JC : 1], A0;
  __device__ __forceinline__ void load(const TermsArgs &T, int64_t b, double (&cj)[J]) {
    const int64_t br = T.batched ? b * JR : 0, bk = T.batched ? b * JC : 0;
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < JR; ++r) { ar[r] = T.ar[br + r]; cj[r] = T.cr[br + r]; sum += ar[r]; }
#pragma unroll
    for (int k = 0; k < JC; ++k) {
      ac[k] = T.ac[bk + k]; bc[k] = T.bc[bk + k]; dc[k] = T.dc[bk + k];
      cj[JR + 2 * k] = cj[JR + 2 * k + 1] = T.cc[bk + k];
      sum += ac[k];   // driver.cpp:456-458: sum of ar, then of ac
    }
    A0 = sum;
  }
  // FAST: every phase dc_k x of the wavefront is known to lie inside the range of the branch-free reduction (phases_fast
  // below); otherwise the library's sincos with its large-argument reduction -- as two instantiations of the whole
  // kernel body, so that the common case carries neither the branch nor the library code in its loop.
  template <bool FAST>
  static __device__ __forceinline__ void sc(double ph, double &sn, double &cs) {
    if constexpr (FAST) sincos_cw_fast(ph, sn, cs);
    else sincos(ph, &sn, &cs);
  }
  // t sorted: the largest |x| of a series sits at one of its ends
  __device__ __forceinline__ bool phases_fast(double x_first, double x_last) const {
    const double xm = fmax(fabs(x_first), fabs(x_last));
    bool ok = true;
#pragma unroll
    for (int k = 0; k < JC; ++k) ok = ok && (fabs(dc[k]) * xm < kSincosFastMax);
    return ok;
  }
  template <bool FAST>
  __device__ __forceinline__ void rows(double xn, double (&u)[J], double (&v)[J]) const {
#pragma unroll
    for (int r = 0; r < JR; ++r) { u[r] = ar[r]; v[r] = 1.0; }
#pragma unroll
    for (int k = 0; k < JC; ++k) {
      double sn, cs;
      sc<FAST>(dc[k] * xn, sn, cs);
      v[JR + 2 * k] = cs; v[JR + 2 * k + 1] = sn;
      u[JR + 2 * k] = fma(ac[k], cs, bc[k] * sn);
      u[JR + 2 * k + 1] = fma(ac[k], sn, -(bc[k] * cs));
    }
  }
  // the same, keeping sin / cos of every complex term for the reverse of the recipe
  template <bool FAST>
  __device__ __forceinline__ void rows_sc(double xn, double (&u)[J], double (&sn)[JC > 0 ? JC : 1],
                                          double (&cs)[JC > 0 ? JC : 1]) const {
#pragma unroll
    for (int r = 0; r < JR; ++r) u[r] = ar[r];
#pragma unroll
    for (int k = 0; k < JC; ++k) {
      sc<FAST>(dc[k] * xn, sn[k], cs[k]);
      u[JR + 2 * k] = fma(ac[k], cs[k], bc[k] * sn[k]);
      u[JR + 2 * k + 1] = fma(ac[k], sn[k], -(bc[k] * cs[k]));
    }
  }
  // decays: one exponential per real column and per complex PAIR
  __device__ __forceinline__ void decay(const double (&cj)[J], double dt, double (&p)[J]) const {
#pragma unroll
    for (int r = 0; r < JR; ++r) p[r] = exp_decay(cj[r] * dt);
#pragma unroll
    for (int k = 0; k < JC; ++k) p[JR + 2 * k] = p[JR + 2 * k + 1] = exp_decay(cj[JR + 2 * k] * dt);
  }
};

// =============================================================================================================
// Forward pass.  REC = false: log-likelihood only.  REC = true: also W rows, (d, z) pairs, checkpoints and the
// stability guard of the backward recursion.
// =============================================================================================================
template <bool REC, bool PAIRED, int JC = -1, bool FAST = true>
__device__ __forceinline__ void fwd_body(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                         const double *__restrict__ c, int64_t c_bs, const double *__restrict__ a,
                                         const double *__restrict__ U, const double *__restrict__ V,
                                         const double *__restrict__ y, double *__restrict__ ll,
                                         int32_t *__restrict__ flag, double *__restrict__ rec, Rec R,
                                         unsigned long long *__restrict__ guard, double *lds,
                                         const TermsArgs T = TermsArgs{}) {
  constexpr bool TERMS = JC >= 0;
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kWave;
  const int last = (int)((B - 1 - b0) < (kWave - 1) ? (B - 1 - b0) : (kWave - 1));
  const int sl = lane < last ? lane : last;
  const int64_t b = b0 + sl;
  const RowIO io(lane, last);
  double *tU = lds, *tV = tU + kWave * RSTR, *tT = tV + kWave * RSTR, *tA = tT + kWave * SSTR, *tY = tA + kWave * SSTR;
  const double *Ub = U + b0 * N * JS, *Vb = V + b0 * N * JS, *ab = a + b0 * N, *yb = y + b0 * N;
  // shared t: every series reads the same grid (stride 0 between series)
  const double *tb = t + (t_bs ? b0 * N : 0);
  const int64_t tN = t_bs ? N : 0;  // series stride of t

  double cj[J];
  TermCoef<JC> tc;
  if constexpr (TERMS) tc.load(T, b, cj);
  else {
#pragma unroll
    for (int j = 0; j < J; ++j) cj[j] = j < JS ? c[b * c_bs + j] : 0.0;
  }
  double cmax = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j) cmax = fmax(cmax, cj[j]);

  // records of this wavefront
#if C2T_TM
  const size_t rsW = (size_t)gridDim.x * (J / 2) * kWave, rs1 = (size_t)gridDim.x * kWave;   // row strides (elements)
  double2 *recW = REC ? reinterpret_cast<double2 *>(rec + R.w) + (size_t)blockIdx.x * (J / 2) * kWave : nullptr;
  double2 *recDZ = REC ? reinterpret_cast<double2 *>(rec + R.dz) + (size_t)blockIdx.x * kWave : nullptr;
#else
  constexpr size_t rsW = (size_t)(J / 2) * kWave, rs1 = kWave;
  double2 *recW = REC ? reinterpret_cast<double2 *>(rec + R.w + (size_t)blockIdx.x * N * J * kWave) : nullptr;
  double2 *recDZ = REC ? reinterpret_cast<double2 *>(rec + R.dz + (size_t)blockIdx.x * N * 2 * kWave) : nullptr;
#endif
  double *recCK = REC ? rec + R.ck + (size_t)blockIdx.x * R.nck * (NS + J) * kWave : nullptr;
  int32_t *recCKR = REC ? reinterpret_cast<int32_t *>(rec + R.ckr + (size_t)blockIdx.x * (((size_t)N + 1) / 2)) : nullptr;
#if C2T_TM
  double *recT = REC ? rec + R.t + (size_t)blockIdx.x * kWave : nullptr;  // the grid, lane-major like (d, z)
#else
  double *recT = REC ? rec + R.t + (size_t)blockIdx.x * N * kWave : nullptr;  // the grid, lane-major like (d, z)
#endif

  // ---- row 0 --------------------------------------------------------------------------------------------------
  double S[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) S[k] = 0.0;
  double F[J], w[J];
  double tprev = t[b * t_bs];
  double d = a[b * N], z = y[b * N];
  if constexpr (TERMS) d += tc.A0;
  double rd = 1.0 / d;
  if constexpr (TERMS) {
    double u0[J], v0[J];
    tc.template rows<FAST>(tprev, u0, v0);
#pragma unroll
    for (int j = 0; j < J; ++j) { F[j] = 0.0; w[j] = v0[j] * rd; }
  } else {
#pragma unroll
    for (int j = 0; j < J; ++j) { F[j] = 0.0; w[j] = j < JS ? V[b * N * JS + j] * rd : 0.0; }
  }
  double prod = d, quad = z * z * rd;
  int eacc = 0;
  int32_t fl = 0;
  int slot = 0, nextra = 0;            // wavefront-uniform: checkpoint slots written, extras among them
  int64_t lastck = 0;                  // row of the last checkpoint (row 0: the recursion starts there)
  // Which rows carry a checkpoint travels with the (d, z) record itself: d is stored as -|d| on those rows (for every lane:
  // the decision is wavefront-uniform) and +|d| on all others, so the reverse sweep reads the flag off a value it loads
  // anyway and keeps no list (one scalar register less in a kernel that has none to spare).
  auto write_ckpt = [&](int64_t row) __attribute__((always_inline)) {   // the state as it stands = state after `row`
    double *ck = recCK + (size_t)slot * (NS + J) * kWave;
#pragma unroll
    for (int k = 0; k < NS; ++k) st1_stream(&ck[k * kWave + lane], S[k]);
#pragma unroll
    for (int j2 = 0; j2 < J; ++j2) st1_stream(&ck[(NS + j2) * kWave + lane], F[j2]);
    if (lane == 0) recCKR[row] = slot;
    ++slot;
    lastck = row;
  };
  double tseg = tprev, gmax = 0.0;  // guard: c_max * (t_n - t of the row behind the last checkpoint), the span the
                                    // backward recursion has to invert decays over
  if (REC) {
#pragma unroll
    for (int q = 0; q < J / 2; ++q) recW[q * kWave + lane] = make_double2(w[2 * q], w[2 * q + 1]);
    recDZ[lane] = make_double2(fabs(d), z);   // (row 0 never carries a checkpoint; a series of ONE row is all seeds)
    recT[lane] = tprev;
  }

  // ---- prologue: tiles start at row 0 so that every run is aligned (128 B row tiles, 64 B scalar tiles); row 0 itself
  // was consumed above and is skipped in the loop
  double su[2 * NI], sv[2 * NI];
  double st_[16], sa_[16], sy_[16];   // 16-row tiles of t, a, y (sc_fetch16), staged into the 8-row LDS tiles half by half
  if constexpr (!TERMS) { row_fetch(Ub, N, 0, io, su); row_fetch(Vb, N, 0, io, sv); }
  sc_fetch16(tb, tN, N, 0, lane, last, st_); sc_fetch16(ab, N, N, 0, lane, last, sa_); sc_fetch16(yb, N, N, 0, lane, last, sy_);

  for (int64_t n0 = 0; n0 < N; n0 += ST) {
    // scalar tile of rows n0 .. n0+ST-1: half (n0 / 8) & 1 of the 16-row tile in registers; behind its second half the
    // next 16 rows are requested (a full 8-row turn ahead of their first use)
    const int half = (int)((n0 >> 3) & 1);
    lds_order();
    sc_stage16(tT, lane, half, st_); sc_stage16(tA, lane, half, sa_); sc_stage16(tY, lane, half, sy_);
    if (half == 1) {
      sc_fetch16(tb, tN, N, n0 + ST, lane, last, st_); sc_fetch16(ab, N, N, n0 + ST, lane, last, sa_);
      sc_fetch16(yb, N, N, n0 + ST, lane, last, sy_);
    }
#pragma unroll
    for (int rt = 0; rt < ST / RT; ++rt) {
      const int64_t nt = n0 + rt * RT;
      if (nt < N) {
        if constexpr (!TERMS) {
          lds_order();
          row_stage(tU, lane, su); row_stage(tV, lane, sv);
          row_fetch(Ub, N, nt + RT, io, su); row_fetch(Vb, N, nt + RT, io, sv);
          lds_order();
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const int64_t n = nt + r;
          if (n < N && n > 0) {
            const int rs = rt * RT + r;
            const double tn = tT[lane * SSTR + rs], yn = tY[lane * SSTR + rs];
            double an = tA[lane * SSTR + rs];
            double u[J], v[J], p[J];
            const double dt = tprev - tn;
            if (REC) {
              // the backward recursion reaches row n-1 from the next checkpoint above by inverting the decays of rows
              // n+.. down to n+1 ... and of row n if the state of row n-1 is not on record: re-anchor there if some series
              // of the wavefront could not afford that (wavefront-uniform decision)
              if (lastck == n - 1) tseg = tn;   // the decay into the row behind a checkpoint is never inverted
              if (__any(cmax * (tn - tseg) > kGuard) && nextra < (int)(R.nck - R.nreg)) {
                write_ckpt(n - 1);
                st2_stream(&recDZ[(size_t)(n - 1) * rs1 + lane], make_double2(-fabs(d), z));   // (d, z still of row n-1)
                ++nextra;
                tseg = tn;
              }
              gmax = fmax(gmax, cmax * (tn - tseg));
            }
            tprev = tn;
            if constexpr (TERMS) {
              an += tc.A0;
              tc.template rows<FAST>(tn, u, v);
              tc.decay(cj, dt, p);
            } else {
              row_read(tU, lane, r, u); row_read(tV, lane, r, v);
              decay<PAIRED>(cj, dt, p);
            }
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
              for (int q = 0; q < J / 2; ++q) st2_stream(&recW[(size_t)n * rsW + q * kWave + lane], make_double2(w[2 * q], w[2 * q + 1]));
              const bool seg_end = (n % C == 0) || (n == N - 1);
              st2_stream(&recDZ[(size_t)n * rs1 + lane], make_double2(seg_end ? -fabs(d) : fabs(d), z));
              st1_stream(&recT[(size_t)n * rs1 + lane], tn);
              if (seg_end) write_ckpt(n);  // uniform over the wavefront
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
    if (REC && lane == 0) recCKR[0] = nextra;   // (row 0 never carries a checkpoint: its entry says whether the wavefront has extras)
  }
  if (REC) {
    // The stability measure of THIS wavefront -- max over its series and segments of c_max * span -- goes to its own word
    // (guard[kGateHeadWords + wavefront]): beyond kGuard its reverse sweep returns at once and the replay kernels take its
    // 64 series.  guard[0] keeps the largest of the launch, guard[1] counts the wavefronts that fell back (diagnostics).
    // NaN-aware: a NaN span must disable the fast path (the replay kernels propagate it like the reference)
    double g = (gmax == gmax) ? gmax : INFINITY;
#pragma unroll
    for (int sft = 1; sft < kWave; sft <<= 1) g = fmax(g, __shfl_xor(g, sft, kWave));
    if (lane == 0) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(g);   // g >= 0: the bit pattern is monotone
      guard[kGateHeadWords + blockIdx.x] = bits;
      atomicMax(guard, bits);
      if (g > kGuard) atomicAdd(guard + 1, 1ull);
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
  for (int k = 0; k < JS / 2; ++k) paired = paired && (c[bb * c_bs + 2 * k] == c[bb * c_bs + 2 * k + 1]);
  if (__all(paired))
    fwd_body<REC, true>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, rec, R, guard, lds);
  else
    fwd_body<REC, false>(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, rec, R, guard, lds);
}

// wavefront-uniform: every phase dc_k x_n of the 64 series inside the range of the branch-free sincos
template <int JC>
__device__ __forceinline__ bool terms_phases_fast(int64_t B, int64_t N, const double *__restrict__ x, int64_t x_bs,
                                                  const TermsArgs &T) {
  const int64_t b0 = (int64_t)blockIdx.x * kWave;
  const int64_t b = (b0 + threadIdx.x) < B ? (b0 + threadIdx.x) : (B - 1);
  TermCoef<JC> tc;
  double cj[J];
  tc.load(T, b, cj);
  return __all(tc.phases_fast(x[b * x_bs], x[b * x_bs + N - 1]));
}

// Coefficient-level forward: same body, rows generated in the lane (TermsArgs).  `diag` is the white-noise diagonal.
template <int JC, bool REC>
__global__ __launch_bounds__(kWave, 1) void k_loglik_tt_fwd(int64_t B, int64_t N, const double *__restrict__ t,
                                                            int64_t t_bs, TermsArgs T, const double *__restrict__ diag,
                                                            const double *__restrict__ y, double *__restrict__ ll,
                                                            int32_t *__restrict__ flag, double *__restrict__ rec, Rec R,
                                                            unsigned long long *__restrict__ guard) {
  __shared__ __attribute__((aligned(16))) double lds[kFwdLds / 8];
  if (terms_phases_fast<JC>(B, N, t, t_bs, T))
    fwd_body<REC, true, JC, true>(B, N, t, t_bs, nullptr, 0, diag, nullptr, nullptr, y, ll, flag, rec, R, guard, lds, T);
  else
    fwd_body<REC, true, JC, false>(B, N, t, t_bs, nullptr, 0, diag, nullptr, nullptr, y, ll, flag, rec, R, guard, lds, T);
}

// =============================================================================================================
// Reverse sweep: fused solve_lower_rev (internal.hpp:225-245) + factor_rev (reverse.hpp:52-84) per row, the
// step derived in c2_loglik.hip, with the forward state S_n / F_n obtained by running the recursion backward
// from the checkpoint at the end of the current segment.  Entering step n: bz, ba, bV = complete cotangents of row n;
// S, F = post-decay state of row n; M = bS + bS^T (symmetric-packed), bF, carry = f_{n+1}.
//
// Memory choreography.  Loads and stores share ONE in-order counter on gfx9 (vmcnt): a wait for a prefetched row also
// waits for every store issued before that row was requested, and behind run-time tile tests the compiler has to assume
// the shortest path and ends up draining the stores it has just issued, once per step.  So the sweep is written as TWO
// step instances, each a fixed instruction sequence: the odd step n = 2p+1 puts the aligned pair of U rows (2p, 2p+1)
// -- one 128-byte line per series, requested two steps earlier -- into a two-row tile, requests the pair below it and,
// at its end, sends the pair of bV rows (2p, 2p+1) out; the even step touches no U in memory and sends the pair of bU
// rows out (bU_n takes the place of U_n in the tile).  W, (d, z) and t of the row below come from the lane-major records
// one step ahead.  Only the scalar tiles (8 rows of ba, by, bt) and the checkpoint turn every 8th / 32nd step, at the
// END of a step.  What does NOT work on a register file this full (512 per lane, all in use): anything staged in
// registers across more than a couple of steps (it lands in scratch, where every load becomes load / wait / spill),
// and loads that pass through `asm volatile` moves one by one (each is a scheduling barrier: one memory latency
// apiece) -- hence t in the records and aload() for the checkpoints.  tools/prof_sections.py shows where a step's
// cycles go (build with -DC2T_PROF).
// =============================================================================================================
constexpr int RS1 = J + 2;   // LDS stride (doubles) of a series in a one-row tile: 80 B, conflict-free b128
constexpr int CS4 = 6;        // LDS stride (doubles) of a series' J / 2 distinct rates (paired): 48 B, conflict-free b128
// U/bU (two rows), bV (two rows with paired rates, else one), ba, by, bt, c, bc -- the larger of the two layouts
constexpr int kRevLds = (2 * kWave * RSTR + 3 * kWave * SSTR + kWave * CS4 + kWave * RS1) * 8;
static_assert(kRevLds <= 40960 && (kWave * RSTR + kWave * RS1 + 3 * kWave * SSTR + 2 * kWave * RS1) * 8 <= kRevLds, "four wavefronts per CU");

// One-row tiles: an instruction moves 64 / PPR series x one row (lane l: series (64 / PPR) i + l / PPR, 16-byte piece l % PPR).  Lanes of
// a partial wavefront are clamped onto the last valid series: they move the same bytes to the same place again.
__device__ __forceinline__ void row1_read(const double *tile, int lane, double (&x)[J]) {
#pragma unroll
  for (int q = 0; q < J / 2; ++q) {
    const double2 v = *reinterpret_cast<const double2 *>(tile + lane * RS1 + 2 * q);
    x[2 * q] = v.x; x[2 * q + 1] = v.y;
  }
}
__device__ __forceinline__ void row1_write(double *tile, int lane, const double (&x)[J]) {
#pragma unroll
  for (int q = 0; q < J / 2; ++q)
    *reinterpret_cast<double2 *>(tile + lane * RS1 + 2 * q) = make_double2(x[2 * q], x[2 * q + 1]);
}
__device__ __forceinline__ void row1_flush(double *__restrict__ base, int64_t N, int64_t n, const double *tile, int lane,
                                           int last) {
#pragma unroll
  for (int i = 0; i < PPR; ++i) {   // an instruction moves 64 / PPR series x one row
    int sr = (kWave / PPR) * i + lane / PPR; sr = sr < last ? sr : last;
    if (piece_in(lane % PPR))
      *reinterpret_cast<double2 *>(base + ((int64_t)sr * N + n) * JS + 2 * (lane % PPR)) =
          *reinterpret_cast<const double2 *>(tile + sr * RS1 + 2 * (lane % PPR));
  }
}

// Coefficient-level form (JC >= 0, see TermsArgs): U_n is generated in the lane; bU_n and bV_n never leave it -- they are
// contracted on the spot with the reverse of the matrix recipe (driver.cpp:456-474 transposed):
//     bar_r += ba_n + bU_n[r]                       bac_k += ba_n + bU0 cos + bU1 sin          bbc_k += bU0 sin - bU1 cos
//     g_k = -bU0 U1 + bU1 U0 - bV0 sin + bV1 cos    bdc_k += g_k x_n                           bx_n = bt_n + sum_k g_k dc_k
// (bU0 = bU_n[Jr + 2k], bU1 = bU_n[Jr + 2k + 1], same for bV, U), bcr = bc[:Jr], bcc_k = bc[Jr + 2k] + bc[Jr + 2k + 1],
// bdiag = ba.  In that form `bt` receives bx, `ba` receives bdiag, bU / bV / U / c are not touched and the per-series sums
// go to TermsGrads; the running sums live in an LDS tile next to the accumulators of bc.
struct TermsGrads {
  double *bar, *bcr, *bac, *bbc, *bcc, *bdc;
};
constexpr int AS1 = 14;   // LDS stride (doubles) of a series in the accumulator tile: 112 B, conflict-free b128

// step(m, phase K) for the rows m = ntop - (RT-1-K), K = RT-1 .. 0, of one tile that exist and are not row 0
template <int K, class Step>
__device__ __forceinline__ void for_phases(Step &step, int64_t ntop, int64_t nf) {
  if constexpr (K >= 0) {
    const int64_t m = ntop - (RT - 1 - K);
    if (m <= nf && m >= 1) step(m, std::integral_constant<int, K>{});
    for_phases<K - 1>(step, ntop, nf);
  }
}

// XCK: the wavefront recorded EXTRA checkpoints (gaps in time): the rows that carry a checkpoint are then read off the sign
// of their d record, every step, and the slot of a row comes from CKR.  Without extras the checkpoints sit at the regular
// rows and slots, known at compile time -- that instance is the round-2 sweep unchanged (a test per step more, anywhere in
// it, costs this kernel 5-10 scratch operations per step pair: 512 of 512 registers are in use).
template <bool PAIRED, int JC = -1, bool FAST = true, bool FULL = false, bool XCK = false>
__device__ __forceinline__ void rev_body(int64_t B, int64_t N, const double *__restrict__ t, int64_t t_bs,
                                         const double *__restrict__ c, int64_t c_bs, const double *__restrict__ U,
                                         const int32_t *__restrict__ flag, const double *__restrict__ rec, Rec R,
                                         double *__restrict__ bt, double *__restrict__ bc, double *__restrict__ ba,
                                         double *__restrict__ bU, double *__restrict__ bV, double *__restrict__ by,
                                         double *lds, const TermsArgs T = TermsArgs{}, const TermsGrads G = TermsGrads{}) {
  constexpr bool TERMS = JC >= 0;
  constexpr int JR = TermCoef<JC>::JR, JCN = JC > 0 ? JC : 1;
  constexpr int NACC = TERMS ? JR + 3 * JC + 1 : 1;   // [sum bU real] [bac, bbc, bdc per complex term] [sum ba]
  static_assert(NACC <= AS1, "accumulator tile");
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kWave;
  const int last = (int)((B - 1 - b0) < (kWave - 1) ? (B - 1 - b0) : (kWave - 1));
  const int sl = lane < last ? lane : last;
  const int64_t b = b0 + sl;
  const RowIOT<FULL> io(lane, last);
  // tU: TWO rows of U per series (the aligned pair 2p, 2p+1 = one 128-byte line, fetched once; single rows cost the
  // line twice, +15 GB per sweep at the bench shape); bU_n takes the place of U_n in it
  // paired rates (matrix-level form): four distinct rates per series, and the room that saves holds a two-row bV tile
  constexpr bool BVPAIR = PAIRED && !TERMS;
  constexpr int CSTR = BVPAIR ? CS4 : RS1;
  double *tU = lds, *tBV = tU + kWave * RSTR, *tBA = tBV + kWave * (BVPAIR ? RSTR : RS1), *tBY = tBA + kWave * SSTR,
         *tBT = tBY + kWave * SSTR, *tC = tBT + kWave * SSTR, *tBC = tC + kWave * CSTR;
  static_assert((kWave * RSTR + kWave * (BVPAIR ? RSTR : RS1) + 3 * kWave * SSTR + kWave * CSTR + kWave * RS1) * 8 <= kRevLds, "LDS");
  double *tACC = lds;   // coefficient-level form: takes the place of the U tile (64 x 14 <= 64 x 18 doubles)
  const double *Ub = U + b0 * N * JS;
  double *bUb = bU + b0 * N * JS, *bVb = bV + b0 * N * JS, *bab = ba + b0 * N, *byb = by + b0 * N, *btb = bt + b0 * N;
#if C2T_TM
  const size_t rsW = (size_t)gridDim.x * (J / 2) * kWave, rs1 = (size_t)gridDim.x * kWave;   // row strides (elements)
  const double2 *recW = reinterpret_cast<const double2 *>(rec + R.w) + (size_t)blockIdx.x * (J / 2) * kWave;
  const double2 *recDZ = reinterpret_cast<const double2 *>(rec + R.dz) + (size_t)blockIdx.x * kWave;
#else
  constexpr size_t rsW = (size_t)(J / 2) * kWave, rs1 = kWave;
  const double2 *recW = reinterpret_cast<const double2 *>(rec + R.w + (size_t)blockIdx.x * N * J * kWave);
  const double2 *recDZ = reinterpret_cast<const double2 *>(rec + R.dz + (size_t)blockIdx.x * N * 2 * kWave);
#endif
  const double *recCK = rec + R.ck + (size_t)blockIdx.x * R.nck * (NS + J) * kWave;
  const int32_t *recCKR = reinterpret_cast<const int32_t *>(rec + R.ckr + (size_t)blockIdx.x * (((size_t)N + 1) / 2));
#if C2T_TM
  const double *recT = rec + R.t + (size_t)blockIdx.x * kWave;
#else
  const double *recT = rec + R.t + (size_t)blockIdx.x * N * kWave;
#endif
  const bool failed = flag[b] != 0;  // NaN gradients for a failed factorisation (see k_loglik_rev)
  const double nan = __builtin_nan("");

  // the rates c_j (read twice per step) and the accumulators of bc (read-modify-write once per step) live in LDS rather
  // than in 32 registers
  TermCoef<JC> tc;
  {
    double cj[J], zero[J];
    if constexpr (TERMS) tc.load(T, b, cj);
    else {
#pragma unroll
      for (int j = 0; j < J; ++j) cj[j] = j < JS ? c[b * c_bs + j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) zero[j] = 0.0;
    if constexpr (BVPAIR) {   // c[2k] == c[2k+1]: the J / 2 distinct rates
#pragma unroll
      for (int k = 0; k < PPR; k += 2)
        *reinterpret_cast<double2 *>(tC + lane * CS4 + k) = make_double2(cj[2 * k], cj[2 * k + 2 < J ? 2 * k + 2 : 2 * k]);
    } else {
      row1_write(tC, lane, cj);
    }
    row1_write(tBC, lane, zero);
    if constexpr (TERMS) {
#pragma unroll
      for (int q = 0; q < AS1 / 2; ++q) *reinterpret_cast<double2 *>(tACC + lane * AS1 + 2 * q) = make_double2(0.0, 0.0);
    }
  }
  auto read_rates = [&](double (&cj)[J]) __attribute__((always_inline)) {
    if constexpr (BVPAIR) {
#pragma unroll
      for (int k = 0; k < PPR; k += 2) {
        const double2 c2k = *reinterpret_cast<const double2 *>(tC + lane * CS4 + k);
        cj[2 * k] = cj[2 * k + 1] = c2k.x;
        if (2 * k + 2 < J) cj[2 * k + 2] = cj[2 * k + 3] = c2k.y;
      }
    } else {
      row1_read(tC, lane, cj);
    }
  };
  // one step's contribution to the running sums (TERMS): gv = -bV0 sin + bV1 cos of the row, bu = bU of the row
  auto accumulate = [&](const double (&u)[J], const double (&bu)[J], const double (&sn)[JCN], const double (&cs)[JCN],
                        const double (&gv)[JCN], double ba_n, double xn) -> double {
    double acc[AS1];
#pragma unroll
    for (int q = 0; q < AS1 / 2; ++q) {
      const double2 v = *reinterpret_cast<const double2 *>(tACC + lane * AS1 + 2 * q);
      acc[2 * q] = v.x; acc[2 * q + 1] = v.y;
    }
    double gsum = 0.0;
#pragma unroll
    for (int r = 0; r < JR; ++r) acc[r] += bu[r];
#pragma unroll
    for (int k = 0; k < JC; ++k) {
      const double b0_ = bu[JR + 2 * k], b1_ = bu[JR + 2 * k + 1];
      acc[JR + 3 * k] = fma(b0_, cs[k], fma(b1_, sn[k], acc[JR + 3 * k]));
      acc[JR + 3 * k + 1] = fma(b0_, sn[k], fma(-b1_, cs[k], acc[JR + 3 * k + 1]));
      const double g = fma(-b0_, u[JR + 2 * k + 1], fma(b1_, u[JR + 2 * k], gv[k]));
      acc[JR + 3 * k + 2] = fma(g, xn, acc[JR + 3 * k + 2]);
      gsum = fma(g, tc.dc[k], gsum);
    }
    acc[NACC - 1] += ba_n;
#pragma unroll
    for (int q = 0; q < AS1 / 2; ++q)
      *reinterpret_cast<double2 *>(tACC + lane * AS1 + 2 * q) = make_double2(acc[2 * q], acc[2 * q + 1]);
    return gsum;
  };

  double F[J], bF[J], bVn[J];
  // Both J x J states -- the forward state S of the current row and the adjoint M = bS + bS^T -- live in AGPRs (144 of
  // the 256 a lane owns); each is read and written once per step.  The arithmetic registers hold the width-J vectors
  // only; the extra accvgpr moves are covered by the HBM time of the step.
  int Slo[NS], Shi[NS], Mlo[NS], Mhi[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) { apark(0.0, Slo[k], Shi[k]); apark(0.0, Mlo[k], Mhi[k]); }
  // Coefficient-level form (C2T_MLDS): its LDS has room the matrix-level form spends on the row tiles -- the rest of the
  // U tile behind the accumulators, the unused bV tile, the tail of the block: nine lane-major slots of 16 bytes -- and the
  // first NML = 18 elements of M (in the order of the pass) can live there instead of in accumulation registers: one
  // ds_read_b128 + one ds_write_b128 per two elements against eight v_accvgpr moves.  Measured on the round-4 kernels: 3 - 5 %
  // slower (the LDS round trip sits in the pass's dependency chains, four wavefronts share the CU's LDS); off.
  constexpr int NML = (TERMS && C2T_MLDS && NS >= 18) ? 18 : 0;
  auto mslot = [&](int q) -> double2 * {   // elements 2 q, 2 q + 1
    return reinterpret_cast<double2 *>(lds + (q < 7 ? kWave * AS1 + q * 2 * kWave : kRevLds / 8 - (9 - q) * 2 * kWave)) + lane;
  };
  static_assert(NML == 0 || kWave * AS1 + 7 * 2 * kWave <= kWave * RSTR + kWave * RS1, "free LDS behind the accumulator tile");
  if constexpr (NML > 0) {
#pragma unroll
    for (int q = 0; q < NML / 2; ++q) *mslot(q) = make_double2(0.0, 0.0);
  }
  double2 mpair = make_double2(0.0, 0.0);
#pragma unroll
  for (int j = 0; j < J; ++j) { F[j] = 0.0; bF[j] = 0.0; bVn[j] = failed ? nan : 0.0; }
  double carry = 0.0;

  // seeds of the last row (reverse.hpp:55-57 with bd = d ll / d d, bz = d ll / d z)
  const double2 dzl = recDZ[(size_t)(N - 1) * rs1 + lane];
  const double rdl = 1.0 / fabs(dzl.x);   // (the sign of a recorded d is the checkpoint flag of its row)
  double ban = 0.5 * rdl * (dzl.y * dzl.y * rdl - 1.0), bzn = -dzl.y * rdl;
  // A failed series gets NaN seeds: every gradient of the series is an arithmetic function of them (bF <- u bz,
  // M <- x = bV + 2 ba u, bp <- F bF + ..., bt <- bp, bc <- bp), so NaN reaches all six outputs by propagation.
  if (failed) { ban = nan; bzn = nan; }
  double tcur = recT[(size_t)(N - 1) * rs1 + lane];

  // The recorded S / F of a checkpointed row replace the recursed ones.  The 36 elements of S go straight into their
  // AGPRs (aload: no arithmetic register in flight), all 80 loads are issued back to back and waited for ONCE.  (Through
  // arithmetic registers, each v_accvgpr_write being a scheduling barrier, the same loads were waited for in ~13 separate
  // groups, 2-3 us each under load: ~90k cycles per checkpoint, a quarter of the sweep.)
  // (WHICH rows carry a checkpoint is the sign of their d record; its slot comes from the row's entry of CKR -- a scalar load
  // at the few rows that need it instead of a counter carried through every step)
  auto load_ckpt = [&](int64_t row) {
    const int64_t slot = XCK ? (int64_t)__builtin_amdgcn_readfirstlane(recCKR[row])
                             : ((row % C == 0) ? row / C - 1 : R.nreg - 1);   // regular rows C, 2C, ..., N-1 in order
    const double *ck = recCK + (size_t)slot * (NS + J) * kWave;
    const unsigned voff = (unsigned)lane * 8u;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const double *base = ck + (k / 8) * 8 * kWave;   // immediate offsets reach 4 KB: one scalar base per 8 elements
      if ((k & 7) == 0) { aload<0>(Slo[k], base, voff); aload<4>(Shi[k], base, voff); }
      if ((k & 7) == 1) { aload<512>(Slo[k], base, voff); aload<516>(Shi[k], base, voff); }
      if ((k & 7) == 2) { aload<1024>(Slo[k], base, voff); aload<1028>(Shi[k], base, voff); }
      if ((k & 7) == 3) { aload<1536>(Slo[k], base, voff); aload<1540>(Shi[k], base, voff); }
      if ((k & 7) == 4) { aload<2048>(Slo[k], base, voff); aload<2052>(Shi[k], base, voff); }
      if ((k & 7) == 5) { aload<2560>(Slo[k], base, voff); aload<2564>(Shi[k], base, voff); }
      if ((k & 7) == 6) { aload<3072>(Slo[k], base, voff); aload<3076>(Shi[k], base, voff); }
      if ((k & 7) == 7) { aload<3584>(Slo[k], base, voff); aload<3588>(Shi[k], base, voff); }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) F[j] = ld1_stream(&ck[(NS + j) * kWave + lane]);
    await_aloads();
  };
  // the grid comes back from the records of the forward pass, lane-major like (d, z): one coalesced 8-byte load per
  // step, requested a step ahead.  (A register-staged tile of t requested eight steps ahead ends up in scratch -- the
  // register file is full -- and scratch turns every one of its loads into load / wait / spill: ~3000 cycles per step.)
  auto t_fetch = [&](int64_t row) { return ld1_stream(&recT[(size_t)(row < 0 ? 0 : row) * rs1 + lane]); };
  auto w_fetch = [&](int64_t row, double (&wv)[J]) {
    row = row < 0 ? 0 : row;
#pragma unroll
    for (int q = 0; q < J / 2; ++q) {
      const double2 v = ld2_stream(&recW[(size_t)row * rsW + q * kWave + lane]);
      wv[2 * q] = v.x; wv[2 * q + 1] = v.y;
    }
  };
  auto dz_fetch = [&](int64_t row) { return ld2_stream(&recDZ[(size_t)(row < 0 ? 0 : row) * rs1 + lane]); };

  if (N >= 2) {
    const int64_t nf = N - 1;
    // ---- prologue ------------------------------------------------------------------------------------------------------
    // bV, ba, by of the last row are pure seeds: they leave at once (their slots in the tiles belong to lower rows)
    // bV of the last row is a pure seed.  One-row bV tile, or the last row an even one (its pair partner lies beyond the
    // series): it leaves at once; otherwise it waits in row 1 of the pair tile for the first (odd) step
    const int ph0 = (int)(nf % RT);          // row of the last row inside its tile
    const bool bv_seed_waits = BVPAIR && ph0 != 0;
    if constexpr (!TERMS) {
      if (bv_seed_waits) row_write(tBV, lane, ph0, bVn);
      else row1_write(tBV, lane, bVn);
    }
    tBA[lane * SSTR + (nf & (ST - 1))] = ban;
    tBY[lane * SSTR + (nf & (ST - 1))] = bzn;
    lds_order();
    if constexpr (!TERMS) {
      if (!bv_seed_waits) row1_flush(bVb, N, nf, tBV, lane, last);
    }
    if ((nf & (ST - 1)) == 0) {  // the last row sits alone at the bottom of its scalar tile
      sc_flush(bab, N, nf, nf, nf, tBA, lane, last);
      sc_flush(byb, N, nf, nf, nf, tBY, lane, last);
    }
    double su[2 * NI];                    // the tile of U rows (RT p .. RT p + RT - 1) staged by the next step n = RT p + RT - 1
    double wa[J];                         // W_{n-1} (requested one step ahead: first used well into the step)
    double2 dza;                          // (d, z)_{n-1}
    double ta;                            // t_{n-1}
    if constexpr (!TERMS) {
      row_fetch(Ub, N, nf - ph0, io, su);
      if (ph0 != RT - 1) {  // the first step is not a staging one: its tile goes into LDS here
        lds_order();
        row_stage(tU, lane, su);
        lds_order();
        row_fetch(Ub, N, nf - ph0 - RT, io, su);
      }
    }
    w_fetch(nf - 1, wa);
    dza = dz_fetch(nf - 1);
    ta = t_fetch(nf - 1);
    load_ckpt(nf);   // the last row always carries one
    lds_order();

#ifdef C2T_PROF
    unsigned long long prof_[6] = {0, 0, 0, 0, 0, 0}, tick_ = __builtin_readcyclecounter();
#endif
    // One step of the sweep, one instance per row PH = n % RT of a tile (matrix-level form; at J = 8: odd / even).  The
    // instance of the tile's top row puts the tile of U rows into LDS and requests the tile below it; the others touch no
    // U in memory.  Each instance is a fixed instruction sequence (see "Memory choreography" above).
    auto step = [&](const int64_t n, auto phase_tag) __attribute__((always_inline)) {
      constexpr int PH = decltype(phase_tag)::value;   // n % RT (matrix-level form)
      constexpr bool STAGE = PH == RT - 1;             // the step that puts its tile of U rows into LDS
      constexpr bool BUOUT = PH == 0;                  // ... whose end sends the tile of bU rows out
      constexpr bool BVOUT = PH == (1 % RT);           // ... whose end sends the tile of bV rows out (row 0 = bV_{n-1})
      __builtin_amdgcn_sched_barrier(0);   // the two instances are scheduled (and their registers allocated) apart
      C2T_TICK(4);
      // ---- fixed part: U_n into its tile, requests for two steps ahead ------------------------------------------------
      double wb[J];
#if C2T_RECFIRST
      // the records of the row below are wanted one step from now, the tile of U rows two: request them in that order (the
      // counter of outstanding memory operations retires in order -- behind the 64 scattered lines of a U tile the records
      // would wait for the slowest of them)
      w_fetch(n - 2, wb);
      const double2 dzb = dz_fetch(n - 2);
      const double tb2 = t_fetch(n - 2);
      __builtin_amdgcn_sched_barrier(0);
#endif
      if constexpr (!TERMS && STAGE) {
        lds_order();
        row_stage(tU, lane, su);
        row_fetch(Ub, N, n - 2 * RT + 1, io, su);
      }
#if !C2T_RECFIRST
      w_fetch(n - 2, wb);
      const double2 dzb = dz_fetch(n - 2);
      const double tb2 = t_fetch(n - 2);
#endif
      if constexpr (!TERMS && STAGE) lds_order();

      // ---- the step ---------------------------------------------------------------------------------------------
      const int rs = (int)((n - 1) & (ST - 1));
      double u[J], p[J], ip[J];
      double sn[JCN], cs[JCN], gv[JCN];
      const double xn = tcur;
      if constexpr (TERMS) {
        tc.template rows_sc<FAST>(xn, u, sn, cs);
#pragma unroll
        for (int k = 0; k < JC; ++k) gv[k] = fma(bVn[JR + 2 * k + 1], cs[k], -(bVn[JR + 2 * k] * sn[k]));
      } else {
        row_read(tU, lane, PH, u);
      }
      const double ba_in = ban;
      const double tm = ta;
      const double dt = tm - tcur;
      tcur = tm;
      {
        double cj[J];
        read_rates(cj);
        if constexpr (TERMS) tc.decay(cj, dt, p);
        else decay<PAIRED>(cj, dt, p);
      }
      C2T_TICK(0);
      // solve_lower_rev part (internal.hpp:232-245)
      double bp[J];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        bF[j] = fma(-u[j], bzn, bF[j]);
        bp[j] = F[j] * bF[j];
        bF[j] *= p[j];
      }
      // factor_rev part (reverse.hpp:65-80) and the backward recursion of the forward state, ONE pass over the 36 packed
      // elements of S and M:
      //   x = bV + 2 ba U;  xs = x S (-> bU2 = -xs);  M -= U^T x + bV^T U;  bp += diag(S M);  M = P M P;  q = W_{n-1} M;
      //   S_{n-1} = P^-1 S_n P^-1 - d_{n-1} w_{n-1}^T w_{n-1}   (where row n-1 is a checkpointed row the result is
      //   replaced at the end of the step)
      const double dm = fabs(dza.x), rdm = rcp_nr(dm), zm = dza.y;
      bool ckm = false;
      if constexpr (XCK) ckm = __builtin_amdgcn_readfirstlane(__double2hiint(dza.x)) < 0;   // row n-1 carries a checkpoint
#if C2T_M2
      // (x_j - 2 ba u_j = bV_j, so M -= u_i x_j + x_i u_j - 2 ba u_i u_j = u_i bV_j + x_i u_j: two fma per element, with
      // bV kept next to x for the length of the pass)
      double x[J];
#else
      // (bV_i = x_i - 2 ba u_i, so M -= u_i x_j + x_i u_j - 2 ba u_i u_j: x takes the registers of bV)
      double (&x)[J] = bVn;
#endif
      double xs[J], q[J];
      const double ba2 = 2.0 * ban;
#pragma unroll
      for (int j = 0; j < J; ++j) { x[j] = fma(ba2, u[j], bVn[j]); xs[j] = 0.0; q[j] = 0.0; ip[j] = rcp_nr(p[j]); }
#pragma unroll
      for (int i = 0; i < J; ++i) {
        const double dwi = dm * wa[i];
#pragma unroll
        for (int j2 = i; j2 < J; ++j2) {
          const int k = sidx(i, j2);
          const double sv = afetch(Slo[k], Shi[k]);
          double m;
          if (k < NML) {
            if ((k & 1) == 0) mpair = *mslot(k / 2);
            m = (k & 1) ? mpair.y : mpair.x;
          } else {
            m = afetch(Mlo[k], Mhi[k]);
          }
          xs[j2] = fma(x[i], sv, xs[j2]);
          if (j2 != i) xs[i] = fma(x[j2], sv, xs[i]);
#if C2T_M2
          m = fma(-u[i], bVn[j2], m);
          m = fma(-x[i], u[j2], m);
#else
          m = fma(-u[i], x[j2], m);
          m = fma(-x[i], u[j2], m);
          m = fma(ba2 * u[i], u[j2], m);
#endif
          bp[j2] = fma(sv, m, bp[j2]);
          if (j2 != i) bp[i] = fma(sv, m, bp[i]);
          m *= p[i] * p[j2];
          if (k < NML) {
            if (k & 1) { mpair.y = m; *mslot(k / 2) = mpair; } else mpair.x = m;
          } else {
            apark(m, Mlo[k], Mhi[k]);
          }
          q[j2] = fma(wa[i], m, q[j2]);
          if (j2 != i) q[i] = fma(wa[j2], m, q[i]);
          apark(fma(-dwi, wa[j2], sv * (ip[i] * ip[j2])), Slo[k], Shi[k]);
        }
      }
      C2T_TICK(1);
      double gsum = 0.0;
      {
        double o[J];
#pragma unroll
        for (int j = 0; j < J; ++j) o[j] = fma(-bzn, F[j], -xs[j]);  // bU_n = -bz_n F_n - x S_n
        if constexpr (TERMS) gsum = accumulate(u, o, sn, cs, gv, ba_in, xn);
        else row_write(tU, lane, PH, o);  // bU_n takes the place of U_n in the tile
      }
      double f = 0.0;
      {
        double cj[J], bcj[J];
        read_rates(cj);
        row1_read(tBC, lane, bcj);
#pragma unroll
        for (int j = 0; j < J; ++j) { bcj[j] = fma(dt, bp[j], bcj[j]); f = fma(cj[j], bp[j], f); }
        row1_write(tBC, lane, bcj);
      }
      const double btn = carry - f + gsum;   // coefficient-level form: bx_n
      carry = f;
#pragma unroll
      for (int j = 0; j < J; ++j) F[j] = fma(-wa[j], zm, F[j] * ip[j]);   // F_{n-1} = P^-1 F_n - w_{n-1} z_{n-1}
      double Gs = 0.0, Q = 0.0;
#pragma unroll
      for (int j = 0; j < J; ++j) { Gs = fma(wa[j], bF[j], Gs); Q = fma(q[j], wa[j], Q); }
      const double zr = zm * rdm;
      bzn = Gs - zr;
#pragma unroll
      for (int j = 0; j < J; ++j) bVn[j] = fma(zr, bF[j], q[j]);
      ban = 0.5 * rdm * (zm * zr - 1.0) - 0.5 * Q - zr * Gs;
      // outputs of rows n-1 (ba, by, bV) and n (bt, bU)
      tBA[lane * SSTR + rs] = ban;
      tBY[lane * SSTR + rs] = bzn;
      tBT[lane * SSTR + (int)(n & (ST - 1))] = btn;
      if constexpr (BVPAIR) row_write(tBV, lane, (PH + RT - 1) % RT, bVn);   // bV_{n-1}
      else if constexpr (!TERMS) row1_write(tBV, lane, bVn);
      lds_order();
      // Outputs of width J leave as whole TILES (RT rows = one aligned 128-byte line per series; at J = 8 a pair of
      // rows; single 64-byte rows cost ~4 % of the sweep in HBM efficiency): the tile of bU rows at the end of the step
      // that fills its row 0; with paired rates the tile of bV rows at the end of the step that produces ITS row 0
      // (unpaired rates keep a one-row bV tile -- their LDS budget is spent on eight rates -- and write bV_{n-1} every
      // step).  All LDS reads first, then all stores.
      if constexpr (!TERMS) {
        double fv[16], fu[16];
        if constexpr (!BVPAIR) {
#pragma unroll
          for (int i = 0; i < PPR; ++i) {
            int sr = (kWave / PPR) * i + lane / PPR; sr = (FULL || sr < last) ? sr : last;
            const double2 v1 = *reinterpret_cast<const double2 *>(tBV + sr * RS1 + 2 * (lane % PPR));
            fv[2 * i] = v1.x; fv[2 * i + 1] = v1.y;
          }
        } else if constexpr (BVOUT) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            int sr = 8 * i + lane / 8; sr = (FULL || sr < last) ? sr : last;
            const double2 v1 = *reinterpret_cast<const double2 *>(tBV + sr * RSTR + 2 * (lane & 7));
            fv[2 * i] = v1.x; fv[2 * i + 1] = v1.y;
          }
        }
        if constexpr (BUOUT) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            int sr = 8 * i + lane / 8; sr = (FULL || sr < last) ? sr : last;
            const double2 v0 = *reinterpret_cast<const double2 *>(tU + sr * RSTR + 2 * (lane & 7));
            fu[2 * i] = v0.x; fu[2 * i + 1] = v0.y;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!BVPAIR) {
#pragma unroll
          for (int i = 0; i < PPR; ++i) {
            int sr = (kWave / PPR) * i + lane / PPR; sr = (FULL || sr < last) ? sr : last;
            if (piece_in(lane % PPR))
              st2_stream(reinterpret_cast<double2 *>(bVb + ((int64_t)sr * N + n - 1) * JS + 2 * (lane % PPR)), make_double2(fv[2 * i], fv[2 * i + 1]));
          }
        } else if constexpr (BVOUT) {
          // rows n-1 .. n-1+RT-1; at J = 8 (pairs) row n always exists, wider tiles may reach beyond the series at the top
          if (RT == 2 || n - 1 + (lane & 7) / PPR <= nf) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              int sr = 8 * i + lane / 8; sr = (FULL || sr < last) ? sr : last;
              if (piece_in(lane & 7))
                st2_stream(reinterpret_cast<double2 *>(bVb + ((int64_t)sr * N + n - 1) * JS + piece_off(lane & 7)), make_double2(fv[2 * i], fv[2 * i + 1]));
            }
          }
        }
        if constexpr (BUOUT) {
          if (n + (lane & 7) / PPR <= nf) {   // the upper rows of the top tile may lie beyond the series
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              int sr = 8 * i + lane / 8; sr = (FULL || sr < last) ? sr : last;
              if (piece_in(lane & 7))
                st2_stream(reinterpret_cast<double2 *>(bUb + ((int64_t)sr * N + n) * JS + piece_off(lane & 7)), make_double2(fu[2 * i], fu[2 * i + 1]));
            }
          }
        }
      }

      C2T_TICK(2);
      // ---- every 8th step: scalar tiles turn; every 32nd: the checkpoint replaces the recursed state -----------------
      if ((n & (ST - 1)) == 0) sc_flush(btb, N, n, 0, N - 1, tBT, lane, FULL ? kWave - 1 : last);  // bt rows n .. n+7
      if (rs == 0) {  // row n-1 is the lowest row of its tile
        sc_flush(bab, N, n - 1, 0, N - 1, tBA, lane, FULL ? kWave - 1 : last);
        sc_flush(byb, N, n - 1, 0, N - 1, tBY, lane, FULL ? kWave - 1 : last);
        // the state of row n-1 is on record (a regular checkpoint every C rows): it replaces the recursed one
        if constexpr (!XCK) { if (n >= 2 && (n - 1) % C == 0) load_ckpt(n - 1); }
      }
      // ... or, with extra checkpoints in front of gaps in time, wherever the record says so
      if constexpr (XCK) { if (ckm && n >= 2) load_ckpt(n - 1); }
      C2T_TICK(5);
#pragma unroll
      for (int j = 0; j < J; ++j) wa[j] = wb[j];
      dza = dzb;
      ta = tb2;
      C2T_TICK(3);
    };
    if constexpr (TERMS) {
      for (int64_t n = nf; n >= 1; --n) step(n, std::integral_constant<int, 0>{});
    } else {
      // RT step instances, one per row of a tile, top row first; rows beyond either end of the series are skipped
      if constexpr (RT == 2 && PAIRED) {   // (same thing spelled out: this spelling keeps the paired instance free of
                                           // scratch, the generic one below the unpaired instance -- register allocation
                                           // at 512 / 512 registers is that sensitive)
        for (int64_t n = nf | 1; n >= 1; n -= 2) {
          if (n <= nf) step(n, std::integral_constant<int, 1>{});
          if (n >= 2) step(n - 1, std::integral_constant<int, 0>{});
        }
      } else {
        for (int64_t ntop = nf | (RT - 1); ntop >= RT - 1; ntop -= RT) for_phases<RT - 1>(step, ntop, nf);
      }
    }
#ifdef C2T_PROF
    if (lane == 0 && blockIdx.x % 97 == 0)
      for (int k = 0; k < 6; ++k) atomicAdd(&c2t_prof[k], prof_[k]);
#endif
    // row 0: bU_0 = 0 (reverse.hpp:83), bt_0 = f_1; ba_0 / by_0 / bV_0 left with the last step
    {
      double zero[J];
#pragma unroll
      for (int j = 0; j < J; ++j) zero[j] = failed ? nan : 0.0;
      lds_order();
      if constexpr (TERMS) {   // row 0: bU_0 = 0, bV_0 and ba_0 complete
        double u0[J], sn[JCN], cs[JCN], gv[JCN], bu0[J];
        tc.template rows_sc<FAST>(tcur, u0, sn, cs);
#pragma unroll
        for (int j = 0; j < J; ++j) bu0[j] = 0.0;
#pragma unroll
        for (int k = 0; k < JC; ++k) gv[k] = fma(bVn[JR + 2 * k + 1], cs[k], -(bVn[JR + 2 * k] * sn[k]));
        carry += accumulate(u0, bu0, sn, cs, gv, ban, tcur);
      } else {
        row_write(tU, lane, 0, zero);   // bU_1 is waiting in row 1 of the tile
      }
      tBT[lane * SSTR] = carry;
      lds_order();
      if constexpr (!TERMS) {
        if ((lane & 7) / PPR <= nf) {   // rows 0 .. min(RT - 1, N - 1)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            int sr = 8 * i + lane / 8; sr = sr < last ? sr : last;
            if (piece_in(lane & 7))
              *reinterpret_cast<double2 *>(bUb + ((int64_t)sr * N) * JS + piece_off(lane & 7)) =
                  *reinterpret_cast<const double2 *>(tU + sr * RSTR + 2 * (lane & 7));
          }
        }
      }
      sc_flush(btb, N, 0, 0, (ST - 1) < (N - 1) ? (ST - 1) : (N - 1), tBT, lane, last);
    }
  } else {  // N == 1: seeds only
    if (lane <= last) {
      bab[(int64_t)lane * N] = ban;
      byb[(int64_t)lane * N] = bzn;
      btb[(int64_t)lane * N] = failed ? nan : 0.0;
      if constexpr (TERMS) {
        tACC[lane * AS1 + NACC - 1] = ban;   // the only row: sum ba = ba_0, every other sum is empty
        if (failed) {
          for (int q = 0; q < NACC; ++q) tACC[lane * AS1 + q] = nan;
          for (int j = 0; j < J; ++j) tBC[lane * RS1 + j] = nan;
        }
      } else {
#pragma unroll
        for (int j = 0; j < JS; ++j) { bUb[(int64_t)lane * N * JS + j] = failed ? nan : 0.0; bVb[(int64_t)lane * N * JS + j] = failed ? nan : 0.0; }
      }
    }
  }
  if (lane <= last) {
    double bcj[J];
    row1_read(tBC, lane, bcj);
    if constexpr (TERMS) {
      const double *acc = tACC + lane * AS1;
      const double sba = acc[NACC - 1];
#pragma unroll
      for (int r = 0; r < JR; ++r) { G.bar[b * JR + r] = sba + acc[r]; G.bcr[b * JR + r] = bcj[r]; }
#pragma unroll
      for (int k = 0; k < JC; ++k) {
        G.bac[b * JC + k] = sba + acc[JR + 3 * k];
        G.bbc[b * JC + k] = acc[JR + 3 * k + 1];
        G.bdc[b * JC + k] = acc[JR + 3 * k + 2];
        G.bcc[b * JC + k] = bcj[JR + 2 * k] + bcj[JR + 2 * k + 1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < JS; ++j) bc[b * JS + j] = bcj[j];
    }
  }
}

__global__ __launch_bounds__(kWave, 1) void k_loglik_t_rev(int64_t B, int64_t N, const double *__restrict__ t,
                                                           int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                           const double *__restrict__ U,
                                                           const int32_t *__restrict__ flag,
                                                           const double *__restrict__ rec, Rec R,
                                                           const unsigned long long *__restrict__ guard,
                                                           double *__restrict__ bt, double *__restrict__ bc,
                                                           double *__restrict__ ba, double *__restrict__ bU,
                                                           double *__restrict__ bV, double *__restrict__ by) {
  __shared__ __attribute__((aligned(16))) double lds[kRevLds / 8];
  if (__longlong_as_double((long long)guard[kGateHeadWords + blockIdx.x]) > kGuard) return;  // the replay kernels take these 64 series
  const int lane = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * kWave;
  const int64_t bb = (b0 + lane) < B ? (b0 + lane) : (B - 1);
  bool paired = true;
#pragma unroll
  for (int k = 0; k < JS / 2; ++k) paired = paired && (c[bb * c_bs + 2 * k] == c[bb * c_bs + 2 * k + 1]);
  const bool full = b0 + kWave <= B;
  const bool xck = __builtin_amdgcn_readfirstlane(
                       reinterpret_cast<const int32_t *>(rec + R.ckr + (size_t)blockIdx.x * (((size_t)N + 1) / 2))[0]) > 0;
#define C2T_REV(P_, F_, X_) rev_body<P_, -1, true, F_, X_>(B, N, t, t_bs, c, c_bs, U, flag, rec, R, bt, bc, ba, bU, bV, by, lds)
  if (__all(paired)) {
    if (full) { if (xck) C2T_REV(true, true, true); else C2T_REV(true, true, false); }
    else { if (xck) C2T_REV(true, false, true); else C2T_REV(true, false, false); }
  } else {
    if (full) { if (xck) C2T_REV(false, true, true); else C2T_REV(false, true, false); }
    else { if (xck) C2T_REV(false, false, true); else C2T_REV(false, false, false); }
  }
#undef C2T_REV
}

template <int JC>
__global__ __launch_bounds__(kWave, 1) void k_loglik_tt_rev(int64_t B, int64_t N, const double *__restrict__ x,
                                                            int64_t x_bs, TermsArgs T, const int32_t *__restrict__ flag,
                                                            const double *__restrict__ rec, Rec R,
                                                            const unsigned long long *__restrict__ guard, TermsGrads G,
                                                            double *__restrict__ bx, double *__restrict__ bdiag,
                                                            double *__restrict__ by) {
  __shared__ __attribute__((aligned(16))) double lds[kRevLds / 8];
  if (__longlong_as_double((long long)guard[kGateHeadWords + blockIdx.x]) > kGuard) return;  // the composed chain takes these 64 series
  const bool xck = __builtin_amdgcn_readfirstlane(
                       reinterpret_cast<const int32_t *>(rec + R.ckr + (size_t)blockIdx.x * (((size_t)N + 1) / 2))[0]) > 0;
#define C2T_TREV(FAST_, X_) rev_body<true, JC, FAST_, false, X_>(B, N, x, x_bs, nullptr, 0, nullptr, flag, rec, R, bx, nullptr, bdiag, nullptr, nullptr, by, lds, T, G)
  if (terms_phases_fast<JC>(B, N, x, x_bs, T)) { if (xck) C2T_TREV(true, true); else C2T_TREV(true, false); }
  else { if (xck) C2T_TREV(false, true); else C2T_TREV(false, false); }
#undef C2T_TREV
}

}  // namespace c2t

using namespace c2t;

extern "C" {

// Forward-only log-likelihood, one lane per series (J == 8).  Internal: dispatched by c2_loglik for large batches.
int C2T_NAME(c2_internal_loglik_t)(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                         const double *a, const double *U, const double *V, const double *y, double *ll,
                         int32_t *flag, c2_stream_t stream) {
  const dim3 grid((unsigned)((B + kWave - 1) / kWave));
  Rec R{};
  hipLaunchKernelGGL((k_loglik_t_fwd<false>), grid, dim3(kWave), 0, (hipStream_t)stream, B, N, t, t_bs, c, c_bs, a, U,
                     V, y, ll, flag, (double *)nullptr, R, (unsigned long long *)nullptr);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// Records of the fwd/rev pair, in doubles (the caller overlays them with the replay kernels' workspace: only one of
// the two paths runs).
size_t C2T_NAME(c2_internal_loglik_t_record_doubles)(int64_t B, int64_t N) { return rec_layout(B, N).total; }

// Forward with records + backward-recursion reverse sweep.  `guard` (device, kGateHeadWords + ceil(B / 64) words, the two
// head words zeroed by the caller on the same stream): word kGateHeadWords + w receives wavefront w's max over series and
// segments of c_max * span; its k_loglik_t_rev returns at once when that exceeds kBackwardGuard, and the caller's replay
// kernels -- gated per group of 64 series by the same words -- then produce the gradients of those series instead.
int C2T_NAME(c2_internal_loglik_t_grad)(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                              const double *a, const double *U, const double *V, const double *y, double *ll,
                              double *bt, double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag,
                              double *rec, unsigned long long *guard, c2_stream_t stream) {
  const dim3 grid((unsigned)((B + kWave - 1) / kWave));
  const Rec R = rec_layout(B, N);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((k_loglik_t_fwd<true>), grid, dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, rec,
                     R, guard);
  if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;
  hipLaunchKernelGGL(k_loglik_t_rev, grid, dim3(kWave), 0, s, B, N, t, t_bs, c, c_bs, U, (const int32_t *)flag,
                     (const double *)rec, R, (const unsigned long long *)guard, bt, bc, ba, bU, bV, by);
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// Coefficient-level forward (J = Jr + 2 Jc == 8): log-likelihood straight from (ar, cr, ac, bc, cc, dc, x, diag, y),
// no U / V / c arrays in memory.
int C2T_NAME(c2_internal_loglik_tt)(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                          const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                          int64_t x_bs, const double *diag, const double *y, double *ll, int32_t *flag,
                          c2_stream_t stream) {
  const dim3 grid((unsigned)((B + kWave - 1) / kWave));
  const TermsArgs T{ar, cr, ac, bc, cc, dc, coef_batched};
  Rec R{};
#define C2_TT(JC_)                                                                                                       \
  hipLaunchKernelGGL((k_loglik_tt_fwd<JC_, false>), grid, dim3(kWave), 0, (hipStream_t)stream, B, N, x, x_bs, T, diag, y, \
                     ll, flag, (double *)nullptr, R, (unsigned long long *)nullptr)
  switch (Jc) {
#if C2T_JS == C2T_J   // the padded build serves the matrix-level entry points only
    case 0: C2_TT(0); break;
    case 1: C2_TT(1); break;
#endif
#if C2T_J >= 4 && C2T_JS == C2T_J
    case 2: C2_TT(2); break;
#endif
#if C2T_J >= 8 && C2T_JS == C2T_J
    case 3: C2_TT(3); break;
    case 4: C2_TT(4); break;
#endif
    default: return C2_ERR_UNSUPPORTED;
  }
#undef C2_TT
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

// Coefficient-level forward with records + reverse sweep (see c2_internal_loglik_t_grad for `guard`).
int C2T_NAME(c2_internal_loglik_tt_grad)(int64_t B, int64_t N, int64_t Jc, int coef_batched, const double *ar, const double *cr,
                               const double *ac, const double *bc, const double *cc, const double *dc, const double *x,
                               int64_t x_bs, const double *diag, const double *y, double *ll, double *bar, double *bcr,
                               double *bac, double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                               int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream) {
  const dim3 grid((unsigned)((B + kWave - 1) / kWave));
  const TermsArgs T{ar, cr, ac, bc, cc, dc, coef_batched};
  const TermsGrads G{bar, bcr, bac, bbc, bcc, bdc};
  const Rec R = rec_layout(B, N);
  hipStream_t s = (hipStream_t)stream;
#define C2_TT(JC_)                                                                                                       \
  do {                                                                                                                   \
    hipLaunchKernelGGL((k_loglik_tt_fwd<JC_, true>), grid, dim3(kWave), 0, s, B, N, x, x_bs, T, diag, y, ll, flag, rec, R, \
                       guard);                                                                                           \
    if (hipGetLastError() != hipSuccess) return C2_ERR_HIP;                                                              \
    hipLaunchKernelGGL((k_loglik_tt_rev<JC_>), grid, dim3(kWave), 0, s, B, N, x, x_bs, T, (const int32_t *)flag,           \
                       (const double *)rec, R, (const unsigned long long *)guard, G, bx, bdiag, by);                     \
  } while (0)
  switch (Jc) {
#if C2T_JS == C2T_J   // the padded build serves the matrix-level entry points only
    case 0: C2_TT(0); break;
    case 1: C2_TT(1); break;
#endif
#if C2T_J >= 4 && C2T_JS == C2T_J
    case 2: C2_TT(2); break;
#endif
#if C2T_J >= 8 && C2T_JS == C2T_J
    case 3: C2_TT(3); break;
    case 4: C2_TT(4); break;
#endif
    default: return C2_ERR_UNSUPPORTED;
  }
#undef C2_TT
  return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP;
}

#ifdef C2T_PROF
// cycles per section summed over the sampled wavefronts (every 97th) since the last call; resets the counters
void c2_internal_prof_read(unsigned long long *out8) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(c2t_prof), sizeof(unsigned long long) * 8);
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipMemcpyToSymbol(HIP_SYMBOL(c2t_prof), z, sizeof(z));
}
#endif

}  // extern "C"
