// c2_loglik.hip -- the north-star hot path: fused batched log-likelihood and its
// reverse-mode gradient, hand-written for gfx950.
//
//   c2_loglik      : factor (forward.hpp:105-134) + solve_lower (internal.hpp:135-145) + the two
//                    reductions of the reference's callers (numpy.py:84-87,104-109; core.py:428) in ONE
//                    pass over (t, a, U, V, y); only ll[b] and flag[b] are written.
//   c2_loglik_grad : the same forward pass, additionally dropping a small CHECKPOINT of the recursion
//                    state every C steps, followed by one reverse sweep that -- segment by segment, last
//                    to first -- recomputes the C forward steps from the checkpoint (S_n, tau_n, W_n, F_n; the
//                    C J x J states wait in accumulation registers) and runs solve_lower_rev
//                    (internal.hpp:225-245) fused with factor_rev (reverse.hpp:52-84) over them.  The
//                    reference's S (N,J,J) and F (N,J) workspaces (1312 B per step at J=8) and even its W
//                    (N,J) are never materialised in HBM.
//
// Mapping (c2_common.hpp): G = 2^ceil(log2 J) lanes per series, 64/G series per wavefront, one
// wavefront per workgroup.  Lane j owns column j of S / M and element j of every width-J vector, in XOR
// order (c2_loglik_helpers.hpp).  Width-J vectors that every lane of a group needs are gathered by DPP lane
// permutes when they sit on the recursion's critical path or when the kernel runs one wavefront per SIMD (the
// reverse sweep), and through a per-wave LDS slot, one step ahead, where a second wavefront hides the LDS
// instructions (the forward pass) -- measured costs in profiles/r01_ubench_instruction_costs.md.
// Scalar all-reduces (d_n, z_n, ...) are DPP butterflies (gsum).  HBM latency is hidden by an explicit
// register prefetch ring (R rows ahead): at 8192 series per GPU and J=8 there is one wavefront per SIMD.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "c2_loglik_helpers.hpp"
#include "../../include/celerite2_amd.h"

extern "C" void c2_internal_set_error(const char *msg);
extern "C" int c2_loglik_grad_composite(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                        const double *c, int64_t c_bs, const double *a, const double *U,
                                        const double *V, const double *y, double *ll, double *bt, double *bc,
                                        double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                                        size_t work_bytes, c2_stream_t stream);
extern "C" size_t c2_loglik_grad_composite_workspace_bytes(int64_t B, int64_t N, int64_t J);

#ifdef C2_REV_TIMING
__device__ unsigned long long c2_dbg[8];
extern "C" void c2_internal_read_dbg(unsigned long long *out) {  // read and clear
  hipMemcpyFromSymbol(out, HIP_SYMBOL(c2_dbg), sizeof(unsigned long long) * 8);
  unsigned long long z[8] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(c2_dbg), z, sizeof(z));
}
#endif
#ifndef C2_LOGLIK_PAIRLINES
#define C2_LOGLIK_PAIRLINES 1
#endif

namespace c2 {

// Checkpoint record of one lane: SX[0..G-1] (column j, XOR order), F_j, w_j, d, z  -> G+4 doubles.
template <int G>
struct Ckpt {
  static constexpr int W = G + 4;
};

// The chain part of one forward step (row n) of factor + solve_lower: everything that depends on the
// previous row's W.  pX/uX are the XOR-gathered p_n and U_n (prepared ahead of time, off the chain).
// On entry SX/F/w/d/z describe row n-1, on exit row n.
template <int G>
__device__ __forceinline__ void fwd_chain(double p, double u, double v, double an, double yn, const double (&pX)[G],
                                          const double (&uX)[G], double (&SX)[G], double &F, double &w, double &d,
                                          double &z, double &rd, double *xslot, int lane, double *tau_out = nullptr) {
  double wX[G];
  xgather_dpp<G>(w, xslot, lane, wX);
  const double dw = d * w;
  double tau0 = 0.0, tau1 = 0.0;
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const double s = (pX[k] * p) * fma(dw, wX[k], SX[k]);  // S = P (S + d w^T w) P   (forward.hpp:115-123)
    SX[k] = s;
    if (k & 1) tau1 = fma(uX[k], s, tau1);                 // tau = U_n S            (forward.hpp:126)
    else tau0 = fma(uX[k], s, tau0);
  }
  const double tau = tau0 + tau1;
  if (tau_out) *tau_out = tau;
  F = p * fma(w, z, F);                                    // F = P (F + W_{n-1}^T z_{n-1})  (internal.hpp:140-143)
  double rd_ = tau * u, rz_ = u * F;
  gsum2<G>(rd_, rz_);
  const double dn = an - rd_;                              // forward.hpp:127
  const double zn = yn - rz_;                              // internal.hpp:144
  rd = rcp_nr(dn);
  w = (v - tau) * rd;                                      // forward.hpp:131
  d = dn;
  z = zn;
}

// =============================================================================
// Forward pass.  R = prefetch ring length (rows), C = checkpoint interval (R % C == 0).
// Software pipeline per step n:   (a) p_{n+1} = exp(c dt_{n+1}) and U_{n+1} go to the LDS slots and are
// gathered back in XOR order for the NEXT step;  (b) the chain of step n runs on the vectors gathered
// during step n-1;  (c) ring slot r is refilled with row n+R.
// =============================================================================
#ifndef C2_FWD_RING
#define C2_FWD_RING 8   // rows of U, V in flight per lane (the register ring of the forward kernels)
#endif
#ifndef C2_FWD_OCC
#define C2_FWD_OCC 1
#endif
#ifndef C2_REV_OCC
#define C2_REV_OCC 1
#endif
#ifndef C2_REV_APARK
#define C2_REV_APARK 1
#endif
#ifndef C2_BACK_EARLY
#define C2_BACK_EARLY 0   // the backward sweep's prefetch a whole segment ahead instead of half: measured, 3 % slower at 8192 series
#endif
// Packed symmetric storage of the C saved S_n columns in LDS.  In XOR order slot k of lane j is S(j^k, j) and
// slot k of lane j^k is its transpose twin S(j, j^k) -- the same number up to rounding -- so for k >= 1 only
// the lane whose bit hb(k) (highest set bit of k) is clear stores it and both lanes read that copy:
// G + (G-1) G/2 doubles per series and step instead of G^2 (36 instead of 64 at G = 8).
template <int G>
struct SymPack {
  static constexpr int PER_STEP = kWave + (G - 1) * (kWave / 2);  // doubles per wavefront per step
  // offset (in doubles) of slot k >= 1 for wave-local lane l
  static __device__ __forceinline__ int off(int l, int k) {
    const int b = 31 - __builtin_clz(k), hb = 1 << b;
    const int o = (l & hb) ? (l ^ k) : l;  // owner lane (bit b clear)
    const int idx = ((o >> (b + 1)) << b) | (o & (hb - 1));
    return kWave + (k - 1) * (kWave / 2) + idx;
  }
};

// Checkpoints are private to a wavefront (written by k_loglik_fwd, read back by the same lanes of
// k_loglik_rev), so they are stored wave-blocked, component-major and SYMMETRIC-PACKED exactly like the LDS copy
// of the saved states above: one record = SymPack<G>::PER_STEP doubles of S (slot 0: 64 contiguous doubles,
// every slot k >= 1: 32 contiguous doubles written by the owner lanes) + 64 doubles of F.  Every store/load
// instruction moves one contiguous 256/512-byte run; 44 instead of 96 bytes per series and step at G = C = 8.
template <int G>
struct CkptRec {
  static constexpr int DOUBLES = SymPack<G>::PER_STEP + 2 * kWave;  // S (packed) + F_j + W_j, per wavefront and checkpoint
};
template <int G>
__device__ __forceinline__ void ckpt_store(double *rec, int lane, const int (&soff)[G], const double (&SX)[G], double F,
                                           double w) {
  const int j = lane & (G - 1);
  rec[lane] = SX[0];
#pragma unroll
  for (int b = 0; (1 << b) < G; ++b) {
    if ((j & (1 << b)) == 0) {  // owner lanes of slots k in [2^b, 2^(b+1))
#pragma unroll
      for (int kk = (1 << b); kk < (2 << b); ++kk) rec[soff[kk]] = SX[kk];
    }
  }
  rec[SymPack<G>::PER_STEP + lane] = F;
  rec[SymPack<G>::PER_STEP + kWave + lane] = w;
}
template <int G>
__device__ __forceinline__ void ckpt_load(const double *rec, int lane, const int (&soff)[G], double (&SX)[G], double &F,
                                          double &w) {
#pragma unroll
  for (int k = 0; k < G; ++k) SX[k] = rec[soff[k]];
  F = rec[SymPack<G>::PER_STEP + lane];
  w = rec[SymPack<G>::PER_STEP + kWave + lane];
}

// the backward-recursion reverse sweep re-anchors at every kAnchor-th checkpoint (every 32 rows at C = 8) ...
constexpr int kAnchor = 4;
// ... where the decays it has to invert in between allow.  One workgroup per wavefront w of the sweep (spw series):
// words[2 w] = max over its series of c_max x (longest time between two anchors kAnchor segments apart), words[2 w + 1] the
// same for single segments.  NaN and negative spans -- unsorted times -- are stored as +inf, which no guard accepts.
__global__ __launch_bounds__(256) void k_anchor_spans(int64_t B, int64_t N, int J, int C, int spw,
                                                      const double *__restrict__ t, int64_t t_bs,
                                                      const double *__restrict__ c, int64_t c_bs,
                                                      unsigned long long *__restrict__ words) {
  __shared__ double cm[kWave], red[2][256 / kWave];
  __shared__ int anybad;
  const int tid = threadIdx.x;
  const int64_t b0 = (int64_t)blockIdx.x * spw;
  if (tid == 0) anybad = 0;
  if (tid < spw) {
    const int64_t b = b0 + tid < B ? b0 + tid : B - 1;
    double m = 0.0;
    for (int j = 0; j < J; ++j) m = fmax(m, c[b * c_bs + j]);
    cm[tid] = m;
  }
  __syncthreads();
  const int64_t nseg = (N - 1 + C - 1) / C, ng = (nseg + kAnchor - 1) / kAnchor;
  double g1 = 0.0, ga = 0.0;
  bool bad = false;
  for (int64_t i = tid; i < (int64_t)spw * ng; i += blockDim.x) {
    const int sl = (int)(i / ng);
    const int64_t grp = i % ng, b = b0 + sl < B ? b0 + sl : B - 1;
    const double *tb = t + b * t_bs;
    const int64_t r0 = grp * kAnchor * C;
    double tp = tb[r0 < N - 1 ? r0 : N - 1];
    const double tfirst = tp;
    double s1 = 0.0;
#pragma unroll
    for (int q = 1; q <= kAnchor; ++q) {
      const int64_t r = r0 + (int64_t)q * C;
      const double tq = tb[r < N - 1 ? r : N - 1];
      bad = bad || !(tq - tp >= 0.0);
      s1 = fmax(s1, tq - tp);
      tp = tq;
    }
    g1 = fmax(g1, cm[sl] * s1);
    ga = fmax(ga, cm[sl] * (tp - tfirst));
  }
  bad = bad || !(g1 >= 0.0) || !(ga >= 0.0);
#pragma unroll
  for (int sft = 1; sft < kWave; sft <<= 1) { g1 = fmax(g1, __shfl_xor(g1, sft, kWave)); ga = fmax(ga, __shfl_xor(ga, sft, kWave)); }
  if (bad) atomicOr(&anybad, 1);
  if ((tid & (kWave - 1)) == 0) { red[0][tid / kWave] = ga; red[1][tid / kWave] = g1; }
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < (int)(blockDim.x / kWave); ++k) { red[0][0] = fmax(red[0][0], red[0][k]); red[1][0] = fmax(red[1][0], red[1][k]); }
    // (+inf, not NaN: every consumer asks `word > kBackwardGuard`, which a NaN would answer with "stable")
    const double inf = __builtin_inf();
    words[2 * blockIdx.x] = (unsigned long long)__double_as_longlong(anybad ? inf : red[0][0]);
    words[2 * blockIdx.x + 1] = (unsigned long long)__double_as_longlong(anybad ? inf : red[1][0]);
  }
}

// ---- coefficient-level form of the eight-lane pair (TT; SURVEY.md section 8f-1, driver.cpp:456-474) ------------------------
// Lane j forms ITS column of U_n and V_n from the celerite coefficients and x_n: a real term r = j (U = ar_r, V = 1), or the
// cos (even) / sin (odd) column of complex term k = (j - Jr) / 2 -- J = Jr + 2 Jc = 8, 4 or 2 makes Jr even, so the two lanes of
// an XOR-1 pair are two real terms or the two columns of ONE complex term.  One sincos per lane and row (dc = 0 for a real term:
// cos = 1, sin = 0 exactly).  No U / V rows are read and no bU / bV rows written: the reverse sweep folds the reverse of the
// recipe (c2_terms.hip: k_terms_rev) into its step.  Only the branch-free sincos lives in these kernels: a group of 64 series
// with a phase beyond its range (raw Julian dates times a fast frequency) is closed in the gate like one beyond the backward
// guard, and the composed chain, launched behind on the same words, takes it.  The rates come as the c array (k_rates).
struct TermsArgs8 {
  const double *ar, *ac, *bc, *dc;
  int batched, Jc;
};
struct TermsGrads8 {
  double *bar, *bcr, *bac, *bbc, *bcc, *bdc;
};
struct LaneTerm {
  double A, Bq, D, A0;
  bool odd, re;
  __device__ __forceinline__ void load(const TermsArgs8 &T, int64_t b, int j, int J) {
    const int JC = T.Jc, JR = J - 2 * JC;
    const int64_t br = T.batched ? b * JR : 0, bk = T.batched ? b * JC : 0;
    re = j < JR;
    odd = !re && ((j - JR) & 1);
    if (re) {
      A = T.ar[br + j]; Bq = 0.0; D = 0.0;
    } else {
      const int k = (j - JR) >> 1;
      const double bb = T.bc[bk + k];
      A = T.ac[bk + k]; Bq = odd ? -bb : bb; D = T.dc[bk + k];
    }
    double sum = 0.0;   // driver.cpp:456-458: the sum of ar, then of ac
    for (int r = 0; r < JR; ++r) sum += T.ar[br + r];
    for (int k = 0; k < JC; ++k) sum += T.ac[bk + k];
    A0 = sum;
  }
  // own column of U_n and of V_n (cos column: U = ac cos + bc sin; sin column: U = ac sin - bc cos)
  __device__ __forceinline__ void uv(double x, double &u, double &v) const {
    double sn, cs;
    const double ph = D * x;
    sincos_cw_fast(ph, sn, cs);
    // (the gate looks at the ENDS of the grid; a row of an UNSORTED grid whose phase leaves the range all the same must not
    // pass for a result: NaN, as k_matrices marks such rows)
    if (!(fabs(ph) < kSincosFastMax)) sn = cs = __builtin_nan("");
    const double p = odd ? sn : cs, q = odd ? cs : sn;
    u = fma(A, p, Bq * q);
    v = p;
  }
};
__device__ __forceinline__ bool tt_group_open(const unsigned long long *gate, int64_t b0) {
  return __longlong_as_double((long long)gate[b0 >> 6]) <= kBackwardGuard;   // (NaN / +inf: closed)
}
// One word per group of 64 series for the coefficient-level pair: the largest single-segment span word of its wavefronts
// (k_anchor_spans, words[2 w + 1]; nullptr: none), +inf if a phase dc x of the group leaves the range of the branch-free sincos
// (x sorted: the largest |x| of a series sits at one of its ends).  One wavefront per group, a lane per series.  head[0]: the
// largest word of the launch, head[1]: closed groups -- both zeroed by the launcher on the same stream.
__global__ __launch_bounds__(kWave) void k_tt8_gate(int64_t B, int64_t N, int64_t nwaves, int wpg,
                                                    const unsigned long long *__restrict__ words, TermsArgs8 T,
                                                    const double *__restrict__ x, int64_t x_bs,
                                                    unsigned long long *__restrict__ head, unsigned long long *__restrict__ gate) {
  const int64_t g = blockIdx.x, b = 64 * g + threadIdx.x;
  bool fast = true;
  if (b < B) {
    const double xm = fmax(fabs(x[b * x_bs]), fabs(x[b * x_bs + N - 1]));
    for (int k = 0; k < T.Jc; ++k) fast = fast && (fabs(T.dc[(T.batched ? b * T.Jc : 0) + k]) * xm < kSincosFastMax);
  }
  double m = 0.0;
  if (words && (int)threadIdx.x < wpg && wpg * g + threadIdx.x < nwaves) {
    m = __longlong_as_double((long long)words[2 * (wpg * g + threadIdx.x) + 1]);
    if (m != m) m = __builtin_inf();
  }
#pragma unroll
  for (int sft = 1; sft < kWave; sft <<= 1) m = fmax(m, __shfl_xor(m, sft, kWave));
  if (!__all(fast)) m = __builtin_inf();
  if (threadIdx.x == 0) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(m);   // (m >= 0: the bit patterns order like the numbers)
    gate[g] = bits;
    atomicMax(head, bits);
    if (!(m <= kBackwardGuard)) atomicAdd(head + 1, 1ull);
  }
}

// DG: the next step's p and U gathered by DPP permutes instead of through LDS.  The LDS form costs the VALU nothing but
// makes the wavefront wait for two LDS round trips per step -- which one would expect to hurt when the grid gives every
// wavefront a SIMD of its own; measured it does not: the DPP form is 2 - 9 % slower there too (launch_fwd).
// LN (G = R = 8, no padding, N even, 16-byte aligned U and V; C2_LOGLIK_LINES=1, OFF by default): the rows of U and V arrive
// as whole aligned 128-byte LINES.  A width-8 row is 64 bytes, so the eight series of a wavefront make a row request eight
// half-lines -- and once every SIMD of a CU has its wavefront these kernels queue at the CU's address unit, which prices a
// request by the runs it touches, not by its bytes (profiles/r05_lines.md: without the U / V requests the forward pass at 8192
// series runs at its clock-scaled floor).  MEASURED: the LDS instructions of the detour cost a lone wavefront what the
// address unit gives back -- 8192 series 4.69 against 4.72 ms, 1024 - 4096 series 9 - 12 % SLOWER -- hence off.  Rows (2P, 2P+1) of a series share a line: one 16-byte piece per lane fetches the pair for all eight
// series (a ring of four pairs in registers, eight rows ahead), a per-wave LDS tile turns pieces into the lanes' own
// elements one pair ahead of their use.
template <int G, int R, int C, int MODE, bool PAD, int OCC = C2_FWD_OCC, bool DG = false, bool LN = false, bool TT = false>
__global__ __launch_bounds__(kWave, OCC) void k_loglik_fwd(int64_t B, int64_t N, int Jrt, const double *__restrict__ t,
                                                         int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                         const double *__restrict__ a,
                                                         const double *__restrict__ U,
                                                         const double *__restrict__ V,
                                                         const double *__restrict__ y, double *__restrict__ ll,
                                                         int32_t *__restrict__ flag, double *__restrict__ ckpt,
                                                         int64_t nseg, double *__restrict__ Wst,
                                                         double2 *__restrict__ DZst,
                                                         const unsigned long long *__restrict__ gate,
                                                         const unsigned long long *__restrict__ segguard = nullptr,
                                                         TermsArgs8 T8 = TermsArgs8{},
                                                         const unsigned long long *__restrict__ tgate = nullptr) {
  // `gate` (nullable): this launch is the fallback of the one-lane-per-series path and runs only when the stability
  // guard that path measured exceeds kBackwardGuard (stream-ordered device decision, no host round trip).
  if (gate_closed(gate, (int64_t)blockIdx.x * (kWave / G))) return;   // (the series of a wavefront share a group of 64)
  // TT: the coefficient-level form (`a` = the white-noise diagonal, U / V not read); `tgate`: one word per group of 64 series
  static_assert(!TT || (G >= 2 && G <= 8 && !PAD && !LN && MODE != 2), "coefficient-level form: full groups of two to eight lanes");
  if constexpr (TT) { if (!tt_group_open(tgate, (int64_t)blockIdx.x * (kWave / G))) return; }
  // MODE 1 with Wst and segguard (the reverse sweep by the BACKWARD recursion, k_loglik_rev<..., BACK>): W rows are
  // recorded as well, and one more checkpoint holds the state after the last row.  segguard[2 w], [2 w + 1] (k_anchor_spans)
  // say what the backward recursion of wavefront w would have to invert: largest c_j x (time spanned by kAnchor segments /
  // by one segment).  Within the guard over the long spans only every kAnchor-th checkpoint is written -- the sweep
  // re-anchors there; otherwise all of them (re-anchoring at every segment, or the replay sweep).
  // MODE 0: log-likelihood only.  MODE 1 (CKPT): also the records of the reverse sweep.  MODE 2 (FACTOR): the
  // same pass used as core::factor -- Wst/DZst are the caller's W (B,N,J) and d (B,N); nothing is stored after
  // the first non-positive pivot, exactly like the reference's early return (forward.hpp:128).
  constexpr bool CKPT = MODE == 1, FACTOR = MODE == 2, REC = MODE != 0;
  static_assert(!CKPT || R % C == 0, "block length must be a multiple of the checkpoint interval");
  constexpr int SPW = kWave / G;         // series per wavefront
  constexpr int NV = (R + G - 1) / G;    // vector loads per scalar stream per block of R rows
  __shared__ __attribute__((aligned(16))) double xs[3][kWave];
  // Per-series scalar streams (t, a, y) are read TRANSPOSED: one instruction fetches R consecutive rows of a
  // series (lane j <-> row j), two blocks are staged here, and each step broadcasts its row to the group with
  // one ds_read.  (A per-step load in which 8 lanes fetch the same 8 bytes costs the CU's address unit as
  // much as a full 512-byte access: profiles/r01_ubench_instruction_costs.md.)
  __shared__ __attribute__((aligned(16))) double sin_[2][3][SPW][R];
  __shared__ __attribute__((aligned(16))) double2 sout[SPW][R];  // (d_n, z_n) of the current block
  const int J = PAD ? Jrt : G;  // PAD == false: J == G is a compile-time constant (immediate row strides)
  const Geo<G> L(B, J);
  const int lane = L.lane, j = L.j, grp = L.lane / G;
  const bool act = PAD ? L.act : true;
  const int64_t ot = (int64_t)L.sl * t_bs, on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + L.jj;
  const double *tb = t + L.b0 * t_bs + ot, *ab = a + L.b0 * N + on, *yb = y + L.b0 * N + on;
  const double *Ub = U + L.b0 * N * J + oj, *Vb = V + L.b0 * N * J + oj;
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  LaneTerm lt;
  if constexpr (TT) lt.load(T8, L.b, j, G);
  // (with W records -- the backward-recursion form -- a wavefront owns one more checkpoint: the state after its last row)
  double *ckw = CKPT ? ckpt + (size_t)blockIdx.x * (nseg + (segguard != nullptr ? 1 : 0)) * CkptRec<G>::DOUBLES : nullptr;
  int soff[G];  // packed-S offsets of this lane (slot 0 lives at [lane])
  soff[0] = lane;
#pragma unroll
  for (int k = 1; k < G; ++k) soff[k] = SymPack<G>::off(lane, k);
  // per-step record for the reverse sweep (CKPT only): (d_n, z_n).  W_n is NOT recorded: the reverse sweep replays
  // W_n = (V_n - tau_n) / d_n bit for bit next to S_n, from the W of the checkpointed row (64 B per step less to
  // write and to read back; the forward pass is HBM-bound).  FACTOR: W is the caller's output.
  const bool wrec = CKPT && Wst != nullptr;   // (uniform)
  double *wst = FACTOR ? Wst + L.b0 * N * J + oj : nullptr;
  // the W record of the backward-recursion sweep is private to the wavefront: lane-major, one dense 512-byte run per row
  double *wrp = wrec ? Wst + (size_t)blockIdx.x * N * kWave + lane : nullptr;
  double2 *dzst = CKPT ? DZst + L.b0 * N + on : nullptr;
  double *dst = FACTOR ? reinterpret_cast<double *>(DZst) + L.b0 * N + on : nullptr;
  const bool stw = PAD ? (L.valid && act) : true;  // duplicate stores of identical values are harmless

  double SX[G];
#pragma unroll
  for (int k = 0; k < G; ++k) SX[k] = 0.0;
  // Row 0 is the first step of block 0, from the neutral state "row -1": S = 0, F = 0, W = 0, z = 0, d = 1 at t_0 (p = 1) --
  // the step then yields d_0 = a_0, W_0 = V_0 / d_0, z_0 = y_0 (forward.hpp:107-108) like any other row.  Blocks therefore
  // cover rows [R b, R b + R): every transposed request of t, a, y and every store of (d, z) / d is a whole aligned run
  // (blocks that started at row 1 straddled two 64-byte sectors per series and instruction: 2 % of the gradient pair at
  // 8192 series, profiles/r05_alignment.md).
  double d = 1.0;
  double rd = 1.0;
  double w = 0.0;
  double z = 0.0;
  double F = 0.0;
  double prod = 1.0;     // running product of pivots, renormalised with frexp -> log det
  int eacc = 0;
  double quad = 0.0;
  int32_t fl = 0;

  // ---- transposed scalar streams: registers hold the rows of block b+2, LDS the rows of blocks b, b+1 ----
  double vt[NV], va[NV], vy[NV];
  auto vload = [&](int64_t nb) {  // rows nb .. nb+R-1, lane j takes rows nb + m*G + j
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t row = nb + m * G + j;
      row = (row < N) ? row : N - 1;
      vt[m] = tb[row]; va[m] = ab[row]; vy[m] = yb[row];
    }
  };
  auto vstage = [&](int q) {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == R || idx < R) {
        sin_[q][0][grp][idx] = vt[m]; sin_[q][1][grp][idx] = va[m]; sin_[q][2][grp][idx] = vy[m];
      }
    }
  };
  vload(0); vstage(0);
  vload(R); vstage(1);
  vload(2 * R);

  // ---- row streams (U_n, V_n): register ring, one row per step, R rows ahead ------------------------------
  // (the ring is RR = min(R, C2_FWD_RING) rows long: blocks of sixteen rows -- scalar requests of whole 128-byte lines -- keep a
  // ring of eight)
  constexpr int RR = R < C2_FWD_RING ? R : C2_FWD_RING;
  double ru[LN ? 1 : RR], rv[LN ? 1 : RR];
  const double *up = Ub, *vp = Vb;  // row n0 of the current block
  auto load_row = [&](int r, int ahead, int64_t n, bool clamp) {  // row n = n0 + ahead
    if constexpr (!LN && !TT) {
      int64_t o = ahead;
      if (clamp && n >= N) o -= n - (N - 1);
      ru[r] = act ? up[o * J] : 0.0; rv[r] = act ? vp[o * J] : 0.0;
    }
  };
  // LN: pair P = rows (2P, 2P+1) of the lane's series, piece j of its 128 bytes; slot P % 4 of the ring
  static_assert(!LN || (G == 8 && R == 8 && !PAD), "line staging: full groups of eight lanes, blocks of eight rows");
  __shared__ __attribute__((aligned(16))) double2 ltile[LN ? 2 : 1][LN ? kWave : 1];   // [U | V][series][piece]
  const double2 *Ul = reinterpret_cast<const double2 *>(U + L.b0 * N * J + (int64_t)L.sl * N * J) + j;
  const double2 *Vl = reinterpret_cast<const double2 *>(V + L.b0 * N * J + (int64_t)L.sl * N * J) + j;
  const int64_t plast = N / 2 - 1;
  double qux[LN ? 4 : 1], quy[LN ? 4 : 1], qvx[LN ? 4 : 1], qvy[LN ? 4 : 1];   // (plain doubles: arrays of double2 end up in scratch)
  double cu[2] = {0.0, 0.0}, cv[2] = {0.0, 0.0};   // own elements of the current pair's two rows
  double nu[2] = {0.0, 0.0}, nv[2] = {0.0, 0.0};   // ... of the next pair
  const double *ltu = reinterpret_cast<const double *>(ltile[0]) + grp * 16 + j;
  const double *ltv = reinterpret_cast<const double *>(ltile[LN ? 1 : 0]) + grp * 16 + j;
  auto pair_load = [&](int slot, int64_t P) {
    const int64_t Pc = P < plast ? P : plast;
    const double2 a2 = Ul[Pc * 8], b2 = Vl[Pc * 8];
    qux[slot] = a2.x; quy[slot] = a2.y; qvx[slot] = b2.x; qvy[slot] = b2.y;
  };
  auto pair_stage = [&](int slot) {
    ltile[0][lane] = make_double2(qux[slot], quy[slot]);
    ltile[LN ? 1 : 0][lane] = make_double2(qvx[slot], qvy[slot]);
  };
  if constexpr (LN) {
    cu[0] = Ub[0]; cu[1] = Ub[J]; cv[0] = Vb[0]; cv[1] = Vb[J];   // pair 0 (rows 0 and 1: the first two steps)
    pair_load(1, 1); pair_load(2, 2); pair_load(3, 3); pair_load(0, 4);   // (the first step stages pair 1 and requests pair 5)
  } else {
#pragma unroll
    for (int r = 0; r < RR; ++r) load_row(r, r, r, true);
  }

  const bool sparse = wrec && !(__longlong_as_double((long long)segguard[2 * blockIdx.x]) > kBackwardGuard);   // (uniform)
  // prepare step 0 (p = 1: the neutral state sits at t_0)
  lds_order();
  double tnext = sin_[0][0][grp][0];
  double pc = 1.0, uc = LN ? cu[0] : ru[0];
  double vcur = 0.0;   // (TT) own column of V of the current row
  if constexpr (TT) lt.uv(tnext, uc, vcur);
  double pXc[G], uXc[G];
  if constexpr (DG) {
    xgather_dpp<G>(pc, xs[0], lane, pXc);
    xgather_dpp<G>(uc, xs[1], lane, uXc);
  } else {
    xs[0][lane] = pc; xs[1][lane] = uc;
    lds_order();
    xgather_lds<G>(xs[0], lane, pXc);
    xgather_lds<G>(xs[1], lane, uXc);
    lds_order();
  }

  auto block = [&](int64_t n0, int q, auto checked_tag) {
    constexpr bool CHECKED = decltype(checked_tag)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t n = n0 + r;
      if (!CHECKED || n < N) {
        if (CKPT && (r % C == 1 % C)) {  // state after row n-1 = C m = checkpoint m (before the step of row C m + 1)
          const int64_t m = (n - 1) / C;
          if (!sparse || (m % kAnchor == 0 && m > 0))   // (uniform; the backward recursion never reads the state after row 0)
            ckpt_store<G>(ckw + m * CkptRec<G>::DOUBLES, lane, soff, SX, F, w);
        }
        const double tn = tnext, yn = sin_[q][2][grp][r];
        const double an = TT ? sin_[q][1][grp][r] + lt.A0 : sin_[q][1][grp][r];
        const double tn1 = (r + 1 < R) ? sin_[q][0][grp][r + 1] : sin_[q ^ 1][0][grp][0];
        // (a) next step's p and U -> LDS -> XOR gathers (consumed by the next iteration)
        const int rn = (r + 1) % R;
        double v, un1;
        if constexpr (LN) {
          // blocks start at multiples of eight: r even <=> n even <=> the first row of pair P = n / 2
          if (r % 2 == 0) {
            v = cv[0]; un1 = cu[1];
            lds_order();
            pair_stage((r / 2 + 1) % 4);                            // pair P + 1 -> tile (read back at the end of this step)
            pair_load((r / 2 + 1) % 4, n / 2 + 5);                  // its slot: pair P + 5, eight rows ahead
          } else {
            v = cv[1]; un1 = nu[0];                                 // (pair P + 1, read back during the step before)
          }
        } else if constexpr (TT) {
          v = vcur;
          lt.uv(tn1, un1, vcur);   // row n + 1 (beyond the last row: the clamped time, unused)
        } else {
          v = rv[LN ? 0 : r % RR]; un1 = ru[LN ? 0 : rn % RR];
        }
        const double pn1 = exp_decay(cj * (tn - tn1));
        double pXn[G], uXn[G];
        if constexpr (DG) {
          xgather_dpp<G>(pn1, xs[0], lane, pXn);
          xgather_dpp<G>(un1, xs[1], lane, uXn);
        } else {
          xs[0][lane] = pn1; xs[1][lane] = un1;
          lds_order();
          xgather_lds<G>(xs[0], lane, pXn);
          xgather_lds<G>(xs[1], lane, uXn);
          lds_order();
        }
        // (b) the chain of step n
        fwd_chain<G>(pc, uc, v, an, yn, pXc, uXc, SX, F, w, d, z, rd, xs[2], lane);
        if (REC) {
          if (FACTOR && stw && (((fl == 0) & (d > 0.0)) | (n == 0))) wst[n * J] = w;   // (W_0 unconditionally: forward.hpp:108)
          if (CKPT && wrec) wrp[(size_t)n * kWave] = w;
          sout[grp][r] = make_double2(d, z);
        }
        // (c) refill ring slot r with row n + R
        load_row(r % RR, r + RR, n + RR, CHECKED);
        // forward.hpp:128: first non-positive pivot (NaN passes, as in the reference); no early exit --
        // a failed series simply runs to the end on garbage, its outputs are flagged.
        fl = ((fl == 0) & (d <= 0.0)) ? (int32_t)n : fl;
        prod *= d;
        quad = fma(z * z, rd, quad);
        if (r % 2 == 1 || r == R - 1) {  // renormalise every second row: safe for pivots in 1e-150 .. 1e150
          int e;
          prod = frexp(prod, &e);
          eacc += e;
        }
        tnext = tn1;
        pc = pn1; uc = un1;
        if (LN && r % 2 == 1) { cu[0] = nu[0]; cu[1] = nu[1]; cv[0] = nv[0]; cv[1] = nv[1]; }
        if (LN && r % 2 == 0) {   // own elements of pair P + 1: used from the next step on, so their latency is hidden
          lds_order();
          nu[0] = ltu[0]; nu[1] = ltu[8]; nv[0] = ltv[0]; nv[1] = ltv[8];
        }
#pragma unroll
        for (int k = 0; k < G; ++k) { pXc[k] = pXn[k]; uXc[k] = uXn[k]; }
      }
    }
    // end of block: flush (d, z) of the block transposed, stage block b+2's scalars, fetch block b+3's
    lds_order();
    if (REC) {
#pragma unroll
      for (int m = 0; m < NV; ++m) {
        const int idx = m * G + j;
        if ((G * NV == R || idx < R) && (!CHECKED || n0 + idx < N)) {
          if (CKPT) dzst[n0 + idx] = sout[grp][idx];
          else if (fl == 0 || n0 + idx <= fl) dst[n0 + idx] = sout[grp][idx].x;  // rows 0..n of d (forward.hpp:127-128)
        }
      }
    }
    vstage(q);
    vload(n0 + 3 * R);
    lds_order();
  };
  int64_t n0 = 0;
  int q = 0;
  auto advance = [&]() { up += R * J; vp += R * J; q ^= 1; };
  for (; n0 + 2 * R <= N; n0 += R) { block(n0, q, std::false_type{}); advance(); }  // every row load in range
  for (; n0 < N; n0 += R) { block(n0, q, std::true_type{}); advance(); }

  if (wrec) ckpt_store<G>(ckw + nseg * CkptRec<G>::DOUBLES, lane, soff, SX, F, w);   // the state after the last row
  if (L.valid && j == 0) {
    flag[L.b] = fl;
    if (!FACTOR) {
      int e;
      prod = frexp(prod, &e);
      const double logdet = log(prod) + (double)(eacc + e) * kLn2;
      ll[L.b] = fl ? -INFINITY : -0.5 * (logdet + (double)N * kLog2Pi) - 0.5 * quad;
    }
  }
}

// =============================================================================
// Reverse sweep with segment recomputation.  For segment k (rows n_lo = 1 + kC ... n_lo + C - 1), last
// segment first:
//   1. reload checkpoint k (state of row n_lo - 1) and the segment's input rows;
//   2. recompute the C forward steps; keep per step: the post-decay state column S_n(:,j) in registers,
//      p_n, U_n, W_{n-1} in LDS (the same slots the forward gathers use), F_n, 1/d_n, z_n, dt_n;
//   3. issue the loads of segment k-1 (they land while step 4 runs);
//   4. run the fused reverse steps n = n_lo + C - 1 ... n_lo.
// Reverse step n (reference internal.hpp:226-245 for the solve, reverse.hpp:58-81 for the factor), with
// bz_n, ba_n, bV_n the already complete cotangents of row n on entry:
//   bU_n  = -bz_n F_n  -  (bV_n + 2 ba_n U_n) S_n                 (S_n post-decay = workspace row * diag(p))
//   bF   += -U_n bz_n ;  M -= U_n^T y + y^T U_n  (y = bV_n + ba_n U_n;  M = bS + bS^T, see k_factor_rev)
//   bp    = F_n.bF + diag(S_n M) ;  bc += dt bp ;  f = sum c bp ;  bt_n = f_{n+1} - f_n
//   bF    = p bF ;  M = P M P ;  q = W_{n-1} M
//   G = W_{n-1}.bF ,  Q = q.W_{n-1}
//   bz_{n-1} = -z_{n-1}/d_{n-1} + G                                 (seed + V_{n-1} bF)
//   bV_{n-1} = z_{n-1} bF / d_{n-1} + q                             (bW_{n-1}/d_{n-1} + W_{n-1}(bS+bS^T))
//   ba_{n-1} = bd_{n-1} - Q/2 - z_{n-1} G / d_{n-1}                 (bd + w bS w^T - W_{n-1}.bV_{n-1})
// with the seeds bd = (z^2/d - 1)/(2d), the derivative of the log-likelihood w.r.t. d (and -z/d w.r.t. z).
// =============================================================================
// A double parked in a pair of accumulation registers (gfx950: 256 AGPRs per lane at one wavefront per SIMD, unused by
// these kernels).  Round trip = 4 v_accvgpr moves (~19 cycles of issue) against ~38 for ds_write_b64 + ds_read_b64.
__device__ __forceinline__ void apark(double x, int &lo, int &hi) {
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(lo) : "v"(__double2loint(x)));
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(hi) : "v"(__double2hiint(x)));
}
__device__ __forceinline__ double afetch(int lo, int hi) {
  int l, h;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(lo));
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(hi));
  return __hiloint2double(h, l);
}

// FR = true turns the same kernel into core::factor_rev (reverse.hpp:10-85) on the reference's own arguments: V is
// then the caller's W, `ckpt` the caller's S workspace -- of which only every C-th row is read, as the checkpoint the
// rows in between are replayed from (64 instead of 512 bytes of S per step at J = 8) --, fr_d / fr_bd / fr_bW the
// pivots and the incoming cotangents of d and W; the solve_lower_rev half (F, z, bF, by) drops out.
// BACK = true: no replay.  The forward pass recorded the W rows and a checkpoint after the last row as well; the sweep
// starts every segment from the checkpoint at its END and runs the recursion backward next to the reverse steps,
//     S_{n-1} = P_n^-1 S_n P_n^-1 - d_{n-1} w_{n-1}^T w_{n-1},   F_{n-1} = P_n^-1 F_n - w_{n-1} z_{n-1}
// (the inverse of forward.hpp:115-123 / internal.hpp:140-143) -- eight rows at most between two anchors, so the decays it
// inverts are bounded by the wavefront's `segguard` word: beyond kBackwardGuard this kernel returns at once and the
// replay form, launched behind it on the same word, takes the wavefront.  Phase B below (a third of the instructions)
// drops out, and so do the 128 accumulation registers the replayed states waited in.
// OCC = 2 (BACK only): two wavefronts per SIMD -- for batches with more wavefronts than the chip has SIMDs; the per-step
// vectors then wait in LDS instead of accumulation registers (256 registers per wavefront all told).
// SC = true (BACK only): the backward recursion and the adjoint recursion in a SCALED FRAME.  With g_n = exp(-c (t_ref - t_n)),
// t_ref the time of the anchor row above (so g <= 1 and, within the guard, >= e^-2), the scaled states
//     S^_n = G_n S_n G_n,   M^_n = G_n^-1 M_n G_n^-1,   F~_n = g_n F_n,   bF-_n = bF_n / g_n,   bV-_n = bV_n / g_n
// obey recursions WITHOUT decay factors: S^_{n-1} = S^_n - d_{n-1} w~ w~^T, F~_{n-1} = F~_n - w~ z_{n-1} (w~ = g_{n-1} W_{n-1});
// M^ and bF- pass from row n to n-1 unchanged; diag(S M) = diag(S^ M^), F.bF = F~.bF-; q = W_{n-1} M = g_{n-1} (w~ M^), so
// bV-_{n-1} = (z/d) bF- + w~ M^ needs no factor at all.  Per step the gathers of p and 1/p and the 32 multiplications by
// p_i p_j, 1/(p_i p_j) drop out (five gathered vectors -> three: u- = u / g_n, x- = bV- + 2 ba u-, w~); what is left of the
// frame is one factor on u, w and on the rows bU_n, bV_n on their way out, and a change of frame at every anchor.
// LN (SC, G = C = 8, no padding, N even, 16-byte aligned U, bU, bV): rows of U, bU, bV move as whole aligned 128-byte lines --
// rows (2P, 2P+1) of a series share one -- through per-wave LDS tiles, as in k_loglik_fwd<..., LN>: four line requests per
// segment for U (rows 8k .. 8k+7; row 8k is handed down to the segment below), one store per completed pair for bU and bV
// (the lane's element goes into a two-row tile; a pair is complete at its even row and leaves during the step after).
template <int G, int C, bool PAD, bool FR, bool BACK = false, int OCC = C2_REV_OCC, bool SC = false, bool LN = false, bool TT = false>
__global__ __launch_bounds__(kWave, OCC) void k_loglik_rev(int64_t B, int64_t N, int Jrt, const double *__restrict__ t,
                                                         int64_t t_bs, const double *__restrict__ c, int64_t c_bs,
                                                         const double *__restrict__ U,
                                                         const double *__restrict__ V,
                                                         const double2 *__restrict__ DZst,
                                                         const double *__restrict__ ckpt, int64_t nseg,
                                                         const int32_t *__restrict__ flag, double *__restrict__ bt,
                                                         double *__restrict__ bc, double *__restrict__ ba,
                                                         double *__restrict__ bU, double *__restrict__ bV,
                                                         double *__restrict__ by, const double *__restrict__ fr_d,
                                                         const double *__restrict__ fr_bd,
                                                         const double *__restrict__ fr_bW,
                                                         const unsigned long long *__restrict__ gate,
                                                         const unsigned long long *__restrict__ segguard = nullptr,
                                                         TermsArgs8 T8 = TermsArgs8{}, TermsGrads8 G8 = TermsGrads8{},
                                                         const unsigned long long *__restrict__ tgate = nullptr) {
  // TT (coefficient-level form): bt, ba, by are bx, bdiag, by; U, bU, bV, bc are not touched; G8 takes the coefficient gradients
  static_assert(!TT || (G >= 2 && G <= 8 && C == 8 && !PAD && !FR && BACK && OCC == 1 && SC && !LN && C2_REV_APARK),
                "coefficient-level form: the scaled-frame backward sweep on full groups of two to eight lanes");
  if constexpr (TT) { if (!tt_group_open(tgate, (int64_t)blockIdx.x * (kWave / G))) return; }
  static_assert(!(BACK && FR), "factor_rev replays from the caller's workspace");
  static_assert(!SC || BACK, "the scaled frame belongs to the backward-recursion sweep");
  static_assert(!LN || (SC && G == 8 && C == 8 && !PAD), "line staging: the scaled-frame sweep on full groups of eight lanes");
  if (gate_closed(gate, (int64_t)blockIdx.x * (kWave / G))) return;  // see k_loglik_fwd
  int astep = kAnchor;   // (BACK, uniform) segments between two anchors of the backward recursion
  if (segguard) {   // this wavefront by the backward recursion, or (the launch behind it) by the replay
    const bool unstable = __longlong_as_double((long long)segguard[2 * blockIdx.x + 1]) > kBackwardGuard;
    if (unstable == BACK) return;
    if (__longlong_as_double((long long)segguard[2 * blockIdx.x]) > kBackwardGuard) astep = 1;
  }
  constexpr int SPW = kWave / G;         // series per wavefront
  constexpr int NV = (C + G - 1) / G;    // vector loads per scalar stream per segment
  // per-step vectors of the current segment: [r][0] = p_n, [1] = U_n, [2] = W_{n-1}; own value at [lane]
  __shared__ __attribute__((aligned(16))) double vv[(C2_REV_APARK && G <= 8 && OCC == 1) ? 1 : C][4][kWave];  // [3] = bW_{n-1} (FR), 1 / p_n (BACK)
  // The replayed S_n columns wait for their reverse step in AGPRs (G <= 8: C*G*2 = 128 of them); wider groups
  // keep them in LDS, symmetric-packed.
  constexpr bool APARK = C2_REV_APARK && G <= 8 && OCC == 1;
  __shared__ __attribute__((aligned(16))) double sfL[(APARK || BACK) ? 1 : C][SymPack<G>::PER_STEP];  // saved S_n columns (packed)
  int sAlo[APARK ? C : 1][G], sAhi[APARK ? C : 1][G];
  int uAlo[APARK ? C : 1], uAhi[APARK ? C : 1], wAlo[APARK ? C : 1], wAhi[APARK ? C : 1];  // own U_n, W_{n-1}
  int xAlo[(APARK && FR) ? C : 1], xAhi[(APARK && FR) ? C : 1];                               // own bW_{n-1} (FR)
  // per-series scalars of rows n_lo-1 .. n_lo+C-1 (entry e <-> row n_lo-1+e): t, d, 1/d, z
  __shared__ __attribute__((aligned(16))) double rowT[C + 1][SPW], rowD[C + 1][SPW], rowR[C + 1][SPW], rowZ[C + 1][SPW];
  // per-series scalar outputs of the segment, flushed transposed as whole aligned runs of C rows (lane j <-> row C k + j):
  // ba_{n-1} and by_{n-1} (complete at the end of step n: rows C k .. C k + C - 1 of segment k); bt_n is complete at step n, so
  // its run C (k + 1) .. C (k + 1) + C - 1 is the segment's top row and the C - 1 rows the segment above left in the other
  // buffer (a segment covers rows C k + 1 .. C k + C: flushed as such, every store straddled two 64-byte sectors)
  __shared__ __attribute__((aligned(16))) double oBA[SPW][C], oBT[2][SPW][C], oBY[SPW][C];
  __shared__ __attribute__((aligned(16))) double xB[kWave];
  const int J = PAD ? Jrt : G;
  const Geo<G> L(B, J);
  const int lane = L.lane, j = L.j;
  const int grp = lane / G;
  const bool act = PAD ? L.act : true;
  // Store predicates.  Lanes of a padding (clamped) series recompute the LAST valid series bit for bit and
  // the G lanes of a group hold identical copies of every group scalar, so duplicate stores of identical
  // values to the same address are harmless: without column padding no store needs an exec mask.
  const bool st = PAD ? (L.valid && act) : true, st0 = PAD ? (L.valid && j == 0) : true;
  const int64_t ot = (int64_t)L.sl * t_bs, on = (int64_t)L.sl * N, oj = (int64_t)L.sl * N * J + L.jj;
  const double *tb = t + L.b0 * t_bs + ot, *Ub = U + L.b0 * N * J + oj, *Vb = V + L.b0 * N * J + oj;
  const double2 *dzb = FR ? nullptr : DZst + L.b0 * N + on;
  const double *ckw = FR ? nullptr : ckpt + (size_t)blockIdx.x * (nseg + (segguard != nullptr ? 1 : 0)) * CkptRec<G>::DOUBLES;
  const double *fdb = FR ? fr_d + L.b0 * N + on : nullptr, *fbdb = FR ? fr_bd + L.b0 * N + on : nullptr;
  const double *fbWb = FR ? fr_bW + L.b0 * N * J + oj : nullptr;
  const double *Scol = FR ? ckpt + (L.b0 + L.sl) * N * J * J + (int64_t)L.jj * J : nullptr;  // S[n, i + J j]: column j
  double *btb = bt + L.b0 * N + on, *bab = ba + L.b0 * N + on, *byb = by + L.b0 * N + on;
  double *bUb = bU + L.b0 * N * J + oj, *bVb = bV + L.b0 * N * J + oj;
  const double cj = act ? c[L.b * c_bs + j] : 0.0;
  if constexpr (!FR) {
    // Failed factorisation (uniform inside a group): the reference raises (driver.hpp:13-19), the batched kernel
    // reports it through flag[b] / ll[b] = -inf and fills the six gradients of THAT series with NaN -- never
    // stale memory -- so that a sum over the batch (shared t or c) cannot silently absorb garbage.
    if (flag[L.b] != 0) {
      const double nan = __builtin_nan("");
      for (int64_t n = j; n < N; n += G) {
        if (PAD ? L.valid : true) { btb[n] = nan; bab[n] = nan; byb[n] = nan; }
      }
      if constexpr (TT) {
        const int JC = T8.Jc, JR = G - 2 * JC;
        if (j < JR) { G8.bar[L.b * JR + j] = nan; G8.bcr[L.b * JR + j] = nan; }
        else if (((j - JR) & 1) == 0) {
          const int64_t o = L.b * JC + ((j - JR) >> 1);
          G8.bac[o] = nan; G8.bbc[o] = nan; G8.bcc[o] = nan; G8.bdc[o] = nan;
        }
        return;
      }
      for (int64_t n = 0; n < N; ++n) {
        if (st) { bUb[n * J] = nan; bVb[n * J] = nan; }
      }
      if (st) bc[L.b * J + j] = nan;
      return;
    }
  }
  LaneTerm lt;
  if constexpr (TT) lt.load(T8, L.b, j, G);
  // (TT) running sums of the lane: bU_own v_own + bU_p v_p (bac; a real term: sum bU_own), bU_own v_p - bU_p v_own (+- bbc),
  // x times the phase cotangent (+- bdc), the sum of ba; the lane's own trig column of every row of the segment parked
  double accA = 0.0, accB = 0.0, accD = 0.0, sba = 0.0;
  const double wD = (TT && !lt.re && !lt.odd) ? lt.D : 0.0;   // the even lane of a complex term carries g dc into bx
  int vAlo[TT ? C : 1], vAhi[TT ? C : 1];

  int soff[G];  // packed-S offsets of this lane (slot 0 lives at [lane])
  soff[0] = lane;
#pragma unroll
  for (int k = 1; k < G; ++k) soff[k] = SymPack<G>::off(lane, k);

  double MX[G];  // column j of M = bS + bS^T, XOR order
#pragma unroll
  for (int k = 0; k < G; ++k) MX[k] = 0.0;
  double bF = 0.0, carry = 0.0, bcj = 0.0;
  double bVn = 0.0, ban = 0.0, bzn = 0.0;

  // Records of the segment about to be replayed (loaded half a segment ahead).  Scalar streams (t and the
  // (d, z) pairs) are fetched TRANSPOSED: lane j takes row n_lo-1+j, i.e. entries 0..C-1; entry C (the last
  // row of the segment) is the previous iteration's entry 0 and is carried in registers.
  double vt[NV];
  double2 vdz[NV];
  double iu[C], iw[C], ibw[FR ? C : 1];
  // LN: raw 16-byte pieces of the segment's four U lines; tiles [series][row of the pair][element]
  double qux[LN ? 4 : 1], quy[LN ? 4 : 1], ucar = 0.0;
  __shared__ __attribute__((aligned(16))) double2 utile[LN ? 4 : 1][LN ? kWave : 1];
  __shared__ __attribute__((aligned(16))) double2 otile[LN ? 2 : 1][2][LN ? kWave : 1];   // [pair parity][bU | bV]
  const double2 *Ul = reinterpret_cast<const double2 *>(U + L.b0 * N * J + (int64_t)L.sl * N * J) + j;
  double2 *bUl = reinterpret_cast<double2 *>(bU + L.b0 * N * J + (int64_t)L.sl * N * J) + j;
  double2 *bVl = reinterpret_cast<double2 *>(bV + L.b0 * N * J + (int64_t)L.sl * N * J) + j;
  const int64_t plast = N / 2 - 1;
  const int lto = grp * 16 + j;   // the lane's element of row 0 of a pair tile, in doubles (row 1: + 8)
  double cS[G], cF = 0.0, cW = 0.0, tck = 0.0;
  auto load_segment = [&](int64_t k) {
    const int64_t n_lo = 1 + k * C;
    const bool full = n_lo + C <= N;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      int64_t row = n_lo - 1 + m * G + j;
      row = (row < N) ? row : N - 1;
      vt[m] = tb[row];
      if constexpr (FR) vdz[m] = make_double2(fdb[row], fbdb[row]);  // (d, bd) take the place of (d, z)
      else vdz[m] = dzb[row];
    }
    if constexpr (LN) {   // pairs 4k .. 4k+3 = rows 8k .. 8k+7 (beyond the last pair: that one again, unused)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t P = 4 * k + i;
        const double2 u2 = Ul[(P < plast ? P : plast) * 8];
        qux[i] = u2.x; quy[i] = u2.y;
      }
    }
#pragma unroll
    for (int r = 0; r < C; ++r) {
      const int64_t n = (full || n_lo + r < N) ? n_lo + r : N - 1;
      if constexpr (!LN && !TT) iu[r] = act ? Ub[n * J] : 0.0;
      if constexpr (BACK) iw[r] = V[((size_t)blockIdx.x * N + (n - 1)) * kWave + lane];   // the recorded W row n-1 (lane-major)
      else iw[r] = act ? Vb[(n - 1) * J] : 0.0;  // V row n-1 (-> W_{n-1} in the replay); FR: the caller's W row n-1
      if constexpr (FR) ibw[r] = act ? fbWb[(n - 1) * J] : 0.0;
    }
    if constexpr (FR) {
      // checkpoint = row m = n_lo-1 of the caller's S workspace: S_ws[m] = diag(p_m)(S + d w^T w) (forward.hpp:120),
      // the state after row m is S_ws[m] diag(p_m).  Column j in XOR order; row 0 of the workspace is zero.
      const int64_t m = n_lo - 1;
#pragma unroll
      for (int kk = 0; kk < G; ++kk) {
        const int i = j ^ kk;
        cS[kk] = (m >= 1 && act && i < J) ? Scol[m * J * J + i] : 0.0;
      }
      tck = tb[m >= 1 ? m - 1 : 0];
    } else if constexpr (BACK) {
      // the state at the END of the segment, where that is an anchor (every kAnchor-th checkpoint, the last row);
      // elsewhere the recursion carries on from the segment above
      if ((k + 1) % astep == 0 || k == nseg - 1)
        ckpt_load<G>(ckw + (k + 1) * CkptRec<G>::DOUBLES, lane, soff, cS, cF, cW);
    } else {
      ckpt_load<G>(ckw + k * CkptRec<G>::DOUBLES, lane, soff, cS, cF, cW);   // the state BEFORE the segment (after row n_lo - 1)
    }
  };

  // entry `cnt` of the first processed segment = row N-1
  double carT = tb[N - 1];
  double2 carDZ = FR ? make_double2(fdb[N - 1], fbdb[N - 1]) : dzb[N - 1];
  double carR = rcp_nr(carDZ.x);
  // FR: the last row's W and bW (seeds: bV_{N-1} = bW_{N-1}/d_{N-1}, ba_{N-1} = bd_{N-1} - W_{N-1}.bV_{N-1})
  const double wlast = (FR && act) ? Vb[(N - 1) * J] : 0.0, bwlast = (FR && act) ? fbWb[(N - 1) * J] : 0.0;
  if (nseg > 0) load_segment(nseg - 1);
  // -DC2_REV_TIMING: s_memtime deltas of the four phases, summed per wavefront into c2_dbg (tools/rev_phase_timing.py)
#ifdef C2_REV_TIMING
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tlast = 0;
#define C2_TCK(i) do { const unsigned long long tn_ = __builtin_readcyclecounter(); if (i) tacc[i] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define C2_TCK(i)
#endif

  double carS[BACK ? G : 1], carF = 0.0;   // (BACK) the recursed state handed from a segment to the one below
  double tref = 0.0, gtop = 1.0, igtop = 1.0;   // (SC) reference time of the frame; g, 1 / g of the row above the current segment's last step
#pragma unroll
  for (int i = 0; i < (BACK ? G : 1); ++i) carS[i] = 0.0;
  constexpr bool HOLD = C2_LOGLIK_PAIRLINES && G == 8 && C == 8 && NV == 1;   // (see the flush at the end of a segment)
  double hA = 0.0, hY = 0.0, hT = 0.0;
  bool hAok = false, hTok = false;
  int bq = 0;   // buffer of oBT the current segment writes
  for (int64_t k = nseg - 1; k >= 0; --k) {
    const int64_t n_lo = 1 + k * C;
    const int cnt = (N - n_lo < C) ? (int)(N - n_lo) : C;

    C2_TCK(0);
    // ---- phase A: scalar rows to LDS (lane-parallel), then p_n (C independent exps); own U_n parked ----------
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if (G * NV == C || idx < C) {
        rowT[idx][grp] = vt[m]; rowD[idx][grp] = vdz[m].x; rowR[idx][grp] = rcp_nr(vdz[m].x); rowZ[idx][grp] = vdz[m].y;
      }
    }
    lds_order();
    rowT[cnt][grp] = carT; rowD[cnt][grp] = carDZ.x; rowR[cnt][grp] = carR; rowZ[cnt][grp] = carDZ.y;
    lds_order();
    double dtv[C], pown[C], ipown[(BACK && APARK) ? C : 1];   // (SC: pown = g, ipown = 1 / g of rows n_lo-1 .. n_lo+C-2)
    const bool anchor = !BACK || (k + 1) % astep == 0 || k == nseg - 1;   // (BACK, uniform) re-anchor, or carry on
    if constexpr (LN) {   // the segment's U lines -> tiles (read back as the lanes' own elements behind the exponentials)
#pragma unroll
      for (int i = 0; i < 4; ++i) utile[i][lane] = make_double2(qux[i], quy[i]);
    }
    if constexpr (SC) {
      if (anchor) {   // change of frame: the anchor row (the segment's last) becomes the reference, g = 1 there
        double gX[G];
        xgather_dpp<G>(gtop, xB, lane, gX);
#pragma unroll
        for (int i = 0; i < G; ++i) MX[i] *= gX[i] * gtop;
        bF *= gtop;
        bVn *= gtop;
        tref = rowT[cnt][grp];
        gtop = 1.0;
        igtop = 1.0;
      }
      double igl[C];
#pragma unroll
      for (int r = 0; r < C; ++r) {
        // (rows of a short last segment beyond its end repeat the last row -- clamped loads -- so g = 1 there as well)
        const double tm = rowT[r][grp];
        dtv[r] = tm - rowT[r + 1][grp];
        pown[r] = exp_decay(cj * (tm - tref));
        igl[r] = rcp_nr(pown[r]);
        if constexpr (APARK) ipown[r] = igl[r];
        else { vv[r][0][lane] = pown[r]; vv[r][3][lane] = igl[r]; }
      }
      if constexpr (LN) {   // rows 8k+1 .. 8k+7 from this segment's lines, row 8k+8 handed down by the segment above
        lds_order();
        iu[C - 1] = ucar;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double *ut = reinterpret_cast<const double *>(utile[i]) + lto;
          if (i == 0) ucar = ut[0];
          else iu[2 * i - 1] = ut[0];
          iu[2 * i] = ut[8];
        }
      }
      if constexpr (TT) {   // own columns of U_n, V_n of rows n_lo .. n_lo + C - 1 (entry r + 1 of rowT)
#pragma unroll
        for (int r = 0; r < C; ++r) {
          double vown;
          lt.uv(rowT[r + 1][grp], iu[r], vown);
          apark(vown, vAlo[r], vAhi[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < C; ++r) {   // u- = U_n / g_n, w~ = W_{n-1} g_{n-1} wait for their step already scaled
        const double ign = (r == C - 1) ? igtop : igl[r + 1 < C ? r + 1 : 0];
        const double us = iu[r] * ign, ws = iw[r] * pown[r];
        if constexpr (APARK) {
          apark(ws, wAlo[r], wAhi[r]);
          apark(us, uAlo[r], uAhi[r]);
        } else {
          vv[r][1][lane] = us;
          vv[r][2][lane] = ws;
        }
      }
    } else {
      double tprev = rowT[0][grp];
#pragma unroll
      for (int r = 0; r < C; ++r) {
        const double tn = rowT[r + 1][grp];
        dtv[r] = tprev - tn;
        tprev = tn;
        pown[r] = exp_decay(cj * dtv[r]);
        if constexpr (BACK && APARK) ipown[r] = rcp_nr(pown[r]);
        if constexpr (APARK) {  // the prefetch registers are refilled half way through phase C
          if constexpr (BACK) apark(iw[r], wAlo[r], wAhi[r]);   // the recorded W_{n-1}
          apark(iu[r], uAlo[r], uAhi[r]);
          if constexpr (FR) apark(ibw[r], xAlo[r], xAhi[r]);
        } else {
          vv[r][0][lane] = pown[r];
          vv[r][1][lane] = iu[r];
          if constexpr (BACK) { vv[r][2][lane] = iw[r]; vv[r][3][lane] = rcp_nr(pown[r]); }
          if constexpr (FR) vv[r][3][lane] = ibw[r];
        }
      }
    }
    // ---- phase B: replay S_n = P (S + d w^T w) P, tau_n = U_n S_n, W_n = (V_n - tau_n) / d_n and
    // F_n = P (F + w z) -- with d, 1/d, z on record there are no reductions and no division.  S_n columns and
    // the lane's own W_{n-1} are parked (AGPRs; LDS for G > 8), F_n and tau_n stay in registers.
    double SX[G];
    const double pck = FR ? exp_decay(cj * (tck - rowT[0][grp])) : 1.0;  // FR: right scaling of the workspace row
#pragma unroll
    for (int i = 0; i < G; ++i) SX[i] = FR ? cS[i] * pck : (anchor ? cS[i] : carS[BACK ? i : 0]);
    double F = anchor ? cF : carF;
    const double Wck = cW;  // W of the checkpointed row n_lo-1
    double Fp[BACK ? 1 : C], tauS[BACK ? 1 : C];
    lds_order();
    C2_TCK(1);
#pragma unroll
    for (int r = 0; r < (BACK ? 0 : C); ++r) {
      if (r < cnt) {
        // W_{n-1} = (V_{n-1} - tau_{n-1}) / d_{n-1}, exactly as the forward pass formed it (forward.hpp:131)
        double wown = Wck;
        if constexpr (FR) wown = iw[r];  // W is an input of factor_rev
        else if (r > 0) wown = (iw[r] - tauS[r > 0 ? r - 1 : 0]) * rowR[r][grp];
        if constexpr (APARK) apark(wown, wAlo[r], wAhi[r]);
        else vv[r][2][lane] = wown;
        double pX[G], uX[G], wX[G];
        if constexpr (G <= 16) {  // the lane's own values are still in registers: gather by DPP, no LDS traffic
          xgather_dpp<G>(pown[r], xB, lane, pX);
          xgather_dpp<G>(iu[r], xB, lane, uX);
          xgather_dpp<G>(wown, xB, lane, wX);
        } else {
          lds_order();
          xgather_lds<G>(vv[r][0], lane, pX);
          xgather_lds<G>(vv[r][1], lane, uX);
          xgather_lds<G>(vv[r][2], lane, wX);
        }
        const double p = pX[0], w = wX[0];
        const double dw = rowD[r][grp] * w;
        double tau0 = 0.0, tau1 = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const double sv = (pX[i] * p) * fma(dw, wX[i], SX[i]);
          SX[i] = sv;
          if (i & 1) tau1 = fma(uX[i], sv, tau1);
          else tau0 = fma(uX[i], sv, tau0);
        }
        tauS[r] = tau0 + tau1;
        if constexpr (!FR) {
          F = p * fma(w, rowZ[r][grp], F);
          Fp[r] = F;
        }
        if constexpr (APARK) {
#pragma unroll
          for (int i = 0; i < G; ++i) apark(SX[i], sAlo[r][i], sAhi[r][i]);
        } else {
          double *sfr = sfL[r];
          sfr[lane] = SX[0];
#pragma unroll
          for (int b = 0; (1 << b) < G; ++b) {
            if ((j & (1 << b)) == 0) {  // owner lanes of slots k in [2^b, 2^(b+1))
#pragma unroll
              for (int kk = (1 << b); kk < (2 << b); ++kk) sfr[soff[kk]] = SX[kk];
            }
          }
        }
      }
    }
    if (k == nseg - 1) {  // cotangents of the last row: pure seeds
      const double rd = rowR[cnt][grp], z = rowZ[cnt][grp];
      if constexpr (FR) {  // reverse.hpp:55-57 and step 6 of the last row (:65); z holds bd_{N-1}
        bVn = bwlast * rd;
        ban = z - gsum<G>(wlast * bwlast) * rd;
      } else {
        ban = 0.5 * rd * (z * z * rd - 1.0);
        bzn = -z * rd;
        bVn = 0.0;
        if (st0) byb[N - 1] = bzn;
      }
      if (st0) bab[N - 1] = ban;   // (the rows below leave with their segments' runs)
    }
    // entry 0 (row n_lo-1) is entry C of the next, earlier segment
    carT = rowT[0][grp]; carDZ = make_double2(rowD[0][grp], rowZ[0][grp]); carR = rowR[0][grp];
    lds_order();
    C2_TCK(2);

    // ---- phase C: fused reverse steps; the next (earlier) segment is fetched half way through ---------
#pragma unroll
    for (int r = C - 1; r >= 0; --r) {
      if (r == (BACK && C2_BACK_EARLY ? C - 1 : C / 2 - 1) || (C == 1)) {
        if (k > 0) load_segment(k - 1);
      }
      if constexpr (SC) {
        if (r < cnt) {   // the step in the scaled frame (see the head of the kernel); F holds F~_n, bF holds bF-, bVn holds bV-_n
          const int64_t n = n_lo + r;
          const double rdm = rowR[r][grp], zm = rowZ[r][grp];
          double gn, ign, dt;   // g_n, 1 / g_n of this row
          if constexpr (APARK) {
            gn = (r == C - 1) ? gtop : pown[r + 1 < C ? r + 1 : 0];
            ign = (r == C - 1) ? igtop : ipown[r + 1 < C ? r + 1 : 0];
            dt = dtv[r];
          } else {
            gn = (r == C - 1) ? gtop : vv[r + 1 < C ? r + 1 : 0][0][lane];
            ign = (r == C - 1) ? igtop : vv[r + 1 < C ? r + 1 : 0][3][lane];
            dt = rowT[r][grp] - rowT[r + 1][grp];
          }
          double uX[G], wX[G], xX[G];
          if constexpr (APARK) {
            xgather_dpp<G>(afetch(uAlo[r], uAhi[r]), xB, lane, uX);
            xgather_dpp<G>(afetch(wAlo[r], wAhi[r]), xB, lane, wX);
          } else {
            xgather_dpp<G>(vv[r][1][lane], xB, lane, uX);
            xgather_dpp<G>(vv[r][2][lane], xB, lane, wX);
          }
          const double u = uX[0], wm = wX[0];   // u-_n and w~_{n-1} of this lane
          // rows n (odd first, then even) of a pair fill its tile; blocks start at odd rows: r odd <=> n even, pair (r + 1) / 2 + 4k
          double *obu = reinterpret_cast<double *>(otile[LN ? ((r + 1) / 2) & 1 : 0][0]) + lto + ((r & 1) ? 0 : 8);
          double *obv = reinterpret_cast<double *>(otile[LN ? ((r + 1) / 2) & 1 : 0][1]) + lto + ((r & 1) ? 0 : 8);
          if constexpr (LN) {
            if (r % 2 == 0 && r + 1 < cnt) {   // the pair completed by the step before (row n + 1, even) leaves now
              lds_order();
              const int64_t P = (n + 1) / 2;
              bUl[P * 8] = otile[((r + 2) / 2) & 1][0][lane];
              bVl[P * 8] = otile[((r + 2) / 2) & 1][1][lane];
            }
          } else if constexpr (!TT) {
            if (st) bVb[n * J] = bVn * gn;
          }
          const double bVout = bVn * gn;
          const double ba_in = ban;   // ba_n
          // x- = bV- + 2 ba u- is the vector gathered on the chain; M^ -= u-^T x- + bV-^T u- = u-_i bV-_j + x-_i u-_j
          const double xv = fma(2.0 * ban, u, bVn);
          xgather_dpp<G>(xv, xB, lane, xX);
          const double bU1 = -bzn * F;            // internal.hpp:232 (F~: the factor 1 / g_n joins below)
          bF = fma(-u, bzn, bF);                  // internal.hpp:233
          const double bp_s = F * bF;             // internal.hpp:236 (F . bF = F~ . bF-)
          double xs0 = 0.0, xs1 = 0.0, bp0 = 0.0, bp1 = 0.0, q0 = 0.0, q1 = 0.0;
#pragma unroll
          for (int i = 0; i < G; ++i) {
            double m = fma(-uX[i], bVn, MX[i]);
            m = fma(-xX[i], u, m);
            MX[i] = m;
            if (i & 1) { xs1 = fma(xX[i], SX[i], xs1); bp1 = fma(SX[i], m, bp1); q1 = fma(wX[i], m, q1); }
            else { xs0 = fma(xX[i], SX[i], xs0); bp0 = fma(SX[i], m, bp0); q0 = fma(wX[i], m, q0); }
          }
          double gs8 = 0.0;   // (TT) sum_k g_nk dc_k of row n: what bx_n has on top of bt_n
          if constexpr (LN) { *obu = ign * (bU1 - (xs0 + xs1)); *obv = bVout; }   // (adjacent tiles: one ds_write2_b64)
          else if constexpr (TT) {
            // the reverse of the recipe for row n: own values and the XOR-1 partner's (the other column of the complex term)
            const double bUo = ign * (bU1 - (xs0 + xs1));          // bU_n, own column
            const double uo = u * gn;                              // U_n, own column (u is u- = U_n / g_n)
            const double vo = afetch(vAlo[r], vAhi[r]);            // V_n, own column
            const double bUp = dpp_mov<kDppXor1>(bUo), bVp = dpp_mov<kDppXor1>(bVout), up = dpp_mov<kDppXor1>(uo);
            const double vp = lt.re ? 0.0 : dpp_mov<kDppXor1>(vo);
            accA = fma(bUo, vo, fma(bUp, vp, accA));
            accB = fma(bUo, vp, fma(-bUp, vo, accB));
            const double Y = fma(-bUo, up, fma(bUp, uo, fma(-bVout, vp, bVp * vo)));   // +- cotangent of the phase dc x_n
            accD = fma(Y, rowT[r + 1][grp], accD);
            gs8 = gsum<G>(Y * wD);
            sba += ba_in;
          }
          else if (st) bUb[n * J] = ign * (bU1 - (xs0 + xs1));     // reverse.hpp:66 + internal.hpp:232
          const double bp = bp_s + (bp0 + bp1);
          bcj = fma(dt, bp, bcj);
          const double q = q0 + q1;               // (w~ M^)_j = q_j / g_{n-1}
          double f = cj * bp, Gs = wm * bF, Q = q * wm;
          gsum3<G>(f, Gs, Q);
          oBT[bq][grp][r] = TT ? carry - f + gs8 : carry - f;   // (TT: bx_n)
          carry = f;
          const double zr = zm * rdm;
          bzn = Gs - zr;
          oBY[grp][r] = bzn;
          bVn = fma(zr, bF, q);                   // bV-_{n-1}: no factor
          ban = 0.5 * rdm * (zm * zr - 1.0) - 0.5 * Q - zr * Gs;
          oBA[grp][r] = ban;                      // ba_{n-1}
          const double dwm = rowD[r][grp] * wm;   // the state of row n-1: no decay to invert
#pragma unroll
          for (int i = 0; i < G; ++i) SX[i] = fma(-dwm, wX[i], SX[i]);
          F = fma(-wm, zm, F);
        }
      } else
      if (r < cnt) {
        const int64_t n = n_lo + r;
        const double Fpn = FR ? 0.0 : (BACK ? F : Fp[BACK ? 0 : r]);
        const double rdm = rowR[r][grp], zm = rowZ[r][grp];  // FR: zm = bd_{n-1}
        const double dt = (BACK && !APARK) ? rowT[r][grp] - rowT[r + 1][grp] : dtv[r];   // (two wavefronts per SIMD: not kept)
        double bWm = 0.0;  // FR: the lane's own bW_{n-1}
        if constexpr (FR) {
          if constexpr (APARK) bWm = afetch(xAlo[r], xAhi[r]);
          else bWm = vv[r][3][lane];
        }
        double uX[G], pX[G], wX[G], bVX[G], Sf[G];
        // own value back from LDS, the rest of the group by DPP: on this chip a ds_read_b64 costs the issuing
        // wavefront ~13 cycles against ~10 for the two DPP moves (profiles/r01_ubench_instruction_costs.md)
        if constexpr (APARK) {
          xgather_dpp<G>(afetch(uAlo[r], uAhi[r]), xB, lane, uX);
          xgather_dpp<G>(pown[r], xB, lane, pX);
          xgather_dpp<G>(afetch(wAlo[r], wAhi[r]), xB, lane, wX);
        } else if constexpr (G <= 16) {
          xgather_dpp<G>(vv[r][1][lane], xB, lane, uX);
          xgather_dpp<G>(vv[r][0][lane], xB, lane, pX);
          xgather_dpp<G>(vv[r][2][lane], xB, lane, wX);
        } else {
          xgather_lds<G>(vv[r][1], lane, uX);
          xgather_lds<G>(vv[r][0], lane, pX);
          xgather_lds<G>(vv[r][2], lane, wX);
        }
        const double p = pX[0], u = uX[0], wm = wX[0];  // slot 0 of an XOR gather is the lane's own element
        double ipX[BACK ? G : 1], tau_n = 0.0;
        if constexpr (BACK) {   // S_n is the carried state; tau_n = U_n S_n as the forward pass formed it
          if constexpr (APARK) xgather_dpp<G>(ipown[r], xB, lane, ipX);
          else xgather_dpp<G>(vv[r][3][lane], xB, lane, ipX);
          double ta0 = 0.0, ta1 = 0.0;
#pragma unroll
          for (int i = 0; i < G; ++i) {
            Sf[i] = SX[i];
            if (i & 1) ta1 = fma(uX[i], Sf[i], ta1);
            else ta0 = fma(uX[i], Sf[i], ta0);
          }
          tau_n = ta0 + ta1;
        } else if constexpr (APARK) {
          tau_n = tauS[BACK ? 0 : r];
#pragma unroll
          for (int i = 0; i < G; ++i) Sf[i] = afetch(sAlo[r][i], sAhi[r][i]);
        } else {
          tau_n = tauS[BACK ? 0 : r];
          const double *sfr = sfL[r];
#pragma unroll
          for (int i = 0; i < G; ++i) Sf[i] = sfr[soff[i]];
        }
        if (st) bVb[n * J] = bVn;
        xgather_dpp<G>(bVn, xB, lane, bVX);
        // solve_lower_rev part (internal.hpp:232-245)
        double bU1 = 0.0, bp_s = 0.0;
        if constexpr (!FR) {
          bU1 = -bzn * Fpn;
          bF = fma(-u, bzn, bF);
          bp_s = Fpn * bF;
          bF *= p;
        }
        // factor_rev part (reverse.hpp:65-80).  With x = bV + 2 ba U (own lane: xv):
        //   bU2_j = -sum_i x_i S(i,j) = -(sum_i bV_i S(i,j) + 2 ba tau_j),  tau_j = sum_i U_i S(i,j)
        //   M    -= U^T y + y^T U  =  U^T x + bV^T U                         (y = bV + ba U)
        const double xv = fma(2.0 * ban, u, bVn);
        double xs0 = 0.0, xs1 = 0.0, bp0 = 0.0, bp1 = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          double m = fma(-uX[i], xv, MX[i]);
          m = fma(-bVX[i], u, m);
          MX[i] = m;
          if (i & 1) { xs1 = fma(bVX[i], Sf[i], xs1); bp1 = fma(Sf[i], m, bp1); }
          else { xs0 = fma(bVX[i], Sf[i], xs0); bp0 = fma(Sf[i], m, bp0); }
        }
        xs0 = fma(2.0 * ban, tau_n, xs0);
        if (st) bUb[n * J] = bU1 - (xs0 + xs1);
        const double bp = bp_s + (bp0 + bp1);
        bcj = fma(dt, bp, bcj);
        double q0 = 0.0, q1 = 0.0;
#pragma unroll
        for (int i = 0; i < G; ++i) {
          MX[i] *= pX[i] * p;
          if (i & 1) q1 = fma(wX[i], MX[i], q1);
          else q0 = fma(wX[i], MX[i], q0);
        }
        const double q = q0 + q1;
        double f = cj * bp, Gs = wm * (FR ? bWm : bF), Q = q * wm;  // FR: Gs = W_{n-1} . bW_{n-1}
        gsum3<G>(f, Gs, Q);
        oBT[bq][grp][r] = carry - f;
        carry = f;
        if constexpr (FR) {
          // bV_{n-1} = bW_{n-1}/d_{n-1} + w M (reverse.hpp:80); ba_{n-1} = bd_{n-1} + w bS w^T (:79) - W_{n-1}.bV_{n-1}
          // (step 6 of the next row, :65) = bd_{n-1} - Q/2 - (W_{n-1}.bW_{n-1})/d_{n-1}
          bVn = fma(bWm, rdm, q);
          ban = zm - 0.5 * Q - Gs * rdm;
          oBA[grp][r] = ban;
        } else {
          const double zr = zm * rdm;
          bzn = Gs - zr;
          oBY[grp][r] = bzn;
          bVn = fma(zr, bF, q);
          ban = 0.5 * rdm * (zm * zr - 1.0) - 0.5 * Q - zr * Gs;
          oBA[grp][r] = ban;
        }
        if constexpr (BACK) {   // the state of row n-1 (rowD[r] = d_{n-1}, zm = z_{n-1}, wX = W_{n-1})
          const double ip = ipX[0], dwm = rowD[r][grp] * wm;
#pragma unroll
          for (int i = 0; i < G; ++i) SX[i] = fma(-dwm, wX[i], (ipX[i] * ip) * Sf[i]);
          F = fma(-wm, zm, F * ip);
        }
      }
    }
    if constexpr (BACK) {
#pragma unroll
      for (int i = 0; i < G; ++i) carS[i] = SX[i];
      carF = F;
    }
    if constexpr (SC) {   // row n_lo - 1 is the row above the next (earlier) segment's last step
      if constexpr (APARK) { gtop = pown[0]; igtop = ipown[0]; }
      else { gtop = vv[0][0][lane]; igtop = vv[0][3][lane]; }
    }
    lds_order();
    C2_TCK(3);
    // flush the per-series scalar outputs, transposed: lane j <-> row n_lo - 1 + j of ba, by; row n_lo + C - 1 + j of bt
    if constexpr (HOLD) {
      // (segments of eight rows at width 8: a run is HALF a 128-byte line, and the two halves of a line written eight steps apart
      // are merged on the memory side one by one -- profiles/r06_halflines.md.  The upper half (rows 16 i + 8 ..) waits in a register
      // per stream and leaves with the lower half, back to back.  C2_LOGLIK_PAIRLINES=0: as they come.)
      if (PAD ? L.valid : true) {
        const double vA = oBA[grp][j], vT = j == 0 ? oBT[bq][grp][C - 1] : oBT[bq ^ 1][grp][j - 1];
        double vY = 0.0;
        if constexpr (!FR) vY = oBY[grp][j];
        const bool tv = n_lo + C - 1 + j < N;
        if (k & 1) {
          hA = vA; hY = vY; hAok = j < cnt;
        } else {
          if (j < cnt) bab[n_lo - 1 + j] = vA;
          if (hAok) bab[n_lo - 1 + C + j] = hA;
          if constexpr (!FR) {
            if (j < cnt) byb[n_lo - 1 + j] = vY;
            if (hAok) byb[n_lo - 1 + C + j] = hY;
          }
          hAok = false;
        }
        if ((k + 1) & 1) {
          hT = vT; hTok = tv;
        } else {
          if (tv) btb[n_lo + C - 1 + j] = vT;
          if (hTok) btb[n_lo + 2 * C - 1 + j] = hT;
          hTok = false;
        }
      }
    } else {
#pragma unroll
    for (int m = 0; m < NV; ++m) {
      const int idx = m * G + j;
      if ((G * NV == C || idx < C) && (PAD ? L.valid : true)) {
        if (idx < cnt) {
          bab[n_lo - 1 + idx] = oBA[grp][idx];
          if constexpr (!FR) byb[n_lo - 1 + idx] = oBY[grp][idx];
        }
        if (n_lo + C - 1 + idx < N) btb[n_lo + C - 1 + idx] = idx == 0 ? oBT[bq][grp][C - 1] : oBT[bq ^ 1][grp][idx - 1];
      }
    }
    }
    bq ^= 1;
    lds_order();
    C2_TCK(4);
  }
#ifdef C2_REV_TIMING
  if (lane == 0) {
    for (int i = 1; i < 5; ++i) atomicAdd(&c2_dbg[i], tacc[i]);
    atomicAdd(&c2_dbg[0], 1ull);
  }
#endif
  if (nseg == 0) {  // N == 1
    const double rd0 = 1.0 / carDZ.x, cz = carDZ.y;
    if constexpr (FR) {
      bVn = bwlast * rd0;
      ban = cz - gsum<G>(wlast * bwlast) * rd0;
    } else {
      ban = 0.5 * rd0 * (cz * cz * rd0 - 1.0);
      bzn = -cz * rd0;
      if (st0) byb[0] = bzn;
    }
  }
  if constexpr (TT) {   // row 0: bU_0 = 0; bV_0 and ba_0 are complete (carT = t_0)
    double u0, v0;
    lt.uv(carT, u0, v0);
    const double bV0 = bVn * gtop, bVp = dpp_mov<kDppXor1>(bV0);
    const double vp = lt.re ? 0.0 : dpp_mov<kDppXor1>(v0);
    const double Y = fma(-bV0, vp, bVp * v0);
    accD = fma(Y, carT, accD);
    carry += gsum<G>(Y * wD);
    sba += ban;
  }
  // row 0 (reverse.hpp:83-84) and the rows of bt the first segment left behind (rows 1 .. C - 1)
  if (st0) bab[0] = ban;
#pragma unroll
  for (int m = 0; m < NV; ++m) {
    const int idx = m * G + j;
    if ((G * NV == C || idx < C) && idx < N && (PAD ? L.valid : true)) btb[idx] = idx == 0 ? carry : oBT[bq ^ 1][grp][idx - 1];
  }
  if constexpr (HOLD) {   // the upper half of bt's first line (rows 8 .. 15) waited for these
    if (hTok && (PAD ? L.valid : true)) btb[C + j] = hT;
  }
  if constexpr (LN) {   // row 0 completes pair 0 (row 1 is in the tile of even pairs since the last step)
    double *obu = reinterpret_cast<double *>(otile[0][0]) + lto, *obv = reinterpret_cast<double *>(otile[0][1]) + lto;
    *obv = bVn * gtop;
    *obu = 0.0;
    lds_order();
    bUl[0] = otile[0][0][lane];
    bVl[0] = otile[0][1][lane];
    bc[L.b * J + j] = bcj;
  } else if constexpr (TT) {
    const int JC = T8.Jc, JR = G - 2 * JC;
    const double bcp = dpp_mov<kDppXor1>(bcj);
    if (lt.re) { G8.bar[L.b * JR + j] = sba + accA; G8.bcr[L.b * JR + j] = bcj; }
    else if (!lt.odd) {
      const int64_t o = L.b * JC + ((j - JR) >> 1);
      G8.bac[o] = sba + accA; G8.bbc[o] = accB; G8.bdc[o] = accD; G8.bcc[o] = bcj + bcp;
    }
  } else if (st) { bVb[0] = SC ? bVn * gtop : bVn; bUb[0] = 0.0; bc[L.b * J + j] = bcj; }
}

}  // namespace c2

using namespace c2;

#ifndef C2_CKPT_C
#define C2_CKPT_C 8   // checkpoint interval for G <= 8
#endif
#ifndef C2_FWD_R
#define C2_FWD_R C2_CKPT_C   // rows per block of the forward kernels (scalar tiles); multiple of the checkpoint interval
#endif
#ifndef C2_FWD_R8
// ... of the eight-lane instances: sixteen, i.e. t, a, y arrive and (d, z) leave as whole 128-byte lines (with eight the second half
// of a line is fetched again eight rows later: forward pass 5.50 -> 5.11 GB at 8192 series = its algorithmic bytes; the ring of
// U, V rows stays eight long, C2_FWD_RING)
#define C2_FWD_R8 16
#endif

static int64_t simd_count();
namespace {
inline int launch_ok() { return hipGetLastError() == hipSuccess ? C2_OK : C2_ERR_HIP; }
// the forward kernel's p / U gathers by DPP instead of through LDS (C2_FWD_DPP_GATHERS=1)
// (measured, round 4: slower even then -- 1024 series 3.76 -> 4.11 ms for the gradient pair, forward-only 1.27 -> 1.30 at 4096
// series: the 28 permutes join the step's dependency chain; kept as a switch, off)
inline bool fwd_dpp_gathers(unsigned waves) {
  (void)waves;
  return opt::has(opt::k_fwd_dpp_gathers) && opt::ival(opt::k_fwd_dpp_gathers) != 0;
}

template <int MODE>
int launch_fwd(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
               const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
               double *ckpt, int64_t nseg, double *Wst, double2 *DZst, hipStream_t s,
               const unsigned long long *gate = nullptr, const unsigned long long *segguard = nullptr, bool occ2 = false) {
  const int G_ = group_size(J);
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
  if (MODE == 1 && occ2 && J == 8) {   // two wavefronts per SIMD (the pair of the backward-recursion sweep, see loglik_grad_group)
    hipLaunchKernelGGL((k_loglik_fwd<8, C2_FWD_R8, C2_CKPT_C, (MODE == 1 ? 1 : 0), false, 2>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c,
                       c_bs, a, U, V, y, ll, flag, ckpt, nseg, Wst, DZst, gate, segguard);
    return launch_ok();
  }
  const bool dg = G_ <= 8 && fwd_dpp_gathers(grid.x);
  // rows of U, V as whole 128-byte lines (k_loglik_fwd<..., LN>): J = 8, an even number of rows, 16-byte aligned arrays
  if (J == 8 && N >= 2 && N % 2 == 0 && ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(V)) & 15) == 0 && !dg &&
      opt::has(opt::k_loglik_lines) && (opt::ival(opt::k_loglik_lines) == 1 || opt::ival(opt::k_loglik_lines) == 2)) {
    hipLaunchKernelGGL((k_loglik_fwd<8, 8, C2_CKPT_C, MODE, false, C2_FWD_OCC, false, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t,
                       t_bs, c, c_bs, a, U, V, y, ll, flag, ckpt, nseg, Wst, DZst, gate, segguard);
    return launch_ok();
  }
#define C2_FWD(G, R, C)                                                                                          \
  do {                                                                                                           \
    if (dg && G <= 8) {                                                                                          \
      if (J == G)                                                                                                \
        hipLaunchKernelGGL((k_loglik_fwd<(G <= 8 ? G : 8), R, C, MODE, false, C2_FWD_OCC, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, \
                           t_bs, c, c_bs, a, U, V, y, ll, flag, ckpt, nseg, Wst, DZst, gate, segguard);         \
      else                                                                                                       \
        hipLaunchKernelGGL((k_loglik_fwd<(G <= 8 ? G : 8), R, C, MODE, true, C2_FWD_OCC, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t,  \
                           t_bs, c, c_bs, a, U, V, y, ll, flag, ckpt, nseg, Wst, DZst, gate, segguard);         \
    } else if (J == G)                                                                                           \
      hipLaunchKernelGGL((k_loglik_fwd<G, R, C, MODE, false>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs,   \
                         c, c_bs, a, U, V, y, ll, flag, ckpt, nseg, Wst, DZst, gate, segguard);                 \
    else                                                                                                         \
      hipLaunchKernelGGL((k_loglik_fwd<G, R, C, MODE, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs,    \
                         c, c_bs, a, U, V, y, ll, flag, ckpt, nseg, Wst, DZst, gate, segguard);                 \
  } while (0)
  switch (G_) {
    case 1: C2_FWD(1, C2_FWD_R, C2_CKPT_C); break;
    case 2: C2_FWD(2, C2_FWD_R, C2_CKPT_C); break;
    case 4: C2_FWD(4, C2_FWD_R, C2_CKPT_C); break;
    case 8: C2_FWD(8, C2_FWD_R8, C2_CKPT_C); break;
    case 16: C2_FWD(16, 8, 4); break;
    default: C2_FWD(32, 4, 2); break;
  }
#undef C2_FWD
  return launch_ok();
}
}  // namespace

// Two-columns-per-lane variant for J == 8 (c2_loglik4.hip).
extern "C" int c2_internal_loglik4(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                   const double *a, const double *U, const double *V, const double *y, double *ll,
                                   int32_t *flag, c2_stream_t stream);
// Four lanes per series, gradient pair in the scaled frame (c2_loglik_q4.hip, J == 8)
extern "C" size_t c2_internal_loglik_q4_record_doubles(int64_t B, int64_t N);
extern "C" int c2_internal_loglik_q4_grad(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                          const double *a, const double *U, const double *V, const double *y, double *ll,
                                          double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                          int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream);

// Which lane mapping serves (B, J)?  Four lanes per series (two columns per lane, J = 8) carry 16 series per wavefront, so
// they need twice the batch to fill the chip.  Forward-only (c2_loglik4.hip): 14-15 % faster than the eight-lane kernel from
// 16384 series up, equal at 8192 (profiles/r01_lanes4.md).  Gradient pair (c2_loglik_q4.hip, scaled frame): the batches that
// give the eight-lane pair two wavefronts per SIMD (profiles/r05_four_lanes.md).  C2_LANES=4 / C2_LANES=8 force one or the other.
static bool use_lanes4(int64_t B, int64_t J, bool grad) {
  if (J != 8) return false;
  const int forced = opt::has(opt::k_lanes) ? (int)opt::ival(opt::k_lanes) : 0;
  if (forced == 4) return true;
  if (forced == 8 || forced == 1 || forced == 2) return false;
  if (grad) return B >= opt::ival(opt::k_lanes4_min_batch_grad) && B <= opt::ival(opt::k_lanes4_max_batch_grad);
  return B >= opt::ival(opt::k_lanes4_min_batch);
}

// One lane per series (c2_loglik_t.hip, J == 8): 64 series per wavefront, so it takes 64 x 1024 series to put one
// wavefront on every SIMD.  Measured on MI355X at N = 4096 (profiles/r02_lane_mappings.md): the forward-only kernel
// wins from 24576 series up (3.3 vs 4.1 ms; 6.7 vs 10.4 ms at 65536), the gradient pair from 24576 up as well (15.7 vs
// 16.0 ms; 17.2 vs 21.6 ms at 32768; 28.2 vs 41.8 ms at 65536).  C2_LANES=1 forces it.
// the same kernels compiled per width (c2_loglik_t.hip with C2T_J = 8, 4, 2; 6 as 8 with two empty columns)
#define C2_DECL_T(J_)                                                                                                  \
  extern "C" int c2_internal_loglik_t##J_(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c,       \
                                          int64_t c_bs, const double *a, const double *U, const double *V,           \
                                          const double *y, double *ll, int32_t *flag, c2_stream_t stream);           \
  extern "C" size_t c2_internal_loglik_t_record_doubles##J_(int64_t B, int64_t N);                                   \
  extern "C" int c2_internal_loglik_t_grad##J_(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, \
                                               int64_t c_bs, const double *a, const double *U, const double *V,      \
                                               const double *y, double *ll, double *bt, double *bc, double *ba,      \
                                               double *bU, double *bV, double *by, int32_t *flag, double *rec,       \
                                               unsigned long long *guard, c2_stream_t stream);
C2_DECL_T(8)
C2_DECL_T(6)   // rows of 6 in memory, computed as rows of 8 (c2_loglik_t6.hip)
C2_DECL_T(4)
C2_DECL_T(2)
#undef C2_DECL_T
// guard words in front of the records of the one-lane path: the head + one per wavefront, rounded to 16 bytes
static size_t lanes1_gate_words(int64_t B) { return (size_t)((kGateHeadWords + (B + kWave - 1) / kWave + 1) & ~(int64_t)1); }
// Two lanes per series (c2_loglik_k2.hip, J == 8): 32 series per wavefront -- the batches that
// give the one-lane mapping half a chip.  C2_LANES=2 forces it.
extern "C" int c2_internal_loglik_k2_ok(int64_t B, int64_t N, int64_t J);
extern "C" int c2_internal_loglik_k2(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                     const double *a, const double *U, const double *V, const double *y, double *ll,
                                     int32_t *flag, c2_stream_t stream);
extern "C" size_t c2_internal_loglik_k2_record_doubles(int64_t B, int64_t N);
extern "C" int c2_internal_loglik_k2_grad(int64_t B, int64_t N, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                          const double *a, const double *U, const double *V, const double *y, double *ll,
                                          double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                          int32_t *flag, double *rec, unsigned long long *guard, c2_stream_t stream);
static bool use_lanes2(int64_t B, int64_t N, int64_t J, bool grad) {
  if (!c2_internal_loglik_k2_ok(B, N, J)) return false;
  const int forced = opt::has(opt::k_lanes) ? (int)opt::ival(opt::k_lanes) : 0;
  if (forced) return forced == 2;
  // N = 4096 on MI355X (profiles/r04_two_lanes.md): a wavefront of the pair walks a row in 0.66 of the one-lane time, so it
  // wins wherever the one-lane mapping leaves SIMDs empty and this one does not need a second round of wavefronts
  return B <= opt::ival(opt::k_lanes2_max_batch) &&
         B >= opt::ival(grad ? opt::k_lanes2_min_batch_grad : opt::k_lanes2_min_batch_fwd);
}
static size_t lanes1_record_doubles(int64_t B, int64_t N, int64_t J) {
  return J == 8 ? c2_internal_loglik_t_record_doubles8(B, N)
       : J == 6 ? c2_internal_loglik_t_record_doubles6(B, N)
       : J == 4 ? c2_internal_loglik_t_record_doubles4(B, N) : c2_internal_loglik_t_record_doubles2(B, N);
}
// Time-parallel forward pass (c2_timepar.hip; widths 4 and 2): batches too small to fill the chip row by row -- below the
// one-lane threshold -- of series long enough to cut into chunks.  C2_TIMEPAR=1 forces it, =0 disables it.
extern "C" size_t c2_internal_timepar_doubles(int64_t B, int64_t N, int64_t J);
extern "C" size_t c2_internal_loglik_timepar_doubles(int64_t B, int64_t N, int64_t J);
extern "C" int c2_internal_loglik_timepar(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                          int64_t c_bs, const double *a, const double *U, const double *V,
                                          const double *y, double *ll, int32_t *flag, double *work,
                                          unsigned long long *guard, c2_stream_t stream);
// The single-rhs SOLVES parallel along time (affine chunk maps): a handful of series (B * J <= 512: every driver.* call) is
// pure latency row by row -- 0.15-0.24 us per row -- against a fixed 40-130 us: level at ~256 / ~420 / ~900 rows (J = 2 / 4 / 8).
static int64_t timepar_min_rows(int64_t B, int64_t J) {
  if (B * J > 512) return opt::ival(opt::k_timepar_min_rows);
  return J == 2 ? 384 : (J == 4 ? 704 : 1024);
}
static bool use_timepar(int64_t B, int64_t N, int64_t J, bool loglik = false) {
  if (J != 4 && J != 2 && !(J == 8 && loglik)) return false;
  if (B > 65535) return false;   // (the chunk kernels put the batch on gridDim.y -- also under C2_TIMEPAR=1)
  if (opt::has(opt::k_timepar)) return opt::ival(opt::k_timepar) != 0 && N >= 2;
  if (opt::has(opt::k_lanes) && opt::ival(opt::k_lanes) != 0) return false;   // a forced lane mapping means the row-by-row kernels
  if (J == 8) {   // forward log-likelihood: chunk elements in lanes, combined by workgroups (k_e8_tree).  Measured (ms,
    // tools/onepass_grid.py 8): 0.10 + 4.7e-6 per combination (B x chunks) + 9e-8 per row against 0.29e-3 per row of the
    // longest series row by row (at least 0.085): 1 x 4096 0.29 vs 1.19 (Newton iterations: 0.56), 1024 x 4096 0.78 vs 1.21,
    // 2048 x 4096 1.52 vs 1.21, 256 x 1000 0.18 vs 0.30, 1024 x 1000 0.49 vs 0.30, 1 x 256 0.097 vs 0.089
    // round 6 (chunk pass with the element spread over a group's lanes, tools/scan8_grid.py): 0.05 + 4.5e-6 per chunk + 5e-8 per
    // row: 1 x 256 0.056 vs 0.078, 256 x 512 0.088 vs 0.150, 512 x 1024 0.225 vs 0.293, 1024 x 1024 0.441 vs 0.295,
    // 1024 x 2048 0.510 vs 0.887, 1024 x 4096 0.649 vs 1.162, 2048 x 4096 1.26 vs 1.17
    int R = 16;
    while (R < 64 && B * ((N + R - 1) / R) > 65536) R *= 2;   // (chunk_rows8 of c2_timepar.hip)
    const double el = 0.05 + 4.5e-6 * (double)(B * ((N + R - 1) / R)) + 5e-8 * (double)B * (double)N;
    const double rows8 = 0.29e-3 * (double)N > 0.078 ? 0.29e-3 * (double)N : 0.078;
    return el * (double)opt::ival(opt::k_timepar_elements_bias) < rows8 * 100.0;
  }
  // Chunk elements (c2_timepar.hip): a wavefront per 4096 rows of a series, 64 chunks of R = 16 / 32 / 64 rows in lock step
  // (chunk_rows), the chip takes 1024 wavefronts a ROUND.  Measured (tools/onepass_grid.py, us; width 2: 0.6 of it):
  //   log-likelihood, one pass:            20 + rounds x (18 + 0.275 R + 12 N / 1024)
  //   factor (states + scan, writing pass): 30 + rounds x (30 + 0.5 R + 30 N / 1024)
  //   row by row: 0.217 (log-likelihood) / 0.25 (factor) us per row, at least 38 / 41 us, whatever the batch up to 8192
  //   series (more beyond: 0.27 / 0.33 / 0.46 us at 12288 / 16384 / 20480 series of width 4)
  const double rounds = (double)((B * ((N + 4095) / 4096) + 1023) / 1024), n1k = (double)(N < 4096 ? N : 4096) / 1024.0;
  const double R = N <= 1024 ? 16.0 : (N <= 2048 ? 32.0 : 64.0);
  double tp = loglik ? 20.0 + rounds * (18.0 + 0.275 * R + 12.0 * n1k) : 30.0 + rounds * (30.0 + 0.5 * R + 30.0 * n1k);
  if (J == 2) tp *= 0.6;
  double rows = (loglik ? 0.217 : 0.25) * (double)N * (1.0 + (B > 8192 ? (double)(B - 8192) / 10240.0 : 0.0));
  rows = rows > (loglik ? 38.0 : 41.0) ? rows : (loglik ? 38.0 : 41.0);
  return tp * (double)opt::ival(opt::k_timepar_elements_bias) < rows * 100.0;
}
// the same decision for the single-rhs solves (affine maps: width 8 as well)
extern "C" int c2_internal_use_timepar_solve(int64_t B, int64_t N, int64_t J) {
  if (J != 8 && J != 4 && J != 2) return 0;
  if (B > 65535) return 0;       // (batch on gridDim.y)
  if (opt::has(opt::k_timepar)) return opt::ival(opt::k_timepar) != 0 && N >= 2;
  return N >= timepar_min_rows(B, J) && B * J <= opt::ival(opt::k_timepar_max_batch_x_width);
}
// Time-parallel GRADIENT (c2_timepar_grad.hip; widths 1 .. 8): small batches of long series.
// C2_TIMEPAR_GRAD=1 forces it, =0 disables it.
// compiled for chunks of 64, 32 and 16 rows; a handful of series (at most 4096 chunks of 64 rows) takes the shorter ones
// -- 16 rows up to 4096 rows per series (one series of 4096 rows 0.92 -> 0.73 ms, 16 x 4096 0.96 -> 0.79 ms at J = 8), 32
// rows beyond: chains of more than 256 chunks cost accuracy first (9000 rows: 1.3e-11 of the largest gradient entry against
// 4e-12) and then time (one series of 1e5 rows 1.42 against 1.25 ms):
// with 32 rows against 64: one series of 1e5 rows 1.68 -> 1.25 ms, 64 x 4096 1.42 -> 1.16 ms (J = 8); beyond 4096 chunks the
// longer chains cost more: 1e6 rows 3.8 vs 5.0 ms, 32 x 50000 2.4 vs 3.1 ms  (C2_TPG_ROWS=16|32|64 overrides)
#define C2_DECL_TPG(R_)                                                                                                    \
  extern "C" size_t c2_internal_timepar_grad_doubles##R_(int64_t B, int64_t N, int64_t J);                               \
  extern "C" int c2_internal_loglik_grad_timepar##R_(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,     \
                                                     const double *c, int64_t c_bs, const double *a, const double *U,    \
                                                     const double *V, const double *y, double *ll, double *bt,           \
                                                     double *bc, double *ba, double *bU, double *bV, double *by,         \
                                                     int32_t *flag, double *work, c2_stream_t stream);                   \
  extern "C" size_t c2_internal_factor_iter_doubles##R_(int64_t B, int64_t N, int64_t J);                                \
  extern "C" int c2_internal_factor_iter##R_(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,             \
                                             const double *c, int64_t c_bs, const double *a, const double *U,            \
                                             const double *V, double *d, double *W, int32_t *flag, double *work,         \
                                             const unsigned long long **last_word, c2_stream_t stream);                  \
  extern "C" size_t c2_internal_loglik_wide_doubles##R_(int64_t B, int64_t N, int64_t J);                                \
  extern "C" int c2_internal_loglik_wide##R_(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,             \
                                             const double *c, int64_t c_bs, const double *a, const double *U,            \
                                             const double *V, const double *y, double *ll, int32_t *flag, double *work,  \
                                             c2_stream_t stream);                                                        \
  extern "C" size_t c2_internal_factor_rev_timepar_doubles##R_(int64_t B, int64_t N, int64_t J);                         \
  extern "C" int c2_internal_factor_rev_timepar##R_(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,      \
                                                    const double *c, int64_t c_bs, const double *U, const double *V,     \
                                                    const double *d, const double *W, const double *bd, const double *bW, \
                                                    double *bt, double *bc, double *ba, double *bU, double *bV,          \
                                                    double *work, c2_stream_t stream);                                   \
  extern "C" size_t c2_internal_s_rows_doubles##R_(int64_t B, int64_t N, int64_t J);                                     \
  extern "C" int c2_internal_s_rows##R_(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, \
                                        int64_t c_bs, const double *d, const double *W, const int32_t *flag, double *Sw, \
                                        double *scratch, c2_stream_t stream);
C2_DECL_TPG(64)
C2_DECL_TPG(32)
C2_DECL_TPG(16)
#undef C2_DECL_TPG
extern "C" int c2_internal_tpg_short_chunks(int64_t B, int64_t N);
// 0: chunks of 64 rows, 1: of 32, 2: of 16
extern "C" int c2_internal_tpg_short_chunks(int64_t B, int64_t N) {
  if (opt::has(opt::k_tpg_rows)) return opt::ival(opt::k_tpg_rows) == 16 ? 2 : (opt::ival(opt::k_tpg_rows) == 32 ? 1 : 0);
  const int64_t k64 = B * ((N + 63) / 64);
  if (k64 > 4096) return 0;
  return N <= opt::ival(opt::k_tpg_rows16_max_rows) ? 2 : 1;   // the shortest chunks while the chains stay short
}
#define C2_TPG_PICK(stem) (c2_internal_tpg_short_chunks(B, N) == 2 ? stem##16 : (c2_internal_tpg_short_chunks(B, N) == 1 ? stem##32 : stem##64))
// ... of the log-likelihood GRADIENT: between 4096 and 9216 chunks of 64 rows, chunks of 32 (round 6, tools/tpg_rows_probe.py, N = 4096, J = 8,
// ms with chunks of 16 / 32 / 64 rows: 80 series 0.94 / 0.75 / 0.89, 128 series 1.13 / 0.89 / 1.02, 160 series 1.37 / 1.10 / 1.09, 192 series
// 1.56 / 1.21 / 1.15 -- the sweeps' work does not depend on the chunk length, the chains over the chunks do, and 64-row chunks fill a
// quarter of the chip at 128 series)
static int tpg_short_chunks_grad(int64_t B, int64_t N) {
  const int sh = c2_internal_tpg_short_chunks(B, N);
  if (sh != 0 || opt::has(opt::k_tpg_rows)) return sh;
  return B * ((N + 63) / 64) <= 9216 ? 1 : 0;
}
#define C2_TPG_PICK_G(stem) (tpg_short_chunks_grad(B, N) == 2 ? stem##16 : (tpg_short_chunks_grad(B, N) == 1 ? stem##32 : stem##64))
static size_t c2_internal_timepar_grad_doubles(int64_t B, int64_t N, int64_t J) {
  return C2_TPG_PICK_G(c2_internal_timepar_grad_doubles)(B, N, J);
}
static int c2_internal_loglik_grad_timepar(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                           int64_t c_bs, const double *a, const double *U, const double *V,
                                           const double *y, double *ll, double *bt, double *bc, double *ba, double *bU,
                                           double *bV, double *by, int32_t *flag, double *work, c2_stream_t stream) {
  return C2_TPG_PICK_G(c2_internal_loglik_grad_timepar)(
      B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, stream);
}
static size_t c2_internal_factor_iter_doubles(int64_t B, int64_t N, int64_t J) {
  return C2_TPG_PICK(c2_internal_factor_iter_doubles)(B, N, J);
}
static int c2_internal_factor_iter(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                   int64_t c_bs, const double *a, const double *U, const double *V, double *d, double *W,
                                   int32_t *flag, double *work, const unsigned long long **last_word, c2_stream_t stream) {
  return C2_TPG_PICK(c2_internal_factor_iter)(
      B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, work, last_word, stream);
}
static size_t c2_internal_loglik_wide_doubles(int64_t B, int64_t N, int64_t J) {
  return C2_TPG_PICK(c2_internal_loglik_wide_doubles)(B, N, J);
}
static int c2_internal_loglik_wide(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                   int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                   double *ll, int32_t *flag, double *work, c2_stream_t stream) {
  return C2_TPG_PICK(c2_internal_loglik_wide)(B, N, J, t, t_bs, c, c_bs, a, U, V,
                                                                                         y, ll, flag, work, stream);
}
constexpr size_t kTimeparVerifyWords = 8;   // = kVerifyWords of c2_timepar_grad.hip: the head of its workspace layout
// Diagnostics (tools/, tests): with a sink set, the verification words of the last time-parallel call are copied there
// (8 doubles; stream-ordered).  Never set in production.
static double *g_debug_sink = nullptr;
extern "C" void c2_internal_set_debug_sink(double *device_ptr) { g_debug_sink = device_ptr; }
extern "C" double *c2_internal_get_debug_sink() { return g_debug_sink; }
// C2_VERIFY_FALLBACK=0 (diagnostics only): leave the result of the time-parallel form in place whatever its words say
static bool verify_fallback_enabled() {
  return !(opt::has(opt::k_verify_fallback) && opt::ival(opt::k_verify_fallback) == 0);
}
static void c2_internal_debug_capture(const double *words, const double *, hipStream_t s) {
  if (g_debug_sink) (void)hipMemcpyAsync(g_debug_sink, words, kTimeparVerifyWords * sizeof(double), hipMemcpyDeviceToDevice, s);
}
static bool use_timepar_grad(int64_t B, int64_t N, int64_t J) {
  if (J < 1 || J > 8) return false;
  if (opt::has(opt::k_timepar_grad)) return opt::ival(opt::k_timepar_grad) != 0 && N >= 2;
  if (opt::has(opt::k_lanes) && opt::ival(opt::k_lanes) != 0) return false;   // a forced lane mapping means the row-by-row kernels
  // Measured (tools/timepar_grad_time.py): one series of 1e5 rows 103 -> 6.0 ms at J = 8, 61 -> 1.6 ms at J = 2; 32 series
  // of 50000 rows at J = 6 57.6 -> 4.6 ms; 256 x 4096 at J = 8 4.3 -> 1.9 ms.  The chunks are walked one per lane with
  // strided rows, which stops paying once they fill the chip several times over (1024 x 4096: 3.0 vs 3.6 ms at J = 4).
  // One series draws level at ~400 rows (J = 2), ~600 (J = 4, 6), ~800 (J = 8): 0.48 -> 0.25 ms at 768 rows, J = 2.
  // (those with chunks of 64 rows; with the 16-row chunks a handful of series takes -- at most 4096 chunks of 64 rows --
  // one series draws level below 200 rows: 256 rows 0.17 -> 0.13 ms at J = 2, 0.30 -> 0.22 at J = 6; 512 rows 0.54 -> 0.33
  // at J = 8, the same for 64 series)
  const bool handful = B * ((N + 63) / 64) <= 4096;
  const int64_t min_rows = handful ? opt::ival(opt::k_timepar_grad_min_rows_handful)
                                   : (J <= 2 ? 512 : (J >= 7 ? opt::ival(opt::k_timepar_grad_min_rows) : 768));
  // (widths up to 4 would keep winning a little further -- 768 x 4096 at J = 4 2.97 -> 1.46 ms, 1024 x 4096 2.98 -> 2.44 ms
  // -- but not by enough to move the limit)
  if (!(N >= min_rows && B * ((N + 63) / 64) <= opt::ival(opt::k_timepar_grad_max_chunks))) return false;
  if (J >= 7 && !handful) {
    // widths 7, 8 beyond a handful of series (tools/scan8_grid.py, round 6; ms parallel along time / row by row): 256 x 1024
    // 0.64 / 0.81, 512 x 1024 1.04 / 0.81, 512 x 2048 1.28 / 1.61, 1024 x 2048 2.58 / 1.62, 512 x 4096 2.53 / 3.21,
    // 1024 x 4096 6.0 / 3.23: 0.40 ms + 6.5e-5 per 64-row chunk against 0.78 us per row
    return 0.40 + 6.5e-5 * (double)(B * ((N + 63) / 64)) < 0.78e-3 * (double)N;
  }
  return true;
}
// widths 1 .. 8: `factor` by Newton iterations on the chunk start states (c2_timepar_grad.hip), the row-by-row kernel
// gated behind; the forward-only log-likelihood composed from it
static bool use_factor_iter(int64_t B, int64_t N, int64_t J) {
  if (J < 1 || J > 8) return false;
  const bool e = opt::has(opt::k_factor_iter);   // 1 forces it (every length; widths 4, 2: long series only), 0 disables it
  if (e && opt::ival(opt::k_factor_iter) == 0) return false;
  // widths 4 and 2 have the exact chunk-start states of c2_timepar.hip (scanned chunk elements): Newton only when forced
  if (J == 2 || J == 4) return e && N >= (J == 4 ? 32768 : 131072) && B * ((N + 63) / 64) <= 32768;
  if (e) return N >= 2;
  if (opt::has(opt::k_lanes) && opt::ival(opt::k_lanes) != 0) return false;
  // five Newton iterations of ~0.15 ms (N = 4096) against 0.3 us per row walked one by one: 0.77 vs 1.22 ms at 4096 rows,
  // 3.9 vs 29.5 ms at 1e5; 2.1 vs 19.9 ms for 32 series of 50000; with 64k chunks in flight (1024 x 4096) a pass no longer has a SIMD per wavefront: 2.3 vs 1.2 ms
  return N >= 2048 && B * ((N + 63) / 64) <= 32768;
}
static bool use_lanes1(int64_t B, int64_t J, bool grad) {
  if (J != 8 && J != 6 && J != 4 && J != 2) return false;
  const int forced = opt::has(opt::k_lanes) ? (int)opt::ival(opt::k_lanes) : 0;
  if (forced == 1) return true;
  if (forced == 4 || forced == 8 || forced == 2) return false;
  // width 6 (rows of 48 bytes: no aligned 128-byte runs) draws level later: gradient 18.5 vs 20.7 ms at 32768 series,
  // 17.2 vs 15.8 ms at 24576; 31.7 vs 41.6 ms at 65536 (forward 7.5 vs 10.8 ms)
  if (grad && J == 6) return B >= opt::ival(opt::k_lanes1_min_batch_grad_j6);
  return B >= opt::ival(grad ? opt::k_lanes1_min_batch_grad : opt::k_lanes1_min_batch_fwd);
}

// wide models (c2_wide.hip; C2_FAST_WIDTH < J <= C2_MAX_WIDTH): the log-likelihood from factor + solve_lower + a reduction, its
// gradient as the literal op chain of c2_fused.hip over the wide kernels, failed series filled with NaN afterwards
extern "C" size_t c2_wide_loglik_doubles(int64_t B, int64_t N, int64_t J);
extern "C" int c2_wide_loglik(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                              const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
                              double *work, c2_stream_t stream);
extern "C" int c2_wide_nan_failed(int64_t B, int64_t N, int64_t J, const int32_t *flag, double *bt, double *bc, double *ba,
                                  double *bU, double *bV, double *by, c2_stream_t stream);
extern "C" size_t c2_loglik_grad_composite_workspace_bytes(int64_t B, int64_t N, int64_t J);
extern "C" int c2_loglik_grad_composite(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                        int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                        double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                        int32_t *flag, void *work, size_t work_bytes, c2_stream_t stream);

extern "C" {

int c2_internal_loglik_grad_replay(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                   int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                   double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                   int32_t *flag, void *work, const unsigned long long *gate, c2_stream_t stream);
size_t c2_internal_loglik_grad_replay_doubles(int64_t B, int64_t N, int64_t J);

int c2_loglik(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
              const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
              c2_stream_t stream) {
  if (B < 1 || N < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  if (!t || !c || !a || !U || !V || !y || !ll || !flag) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH) {   // a wide model: factor + solve_lower + reduction on the workgroup-per-series kernels
    hipStream_t ws = (hipStream_t)stream;
    hipStreamCaptureStatus wcap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(ws, &wcap);
    if (wcap != hipStreamCaptureStatusNone) {   // its temporary is a stream-ordered allocation: not inside a graph capture
      c2_internal_set_error("c2_loglik at J > 32 allocates a temporary and cannot be captured in a HIP graph; c2_loglik_grad takes a caller workspace");
      return C2_ERR_UNSUPPORTED;
    }
    void *tmp = nullptr;
    if (c2::temp_alloc(&tmp, c2_wide_loglik_doubles(B, N, J) * sizeof(double), ws) != hipSuccess) return C2_ERR_HIP;
    int rc = c2_wide_loglik(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, (double *)tmp, stream);
    if (hipFreeAsync(tmp, ws) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
    return rc;
  }
  if (use_lanes2(B, N, J, false)) return c2_internal_loglik_k2(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, stream);
  if (use_lanes1(B, J, false)) {
    if (J == 8) return c2_internal_loglik_t8(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, stream);
    if (J == 6) return c2_internal_loglik_t6(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, stream);
    if (J == 4) return c2_internal_loglik_t4(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, stream);
    return c2_internal_loglik_t2(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, stream);
  }
  if (use_lanes4(B, J, false)) return c2_internal_loglik4(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, flag, stream);
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &capturing);   // (its temporary is a stream-ordered allocation: kept out of graph captures)
  // widths 4 and 2 in one pass (chunk elements combined in a tree: c2_timepar.hip) whatever the length; the Newton
  // iterations below are for the other widths
  const bool onepass = (J == 8 || J == 4 || J == 2) && use_timepar(B, N, J, true);
  if (capturing == hipStreamCaptureStatusNone && !onepass && use_factor_iter(B, N, J)) {
    // d, W by Newton iterations on the chunk start states, z by the chunk-map solve, a reduction
    const size_t nd = c2_internal_loglik_wide_doubles(B, N, J);
    void *tmp = nullptr;
    if (nd > 0 && c2::temp_alloc(&tmp, nd * sizeof(double), s) == hipSuccess) {
      int rc = c2_internal_loglik_wide(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, (double *)tmp, stream);
      if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
      return rc;
    }
    (void)hipGetLastError();
  }
  if (capturing == hipStreamCaptureStatusNone && use_timepar(B, N, J, true)) {
    // Small batch of long series: parallel along time (c2_timepar.hip), verified on the device; the ordinary kernel
    // below runs behind the verification word and does nothing unless it failed.  Scratch is a stream-ordered
    // temporary; without it (allocation refused) the ordinary kernel runs alone.
    const size_t nd = c2_internal_loglik_timepar_doubles(B, N, J);
    void *tmp = nullptr;
    if (nd > 0 && c2::temp_alloc(&tmp, (nd + 2) * sizeof(double), s) == hipSuccess) {
      unsigned long long *guard = (unsigned long long *)tmp;
      int rc = c2_internal_loglik_timepar(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, (double *)tmp + 2, guard, stream);
      if (rc == C2_OK)
        rc = launch_fwd<0>(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, nullptr, 0, nullptr, nullptr, s, guard + 1);
      if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
      return rc;
    }
    (void)hipGetLastError();
  }
  return launch_fwd<0>(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, nullptr, 0, nullptr, nullptr, s);
}

// Checkpoint interval per group size (must match launch_fwd / launch_rev below).
static inline int ckpt_interval(int G_) { return G_ <= 8 ? C2_CKPT_C : (G_ == 16 ? 4 : 2); }

// Workspace layout: [checkpoints: waves*nseg*CkptRec<G>::DOUBLES] [(d,z) pairs B*N*2]  (doubles)
// `back`: the form whose reverse sweep runs the recursion backward (k_loglik_rev<..., BACK>): one more checkpoint (the
// state after the last row), the W rows, one stability word per wavefront
struct GradWs {
  size_t ck, w, dz, guard, total;
};
static inline GradWs grad_ws(int64_t B, int64_t N, int64_t J, bool back = false) {
  const int G_ = group_size(J), C_ = ckpt_interval(G_);
  const int64_t nseg = (N - 1 + C_ - 1) / C_;
  GradWs g;
  const size_t waves = ((size_t)B * G_ + kWave - 1) / kWave;       // checkpoints are wave-blocked
  g.ck = waves * (size_t)(nseg + (back ? 1 : 0)) * ((size_t)kWave + (size_t)(G_ - 1) * (kWave / 2) + 2 * kWave);  // CkptRec<G>::DOUBLES
  g.ck = (g.ck + 1) & ~(size_t)1;  // keep the following arrays 16-byte aligned
  g.w = back ? waves * (size_t)N * kWave : 0;  // W rows: recorded (lane-major per wavefront), or replayed
  g.dz = (size_t)B * N * 2;
  g.guard = back ? 2 * waves : 0;   // k_anchor_spans
  g.total = g.ck + g.w + g.dz + g.guard;
  return g;
}
// The backward-recursion form serves the group mappings up to eight lanes.  Sixteen lanes (C = 4) measured on the bench's
// recipe at N = 4096: J = 16 every wavefront beyond the guard (the replay sweep answers: 8.0 ms either way), J = 12 re-anchored at
// every segment 8.9 -> 13.7 ms, J = 10 8.9 -> 7.3 -- not a gain one can count on; left on the replay.  C2_LOGLIK_BACK=0 keeps the replay (A/B runs).
static int64_t simd_count() {
  static int64_t n = 0;
  if (n == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    n = 4 * (int64_t)cus;
  }
  return n;
}
static bool occ2_enabled() { return !(opt::has(opt::k_loglik_back_occ2) && opt::ival(opt::k_loglik_back_occ2) == 0); }
static bool use_back(int64_t N, int64_t J) {
  if (opt::has(opt::k_loglik_back) && opt::ival(opt::k_loglik_back) == 0) return false;
  return group_size(J) <= 8 && N >= 2;
}

// core::factor without the S workspace (interface.hpp:37-48) on the fused forward kernel: d and W straight
// into the caller's arrays (d == a and W == V allowed: every row is read blocks ahead of the row being written).
extern "C" int c2_internal_factor_timepar(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                          int64_t c_bs, const double *a, const double *U, const double *V, double *d,
                                          double *W, int32_t *flag, double *work, unsigned long long *guard,
                                          c2_stream_t stream);
// allow_timepar: 0 row by row only; 1 the dispatch's choice; 2 what the time-parallel gradient builds on: from 2048 rows the
// scanned chunk elements at widths 4 / 2 (start states exact to rounding) and the Newton iterations at the other widths
// (1 .. 8), the row-by-row kernel below that
// `scratch` (nullable; c2_internal_factor_scratch_doubles): caller-provided room for the time-parallel forms -- without it
// they use a stream-ordered temporary and stay out of graph captures.
size_t c2_internal_factor_scratch_doubles(int64_t B, int64_t N, int64_t J) {
  const size_t n1 = c2_internal_factor_iter_doubles(B, N, J), n2 = c2_internal_timepar_doubles(B, N, J) + 2;
  return ((n1 > n2 ? n1 : n2) + 1) & ~(size_t)1;
}
int c2_internal_factor_fused_ws(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                int64_t c_bs, const double *a, const double *U, const double *V, double *d, double *W,
                                int32_t *flag, int allow_timepar, double *scratch, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &capturing);
  const bool may_alloc = capturing == hipStreamCaptureStatusNone;
  auto room = [&](size_t nd, void **tmp) {   // the caller's scratch, else a stream-ordered temporary
    if (scratch) { *tmp = scratch; return true; }
    return may_alloc && c2::temp_alloc(tmp, nd * sizeof(double), s) == hipSuccess;
  };
  auto release = [&](void *tmp, int rc) {
    if (!scratch && hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
    return rc;
  };
  // allow_timepar == 2 (the time-parallel gradient builds on d, W): the time-parallel forms from 2048 rows (Newton forced:
  // every length), the row-by-row kernel below that (0.15 us per row against several launches)
  const bool forced_iter = opt::has(opt::k_factor_iter) && opt::ival(opt::k_factor_iter) != 0;
  const bool scan_widths = (J == 2 || J == 4) && !forced_iter;   // chunk-start states by scanned elements (c2_timepar.hip)
  const bool long_enough = N >= 2048 && B * ((N + 63) / 64) <= 32768;
  bool newton = scan_widths ? false
                : allow_timepar == 2
                    ? (J >= 1 && J <= 8 && N >= 2 && (forced_iter || long_enough) &&
                       !(opt::has(opt::k_factor_iter) && opt::ival(opt::k_factor_iter) == 0))
                    : use_factor_iter(B, N, J);
  if (J == 8 && !opt::has(opt::k_factor_iter) && !(opt::has(opt::k_lanes) && opt::ival(opt::k_lanes) != 0) &&
      !(opt::has(opt::k_factor_scan8) && opt::ival(opt::k_factor_scan8) == 0) && B <= 65535) {
    // width 8, round 6: the chunk start states come from the scanned chunk elements, not from Newton iterations -- 0.10 ms +
    // 1.7e-5 per 64 rows against 0.30 us per row walked one by one (tools/scan8_grid.py, ms scan / rows: 1 x 384 0.115 /
    // 0.119, 1 x 1024 0.164 / 0.308, 1 x 4096 0.192 / 1.213, 256 x 512 0.140 / 0.158, 512 x 1024 0.252 / 0.309, 512 x 4096
    // 0.590 / 1.216, 1024 x 1024 0.403 / 0.310, 1024 x 2048 0.652 / 0.618, 1024 x 4096 1.198 / 1.223)
    const double scan_ms = 0.10 + 1.7e-5 * (double)(B * ((N + 63) / 64));
    const double rows_ms = 0.30e-3 * (double)N > 0.08 ? 0.30e-3 * (double)N : 0.08;
    newton = scan_ms < rows_ms;
  }
  if (allow_timepar && d != a && W != V && newton) {
    const size_t nd = c2_internal_factor_iter_doubles(B, N, J);
    void *tmp = nullptr;
    if (nd > 0 && room(nd, &tmp)) {
      const unsigned long long *last = nullptr;
      int rc = c2_internal_factor_iter(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, (double *)tmp, &last, stream);
      if (rc == C2_OK && c2_internal_get_debug_sink())   // diagnostics: the iterations' / the scan's words (tools/e8_words.py)
        (void)hipMemcpyAsync(c2_internal_get_debug_sink() + 8, tmp, 30 * sizeof(double), hipMemcpyDeviceToDevice, s);
      if (rc == C2_OK)
        rc = launch_fwd<2>(B, N, J, t, t_bs, c, c_bs, a, U, V, a, nullptr, flag, nullptr, 0, W,
                           reinterpret_cast<double2 *>(d), s, last);
      return release(tmp, rc);
    }
    (void)hipGetLastError();
  }
  // small batch of long series, out of place: parallel along time, verified, the row-by-row kernel gated behind it
  // (in place -- d == a or W == V -- stays row by row: the fallback would read what the time-parallel pass overwrote)
  if (allow_timepar && d != a && W != V && scan_widths && use_timepar(B, N, J)) {
    const size_t nd = c2_internal_timepar_doubles(B, N, J);
    void *tmp = nullptr;
    if (nd > 0 && room(nd + 2, &tmp)) {
      unsigned long long *guard = (unsigned long long *)tmp;
      int rc = c2_internal_factor_timepar(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, (double *)tmp + 2, guard, stream);
      if (rc == C2_OK)
        rc = launch_fwd<2>(B, N, J, t, t_bs, c, c_bs, a, U, V, a, nullptr, flag, nullptr, 0, W,
                           reinterpret_cast<double2 *>(d), s, guard);
      return release(tmp, rc);
    }
    (void)hipGetLastError();
  }
  return launch_fwd<2>(B, N, J, t, t_bs, c, c_bs, a, U, V, /*y (unused: any readable (B,N) array)*/ a, nullptr, flag,
                       nullptr, 0, W, reinterpret_cast<double2 *>(d), (hipStream_t)stream);
}
int c2_internal_factor_fused(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                             int64_t c_bs, const double *a, const double *U, const double *V, double *d, double *W,
                             int32_t *flag, int allow_timepar, c2_stream_t stream) {
  return c2_internal_factor_fused_ws(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, allow_timepar, nullptr, stream);
}
// shortest series the time-parallel factor_rev / factor with S take (C2_DROPIN_LONG_ROWS overrides)
// tools/bench_ops.py, J = 8, ms row by row -> parallel along time: factor_rev 1 x 512 0.38 -> 0.15, 64 x 1024 0.75 -> 0.19,
// 256 x 2048 1.50 -> 0.70, 512 x 4096 3.00 -> 1.59; factor + S 1 x 1024 0.74 -> 0.40, 64 x 2048 1.22 -> 0.63, 512 x 4096
// 2.43 -> 1.69 (1 x 512: 0.37 either way)
static int64_t drop_in_long_rows() {
  const int64_t v = opt::ival(opt::k_dropin_long_rows);
  return v >= 128 ? v : 512;
}
static bool drop_in_long_shape(int64_t B, int64_t N) {
  const int64_t k64 = B * ((N + 63) / 64);
  return N >= drop_in_long_rows() && k64 <= (N >= 2048 ? 32768 : 4096);   // (short series: measured up to 256 x 1024)
}
// factor_rev on a small batch of long series (widths 1 .. 8): the reverse pass of the time-parallel gradient with the
// adjoints of d, W handed in (c2_timepar_grad.hip, run_factor_rev); the S workspace is not read -- the states are replayed
// from d, W.  One series of 1e5 rows, J = 8: 73 ms row by row.  C2_TIMEPAR_GRAD=0 disables it.
extern "C" int c2_internal_factor_rev_replay(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                             int64_t c_bs, const double *U, const double *d, const double *W,
                                             const double *S, const double *bd, const double *bW, double *bt, double *bc,
                                             double *ba, double *bU, double *bV, const unsigned long long *gate,
                                             c2_stream_t stream);
extern "C" int c2_internal_factor_rev_long(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                           int64_t c_bs, const double *U, const double *d, const double *W, const double *S,
                                           const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                                           double *bV, c2_stream_t stream) {
  if (J < 1 || J > 8 || !drop_in_long_shape(B, N)) return C2_ERR_UNSUPPORTED;
  if (opt::has(opt::k_timepar_grad) && opt::ival(opt::k_timepar_grad) == 0) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &capturing);
  if (capturing != hipStreamCaptureStatusNone) return C2_ERR_UNSUPPORTED;   // a stream-ordered temporary below
  const size_t nd = C2_TPG_PICK(c2_internal_factor_rev_timepar_doubles)(B, N, J);
  void *tmp = nullptr;
  if (nd == 0 || c2::temp_alloc(&tmp, nd * sizeof(double), s) != hipSuccess) {
    (void)hipGetLastError();
    return C2_ERR_UNSUPPORTED;
  }
  int rc = C2_TPG_PICK(c2_internal_factor_rev_timepar)(B, N, J, t, t_bs, c, c_bs, U, U, d, W, bd, bW, bt, bc, ba, bU, bV,
                                                       (double *)tmp, stream);
  // verified on the device (word 0 of the scratch, c2_timepar_grad.hip): should the chunk boundaries disagree, the
  // segment-replay kernel -- behind that word, empty otherwise -- recomputes the batch from the caller's S rows
  if (rc == C2_OK && verify_fallback_enabled())
    rc = c2_internal_factor_rev_replay(B, N, J, t, t_bs, c, c_bs, U, d, W, S, bd, bW, bt, bc, ba, bU, bV,
                                       (const unsigned long long *)tmp, stream);
  c2_internal_debug_capture((const double *)tmp, nullptr, s);
  if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
  return rc;
}
// factor WITH the S workspace of the drop-in on a small batch of long series (widths 1 .. 8): d, W by the Newton iterations
// on the chunk start states, then the S rows -- a linear recurrence once d, W are known -- by chunks (k_s_rows of
// c2_timepar_grad.hip).  One series of 1e5 rows, J = 8: 71 ms row by row.  C2_ERR_UNSUPPORTED: not this shape.
extern "C" int c2_internal_factor_states_timepar(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs,
                                                 const double *c, int64_t c_bs, const double *a, const double *U,
                                                 const double *V, double *d, double *W, double *S, int32_t *flag,
                                                 c2_stream_t stream) {
  if (J < 1 || J > 8 || !drop_in_long_shape(B, N) || d == a || W == V) return C2_ERR_UNSUPPORTED;
  if (opt::has(opt::k_factor_iter) && opt::ival(opt::k_factor_iter) == 0) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &capturing);
  if (capturing != hipStreamCaptureStatusNone) return C2_ERR_UNSUPPORTED;   // stream-ordered temporaries below
  const size_t nd = C2_TPG_PICK(c2_internal_s_rows_doubles)(B, N, J);
  void *tmp = nullptr;
  if (nd == 0 || c2::temp_alloc(&tmp, nd * sizeof(double), s) != hipSuccess) {
    (void)hipGetLastError();
    return C2_ERR_UNSUPPORTED;
  }
  int rc = c2_internal_factor_fused_ws(B, N, J, t, t_bs, c, c_bs, a, U, V, d, W, flag, /*Newton iterations*/ 2, nullptr, stream);
  if (rc == C2_OK)
    rc = C2_TPG_PICK(c2_internal_s_rows)(B, N, J, t, t_bs, c, c_bs, d, W, flag, S, (double *)tmp, stream);
  if (hipFreeAsync(tmp, s) != hipSuccess && rc == C2_OK) rc = C2_ERR_HIP;
  return rc;
}


size_t c2_internal_loglik_grad_replay_doubles(int64_t B, int64_t N, int64_t J) { return grad_ws(B, N, J).total; }

// k_anchor_spans for the other lane mappings (c2_loglik_q4.hip): one workgroup per wavefront of `spw` series
int c2_internal_anchor_spans(int64_t B, int64_t N, int64_t J, int C, int spw, const double *t, int64_t t_bs, const double *c,
                             int64_t c_bs, unsigned long long *words, c2_stream_t stream) {
  const dim3 grid((unsigned)((B + spw - 1) / spw));
  hipLaunchKernelGGL(k_anchor_spans, grid, dim3(256), 0, (hipStream_t)stream, B, N, (int)J, C, spw, t, t_bs, c, c_bs, words);
  return launch_ok();
}

size_t c2_loglik_grad_workspace_bytes(int64_t B, int64_t N, int64_t J) {
  if (B < 1 || N < 1 || J < 1 || J > C2_MAX_WIDTH) return 0;
  if (J > C2_FAST_WIDTH) return c2_loglik_grad_composite_workspace_bytes(B, N, J);   // (d, W, S, z, F, seeds: the op chain)
  size_t n = grad_ws(B, N, J, use_back(N, J)).total;
  if (use_lanes4(B, J, true)) {   // [guard words] [records of the four-lane pair | workspace of the replay fallback]
    const size_t r = c2_internal_loglik_q4_record_doubles(B, N), f = grad_ws(B, N, J).total;
    const size_t n4 = lanes1_gate_words(B) + (r > f ? r : f);
    n = n4 > n ? n4 : n;   // (arrays off a 16-byte boundary take the eight-lane pair on the same workspace)
  }
  if (use_lanes1(B, J, true)) {  // [guard words: head + one per wavefront] [records of the one-lane path | workspace of the replay fallback]
    const size_t r = lanes1_record_doubles(B, N, J), f = grad_ws(B, N, J).total;
    n = lanes1_gate_words(B) + (r > f ? r : f);
  }
  if (use_lanes2(B, N, J, true)) {  // the same layout with the records of the two-lane path
    const size_t f = grad_ws(B, N, J).total, r = c2_internal_loglik_k2_record_doubles(B, N);
    n = lanes1_gate_words(B) + (r > f ? r : f);
  }
  if (use_timepar_grad(B, N, J)) {   // [verification words | its scratch, or the workspace of the gated row-by-row pair]
    const size_t r = c2_internal_timepar_grad_doubles(B, N, J), f = grad_ws(B, N, J).total + kTimeparVerifyWords;
    n = r > f ? r : f;
  }
  return n * sizeof(double);
}

static int loglik_grad_impl(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                            const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                            double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                            size_t work_bytes, bool allow_timepar, c2_stream_t stream);
static int loglik_grad_group(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                             int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                             double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                             int32_t *flag, void *work, const unsigned long long *gate, bool back, c2_stream_t stream);
int c2_loglik_grad(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                   const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                   double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                   size_t work_bytes, c2_stream_t stream) {
  return loglik_grad_impl(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, work_bytes, true,
                          stream);
}
// the same, never parallel along time (the interleaved 2-D construction: runs of identical times, conditioning 1e3-1e4 --
// the row-by-row kernels hold every element to 1e-12 of the largest there, the chunk maps to 3e-12)
int c2_internal_loglik_grad_rows(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                 int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                 double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                 int32_t *flag, void *work, size_t work_bytes, c2_stream_t stream) {
  return loglik_grad_impl(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, work_bytes, false,
                          stream);
}
static int loglik_grad_impl(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                            const double *a, const double *U, const double *V, const double *y, double *ll, double *bt,
                            double *bc, double *ba, double *bU, double *bV, double *by, int32_t *flag, void *work,
                            size_t work_bytes, bool allow_timepar, c2_stream_t stream) {
  if (B < 1 || N < 1 || J < 1) return C2_ERR_INVALID;
  if (J > C2_MAX_WIDTH) return C2_ERR_UNSUPPORTED;
  if (!t || !c || !a || !U || !V || !y || !ll || !bt || !bc || !ba || !bU || !bV || !by || !flag || !work)
    return C2_ERR_INVALID;
  if (work_bytes < c2_loglik_grad_workspace_bytes(B, N, J)) return C2_ERR_INVALID;
  if (J > C2_FAST_WIDTH) {
    if (int e = c2_loglik_grad_composite(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work,
                                         work_bytes, stream))
      return e;
    return c2_wide_nan_failed(B, N, J, flag, bt, bc, ba, bU, bV, by, stream);
  }
  if (allow_timepar && use_timepar_grad(B, N, J)) {   // small batch of long series: parallel along time
    if (int e = c2_internal_loglik_grad_timepar(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag,
                                                (double *)work, stream))
      return e;
    // ... verified on the device: word 0 of the workspace holds the worst disagreement at the chunk boundaries in units of
    // half its tolerance (c2_timepar_grad.hip, k_verify_combine).  Beyond it the row-by-row pair below -- launched behind
    // the word, empty otherwise -- recomputes the batch (the same outputs; its workspace overlays the scratch).
    c2_internal_debug_capture((const double *)work, nullptr, (hipStream_t)stream);
    if (!verify_fallback_enabled()) return C2_OK;
    return c2_internal_loglik_grad_replay(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag,
                                          (double *)work + kTimeparVerifyWords, (const unsigned long long *)work, stream);
  }
  const unsigned long long *gate = nullptr;
  // (the four-lane pair moves U, V, bU, bV, bc and its records as double2: the C ABI promises 8-byte alignment only, so
  // arrays off a 16-byte boundary -- a view starting at an odd element -- take the eight-lane pair instead)
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(V) | reinterpret_cast<uintptr_t>(bU) |
                           reinterpret_cast<uintptr_t>(bV) | reinterpret_cast<uintptr_t>(bc) | reinterpret_cast<uintptr_t>(work)) & 15) == 0;
  if (use_lanes4(B, J, true) && aligned16) {
    // Four lanes per series, scaled frame.  Groups of 64 series whose anchor intervals are beyond the guard (gaps in time)
    // are left to the replay pair below, gated per group by the words k_q4_gate leaves (decided on the device).
    unsigned long long *guard = (unsigned long long *)work;
    work = (double *)work + lanes1_gate_words(B);
    if (int e = c2_internal_loglik_q4_grad(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag,
                                           (double *)work, guard, stream))
      return e;
    gate = gate_per_wave(guard + kGateHeadWords);
  } else if (use_lanes2(B, N, J, true)) {
    // Two lanes per series: as below, the two wavefronts of a group of 64 series raising its guard word together
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *guard = (unsigned long long *)work;
    if (hipMemsetAsync(guard, 0, 8 * lanes1_gate_words(B), s) != hipSuccess) return C2_ERR_HIP;
    work = (double *)work + lanes1_gate_words(B);
    if (int e = c2_internal_loglik_k2_grad(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag,
                                           (double *)work, guard, stream))
      return e;
    gate = gate_per_wave(guard + kGateHeadWords);
  } else if (use_lanes1(B, J, true)) {
    hipStream_t s = (hipStream_t)stream;
    // One lane per series: forward with records, then the backward-recursion reverse sweep.  The forward pass leaves
    // the stability measure of every WAVEFRONT in its own guard word; where it exceeds kBackwardGuard that wavefront's
    // reverse sweep returns at once and the replay pair below -- gated per group of 64 series by the same words --
    // produces the gradients of those series (same outputs; its workspace overlays the records, which nobody reads any
    // more by then; decided on the device).
    unsigned long long *guard = (unsigned long long *)work;
    if (hipMemsetAsync(guard, 0, 8 * kGateHeadWords, s) != hipSuccess) return C2_ERR_HIP;
    work = (double *)work + lanes1_gate_words(B);
    auto one_lane = J == 8 ? c2_internal_loglik_t_grad8
                  : J == 6 ? c2_internal_loglik_t_grad6
                  : J == 4 ? c2_internal_loglik_t_grad4 : c2_internal_loglik_t_grad2;
    if (int e = one_lane(B, N, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, (double *)work, guard,
                         stream))
      return e;
    gate = gate_per_wave(guard + kGateHeadWords);   // per group of 64 series: only the wavefronts that fell back are replayed
  }
  if (gate == nullptr && use_back(N, J))   // the primary path of this batch: reverse sweep by the backward recursion
    return loglik_grad_group(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, nullptr, true,
                             stream);
  return c2_internal_loglik_grad_replay(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, gate,
                                        stream);
}

// The checkpoint / replay pair (lanes of a series share its state), every kernel behind `gate` (gate_closed; nullptr: run).
// `work`: grad_ws(B, N, J).total doubles.
int c2_internal_loglik_grad_replay(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                   int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                                   double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                                   int32_t *flag, void *work, const unsigned long long *gate, c2_stream_t stream) {
  return loglik_grad_group(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, bt, bc, ba, bU, bV, by, flag, work, gate, false, stream);
}
// `back`: forward pass with W records, reverse sweep by the backward recursion, the replay sweep behind it for the
// wavefronts whose segments it cannot invert (`work`: grad_ws(B, N, J, true).total doubles).
static int loglik_grad_group(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                             int64_t c_bs, const double *a, const double *U, const double *V, const double *y,
                             double *ll, double *bt, double *bc, double *ba, double *bU, double *bV, double *by,
                             int32_t *flag, void *work, const unsigned long long *gate, bool back, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int G_ = group_size(J), C_ = ckpt_interval(G_);
  const int64_t nseg = (N - 1 + C_ - 1) / C_;
  const GradWs ws = grad_ws(B, N, J, back);
  double *ckpt = (double *)work;
  double *Wrec = back ? ckpt + ws.ck : nullptr;
  double2 *DZst = reinterpret_cast<double2 *>(ckpt + ws.ck + ws.w);
  unsigned long long *segg = back ? reinterpret_cast<unsigned long long *>(ckpt + ws.ck + ws.w + ws.dz) : nullptr;
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
  if (back) {   // what the backward recursion of every wavefront would have to invert
    hipLaunchKernelGGL(k_anchor_spans, grid, dim3(256), 0, s, B, N, (int)J, C_, kWave / G_, t, t_bs, c, c_bs, segg);
    if (int e = launch_ok()) return e;
  }
  // More wavefronts than SIMDs (8192 < B <= 16384 at J = 8; the two-lane pair takes over beyond): both kernels of the
  // backward form as instances that fit two wavefronts per SIMD, so the second half of the batch runs next to the first
  // instead of behind it
  const bool occ2 = back && J == 8 && occ2_enabled() && (int64_t)grid.x > simd_count();
  if (int e = launch_fwd<1>(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, ckpt, nseg, Wrec, DZst, s, gate, segg, occ2)) return e;
  // the sweep in the scaled frame (k_loglik_rev<..., SC>: no decay factors in the step); C2_LOGLIK_SCALED=0: the plain form (A/B runs)
  const bool sc = !(opt::has(opt::k_loglik_scaled) && opt::ival(opt::k_loglik_scaled) == 0);
#define C2_REVB_ARGS grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U, (const double *)Wrec, (const double2 *)DZst, \
                     (const double *)ckpt, nseg, (const int32_t *)flag, bt, bc, ba, bU, bV, by, nullptr, nullptr, nullptr, gate, \
                     (const unsigned long long *)segg
  // ... with the rows of U, bU, bV as whole 128-byte lines (k_loglik_rev<..., LN>): J = 8, an even number of rows, aligned arrays
  const bool ln = sc && J == 8 && N >= 2 && N % 2 == 0 &&
                  ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(bU) | reinterpret_cast<uintptr_t>(bV)) & 15) == 0 &&
                  opt::has(opt::k_loglik_lines) && (opt::ival(opt::k_loglik_lines) == 1 || opt::ival(opt::k_loglik_lines) == 3);
  if (back && occ2) {
    if (ln) hipLaunchKernelGGL((k_loglik_rev<8, C2_CKPT_C, false, false, true, 2, true, true>), C2_REVB_ARGS);
    else if (sc) hipLaunchKernelGGL((k_loglik_rev<8, C2_CKPT_C, false, false, true, 2, true>), C2_REVB_ARGS);
    else hipLaunchKernelGGL((k_loglik_rev<8, C2_CKPT_C, false, false, true, 2>), C2_REVB_ARGS);
    if (int e = launch_ok()) return e;
  } else if (back && ln) {
    hipLaunchKernelGGL((k_loglik_rev<8, C2_CKPT_C, false, false, true, C2_REV_OCC, true, true>), C2_REVB_ARGS);
    if (int e = launch_ok()) return e;
  } else if (back) {
#define C2_REVB(G, C)                                                                                         \
  do {                                                                                                        \
    if (J == G) {                                                                                             \
      if (sc) hipLaunchKernelGGL((k_loglik_rev<G, C, false, false, true, C2_REV_OCC, true>), C2_REVB_ARGS);   \
      else hipLaunchKernelGGL((k_loglik_rev<G, C, false, false, true>), C2_REVB_ARGS);                        \
    } else {                                                                                                  \
      if (sc) hipLaunchKernelGGL((k_loglik_rev<G, C, true, false, true, C2_REV_OCC, true>), C2_REVB_ARGS);    \
      else hipLaunchKernelGGL((k_loglik_rev<G, C, true, false, true>), C2_REVB_ARGS);                         \
    }                                                                                                         \
  } while (0)
    switch (G_) {
      case 1: C2_REVB(1, C2_CKPT_C); break;
      case 2: C2_REVB(2, C2_CKPT_C); break;
      case 4: C2_REVB(4, C2_CKPT_C); break;
      default: C2_REVB(8, C2_CKPT_C); break;
    }
#undef C2_REVB
    if (int e = launch_ok()) return e;
  }
#undef C2_REVB_ARGS
  const unsigned long long *segc = segg;   // (the replay sweep: every wavefront, or those the sweep above left)
#define C2_REV(G, C)                                                                                              \
  do {                                                                                                            \
    if (J == G)                                                                                                   \
      hipLaunchKernelGGL((k_loglik_rev<G, C, false, false>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U,  \
                         V, (const double2 *)DZst, (const double *)ckpt, nseg,                  \
                         (const int32_t *)flag, bt, bc, ba, bU, bV, by, nullptr, nullptr, nullptr, gate, segc);   \
    else                                                                                                          \
      hipLaunchKernelGGL((k_loglik_rev<G, C, true, false>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, U,   \
                         V, (const double2 *)DZst, (const double *)ckpt, nseg,                  \
                         (const int32_t *)flag, bt, bc, ba, bU, bV, by, nullptr, nullptr, nullptr, gate, segc);   \
  } while (0)
  switch (G_) {
    case 1: C2_REV(1, C2_CKPT_C); break;
    case 2: C2_REV(2, C2_CKPT_C); break;
    case 4: C2_REV(4, C2_CKPT_C); break;
    case 8: C2_REV(8, C2_CKPT_C); break;
    case 16: C2_REV(16, 4); break;
    default: C2_REV(32, 2); break;
  }
#undef C2_REV
  return launch_ok();
}

// Coefficient-level log-likelihood + gradient on the group mappings (J = Jr + 2 Jc = 8, 4 or 2 lanes per series;
// k_loglik_fwd / k_loglik_rev<..., TT>): batches of at most one wavefront per SIMD.  `c`: the rates (B, J) (c2_terms.hip:
// k_rates); `work`: grad_ws(B, N, J, back) doubles; `guard`: kGateHeadWords + ceil(B / 64) words, written here -- a group of 64
// series whose word exceeds kBackwardGuard (a span the backward recursion cannot cross, unsorted times, a phase beyond the
// branch-free sincos) is left to the caller's composed chain, gated on the same words.
size_t c2_internal_loglik_g8_tt_doubles(int64_t B, int64_t N, int64_t J) { return grad_ws(B, N, J, true).total; }
int c2_internal_loglik_g8_tt_ok(int64_t B, int64_t N, int64_t J) {
  return (J == 8 || J == 4 || J == 2) && N >= 2 && B >= 1 && (B * J + kWave - 1) / kWave <= simd_count();
}
int c2_internal_loglik_g8_tt_grad(int64_t B, int64_t N, int64_t J, int64_t Jc, int coef_batched, const double *ar, const double *ac,
                                  const double *bc, const double *dc, const double *c, const double *x, int64_t x_bs,
                                  const double *diag, const double *y, double *ll, double *bar, double *bcr, double *bac,
                                  double *bbc, double *bcc, double *bdc, double *bx, double *bdiag, double *by,
                                  int32_t *flag, double *work, unsigned long long *guard, c2_stream_t stream) {
  if (Jc < 0 || 2 * Jc > J || !c2_internal_loglik_g8_tt_ok(B, N, J)) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  constexpr int C_ = C2_CKPT_C;
  const int64_t nseg = (N - 1 + C_ - 1) / C_;
  const GradWs ws = grad_ws(B, N, J, true);
  double *ckpt = work, *Wrec = ckpt + ws.ck;
  double2 *DZst = reinterpret_cast<double2 *>(ckpt + ws.ck + ws.w);
  unsigned long long *segg = reinterpret_cast<unsigned long long *>(ckpt + ws.ck + ws.w + ws.dz);
  const dim3 grid((unsigned)((B * J + kWave - 1) / kWave));
  const TermsArgs8 T{ar, ac, bc, dc, coef_batched, (int)Jc};
  const TermsGrads8 G{bar, bcr, bac, bbc, bcc, bdc};
  unsigned long long *tgate = guard + kGateHeadWords;
  hipLaunchKernelGGL(k_anchor_spans, grid, dim3(256), 0, s, B, N, (int)J, C_, kWave / (int)J, x, x_bs, c, J, segg);
  if (int e = launch_ok()) return e;
  if (hipMemsetAsync(guard, 0, 8 * kGateHeadWords, s) != hipSuccess) return C2_ERR_HIP;
  hipLaunchKernelGGL(k_tt8_gate, dim3((unsigned)((B + 63) / 64)), dim3(kWave), 0, s, B, N, (int64_t)grid.x, (int)J,
                     (const unsigned long long *)segg, T, x, x_bs, guard, tgate);
  if (int e = launch_ok()) return e;
#define C2_TTG(G_, R_)                                                                                                                \
  do {                                                                                                                                \
    hipLaunchKernelGGL((k_loglik_fwd<G_, R_, C2_CKPT_C, 1, false, 1, false, false, true>), grid, dim3(kWave), 0, s, B, N, G_, x, x_bs, \
                       c, J, diag, (const double *)nullptr, (const double *)nullptr, y, ll, flag, ckpt, nseg, Wrec, DZst,             \
                       (const unsigned long long *)nullptr, (const unsigned long long *)segg, T, (const unsigned long long *)tgate);  \
    if (int e = launch_ok()) return e;                                                                                                \
    hipLaunchKernelGGL((k_loglik_rev<G_, C2_CKPT_C, false, false, true, 1, true, false, true>), grid, dim3(kWave), 0, s, B, N, G_, x,  \
                       x_bs, c, J, (const double *)nullptr, (const double *)Wrec, (const double2 *)DZst, (const double *)ckpt, nseg,  \
                       (const int32_t *)flag, bx, (double *)nullptr, bdiag, (double *)nullptr, (double *)nullptr, by,                 \
                       (const double *)nullptr, (const double *)nullptr, (const double *)nullptr,                                     \
                       (const unsigned long long *)nullptr, (const unsigned long long *)segg, T, G, (const unsigned long long *)tgate); \
  } while (0)
  if (J == 8) C2_TTG(8, C2_FWD_R8);
  else if (J == 4) C2_TTG(4, C2_FWD_R);
  else C2_TTG(2, C2_FWD_R);
#undef C2_TTG
  return launch_ok();
}
// The forward-only form: `guard` as above (here a word is 0 or +inf: only the phases decide) ...
int c2_internal_loglik_g8_tt(int64_t B, int64_t N, int64_t J, int64_t Jc, int coef_batched, const double *ar, const double *ac,
                             const double *bc, const double *dc, const double *c, const double *x, int64_t x_bs,
                             const double *diag, const double *y, double *ll, int32_t *flag, unsigned long long *guard,
                             c2_stream_t stream) {
  if (!(J == 8 || J == 4 || J == 2) || Jc < 0 || 2 * Jc > J || B < 1 || N < 1) return C2_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((B * J + kWave - 1) / kWave));
  const TermsArgs8 T{ar, ac, bc, dc, coef_batched, (int)Jc};
  unsigned long long *tgate = guard + kGateHeadWords;
  if (hipMemsetAsync(guard, 0, 8 * kGateHeadWords, s) != hipSuccess) return C2_ERR_HIP;
  hipLaunchKernelGGL(k_tt8_gate, dim3((unsigned)((B + 63) / 64)), dim3(kWave), 0, s, B, N, (int64_t)grid.x, (int)J,
                     (const unsigned long long *)nullptr, T, x, x_bs, guard, tgate);
  if (int e = launch_ok()) return e;
#define C2_TTF(G_, R_)                                                                                                                \
  hipLaunchKernelGGL((k_loglik_fwd<G_, R_, C2_CKPT_C, 0, false, 1, false, false, true>), grid, dim3(kWave), 0, s, B, N, G_, x, x_bs, c, \
                     J, diag, (const double *)nullptr, (const double *)nullptr, y, ll, flag, (double *)nullptr, (int64_t)0,           \
                     (double *)nullptr, (double2 *)nullptr, (const unsigned long long *)nullptr, (const unsigned long long *)nullptr, \
                     T, (const unsigned long long *)tgate)
  if (J == 8) C2_TTF(8, C2_FWD_R8);
  else if (J == 4) C2_TTF(4, C2_FWD_R);
  else C2_TTF(2, C2_FWD_R);
#undef C2_TTF
  return launch_ok();
}
// ... and the matrix-level forward kernel of the same mapping for the groups it declined (`gate`: per group of 64 series, gate_per_wave)
int c2_internal_loglik_g8_gated(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c, int64_t c_bs,
                                const double *a, const double *U, const double *V, const double *y, double *ll, int32_t *flag,
                                const unsigned long long *gate, c2_stream_t stream) {
  return launch_fwd<0>(B, N, J, t, t_bs, c, c_bs, a, U, V, y, ll, flag, nullptr, 0, nullptr, nullptr, (hipStream_t)stream, gate);
}

// core::factor_rev on the segment-replay kernel (FR mode of k_loglik_rev): the caller's S workspace serves as the
// checkpoint every C rows, everything in between is replayed from d and W.
int c2_internal_factor_rev_replay(int64_t B, int64_t N, int64_t J, const double *t, int64_t t_bs, const double *c,
                                  int64_t c_bs, const double *U, const double *d, const double *W, const double *S,
                                  const double *bd, const double *bW, double *bt, double *bc, double *ba, double *bU,
                                  double *bV, const unsigned long long *gate, c2_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int G_ = group_size(J), C_ = ckpt_interval(G_);
  const int64_t nseg = (N - 1 + C_ - 1) / C_;
  const dim3 grid((unsigned)((B * G_ + kWave - 1) / kWave));
#define C2_FREV(G, C)                                                                                              \
  do {                                                                                                             \
    if (J == G)                                                                                                    \
      hipLaunchKernelGGL((k_loglik_rev<G, C, false, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs, \
                         U, W, (const double2 *)nullptr, S, nseg, (const int32_t *)nullptr, bt, bc, ba, bU, bV,    \
                         (double *)nullptr, d, bd, bW, gate);                       \
    else                                                                                                           \
      hipLaunchKernelGGL((k_loglik_rev<G, C, true, true>), grid, dim3(kWave), 0, s, B, N, (int)J, t, t_bs, c, c_bs,  \
                         U, W, (const double2 *)nullptr, S, nseg, (const int32_t *)nullptr, bt, bc, ba, bU, bV,    \
                         (double *)nullptr, d, bd, bW, gate);                       \
  } while (0)
  switch (G_) {
    case 1: C2_FREV(1, C2_CKPT_C); break;
    case 2: C2_FREV(2, C2_CKPT_C); break;
    case 4: C2_FREV(4, C2_CKPT_C); break;
    case 8: C2_FREV(8, C2_CKPT_C); break;
    case 16: C2_FREV(16, 4); break;
    default: C2_FREV(32, 2); break;
  }
#undef C2_FREV
  return launch_ok();
}

}  // extern "C"
