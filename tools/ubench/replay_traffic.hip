// replay_traffic.hip -- memory-movement skeletons of the one-lane forward / reverse pair (c2_loglik_t.hip), no arithmetic:
// does a reverse pass that REPLAYS an anchor interval forward into a per-wavefront scratch ring (re-used every interval,
// so it can live in L2 / Infinity Cache) and then sweeps it backward move its bytes faster than today's recorded pair?
//
//   A  (today):   fwd  reads  t a y U V (152 B per series-row), writes W (64) + (d, z) (16) + t (8) + checkpoints (11)
//                 rev  reads  U (64) + W (64) + (d, z) (16) + t (8) + checkpoints (11), writes bt ba by bU bV (152)
//   B  (replay):  fwd  reads  152, writes checkpoints (11)
//                 rev  per interval of C rows: reads checkpoint + t a y U V (152) forward, writes W + (d, z) (80) to the
//                      wavefront's ring; then backward: reads the ring (80) + U + t again (72), writes 152
// Access shapes follow the kernels: rows of U, V, bU, bV as 128-byte runs per series (8 series per instruction, 16 bytes
// per lane), per-series scalars as 64-byte runs (8 rows x 8 series per instruction), records lane-major (one contiguous
// KB per instruction).  One wavefront per workgroup, grid = series / 64, __launch_bounds__(64, 1) like the kernels.
//
// hipcc -O3 --offload-arch=gfx950 -o replay_traffic replay_traffic.hip ; ./replay_traffic [series] [rows] [C]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int J = 8;
constexpr int NSF = 44;   // packed S (36) + F (8)

struct Bufs {
  const double *t, *a, *y, *U, *V;
  double *bt, *ba, *by, *bU, *bV;
  double *W, *DZ, *T, *CK;     // records of A (lane-major per wavefront)
  double *ring;                // B: per wavefront C rows x 10 doubles x 64 lanes
};

// rows [n0, n0 + 4) of a (B, N, 8) array for the 64 series of the wavefront: 16 instructions of 16 bytes per lane
__device__ __forceinline__ void rows4_load(const double *X, int64_t N, int64_t b0, int64_t n0, int lane, double2 (&r)[16]) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t s = b0 + 8 * i + lane / 8;
      r[8 * h + i] = *reinterpret_cast<const double2 *>(X + (s * N + n0 + 2 * h) * J + 2 * (lane % 8));
    }
}
__device__ __forceinline__ void rows4_store(double *X, int64_t N, int64_t b0, int64_t n0, int lane, const double2 (&r)[16]) {
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t s = b0 + 8 * i + lane / 8;
      *reinterpret_cast<double2 *>(X + (s * N + n0 + 2 * h) * J + 2 * (lane % 8)) = r[8 * h + i];
    }
}
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ldnt2(const double *p) { const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p)); return make_double2(v.x, v.y); }
__device__ __forceinline__ void stnt2(double *p, double2 v) { d2v w; w.x = v.x; w.y = v.y; __builtin_nontemporal_store(w, reinterpret_cast<d2v *>(p)); }
// rows [n0, n0 + 8) of a (B, N) array: 8 instructions of 8 bytes per lane
__device__ __forceinline__ void sc8_load(const double *x, int64_t N, int64_t b0, int64_t n0, int lane, double (&r)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = x[(b0 + 8 * i + lane / 8) * N + n0 + lane % 8];
}
__device__ __forceinline__ void sc8_store(double *x, int64_t N, int64_t b0, int64_t n0, int lane, const double (&r)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) x[(b0 + 8 * i + lane / 8) * N + n0 + lane % 8] = r[i];
}
// rows [n0, n0 + 16) of a (B, N) array: 16 instructions of 8 bytes per lane, 4 series x 128 bytes each
__device__ __forceinline__ void sc16_store(double *x, int64_t N, int64_t b0, int64_t n0, int lane, const double (&r)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) x[(b0 + 4 * i + lane / 16) * N + n0 + lane % 16] = r[i];
}
__device__ __forceinline__ double fold(const double2 (&r)[16]) {
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
  return s;
}

// MODE 0: A fwd, 1: A rev, 2: B fwd, 3: B rev, 4: B rev without the ring traffic (what the HBM side alone costs),
// 5: B rev with the ring but U, t not read a second time (kept on chip: an upper bound of what that would give)
template <int MODE>
__global__ __launch_bounds__(64, 1) void k(Bufs p, int64_t B, int64_t N, int C) {
  const int lane = threadIdx.x;
  const int64_t wave = blockIdx.x, b0 = wave * 64;
  double acc = 0;
  double *Wr = p.W + wave * N * J * 64, *DZr = p.DZ + wave * N * 2 * 64, *Tr = p.T + wave * N * 64;
  double *CKr = p.CK + wave * ((N + C - 1) / C) * NSF * 64;
  double *ring = p.ring + wave * (int64_t)C * 10 * 64;
  double2 u[16], v[16];
  double st[8], sa[8], sy[8];

  if (MODE == 0 || MODE == 2) {
    for (int64_t n0 = 0; n0 < N; n0 += 8) {
      sc8_load(p.t, N, b0, n0, lane, st);
      sc8_load(p.a, N, b0, n0, lane, sa);
      sc8_load(p.y, N, b0, n0, lane, sy);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t n = n0 + 4 * q;
        rows4_load(p.U, N, b0, n, lane, u);
        rows4_load(p.V, N, b0, n, lane, v);
        acc += fold(u) + fold(v);
        if (MODE == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc)
              *reinterpret_cast<double2 *>(Wr + ((n + r) * 4 + pc) * 128 + 2 * lane) = make_double2(acc + pc, u[4 * r + pc].x);
            *reinterpret_cast<double2 *>(DZr + (n + r) * 128 + 2 * lane) = make_double2(acc, v[r].y);
            Tr[(n + r) * 64 + lane] = acc + r;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += st[i] + sa[i] + sy[i];
      if ((n0 + 8) % C == 0) {
        double *ck = CKr + (n0 / C) * NSF * 64;
        for (int e = 0; e < NSF; ++e) ck[e * 64 + lane] = acc + e;
      }
    }
  } else if (MODE == 1) {
    for (int64_t n0 = N - 8; n0 >= 0; n0 -= 8) {
      if ((n0 + 8) % C == 0) {
        const double *ck = CKr + (n0 / C) * NSF * 64;
        for (int e = 0; e < NSF; ++e) acc += ck[e * 64 + lane];
      }
#pragma unroll
      for (int q = 1; q >= 0; --q) {
        const int64_t n = n0 + 4 * q;
        rows4_load(p.U, N, b0, n, lane, u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int pc = 0; pc < 4; ++pc) v[4 * r + pc] = *reinterpret_cast<const double2 *>(Wr + ((n + r) * 4 + pc) * 128 + 2 * lane);
          const double2 dz = *reinterpret_cast<const double2 *>(DZr + (n + r) * 128 + 2 * lane);
          acc += dz.x + dz.y + Tr[(n + r) * 64 + lane];
        }
        acc += fold(u) + fold(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) { u[i].x += acc; v[i].y += acc; }
        rows4_store(p.bU, N, b0, n, lane, u);
        rows4_store(p.bV, N, b0, n, lane, v);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { st[i] = acc + i; sa[i] = acc - i; sy[i] = acc * i; }
      sc8_store(p.bt, N, b0, n0, lane, st);
      sc8_store(p.ba, N, b0, n0, lane, sa);
      sc8_store(p.by, N, b0, n0, lane, sy);
    }
  } else if (MODE == 11) {   // A rev as today (halves 8 rows apart), every OTHER stream with the non-temporal hint: do the half lines survive in L2?
    for (int64_t n0 = N - 8; n0 >= 0; n0 -= 8) {
      if ((n0 + 8) % C == 0) {
        const double *ck = CKr + (n0 / C) * NSF * 64;
        for (int e = 0; e < NSF; ++e) acc += __builtin_nontemporal_load(&ck[e * 64 + lane]);
      }
#pragma unroll
      for (int q = 1; q >= 0; --q) {
        const int64_t n = n0 + 4 * q;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < 8; ++i) u[8 * hh + i] = ldnt2(p.U + ((b0 + 8 * i + lane / 8) * N + n + 2 * hh) * J + 2 * (lane % 8));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int pc = 0; pc < 4; ++pc) v[4 * r + pc] = ldnt2(Wr + ((n + r) * 4 + pc) * 128 + 2 * lane);
          const double2 dz = ldnt2(DZr + (n + r) * 128 + 2 * lane);
          acc += dz.x + dz.y + __builtin_nontemporal_load(&Tr[(n + r) * 64 + lane]);
        }
        acc += fold(u) + fold(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) { u[i].x += acc; v[i].y += acc; }
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            stnt2(p.bU + ((b0 + 8 * i + lane / 8) * N + n + 2 * hh) * J + 2 * (lane % 8), u[8 * hh + i]);
            stnt2(p.bV + ((b0 + 8 * i + lane / 8) * N + n + 2 * hh) * J + 2 * (lane % 8), v[8 * hh + i]);
          }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { st[i] = acc + i; sa[i] = acc - i; sy[i] = acc * i; }
      sc8_store(p.bt, N, b0, n0, lane, st);
      sc8_store(p.ba, N, b0, n0, lane, sa);
      sc8_store(p.by, N, b0, n0, lane, sy);
    }
  } else if (MODE == 6 || MODE == 9 || MODE == 10) {   // A rev in blocks of 16 rows; the scalar gradients leave as 6: 16-row tiles (whole 128-byte lines), 9: the two 64-byte halves of every line back to back, 10: the halves 8 rows apart in time (today)
    double s16a[16], s16b[16], s16c[16];
    for (int64_t n0 = N - 16; n0 >= 0; n0 -= 16) {
      if ((n0 + 16) % C == 0) {
        const double *ck = CKr + (n0 / C) * NSF * 64;
        for (int e = 0; e < NSF; ++e) acc += ck[e * 64 + lane];
      }
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        const int64_t n = n0 + 4 * q;
        rows4_load(p.U, N, b0, n, lane, u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int pc = 0; pc < 4; ++pc) v[4 * r + pc] = *reinterpret_cast<const double2 *>(Wr + ((n + r) * 4 + pc) * 128 + 2 * lane);
          const double2 dz = *reinterpret_cast<const double2 *>(DZr + (n + r) * 128 + 2 * lane);
          acc += dz.x + dz.y + Tr[(n + r) * 64 + lane];
        }
        acc += fold(u) + fold(v);
#pragma unroll
        for (int i = 0; i < 16; ++i) { u[i].x += acc; v[i].y += acc; }
        rows4_store(p.bU, N, b0, n, lane, u);
        rows4_store(p.bV, N, b0, n, lane, v);
        if (MODE == 10 && (q == 2 || q == 0)) {   // rows n0 + 8 .. n0 + 15 after the upper two row tiles, rows n0 .. n0 + 7 after the lower two
#pragma unroll
          for (int i = 0; i < 8; ++i) { st[i] = acc + i; sa[i] = acc - i; sy[i] = acc * i; }
          sc8_store(p.bt, N, b0, n0 + 4 * q, lane, st);
          sc8_store(p.ba, N, b0, n0 + 4 * q, lane, sa);
          sc8_store(p.by, N, b0, n0 + 4 * q, lane, sy);
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) { s16a[i] = acc + i; s16b[i] = acc - i; s16c[i] = acc * i; }
      if (MODE == 6) {
        sc16_store(p.bt, N, b0, n0, lane, s16a);
        sc16_store(p.ba, N, b0, n0, lane, s16b);
        sc16_store(p.by, N, b0, n0, lane, s16c);
      } else if (MODE == 9) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // lower and upper half of the same lines in consecutive instructions
          const int64_t o = (b0 + 8 * i + lane / 8) * N + n0 + lane % 8;
          p.bt[o] = s16a[i]; p.bt[o + 8] = s16a[8 + i];
          p.ba[o] = s16b[i]; p.ba[o + 8] = s16b[8 + i];
          p.by[o] = s16c[i]; p.by[o + 8] = s16c[8 + i];
        }
      }
    }
  } else if (MODE == 7 || MODE == 8) {   // the same bytes as A rev (7) / A fwd (8) with EVERY stream lane-major: what the layout of the API arrays costs
    const double *in1 = p.U + wave * N * J * 64, *in2 = p.V + wave * N * J * 64;
    double *o1 = p.bU + wave * N * J * 64, *o2 = p.bV + wave * N * J * 64, *o3 = p.bt + wave * N * 64, *o4 = p.ba + wave * N * 64, *o5 = p.by + wave * N * 64;
    const double *i3 = p.t + wave * N * 64, *i4 = p.a + wave * N * 64, *i5 = p.y + wave * N * 64;
    for (int64_t n = 0; n < N; ++n) {
      double2 x[4], w2[4];
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) x[pc] = *reinterpret_cast<const double2 *>(in1 + (n * 4 + pc) * 128 + 2 * lane);
      if (MODE == 7) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) w2[pc] = *reinterpret_cast<const double2 *>(Wr + (n * 4 + pc) * 128 + 2 * lane);
        const double2 dz = *reinterpret_cast<const double2 *>(DZr + n * 128 + 2 * lane);
        acc += dz.x + dz.y + Tr[n * 64 + lane];
      } else {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) w2[pc] = *reinterpret_cast<const double2 *>(in2 + (n * 4 + pc) * 128 + 2 * lane);
        acc += i3[n * 64 + lane] + i4[n * 64 + lane] + i5[n * 64 + lane];
      }
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) acc += x[pc].x + x[pc].y + w2[pc].x + w2[pc].y;
      if (MODE == 7) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
          *reinterpret_cast<double2 *>(o1 + (n * 4 + pc) * 128 + 2 * lane) = make_double2(acc, x[pc].x);
          *reinterpret_cast<double2 *>(o2 + (n * 4 + pc) * 128 + 2 * lane) = make_double2(acc, w2[pc].y);
        }
        o3[n * 64 + lane] = acc; o4[n * 64 + lane] = acc + 1; o5[n * 64 + lane] = acc + 2;
      } else {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) *reinterpret_cast<double2 *>(Wr + (n * 4 + pc) * 128 + 2 * lane) = make_double2(acc + pc, x[pc].x);
        *reinterpret_cast<double2 *>(DZr + n * 128 + 2 * lane) = make_double2(acc, w2[0].y);
        Tr[n * 64 + lane] = acc;
      }
      if ((n + 1) % C == 0) {
        double *ck = CKr + (n / C) * NSF * 64;
        if (MODE == 8) { for (int e = 0; e < NSF; ++e) ck[e * 64 + lane] = acc + e; }
        else { for (int e = 0; e < NSF; ++e) acc += ck[e * 64 + lane]; }
      }
    }
  } else {
    for (int64_t top = N; top > 0; top -= C) {
      const int64_t base = top - C;
      {  // checkpoint at the interval's start
        const double *ck = CKr + (base / C) * NSF * 64;
        for (int e = 0; e < NSF; ++e) acc += ck[e * 64 + lane];
      }
      // replay forward
      for (int64_t n0 = base; n0 < top; n0 += 8) {
        sc8_load(p.t, N, b0, n0, lane, st);
        sc8_load(p.a, N, b0, n0, lane, sa);
        sc8_load(p.y, N, b0, n0, lane, sy);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int64_t n = n0 + 4 * q;
          rows4_load(p.U, N, b0, n, lane, u);
          rows4_load(p.V, N, b0, n, lane, v);
          acc += fold(u) + fold(v);
          if (MODE != 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double *rr = ring + (n + r - base) * 10 * 64;
#pragma unroll
              for (int pc = 0; pc < 4; ++pc) *reinterpret_cast<double2 *>(rr + pc * 128 + 2 * lane) = make_double2(acc + pc, u[4 * r + pc].x);
              *reinterpret_cast<double2 *>(rr + 4 * 128 + 2 * lane) = make_double2(acc, v[r].y);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += st[i] + sa[i] + sy[i];
      }
      // sweep backward
      for (int64_t n0 = top - 8; n0 >= base; n0 -= 8) {
        if (MODE != 5) { sc8_load(p.t, N, b0, n0, lane, st);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc += st[i]; }
#pragma unroll
        for (int q = 1; q >= 0; --q) {
          const int64_t n = n0 + 4 * q;
          if (MODE != 5) rows4_load(p.U, N, b0, n, lane, u);
          if (MODE != 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double *rr = ring + (n + r - base) * 10 * 64;
#pragma unroll
              for (int pc = 0; pc < 4; ++pc) v[4 * r + pc] = *reinterpret_cast<const double2 *>(rr + pc * 128 + 2 * lane);
              const double2 dz = *reinterpret_cast<const double2 *>(rr + 4 * 128 + 2 * lane);
              acc += dz.x + dz.y;
            }
          }
          acc += fold(u) + fold(v);
#pragma unroll
          for (int i = 0; i < 16; ++i) { u[i].x += acc; v[i].y += acc; }
          rows4_store(p.bU, N, b0, n, lane, u);
          rows4_store(p.bV, N, b0, n, lane, v);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { st[i] = acc + i; sa[i] = acc - i; sy[i] = acc * i; }
        sc8_store(p.bt, N, b0, n0, lane, st);
        sc8_store(p.ba, N, b0, n0, lane, sa);
        sc8_store(p.by, N, b0, n0, lane, sy);
      }
    }
  }
  if (acc == 12345.678) p.bt[0] = acc;
}

template <int MODE>
static float run(const Bufs &p, int64_t B, int64_t N, int C, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<MODE>, dim3(B / 64), dim3(64), 0, 0, p, B, N, C);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(B / 64), dim3(64), 0, 0, p, B, N, C);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 65536, N = argc > 2 ? atoll(argv[2]) : 4096;
  const int C = argc > 3 ? atoi(argv[3]) : 32;
  const int reps = 5;
  Bufs p;
  auto alloc = [](size_t doubles) { double *q; CHECK(hipMalloc(&q, doubles * 8)); CHECK(hipMemset(q, 0, doubles * 8)); return q; };
  p.t = alloc(B * N); p.a = alloc(B * N); p.y = alloc(B * N); p.U = alloc(B * N * J); p.V = alloc(B * N * J);
  p.bt = alloc(B * N); p.ba = alloc(B * N); p.by = alloc(B * N); p.bU = alloc(B * N * J); p.bV = alloc(B * N * J);
  p.W = alloc(B * N * J); p.DZ = alloc(B * N * 2); p.T = alloc(B * N); p.CK = alloc((B / 64) * ((N + C - 1) / C) * NSF * 64);
  p.ring = alloc((B / 64) * (size_t)C * 10 * 64);
  const double rows = (double)B * N;
  printf("series %lld rows %lld C %d   (ring %.1f KB per wavefront, %.1f MB in all)\n", (long long)B, (long long)N, C, C * 10 * 64 * 8 / 1024.0,
         (B / 64) * (double)C * 10 * 64 * 8 / 1e6);
  const double ck = 2.0 * NSF * 8 / C / 2;  // checkpoint bytes per row, one direction
  struct { const char *name; double hbm, cache; float ms; } r[12] = {
      {"A fwd (records)", 152 + 88 + ck, 0, 0},          {"A rev (records)", 152 + ck + 152, 0, 0},
      {"B fwd (checkpoints)", 152 + ck, 0, 0},           {"B rev (replay, ring)", 152 + ck + 152, 80 + 80 + 72, 0},
      {"B rev, ring traffic removed", 152 + ck + 152, 72, 0}, {"B rev, ring, U t not re-read", 152 + ck + 152, 160, 0},
      {"A rev, scalars as 128-B runs", 152 + ck + 152, 0, 0}, {"A rev, every stream lane-major", 152 + ck + 152, 0, 0},
      {"A fwd, every stream lane-major", 152 + 88 + ck, 0, 0},
      {"A rev 16-row blocks, halves back to back", 152 + ck + 152, 0, 0}, {"A rev 16-row blocks, halves 8 rows apart", 152 + ck + 152, 0, 0},
      {"A rev today, other streams non-temporal", 152 + ck + 152, 0, 0}};
  r[0].ms = run<0>(p, B, N, C, reps);
  r[1].ms = run<1>(p, B, N, C, reps);
  r[2].ms = run<2>(p, B, N, C, reps);
  r[3].ms = run<3>(p, B, N, C, reps);
  r[4].ms = run<4>(p, B, N, C, reps);
  r[5].ms = run<5>(p, B, N, C, reps);
  r[6].ms = run<6>(p, B, N, C, reps);
  r[7].ms = run<7>(p, B, N, C, reps);
  r[8].ms = run<8>(p, B, N, C, reps);
  r[9].ms = run<9>(p, B, N, C, reps);
  r[10].ms = run<10>(p, B, N, C, reps);
  r[11].ms = run<11>(p, B, N, C, reps);
  for (auto &x : r)
    printf("%-42s %7.3f ms   first-touch bytes %.1f GB (%.2f TB/s)   re-touched %.1f GB   all %.2f TB/s\n", x.name, x.ms, x.hbm * rows / 1e9,
           x.hbm * rows / 1e9 / x.ms, x.cache * rows / 1e9, (x.hbm + x.cache) * rows / 1e9 / x.ms);
  printf("A pair %.3f ms   B pair %.3f ms   (B with U, t kept on chip %.3f)\n", r[0].ms + r[1].ms, r[2].ms + r[3].ms, r[2].ms + r[5].ms);
  return 0;
}
