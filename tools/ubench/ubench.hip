// tools/ubench/ubench.hip -- instruction-cost microbenchmarks for gfx950 (one wavefront per workgroup).
// Each kernel runs ITER iterations of an unrolled block of K identical instructions (inline asm) and
// reports shader cycles (s_memtime) per instruction for: independent / dependent fp64 FMA, fp64 mul,
// 32-bit DPP moves (quad_perm, row_half_mirror), v_mov_b32, ds_read_b64, ds_write_b64, s_nop.
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench ; run: ./ubench [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP2(x) x x
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// VMEM instruction cost: PATTERN 0 = 64 lanes x 8 B contiguous (one 512-B run); 1 = eight 64-B runs 256 KB apart
// (the sub-wave-packed row access); 2 = eight 8-B words, each shared by 8 lanes (a per-series scalar).
template <int STORE, int PATTERN>
__global__ __launch_bounds__(64) void kmem(unsigned long long *out, double *buf, int iters) {
  const int l = threadIdx.x;
  size_t off;
  if (PATTERN == 0) off = (size_t)blockIdx.x * 4096 + l;
  else if (PATTERN == 1) off = ((size_t)blockIdx.x * 8 + (l >> 3)) * 32768 + (l & 7);
  else off = ((size_t)blockIdx.x * 8 + (l >> 3)) * 32768;
  double *p = buf + off;
  double acc = 0.0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (STORE) p[(it * 16 + k) * 8 % 2048] = acc + k;
      else acc += p[(it * 16 + k) * 8 % 2048];
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (l == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.678) out[0] = 0;
}

template <int STORE, int PATTERN>
void runmem(const char *name, int waves_per_simd, unsigned long long *d_out, double *buf) {
  const int blocks = getenv("UB_BLOCKS") ? atoi(getenv("UB_BLOCKS")) : 1024 * waves_per_simd, iters = 500;
  hipLaunchKernelGGL((kmem<STORE, PATTERN>), dim3(blocks), dim3(64), 0, 0, d_out, buf, 5);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((kmem<STORE, PATTERN>), dim3(blocks), dim3(64), 0, 0, d_out, buf, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  fprintf(stderr, "%-40s waves/SIMD=%d  wall %.3f ms  wall-ns per VMEM instr per wave %.1f  (per CU: %.1f ns)\n", name,
          waves_per_simd, ms, ms * 1e6 / (16.0 * iters), ms * 1e6 / (16.0 * iters) / (4 * waves_per_simd));
}

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned long long *out, int iters, double seed) {
  __shared__ double lds[1024];
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 q0, q1, q2, q3, q4, q5, q6, q7;
  q0 = q1 = q2 = q3 = q4 = q5 = q6 = q7 = (d2)(0.0);
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double m = 1.0000001, c = 1e-9;
  double b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, j0 = i0 + 4, j1 = i0 + 5, j2 = i0 + 6, j3 = i0 + 7;
  lds[threadIdx.x] = a0;
  int addr = (threadIdx.x ^ 3) * 8;
  int addr16 = (threadIdx.x ^ 3) * 16;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 8 independent fp64 FMA chains
      REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                        "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (MODE == 1) {  // one dependent fp64 FMA chain
      REP64(asm volatile("v_fma_f64 %0, %0, %1, %2\n" : "+v"(a0) : "v"(m), "v"(c));)
    } else if (MODE == 2) {  // independent fp64 MUL
      REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                        "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
    } else if (MODE == 3) {  // independent DPP quad_perm moves (4 destinations)
      REP8(asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %1, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %6 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %5, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %6, %2 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %7, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3) :);)
    } else if (MODE == 4) {  // row_half_mirror DPP moves
      REP8(asm volatile("v_mov_b32_dpp %0, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %1, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %3, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %0, %2 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %1, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %2, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %3, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) :);)
    } else if (MODE == 5) {  // plain v_mov_b32 (independent)
      REP8(asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %2, %0\n v_mov_b32 %3, %1\n"
                        "v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %2, %0\n v_mov_b32 %3, %1\n"
                        : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) :);)
    } else if (MODE == 6) {  // ds_read_b64, 8 in flight then wait
      REP8(asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:64\n ds_read_b64 %2, %8 offset:128\n ds_read_b64 %3, %8 offset:192\n"
                        "ds_read_b64 %4, %8 offset:256\n ds_read_b64 %5, %8 offset:320\n ds_read_b64 %6, %8 offset:384\n ds_read_b64 %7, %8 offset:448\n s_waitcnt lgkmcnt(0)\n"
                        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(addr) : "memory");)
    } else if (MODE == 7) {  // fp64 add with the DPP'd value: butterfly level = 2 dpp + 1 add (dependent, like gsum)
      REP8(asm volatile("v_mov_b32_dpp %2, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %3, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f64 %4, %4, %5\n"
                        "v_mov_b32_dpp %0, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %1, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f64 %5, %5, %4\n"
                        "v_mov_b32_dpp %2, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mov_b32_dpp %3, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(a0), "+v"(a1) :);)
    } else if (MODE == 8) {  // s_nop 0
      REP64(asm volatile("s_nop 0\n");)
    } else if (MODE == 10) {  // the kernels' mix: 3 fp64 FMA per ds_read_b64 (24 + 8 per body, reads drained at the end)
      REP2(asm volatile("v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n ds_read_b64 %8, %18\n"
                        "v_fma_f64 %3, %3, %16, %17\n v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n ds_read_b64 %9, %18 offset:64\n"
                        "v_fma_f64 %6, %6, %16, %17\n v_fma_f64 %7, %7, %16, %17\n v_fma_f64 %0, %0, %16, %17\n ds_read_b64 %10, %18 offset:128\n"
                        "v_fma_f64 %1, %1, %16, %17\n v_fma_f64 %2, %2, %16, %17\n v_fma_f64 %3, %3, %16, %17\n ds_read_b64 %11, %18 offset:192\n"
                        "v_fma_f64 %4, %4, %16, %17\n v_fma_f64 %5, %5, %16, %17\n v_fma_f64 %6, %6, %16, %17\n ds_read_b64 %12, %18 offset:256\n"
                        "v_fma_f64 %7, %7, %16, %17\n v_fma_f64 %0, %0, %16, %17\n v_fma_f64 %1, %1, %16, %17\n ds_read_b64 %13, %18 offset:320\n"
                        "v_fma_f64 %2, %2, %16, %17\n v_fma_f64 %3, %3, %16, %17\n v_fma_f64 %4, %4, %16, %17\n ds_read_b64 %14, %18 offset:384\n"
                        "v_fma_f64 %5, %5, %16, %17\n v_fma_f64 %6, %6, %16, %17\n v_fma_f64 %7, %7, %16, %17\n ds_read_b64 %15, %18 offset:448\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2),
                          "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                        : "v"(m), "v"(c), "v"(addr)
                        : "memory");)
    } else if (MODE == 12) {  // ds_read_b128, 8 in flight then wait
      REP8(asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:64\n ds_read_b128 %2, %8 offset:128\n ds_read_b128 %3, %8 offset:192\n"
                        "ds_read_b128 %4, %8 offset:256\n ds_read_b128 %5, %8 offset:320\n ds_read_b128 %6, %8 offset:384\n ds_read_b128 %7, %8 offset:448\n s_waitcnt lgkmcnt(0)\n"
                        : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr16) : "memory");)
    } else if (MODE == 13) {  // ds_read2st64_b64 (two 8-byte words 512 B apart), 8 in flight then wait
      REP8(asm volatile("ds_read2st64_b64 %0, %8 offset1:1\n ds_read2st64_b64 %1, %8 offset0:2 offset1:3\n ds_read2st64_b64 %2, %8 offset0:4 offset1:5\n ds_read2st64_b64 %3, %8 offset0:6 offset1:7\n"
                        "ds_read2st64_b64 %4, %8 offset1:1\n ds_read2st64_b64 %5, %8 offset0:2 offset1:3\n ds_read2st64_b64 %6, %8 offset0:4 offset1:5\n ds_read2st64_b64 %7, %8 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)\n"
                        : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr) : "memory");)
    } else if (MODE == 14) {  // ds_write_b64, 8 then wait
      REP8(asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2 offset:1024\n ds_write_b64 %8, %3 offset:1536\n"
                        "ds_write_b64 %8, %4\n ds_write_b64 %8, %5 offset:512\n ds_write_b64 %8, %6 offset:1024\n ds_write_b64 %8, %7 offset:1536\n s_waitcnt lgkmcnt(0)\n"
                        : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(addr) : "memory");)
    } else if (MODE == 11) {  // fp64 FMA with three distinct, rotating VGPR-pair operands (register-file port pressure)
      REP8(asm volatile("v_fma_f64 %0, %1, %2, %3\n v_fma_f64 %1, %2, %3, %4\n v_fma_f64 %2, %3, %4, %5\n v_fma_f64 %3, %4, %5, %6\n"
                        "v_fma_f64 %4, %5, %6, %7\n v_fma_f64 %5, %6, %7, %0\n v_fma_f64 %6, %7, %0, %1\n v_fma_f64 %7, %0, %1, %2\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :);)
    } else if (MODE == 9) {  // SALU add
      int s = it;
      REP64(asm volatile("s_add_u32 %0, %0, 1\n" : "+s"(s));)
      i0 += s;
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 0.123) out[1] = 0;
  if (q0.x + q1.x + q2.y + q3.x + q4.x + q5.y + q6.x + q7.x + addr16 == 0.123) out[2] = 0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + j0 + j1 + j2 + j3 + addr == 12345.678) out[0] = 0;
}

template <int MODE>
void run(const char *name, int waves_per_simd, unsigned long long *d_out) {
  fprintf(stderr, "start %s\n", name);
  const int blocks = getenv("UB_BLOCKS") ? atoi(getenv("UB_BLOCKS")) : 1024 * waves_per_simd, iters = getenv("UB_ITERS") ? atoi(getenv("UB_ITERS")) : 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, 10, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d_out, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += (double)v;
  avg /= blocks;
  const double n_inst = 64.0 * iters;
  fprintf(stderr, "%-34s waves/SIMD=%d  wall %.3f ms  counter/inst %.2f  wall-ns/inst/wave %.3f\n", name, waves_per_simd, ms,
         avg / n_inst, ms * 1e6 / n_inst);
}

int main(int argc, char **argv) {
  const int w = argc > 1 ? atoi(argv[1]) : 1;
  unsigned long long *d_out;
  hipMalloc(&d_out, 1024 * 8 * sizeof(unsigned long long));
  run<0>("fma_f64 independent x8", w, d_out);
  run<1>("fma_f64 dependent chain", w, d_out);
  run<2>("mul_f64 independent x8", w, d_out);
  run<3>("dpp quad_perm mov_b32", w, d_out);
  run<4>("dpp row_half_mirror mov_b32", w, d_out);
  run<5>("v_mov_b32", w, d_out);
  run<6>("ds_read_b64 x8 + wait", w, d_out);
  run<7>("gsum-like (6 dpp + 2 add)/8", w, d_out);
  run<8>("s_nop 0", w, d_out);
  run<10>("mix 3 fma_f64 : 1 ds_read_b64", w, d_out);
  run<12>("ds_read_b128 x8 + wait", w, d_out);
  run<13>("ds_read2st64_b64 x8 + wait", w, d_out);
  run<14>("ds_write_b64 x8 + wait", w, d_out);
  run<11>("fma_f64 3 distinct operands", w, d_out);
  if (getenv("UB_NOMEM")) return 0;
  double *buf;
  hipMalloc(&buf, (size_t)4096 * 8 * 32768 * sizeof(double) + (1 << 24));
  runmem<0, 0>("load dwordx2, 512 B contiguous", w, d_out, buf);
  runmem<0, 1>("load dwordx2, 8 x 64 B runs", w, d_out, buf);
  runmem<0, 2>("load dwordx2, 8 x 8 B (8 lanes each)", w, d_out, buf);
  runmem<1, 0>("store dwordx2, 512 B contiguous", w, d_out, buf);
  runmem<1, 1>("store dwordx2, 8 x 64 B runs", w, d_out, buf);
  runmem<1, 2>("store dwordx2, 8 x 8 B (8 lanes each)", w, d_out, buf);

  return 0;
}
