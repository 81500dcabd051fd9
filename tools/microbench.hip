// microbench.hip -- issue rate of the VALU instructions the scan kernels are made of, measured on
// the whole chip (every SIMD busy) so the numbers are directly comparable with k_scan's op roof.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void bench(double *out, double a, double b, float fa, float fb) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
  v2f p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
  unsigned cnt = 0;
  d4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0, m4 = m0, m5 = m0, m6 = m0, m7 = m0;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) { asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a)); }
    if (OP == 1) { asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(b)); }
    if (OP == 2) { asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(b)); }
    if (OP == 3) { asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "s"(fa)); }
    if (OP == 4) { asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "s"(fb)); }
    if (OP == 5) { asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(a)); }
    if (OP == 6) { asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(b)); }
    if (OP == 7) { unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
      asm volatile("v_cmp_lt_f64 %0, %8, %16\n v_cmp_lt_f64 %1, %9, %16\n v_cmp_lt_f64 %2, %10, %16\n v_cmp_lt_f64 %3, %11, %16\n v_cmp_lt_f64 %4, %12, %16\n v_cmp_lt_f64 %5, %13, %16\n v_cmp_lt_f64 %6, %14, %16\n v_cmp_lt_f64 %7, %15, %16" : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3), "=s"(m4), "=s"(m5), "=s"(m6), "=s"(m7) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "s"(a));
      cnt += __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3) + __builtin_popcountll(m4) + __builtin_popcountll(m5) + __builtin_popcountll(m6) + __builtin_popcountll(m7); }
    if (OP == 8) { unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
      asm volatile("v_cmp_lt_f32 %0, %8, %16\n v_cmp_lt_f32 %1, %9, %16\n v_cmp_lt_f32 %2, %10, %16\n v_cmp_lt_f32 %3, %11, %16\n v_cmp_lt_f32 %4, %12, %16\n v_cmp_lt_f32 %5, %13, %16\n v_cmp_lt_f32 %6, %14, %16\n v_cmp_lt_f32 %7, %15, %16" : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3), "=s"(m4), "=s"(m5), "=s"(m6), "=s"(m7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7), "s"(fa));
      cnt += __builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3) + __builtin_popcountll(m4) + __builtin_popcountll(m5) + __builtin_popcountll(m6) + __builtin_popcountll(m7); }
    if (OP == 10) { unsigned c0;
      asm volatile("v_cmp_lt_f64 vcc, %1, %9\n s_bcnt1_i32_b64 %0, vcc\n v_cmp_lt_f64 vcc, %2, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f64 vcc, %3, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f64 vcc, %4, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f64 vcc, %5, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f64 vcc, %6, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f64 vcc, %7, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f64 vcc, %8, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60" : "=s"(c0) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "s"(a) : "vcc", "s60", "scc");
      cnt += c0; }
    if (OP == 11) { unsigned c0;
      asm volatile("v_cmp_lt_f32 vcc, %1, %9\n s_bcnt1_i32_b64 %0, vcc\n v_cmp_lt_f32 vcc, %2, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f32 vcc, %3, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f32 vcc, %4, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f32 vcc, %5, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f32 vcc, %6, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f32 vcc, %7, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60\n v_cmp_lt_f32 vcc, %8, %9\n s_bcnt1_i32_b64 s60, vcc\n s_add_u32 %0, %0, s60" : "=s"(c0) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7), "s"(fa) : "vcc", "s60", "scc");
      cnt += c0; }
    if (OP == 12) { unsigned c0 = 0, t;
      asm volatile("v_add_co_u32 %1, vcc, %2, %2\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_add_co_u32 %1, vcc, %3, %3\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_add_co_u32 %1, vcc, %4, %4\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_add_co_u32 %1, vcc, %5, %5\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_add_co_u32 %1, vcc, %6, %6\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_add_co_u32 %1, vcc, %7, %7\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_add_co_u32 %1, vcc, %8, %8\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_add_co_u32 %1, vcc, %9, %9\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(c0), "=&v"(t) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7) : "vcc");
      cnt += c0; }
    if (OP == 13) { // 4 independent arithmetic ops between compares: is the compare cost hidden?
      unsigned long long m0, m1;
      asm volatile("v_cmp_lt_f64 %0, %2, %10\n v_add_f64 %2, %2, %10\n v_add_f64 %3, %3, %10\n v_add_f64 %4, %4, %10\n v_add_f64 %5, %5, %10\n v_cmp_lt_f64 %1, %6, %10\n v_add_f64 %6, %6, %10\n v_add_f64 %7, %7, %10\n v_add_f64 %8, %8, %10\n v_add_f64 %9, %9, %10" : "=s"(m0), "=s"(m1), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(a));
      cnt += __builtin_popcountll(m0) + __builtin_popcountll(m1); }
    if (OP == 14) { // one dependent chain: t = x_k * s ; acc += t   (8 pairs, the dense agree() pattern)
      asm volatile("v_mul_f64 %1, %2, %10\n v_add_f64 %0, %0, %1\n v_mul_f64 %1, %3, %10\n v_add_f64 %0, %0, %1\n v_mul_f64 %1, %4, %10\n v_add_f64 %0, %0, %1\n v_mul_f64 %1, %5, %10\n v_add_f64 %0, %0, %1\n v_mul_f64 %1, %6, %10\n v_add_f64 %0, %0, %1\n v_mul_f64 %1, %7, %10\n v_add_f64 %0, %0, %1\n v_mul_f64 %1, %8, %10\n v_add_f64 %0, %0, %1\n v_mul_f64 %1, %9, %10\n v_add_f64 %0, %0, %1" : "+v"(x0), "=&v"(x1) : "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(x2), "v"(x3), "s"(b)); }
    if (OP == 15) { // two interleaved chains with separate temporaries
      double t0, t1;
      asm volatile("v_mul_f64 %2, %4, %8\n v_mul_f64 %3, %5, %8\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %3\n v_mul_f64 %2, %6, %8\n v_mul_f64 %3, %7, %8\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %3\n v_mul_f64 %2, %4, %8\n v_mul_f64 %3, %5, %8\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %3\n v_mul_f64 %2, %6, %8\n v_mul_f64 %3, %7, %8\n v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %3" : "+v"(x0), "+v"(x1), "=&v"(t0), "=&v"(t1) : "v"(x2), "v"(x3), "v"(x4), "v"(x5), "s"(b)); }
    if (OP == 16) { // 8 independent fp64 MFMAs 16x16x4 (2048 flop each)
      static_assert(true, "");
      m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, m1, 0, 0, 0);
      m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x3, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, x4, m3, 0, 0, 0);
      m4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x4, x5, m4, 0, 0, 0); m5 = __builtin_amdgcn_mfma_f64_16x16x4f64(x5, x6, m5, 0, 0, 0);
      m6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x6, x7, m6, 0, 0, 0); m7 = __builtin_amdgcn_mfma_f64_16x16x4f64(x7, x0, m7, 0, 0, 0); }
    if (OP == 17) { asm volatile("v_min3_f32 %0, |%0|, |%1|, %2\n v_min3_f32 %1, |%1|, |%2|, %3\n v_min3_f32 %2, |%2|, |%3|, %4\n v_min3_f32 %3, |%3|, |%4|, %5\n v_min3_f32 %4, |%4|, |%5|, %6\n v_min3_f32 %5, |%5|, |%6|, %7\n v_min3_f32 %6, |%6|, |%7|, %0\n v_min3_f32 %7, |%7|, |%0|, %1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
    if (OP == 18) { unsigned u0 = __builtin_bit_cast(unsigned, f0), u1 = __builtin_bit_cast(unsigned, f1), u2 = __builtin_bit_cast(unsigned, f2), u3 = __builtin_bit_cast(unsigned, f3);
      asm volatile("v_min_u32 %0, %0, %1\n v_min_u32 %1, %1, %2\n v_min_u32 %2, %2, %3\n v_min_u32 %3, %3, %0\n v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %3\n v_min_u32 %2, %2, %0\n v_min_u32 %3, %3, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
      f0 = __builtin_bit_cast(float, u0); f1 = __builtin_bit_cast(float, u1); f2 = __builtin_bit_cast(float, u2); f3 = __builtin_bit_cast(float, u3); }
    if (OP == 19) {  // the level-2 body of the two-level plane scan per packed pair of observations (cells.h: cells_survivors):
      // 3 v_pk_fma_f32 (measure), v_min3 (candidate test), 2 compares into SGPR pairs (inlier ballots), 2 v_sub + 2 v_min_u32
      // (band test): 10 vector instructions; twice per iteration = 20 (run<> prices 8 per iteration: scaled in main)
      unsigned long long b0, b1, b2, b3; unsigned dm = 0xFFFFFFFFu; float t0, t1;
      asm volatile("v_pk_fma_f32 %6, %7, %12, %6\n v_pk_fma_f32 %6, %8, %12, %6\n v_pk_fma_f32 %6, %9, %12, %6\n"
                   "v_min3_f32 %10, |%10|, |%11|, %10\n"
                   "v_cmp_lt_f32 %0, |%10|, %13\n v_cmp_lt_f32 %1, |%11|, %13\n"
                   "v_sub_f32 %4, |%10|, %13\n v_sub_f32 %5, |%11|, %13\n v_min_u32 %14, %14, %4\n v_min_u32 %14, %14, %5\n"
                   "v_pk_fma_f32 %7, %6, %12, %7\n v_pk_fma_f32 %7, %8, %12, %7\n v_pk_fma_f32 %7, %9, %12, %7\n"
                   "v_min3_f32 %11, |%11|, |%10|, %11\n"
                   "v_cmp_lt_f32 %2, |%11|, %13\n v_cmp_lt_f32 %3, |%10|, %13\n"
                   "v_sub_f32 %4, |%11|, %13\n v_sub_f32 %5, |%10|, %13\n v_min_u32 %14, %14, %4\n v_min_u32 %14, %14, %5"
                   : "=s"(b0), "=s"(b1), "=s"(b2), "=s"(b3), "=&v"(t0), "=&v"(t1), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(f0), "+v"(f1)
                   : "s"(b), "s"(fa), "v"(dm));
      cnt += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) + __builtin_popcountll(b3) + (dm & 1); }
    if (OP == 21) {  // r04 / r05 level-2 body (filter on squares, cells.h: cells_filter_squares) per TWO packed pairs of
      // observations: 6 v_pk_fma_f32 (measure) + 2 v_pk_fma_f32 (d = v v - a), 4 v_cmp_lt_f32 of the halves into SGPR
      // pairs (inlier ballots), 2 v_min3_u32 (band minimum over the bit patterns of d): 14 vector instructions
      unsigned long long b0, b1, b2, b3; unsigned dm = 0xFFFFFFFFu;
      asm volatile("v_pk_fma_f32 %4, %5, %10, %4\n v_pk_fma_f32 %4, %6, %10, %4\n v_pk_fma_f32 %4, %7, %10, %4\n"
                   "v_pk_fma_f32 %5, %4, %10, %5\n v_pk_fma_f32 %5, %6, %10, %5\n v_pk_fma_f32 %5, %7, %10, %5\n"
                   "v_pk_fma_f32 %6, %4, %4, %6\n v_pk_fma_f32 %7, %5, %5, %7\n"
                   "v_cmp_lt_f32 %0, %8, %11\n v_cmp_lt_f32 %1, %9, %11\n v_cmp_gt_f32 %2, %8, %11\n v_cmp_gt_f32 %3, %9, %11\n"
                   "v_min3_u32 %12, %12, %8, %9\n v_min3_u32 %12, %12, %9, %8"
                   : "=s"(b0), "=s"(b1), "=s"(b2), "=s"(b3), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(f0), "+v"(f1)
                   : "s"(b), "s"(fa), "v"(dm));
      cnt += __builtin_popcountll(b0) + __builtin_popcountll(b1) + __builtin_popcountll(b2) + __builtin_popcountll(b3) + (dm & 1); }
    if (OP == 20) { unsigned r0, r1, r2, r3;   // v_readlane_b32 x 4 + v_writelane_b32 x 4 (the hypothesis broadcast / vote scatter)
      asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 7\n v_readlane_b32 %2, %6, 11\n v_readlane_b32 %3, %7, 13\n s_nop 3\n"
                   "v_writelane_b32 %4, %0, 5\n v_writelane_b32 %5, %1, 9\n v_writelane_b32 %6, %2, 17\n v_writelane_b32 %7, %3, 21"
                   : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)); }
    if (OP == 9) { asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(b)); }
  }
  out[(blockIdx.x & 255) * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + cnt + m0[0] + m1[1] + m2[2] + m3[3] + m4[0] + m5[1] + m6[2] + m7[3];
}
static FILE *g_json = nullptr;
static bool g_first = true;
template <int OP>
void run(const char *name, int lanes_per_inst, double *d, int blocks_per_cu, int inst_per_iter = 8) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  bench<OP><<<grid, 256>>>(d, 1e-9, 1.0000001, 1e-9f, 1.0000001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) bench<OP><<<grid, 256>>>(d, 1e-9, 1.0000001, 1e-9f, 1.0000001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double inst = (double)grid * 4 /*waves*/ * ITERS * inst_per_iter;  // wave-instructions
  double per_simd_cycles = ms * 1e-3 * 2.4e9 / (inst / 1024.0);       // cycles at 2.4 GHz nominal per wave-instruction per SIMD
  if (g_json) {
    fprintf(g_json, "%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"nominal_cycles_per_wave_instruction\": %.3f}",
            g_first ? "" : ",", name, blocks_per_cu, ms, per_simd_cycles);
    g_first = false;
  }
  printf("%-14s blocks/CU %d: %.3f ms  %.2f Tinst-lanes/s  (%.2f nominal cycles per wave-instruction per SIMD)\n", name, blocks_per_cu, ms, inst * 64 * (lanes_per_inst / 64.0) / (ms * 1e-3) / 1e12, per_simd_cycles);
}
int main(int argc, char **argv) {
  double *d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
  if (argc > 1) { g_json = fopen(argv[1], "w"); if (g_json) fprintf(g_json, "{\"what\": \"tools/microbench.hip: whole-chip issue rate of the vector instructions the scan kernels are made of; 256 x waves_per_simd workgroups of 256 threads (one workgroup = one wave on each SIMD of a CU), cycles = time x 2.4 GHz / (wave-instructions per SIMD)\", \"rows\": ["); }
  for (int b : {8, 4, 2, 1}) {
    run<0>("v_add_f64", 64, d, b); run<1>("v_mul_f64", 64, d, b); run<2>("v_fma_f64", 64, d, b);
    run<3>("v_add_f32", 64, d, b); run<4>("v_fma_f32", 64, d, b);
    run<5>("v_pk_add_f32", 128, d, b); run<9>("v_pk_mul_f32", 128, d, b); run<6>("v_pk_fma_f32", 128, d, b);
    run<7>("v_cmp_lt_f64", 64, d, b); run<8>("v_cmp_lt_f32", 64, d, b);
    run<10>("cmp64vcc+bcnt", 64, d, b); run<11>("cmp32vcc+bcnt", 64, d, b); run<12>("addco+addc x8", 128, d, b);
    run<13>("2cmp+8add f64", 80, d, b);
    run<17>("v_min3_f32_abs", 64, d, b); run<18>("v_min_u32", 64, d, b); run<19>("level2_mix_20", 64, d, b, 20); run<21>("level2_squares_14", 64, d, b, 14); run<20>("readlane4+writelane4", 64, d, b, 8);
    run<16>("mfma_f64_16x16x4", 64, d, b);
    run<14>("dep mul->add", 128, d, b); run<15>("2 chains", 128, d, b);
    printf("\n");
  }
  if (g_json) { fprintf(g_json, "\n]}\n"); fclose(g_json); }
  return 0;
}
