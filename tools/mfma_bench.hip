// mfma_bench.hip -- would level 2 of the plane's scan run faster on the matrix cores?  (r03: measured, NOT built.)
// The idea: v_mfma_f32_16x16x4_f32 evaluates s = n . x' + d0 for 16 observations x 16 hypotheses in one instruction
// (A row = (x', 1), B column = (n32, d0)); each lane then holds 4 values of one hypothesis, squares them against its
// own threshold (v_fma_f32), shifts the sign bits into a bit register (v_alignbit_b32: no ballots, no scalar unit) and
// tracks the band minimum (v_min3_u32): 10 VALU instructions beside each MFMA instead of 15.5 without it.
// This program times that inner loop in isolation (and checks the matrix core's arithmetic):
//   * the fp32 MFMA is bit-equal to the ascending chain of four fmaf -- the error analysis of cells.h would carry over;
//   * but an fp32 MFMA and the VALU instructions of other waves on the same SIMD do NOT overlap: MFMAs alone 48 nominal
//     cycles each, the consumers alone 29, together 82 - 85 whatever the arrangement (1 / 2 / 4 MFMAs in flight, 32x32x2)
//     and with v_pk_fma_f32 for the squares 90 (packed fp32 beside MFMAs is an anti-lever, MI355X_MICROARCH.md) --
//     against 66 cycles for the same 4 values per lane in k_scan_pairs.  A full kernel built on it (same counting pass
//     and shares as k_scan_pairs, votes bit-identical) took 1.58 ms per 4096-hypothesis batch against 1.12 ms.
// Results: profiles/r03_mfma_bench.txt.
// build: hipcc --offload-arch=gfx950 -O3 -DSCALAR_FMA -o tools/mfma_bench tools/mfma_bench.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void consume(const v4f C, const v2f na, uint32_t &R0, uint32_t &R1, uint32_t &R2, uint32_t &R3,
                                        uint32_t &dmin) {
  v2f s01, s23;
  s01.x = C.x, s01.y = C.y, s23.x = C.z, s23.y = C.w;
#ifdef SCALAR_FMA
  v2f d01, d23;
  asm("v_fma_f32 %0, %1, %1, %2" : "=v"(d01.x) : "v"(C.x), "v"(na.x));
  asm("v_fma_f32 %0, %1, %1, %2" : "=v"(d01.y) : "v"(C.y), "v"(na.x));
  asm("v_fma_f32 %0, %1, %1, %2" : "=v"(d23.x) : "v"(C.z), "v"(na.x));
  asm("v_fma_f32 %0, %1, %1, %2" : "=v"(d23.y) : "v"(C.w), "v"(na.x));
#else
  const v2f d01 = __builtin_elementwise_fma(s01, s01, na), d23 = __builtin_elementwise_fma(s23, s23, na);
#endif
  const v2u u01 = __builtin_bit_cast(v2u, d01), u23 = __builtin_bit_cast(v2u, d23);
  R0 = __builtin_amdgcn_alignbit(R0, u01.x, 31);
  R1 = __builtin_amdgcn_alignbit(R1, u01.y, 31);
  R2 = __builtin_amdgcn_alignbit(R2, u23.x, 31);
  R3 = __builtin_amdgcn_alignbit(R3, u23.y, 31);
  asm("v_min3_u32 %0, %0, %1, %2" : "+v"(dmin) : "v"(u01.x), "v"(u01.y));
  asm("v_min3_u32 %0, %0, %1, %2" : "+v"(dmin) : "v"(u23.x), "v"(u23.y));
}

// MODE 0: one MFMA, then its consumers (what the compiler makes of the plain loop)
// MODE 1: two MFMAs in flight (consume j while j + 1 runs)
// MODE 2: four MFMAs in flight
// MODE 3: MFMAs only      MODE 4: consumers only (C from registers)
// MODE 5: v_mfma_f32_32x32x2_f32 pairs (16 results per two instructions), two pairs in flight
template <int MODE>
__global__ __launch_bounds__(256) void k_loop(const float *__restrict__ in, uint32_t *__restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  float A[32];
#pragma unroll
  for (int j = 0; j < 32; j++) A[j] = in[j * 64 + lane];
  const float B = in[2048 + lane];
  v2f na;
  na.x = -in[2112 + lane], na.y = na.x;
  uint32_t R0 = 0, R1 = 0, R2 = 0, R3 = 0, dmin = 0xFFFFFFFFu, acc = 0;
  const v4f zero = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 32; j++) {
        const v4f C = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], B, zero, 0, 0, 0);
        consume(C, na, R0, R1, R2, R3, dmin);
      }
    } else if (MODE == 1) {
      v4f C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[0], B, zero, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 32; j++) {
        v4f C1 = zero;
        if (j + 1 < 32) C1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j + 1], B, zero, 0, 0, 0);
        consume(C0, na, R0, R1, R2, R3, dmin);
        C0 = C1;
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 32; j += 4) {
        const v4f C0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], B, zero, 0, 0, 0);
        const v4f C1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j + 1], B, zero, 0, 0, 0);
        const v4f C2 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j + 2], B, zero, 0, 0, 0);
        const v4f C3 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j + 3], B, zero, 0, 0, 0);
        consume(C0, na, R0, R1, R2, R3, dmin);
        consume(C1, na, R0, R1, R2, R3, dmin);
        consume(C2, na, R0, R1, R2, R3, dmin);
        consume(C3, na, R0, R1, R2, R3, dmin);
      }
    } else if (MODE == 3) {
      v4f S = zero;
#pragma unroll
      for (int j = 0; j < 32; j++) {
        const v4f C = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j], B, zero, 0, 0, 0);
        asm volatile("" ::"v"(C));
      }
      R0 += __builtin_bit_cast(uint32_t, S.x);
    } else if (MODE == 4) {
      v4f C;
      C.x = A[0], C.y = A[1], C.z = A[2], C.w = A[3];
#pragma unroll
      for (int j = 0; j < 32; j++) {
        consume(C, na, R0, R1, R2, R3, dmin);
        C.x += 1.0f;
      }
    } else if (MODE == 5) {
      const v16f z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 32; j += 8) {  // (same number of results: 8 x 4 = 32 = 2 x 16)
        v16f C = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j], B, z16, 0, 0, 0);
        C = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j + 1], B, C, 0, 0, 0);
        v16f D = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j + 2], B, z16, 0, 0, 0);
        D = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j + 3], B, D, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
          v4f c4;
          c4.x = C[k], c4.y = C[k + 1], c4.z = C[k + 2], c4.w = C[k + 3];
          consume(c4, na, R0, R1, R2, R3, dmin);
        }
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
          v4f c4;
          c4.x = D[k], c4.y = D[k + 1], c4.z = D[k + 2], c4.w = D[k + 3];
          consume(c4, na, R0, R1, R2, R3, dmin);
        }
      }
    }
    acc += (uint32_t)(__builtin_popcount(R0) + __builtin_popcount(R1) + __builtin_popcount(R2) + __builtin_popcount(R3));
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + dmin;
}

// accuracy: C = A(16 x 4) B(4 x 16) through the matrix core against fp64
__global__ void k_acc(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ c) {
  const int lane = threadIdx.x;
  const v4f zero = {0.f, 0.f, 0.f, 0.f};
  const v4f C = __builtin_amdgcn_mfma_f32_16x16x4f32(a[blockIdx.x * 64 + lane], b[blockIdx.x * 64 + lane], zero, 0, 0, 0);
  for (int i = 0; i < 4; i++) c[blockIdx.x * 256 + i * 64 + lane] = C[i];
}

template <int MODE>
static void run(const char *name, const float *d_in, uint32_t *d_out, int per_cu) {
  const int iters = 400;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_loop<MODE>, dim3(256 * per_cu), dim3(256), 0, 0, d_in, d_out, 10);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_loop<MODE>, dim3(256 * per_cu), dim3(256), 0, 0, d_in, d_out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // per SIMD: per_cu waves, each iters * 32 MFMA-equivalents (4 results per lane each)
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)per_cu * iters * 32);
  printf("%-44s waves/SIMD %d: %.3f ms, %.1f nominal cycles per (MFMA + consumers) per SIMD\n", name, per_cu, ms, cyc);
}

int main() {
  std::vector<float> h(2176);
  srand(1);
  for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
  float *d_in;
  uint32_t *d_out;
  CHECK(hipMalloc(&d_in, h.size() * 4));
  CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  CHECK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int w : {1, 2, 4}) {
    run<0>("one MFMA, then its consumers", d_in, d_out, w);
    run<1>("two MFMAs in flight", d_in, d_out, w);
    run<2>("four MFMAs in flight", d_in, d_out, w);
    run<3>("MFMAs only", d_in, d_out, w);
    run<4>("consumers only", d_in, d_out, w);
    run<5>("32x32x2 pairs, two in flight", d_in, d_out, w);
  }
  // accuracy and layout
  const int NB = 4096;
  std::vector<float> a(NB * 64), b(NB * 64), c(NB * 256);
  for (auto &v : a) v = ((float)rand() / RAND_MAX - 0.5f) * 40.0f;
  for (auto &v : b) v = ((float)rand() / RAND_MAX - 0.5f) * 2.0f;
  float *da, *db, *dc;
  CHECK(hipMalloc(&da, a.size() * 4));
  CHECK(hipMalloc(&db, b.size() * 4));
  CHECK(hipMalloc(&dc, c.size() * 4));
  CHECK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_acc, dim3(NB), dim3(64), 0, 0, da, db, dc);
  CHECK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst_fma = 0;
  long exact_chain[2] = {0, 0}, total = 0;
  for (int blk = 0; blk < NB; blk++)
    for (int i = 0; i < 4; i++)
      for (int lane = 0; lane < 64; lane++) {
        const int row = 4 * (lane / 16) + i, col = lane % 16;  // C layout: lane l, register i
        double ref = 0, mag = 0;
        float t[4];
        for (int k = 0; k < 4; k++) {
          const float av = a[blk * 64 + k * 16 + row], bv = b[blk * 64 + k * 16 + col];  // A: lane = k * 16 + row
          ref += (double)av * bv;
          mag += fabs((double)av * bv);
          t[k] = av * bv;
          (void)t;
        }
        const float got = c[blk * 256 + i * 64 + lane];
        const double err = fabs((double)got - ref) / (mag * 5.9604644775390625e-08);
        if (err > worst) worst = err;
        // chained FMAs, k ascending / descending
        float f0 = 0.f, f1 = 0.f;
        for (int k = 0; k < 4; k++) f0 = fmaf(a[blk * 64 + k * 16 + row], b[blk * 64 + k * 16 + col], f0);
        for (int k = 3; k >= 0; k--) f1 = fmaf(a[blk * 64 + k * 16 + row], b[blk * 64 + k * 16 + col], f1);
        exact_chain[0] += got == f0;
        exact_chain[1] += got == f1;
        total++;
        const double ef = fabs((double)f0 - ref) / (mag * 5.9604644775390625e-08);
        if (ef > worst_fma) worst_fma = ef;
      }
  printf("accuracy over %ld results: worst |C - exact| = %.3f u sum|a_k b_k| (chained fmaf: %.3f); bit-equal to the ascending "
         "fmaf chain: %ld, to the descending one: %ld\n", total, worst, worst_fma, exact_chain[0], exact_chain[1]);
  return 0;
}
