// tools/mfma_rate.hip -- what one SIMD sustains of v_mfma_f32_32x32x16_f16 when EVERY SIMD of the chip issues them back to
// back: time per instruction, shader cycles per instruction (s_memtime), hence the clock under this load; and with V
// independent v_fma_f32 behind every matrix instruction (do they run in its shadow?).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_rate tools/mfma_rate.hip;  tools/mfma_rate [waves per SIMD] [chains] [V] [0 v_fma_f32 | 1 v_pk_fma_f32 | 2 v_alignbit_b32 | 3 v_min3_u32]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH, int V, int KIND>
__global__ __launch_bounds__(256) void k_rate(float *out, unsigned long long *cyc, int iters, float seed) {
  h16x8 a, b;
  for (int i = 0; i < 8; i++) a[i] = (_Float16)(seed + threadIdx.x * 1e-3f + i), b[i] = (_Float16)(seed * 0.5f + i);
  f32x16 acc[CH];
  for (int c = 0; c < CH; c++)
    for (int i = 0; i < 16; i++) acc[c][i] = 0.0f;
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = seed * (i + 1);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 w[5];
  for (int i = 0; i < 5; i++) w[i] = (f32x2){seed * i, seed + i};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < CH; c++) {
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < V; j++) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(seed));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[j & 3]) : "v"(w[4]));
        if (KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(v[j & 7]) : "v"(seed));
        if (KIND == 3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(seed), "v"(v[(j + 1) & 7]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
  for (int c = 0; c < CH; c++)
    for (int i = 0; i < 16; i++) s += acc[c][i];
  for (int i = 0; i < 8; i++) s += v[i];
  for (int i = 0; i < 5; i++) s += w[i].x + w[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int KIND>
static void launch2(int ch, int blocks, float *out, unsigned long long *cyc, int iters) {
  if (ch == 1) hipLaunchKernelGGL((k_rate<1, V, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
  else if (ch == 2) hipLaunchKernelGGL((k_rate<2, V, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
  else hipLaunchKernelGGL((k_rate<4, V, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
}
static int g_kind = 0;  // 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_alignbit_b32, 3 v_min3_u32
template <int V>
static void launch(int ch, int blocks, float *out, unsigned long long *cyc, int iters) {
  if (g_kind == 1) launch2<V, 1>(ch, blocks, out, cyc, iters);
  else if (g_kind == 2) launch2<V, 2>(ch, blocks, out, cyc, iters);
  else if (g_kind == 3) launch2<V, 3>(ch, blocks, out, cyc, iters);
  else launch2<V, 0>(ch, blocks, out, cyc, iters);
}

int main(int argc, char **argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;   // waves per SIMD
  const int ch = argc > 2 ? atoi(argv[2]) : 4;    // independent accumulator chains per wave
  const int nv = argc > 3 ? atoi(argv[3]) : 0;     // v_fma_f32 behind every matrix instruction
  g_kind = argc > 4 ? atoi(argv[4]) : 0;
  const int iters = 20000;
  const int blocks = 256 * wps;                   // 256 threads = 4 waves = one per SIMD of a CU
  float *out;
  unsigned long long *cyc;
  hipMalloc(&out, sizeof(float) * 256 * blocks);
  hipMalloc(&cyc, 8 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    if (nv == 0) launch<0>(ch, blocks, out, cyc, iters);
    else if (nv <= 2) launch<2>(ch, blocks, out, cyc, iters);
    else if (nv <= 4) launch<4>(ch, blocks, out, cyc, iters);
    else if (nv <= 6) launch<6>(ch, blocks, out, cyc, iters);
    else if (nv <= 8) launch<8>(ch, blocks, out, cyc, iters);
    else launch<12>(ch, blocks, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c0;
    hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    const double n_per_simd = (double)iters * (ch == 1 ? 1 : ch == 2 ? 2 : 4) * wps;
    printf("waves/SIMD %d, chains %d, %d x kind %d behind each: %.3f ms, %.2f ns per instruction and SIMD = %.1f TFLOP/s on 1024 SIMDs; counter ticks per "
           "instruction of one wave %.1f\n", wps, ch, nv, g_kind, ms, ms * 1e6 / n_per_simd, 32768.0 * 1024 / (ms * 1e6 / n_per_simd) / 1e3,
           (double)c0 / (iters * (ch == 1 ? 1 : ch == 2 ? 2 : 4)));
  }
  return 0;
}
