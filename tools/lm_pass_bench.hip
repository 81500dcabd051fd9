// tools/lm_pass_bench.hip -- the LM tile pass of the US calibration (kernels.h: k_lm_pass_mfma_t) alone: 500 k records
// in the tile layout, launched back to back; the whole pass, its loads alone, loads + row formation without the LDS
// transposition and the matrix instructions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o tools/lm_pass_bench tools/lm_pass_bench.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../lsqrrecipes_amd/csrc/models.h"
#include "../lsqrrecipes_amd/csrc/us.h"
#include "../lsqrrecipes_amd/csrc/kernels.h"

using namespace lsqr;
#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// the pass's loop skeleton with parts switched off (the product's kernel has no such switches): V 1 = the loads alone,
// 2 = loads + rows (no LDS transposition, no matrix instructions), 3 = everything, but all sixteen LDS operands read
// before the matrix instructions, 4 = 3 with two accumulator chains, 5 = everything, two tiles requested ahead
template <class M, int V>
__global__ __launch_bounds__(kBlock) void k_pass_parts(const double *__restrict__ tiles, size_t n, typename M::LmCoef coef,
                                                       double *__restrict__ partials) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int P = 17, REC = M::REC, PF = V == 5 ? 2 : 1;
  __shared__ double s_z[kBlock / 64][64 * P];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 4, c16 = lane & 15;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  double *tile = s_z[wave];
  const size_t ntiles = (n + 63) / 64;
  const size_t W = (size_t)gridDim.x * (kBlock / 64), w0 = (size_t)blockIdx.x * (kBlock / 64) + wave;
  double nx[PF][REC];
  auto fetch = [&](size_t t, double *dst) {
    const double *src = tiles + t * (size_t)(REC * 64) + lane;
#pragma unroll
    for (int j = 0; j < REC; j++) dst[j] = (j == 12) ? 0.0 : src[(size_t)j * 64];
  };
  auto process = [&](size_t t, const double *x) {
    if (V == 1) {
      double sacc = 0.0;
#pragma unroll
      for (int j = 0; j < REC; j++) sacc += x[j];
      acc[0] += sacc;
      return;
    }
    double z[16];
#pragma unroll
    for (int j = 0; j < 16; j++) z[j] = 0.0;
    if (t * 64 + lane < n) M::lm_row(x, coef, z);
    if (V == 2) {
      double sacc = 0.0;
#pragma unroll
      for (int j = 0; j < 16; j++) sacc += z[j];
      acc[0] += sacc;
      return;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) tile[lane * P + j] = z[j];
    __builtin_amdgcn_wave_barrier();
    if (V == 3 || V == 4) {
      double v[16];
#pragma unroll
      for (int s = 0; s < 16; s++) v[s] = tile[(4 * s + k) * P + c16];
      __builtin_amdgcn_sched_barrier(0);
      if (V == 3) {
#pragma unroll
        for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[s], v[s], acc, 0, 0, 0);
      } else {
        d4 acc2 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[s], v[s], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(v[s + 1], v[s + 1], acc2, 0, 0, 0);
        }
        acc += acc2;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; s++) {
        const double v = tile[(4 * s + k) * P + c16];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
#pragma unroll
  for (int p = 0; p < PF; p++)
    if (w0 + p * W < ntiles) fetch(w0 + p * W, nx[p]);
  for (size_t t = w0; t < ntiles; t += PF * W) {
#pragma unroll
    for (int p = 0; p < PF; p++) {
      const size_t tt = t + p * W;
      if (tt >= ntiles) break;
      double x[REC];
#pragma unroll
      for (int j = 0; j < REC; j++) x[j] = nx[p][j];
      if (tt + PF * W < ntiles) fetch(tt + PF * W, nx[p]);
      process(tt, x);
    }
  }
  // (the product folds the four waves through LDS and writes 91 sums per workgroup; here one value per lane)
  partials[((size_t)blockIdx.x * kBlock + threadIdx.x) % (2048 * 128)] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main(int argc, char **argv) {
  typedef USModel<true> M;
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 500000;
  const int reps = argc > 2 ? atoi(argv[2]) : 300;
  const size_t ntiles = (n + 63) / 64;
  std::vector<double> h(ntiles * 64 * M::REC);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (auto &v : h) v = U(g) * 100.0;
  double *d_tiles, *d_part;
  CK(hipMalloc(&d_tiles, h.size() * 8));
  CK(hipMemcpy(d_tiles, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_part, sizeof(double) * 2048 * 128));
  double xk[11] = {1, 2, 3, 10, 20, 30, 0.1, 0.2, 0.3, 0.05, 0.06};
  M::LmCoef coef;
  M::lm_coef(xk, coef);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](auto kern, int nb, const char *what) {
    hipLaunchKernelGGL(kern, dim3(nb), dim3(kBlock), 0, 0, d_tiles, n, coef, d_part);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(kern, dim3(nb), dim3(kBlock), 0, 0, d_tiles, n, coef, d_part);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-58s %4d workgroups: %6.2f us per launch = %5.2f TB/s of the %zu MB the pass reads\n", what, nb, us,
           (double)n * 14 * 8 / (us * 1e-6) * 1e-12, n * 14 * 8 / 1000000);
  };
  for (int nb : {488, 976, 1952}) {
    run(k_lm_pass_mfma_t<M>, nb, "the pass (kernels.h: k_lm_pass_mfma_t)");
    run(k_pass_parts<M, 1>, nb, "its loads alone");
    run(k_pass_parts<M, 2>, nb, "loads + rows (no LDS transposition, no matrix instructions)");
    run(k_pass_parts<M, 3>, nb, "everything, the sixteen LDS operands read before the matrix instructions");
    run(k_pass_parts<M, 4>, nb, "... and two accumulator chains");
    run(k_pass_parts<M, 5>, nb, "everything, two tiles requested ahead");
  }
  // several passes at once (several fits in flight): the same kernel on S streams, each with its own copy of the tiles
  for (int S : {2, 4, 8}) {
    std::vector<hipStream_t> st(S);
    std::vector<double *> buf(S);
    for (int i = 0; i < S; i++) {
      CK(hipStreamCreate(&st[i]));
      CK(hipMalloc(&buf[i], h.size() * 8));
      CK(hipMemcpy(buf[i], h.data(), h.size() * 8, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < S; i++)
      hipLaunchKernelGGL((k_lm_pass_mfma_t<M>), dim3(488), dim3(kBlock), 0, st[i], buf[i], n, coef, d_part + (size_t)i * 512 * 128 % (2048 * 128 - 488 * 128));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; r++)
      for (int i = 0; i < S; i++)
        hipLaunchKernelGGL((k_lm_pass_mfma_t<M>), dim3(488), dim3(kBlock), 0, st[i], buf[i], n, coef, d_part);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * S);
    printf("the pass on %d streams at once (a set of tiles each): %6.2f us per pass in the aggregate = %5.2f TB/s\n", S, us,
           (double)n * 14 * 8 / (us * 1e-6) * 1e-12);
    for (int i = 0; i < S; i++) {
      CK(hipFree(buf[i]));
      CK(hipStreamDestroy(st[i]));
    }
  }
  return 0;
}
