// tools/ph16_bench.hip -- standalone check + timing of phantom_h16.h (the plane phantom's agree() scan on the fp16 matrix cores).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o tools/ph16_bench tools/ph16_bench.hip
//   tools/ph16_bench [frames] [hypotheses] [reps]
// Synthetic frames (rotation entries in [-1, 1], translations to 300, pixels to 640) and hypotheses (a plane fitted to
// nothing: random coefficients scaled so that a share of the frames falls near the threshold); rows -> prep -> scan ->
// exact decision of the worklist; every vote is compared with a brute-force fp64 count of PhantomModel::agree.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../lsqrrecipes_amd/csrc/models.h"
#include "../lsqrrecipes_amd/csrc/phantom_h16.h"

using namespace lsqr;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

__global__ void k_brute(const double *__restrict__ data, size_t stride, size_t n, const double *__restrict__ sp, uint32_t H,
                        ModelConsts mc, uint32_t *__restrict__ votes) {
  const uint32_t h = blockIdx.y;
  uint32_t c = 0;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (size_t)gridDim.x * blockDim.x) {
    double x[PhantomModel::REC];
    PhantomModel::load(data + r * stride, mc, x);
    c += PhantomModel::agree(sp + (size_t)h * PhantomModel::SP, x, mc) ? 1u : 0u;
  }
  if (c) atomicAdd(&votes[h], c);
}

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
  const uint32_t H = argc > 2 ? (uint32_t)atoi(argv[2]) : 4096;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const size_t stride = 15;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> data(n * stride), sp((size_t)H * PhantomModel::SP, 0.0);
  for (size_t i = 0; i < n; i++) {
    double *p = &data[i * stride];
    for (int k = 0; k < 9; k++) p[k] = U(rng);
    for (int k = 9; k < 12; k++) p[k] = 300.0 * U(rng);
    p[12] = 0.0;
    p[13] = 320.0 * (1.0 + U(rng)), p[14] = 240.0 * (1.0 + U(rng));
  }
  for (uint32_t h = 0; h < H; h++) {
    double *q = &sp[(size_t)h * PhantomModel::SP];
    for (int k = 0; k < 18; k++) q[11 + k] = 1e-3 * U(rng);
    for (int k = 18; k < 27; k++) q[11 + k] = 0.5 * U(rng);
    for (int k = 27; k < 30; k++) q[11 + k] = 1e-2 * U(rng);
    q[2] = U(rng);
  }
  ModelConsts mc{};
  mc.delta = 2.0, mc.delta_sq = 4.0, mc.thr = 2.0;
  double X = 0.0, Rm = 0.0;
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 15; k++) {
      const double a = fabs(data[i * stride + k]);
      if (k != 12) X = a > X ? a : X;
      if (k < 9) Rm = a > Rm ? a : Rm;
    }
  mc.absmax = X, mc.absmax_rot = Rm;
  const Us16Scales sc = us16_scales<true>(X, Rm);
  const size_t n_tiles = (n + 31) / 32 + 2;
  double *d_data, *d_sp;
  uint4 *d_a, *d_x;
  float *d_thr;
  uint32_t *d_votes, *d_ref;
  unsigned long long *d_amb;
  unsigned int *d_seg, *d_max;
  const uint32_t seg_cap = 1u << 14;
  CK(hipMalloc(&d_data, data.size() * 8));
  CK(hipMalloc(&d_sp, sp.size() * 8));
  CK(hipMalloc(&d_a, n_tiles * (size_t)kPh16FrameTile));
  CK(hipMalloc(&d_x, (size_t)((H + 31) / 32) * 4096));
  CK(hipMalloc(&d_thr, sizeof(float) * 4 * (H + 32)));
  CK(hipMalloc(&d_votes, 4 * (size_t)H));
  CK(hipMalloc(&d_ref, 4 * (size_t)H));
  CK(hipMalloc(&d_amb, 8ull * 256 * seg_cap));
  CK(hipMalloc(&d_seg, 4 * 1024));
  CK(hipMalloc(&d_max, 8));
  CK(hipMemcpy(d_data, data.data(), data.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sp, sp.data(), sp.size() * 8, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void *)k_scan_phantom_h16, hipFuncAttributeMaxDynamicSharedMemorySize,
                         (int)phantom_h16_lds(kPh16HypChunk)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_phantom_rows_h16, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, 0, d_data, stride, n, sc, d_a, n_tiles);
  CK(hipDeviceSynchronize());
  float best_scan = 1e9f, best_all = 1e9f;
  unsigned long long amb_total = 0;
  for (int rep = 0; rep < reps; rep++) {
    CK(hipMemset(d_votes, 0, 4 * (size_t)H));
    CK(hipMemset(d_seg, 0, 4 * 1024));
    CK(hipMemset(d_max, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_phantom_prep_h16, dim3((H + 31 + 255) / 256), dim3(256), 0, 0, d_sp, (int)PhantomModel::SP, H,
                       sqrt(mc.delta_sq), X, Rm, sc, d_x, d_thr);
    const size_t passes = (n + kPh16Wg - 1) / kPh16Wg;
    const unsigned nblk = (unsigned)(passes < 256 ? passes : 256);
    float scan_ms = 0.0f;
    for (size_t h0 = 0; h0 < H; h0 += kPh16HypChunk) {
      const uint32_t hc = (uint32_t)(H - h0 < kPh16HypChunk ? H - h0 : kPh16HypChunk);
      hipEvent_t s0, s1;
      CK(hipEventCreate(&s0));
      CK(hipEventCreate(&s1));
      CK(hipEventRecord(s0));
      hipLaunchKernelGGL(k_scan_phantom_h16, dim3(nblk), dim3(kPh16Wg), phantom_h16_lds(hc), 0, d_a, n, (size_t)0, n,
                         d_x + (h0 / 32) * 256, d_thr + 4 * h0, hc, d_votes, d_amb, d_seg, seg_cap, (uint32_t)h0,
                         (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
      CK(hipEventRecord(s1));
      hipLaunchKernelGGL((k_us_recheck_seg<PhantomModel>), dim3(256), dim3(1024), 0, 0, d_data, stride, d_sp,
                         (int)PhantomModel::SP, mc, d_amb, d_seg, seg_cap, d_votes, d_max);
      CK(hipEventSynchronize(s1));
      float ms;
      CK(hipEventElapsedTime(&ms, s0, s1));
      scan_ms += ms;
      CK(hipEventDestroy(s0));
      CK(hipEventDestroy(s1));
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best_all = ms < best_all ? ms : best_all;
    best_scan = scan_ms < best_scan ? scan_ms : best_scan;
  }
  unsigned int mx[2];
  CK(hipMemcpy(mx, d_max, 8, hipMemcpyDeviceToHost));
  CK(hipMemset(d_ref, 0, 4 * (size_t)H));
  hipLaunchKernelGGL(k_brute, dim3(64, H), dim3(256), 0, 0, d_data, stride, n, d_sp, H, mc, d_ref);
  std::vector<uint32_t> v(H), r(H);
  CK(hipMemcpy(v.data(), d_votes, 4 * (size_t)H, hipMemcpyDeviceToHost));
  CK(hipMemcpy(r.data(), d_ref, 4 * (size_t)H, hipMemcpyDeviceToHost));
  uint32_t bad = 0;
  for (uint32_t h = 0; h < H; h++) bad += v[h] != r[h];
  const double mf = 6.0 * (double)((n + 31) / 32) * ((H + 31) / 32);
  printf("scan kernels: %.3f ms (%zu frames x %u hypotheses; %.1f ns per matrix instruction and SIMD); with prep + recheck + "
         "event gaps %.3f ms\n", best_scan, n, H, best_scan * 1e6 / (mf / 1024.0), best_all);
  printf("fullest worklist segment of a launch: %u of %u\n", mx[0], seg_cap);
  printf("votes: %u of %u hypotheses differ; votes[0..3] = %u %u %u %u (ref %u %u %u %u)\n", bad, H, v[0], v[1], v[2], v[3], r[0],
         r[1], r[2], r[3]);
  (void)amb_total;
  return bad != 0;
}
