// tools/h16_bench.hip -- standalone check + timing of dense_h16.h (the fp16-split matrix-core filter of the dense scan).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -o tools/h16_bench tools/h16_bench.hip
//   tools/h16_bench [rows] [hypotheses] [reps]
// Builds synthetic rows like lsqrrecipes_amd/synth.py: dense(), runs rows -> prep -> scan -> exact decision of the
// worklist, compares every vote with a brute-force fp64 count (the reference's running sum), and measures how far the
// matrix unit's r'' is from the exact residual in units of u S (the header's bound assumes <= 83).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../lsqrrecipes_amd/csrc/dense_h16.h"

using namespace lsqr;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

__global__ void k_brute(const double *__restrict__ data, size_t stride, size_t n, const double *__restrict__ sp,
                        uint32_t H, double delta, uint32_t *__restrict__ votes) {
  const uint32_t h = blockIdx.y * 64 + (threadIdx.x & 63);
  __shared__ double s_row[4][65];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double x[64];
  for (int k = 0; k < 64; k++) x[k] = sp[(size_t)(h < H ? h : 0) * 64 + k];
  uint32_t c = 0;
  for (size_t r = (size_t)blockIdx.x * 4 + w; r < n; r += (size_t)gridDim.x * 4) {
    s_row[w][lane] = data[r * stride + lane];
    if (lane == 0) s_row[w][64] = data[r * stride + 64];
    __builtin_amdgcn_wave_barrier();
    double sum = 0.0;
    for (int k = 0; k < 64; k++) sum += s_row[w][k] * x[k];
    sum -= s_row[w][64];
    c += fabs(sum) < delta ? 1u : 0u;
    __builtin_amdgcn_wave_barrier();
  }
  if (h < H && c) atomicAdd(&votes[h], c);
}

__global__ void k_recheck(const double *__restrict__ data, size_t stride, const double *__restrict__ sp, double delta,
                          const unsigned long long *__restrict__ amb_list, const unsigned int *__restrict__ amb_counts,
                          uint32_t seg_cap, uint32_t *__restrict__ votes, unsigned long long *__restrict__ total) {
  const unsigned filled = amb_counts[blockIdx.x];
  const unsigned tot = filled < seg_cap ? filled : seg_cap;
  if (threadIdx.x == 0 && filled) atomicAdd(total, (unsigned long long)filled);
  for (unsigned e = threadIdx.x; e < tot; e += blockDim.x) {
    const unsigned long long v = amb_list[(size_t)blockIdx.x * seg_cap + e];
    const size_t row = (size_t)(v >> 32);
    const uint32_t h = (uint32_t)(v & 0xffffffffu);
    double sum = 0.0;
    for (int k = 0; k < 64; k++) sum += data[row * stride + k] * sp[(size_t)h * 64 + k];
    sum -= data[row * stride + 64];
    if (fabs(sum) < delta) atomicAdd(&votes[h], 1u);
  }
}

// r'' of rows 0..31 x hypotheses 0..31, the scan's instruction order
__global__ void k_probe(const uint4 *__restrict__ afrag, const float *__restrict__ bs, const _Float16 *__restrict__ xh,
                        const float *__restrict__ thr4, float *__restrict__ out) {
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  h16x8 a[4][2], x[4][2];
  for (int kb = 0; kb < 4; kb++)
    for (int part = 0; part < 2; part++) {
      a[kb][part] = __builtin_bit_cast(h16x8, afrag[((0 * 4 + kb) * 2 + part) * 64 + lane]);
      x[kb][part] = *(const h16x8 *)(xh + (size_t)col * 128 + part * 64 + kb * 16 + 8 * half);
    }
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  for (int kb = 0; kb < 4; kb++) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kb][0], x[kb][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kb][1], x[kb][0], acc, 0, 0, 0);
  }
  for (int kb = 0; kb < 4; kb++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kb][0], x[kb][0], acc, 0, 0, 0);
  const float nph = thr4[4 * col + 2];
  for (int i = 0; i < 16; i++) {
    const int row = 8 * (i / 4) + 4 * half + i % 4;
    out[row * 32 + col] = __builtin_fmaf(bs[row], nph, acc[i]);
  }
}

int main(int argc, char **argv) {
  const size_t N = argc > 1 ? strtoull(argv[1], nullptr, 10) : 524288;
  const uint32_t H = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const int weird = argc > 4 ? atoi(argv[4]) : 0;  // 1: near-model hypotheses and ones the filter cannot take
  const int n = 64;
  const size_t stride = 65;
  const double delta = argc > 5 ? atof(argv[5]) : 0.1;
  std::mt19937_64 g(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0), U01(0.0, 1.0);
  std::vector<double> data(N * stride), xt(64), sp((size_t)H * 64);
  for (auto &v : xt) v = U(g);
  double amax = 0.0, bmax = 0.0;
  for (size_t r = 0; r < N; r++) {
    double s = 0.0;
    for (int k = 0; k < 64; k++) {
      const double a = U(g);
      data[r * stride + k] = a;
      s += a * xt[k];
      amax = std::max(amax, fabs(a));
    }
    s *= 1.0 + 0.05 * U(g);
    if (U01(g) < 0.05) s *= 20.0;
    data[r * stride + 64] = s;
    bmax = std::max(bmax, fabs(s));
  }
  std::normal_distribution<double> G(0.0, 1.0);
  for (uint32_t h = 0; h < H; h++) {
    const double sc = weird ? (h < 16 ? 1e-4 * h : pow(10.0, -3.0 + 6.0 * U01(g))) : pow(10.0, -2.0 + 4.0 * U01(g));
    for (int k = 0; k < 64; k++) sp[(size_t)h * 64 + k] = xt[k] + sc * G(g);
  }
  if (H > 40 && weird) {  // a hypothesis that does not fit and a NaN one
    for (int k = 0; k < 64; k++) sp[(size_t)33 * 64 + k] *= 1e20;
    sp[(size_t)34 * 64 + 5] = NAN;
    for (int k = 0; k < 64; k++) sp[(size_t)35 * 64 + k] = 0.0;
  }
  const double pa = 32768.0 / amax;
  const size_t n_tiles = (N + 31) / 32 + 8;  // padded: a workgroup pass may read up to 256 rows past the end
  double *d_data, *d_sp;
  uint4 *d_afrag;
  float *d_bs, *d_thr4, *d_probe;
  _Float16 *d_xh;
  uint32_t *d_votes, *d_votes_ref;
  unsigned long long *d_amb, *d_total;
  unsigned int *d_segcnt;
  const uint32_t seg_cap = (1u << 22) / 1024;
  CK(hipMalloc(&d_data, data.size() * 8));
  CK(hipMalloc(&d_sp, sp.size() * 8));
  CK(hipMalloc(&d_afrag, n_tiles * kH16TileBytes));
  CK(hipMalloc(&d_bs, n_tiles * 32 * 4));
  CK(hipMalloc(&d_thr4, (size_t)H * 16));
  CK(hipMalloc(&d_xh, (size_t)H * 256));
  CK(hipMalloc(&d_votes, H * 4));
  CK(hipMalloc(&d_votes_ref, H * 4));
  CK(hipMalloc(&d_amb, sizeof(unsigned long long) << 22));
  CK(hipMalloc(&d_segcnt, 1024 * 4));
  CK(hipMalloc(&d_total, 8));
  CK(hipMalloc(&d_probe, 32 * 32 * 4));
  CK(hipMemcpy(d_data, data.data(), data.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sp, sp.data(), sp.size() * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_dense_rows_h16, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, 0, d_data, stride, N, n, pa,
                     d_afrag, d_bs, n_tiles);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("rows_h16: %.3f ms for %zu rows\n", ms, N);
  const size_t passes = (N + kH16RowsPerWg - 1) / kH16RowsPerWg;
  const size_t nb = std::min<size_t>(passes, getenv("H16_NB") ? (size_t)atoi(getenv("H16_NB")) : 512);
  const size_t rpb = (passes + nb - 1) / nb * kH16RowsPerWg;
  const size_t nblk = (N + rpb - 1) / rpb;
  const size_t lds = dense_h16_lds(H);
  CK(hipFuncSetAttribute((const void *)k_scan_dense_h16<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  float best = 1e9f, best_prep = 1e9f, best_re = 1e9f;
  for (int it = 0; it < reps; it++) {
    CK(hipMemset(d_votes, 0, H * 4));
    CK(hipMemset(d_segcnt, 0, 1024 * 4));
    CK(hipMemset(d_total, 0, 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_dense_prep_h16, dim3((H + 3) / 4), dim3(256), 0, 0, d_sp, H, n, 64, delta, amax, bmax, pa,
                       d_xh, d_thr4);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    best_prep = std::min(best_prep, ms);
    if (getenv("H16_DBG")) {
      auto run = [&](auto kern, const char *what) {
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, 0, d_afrag, d_bs, (size_t)0, N, rpb, d_xh, d_thr4, H,
                           d_votes, d_amb, d_segcnt, seg_cap, 0u, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                           (const uint32_t *)nullptr);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %s: %.3f ms\n", what, ms);
      };
      run(k_scan_dense_h16<64, true, 1>, "matrix instructions alone");
      run(k_scan_dense_h16<64, true, 2>, "classification alone");
      run(k_scan_dense_h16<64, true, 3>, "matrix instructions alone, no barrier / tile requests (LDS reads stay)");
      run(k_scan_dense_h16<64, true, 0>, "filter alone");
      run(k_scan_dense_h16<64, true, 8>, "filter alone, packed classification (until r05)");
      auto phases = [&]() {
        std::vector<unsigned long long> tp((size_t)nblk * 4 * 8);
        CK(hipMemcpy(tp.data(), d_amb, tp.size() * 8, hipMemcpyDeviceToHost));
        double sum[6] = {0, 0, 0, 0, 0, 0};
        for (size_t w = 0; w < (size_t)nblk * 4; w++)
          for (int i = 0; i < 6; i++) sum[i] += (double)tp[w * 8 + i];
        const double tiles = sum[4];
        printf("  per tile and wave (shader-clock ticks of s_memtime): barrier %.0f, x fragments from LDS %.0f, 24 matrix instructions issued %.0f, "
               "results + classification %.0f; tiles per wave %.0f; whole wave %.0f ticks\n",
               sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, tiles / (nblk * 4.0), sum[5] / (nblk * 4.0));
      };
      run(k_scan_dense_h16<64, true, 7>, "filter alone, phase clocks");
      phases();
      CK(hipMemset(d_segcnt, 0, 1024 * 4));
      run(k_scan_dense_h16<64, false, 7>, "with worklist, phase clocks (the clocks' dump overwrites worklist entries)");
      phases();
      CK(hipMemset(d_segcnt, 0, 1024 * 4));
      run(k_scan_dense_h16<64, false, 8>, "with worklist, packed classification (until r05)");
      CK(hipMemset(d_segcnt, 0, 1024 * 4));
      run(k_scan_dense_h16<64, false, 0>, "with worklist");
      CK(hipMemset(d_votes, 0, H * 4));
      CK(hipMemset(d_segcnt, 0, 1024 * 4));
      CK(hipMemset(d_votes, 0, H * 4));
    }
    if (getenv("H16_SKIP_AMB")) {
      CK(hipFuncSetAttribute((const void *)k_scan_dense_h16<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_scan_dense_h16<64, true>), dim3((unsigned)nblk), dim3(256), lds, 0, d_afrag, d_bs, (size_t)0, N,
                         rpb, d_xh, d_thr4, H, d_votes, d_amb, d_segcnt, seg_cap, 0u, (const uint32_t *)nullptr,
                         (const uint32_t *)nullptr, (const uint32_t *)nullptr);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  filter alone (no worklist): %.3f ms\n", ms);
      CK(hipMemset(d_votes, 0, H * 4));
    }
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_scan_dense_h16<64>), dim3((unsigned)nblk), dim3(256), lds, 0, d_afrag, d_bs, (size_t)0, N, rpb,
                       d_xh, d_thr4, H, d_votes, d_amb, d_segcnt, seg_cap, 0u, (const uint32_t *)nullptr,
                       (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_recheck, dim3((unsigned)nblk), dim3(256), 0, 0, d_data, stride, d_sp, delta, d_amb, d_segcnt,
                       seg_cap, d_votes, d_total);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
    best_re = std::min(best_re, ms);
  }
  unsigned long long total = 0;
  CK(hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost));
  std::vector<unsigned int> seg(1024);
  CK(hipMemcpy(seg.data(), d_segcnt, 1024 * 4, hipMemcpyDeviceToHost));
  unsigned segmax = 0;
  for (auto v : seg) segmax = std::max(segmax, v);
  const double flops = 2.0 * (double)N * 64.0 * (double)H;
  printf("scan: %.3f ms (%zu workgroups, %zu B LDS) = %.1f TFLOP/s of the logical product; prep %.3f ms; recheck %.3f ms\n",
         best, nblk, lds, flops / best * 1e-9, best_prep, best_re);
  printf("worklist: %llu pairs = %.3e of all; fullest segment %u of %u\n", total, (double)total / ((double)N * H), segmax,
         seg_cap);
  CK(hipMemset(d_votes_ref, 0, H * 4));
  hipLaunchKernelGGL(k_brute, dim3(1024, (H + 63) / 64), dim3(256), 0, 0, d_data, stride, N, d_sp, H, delta, d_votes_ref);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> v(H), vr(H);
  CK(hipMemcpy(v.data(), d_votes, H * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(vr.data(), d_votes_ref, H * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (uint32_t h = 0; h < H; h++)
    if (v[h] != vr[h]) {
      if (bad < 10) printf("  MISMATCH h=%u filter %u exact %u\n", h, v[h], vr[h]);
      bad++;
    }
  printf("votes: %zu of %u hypotheses differ; votes[0..3] = %u %u %u %u\n", bad, H, vr[0], vr[1], vr[2], vr[3]);
  // the matrix unit against the exact residual
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_afrag, d_bs, d_xh, d_thr4, d_probe);
  CK(hipDeviceSynchronize());
  std::vector<float> pr(1024), thr((size_t)H * 4);
  CK(hipMemcpy(pr.data(), d_probe, 4096, hipMemcpyDeviceToHost));
  CK(hipMemcpy(thr.data(), d_thr4, (size_t)H * 16, hipMemcpyDeviceToHost));
  printf("thr4[0] = %g %g (bits %08x) %g; probe[0][0] = %g\n", thr[0], thr[1], *(unsigned *)&thr[1], thr[2], pr[0]);
  {
    std::vector<unsigned long long> am(8);
    CK(hipMemcpy(am.data(), d_amb, 64, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) printf("  amb[%d] = row %llu hyp %llu\n", i, am[i] >> 32, am[i] & 0xffffffffull);
    int na = 0, ni = 0;
    for (int r = 0; r < 32; r++)
      for (int h = 0; h < 32; h++) {
        const float rr = pr[r * 32 + h];
        const float d = fmaf(rr, rr, thr[4 * h]);
        unsigned du;
        memcpy(&du, &d, 4);
        unsigned bb;
        memcpy(&bb, &thr[4 * h + 1], 4);
        if (du <= bb) na++;
        if (du >> 31) ni++;
      }
    printf("  host classification of the probe tile: %d ambiguous, %d certain inliers of 1024\n", na, ni);
  }
  double worst = 0.0;
  for (int r = 0; r < 32; r++)
    for (int h = 0; h < 32 && h < (int)H; h++) {
      const double ph = -(double)thr[4 * h + 2];
      long double s = 0.0L, l1 = 0.0L;
      for (int k = 0; k < 64; k++) {
        s += (long double)data[r * stride + k] * (long double)sp[(size_t)h * 64 + k];
        l1 += fabsl((long double)sp[(size_t)h * 64 + k]);
      }
      s -= (long double)data[r * stride + 64];
      const double exact = (double)(s * (long double)pa * (long double)ph);
      const double S = 32768.0 * (double)l1 * ph;
      const double dev = fabs((double)pr[r * 32 + h] - exact) / (5.9604644775390625e-08 * S);
      worst = std::max(worst, dev);
    }
  printf("probe: largest |r'' - res''| = %.3f u S (bound assumed by the thresholds: 83)\n", worst);
  {
    float *d_al;
    CK(hipMalloc(&d_al, 64));
    hipLaunchKernelGGL(k_dense_h16_probe, dim3(1), dim3(64), 0, 0, d_al);
    CK(hipDeviceSynchronize());
    float al[8];
    CK(hipMemcpy(al, d_al, 32, hipMemcpyDeviceToHost));
    for (int v = 0; v < 8; v++) {
      const double sm = (double)(float)(_Float16)dense_h16_probe_small(v);
      const int nsmall = v == 1 ? 12 + 1 : 15;
      const double exact = 1048576.0 + nsmall * sm;
      printf("align probe %d: %d terms of %+.5f beside 2^20 (ulp 0.125): result %.4f exact %.4f -> off by %+.3f ulp = %+.2f u of the sum of magnitudes\n",
             v, nsmall, sm, (double)al[v], exact, ((double)al[v] - exact) / 0.125,
             ((double)al[v] - exact) / (5.9604644775390625e-08 * (1048576.0 + nsmall * fabs(sm))));
    }
    printf("align probe: worst %.2f u of the sum of magnitudes (the library refuses the fp16 filter beyond %.1f)\n",
           dense_h16_probe_worst(al), kH16ProbeLimit);
    for (int rounds : {64, 4096}) {
      hipLaunchKernelGGL(k_dense_h16_probe_random, dim3(1), dim3(64), 0, 0, d_al, rounds);
      CK(hipDeviceSynchronize());
      float w;
      CK(hipMemcpy(&w, d_al, 4, hipMemcpyDeviceToHost));
      printf("random probe: %d instructions x 1024 outputs, worst |result - exact| = %.3f u of the largest magnitude involved\n",
             rounds, (double)w);
    }
  }
  return bad ? 1 : 0;
}
