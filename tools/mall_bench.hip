// tools/mall_bench.hip -- how fast does a kernel re-read a buffer that fits the 256 MB Infinity Cache, against one
// that does not?  (The LM fit of the US calibration re-reads its 56 MB consensus set 5000 times: DESIGN 3.4b.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/mall_bench tools/mall_bench.hip && tools/mall_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// every thread streams uint4s with a grid stride, U loads in flight
template <int U>
__global__ __launch_bounds__(256) void k_read(const uint4 *__restrict__ p, size_t n, uint32_t *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = p[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  for (; i < n; i += stride) acc ^= p[i].x;
  if (acc == 0x12345678u) out[0] = acc;  // (never: keeps the loads)
}

// the same with 8-byte loads (the LM pass's tiles are read field by field, one double per lane and load)
template <int U>
__global__ __launch_bounds__(256) void k_read8(const uint2 *__restrict__ p, size_t n, uint32_t *__restrict__ out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    uint2 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) v[k] = p[i + k * stride];
#pragma unroll
    for (int k = 0; k < U; k++) acc ^= v[k].x ^ v[k].y;
  }
  for (; i < n; i += stride) acc ^= p[i].x;
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  uint32_t *d_out;
  CK(hipMalloc(&d_out, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t sizes_mb[] = {14, 28, 56, 112, 168, 224, 256, 320, 512, 1024, 4096};
  const int grids[] = {256, 512, 1024, 2048, 4096};
  printf("# buffer re-read %d times back to back; GB/s by buffer size (MB) and workgroups of 256 threads, 8 x 16 B in flight per thread\n", reps);
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20, n = bytes / 16;
    uint4 *d;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    printf("%5zu MB:", mb);
    for (int g : grids) {
      hipLaunchKernelGGL(k_read<8>, dim3(g), dim3(256), 0, 0, d, n, d_out);  // warm
      CK(hipDeviceSynchronize());
      const int r = mb >= 1024 ? reps / 8 + 1 : reps;
      CK(hipEventRecord(e0));
      for (int k = 0; k < r; k++) hipLaunchKernelGGL(k_read<8>, dim3(g), dim3(256), 0, 0, d, n, d_out);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  %4d wg %7.0f", g, (double)bytes * r / (ms * 1e-3) * 1e-9);
    }
    printf("\n");
    if (mb == 56 || mb == 4096) {
      printf("%5zu MB, 8-byte loads, 16 in flight per thread:", mb);
      for (int g : grids) {
        hipLaunchKernelGGL(k_read8<16>, dim3(g), dim3(256), 0, 0, (const uint2 *)d, bytes / 8, d_out);
        CK(hipDeviceSynchronize());
        const int r = mb >= 1024 ? reps / 8 + 1 : reps;
        CK(hipEventRecord(e0));
        for (int k = 0; k < r; k++) hipLaunchKernelGGL(k_read8<16>, dim3(g), dim3(256), 0, 0, (const uint2 *)d, bytes / 8, d_out);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %4d wg %7.0f", g, (double)bytes * r / (ms * 1e-3) * 1e-9);
      }
      printf("\n");
    }
    CK(hipFree(d));
  }
  return 0;
}
