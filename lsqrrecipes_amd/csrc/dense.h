// dense.h -- DenseLinearEquationSystemParametersEstimator<double,n> on the device.
//   AugmentedRow<double,n> = n+1 doubles (aValues[n], bValue), DenseLinear...Estimator.h:133-134
//   n is a template argument in the reference; the C ABI takes it at run time (1..64) and the
//   kernels are instantiated for the padded sizes NR in {8,16,32,64}: the padding terms are
//   0.0*0.0 products, which leave the reference's running sum bit-identical.
#pragma once
#include <hip/hip_runtime.h>

#include "dense_model.h"
#include "models.h"
#include "wave_linalg.h"

namespace lsqr {

// K3 dense: consensus mask of one model.  The generic k_mask gives every lane its own 520-byte row,
// i.e. 64 cache lines per load instruction (measured 1.3 TB/s); here each wave stages 32 rows in LDS
// with coalesced loads (pitch n+1 padded to odd: conflict-free row reads) and lane r < 32 evaluates row
// r with the reference's running sum (bit-identical to DenseModel::agree).
template <int NR>
__global__ __launch_bounds__(256) void k_mask_dense(const double *__restrict__ data, size_t stride,
                                                    size_t begin, size_t end, int n,
                                                    const double *__restrict__ par, double delta,
                                                    uint8_t *__restrict__ mask,
                                                    unsigned long long *__restrict__ counter) {
  constexpr int ROWS = 32;
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pitch = (n + 1) | 1;
  double *tile = sm + (size_t)wave * ROWS * pitch;
  double x[NR];
#pragma unroll
  for (int i = 0; i < NR; i++) x[i] = i < n ? par[i] : 0.0;
  uint32_t local = 0;
  const size_t nw = (size_t)gridDim.x * 4;
  for (size_t base = begin + ((size_t)blockIdx.x * 4 + wave) * ROWS; base < end; base += nw * ROWS) {
    const int rows = (int)(end - base < (size_t)ROWS ? end - base : (size_t)ROWS);
    const int total = rows * (n + 1);
    // stride == n + 1 doubles is the tight layout; any stride is handled row by row.  Eight loads per lane
    // are in flight before the first LDS store (two waves per SIMD fit beside the 66 KB tiles: nothing else
    // hides the HBM latency); (r, c) advance incrementally instead of dividing per element
    {
      constexpr int U = 8;
      const int w = n + 1;
      int r = lane / w, c = lane - r * w;
      const int dr = 64 / w, dc = 64 - dr * w;  // one step of 64 elements
      for (int idx = lane; idx < total; idx += 64 * U) {
        double v[U];
        int rr[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          rr[u] = r, cc[u] = c;
          v[u] = idx + 64 * u < total ? data[(base + r) * stride + c] : 0.0;
          r += dr, c += dc;
          if (c >= w) c -= w, r++;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
          if (idx + 64 * u < total) tile[rr[u] * pitch + cc[u]] = v[u];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < rows) {
      const double *row = tile + lane * pitch;
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < NR; i++) sum += (i < n ? row[i] : 0.0) * x[i];
      sum -= row[n];
      const bool a = fabs(sum) < delta;
      mask[base + lane] = a ? 1 : 0;
      local += a ? 1u : 0u;
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if (lane == 0 && local) atomicAdd(counter, (unsigned long long)local);
}

// K1 dense, fast path (r04): FOUR hypotheses per workgroup, one wave each, the n x n system in the wave's own LDS
// area, wave_gepp_solve (wave_linalg.h) -- bit-identical to block_gepp_solve, no workgroup barriers: 0.30 -> ~0.05 ms
// per 1024 hypotheses at n = 64.  A system the elimination refuses (a pivot below 1e-8 max|A|: near the rank
// decision) is marked valid[h] = 2 and k_estimate_dense, launched behind this kernel with `only_marked`, takes it
// through the SVD pseudo-inverse exactly as before.
__global__ __launch_bounds__(256) void k_estimate_dense_w4(const double *__restrict__ data, size_t stride, size_t nobs,
                                                          const uint32_t *__restrict__ subsets, uint32_t H, int n,
                                                          int sp_stride, double *__restrict__ hparams,
                                                          uint8_t *__restrict__ valid) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t h = blockIdx.x * (blockDim.x >> 6) + wave;   // blockDim.x / 64 systems per workgroup
  if (h >= H) return;                              // wave-uniform; no workgroup barrier below
  const int lda = n | 1;
  double *A = sm + (size_t)wave * (n * lda + 2 * n), *b = A + n * lda, *x = b + n;
  bool in_range = true;
  for (int l = 0; l < n; l++) {                    // lane = column: one coalesced read of a row's n doubles
    size_t i = subsets[(size_t)h * n + l];
    if (i >= nobs) {
      in_range = false;
      i = 0;
    }
    if (lane < n) A[lane * lda + l] = data[i * stride + lane];
    if (lane == 0) b[l] = data[i * stride + n];
  }
  __builtin_amdgcn_wave_barrier();
  const bool solved = wave_gepp_solve(n, A, lda, b, x);
  __builtin_amdgcn_wave_barrier();
  const bool ok = solved && in_range;
  const double qnan = __builtin_nan("");
  for (int j = lane; j < sp_stride; j += 64) hparams[(size_t)h * sp_stride + j] = j < n ? (ok ? x[j] : qnan) : 0.0;
  if (lane == 0) valid[h] = ok ? 1 : (in_range ? 2 : 0);   // 2: to the SVD path; out-of-range subset: invalid either way
}

// K1 dense at n = 64 (r04, default): one wave per hypothesis, the 64 x 64 system IN REGISTERS -- lane = row -- and
// wave_gepp_solve_reg64 (wave_linalg.h): the pivot row travels by v_readlane, no LDS round trip per column.  Results
// bit-identical to k_estimate_dense_w4 / block_gepp_solve; what the elimination refuses is marked valid[h] = 2 for the
// SVD path exactly as there.
__global__ __launch_bounds__(64) void k_estimate_dense_r64(const double *__restrict__ data, size_t stride, size_t nobs,
                                                           const uint32_t *__restrict__ subsets, uint32_t H, int sp_stride,
                                                           double *__restrict__ hparams, uint8_t *__restrict__ valid) {
  const int lane = threadIdx.x;
  const uint32_t h = blockIdx.x;
  if (h >= H) return;
  size_t i = subsets[(size_t)h * 64 + lane];
  const bool in_range = __ballot(i >= nobs) == 0;
  if (i >= nobs) i = 0;
  double a[64], xv;
  const double *row = data + i * stride;
#pragma unroll
  for (int c = 0; c < 64; c++) a[c] = row[c];
  const bool solved = wave_gepp_solve_reg64(a, row[64], xv);
  const bool ok = solved && in_range;
  const double qnan = __builtin_nan("");
  for (int j = lane; j < sp_stride; j += 64) hparams[(size_t)h * sp_stride + j] = j < 64 ? (ok ? xv : qnan) : 0.0;
  if (lane == 0) valid[h] = ok ? 1 : (in_range ? 2 : 0);
}

// K1 dense: one wave per hypothesis; n x n system in LDS; x = pinv(A) b, singular if any
// sigma <= EPS (DenseLinear...Estimator.hxx:17-49)
__global__ __launch_bounds__(256) void k_estimate_dense(const double *__restrict__ data,
                                                       size_t stride, size_t nobs,
                                                       const uint32_t *__restrict__ subsets,
                                                       uint32_t H, int n, int sp_stride, int fast,
                                                       double *__restrict__ hparams,
                                                       uint8_t *__restrict__ valid, int only_marked) {
  extern __shared__ double sm[];
  __shared__ int s_bad;
  const int lane = threadIdx.x;
  const uint32_t h = blockIdx.x;
  if (only_marked && valid[h] != 2) return;      // (behind k_estimate_dense_w4: the systems its elimination refused)
  const int lda = n | 1;
  double *A = sm, *V = A + n * lda, *b = V + n * lda, *cw = b + n, *x = cw + n;
  bool in_range = true;
  if (lane == 0) s_bad = 0;
  auto load_system = [&]() {
    for (int idx = lane; idx < n * n; idx += 256) {
      int l = idx / n, c = idx % n;
      size_t i = subsets[(size_t)h * n + l];
      if (i >= nobs) {
        in_range = false;
        i = 0;
      }
      A[c * lda + l] = data[i * stride + c];
    }
    for (int l = lane; l < n; l += 256) {
      size_t i = subsets[(size_t)h * n + l];
      if (i >= nobs) i = 0;
      b[l] = data[i * stride + n];
    }
  };
  load_system();
  __syncthreads();
  if (!in_range) s_bad = 1;
  // well-conditioned systems (all of them on ordinary data): elimination with partial pivoting;
  // anything near the rank decision goes through the SVD pseudo-inverse as the reference does
  int rank = n;
  if (!fast || !block_gepp_solve<256>(n, A, lda, b, x)) {
    __syncthreads();
    load_system();
    __syncthreads();
    rank = block_pinv_solve<256>(n, n, A, lda, V, lda, b, kEPS, 0.0, x, cw);
  }
  __syncthreads();
  bool ok = (rank == n) && !s_bad;
  const double qnan = __builtin_nan("");
  for (int j = lane; j < sp_stride; j += 256)
    hparams[(size_t)h * sp_stride + j] = j < n ? (ok ? x[j] : qnan) : 0.0;
  if (lane == 0) valid[h] = ok ? 1 : 0;
}


// K2 dense on the matrix cores.  The residuals of a block of rows against a block of hypotheses are
// a GEMM; v_mfma_f64_16x16x4 evaluates it with fused multiply-adds, which round differently from the
// reference's separate multiply and add -- so the MFMA result is used as a FILTER with a rigorous
// band, exactly like the fp32 pre-filter of the point models:
//   both the reference's running sum and any fma-ordered evaluation of a.x - b are within
//   g = gamma_{n+1} (sum |a_i||x_i| + |b|) <= gamma_{n+1} Amax (||x||_1 + 1) of the exact value
//   (gamma_k = k u / (1 - k u), u = 2^-53, Amax = max |entry| of the uploaded rows), hence within
//   E_h = 2.02 g of each other.  |res| < delta - E_h => agrees; |res| >= delta + E_h => does not;
//   pairs in between (probability ~1e-13) go to a worklist and are decided by the exact formula
//   (k_dense_recheck).  Votes are therefore bit-identical to k_scan<DenseModel>.
// Tiling: workgroup = 64 rows x 64 hypotheses, both staged in LDS with a 66-double pitch (the
// 16 x 4 fragment reads of a wave hit 32 distinct 8-byte slots); wave w owns rows 16w..16w+15 and
// four 16 x 16 accumulator tiles; each lane counts the inliers of its own hypothesis columns.
constexpr int kDmPitch = 66;
constexpr unsigned kAmbCap = 1u << 22;  // worklist entries (the fp32 filter sends ~1e-4 of the pairs there)
constexpr int kDmHypChunk = 2048;  // hypotheses per launch (LDS vote counters: 8 KiB)

// per hypothesis thresholds {delta - E_h, delta + E_h} of the MFMA filter ({-1,-1}: never agrees)
__global__ void k_dense_thresholds(const double *__restrict__ sp, uint32_t H, int n, int nr,
                                   double delta, double amax, double *__restrict__ thr) {
  uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  double l1 = 0.0;
  for (int k = 0; k < n; k++) l1 += fabs(sp[(size_t)h * nr + k]);
  const double u = 1.1102230246251565e-16, kg = (double)(n + 1) * u;
  const double E = 2.02 * (kg / (1.0 - kg)) * amax * (l1 + 1.0);
  bool live = l1 == l1 && l1 < 1e300;  // NaN / non-finite model: never agrees
  thr[2 * (size_t)h] = live ? delta - E : -1.0;
  thr[2 * (size_t)h + 1] = live ? delta + E : -1.0;
}

// Each workgroup owns a range of rows: a 64-row tile stays in LDS while all hypothesis blocks
// (64 each, L2 resident) stream past it, so the rows are read from HBM once per launch.
template <int NR>
__global__ __launch_bounds__(256) void k_scan_dense_mfma(
    const double *__restrict__ data, size_t stride, size_t m, size_t rows_per_block,
    const double *__restrict__ sp, const double *__restrict__ thr, uint32_t H, int n,
    uint32_t *__restrict__ votes, unsigned long long *__restrict__ amb_list,
    unsigned int *__restrict__ amb_count, uint32_t hyp_base) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  extern __shared__ double sm[];
  double *At = sm;                      // 64 rows x pitch
  double *Bt = At + 64 * kDmPitch;      // 64 hypotheses x pitch
  double *bv = Bt + 64 * kDmPitch;      // 64 right-hand sides
  double *tin = bv + 64, *tout = tin + 64;  // thresholds of the current hypothesis block
  uint32_t *s_cnt = (uint32_t *)(tout + 64);  // H counters
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, k4 = lane >> 4;
  const uint32_t nhb = (H + 63) / 64;
  for (uint32_t h = tid; h < H; h += 256) s_cnt[h] = 0;
  size_t lo = (size_t)blockIdx.x * rows_per_block;
  size_t hi = lo + rows_per_block < m ? lo + rows_per_block : m;
  // hypothesis tiles are fetched into registers one block ahead of their use
  double nb[16], nti = -1.0, nto = -1.0;
  auto fetch_hyp = [&](uint32_t hb) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      int idx = tid + q * 256, hh = idx >> 6, kk = idx & 63;
      uint32_t h = hb * 64 + hh;
      nb[q] = (h < H && kk < n) ? sp[(size_t)h * NR + kk] : 0.0;
    }
    if (tid < 64) {
      uint32_t h = hb * 64 + tid;
      nti = h < H ? thr[2 * (size_t)h] : -1.0;
      nto = h < H ? thr[2 * (size_t)h + 1] : -1.0;
    }
  };
  if (lo < hi) fetch_hyp(0);
  for (size_t base = lo; base < hi; base += 64) {
    __syncthreads();
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      int r = idx >> 6, kk = idx & 63;
      size_t row = base + r;
      At[r * kDmPitch + kk] = (row < hi && kk < n) ? data[row * stride + kk] : 0.0;
    }
    if (tid < 64) {
      size_t row = base + tid;
      bv[tid] = row < hi ? data[row * stride + n] : __builtin_nan("");  // NaN: row never counts
    }
    for (uint32_t hb = 0; hb < nhb; hb++) {
      __syncthreads();  // previous block's fragment reads are done
#pragma unroll
      for (int q = 0; q < 16; q++) {
        int idx = tid + q * 256;
        Bt[(idx >> 6) * kDmPitch + (idx & 63)] = nb[q];
      }
      if (tid < 64) {
        tin[tid] = nti;
        tout[tid] = nto;
      }
      __syncthreads();
      fetch_hyp(hb + 1 < nhb ? hb + 1 : 0);  // next block (or block 0 for the next row tile)
      d4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
      const double *ap = At + (wave * 16 + c16) * kDmPitch + k4;
      const double *bp = Bt + c16 * kDmPitch + k4;
#pragma unroll 4
      for (int s4 = 0; s4 < 64; s4 += 4) {
        double a = ap[s4];
#pragma unroll
        for (int t = 0; t < 4; t++)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[t * 16 * kDmPitch + s4], acc[t], 0, 0, 0);
      }
      // D layout: column (hypothesis) = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int hh = t * 16 + c16;
        const double ti = tin[hh], to = tout[hh];
        uint32_t c = 0;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int r = wave * 16 + k4 + 4 * rg;
          const double res = fabs(acc[t][rg] - bv[r]);
          c += res < ti ? 1u : 0u;
          if (res >= ti && res < to) {  // ambiguous: decided exactly by k_dense_recheck
            unsigned slot = atomicAdd(amb_count, 1u);
            if (slot < kAmbCap)
              amb_list[slot] = ((unsigned long long)(base + r) << 32) |
                               (unsigned long long)(hyp_base + hb * 64 + hh);
          }
        }
        c += __shfl_xor(c, 16);  // lanes l, l^16, l^32, l^48 hold the same hypothesis column
        c += __shfl_xor(c, 32);
        if (k4 == 0 && c) atomicAdd(&s_cnt[hb * 64 + hh], c);
      }
    }
  }
  __syncthreads();
  for (uint32_t h = tid; h < H; h += 256) {
    uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[h], c);
  }
}

// Second arrangement of the same filter (default): wave w owns hypotheses 16w..16w+15 of the current
// 64-hypothesis block and ALL 64 rows of the tile (four 16 x 16 accumulators, one per row group).  Its B
// fragments -- 16 doubles per lane -- come straight from global memory / L2 into registers, one block
// ahead, so the hypothesis loop has no LDS writes and no barriers (the row tile in LDS is read-only
// there); LDS per workgroup drops to 34 KB + counters: three workgroups per CU.
template <int NR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_scan_dense_mfma2(
    const double *__restrict__ data, size_t stride, size_t m, size_t rows_per_block,
    const double *__restrict__ sp, const double *__restrict__ thr, uint32_t H, int n,
    uint32_t *__restrict__ votes, unsigned long long *__restrict__ amb_list,
    unsigned int *__restrict__ amb_count, uint32_t hyp_base) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  static_assert(NR == 64, "fragment bookkeeping below assumes 64 padded unknowns");
  extern __shared__ double sm[];
  double *At = sm;                  // 64 rows x pitch
  double *bv = At + 64 * kDmPitch;  // 64 right-hand sides
  uint32_t *s_cnt = (uint32_t *)(bv + 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, k4 = lane >> 4;
  const uint32_t nhb = (H + 63) / 64;
  for (uint32_t h = tid; h < H; h += 256) s_cnt[h] = 0;
  size_t lo = (size_t)blockIdx.x * rows_per_block;
  size_t hi = lo + rows_per_block < m ? lo + rows_per_block : m;
  double nb[16], nti = -1.0, nto = -1.0;
  auto fetch_hyp = [&](uint32_t hb) {
    const uint32_t h = hb * 64 + wave * 16 + c16;
    const double *row = sp + (size_t)(h < H ? h : 0) * NR;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int kk = 4 * q + k4;
      nb[q] = (h < H && kk < n) ? row[kk] : 0.0;
    }
    nti = h < H ? thr[2 * (size_t)h] : -1.0;
    nto = h < H ? thr[2 * (size_t)h + 1] : -1.0;
  };
  if (lo < hi) fetch_hyp(0);
  for (size_t base = lo; base < hi; base += 64) {
    __syncthreads();  // the previous tile's fragment reads are done
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      int r = idx >> 6, kk = idx & 63;
      size_t row = base + r;
      At[r * kDmPitch + kk] = (row < hi && kk < n) ? data[row * stride + kk] : 0.0;
    }
    if (tid < 64) {
      size_t row = base + tid;
      bv[tid] = row < hi ? data[row * stride + n] : __builtin_nan("");  // NaN: row never counts
    }
    __syncthreads();
    for (uint32_t hb = 0; hb < nhb; hb++) {
      double b[16];
#pragma unroll
      for (int q = 0; q < 16; q++) b[q] = nb[q];
      const double ti = nti, to = nto;
      fetch_hyp(hb + 1 < nhb ? hb + 1 : 0);  // next block (or block 0 for the next row tile)
      d4 acc[4];
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
      const double *ap = At + c16 * kDmPitch + k4;
#pragma unroll
      for (int q = 0; q < 16; q++) {
#pragma unroll
        for (int t = 0; t < 4; t++)
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[t * 16 * kDmPitch + 4 * q], b[q], acc[t], 0, 0, 0);
      }
      // D layout: column (hypothesis) = lane & 15, row = (lane >> 4) + 4 * reg (+ 16 * row group)
      uint32_t c = 0;
#pragma unroll
      for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int r = t * 16 + k4 + 4 * rg;
          const double res = fabs(acc[t][rg] - bv[r]);
          c += res < ti ? 1u : 0u;
          if (res >= ti && res < to) {  // ambiguous: decided exactly by k_dense_recheck
            unsigned slot = atomicAdd(amb_count, 1u);
            if (slot < kAmbCap)
              amb_list[slot] = ((unsigned long long)(base + r) << 32) |
                               (unsigned long long)(hyp_base + hb * 64 + wave * 16 + c16);
          }
        }
      }
      c += __shfl_xor(c, 16);  // lanes l, l^16, l^32, l^48 hold the same hypothesis column
      c += __shfl_xor(c, 32);
      if (k4 == 0 && c) atomicAdd(&s_cnt[hb * 64 + wave * 16 + c16], c);
    }
  }
  __syncthreads();
  for (uint32_t h = tid; h < H; h += 256) {
    uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[h], c);
  }
}

// Third arrangement (default at n > 32): the same tiling as k_scan_dense_mfma2 in SINGLE precision
// (v_mfma_f32_16x16x4_f32: twice the fp64 matrix rate, half the LDS bytes).  The filter's band widens from ~1e-13
// to ~1e-4 of the pairs, which is still a worklist (k_dense_recheck decides them with the exact fp64 formula), so
// the votes stay bit-identical.  Error of the fp32 evaluation r32 = fl(fl(sum fl(a_i) fl(x_i)) - fl(b)) against
// the exact residual, u = 2^-24: inputs 2u |a_i x_i| each, n fused accumulations gamma_n sum|a_i x_i|, the
// subtraction and fl(b): 2u (|S| + |b|)  =>  |r32 - res| <= (n + 4) u (1 + 1e-3) (Amax ||x||_1 + Bmax), Amax / Bmax
// the largest |coefficient| / |right-hand side| of the upload (taken separately: a random minimal solve has
// ||x||_1 in the thousands, and pricing it with the right-hand sides' magnitude put 1.5 % of the pairs in the band);
// the reference's fp64 running sum is within 1e-14 of that scale of res.  k_dense_thresholds32 takes
// E_h = 1.5 (n + 4) u (Amax ||x||_1 + Bmax) + 1e-30 (denormal products may be flushed) and rounds the two
// thresholds outwards to fp32.
__global__ void k_dense_thresholds32(const double *__restrict__ sp, uint32_t H, int n, int nr, double delta,
                                     double amax, double bmax, float *__restrict__ thr,
                                     float *__restrict__ sp32) {
  uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  double l1 = 0.0;
  for (int k = 0; k < nr; k++) {
    const double v = k < n ? sp[(size_t)h * nr + k] : 0.0;
    l1 += fabs(v);
    // FRAGMENT order: unknown k = 4 q + k4 is stored at k4 * 16 + q, so that the 16 values a lane of the scan
    // needs (its k4, q = 0..15) are 64 contiguous bytes = four 16-byte loads instead of sixteen 4-byte ones
    sp32[(size_t)h * nr + (k & 3) * (nr / 4) + (k >> 2)] = (float)v;
  }
  const double u = 5.9604644775390625e-08;
  const double E = 1.5 * (double)(n + 4) * u * (amax * l1 + bmax) + 1e-30;
  // fp32 must be able to hold the products: magnitudes beyond 1e15 (or a NaN model) never agree through the filter
  const bool live = l1 == l1 && l1 < 1e15 && amax < 1e15 && bmax < 1e15;
  float ti = (float)(delta - E), to = (float)(delta + E);
  if ((double)ti > delta - E) ti = nextafterf(ti, -INFINITY);
  if ((double)to < delta + E) to = nextafterf(to, INFINITY);
  thr[2 * (size_t)h] = live ? ti : -1.0f;
  thr[2 * (size_t)h + 1] = live ? to : (l1 == l1 ? INFINITY : -1.0f);  // not live but finite: everything re-checked
}

constexpr int kDmPitch32 = 68;  // floats: the 16 x 4 fragment reads of a wave hit 64 distinct banks
// The fp32 filter with the hypothesis fragments prefetched THROUGH LDS, two blocks ahead.  (Its predecessor kept the
// fragments in registers -- 64 floats of A per lane read once per tile, two accumulator sets so that the counting of
// block hb runs beside the 64 matrix instructions of block hb + 1; removed in r03, 3.27 against 2.99 ms.)  There a
// wave asks for the 64 bytes per lane of hypothesis block hb + 1 while it works on block hb -- 64 matrix instructions,
// 2048 cycles -- and an L2 hit on another XCD's slice takes about as long: with two waves per SIMD nothing else hides
// the rest of the wait, and there are no registers for a second block in flight (254 of 256).  global_load_lds writes
// the rows straight into a per-wave ring in LDS (no staging registers, no waiting at issue): the loads for block
// it + 2 are issued when block it is read out of its slot, so every fetch has two blocks' worth of matrix work
// (~1.7 us) to land.  The thresholds of the whole batch sit in LDS as well.
template <int NR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_scan_dense_mfma32r(
    const double *__restrict__ data, size_t stride, size_t row_begin, size_t row_end, size_t rows_per_block,
    const float *__restrict__ sp32, const float *__restrict__ thr, uint32_t H, int n,
    uint32_t *__restrict__ votes, unsigned long long *__restrict__ amb_list,
    unsigned int *__restrict__ amb_counts, uint32_t seg_cap, uint32_t hyp_base,
    const uint32_t *__restrict__ h_dev, const uint32_t *__restrict__ sel, const uint32_t *__restrict__ range_dev) {
  // (range_dev: the row range comes from device memory -- planned by k_ee_plan -- and is cut into gridDim.x equal runs
  // of whole tiles here)
  if (range_dev) {
    row_begin = range_dev[0];
    row_end = range_dev[1];
    if (row_begin >= row_end) return;  // workgroup-uniform
    const size_t tiles = (row_end - row_begin + 63) / 64;
    rows_per_block = (tiles + gridDim.x - 1) / gridDim.x * 64;
  }
  // Rows [row_begin, row_end) against hypotheses [hyp_base, hyp_base + H) of a batch -- the context's batch itself
  // (sel == null: hypothesis index = position) or a compacted selection of it (early exit, earlyexit.h: sel[] maps
  // positions to hypothesis indices, *h_dev is the selection's size; sp32 / thr are the selection's compact rows,
  // already offset to hyp_base).  Votes and worklist entries carry hypothesis indices.
  typedef float f4 __attribute__((ext_vector_type(4)));
  static_assert(NR == 64, "fragment bookkeeping below assumes 64 padded unknowns");
  extern __shared__ float smf[];
  if (h_dev) {
    const uint32_t ht = *h_dev, hd = ht > hyp_base ? ht - hyp_base : 0u;
    H = hd < H ? hd : H;
    if (H == 0) return;  // workgroup-uniform
  }
  const uint32_t nhb = (H + 63) / 64, nhb2 = (nhb + 1) & ~1u;  // blocks past the batch: thresholds that never pass
  float *ring = smf;                              // 4 waves x 2 slots x 4 chunks of 1 KiB (16-byte aligned: first)
  float *At = ring + 4 * 2 * 1024;                // 64 rows x pitch
  float *bv = At + 64 * kDmPitch32;               // 64 right-hand sides
  float *thl = bv + 64;                           // (t_in, t_out) of every hypothesis of the launch
  uint32_t *s_cnt = (uint32_t *)(thl + 2 * 64 * nhb2);
  uint32_t *s_amb = s_cnt + H;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, k4 = lane >> 4;
  for (uint32_t h = tid; h < H; h += 256) s_cnt[h] = 0;
  for (uint32_t h = tid; h < 64 * nhb2; h += 256) {
    thl[2 * h] = h < H ? thr[2 * (size_t)h] : -1.0f;
    thl[2 * h + 1] = h < H ? thr[2 * (size_t)h + 1] : -1.0f;
  }
  if (tid == 0) *s_amb = amb_counts[blockIdx.x];
  size_t lo = row_begin + (size_t)blockIdx.x * rows_per_block;
  size_t hi = lo + rows_per_block < row_end ? lo + rows_per_block : row_end;
  if (lo > hi) lo = hi;
  auto hyp_id = [&](uint32_t h) -> uint32_t { return sel ? sel[hyp_base + h] : hyp_base + h; };
  // hypothesis block hb -> slot: the rows of my 16 hypotheses in fragment order, 64 bytes per lane, as four
  // 16-byte pieces; piece j of all lanes lands contiguously (lane * 16 bytes) in chunk j of the slot
  auto issue = [&](uint32_t hb, uint32_t slot) {
    const uint32_t h = hb * 64 + wave * 16 + c16;
    const float *g = sp32 + (size_t)(h < H ? h : 0) * NR + k4 * 16;
    float *dst = ring + ((wave * 2 + slot) * 4) * 256;
#pragma unroll
    for (int j = 0; j < 4; j++)
      __builtin_amdgcn_global_load_lds(g + 4 * j, (__attribute__((address_space(3))) void *)(dst + j * 256), 16, 0, 0);
  };
  uint32_t it = 0;  // hypothesis blocks this wave has started, over all tiles (slot = it & 1, block = it % nhb2)
  // The NEXT tile of rows travels in registers while the current one is multiplied (the ring freed the registers the
  // hypothesis fragments used to occupy): 16 coefficients and one right-hand side per thread, 17 loads per wave.
  double pre[16], preb;
  auto load_tile = [&](size_t base) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int idx = tid + 256 * i, r = idx >> 6, kk = idx & 63;
      const size_t row = base + r;
      pre[i] = data[(row < hi ? row : lo) * stride + (kk < n ? kk : 0)];
    }
    const size_t rowb = base + (tid & 63);
    preb = data[(rowb < hi ? rowb : lo) * stride + n];
  };
  if (lo < hi) {
    issue(0, 0);
    issue(1 % nhb2, 1);
    load_tile(lo);
  }
  for (size_t base = lo; base < hi; base += 64) {
    __syncthreads();  // the previous tile's fragment reads are done
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int idx = tid + 256 * i, r = idx >> 6, kk = idx & 63;
      At[r * kDmPitch32 + kk] = (base + r < hi && kk < n) ? (float)pre[i] : 0.0f;
    }
    if (tid < 64) bv[tid] = base + tid < hi ? (float)preb : __builtin_nanf("");  // NaN: row never counts
    const bool more = base + 64 < hi;  // workgroup-uniform
    if (more) load_tile(base + 64);    // in flight during the whole tile
    __syncthreads();
    float a[4][16];
    {
      const float *ap = At + c16 * kDmPitch32 + k4;
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int q = 0; q < 16; q++) a[t][q] = ap[t * 16 * kDmPitch32 + 4 * q];
    }
    // (the 16 right-hand sides of a lane are read from LDS where they are used: their registers carry the next tile)
    // block `hb` out of its slot into registers, the slot handed to block it + 2, then the 64 matrix instructions
    auto start_block = [&](f4(&acc)[4], uint32_t hb, float &ti, float &to) {
      const uint32_t slot = it & 1u;
      // My slot's four loads have landed.  Loads complete in order: younger than them are the four of block it + 1
      // and, for the first two blocks of a tile, the 17 loads of the next tile issued at the top of this one.
      if (more && hb < 2)
        asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      float b[16];
      const f4 *src = (const f4 *)(ring + ((wave * 2 + slot) * 4) * 256) + lane;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f4 v = src[j * 64];
        b[4 * j] = v.x, b[4 * j + 1] = v.y, b[4 * j + 2] = v.z, b[4 * j + 3] = v.w;
      }
      const uint32_t h = hb * 64 + wave * 16 + c16;
      ti = thl[2 * h];
      to = thl[2 * h + 1];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot has been read: it may be overwritten
      issue((hb + 2) % nhb2, slot);
      it++;
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < 16; q++) {
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][q], b[q], acc[t], 0, 0, 0);
      }
    };
    // The same with the branch-free part of the OTHER accumulator set's counting in the same basic block, and the
    // scheduler told to alternate: a wave issues its instructions in order, so 64 matrix instructions in a row keep
    // its own ~100 counting instructions out of their shadow (each occupies the matrix pipe for 32 cycles, the
    // vector issue port for 4) -- only the second wave of the SIMD filled that gap.
    auto start_and_count = [&](f4(&acc)[4], uint32_t hb, float &ti, float &to, const f4(&old)[4], float oti,
                               float oto, uint32_t &c_out, uint32_t &may_out) {
      const uint32_t slot = it & 1u;
      if (more && hb < 2)
        asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      float b[16];
      const f4 *src = (const f4 *)(ring + ((wave * 2 + slot) * 4) * 256) + lane;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f4 v = src[j * 64];
        b[4 * j] = v.x, b[4 * j + 1] = v.y, b[4 * j + 2] = v.z, b[4 * j + 3] = v.w;
      }
      const uint32_t h = hb * 64 + wave * 16 + c16;
      ti = thl[2 * h];
      to = thl[2 * h + 1];
      f4 rhs[4];
#pragma unroll
      for (int t = 0; t < 4; t++) rhs[t] = *(const f4 *)(bv + t * 16 + 4 * k4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot has been read: it may be overwritten
      issue((hb + 2) % nhb2, slot);
      it++;
#pragma unroll
      for (int t = 0; t < 4; t++) acc[t] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
      uint32_t c = 0, may = 0;
#pragma unroll
      for (int q = 0; q < 16; q++) {
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][q], b[q], acc[t], 0, 0, 0);
        {  // one of the 16 residuals of the other set per four matrix instructions
          const float res = __builtin_fabsf(old[q >> 2][q & 3] - rhs[q >> 2][q & 3]);
          c += res < oti ? 1u : 0u;
          may += res < oto ? 1u : 0u;
        }
      }
      // (groups of 4 + 6; strict 1 : 1 alternation was measured slower than no interleaving at all: 3.35 against
      // 3.20 ms; this pattern 3.11 ms)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 matrix instructions
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);  // the ~6 vector instructions of one residual
      }
      c_out = c;
      may_out = may;
    };
    auto count_block = [&](const f4(&acc)[4], float ti, float to, uint32_t hb) {
      uint32_t c = 0, may = 0;
      f4 rhs[4];
#pragma unroll
      for (int t = 0; t < 4; t++) rhs[t] = *(const f4 *)(bv + t * 16 + 4 * k4);
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const float res = __builtin_fabsf(acc[t][rg] - rhs[t][rg]);
          c += res < ti ? 1u : 0u;
          may += res < to ? 1u : 0u;
        }
      if (may != c) {  // rare: some pair sits in the band -> worklist, decided exactly by k_dense_recheck_seg
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            const float res = __builtin_fabsf(acc[t][rg] - rhs[t][rg]);
            if (res >= ti && res < to) {
              const unsigned slot = atomicAdd(s_amb, 1u);
              if (slot < seg_cap)
                amb_list[(size_t)blockIdx.x * seg_cap + slot] =
                    ((unsigned long long)(base + t * 16 + 4 * k4 + rg) << 32) |
                    (unsigned long long)hyp_id(hb * 64 + wave * 16 + c16);
            }
          }
      }
      c += __shfl_xor(c, 16);  // lanes l, l^16, l^32, l^48 hold the same hypothesis column
      c += __shfl_xor(c, 32);
      if (k4 == 0 && c) atomicAdd(&s_cnt[hb * 64 + wave * 16 + c16], c);
    };
    // what is left of a block's counting once c / may are known: the rare worklist appends, the fold over the four
    // lanes of a hypothesis column, one LDS atomic
    auto count_tail = [&](const f4(&acc)[4], float ti, float to, uint32_t hb, uint32_t c, uint32_t may) {
      if (may != c) {
        f4 rhs[4];
#pragma unroll
        for (int t = 0; t < 4; t++) rhs[t] = *(const f4 *)(bv + t * 16 + 4 * k4);
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            const float res = __builtin_fabsf(acc[t][rg] - rhs[t][rg]);
            if (res >= ti && res < to) {
              const unsigned slot = atomicAdd(s_amb, 1u);
              if (slot < seg_cap)
                amb_list[(size_t)blockIdx.x * seg_cap + slot] =
                    ((unsigned long long)(base + t * 16 + 4 * k4 + rg) << 32) |
                    (unsigned long long)hyp_id(hb * 64 + wave * 16 + c16);
            }
          }
      }
      c += __shfl_xor(c, 16);
      c += __shfl_xor(c, 32);
      if (k4 == 0 && c) atomicAdd(&s_cnt[hb * 64 + wave * 16 + c16], c);
    };
    f4 accA[4], accB[4];
    float tiA, toA, tiB, toB;
    uint32_t cc = 0, cm = 0;
    start_block(accA, 0, tiA, toA);
    for (uint32_t hb = 0; hb < nhb2; hb += 2) {
      start_and_count(accB, hb + 1, tiB, toB, accA, tiA, toA, cc, cm);
      count_tail(accA, tiA, toA, hb, cc, cm);
      if (hb + 2 < nhb2) {
        start_and_count(accA, hb + 2, tiA, toA, accB, tiB, toB, cc, cm);
        count_tail(accB, tiB, toB, hb + 1, cc, cm);
      } else {
        count_block(accB, tiB, toB, hb + 1);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS write of mine may still be in flight when the LDS is freed
  __syncthreads();
  for (uint32_t h = tid; h < H; h += 256) {
    uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[hyp_id(h)], c);
  }
  if (tid == 0) amb_counts[blockIdx.x] = *s_amb;
}

// exact decision of the segmented worklist (one block per segment); out_max[0] = largest segment fill (overflow check)
template <int NR>
__global__ __launch_bounds__(256) void k_dense_recheck_seg(const double *__restrict__ data, size_t stride,
                                                           const double *__restrict__ sp, ModelConsts mc,
                                                           const unsigned long long *__restrict__ amb_list,
                                                           unsigned int *__restrict__ amb_counts,
                                                           uint32_t seg_cap, uint32_t *__restrict__ votes,
                                                           unsigned int *__restrict__ out_max) {
  typedef DenseModel<NR> M;
  // votes per workgroup first (us_h16.h: k_us_recheck_seg -- device-wide atomics on the few counters of the good
  // hypotheses are served one at a time), in a direct-mapped LDS table keyed by the hypothesis index
  constexpr unsigned kSlots = 1024;
  __shared__ uint32_t s_tag[kSlots], s_votes[kSlots];
  const unsigned filled = amb_counts[blockIdx.x];
  if (filled == 0) return;  // workgroup-uniform
  for (unsigned i = threadIdx.x; i < kSlots; i += 256) s_tag[i] = 0xFFFFFFFFu, s_votes[i] = 0;
  if (threadIdx.x == 0) atomicMax(out_max, filled);
  __syncthreads();
  const unsigned total = filled < seg_cap ? filled : seg_cap;
  for (unsigned e = threadIdx.x; e < total; e += 256) {
    unsigned long long v = amb_list[(size_t)blockIdx.x * seg_cap + e];
    size_t row = (size_t)(v >> 32);
    uint32_t h = (uint32_t)(v & 0xffffffffu);
    double x[M::REC];
    M::load(data + row * stride, mc, x);
    if (M::agree(sp + (size_t)h * NR, x, mc)) {
      const unsigned slot = h & (kSlots - 1);
      const uint32_t was = atomicCAS(&s_tag[slot], 0xFFFFFFFFu, h);
      if (was == 0xFFFFFFFFu || was == h)
        atomicAdd(&s_votes[slot], 1u);
      else
        atomicAdd(&votes[h], 1u);
    }
  }
  __syncthreads();  // every thread has read `filled`
  for (unsigned i = threadIdx.x; i < kSlots; i += 256)
    if (s_votes[i]) atomicAdd(&votes[s_tag[i]], s_votes[i]);
  if (threadIdx.x == 0) amb_counts[blockIdx.x] = 0;  // decided: the segment is free for the next chunk's appends
}

// exact decision of the pairs the MFMA filter could not classify
template <int NR>
__global__ __launch_bounds__(256) void k_dense_recheck(const double *__restrict__ data,
                                                       size_t stride, const double *__restrict__ sp,
                                                       ModelConsts mc,
                                                       const unsigned long long *__restrict__ amb_list,
                                                       const unsigned int *__restrict__ amb_count,
                                                       uint32_t *__restrict__ votes) {
  typedef DenseModel<NR> M;
  unsigned total = *amb_count;
  if (total > kAmbCap) total = kAmbCap;
  for (unsigned e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    unsigned long long v = amb_list[e];
    size_t row = (size_t)(v >> 32);
    uint32_t h = (uint32_t)(v & 0xffffffffu);
    double x[M::REC];
    M::load(data + row * stride, mc, x);
    if (M::agree(sp + (size_t)h * NR, x, mc)) atomicAdd(&votes[h], 1u);
  }
}

// K4 dense: upper triangle of sum z z^T, z = [a, b] (n+1 entries) -> normal equations A^T A,
// A^T b (+ b^T b, + count).  Rows are staged through LDS in tiles; each thread owns a fixed set
// of matrix entries; per-block partial sums are reduced later in a fixed order.
constexpr int kSyrkTile = 32;  // rows per LDS tile

__global__ __launch_bounds__(256) void k_syrk_dense(const double *__restrict__ data, size_t stride,
                                                    size_t begin, size_t end, size_t chunk, int n,
                                                    const uint8_t *__restrict__ mask, int use_mask,
                                                    int pstride, double *__restrict__ partials) {
  extern __shared__ double sm[];  // kSyrkTile * ld doubles + kSyrkTile flags
  const int nz = n + 1, ld = nz | 1;
  const int ne = nz * (nz + 1) / 2;
  unsigned char *flag = (unsigned char *)(sm + kSyrkTile * ld);
  constexpr int EPT = 9;  // ceil(65*66/2 / 256)
  int ei[EPT], ej[EPT];
  double acc[EPT];
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    int e = threadIdx.x + q * 256;
    acc[q] = 0.0;
    ei[q] = 0;
    ej[q] = 0;
    if (e < ne) {  // e -> (i,j), i <= j, row-major upper triangle
      int i = 0, rem = e;
      while (rem >= nz - i) {
        rem -= nz - i;
        i++;
      }
      ei[q] = i;
      ej[q] = i + rem;
    }
  }
  double cnt = 0.0;
  size_t lo = begin + (size_t)blockIdx.x * chunk;
  size_t hi = lo + chunk < end ? lo + chunk : end;
  for (size_t base = lo; base < hi; base += kSyrkTile) {
    int rows = (int)((hi - base) < (size_t)kSyrkTile ? (hi - base) : (size_t)kSyrkTile);
    __syncthreads();
    for (int idx = threadIdx.x; idx < rows * nz; idx += 256) {
      int r = idx / nz, c = idx % nz;
      sm[r * ld + c] = data[(base + r) * stride + c];
    }
    for (int r = threadIdx.x; r < rows; r += 256) flag[r] = use_mask ? mask[base + r] : 1;
    __syncthreads();
    for (int r = 0; r < rows; r++) {
      if (!flag[r]) continue;  // workgroup-uniform
      const double *z = sm + r * ld;
#pragma unroll
      for (int q = 0; q < EPT; q++) acc[q] = fma(z[ei[q]], z[ej[q]], acc[q]);
      cnt += 1.0;
    }
  }
  double *out = partials + (size_t)blockIdx.x * pstride;
#pragma unroll
  for (int q = 0; q < EPT; q++) {
    int e = threadIdx.x + q * 256;
    if (e < ne) out[e] = acc[q];
  }
  if (threadIdx.x == 0) out[ne] = cnt;
}

// K4 dense on the matrix cores: the same block of sums as k_syrk_dense with v_mfma_f64_16x16x4_f64.
// z = [a, b] is padded to 80 = 5 x 16 columns; for 4 rows at a time lane l holds z_k[16 b + (l & 15)],
// k = l >> 4, for b = 0..4 -- which is at once the A operand (A[i][k]) of block-row b and the B operand
// (B[k][j]) of block-column b, so the 15 upper 16x16 blocks of sum z z^T need 5 loads and 15 MFMAs per
// 4 rows and no LDS traffic in the loop.  (fp64 MFMA peaks at the vector FMA rate on MI355X; what it
// buys here is operand reuse.)  The four waves of a workgroup fold their accumulators in wave order
// through LDS, so the per-block partial is deterministic.
typedef double d4 __attribute__((ext_vector_type(4)));

// NB = number of 16-column blocks that hold anything (n + 1 <= 16 NB): the plane phantom's 32-column rows need two of
// the five -- 3 matrix instructions and 2 loads per four rows instead of 15 and 5 (until r05 the three idle blocks
// were loaded from column 0 and multiplied as zeros: 186 us per 1 M x 32 matrix, 1.4 TB/s), with PF groups in flight.
template <int VAR, int NB = 5, int PF = 4>
__global__ __launch_bounds__(256) void k_syrk_mfma(const double *__restrict__ data, size_t stride,
                                                   size_t begin, size_t end, size_t chunk, int n,
                                                   const uint8_t *__restrict__ mask, int use_mask,
                                                   int pstride, double *__restrict__ partials) {
  __shared__ double fold[15 * 256];
  __shared__ unsigned s_rows[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane >> 4, c16 = lane & 15;
  const int nz = n + 1;
  d4 acc[15];
#pragma unroll
  for (int t = 0; t < 15; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  unsigned rows_used = 0;
  size_t lo = begin + (size_t)blockIdx.x * chunk;
  size_t hi = lo + chunk < end ? lo + chunk : end;
  // software pipeline: the loads of the next 4-row group are issued before the 15 MFMAs of the
  // current one (one wave per SIMD: nothing else would cover the HBM latency)
  double z[PF][NB];  // PF groups in flight
  unsigned char mk[PF];
  // Loop-invariant column offsets (columns past the row are read from column 0 and zeroed below);
  // loads never depend on the mask byte (a dependent load would drain the whole ring with vmcnt(0)).
  int off[NB];
  bool colok[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    colok[b] = 16 * b + c16 < nz;
    off[b] = colok[b] ? 16 * b + c16 : 0;
  }
  const size_t last = hi > lo ? hi - 1 : lo;
  auto fetch = [&](size_t g, double *zz, unsigned char &m) {
    size_t row = g + k;
    bool in = row < hi;
    const double *rp = data + (in ? row : last) * stride;
    m = in ? (use_mask ? mask[row] : (unsigned char)1) : (unsigned char)0;
#pragma unroll
    for (int b = 0; b < NB; b++) zz[b] = rp[off[b]];
  };
  size_t g = lo + (size_t)wave * 4;
#pragma unroll
  for (int p = 0; p < PF; p++) fetch(g + (size_t)p * 16, z[p], mk[p]);
  for (; g < hi; g += 16 * PF) {
#pragma unroll
    for (int p = 0; p < PF; p++) {
      double zc[NB];
      bool vc = mk[p] != 0;
#pragma unroll
      for (int b = 0; b < NB; b++) zc[b] = (vc && colok[b]) ? z[p][b] : 0.0;
      if (VAR != 2) fetch(g + (size_t)(p + PF) * 16, z[p], mk[p]);  // refill this slot for the next round
      rows_used += (unsigned)__builtin_popcountll(__ballot(vc && c16 == 0));
      if (VAR == 1) {  // diagnostic: loads only
#pragma unroll
        for (int b = 0; b < NB; b++) acc[b][0] += zc[b];
      } else {
        int t = 0;
#pragma unroll
        for (int bi = 0; bi < 5; bi++)
#pragma unroll
          for (int bj = bi; bj < 5; bj++, t++)
            if (bi < NB && bj < NB)
              acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[bi < NB ? bi : 0], zc[bj < NB ? bj : 0], acc[t], 0, 0, 0);
      }
    }
  }
  // fold the four waves in order 0,1,2,3; D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  for (int w = 0; w < 4; w++) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 15; t++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          int pos = t * 256 + (k + 4 * rg) * 16 + c16;
          fold[pos] = (w == 0 ? 0.0 : fold[pos]) + acc[t][rg];
        }
      if (lane == 0) s_rows[w] = rows_used;
    }
    __syncthreads();
  }
  double *out = partials + (size_t)blockIdx.x * pstride;
  const int ne = nz * (nz + 1) / 2;
  for (int pos = threadIdx.x; pos < 15 * 256; pos += 256) {
    int t = pos >> 8, r = (pos >> 4) & 15, cc = pos & 15;
    int bi = 0, rem = t;  // t -> (bi, bj) of the upper block triangle
    while (rem >= 5 - bi) {
      rem -= 5 - bi;
      bi++;
    }
    int bj = bi + rem;
    int i = 16 * bi + r, j = 16 * bj + cc;
    if (i <= j && j < nz) out[i * nz - i * (i - 1) / 2 + (j - i)] = fold[pos];
  }
  if (threadIdx.x == 0) out[ne] = (double)(s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3]);
}

// K3 + K4 dense in ONE pass over the rows: consensus mask of the model `par` (bit-identical to DenseModel::agree: the
// reference's running sum, one row per lane) and, from the same rows while they sit in LDS, the block of sums
// sum z z^T over the agreeing rows on the matrix cores (same partial layout as k_syrk_mfma).  Tight records only
// (stride == n + 1), so that a tile of 16 rows is one contiguous run of 128 (n + 1) bytes:
//   * every wave owns a ring of four 16-row tile buffers in LDS and fills them with global_load_lds (16 bytes per
//     lane, no staging registers): the LDS image is the memory image, and because a row is an ODD number of doubles
//     the lanes that walk one row each hit distinct bank pairs -- no padding needed.  Three tiles are in flight while
//     one is evaluated (counted s_waitcnt vmcnt): 4 waves x 25 KB per CU outstanding at all times -- with one tile in
//     flight per wave the kernel ran at the latency of a tile, 3.4 TB/s;
//   * four lanes evaluate a row (one interleaved chain of 16 columns each); the ballot of the decisions is the tile's
//     piece of the mask;
//   * only the agreeing rows are multiplied: their indices are compacted (rank of the lane's bit) and fed four at a
//     time to v_mfma_f64_16x16x4 -- lane (k, c) holds z_k[16 b + c] for block b: at once the A operand of block-row
//     b and the B operand of block-column b (as k_syrk_mfma).  With 30 % of the rows agreeing that is a third of the
//     matrix work of the unconditional SYRK, and the rows are read from HBM once instead of twice.
template <int NA16, int NBUF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NBUF == 2 ? 2 : 1, NBUF == 2 ? 2 : 1))) void k_mask_syrk_dense(
    const double *__restrict__ data, size_t begin, size_t end, size_t chunk, int n, const double *__restrict__ par,
    double delta, uint8_t *__restrict__ mask, unsigned long long *__restrict__ counter, int pstride,
    double *__restrict__ partials, double amax, double bmax, double band_scale, int diag) {
  constexpr int TR = 16;   // rows per tile
  // NBUF = ring of tile buffers per wave.  4: one workgroup per CU, three tiles in flight per wave (n = 64); 2: two
  // workgroups (eight waves) per CU, one tile in flight per wave while the other is evaluated.
  // NA16 = 16-column blocks that cover the n columns of A; the right-hand side b = z[n] is not a matrix block (below).
  extern __shared__ double smd[];
  __shared__ unsigned s_rows[4];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k4 = lane >> 4, c16 = lane & 15;
  const int nz = n + 1;
  const int tile_d = TR * nz;                       // doubles per tile
  double *ring = smd + (size_t)wave * NBUF * tile_d;
  double *xs = smd + (size_t)4 * NBUF * tile_d;     // the model, read with uniform addresses (64 doubles)
  double *pend = xs + 64 + (size_t)wave * 3 * nz;           // up to three agreeing rows waiting for a full group
  if (threadIdx.x < 64) xs[threadIdx.x] = (int)threadIdx.x < n ? par[threadIdx.x] : 0.0;
  __syncthreads();
  // Band of the four-chain evaluation below against the reference's single running sum: both add the same 64 rounded
  // products, in different orders; each order is within (n - 1) u sum |p_i| (1 + O(u)) of the exact sum, u = 2^-53,
  // and sum |p_i| <= Amax ||x||_1.  E = 2.2 n u (Amax ||x||_1 + Bmax) covers both and the final subtraction.
  // Non-finite magnitudes (or a NaN model) give E = NaN / inf: every tile takes the serial evaluation.
  double l1 = 0.0;
  for (int i = 0; i < 64; i++) l1 += fabs(xs[i]);
  double xq[16];  // the model entries of this lane's chain (columns q, q + 4, ...), in registers for the whole pass
#pragma unroll
  for (int i = 0; i < 16; i++) xq[i] = xs[4 * i + (lane & 3)];
  const double eband = band_scale * 2.2 * 64.0 * 1.1102230246251565e-16 * (amax * l1 + bmax);
  constexpr int NT = NA16 * (NA16 + 1) / 2;
  d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
  double vb = 0.0, bb = 0.0;  // A^T b (lane = column) and b^T b, on the vector unit: see group()
  int off[NA16];
  bool colok[NA16];
#pragma unroll
  for (int b = 0; b < NA16; b++) {
    colok[b] = 16 * b + c16 < n;
    off[b] = colok[b] ? 16 * b + c16 : 0;
  }
  const size_t lo = begin + (size_t)blockIdx.x * chunk;
  const size_t hi = lo + chunk < end ? lo + chunk : end;
  unsigned rows_used = 0;
  int npend = 0;
  // a tile = TR consecutive rows = one contiguous run of bytes; 16-byte pieces straight into LDS (a whole tile is a
  // multiple of 16 bytes; the last, shorter tile of a range may end in one 8-byte piece)
  auto issue = [&](size_t base, double *dst) {
    const size_t rows = hi - base < (size_t)TR ? hi - base : (size_t)TR;
    const size_t bytes = rows * (size_t)nz * 8;
    const char *g = (const char *)(data + base * (size_t)nz);
    if (NA16 == 4 && n == 64 && rows == (size_t)TR) {
      // a whole tile of 65-double rows: 8320 bytes = eight full pieces and 128 bytes; one per-lane address and
      // immediate offsets instead of nine 64-bit address computations and compares (one wave per SIMD: every
      // instruction of the wave is on the tile's critical path)
      typedef __attribute__((address_space(3))) void *lds_t;
      const char *g0 = g + (size_t)lane * 16, *g1 = g0 + 4096;
      char *d = (char *)dst;
      // (the instruction's immediate offset moves the global AND the LDS address)
      __builtin_amdgcn_global_load_lds((const void *)g0, (lds_t)(d), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void *)g0, (lds_t)(d), 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((const void *)g0, (lds_t)(d), 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((const void *)g0, (lds_t)(d), 16, 3072, 0);
      __builtin_amdgcn_global_load_lds((const void *)g1, (lds_t)(d + 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void *)g1, (lds_t)(d + 4096), 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((const void *)g1, (lds_t)(d + 4096), 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((const void *)g1, (lds_t)(d + 4096), 16, 3072, 0);
      if (lane < 8) __builtin_amdgcn_global_load_lds((const void *)(g1 + 4096), (lds_t)(d + 8192), 16, 0, 0);
      return;
    }
    for (size_t o = 0; o < bytes; o += 1024) {      // wave-uniform trip count
      const size_t mine = o + (size_t)lane * 16;
      if (mine + 16 <= bytes)
        __builtin_amdgcn_global_load_lds((const void *)(g + mine),
                                         (__attribute__((address_space(3))) void *)((char *)dst + o), 16, 0, 0);
      else if (mine + 8 <= bytes)
        ((double *)dst)[mine / 8] = *(const double *)(g + mine);
    }
  };
  // Four agreeing rows -- the waiting ones first, then rows of the tile in index order -- into the sums:
  //   A^T A (the NA16 (NA16 + 1) / 2 upper 16 x 16 blocks) on the matrix cores: lane (k, c) holds z_k[16 b + c] for
  //   block b, at once the A operand of block-row b and the B operand of block-column b (as k_syrk_mfma);
  //   A^T b and b^T b on the vector unit, lane = column: one fused multiply-add per row.  As a fifth block row /
  //   column of the matrix product (r03 before) the right-hand side cost 5 of 15 MFMAs for 65 useful entries.
  // The group's rows are `nlive` <= 4 wave-uniform LDS pointers p[]: no index list in memory, no dependent reads --
  // with one wave per SIMD every LDS round trip in front of the MFMAs is idle time (r03: an index list in LDS and a
  // row loop around the vector part cost 1100 cycles a tile on top of the 640 of the MFMAs).
  auto group = [&](const double *const (&p)[4], int nlive) {
    const double *rp = k4 == 0 ? p[0] : k4 == 1 ? p[1] : k4 == 2 ? p[2] : p[3];
    const bool live = k4 < nlive;
    double zc[NA16], zl[4], zb[4];
#pragma unroll
    for (int b = 0; b < NA16; b++) zc[b] = rp[off[b]];
#pragma unroll
    for (int j = 0; j < 4; j++) zl[j] = p[j][lane < n ? lane : 0], zb[j] = p[j][n];
#pragma unroll
    for (int b = 0; b < NA16; b++) zc[b] = (live && colok[b]) ? zc[b] : 0.0;
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < NA16; bi++)
#pragma unroll
      for (int bj = bi; bj < NA16; bj++, t++)
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[bi], zc[bj], acc[t], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const double l = (j < nlive && lane < n) ? zl[j] : 0.0, r = j < nlive ? zb[j] : 0.0;
      vb = fma(l, r, vb);  // (a dead slot adds 0 * 0)
      bb = fma(r, r, bb);
    }
  };
  // tile j of this wave: rows [lo + (4 j + wave) * TR, + TR)
  const size_t step = (size_t)4 * TR;
  size_t base = lo + (size_t)wave * TR;
#pragma unroll
  for (int p = 0; p < NBUF - 1; p++)
    if (base + p * step < hi) issue(base + p * step, ring + (size_t)p * tile_d);
  for (unsigned j = 0; base < hi; base += step, j++) {
    double *tile = ring + (size_t)(j % NBUF) * tile_d;
    const size_t ahead = base + (size_t)(NBUF - 1) * step;
    // My tile has landed when all but the operations issued after it are done.  Loads, LDS-DMA and stores retire in
    // order; after tile j's pieces came, per later iteration, one mask store and the pieces of one more tile.  Only
    // the full case (n = 64: nine pieces per 16-row tile, every later tile whole) is counted; anything else drains.
    if (NBUF == 4 && NA16 == 4 && n == 64 && ahead + TR <= hi)
      asm volatile("s_waitcnt vmcnt(21) lgkmcnt(0)" ::: "memory");   // 2 x (1 store + 9 pieces) + 1 store
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (ahead < hi) issue(ahead, ring + (size_t)((j + NBUF - 1) % NBUF) * tile_d);  // the buffer read last iteration
    const int rows = (int)(hi - base < (size_t)TR ? hi - base : (size_t)TR);
    bool a;
    {
      // FOUR lanes walk a row: lane 4 r + q takes the columns i = q (mod 4) of row r -- one of the four interleaved
      // chains (rows past the tile's end walk row 0: no divergence, their result is dropped); multiply and add
      // unfused (-ffp-contract=off).  The reference's single running sum is one chain of 128 dependent fp64
      // operations; the four chains, combined as (s0 + s1) + (s2 + s3), give the residual up to the band E above, and
      // only a tile with a row inside |  |r| - delta | <= E (one in ~1e11 rows) is walked again serially.  (r03: one
      // lane per row walked all four chains itself, 16 of the 64 lanes busy for 128 operations: 1770 cycles a tile.)
      const int rr = lane >> 2, q = lane & 3;
      const double *row = tile + (rr < rows ? rr : 0) * nz;
      double sq = 0.0;
      if (diag & 2) {  // timing diagnostics: no evaluation
        sq = row[0];
      } else if (NA16 == 4 && n == 64) {
#pragma unroll
        for (int i = 0; i < 16; i++) sq += row[4 * i + q] * xq[i];
      } else {
        for (int i = 0; 4 * i + q < (n & ~3); i++) sq += row[4 * i + q] * xq[i];
        if (q == 0)
          for (int i = n & ~3; i < n; i++) sq += row[i] * xs[i];  // (the tail rides on chain 0)
      }
      const double s01 = sq + __shfl_down(sq, 1);     // q = 0: s0 + s1,  q = 2: s2 + s3
      const double s = s01 + __shfl_down(s01, 2);     // q = 0: (s0 + s1) + (s2 + s3)
      const double rq = fabs(s - row[n]);
      const bool mine = q == 0 && rr < rows;
      a = rq < delta;
      const bool unsure = !(fabs(rq - delta) > eband);      // also true for NaN
      if (__ballot(unsure && mine)) {                       // wave-uniform, rare: the reference's own order
        double sum = 0.0;
        for (int i = 0; i < n; i++) sum += row[i] * xs[i];
        sum -= row[n];
        a = fabs(sum) < delta;
      }
      a = a && mine;
      if (mine) mask[base + rr] = a ? 1 : 0;
    }
    const unsigned long long in = __ballot(a);   // bit 4 r: row r agrees
    const int cnt = __builtin_popcountll(in);
    rows_used += (unsigned)cnt;
    if (diag & 4) __builtin_amdgcn_s_sleep(32);  // timing diagnostics: a delay of ~2000 cycles instead of the matrix work
    if (!(diag & 1)) {
      // whole groups of four out of (waiting rows, then the tile's agreeing rows in index order); what is left over
      // (< 4 rows) waits in `pend` for the next tile -- a tile holds 4.6 agreeing rows on average at 29 % inliers, and
      // rounding every tile up to whole groups by itself cost 1.7 groups a tile instead of 1.16
      unsigned long long rest = in;
      auto next_row = [&]() -> const double * {  // (scalar unit)
        const int b = __builtin_ctzll(rest);
        rest &= rest - 1;
        return tile + (b >> 2) * nz;
      };
      int total = npend + cnt, used = 0;          // used = waiting rows consumed
      while (total >= 4) {                        // wave-uniform
        const double *p[4];
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = used < npend ? pend + (used++) * nz : next_row();
        group(p, 4);
        total -= 4;
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the waiting rows have been read)
      // what remains: either the untouched waiting rows plus tile rows (no group ran) or tile rows only
      int slot = used < npend ? npend : 0;
      while (rest) {                              // <= 3 rows
        const double *src = next_row();
        double *dst = pend + slot * nz;
        for (int i = lane; i < nz; i += 64) dst[i] = src[i];
        slot++;
      }
      npend = total;
    }
    __builtin_amdgcn_wave_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (npend) {  // the last, partial group
    const double *p[4] = {pend, pend + (npend > 1 ? nz : 0), pend + (npend > 2 ? 2 * nz : 0), pend};
    group(p, npend);
    npend = 0;
  }
  __syncthreads();  // every wave is done with its tiles: the LDS becomes the fold area
  double *fold = smd, *fvb = smd + NT * 256;
  for (int w = 0; w < 4; w++) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int pos = t * 256 + (k4 + 4 * rg) * 16 + c16;
          fold[pos] = (w == 0 ? 0.0 : fold[pos]) + acc[t][rg];
        }
      fvb[lane] = (w == 0 ? 0.0 : fvb[lane]) + vb;
      if (lane == 0) {
        fvb[64] = (w == 0 ? 0.0 : fvb[64]) + bb;
        s_rows[w] = rows_used;
      }
    }
    __syncthreads();
  }
  double *out = partials + (size_t)blockIdx.x * pstride;
  const int ne = nz * (nz + 1) / 2;
  for (int pos = threadIdx.x; pos < NT * 256; pos += 256) {
    const int t = pos >> 8, r = (pos >> 4) & 15, cc = pos & 15;
    int bi = 0, rem = t;
    while (rem >= NA16 - bi) {
      rem -= NA16 - bi;
      bi++;
    }
    const int bj = bi + rem;
    const int i = 16 * bi + r, j = 16 * bj + cc;
    if (i <= j && j < n) out[i * nz - i * (i - 1) / 2 + (j - i)] = fold[pos];
  }
  if ((int)threadIdx.x < n) out[threadIdx.x * nz - (int)threadIdx.x * ((int)threadIdx.x - 1) / 2 + (n - (int)threadIdx.x)] = fvb[threadIdx.x];
  if (threadIdx.x == 64) out[ne - 1] = fvb[64];
  if (threadIdx.x == 0) {
    const unsigned tot = s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3];
    out[ne] = (double)tot;
    if (tot) atomicAdd(counter, (unsigned long long)tot);
  }
}

// K5 dense: x = pinv(A) b from the normal equations block (DenseLinear...Estimator.hxx:64-96:
// rank(A) < n -> empty).  One wave; G = A^T A (n x n) in LDS.
// flag (nullable): raised instead of taking the eigen / SVD path on G when the elimination is refused -- the caller
// then solves the system again from the rows (k_gram_dd_dense / k_dense_dd_solve below); with flag == nullptr (only
// the block is at hand: lsqr_solve_moments, the multi-GPU sum) the pseudo-inverse of G decides, rank test relative.
__global__ __launch_bounds__(256) void k_solve_dense(const double *__restrict__ mom, int n, int fast,
                                                     SolveOut *__restrict__ out, int *__restrict__ flag) {
  extern __shared__ double sm[];
  const int tid = threadIdx.x, nz = n + 1, lda = n | 1;
  const int ne = nz * (nz + 1) / 2;
  double *G = sm, *V = G + n * lda, *rhs = V + n * lda, *cw = rhs + n, *x = cw + n;
  auto load_system = [&]() {
    for (int idx = tid; idx < n * n; idx += 256) {
      int i = idx / n, j = idx % n;
      int a = i < j ? i : j, bb = i < j ? j : i;
      int e = a * nz - a * (a - 1) / 2 + (bb - a);
      G[j * lda + i] = mom[e];
    }
    for (int i = tid; i < n; i += 256) {
      int e = i * nz - i * (i - 1) / 2 + (n - i);
      rhs[i] = mom[e];
    }
  };
  load_system();
  __syncthreads();
  double count = mom[ne];
  // sigma(A)^2 are the singular values of G: rank test relative to the largest one (the
  // reference's absolute sigma <= 2.2e-16 test only ever fires for exactly singular systems).
  // Well-conditioned normal equations (every pivot > 1e-8 max|G|) are solved by elimination; anything
  // closer to the rank decision goes through the eigen/SVD path that makes it.
  int rank = n;
  // (with the rows at hand the elimination is only trusted while eps cond(G) stays far below the 1e-6 bar: pivots
  // above 1e-6 max|G|; without them, as for the minimal solves, above 1e-8)
  // (r04: the elimination by ONE wave -- wave_linalg.h: wave_gepp_solve, bit-identical to block_gepp_solve<256> --
  // while the other three wait: 0.14 -> ~0.05 ms for the 64 x 64 system, the workgroup version spends its time in
  // barriers)
  __shared__ int s_solved, s_hint;
  if (tid == 0) s_hint = 0;
  if (fast) {
    if (tid < 64) {
      bool okw;
      if (n == 64) {  // the system in registers (wave_linalg.h: wave_gepp_solve_reg64, bit-identical): lane = row
        double a[64], xv;
#pragma unroll
        for (int c = 0; c < 64; c++) a[c] = G[c * lda + tid];
        okw = wave_gepp_solve_reg64(a, rhs[tid], xv, 1e-6);
        if (!okw && !flag) {  // only the block is at hand: solve all the same, but say that 1e-6 is not guaranteed (r05)
          if (tid == 0) s_hint = 1;
#pragma unroll
          for (int c = 0; c < 64; c++) a[c] = G[c * lda + tid];
          okw = wave_gepp_solve_reg64(a, rhs[tid], xv, 1e-8);
        }
        x[tid] = xv;
      } else {
        okw = wave_gepp_solve(n, G, lda, rhs, x, 1e-6);
        if (!okw && !flag) {
          if (tid == 0) s_hint = 1;
          __builtin_amdgcn_wave_barrier();
          for (int idx = tid; idx < n * n; idx += 64) {  // the elimination worked in place: the system again
            int i = idx / n, j = idx % n;
            int a = i < j ? i : j, bb = i < j ? j : i;
            G[j * lda + i] = mom[a * nz - a * (a - 1) / 2 + (bb - a)];
          }
          for (int i = tid; i < n; i += 64) rhs[i] = mom[i * nz - i * (i - 1) / 2 + (n - i)];
          __builtin_amdgcn_wave_barrier();
          okw = wave_gepp_solve(n, G, lda, rhs, x, 1e-8);
        }
      }
      if (tid == 0) s_solved = okw ? 1 : 0;
    }
    __syncthreads();
  }
  if (!fast || !s_solved) {
    if (flag) {              // the rows are at hand: the double-double route decides (workgroup-uniform branch)
      if (tid == 0) {
        *flag = 1;
        out->ok = 0;
        out->n_params = 0;
      }
      return;
    }
    __syncthreads();
    load_system();
    __syncthreads();
    rank = block_pinv_solve<256>(n, n, G, lda, V, lda, rhs, 0.0, 1e-13, x, cw);
    if (tid == 0) s_hint = 1;
  }
  __syncthreads();
  bool ok = rank == n && count >= (double)n;
  if (tid == 0) {
    out->ok = ok ? 1 : 0;
    out->n_params = ok ? n : 0;
    out->lm_info = 0;
    out->lm_nfev = 0;
    out->cont = 0;
    out->cost = 0.0;
    // 2: solved from the block alone although a pivot fell below 1e-6 max|G| (cond(A) >~ 1e3): the result carries
    // eps cond(A)^2 and the caller who holds the rows should fit again from them (lsqr_ls_fit: the double-double
    // route) -- what the multi-GPU finish does, so that an N-GPU fit equals the 1-GPU fit on such systems too
    out->pad = s_hint ? 2 : 0;
  }
  for (int j = tid; j < n; j += 256) out->params[j] = ok ? x[j] : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense least squares on ILL-CONDITIONED systems (r04): the reference's result, not the normal equations' one.
//
// DenseLinearEquationSystemParametersEstimator.hxx:64-96 solves min |Ax - b| by the SVD pseudo-inverse of A itself
// (vnl_matrix_inverse) and declares rank(A) < n only for a singular value <= EPS = 2.2e-16 ABSOLUTE (:88-91).  The
// one-pass route above forms G = A^T A in double: its rounding (eps |A|^2) swamps sigma_min^2 once cond(A) >~ 1e7 and
// the solution loses eps cond(A)^2 -- the 1e-6 bar of this repository holds to cond(A) ~ 6e4 only.  When the
// elimination of k_solve_dense meets a pivot below 1e-8 max|G| it therefore raises `flag`, and the system is solved
// again from the rows, in three steps that keep every digit the SVD of A would have:
//   k_gram_dd_dense   the augmented Gram matrix (A|b)^T (A|b) of the rows in use accumulated in DOUBLE-DOUBLE:
//                     products exact (TwoProduct by fma), sums compensated (TwoSum; Ogita-Rump-Oishi Dot2) --
//                     relative error ~1e-32 m instead of 1e-16 m, so sigma_min^2 survives up to cond(A) ~ 1e15;
//   k_dense_dd_solve  (one workgroup) fixed-order double-double sum of the blocks' partials, Cholesky factor
//                     (R | z ; 0 rho) of the augmented Gram matrix in double-double -- R is the triangular factor of
//                     A's QR decomposition (A = QR) and z = Q^T b, to eps_dd cond(A)^2 --, both rounded to double,
//                     the one-sided Jacobi SVD of R (wave_linalg.h) gives the singular values of A for the
//                     reference's ABSOLUTE rank test (:88-91), and with full rank x = pinv(A) b = R^-1 z is taken by
//                     back substitution in double-double: the exact least-squares solution to ~1e-12 at cond 1e10,
//                     i.e. what remains between this result and an SVD route's is the SVD's own eps cond(A).
// A Cholesky breakdown (a column in the span of the earlier ones to 14 digits: cond(A) beyond ~1e14, exact dependencies
// included) is reported as EMPTY -- as the oracle's Jacobi SVD reports exactly dependent columns.
// Both kernels return at once while `flag` is 0 (well-conditioned system: the elimination's result stands).
struct dd_t {
  double hi, lo;
};
__device__ __forceinline__ dd_t dd_two_sum(double a, double b) {
  const double s = a + b, bb = s - a;
  return {s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd_t dd_quick_two_sum(double a, double b) {  // |a| >= |b|
  const double s = a + b;
  return {s, b - (s - a)};
}
__device__ __forceinline__ dd_t dd_two_prod(double a, double b) {
  const double p = a * b;
  return {p, fma(a, b, -p)};
}
__device__ __forceinline__ dd_t dd_add(dd_t a, dd_t b) {
  dd_t s = dd_two_sum(a.hi, b.hi);
  const dd_t t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = dd_quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return dd_quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ dd_t dd_sub(dd_t a, dd_t b) { return dd_add(a, dd_t{-b.hi, -b.lo}); }
__device__ __forceinline__ dd_t dd_mul(dd_t a, dd_t b) {
  dd_t p = dd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd_t dd_mul_d(dd_t a, double b) {
  dd_t p = dd_two_prod(a.hi, b);
  p.lo += a.lo * b;
  return dd_quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ dd_t dd_div(dd_t a, dd_t b) {  // three quotient digits
  const double q1 = a.hi / b.hi;
  dd_t r = dd_sub(a, dd_mul_d(b, q1));
  const double q2 = r.hi / b.hi;
  r = dd_sub(r, dd_mul_d(b, q2));
  const double q3 = r.hi / b.hi;
  dd_t q = dd_quick_two_sum(q1, q2);
  return dd_add(q, dd_t{q3, 0.0});
}
__device__ __forceinline__ dd_t dd_sqrt(dd_t a) {  // a > 0; one Newton step on the double root
  const double x = 1.0 / sqrt(a.hi), ax = a.hi * x;
  const dd_t d = dd_sub(a, dd_two_prod(ax, ax));
  return dd_add(dd_t{ax, 0.0}, dd_t{d.hi * (x * 0.5), 0.0});
}

constexpr int kDdNz = 66;                          // 64 columns + right-hand side, padded to 22 blocks of 3
constexpr int kDdNe = kDdNz * (kDdNz + 1) / 2;     // packed upper triangle
constexpr int kDdBlocks = 256;                     // workgroups of the Gram pass = partial blocks summed by the solve
__host__ __device__ inline int dd_packed(int i, int j) { return i * kDdNz - i * (i - 1) / 2 + (j - i); }  // i <= j

// part: [gridDim.x][2][kDdNe] -- hi then lo of the block's share of sum z z^T, z = (row | b), rows [begin, end) in use
// (mask nullable: every row).  Thread t < 253 owns the 3 x 3 block (bi, bj), bi <= bj, of the 22 x 22 block matrix.
template <int TR>
__global__ __launch_bounds__(256) void k_gram_dd_dense(const double *__restrict__ data, size_t stride, size_t begin,
                                                       size_t end, int n, const uint8_t *__restrict__ mask,
                                                       const int *__restrict__ flag, double *__restrict__ part) {
  if (*flag == 0) return;
  __shared__ double s_rows[TR][kDdNz + 1];
  __shared__ int s_use[TR];
  const int t = threadIdx.x;
  int bi = 0, rem = t;
  while (bi < 22 && rem >= 22 - bi) {
    rem -= 22 - bi;
    bi++;
  }
  const bool act = bi < 22;
  const int bj = act ? bi + rem : 0;
  if (!act) bi = 0;
  double hi[3][3], lo[3][3];
#pragma unroll
  for (int p = 0; p < 3; p++)
#pragma unroll
    for (int q = 0; q < 3; q++) hi[p][q] = lo[p][q] = 0.0;
  const size_t total = end - begin;
  const size_t chunk = (total + gridDim.x - 1) / gridDim.x;
  const size_t r0 = begin + (size_t)blockIdx.x * chunk;
  const size_t r1 = r0 + chunk < end ? r0 + chunk : end;
  for (size_t base = r0; base < r1; base += TR) {
    __syncthreads();
    if (t < TR) {
      const size_t row = base + t;
      s_use[t] = row < r1 && (!mask || mask[row]) ? 1 : 0;
    }
    __syncthreads();
    for (int idx = t; idx < TR * kDdNz; idx += 256) {
      const int r = idx / kDdNz, cc = idx - r * kDdNz;
      s_rows[r][cc] = (s_use[r] && cc <= n) ? data[(base + r) * stride + cc] : 0.0;
    }
    __syncthreads();
    if (act)
      for (int r = 0; r < TR; r++) {
        if (!s_use[r]) continue;  // workgroup-uniform
        double a[3], b[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
          a[p] = s_rows[r][3 * bi + p];
          b[p] = s_rows[r][3 * bj + p];
        }
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
          for (int q = 0; q < 3; q++) {  // Dot2: exact product, compensated sum
            const dd_t pr = dd_two_prod(a[p], b[q]);
            const dd_t sm = dd_two_sum(hi[p][q], pr.hi);
            hi[p][q] = sm.hi;
            lo[p][q] += sm.lo + pr.lo;
          }
      }
  }
  if (act) {
    double *out = part + (size_t)blockIdx.x * 2 * kDdNe;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = 0; q < 3; q++) {
        const int i = 3 * bi + p, j = 3 * bj + q;
        if (i <= j) {
          out[dd_packed(i, j)] = hi[p][q];
          out[kDdNe + dd_packed(i, j)] = lo[p][q];
        }
      }
  }
}

// dynamic LDS: packed double-double Gram / factor (2 * kDdNe), then R (n x lda), V (n x lda), z, cw, x: dense_dd_lds()
__host__ __device__ inline size_t dense_dd_lds(int n) {
  return sizeof(double) * ((size_t)2 * kDdNe + (size_t)2 * n * (n | 1) + 3 * n);
}
__global__ __launch_bounds__(256) void k_dense_dd_solve(const double *__restrict__ part, int nblocks, int n,
                                                        const double *__restrict__ mom, const int *__restrict__ flag,
                                                        SolveOut *__restrict__ out) {
  if (*flag == 0) return;
  extern __shared__ double sm[];
  __shared__ int s_bad;
  const int tid = threadIdx.x, nz = n + 1, lda = n | 1;
  double *Ghi = sm, *Glo = sm + kDdNe;
  double *R = Glo + kDdNe, *V = R + n * lda, *z = V + n * lda, *cw = z + n, *x = cw + n;
  if (tid == 0) s_bad = 0;
  // fixed-order double-double sum of the partial blocks
  for (int e = tid; e < kDdNe; e += 256) {
    dd_t acc = {0.0, 0.0};
    for (int b = 0; b < nblocks; b++) {
      const double *p = part + (size_t)b * 2 * kDdNe;
      acc = dd_add(acc, dd_quick_two_sum(p[e], p[kDdNe + e]));   // (|hi| >= |lo| by construction: lo = error terms)
    }
    Ghi[e] = acc.hi;
    Glo[e] = acc.lo;
  }
  __syncthreads();
  // Cholesky of the augmented Gram matrix, row by row, in place (upper factor): row i of the factor needs rows < i
  for (int i = 0; i < nz; i++) {
    // diagonal: thread 0 (a chain of i terms); the others wait
    if (tid == 0) {
      dd_t d = {Ghi[dd_packed(i, i)], Glo[dd_packed(i, i)]};
      const double g0 = d.hi;                               // |a_i|^2
      for (int k = 0; k < i; k++) {
        const dd_t r = {Ghi[dd_packed(k, i)], Glo[dd_packed(k, i)]};
        d = dd_sub(d, dd_mul(r, r));
      }
      // breakdown: column i lies in the span of the columns before it to 14 digits (what is left of |a_i|^2 is below
      // 1e-28 of it: cond(A) beyond ~1e14, where the double-double Gram matrix has no digits left either)
      if (i < n && !(d.hi > 1e-28 * g0)) s_bad = 1;
      const dd_t rii = d.hi > 0.0 ? dd_sqrt(d) : dd_t{0.0, 0.0};  // (i == n: the residual norm; may round to <= 0)
      Ghi[dd_packed(i, i)] = rii.hi;
      Glo[dd_packed(i, i)] = rii.lo;
    }
    __syncthreads();
    if (s_bad) break;
    const dd_t rii = {Ghi[dd_packed(i, i)], Glo[dd_packed(i, i)]};
    for (int j = i + 1 + tid; j < nz; j += 256) {
      dd_t v = {Ghi[dd_packed(i, j)], Glo[dd_packed(i, j)]};
      for (int k = 0; k < i; k++) {
        const dd_t a = {Ghi[dd_packed(k, i)], Glo[dd_packed(k, i)]}, b = {Ghi[dd_packed(k, j)], Glo[dd_packed(k, j)]};
        v = dd_sub(v, dd_mul(a, b));
      }
      v = rii.hi > 0.0 ? dd_div(v, rii) : dd_t{0.0, 0.0};
      Ghi[dd_packed(i, j)] = v.hi;
      Glo[dd_packed(i, j)] = v.lo;
    }
    __syncthreads();
  }
  int rank = 0;
  __shared__ double s_zh[64], s_zl[64], s_xh[64];
  if (!s_bad) {
    // The rank decision is the reference's: singular values of R (= those of A) against EPS, absolute (:88-91) --
    // one-sided Jacobi SVD of R rounded to double.
    for (int idx = tid; idx < n * n; idx += 256) {
      const int i = idx % n, j = idx / n;                    // column-major
      R[j * lda + i] = i <= j ? Ghi[dd_packed(i, j)] + Glo[dd_packed(i, j)] : 0.0;
    }
    for (int i = tid; i < n; i += 256) z[i] = Ghi[dd_packed(i, n)] + Glo[dd_packed(i, n)];
    __syncthreads();
    rank = block_pinv_solve<256>(n, n, R, lda, V, lda, z, kEPS, 0.0, x, cw);
    __syncthreads();
    // With full rank pinv(A) b is THE least-squares solution R^-1 z: taken by back substitution in double-double
    // from the double-double factor, it carries none of an SVD's eps cond(A) rounding (measured against the exact
    // solution of consistent systems at cond 1e10: 1e-12, where the Jacobi SVD of R gave 1e-6 and the oracle's /
    // LAPACK's SVD of A 5e-8 / 1e-8 -- tests/test_gpu_dense_cond.py).
    if (rank == n) {
      if (tid < n) {
        s_zh[tid] = Ghi[dd_packed(tid, n)];
        s_zl[tid] = Glo[dd_packed(tid, n)];
      }
      __syncthreads();
      for (int i = n - 1; i >= 0; i--) {
        const dd_t xi = dd_div(dd_t{s_zh[i], s_zl[i]}, dd_t{Ghi[dd_packed(i, i)], Glo[dd_packed(i, i)]});
        __syncthreads();                                     // (everyone has read z_i)
        if (tid < i) {
          const dd_t zj = dd_sub(dd_t{s_zh[tid], s_zl[tid]},
                                 dd_mul(dd_t{Ghi[dd_packed(tid, i)], Glo[dd_packed(tid, i)]}, xi));
          s_zh[tid] = zj.hi;
          s_zl[tid] = zj.lo;
        }
        if (tid == 0) s_xh[i] = xi.hi + xi.lo;
        __syncthreads();
      }
      for (int j = tid; j < n; j += 256) x[j] = s_xh[j];
      __syncthreads();
    }
  }
  const int ne = nz * (nz + 1) / 2;
  const bool ok = !s_bad && rank == n && mom[ne] >= (double)n;
  if (tid == 0) {
    out->ok = ok ? 1 : 0;
    out->n_params = ok ? n : 0;
    out->lm_info = 0;
    out->lm_nfev = 0;
    out->cont = 0;
    out->cost = 0.0;
    out->pad = 1;   // (diagnostics: the double-double route produced this result)
  }
  for (int j = tid; j < n; j += 256) out->params[j] = ok ? x[j] : 0.0;
}

}  // namespace lsqr
