// cells.h -- two-level agree() scan for the point models (plane / sphere / line).
//
// The batched scan (kernels.h: k_scan, k_scan_f32) evaluates every (hypothesis, observation) pair.
// Almost all of those pairs are far from the model: a random plane through a +-1000 box with
// delta = 0.5 is near 0.05 % of the points.  Because the observations of one upload are fixed for
// all hypotheses, they are binned ONCE into spatially compact cells of kCellPts observations (a
// linear BVH: Morton order, consecutive runs), each with a conservative fp32 bounding box.  The
// scan then runs in two levels:
//
//   level 1  lane = hypothesis (64 at a time), cell box in SGPRs: one conservative fp32 test says
//            whether the model can be within its candidate threshold of ANY point of the box
//            (8 VALU slots per 64 hypotheses x 128 observations);
//   level 2  for each surviving (hypothesis, cell) the hypothesis is broadcast (v_readlane) and the
//            cell's 128 observations (two per lane, packed fp32) go through the same pre-filter +
//            exact fp64 re-check as k_scan_f32.
//
// Votes are counts, so the permutation of the observations does not matter; a culled cell holds
// only observations that certainly do not agree (bound below), so votes are bit-identical to the
// exhaustive kernels (tests/test_gpu_parity.py::test_cell_scan_*).
//
// Index build (k_bounds -> k_keys -> 3-kernel prefix sum -> k_scatter -> k_cell_boxes) is a counting
// sort on Morton keys: once per upload, a few HBM passes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "models.h"

#pragma clang diagnostic ignored "-Winline-asm"  // m0 is clobbered on purpose (k_scan_cells)

namespace lsqr {

constexpr int kCellPtsMin = 128;  // cell sizes are multiples of one packed fp32 pair per lane

struct CellBox {  // 32 B: one s_load_dwordx8
  float c[3];     // centre (exactly representable, inside the box)
  float h[3];     // half extents, inflated: every point satisfies |x_i - c_i| * (1 + 2^-20) <= h_i
  float pad[2];
};

// ---- index build ---------------------------------------------------------------------------------
__device__ inline unsigned long long ord_u64(double v) {  // monotone map double -> uint64
  unsigned long long b;
  __builtin_memcpy(&b, &v, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
inline double ord_u64_inv(unsigned long long o) {
  unsigned long long b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFULL) : ~o;
  double v;
  memcpy(&v, &b, 8);
  return v;
}

// out[0..2] = min per dimension, out[3..5] = max (ordered encoding); non-finite records are skipped
template <int D>
__global__ __launch_bounds__(256) void k_bounds(const double *__restrict__ data, size_t stride,
                                                size_t n, unsigned long long *__restrict__ out) {
  unsigned long long mn[D], mx[D];
  for (int d = 0; d < D; d++) mn[d] = ~0ULL, mx[d] = 0ULL;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    double x[D];
    bool fin = true;
    for (int d = 0; d < D; d++) {
      x[d] = data[i * stride + d];
      fin = fin && (fabs(x[d]) <= 1.7976931348623157e308);  // false for NaN and inf
    }
    if (!fin) continue;
    for (int d = 0; d < D; d++) {
      unsigned long long o = ord_u64(x[d]);
      mn[d] = o < mn[d] ? o : mn[d];
      mx[d] = o > mx[d] ? o : mx[d];
    }
  }
  for (int d = 0; d < D; d++) {
    for (int o = 32; o > 0; o >>= 1) {
      unsigned long long a = __shfl_down(mn[d], o), b = __shfl_down(mx[d], o);
      mn[d] = a < mn[d] ? a : mn[d];
      mx[d] = b > mx[d] ? b : mx[d];
    }
    if ((threadIdx.x & 63) == 0) {
      if (mn[d] != ~0ULL) atomicMin(&out[d], mn[d]);
      if (mx[d] != 0ULL) atomicMax(&out[3 + d], mx[d]);
    }
  }
}

struct IndexGrid {
  double lo[3], scale[3];  // bin_d = min((unsigned)((x_d - lo_d) * scale_d), nb - 1)
  uint32_t bits;           // bits per dimension (nb = 1 << bits)
  uint32_t nbins;          // (1 << (bits * D)); key nbins = "non-finite record", sorted to the tail
};

__device__ inline uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3FF;
  v = (v | (v << 16)) & 0x30000FF;
  v = (v | (v << 8)) & 0x300F00F;
  v = (v | (v << 4)) & 0x30C30C3;
  v = (v | (v << 2)) & 0x9249249;
  return v;
}
__device__ inline uint32_t spread2(uint32_t v) {  // 16 bits -> every second bit
  v &= 0xFFFF;
  v = (v | (v << 8)) & 0x00FF00FF;
  v = (v | (v << 4)) & 0x0F0F0F0F;
  v = (v | (v << 2)) & 0x33333333;
  v = (v | (v << 1)) & 0x55555555;
  return v;
}

template <int D>
__global__ __launch_bounds__(256) void k_keys(const double *__restrict__ data, size_t stride,
                                              size_t n, IndexGrid g, uint32_t *__restrict__ keys,
                                              uint32_t *__restrict__ hist) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t q[3] = {0, 0, 0};
  bool fin = true;
  const uint32_t nb1 = (1u << g.bits) - 1;
  for (int d = 0; d < D; d++) {
    double x = data[i * stride + d];
    fin = fin && (fabs(x) <= 1.7976931348623157e308);
    double t = (x - g.lo[d]) * g.scale[d];
    uint32_t b = t >= 0.0 ? (t < (double)nb1 ? (uint32_t)t : nb1) : 0u;  // NaN -> 0
    q[d] = b;
  }
  uint32_t key;
  if (!fin) key = g.nbins;
  else if (D == 3) key = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
  else key = spread2(q[0]) | (spread2(q[1]) << 1);
  keys[i] = key;
  atomicAdd(&hist[key], 1u);
}

// exclusive prefix sum of hist[0..m) in three kernels (4096 entries per block)
constexpr int kScanSpan = 4096;
__global__ __launch_bounds__(256) void k_hist_blocksum(const uint32_t *__restrict__ hist, uint32_t m,
                                                       uint32_t *__restrict__ bsum) {
  __shared__ uint32_t s[4];
  uint32_t base = blockIdx.x * kScanSpan, t = 0;
  for (int k = 0; k < kScanSpan / 256; k++) {
    uint32_t i = base + k * 256 + threadIdx.x;
    t += i < m ? hist[i] : 0u;
  }
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// exclusive scan of the block sums in place (nb <= kMaxScanBlocks, one workgroup): staged in LDS,
// each of the first 64 threads owns a consecutive chunk, wave scan of the chunk sums
constexpr int kMaxScanBlocks = 4352;
__global__ __launch_bounds__(256) void k_hist_scan_bsum(uint32_t *bsum, uint32_t nb) {
  __shared__ uint32_t s[kMaxScanBlocks];
  for (uint32_t i = threadIdx.x; i < nb; i += 256) s[i] = bsum[i];
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t per = (nb + 63) / 64, lo = threadIdx.x * per;
    uint32_t t = 0;
    for (uint32_t k = 0; k < per; k++) t += lo + k < nb ? s[lo + k] : 0u;
    uint32_t inc = t;
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t a = __shfl_up(inc, o);
      if ((int)threadIdx.x >= o) inc += a;
    }
    uint32_t run = inc - t;
    for (uint32_t k = 0; k < per; k++)
      if (lo + k < nb) {
        uint32_t v = s[lo + k];
        bsum[lo + k] = run;
        run += v;
      }
  }
}
__global__ __launch_bounds__(256) void k_hist_apply(uint32_t *__restrict__ hist, uint32_t m,
                                                    const uint32_t *__restrict__ bsum) {
  __shared__ uint32_t s[256];
  constexpr int per = kScanSpan / 256;  // consecutive entries per thread
  uint32_t base = blockIdx.x * kScanSpan + threadIdx.x * per;
  uint32_t v[per], t = 0;
  for (int k = 0; k < per; k++) {
    v[k] = base + k < m ? hist[base + k] : 0u;
    t += v[k];
  }
  s[threadIdx.x] = t;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan of the 256 thread sums
    uint32_t a = threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
    __syncthreads();
    s[threadIdx.x] += a;
    __syncthreads();
  }
  uint32_t run = bsum[blockIdx.x] + s[threadIdx.x] - t;
  for (int k = 0; k < per; k++) {
    if (base + k < m) hist[base + k] = run;
    run += v[k];
  }
}

// sorted[pos] = record i (tight D doubles); offs = exclusive prefix (advanced atomically)
template <int D>
__global__ __launch_bounds__(256) void k_scatter(const double *__restrict__ data, size_t stride,
                                                 size_t n, const uint32_t *__restrict__ keys,
                                                 uint32_t *__restrict__ offs,
                                                 double *__restrict__ sorted) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t pos = atomicAdd(&offs[keys[i]], 1u);
  for (int d = 0; d < D; d++) sorted[(size_t)pos * D + d] = data[i * stride + d];
}

__device__ inline float f32_up(double v) {  // smallest float >= v (v finite, >= 0)
  float f = (float)v;
  if ((double)f < v) f = nextafterf(f, INFINITY);
  return f;
}

// one wave per cell: conservative fp32 box of sorted[cell*cell_pts .. +cell_pts) ∩ [0, ns)
template <int D>
__global__ __launch_bounds__(256) void k_cell_boxes(const double *__restrict__ sorted, size_t ns,
                                                    uint32_t ncells, uint32_t cell_pts,
                                                    CellBox *__restrict__ boxes) {
  uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= ncells) return;
  const int lane = threadIdx.x & 63;
  double mn[D], mx[D];
  for (int d = 0; d < D; d++) mn[d] = __builtin_inf(), mx[d] = -__builtin_inf();
  for (uint32_t k = 0; k < cell_pts / 64; k++) {
    size_t i = (size_t)cell * cell_pts + k * 64 + lane;
    if (i < ns)
      for (int d = 0; d < D; d++) {
        double x = sorted[i * D + d];
        mn[d] = x < mn[d] ? x : mn[d];
        mx[d] = x > mx[d] ? x : mx[d];
      }
  }
  for (int d = 0; d < D; d++)
    for (int o = 32; o > 0; o >>= 1) {
      double a = __shfl_xor(mn[d], o), b = __shfl_xor(mx[d], o);
      mn[d] = a < mn[d] ? a : mn[d];
      mx[d] = b > mx[d] ? b : mx[d];
    }
  if (lane == 0) {
    CellBox bx;
    for (int d = 0; d < 3; d++) {
      if (d < D) {
        // 0.5*mn + 0.5*mx cannot overflow; the float centre may sit outside [mn, mx] by rounding,
        // the half extent is measured from the float centre so the box still covers every point
        float cf = (float)(0.5 * mn[d] + 0.5 * mx[d]);
        if (!(fabsf(cf) <= 3.4028234e38f)) cf = cf > 0 ? 3.4028234e38f : -3.4028234e38f;
        double hd = fmax(mx[d] - (double)cf, (double)cf - mn[d]);
        hd = hd * (1.0 + 9.5367431640625e-07) * (1.0 + 1e-12);
        float hf = (hd <= 3.4e38) ? f32_up(hd) : __builtin_inff();
        bx.c[d] = cf;
        bx.h[d] = hf;
      } else {
        bx.c[d] = 0.0f;
        bx.h[d] = 0.0f;
      }
    }
    bx.pad[0] = bx.pad[1] = 0.0f;
    boxes[cell] = bx;
  }
}

// ---- level-1 tests (lane = hypothesis; hf = the hypothesis' fp32 filter parameters) ---------------
// Each returns true unless NO point of the box can pass the model's candidate test |value| < tout
// (models.h: prepare_f32), with tc = tout * (1 + 2^-20) rounded up.
//
// plane: candidate value s32(x) = fma chain of n32.x32 - c32, |s32(x) - n.(x-a)| <= B < E (prepare_f32).
//   d32 = the same chain at the (fp32-exact) box centre, so |d32 - n.(ctr-a)| <= B;
//   r32 = fma(|n0|,h0, fma(|n1|,h1, fma(|n2|,h2, tc))) >= (sum|n_i|h_i + tc)(1 - 4u)
//                                                       >= sum|n_i| hx_i + tout      (hx_i = h_i/(1+2^-20), 4u < 2^-20)
//   A point x of the box with |n.(x-a)| < T + 1e-14 X (everything the reference can count) has
//   |n.(ctr-a)| < T + 1e-14 X + sum|n_i| hx_i, hence |d32| < T + B + 1e-14 X + sum|n_i| hx_i <= r32
//   because tout >= T + 1.01 B + 1e-12 X.  So "|d32| < r32 is false" => no point of the cell agrees.
//   tout = +inf (filter disabled) keeps every cell; tout = NaN (NaN model) drops every cell.
template <int D>
__device__ inline bool cell_survives(const PlaneModel<D> *, const float *hf, float tc,
                                     const CellBox &b) {
  float d = hf[3];
  float r = tc;
  if (D == 3) {
    d = __builtin_fmaf(hf[2], b.c[2], d);
    r = __builtin_fmaf(__builtin_fabsf(hf[2]), b.h[2], r);
  }
  d = __builtin_fmaf(hf[1], b.c[1], d);
  r = __builtin_fmaf(__builtin_fabsf(hf[1]), b.h[1], r);
  d = __builtin_fmaf(hf[0], b.c[0], d);
  r = __builtin_fmaf(__builtin_fabsf(hf[0]), b.h[0], r);
  return __builtin_fabsf(d) < r;
}

// ---- the scan ----------------------------------------------------------------------------------------
// A cell is 128*PP consecutive records of the sorted copy (PP packed pairs per lane); a wave tile is
// CPT cells whose observations stay in registers for the whole hypothesis loop.  Tiles are handed
// out through an atomic counter (near-model cells cost several times more than far ones, a static
// assignment leaves a long tail); the next 64 hypotheses' parameters are prefetched while the
// current 64 are processed.
template <class M, int PP, int CPT>
__global__ __launch_bounds__(256) void k_scan_cells(const double *__restrict__ sorted, size_t ns,
                                                    const CellBox *__restrict__ boxes,
                                                    uint32_t ncells, const double *__restrict__ sp,
                                                    const float *__restrict__ spf, uint32_t H,
                                                    ModelConsts mc, uint32_t *__restrict__ votes,
                                                    uint32_t *__restrict__ next_tile) {
  constexpr int D = M::ND;
  constexpr int NF = M::NF;
  constexpr int CP = 128 * PP;  // observations per cell
  constexpr int NV = M::SPF / 4;
  static_assert(M::SPF % 4 == 0 && 2 * NF + 2 <= M::SPF, "fp32 parameter block layout");
  extern __shared__ uint32_t s_cnt[];
  for (uint32_t h = threadIdx.x; h < H; h += 256) s_cnt[h] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t wtiles = (ncells + CPT - 1) / CPT;
  for (;;) {
    uint32_t wt = 0;
    if (lane == 0) wt = atomicAdd(next_tile, 1u);
    wt = __builtin_amdgcn_readfirstlane(wt);
    if (wt >= wtiles) break;
    v2f xs[CPT][PP][3];
    CellBox bx[CPT];
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      const uint32_t cell = wt * CPT + q;
#pragma unroll
      for (int p = 0; p < PP; p++) {
        const size_t i0 = (size_t)cell * CP + p * 128 + lane, i1 = i0 + 64;
        // rows past the end are NaN (never candidates); the loads themselves are unconditional
        const double *p0 = sorted + (i0 < ns ? i0 : 0) * D, *p1 = sorted + (i1 < ns ? i1 : 0) * D;
#pragma unroll
        for (int d = 0; d < 3; d++) {
          const float a0 = d < D ? (float)p0[d < D ? d : 0] : 0.0f;
          const float a1 = d < D ? (float)p1[d < D ? d : 0] : 0.0f;
          xs[q][p][d].x = (d < D && !(i0 < ns)) ? __builtin_nanf("") : a0;
          xs[q][p][d].y = (d < D && !(i1 < ns)) ? __builtin_nanf("") : a1;
        }
      }
      if (cell < ncells) {
        bx[q] = boxes[cell];  // wave-uniform address -> scalar load
      } else {
        for (int d = 0; d < 3; d++) bx[q].c[d] = 0.0f, bx[q].h[d] = __builtin_nanf("");  // never survives
      }
    }
    // the hypothesis' fp32 block {v0,v0, v1,v1, ..., tin, tout, 0, 0} as 16-byte loads
    float4 nxt[NV];
    {
      const float4 *f4 = (const float4 *)(spf + (size_t)((uint32_t)lane < H ? lane : 0) * M::SPF);
#pragma unroll
      for (int k = 0; k < NV; k++) nxt[k] = f4[k];
    }
    for (uint32_t h0 = 0; h0 < H; h0 += 64) {
      const uint32_t h = h0 + lane;
      float fl[M::SPF];
#pragma unroll
      for (int k = 0; k < NV; k++)
        fl[4 * k] = nxt[k].x, fl[4 * k + 1] = nxt[k].y, fl[4 * k + 2] = nxt[k].z, fl[4 * k + 3] = nxt[k].w;
      if (h0 + 64 < H) {  // prefetch the next group
        const uint32_t hn = h + 64;
        const float4 *f4 = (const float4 *)(spf + (size_t)(hn < H ? hn : 0) * M::SPF);
#pragma unroll
        for (int k = 0; k < NV; k++) nxt[k] = f4[k];
      }
      float hf[NF];
#pragma unroll
      for (int k = 0; k < NF; k++) hf[k] = fl[2 * k];
      const float tin = fl[2 * NF];
      const float tout = h < H ? fl[2 * NF + 1] : __builtin_nanf("");
      const float tc = tout * 1.00000096f;  // >= tout * (1 + 2^-20) after rounding
      uint32_t accv = 0;  // lane b: votes of hypothesis h0 + b collected from this tile
#pragma unroll
      for (int q = 0; q < CPT; q++) {
        unsigned long long surv = __ballot(cell_survives((const M *)nullptr, hf, tc, bx[q]));
        while (surv) {
          const int b = __builtin_ctzll(surv);
          asm("s_bitset0_b64 %0, %1" : "+s"(surv) : "s"(b));  // surv &= ~(1 << b)
          v2f fp[NF];
#pragma unroll
          for (int k = 0; k < NF; k++) {
            float v = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hf[k]), b));
            fp[k].x = v;
            fp[k].y = v;
          }
          const float btout =
              __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tout), b));
          // candidate test on the smallest |value| of the lane: NaN rows are ignored by min
          v2f s[PP];
          float m = __builtin_inff();
#pragma unroll
          for (int p = 0; p < PP; p++) {
            s[p] = M::filter_value(xs[q][p], fp);
            m = __builtin_fminf(m, __builtin_fminf(__builtin_fabsf(s[p].x), __builtin_fabsf(s[p].y)));
          }
          if (__ballot(m < btout) == 0) continue;
          const float btin =
              __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tin), b));
          unsigned long long in[2 * PP], amb = 0;
#pragma unroll
          for (int p = 0; p < PP; p++) {
            in[2 * p] = __ballot(__builtin_fabsf(s[p].x) < btin);
            in[2 * p + 1] = __ballot(__builtin_fabsf(s[p].y) < btin);
            amb |= (in[2 * p] ^ __ballot(__builtin_fabsf(s[p].x) < btout)) |
                   (in[2 * p + 1] ^ __ballot(__builtin_fabsf(s[p].y) < btout));
          }
          if (amb) {  // some observation sits in the band: exact fp64 predicate for the whole cell
            const double *hp = sp + (size_t)(h0 + b) * M::SP;  // wave-uniform -> scalar loads
#pragma unroll
            for (int p = 0; p < PP; p++) {
              const size_t i0 = (size_t)(wt * CPT + q) * CP + p * 128 + lane, i1 = i0 + 64;
              double r0[D], r1[D];
#pragma unroll
              for (int d = 0; d < D; d++) {
                r0[d] = i0 < ns ? sorted[i0 * D + d] : __builtin_nan("");
                r1[d] = i1 < ns ? sorted[i1 * D + d] : __builtin_nan("");
              }
              in[2 * p] = __ballot(M::agree(hp, r0, mc));
              in[2 * p + 1] = __ballot(M::agree(hp, r1, mc));
            }
          }
          uint32_t cnt = 0;
#pragma unroll
          for (int p = 0; p < 2 * PP; p++) cnt += (uint32_t)__builtin_popcountll(in[p]);
          cnt += (uint32_t)__builtin_amdgcn_readlane((int)accv, b);
          // v_writelane takes one SGPR on the constant bus: the lane select goes through m0
          asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0"
                       : "+v"(accv)
                       : "s"(cnt), "s"(b)
                       : "m0");
        }
      }
      if (accv) atomicAdd(&s_cnt[h], accv);  // h < H whenever accv != 0 (lanes past H never survive)
    }
  }
  __syncthreads();
  for (uint32_t h = threadIdx.x; h < H; h += 256) {
    uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[h], c);
  }
}

}  // namespace lsqr
