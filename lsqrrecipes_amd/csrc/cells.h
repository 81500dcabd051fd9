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
// Two drivers of the two levels:
//   k_scan_cells   plain scan (lsqr_scan, small batches): a wave owns a tile (cell) for the whole hypothesis loop,
//                  tiles handed out from shared atomic counters;
//   bounded scan   (batch entry points, >= 1024 hypotheses): k_cells_bounds gives every hypothesis an upper bound on
//                  its votes (level 1 alone, on boxes of four merged cells), a few pilots and then only the hypotheses
//                  that can still become the running maximum are counted by k_scan_pairs -- level 1 is run first to
//                  COUNT the surviving (hypothesis, cell) pairs, and every wave takes an equal share of them
//                  ("statically balanced level 2" below).
//
// Index build, once per upload: k_bounds (min / max / max |x| in one pass) -> k_keys (Morton key + identity) ->
// stable radix sort of the (key, index) pairs (sort.hip) -> k_gather_boxes (sorted copy + cell boxes).  No
// per-record global atomics anywhere: the build time does not depend on how the observations cluster.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "models.h"

#pragma clang diagnostic ignored "-Winline-asm"  // m0 is clobbered on purpose (k_scan_cells)

namespace lsqr {

constexpr int kCellPtsMin = 128;
constexpr uint32_t kGroupPad = 4, kCellPad = 16;  // k_scan_pairs: cost padding per group / per cell, in pairs (see there)
constexpr uint32_t kQueues = 32;   // work queues of k_scan_cells (one atomic counter each)
constexpr uint32_t kQueuePitch = 1088;  // uint32 words between counters: separate cache lines / channels  // cell sizes are multiples of one packed fp32 pair per lane

struct CellBox {  // 32 B: one s_load_dwordx8
  float c[3];     // centre (exactly representable, inside the box)
  float h[3];     // half extents, inflated: every point satisfies |x_i - c_i| * (1 + 2^-20) <= h_i
  float pad[2];   // pad[0]: bounding radius |h|, rounded up
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

// ---- pass 1: bounds -------------------------------------------------------------------------------------
// One pass over the records gives everything the upload needs to know about its coordinates: per-dimension
// min / max of the finite records (ordered encoding) for the Morton grid, the number of non-finite records
// (they can never agree and are sorted to the tail), and max |coordinate| -- the X of the fp32 filters' error
// bands (k_absmax's result: +inf as soon as one record is not finite, which switches the filters off).
// Two stages: one row of 8 values per block, then a single-block reduction (same-address atomics cost
// ~14 ns each: a few thousand of them on one cache line would take longer than the pass itself).
// 46 us for 240 MB at 10 M points = 0.65 of the 8 TB/s peak (r04, HIP events around the kernel: tools/bounds_time.py;
// r03 quoted 67 us / 0.44 because its profile scope included the final reduction, the copy back and the host
// synchronisation).  A flat variant -- the array as one run of 16-byte pieces, fully coalesced, two (min, max) pairs per
// lane -- was built and measured no faster (49 - 52 us): the record-per-lane reads are not what limits the pass.
struct BoundsRow {
  unsigned long long mn[3], mx[3], amax, nonfinite;
};
template <int D>
__global__ __launch_bounds__(256) void k_bounds(const double *__restrict__ data, size_t stride,
                                                size_t n, BoundsRow *__restrict__ rows) {
  unsigned long long mn[D], mx[D], am = 0, bad = 0;
  for (int d = 0; d < D; d++) mn[d] = ~0ULL, mx[d] = 0ULL;
  auto one = [&](const double(&x)[D]) {
    bool fin = true;
    for (int d = 0; d < D; d++) fin = fin && (fabs(x[d]) <= 1.7976931348623157e308);  // false for NaN and inf
    if (!fin) {
      bad++;
      return;
    }
    for (int d = 0; d < D; d++) {
      unsigned long long o = ord_u64(x[d]);
      mn[d] = o < mn[d] ? o : mn[d];
      mx[d] = o > mx[d] ? o : mx[d];
      const unsigned long long a = (unsigned long long)__builtin_bit_cast(long long, fabs(x[d]));
      am = a > am ? a : am;  // bit patterns of non-negative doubles are ordered like the values
    }
  };
  // four records per lane in flight (a minimum / maximum does not care about the order)
  const size_t step = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < n; i += 4 * step) {
    double x[4][D];
#pragma unroll
    for (int u = 0; u < 4; u++)
      for (int d = 0; d < D; d++) x[u][d] = data[(i + u * step) * stride + d];
#pragma unroll
    for (int u = 0; u < 4; u++) one(x[u]);
  }
  for (; i < n; i += step) {
    double x[D];
    for (int d = 0; d < D; d++) x[d] = data[i * stride + d];
    one(x);
  }
  __shared__ unsigned long long s_v[4][2 * D + 2];
  for (int o = 32; o > 0; o >>= 1) {
    for (int d = 0; d < D; d++) {
      unsigned long long a = __shfl_down(mn[d], o), b = __shfl_down(mx[d], o);
      mn[d] = a < mn[d] ? a : mn[d];
      mx[d] = b > mx[d] ? b : mx[d];
    }
    unsigned long long a = __shfl_down(am, o);
    am = a > am ? a : am;
    bad += __shfl_down(bad, o);
  }
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *v = s_v[threadIdx.x >> 6];
    for (int d = 0; d < D; d++) v[d] = mn[d], v[D + d] = mx[d];
    v[2 * D] = am;
    v[2 * D + 1] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    BoundsRow r;
    for (int d = 0; d < 3; d++) r.mn[d] = ~0ULL, r.mx[d] = 0ULL;
    r.amax = 0, r.nonfinite = 0;
    for (int w = 0; w < 4; w++) {
      for (int d = 0; d < D; d++) {
        r.mn[d] = s_v[w][d] < r.mn[d] ? s_v[w][d] : r.mn[d];
        r.mx[d] = s_v[w][D + d] > r.mx[d] ? s_v[w][D + d] : r.mx[d];
      }
      r.amax = s_v[w][2 * D] > r.amax ? s_v[w][2 * D] : r.amax;
      r.nonfinite += s_v[w][2 * D + 1];
    }
    rows[blockIdx.x] = r;
  }
}
__global__ __launch_bounds__(256) void k_bounds_final(const BoundsRow *__restrict__ rows, int nb,
                                                      BoundsRow *__restrict__ out) {
  BoundsRow r;
  for (int d = 0; d < 3; d++) r.mn[d] = ~0ULL, r.mx[d] = 0ULL;
  r.amax = 0, r.nonfinite = 0;
  for (int b = threadIdx.x; b < nb; b += 256) {
    const BoundsRow q = rows[b];
    for (int d = 0; d < 3; d++) {
      r.mn[d] = q.mn[d] < r.mn[d] ? q.mn[d] : r.mn[d];
      r.mx[d] = q.mx[d] > r.mx[d] ? q.mx[d] : r.mx[d];
    }
    r.amax = q.amax > r.amax ? q.amax : r.amax;
    r.nonfinite += q.nonfinite;
  }
  __shared__ BoundsRow s_r[4];
  for (int o = 32; o > 0; o >>= 1) {
    for (int d = 0; d < 3; d++) {
      unsigned long long a = __shfl_down(r.mn[d], o), b = __shfl_down(r.mx[d], o);
      r.mn[d] = a < r.mn[d] ? a : r.mn[d];
      r.mx[d] = b > r.mx[d] ? b : r.mx[d];
    }
    unsigned long long a = __shfl_down(r.amax, o);
    r.amax = a > r.amax ? a : r.amax;
    r.nonfinite += __shfl_down(r.nonfinite, o);
  }
  if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      for (int d = 0; d < 3; d++) {
        r.mn[d] = s_r[w].mn[d] < r.mn[d] ? s_r[w].mn[d] : r.mn[d];
        r.mx[d] = s_r[w].mx[d] > r.mx[d] ? s_r[w].mx[d] : r.mx[d];
      }
      r.amax = s_r[w].amax > r.amax ? s_r[w].amax : r.amax;
      r.nonfinite += s_r[w].nonfinite;
    }
    *out = r;
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

// ---- pass 2: Morton key of every record (non-finite records: key nbins, i.e. after every finite one) and the
// identity permutation; the (key, index) pairs are then sorted by a stable radix sort (sort.hip)
template <int D>
__global__ __launch_bounds__(256) void k_keys(const double *__restrict__ data, size_t stride,
                                              size_t n, IndexGrid g, uint32_t *__restrict__ keys,
                                              uint32_t *__restrict__ vals, int presorted) {
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
  else if (presorted) key = 0;  // "scan_presorted": the upload order IS the cell order (stable sort: finite records keep it)
  else if (D == 3) key = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
  else key = spread2(q[0]) | (spread2(q[1]) << 1);
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// ---- pass 2b (r04): local k-d refinement of the Morton order ----------------------------------------------------------
// A cell is a run of 128 * PP consecutive records of the sorted copy, and a run of a Z-order curve straddles octant
// boundaries at every level: its bounding box is 2 - 3 times the volume of a compact region holding as many records, and
// the first level of the scan lets that many more (hypothesis, cell) pairs through to the second.  Measured on the
// bench's uploads (tools/ab_order_kd.py, host-side k-d order against the library's): plane 10.3 M -> 8.8 M surviving
// pairs per 4096 hypotheses (full count 0.96 -> 0.81 ms), sphere 22.5 M -> 11.7 M (1.73 -> 1.24 ms), line 20.2 M ->
// 13.4 M (2.28 -> 1.66 ms).  A global k-d build is ~15 segmented sorts; what is built instead keeps the Morton radix
// sort as the coarse order and re-partitions every SUPER-RUN of kRunPts = 8192 consecutive records (16 cells of 512, 32
// of 256) as a k-d tree, one workgroup per run, entirely in LDS: per level every segment takes the axis of its widest
// extent and is split at its median along it (radix select + stable partition, below).  Only the permutation changes: votes do not
// depend on the order of the observations (every parity test runs on the refined order), the boxes are formed from
// the gathered records as before.  Simulation at 2 M points: Morton 19.9 % of the (plane, cell) pairs survive, runs of
// 8192 refined 16.2 %, a full k-d partition 15.0 %.
constexpr uint32_t kRunPts = 8192;
__device__ __forceinline__ uint32_t ord_u32(float v) {  // monotone map float -> uint32 (NaN above +inf)
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_u32_inv(uint32_t o) {
  const uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __builtin_bit_cast(float, b);
}

// ---- pass 2a (r05): k-d levels ABOVE the runs, by segmented sorts ------------------------------------------------------
// The refinement below re-partitions runs of 8192 records; the Morton order above them still straddles octant
// boundaries.  Simulated at the bench's 10 M points (tools/kd_order_sim.py, share of (random plane, cell) pairs that
// survive level 1): Morton runs 11.61 %, + k-d inside runs of 8192 9.86 %, of 65 536 9.53 %, of 262 144 9.37 %, of 1 M
// 9.28 %, the exact k-d partition 9.32 %.  Medians have to be exact here (regions cut at sampled medians do not align
// with the 8192-record runs of the stage behind them and come out WORSE than the plain refinement), so a level is a
// SORT: every segment of S consecutive positions of the current order (S = 2^k, aligned) takes the axis of its widest
// extent (k_seg_extent), every record gets the key (segment << 16) | its coordinate along that axis in 65 535 steps of
// the extent (k_seg_keys), one stable radix sort of the (key, index) pairs orders every segment along its axis, and the
// next level's segments are the halves by position.  `kd_levels` levels from S = 8192 << kd_levels down to 16 384.
struct SegExtent {
  uint32_t lo[3], hi[3];  // ord_u32 of the fp32 coordinates
};
// (float)double one value at a time: hipcc 7.2 crashes in instruction selection ("AMDGPU DAG->DAG") on the
// <2 x double> -> <2 x float> conversion the vectoriser forms from two neighbouring ones in these kernels
__device__ __forceinline__ float seg_f32(double d) {
  asm("" : "+v"(d));
  return (float)d;
}
// ... and ord_u32_inv likewise (the select on a <2 x i32> formed from two neighbouring extents)
__device__ __forceinline__ float seg_ord_inv(uint32_t o) {
  asm("" : "+v"(o));
  return ord_u32_inv(o);
}
__global__ __launch_bounds__(256) void k_seg_extent_init(SegExtent *__restrict__ ext, uint32_t nseg) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nseg) return;
  for (int d = 0; d < 3; d++) ext[s].lo[d] = 0xFFFFFFFFu, ext[s].hi[d] = 0u;
}
// The levels work on a compact copy of the coordinates in MORTON order (xyz: D floats per record, k_seg_gather) and on
// positions q into it: a segment's records sit in one window of that copy (3 - 12 MB: the L2s hold it), where the
// original records are 24 B each anywhere in the upload (the first version read them through the permutation: two
// random gathers of 240 MB per level, 0.44 of a level's 0.8 ms).  k_seg_compose turns the final positions back into
// record indices.
template <int D>
__global__ __launch_bounds__(256) void k_seg_gather(const double *__restrict__ data, size_t stride,
                                                    const uint32_t *__restrict__ perm, size_t n, size_t ns,
                                                    float *__restrict__ xyz, uint32_t *__restrict__ q) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  q[i] = (uint32_t)i;
  if (i >= ns) return;
  const double *p = data + (size_t)perm[i] * stride;
  for (int d = 0; d < D; d++) xyz[i * D + d] = seg_f32(p[d]);
}
__global__ __launch_bounds__(256) void k_seg_compose(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ q,
                                                     size_t n, uint32_t *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = perm[q[i]];
}
// one workgroup per 1024 consecutive positions (S is a multiple of 1024: all of one segment)
template <int D>
__global__ __launch_bounds__(256) void k_seg_extent(const float *__restrict__ xyz, const uint32_t *__restrict__ q, size_t ns,
                                                    uint32_t seg_shift, SegExtent *__restrict__ ext) {
  __shared__ uint32_t s_lo[D], s_hi[D];
  if (threadIdx.x < D) s_lo[threadIdx.x] = 0xFFFFFFFFu, s_hi[threadIdx.x] = 0u;
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * 1024;
  uint32_t lo[D], hi[D];
  for (int d = 0; d < D; d++) lo[d] = 0xFFFFFFFFu, hi[d] = 0u;
  for (int u = 0; u < 4; u++) {
    const size_t i = base + (size_t)u * 256 + threadIdx.x;
    if (i < ns) {
      const float *p = xyz + (size_t)q[i] * D;
      for (int d = 0; d < D; d++) {
        const uint32_t o = ord_u32(p[d]);
        lo[d] = o < lo[d] ? o : lo[d];
        hi[d] = o > hi[d] ? o : hi[d];
      }
    }
  }
  for (int d = 0; d < D; d++) {
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t a = __shfl_xor(lo[d], off), b = __shfl_xor(hi[d], off);
      lo[d] = a < lo[d] ? a : lo[d];
      hi[d] = b > hi[d] ? b : hi[d];
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&s_lo[d], lo[d]);
      atomicMax(&s_hi[d], hi[d]);
    }
  }
  __syncthreads();
  if (threadIdx.x < D && base < ns) {
    SegExtent *e = ext + (base >> seg_shift);
    atomicMin(&e->lo[threadIdx.x], s_lo[threadIdx.x]);
    atomicMax(&e->hi[threadIdx.x], s_hi[threadIdx.x]);
  }
}
// key = (segment << 16) | coordinate along the segment's widest axis; positions past the finite records keep their
// place behind everything (segment number nseg)
template <int D>
__global__ __launch_bounds__(256) void k_seg_keys(const float *__restrict__ xyz, const uint32_t *__restrict__ q, size_t n,
                                                  size_t ns, uint32_t seg_shift, uint32_t nseg,
                                                  const SegExtent *__restrict__ ext, uint32_t *__restrict__ keys,
                                                  uint32_t *__restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t pos = q[i];
  vals[i] = pos;
  if (i >= ns) {
    keys[i] = nseg << 16;
    return;
  }
  const uint32_t seg = (uint32_t)(i >> seg_shift);
  const SegExtent *e = ext + seg;
  const float *p = xyz + (size_t)pos * D;
  // the widest axis (ties: the lowest) and the record's coordinate along it
  float lo = seg_ord_inv(e->lo[0]), w = seg_ord_inv(e->hi[0]) - lo, x = p[0];
  {
    const float l1 = seg_ord_inv(e->lo[1]), w1 = seg_ord_inv(e->hi[1]) - l1, x1 = p[1];
    const bool take = w1 > w;
    lo = take ? l1 : lo, x = take ? x1 : x, w = take ? w1 : w;
  }
  if (D == 3) {
    const float l2 = seg_ord_inv(e->lo[2]), w2 = seg_ord_inv(e->hi[2]) - l2, x2 = p[D - 1];
    const bool take = w2 > w;
    lo = take ? l2 : lo, x = take ? x2 : x, w = take ? w2 : w;
  }
  float t = w > 0.0f ? (x - lo) / w * 65535.0f : 0.0f;
  t = t > 0.0f ? (t < 65535.0f ? t : 65535.0f) : 0.0f;  // (NaN cannot occur: the records below ns are finite)
  keys[i] = (seg << 16) | (uint32_t)t;
}

constexpr uint32_t kRunMaxSeg = 16;  // segments of a level: runs of kRunPts are split down to 512-record segments
__host__ __device__ inline size_t refine_lds_bytes(int D) {
  return (size_t)kRunPts * (sizeof(float) * D + 2 * sizeof(uint16_t)) + kRunMaxSeg * 256 * sizeof(uint32_t) + 1024;
}
// Position p = 8 t + r belongs to thread t (r = 0..7): a wave covers 512 consecutive positions, a segment (>= 512 long,
// a power of two) is a whole number of waves.  Per level and segment: widest axis -> 16-bit keys -> the key K of
// rank segn/2 - 1 by radix select (two 8-bit passes over per-segment LDS histograms) -> stable partition into
// {key < K, the first t_eq keys equal to K} | rest, each side keeping its order: exactly segn/2 records left, ties
// split by position, so the result is deterministic.  O(n) per level (a sort per level, the first version of this
// kernel, took 2.2 ms per 10 M records; this takes ~0.2).
template <int D>
__global__ __launch_bounds__(1024) void k_refine_runs(const double *__restrict__ data, size_t stride,
                                                      uint32_t *__restrict__ perm, size_t ns, uint32_t cell_pts) {
  extern __shared__ unsigned char refine_smem[];
  float *cx = (float *)refine_smem;                          // [D][kRunPts], by position in the ORIGINAL run
  uint16_t *idxa = (uint16_t *)(cx + (size_t)D * kRunPts);    // [2][kRunPts]: current / next order
  uint32_t *hist = (uint32_t *)(idxa + 2 * kRunPts);          // [kRunMaxSeg][256]
  uint32_t *sbox = hist + kRunMaxSeg * 256;                   // [kRunMaxSeg][3][2]
  uint32_t *spref = sbox + kRunMaxSeg * 6;                    // [kRunMaxSeg] key prefix found so far
  uint32_t *srk = spref + kRunMaxSeg;                         // [kRunMaxSeg] rank still to resolve inside the prefix
  uint32_t *saxis = srk + kRunMaxSeg;                         // [kRunMaxSeg]
  uint32_t *wtot = saxis + kRunMaxSeg;                        // [16 waves]
  const size_t base = (size_t)blockIdx.x * kRunPts;
  if (base >= ns) return;
  const uint32_t m = (uint32_t)(ns - base < kRunPts ? ns - base : kRunPts);  // records of this run; the rest is padding
  const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
  uint32_t mine[kRunPts / 1024];
#pragma unroll
  for (uint32_t k = 0; k < kRunPts / 1024; k++) {
    const uint32_t j = k * 1024 + t;
    uint32_t src = 0;
    if (j < m) {
      src = perm[base + j];
      const double *r = data + (size_t)src * stride;
#pragma unroll
      for (int d = 0; d < D; d++) cx[(size_t)d * kRunPts + j] = (float)r[d];
    } else {
#pragma unroll
      for (int d = 0; d < D; d++) cx[(size_t)d * kRunPts + j] = __builtin_inff();
    }
    mine[k] = src;
    idxa[j] = (uint16_t)j;
  }
  __syncthreads();
  uint32_t cur = 0;
  const uint32_t stop = cell_pts < 256 ? 256u : cell_pts;  // a segment is >= one wave (512 positions) when it is split:
                                                           // 128-record cells keep 256-record leaves
  for (uint32_t segn = kRunPts; segn > stop; segn >>= 1) {
    const uint16_t *idx = idxa + cur * kRunPts;
    uint16_t *idn = idxa + (cur ^ 1) * kRunPts;
    const uint32_t nseg = kRunPts / segn, wps = segn / 512;  // waves per segment
    const uint32_t sg = wave / wps, p0 = t * 8;
    uint32_t e[8];
    {
      const uint4 raw = *(const uint4 *)(idx + p0);             // 8 x u16
      e[0] = raw.x & 0xFFFFu, e[1] = raw.x >> 16, e[2] = raw.y & 0xFFFFu, e[3] = raw.y >> 16;
      e[4] = raw.z & 0xFFFFu, e[5] = raw.z >> 16, e[6] = raw.w & 0xFFFFu, e[7] = raw.w >> 16;
    }
    for (uint32_t k = t; k < nseg * 6; k += 1024) sbox[k] = (k & 1) ? 0u : 0xFFFFFFFFu;
    __syncthreads();
    // ---- extent of every segment over its records (padding sits at the tail of its segment at every level)
#pragma unroll
    for (int d = 0; d < D; d++) {
      uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
      for (int r = 0; r < 8; r++)
        if (e[r] < m) {
          const uint32_t u = ord_u32(cx[(size_t)d * kRunPts + e[r]]);
          lo = u < lo ? u : lo;
          hi = u > hi ? u : hi;
        }
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(lo, o), b = __shfl_xor(hi, o);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
      }
      if (lane == 0 && hi >= lo) {
        atomicMin(&sbox[(sg * 3 + d) * 2], lo);
        atomicMax(&sbox[(sg * 3 + d) * 2 + 1], hi);
      }
    }
    __syncthreads();
    if (t < nseg) {
      int ax = 0;
      float best = -1.0f;
#pragma unroll
      for (int d = 0; d < D; d++) {
        const uint32_t lo = sbox[(t * 3 + d) * 2], hi = sbox[(t * 3 + d) * 2 + 1];
        const float ext = hi >= lo ? ord_u32_inv(hi) - ord_u32_inv(lo) : -1.0f;  // (segment of padding only: -1)
        if (ext > best) best = ext, ax = d;
      }
      saxis[t] = (uint32_t)ax;
      spref[t] = 0;
      srk[t] = segn / 2 - 1;
    }
    __syncthreads();
    // keys: the coordinate along the segment's axis in 65535 steps of the segment's extent (padding: 65535, above every
    // record).  A median found on these keys is off by at most extent / 65535 -- nothing for the shape of the cells --
    // and two 8-bit passes select it where the 32-bit key took four (the kernel is bound by its barriers).
    uint32_t key[8];
    {
      const uint32_t ax = saxis[sg];
      const float *ca = cx + (size_t)ax * kRunPts;
      const uint32_t ulo = sbox[(sg * 3 + ax) * 2], uhi = sbox[(sg * 3 + ax) * 2 + 1];
      const float lo = uhi >= ulo ? ord_u32_inv(ulo) : 0.0f, ext = uhi >= ulo ? ord_u32_inv(uhi) - lo : 0.0f;
      const float sc = ext > 0.0f && ext < 3.0e38f ? 65534.0f / ext : 0.0f;
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const float q = (ca[e[r] < m ? e[r] : 0] - lo) * sc;
        const uint32_t qi = q >= 65534.0f ? 65534u : (q > 0.0f ? (uint32_t)q : 0u);   // (NaN -> 0)
        key[r] = e[r] < m ? qi : 65535u;
      }
    }
    // ---- radix select: the key of rank segn/2 - 1 of every segment
    for (int shift = 8; shift >= 0; shift -= 8) {
      for (uint32_t k = t; k < nseg * 256; k += 1024) hist[k] = 0;
      __syncthreads();
      {
        const uint32_t pref = spref[sg], himask = shift == 8 ? 0u : 0xFF00u;
#pragma unroll
        for (int r = 0; r < 8; r++)
          if ((key[r] & himask) == pref) atomicAdd(&hist[sg * 256 + ((key[r] >> shift) & 255u)], 1u);
      }
      __syncthreads();
      if (wave < nseg) {  // wave s resolves segment s: lane l owns digits 4l .. 4l + 3
        const uint32_t *h = hist + wave * 256 + 4 * lane;
        const uint32_t c0 = h[0], c1 = h[1], c2 = h[2], c3 = h[3];
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t inc = tot;
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t a = __shfl_up(inc, o);
          inc += (int)lane >= o ? a : 0u;
        }
        const uint32_t before = inc - tot, rk = srk[wave];
        const bool here = rk >= before && rk < inc;   // exactly one lane (the candidates number more than rk)
        if (here) {
          uint32_t b = 0, cb = before;
          if (rk >= cb + c0) {
            cb += c0, b = 1;
            if (rk >= cb + c1) {
              cb += c1, b = 2;
              if (rk >= cb + c2) cb += c2, b = 3;
            }
          }
          spref[wave] |= (4 * lane + b) << shift;
          srk[wave] = rk - cb;
        }
      }
      __syncthreads();
    }
    // ---- stable partition: {key < K, the first srk + 1 keys equal to K} go left
    const uint32_t K = spref[sg], teq = srk[sg] + 1;
    uint32_t nl = 0, ne = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) ne += key[r] == K ? 1u : 0u;
    // exclusive scan of the equal-key counts over the segment (position order)
    uint32_t inc = ne;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t a = __shfl_up(inc, o);
      inc += (int)lane >= o ? a : 0u;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t eqb = inc - ne;
    for (uint32_t w = sg * wps; w < wave; w++) eqb += wtot[w];
    __syncthreads();
    bool left[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const bool eq = key[r] == K;
      left[r] = key[r] < K || (eq && eqb < teq);
      eqb += eq ? 1u : 0u;
      nl += left[r] ? 1u : 0u;
    }
    inc = nl;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t a = __shfl_up(inc, o);
      inc += (int)lane >= o ? a : 0u;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t lb = inc - nl;                                   // lefts before my first position, in the segment
    for (uint32_t w = sg * wps; w < wave; w++) lb += wtot[w];
    const uint32_t sbase = sg * segn;
    uint32_t rb = (p0 - sbase) - lb;                          // rights before my first position
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const uint32_t dst = left[r] ? sbase + lb : sbase + segn / 2 + rb;
      idn[dst] = (uint16_t)e[r];
      lb += left[r] ? 1u : 0u;
      rb += left[r] ? 0u : 1u;
    }
    __syncthreads();
    cur ^= 1;
  }
  // the refined order of the run: position j takes the record that sat at position idx[j] of the Morton run
  const uint16_t *idx = idxa + cur * kRunPts;
  uint32_t *srcs = (uint32_t *)cx;  // (the coordinates are not needed any more)
#pragma unroll
  for (uint32_t k = 0; k < kRunPts / 1024; k++) srcs[k * 1024 + t] = mine[k];
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < kRunPts / 1024; k++) {
    const uint32_t j = k * 1024 + t;
    if (j < m) perm[base + j] = srcs[idx[j]];
  }
}

__device__ inline float f32_up(double v) {  // smallest float >= v (v finite, >= 0)
  float f = (float)v;
  if ((double)f < v) f = nextafterf(f, INFINITY);
  return f;
}

// ---- pass 3: one wave per cell gathers its records in sorted order (order[] = the sorted record indices) into
// the tight copy and, from the same registers, forms the cell's conservative fp32 box
template <int D>
__global__ __launch_bounds__(256) void k_gather_boxes(const double *__restrict__ data, size_t stride,
                                                      const uint32_t *__restrict__ order, size_t ns,
                                                      uint32_t ncells, uint32_t cell_pts,
                                                      double *__restrict__ sorted,
                                                      CellBox *__restrict__ boxes) {
  uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= ncells) return;
  const int lane = threadIdx.x & 63;
  double mn[D], mx[D];
  for (int d = 0; d < D; d++) mn[d] = __builtin_inf(), mx[d] = -__builtin_inf();
  for (uint32_t k = 0; k < cell_pts / 64; k++) {
    size_t i = (size_t)cell * cell_pts + k * 64 + lane;
    if (i < ns) {
      const double *src = data + (size_t)order[i] * stride;
      for (int d = 0; d < D; d++) {
        double x = src[d];
        sorted[i * D + d] = x;
        mn[d] = x < mn[d] ? x : mn[d];
        mx[d] = x > mx[d] ? x : mx[d];
      }
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
    // bounding radius |h| (rounded up): every point of the cell is within pad[0] of the centre
    double r2 = 0.0;
    for (int d = 0; d < D; d++) r2 += (double)bx.h[d] * (double)bx.h[d];
    const double rad = sqrt(r2) * (1.0 + 1e-6);
    bx.pad[0] = rad <= 3.4e38 ? f32_up(rad) : __builtin_inff();
    bx.pad[1] = 0.0f;
    boxes[cell] = bx;
  }
}

// ---- coarser boxes for the vote bounds: one box per `merge` consecutive cells (Morton order keeps them close
// together).  The bounds pass of the bounded scan only has to tell near-model hypotheses from the rest: with four
// cells per box it evaluates a quarter of the (hypothesis, box) tests, and the bound -- the population of the
// surviving boxes -- stays valid (a merged box contains its cells' boxes) if a little looser.
template <int D>
__global__ __launch_bounds__(256) void k_super_boxes(const CellBox *__restrict__ boxes, uint32_t ncells,
                                                     CellBox *__restrict__ sboxes, uint32_t nsuper, uint32_t merge) {
  const uint32_t sc = blockIdx.x * 256 + threadIdx.x;
  if (sc >= nsuper) return;
  double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()},
         hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
  bool nan = false;
  for (uint32_t c = sc * merge; c < (sc + 1) * merge && c < ncells; c++) {
    const CellBox b = boxes[c];
    for (int d = 0; d < D; d++) {  // the child box covers [c - h, c + h] (h already inflated); exact in fp64
      const double l = (double)b.c[d] - (double)b.h[d], u = (double)b.c[d] + (double)b.h[d];
      nan = nan || !(l == l) || !(u == u);
      lo[d] = l < lo[d] ? l : lo[d];
      hi[d] = u > hi[d] ? u : hi[d];
    }
  }
  CellBox sb;
  double r2 = 0.0;
  for (int d = 0; d < 3; d++) {
    if (d < D) {
      float cf = (float)(0.5 * lo[d] + 0.5 * hi[d]);
      if (!(__builtin_fabsf(cf) <= 3.4028234e38f)) cf = cf > 0 ? 3.4028234e38f : -3.4028234e38f;
      const double hd = fmax(hi[d] - (double)cf, (double)cf - lo[d]) * (1.0 + 1e-12);
      sb.c[d] = cf;
      // a box nothing is known about (NaN child, overflow) must SURVIVE every test of a bound: the largest finite
      // extent (not inf: 0 * inf would poison the plane's margin).  Cannot happen on an indexed upload (the index is
      // only used when max |coordinate| <= 1e15 and holds finite records only); kept conservative all the same.
      sb.h[d] = (!nan && hd <= 3.4e38) ? f32_up(hd) : 3.4028234e38f;
      r2 += (double)sb.h[d] * (double)sb.h[d];
    } else {
      sb.c[d] = 0.0f;
      sb.h[d] = 0.0f;
    }
  }
  const double rad = sqrt(r2) * (1.0 + 1e-6);
  sb.pad[0] = rad <= 3.4e38 ? f32_up(rad) : 3.4028234e38f;
  sb.pad[1] = 0.0f;
  sboxes[sc] = sb;
}

// ---- per-model cell logic ------------------------------------------------------------------------------
// A cell model CM provides
//   Hyp                      per-lane hypothesis state (lane = hypothesis), built by load() from the
//                            hypothesis' fp64 scan parameters (M::SP doubles)
//   level1(hyp, box, cc, bc) conservative test "can any observation inside the box agree?" and the
//                            NB per-(hypothesis, cell) values bc[] the second level needs:
//                            bc[0..NV) feed value(), bc[NB-2] = tin, bc[NB-1] = tout
//   value(xs, fp)            packed fp32 filter measure v of two observations with the guarantee
//                            |v| < tin => agrees,  |v| >= tout => does not agree   (exact fp64 predicate
//                            decides in between)
// CellConsts are per-launch constants derived on the host (cell_consts<CM>()).
struct CellConsts {
  float f[8];
};
// an fp64 value fetched as two 32-bit words of a float row
__device__ inline double f64_from(const float *w) {
  const unsigned long long bits = (unsigned long long)__builtin_bit_cast(uint32_t, w[0]) |
                                  ((unsigned long long)__builtin_bit_cast(uint32_t, w[1]) << 32);
  return __builtin_bit_cast(double, bits);
}

// Plane.  Observations are stored relative to the (fp32-exact) cell centre, x' = fl32(x - ctr), and the
// level-1 pass evaluates d0 = n.ctr - n.a in fp64, so the fp32 arithmetic only ever sees cell-sized
// numbers.  With u = 2^-24, h_i the half extents, rr* = sum|n_i|h_i, s* = n.(x-a) = n.x' + d0:
//   s32 = fma(n32_0,x'_0, fma(n32_1,x'_1, fma(n32_2,x'_2, fl32(d0))))
//   |s32 - s*| <= u rr* (x' rounding) + u rr* (n rounding) + u|d0| (d0 rounding)
//                 + 3u(|d0| + rr*)(1+4u) (three fma results) + fp64 noise
//              <= u(5.1 rr* + 4.1|d0|) + 2e-12 X
// and a cell that passed level 1 has |d0| <= (rr + tout)(1 + 2^-18), so for every observation of a
// surviving cell |s32 - s*| <= u(9.3 rr + 4.2 tout) + 2e-12 X <= E := 1.01u(9.4 rr + 4.3 T) + 3e-12 X.
// The reference's fp64 s differs from s* by < 1e-13 X.  Hence with tin = (Tdn - E)(1 - 2^-22),
// tout = (Tup + E)(1 + 2^-22) (Tdn <= T <= Tup the fp32 neighbours of the exact threshold T):
//   |s32| < tin => |s_ref| < T (agrees);   |s32| >= tout => |s_ref| >= T (does not).
// Level 1: an observation of the box with |s_ref| < T has |d0| < T + rr* + 1e-13 X, so
// |fl32(d0)| < (rr + tout)(1 + 2^-19); "false" therefore proves that nothing in the cell agrees.
// A NaN model (or a lane past the batch) gives d0 = NaN: never survives.  |n_i| > 1 (never produced by
// estimate()) sets E = inf: every cell survives and every observation takes the exact path.
template <int D>
struct PlaneCell {
  typedef PlaneModel<D> M;
  enum { NB = 6, NV = 4, RELATIVE = 1, ROW = 4 * D, ROW_F32 = 0, ROW2 = 0, ROW2_OFF = 0 };  // row = fp64 scan parameters
  enum { DEFAULT_CELL = 512, LDS_BROADCAST = 0, MIN_WAVES = 6 };  // measured best (tools/ab_cells.py)
  // k_scan_pairs counts near-model hypotheses, 80 % of whose (hypothesis, cell) pairs take the full path: there the
  // six v_readlane per pair are vector-issue slots the LDS broadcast gives back (0.84 -> 0.77 ms per step)
  enum { LDS_BROADCAST_PAIRS = 1 };
  enum { FULL_COUNT_PAIRS = 1 };  // plain scans of >= 1024 hypotheses through k_scan_pairs too (r03: 1.24 -> 1.13 ms)
  // (1024-point cells / 8 packed pairs per lane, 4 waves per SIMD: second pass of the bounded scan 642 us against
  // 510 us -- measured and not built; `enum { MAX_PP = 8 };` here brings the instantiation back)
  enum { USE_BOUND = 1 };  // bounded scan pays: a random plane still cuts ~13 % of the cells
  // bounds on boxes of four cells: 47 -> 15 us for the bounds pass, the selection hardly changes (0.73 -> 0.71 ms / step)
  enum { BOUND_MERGE = 4 };
  struct Hyp {
    double n[3], c;
    float nf[3], e0;
  };
  static __device__ inline void load(const float *row, const float *, bool valid,
                                     const CellConsts &cc, Hyp &h) {
    double r[2 * D];
#pragma unroll
    for (int i = 0; i < 2 * D; i++) r[i] = f64_from(row + 2 * i);
    bool ok = true;
    h.c = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double ni = i < D ? (valid ? r[i] : __builtin_nan("")) : 0.0;
      h.n[i] = ni;
      h.nf[i] = (float)ni;
      if (i < D) {
        h.c += ni * r[D + i];
        ok = ok && fabs(ni) <= 1.0000001;
      }
    }
    h.e0 = ok ? cc.f[2] : __builtin_inff();
  }
  static __device__ inline bool level1(const Hyp &h, const CellBox &b, const double *ctr,
                                       const CellConsts &cc, float *bc) {
    double d = -h.c;
    float rr = 0.0f;
#pragma unroll
    for (int i = D - 1; i >= 0; i--) {
      d = fma(h.n[i], ctr[i], d);
      rr = __builtin_fmaf(__builtin_fabsf(h.nf[i]), b.h[i], rr);
    }
    const float d0 = (float)d;
    const float E = __builtin_fmaf(rr, cc.f[3], h.e0);
    const float tout = (cc.f[1] + E) * 1.0000003f;
    const float tin = (cc.f[0] - E) * 0.9999997f;
    bc[0] = h.nf[0], bc[1] = h.nf[1], bc[2] = h.nf[2], bc[3] = d0, bc[4] = tin, bc[5] = tout;
    return __builtin_fabsf(d0) < (rr + tout) * 1.000004f;
  }
  static __device__ inline v2f value(const v2f *xs, const v2f *fp) {
    v2f s = fp[3];
    if (D == 3) s = __builtin_elementwise_fma(xs[2], fp[2], s);
    s = __builtin_elementwise_fma(xs[1], fp[1], s);
    s = __builtin_elementwise_fma(xs[0], fp[0], s);
    return s;
  }
};

inline float f32_up_host(double v) {
  float f = (float)v;
  if ((double)f < v) f = nextafterf(f, INFINITY);
  return f;
}
inline float f32_down_host(double v) {
  float f = (float)v;
  if ((double)f > v) f = nextafterf(f, -INFINITY);
  return f;
}
template <int D>
inline CellConsts cell_consts(const PlaneCell<D> *, const ModelConsts &mc) {
  const double u = 5.9604644775390625e-08;
  CellConsts cc;
  memset(&cc, 0, sizeof cc);
  cc.f[0] = f32_down_host(mc.thr);
  cc.f[1] = f32_up_host(mc.thr);
  cc.f[2] = f32_up_host((1.01 * 4.3 * u * mc.thr + 3e-12 * mc.absmax) * (1.0 + 1e-6));
  cc.f[3] = f32_up_host(1.01 * 9.4 * u * (1.0 + 1e-6));
  return cc;
}

// Sphere.  Like the plane and the line, observations relative to the cell centre: x = ctr + y, x' = fl32(y), and
// with w = ctr - c (fp64, per hypothesis and cell)
//   D* = |x - c|^2 = |w|^2 + 2 w.y + |y|^2,      T* = D* - mid = k0 + sum (2 w_i + y_i) y_i,   k0 = |w|^2 - mid,
// so the fp32 arithmetic sees 2 r R-sized numbers instead of r^2-sized ones (r radius, R cell radius):
//   t32 = fma chain  s = fl32(fl32(k0) + q32);  s = fma(fl32(2 w_i), x'_i, s),   q32 = the fma chain of |x'|^2
// -- q32 is a property of the observation alone and is formed once per cell (cells_load, xs[.][3]): 4 packed
// instructions per pair of observations and hypothesis instead of 6 (r03; before: s = fma(x'_i + 2 w_i, x'_i, s)).
// Error of t32 against T* (u = 2^-24, Y_i = |y_i| <= h_i, W_i = |w_i|): q32 against |y|^2: x' rounding 2u Y_i^2 per
// term, three chain roundings 3u |y|^2: <= 5u sum Y_i^2;  2 w_i y_i: rounding of 2w_i 2u W_i Y_i, of x' 2u W_i Y_i;
// fl32(k0) u |k0|, the sum k0 + q: u (|k0| + |y|^2); three fma results: 3u S with S <= |k0| + sum Y_i^2 + 2 sum W_i Y_i;
// in total <= u (10 sum W_i Y_i + 9 sum Y_i^2 + 5 |k0|) <= u (10 |w| R + 9 R^2 + 5 |k0|).
// A cell that can hold a candidate has |k0| <= K := half + 2|w|R + R^2 + slack (else level 1 drops it), and the
// reference's fp64 D_ref, the fp64 evaluation of w and k0 are within eta = 1e-12 ((|w| + R)^2 + mid) of the
// exact values.  With E = 1.01 u (12 |w| R + 9 R^2 + 5 K) + eta (evaluated in fp32 with |w| rounded up):
//   |t32| < (half - E)(1 - 2^-21) => D_ref in [dlo, dhi] (agrees);   |t32| >= (half + E)(1 + 2^-21) => does not.
// Level 1: T* over the box lies in [k0 - 2 rr, k0 + 2 rr + R^2], rr = sum W_i h_i; the cell is dropped when that
// range (widened by E and 1e-5 relative for the fp32 evaluation of the bounds) misses [-tout, tout].
// Hypotheses whose filter is off (literal formula / magnitudes outside the validated range: f[21] = 0) keep every
// cell and send every observation the exact way (tin = -inf, tout = +inf); a NaN model never survives.
template <int D>
struct SphereCell {
  typedef SphereModel<D> M;
  enum { NB = 6, NV = 4, RELATIVE = 1, ROW = M::SPF, ROW_F32 = 1, ROW2 = 0, ROW2_OFF = 0 };
  enum { XQ = 1 };         // cells_load keeps q = |x'|^2 per observation in xs[.][3]
  enum { MIN_WAVES = 4 };  // 72 VGPRs = 7 waves per SIMD as compiled
  // r04, with the k-d refined index (tools/ab_cell_size.py, 4096 x 10 M): 512-record cells 1.30 ms full count / 0.37 ms
  // bounded against 1.36 / 0.41 ms with 256 (r03, plain Morton runs: 256 was the faster one, 1.9 against 2.1 ms)
  enum { DEFAULT_CELL = 512, LDS_BROADCAST = 1 };
  enum { USE_BOUND = 1, BOUND_MERGE = 4 };  // the sphere's box test is 61 instructions: 0.29 -> 0.08 ms, 0.87 -> 0.70 ms / step
  // r04, refined index: plain scans of >= 1024 hypotheses through the counted, statically balanced k_scan_pairs as
  // well (tools/ab_pairs.py, 4096 x 10 M: 1.01 ms against 1.27 - 1.33 ms with k_scan_cells' dynamic tiles; on plain
  // Morton runs, r03, the dynamic tiles were the faster ones).  bench.py --option scan_pairs=2 against the default on
  // the same box: one stream 2.40 -> 2.63 M hypotheses/s; four streams 3.6 - 3.7 M either way (launches of the
  // dynamic kernel share the chip more gracefully than four statically split ones)
  enum { FULL_COUNT_PAIRS = 1 };
  struct Hyp {
    double c[3], mid;
    float half;
    bool off, nan;
  };
  static __device__ inline void load(const float *row, const float *, bool valid,
                                     const CellConsts &, Hyp &h) {
#pragma unroll
    for (int i = 0; i < 3; i++) h.c[i] = i < D ? f64_from(row + 12 + 2 * i) : 0.0;
    h.mid = f64_from(row + 18);
    h.half = row[20];
    const float flag = row[21];
    h.nan = !valid || !(flag == flag);
    h.off = valid && flag == 0.0f;
  }
  static __device__ inline bool level1(const Hyp &h, const CellBox &b, const double *ctr,
                                       const CellConsts &cc, float *bc) {
    double w[3], k0 = -h.mid;
    float rr = 0.0f;
#pragma unroll
    for (int i = D - 1; i >= 0; i--) {
      w[i] = ctr[i] - h.c[i];
      k0 = fma(w[i], w[i], k0);
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float wf = i < D ? (float)w[i] : 0.0f;
      bc[i] = 2.0f * wf;
      if (i < D) rr = __builtin_fmaf(__builtin_fabsf(wf), b.h[i], rr);
    }
    const float k0f = (float)k0;
    const float R = b.pad[0];
    // |w| rounded up: |w|^2 = k0 + mid
    const float wn = __builtin_sqrtf(__builtin_fmaxf((float)(k0 + h.mid), 0.0f)) * 1.000001f;
    const float wr = wn * R, r2 = R * R;
    const float K = (h.half + 2.0f * wr + r2) * 1.0001f;
    // cc.f: 0 = 1.01 u, 1 = 1e-12 (both rounded up)
    float E = cc.f[0] * (12.0f * wr + 9.0f * r2 + 5.0f * K) * 1.00001f +
              cc.f[1] * ((wn + R) * (wn + R) + (float)h.mid) * 1.00001f;
    // K's 1e-4 head room has to cover E and the level-1 slack (see the bound on |k0| of a surviving cell)
    if (h.off || !(E <= 5e-5f * K)) E = __builtin_inff();
    const float tin = (h.half - E) * 0.9999995f;
    const float tout = (h.half + E) * 1.0000005f;
    bc[3] = k0f, bc[4] = tin, bc[5] = tout;
    // range of T* over the box, widened for the fp32 evaluation of its end points
    const float span = 2.0f * rr * 1.00001f, mag = (__builtin_fabsf(k0f) + span + r2) * 1e-5f;
    const bool miss = (k0f - span - mag > tout) | (k0f + span + r2 * 1.00001f + mag < -tout);
    return !h.nan & (h.off | !miss);
  }
  static __device__ inline v2f value(const v2f *xs, const v2f *fp) {
    v2f s = fp[3] + xs[3];  // k0 + |x'|^2
#pragma unroll
    for (int i = D - 1; i >= 0; i--) s = __builtin_elementwise_fma(fp[i], xs[i], s);
    return s;
  }
};
template <int D>
inline CellConsts cell_consts(const SphereCell<D> *, const ModelConsts &) {
  const double u = 5.9604644775390625e-08;
  CellConsts cc;
  memset(&cc, 0, sizeof cc);
  cc.f[0] = f32_up_host(1.01 * u * (1.0 + 1e-6));
  cc.f[1] = f32_up_host(1e-12 * (1.0 + 1e-6));
  return cc;
}

// Line.  Like the plane: observations relative to the cell centre, and the level-1 pass works in fp64.
//   v = ctr - a,  t = v.n,  w = v - t n   (offset of the centre from the line, w = ctr - p0, p0 on the line)
// so that for an observation x = ctr + x' of the cell  (x - p0) = x' + w  and the model's measure
// |(x - a) x n|^2 = |(x' + w) x n|^2  only involves cell-sized numbers.  The second level is
// LineModel::filter_value() on (x', +w, n); its bound (models.h: prepare_f32) holds verbatim with
// W replaced by Wc = 2R + rho >= max |x'_i| + |w|  (R = bounding radius of the cell, |w| <= R + rho for a
// surviving cell) plus eta = 1e-12 (X + A) for the fp64 evaluation of w:
//   ec = 6u Wc (1 + 4u) + u sqrt(cap) + 2 eta,   E32 = 2 sqrt3 sqrt(cap) ec + 3 ec^2 + 4u cap
//   E = 1.01 E32 + 1.01 (Eref + Enn)      (the second term per hypothesis: f[15])
// valid when 6 sqrt3 u Wc <= delta/4 and E <= delta^2/4 (otherwise tin = -inf, tout = +inf: exact path).
// Level 1: dist = |w| (fp32 norm of fl32(w): within 4u dist + eta of exact); an observation of the cell
// has |(x - a) x n| >= dist - R|n|, and D_ref >= |(x - a) x n|^2 - (Enn + Eref), so
//   dist > R + rho,  rho = delta(1 + 1e-6) + sqrt(Enn + Eref) + 1.01 * 24u(X + A)   (f[14])
// proves D_ref > delta^2 for the whole cell.  tout_abs = +inf (f[13]: the hypothesis' filter is off)
// keeps every cell and sends every observation the exact way; NaN never survives.
template <int D>
struct LineCell {
  typedef LineModel<D> M;
  enum { NB = 8, NV = 6, RELATIVE = 1, ROW = 4 * D, ROW_F32 = 0, ROW2 = 4, ROW2_OFF = 12 };
  enum { DEFAULT_CELL = 256, LDS_BROADCAST = 1, MIN_WAVES = 4 };  // measured (tools/ab_cells.py): 2.17 ms against 2.34 ms with 512 / v_readlane
  // a line that misses the inliers touches < 1 % of the cells: the consensus candidates (a quarter of the batch at
  // 50 % outliers) are 99 % of the work with or without the bound; with the statically balanced second level the
  // bounded path is the faster one all the same (2.17 against 2.34 ms per 4096 hypotheses)
  enum { USE_BOUND = 1, BOUND_MERGE = 4 };
  // (r04, refined index: a plain scan through k_scan_pairs is 1.60 - 1.67 ms against 1.75 - 1.85 ms with k_scan_cells
  // -- tools/ab_pairs.py -- but its counting pass costs the step what the scan gains: bench.py --option scan_pairs=1
  // gives the same hypotheses/s on one stream and on four; the line keeps the dynamic tiles)
  struct Hyp {
    double n[3], a[3];
    float nf[3], rho, eh;
    bool off;
  };
  static __device__ inline void load(const float *row, const float *row2, bool valid,
                                     const CellConsts &, Hyp &h) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      if (i < D) {
        h.n[i] = valid ? f64_from(row + 2 * i) : __builtin_nan("");
        h.a[i] = f64_from(row + 2 * (D + i));
      } else {
        h.n[i] = 0.0, h.a[i] = 0.0;
      }
      h.nf[i] = (float)h.n[i];
    }
    h.off = valid && row2[1] == __builtin_inff();
    h.rho = valid ? row2[2] : __builtin_nanf("");
    h.eh = row2[3];
  }
  static __device__ inline bool level1(const Hyp &h, const CellBox &b, const double *ctr,
                                       const CellConsts &cc, float *bc) {
    double v[3], t = 0.0;
#pragma unroll
    for (int i = 0; i < D; i++) {
      v[i] = ctr[i] - h.a[i];
      t = fma(v[i], h.n[i], t);
    }
    float wf[3] = {0.0f, 0.0f, 0.0f}, d2 = 0.0f;
#pragma unroll
    for (int i = 0; i < D; i++) {
      wf[i] = (float)fma(-t, h.n[i], v[i]);
      d2 = __builtin_fmaf(wf[i], wf[i], d2);
    }
    const float dist = __builtin_sqrtf(d2);
    const float R = b.pad[0];
    // cc.f: 0 Tdn, 1 Tup, 2 6u(1+4u), 3 u sqrt(cap) + 2 eta, 4 2 sqrt3 sqrt(cap), 5 4u cap,
    //       6 delta / (4 * 6 sqrt3 u), 7 delta^2 / 4
    const float Wc = (2.0f * R + h.rho) * 1.000001f;
    const float ec = __builtin_fmaf(Wc, cc.f[2], cc.f[3]);
    const float E32 = __builtin_fmaf(ec, cc.f[4], __builtin_fmaf(3.0f * ec, ec, cc.f[5]));
    float E = __builtin_fmaf(E32, 1.0100001f, h.eh);
    const bool ok = (Wc <= cc.f[6]) & (E <= cc.f[7]) & !h.off;
    E = ok ? E : __builtin_inff();
    bc[0] = wf[0], bc[1] = wf[1], bc[2] = wf[2];
    bc[3] = h.nf[0], bc[4] = h.nf[1], bc[5] = h.nf[2];
    bc[6] = (cc.f[0] - E) * 0.9999997f;
    bc[7] = (cc.f[1] + E) * 1.0000003f;
    return h.off | (dist <= R + h.rho);
  }
  static __device__ inline v2f value(const v2f *xs, const v2f *fp) {
    return M::filter_value(xs, fp);
  }
};
template <int D>
inline CellConsts cell_consts(const LineCell<D> *, const ModelConsts &mc) {
  const double u = 5.9604644775390625e-08, s3 = 1.7320508075688774;
  const double rc = 2.0 * mc.delta, cap = 4.0 * mc.delta_sq;
  const double eta = 1e-12 * 2.0 * mc.absmax;  // X + A <= 2X is enforced by prepare_f32 (A <= X: a is a datum)
  CellConsts cc;
  memset(&cc, 0, sizeof cc);
  cc.f[0] = f32_down_host(mc.delta_sq);
  cc.f[1] = f32_up_host(mc.delta_sq);
  cc.f[2] = f32_up_host(6.0 * u * (1.0 + 4.0 * u) * (1.0 + 1e-6));
  cc.f[3] = f32_up_host((u * rc + 2.0 * eta) * (1.0 + 1e-6));
  cc.f[4] = f32_up_host(2.0 * s3 * rc * (1.0 + 1e-6));
  cc.f[5] = f32_up_host(4.0 * u * cap * (1.0 + 1e-6));
  cc.f[6] = f32_down_host(0.25 * mc.delta / (6.0 * s3 * u) * (1.0 - 1e-6));
  cc.f[7] = f32_down_host(0.25 * mc.delta_sq * (1.0 - 1e-6));
  return cc;
}

// ---- level 2 of one (cell, group of 64 hypotheses): the surviving hypotheses of the group (bits of `surv`) against
// the cell's observations held in registers.  bc[] are the lane's (= hypothesis') per-cell constants from level 1;
// votes are added to lane b of accv for hypothesis b of the group.
//
// The filter works on SQUARES (r03; the |v| form it replaces cost 50 vector instructions a pair, this one 32 -- the
// loop is bound by vector issue, every wave64 instruction taking 4 cycles: profiles/r03_microbench.json):
//   level 1 gives t_in <= t_out with  |v32| < t_in => agrees,  |v32| >= t_out => does not.  With
//   a = RD(t_in^2) (0 when t_in <= 0: nothing is certain) and c = RU(t_out^2):
//     d = fma(v32, v32, -a)  has the sign of v32^2 - a exactly (one rounding), so  d < 0  =>  v32^2 < a <= t_in^2: a
//     certain inlier (float compare: NaN rows never count);  an observation outside the certain set is AMBIGUOUS iff
//     v32^2 < c, and  v32^2 < c  =>  d = RN(v32^2 - a) <= RN(c - a) =: band  (rounding is monotone) -- tested once per
//     lane on the unsigned minimum of the bit patterns of d (a negative d has its sign bit set and looks huge, a NaN
//     larger than any finite number).  Flagging a few observations with v32^2 >= c too only costs a re-check.
//   A hypothesis whose filter is off (t_out = +inf or NaN: literal-formula sphere, |n_i| > 1, out-of-range magnitudes)
//   gets a = 0 and band = 0xFFFFFFFF: every cell it survives in takes the exact predicate, whatever v32 is.
// cells_filter_squares() turns level 1's (t_in, t_out) in bc[NB-2], bc[NB-1] into (a, band bits).
template <int NB>
__device__ __forceinline__ void cells_filter_squares(float (&bc)[NB]) {
  const float tin = bc[NB - 2], tout = bc[NB - 1];
  const bool off = !(tout < __builtin_inff());
  const float a = tin > 0.0f ? (tin * tin) * 0.9999998f : 0.0f;  // RD with room: fl(t^2)(1 - 2^-22) <= t^2
  const float c = (tout * tout) * 1.0000002f;                    // RU with room (inf when it overflows: band = inf)
  const float band = c - a;
  bc[NB - 2] = off ? 0.0f : a;
  bc[NB - 1] = off ? __builtin_bit_cast(float, 0xFFFFFFFFu) : band;
}

template <class CM, int PP, bool LDSB>
__device__ __forceinline__ void cells_survivors(const v2f (&xs)[PP][4], const float (&bc)[CM::NB], float *s_bc,
                                                unsigned long long surv, const int lane,
                                                const double *__restrict__ sorted, const size_t ns, const size_t cell,
                                                const double *__restrict__ spg, const ModelConsts &mc,
                                                uint32_t &accv) {
  typedef typename CM::M M;
  constexpr int D = M::ND, NB = CM::NB, NV = CM::NV, SPD = M::SP, CP = 128 * PP;
  (void)D, (void)NV, (void)CP;
  if (LDSB && surv) {  // the lane's values -> LDS; survivors are fetched with uniform-address reads
    float4 w0, w1;
    w0.x = bc[0], w0.y = NB > 1 ? bc[1 < NB ? 1 : 0] : 0.0f, w0.z = NB > 2 ? bc[2 < NB ? 2 : 0] : 0.0f,
    w0.w = NB > 3 ? bc[3 < NB ? 3 : 0] : 0.0f;
    w1.x = NB > 4 ? bc[4 < NB ? 4 : 0] : 0.0f, w1.y = NB > 5 ? bc[5 < NB ? 5 : 0] : 0.0f,
    w1.z = NB > 6 ? bc[6 < NB ? 6 : 0] : 0.0f, w1.w = NB > 7 ? bc[7 < NB ? 7 : 0] : 0.0f;
    ((float4 *)s_bc)[2 * lane] = w0;
    ((float4 *)s_bc)[2 * lane + 1] = w1;
  }
  while (surv) {
    const int b = __builtin_ctzll(surv);
    asm("s_bitset0_b64 %0, %1" : "+s"(surv) : "s"(b));  // surv &= ~(1 << b)
    v2f fp[NV], na;
    uint32_t band;
    if (LDSB) {
      static_assert(NB <= 8, "broadcast area holds 8 floats per lane");
      const float4 r0 = ((const float4 *)s_bc)[2 * b], r1 = ((const float4 *)s_bc)[2 * b + 1];
      const float rb[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
      for (int k = 0; k < NV; k++) fp[k].x = rb[k], fp[k].y = rb[k];
      na.x = -rb[NB - 2], na.y = -rb[NB - 2];
      // (the same value in every lane: compared vector against vector -- r04; moving it to a scalar register first was
      // one of the pair's vector instructions)
      band = __builtin_bit_cast(uint32_t, rb[NB - 1]);
    } else {
#pragma unroll
      for (int k = 0; k < NV; k++) {
        float v = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bc[k]), b));
        fp[k].x = v;
        fp[k].y = v;
      }
      const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bc[NB - 2]), b));
      na.x = -a, na.y = -a;
      band = (uint32_t)__builtin_amdgcn_readlane(__builtin_bit_cast(int, bc[NB - 1]), b);
    }
    // (An early test "does any observation of the cell come near?" before the ballots -- the |v| form had one -- is
    // not made: at the bench's thresholds nearly every pair that survives level 1 holds candidates, and after d the
    // rest of the body is 13 instructions.  Two compares per value with the mask logic and the counts on the scalar
    // unit measured SLOWER in r02: the scalar unit is nearly as busy as the vector unit in this loop.)
    unsigned long long in[2 * PP];
    uint32_t dmin = 0;
    v2f sv[PP];  // (all values first: PP independent chains for the scheduler to interleave)
#pragma unroll
    for (int p = 0; p < PP; p++) sv[p] = CM::value(xs[p], fp);
#pragma unroll
    for (int p = 0; p < PP; p++) {
      const v2f s = sv[p];
      const v2f d = __builtin_elementwise_fma(s, s, na);
      in[2 * p] = __ballot(d.x < 0.0f);
      in[2 * p + 1] = __ballot(d.y < 0.0f);
      // (the two halves are compared with each other FIRST: written as two running minima -- dmin = min(dmin, dx),
      // dmin = min(dmin, dy) -- hipcc 7.2 drops the high half of a packed fma's result from the chain; seen in the
      // ISA, a standalone kernel reproduces it, 50 votes in 4 M were lost)
      typedef uint32_t v2u __attribute__((ext_vector_type(2)));
      const v2u du = __builtin_bit_cast(v2u, d);
      const uint32_t m = du.x < du.y ? du.x : du.y;
      dmin = p == 0 ? m : (m < dmin ? m : dmin);   // (the backend fuses the two minima into one v_min3_u32; the first
    }                                              // pair seeds the chain: no register to initialise)
    const unsigned long long amb = __ballot(dmin <= band);
    if (amb) {  // some observation sits in the band: exact fp64 predicate for the whole cell
      const double *hp = spg + (size_t)b * SPD;  // wave-uniform -> scalar loads
#pragma unroll
      for (int p = 0; p < PP; p++) {
        const size_t i0 = cell * CP + p * 128 + lane, i1 = i0 + 64;
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
    // (lane b of accv is written once per cell and group: every hypothesis of the group is visited once)
    // v_writelane takes one SGPR on the constant bus: the lane select goes through m0
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0"
                 : "+v"(accv)
                 : "s"(cnt), "s"(b)
                 : "m0");
  }
}

// the observations of cell `cell` as packed fp32 pairs, relative to the cell centre where the model asks for it
template <class CM, int PP>
__device__ __forceinline__ void cells_load(const double *__restrict__ sorted, const size_t ns, const size_t cell,
                                           const int lane, const double (&ctr)[3], v2f (&xs)[PP][4]) {
  constexpr int D = CM::M::ND, CP = 128 * PP;
#pragma unroll
  for (int p = 0; p < PP; p++) {
    const size_t i0 = cell * CP + p * 128 + lane, i1 = i0 + 64;
    // rows past the end are NaN (never candidates); the loads themselves are unconditional
    const double *p0 = sorted + (i0 < ns ? i0 : 0) * D, *p1 = sorted + (i1 < ns ? i1 : 0) * D;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const double off = CM::RELATIVE ? ctr[d] : 0.0;
      const float a0 = d < D ? (float)(p0[d < D ? d : 0] - off) : 0.0f;
      const float a1 = d < D ? (float)(p1[d < D ? d : 0] - off) : 0.0f;
      xs[p][d].x = (d < D && !(i0 < ns)) ? __builtin_nanf("") : a0;
      xs[p][d].y = (d < D && !(i1 < ns)) ? __builtin_nanf("") : a1;
    }
    // xs[p][3]: a per-observation constant of the model's measure, formed once per cell instead of once per
    // (hypothesis, cell) -- sphere: q = |x'|^2 (fma chain from the last coordinate down); unused otherwise
    v2f q = {0.0f, 0.0f};
    if constexpr (requires { CM::XQ; }) {
      q = xs[p][D - 1] * xs[p][D - 1];
#pragma unroll
      for (int d = D - 2; d >= 0; d--) q = __builtin_elementwise_fma(xs[p][d], xs[p][d], q);
    }
    xs[p][3] = q;
  }
}

// ---- the scan ----------------------------------------------------------------------------------------
// A cell is 128*PP consecutive records of the sorted copy (PP packed pairs per lane); a wave tile is
// CPT cells whose observations stay in registers for the whole hypothesis loop.  Tiles are handed
// out through an atomic counter (near-model cells cost several times more than far ones, a static
// assignment leaves a long tail); the next 64 hypotheses' parameters are prefetched while the
// current 64 are processed.  Votes of the 64 hypotheses of a group are collected in one VGPR
// (lane b = hypothesis h0 + b, v_readlane / v_writelane) and flushed with one LDS atomic per group.
template <class CM, int PP, int CPT, int BS, bool LDSB = false>
// CM::MIN_WAVES (default tile shape only): the register budget that buys the occupancy measured best -- plane
// 80 VGPRs = 6 waves per SIMD (98 VGPRs / 4 waves without the hint: 1.44 -> 1.31 ms; 7 waves spill: 1.40 ms)
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(CPT == 1 ? (PP >= 8 ? 4 : CM::MIN_WAVES) : 1, 8))) void k_scan_cells(const double *__restrict__ sorted, size_t ns,
                                                    const CellBox *__restrict__ boxes,
                                                    uint32_t ncells, const double *__restrict__ sp,
                                                    const float *__restrict__ rows,
                                                    const float *__restrict__ spf, uint32_t H,
                                                    ModelConsts mc, CellConsts cc,
                                                    uint32_t *__restrict__ votes,
                                                    uint32_t *__restrict__ next_tile, uint32_t grab,
                                                    uint32_t hsplit, const uint32_t *__restrict__ h_dev) {
  typedef typename CM::M M;
  constexpr int NB = CM::NB;
  // the number of hypotheses may be decided on the device (bounded scan: the batch is a compacted selection);
  // `H` then is the capacity the launch was sized for
  if (h_dev) {
    const uint32_t hd = *h_dev;
    H = hd < H ? hd : H;
  }
  constexpr int SPD = M::SP;
  constexpr int ROW = CM::ROW, NR4 = ROW / 4;  // per-hypothesis row of the level-1 pass, 16-byte loads
  constexpr int NR2 = CM::ROW2 / 4;            // optional second piece, taken from the fp32 block
  static_assert(ROW % 4 == 0, "hypothesis rows are fetched as 16-byte loads");
  extern __shared__ uint32_t s_cnt[];
  for (uint32_t h = threadIdx.x; h < H; h += BS) s_cnt[h] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // LDSB: per-wave broadcast area behind the counters (H rounded up to 4): 64 lanes x 8 floats
  float *s_bc = (float *)(s_cnt + ((H + 3) & ~3u)) + (size_t)(threadIdx.x >> 6) * 512;
  const uint32_t wtiles = (ncells + CPT - 1) / CPT;
  // Work unit = (wave tile, segment of the hypothesis range), handed out dynamically: near-model tiles
  // cost 4x more than far ones and a wave only sees a handful of them.  One same-address atomic costs
  // ~14 ns of L2 time (39 000 grabs on one counter serialise into 0.55 ms -- measured: the whole
  // launch at H = 256), so the units are split over kQueues counters; a wave starts at its
  // workgroup's queue and moves on (after a plain load says "empty") when a queue runs dry.
  const uint32_t hseg = ((H + 63) / 64 + hsplit - 1) / hsplit * 64;  // hypotheses per segment
  const uint32_t units = wtiles * hsplit;
  const uint32_t qsize = (units + kQueues - 1) / kQueues;
  uint32_t q = blockIdx.x % kQueues, tried = 0;
  for (uint32_t un = 0, un_end = 0;; un++) {
    if (un >= un_end) {
      uint32_t t = ~0u;
      while (tried < kQueues) {  // wave-uniform
        const uint32_t qbase = q * qsize;
        const uint32_t qlen = qbase >= units ? 0u : (units - qbase < qsize ? units - qbase : qsize);
        uint32_t got = ~0u;
        // the plain load may be stale (the counter only grows): at worst one useless atomic
        if (lane == 0 && *(volatile const uint32_t *)&next_tile[q * kQueuePitch] < qlen)
          got = atomicAdd(&next_tile[q * kQueuePitch], grab);
        got = __builtin_amdgcn_readfirstlane(got);
        if (got < qlen) {
          t = qbase + got;
          un_end = qbase + (got + grab < qlen ? got + grab : qlen);
          break;
        }
        q = q + 1 == kQueues ? 0 : q + 1;
        tried++;
      }
      if (t == ~0u) break;
      un = t;
    }
    const uint32_t wt = un / hsplit;
    const uint32_t hbeg = (un - wt * hsplit) * hseg, hend = hbeg + hseg < H ? hbeg + hseg : H;
    if (hbeg >= H) continue;
    v2f xs[CPT][PP][4];
    CellBox bx[CPT];
    double ctr[CPT][3];
#pragma unroll
    for (int q = 0; q < CPT; q++) {
      const uint32_t cell = wt * CPT + q;
      if (cell < ncells) {
        bx[q] = boxes[cell];  // wave-uniform address -> scalar load
      } else {
        for (int d = 0; d < 3; d++) bx[q].c[d] = 0.0f, bx[q].h[d] = __builtin_nanf("");  // never survives
      }
#pragma unroll
      for (int d = 0; d < 3; d++) ctr[q][d] = (double)bx[q].c[d];
      cells_load<CM, PP>(sorted, ns, (size_t)cell, lane, ctr[q], xs[q]);
    }
    float4 nxt[NR4];
    {
      const uint32_t hl = hbeg + lane;
      const float4 *r4 = (const float4 *)(rows + (size_t)(hl < H ? hl : 0) * ROW);
#pragma unroll
      for (int k = 0; k < NR4; k++) nxt[k] = r4[k];
    }
    float4 nxt2[NR2 ? NR2 : 1];
    if constexpr (NR2 > 0) {
      const uint32_t hl = hbeg + lane;
      const float4 *r4 = (const float4 *)(spf + (size_t)(hl < H ? hl : 0) * M::SPF + CM::ROW2_OFF);
#pragma unroll
      for (int k = 0; k < NR2; k++) nxt2[k] = r4[k];
    }
    for (uint32_t h0 = hbeg; h0 < hend; h0 += 64) {
      const uint32_t h = h0 + lane;
      float row[ROW];
#pragma unroll
      for (int k = 0; k < NR4; k++)
        row[4 * k] = nxt[k].x, row[4 * k + 1] = nxt[k].y, row[4 * k + 2] = nxt[k].z, row[4 * k + 3] = nxt[k].w;
      float row2[NR2 ? 4 * NR2 : 4];
      if constexpr (NR2 > 0) {
#pragma unroll
        for (int k = 0; k < NR2; k++)
          row2[4 * k] = nxt2[k].x, row2[4 * k + 1] = nxt2[k].y, row2[4 * k + 2] = nxt2[k].z,
                   row2[4 * k + 3] = nxt2[k].w;
      }
      if (h0 + 64 < hend) {  // prefetch the next group
        const uint32_t hn = h + 64;
        const float4 *r4 = (const float4 *)(rows + (size_t)(hn < H ? hn : 0) * ROW);
#pragma unroll
        for (int k = 0; k < NR4; k++) nxt[k] = r4[k];
        if constexpr (NR2 > 0) {
          const float4 *q4 = (const float4 *)(spf + (size_t)(hn < H ? hn : 0) * M::SPF + CM::ROW2_OFF);
#pragma unroll
          for (int k = 0; k < NR2; k++) nxt2[k] = q4[k];
        }
      }
      typename CM::Hyp hy;
      CM::load(row, row2, h < H, cc, hy);
      static_assert(CPT == 1, "cells_survivors writes lane b of accv once per cell: one cell per wave tile");
      uint32_t accv = 0;  // lane b: votes of hypothesis h0 + b collected from this tile
#pragma unroll
      for (int q = 0; q < CPT; q++) {
        float bc[NB];
        unsigned long long surv = __ballot(CM::level1(hy, bx[q], ctr[q], cc, bc));
        cells_filter_squares(bc);
        cells_survivors<CM, PP, LDSB>(xs[q], bc, s_bc, surv, lane, sorted, ns, (size_t)(wt * CPT + q),
                                      sp + (size_t)h0 * SPD, mc, accv);
      }
      if (accv) atomicAdd(&s_cnt[h], accv);  // h < H whenever accv != 0 (lanes past H never survive)
    }
  }
  __syncthreads();
  // flush: every workgroup starts at a different hypothesis so that the global atomics of the
  // workgroups finishing together do not all queue on the same addresses
  const uint32_t rot = (blockIdx.x * 61u) % (H ? H : 1u);
  for (uint32_t i = threadIdx.x; i < H; i += BS) {
    const uint32_t h = i + rot < H ? i + rot : i + rot - H;
    const uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[h], c);
  }
}

// ---- bounded scan: hypotheses that cannot win are not counted --------------------------------------------------
// RANSAC.hxx:94 abandons a hypothesis as soon as it can no longer overtake the best one; only hypotheses that become
// the best-so-far ever change the loop's state (strict '>', :100).  The batched equivalent: level 1 alone gives
// every hypothesis an UPPER bound ub[h] on its votes (k_cells_bounds: population of its surviving cells); a few
// PILOTS -- the earliest hypotheses with a large bound -- are counted exactly, and every other hypothesis h is
// counted only if  ub[h] > L[h],  L[h] = max(best of earlier batches, exact votes of the pilots before h)  -- a lower
// bound of the running maximum at h.  A skipped hypothesis has votes <= ub[h] <= L[h] <= the running maximum, so the
// serial loop would not have updated on it either: winner, iteration count and consensus set are unchanged.  Skipped
// hypotheses report 0 votes.  (50 % outliers, 4096 random planes: 12.5 % of the hypotheses are worth counting.)
constexpr int kPilots = 64;
struct BoundSel {            // device-side state of one bounded scan
  uint32_t n_pilot, n_rest;  // number of selected hypotheses of the two passes
  uint32_t n_cand;           // rank bounds (axis.h): hypotheses whose bounds were refined by rank
  uint32_t known;            // 1: a lower bound of the running maximum was known without pilots (k_pick_pilots)
};

// single block of 1024 threads, H <= 8192: pilots = the first kPilots valid hypotheses (index order) whose bound is
// at least half the largest bound.  lo (nullable): per-hypothesis LOWER vote bounds (rank bounds, axis.h) -- they
// count as known lower bounds of the running maximum like the best of earlier batches.
__global__ __launch_bounds__(1024) void k_pick_pilots(const uint32_t *__restrict__ ub,
                                                      const uint8_t *__restrict__ valid, uint32_t H,
                                                      uint32_t *__restrict__ sel, BoundSel *__restrict__ st,
                                                      uint32_t *__restrict__ votes, uint32_t best_before,
                                                      const uint32_t *__restrict__ lo, int no_pilots) {
  __shared__ uint32_t s_red[16], s_redl[16], s_scan[1024];
  const int t = threadIdx.x;
  if (H > kSelCap) return;  // 1024 threads x 8 hypotheses (models.h)
  for (uint32_t h = t; h < H; h += 1024) votes[h] = 0;  // a hypothesis that is not counted reports 0 votes
  uint32_t mx = 0, ml = 0;
  for (uint32_t h = t; h < H; h += 1024) {
    mx = valid[h] && ub[h] > mx ? ub[h] : mx;
    if (lo) ml = valid[h] && lo[h] > ml ? lo[h] : ml;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = __shfl_down(mx, o), b = __shfl_down(ml, o);
    mx = a > mx ? a : mx;
    ml = b > ml ? b : ml;
  }
  if ((t & 63) == 0) s_red[t >> 6] = mx, s_redl[t >> 6] = ml;
  __syncthreads();
  mx = 0, ml = best_before;
  for (int w = 0; w < 16; w++) mx = s_red[w] > mx ? s_red[w] : mx, ml = s_redl[w] > ml ? s_redl[w] : ml;
  const uint32_t thr = mx - mx / 2;  // ceil(mx / 2)
  // each thread owns 8 consecutive hypotheses: flags, block-wide exclusive scan, ordered write
  uint32_t f[8], cnt = 0;
  for (int k = 0; k < 8; k++) {
    const uint32_t h = t * 8 + k;
    f[k] = (h < H && valid[h] && mx > 0 && ub[h] >= thr) ? 1u : 0u;
    cnt += f[k];
  }
  s_scan[t] = cnt;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const uint32_t a = t >= o ? s_scan[t - o] : 0u;
    __syncthreads();
    s_scan[t] += a;
    __syncthreads();
  }
  uint32_t pos = s_scan[t] - cnt;
  for (int k = 0; k < 8; k++)
    if (f[k]) {
      if (pos < kPilots) sel[pos] = t * 8 + k;
      pos++;
    }
  if (t == 1023) {
    // Pilots only exist to give the second pass a lower bound of the running maximum.  When an earlier batch of the
    // same RANSAC run already holds a maximum of at least half the largest bound (the pilots' own entry level), that
    // IS the bound: no pilots, the counting launches of the first pass find an empty cost table and return at once.
    // (The rank bounds' lower bounds count the same way.)
    // no_pilots: the host did not launch a pilot pass for this batch (the last batch of the upload needed none:
    // run_scan_bounded); `known` still says whether one would have been needed, which re-arms it for the next batch.
    const bool known = ml >= thr && ml > 0;
    st->known = known ? 1u : 0u;
    st->n_pilot = (known || no_pilots) ? 0u : (s_scan[1023] < kPilots ? s_scan[1023] : kPilots);
    st->n_rest = 0;
    if (!lo) st->n_cand = 0;
  }
}

// the rest: every valid non-pilot hypothesis whose bound exceeds the lower bound of the running maximum at its index
__global__ __launch_bounds__(1024) void k_pick_rest(const uint32_t *__restrict__ ub,
                                                    const uint8_t *__restrict__ valid, uint32_t H,
                                                    const uint32_t *__restrict__ pilots,
                                                    const uint32_t *__restrict__ pilot_votes, uint32_t best_before,
                                                    uint32_t *__restrict__ sel, BoundSel *__restrict__ st,
                                                    const uint32_t *__restrict__ lo) {
  __shared__ uint32_t s_pi[kPilots], s_pm[kPilots], s_scan[1024];
  const int t = threadIdx.x;
  if (H > kSelCap) return;
  const uint32_t np = st->n_pilot;
  // lo (nullable): lower vote bounds per hypothesis; lob[k] = max of lo over the hypotheses before t * 8 + k
  uint32_t lob[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lo) {
    uint32_t lm = 0;
    for (int k = 0; k < 8; k++) {
      const uint32_t h = t * 8 + k;
      lob[k] = lm;
      const uint32_t v = (h < H && valid[h]) ? lo[h] : 0u;
      lm = v > lm ? v : lm;
    }
    s_scan[t] = lm;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const uint32_t a = t >= o ? s_scan[t - o] : 0u;
      __syncthreads();
      s_scan[t] = a > s_scan[t] ? a : s_scan[t];
      __syncthreads();
    }
    const uint32_t before = t ? s_scan[t - 1] : 0u;
    __syncthreads();
    for (int k = 0; k < 8; k++) lob[k] = before > lob[k] ? before : lob[k];
  }
  if (t < kPilots) s_pi[t] = t < (int)np ? pilots[t] : 0xFFFFFFFFu;
  if (t == 0) {  // running maximum over the pilots in index order (they are sorted by index)
    uint32_t m = best_before;
    for (uint32_t j = 0; j < np; j++) {
      m = pilot_votes[j] > m ? pilot_votes[j] : m;
      s_pm[j] = m;
    }
  }
  __syncthreads();
  uint32_t f[8], cnt = 0;
  for (int k = 0; k < 8; k++) {
    const uint32_t h = t * 8 + k;
    f[k] = 0;
    if (h < H && valid[h]) {
      uint32_t before = 0;  // pilots with a smaller index: lower bound in the sorted list (0xFFFFFFFF pads it)
#pragma unroll
      for (uint32_t step = kPilots / 2; step > 0; step >>= 1)
        before += s_pi[before + step - 1] < h ? step : 0u;
      before += (before < kPilots && s_pi[before] < h) ? 1u : 0u;
      const bool is_pilot = before < kPilots && s_pi[before] == h;
      uint32_t L = before ? s_pm[before - 1] : best_before;
      L = lob[k] > L ? lob[k] : L;
      f[k] = (!is_pilot && ub[h] > L) ? 1u : 0u;
    }
    cnt += f[k];
  }
  s_scan[t] = cnt;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const uint32_t a = t >= o ? s_scan[t - o] : 0u;
    __syncthreads();
    s_scan[t] += a;
    __syncthreads();
  }
  uint32_t pos = s_scan[t] - cnt;
  for (int k = 0; k < 8; k++)
    if (f[k]) sel[pos++] = t * 8 + k;
  if (t == 1023) st->n_rest = s_scan[1023];
}

// rows of the selected hypotheses -> a compact batch (fp64 scan parameters and the fp32 block), NaN rows up to `cap`
__global__ __launch_bounds__(256) void k_gather_rows(const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel,
                                                     uint32_t cap, const double *__restrict__ sp, int spd,
                                                     const float *__restrict__ spf, int spfd,
                                                     double *__restrict__ sp_out, float *__restrict__ spf_out) {
  const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= cap) return;
  const bool live = j < *n_sel;
  const uint32_t h = live ? sel[j] : 0;
  for (int k = lane; k < spd; k += 64) sp_out[(size_t)j * spd + k] = live ? sp[(size_t)h * spd + k] : __builtin_nan("");
  for (int k = lane; k < spfd; k += 64) spf_out[(size_t)j * spfd + k] = live ? spf[(size_t)h * spfd + k] : __builtin_nanf("");
}
// diagnostics: sum of per-hypothesis values over a selection
__global__ __launch_bounds__(256) void k_sum_selected(const uint32_t *__restrict__ sel, const uint32_t *__restrict__ n_sel,
                                                      const uint32_t *__restrict__ v, unsigned long long *__restrict__ out) {
  unsigned long long t = 0;
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < *n_sel; j += gridDim.x * 256) t += v[sel[j]];
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if ((threadIdx.x & 63) == 0 && t) atomicAdd(out, t);
}
// exact votes of the two counted selections back to their places in the batch (block 0: the pilots)
__global__ __launch_bounds__(256) void k_scatter_votes(const uint32_t *__restrict__ sel_a,
                                                       const uint32_t *__restrict__ n_a,
                                                       const uint32_t *__restrict__ v_a,
                                                       const uint32_t *__restrict__ sel_b,
                                                       const uint32_t *__restrict__ n_b,
                                                       const uint32_t *__restrict__ v_b,
                                                       uint32_t *__restrict__ votes) {
  if (blockIdx.x == 0) {
    static_assert(kPilots <= 256, "one block scatters the pilots");
    if (threadIdx.x < *n_a) votes[sel_a[threadIdx.x]] = v_a[threadIdx.x];
    return;
  }
  const uint32_t j = (blockIdx.x - 1) * 256 + threadIdx.x;
  if (j < *n_b) votes[sel_b[j]] = v_b[j];
}

// Level 1 alone (measurement, and the vote bound of the two-pass scan): per hypothesis the summed population
// of its surviving cells -- an upper bound on its votes, every agreeing observation lies in a surviving cell --
// and, per launch, the number of surviving (hypothesis, cell) pairs, i.e. what level 2 has to look at.
// Lane = hypothesis; blockIdx.y * 4 + wave = group of 64 hypotheses; blockIdx.x = a run of cells.
template <class CM, int PP>
__global__ __launch_bounds__(256) void k_cells_bounds(const CellBox *__restrict__ boxes, uint32_t ncells,
                                                      size_t ns, const float *__restrict__ rows,
                                                      const float *__restrict__ spf, uint32_t H,
                                                      CellConsts cc, uint32_t cells_per_block,
                                                      uint32_t *__restrict__ ub,
                                                      unsigned long long *__restrict__ total,
                                                      uint32_t *__restrict__ ncells_out,
                                                      uint8_t *__restrict__ cnt, uint32_t gstride,
                                                      const uint32_t *__restrict__ h_dev, uint32_t h_off,
                                                      uint32_t box_pop) {
  typedef typename CM::M M;
  constexpr int ROW = CM::ROW, NR4 = ROW / 4, NR2 = CM::ROW2 / 4;
  const size_t CP = box_pop;  // observations per box: 128 * PP for the cells, a multiple for merged boxes
  const int lane = threadIdx.x & 63;
  if (h_dev) {
    const uint32_t hd = *h_dev > h_off ? *h_dev - h_off : 0u;  // hypotheses [h_off, h_off + H) of the selection
    H = hd < H ? hd : H;
  }
  const uint32_t grp = blockIdx.y * 4 + (threadIdx.x >> 6);
  const uint32_t h = grp * 64 + lane;
  if (blockIdx.y * 256 >= H) return;   // workgroup-uniform: the barriers below are for whole workgroups
  const bool active = h - lane < H;    // wave-uniform
  float row[ROW], row2[NR2 ? 4 * NR2 : 4];
  {
    const float4 *r4 = (const float4 *)(rows + (size_t)(h < H ? h : 0) * ROW);
#pragma unroll
    for (int k = 0; k < NR4; k++) {
      const float4 v = r4[k];
      row[4 * k] = v.x, row[4 * k + 1] = v.y, row[4 * k + 2] = v.z, row[4 * k + 3] = v.w;
    }
    if constexpr (NR2 > 0) {
      const float4 *q4 = (const float4 *)(spf + (size_t)(h < H ? h : 0) * M::SPF + CM::ROW2_OFF);
#pragma unroll
      for (int k = 0; k < NR2; k++) {
        const float4 v = q4[k];
        row2[4 * k] = v.x, row2[4 * k + 1] = v.y, row2[4 * k + 2] = v.z, row2[4 * k + 3] = v.w;
      }
    }
  }
  typename CM::Hyp hy;
  CM::load(row, row2, h < H, cc, hy);
  const uint32_t c0 = blockIdx.x * cells_per_block;
  const uint32_t c1 = c0 + cells_per_block < ncells ? c0 + cells_per_block : ncells;
  uint32_t u = 0, nc = 0;
  // The boxes go through LDS, 128 at a time: a scalar load per cell leaves every iteration waiting for its own
  // s_load (SMEM returns out of order, so the wait is always lgkmcnt(0) and nothing can be fetched ahead) --
  // measured 141 us for 1.25 M plane evaluations of ~20 instructions; LDS reads of one address are broadcasts that
  // return in order and pipeline across the unrolled loop.
  constexpr uint32_t BCH = 128;
  __shared__ CellBox s_box[BCH];
  __shared__ double s_ctr[BCH][3];
  for (uint32_t cb = c0; cb < c1; cb += BCH) {
    const uint32_t n = c1 - cb < BCH ? c1 - cb : BCH;
    __syncthreads();  // the previous chunk has been consumed
    if (threadIdx.x < 2 * n) ((float4 *)s_box)[threadIdx.x] = ((const float4 *)(boxes + cb))[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < n) {
#pragma unroll
      for (int d = 0; d < 3; d++) s_ctr[threadIdx.x][d] = (double)s_box[threadIdx.x].c[d];
    }
    __syncthreads();
    if (!active) continue;
    auto one = [&](const uint32_t i, const CellBox &bx, const double (&ctr)[3]) {
      const uint32_t c = cb + i;
      float bc[CM::NB];
      const bool s = CM::level1(hy, bx, ctr, cc, bc);
      const size_t first = (size_t)c * CP;
      const uint32_t pop = first + CP <= ns ? (uint32_t)CP : (uint32_t)(ns - first);
      u += s ? pop : 0u;
      nc += s ? 1u : 0u;
      if (cnt) {  // survivors of this group in this cell: the cost table of k_scan_pairs
        const uint32_t pc = (uint32_t)__builtin_popcountll(__ballot(s));
        if (lane == 0) cnt[(size_t)c * gstride + grp] = (uint8_t)(pc ? pc + kGroupPad : 0u);  // <= 64 + pad
      }
    };
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) {  // four boxes in flight
      const CellBox b0 = s_box[i], b1 = s_box[i + 1], b2 = s_box[i + 2], b3 = s_box[i + 3];
      const double t0[3] = {s_ctr[i][0], s_ctr[i][1], s_ctr[i][2]};
      const double t1[3] = {s_ctr[i + 1][0], s_ctr[i + 1][1], s_ctr[i + 1][2]};
      const double t2[3] = {s_ctr[i + 2][0], s_ctr[i + 2][1], s_ctr[i + 2][2]};
      const double t3[3] = {s_ctr[i + 3][0], s_ctr[i + 3][1], s_ctr[i + 3][2]};
      one(i, b0, t0);
      one(i + 1, b1, t1);
      one(i + 2, b2, t2);
      one(i + 3, b3, t3);
    }
    for (; i < n; i++) {
      const CellBox b0 = s_box[i];
      const double t0[3] = {s_ctr[i][0], s_ctr[i][1], s_ctr[i][2]};
      one(i, b0, t0);
    }
  }
  if (ub && h < H && u) atomicAdd(&ub[h], u);
  if (ncells_out && h < H && nc) atomicAdd(&ncells_out[h], nc);  // surviving cells of the hypothesis (diagnostics)
  if (total) {
    for (int o = 32; o > 0; o >>= 1) nc += __shfl_down(nc, o);
    if (lane == 0 && nc) atomicAdd(total, (unsigned long long)nc);
  }
}

// ---- statically balanced level 2 ------------------------------------------------------------------------------
// Handing (tile, hypothesis range) units to waves dynamically does not work for the light launches of the bounded
// scan: a returning atomic on a counter shared across the eight XCDs costs ~120 ns of serialised memory-side time
// (19 532 grabs on 32 counters add 75 us to a 140 us launch; finer units cost proportionally more), while whole tiles
// are too coarse -- with ~500 near-model hypotheses a near-model tile is 280 us of one wave's time in a 660 us launch
// and half of the waves' lifetime is spent waiting for the slowest (measured with s_memrealtime).  So the work is
// COUNTED first and then cut into equal pieces:
//   k_cells_bounds(cnt)  level 1 alone: cnt[cell][group] = surviving hypotheses of the group in the cell
//   k_tile_costs         cost[cell] = sum over groups, csum[chunk] = sum over the 128 cells of a chunk
//   k_scan_pairs         wave w of W takes the (hypothesis, cell) pairs [w C / W, (w + 1) C / W) of the
//                        cell-major, group-major, lane-major enumeration: it finds its first cell from csum / cost
//                        (two wave-wide prefix sums), walks cells and groups from there, repeats level 1 for the
//                        groups it touches (bit-identical to the counting pass) and runs cells_survivors on its part
//                        of the survivor mask.  No atomics except the vote flush, no waiting, every wave the same
//                        number of pairs (a pair costs 31 vector instructions, an exact re-check a few hundred:
//                        the 800+ pairs of a wave average out).
constexpr uint32_t kChunkCells = 128;
// A pair is the unit of cost; what a wave pays per group it evaluates (rows, level 1: ~2 pairs' worth of instructions
// and a dependent load) and per cell it opens (box, 12 KB of observations) is charged as padding in front of the
// group's / cell's pairs, so that a stretch of far cells with one or two survivors per group is not handed to one
// wave as if it were free (measured before the padding: one wave walking 574 groups for 662 pairs, 1.5 ms against a
// mean of 0.44 ms).

__global__ __launch_bounds__(128) void k_tile_costs(const uint8_t *__restrict__ cnt, uint32_t gstride, uint32_t H,
                                                    const uint32_t *__restrict__ h_dev, uint32_t ncells,
                                                    uint32_t *__restrict__ cost, uint32_t *__restrict__ csum,
                                                    uint32_t *__restrict__ votes, uint32_t h_off) {
  for (uint32_t i = blockIdx.x * kChunkCells + threadIdx.x; i < H; i += gridDim.x * kChunkCells) votes[i] = 0;
  if (h_dev) {
    const uint32_t hd = *h_dev > h_off ? *h_dev - h_off : 0u;  // hypotheses [h_off, h_off + H) of the selection
    H = hd < H ? hd : H;
  }
  const uint32_t G = (H + 63) / 64;
  const uint32_t c = blockIdx.x * kChunkCells + threadIdx.x;
  uint32_t t = 0;
  if (c < ncells) {
    const uint8_t *p = cnt + (size_t)c * gstride;
    // (only the groups below G have been written by the counting pass)
    uint32_t g = 0;
    if ((gstride & 15u) == 0)
      for (; g + 16 <= G; g += 16) {
        const uint4 v = *(const uint4 *)(p + g);
        t += __builtin_amdgcn_sad_u8(v.x, 0u, 0u) + __builtin_amdgcn_sad_u8(v.y, 0u, 0u) +
             __builtin_amdgcn_sad_u8(v.z, 0u, 0u) + __builtin_amdgcn_sad_u8(v.w, 0u, 0u);
      }
    for (; g < G; g++) t += p[g];
    t = t ? t + kCellPad : 0u;
    cost[c] = t;
  }
  __shared__ uint32_t s_w[2];
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) csum[blockIdx.x] = s_w[0] + s_w[1];
}

// inclusive prefix sum across the wave (lane i: sum of lanes 0..i)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t a = __shfl_up(v, o);
    v += lane >= o ? a : 0u;
  }
  return v;
}

template <class CM, int PP, int BS, bool LDSB>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(PP >= 8 ? 4 : CM::MIN_WAVES, 8))) void k_scan_pairs(
    const double *__restrict__ sorted, size_t ns, const CellBox *__restrict__ boxes, uint32_t ncells,
    const double *__restrict__ sp, const float *__restrict__ rows, const float *__restrict__ spf, uint32_t H,
    ModelConsts mc, CellConsts cc, uint32_t *__restrict__ vpart, uint32_t vstride,
    const uint32_t *__restrict__ h_dev, const uint8_t *__restrict__ cnt, uint32_t gstride, const uint32_t *__restrict__ cost,
    const uint32_t *__restrict__ csum, uint32_t nchunks, uint32_t h_off) {
  typedef typename CM::M M;
  constexpr int NB = CM::NB;
  constexpr int SPD = M::SP;
  constexpr int ROW = CM::ROW, NR4 = ROW / 4, NR2 = CM::ROW2 / 4;
  if (h_dev) {
    const uint32_t hd = *h_dev > h_off ? *h_dev - h_off : 0u;  // hypotheses [h_off, h_off + H) of the selection
    H = hd < H ? hd : H;
  }
  extern __shared__ uint32_t s_cnt[];
  for (uint32_t h = threadIdx.x; h < H; h += BS) s_cnt[h] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float *s_bc = (float *)(s_cnt + ((H + 3) & ~3u)) + (size_t)(threadIdx.x >> 6) * 512;
  const uint32_t W = gridDim.x * (BS / 64);
  const uint32_t wid = blockIdx.x * (BS / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

  // ---- my share of the pairs: [t0, t0 + budget) of C
  unsigned long long C = 0;
  for (uint32_t k0 = 0; k0 < nchunks; k0 += 64) {
    uint32_t v = k0 + lane < nchunks ? csum[k0 + lane] : 0u;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    C += (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
  }
  const unsigned long long t0 = C * wid / W, t1 = C * (wid + 1) / W;
  uint32_t budget = (uint32_t)(t1 - t0);
  uint32_t cell = ncells, skip = 0;
  if (budget) {  // wave-uniform
    // the chunk that holds pair t0 ...
    unsigned long long base = 0;
    uint32_t chunk = 0;
    for (uint32_t k0 = 0; k0 < nchunks; k0 += 64) {
      const uint32_t v = k0 + lane < nchunks ? csum[k0 + lane] : 0u;
      const uint32_t inc = wave_incl_scan(v, lane);
      const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      if (base + tot > t0) {
        const unsigned long long hit = __ballot(base + inc > t0);
        const int l = __builtin_ctzll(hit);
        chunk = k0 + l;
        base += (uint32_t)__builtin_amdgcn_readlane((int)(inc - v), l);
        break;
      }
      base += tot;
    }
    // ... and the cell inside it (kChunkCells = 2 per lane)
    const uint32_t cb = chunk * kChunkCells + 2 * lane;
    const uint32_t v0 = cb < ncells ? cost[cb] : 0u, v1 = cb + 1 < ncells ? cost[cb + 1] : 0u;
    const uint32_t inc = wave_incl_scan(v0 + v1, lane);
    const uint32_t r = (uint32_t)(t0 - base);  // pairs of the chunk before mine
    const unsigned long long hit = __ballot(inc > r);
    const int l = hit ? __builtin_ctzll(hit) : 63;
    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(inc - v0 - v1), l);
    const uint32_t c0v = (uint32_t)__builtin_amdgcn_readlane((int)v0, l);
    const bool second = r - before >= c0v;
    cell = chunk * kChunkCells + 2 * l + (second ? 1u : 0u);
    skip = r - before - (second ? c0v : 0u);
  }

  const uint32_t G = (H + 63) / 64;
  // cost units [skip, skip + budget) of the enumeration that starts at `cell`: a padded item of P units in front of
  // the position consumes what of it lies in my range
  auto pad = [&](uint32_t P) {
    if (skip >= P) {
      skip -= P;
    } else {
      const uint32_t take = P - skip < budget ? P - skip : budget;
      budget -= take;
      skip = 0;
    }
  };
  auto load_rows = [&](uint32_t g, float4(&r)[NR4], float4(&r2)[NR2 ? NR2 : 1]) {
    const uint32_t h = g * 64 + lane;
    const float4 *r4 = (const float4 *)(rows + (size_t)(h < H ? h : 0) * ROW);
#pragma unroll
    for (int k = 0; k < NR4; k++) r[k] = r4[k];
    if constexpr (NR2 > 0) {
      const float4 *q4 = (const float4 *)(spf + (size_t)(h < H ? h : 0) * M::SPF + CM::ROW2_OFF);
#pragma unroll
      for (int k = 0; k < NR2; k++) r2[k] = q4[k];
    }
  };
  while (budget && cell < ncells) {
    // jump over cells nothing survives in
    {
      const uint32_t v = cell + lane < ncells ? cost[cell + lane] : 0u;
      const unsigned long long nz = __ballot(v != 0);
      if (!nz) {
        cell += 64;
        continue;
      }
      cell += (uint32_t)__builtin_ctzll(nz);
    }
    pad(kCellPad);
    // padded survivor counts of the cell's groups: lane g <-> group g (the host cuts batches of more than 4096
    // hypotheses into launches of 4096: a third loop level here costs the 4096 case 28 spilled registers)
    const uint32_t gc = (uint32_t)lane < G ? (uint32_t)cnt[(size_t)cell * gstride + lane] : 0u;
    unsigned long long gm = __ballot(gc != 0);
    // groups that lie before my range altogether
    while (gm && budget) {
      const int g = __builtin_ctzll(gm);
      const uint32_t cg = (uint32_t)__builtin_amdgcn_readlane((int)gc, g);
      if (skip < cg) break;
      skip -= cg;
      gm &= gm - 1;
    }
    if (!gm || !budget) {
      cell++;
      continue;
    }
    const CellBox bx = boxes[cell];  // wave-uniform address -> scalar load
    double ctr[3];
#pragma unroll
    for (int d = 0; d < 3; d++) ctr[d] = (double)bx.c[d];
    v2f xs[PP][4];
    cells_load<CM, PP>(sorted, ns, (size_t)cell, lane, ctr, xs);
    float4 nxt[NR4], nxt2[NR2 ? NR2 : 1];
    load_rows((uint32_t)__builtin_ctzll(gm), nxt, nxt2);
    while (gm && budget) {
      const int g = __builtin_ctzll(gm);
      gm &= gm - 1;
      const uint32_t cg = (uint32_t)__builtin_amdgcn_readlane((int)gc, g);
      float row[ROW], row2[NR2 ? 4 * NR2 : 4];
#pragma unroll
      for (int k = 0; k < NR4; k++)
        row[4 * k] = nxt[k].x, row[4 * k + 1] = nxt[k].y, row[4 * k + 2] = nxt[k].z, row[4 * k + 3] = nxt[k].w;
      if constexpr (NR2 > 0) {
#pragma unroll
        for (int k = 0; k < NR2; k++)
          row2[4 * k] = nxt2[k].x, row2[4 * k + 1] = nxt2[k].y, row2[4 * k + 2] = nxt2[k].z,
                   row2[4 * k + 3] = nxt2[k].w;
      }
      if (gm) load_rows((uint32_t)__builtin_ctzll(gm), nxt, nxt2);  // the next group's rows meanwhile
      // my part [lo, hi) of the group's cost units; pair j sits at unit kGroupPad + j
      const uint32_t lo = skip, hi = cg < skip + budget ? cg : skip + budget;  // skip < cg here
      budget -= hi - lo;
      skip = 0;
      const uint32_t jlo = (lo > kGroupPad ? lo : kGroupPad) - kGroupPad,
                     jhi = (hi > kGroupPad ? hi : kGroupPad) - kGroupPad;
      if (jhi <= jlo) continue;
      const uint32_t h0 = (uint32_t)g * 64, h = h0 + lane;
      typename CM::Hyp hy;
      CM::load(row, row2, h < H, cc, hy);
      float bc[NB];
      const bool l1 = CM::level1(hy, bx, ctr, cc, bc);
      unsigned long long surv = __ballot(l1);  // == the counting pass: cg - pad bits
      cells_filter_squares(bc);
      for (uint32_t k = 0; k < jlo; k++) surv &= surv - 1;                  // the first jlo are not mine
      if (jhi < cg - kGroupPad) {  // the tail belongs to the next wave: keep the lowest jhi - jlo bits
        unsigned long long keep = 0, m = surv;
        for (uint32_t k = jlo; k < jhi; k++) {
          keep |= m & (0ull - m);
          m &= m - 1;
        }
        surv = keep;
      }
      uint32_t accv = 0;
      cells_survivors<CM, PP, LDSB>(xs, bc, s_bc, surv, lane, sorted, ns, (size_t)cell, sp + (size_t)h0 * SPD, mc,
                                    accv);
      if (accv) atomicAdd(&s_cnt[h], accv);
    }
    cell++;
  }
  __syncthreads();
  // Every workgroup gets here at the same moment (that is the point of the static split), so H atomics per
  // workgroup would arrive as one burst on the few cache lines of votes[] (measured: 770 k atomics on 16 lines take
  // ~0.9 ms to drain, three times the counting itself).  The partial counts go out as plain coalesced stores and
  // k_votes_reduce adds them up.
  for (uint32_t i = threadIdx.x; i < H; i += BS) vpart[(size_t)blockIdx.x * vstride + i] = s_cnt[i];
}

// ---- hypothesis order of a full count (plane) -------------------------------------------------------------------
// k_scan_pairs evaluates level 1 for every (cell, 64-hypothesis group) in which the counting pass found a survivor.
// With the hypotheses in sampling order a group is 64 unrelated planes and some of them cut almost every cell: 95 % of
// the (cell, group) pairs have to be evaluated.  Sorted by a Morton key of (normal direction, offset) a group is 64
// SIMILAR planes that miss the same cells: 52 % (10 M points, 4096 hypotheses) -- the scan walks the batch through a
// permutation (rows gathered, votes scattered back: the count of a hypothesis does not depend on its position).
// One workgroup of 1024 threads, H <= 4096: keys, bitonic sort of (key, index) in LDS; perm[j] = index of the j-th
// hypothesis in key order, *count = H (the device-side count k_gather_rows takes).
template <int SPD>
__global__ __launch_bounds__(1024) void k_plane_order(const double *__restrict__ sp, uint32_t H, double xabs,
                                                      uint32_t *__restrict__ perm, uint32_t *__restrict__ count) {
  __shared__ unsigned long long s_k[kOrderCap];
  static_assert(kOrderCap == 4096, "the key loop and the bitonic network below are written for 4096 slots");
  const int t = threadIdx.x;
  if (H > kOrderCap) return;
  const double inv = 1.0 / (1.7320508075688774 * (xabs > 0.0 ? xabs : 1.0));
  for (uint32_t h = t; h < 4096; h += 1024) {
    unsigned long long key = ~0ull;  // past the batch: sorts to the end
    if (h < H) {
      const double *r = sp + (size_t)h * SPD;
      const double n0 = r[0], n1 = r[1], n2 = r[2];
      double c = n0 * r[3] + n1 * r[4] + n2 * r[5];
      uint32_t m = 0x8000u;  // NaN models after all others
      if (n0 == n0 && c == c) {
        const double sg = (n2 != 0.0 ? n2 : n1 != 0.0 ? n1 : n0) < 0.0 ? -1.0 : 1.0;  // one of the two signs of a plane
        auto q5 = [](double v) {  // [-1, 1] -> 0..31
          const double w = (v + 1.0) * 16.0;
          return (uint32_t)(w < 0.0 ? 0.0 : w > 31.0 ? 31.0 : w);
        };
        const uint32_t qa = q5(sg * n0), qb = q5(sg * n1), qd = q5(sg * c * inv);
        m = 0;
#pragma unroll
        for (int b = 0; b < 5; b++)
          m |= ((qa >> b) & 1u) << (3 * b) | ((qb >> b) & 1u) << (3 * b + 1) | ((qd >> b) & 1u) << (3 * b + 2);
      }
      key = ((unsigned long long)m << 32) | h;
    }
    s_k[h] = key;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= 4096; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = t; i < 4096; i += 1024) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const unsigned long long a = s_k[i], b = s_k[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) s_k[i] = b, s_k[l] = a;
        }
      }
      __syncthreads();
    }
  for (uint32_t i = t; i < H; i += 1024) perm[i] = (uint32_t)(s_k[i] & 0xFFFFFFFFull);
  if (t == 0) *count = H;
}
// votes[perm[j]] = v[j]
__global__ __launch_bounds__(256) void k_scatter_perm(const uint32_t *__restrict__ perm, uint32_t H,
                                                      const uint32_t *__restrict__ v, uint32_t *__restrict__ votes) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j < H) votes[perm[j]] = v[j];
}

// votes[h] += sum over the workgroups of k_scan_pairs of their partial counts; blockIdx.x = 64 hypotheses,
// blockIdx.y = a slice of the workgroups
__global__ __launch_bounds__(256) void k_votes_reduce(const uint32_t *__restrict__ vpart, uint32_t vstride,
                                                      uint32_t nparts, uint32_t H,
                                                      const uint32_t *__restrict__ h_dev,
                                                      uint32_t *__restrict__ votes, uint32_t h_off) {
  if (h_dev) {
    const uint32_t hd = *h_dev > h_off ? *h_dev - h_off : 0u;  // hypotheses [h_off, h_off + H) of the selection
    H = hd < H ? hd : H;
  }
  const uint32_t h = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
  if (blockIdx.x * 64 >= H) return;
  uint32_t t = 0;
  if (h < H)
    for (uint32_t r = blockIdx.y * 4 + ty; r < nparts; r += 4 * gridDim.y) t += vpart[(size_t)r * vstride + h];
  __shared__ uint32_t s_t[256];
  s_t[threadIdx.x] = t;
  __syncthreads();
  if (ty == 0 && h < H) {
    t += s_t[64 + threadIdx.x] + s_t[128 + threadIdx.x] + s_t[192 + threadIdx.x];
    if (t) atomicAdd(&votes[h], t);
  }
}

}  // namespace lsqr
