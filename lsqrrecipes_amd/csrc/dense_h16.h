// dense_h16.h -- the dense system's agree() scan (DenseLinearEquationSystemParametersEstimator.hxx:111-119) as a
// FILTER on the fp16 matrix cores: every coefficient and every unknown is split into two fp16 numbers and the
// residual of a (row, hypothesis) pair is accumulated from three of the four partial products in fp32 --
// `v_mfma_f32_32x32x16_f16` runs at 16 times the rate of the fp32 matrix instruction dense.h's filter uses, three
// of them replace one.  What the filter cannot decide (|residual| within E of delta) goes to the same worklist and
// the same exact fp64 re-check as before (dense.h: k_dense_recheck_seg): votes stay bit-identical.
//
// Scaling (fp16 holds 11 bits between 2^-14 and 65504).  Per upload  pa = 2^15 / Amax  (Amax = largest |coefficient|),
// per hypothesis  ph = 2^15 / max_k |x_k|.  With a'' = a pa, x'' = x ph (formed in fp64: error 2^-53, ignored):
//   a1 = fp16(a''), a2 = fp16(a'' - a1)   (the difference is exact in fp64),     likewise x1, x2.
//   |a'' - a1| <= 2^-11 |a''| (half an ulp of an 11-bit significand), so |a2| <= 2^-11 |a''| and
//   |a'' - a1 - a2| <= 2^-23 |a''|  when a2 is a normal fp16 number; values below 2^-14 may be flushed
//   by the matrix unit: absolute error <= 2^-14 per entry, i.e. <= 2^-29 of the scale 2^15.
// The filter evaluates  r'' = sum_k (a1 x2 + a2 x1) + sum_k a1 x1 - b''_i ph,   b''_i = fl32(b_i pa),
// against the exact  res'' = (sum_k a_k x_k - b_i) pa ph.  With S = 2^15 sum_k |x''_k| (>= sum |a''_k x''_k|),
// B = max |b''_i| ph, u = 2^-24:
//   representation      (2 u + 2 u) (1 + 2^-11) S  +  2^-14 (sum|x''| + sum|a''|)  <=  4.01 u S + 4.1 u S   (both factors of a
//                       product carry 2^-23 = 2 u; sum|a''| <= 2^21, sum|x''| >= 2^14, so 2^-14 2^21 <= 2^-22 S)
//   dropped a2 x2       2^-22 S                                                              <=  4 u S
//   accumulation        the matrix unit's fp32 accumulation is not specified bit by bit.  MEASURED (tools/h16_bench.hip,
//                       k_dense_h16_probe below): the products of one instruction and its addend are aligned to the largest of
//                       them with two guard bits and truncated -- fifteen terms just below an ulp of a 2^20 term lose
//                       3.8 ulp of it, a quarter ulp = 0.5 u of the largest magnitude each.  ASSUMED: twice that, 1 u
//                       of the largest magnitude involved per product and addend (r05: 65 536 sums of RANDOM
//                       operands per context, k_dense_h16_probe_random: worst 4 u of the largest magnitude, 6 u over
//                       4 M sums -- the assumption allows 17).  The eight instructions of the small
//                       terms come first (partial sums <= 2^-10 S):  136 2^-10 u S <= 0.14 u S;  the four of a1 x1:
//                       68 (1 + 2^-9) u S <= 68.2 u S.   (tests/test_gpu_dense_h16.py measures the whole chain on every
//                       run: the largest deviation seen is below 1.5 u S.)
//   b'' and the final fma   u B (fl32 of b pa) + u (S + B) (rounding of the result)
//   in total  |r'' - res''| <= (82.5 S + 2 B) u;   k_dense_prep_h16 takes E'' = 1.01 (83 S + 2 B) u and, like
//   dense.h's fp32 filter, the reference's fp64 running sum is within 1e-14 of that scale of res''.
// Thresholds on SQUARES as in cells.h (cells_filter_squares): a = RD(t_in^2), band = RN(RU(t_out^2) - a),
//   d = fma(r'', r'', -a):  d < 0 => certain inlier;  0 <= d <= band => ambiguous;  else certain outlier.
// A hypothesis whose numbers do not fit (infinite, thresholds beyond 2^63) gets x = 0, a = 0, band = the largest
// finite float: every real row is ambiguous and decided exactly; one with a NaN among its unknowns -a = +inf, band = 0:
// never counted, as the reference's comparison with a NaN.  Rows past the end / outside the launch's range carry
// b'' = +inf: r'' = -inf, d = +inf, never counted, never ambiguous.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace lsqr {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kH16Slots = 6;          // LDS ring of hypothesis tiles (32 hypotheses = 8 KiB each)
constexpr int kH16TileBytes = 8192;   // 32 rows (or hypotheses) x 64 unknowns x 2 parts x 2 B
constexpr int kH16RowsPerWg = 256;    // four waves x two 32-row tiles

// two-fp16 split of a scaled fp64 value
__device__ __forceinline__ void h16_split(double v, _Float16 &hi, _Float16 &lo) {
  hi = (_Float16)(float)v;
  lo = (_Float16)(float)(v - (double)(float)hi);
}

// Once per upload: the rows in fragment order.  afrag[((tile * 4 + kb) * 2 + part) * 64 + lane] = the eight fp16
// values part (0: a1, 1: a2) of row tile * 32 + lane % 32, unknowns kb * 16 + 8 (lane / 32) + 0..7 -- exactly what lane
// `lane` hands to the matrix instruction; bs[row] = fl32(b pa) (+inf past the end).  One wave per 32-row tile.
__global__ __launch_bounds__(256) void k_dense_rows_h16(const double *__restrict__ data, size_t stride, size_t n_rows,
                                                        int n, double pa, uint4 *__restrict__ afrag,
                                                        float *__restrict__ bs, size_t n_tiles) {
  const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const size_t row = tile * 32 + r;
  const bool live = row < n_rows;
  const double *p = data + (live ? row : 0) * stride;
#pragma unroll
  for (int kb = 0; kb < 4; kb++) {
    h16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int k = kb * 16 + 8 * half + i;
      const double v = (live && k < n) ? p[k] * pa : 0.0;
      _Float16 a, b;
      h16_split(v, a, b);
      hi[i] = a, lo[i] = b;
    }
    afrag[((tile * 4 + kb) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    afrag[((tile * 4 + kb) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
  }
  if (half == 0) bs[row] = live ? (float)(p[n] * pa) : __builtin_inff();
}

// Once per batch: per hypothesis 256 B  [x1: 64 fp16 | x2: 64 fp16]  and  thr4 = (-a, band bits, -ph, 0).
// One wave per hypothesis, lane = unknown (r05; until then one THREAD per hypothesis: 16 waves on the whole device walking
// 64 unknowns each, 43-48 us per 1024 hypotheses -- a twentieth of a step).
__global__ __launch_bounds__(256) void k_dense_prep_h16(const double *__restrict__ sp, uint32_t H, int n, int nr,
                                                        double delta, double amax, double bmax, double pa,
                                                        _Float16 *__restrict__ xh, float *__restrict__ thr4) {
  const uint32_t h = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (h >= H) return;  // whole waves
  const int k = threadIdx.x & 63;
  const double xk = k < n ? sp[(size_t)h * nr + k] : 0.0;
  double l1 = fabs(xk), xm = fabs(xk);  // (a NaN unknown: l1 is NaN on every lane after the sum; xm is not used then)
  for (int o = 32; o > 0; o >>= 1) {
    l1 += __shfl_xor(l1, o);
    const double t = __shfl_xor(xm, o);
    xm = t > xm ? t : xm;
  }
  const double u = 5.9604644775390625e-08;
  // ph is an fp32 number (it is a factor of the fma that subtracts b'' ph): everything below uses the rounded value
  const float phf = xm > 0.0 ? (float)(32768.0 / xm) : 1.0f;
  const double ph = (double)phf;
  const double scale = pa * ph;
  // the bound of the header in the scaled domain; S = amax pa * l1 ph (amax pa = 2^15 up to rounding)
  const double S = amax * pa * l1 * ph, B = bmax * scale;
  const double E = 1.01 * (83.0 * S + 2.0 * B) * u + 1e-14 * (S + B);
  const double tin = delta * scale - E, tout = delta * scale + E;
  const bool live = l1 == l1 && l1 < 1e15 && amax < 1e15 && bmax < 1e15 && xm > 0.0 && ph < 1e30 && ph > 1e-30 &&
                    scale < 1e30 && scale > 1e-30 && tout < 9.0e18 && tout == tout;
  // not live: x = 0 and ph = 1, i.e. r'' = -b'' (finite: the host only takes this path when bmax pa < 1e18), a = 0,
  // band = the largest finite float: every real row is ambiguous
  float a = 0.0f, band = __builtin_bit_cast(float, 0x7F7FFFFFu), nph = -1.0f;
  if (live) {
    if (tin > 0.0) {
      a = (float)(tin * tin);
      if ((double)a > tin * tin) a = nextafterf(a, 0.0f);
      a *= 0.9999998f;  // RD with room for the rounding of tin * tin itself
    }
    float c = (float)(tout * tout);
    if ((double)c < tout * tout) c = nextafterf(c, INFINITY);
    c *= 1.0000002f;
    band = c - a;
    nph = -phf;
  }
  // an unknown that is NaN (a minimal system that was refused): the reference's |residual| < delta is false for every
  // row, so the hypothesis is never counted and nothing of it is ambiguous (d = r''^2 + inf) -- not a row per
  // observation in the worklist
  if (!(l1 == l1)) a = -INFINITY, band = 0.0f;
  if (k == 0) {
    thr4[4 * (size_t)h] = -a;
    thr4[4 * (size_t)h + 1] = band;
    thr4[4 * (size_t)h + 2] = nph;
    thr4[4 * (size_t)h + 3] = 0.0f;
  }
  const double v = live ? xk * (double)(-nph) : 0.0;
  _Float16 x1, x2;
  h16_split(v, x1, x2);
  _Float16 *o = xh + (size_t)h * 128;
  o[k] = x1;
  o[64 + k] = x2;
}

// LDS atomics as inline assembly: the compiler orders every LDS write or atomic that may alias the destination of a
// pending global_load_lds behind an s_waitcnt vmcnt(0) (it does not tell the ring from the counters, static array
// or not) -- one per tile for the vote counters, i.e. a wait for the tiles requested a moment ago.  LDS operations
// complete in order, so an operation the wait-count pass does not know about only makes its lgkmcnt waits stricter.
__device__ __forceinline__ void lds_add_u32(uint32_t *p, uint32_t v) {
  asm volatile("ds_add_u32 %0, %1" ::"v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)p), "v"(v)
               : "memory");
}
__device__ __forceinline__ uint32_t lds_add_rtn_u32(uint32_t *p, uint32_t v) {
  uint32_t r;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
               : "=v"(r)
               : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)p), "v"(v)
               : "memory");
  return r;
}

// How does THIS device's matrix unit align the 16 products of one instruction (and its addend) before it sums them?
// The thresholds above assume at most 1 u of the largest magnitude per product; the probe hands the unit one large term
// (2^20) and up to 15 terms just below fractions of its ulp -- a unit that truncated every aligned addend to the large
// term's ulp would lose ~0.9 ulp per small term (1.8 u), gfx950 loses a quarter ulp (0.5 u: two guard bits).
// out[v] = the result of variant v; dense_h16_probe_worst() compares with the exact sums.  Run once per context before the
// filter is used: a unit that does not keep the assumption leaves the scan to the fp32 filter.
constexpr int kH16ProbeVariants = 8;
__device__ __host__ inline float dense_h16_probe_small(int v) {
  const float smalls[kH16ProbeVariants] = {0.98975f, 0.98975f, -0.98975f, 0.0615f, 0.1245f, 0.49f, 0.98975f, 0.98975f};
  return smalls[v];
}
__global__ __launch_bounds__(64) void k_dense_h16_probe(float *__restrict__ out) {
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  for (int v = 0; v < kH16ProbeVariants; v++) {
    h16x8 a, b;
    for (int i = 0; i < 8; i++) a[i] = (_Float16)0.0f, b[i] = (_Float16)0.0f;
    const int bigk[kH16ProbeVariants] = {0, 0, 0, 0, 0, 0, 15, 7};  // slot k = 8 half + i of row 0 / column 0
    if (col == 0) {
      for (int i = 0; i < 8; i++) {
        const int k = 8 * half + i;
        if (k == bigk[v]) {
          a[i] = (_Float16)1024.0f, b[i] = (_Float16)1024.0f;
        } else if (v != 1 || k < 13) {
          a[i] = (_Float16)dense_h16_probe_small(v), b[i] = (_Float16)1.0f;
        }
      }
    }
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    if (v == 1 && lane == 0) acc[0] = dense_h16_probe_small(v);  // the addend as one more small term (12 products + C)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[v] = acc[0];
  }
}
// worst deviation of the probe's results from the exact sums, in u of the sum of the magnitudes (gfx950: 7.6)
inline double dense_h16_probe_worst(const float *res) {
  double worst = 0.0;
  for (int v = 0; v < kH16ProbeVariants; v++) {
    const double sm = (double)(float)(_Float16)dense_h16_probe_small(v);
    const int nsmall = v == 1 ? 13 : 15;
    const double exact = 1048576.0 + nsmall * sm, mag = 1048576.0 + nsmall * fabs(sm);
    const double dev = fabs((double)res[v] - exact) / (5.9604644775390625e-08 * mag);
    if (!(dev <= worst)) worst = dev;  // (a NaN result is the worst)
  }
  return worst;
}
// the assumption is 1 u per product and addend of the largest magnitude: 17 u for a full instruction
constexpr double kH16ProbeLimit = 12.0;

// The same question on RANDOM operands (r05): `rounds` instructions, each 32 x 32 outputs whose 16 products and addend
// have random signs, random 11- / 24-bit significands and exponents spread over 13 (the fp16 factors) and 32 (the
// addend) binades -- every alignment the eight hand-made patterns above do not think of.  Every lane rebuilds the
// operands of its 16 outputs from the same hash, adds the 17 terms in fp64 (the products of two fp16 numbers are exact
// there, the sum is good to 2^-53 of the largest term) and compares: out[0] = the worst |result - exact| in u of the
// largest magnitude among the products, the addend and the exact result (the result's own rounding is half an ulp
// = up to 1 u of it).  65 536 cases per context at 64 rounds, a few microseconds.
__device__ __forceinline__ uint32_t h16_probe_hash(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return h;
}
// a random fp16 number: sign, 10 random fraction bits, exponent -8 .. +4
__device__ __forceinline__ _Float16 h16_probe_f16(uint32_t h) {
  const int e = (int)((h >> 11) % 13u) - 8;
  const float m = 1.0f + (float)(h & 1023u) * (1.0f / 1024.0f);
  const float v = __builtin_ldexpf(m, e);
  return (_Float16)((h >> 31) ? -v : v);
}
// a random fp32 addend: sign, 23 random fraction bits, exponent -20 .. +11
__device__ __forceinline__ float h16_probe_f32(uint32_t h, uint32_t h2) {
  const int e = (int)((h2 >> 8) % 32u) - 20;
  const float m = 1.0f + (float)(h & 0x7FFFFFu) * (1.0f / 8388608.0f);
  return __builtin_ldexpf((h >> 31) ? -m : m, e);
}
__global__ __launch_bounds__(64) void k_dense_h16_probe_random(float *__restrict__ out, int rounds) {
  const int lane = threadIdx.x, col = lane & 31, half = lane >> 5;
  double worst = 0.0;
  for (int r = 0; r < rounds; r++) {
    // operand element A[row][k] = f16(hash(r, row, k)), B[k][col] = f16(hash(r, 64 + col, k)), C[row][col]
    h16x8 a, b;
    for (int i = 0; i < 8; i++) {
      const int k = 8 * half + i;
      a[i] = h16_probe_f16(h16_probe_hash((uint32_t)r, (uint32_t)col, (uint32_t)k));        // my A row is `col`
      b[i] = h16_probe_f16(h16_probe_hash((uint32_t)r, 64u + (uint32_t)col, (uint32_t)k));
    }
    f32x16 acc;
    for (int i = 0; i < 16; i++) {
      const int row = 8 * (i / 4) + 4 * half + i % 4;
      const uint32_t hc = h16_probe_hash((uint32_t)r, 128u + (uint32_t)row, (uint32_t)col);
      acc[i] = h16_probe_f32(hc, h16_probe_hash(hc, 7u, 9u));
    }
    const f32x16 c0 = acc;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; i++) {
      const int row = 8 * (i / 4) + 4 * half + i % 4;
      double exact = (double)c0[i], big = fabs((double)c0[i]);
      for (int k = 0; k < 16; k++) {
        const double p = (double)(float)h16_probe_f16(h16_probe_hash((uint32_t)r, (uint32_t)row, (uint32_t)k)) *
                         (double)(float)h16_probe_f16(h16_probe_hash((uint32_t)r, 64u + (uint32_t)col, (uint32_t)k));
        exact += p;
        big = fabs(p) > big ? fabs(p) : big;
      }
      big = fabs(exact) > big ? fabs(exact) : big;
      const double dev = fabs((double)acc[i] - exact) / (5.9604644775390625e-08 * big);
      if (!(dev <= worst)) worst = dev;  // (a NaN is the worst)
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double t = __shfl_xor(worst, o);
    if (!(t <= worst)) worst = t;
  }
  if (lane == 0) out[0] = (float)worst;
}

// The scan.  One workgroup = four waves; a wave owns 64 rows (two 32-row tiles whose fragments stay in registers
// for a whole pass over the batch), the workgroup 256.  The hypotheses come through a ring of 32-hypothesis tiles in
// LDS, filled two tiles ahead by global_load_lds (no staging registers), one barrier per tile.
template <int NR, bool SKIP_AMB = false, int DBG = 0>  // SKIP_AMB / DBG: tools/h16_bench.hip times parts of the loop
                                                        // (1-3: parts alone; 7: s_memtime per phase; 8: packed classification)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_scan_dense_h16(
    const uint4 *__restrict__ afrag, const float *__restrict__ bs, size_t row_begin, size_t row_end,
    size_t rows_per_block, const _Float16 *__restrict__ xh, const float *__restrict__ thr4, uint32_t H,
    uint32_t *__restrict__ votes, unsigned long long *__restrict__ amb_list, unsigned int *__restrict__ amb_counts,
    uint32_t seg_cap, uint32_t hyp_base, const uint32_t *__restrict__ h_dev, const uint32_t *__restrict__ sel,
    const uint32_t *__restrict__ range_dev) {
  static_assert(NR == 64, "fragment bookkeeping assumes 64 padded unknowns");
  if (range_dev) {
    row_begin = range_dev[0];
    row_end = range_dev[1];
    if (row_begin >= row_end) return;  // workgroup-uniform
    const size_t passes = (row_end - row_begin + kH16RowsPerWg - 1) / kH16RowsPerWg;
    rows_per_block = (passes + gridDim.x - 1) / gridDim.x * kH16RowsPerWg;
  }
  if (h_dev) {
    const uint32_t ht = *h_dev, hd = ht > hyp_base ? ht - hyp_base : 0u;
    H = hd < H ? hd : H;
    if (H == 0) return;  // workgroup-uniform
  }
  // The ring is an object of its own: the compiler orders every LDS atomic / write that MAY alias a pending
  // global_load_lds behind an s_waitcnt vmcnt(0) -- with the ring inside the dynamic array that was every vote and
  // worklist counter update, i.e. a wait for the tiles just requested
  __shared__ __attribute__((aligned(16))) unsigned char ring[kH16Slots * kH16TileBytes];
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const uint32_t NT = (H + 31) / 32;
  // per hypothesis (padded to tiles) (-a, -a, -ph, -ph): the two packed operands as they are used -- built from one
  // loaded vector with `{th.x, th.x}` hipcc 7.2 compared the running minimum below with th.x instead of the band
  // (seen in the ISA; every pair went to the worklist) -- and the band's bit pattern in its own array
  float *thl = (float *)smraw;
  float *s_b = thl + 4 * 32 * NT;                                      // 2 x 256 right-hand sides
  uint32_t *s_cnt = (uint32_t *)(s_b + 2 * kH16RowsPerWg);             // 32 NT vote counters
  uint32_t *s_band = s_cnt + 32 * NT;
  // hypothesis index of every position (selections: sel[]), staged here so that the tile loop holds no load into a
  // register: with the sel[] read in the worklist branch the wait-count pass put an s_waitcnt vmcnt(0) into EVERY
  // iteration (a pending load into a register the loop reuses), i.e. it waited for the tiles just requested
  uint32_t *s_hid = s_band + 32 * NT;
  uint32_t *s_amb = s_hid + 32 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  for (uint32_t h = tid; h < 32 * NT; h += 256) {
    s_cnt[h] = 0;
    const bool in = h < H;
    // past the batch: -a = +inf, band = 0 -- d = +inf for every row, never counted, never ambiguous
    const float na = in ? thr4[4 * (size_t)h] : __builtin_inff(), nph = in ? thr4[4 * (size_t)h + 2] : -1.0f;
    thl[4 * h] = na, thl[4 * h + 1] = na;
    thl[4 * h + 2] = nph, thl[4 * h + 3] = nph;
    s_band[h] = in ? __builtin_bit_cast(uint32_t, thr4[4 * (size_t)h + 1]) : 0u;
    s_hid[h] = in ? (sel ? sel[hyp_base + h] : hyp_base + h) : 0u;
  }
  if (tid == 0) *s_amb = amb_counts[blockIdx.x];
  size_t lo = row_begin + (size_t)blockIdx.x * rows_per_block;
  size_t hi = lo + rows_per_block < row_end ? lo + rows_per_block : row_end;
  if (lo >= hi) return;  // workgroup-uniform (nothing of mine is in flight yet)
  // tile T of the batch -> ring slot: wave w copies the pieces (kb = w, part 0 and 1); lane l fetches the 16 bytes of
  // hypothesis T * 32 + l % 32, unknowns w * 16 + 8 (l / 32) + 0..7
  auto issue = [&](uint32_t T, uint32_t slot) {
    const uint32_t h = T * 32 + col;
    const _Float16 *g = xh + (size_t)(h < H ? h : 0) * 128 + wave * 16 + 8 * half;
    unsigned char *dst = ring + slot * kH16TileBytes + (wave * 2) * 1024;
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(g + 64, (__attribute__((address_space(3))) void *)(dst + 1024), 16, 0, 0);
  };
  unsigned long long tphase[5] = {0, 0, 0, 0, 0};
  const unsigned long long tstart = DBG == 7 ? __builtin_readcyclecounter() : 0;
  uint32_t q = 0;  // tiles started, over all passes: slot = q % 3, hypothesis tile = q % NT
  for (uint32_t k = 0; k + 1 < (uint32_t)kH16Slots; k++) issue(k % NT, k);
  uint32_t pass = 0;
  for (size_t base = lo; base < hi; base += kH16RowsPerWg, pass++) {
    // my 64 rows' fragments: 16 coalesced 16-byte loads per lane
    h16x8 a[2][4][2];
    {
      const size_t t0 = (base + 64 * (size_t)wave) / 32;
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int kb = 0; kb < 4; kb++)
#pragma unroll
          for (int part = 0; part < 2; part++)
            a[m][kb][part] =
                __builtin_bit_cast(h16x8, afrag[(((t0 + m) * 4 + kb) * 2 + part) * 64 + lane]);
    }
    float *sb = s_b + (pass & 1u) * kH16RowsPerWg;
    {
      const size_t row = base + tid;
      sb[tid] = row < hi ? bs[row] : __builtin_inff();
    }
    for (uint32_t T = 0; T < NT; T++, q++) {
      const uint32_t slot = q % kH16Slots;
      unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0;
      if (DBG == 7) {
        __builtin_amdgcn_sched_barrier(0);
        tq0 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
      }
      // the two loads of tile q have landed (younger: the two of tile q + 1; at the top of a pass everything is
      // waited for -- the fragments are needed now)
      if (DBG != 3) {
        // (r05, measured and not kept: one barrier per TWO tiles -- the barrier phase of tools/h16_bench's s_memtime
        // clocks falls from 470 to 310 cycles per tile, the launch takes 0.84 instead of 0.825 ms: the requests run one
        // tile less ahead)
        if (T == 0)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kH16Slots - 2)) : "memory");
        __syncthreads();  // tile q is complete in LDS; everybody is done reading tile q - 1
        issue((T + kH16Slots - 1) % NT, (q + kH16Slots - 1) % kH16Slots);
      }
      if (DBG == 7) {
        __builtin_amdgcn_sched_barrier(0);
        tq1 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
      }
      h16x8 x[4][2];
      const unsigned char *sl = ring + slot * kH16TileBytes + lane * 16;
#pragma unroll
      for (int kb = 0; kb < 4; kb++)
#pragma unroll
        for (int part = 0; part < 2; part++) x[kb][part] = *(const h16x8 *)(sl + (kb * 2 + part) * 1024);
      const f32x2 na = *(const f32x2 *)(thl + 4 * (T * 32 + col)), nph = *(const f32x2 *)(thl + 4 * (T * 32 + col) + 2);
      const uint32_t band = s_band[T * 32 + col];
      if (DBG == 7) {
        __builtin_amdgcn_sched_barrier(0);
        tq2 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[m][i] = 0.0f;
      // the small terms first, then a1 x1 (header: accumulation)
      if (DBG == 2) {
        acc[0][0] = (float)x[0][0][0] + (float)x[1][1][1] + (float)x[2][0][2] + (float)x[3][1][3];
        acc[1][5] = (float)x[0][1][0] + (float)x[1][0][1] + (float)x[2][1][2] + (float)x[3][0][3] + (float)a[1][2][1][2];
      }
#pragma unroll
      for (int kb = 0; DBG != 2 && kb < 4; kb++) {
#pragma unroll
        for (int m = 0; m < 2; m++)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][kb][0], x[kb][1], acc[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 2; m++)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][kb][1], x[kb][0], acc[m], 0, 0, 0);
      }
#pragma unroll
      for (int kb = 0; DBG != 2 && kb < 4; kb++)
#pragma unroll
        for (int m = 0; m < 2; m++)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][kb][0], x[kb][0], acc[m], 0, 0, 0);
      if (DBG == 1 || DBG == 3) {  // matrix instructions alone: one value of each accumulator kept alive
        const uint32_t cc = (__builtin_bit_cast(uint32_t, acc[0][3]) >> 31) + (__builtin_bit_cast(uint32_t, acc[1][7]) >> 31);
        if (cc) lds_add_u32(&s_cnt[T * 32 + col], cc);
        continue;
      }
      if (DBG == 7) {
        __builtin_amdgcn_sched_barrier(0);
        tq3 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
      }
      // classify: lane = hypothesis column T * 32 + col; register i of tile m = row 8 (i / 4) + 4 half + i % 4.
      // The sign bits of d are shifted into `bits` value by value: value j = 16 m + 4 g + rr ends at bit 31 - j.
      uint32_t bits = 0, dmin[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int m = 0; m < 2; m++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 bq = *(const f32x4 *)(sb + 64 * wave + 32 * m + 8 * g + 4 * half);
#pragma unroll
          for (int p = 0; p < 2; p++) {
            if (DBG != 8) {
              // plain v_fma_f32, not v_pk_fma_f32: a packed fp32 instruction takes two issue slots and does not run beside
              // the other wave's matrix instructions (tools/mfma_rate), and with the worklist branch compiled in the
              // packed form of this loop came with 16 s_nop per tile -- measured (tools/h16_bench, 2 M x 1024, r05): 0.82 ms
              // against 0.90-0.95 ms at the bench's band, 0.73-0.75 against 0.80-0.88 ms with an empty band
              const float r0 = __builtin_fmaf(p ? bq.z : bq.x, nph.x, acc[m][4 * g + 2 * p]);
              const float r1 = __builtin_fmaf(p ? bq.w : bq.y, nph.x, acc[m][4 * g + 2 * p + 1]);
              const uint32_t d0 = __builtin_bit_cast(uint32_t, __builtin_fmaf(r0, r0, na.x));
              const uint32_t d1 = __builtin_bit_cast(uint32_t, __builtin_fmaf(r1, r1, na.x));
              bits = __builtin_amdgcn_alignbit(bits, d0, 31);
              bits = __builtin_amdgcn_alignbit(bits, d1, 31);
              const uint32_t mn = d0 < d1 ? d0 : d1;
              dmin[m] = mn < dmin[m] ? mn : dmin[m];
              continue;
            }
            // DBG 8 (tools/h16_bench): the packed form, until r05 the product's
            const f32x2 ac = {acc[m][4 * g + 2 * p], acc[m][4 * g + 2 * p + 1]};
            const f32x2 bb = {p ? bq.z : bq.x, p ? bq.w : bq.y};
            const f32x2 r = __builtin_elementwise_fma(bb, nph, ac);
            const f32x2 d = __builtin_elementwise_fma(r, r, na);
            const u32x2 du = __builtin_bit_cast(u32x2, d);
            bits = __builtin_amdgcn_alignbit(bits, du.x, 31);
            bits = __builtin_amdgcn_alignbit(bits, du.y, 31);
            const uint32_t mn = du.x < du.y ? du.x : du.y;  // (halves first: cells.h on hipcc 7.2 and packed results)
            dmin[m] = mn < dmin[m] ? mn : dmin[m];
          }
        }
      }
      const uint32_t c = (uint32_t)__builtin_popcount(bits);
      // Rare per lane (some pair of the lane's 32 sits in the band), not per wave: at the bench's band one wave-tile in
      // five has such a lane, and a branch is taken by the wave.  So the branch is kept short: only the 32-row half whose
      // minimum fell into the band is evaluated again, and the band's mask comes out of sign bits like the inliers' --
      // z = band - d is negative exactly when d > band (a float difference has the sign of the exact one), `bits` already
      // holds d < 0 in the same order: ambiguous = neither.  (Until r05: all 32 values again with a compare, a select and
      // an or each -- 150 vector instructions per taken branch against 135 for the whole tile.)
      if (!SKIP_AMB && (dmin[0] < dmin[1] ? dmin[0] : dmin[1]) <= band) {
        const float bandf = __builtin_bit_cast(float, band);
        uint32_t am = 0;  // bit 31 - j: value j = 16 m + 4 g + rr is in the band
#pragma unroll
        for (int m = 0; m < 2; m++) {
          if (dmin[m] > band) continue;  // (lanes of the branch whose other half was hit)
          uint32_t zb = 0;
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const f32x4 bq = *(const f32x4 *)(sb + 64 * wave + 32 * m + 8 * g + 4 * half);
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
              const float r = __builtin_fmaf(bq[rr], nph.x, acc[m][4 * g + rr]);
              const float d = __builtin_fmaf(r, r, na.x);
              zb = __builtin_amdgcn_alignbit(zb, __builtin_bit_cast(uint32_t, bandf - d), 31);
            }
          }
          const uint32_t neg = (bits >> (16 * (1 - m))) & 0xFFFFu;  // d < 0, value 4 g + rr of this half at bit 15 - (4 g + rr)
          am |= (~zb & ~neg & 0xFFFFu) << (16 * (1 - m));
        }
        unsigned slot_w = lds_add_rtn_u32(s_amb, (uint32_t)__builtin_popcount(am));
        const unsigned long long hid = (unsigned long long)s_hid[T * 32 + col];
        while (am) {
          const int j = __builtin_clz(am);  // value j = 16 m + 4 g + rr -> row 32 m + 8 g + 4 half + rr
          am &= ~(0x80000000u >> j);
          if (slot_w < seg_cap)
            amb_list[(size_t)blockIdx.x * seg_cap + slot_w] =
                ((unsigned long long)(base + 64 * wave + 32 * (j >> 4) + 8 * ((j >> 2) & 3) + 4 * half + (j & 3)) << 32) |
                hid;
          slot_w++;
        }
      }
      if (c) lds_add_u32(&s_cnt[T * 32 + col], c);
      if (DBG == 7) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long tq4 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        tphase[0] += tq1 - tq0, tphase[1] += tq2 - tq1, tphase[2] += tq3 - tq2, tphase[3] += tq4 - tq3;
        tphase[4] += 1;
      }
    }
  }
  if (DBG == 7 && lane == 0) {
    for (int i = 0; i < 5; i++) amb_list[((size_t)blockIdx.x * 4 + wave) * 8 + i] = tphase[i];
    amb_list[((size_t)blockIdx.x * 4 + wave) * 8 + 5] = __builtin_readcyclecounter() - tstart;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS write of mine may still be in flight when the LDS is freed
  __syncthreads();
  for (uint32_t h = tid; h < H; h += 256) {
    const uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[s_hid[h]], c);
  }
  if (tid == 0) amb_counts[blockIdx.x] = *s_amb;
}

inline size_t dense_h16_lds(uint32_t H) {
  const size_t NT = (H + 31) / 32;
  return sizeof(float) * (4 * 32 * NT + 2 * kH16RowsPerWg) + sizeof(uint32_t) * (3 * 32 * NT + 1);  // + the static ring
}

}  // namespace lsqr
