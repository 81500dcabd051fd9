// us_h16.h -- the US calibrations' agree() scan (SinglePointTargetUSCalibrationParametersEstimator.cxx:74-107 /
// :728-766) as a FILTER on the fp16 matrix cores, the construction of dense_h16.h.
//
// Every component of a frame's error vector is a dot product of a DATA row and a HYPOTHESIS column,
//     e_c = sum_j (u R2_cj) c0_j + (v R2_cj) c1_j + R2_cj t3_j + t2_c * 1 - t1_c             (pointer target: t2_c - p_c, no t1)
// (c0 = T3(:,0), c1 = T3(:,1), t3 = T3(:,3): the scan parameters of us.h), 13 terms, so the three components of 32
// frames against 32 hypotheses are three 32 x 32 x 16 matrix products -- with both sides as two-way fp16 splits, three
// instructions each (lo x hi, hi x lo first, hi x hi last: dense_h16.h, accumulation).  The K slots:
//     A (frame, component c)   [u R2_c0 u R2_c1 u R2_c2 | v R2_c. | R2_c. | t2_c | (c == 0) (c == 1) (c == 2) | 0 0 0]
//     B (hypothesis)           [c0_0 c0_1 c0_2          | c1_.    | t3_.  | 1    | -t1_0 -t1_1 -t1_2          | 0 0 0]
// Slot GROUPS have very different magnitudes (u R ~ 640, R ~ 1, t2 ~ 300), so each group has its own scale on the data
// side, pa_k = 2^15 / (bound of the group's entries: X Rm, Rm, X or 2 X, 1 with X = max |entry|, Rm = max |rotation
// entry| of the upload), and the hypothesis side carries x''_k = x_k G / pa_k with ONE factor G per hypothesis chosen so
// that max_k |x''_k| = 2^15:  e'' = e G.
// Error of a component against the exact e'' (u = 2^-24, S'' = 2^15 sum_k |x''_k| >= sum |a''_k x''_k|):
//   splits' remainders 4 u S'' (2^-23 per factor), flushed operands (< 2^-14) 0.8 u S'', dropped lo x lo 4 u S'' (2^-11 of
//   either factor), accumulation -- measured two
//   guard bits, assumed 1 u of the largest magnitude per product and addend (dense_h16.h) -- 14 u S'' for the hi x hi
//   instruction, 0.03 u S'' for the two before it:  |e16 - e''| <= 23 u S'';  the reference's fp64 components are within
//   1e-13 S'' of exact (us.h: Ee64).   E = 1.01 (23 u + 1e-13) S''.
// | |e16| - |e_ref| | <= sqrt(3) E, so with D = delta G:   v = |e16|^2 <  (D - sqrt3 E)^2 (1 - 1e-6) => agrees,
//   v >= (D + sqrt3 E)^2 (1 + 1e-6) => does not (1e-6: the three roundings of the fp32 sum of squares and of delta^2),
//   in between the exact predicate decides (worklist, k_us_recheck_seg).  Thresholds on d = v - a as in dense_h16.h.
// A hypothesis whose numbers do not fit gets zero operands and a band that holds every finite value; one with a NaN among
// its parameters -a = +inf and band 0 (never counted: the reference compares a NaN).
#pragma once
#include <hip/hip_runtime.h>

#include "dense_h16.h"
#include "us.h"

namespace lsqr {

constexpr int kUs16Wg = 512;           // threads per workgroup: eight waves x 64 frames = 512 frames per pass
constexpr int kUs16FrameTile = 6144;   // bytes of fragments per 32 frames: 3 components x 2 parts x 64 lanes x 16 B
constexpr int kUs16HypChunk = 1024;    // hypotheses per launch: their fragments (64 KiB) stay in LDS

struct Us16Scales {
  double pa[4];  // slot groups: u / v R (0..5), R (6..8), t2 (9), the ones (10..12)
};
template <bool SINGLE>
inline Us16Scales us16_scales(double X, double Rm) {
  Us16Scales s;
  s.pa[0] = 32768.0 / (X * Rm > 0.0 ? X * Rm : 1.0);
  s.pa[1] = 32768.0 / (Rm > 0.0 ? Rm : 1.0);
  s.pa[2] = 32768.0 / ((SINGLE ? 1.0 : 2.0) * (X > 0.0 ? X : 1.0));
  s.pa[3] = 32768.0;
  return s;
}
__device__ __host__ inline int us16_group(int k) { return k < 6 ? 0 : k < 9 ? 1 : k == 9 ? 2 : 3; }

// Once per upload: the frames as A fragments.  afrag[((tile * 3 + c) * 2 + part) * 64 + lane] = the eight fp16 values
// (part 0: hi, 1: lo) of frame tile * 32 + lane % 32, component c, slots 8 (lane / 32) + 0..7.  One wave per tile.
template <bool SINGLE>
__global__ __launch_bounds__(256) void k_us_rows_h16(const double *__restrict__ data, size_t stride, size_t n,
                                                     Us16Scales sc, uint4 *__restrict__ afrag, size_t n_tiles) {
  const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const size_t fr = tile * 32 + r;
  const bool live = fr < n;
  const double *p = data + (live ? fr : 0) * stride;
  const double u = p[13], v = p[14];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    h16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int k = 8 * half + i;
      double a = 0.0;
      if (k < 3) a = u * p[3 * c + k];
      else if (k < 6) a = v * p[3 * c + k - 3];
      else if (k < 9) a = p[3 * c + k - 6];
      else if (k == 9) a = SINGLE ? p[9 + c] : p[9 + c] - p[15 + c];
      else if (k < 13) a = SINGLE && k - 10 == c ? 1.0 : 0.0;
      const double s = live ? a * sc.pa[us16_group(k)] : 0.0;
      _Float16 h1, h2;
      h16_split(s, h1, h2);
      hi[i] = h1, lo[i] = h2;
    }
    afrag[((tile * 3 + c) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    afrag[((tile * 3 + c) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
  }
}

// Once per batch (or compact selection): per 32-hypothesis tile the B fragments xfrag[(tile * 2 + part) * 64 + lane]
// (lane = hypothesis % 32 + 32 half, slots 8 half + 0..7) and per hypothesis thr4 = (-a, band bits, 0, 0).
template <bool SINGLE>
__global__ __launch_bounds__(256) void k_us_prep_h16(const double *__restrict__ sp, int sp_stride, uint32_t H, double delta_sq,
                                                     double X, double Rm, Us16Scales sc, uint4 *__restrict__ xfrag,
                                                     float *__restrict__ thr4) {
  typedef USModel<SINGLE> M;
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= ((H + 31) / 32) * 32) return;
  double x[16];
  for (int k = 0; k < 16; k++) x[k] = 0.0;
  bool finite = h < H;
  if (h < H) {
    const double *par = sp + (size_t)h * sp_stride;
    for (int j = 0; j < 3; j++) {
      x[j] = par[M::T3C + j];
      x[3 + j] = par[M::T3C + 3 + j];
      x[6 + j] = par[M::T3T + j];
      if (SINGLE) x[10 + j] = -par[j];
    }
    x[9] = 1.0;
    for (int k = 0; k < 13; k++) finite = finite && x[k] == x[k] && fabs(x[k]) < 1e100;
  }
  double wmax = 0.0;
  for (int k = 0; k < 13; k++) {
    const double w = fabs(x[k]) / sc.pa[us16_group(k)];
    wmax = w > wmax ? w : wmax;
  }
  const double u = 5.9604644775390625e-08;
  const float Gf = finite && wmax > 0.0 ? (float)(32768.0 / wmax * (1.0 - 1e-7)) : 1.0f;  // (rounded: never above 2^15)
  const double G = (double)Gf;
  double xs[16], S = 0.0;
  for (int k = 0; k < 16; k++) {
    xs[k] = k < 13 ? x[k] * G / sc.pa[us16_group(k)] : 0.0;
    S += 32768.0 * fabs(xs[k]);
  }
  const double E = 1.01 * (23.0 * u + 1e-13) * S;
  const double D = sqrt(delta_sq > 0.0 ? delta_sq : 0.0) * G;
  const double tin = D - 1.7320508075688774 * E, tout = D + 1.7320508075688774 * E;
  const bool live = finite && delta_sq > 0.0 && X < 1e15 && Rm < 1e15 && G < 1e30 && G > 1e-30 && tout < 9.0e18 &&
                    tout == tout && S == S;
  float a = 0.0f, band = __builtin_bit_cast(float, 0x7F7FFFFFu);
  if (live) {
    if (tin > 0.0) a = (float)(tin * tin * (1.0 - 1e-6)) * 0.9999998f;
    float c = (float)(tout * tout * (1.0 + 1e-6)) * 1.0000002f;
    band = c - a;
  }
  // a NaN among the parameters (a minimal solve that was refused): the reference's squared distance is NaN for every
  // frame and '<' false -- never counted, nothing ambiguous (d = v + inf), instead of a worklist entry per frame
  bool has_nan = false;
  for (int k = 0; k < 13; k++) has_nan = has_nan || x[k] != x[k];
  if (has_nan) a = -INFINITY, band = 0.0f;
  if (h < H) {
    thr4[4 * (size_t)h] = -a;
    thr4[4 * (size_t)h + 1] = band;
    thr4[4 * (size_t)h + 2] = 0.0f;
    thr4[4 * (size_t)h + 3] = 0.0f;
  }
  const uint32_t tile = h >> 5, col = h & 31;
#pragma unroll
  for (int half = 0; half < 2; half++) {
    h16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      _Float16 h1, h2;
      h16_split(live ? xs[8 * half + i] : 0.0, h1, h2);
      hi[i] = h1, lo[i] = h2;
    }
    xfrag[((size_t)tile * 2 + 0) * 64 + col + 32 * half] = __builtin_bit_cast(uint4, hi);
    xfrag[((size_t)tile * 2 + 1) * 64 + col + 32 * half] = __builtin_bit_cast(uint4, lo);
  }
}

// The scan.  The fragments of all (<= 1024) hypotheses of the launch and their thresholds are copied into LDS once;
// after that a wave works alone -- 64 frames (two tiles, 12 fragments in registers) per pass against the 32 tiles of
// hypotheses, nine matrix instructions and the classification of 16 (frame, hypothesis) pairs per lane at a time --
// no barrier, no request in flight inside the loop.
template <bool SINGLE>
__global__ __launch_bounds__(kUs16Wg) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_scan_us_h16(
    const uint4 *__restrict__ afrag, size_t n, size_t row_begin, size_t row_end, const uint4 *__restrict__ xfrag,
    const float *__restrict__ thr4, uint32_t H, uint32_t *__restrict__ votes, unsigned long long *__restrict__ amb_list,
    unsigned int *__restrict__ amb_counts, uint32_t seg_cap, uint32_t hyp_base, const uint32_t *__restrict__ h_dev,
    const uint32_t *__restrict__ sel, const uint32_t *__restrict__ range_dev) {
  if (range_dev) {
    row_begin = range_dev[0];
    row_end = range_dev[1];
    if (row_begin >= row_end) return;  // workgroup-uniform
  }
  if (h_dev) {
    const uint32_t ht = *h_dev, hd = ht > hyp_base ? ht - hyp_base : 0u;
    H = hd < H ? hd : H;
    if (H == 0) return;  // workgroup-uniform
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
  const uint32_t NT = (H + 31) / 32;
  uint4 *s_x = (uint4 *)smraw;                                  // NT tiles x 2 parts x 64 lanes
  float *s_na = (float *)(s_x + (size_t)NT * 128);              // (-a, -a) per hypothesis: the packed operand as used
  uint32_t *s_band = (uint32_t *)(s_na + 2 * 32 * NT);
  uint32_t *s_hid = s_band + 32 * NT;
  uint32_t *s_cnt = s_hid + 32 * NT;
  uint32_t *s_amb = s_cnt + 32 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  for (uint32_t i = tid; i < NT * 128; i += kUs16Wg) s_x[i] = xfrag[i];
  for (uint32_t h = tid; h < 32 * NT; h += kUs16Wg) {
    const bool in = h < H;
    // past the batch: -a = +inf, band = 0 -- d = +inf for every frame, never counted, never ambiguous
    const float na = in ? thr4[4 * (size_t)h] : __builtin_inff();
    s_na[2 * h] = na, s_na[2 * h + 1] = na;
    s_band[h] = in ? __builtin_bit_cast(uint32_t, thr4[4 * (size_t)h + 1]) : 0u;
    s_hid[h] = in ? (sel ? sel[hyp_base + h] : hyp_base + h) : 0u;
    s_cnt[h] = 0;
  }
  if (tid == 0) *s_amb = amb_counts[blockIdx.x];
  __syncthreads();
  const size_t passes = (row_end - row_begin + kUs16Wg - 1) / kUs16Wg;
  const size_t per_wg = (passes + gridDim.x - 1) / gridDim.x;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  for (size_t ps = (size_t)blockIdx.x * per_wg; ps < passes && ps < ((size_t)blockIdx.x + 1) * per_wg; ps++) {
    const size_t base = row_begin + ps * kUs16Wg + 64 * (size_t)wave;  // my 64 frames (row_begin is a multiple of 32)
    if (base >= row_end) break;                                         // wave-uniform
    h16x8 a[2][3][2];
    const size_t t0 = base / 32;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int c = 0; c < 3; c++)
#pragma unroll
        for (int part = 0; part < 2; part++)
          a[m][c][part] = __builtin_bit_cast(h16x8, afrag[(((t0 + m) * 3 + c) * 2 + part) * 64 + lane]);
    // which of my 16 rows per tile are frames of the range: register i = row 8 (i / 4) + 4 half + i % 4, bit 15 - i
    uint32_t vmask[2];
#pragma unroll
    for (int m = 0; m < 2; m++) {
      uint32_t k = 0;
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const size_t fr = base + 32 * m + 8 * (i / 4) + 4 * half + (i % 4);
        k |= fr < row_end ? 1u << (15 - i) : 0u;
      }
      vmask[m] = k;
    }
    // Software pipeline over the steps (T, m): the nine matrix instructions of a step are issued with the
    // classification of the PREVIOUS step's accumulators between them -- one chunk (a packed pair of values: three
    // packed FMAs, two v_alignbit, one v_min3) behind each of the first eight -- in SOURCE order, a scheduling barrier
    // behind every group: left to itself the compiler issues the nine back to back and the vector work after them,
    // and the two do not overlap across the waves of a SIMD (measured: 57 cycles per matrix instruction, the sum of
    // both).  Two accumulator sets: A for m = 0, B for m = 1.
    struct Meta {
      f32x2 na;
      uint32_t band, idx, vm;   // idx = T * 32 + col of the step being classified, vm its row mask
      size_t fbase;             // first frame of its tile
    };
    auto step = [&](f32x16(&cur)[3], const h16x8(&am)[3][2], const h16x8 &x1, const h16x8 &x2, const f32x16(&prv)[3],
                    const Meta &pm, uint32_t &bits, uint32_t &dmin) __attribute__((always_inline)) {
      bits = 0, dmin = 0xFFFFFFFFu;
      auto chunk = [&](int p) __attribute__((always_inline)) {
        // (r05: plain v_fma_f32 instead of the packed form -- what the dense filter's classification gained 10 % from --
        // costs THIS loop 17 %, 1.18 -> 1.38 ms per 4096 x 1 M, and the plane phantom's 5 %: here the chunks sit in
        // source order between the matrix instructions and the packed form is the shorter stream; tools/scan_ab.py)
        const f32x2 e0 = {prv[0][2 * p], prv[0][2 * p + 1]}, e1 = {prv[1][2 * p], prv[1][2 * p + 1]},
                    e2 = {prv[2][2 * p], prv[2][2 * p + 1]};
        f32x2 d = __builtin_elementwise_fma(e0, e0, pm.na);
        d = __builtin_elementwise_fma(e1, e1, d);
        d = __builtin_elementwise_fma(e2, e2, d);
        const u32x2 du = __builtin_bit_cast(u32x2, d);
        bits = __builtin_amdgcn_alignbit(bits, du.x, 31);
        bits = __builtin_amdgcn_alignbit(bits, du.y, 31);
        const uint32_t mn = du.x < du.y ? du.x : du.y;  // (halves first: cells.h on hipcc 7.2 and packed results)
        dmin = mn < dmin ? mn : dmin;
      };
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {
#pragma unroll
        for (int i = 0; i < 16; i++) cur[cc][i] = 0.0f;
        cur[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[cc][0], x2, cur[cc], 0, 0, 0);
        chunk(2 * cc);
        __builtin_amdgcn_sched_barrier(0);
        cur[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[cc][1], x1, cur[cc], 0, 0, 0);
        chunk(2 * cc + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {
        cur[cc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[cc][0], x1, cur[cc], 0, 0, 0);
        if (cc < 2) chunk(6 + cc);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // what follows the vector part: votes of the step, the rare worklist appends; returns the step's count
    auto finish = [&](const f32x16(&prv)[3], const Meta &pm, uint32_t bits, uint32_t dmin) -> uint32_t {
      const uint32_t c = (uint32_t)__builtin_popcount(bits & pm.vm);
      if (dmin <= pm.band) {  // rare: a pair of this lane in the band -> worklist (one counter update per lane)
        uint32_t am = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          float d1 = __builtin_fmaf(prv[0][i], prv[0][i], pm.na.x);
          d1 = __builtin_fmaf(prv[1][i], prv[1][i], d1);
          d1 = __builtin_fmaf(prv[2][i], prv[2][i], d1);
          am |= __builtin_bit_cast(uint32_t, d1) <= pm.band ? 1u << (15 - i) : 0u;
        }
        am &= pm.vm;
        if (am) {
          unsigned slot_w = atomicAdd(s_amb, (unsigned)__builtin_popcount(am));
          const unsigned long long hid = (unsigned long long)s_hid[pm.idx];
          while (am) {
            const int b = 31 - __builtin_clz(am);  // bit 15 - i
            am &= ~(1u << b);
            const int i = 15 - b;
            if (slot_w < seg_cap)
              amb_list[(size_t)blockIdx.x * seg_cap + slot_w] =
                  ((unsigned long long)(pm.fbase + 8 * (i >> 2) + 4 * half + (i & 3)) << 32) | hid;
            slot_w++;
          }
        }
      }
      return c;
    };
    f32x16 accA[3], accB[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++)
#pragma unroll
      for (int i = 0; i < 16; i++) accB[cc][i] = 0.0f;
    Meta mA, mB;  // of the step whose results sit in accA / accB
    mB.na = (f32x2){__builtin_inff(), __builtin_inff()}, mB.band = 0, mB.idx = col, mB.vm = 0, mB.fbase = base;
    uint32_t cA = 0;  // votes of (T, m = 0), added to (T, m = 1)'s when that step is classified
    for (uint32_t T = 0; T < NT; T++) {
      const h16x8 x1 = __builtin_bit_cast(h16x8, s_x[(T * 2 + 0) * 64 + lane]);
      const h16x8 x2 = __builtin_bit_cast(h16x8, s_x[(T * 2 + 1) * 64 + lane]);
      const f32x2 na = *(const f32x2 *)(s_na + 2 * (T * 32 + col));
      const uint32_t band = s_band[T * 32 + col];
      uint32_t bits, dmin;
      // (T, 0) -> A while (T - 1, 1) in B is classified
      step(accA, a[0], x1, x2, accB, mB, bits, dmin);
      {
        const uint32_t c = cA + finish(accB, mB, bits, dmin);
        if (c) atomicAdd(&s_cnt[mB.idx], c);
      }
      mA.na = na, mA.band = band, mA.idx = T * 32 + col, mA.vm = vmask[0], mA.fbase = base;
      // (T, 1) -> B while (T, 0) in A is classified
      step(accB, a[1], x1, x2, accA, mA, bits, dmin);
      cA = finish(accA, mA, bits, dmin);
      mB = mA, mB.vm = vmask[1], mB.fbase = base + 32;
    }
    {  // the last step's accumulators
      uint32_t bits = 0, dmin = 0xFFFFFFFFu;
#pragma unroll
      for (int p = 0; p < 8; p++) {
        const f32x2 e0 = {accB[0][2 * p], accB[0][2 * p + 1]}, e1 = {accB[1][2 * p], accB[1][2 * p + 1]},
                    e2 = {accB[2][2 * p], accB[2][2 * p + 1]};
        f32x2 d = __builtin_elementwise_fma(e0, e0, mB.na);
        d = __builtin_elementwise_fma(e1, e1, d);
        d = __builtin_elementwise_fma(e2, e2, d);
        const u32x2 du = __builtin_bit_cast(u32x2, d);
        bits = __builtin_amdgcn_alignbit(bits, du.x, 31);
        bits = __builtin_amdgcn_alignbit(bits, du.y, 31);
        const uint32_t mn = du.x < du.y ? du.x : du.y;
        dmin = mn < dmin ? mn : dmin;
      }
      const uint32_t c = cA + finish(accB, mB, bits, dmin);
      if (c) atomicAdd(&s_cnt[mB.idx], c);
    }
  }
  __syncthreads();
  for (uint32_t h = tid; h < H; h += kUs16Wg) {
    const uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[s_hid[h]], c);
  }
  if (tid == 0) amb_counts[blockIdx.x] = *s_amb;
}

inline size_t us_h16_lds(uint32_t H) {
  const size_t NT = (H + 31) / 32;
  return NT * 128 * 16 + sizeof(float) * 2 * 32 * NT + sizeof(uint32_t) * (3 * 32 * NT + 1);
}

// exact decision of the segmented worklist (one block per segment); out_max[0] = largest segment fill
template <class M>
__global__ __launch_bounds__(1024) void k_us_recheck_seg(const double *__restrict__ data, size_t stride,
                                                        const double *__restrict__ sp, int sp_stride, ModelConsts mc,
                                                        const unsigned long long *__restrict__ amb_list,
                                                        unsigned int *__restrict__ amb_counts, uint32_t seg_cap,
                                                        uint32_t *__restrict__ votes, unsigned int *__restrict__ out_max) {
  // The band's pairs belong to few hypotheses (the good ones: every frame near their threshold), and a device-wide
  // atomic on one address is served by the memory side, one at a time, for all eight XCDs: 150 k votes on a few dozen
  // counters were most of this kernel's 59 us (plane phantom, r05).  Votes are first added up per workgroup in a
  // direct-mapped LDS table keyed by the hypothesis index (a launch's 1024 consecutive hypotheses never collide; a
  // key whose slot is taken by another goes to the global counter as before), one global atomic per used slot at the end.
  constexpr unsigned kSlots = 4096;
  __shared__ uint32_t s_tag[kSlots], s_votes[kSlots];
  const unsigned filled = amb_counts[blockIdx.x];
  if (filled == 0) return;  // workgroup-uniform
  for (unsigned i = threadIdx.x; i < kSlots; i += blockDim.x) s_tag[i] = 0xFFFFFFFFu, s_votes[i] = 0;
  if (threadIdx.x == 0) atomicMax(out_max, filled);
  __syncthreads();
  const unsigned total = filled < seg_cap ? filled : seg_cap;
  for (unsigned e = threadIdx.x; e < total; e += blockDim.x) {  // (dependent loads: one entry per thread where it fits)
    const unsigned long long v = amb_list[(size_t)blockIdx.x * seg_cap + e];
    const size_t row = (size_t)(v >> 32);
    const uint32_t h = (uint32_t)(v & 0xffffffffu);
    double x[M::REC];
    M::load(data + row * stride, mc, x);
    if (M::agree(sp + (size_t)h * sp_stride, x, mc)) {
      const unsigned slot = h & (kSlots - 1);
      const uint32_t was = atomicCAS(&s_tag[slot], 0xFFFFFFFFu, h);
      if (was == 0xFFFFFFFFu || was == h)
        atomicAdd(&s_votes[slot], 1u);
      else
        atomicAdd(&votes[h], 1u);
    }
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < kSlots; i += blockDim.x)
    if (s_votes[i]) atomicAdd(&votes[s_tag[i]], s_votes[i]);
  if (threadIdx.x == 0) amb_counts[blockIdx.x] = 0;
}

}  // namespace lsqr
