// phantom_h16.h -- the plane phantom's agree() scan (PlanePhantomUSCalibrationParametersEstimator.cxx:73-135) as a FILTER on
// the fp16 matrix cores, the arrangement of us_h16.h.
//
// The reference's error IS a dot product: 31 terms, the data row  a = [u R2 (9) | v R2 (9) | R2 (9) | t2 (3) | 1]  against
// the hypothesis column  e = [par[11 .. 40] | par[2]]  (phantom.h: row_entry / err) -- so 32 frames against 32
// hypotheses are ONE 32 x 32 x 32 matrix product, two 16-slot blocks, and with both sides as two-way fp16 splits six
// instructions (lo x hi, hi x lo of both blocks first, hi x hi last: dense_h16.h, accumulation).  No factorisation of the
// 30 products is needed (the packed fp32 filter of phantom.h evaluates the factored form and has to check it).
// Slot groups and their scales as in us_h16.h: u / v R (slots 0 .. 17) X Rm, R (18 .. 26) Rm, t2 (27 .. 29) X, the one 1;
// the hypothesis side carries x''_k = x_k G / pa_k, max |x''_k| = 2^15, err'' = err G.
// Error against the exact err'' (u = 2^-24, S'' = 2^15 sum |x''_k|): splits 4 u (2^-23 per factor), flushed operands 2 u,
// dropped lo x lo 4 u, accumulation 1 u per product and addend (measured 0.5 u: dense_h16.h) over the two hi x hi
// instructions 34 u, the four before them 0.1 u:  |e16 - err''| <= 45 u S'';  the reference's fp64 31-term sum is within
// 1e-13 S'' of exact (phantom.h: 64 u64 W).   E = 1.01 (45 u + 1e-13) S''.
// The reference compares err^2 with delta^2, i.e. |err| with T = mc.thr (models.h: square_threshold); with D = T G:
//   |e16| < D - E => agrees,  |e16| > D + E => does not;  thresholds on d = e16^2 - a as in dense_h16.h; in between the
//   exact predicate decides (worklist, k_us_recheck_seg<PhantomModel>).
#pragma once
#include <hip/hip_runtime.h>

#include "dense_h16.h"
#include "phantom.h"
#include "us_h16.h"

namespace lsqr {

constexpr int kPh16Wg = 512;           // eight waves x 64 frames = 512 frames per pass
constexpr int kPh16FrameTile = 4096;   // bytes of fragments per 32 frames: 2 blocks x 2 parts x 64 lanes x 16 B
constexpr int kPh16HypChunk = 1024;    // hypotheses per launch: their fragments (128 KiB) stay in LDS

__device__ __host__ inline int ph16_group(int k) { return k < 18 ? 0 : k < 27 ? 1 : k < 30 ? 2 : 3; }

// Once per upload: afrag[((tile * 2 + kb) * 2 + part) * 64 + lane] = the eight fp16 values (part 0: hi, 1: lo) of
// frame tile * 32 + lane % 32, slots kb * 16 + 8 (lane / 32) + 0..7.  One wave per tile.
__global__ __launch_bounds__(256) void k_phantom_rows_h16(const double *__restrict__ data, size_t stride, size_t n,
                                                          Us16Scales sc, uint4 *__restrict__ afrag, size_t n_tiles) {
  const size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const size_t fr = tile * 32 + r;
  const bool live = fr < n;
  const double *p = data + (live ? fr : 0) * stride;
  double rec[PhantomModel::REC];
#pragma unroll
  for (int i = 0; i < PhantomModel::ND; i++) rec[i] = (i == 12) ? 0.0 : p[i];
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    h16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int k = kb * 16 + 8 * half + i;
      const double a = k < 31 ? PhantomModel::row_entry(rec, k) : 0.0;
      const double s = live && k < 31 ? a * sc.pa[ph16_group(k)] : 0.0;
      _Float16 h1, h2;
      h16_split(s, h1, h2);
      hi[i] = h1, lo[i] = h2;
    }
    afrag[((tile * 2 + kb) * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    afrag[((tile * 2 + kb) * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
  }
}

// Once per batch (or compact selection): per 32-hypothesis tile xfrag[((tile * 2 + kb) * 2 + part) * 64 + lane] and per
// hypothesis thr4 = (-a, band bits, 0, 0).
__global__ __launch_bounds__(256) void k_phantom_prep_h16(const double *__restrict__ sp, int sp_stride, uint32_t H, double thr,
                                                          double X, double Rm, Us16Scales sc, uint4 *__restrict__ xfrag,
                                                          float *__restrict__ thr4) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= ((H + 31) / 32) * 32) return;
  double x[32];
  for (int k = 0; k < 32; k++) x[k] = 0.0;
  bool finite = h < H;
  if (h < H) {
    const double *par = sp + (size_t)h * sp_stride;
    for (int k = 0; k < 30; k++) x[k] = par[11 + k];
    x[30] = par[2];
    for (int k = 0; k < 31; k++) finite = finite && x[k] == x[k] && fabs(x[k]) < 1e100;
  }
  double wmax = 0.0;
  for (int k = 0; k < 31; k++) {
    const double w = fabs(x[k]) / sc.pa[ph16_group(k)];
    wmax = w > wmax ? w : wmax;
  }
  const double u = 5.9604644775390625e-08;
  const float Gf = finite && wmax > 0.0 ? (float)(32768.0 / wmax * (1.0 - 1e-7)) : 1.0f;  // (rounded: never above 2^15)
  const double G = (double)Gf;
  double xs[32], S = 0.0;
  for (int k = 0; k < 32; k++) {
    xs[k] = k < 31 ? x[k] * G / sc.pa[ph16_group(k)] : 0.0;
    S += 32768.0 * fabs(xs[k]);
  }
  const double E = 1.01 * (45.0 * u + 1e-13) * S;
  // thr = sqrt(delta^2); the reference compares fl(err err) with delta^2: 1e-12 covers the roundings of both
  const double tin = thr * G * (1.0 - 1e-12) - E, tout = thr * G * (1.0 + 1e-12) + E;
  const bool live = finite && thr > 0.0 && thr < 1e150 && X < 1e15 && Rm < 1e15 && G < 1e30 && G > 1e-30 && tout < 9.0e18 &&
                    tout == tout && S == S;
  float a = 0.0f, band = __builtin_bit_cast(float, 0x7F7FFFFFu);
  if (live) {
    if (tin > 0.0) {
      a = (float)(tin * tin);
      if ((double)a > tin * tin) a = nextafterf(a, 0.0f);
      a *= 0.9999998f;
    }
    float c = (float)(tout * tout);
    if ((double)c < tout * tout) c = nextafterf(c, INFINITY);
    c *= 1.0000002f;
    band = c - a;
  }
  // a NaN among the coefficients (k_estimate_phantom refused the subset): err is NaN for every frame and the reference's
  // '<' false -- never counted, nothing ambiguous (d = e^2 + inf), instead of a worklist entry per frame
  bool has_nan = false;
  for (int k = 0; k < 31; k++) has_nan = has_nan || x[k] != x[k];
  if (has_nan) a = -INFINITY, band = 0.0f;
  if (h < H) {
    thr4[4 * (size_t)h] = -a;
    thr4[4 * (size_t)h + 1] = band;
    thr4[4 * (size_t)h + 2] = 0.0f;
    thr4[4 * (size_t)h + 3] = 0.0f;
  }
  const uint32_t tile = h >> 5, col = h & 31;
#pragma unroll
  for (int kb = 0; kb < 2; kb++)
#pragma unroll
    for (int half = 0; half < 2; half++) {
      h16x8 hi, lo;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        _Float16 h1, h2;
        h16_split(live ? xs[kb * 16 + 8 * half + i] : 0.0, h1, h2);
        hi[i] = h1, lo[i] = h2;
      }
      xfrag[(((size_t)tile * 2 + kb) * 2 + 0) * 64 + col + 32 * half] = __builtin_bit_cast(uint4, hi);
      xfrag[(((size_t)tile * 2 + kb) * 2 + 1) * 64 + col + 32 * half] = __builtin_bit_cast(uint4, lo);
    }
}

// The scan: us_h16.h's arrangement with one component and two 16-slot blocks.  A wave keeps its 64 frames' fragments in
// registers and walks the hypothesis tiles in LDS; per tile twelve matrix instructions (two frame tiles x six), the
// previous tile's classification (per pair of values one packed FMA, two v_alignbit, one v_min3) between them.
__global__ __launch_bounds__(kPh16Wg) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_scan_phantom_h16(
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
  uint4 *s_x = (uint4 *)smraw;                                  // NT tiles x (2 blocks x 2 parts) x 64 lanes
  float *s_na = (float *)(s_x + (size_t)NT * 256);              // (-a, -a) per hypothesis
  uint32_t *s_band = (uint32_t *)(s_na + 2 * 32 * NT);
  uint32_t *s_hid = s_band + 32 * NT;
  uint32_t *s_cnt = s_hid + 32 * NT;
  uint32_t *s_amb = s_cnt + 32 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  for (uint32_t i = tid; i < NT * 256; i += kPh16Wg) s_x[i] = xfrag[i];
  for (uint32_t h = tid; h < 32 * NT; h += kPh16Wg) {
    const bool in = h < H;
    const float na = in ? thr4[4 * (size_t)h] : __builtin_inff();  // past the batch: never counted, never ambiguous
    s_na[2 * h] = na, s_na[2 * h + 1] = na;
    s_band[h] = in ? __builtin_bit_cast(uint32_t, thr4[4 * (size_t)h + 1]) : 0u;
    s_hid[h] = in ? (sel ? sel[hyp_base + h] : hyp_base + h) : 0u;
    s_cnt[h] = 0;
  }
  if (tid == 0) *s_amb = amb_counts[blockIdx.x];
  __syncthreads();
  const size_t passes = (row_end - row_begin + kPh16Wg - 1) / kPh16Wg;
  const size_t per_wg = (passes + gridDim.x - 1) / gridDim.x;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  struct Meta {
    f32x2 na;
    uint32_t band, idx;
  };
  const size_t ps_begin = (size_t)blockIdx.x * per_wg;
  const size_t ps_end = passes < ps_begin + per_wg ? passes : ps_begin + per_wg;
  // a pass = my 64 frames' fragments (eight 16-byte loads per lane) against every hypothesis tile.  (Requesting the next
  // pass's fragments a pass ahead was measured and changes nothing: 0.83 ms against 0.82 ms per 4096 x 1 M.)
  for (size_t ps = ps_begin; ps < ps_end; ps++) {
    const size_t base = row_begin + ps * kPh16Wg + 64 * (size_t)wave;  // my 64 frames (row_begin is a multiple of 32)
    if (base >= row_end) break;                                         // wave-uniform
    h16x8 a[2][2][2];
    const size_t t0 = base / 32;
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int part = 0; part < 2; part++)
          a[m][kb][part] = __builtin_bit_cast(h16x8, afrag[(((t0 + m) * 2 + kb) * 2 + part) * 64 + lane]);
    uint32_t vmask[2];  // register i = row 8 (i / 4) + 4 half + i % 4 of the tile, bit 15 - i
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
    auto chunk = [&](const f32x16 &prv, const f32x2 na, int p, uint32_t &bits, uint32_t &dmin) __attribute__((always_inline)) {
      const f32x2 e = {prv[2 * p], prv[2 * p + 1]};
      const f32x2 d = __builtin_elementwise_fma(e, e, na);
      const u32x2 du = __builtin_bit_cast(u32x2, d);
      bits = __builtin_amdgcn_alignbit(bits, du.x, 31);
      bits = __builtin_amdgcn_alignbit(bits, du.y, 31);
      const uint32_t mn = du.x < du.y ? du.x : du.y;  // (halves first: cells.h on hipcc 7.2 and packed results)
      dmin = mn < dmin ? mn : dmin;
    };
    // one step: hypothesis tile T against BOTH frame tiles (its four fragments are read from LDS once) -- twelve matrix
    // instructions in two independent chains, the sixteen chunks of the previous tile's two accumulators between them
    // (the tile's fragments x[kb * 2 + part] were read from LDS a step earlier; the next tile's are read into xn here,
    // ahead of the matrix instructions, so that their latency is not waited for)
    auto step = [&](f32x16(&cur)[2], const h16x8(&x)[4], h16x8(&xn)[4], const uint4 *sl_next, const f32x16(&prv)[2],
                    const Meta &pm, uint32_t(&bits)[2], uint32_t(&dmin)[2]) __attribute__((always_inline)) {
      bits[0] = bits[1] = 0, dmin[0] = dmin[1] = 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < 4; j++) xn[j] = __builtin_bit_cast(h16x8, sl_next[64 * j]);
      const h16x8 x00 = x[0], x01 = x[1], x10 = x[2], x11 = x[3];
      const f32x16 Z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0][0], x01, Z, 0, 0, 0);
      chunk(prv[0], pm.na, 0, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0][0], x01, Z, 0, 0, 0);
      chunk(prv[0], pm.na, 1, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0][1], x00, cur[0], 0, 0, 0);
      chunk(prv[0], pm.na, 2, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0][1], x00, cur[1], 0, 0, 0);
      chunk(prv[0], pm.na, 3, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1][0], x11, cur[0], 0, 0, 0);
      chunk(prv[0], pm.na, 4, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1][0], x11, cur[1], 0, 0, 0);
      chunk(prv[0], pm.na, 5, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1][1], x10, cur[0], 0, 0, 0);
      chunk(prv[0], pm.na, 6, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1][1], x10, cur[1], 0, 0, 0);
      chunk(prv[0], pm.na, 7, bits[0], dmin[0]);
      __builtin_amdgcn_sched_barrier(0);
      cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][0][0], x00, cur[0], 0, 0, 0);
      chunk(prv[1], pm.na, 0, bits[1], dmin[1]);
      chunk(prv[1], pm.na, 1, bits[1], dmin[1]);
      __builtin_amdgcn_sched_barrier(0);
      cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][0][0], x00, cur[1], 0, 0, 0);
      chunk(prv[1], pm.na, 2, bits[1], dmin[1]);
      chunk(prv[1], pm.na, 3, bits[1], dmin[1]);
      __builtin_amdgcn_sched_barrier(0);
      cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][1][0], x10, cur[0], 0, 0, 0);
      chunk(prv[1], pm.na, 4, bits[1], dmin[1]);
      chunk(prv[1], pm.na, 5, bits[1], dmin[1]);
      __builtin_amdgcn_sched_barrier(0);
      cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][1][0], x10, cur[1], 0, 0, 0);
      chunk(prv[1], pm.na, 6, bits[1], dmin[1]);
      chunk(prv[1], pm.na, 7, bits[1], dmin[1]);
      __builtin_amdgcn_sched_barrier(0);
      // (the results are needed HERE: without this the compiler sinks the sign collection behind the worklist branch
      // of finish(), out of the matrix instructions' shadow)
      asm volatile("" : "+v"(bits[0]), "+v"(bits[1]), "+v"(dmin[0]), "+v"(dmin[1]));
    };
    // popcount of the certain inliers of one accumulator; a lane with a pair in the band writes the worklist
    auto finish = [&](const f32x16 &prv, const Meta &pm, int m, uint32_t bits, uint32_t dmin) -> uint32_t {
      const uint32_t c = (uint32_t)__builtin_popcount(bits & vmask[m]);
      if (dmin <= pm.band) {  // rare (one counter update per lane)
        uint32_t am = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
          const float d1 = __builtin_fmaf(prv[i], prv[i], pm.na.x);
          am |= __builtin_bit_cast(uint32_t, d1) <= pm.band ? 1u << (15 - i) : 0u;
        }
        am &= vmask[m];
        if (am) {
          unsigned slot_w = atomicAdd(s_amb, (unsigned)__builtin_popcount(am));
          const unsigned long long hid = (unsigned long long)s_hid[pm.idx];
          while (am) {
            const int b = 31 - __builtin_clz(am);  // bit 15 - i
            am &= ~(1u << b);
            const int i = 15 - b;
            if (slot_w < seg_cap)
              amb_list[(size_t)blockIdx.x * seg_cap + slot_w] =
                  ((unsigned long long)(base + 32 * m + 8 * (i >> 2) + 4 * half + (i & 3)) << 32) | hid;
            slot_w++;
          }
        }
      }
      return c;
    };
    auto settle = [&](const f32x16(&prv)[2], const Meta &pm, const uint32_t(&bits)[2], const uint32_t(&dmin)[2]) {
      const uint32_t c = finish(prv[0], pm, 0, bits[0], dmin[0]) + finish(prv[1], pm, 1, bits[1], dmin[1]);
      if (c) atomicAdd(&s_cnt[pm.idx], c);
    };
    auto meta_of = [&](uint32_t T) {
      Meta m;
      m.na = *(const f32x2 *)(s_na + 2 * (T * 32 + col));
      m.band = s_band[T * 32 + col];
      m.idx = T * 32 + col;
      return m;
    };
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int i = 0; i < 16; i++) accB[0][i] = accB[1][i] = 0.0f;
    Meta mA, mB;  // of the tile whose results sit in accA / accB
    mB.na = (f32x2){__builtin_inff(), __builtin_inff()}, mB.band = 0, mB.idx = col;  // nothing yet: never counted
    uint32_t bits[2], dmin[2];
    h16x8 xa[4], xb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) xa[j] = __builtin_bit_cast(h16x8, s_x[lane + 64 * j]);
    for (uint32_t T = 0; T < NT; T += 2) {
      const uint32_t T1 = T + 1 < NT ? T + 1 : T, T2 = T + 2 < NT ? T + 2 : T1;  // (past the end: read again, unused)
      step(accA, xa, xb, s_x + (size_t)T1 * 256 + lane, accB, mB, bits, dmin);  // T -> A while T - 1 in B is classified
      settle(accB, mB, bits, dmin);
      mA = meta_of(T);
      if (T + 1 < NT) {  // (workgroup-uniform)
        step(accB, xb, xa, s_x + (size_t)T2 * 256 + lane, accA, mA, bits, dmin);  // T + 1 -> B while T in A is classified
        settle(accA, mA, bits, dmin);
        mB = meta_of(T + 1);
      } else {  // odd count: the last tile sits in A
        accB[0] = accA[0], accB[1] = accA[1], mB = mA;
      }
    }
    {  // the last tile's accumulators
      bits[0] = bits[1] = 0, dmin[0] = dmin[1] = 0xFFFFFFFFu;
#pragma unroll
      for (int p = 0; p < 8; p++) chunk(accB[0], mB.na, p, bits[0], dmin[0]);
#pragma unroll
      for (int p = 0; p < 8; p++) chunk(accB[1], mB.na, p, bits[1], dmin[1]);
      settle(accB, mB, bits, dmin);
    }
  }
  __syncthreads();
  for (uint32_t h = tid; h < H; h += kPh16Wg) {
    const uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[s_hid[h]], c);
  }
  if (tid == 0) amb_counts[blockIdx.x] = *s_amb;
}

inline size_t phantom_h16_lds(uint32_t H) {
  const size_t NT = (H + 31) / 32;
  return NT * 256 * 16 + sizeof(float) * 2 * 32 * NT + sizeof(uint32_t) * (3 * 32 * NT + 1);
}

}  // namespace lsqr
