// cells_h16.h -- level 2 of the plane's two-level scan (cells.h: k_scan_pairs) on the fp16 matrix cores.
//
// k_scan_pairs evaluates s = n . x' + d0 for ONE surviving hypothesis and the 512 observations of a cell with 12 packed
// fp32 FMAs, then needs 8 compares / ballots and 4 minima to classify them: 32 vector instructions a pair, vector issue
// bound.  Here the survivors of a cell are COMPACTED into tiles of 32 and one `v_mfma_f32_32x32x16_f16` evaluates s for
// 32 hypotheses x 32 observations: both operands as two-way fp16 splits (dense_h16.h), the 16 K-slots hold the three
// partial products of the 4-term sum at once --
//     A (observations, per cell)   [p1x p1y p1z ONE | p1x p1y p1z ONE | p2x p2y p2z 0 | 0 0 0 0]
//     B (hypotheses, per tile)     [n1x n1y n1z d1  | n2x n2y n2z d2  | n1x n1y n1z 0 | 0 0 0 0]
// with p = x' px (px = 2^14 / the power of two above the cell's largest half extent), n'' = n 2^14, ONE = 2^14,
// d = d0 px, each split hi + lo in fp16 -- and the result tile has the hypothesis on the LANE and 16 observations in
// the registers: a hypothesis' votes are counted per lane (sign bits collected with v_alignbit_b32, no ballots, no
// scalar unit), 2 vector instructions per value instead of 4.
//
// Error of s'' = s 2^14 px against the exact n . (x - a) 2^14 px, u = 2^-24, rr = sum |n_i| h_i, R = the power of two
// above max h_i:
//   as in cells.h: x' = fl32(x - ctr), n -> fp32, d0 -> fp32                      u (2 rr + |d0|)
//   remainders of the two-way splits of p, n'', d (relative 2^-23 = 2 u each)      u (4 rr + 2 |d0|)
//   operands below 2^-14 may be flushed by the matrix unit: <= 2^-14 per slot against the other operand's <= 2^14
//   (seven slots), in units of s:  7 / (2^14 px) = 7 R 2^-28                       Eflush = 8 R 2^-28
//   the dropped lo x lo products (each lo <= 2^-11 of its value)                    4 u rr
//   accumulation inside the instruction: not specified bit by bit; ASSUMED (as in dense_h16.h) at most one ulp of the
//   largest magnitude involved per product, 13 terms:                               26 u (rr + |d0|)
// |s16 - s*| <= u (36 rr + 29 |d0|) + Eflush, and a cell that passed level 1 has |d0| <= (rr + tout)(1 + 2^-18):
//   E16 = 1.01 u (65.1 rr + 29.1 T) + 8 R 2^-28 + 3e-12 X.    (cells.h's fp32 chain: 1.01 u (9.4 rr + 4.3 T).)
// The band is wider, the re-check is narrower: a lane whose hypothesis has a value in the band evaluates the exact fp64
// predicate for THAT observation (k_scan_pairs re-evaluates all 512 of the cell).
// Level 1, the counted shares and the output are k_scan_pairs' own: same survivor masks, same (cell, group, lane)
// enumeration, votes bit-identical (tests/test_gpu_allvotes.py::test_plane_level2_on_the_matrix_cores).
//
// STATUS (r04): exact, and NOT the default (`scan_pairs_mfma` 0).  Measured at 4096 x 10 M: scan phase 0.97 ms with two
// waves per SIMD, 1.04 ms with three (17 registers spilled), against 0.78 ms for the packed fp32 level 2.  Per 32 x 32
// tile the compiler emits ~60 issue slots (the 36 of the classification, 16 moves around the accumulator, waits behind
// the matrix instruction) at two waves per SIMD where k_scan_pairs runs six; the matrix unit aligns the products of one
// instruction with two guard bits (tools/h16_bench.hip, "align probe": 15 small terms beside a large one lose 3.8 ulp
// of it), so the band cannot be narrowed much below the 65 rr u taken here, and deferring the exact decisions to
// per-lane queues (below) changed nothing -- with the band switched off altogether (timing experiment, wrong votes) the
// scan phase is the same 0.97 ms: the loop is issue bound.  40 issue slots per matrix instruction = 20 per (hypothesis,
// cell) pair, plus ~7 per pair for level 1 and the split operands of a group, against k_scan_pairs' 37 -- a quarter
// fewer, at a third of its occupancy.
#pragma once
#include <type_traits>
#include <utility>

#include "cells.h"
#include "dense_h16.h"

namespace lsqr {

struct PairsH16Consts {
  float e_rr, e_t;  // E16 = rr * e_rr + e_t + R * e_r
  float e_r;
  float tdn, tup;   // fp32 neighbours of the exact threshold T
};
inline PairsH16Consts pairs_h16_consts(const ModelConsts &mc) {
  const double u = 5.9604644775390625e-08;
  PairsH16Consts k;
  k.e_rr = f32_up_host(1.01 * 65.1 * u * (1.0 + 1e-6));
  k.e_t = f32_up_host((1.01 * 29.1 * u * mc.thr + 3e-12 * mc.absmax) * (1.0 + 1e-6));
  k.e_r = f32_up_host(8.0 / 268435456.0 * (1.0 + 1e-6));
  k.tdn = f32_down_host(mc.thr);
  k.tup = f32_up_host(mc.thr);
  return k;
}

// one 32 x 32 tile: the matrix instruction, the votes of my column among the tile's 16 rows of my half, and the exact
// predicate for the values in the band.  Register i = observation row 8 (i / 4) + 4 half + i % 4; bit 15 - i of `bits`.
// Values in the band are not decided where they are found: a lane would wait a memory round trip (an observation and
// six parameters, ~1.5 us with two waves per SIMD) for each of them, and with a band six times the fp32 chain's one
// matrix instruction in ten finds one.  Every lane keeps its own short queue of (hypothesis, observation) in LDS; the
// queues are emptied by all lanes together when one is half full (pairs_h16_drain): one round trip for ~100 of them.
constexpr uint32_t kPairsQueue = 8;
__device__ __forceinline__ uint32_t pairs_h16_tile(const h16x8 &af, const h16x8 &bf, const float na1, const uint32_t band,
                                                   const int t, const bool partial, const size_t cell0, const int half,
                                                   const size_t ns, const double *__restrict__ sorted,
                                                   const double *__restrict__ hp, const ModelConsts &mc,
                                                   const uint32_t hid, uint2 *q_ent, uint32_t &qn) {
  typedef PlaneModel<3> M;
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const f32x2 na = {na1, na1};
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc, 0, 0, 0);
  uint32_t bits = 0, dmin = 0xFFFFFFFFu;
#pragma unroll
  for (int p = 0; p < 8; p++) {
    const f32x2 s = {acc[2 * p], acc[2 * p + 1]};
    const f32x2 d = __builtin_elementwise_fma(s, s, na);
    const u32x2 du = __builtin_bit_cast(u32x2, d);
    bits = __builtin_amdgcn_alignbit(bits, du.x, 31);
    bits = __builtin_amdgcn_alignbit(bits, du.y, 31);
    const uint32_t mn = du.x < du.y ? du.x : du.y;  // (halves first: cells.h on hipcc 7.2 and packed results)
    dmin = mn < dmin ? mn : dmin;
  }
  if (partial) {  // rows past the end of the upload never count (their fragments are zeros)
    uint32_t keep = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const size_t gi = cell0 + t * 32 + 8 * (i / 4) + 4 * half + (i % 4);
      keep |= gi < ns ? 1u << (15 - i) : 0u;
    }
    bits &= keep;
  }
  uint32_t c = (uint32_t)__builtin_popcount(bits);
  if (dmin <= band) {  // a value of my hypothesis in the band: the exact predicate on that observation
    uint32_t am = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const float s1 = acc[i];
      const float d1 = __builtin_fmaf(s1, s1, na1);
      am |= __builtin_bit_cast(uint32_t, d1) <= band ? 1u << i : 0u;
    }
    while (am) {  // (a loop, not 16 branches: sixteen tiles of sixteen nested branches spilled a thousand registers)
      const int i = __builtin_ctz(am);
      am &= am - 1;
      const size_t gi = cell0 + t * 32 + 8 * (i >> 2) + 4 * half + (i & 3);
      if (gi < ns) {
        if (qn < kPairsQueue) {  // decided later, together with the other lanes' (pairs_h16_drain)
          q_ent[qn++] = (uint2){hid, (uint32_t)gi};
        } else {
          double r[3];
#pragma unroll
          for (int d = 0; d < 3; d++) r[d] = sorted[gi * 3 + d];
          c += M::agree(hp, r, mc) ? 1u : 0u;
        }
      }
    }
  }
  return c;
}

template <int BS, int WPE>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_scan_pairs_h16(
    const double *__restrict__ sorted, size_t ns, const CellBox *__restrict__ boxes, uint32_t ncells,
    const double *__restrict__ sp, const float *__restrict__ rows, const float *__restrict__ spf, uint32_t H,
    ModelConsts mc, CellConsts cc, PairsH16Consts kc, uint32_t *__restrict__ vpart, uint32_t vstride,
    const uint32_t *__restrict__ h_dev, const uint8_t *__restrict__ cnt, uint32_t gstride, const uint32_t *__restrict__ cost,
    const uint32_t *__restrict__ csum, uint32_t nchunks, uint32_t h_off) {
  typedef PlaneCell<3> CM;
  typedef typename CM::M M;
  constexpr int NB = CM::NB, SPD = M::SP, ROW = CM::ROW, NR4 = ROW / 4, D = 3, CP = 512;
  if (h_dev) {
    const uint32_t hd = *h_dev > h_off ? *h_dev - h_off : 0u;  // hypotheses [h_off, h_off + H) of the selection
    H = hd < H ? hd : H;
  }
  extern __shared__ uint32_t s_cnt[];
  for (uint32_t h = threadIdx.x; h < H; h += BS) s_cnt[h] = 0;
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // per wave: the tile being filled -- B fragments [2 halves][32 columns] of 16 bytes, then -a, band, hypothesis
  uint4 *t_frag = (uint4 *)(s_cnt + ((H + 3) & ~3u)) + (size_t)wv * 88;
  float *t_na = (float *)(t_frag + 64);
  uint32_t *t_band = (uint32_t *)(t_na + 32);
  uint32_t *t_hid = t_band + 32;
  // my lane's queue of undecided (hypothesis, observation) pairs, behind the four tile buffers
  uint2 *q_ent = (uint2 *)((uint4 *)(s_cnt + ((H + 3) & ~3u)) + (size_t)(BS / 64) * 88) +
                 ((size_t)wv * 64 + lane) * kPairsQueue;
  uint32_t qn = 0;
  auto drain = [&]() {
    for (uint32_t k = 0; __ballot(k < qn); k++) {
      if (k < qn) {
        const uint2 e = q_ent[k];
        double r[3];
#pragma unroll
        for (int d = 0; d < 3; d++) r[d] = sorted[(size_t)e.y * 3 + d];
        if (M::agree(sp + (size_t)e.x * SPD, r, mc)) atomicAdd(&s_cnt[e.x], 1u);
      }
    }
    qn = 0;
  };
  t_frag[lane] = (uint4){0u, 0u, 0u, 0u};
  if (lane < 32) t_na[lane] = __builtin_inff(), t_band[lane] = 0u, t_hid[lane] = 0u;
  __syncthreads();
  const uint32_t W = gridDim.x * (BS / 64);
  const uint32_t wid = blockIdx.x * (BS / 64) + (uint32_t)wv;

  // ---- my share of the pairs: [t0, t0 + budget) of C (k_scan_pairs, verbatim)
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
    const uint32_t cb = chunk * kChunkCells + 2 * lane;
    const uint32_t v0 = cb < ncells ? cost[cb] : 0u, v1 = cb + 1 < ncells ? cost[cb + 1] : 0u;
    const uint32_t inc = wave_incl_scan(v0 + v1, lane);
    const uint32_t r = (uint32_t)(t0 - base);
    const unsigned long long hit = __ballot(inc > r);
    const int l = hit ? __builtin_ctzll(hit) : 63;
    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(inc - v0 - v1), l);
    const uint32_t c0v = (uint32_t)__builtin_amdgcn_readlane((int)v0, l);
    const bool second = r - before >= c0v;
    cell = chunk * kChunkCells + 2 * l + (second ? 1u : 0u);
    skip = r - before - (second ? c0v : 0u);
  }
  const uint32_t G = (H + 63) / 64;
  auto pad = [&](uint32_t P) {
    if (skip >= P) {
      skip -= P;
    } else {
      const uint32_t take = P - skip < budget ? P - skip : budget;
      budget -= take;
      skip = 0;
    }
  };
  auto load_rows = [&](uint32_t g, float4(&r)[NR4]) {
    const uint32_t h = g * 64 + lane;
    const float4 *r4 = (const float4 *)(rows + (size_t)(h < H ? h : 0) * ROW);
#pragma unroll
    for (int k = 0; k < NR4; k++) r[k] = r4[k];
  };
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  while (budget && cell < ncells) {
    {  // jump over cells nothing survives in
      const uint32_t v = cell + lane < ncells ? cost[cell + lane] : 0u;
      const unsigned long long nz = __ballot(v != 0);
      if (!nz) {
        cell += 64;
        continue;
      }
      cell += (uint32_t)__builtin_ctzll(nz);
    }
    pad(kCellPad);
    const uint32_t gc = (uint32_t)lane < G ? (uint32_t)cnt[(size_t)cell * gstride + lane] : 0u;
    unsigned long long gm = __ballot(gc != 0);
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
    // the cell's scale: px = 2^14 / R, R the power of two above the largest half extent (exact scalings throughout)
    const float hmax = __builtin_fmaxf(bx.h[0], __builtin_fmaxf(bx.h[1], bx.h[2]));
    int ex = 0;
    (void)frexpf(hmax > 0.0f ? hmax : 1.0f, &ex);                  // hmax = m 2^ex, m in [0.5, 1)  =>  hmax < 2^ex
    const float Rp2 = __builtin_ldexpf(1.0f, ex), px = __builtin_ldexpf(1.0f, 14 - ex);
    const float sc = px * 16384.0f;                                // s'' = s * sc
    const bool partial = (size_t)cell * CP + CP > ns;              // the upload's last cell (wave-uniform)
    // the observations as A fragments: 16 tiles of 32, lane = observation l % 32, slots by half (header)
    h16x8 af[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
      const size_t i = (size_t)cell * CP + t * 32 + col;
      const double *p = sorted + (i < ns ? i : 0) * D;
      h16x8 f;
      _Float16 hi[3], lo[3];
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const float xs = i < ns ? (float)(p[d] - ctr[d]) * px : 0.0f;
        hi[d] = (_Float16)xs;
        lo[d] = (_Float16)(xs - (float)hi[d]);
      }
      const _Float16 one = (_Float16)16384.0f, zero = (_Float16)0.0f;
      if (half == 0)
        f = (h16x8){hi[0], hi[1], hi[2], one, hi[0], hi[1], hi[2], one};
      else
        f = (h16x8){lo[0], lo[1], lo[2], zero, zero, zero, zero, zero};
      af[t] = f;
    }
    uint32_t fill = 0;  // entries of the tile being filled (wave-uniform)
    // 32 (or, at the end of the cell, `n`) columns against the 512 observations
    auto flush = [&](uint32_t n) __attribute__((always_inline)) {
      const h16x8 bf = __builtin_bit_cast(h16x8, t_frag[half * 32 + col]);
      const bool live = (uint32_t)col < n;
      const float na1 = live ? t_na[col] : __builtin_inff();
      const uint32_t band = live ? t_band[col] : 0u;
      const uint32_t hid = t_hid[col];
      uint32_t c = 0;
      // (sixteen explicit calls: `#pragma unroll` left the 16 tiles in a loop of 8 and the fragments, indexed by a
      // variable, in scratch memory)
      // (a scheduling barrier behind every tile: left alone the scheduler issues the 16 matrix instructions first and
      // keeps 256 accumulator registers alive -- 777 spilled)
#define LSQR_TILE(T)                                                                                                  \
  c += pairs_h16_tile(af[T], bf, na1, band, T, partial, (size_t)cell * 512, half, ns, sorted, sp + (size_t)hid * SPD, mc, hid, q_ent, qn); \
  __builtin_amdgcn_sched_barrier(0)
      LSQR_TILE(0); LSQR_TILE(1); LSQR_TILE(2); LSQR_TILE(3); LSQR_TILE(4); LSQR_TILE(5); LSQR_TILE(6); LSQR_TILE(7);
      LSQR_TILE(8); LSQR_TILE(9); LSQR_TILE(10); LSQR_TILE(11); LSQR_TILE(12); LSQR_TILE(13); LSQR_TILE(14); LSQR_TILE(15);
#undef LSQR_TILE
      c += __shfl_xor(c, 32);  // the two halves hold different rows of the same column
      if (half == 0 && live && c) atomicAdd(&s_cnt[hid], c);
      if (__ballot(qn >= kPairsQueue / 2)) drain();
    };
    float4 nxt[NR4];
    load_rows((uint32_t)__builtin_ctzll(gm), nxt);
    // one loop for the groups AND the end of the cell (a last round that only empties the tile): flush() has a single
    // call site -- with two the fragments of the cell went to scratch memory
    for (;;) {
      const bool more = gm && budget;  // wave-uniform
      uint32_t total = fill, pos = 0, band = 0, h = 0;
      bool mine = false;
      float a = 0.0f;
      h16x8 f0, f1;
#pragma unroll
      for (int i = 0; i < 8; i++) f0[i] = (_Float16)0.0f, f1[i] = (_Float16)0.0f;
      if (more) {
        const int g = __builtin_ctzll(gm);
        gm &= gm - 1;
        const uint32_t cg = (uint32_t)__builtin_amdgcn_readlane((int)gc, g);
        float row[ROW], row2[4];
#pragma unroll
        for (int k = 0; k < NR4; k++)
          row[4 * k] = nxt[k].x, row[4 * k + 1] = nxt[k].y, row[4 * k + 2] = nxt[k].z, row[4 * k + 3] = nxt[k].w;
        if (gm) load_rows((uint32_t)__builtin_ctzll(gm), nxt);  // the next group's rows meanwhile
        const uint32_t lo = skip, hi = cg < skip + budget ? cg : skip + budget;  // skip < cg here
        budget -= hi - lo;
        skip = 0;
        const uint32_t jlo = (lo > kGroupPad ? lo : kGroupPad) - kGroupPad,
                       jhi = (hi > kGroupPad ? hi : kGroupPad) - kGroupPad;
        if (jhi <= jlo) continue;
        const uint32_t h0 = (uint32_t)g * 64;
        h = h0 + lane;
        typename CM::Hyp hy;
        CM::load(row, row2, h < H, cc, hy);
        float bc[NB];
        const bool l1 = CM::level1(hy, bx, ctr, cc, bc);
        unsigned long long surv = __ballot(l1);  // == the counting pass: cg - pad bits
        for (uint32_t k = 0; k < jlo; k++) surv &= surv - 1;                  // the first jlo are not mine
        if (jhi < cg - kGroupPad) {  // the tail belongs to the next wave: keep the lowest jhi - jlo bits
          unsigned long long keep = 0, m = surv;
          for (uint32_t k = jlo; k < jhi; k++) {
            keep |= m & (0ull - m);
            m &= m - 1;
          }
          surv = keep;
        }
        if (!surv) continue;
        // my column: thresholds of the fp16 evaluation (header) and the split operands, in the tile's scale
        float rr = 0.0f;
#pragma unroll
        for (int i = D - 1; i >= 0; i--) rr = __builtin_fmaf(__builtin_fabsf(hy.nf[i]), bx.h[i], rr);
        // level 1 switched the filter off (|n_i| > 1, NaN), or d0 px does not fit an fp16 number (a threshold far above
        // the cell's size): zero operands and a band that holds every finite value -- the exact predicate decides
        const bool off = !(bc[NB - 1] < __builtin_inff()) || !(__builtin_fabsf(bc[3] * px) < 60000.0f);
        const float E = __builtin_fmaf(rr, kc.e_rr, __builtin_fmaf(Rp2, kc.e_r, kc.e_t));
        const float tin = (kc.tdn - E) * 0.9999997f * sc, tout = (kc.tup + E) * 1.0000003f * sc;
        a = tin > 0.0f ? (tin * tin) * 0.9999998f : 0.0f;
        const float cq = (tout * tout) * 1.0000002f;
        band = __builtin_bit_cast(uint32_t, cq - a);
        if (off || !(cq < __builtin_inff())) a = 0.0f, band = 0x7F7FFFFFu;  // everything finite is ambiguous
        _Float16 v1[4], v2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float v = k < 3 ? hy.nf[k] * 16384.0f : bc[3] * px;
          if (off) v = 0.0f;
          v1[k] = (_Float16)v;
          v2[k] = (_Float16)(v - (float)v1[k]);
        }
        f0 = (h16x8){v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
        const _Float16 zero = (_Float16)0.0f;
        f1 = (h16x8){v1[0], v1[1], v1[2], zero, zero, zero, zero, zero};
        mine = (surv >> lane) & 1ull;
        const uint32_t below =
            __builtin_amdgcn_mbcnt_hi((uint32_t)(surv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)surv, 0u));
        pos = fill + below;
        total = fill + (uint32_t)__builtin_popcountll(surv);
      }
      uint32_t r0 = 0;
      for (;;) {
        if (mine && pos >= r0 && pos < r0 + 32) {
          const uint32_t sl = pos - r0;
          t_frag[sl] = __builtin_bit_cast(uint4, f0);
          t_frag[32 + sl] = __builtin_bit_cast(uint4, f1);
          t_na[sl] = -a;
          t_band[sl] = band;
          t_hid[sl] = h;
        }
        uint32_t ncols;
        if (total >= r0 + 32)
          ncols = 32;
        else if (!more && total > r0)
          ncols = total - r0;  // the end of the cell: what is left in the tile
        else
          break;
        flush(ncols);
        r0 += 32;
        if (!more) total = r0;
      }
      fill = total - r0;
      if (!more) break;
    }
    cell++;
  }
  drain();
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < H; i += BS) vpart[(size_t)blockIdx.x * vstride + i] = s_cnt[i];
}

}  // namespace lsqr
