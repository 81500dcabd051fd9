// axis.h -- axis-sorted cells: near-parallel hypotheses counted by rank instead of by evaluation (plane, 3-D).
//
// The hypotheses a RANSAC batch has to count are dominated by the near-model ones (all-inlier samples): they survive
// the box test in every cell the structure passes through -- half of the (hypothesis, cell) pairs of a full count,
// nearly all of the bounded scan's second pass -- and there level 2 evaluates every observation of the cell although
// the outcome is almost the same for all of them.  The cells of such a structure are thin slabs; so, ONCE PER UPLOAD
// (data only, like the Morton order and the boxes):
//   * every cell gets an axis e -- the direction of least spread of its own observations, or of its group of eight
//     consecutive cells when that serves the cell as well (more observations: a quieter estimate; k_cell_axes);
//   * the observations of a cell are re-ordered by t = e . (x - ctr) (a bitonic sort inside the wave; the order inside
//     a cell is irrelevant to every other kernel) and the sorted t are kept as fp32 T[] (k_cell_sort).
// For a hypothesis (n, a) and a cell:  s(x) = n . (x - a) = alpha t + d0 + m . (x - ctr),  alpha = n.e / e.e,
// m = n - alpha e,  d0 = n . (ctr - a)  -- an identity for ANY e.  When n is nearly parallel to e, |m . (x - ctr)| <=
// rho = sum |m_i| h_i is small, and with E = rho + (rounding, below)
//      |alpha T + d0| <  thr - E   =>  the observation agrees,         |alpha T + d0| >= thr + E   =>  it does not:
// two intervals of T, i.e. four binary searches in the cell's sorted T[], give per cell how many observations CERTAINLY
// agree (r2 - r1) and how many CAN agree (r3 - r0) without evaluating one of them.  Summed over the cells that is a
// LOWER and an UPPER bound on the hypothesis' votes, within a few per cent of each other for a near-model hypothesis --
// where the bounded scan's box-population bound is 2.7 x the votes (every cell the structure passes through counts
// whole) and cannot tell one near-model hypothesis from another.  k_bound_axis computes the two bounds for the bounded
// scan's candidates; with them only the hypotheses whose upper bound exceeds the best LOWER bound before them are
// counted exactly (plane, 10 M points, 50 % outliers: 169 of 4096 instead of 511, and no pilots; sparse uploads whose
// cells are not flat keep the pilots: cells.h, k_pick_pilots).
// (Tried first, r03: settling the pairs themselves by rank and evaluating only the shells, lane = hypothesis -- exact,
// 3.5 M of 5.1 M near-model pairs settled, but the serial shell loops of 64 lanes run at the length of the longest and
// at the latency of their LDS reads: 0.86 ms against the 0.31 ms of level-2 work they saved.  Removed.)
//
// Rounding.  ctr and e are the stored floats taken as exact reals; t is evaluated in fp64 and rounded to fp32:
// |T - t| <= 2^-24 |t| + 1e-13 X (X = max |coordinate|).  alpha, m, d0 are evaluated in fp64 (errors <= 1e-13 X in
// the identity above, m being formed from the rounded alpha so that the identity holds for it).  The reference's
// decision is |s_ref| < thr with |s_ref - s| < 1e-13 X (cells.h).  E = rho (1 + 2^-20) + |alpha| 2^-23 Tmax + 4e-12 X
// covers all of it (k_bound_axis evaluates the same quantities in fp32 and prices that as well: see there).  The four
// boundaries (+-thr +- E - d0) / alpha are moved outwards (certain sets shrink) by the error of their own evaluation
// before the search, and the searches compare fp32 with fp32 exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cells.h"

namespace lsqr {

struct CellAxis {  // 16 B: one s_load_dwordx4
  float e[3];      // the axis (not normalised exactly: any vector serves the identity)
  float tmax;      // max |T| over the cell's observations
};
constexpr int kAxisGroup = 8;      // cells per group of k_cell_axes

// ---- build 1: per-cell moments about the cell centre: {count, sum x (3), sum x x^T upper (6)} ------------------------
template <int NPT>
__global__ __launch_bounds__(256) void k_cell_moments(const double *__restrict__ sorted, size_t ns, uint32_t ncells,
                                                      const CellBox *__restrict__ boxes, double *__restrict__ mom) {
  const uint32_t cell = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cell >= ncells) return;
  const int lane = threadIdx.x & 63;
  constexpr uint32_t CP = 64 * NPT;
  const CellBox bx = boxes[cell];
  double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < NPT; k++) {
    const size_t i = (size_t)cell * CP + k * 64 + lane;
    if (i < ns) {
      const double x = sorted[i * 3] - (double)bx.c[0], y = sorted[i * 3 + 1] - (double)bx.c[1],
                   z = sorted[i * 3 + 2] - (double)bx.c[2];
      s[0] += 1.0, s[1] += x, s[2] += y, s[3] += z;
      s[4] += x * x, s[5] += x * y, s[6] += x * z, s[7] += y * y, s[8] += y * z, s[9] += z * z;
    }
  }
#pragma unroll
  for (int q = 0; q < 10; q++)
    for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o);
  if (lane < 10) {
    double v = s[0];
#pragma unroll
    for (int q = 1; q < 10; q++) v = lane == q ? s[q] : v;
    mom[(size_t)cell * 10 + lane] = v;
  }
}

// smallest-eigenvalue direction of the covariance behind {n, sum x, sum x x^T}; returns its variance
__device__ inline double axis_of(const double *s, double *e) {
  const double n = s[0] > 0 ? s[0] : 1.0;
  const double mx = s[1] / n, my = s[2] / n, mz = s[3] / n;
  double a[9], w[3], v[9];
  a[0] = s[4] / n - mx * mx, a[1] = s[5] / n - mx * my, a[2] = s[6] / n - mx * mz;
  a[4] = s[7] / n - my * my, a[5] = s[8] / n - my * mz, a[8] = s[9] / n - mz * mz;
  a[3] = a[1], a[6] = a[2], a[7] = a[5];
  sym_eig(3, a, w, v);
  e[0] = v[0], e[1] = v[3], e[2] = v[6];  // column 0: the smallest eigenvalue
  return w[0];
}

// ---- build 2: one thread per group of kAxisGroup cells: the group's axis, each cell's own, the choice ----------------
__global__ __launch_bounds__(256) void k_cell_axes(const double *__restrict__ mom, const CellBox *__restrict__ boxes,
                                                   uint32_t ncells, CellAxis *__restrict__ axis) {
  const uint32_t g = blockIdx.x * 256 + threadIdx.x;
  const uint32_t c0 = g * kAxisGroup;
  if (c0 >= ncells) return;
  const uint32_t c1 = c0 + kAxisGroup < ncells ? c0 + kAxisGroup : ncells;
  // the group's moments about the first cell's centre
  const CellBox b0 = boxes[c0];
  double G[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t c = c0; c < c1; c++) {
    const CellBox b = boxes[c];
    const double *s = mom + (size_t)c * 10;
    const double dx = (double)b.c[0] - (double)b0.c[0], dy = (double)b.c[1] - (double)b0.c[1],
                 dz = (double)b.c[2] - (double)b0.c[2];
    const double n = s[0];
    G[0] += n;
    G[1] += s[1] + n * dx, G[2] += s[2] + n * dy, G[3] += s[3] + n * dz;
    G[4] += s[4] + 2 * dx * s[1] + n * dx * dx;
    G[5] += s[5] + dx * s[2] + dy * s[1] + n * dx * dy;
    G[6] += s[6] + dx * s[3] + dz * s[1] + n * dx * dz;
    G[7] += s[7] + 2 * dy * s[2] + n * dy * dy;
    G[8] += s[8] + dy * s[3] + dz * s[2] + n * dy * dz;
    G[9] += s[9] + 2 * dz * s[3] + n * dz * dz;
  }
  double eg[3];
  axis_of(G, eg);
  for (uint32_t c = c0; c < c1; c++) {
    const double *s = mom + (size_t)c * 10;
    double ec[3];
    const double vc = axis_of(s, ec);
    // the cell's variance along the group's axis: if the group's axis serves the cell (almost) as well as its own,
    // take it (eight times the observations behind it)
    const double n = s[0] > 0 ? s[0] : 1.0, mx = s[1] / n, my = s[2] / n, mz = s[3] / n;
    const double cxx = s[4] / n - mx * mx, cxy = s[5] / n - mx * my, cxz = s[6] / n - mx * mz, cyy = s[7] / n - my * my,
                 cyz = s[8] / n - my * mz, czz = s[9] / n - mz * mz;
    const double vg = eg[0] * (cxx * eg[0] + 2 * cxy * eg[1] + 2 * cxz * eg[2]) + eg[1] * (cyy * eg[1] + 2 * cyz * eg[2]) +
                      eg[2] * czz * eg[2];
    const bool use_group = vg <= 1.5 * vc + 1e-300 && vg == vg;
    CellAxis a;
    for (int d = 0; d < 3; d++) {
      const double v = use_group ? eg[d] : ec[d];
      a.e[d] = (v == v) ? (float)v : (d == 0 ? 1.0f : 0.0f);
    }
    if (!(a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2] > 0.25f)) a.e[0] = 1.0f, a.e[1] = 0.0f, a.e[2] = 0.0f;
    a.tmax = 0.0f;  // (set by k_cell_sort)
    axis[c] = a;
  }
}

// ---- build 3: one wave per cell: t along the axis, bitonic sort of (t, index) in LDS, T[] and the cell's records
// in that order ----------------------------------------------------------------------------------------------------
template <int NPT>
__global__ __launch_bounds__(256) void k_cell_sort(double *__restrict__ sorted, size_t ns, uint32_t ncells,
                                                   const CellBox *__restrict__ boxes, CellAxis *__restrict__ axis,
                                                   float *__restrict__ cellT) {
  constexpr uint32_t CP = 64 * NPT;
  __shared__ unsigned long long s_key[4][CP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t cell_raw = blockIdx.x * 4 + wave;
  const bool live = cell_raw < ncells;
  const uint32_t cell = live ? cell_raw : ncells - 1;  // (waves past the end walk the last cell and store nothing)
  unsigned long long *s = s_key[wave];
  const CellBox bx = boxes[cell];
  const CellAxis ax = axis[cell];
  const double e0 = (double)ax.e[0], e1 = (double)ax.e[1], e2 = (double)ax.e[2];
  float tmax = 0.0f;
  for (int k = 0; k < NPT; k++) {
    const uint32_t j = k * 64 + lane;
    const size_t i = (size_t)cell * CP + j;
    uint32_t key = 0xFFFFFFFFu;  // padding sorts to the end
    if (i < ns) {
      const double t = e0 * (sorted[i * 3] - (double)bx.c[0]) + e1 * (sorted[i * 3 + 1] - (double)bx.c[1]) +
                       e2 * (sorted[i * 3 + 2] - (double)bx.c[2]);
      float tf = (float)t;
      if (!(tf == tf)) tf = 3.0e38f;  // (cannot happen on an indexed upload: finite records only)
      tmax = fmaxf(tmax, fabsf(tf));
      uint32_t u = __builtin_bit_cast(uint32_t, tf);
      u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // order-preserving map float -> uint32
      key = u < 0xFFFFFFFEu ? u : 0xFFFFFFFEu;
    }
    s[j] = ((unsigned long long)key << 32) | j;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= CP; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t t = lane; t < CP / 2; t += 64) {
        const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const unsigned long long a = s[i], b = s[i + j];
        const bool up = (i & k) == 0;
        if ((a > b) == up) {
          s[i] = b;
          s[i + j] = a;
        }
      }
      __syncthreads();
    }
  for (int o = 32; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o));
  // every record of the cell is read before any is written back (the stores depend on the loads, and all loads are
  // issued first: the order inside a cell changes in place)
  double px[NPT], py[NPT], pz[NPT];
  float tv[NPT];
  bool ok[NPT];
#pragma unroll
  for (int k = 0; k < NPT; k++) {
    const unsigned long long v = s[k * 64 + lane];
    const uint32_t src = (uint32_t)(v & 0xFFFFFFFFu), key = (uint32_t)(v >> 32);
    ok[k] = key != 0xFFFFFFFFu;
    const size_t i = (size_t)cell * CP + (ok[k] ? src : 0);
    const size_t ic = i < ns ? i : 0;
    px[k] = sorted[ic * 3], py[k] = sorted[ic * 3 + 1], pz[k] = sorted[ic * 3 + 2];
    uint32_t u = key;
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;  // inverse map
    tv[k] = ok[k] ? __builtin_bit_cast(float, u) : __builtin_inff();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!live) return;
#pragma unroll
  for (int k = 0; k < NPT; k++) {
    const size_t i = (size_t)cell * CP + k * 64 + lane;
    cellT[i] = tv[k];
    if (ok[k] && i < ns) sorted[i * 3] = px[k], sorted[i * 3 + 1] = py[k], sorted[i * 3 + 2] = pz[k];
  }
  if (lane == 0) axis[cell].tmax = tmax;
}

// ---- vote bounds by rank ------------------------------------------------------------------------------------------
// For a (compacted) batch of candidate hypotheses: per hypothesis an UPPER and a LOWER bound on its votes, cell by
// cell -- rank counts where the pair is nearly parallel (header comment; no observation is evaluated), the cell's
// population / zero elsewhere.  lane = hypothesis, blockIdx.y * 8 + wave = group of 64, blockIdx.x = a run of cells;
// eight cells per round are staged in LDS (boxes, axes and their T[]: 2 KB a cell, read once per workgroup -- 40 MB per
// 512 candidates at 10 M points) and the four searches of a pair step together over LDS (11 dependent rounds of four
// reads; from global memory the same searches took 278 us per batch, from LDS 93 us, in 8-cell runs 71 us).
// All per-pair quantities in fp32 here (a bound only has to be conservative): with u = 2^-24,
//   alpha32 = fl(n32 . e) / fl(e . e):           |alpha32 - alpha| <= 8u (|n| <= 1.0000001, |e| ~ 1)
//   m32 = n32 - alpha32 e,  rho32 = sum |m32_i| h_i:   the identity s = alpha32 t + d0 + m . (x - ctr) holds for
//         m = n - alpha32 e exactly; |m - m32| <= 3u per component (n32 rounding, product, difference)
//   d0 = fl32 of level 1's fp64 d:               |d0 - d| <= u |d|,  |d| <= rr + tout (a surviving cell)
//   E = 1.001 (rho32 + 3.1 u R1) + |alpha32| 2u Tmax + u |d0| (1 + 2u) + 4e-12 X,   R1 = h_0 + h_1 + h_2
// and every boundary is moved outwards by its own fp32 evaluation error (see the code).
template <int PP>
__global__ __launch_bounds__(512) void k_bound_axis(const CellBox *__restrict__ boxes, const CellAxis *__restrict__ axis,
                                                    const float *__restrict__ cellT, size_t ns, uint32_t ncells,
                                                    const float *__restrict__ rows, uint32_t H,
                                                    const uint32_t *__restrict__ h_dev, CellConsts cc, float thr_up,
                                                    float thr_dn, float xabs, uint32_t cells_per_block,
                                                    uint32_t *__restrict__ ub, uint32_t *__restrict__ lb) {
  typedef PlaneCell<3> CM;
  constexpr uint32_t CP = 128 * PP;
  constexpr int ROW = CM::ROW, NR4 = ROW / 4;
  const int lane = threadIdx.x & 63;
  if (h_dev) {
    const uint32_t hd = *h_dev;
    H = hd < H ? hd : H;
  }
  const uint32_t grp = blockIdx.y * 8 + (threadIdx.x >> 6);
  const uint32_t h = grp * 64 + lane;
  if (blockIdx.y * 512 >= H) return;   // workgroup-uniform
  const bool active = h - lane < H;    // wave-uniform
  float row[ROW];
  {
    const float4 *r4 = (const float4 *)(rows + (size_t)(h < H ? h : 0) * ROW);
#pragma unroll
    for (int k = 0; k < NR4; k++) {
      const float4 v = r4[k];
      row[4 * k] = v.x, row[4 * k + 1] = v.y, row[4 * k + 2] = v.z, row[4 * k + 3] = v.w;
    }
  }
  typename CM::Hyp hy;
  CM::load(row, row, h < H, cc, hy);
  const uint32_t c0 = blockIdx.x * cells_per_block;
  const uint32_t c1 = c0 + cells_per_block < ncells ? c0 + cells_per_block : ncells;
  uint32_t u_acc = 0, l_acc = 0;
  constexpr uint32_t BCH = 8;             // cells per round: their boxes, axes and T[] staged in LDS
  __shared__ CellBox s_box[BCH];
  __shared__ CellAxis s_ax[BCH];
  __shared__ float4 s_T[BCH * CP / 4];
  for (uint32_t cb = c0; cb < c1; cb += BCH) {
    const uint32_t n = c1 - cb < BCH ? c1 - cb : BCH;
    __syncthreads();
    if (threadIdx.x < 2 * n) ((float4 *)s_box)[threadIdx.x] = ((const float4 *)(boxes + cb))[threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + n) ((float4 *)s_ax)[threadIdx.x - 64] = ((const float4 *)(axis + cb))[threadIdx.x - 64];
    {
      const float4 *src = (const float4 *)(cellT + (size_t)cb * CP);
      for (uint32_t k = threadIdx.x; k < n * (CP / 4); k += 512) s_T[k] = src[k];
    }
    __syncthreads();
    if (!active) continue;
    // Per cell: level 1 and the four boundaries (prep), then the searches.  The searches are a chain of 11 dependent
    // LDS reads; TWO cells are searched together (eight independent reads per round) -- with one the kernel ran at
    // the latency of that chain (93 us per batch of ~500 candidates; r03).
    struct Prep {
      bool s, pre;
      float ol, oh, il, ih;
      uint32_t pop;
      const float *T;
    };
    auto prep = [&](uint32_t i, Prep &p) -> bool {  // false: no lane of the wave survives in the cell
      const CellBox bx = s_box[i];
      const CellAxis ax = s_ax[i];
      const double ctr[3] = {(double)bx.c[0], (double)bx.c[1], (double)bx.c[2]};
      float bc[CM::NB];
      p.s = CM::level1(hy, bx, ctr, cc, bc);
      p.pre = false;
      if (!__ballot(p.s)) return false;  // wave-uniform
      const size_t first = (size_t)(cb + i) * CP;
      p.pop = first + CP <= ns ? CP : (uint32_t)(ns - first);
      p.T = (const float *)s_T + i * CP;
      const float iee = __builtin_amdgcn_rcpf(ax.e[0] * ax.e[0] + ax.e[1] * ax.e[1] + ax.e[2] * ax.e[2]);
      const float al = (bc[0] * ax.e[0] + bc[1] * ax.e[1] + bc[2] * ax.e[2]) * iee;  // bc[0..2] = n32
      const float m0 = bc[0] - al * ax.e[0], m1 = bc[1] - al * ax.e[1], m2 = bc[2] - al * ax.e[2];
      const float rho = __builtin_fabsf(m0) * bx.h[0] + __builtin_fabsf(m1) * bx.h[1] + __builtin_fabsf(m2) * bx.h[2];
      const float d0 = bc[3];
      const float E = (1.001f * (rho + 1.9e-7f * (bx.h[0] + bx.h[1] + bx.h[2])) + __builtin_fabsf(al) * 1.2e-7f * ax.tmax +
                       6.1e-8f * __builtin_fabsf(d0) + 4e-12f * xabs) * 1.00001f + 1e-30f;
      p.pre = p.s && __builtin_fabsf(al) >= 0.25f && E < 0.5f * thr_dn && bc[5] < __builtin_inff();
      const float ia = __builtin_amdgcn_rcpf(al);
      const float b_ol = (-thr_up - E - d0) * ia, b_oh = (thr_up + E - d0) * ia;
      const float b_il = (-thr_dn + E - d0) * ia, b_ih = (thr_dn - E - d0) * ia;
      float ol = fminf(b_ol, b_oh), oh = fmaxf(b_ol, b_oh), il = fminf(b_il, b_ih), ih = fmaxf(b_il, b_ih);
      // fp32 evaluation of a boundary: the numerator (three terms of magnitude <= |d0| + thr + E) is off by at most
      // 3u of that magnitude, the reciprocal and the product by 3u relative: move every boundary outwards by
      // mar = 4e-7 (|d0| + thr + E) |1 / alpha| + 2e-6 |b| (certain sets shrink)
      const float mag = 4e-7f * (__builtin_fabsf(d0) + thr_up + E) * __builtin_fabsf(ia);
      p.ol = ol - (mag + 2e-6f * __builtin_fabsf(ol));
      p.oh = oh + (mag + 2e-6f * __builtin_fabsf(oh));
      p.il = il + (mag + 2e-6f * __builtin_fabsf(il));
      p.ih = ih - (mag + 2e-6f * __builtin_fabsf(ih));
      return true;
    };
    // r0 = #{T < ol}, r3 = #{T <= oh}, r1 = #{T <= il}, r2 = #{T < ih} for the lanes with `pre`, both cells' four
    // searches stepping together; then the cell's contribution to the two bounds
    auto search2 = [&](const Prep &A, const Prep &B, bool haveB) {
      uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
      const bool pb = haveB && B.pre;
      if (A.pre || pb) {
        const float *TA = A.T, *TB = haveB ? B.T : A.T;
        // (a lane without `pre` in one of the cells searches with boundaries that keep its ranks at 0)
        const float ninf = -__builtin_inff();
        const float aol = A.pre ? A.ol : ninf, aoh = A.pre ? A.oh : ninf, ail = A.pre ? A.il : ninf,
                    aih = A.pre ? A.ih : ninf;
        const float bol = pb ? B.ol : ninf, boh = pb ? B.oh : ninf, bil = pb ? B.il : ninf, bih = pb ? B.ih : ninf;
#pragma unroll
        for (uint32_t step = CP / 2; step > 0; step >>= 1) {
          const float ta0 = TA[a0 + step - 1], ta3 = TA[a3 + step - 1], ta1 = TA[a1 + step - 1], ta2 = TA[a2 + step - 1];
          const float tb0 = TB[b0 + step - 1], tb3 = TB[b3 + step - 1], tb1 = TB[b1 + step - 1], tb2 = TB[b2 + step - 1];
          a0 += ta0 < aol ? step : 0u;
          a3 += ta3 <= aoh ? step : 0u;
          a1 += ta1 <= ail ? step : 0u;
          a2 += ta2 < aih ? step : 0u;
          b0 += tb0 < bol ? step : 0u;
          b3 += tb3 <= boh ? step : 0u;
          b1 += tb1 <= bil ? step : 0u;
          b2 += tb2 < bih ? step : 0u;
        }
        {  // (the branchless search stops one short of CP)
          const float ta0 = TA[a0], ta3 = TA[a3], ta1 = TA[a1], ta2 = TA[a2];
          const float tb0 = TB[b0], tb3 = TB[b3], tb1 = TB[b1], tb2 = TB[b2];
          a0 += ta0 < aol ? 1u : 0u;
          a3 += ta3 <= aoh ? 1u : 0u;
          a1 += ta1 <= ail ? 1u : 0u;
          a2 += ta2 < aih ? 1u : 0u;
          b0 += tb0 < bol ? 1u : 0u;
          b3 += tb3 <= boh ? 1u : 0u;
          b1 += tb1 <= bil ? 1u : 0u;
          b2 += tb2 < bih ? 1u : 0u;
        }
      }
      auto fold = [&](const Prep &P, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3) {
        r3 = r3 < P.pop ? r3 : P.pop;  // padding carries T = +inf and never counts
        r0 = r0 < r3 ? r0 : r3;
        r1 = r1 < r0 ? r0 : (r1 > r3 ? r3 : r1);
        r2 = r2 < r1 ? r1 : (r2 > r3 ? r3 : r2);
        u_acc += P.s ? (P.pre ? r3 - r0 : P.pop) : 0u;
        l_acc += P.pre ? r2 - r1 : 0u;
      };
      fold(A, a0, a1, a2, a3);
      if (haveB) fold(B, b0, b1, b2, b3);
    };
    Prep pa, pb;
    bool havea = false;
    for (uint32_t i = 0; i < n; i++) {  // (wave-uniform control flow throughout)
      if (!havea) {
        havea = prep(i, pa);
      } else if (prep(i, pb)) {
        search2(pa, pb, true);
        havea = false;
      }
    }
    if (havea) search2(pa, pa, false);
  }
  if (h < H) {
    if (u_acc) atomicAdd(&ub[h], u_acc);
    if (l_acc) atomicAdd(&lb[h], l_acc);
  }
}

// candidates of the rank bounds: valid hypotheses whose box-population bound is at least half the largest one and
// above the best of earlier batches (index order); st->n_cand = their number
__global__ __launch_bounds__(1024) void k_pick_cands(const uint32_t *__restrict__ ub, const uint8_t *__restrict__ valid,
                                                     uint32_t H, uint32_t best_before, uint32_t *__restrict__ sel,
                                                     BoundSel *__restrict__ st, uint32_t *__restrict__ votes,
                                                     uint32_t *__restrict__ ub2, uint32_t *__restrict__ lb2,
                                                     uint32_t *__restrict__ lo) {
  __shared__ uint32_t s_red[16], s_scan[1024];
  const int t = threadIdx.x;
  if (H > kSelCap) return;  // 1024 threads x 8 hypotheses (models.h)
  for (uint32_t h = t; h < H; h += 1024) votes[h] = 0, ub2[h] = 0, lb2[h] = 0, lo[h] = 0;  // (uncounted hypotheses report 0)
  uint32_t mx = 0;
  for (uint32_t h = t; h < H; h += 1024) mx = valid[h] && ub[h] > mx ? ub[h] : mx;
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = __shfl_down(mx, o);
    mx = a > mx ? a : mx;
  }
  if ((t & 63) == 0) s_red[t >> 6] = mx;
  __syncthreads();
  mx = 0;
  for (int w = 0; w < 16; w++) mx = s_red[w] > mx ? s_red[w] : mx;
  const uint32_t thr = mx - mx / 2;
  uint32_t f[8], cnt = 0;
  for (int k = 0; k < 8; k++) {
    const uint32_t h = t * 8 + k;
    f[k] = (h < H && valid[h] && mx > 0 && ub[h] >= thr && ub[h] > best_before) ? 1u : 0u;
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
  if (t == 1023) {
    st->n_cand = s_scan[1023];
    st->n_pilot = 0;
    st->n_rest = 0;
    st->known = 0;
  }
}

// The rank bounds of the candidates (compact order of `cand`) folded into per-hypothesis bounds: ub[h] = min(box
// population, rank upper bound) -- both are valid upper bounds -- and lo[h] = rank lower bound (0 for the others:
// zeroed by k_pick_cands).  k_pick_pilots / k_pick_rest (cells.h) then select on these: h is counted iff
//   ub[h] > L[h],   L[h] = max(best of earlier batches, max over h' < h of lo[h'], exact votes of pilots before h)
// -- a lower bound of the serial loop's running maximum when it reaches h.  A hypothesis left out has votes <= ub[h]
// <= L[h] <= running maximum: no update there (strict '>'); a hypothesis the serial loop DOES update on has votes >
// every earlier count >= every earlier lo, so ub >= votes > L: it is counted.  Winner, iteration count and consensus
// set are those of counting everything.  Pilots are only counted when the rank lower bounds are weak (max lo below
// half the largest upper bound: sparse uploads, whose cells are not flat enough for rank bounds).
__global__ __launch_bounds__(256) void k_refine_bounds(const uint32_t *__restrict__ cand, const BoundSel *__restrict__ st,
                                                       const uint32_t *__restrict__ ub2, const uint32_t *__restrict__ lb2,
                                                       uint32_t *__restrict__ ub, uint32_t *__restrict__ lo) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= st->n_cand) return;
  const uint32_t h = cand[j];
  ub[h] = ub2[j] < ub[h] ? ub2[j] : ub[h];
  lo[h] = lb2[j];
}

}  // namespace lsqr
