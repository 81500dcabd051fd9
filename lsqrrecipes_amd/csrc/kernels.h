// kernels.h -- the HIP kernels of the hot path (gfx950 / CDNA4, wave64).
//
//   k_sample     K0  counter-based minimal subsets, one lane per hypothesis
//   k_estimate   K1  minimal-subset solve, one lane per hypothesis (small closed-form models)
//   k_scan       K2  agree() count of H hypotheses over all observations.  Each lane keeps PPL
//                    observations in registers and streams the hypotheses past them (their
//                    parameters arrive through the scalar cache, uniform per wave); the per-
//                    hypothesis inlier count of a wave is a ballot + s_bcnt1 (wavefront reduction),
//                    accumulated per workgroup in LDS and flushed once with one global atomic per
//                    hypothesis per workgroup.  One HBM pass over the observations serves the
//                    whole batch: the kernel is bound by the fp64 VALU rate, not by HBM.
//   k_mask       K3  consensus mask of one model + inlier count
//   k_moments    K4  (masked) reduction of the observations to the model's moment block, fixed
//                    reduction tree (lane-strided partial sums -> wave shuffle tree -> LDS ->
//                    per-block partial) so results are run-to-run deterministic
//   k_reduce / k_solve / k_lm_*  K5  fixed-order sum of the block partials and the small solves
//                    (3x3 / 4x4 Jacobi eigen, LM step on n x n normal equations), single wave
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "models.h"
#include "sampler.h"

namespace lsqr {

constexpr int kBlock = 256;       // 4 waves
constexpr int kMaxPartials = 1024;  // blocks of the moment reduction
// smallest chunk of a block of the mask / moment passes: 1 M records still make ~1000 workgroups (r03; 16 * kBlock
// left a 1 M-frame pass with 244 workgroups of four waves -- one per CU -- at a quarter of the HBM rate)
constexpr int kMomChunk = kBlock * 4;
// ... except where a block's own reduction is long (US / phantom: 80 - 150 sums, a shuffle tree each): 977 instead of
// 244 workgroups made the 1 M-frame mask + moments pass SLOWER (60 -> 88 us)
constexpr int kMomChunkWide = kBlock * 16;

template <int KMAX>
__global__ __launch_bounds__(kBlock) void k_sample(uint64_t seed, uint64_t first, uint32_t H,
                                                   uint64_t n, int K,
                                                   uint32_t *__restrict__ subsets) {
  uint32_t h = blockIdx.x * kBlock + threadIdx.x;
  if (h >= H) return;
  uint32_t idx[KMAX], sorted[KMAX];
  ctr_subset(seed, first + h, n, K, idx, sorted);
  for (int l = 0; l < K; l++) subsets[(size_t)h * K + l] = idx[l];
}

// K0 for large subsets (the dense system draws 64 rows per hypothesis): one WAVE per hypothesis.
// ctr_subset() walks the sorted list of earlier draws (O(k^2) dependent steps on one lane: 0.7 ms for
// 1024 x 64 draws); here lane j holds sorted[j] and the walk becomes the fixed point
//     v = rank + #{chosen <= v}      (ballot + popcount, a few rounds)
// followed by a one-lane shift to insert v.  Same selection rule, same output (test_device_sampler_*).
__global__ __launch_bounds__(64) void k_sample_wave(uint64_t seed, uint64_t first, uint32_t H,
                                                    uint64_t n, int K,
                                                    uint32_t *__restrict__ subsets) {
  const uint32_t hyp = blockIdx.x;
  if (hyp >= H) return;
  const int lane = threadIdx.x;
  const uint64_t h = first + hyp;
  uint32_t sorted = 0xFFFFFFFFu;  // lanes >= number of draws so far hold the sentinel
  uint32_t mine = 0;              // lane l keeps draw l (draw order)
  for (int l = 0; l < K; l++) {
    const uint64_t u = mix64(seed + 0x9E3779B97F4A7C15ULL * (h * 64ULL + (uint64_t)l + 1ULL));
    const uint32_t rank = (uint32_t)mulhi64(u, n - (uint64_t)l);
    uint32_t v = rank, cnt = 0;
    for (;;) {
      cnt = (uint32_t)__builtin_popcountll(__ballot(sorted <= v));
      const uint32_t nv = rank + cnt;
      if (nv == v) break;
      v = nv;
    }
    const uint32_t up = __shfl_up(sorted, 1);
    if ((uint32_t)lane == cnt) sorted = v;
    else if ((uint32_t)lane > cnt) sorted = up;
    if (lane == l) mine = v;
  }
  if (lane < K) subsets[(size_t)hyp * K + lane] = mine;
}

template <class M>
__global__ __launch_bounds__(kBlock) void k_estimate(const double *__restrict__ data,
                                                     size_t stride, size_t n,
                                                     const uint32_t *__restrict__ subsets,
                                                     uint32_t H, ModelConsts mc,
                                                     double *__restrict__ hparams,
                                                     float *__restrict__ hparams_f32,
                                                     uint8_t *__restrict__ valid) {
  uint32_t h = blockIdx.x * kBlock + threadIdx.x;
  if (h >= H) return;
  double r[M::K][M::ND];
  bool ok = true;
  for (int l = 0; l < M::K; l++) {
    size_t i = subsets[(size_t)h * M::K + l];
    if (i >= n) {  // never index outside the observation buffer
      ok = false;
      i = 0;
    }
    for (int j = 0; j < M::ND; j++) r[l][j] = data[i * stride + j];
  }
  double par[M::P];
  ok = ok && M::estimate(r, mc, par);
  const double qnan = __builtin_nan("");
  double sp[M::SP];
  for (int j = 0; j < M::P; j++) sp[j] = ok ? par[j] : qnan;
  for (int j = M::P; j < M::SP; j++) sp[j] = 0.0;
  M::prepare(sp, mc);
  for (int j = 0; j < M::SP; j++) hparams[(size_t)h * M::SP + j] = sp[j];
  if constexpr (requires { M::SPF; }) {
    float f[M::SPF];
    M::prepare_f32(sp, mc, f);
    for (int j = 0; j < M::SPF; j++) hparams_f32[(size_t)h * M::SPF + j] = f[j];
  }
  valid[h] = ok ? 1 : 0;
}

// K2.  grid-stride over tiles of kBlock*PPL observations; dynamic LDS = H counters.
template <class M, int PPL, bool FILT = true>
__global__ __launch_bounds__(kBlock) void k_scan(const double *__restrict__ data, size_t stride,
                                                 size_t n, const double *__restrict__ sp,
                                                 uint32_t H, ModelConsts mc,
                                                 uint32_t *__restrict__ votes) {
  extern __shared__ uint32_t s_cnt[];
  for (uint32_t h = threadIdx.x; h < H; h += kBlock) s_cnt[h] = 0;
  __syncthreads();
  const size_t tile = (size_t)kBlock * PPL;
  const double qnan = __builtin_nan("");
  const bool leader = (threadIdx.x & 63) == 0;
  for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
    double rec[PPL][M::REC];
#pragma unroll
    for (int j = 0; j < PPL; j++) {
      size_t i = base + (size_t)j * kBlock + threadIdx.x;
      bool in = i < n;
      M::load(data + (in ? i : 0) * stride, mc, rec[j]);
      if (!in) {
#pragma unroll
        for (int d = 0; d < M::REC; d++) rec[j][d] = qnan;  // NaN never agrees
      }
    }
    for (uint32_t h = 0; h < H; h++) {
      const double *hp = sp + (size_t)h * M::SP;  // wave-uniform -> scalar loads
      uint32_t c = 0;
      if constexpr (requires { M::use_literal(hp); }) {
        // two formulations of the same predicate; the choice is per hypothesis, so branch on a
        // scalar instead of letting the compiler evaluate both
        if (__builtin_amdgcn_readfirstlane((int)M::use_literal(hp))) {
#pragma unroll
          for (int j = 0; j < PPL; j++)
            c += (uint32_t)__builtin_popcountll(__ballot(M::agree_literal(hp, rec[j], mc)));
        } else {
#pragma unroll
          for (int j = 0; j < PPL; j++)
            c += (uint32_t)__builtin_popcountll(__ballot(M::agree_interval(hp, rec[j])));
        }
      } else if constexpr (FILT && requires { M::filter_value(hp, rec[0]); }) {
        // fused / re-associated fp64 filter with a rigorous band (M::prepare); ambiguous tiles are
        // re-evaluated with the exact predicate
        const double tin = hp[M::P], tout = hp[M::P + 1];
        unsigned long long amb = 0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
          double v = M::filter_value(hp, rec[j]);
          unsigned long long in = __ballot(v < tin), may = __ballot(v < tout);
          c += (uint32_t)__builtin_popcountll(in);
          amb |= in ^ may;
        }
        if (amb) {
          c = 0;
#pragma unroll
          for (int j = 0; j < PPL; j++)
            c += (uint32_t)__builtin_popcountll(__ballot(M::agree(hp, rec[j], mc)));
        }
      } else {
#pragma unroll
        for (int j = 0; j < PPL; j++)
          c += (uint32_t)__builtin_popcountll(__ballot(M::agree(hp, rec[j], mc)));
      }
      if (leader && c) atomicAdd(&s_cnt[h], c);
    }
  }
  __syncthreads();
  for (uint32_t h = threadIdx.x; h < H; h += kBlock) {
    uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[h], c);
  }
}

// derive the scan parameters (M::SP doubles) of one model in place
template <class M>
__global__ void k_prepare(double *par, ModelConsts mc) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double sp[M::SP];
  for (int j = 0; j < M::SP; j++) sp[j] = j < M::P ? par[j] : 0.0;
  M::prepare(sp, mc);
  for (int j = M::P; j < M::SP; j++) par[j] = sp[j];
}

// K2 with an fp32 pre-filter (plane, sphere): the packed-fp32 evaluation (two observations per
// v_pk_* instruction) classifies every observation as certain inlier / certain outlier / ambiguous
// (M::prepare_f32 states the error bound); a wave whose tile holds an ambiguous observation for the
// current hypothesis re-evaluates that tile with the exact fp64 predicate, so the votes are
// bit-identical to k_scan's.
template <class M, int PPL, int GRAN = 0>
__global__ __launch_bounds__(kBlock) void k_scan_f32(const double *__restrict__ data,
                                                           size_t stride, size_t n,
                                                           const double *__restrict__ sp,
                                                           const float *__restrict__ spf,
                                                           uint32_t H, ModelConsts mc,
                                                           uint32_t *__restrict__ votes) {
  constexpr int D = M::ND;
  static_assert(PPL % 2 == 0, "observations are processed in packed pairs");
  extern __shared__ uint32_t s_cnt[];
  for (uint32_t h = threadIdx.x; h < H; h += kBlock) s_cnt[h] = 0;
  __syncthreads();
  const size_t tile = (size_t)kBlock * PPL;
  const bool leader = (threadIdx.x & 63) == 0;
  for (size_t base = (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
    // fp64 records (exact re-check) and their fp32 copies (filter) both stay in registers
    double rec[PPL][D];
    v2f xs[PPL / 2][3];
#pragma unroll
    for (int j = 0; j < PPL; j++) {
      size_t i = base + (size_t)j * kBlock + threadIdx.x;
      const double *p = data + (i < n ? i : 0) * stride;
#pragma unroll
      for (int d = 0; d < D; d++) rec[j][d] = i < n ? p[d] : __builtin_nan("");  // never agrees
    }
#pragma unroll
    for (int q = 0; q < PPL / 2; q++)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        xs[q][d].x = d < D ? (float)rec[2 * q][d] : 0.0f;  // NaN stays NaN: never passes a '<'
        xs[q][d].y = d < D ? (float)rec[2 * q + 1][d] : 0.0f;
      }
    for (uint32_t h = 0; h < H; h++) {
      const v2f *f = (const v2f *)(spf + (size_t)h * M::SPF);  // wave-uniform -> scalar loads
      v2f fp[M::NF];
#pragma unroll
      for (int q = 0; q < M::NF; q++) fp[q] = f[q];
      const float tin = f[M::NF].x, tout = f[M::NF].y;
      v2f a[PPL / 2];
      unsigned long long may[PPL], any = 0;
#pragma unroll
      for (int q = 0; q < PPL / 2; q++) {
        v2f s = M::filter_value(xs[q], fp);
        a[q] = s;
        may[2 * q] = __ballot(__builtin_fabsf(s.x) < tout);
        may[2 * q + 1] = __ballot(__builtin_fabsf(s.y) < tout);
        any |= may[2 * q] | may[2 * q + 1];
      }
      if (any == 0) continue;  // wave-uniform: no observation of this tile is near the model
      uint32_t c = 0;
      if constexpr (GRAN == 0) {
        unsigned long long amb = 0;
#pragma unroll
        for (int q = 0; q < PPL / 2; q++) {
          unsigned long long in0 = __ballot(__builtin_fabsf(a[q].x) < tin);
          unsigned long long in1 = __ballot(__builtin_fabsf(a[q].y) < tin);
          c += (uint32_t)__builtin_popcountll(in0) + (uint32_t)__builtin_popcountll(in1);
          amb |= (in0 ^ may[2 * q]) | (in1 ^ may[2 * q + 1]);
        }
        if (amb) {  // rare, wave-uniform: exact fp64 predicate for this tile and hypothesis
          const double *hp = sp + (size_t)h * M::SP;
          c = 0;
#pragma unroll
          for (int j = 0; j < PPL; j++)
            c += (uint32_t)__builtin_popcountll(__ballot(M::agree(hp, rec[j], mc)));
        }
      } else {  // re-check per packed pair (64 lanes x 2 observations) instead of per tile
        const double *hp = sp + (size_t)h * M::SP;
#pragma unroll
        for (int q = 0; q < PPL / 2; q++) {
          unsigned long long in0 = __ballot(__builtin_fabsf(a[q].x) < tin);
          unsigned long long in1 = __ballot(__builtin_fabsf(a[q].y) < tin);
          if ((in0 ^ may[2 * q]) | (in1 ^ may[2 * q + 1])) {
            in0 = __ballot(M::agree(hp, rec[2 * q], mc));
            in1 = __ballot(M::agree(hp, rec[2 * q + 1], mc));
          }
          c += (uint32_t)__builtin_popcountll(in0) + (uint32_t)__builtin_popcountll(in1);
        }
      }
      if (leader && c) atomicAdd(&s_cnt[h], c);
    }
  }
  __syncthreads();
  for (uint32_t h = threadIdx.x; h < H; h += kBlock) {
    uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[h], c);
  }
}

// max |x| over the first nd doubles of every record (bit patterns of non-negative doubles are
// ordered like the values, so an integer atomicMax works).  The records are read as ONE contiguous stream --
// thread t of a block takes doubles t, t + 256, ... of the block's 256 records and tracks its slot index
// incrementally -- so the pass is coalesced whatever the record size (a lane-per-record loop over 65 or 15
// slots reads 8 B per lane at a 520 B / 120 B stride: 0.13 of the HBM rate).
__global__ __launch_bounds__(kBlock) void k_absmax(const double *__restrict__ data, size_t stride,
                                                   size_t n, int nd, int skip, int nrot,
                                                   unsigned long long *__restrict__ out) {
  // out[0]: over all nd slots (but `skip`); out[1]: over the first nrot slots only
  unsigned long long m = 0, mr = 0;
  const uint32_t st = (uint32_t)stride, step = (uint32_t)(kBlock % stride);
  for (size_t r0 = (size_t)blockIdx.x * kBlock; r0 < n; r0 += (size_t)gridDim.x * kBlock) {
    const size_t rows = n - r0 < (size_t)kBlock ? n - r0 : (size_t)kBlock;
    const size_t cnt = rows * stride;
    const double *base = data + r0 * stride;
    uint32_t d = (uint32_t)(threadIdx.x % stride);
    // eight loads per lane in flight (one at a time ran at 1.7 TB/s: nothing else covers the HBM latency of a pass
    // that does no arithmetic); slots that hold no double (skip) are loaded and dropped
    constexpr int U = 8;
    for (size_t e = threadIdx.x; e < cnt; e += (size_t)U * kBlock) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const size_t eu = e + (size_t)u * kBlock;
        v[u] = base[eu < cnt ? eu : cnt - 1];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool in = e + (size_t)u * kBlock < cnt;
        if (in && (int)d < nd && (int)d != skip) {  // US records: slot 12 holds an int + padding, not a double
          double a = fabs(v[u]);
          unsigned long long b;
          if (!(a == a)) a = __builtin_inf();  // NaN observation: disables the filter
          __builtin_memcpy(&b, &a, 8);
          m = b > m ? b : m;
          if ((int)d < nrot) mr = b > mr ? b : mr;
        }
        d += step;
        if (d >= st) d -= st;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_down(m, o), tr = __shfl_down(mr, o);
    m = t > m ? t : m;
    mr = tr > mr ? tr : mr;
  }
  // one pair of atomics per BLOCK: same-address atomics serialise at ~14 ns each, and with one pair per wave
  // (8192 of them) they, not the 8 TB/s stream, set the time of this pass (0.24 ms for 240 MB)
  __shared__ unsigned long long s_v[kBlock / 64][2];
  if ((threadIdx.x & 63) == 0) s_v[threadIdx.x >> 6][0] = m, s_v[threadIdx.x >> 6][1] = mr;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      m = s_v[w][0] > m ? s_v[w][0] : m;
      mr = s_v[w][1] > mr ? s_v[w][1] : mr;
    }
    if (m) atomicMax(out, m);
    if (mr) atomicMax(out + 1, mr);
  }
}

template <class M>
__global__ __launch_bounds__(kBlock) void k_mask(const double *__restrict__ data, size_t stride,
                                                 size_t begin, size_t end,
                                                 const double *__restrict__ par, ModelConsts mc,
                                                 uint8_t *__restrict__ mask,
                                                 unsigned long long *__restrict__ counter) {
  __shared__ uint32_t s_c[kBlock / 64];
  uint32_t local = 0;
  double sp[M::SP];
  for (int j = 0; j < M::SP; j++) sp[j] = par[j];  // prepared by k_prepare
  for (size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x; i < end;
       i += (size_t)gridDim.x * kBlock) {
    double x[M::REC];
    M::load(data + i * stride, mc, x);
    bool a = M::agree(sp, x, mc);
    mask[i] = a ? 1 : 0;
    local += a ? 1u : 0u;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kBlock / 64; w++) t += s_c[w];
    if (t) atomicAdd(counter, t);
  }
}

// accumulate functors select the phase of a model's moment block
template <class M>
struct AccLs {
  enum { N = M::NMOM };
  static __device__ void acc(const double *x, const double *ctx, double *m) {
    M::accumulate(x, ctx, m);
  }
};
template <class M>
struct AccLm {
  enum { N = M::NMOM_LM };
  static __device__ void acc(const double *x, const double *ctx, double *m) {
    M::accumulate_lm(x, ctx, m);
  }
};

// K4.  block b reduces the contiguous chunk [begin + b*chunk, begin + (b+1)*chunk) ∩ [begin,end).
template <class M, class A, bool USE_MASK>
__global__ __launch_bounds__(kBlock) void k_moments(const double *__restrict__ data,
                                                    size_t stride, size_t begin, size_t end,
                                                    size_t chunk,
                                                    const uint8_t *__restrict__ mask,
                                                    const double *__restrict__ ctxv,
                                                    ModelConsts mc,
                                                    double *__restrict__ partials) {
  __shared__ double s_m[kBlock / 64][A::N];
  double acc[A::N];
#pragma unroll
  for (int k = 0; k < A::N; k++) acc[k] = 0.0;
  double cv[M::P > M::REC ? M::P : M::REC];
  for (int k = 0; k < (M::P > M::REC ? M::P : M::REC); k++) cv[k] = ctxv[k];
  size_t lo = begin + (size_t)blockIdx.x * chunk;
  size_t hi = lo + chunk < end ? lo + chunk : end;
  for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
    if (USE_MASK && !mask[i]) continue;
    double x[M::REC];
    M::load(data + i * stride, mc, x);
    A::acc(x, cv, acc);
  }
#pragma unroll
  for (int k = 0; k < A::N; k++) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < A::N) {
    double t = 0.0;
    for (int w = 0; w < kBlock / 64; w++) t += s_m[w][threadIdx.x];
    partials[(size_t)blockIdx.x * MOM_MAX + threadIdx.x] = t;
  }
}

// K3 + K4 in one pass over the records: consensus mask of the (prepared) model `par` and the phase-0
// moment block of the agreeing records about `ctxv`.  Same chunking, same per-thread order and the same
// partial sums as k_mask followed by k_moments<M, AccLs<M>, true> -- the block is bit-identical -- but the
// records are read once (the winner's mask + fit at 10 M points: 0.115 -> 0.07 ms).
template <class M>
__global__ __launch_bounds__(kBlock) void k_mask_moments(const double *__restrict__ data, size_t stride,
                                                         size_t begin, size_t end, size_t chunk,
                                                         const double *__restrict__ par,
                                                         const double *__restrict__ ctxv, ModelConsts mc,
                                                         uint8_t *__restrict__ mask,
                                                         unsigned long long *__restrict__ counter,
                                                         double *__restrict__ partials) {
  typedef AccLs<M> A;
  __shared__ double s_m[kBlock / 64][A::N];
  __shared__ uint32_t s_c[kBlock / 64];
  double acc[A::N];
#pragma unroll
  for (int k = 0; k < A::N; k++) acc[k] = 0.0;
  double sp[M::SP];
  for (int j = 0; j < M::SP; j++) sp[j] = par[j];  // prepared by k_prepare
  double cv[M::P > M::REC ? M::P : M::REC];
  for (int k = 0; k < (M::P > M::REC ? M::P : M::REC); k++) cv[k] = ctxv[k];
  uint32_t local = 0;
  size_t lo = begin + (size_t)blockIdx.x * chunk;
  size_t hi = lo + chunk < end ? lo + chunk : end;
  // small records: the loads of U records are issued before the first is used (one 24-byte record per lane in
  // flight is 24 KB per CU at 16 waves, a third of what 8 TB/s x the memory latency needs); the records are still
  // accumulated one by one in index order, so the sums keep their bits
  constexpr int U = M::REC <= 4 ? 4 : 1;
  size_t i = lo + threadIdx.x;
  if constexpr (U > 1) {
    for (; i + (size_t)(U - 1) * kBlock < hi; i += (size_t)U * kBlock) {
      double x[U][M::REC];
#pragma unroll
      for (int u = 0; u < U; u++) M::load(data + (i + (size_t)u * kBlock) * stride, mc, x[u]);
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool a = M::agree(sp, x[u], mc);
        mask[i + (size_t)u * kBlock] = a ? 1 : 0;
        if (a) {
          local++;
          A::acc(x[u], cv, acc);
        }
      }
    }
  }
  // (Wide records -- US: 120 B / 144 B -- staged through LDS as a coalesced stream, as k_lm_pass_mfma does, measured
  // SLOWER here: 74 us against 45 us for the 1 M-frame pass; tools/mask_time.py.)
  for (; i < hi; i += kBlock) {
    double x[M::REC];
    M::load(data + i * stride, mc, x);
    const bool a = M::agree(sp, x, mc);
    mask[i] = a ? 1 : 0;
    if (!a) continue;
    local++;
    A::acc(x, cv, acc);
  }
#pragma unroll
  for (int k = 0; k < A::N; k++) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6][k] = v;
  }
  for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x < A::N) {
    double t = 0.0;
    for (int w = 0; w < kBlock / 64; w++) t += s_m[w][threadIdx.x];
    partials[(size_t)blockIdx.x * MOM_MAX + threadIdx.x] = t;
  }
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kBlock / 64; w++) t += s_c[w];
    if (t) atomicAdd(counter, t);
  }
}

// ---- stable compaction of the consensus set (before an iterative fit) ---------------------------------------
// An iterative fit passes over the SAME consensus set once per evaluation (sphere: ~15 times, the US calibration
// with the reference's tolerances: up to 5000 times).  Reading it through the mask touches every cache line of the
// upload (24 B to 144 B records, ~40-50 % of them agreeing: no line is skipped), so the set is first copied, in
// order, into a tight buffer: every later pass reads n_in * sizeof(record) bytes, coalesced, with all lanes busy.
//   k_compact_count   block b counts the set bytes of mask[b * chunk, (b + 1) * chunk)
//   k_compact_scan    exclusive prefix over the (<= 1024) block counts, total -> offs[nb]
//   k_compact_write   block b writes its agreeing records, in order, from offs[b] on
__global__ __launch_bounds__(kBlock) void k_compact_count(const uint8_t *__restrict__ mask, size_t n, size_t chunk,
                                                          uint32_t *__restrict__ counts) {
  __shared__ uint32_t s_c[kBlock / 64];
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  uint32_t c = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) c += mask[i] ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
__global__ __launch_bounds__(1024) void k_compact_scan(const uint32_t *__restrict__ counts, int nb,
                                                       uint32_t *__restrict__ offs) {
  __shared__ uint32_t s[1024];
  const int t = threadIdx.x;
  const uint32_t v = t < nb ? counts[t] : 0u;
  s[t] = v;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
    const uint32_t a = t >= o ? s[t - o] : 0u;
    __syncthreads();
    s[t] += a;
    __syncthreads();
  }
  if (t < nb) offs[t] = s[t] - v;
  if (t == 0) offs[nb] = s[nb > 0 ? nb - 1 : 0];
}
template <int ND>
__global__ __launch_bounds__(kBlock) void k_compact_write(const double *__restrict__ data, size_t stride,
                                                          const uint8_t *__restrict__ mask, size_t n, size_t chunk,
                                                          const uint32_t *__restrict__ offs,
                                                          double *__restrict__ dst) {
  __shared__ uint32_t s_w[kBlock / 64];
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  size_t base = offs[blockIdx.x];
  for (size_t i0 = lo; i0 < hi; i0 += kBlock) {
    const size_t i = i0 + threadIdx.x;
    const bool m = i < hi && mask[i];
    const unsigned long long b = __ballot(m);
    const uint32_t before = (uint32_t)__builtin_popcountll(b & ((1ULL << lane) - 1ULL));
    if (lane == 0) s_w[wave] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int w = 0; w < kBlock / 64; w++) {
      if (w < wave) woff += s_w[w];
      tot += s_w[w];
    }
    if (m) {
      const double *src = data + i * stride;
      double *d = dst + (base + woff + before) * (size_t)ND;
#pragma unroll
      for (int k = 0; k < ND; k++) d[k] = src[k];
    }
    base += tot;
    __syncthreads();
  }
}

// the same compaction into TILES of 64 records stored field-major -- field k of record q at
// dst[((q >> 6) * ND + k) * 64 + (q & 63)] -- so that the matrix-core LM pass (k_lm_pass_mfma_t) reads a wave's 64
// records with ND perfectly coalesced loads, lane = record, and needs no transposition through LDS
template <int ND>
__global__ __launch_bounds__(kBlock) void k_compact_write_tiles(const double *__restrict__ data, size_t stride,
                                                                const uint8_t *__restrict__ mask, size_t n,
                                                                size_t chunk, const uint32_t *__restrict__ offs,
                                                                double *__restrict__ dst) {
  __shared__ uint32_t s_w[kBlock / 64];
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  size_t base = offs[blockIdx.x];
  for (size_t i0 = lo; i0 < hi; i0 += kBlock) {
    const size_t i = i0 + threadIdx.x;
    const bool m = i < hi && mask[i];
    const unsigned long long b = __ballot(m);
    const uint32_t before = (uint32_t)__builtin_popcountll(b & ((1ULL << lane) - 1ULL));
    if (lane == 0) s_w[wave] = (uint32_t)__builtin_popcountll(b);
    __syncthreads();
    uint32_t woff = 0, tot = 0;
    for (int w = 0; w < kBlock / 64; w++) {
      if (w < wave) woff += s_w[w];
      tot += s_w[w];
    }
    if (m) {
      const double *src = data + i * stride;
      const size_t q = base + woff + before;
      double *d = dst + (q >> 6) * (size_t)(ND * 64) + (q & 63);
#pragma unroll
      for (int k = 0; k < ND; k++) d[(size_t)k * 64] = src[k];
    }
    base += tot;
    __syncthreads();
  }
}

// One Levenberg-Marquardt evaluation without a host synchronisation: k_lm_pass is the (masked) reduction of
// k_moments<M, AccLm<M>> at the trial point `xk` -- passed BY VALUE, so consecutive evaluations need no staging
// copy -- and k_lm_publish, next on the stream, sums the block partials (one wave per moment, in parallel across
// the chip: a few microseconds; summing them in the last-arriving block of the pass itself was measured at
// 30-40 us of serial tail) and PUBLISHES the block to host-visible pinned memory: the wave that draws the last
// ticket writes the sequence flag the host's MINPACK control flow polls.  The kernel boundary makes the partials
// visible; the order of every sum depends only on the grid: deterministic.
struct LmX {
  double x[LM_NMAX];
};
template <class M, bool USE_MASK>
__global__ __launch_bounds__(kBlock) void k_lm_pass(const double *__restrict__ data, size_t stride,
                                                    size_t begin, size_t end, size_t chunk,
                                                    const uint8_t *__restrict__ mask, LmX xk, ModelConsts mc,
                                                    double *__restrict__ partials) {
  constexpr int N = M::NMOM_LM;
  __shared__ double s_m[kBlock / 64][N];
  double acc[N];
#pragma unroll
  for (int k = 0; k < N; k++) acc[k] = 0.0;
  size_t lo = begin + (size_t)blockIdx.x * chunk;
  size_t hi = lo + chunk < end ? lo + chunk : end;
  if constexpr (requires(typename M::LmCoef k, double *m) { M::accumulate_lm_fast(m, k, m); }) {
    typename M::LmCoef coef;
    M::lm_coef(xk.x, coef);
    for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
      if (USE_MASK && !mask[i]) continue;
      double x[M::REC];
      M::load(data + i * stride, mc, x);
      M::accumulate_lm_fast(x, coef, acc);
    }
  } else {
    // small records: U loads in flight per lane (see k_mask_moments); accumulated one by one in index order
    constexpr int U = M::REC <= 4 ? 4 : 1;
    size_t i = lo + threadIdx.x;
    if constexpr (U > 1) {
      for (; i + (size_t)(U - 1) * kBlock < hi; i += (size_t)U * kBlock) {
        double x[U][M::REC];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          in[u] = !USE_MASK || mask[i + (size_t)u * kBlock];
          M::load(data + (i + (size_t)u * kBlock) * stride, mc, x[u]);
        }
#pragma unroll
        for (int u = 0; u < U; u++)
          if (in[u]) M::accumulate_lm(x[u], xk.x, acc);
      }
    }
    for (; i < hi; i += kBlock) {
      if (USE_MASK && !mask[i]) continue;
      double x[M::REC];
      M::load(data + i * stride, mc, x);
      M::accumulate_lm(x, xk.x, acc);
    }
  }
#pragma unroll
  for (int k = 0; k < N; k++) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double t = 0.0;
    for (int w = 0; w < kBlock / 64; w++) t += s_m[w][threadIdx.x];
    partials[(size_t)blockIdx.x * MOM_MAX + threadIdx.x] = t;
  }
}

// The same evaluation on the matrix cores (north_star: "MFMA only if J^T J actually becomes a large dense GEMM" --
// over 1 M frames it is one: C = Z^T Z with Z = (J | f), 12 columns, a million rows).  Lane = record: each lane forms
// its row z (M::lm_row: NLM Jacobian entries and the residual), the wave parks its 64 rows in LDS and feeds them, four
// records per instruction, to v_mfma_f64_16x16x4 with the SAME register as A and B operand (lane (k, c) holds
// z_{record k}[c] for both).  The 16 x 16 accumulator (4 doubles per lane) replaces the 78 per-lane accumulators and
// their 78 shuffle trees: 8 accumulator VGPRs instead of 156 (six to eight waves per SIMD instead of two), no
// cross-lane reduction at all.  C[p][q] = (J^T J)_pq, C[p][NLM] = (J^T f)_p, C[NLM][NLM] = sum f^2.
template <class M>
__global__ __launch_bounds__(kBlock) void k_lm_pass_mfma(const double *__restrict__ data, size_t stride, size_t n,
                                                         typename M::LmCoef coef, ModelConsts mc,
                                                         double *__restrict__ partials) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int NZ = M::NLM + 1, P = 18;  // row pitch (doubles) of the LDS tile
  static_assert(NZ <= 16, "one 16 x 16 accumulator tile");
  __shared__ double s_z[kBlock / 64][64 * P];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 4, c16 = lane & 15;
  // `coef` -- the per-evaluation constants (rotation products and their derivatives for the US model) -- is formed on
  // the host and arrives as a kernel argument: wave-uniform, in scalar registers, no sin / cos per lane
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  double *tile = s_z[wave];
  for (size_t base = (size_t)blockIdx.x * kBlock; base < n; base += (size_t)gridDim.x * kBlock) {
    const size_t i = base + threadIdx.x;
    double z[16];
#pragma unroll
    for (int j = 0; j < 16; j++) z[j] = 0.0;
    if constexpr (M::REC > 8 && M::REC <= P) {
      // wide records (US: 120 B / 144 B): the wave's 64 records are one contiguous run when the input is the
      // compacted consensus set (stride == REC).  Read it as a coalesced stream (lane l takes doubles l, l + 64, ...)
      // into the wave's LDS rows and pick the own record up from there -- a lane-per-record read touches 64
      // different cache lines per load instruction.
      const size_t w0 = base + (size_t)wave * 64;  // first record of this wave
      if (stride == (size_t)M::REC && w0 + 64 <= n) {
        const double *src = data + w0 * M::REC;
#pragma unroll
        for (int j = 0; j < M::REC; j++) {
          const int e = lane + 64 * j;             // element of the run
          tile[(e / M::REC) * P + (e % M::REC)] = src[e];
        }
        __builtin_amdgcn_wave_barrier();
        double x[M::REC];
#pragma unroll
        for (int j = 0; j < M::REC; j++) x[j] = (j == 12 && M::IS_US) ? 0.0 : tile[lane * P + j];
        __builtin_amdgcn_wave_barrier();
        M::lm_row(x, coef, z);
      } else if (i < n) {
        double x[M::REC];
        M::load(data + i * stride, mc, x);
        M::lm_row(x, coef, z);
      }
    } else if (i < n) {
      double x[M::REC];
      M::load(data + i * stride, mc, x);
      M::lm_row(x, coef, z);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) tile[lane * P + j] = z[j];
    // the tile is private to the wave and the LDS serves a wave's requests in order: a wave barrier (no instruction,
    // it only stops the compiler from reordering) is all the write -> transposed read hand-over needs
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const double v = tile[(4 * s + k) * P + c16];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // fold the four waves in order; D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double *fold = &s_z[0][0];
  __syncthreads();
  for (int w = 0; w < kBlock / 64; w++) {
    if (wave == w) {
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int pos = (k + 4 * rg) * 16 + c16;
        fold[pos] = (w == 0 ? 0.0 : fold[pos]) + acc[rg];
      }
    }
    __syncthreads();
  }
  // -> the moment block layout {sum f^2, J^T J upper (packed row-major), J^T f}
  constexpr int NL = M::NLM;
  const int t = threadIdx.x;
  if (t < 256) {
    const int r = t >> 4, cc = t & 15;
    double *out = partials + (size_t)blockIdx.x * MOM_MAX;
    if (r < NL && cc < NL && r <= cc) out[1 + r * NL - r * (r - 1) / 2 + (cc - r)] = fold[t];
    if (r < NL && cc == NL) out[1 + NL * (NL + 1) / 2 + r] = fold[t];
    if (r == NL && cc == NL) out[0] = fold[t];
  }
}

// The matrix-core LM pass over the TILE layout of k_compact_write_tiles (r04): lane = record, field j of the wave's
// tile is one coalesced 512-byte load; the NEXT tile's loads are issued before the current rows are formed, so the
// memory latency of a tile is covered by the arithmetic of the one before (k_lm_pass_mfma started every tile by
// waiting for its own loads and moved every record through LDS twice).  Same rows, same accumulation, same block.
template <class M>
__global__ __launch_bounds__(kBlock) void k_lm_pass_mfma_t(const double *__restrict__ tiles, size_t n,
                                                           typename M::LmCoef coef, double *__restrict__ partials) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int NZ = M::NLM + 1, P = 17, REC = M::REC;
  static_assert(NZ <= 16, "one 16 x 16 accumulator tile");
  __shared__ double s_z[kBlock / 64][64 * P];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 4, c16 = lane & 15;
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  double *tile = s_z[wave];
  const size_t ntiles = (n + 63) / 64;
  const size_t W = (size_t)gridDim.x * (kBlock / 64), w0 = (size_t)blockIdx.x * (kBlock / 64) + wave;
  double nx[REC];
  auto fetch = [&](size_t t) {
    const double *src = tiles + t * (size_t)(REC * 64) + lane;
#pragma unroll
    for (int j = 0; j < REC; j++) nx[j] = (j == 12 && M::IS_US) ? 0.0 : src[(size_t)j * 64];
  };
  if (w0 < ntiles) fetch(w0);
  for (size_t t = w0; t < ntiles; t += W) {
    double x[REC];
#pragma unroll
    for (int j = 0; j < REC; j++) x[j] = nx[j];
    if (t + W < ntiles) fetch(t + W);
    double z[16];
#pragma unroll
    for (int j = 0; j < 16; j++) z[j] = 0.0;
    if (t * 64 + lane < n) M::lm_row(x, coef, z);   // (lanes past the end of the last tile: a zero row)
#pragma unroll
    for (int j = 0; j < 16; j++) tile[lane * P + j] = z[j];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const double v = tile[(4 * s + k) * P + c16];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  double *fold = &s_z[0][0];
  __syncthreads();
  for (int w = 0; w < kBlock / 64; w++) {
    if (wave == w) {
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int pos = (k + 4 * rg) * 16 + c16;
        fold[pos] = (w == 0 ? 0.0 : fold[pos]) + acc[rg];
      }
    }
    __syncthreads();
  }
  constexpr int NL = M::NLM;
  const int t = threadIdx.x;
  const int r = t >> 4, cc = t & 15;
  double *out = partials + (size_t)blockIdx.x * MOM_MAX;
  if (r < NL && cc < NL && r <= cc) out[1 + r * NL - r * (r - 1) / 2 + (cc - r)] = fold[t];
  if (r < NL && cc == NL) out[1 + NL * (NL + 1) / 2 + r] = fold[t];
  if (r == NL && cc == NL) out[0] = fold[t];
}

// K3 + K4 of the US calibrations on the matrix cores (r04).  k_mask_moments<US> kept the 91 sums of the analytic
// system {N, A^T A, A^T b} (...Estimator.cxx:137-190: three rows [u R2_j, v R2_j, R2_j, -e_j | -t2_j] per frame) as
// per-lane fp64 accumulators -- 182 VGPRs, one or two waves per SIMD, 16 dependent 120-byte records per lane: 48 us
// for the 121 MB of 1 M frames (0.31 of the HBM peak).  Here, as in k_lm_pass_mfma: the wave's 64 records arrive as
// one coalesced stream (the NEXT tile's loads are issued before the current one is used), every lane evaluates agree()
// on its record (exact predicate, the reference's operation order: the mask keeps its bits), and the rows of the
// AGREEING lanes -- compacted to the front of the wave's LDS tile by their rank in the ballot -- are fed four at a
// time to v_mfma_f64_16x16x4 with the same register as A and B operand: C = Z^T Z, Z = (A | b), 13 (10) columns.  One
// 16 x 16 accumulator (8 VGPRs) per wave, no cross-lane reduction; with half of the frames agreeing 3 x 8 matrix
// instructions per 64 frames.  The sums differ from the per-lane version's in the last bits only (another order).
template <class M>
__global__ __launch_bounds__(kBlock) void k_mask_moments_us_mfma(const double *__restrict__ data, size_t begin, size_t end,
                                                                 const double *__restrict__ par, ModelConsts mc,
                                                                 uint8_t *__restrict__ mask,
                                                                 unsigned long long *__restrict__ counter,
                                                                 double *__restrict__ partials) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int NC = M::NC, NZ = NC + 1, REC = M::REC, P = REC > 16 ? REC + 1 : 17;  // odd row pitch (doubles)
  static_assert(NZ <= 16, "one 16 x 16 accumulator tile");
  __shared__ double s_z[kBlock / 64][64 * P];
  __shared__ uint32_t s_c[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = lane >> 4, c16 = lane & 15;
  double sp[M::SP];
  for (int j = 0; j < M::SP; j++) sp[j] = par[j];  // prepared by k_prepare; wave-uniform
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  uint32_t local = 0;
  double *tile = s_z[wave];
  const size_t total = end - begin, ntiles = (total + 63) / 64;
  const size_t W = (size_t)gridDim.x * (kBlock / 64), w0 = (size_t)blockIdx.x * (kBlock / 64) + wave;
  double nx[REC];
  auto fetch = [&](size_t t) {  // tile t of the range as a coalesced stream: lane l takes doubles l, l + 64, ...
    const size_t b0 = begin + t * 64;
    const double *src = data + b0 * REC;
    const size_t avail = (end - b0) * REC;  // doubles of the range from this tile on
#pragma unroll
    for (int j = 0; j < REC; j++) {
      const size_t e = (size_t)lane + 64 * (size_t)j;
      nx[j] = e < avail ? src[e] : 0.0;
    }
  };
  if (w0 < ntiles) fetch(w0);
  for (size_t t = w0; t < ntiles; t += W) {
#pragma unroll
    for (int j = 0; j < REC; j++) {
      const int e = lane + 64 * j;
      tile[(e / REC) * P + (e % REC)] = nx[j];
    }
    if (t + W < ntiles) fetch(t + W);          // in flight while this tile is evaluated
    __builtin_amdgcn_wave_barrier();
    double x[REC];
#pragma unroll
    for (int j = 0; j < REC; j++) x[j] = (j == 12) ? 0.0 : tile[lane * P + j];  // (slot 12: the int outputFormat)
    __builtin_amdgcn_wave_barrier();
    const size_t i = begin + t * 64 + lane;
    const bool a = i < end && M::agree(sp, x, mc);
    if (i < end) mask[i] = a ? 1 : 0;
    const unsigned long long bal = __ballot(a);
    const int cnt = __builtin_popcountll(bal);
    const int r = __builtin_popcountll(bal & ((1ULL << lane) - 1ULL));  // my rank among the agreeing lanes
    if (lane == 0) local += (uint32_t)cnt;
    const int groups = (cnt + 3) >> 2;                                    // wave-uniform
    if (groups == 0) continue;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (a) {
        double arow[NC];
        const double bj = M::row(x, j, arow);
#pragma unroll
        for (int c = 0; c < NC; c++) tile[r * P + c] = arow[c];
        tile[r * P + NC] = bj;
#pragma unroll
        for (int c = NZ; c < 16; c++) tile[r * P + c] = 0.0;
      }
      if (lane < 48) {  // the rows that complete the last group of four
        const int q = cnt + (lane >> 4);
        if (q < 4 * groups) tile[q * P + c16] = 0.0;
      }
      __builtin_amdgcn_wave_barrier();
      for (int s = 0; s < groups; s++) {
        const double v = tile[(4 * s + k) * P + c16];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  // fold the four waves in order; D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
  double *fold = &s_z[0][0];
  if (lane == 0) s_c[wave] = local;
  __syncthreads();
  for (int w = 0; w < kBlock / 64; w++) {
    if (wave == w) {
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int pos = (k + 4 * rg) * 16 + c16;
        fold[pos] = (w == 0 ? 0.0 : fold[pos]) + acc[rg];
      }
    }
    __syncthreads();
  }
  // -> the moment block layout of USModel::accumulate: {N, A^T A upper (packed row-major), A^T b}
  const int t = threadIdx.x;
  unsigned long long tot = 0;
  for (int w = 0; w < kBlock / 64; w++) tot += s_c[w];
  {
    const int rr = t >> 4, cc = t & 15;
    double *out = partials + (size_t)blockIdx.x * MOM_MAX;
    if (rr < NC && cc < NC && rr <= cc) out[1 + rr * NC - rr * (rr - 1) / 2 + (cc - rr)] = fold[t];
    if (rr < NC && cc == NC) out[1 + NC * (NC + 1) / 2 + rr] = fold[t];
    if (t == 0) {
      out[0] = (double)tot;
      if (tot) atomicAdd(counter, tot);
    }
  }
}

// grid = nmom blocks of one wave.  out: host-visible pinned memory, 2 * nmom 8-byte GRANULES: moment k is published
// as {high word | seq} and {low word | seq}, each one aligned 8-byte store -- a granule is valid the moment its tag
// equals the sequence number of the evaluation the host is waiting for, so no fence, no ticket and no flag ordering
// are needed (r02 published doubles + a flag behind __threadfence_system() per block and an agent-scope ticket:
// 10.7 us per evaluation; MI355X_MICROARCH.md, "handoff-1to1": data-tagged granules).
__global__ __launch_bounds__(64) void k_lm_publish(const double *__restrict__ partials, int nblocks, int nmom,
                                                   unsigned long long *__restrict__ out, uint32_t seq) {
  const int k = blockIdx.x;
  double t = 0.0;
  // fixed order: lane l sums blocks l, l + 64, ...; four loads in flight per lane
  int b = threadIdx.x;
  for (; b + 192 < nblocks; b += 256) {
    const double a0 = partials[(size_t)b * MOM_MAX + k], a1 = partials[(size_t)(b + 64) * MOM_MAX + k],
                 a2 = partials[(size_t)(b + 128) * MOM_MAX + k], a3 = partials[(size_t)(b + 192) * MOM_MAX + k];
    t += a0;
    t += a1;
    t += a2;
    t += a3;
  }
  for (; b < nblocks; b += 64) t += partials[(size_t)b * MOM_MAX + k];
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if (threadIdx.x != 0) return;
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, t);
  __hip_atomic_store(&out[2 * k], (bits & 0xFFFFFFFF00000000ULL) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&out[2 * k + 1], (bits << 32) | seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// fixed-order sum of the per-block partials -> mom[0..nmom): one wave per moment, lane-strided
// partial sums then a shuffle tree (the order depends only on nblocks).
__global__ __launch_bounds__(64) void k_reduce(const double *__restrict__ partials, int nblocks,
                                               int pstride, int nmom, double *__restrict__ mom) {
  int k = blockIdx.x;
  if (k >= nmom) return;
  double t = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) t += partials[(size_t)b * pstride + k];
  for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o);
  if (threadIdx.x == 0) mom[k] = t;
}

// result block written by the solve kernels: {status(1 ok / 0 empty), n_params, lm_info, lm_nfev,
// continue flag}
struct SolveOut {
  int ok, n_params, lm_info, lm_nfev, cont, pad;
  double cost;
  double params[64];
};

template <class M>
__global__ __launch_bounds__(64) void k_solve(const double *__restrict__ mom,
                                              const double *__restrict__ org, ModelConsts mc,
                                              SolveOut *__restrict__ out) {
  __shared__ double m[MOM_MAX];
  __shared__ double ws[M::NMOM > 40 ? 512 : 8];  // workspace for the larger normal-equation solves
  for (int i = threadIdx.x; i < (int)M::NMOM; i += 64) m[i] = mom[i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  double par[M::P];
  bool ok;
  if constexpr (requires { M::solve_ws(m, org, mc, par, ws); }) ok = M::solve_ws(m, org, mc, par, ws);
  else ok = M::solve(m, org, mc, par);
  out->ok = ok ? 1 : 0;
  out->n_params = ok ? M::P : 0;
  out->lm_info = 0;
  out->lm_nfev = 0;
  out->cont = 0;
  out->cost = 0.0;
  for (int j = 0; j < M::P; j++) out->params[j] = ok ? par[j] : 0.0;
}

// LM: state lives in device memory; the trial point the next pass must evaluate is state->xtrial.
__global__ void k_lm_init(LmState *st, const SolveOut *init, int n, double ftol, double xtol,
                          double gtol, int maxfev, double factor) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  lm_init(*st, n, init->params, ftol, xtol, gtol, maxfev, factor);
}

// The step runs on one lane; its state and the moment block are staged in LDS (the state machine
// indexes small matrices dynamically -- from global memory every access would cost an L2 round trip).
template <class M>
__global__ __launch_bounds__(64) void k_lm_advance(LmState *st, const double *__restrict__ mom,
                                                   SolveOut *out) {
  __shared__ LmState s;
  __shared__ double m[LM_MOM_MAX];
  {
    const int nw = (int)(sizeof(LmState) / sizeof(int));
    const int *src = (const int *)st;
    int *dst = (int *)&s;
    for (int i = threadIdx.x; i < nw; i += 64) dst[i] = src[i];
    for (int i = threadIdx.x; i < (int)M::NMOM_LM; i += 64) m[i] = mom[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    bool cont = lm_advance(s, m);
    out->cont = cont ? 1 : 0;
    out->lm_info = s.info;
    out->lm_nfev = s.nfev;
    out->pad = s.stall;
    if (!cont) {
      bool ok = s.info >= 1 && s.info <= 4;  // vnl_levenberg_marquardt::minimize -> true
      out->ok = ok ? 1 : 0;
      out->cost = s.fnorm * s.fnorm;
      int np = M::lm_finalize(s.x, out->params);
      out->n_params = ok ? np : 0;
    }
  }
  __syncthreads();
  {
    const int nw = (int)(sizeof(LmState) / sizeof(int));
    const int *src = (const int *)&s;
    int *dst = (int *)st;
    for (int i = threadIdx.x; i < nw; i += 64) dst[i] = src[i];
  }
}

// first-max winner of a vote array: (votes << 32) | (0xFFFFFFFF - index)
// `base`: offset of this rank's hypotheses inside a multi-GPU batch (the packed values of all ranks are
// then comparable: one all-reduce MAX picks the earliest best hypothesis of the whole batch)
__global__ __launch_bounds__(kBlock) void k_best(const uint32_t *__restrict__ votes,
                                                 const uint8_t *__restrict__ valid, uint32_t H,
                                                 unsigned long long *__restrict__ out, uint32_t base = 0) {
  __shared__ unsigned long long s_b[kBlock / 64];
  unsigned long long best = 0;
  for (uint32_t h = threadIdx.x; h < H; h += kBlock) {
    if (!valid[h]) continue;
    unsigned long long p = ((unsigned long long)votes[h] << 32) | (0xFFFFFFFFu - (base + h));
    best = p > best ? p : best;
  }
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_down(best, o);
    best = other > best ? other : best;
  }
  if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++) best = s_b[w] > best ? s_b[w] : best;
    *out = best;
  }
}

// the winner's subset from the packed winner in device memory (multi-GPU step: no host round trip between
// the all-reduce and the re-derivation of the winning hypothesis); same draw rule as k_sample / k_sample_wave
__global__ __launch_bounds__(64) void k_winner_subset(uint64_t seed, uint64_t batch_first,
                                                      const unsigned long long *__restrict__ packed,
                                                      uint64_t n, int K, uint32_t *__restrict__ subsets) {
  if (threadIdx.x != 0) return;
  const unsigned long long pk = *packed;
  const uint64_t in_batch = pk ? 0xFFFFFFFFull - (pk & 0xFFFFFFFFull) : 0;
  uint32_t idx[64], sorted[64];
  ctr_subset(seed, batch_first + in_batch, n, K, idx, sorted);
  for (int l = 0; l < K; l++) subsets[l] = idx[l];
}

// {moment block, inlier count of the slice} -> the exchange buffer of the multi-GPU step
__global__ __launch_bounds__(256) void k_pack_block(const double *__restrict__ mom, int nmom,
                                                    const unsigned long long *__restrict__ count,
                                                    double *__restrict__ out) {
  for (int i = threadIdx.x; i < nmom; i += 256) out[i] = mom[i];
  if (threadIdx.x == 0) out[nmom] = (double)*count;
}

// residual statistics {min, max, sum, sumsq, count} per block
template <class M, bool USE_MASK>
__global__ __launch_bounds__(kBlock) void k_stats(const double *__restrict__ data, size_t stride,
                                                  size_t n, size_t chunk,
                                                  const uint8_t *__restrict__ mask,
                                                  const double *__restrict__ par,
                                                  ModelConsts mc,
                                                  double *__restrict__ partials) {
  __shared__ double s_m[kBlock / 64][5];
  double mn = __builtin_inf(), mx = -__builtin_inf(), sum = 0, sq = 0, cnt = 0;
  double pv[M::P];
  for (int k = 0; k < M::P; k++) pv[k] = par[k];
  size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += kBlock) {
    if (USE_MASK && !mask[i]) continue;
    double x[M::REC];
    M::load(data + i * stride, mc, x);
    double r = M::residual(pv, x, mc);
    mn = r < mn ? r : mn;
    mx = r > mx ? r : mx;
    sum += r;
    sq = fma(r, r, sq);
    cnt += 1.0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    double a = __shfl_down(mn, o), b = __shfl_down(mx, o);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
    sum += __shfl_down(sum, o);
    sq += __shfl_down(sq, o);
    cnt += __shfl_down(cnt, o);
  }
  if ((threadIdx.x & 63) == 0) {
    double *s = s_m[threadIdx.x >> 6];
    s[0] = mn; s[1] = mx; s[2] = sum; s[3] = sq; s[4] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; w++) {
      mn = s_m[w][0] < mn ? s_m[w][0] : mn;
      mx = s_m[w][1] > mx ? s_m[w][1] : mx;
      sum += s_m[w][2];
      sq += s_m[w][3];
      cnt += s_m[w][4];
    }
    double *o = partials + (size_t)blockIdx.x * 8;
    o[0] = mn; o[1] = mx; o[2] = sum; o[3] = sq; o[4] = cnt;
  }
}

// the model's residual of every record in [begin, end) (PlanePhantom...Estimator.cxx:455-549 returns
// the whole vector next to the statistics)
template <class M>
__global__ __launch_bounds__(kBlock) void k_residuals(const double *__restrict__ data, size_t stride,
                                                      size_t begin, size_t end,
                                                      const double *__restrict__ par, ModelConsts mc,
                                                      double *__restrict__ out) {
  double pv[M::P];
  for (int k = 0; k < M::P; k++) pv[k] = par[k];
  for (size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x; i < end;
       i += (size_t)gridDim.x * kBlock) {
    double x[M::REC];
    M::load(data + i * stride, mc, x);
    out[i - begin] = M::residual(pv, x, mc);
  }
}

}  // namespace lsqr
