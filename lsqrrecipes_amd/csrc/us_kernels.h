// us_kernels.h -- K1 of the ultrasound calibration estimators: the minimal-subset solve is the
// analytic least squares on exactly 4 (3) frames (...Estimator.cxx:17-25 -> :120-270), a 12x12
// (9x9) pseudo-inverse solve with singular values <= FLT_EPSILON zeroed.  One wave per hypothesis,
// system in LDS (wave_linalg.h), post-processing by lane 0 (us.h finish()).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "us.h"
#include "wave_linalg.h"

namespace lsqr {

template <bool SINGLE>
__global__ __launch_bounds__(64) void k_estimate_us(const double *__restrict__ data, size_t stride,
                                                    size_t nobs,
                                                    const uint32_t *__restrict__ subsets,
                                                    uint32_t H, ModelConsts mc,
                                                    double *__restrict__ hparams,
                                                    uint8_t *__restrict__ valid, int fast) {
  typedef USModel<SINGLE> M;
  constexpr int NC = M::NC, MR = 3 * M::K, LDA = 13;
  __shared__ double A[NC * LDA], V[NC * LDA], b[MR], cw[NC], x[NC], recs[M::K][M::ND];
  const int lane = threadIdx.x;
  const uint32_t h = blockIdx.x;
  bool in_range = true;
  for (int idx = lane; idx < M::K * M::ND; idx += 64) {
    int l = idx / M::ND, c = idx % M::ND;
    size_t i = subsets[(size_t)h * M::K + l];
    if (i >= nobs) {
      in_range = false;
      i = 0;
    }
    recs[l][c] = (c == 12) ? 0.0 : data[i * stride + c];
  }
  __syncthreads();
  if (lane < MR) {
    double a[NC];
    b[lane] = M::row(recs[lane / 3], lane % 3, a);
    for (int c = 0; c < NC; c++) A[c * LDA + lane] = a[c];
  }
  __syncthreads();
  // The minimal system is square (12 x 12 / 9 x 9).  Fast path (r04, `us_fast_solve` 1): elimination with partial
  // pivoting by the wave (wave_linalg.h: wave_gepp_solve) -- for a well-conditioned system the reference's SVD
  // pseudo-inverse (...Estimator.cxx:137-201) gives the same solution to cond * eps.  It only ACCEPTS when every pivot
  // exceeds 1e-3 -- four orders above the reference's rank threshold on the singular values (FLT_EPSILON, absolute) --
  // and otherwise the system is rebuilt and takes the SVD below, which makes the reference's rank decision.
  bool solved = false;
  if constexpr (MR == NC) {
    if (fast) {
      solved = wave_gepp_solve(NC, A, LDA, b, x, 1e-8, 1e-3);
      __syncthreads();
      if (!solved) {
        if (lane < MR) {
          double a[NC];
          b[lane] = M::row(recs[lane / 3], lane % 3, a);
          for (int c = 0; c < NC; c++) A[c * LDA + lane] = a[c];
        }
        __syncthreads();
      }
    }
  }
  int rank = NC;
  if (!solved) rank = block_pinv_solve<64, MR, NC>(MR, NC, A, LDA, V, LDA, b, kUsSvEps, 0.0, x, cw);
  bool ok = (rank == NC) && !__any(!in_range);
  if (lane == 0) {
    double par[M::SP];
    if (ok) M::finish(x, par);
    const double qnan = __builtin_nan("");
    for (int j = 0; j < M::P; j++) par[j] = ok ? par[j] : qnan;
    M::prepare(par, mc);
    for (int j = 0; j < M::SP; j++) hparams[(size_t)h * M::SP + j] = par[j];
    valid[h] = ok ? 1 : 0;
  }
}

// fp32 filter block of every hypothesis (after k_estimate_us)
template <class M>
__global__ __launch_bounds__(256) void k_prepare_f32_us(const double *__restrict__ hparams, uint32_t H,
                                                        ModelConsts mc, float *__restrict__ spf) {
  const uint32_t h = blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  double sp[M::SP];
  for (int j = 0; j < M::SP; j++) sp[j] = hparams[(size_t)h * M::SP + j];
  float f[M::SPF];
  M::prepare_f32(sp, mc, f);
  for (int j = 0; j < M::SPF; j++) spf[(size_t)h * M::SPF + j] = f[j];
}

// K2 for the US estimators (cross-wire, pointer, plane phantom: M = USModel<>, PhantomModel) with the
// packed fp32 pre-filter: every lane keeps NP pairs of frames as
// fp32 fields in registers, the hypotheses' M::NF32 fp32 parameters arrive through the scalar cache; a
// v_min over the lane's values gives the one-compare "any candidate in this tile?" test; tiles with a
// candidate take the inlier ballots, tiles with an observation inside the error band re-read the fp64
// records and evaluate the exact predicate (bit-identical votes; same counting scheme as k_scan).
template <class M, int NP>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_scan_us_f32(const double *__restrict__ data, size_t stride,
                                                        size_t n_begin, size_t n, const double *__restrict__ sp,
                                                        const float *__restrict__ spf, uint32_t H,
                                                        ModelConsts mc, uint32_t *__restrict__ votes,
                                                        const uint32_t *__restrict__ h_dev,
                                                        const uint32_t *__restrict__ sel,
                                                        const uint32_t *__restrict__ range_dev) {
  if (range_dev) {  // the frame range comes from device memory (planned by k_ee_plan)
    n_begin = range_dev[0];
    n = range_dev[1];
  }
  // Frames [n_begin, n) against the batch's hypotheses -- the context's batch itself (sel == null) or a compacted
  // selection of it (early exit, earlyexit.h: sp / spf are the selection's compact rows, sel[] maps positions to
  // hypothesis indices, *h_dev is the selection's size).  votes[] is indexed by hypothesis index.
  constexpr int NFLD = M::NFLD, NF = M::NF32;
  extern __shared__ uint32_t s_cnt[];
  if (h_dev) {
    // (read through a vector load: made scalar again, or the per-hypothesis address arithmetic of the loop below
    // runs on the vector unit -- 7 of its ~57 instructions)
    const uint32_t hd = __builtin_amdgcn_readfirstlane(*h_dev);
    H = hd < H ? hd : H;
  }
  // blockIdx.y selects a segment of the hypothesis range: more resident waves when the observations
  // alone give fewer tiles than the chip has wave slots
  // (everything the hypothesis loop addresses with is wave-uniform; said explicitly -- readfirstlane -- because values
  // derived from the vector load of *h_dev otherwise stay in vector registers and drag the per-hypothesis address
  // arithmetic onto the vector unit)
  uint32_t hb0;
  {
    const uint32_t hseg = (H + gridDim.y - 1) / gridDim.y;
    const uint32_t hb = __builtin_amdgcn_readfirstlane(blockIdx.y * hseg);
    if (hb >= H) return;
    sp += (size_t)hb * M::SP;
    spf += (size_t)hb * M::SPF;
    hb0 = hb;
    H = __builtin_amdgcn_readfirstlane(hb + hseg < H ? hseg : H - hb);
  }
  for (uint32_t h = threadIdx.x; h < H; h += kBlock) s_cnt[h] = 0;
  __syncthreads();
  const size_t tile = (size_t)kBlock * 2 * NP;
  const bool leader = (threadIdx.x & 63) == 0;
  for (size_t base = n_begin + (size_t)blockIdx.x * tile; base < n; base += (size_t)gridDim.x * tile) {
    v2f xs[NP][NFLD];
#pragma unroll
    for (int q = 0; q < NP; q++) {
      const size_t i0 = base + (size_t)(2 * q) * kBlock + threadIdx.x, i1 = i0 + kBlock;
      const double *p0 = data + (i0 < n ? i0 : 0) * stride, *p1 = data + (i1 < n ? i1 : 0) * stride;
#pragma unroll
      for (int k = 0; k < NFLD; k++) {
        const int slot = k < 12 ? k : k + 1;  // skip the int outputFormat slot 12
        xs[q][k].x = i0 < n ? (float)p0[slot] : __builtin_nanf("");  // NaN never passes a '<'
        xs[q][k].y = i1 < n ? (float)p1[slot] : __builtin_nanf("");
      }
    }
    // The hypothesis' fp32 block arrives through the scalar cache.  The loads of hypothesis h + 1 are ISSUED at the top
    // of iteration h -- the scheduling barrier pins them there; left alone the compiler sinks them to the end of the
    // body and every iteration starts by waiting out a scalar-cache round trip (r03: vector issue 68 % busy) -- so
    // that the arithmetic of h covers their latency.
    float nx[NF];
#pragma unroll
    for (int k = 0; k < NF; k++) nx[k] = spf[k];
    // (the first block is waited for HERE: with loads still pending at the loop header the compiler's wait-count
    // pass puts an s_waitcnt lgkmcnt(0) behind the loads at the top of every iteration -- scalar loads return out
    // of order, so that waits for the block just requested as well)
#pragma unroll
    for (int k = 0; k < NF; k++) asm volatile("" ::"s"(nx[k]));
    for (uint32_t h = 0; h < H; h++) {
      float fl[NF];
#pragma unroll
      for (int k = 0; k < NF; k++) fl[k] = nx[k];
      {
        const float *f = spf + (size_t)(h + 1 < H ? h + 1 : h) * M::SPF;  // wave-uniform
#pragma unroll
        for (int k = 0; k < NF; k++) nx[k] = f[k];
      }
      __builtin_amdgcn_sched_barrier(0);
      const float tin = fl[M::TIN], tout = fl[M::TIN + 1];
      v2f v[NP];
      float m = __builtin_inff();
#pragma unroll
      for (int q = 0; q < NP; q++) {
        v[q] = M::filter_value_f32(xs[q], fl);
        m = __builtin_fminf(m, __builtin_fminf(v[q].x, v[q].y));
      }
      if (__ballot(m < tout) == 0) continue;  // no frame of this tile is near the target
      uint32_t c = 0;
#pragma unroll
      for (int q = 0; q < NP; q++) {
        bool a0 = v[q].x < tin, a1 = v[q].y < tin;
        const bool b0 = (v[q].x < tout) != a0, b1 = (v[q].y < tout) != a1;  // inside the filter's band
        if (__ballot(b0 || b1)) {
          // exact fp64 predicate for THE LANES whose frame sits in the band, each on its own record.  (r03 had the
          // whole wave re-read both 64-frame halves of the pair -- 15 KB per event, 13.6 GB of fabric reads per launch
          // for a 120 MB upload: profiles/r03_us_full_count_scan_counters.json.)
          const double *hp = sp + (size_t)h * M::SP;
#pragma nounroll
          for (int half = 0; half < 2; half++) {
            if (half ? b1 : b0) {
              const size_t i = base + (size_t)(2 * q + half) * kBlock + threadIdx.x;
              bool ex = false;
              if (i < n) {
                double r[M::REC];
                M::load(data + i * stride, mc, r);
                ex = M::agree(hp, r, mc);
              }
              if (half) a1 = ex;
              else a0 = ex;
            }
          }
        }
        c += (uint32_t)__builtin_popcountll(__ballot(a0)) + (uint32_t)__builtin_popcountll(__ballot(a1));
      }
      if (leader && c) atomicAdd(&s_cnt[h], c);
    }
  }
  __syncthreads();
  for (uint32_t h = threadIdx.x; h < H; h += kBlock) {
    uint32_t c = s_cnt[h];
    if (c) atomicAdd(&votes[sel ? sel[hb0 + h] : hb0 + h], c);
  }
}

}  // namespace lsqr
