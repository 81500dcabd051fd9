// us_kernels.h -- K1 of the ultrasound calibration estimators: the minimal-subset solve is the
// analytic least squares on exactly 4 (3) frames (...Estimator.cxx:17-25 -> :120-270), a 12x12
// (9x9) pseudo-inverse solve with singular values <= FLT_EPSILON zeroed.  One wave per hypothesis,
// system in LDS (wave_linalg.h), post-processing by lane 0 (us.h finish()).
#pragma once
#include <hip/hip_runtime.h>

#include "us.h"
#include "wave_linalg.h"

namespace lsqr {

template <bool SINGLE>
__global__ __launch_bounds__(64) void k_estimate_us(const double *__restrict__ data, size_t stride,
                                                    size_t nobs,
                                                    const uint32_t *__restrict__ subsets,
                                                    uint32_t H, ModelConsts mc,
                                                    double *__restrict__ hparams,
                                                    uint8_t *__restrict__ valid) {
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
  int rank = wave_pinv_solve(MR, NC, A, LDA, V, LDA, b, kUsSvEps, 0.0, x, cw);
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

}  // namespace lsqr
