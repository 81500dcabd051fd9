// dense_model.h -- the per-row arithmetic of DenseLinearEquationSystemParametersEstimator<double,n> (load, agree,
// residual): plain LSQR_HD code, so that the host entry points (host_entry.h) and the CPU sanitizer build compile it
// without the HIP runtime; the kernels are in dense.h.
#pragma once
#include "models.h"

namespace lsqr {

template <int NRp>
struct DenseModel {
  enum { NR = NRp, REC = NRp + 1, SP = NRp, P = NRp, PPL = 1, IS_DENSE = 1, IS_US = 0 };

  static LSQR_HD void load(const double *p, const ModelConsts &c, double *rec) {
    const int n = c.dim;
#pragma unroll
    for (int i = 0; i < NR; i++) rec[i] = i < n ? p[i] : 0.0;
    rec[NR] = p[n];
  }
  // DenseLinearEquationSystemParametersEstimator.hxx:111-119
  static LSQR_HD double signed_res(const double *sp, const double *x) {
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < NR; i++) sum += x[i] * sp[i];
    sum -= x[NR];
    return sum;
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return fabs(signed_res(sp, x)) < c.delta;
  }
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &) {
    return fabs(signed_res(sp, x));
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}
};

}  // namespace lsqr
