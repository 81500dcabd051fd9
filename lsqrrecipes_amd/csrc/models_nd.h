// models_nd.h -- plane / sphere / line in dimension 4 and up (the reference's templates take any
// dimension: PlaneParametersEstimator<dimension>, SphereParametersEstimator<dimension>,
// LineParametersEstimator<dimension>; its own sphere test runs a 4-D case,
// testing/SphereParametersEstimatorTest.cxx:432-468).
//
// agree(), the moment accumulation and the final fits of models.h are dimension-generic and are reused
// as they are; what the general dimension adds is the reference's SVD formulation of the minimal solve:
//   plane  (PlaneParametersEstimator.hxx:70-104)  null vector of the d x (d+1) matrix [p_i, -1],
//          rank decided at EPS = 2.2e-16 on the singular values (vnl_svd + zero_out_absolute)
//   sphere (SphereParametersEstimator.hxx:169-202) x = pinv(A) b / 2 with A_ij = p0_j - p_{i+1,j},
//          b_i = sum_j A_ij (p0_j + p_{i+1,j}), rank < d -> degenerate
// restated with the one-sided Jacobi SVD of small_linalg.h (VNL is absent, so the bit-level behaviour of
// vnl_svd is unpinned: parity of these minimal solves is 1e-6 relative, null-vector sign arbitrary;
// agree() on the resulting model is bit-exact as in every other dimension).
//
// These models carry no packed-fp32 filter and no cell model: the scan is the exact fp64 k_scan.
#pragma once
#include "models.h"

namespace lsqr {

template <int D>
struct PlaneModelN {
  static_assert(D >= 4, "dimensions 2 and 3 are PlaneModel<D>");
  enum { ND = D, K = D, P = 2 * D, SP = 2 * D, REC = D, PPL = 2, IS_DENSE = 0, IS_US = 0 };
  enum { NMOM = PlaneModel<D>::NMOM };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < D; i++) rec[i] = p[i];
  }
  // PlaneParametersEstimator.hxx:70-104,107-108.  The d x (d+1) matrix is padded with a zero row to a
  // square one (same null space, same non-zero singular values); its d+1 singular values are what
  // vnl_svd reports for the d x (d+1) matrix (min(m+1, n) values from LINPACK dsvdc).
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &, double *par) {
    constexpr int n = D + 1;
    double a[n * n], s[n], v[n * n];
    for (int i = 0; i < D; i++) {
      for (int j = 0; j < D; j++) a[i * n + j] = r[i][j];
      a[i * n + D] = -1.0;
    }
    for (int j = 0; j < n; j++) a[D * n + j] = 0.0;
    svd_jacobi(n, n, a, n, s, v);
    int rank = 0, jmin = 0;
    for (int j = 0; j < n; j++) {
      if (s[j] > kEPS) rank++;          // zero_out_absolute(EPS); rank() counts what is left
      if (s[j] < s[jmin]) jmin = j;
    }
    if (rank < D) return false;         // :87-88
    double norm = 0.0;                  // nullvector() = the right singular vector of the smallest value
    for (int i = 0; i < D; i++) norm += v[i * n + jmin] * v[i * n + jmin];
    if (!(norm > 0.0)) return false;    // null vector along the homogeneous axis: no normal (points at infinity)
    norm = 1.0 / sqrt(norm);
    for (int i = 0; i < D; i++) par[i] = v[i * n + jmin] * norm;
    for (int i = 0; i < D; i++) par[D + i] = r[0][i];
    return true;
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return PlaneModel<D>::agree(sp, x, c);        // PlaneParametersEstimator.hxx:196-203
  }
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &c) {
    return PlaneModel<D>::residual(sp, x, c);
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}
  static LSQR_HD void accumulate(const double *x, const double *org, double *m) {
    PlaneModel<D>::accumulate(x, org, m);
  }
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &c, double *par) {
    return PlaneModel<D>::solve(m, org, c, par);  // :129-172
  }
};

template <int D>
struct LineModelN {
  static_assert(D >= 4, "dimensions 2 and 3 are LineModel<D>");
  enum { ND = D, K = 2, P = 2 * D, SP = 2 * D, REC = D, PPL = 2, IS_DENSE = 0, IS_US = 0 };
  enum { NMOM = PlaneModel<D>::NMOM };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < D; i++) rec[i] = p[i];
  }
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &c, double *par) {
    return LineModel<D>::estimate(r, c, par);     // LineParametersEstimator.hxx:23-48 (any dimension)
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return LineModel<D>::agree(sp, x, c);         // :135-150
  }
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &c) {
    return LineModel<D>::residual(sp, x, c);
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}
  static LSQR_HD void accumulate(const double *x, const double *org, double *m) {
    PlaneModel<D>::accumulate(x, org, m);
  }
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &c, double *par) {
    return LineModel<D>::solve(m, org, c, par);   // :68-111
  }
};

template <int D>
struct SphereModelN {
  static_assert(D >= 4, "dimensions 2 and 3 are SphereModel<D>");
  typedef SphereModel<D> B;
  enum { ND = D, K = D + 1, P = D + 1, SP = D + 3, REC = D, PPL = 2, IS_DENSE = 0, IS_US = 0 };
  enum { NMOM = B::NMOM, NLM = B::NLM, NMOM_LM = B::NMOM_LM };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < D; i++) rec[i] = p[i];
  }
  // SphereParametersEstimator.hxx:169-202
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &, double *par) {
    double a[D * D], b[D], x[D], s[D], v[D * D];
    for (int i = 0; i < D; i++) {
      b[i] = 0.0;
      for (int j = 0; j < D; j++) {
        a[i * D + j] = r[0][j] - r[i + 1][j];
        b[i] += a[i * D + j] * (r[0][j] + r[i + 1][j]);
      }
    }
    if (pinv_solve(D, D, a, D, b, kEPS, x, s, v) < D) return false;  // :190-193
    double r2 = 0.0;
    for (int i = 0; i < D; i++) {
      par[i] = x[i] * 0.5;                                            // x = Ainv * b * 0.5
      r2 += (r[0][i] - par[i]) * (r[0][i] - par[i]);
    }
    par[D] = sqrt(r2);
    return true;
  }
  static LSQR_HD void prepare(double *sp, const ModelConsts &c) { B::prepare(sp, c); }
  static LSQR_HD bool use_literal(const double *sp) { return B::use_literal(sp); }
  static LSQR_HD bool agree_literal(const double *sp, const double *x, const ModelConsts &c) {
    return B::agree_literal(sp, x, c);            // :255-264
  }
  static LSQR_HD bool agree_interval(const double *sp, const double *x) { return B::agree_interval(sp, x); }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return B::agree(sp, x, c);
  }
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &c) {
    return B::residual(sp, x, c);
  }
  static LSQR_HD void accumulate(const double *x, const double *org, double *m) { B::accumulate(x, org, m); }
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &c, double *par) {
    return B::solve(m, org, c, par);              // :267-307
  }
  static LSQR_HD int lm_finalize(const double *x, double *par) { return B::lm_finalize(x, par); }
  typedef typename B::LmCoef LmCoef;
  static LSQR_HD void lm_coef(const double *xk, LmCoef &k) { B::lm_coef(xk, k); }
  static LSQR_HD void lm_row(const double *x, const LmCoef &k, double *z) { B::lm_row(x, k, z); }
  static LSQR_HD void accumulate_lm(const double *x, const double *xk, double *m) {
    B::accumulate_lm(x, xk, m);                   // :394-431
  }
};

}  // namespace lsqr
