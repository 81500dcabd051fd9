// small_linalg.h -- fixed-size dense kernels used by the minimal-subset solves and the final
// fits: cyclic Jacobi symmetric eigen (what the reference takes from vnl_symmetric_eigensystem,
// PlaneParametersEstimator.hxx:163), one-sided Jacobi SVD / pseudo-inverse solve (vnl_svd,
// vnl_matrix_inverse + zero_out_absolute, DenseLinear...Estimator.hxx:38-45) and pivoted
// Cholesky (normal equations of the LM step).  Everything is LSQR_HD so that the same source
// runs in device kernels and in the host-compiled unit tests (tests/host_math/).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define LSQR_HD __host__ __device__ inline
#else
#define LSQR_HD inline
#endif

namespace lsqr {

// sin and cos with the SAME BITS on the host and on the device.  The Levenberg-Marquardt fits of the US calibrations
// form their per-evaluation coefficients from three Euler angles; the step runs on the host (lsqr_hip.hip) or inside
// the persistent kernel (lm_persist.h), and libm's and the device library's sin / cos differ in the last bit for some
// arguments -- enough to send two runs of an ill-conditioned minimisation down different iterates.  This is the
// classical argument reduction by three-part pi/2 (Cody & Waite) and the minimax polynomials on [-pi/4, pi/4] (the
// published fdlibm coefficients), written with plain IEEE multiplies, adds and rint only (the library is compiled
// with -ffp-contract=off), so every operation rounds identically everywhere.  Error < 1 ulp for |x| < 2^19 pi/2;
// beyond that (never an Euler angle of a minimisation) the platform's own functions answer.
LSQR_HD double sc_kernel_sin(double x, double y, bool have_y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double z = x * x, v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (!have_y) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
LSQR_HD double sc_kernel_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double ax = fabs(x);
  if (ax < 0.3) return 1.0 - (0.5 * z - (z * r - x * y));
  // qx ~ x^2 / 2 with few significant bits, so that 1 - qx and x^2 / 2 - qx are exact (fdlibm clears the low word of
  // x / 4; rounding to single precision serves the same purpose and is an IEEE operation on both sides)
  const double qx = ax > 0.78125 ? 0.28125 : (double)(float)(0.25 * ax);
  const double hz = 0.5 * z - qx, a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}
LSQR_HD void lsqr_sincos(double x, double *s, double *c) {
  const double ax = fabs(x);
  if (!(ax < 8.2e5)) {  // huge, infinite or NaN: the platform's functions
    *s = sin(x);
    *c = cos(x);
    return;
  }
  if (ax <= 0.78539816339744830962) {
    *s = sc_kernel_sin(x, 0.0, false);
    *c = sc_kernel_cos(x, 0.0);
    return;
  }
  const double invpio2 = 6.36619772367581382433e-01, p1 = 1.57079632673412561417e+00, p1t = 6.07710050650619224932e-11,
               p2 = 6.07710050630396597660e-11, p2t = 2.02226624879595063154e-21, p3 = 2.02226624871116645580e-21,
               p3t = 8.47842766036889956997e-32;
  const double fn = rint(x * invpio2);
  // three rounds of the reduction, unconditionally (fdlibm stops early when the first difference kept enough bits;
  // running all three costs a dozen operations and removes a data-dependent branch): r - w stays exact to ~118 bits
  double r = x - fn * p1, w = fn * p1t;
  {
    double t = r;
    w = fn * p2;
    r = t - w;
    w = fn * p2t - ((t - r) - w);
    t = r;
    w = fn * p3;
    r = t - w;
    w = fn * p3t - ((t - r) - w);
  }
  const double y0 = r - w, y1 = (r - y0) - w;
  const double ks = sc_kernel_sin(y0, y1, true), kc = sc_kernel_cos(y0, y1);
  switch ((long long)fn & 3) {
    case 0: *s = ks, *c = kc; break;
    case 1: *s = kc, *c = -ks; break;
    case 2: *s = -ks, *c = -kc; break;
    default: *s = -kc, *c = ks; break;
  }
}

// Symmetric eigen decomposition, N <= 64.  a: n*n row-major (destroyed), w ascending,
// v: columns are unit eigenvectors.
LSQR_HD void sym_eig(int n, double *a, double *w, double *v) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) v[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0.0, dg = 0.0;
    for (int i = 0; i < n; i++) {
      dg += a[i * n + i] * a[i * n + i];
      for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
    }
    if (off == 0.0 || off <= 1e-34 * dg) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = a[p * n + q];
        if (apq == 0.0) continue;
        double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq;
          v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) w[i] = a[i * n + i];
  for (int i = 0; i < n - 1; i++) {
    int m = i;
    for (int j = i + 1; j < n; j++)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      double t = w[i];
      w[i] = w[m];
      w[m] = t;
      for (int j = 0; j < n; j++) {
        t = v[j * n + i];
        v[j * n + i] = v[j * n + m];
        v[j * n + m] = t;
      }
    }
  }
}

// Serial one-sided Jacobi SVD of an m*n matrix (m >= n, row-major with leading dimension lda,
// overwritten by U*diag(s) then normalised to U); s unsorted; v n*n.
LSQR_HD void svd_jacobi(int m, int n, double *a, int lda, double *s, double *v) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) v[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < m; k++) {
          double ui = a[k * lda + i], uj = a[k * lda + j];
          al += ui * ui;
          be += uj * uj;
          ga += ui * uj;
        }
        if (ga == 0.0 || fabs(ga) <= 1e-16 * sqrt(al * be)) continue;
        rotated = true;
        double zeta = (be - al) / (2.0 * ga);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int k = 0; k < m; k++) {
          double ui = a[k * lda + i], uj = a[k * lda + j];
          a[k * lda + i] = c * ui - sn * uj;
          a[k * lda + j] = sn * ui + c * uj;
        }
        for (int k = 0; k < n; k++) {
          double vi = v[k * n + i], vj = v[k * n + j];
          v[k * n + i] = c * vi - sn * vj;
          v[k * n + j] = sn * vi + c * vj;
        }
      }
    if (!rotated) break;
  }
  for (int j = 0; j < n; j++) {
    double nrm = 0;
    for (int k = 0; k < m; k++) nrm += a[k * lda + j] * a[k * lda + j];
    nrm = sqrt(nrm);
    s[j] = nrm;
    if (nrm > 0)
      for (int k = 0; k < m; k++) a[k * lda + j] /= nrm;
  }
}

// x = pinv(A) b, singular values <= tol zeroed; returns the rank.  a (m*n, lda) is destroyed;
// s (n) and v (n*n) are scratch.
LSQR_HD int pinv_solve(int m, int n, double *a, int lda, const double *b, double tol, double *x,
                       double *s, double *v) {
  svd_jacobi(m, n, a, lda, s, v);
  int rank = 0;
  for (int k = 0; k < n; k++) x[k] = 0.0;
  for (int j = 0; j < n; j++) {
    if (!(s[j] > tol)) continue;
    rank++;
    double d = 0;
    for (int k = 0; k < m; k++) d += a[k * lda + j] * b[k];
    d /= s[j];
    for (int k = 0; k < n; k++) x[k] += v[k * n + j] * d;
  }
  return rank;
}

// Symmetric positive (semi)definite solve through the eigen decomposition, with the diagonal
// scaled to 1 first; eigenvalues <= rtol * max are treated as zero (rank deficiency).  g (n*n,
// full symmetric, destroyed), rhs n -> x n.  Returns the rank.  work: 2*n*n + 3*n doubles.
LSQR_HD int spd_solve_eig(int n, double *g, const double *rhs, double rtol, double *x,
                          double *work) {
  double *d = work, *w = d + n, *y = w + n, *v = y + n, *gs = v + n * n;
  for (int i = 0; i < n; i++) d[i] = g[i * n + i] > 0 ? 1.0 / sqrt(g[i * n + i]) : 0.0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) gs[i * n + j] = g[i * n + j] * d[i] * d[j];
  sym_eig(n, gs, w, v);
  double wmax = w[n - 1];
  int rank = 0;
  for (int j = 0; j < n; j++) {
    double t = 0;
    for (int i = 0; i < n; i++) t += v[i * n + j] * (rhs[i] * d[i]);
    if (w[j] > rtol * wmax && w[j] > 0) {
      y[j] = t / w[j];
      rank++;
    } else
      y[j] = 0;
  }
  for (int i = 0; i < n; i++) {
    double t = 0;
    for (int j = 0; j < n; j++) t += v[i * n + j] * y[j];
    x[i] = t * d[i];
  }
  return rank;
}

// Fast path of spd_solve_eig for well-conditioned systems: Cholesky of the diagonally scaled matrix
// (unit diagonal).  Returns false -- nothing written to x -- unless every pivot exceeds 1e-8, i.e. far
// from the 1e-13 rank decision of spd_solve_eig, which the caller then takes.  g is NOT modified;
// work: n*n + n doubles.
LSQR_HD bool spd_solve_chol(int n, const double *g, const double *rhs, double *x, double *work) {
  double *l = work, *d = work + n * n;
  for (int i = 0; i < n; i++) {
    if (!(g[i * n + i] > 0.0)) return false;
    d[i] = 1.0 / sqrt(g[i * n + i]);
  }
  for (int j = 0; j < n; j++) {
    double s = g[j * n + j] * d[j] * d[j];
    for (int k = 0; k < j; k++) s -= l[j * n + k] * l[j * n + k];
    if (!(s > 1e-8)) return false;
    const double ljj = sqrt(s);
    l[j * n + j] = ljj;
    for (int i = j + 1; i < n; i++) {
      double t = g[i * n + j] * d[i] * d[j];
      for (int k = 0; k < j; k++) t -= l[i * n + k] * l[j * n + k];
      l[i * n + j] = t / ljj;
    }
  }
  for (int i = 0; i < n; i++) {  // L y = D rhs
    double t = rhs[i] * d[i];
    for (int k = 0; k < i; k++) t -= l[i * n + k] * x[k];
    x[i] = t / l[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {  // L^T z = y, x = D z
    double t = x[i];
    for (int k = i + 1; k < n; k++) t -= l[k * n + i] * x[k];
    x[i] = t / l[i * n + i];
  }
  for (int i = 0; i < n; i++) x[i] *= d[i];
  return true;
}

}  // namespace lsqr
