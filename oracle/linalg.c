/*
 * linalg.c -- oracle (TEST INFRASTRUCTURE ONLY, see lsqr_oracle.h): the numerics the
 * reference takes from VNL, restated from their published algorithms.
 *
 *   orc_sym_eig    <- vnl_symmetric_eigensystem (PlaneParametersEstimator.hxx:163-169,
 *                     LineParametersEstimator.hxx:102-108): ascending eigenvalues, unit
 *                     eigenvectors in the columns of V, sign arbitrary.  Cyclic Jacobi.
 *   orc_svd / orc_pinv_solve <- vnl_svd / vnl_matrix_inverse + zero_out_absolute(tol)
 *                     (DenseLinearEquationSystemParametersEstimator.hxx:38-45,85-92,
 *                     SphereParametersEstimator.hxx:187-194,288-294,
 *                     SinglePointTargetUSCalibrationParametersEstimator.cxx:192-201,228-229).
 *                     One-sided (Hestenes) Jacobi.
 *   orc_lmder      <- vnl_levenberg_marquardt with use_gradient (SphereParametersEstimator.hxx:
 *                     319-331, SinglePointTarget...cxx:282-297): MINPACK lmder + lmpar +
 *                     qrfac + qrsolv, mode 1, factor 100.
 */
#include "lsqr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ symmetric eigen */
void orc_sym_eig(int n, double *A, double *w, double *V) {
  int i, j, p, q, sweep;
  for (i = 0; i < n; i++)
    for (j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (sweep = 0; sweep < 100; sweep++) {
    double off = 0.0, diag = 0.0;
    for (i = 0; i < n; i++) {
      diag += A[i * n + i] * A[i * n + i];
      for (j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
    }
    if (off == 0.0 || off <= 1e-34 * diag) break;
    for (p = 0; p < n - 1; p++)
      for (q = p + 1; q < n; q++) {
        double apq = A[p * n + q];
        if (apq == 0.0) continue;
        {
          double app = A[p * n + p], aqq = A[q * n + q];
          double theta = (aqq - app) / (2.0 * apq);
          double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
          int k;
          for (k = 0; k < n; k++) {
            double akp = A[k * n + p], akq = A[k * n + q];
            A[k * n + p] = c * akp - s * akq;
            A[k * n + q] = s * akp + c * akq;
          }
          for (k = 0; k < n; k++) {
            double apk = A[p * n + k], aqk = A[q * n + k];
            A[p * n + k] = c * apk - s * aqk;
            A[q * n + k] = s * apk + c * aqk;
          }
          for (k = 0; k < n; k++) {
            double vkp = V[k * n + p], vkq = V[k * n + q];
            V[k * n + p] = c * vkp - s * vkq;
            V[k * n + q] = s * vkp + c * vkq;
          }
        }
      }
  }
  for (i = 0; i < n; i++) w[i] = A[i * n + i];
  /* ascending order, columns follow */
  for (i = 0; i < n - 1; i++) {
    int m = i;
    for (j = i + 1; j < n; j++)
      if (w[j] < w[m]) m = j;
    if (m != i) {
      double t = w[i];
      w[i] = w[m];
      w[m] = t;
      for (j = 0; j < n; j++) {
        t = V[j * n + i];
        V[j * n + i] = V[j * n + m];
        V[j * n + m] = t;
      }
    }
  }
}

/* ------------------------------------------------------------------ one-sided Jacobi SVD */
void orc_svd(int m, int n, const double *A, double *U, double *s, double *V) {
  int i, j, k, sweep;
  memcpy(U, A, sizeof(double) * (size_t)m * n);
  for (i = 0; i < n; i++)
    for (j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (sweep = 0; sweep < 60; sweep++) {
    int rotated = 0;
    for (i = 0; i < n - 1; i++)
      for (j = i + 1; j < n; j++) {
        double a = 0, b = 0, g = 0;
        for (k = 0; k < m; k++) {
          double ui = U[k * n + i], uj = U[k * n + j];
          a += ui * ui;
          b += uj * uj;
          g += ui * uj;
        }
        if (g == 0.0 || fabs(g) <= 1e-16 * sqrt(a * b)) continue;
        rotated = 1;
        {
          double zeta = (b - a) / (2.0 * g);
          double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
          for (k = 0; k < m; k++) {
            double ui = U[k * n + i], uj = U[k * n + j];
            U[k * n + i] = c * ui - sn * uj;
            U[k * n + j] = sn * ui + c * uj;
          }
          for (k = 0; k < n; k++) {
            double vi = V[k * n + i], vj = V[k * n + j];
            V[k * n + i] = c * vi - sn * vj;
            V[k * n + j] = sn * vi + c * vj;
          }
        }
      }
    if (!rotated) break;
  }
  for (j = 0; j < n; j++) {
    double nrm = 0;
    for (k = 0; k < m; k++) nrm += U[k * n + j] * U[k * n + j];
    nrm = sqrt(nrm);
    s[j] = nrm;
    if (nrm > 0)
      for (k = 0; k < m; k++) U[k * n + j] /= nrm;
  }
  /* descending singular values */
  for (i = 0; i < n - 1; i++) {
    int mx = i;
    for (j = i + 1; j < n; j++)
      if (s[j] > s[mx]) mx = j;
    if (mx != i) {
      double t = s[i];
      s[i] = s[mx];
      s[mx] = t;
      for (k = 0; k < m; k++) {
        t = U[k * n + i];
        U[k * n + i] = U[k * n + mx];
        U[k * n + mx] = t;
      }
      for (k = 0; k < n; k++) {
        t = V[k * n + i];
        V[k * n + i] = V[k * n + mx];
        V[k * n + mx] = t;
      }
    }
  }
}

int orc_pinv_solve(int m, int n, const double *A, const double *b, double tol, double *x) {
  double *U = (double *)malloc(sizeof(double) * (size_t)m * n);
  double *s = (double *)malloc(sizeof(double) * n);
  double *V = (double *)malloc(sizeof(double) * (size_t)n * n);
  int j, k, rank = 0;
  orc_svd(m, n, A, U, s, V);
  for (k = 0; k < n; k++) x[k] = 0.0;
  for (j = 0; j < n; j++) {
    double d = 0;
    if (!(s[j] > tol)) continue; /* zero_out_absolute: |sigma| <= tol -> 0 */
    rank++;
    for (k = 0; k < m; k++) d += U[k * n + j] * b[k];
    d /= s[j];
    for (k = 0; k < n; k++) x[k] += V[k * n + j] * d;
  }
  free(U);
  free(s);
  free(V);
  return rank;
}

/* ------------------------------------------------------------------ MINPACK lmder */
static double enorm(int n, const double *x, int inc) {
  /* MINPACK enorm guards against over/underflow; the value is sqrt(sum x^2). */
  double scale = 0.0, ssq = 1.0;
  int i;
  for (i = 0; i < n; i++) {
    double a = fabs(x[(size_t)i * inc]);
    if (a == 0.0) continue;
    if (scale < a) {
      ssq = 1.0 + ssq * (scale / a) * (scale / a);
      scale = a;
    } else
      ssq += (a / scale) * (a / scale);
  }
  return scale * sqrt(ssq);
}

#define R_(i, j) r[(size_t)(i) * ldr + (j)]

static void qrsolv(int n, double *r, int ldr, const int *ipvt, const double *diag,
                   const double *qtb, double *x, double *sdiag, double *wa) {
  int i, j, k, l, nsing;
  for (j = 0; j < n; j++) {
    for (i = j; i < n; i++) R_(i, j) = R_(j, i);
    x[j] = R_(j, j);
    wa[j] = qtb[j];
  }
  for (j = 0; j < n; j++) {
    l = ipvt[j];
    if (diag[l] != 0.0) {
      double qtbpj = 0.0;
      for (k = j; k < n; k++) sdiag[k] = 0.0;
      sdiag[j] = diag[l];
      for (k = j; k < n; k++) {
        double c, s, t;
        if (sdiag[k] == 0.0) continue;
        if (fabs(R_(k, k)) < fabs(sdiag[k])) {
          double cotan = R_(k, k) / sdiag[k];
          s = 0.5 / sqrt(0.25 + 0.25 * cotan * cotan);
          c = s * cotan;
        } else {
          double tn = sdiag[k] / R_(k, k);
          c = 0.5 / sqrt(0.25 + 0.25 * tn * tn);
          s = c * tn;
        }
        R_(k, k) = c * R_(k, k) + s * sdiag[k];
        t = c * wa[k] + s * qtbpj;
        qtbpj = -s * wa[k] + c * qtbpj;
        wa[k] = t;
        for (i = k + 1; i < n; i++) {
          t = c * R_(i, k) + s * sdiag[i];
          sdiag[i] = -s * R_(i, k) + c * sdiag[i];
          R_(i, k) = t;
        }
      }
    }
    sdiag[j] = R_(j, j);
    R_(j, j) = x[j];
  }
  nsing = n;
  for (j = 0; j < n; j++) {
    if (sdiag[j] == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa[j] = 0.0;
  }
  for (k = 1; k <= nsing; k++) {
    double sum = 0.0;
    j = nsing - k;
    for (i = j + 1; i < nsing; i++) sum += R_(i, j) * wa[i];
    wa[j] = (wa[j] - sum) / sdiag[j];
  }
  for (j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}

static void lmpar(int n, double *r, int ldr, const int *ipvt, const double *diag,
                  const double *qtb, double delta, double *par, double *x, double *sdiag,
                  double *wa1, double *wa2) {
  const double dwarf = DBL_MIN;
  int i, j, k, l, nsing = n, iter = 0;
  double dxnorm, fp, gnorm, parc, parl, paru, temp;
  for (j = 0; j < n; j++) {
    wa1[j] = qtb[j];
    if (R_(j, j) == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa1[j] = 0.0;
  }
  for (k = 1; k <= nsing; k++) {
    j = nsing - k;
    wa1[j] /= R_(j, j);
    temp = wa1[j];
    for (i = 0; i < j; i++) wa1[i] -= R_(i, j) * temp;
  }
  for (j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
  for (j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
  dxnorm = enorm(n, wa2, 1);
  fp = dxnorm - delta;
  if (fp <= 0.1 * delta) {
    *par = 0.0;
    return;
  }
  parl = 0.0;
  if (nsing >= n) {
    for (j = 0; j < n; j++) {
      l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (j = 0; j < n; j++) {
      double sum = 0.0;
      for (i = 0; i < j; i++) sum += R_(i, j) * wa1[i];
      wa1[j] = (wa1[j] - sum) / R_(j, j);
    }
    temp = enorm(n, wa1, 1);
    parl = ((fp / delta) / temp) / temp;
  }
  for (j = 0; j < n; j++) {
    double sum = 0.0;
    for (i = 0; i <= j; i++) sum += R_(i, j) * qtb[i];
    l = ipvt[j];
    wa1[j] = sum / diag[l];
  }
  gnorm = enorm(n, wa1, 1);
  paru = gnorm / delta;
  if (paru == 0.0) paru = dwarf / (delta < 0.1 ? delta : 0.1);
  if (*par < parl) *par = parl;
  if (*par > paru) *par = paru;
  if (*par == 0.0) *par = gnorm / dxnorm;
  for (;;) {
    iter++;
    if (*par == 0.0) *par = (dwarf > 0.001 * paru) ? dwarf : 0.001 * paru;
    temp = sqrt(*par);
    for (j = 0; j < n; j++) wa1[j] = temp * diag[j];
    qrsolv(n, r, ldr, ipvt, wa1, qtb, x, sdiag, wa2);
    for (j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    dxnorm = enorm(n, wa2, 1);
    temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10)
      break;
    for (j = 0; j < n; j++) {
      l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (j = 0; j < n; j++) {
      wa1[j] /= sdiag[j];
      temp = wa1[j];
      for (i = j + 1; i < n; i++) wa1[i] -= R_(i, j) * temp;
    }
    temp = enorm(n, wa1, 1);
    parc = ((fp / delta) / temp) / temp;
    if (fp > 0.0 && parl < *par) parl = *par;
    if (fp < 0.0 && paru > *par) paru = *par;
    *par = (parl > *par + parc) ? parl : *par + parc;
  }
}
#undef R_

#define A_(i, j) a[(size_t)(i) * n + (j)]

static void qrfac(int m, int n, double *a, int *ipvt, double *rdiag, double *acnorm,
                  double *wa) {
  const double epsmch = DBL_EPSILON;
  int i, j, k, minmn = m < n ? m : n;
  for (j = 0; j < n; j++) {
    acnorm[j] = enorm(m, &A_(0, j), n);
    rdiag[j] = acnorm[j];
    wa[j] = rdiag[j];
    ipvt[j] = j;
  }
  for (j = 0; j < minmn; j++) {
    int kmax = j;
    double ajnorm;
    for (k = j; k < n; k++)
      if (rdiag[k] > rdiag[kmax]) kmax = k;
    if (kmax != j) {
      for (i = 0; i < m; i++) {
        double t = A_(i, j);
        A_(i, j) = A_(i, kmax);
        A_(i, kmax) = t;
      }
      rdiag[kmax] = rdiag[j];
      wa[kmax] = wa[j];
      k = ipvt[j];
      ipvt[j] = ipvt[kmax];
      ipvt[kmax] = k;
    }
    ajnorm = enorm(m - j, &A_(j, j), n);
    if (ajnorm != 0.0) {
      if (A_(j, j) < 0.0) ajnorm = -ajnorm;
      for (i = j; i < m; i++) A_(i, j) /= ajnorm;
      A_(j, j) += 1.0;
      for (k = j + 1; k < n; k++) {
        double sum = 0.0, temp;
        for (i = j; i < m; i++) sum += A_(i, j) * A_(i, k);
        temp = sum / A_(j, j);
        for (i = j; i < m; i++) A_(i, k) -= temp * A_(i, j);
        if (rdiag[k] != 0.0) {
          double d;
          temp = A_(j, k) / rdiag[k];
          d = 1.0 - temp * temp;
          rdiag[k] *= sqrt(d > 0.0 ? d : 0.0);
          d = rdiag[k] / wa[k];
          if (0.05 * d * d <= epsmch) {
            rdiag[k] = enorm(m - j - 1, &A_(j + 1, k), n);
            wa[k] = rdiag[k];
          }
        }
      }
    }
    rdiag[j] = -ajnorm;
  }
}

int orc_lmder(orc_lm_fcn fcn, void *ctx, int m, int n, double *x, double ftol, double xtol,
              double gtol, int maxfev, double factor, int *nfev_out, int *njev_out,
              double *fnorm_out) {
  const double epsmch = DBL_EPSILON;
  int info = 0, nfev = 0, njev = 0, iter = 1, i, j, l;
  double par = 0.0, fnorm = 0, fnorm1, gnorm = 0, delta = 0, xnorm = 0, pnorm, actred, prered,
         dirder, ratio, temp, temp1, temp2;
  double *a, *fvec, *diag, *qtf, *wa1, *wa2, *wa3, *wa4, *r;
  int *ipvt;
  if (n <= 0 || m < n || ftol < 0 || xtol < 0 || gtol < 0 || maxfev <= 0 || factor <= 0) {
    if (nfev_out) *nfev_out = 0;
    if (njev_out) *njev_out = 0;
    return 0;
  }
  a = (double *)malloc(sizeof(double) * (size_t)m * n);
  fvec = (double *)malloc(sizeof(double) * m);
  wa4 = (double *)malloc(sizeof(double) * m);
  diag = (double *)malloc(sizeof(double) * n * 5);
  qtf = diag + n;
  wa1 = qtf + n;
  wa2 = wa1 + n;
  wa3 = wa2 + n;
  r = (double *)malloc(sizeof(double) * (size_t)n * n);
  ipvt = (int *)malloc(sizeof(int) * n);

  fcn(ctx, m, n, x, fvec, NULL, 1);
  nfev = 1;
  fnorm = enorm(m, fvec, 1);
  for (;;) { /* outer loop */
    fcn(ctx, m, n, x, NULL, a, 2);
    njev++;
    qrfac(m, n, a, ipvt, wa1, wa2, wa3);
    if (iter == 1) {
      for (j = 0; j < n; j++) {
        diag[j] = wa2[j];
        if (wa2[j] == 0.0) diag[j] = 1.0;
      }
      for (j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
      xnorm = enorm(n, wa3, 1);
      delta = factor * xnorm;
      if (delta == 0.0) delta = factor;
    }
    memcpy(wa4, fvec, sizeof(double) * m);
    for (j = 0; j < n; j++) {
      if (A_(j, j) != 0.0) {
        double sum = 0.0;
        for (i = j; i < m; i++) sum += A_(i, j) * wa4[i];
        temp = -sum / A_(j, j);
        for (i = j; i < m; i++) wa4[i] += A_(i, j) * temp;
      }
      A_(j, j) = wa1[j];
      qtf[j] = wa4[j];
    }
    /* upper triangle of the factored jacobian -> r */
    for (i = 0; i < n; i++)
      for (j = 0; j < n; j++) r[(size_t)i * n + j] = (j >= i) ? A_(i, j) : 0.0;
    gnorm = 0.0;
    if (fnorm != 0.0)
      for (j = 0; j < n; j++) {
        l = ipvt[j];
        if (wa2[l] != 0.0) {
          double sum = 0.0;
          for (i = 0; i <= j; i++) sum += r[(size_t)i * n + j] * (qtf[i] / fnorm);
          temp = fabs(sum / wa2[l]);
          if (temp > gnorm) gnorm = temp;
        }
      }
    if (gnorm <= gtol) {
      info = 4;
      break;
    }
    for (j = 0; j < n; j++)
      if (wa2[j] > diag[j]) diag[j] = wa2[j];
    for (;;) { /* inner loop */
      double *sdiag = wa2; /* lmpar's sdiag output reuses wa2 as in MINPACK */
      lmpar(n, r, n, ipvt, diag, qtf, delta, &par, wa1, sdiag, wa3, wa4);
      for (j = 0; j < n; j++) {
        wa1[j] = -wa1[j];
        wa2[j] = x[j] + wa1[j];
        wa3[j] = diag[j] * wa1[j];
      }
      pnorm = enorm(n, wa3, 1);
      if (iter == 1 && pnorm < delta) delta = pnorm;
      fcn(ctx, m, n, wa2, wa4, NULL, 1);
      nfev++;
      fnorm1 = enorm(m, wa4, 1);
      actred = -1.0;
      if (0.1 * fnorm1 < fnorm) {
        temp = fnorm1 / fnorm;
        actred = 1.0 - temp * temp;
      }
      for (j = 0; j < n; j++) {
        wa3[j] = 0.0;
        l = ipvt[j];
        temp = wa1[l];
        for (i = 0; i <= j; i++) wa3[i] += r[(size_t)i * n + j] * temp;
      }
      temp1 = enorm(n, wa3, 1) / fnorm;
      temp2 = (sqrt(par) * pnorm) / fnorm;
      prered = temp1 * temp1 + temp2 * temp2 / 0.5;
      dirder = -(temp1 * temp1 + temp2 * temp2);
      ratio = 0.0;
      if (prered != 0.0) ratio = actred / prered;
      if (ratio <= 0.25) {
        if (actred >= 0.0) temp = 0.5;
        else temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * (delta < pnorm / 0.1 ? delta : pnorm / 0.1);
        par /= temp;
      } else if (par == 0.0 || ratio >= 0.75) {
        delta = pnorm / 0.5;
        par *= 0.5;
      }
      if (ratio >= 1e-4) {
        for (j = 0; j < n; j++) {
          x[j] = wa2[j];
          wa2[j] = diag[j] * x[j];
        }
        memcpy(fvec, wa4, sizeof(double) * m);
        xnorm = enorm(n, wa2, 1);
        fnorm = fnorm1;
        iter++;
      }
      if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) info = 1;
      if (delta <= xtol * xnorm) info = 2;
      if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0 && info == 2) info = 3;
      if (info != 0) goto done;
      if (nfev >= maxfev) info = 5;
      if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) info = 6;
      if (delta <= epsmch * xnorm) info = 7;
      if (gnorm <= epsmch) info = 8;
      if (info != 0) goto done;
      if (ratio >= 1e-4) break;
    }
  }
done:
  if (nfev_out) *nfev_out = nfev;
  if (njev_out) *njev_out = njev;
  if (fnorm_out) *fnorm_out = fnorm;
  free(a);
  free(fvec);
  free(wa4);
  free(diag);
  free(r);
  free(ipvt);
  return info;
}
#undef A_
