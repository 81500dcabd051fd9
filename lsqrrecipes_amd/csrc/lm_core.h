// lm_core.h -- Levenberg-Marquardt on device-reduced normal equations.
//
// The reference minimises with vnl_levenberg_marquardt (MINPACK lmder, QR of the N x n
// Jacobian: SphereParametersEstimator.hxx:319-331, SinglePointTarget...Estimator.cxx:282-297).
// Here one HBM pass per function evaluation reduces the observations to
//     mom = { sum f_i^2,  J^T J (upper, packed row-major),  J^T f }
// and this state machine replays lmder's control flow on those n x n quantities:
// R^T R = P^T J^T J P by pivoted Cholesky (same pivot rule as qrfac), Q^T f = R^-T P^T J^T f,
// then lmpar/qrsolv, the actual/predicted reduction test, the trust-region update and the
// info 1..8 stopping rules exactly as lmder orders them.  LSQR_HD: the same source runs in the
// single-thread solve kernel and in the host-compiled unit tests.
#pragma once
#include <float.h>
#include <math.h>

#include "small_linalg.h"

namespace lsqr {

enum { LM_NMAX = 11, LM_MOM_MAX = 1 + LM_NMAX * (LM_NMAX + 1) / 2 + LM_NMAX };

struct LmState {
  int n, iter, nfev, info, maxfev, started;
  int stall, pad0;  // stall: the last evaluation at which the cost still fell by more than 1e-7 relative (diagnostics)
  double ftol, xtol, gtol, factor;
  double x[LM_NMAX], xtrial[LM_NMAX], diag[LM_NMAX], qtf[LM_NMAX], acnorm[LM_NMAX], p[LM_NMAX];
  double r[LM_NMAX * LM_NMAX];
  int ipvt[LM_NMAX];
  double fnorm, par, delta, xnorm, gnorm, pnorm;
  double cstall;    // the cost at `stall`
};

LSQR_HD int lm_mom_len(int n) { return 1 + n * (n + 1) / 2 + n; }

LSQR_HD double lm_enorm(int n, const double *x) {
  double s = 0;
  for (int i = 0; i < n; i++) s += x[i] * x[i];
  return sqrt(s);
}

#define LSQR_R(i, j) r[(i) * n + (j)]

LSQR_HD void lm_qrsolv(int n, double *r, const int *ipvt, const double *diag, const double *qtb,
                       double *x, double *sdiag, double *wa) {
  for (int j = 0; j < n; j++) {
    for (int i = j; i < n; i++) LSQR_R(i, j) = LSQR_R(j, i);
    x[j] = LSQR_R(j, j);
    wa[j] = qtb[j];
  }
  for (int j = 0; j < n; j++) {
    int l = ipvt[j];
    if (diag[l] != 0.0) {
      for (int k = j; k < n; k++) sdiag[k] = 0.0;
      sdiag[j] = diag[l];
      double qtbpj = 0.0;
      for (int k = j; k < n; k++) {
        if (sdiag[k] == 0.0) continue;
        double c, s;
        if (fabs(LSQR_R(k, k)) < fabs(sdiag[k])) {
          double ct = LSQR_R(k, k) / sdiag[k];
          s = 0.5 / sqrt(0.25 + 0.25 * ct * ct);
          c = s * ct;
        } else {
          double tn = sdiag[k] / LSQR_R(k, k);
          c = 0.5 / sqrt(0.25 + 0.25 * tn * tn);
          s = c * tn;
        }
        LSQR_R(k, k) = c * LSQR_R(k, k) + s * sdiag[k];
        double t = c * wa[k] + s * qtbpj;
        qtbpj = -s * wa[k] + c * qtbpj;
        wa[k] = t;
        for (int i = k + 1; i < n; i++) {
          t = c * LSQR_R(i, k) + s * sdiag[i];
          sdiag[i] = -s * LSQR_R(i, k) + c * sdiag[i];
          LSQR_R(i, k) = t;
        }
      }
    }
    sdiag[j] = LSQR_R(j, j);
    LSQR_R(j, j) = x[j];
  }
  int nsing = n;
  for (int j = 0; j < n; j++) {
    if (sdiag[j] == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa[j] = 0.0;
  }
  for (int k = 1; k <= nsing; k++) {
    int j = nsing - k;
    double sum = 0.0;
    for (int i = j + 1; i < nsing; i++) sum += LSQR_R(i, j) * wa[i];
    wa[j] = (wa[j] - sum) / sdiag[j];
  }
  for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}

LSQR_HD void lm_lmpar(int n, double *r, const int *ipvt, const double *diag, const double *qtb,
                      double delta, double *par, double *x) {
  const double dwarf = DBL_MIN;
  double sdiag[LM_NMAX], wa1[LM_NMAX], wa2[LM_NMAX];
  int nsing = n;
  for (int j = 0; j < n; j++) {
    wa1[j] = qtb[j];
    if (LSQR_R(j, j) == 0.0 && nsing == n) nsing = j;
    if (nsing < n) wa1[j] = 0.0;
  }
  for (int k = 1; k <= nsing; k++) {
    int j = nsing - k;
    wa1[j] /= LSQR_R(j, j);
    double t = wa1[j];
    for (int i = 0; i < j; i++) wa1[i] -= LSQR_R(i, j) * t;
  }
  for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
  for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
  double dxnorm = lm_enorm(n, wa2);
  double fp = dxnorm - delta;
  if (fp <= 0.1 * delta) {
    *par = 0.0;
    return;
  }
  double parl = 0.0;
  if (nsing >= n) {
    for (int j = 0; j < n; j++) {
      int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; j++) {
      double sum = 0.0;
      for (int i = 0; i < j; i++) sum += LSQR_R(i, j) * wa1[i];
      wa1[j] = (wa1[j] - sum) / LSQR_R(j, j);
    }
    double t = lm_enorm(n, wa1);
    parl = ((fp / delta) / t) / t;
  }
  for (int j = 0; j < n; j++) {
    double sum = 0.0;
    for (int i = 0; i <= j; i++) sum += LSQR_R(i, j) * qtb[i];
    wa1[j] = sum / diag[ipvt[j]];
  }
  double gnorm = lm_enorm(n, wa1);
  double paru = gnorm / delta;
  if (paru == 0.0) paru = dwarf / (delta < 0.1 ? delta : 0.1);
  if (*par < parl) *par = parl;
  if (*par > paru) *par = paru;
  if (*par == 0.0) *par = gnorm / dxnorm;
  for (int iter = 1;; iter++) {
    if (*par == 0.0) *par = (dwarf > 0.001 * paru) ? dwarf : 0.001 * paru;
    double t = sqrt(*par);
    for (int j = 0; j < n; j++) wa1[j] = t * diag[j];
    lm_qrsolv(n, r, ipvt, wa1, qtb, x, sdiag, wa2);
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    dxnorm = lm_enorm(n, wa2);
    t = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= t && t < 0.0) || iter == 10) break;
    for (int j = 0; j < n; j++) {
      int l = ipvt[j];
      wa1[j] = diag[l] * (wa2[l] / dxnorm);
    }
    for (int j = 0; j < n; j++) {
      wa1[j] /= sdiag[j];
      double tt = wa1[j];
      for (int i = j + 1; i < n; i++) wa1[i] -= LSQR_R(i, j) * tt;
    }
    t = lm_enorm(n, wa1);
    double parc = ((fp / delta) / t) / t;
    if (fp > 0.0 && parl < *par) parl = *par;
    if (fp < 0.0 && paru > *par) paru = *par;
    *par = (parl > *par + parc) ? parl : *par + parc;
  }
}

// factor the normal equations at the current x (lmder's "outer loop" head).  Returns false when
// the gradient test (info 4) fires.
LSQR_HD bool lm_outer(LmState &s, const double *mom) {
  const int n = s.n;
  double *r = s.r;
  double S[LM_NMAX * LM_NMAX], g[LM_NMAX];
  const double *pk = mom + 1;
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) S[i * n + j] = S[j * n + i] = *pk++;
  for (int i = 0; i < n; i++) g[i] = pk[i];
  for (int j = 0; j < n; j++) {
    s.acnorm[j] = sqrt(S[j * n + j] > 0 ? S[j * n + j] : 0.0);
    s.ipvt[j] = j;
  }
  for (int i = 0; i < n * n; i++) r[i] = 0.0;
  // pivoted Cholesky == qrfac's column-pivoted Householder QR up to the signs of R's rows
  for (int j = 0; j < n; j++) {
    int kmax = j;
    for (int k = j; k < n; k++)
      if (S[k * n + k] > S[kmax * n + kmax]) kmax = k;
    if (kmax != j) {
      for (int i = 0; i < n; i++) {
        double t = S[i * n + j];
        S[i * n + j] = S[i * n + kmax];
        S[i * n + kmax] = t;
      }
      for (int i = 0; i < n; i++) {
        double t = S[j * n + i];
        S[j * n + i] = S[kmax * n + i];
        S[kmax * n + i] = t;
      }
      for (int i = 0; i < j; i++) {
        double t = LSQR_R(i, j);
        LSQR_R(i, j) = LSQR_R(i, kmax);
        LSQR_R(i, kmax) = t;
      }
      int t = s.ipvt[j];
      s.ipvt[j] = s.ipvt[kmax];
      s.ipvt[kmax] = t;
    }
    double d = S[j * n + j];
    if (!(d > 0.0)) break;  // remaining Schur complement is zero: rank deficient
    d = sqrt(d);
    LSQR_R(j, j) = d;
    for (int k = j + 1; k < n; k++) LSQR_R(j, k) = S[j * n + k] / d;
    for (int k = j + 1; k < n; k++)
      for (int l = k; l < n; l++) {
        S[k * n + l] -= LSQR_R(j, k) * LSQR_R(j, l);
        S[l * n + k] = S[k * n + l];
      }
  }
  if (s.iter == 1) {
    for (int j = 0; j < n; j++) {
      s.diag[j] = s.acnorm[j];
      if (s.acnorm[j] == 0.0) s.diag[j] = 1.0;
    }
    double wa[LM_NMAX];
    for (int j = 0; j < n; j++) wa[j] = s.diag[j] * s.x[j];
    s.xnorm = lm_enorm(n, wa);
    s.delta = s.factor * s.xnorm;
    if (s.delta == 0.0) s.delta = s.factor;
  }
  for (int j = 0; j < n; j++) {
    double sum = g[s.ipvt[j]];
    for (int i = 0; i < j; i++) sum -= LSQR_R(i, j) * s.qtf[i];
    s.qtf[j] = (LSQR_R(j, j) != 0.0) ? sum / LSQR_R(j, j) : 0.0;
  }
  s.gnorm = 0.0;
  if (s.fnorm != 0.0)
    for (int j = 0; j < n; j++) {
      int l = s.ipvt[j];
      if (s.acnorm[l] != 0.0) {
        double sum = 0.0;
        for (int i = 0; i <= j; i++) sum += LSQR_R(i, j) * (s.qtf[i] / s.fnorm);
        double t = fabs(sum / s.acnorm[l]);
        if (t > s.gnorm) s.gnorm = t;
      }
    }
  if (s.gnorm <= s.gtol) {
    s.info = 4;
    return false;
  }
  for (int j = 0; j < n; j++)
    if (s.acnorm[j] > s.diag[j]) s.diag[j] = s.acnorm[j];
  return true;
}

LSQR_HD void lm_trial(LmState &s) {
  const int n = s.n;
  lm_lmpar(n, s.r, s.ipvt, s.diag, s.qtf, s.delta, &s.par, s.p);
  double wa3[LM_NMAX];
  for (int j = 0; j < n; j++) {
    s.p[j] = -s.p[j];
    s.xtrial[j] = s.x[j] + s.p[j];
    wa3[j] = s.diag[j] * s.p[j];
  }
  s.pnorm = lm_enorm(n, wa3);
  if (s.iter == 1 && s.pnorm < s.delta) s.delta = s.pnorm;
}

LSQR_HD void lm_init(LmState &s, int n, const double *x0, double ftol, double xtol, double gtol,
                     int maxfev, double factor) {
  s.n = n;
  s.iter = 1;
  s.nfev = 0;
  s.info = 0;
  s.maxfev = maxfev;
  s.started = 0;
  s.stall = 0;
  s.pad0 = 0;
  s.cstall = 0.0;
  s.ftol = ftol;
  s.xtol = xtol;
  s.gtol = gtol;
  s.factor = factor;
  s.par = 0.0;
  s.fnorm = s.delta = s.xnorm = s.gnorm = s.pnorm = 0.0;
  for (int j = 0; j < n; j++) s.x[j] = s.xtrial[j] = x0[j];
}

// Consume the moments of the pass evaluated at s.xtrial.  Returns true while another pass (at
// the new s.xtrial) is needed; false when finished (s.info holds the MINPACK code).
LSQR_HD bool lm_advance(LmState &s, const double *mom) {
  const int n = s.n;
  const double epsmch = DBL_EPSILON;
  double *r = s.r;
  s.nfev++;
  if (!s.started) {
    s.started = 1;
    s.fnorm = sqrt(mom[0]);
    s.stall = s.nfev;
    s.cstall = mom[0];
    if (!lm_outer(s, mom)) return false;
    lm_trial(s);
    return true;
  }
  double fnorm1 = sqrt(mom[0]);
  double actred = -1.0;
  if (0.1 * fnorm1 < s.fnorm) {
    double t = fnorm1 / s.fnorm;
    actred = 1.0 - t * t;
  }
  double wa3[LM_NMAX];
  for (int j = 0; j < n; j++) wa3[j] = 0.0;
  for (int j = 0; j < n; j++) {
    double t = s.p[s.ipvt[j]];
    for (int i = 0; i <= j; i++) wa3[i] += LSQR_R(i, j) * t;
  }
  double temp1 = lm_enorm(n, wa3) / s.fnorm;
  double temp2 = (sqrt(s.par) * s.pnorm) / s.fnorm;
  double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
  double dirder = -(temp1 * temp1 + temp2 * temp2);
  double ratio = (prered != 0.0) ? actred / prered : 0.0;
  if (ratio <= 0.25) {
    double t;
    if (actred >= 0.0) t = 0.5;
    else t = 0.5 * dirder / (dirder + 0.5 * actred);
    if (0.1 * fnorm1 >= s.fnorm || t < 0.1) t = 0.1;
    s.delta = t * (s.delta < s.pnorm / 0.1 ? s.delta : s.pnorm / 0.1);
    s.par /= t;
  } else if (s.par == 0.0 || ratio >= 0.75) {
    s.delta = s.pnorm / 0.5;
    s.par *= 0.5;
  }
  bool accepted = ratio >= 1e-4;
  if (accepted) {
    double wa2[LM_NMAX];
    for (int j = 0; j < n; j++) {
      s.x[j] = s.xtrial[j];
      wa2[j] = s.diag[j] * s.x[j];
    }
    s.xnorm = lm_enorm(n, wa2);
    s.fnorm = fnorm1;
    s.iter++;
    if (mom[0] < s.cstall * (1.0 - 1e-7)) {
      s.cstall = mom[0];
      s.stall = s.nfev;
    }
  }
  bool small = fabs(actred) <= s.ftol && prered <= s.ftol && 0.5 * ratio <= 1.0;
  if (small) s.info = 1;
  if (s.delta <= s.xtol * s.xnorm) s.info = 2;
  if (small && s.info == 2) s.info = 3;
  if (s.info != 0) return false;
  if (s.nfev >= s.maxfev) s.info = 5;
  if (fabs(actred) <= epsmch && prered <= epsmch && 0.5 * ratio <= 1.0) s.info = 6;
  if (s.delta <= epsmch * s.xnorm) s.info = 7;
  if (s.gnorm <= epsmch) s.info = 8;
  if (s.info != 0) return false;
  if (accepted && !lm_outer(s, mom)) return false;
  lm_trial(s);
  return true;
}

#undef LSQR_R

}  // namespace lsqr
