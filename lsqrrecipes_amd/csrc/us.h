// us.h -- SingleUnknownPointTarget / CalibratedPointerTarget ultrasound calibration estimators
// (parametersEstimators/SinglePointTargetUSCalibrationParametersEstimator.{h,cxx}) on the device.
//
// Record layout (doubles): Frame::rotation[3][3] 0..8 (row-major), Frame::translation 9..11,
// slot 12 = Frame::outputFormat (int) + padding (never read), Point2D q 13..14, and for the
// calibrated-pointer variant Point3D p 15..17  (common/Frame.h:30-31,41; ...Estimator.h:45-48,
// 335-339).  Parameter vectors: SINGLE 20 = [t1(3), t3(3), wz, wy, wx, mx, my, mx*R3(:,1),
// my*R3(:,2), R3(:,3)]; POINTER 17 = the same without t1.
#pragma once
#include <math.h>

#include "lm_core.h"
#include "models.h"
#include "small_linalg.h"

namespace lsqr {

static const double kUsSvEps = 1.192092896e-07;  // ...Estimator.cxx:196,843 (FLT_EPSILON)

template <bool SINGLE>
struct USModel {
  enum {
    ND = SINGLE ? 15 : 18, REC = ND, K = SINGLE ? 4 : 3, P = SINGLE ? 20 : 17, SP = P + 2, PPL = 2,
    IS_DENSE = 0, IS_US = 1,
    NC = SINGLE ? 12 : 9,                       // unknowns of the analytic system
    NMOM = 1 + NC * (NC + 1) / 2 + NC,          // {N, A^T A upper, A^T b}
    NLM = SINGLE ? 11 : 8, NMOM_LM = 1 + NLM * (NLM + 1) / 2 + NLM,
    T3C = SINGLE ? 11 : 8, T3T = SINGLE ? 3 : 0  // offsets of the T3 columns / translation
  };

  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
#pragma unroll
    for (int i = 0; i < ND; i++) rec[i] = (i == 12) ? 0.0 : p[i];
  }

  // q' = T2*T3*[u,v,0,1] (...Estimator.cxx:74-107 / :728-766).  The reference forms the 4x4
  // product T2*T3 and then multiplies by q with vnl's running sums from 0.  The terms dropped here
  // are products with the exact constants 0 and 1 of the homogeneous rows/entries (x*0 = 0,
  // s+0 = s, x*1 = x for finite x), so the remaining roundings are the reference's.
  static LSQR_HD void map(const double *par, const double *x, double q[3]) {
    const double u = x[13], v = x[14];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double r0 = x[3 * i], r1 = x[3 * i + 1], r2 = x[3 * i + 2];
      double m0 = r0 * par[T3C + 0];
      m0 += r1 * par[T3C + 1];
      m0 += r2 * par[T3C + 2];
      double m1 = r0 * par[T3C + 3];
      m1 += r1 * par[T3C + 4];
      m1 += r2 * par[T3C + 5];
      double m3 = r0 * par[T3T + 0];
      m3 += r1 * par[T3T + 1];
      m3 += r2 * par[T3T + 2];
      m3 += x[9 + i];
      double s = m0 * u;
      s += m1 * v;
      s += m3;
      q[i] = s;
    }
  }
  static LSQR_HD double dist_sq(const double *par, const double *x) {
    double q[3];
    map(par, x, q);
    double ex, ey, ez;
    if (SINGLE) {
      ex = q[0] - par[0];
      ey = q[1] - par[1];
      ez = q[2] - par[2];
    } else {
      ex = q[0] - x[15];
      ey = q[1] - x[16];
      ez = q[2] - x[17];
    }
    return ex * ex + ey * ey + ez * ez;
  }
  static LSQR_HD bool agree(const double *par, const double *x, const ModelConsts &c) {
    return dist_sq(par, x) < c.delta_sq;
  }
  static LSQR_HD double residual(const double *par, const double *x, const ModelConsts &) {
    return sqrt(dist_sq(par, x));
  }
  // ---- fp64 fused / re-associated pre-filter ---------------------------------------------------
  // filter_value() evaluates the same squared distance as dist_sq() but as R2*(T3*[u,v,0,1]) + t2 with
  // fused multiply-adds (21 instead of 66 fp64 operations).  With u64 = 2^-53, X = max |entry| of the
  // observations and S3 = max_j (|T3_j0| + |T3_j1|) X + |T3_j3|, every component of the mapped point is
  // bounded by Q = sqrt(3) S3 + X and either evaluation is within Ee = 16 u64 (Q + 2X + |t1|max) of the
  // exact component error e_i; for observations whose exact squared distance is <= 4 delta^2 the two
  // squared distances are therefore within Ed = 4 sqrt(3) delta Ee + 3 Ee^2 + 16 u64 delta^2 of the exact
  // one, i.e. within E = 2.02 Ed of each other (observations beyond 4 delta^2 are far from the test in
  // both; requires E <= delta^2 / 4, else the filter is off for the hypothesis):
  //     v <  delta^2 - E  =>  agrees;   v >= delta^2 + E  =>  does not;   otherwise dist_sq() decides.
  static LSQR_HD void prepare(double *sp, const ModelConsts &c) {
    const double X = c.absmax, u64 = 1.1102230246251565e-16, d2 = c.delta_sq;
    double S3 = 0.0, t1 = 0.0;
    bool finite = true;
    for (int j = 0; j < 3; j++) {
      double s = (fabs(sp[T3C + j]) + fabs(sp[T3C + 3 + j])) * X + fabs(sp[T3T + j]);
      S3 = s > S3 ? s : S3;
      if (SINGLE) t1 = fabs(sp[j]) > t1 ? fabs(sp[j]) : t1;
    }
    for (int j = 0; j < P; j++) finite = finite && sp[j] == sp[j];
    const double Q = 1.7320508075688774 * S3 + X;
    const double Ee = 16.0 * u64 * (Q + 2.0 * X + t1);
    const double Ed = 4.0 * 1.7320508075688774 * c.delta * Ee + 3.0 * Ee * Ee + 16.0 * u64 * d2;
    const double E = 2.02 * Ed;
    bool ok = finite && X <= 1e100 && S3 <= 1e150 && E <= 0.25 * d2 && d2 > 0.0;
    if (!finite) {  // NaN model: never agrees
      sp[P] = sp[P + 1] = __builtin_nan("");
    } else {
      sp[P] = ok ? d2 - E : -INFINITY;      // below: certainly agrees
      sp[P + 1] = ok ? d2 + E : INFINITY;   // at or above: certainly does not
    }
  }
  static LSQR_HD double filter_value(const double *par, const double *x) {
    const double u = x[13], v = x[14];
    double p[3];
#pragma unroll
    for (int j = 0; j < 3; j++) p[j] = fma(par[T3C + j], u, fma(par[T3C + 3 + j], v, par[T3T + j]));
    double d2 = 0.0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      double q = fma(x[3 * i], p[0], fma(x[3 * i + 1], p[1], fma(x[3 * i + 2], p[2], x[9 + i])));
      double e = q - (SINGLE ? par[i] : x[15 + i]);
      d2 = fma(e, e, d2);
    }
    return d2;
  }


  // ---- packed fp32 pre-filter (us_kernels.h: k_scan_us_f32) ------------------------------------------
  // Same measure as filter_value() in fp32, two frames per v_pk_* instruction.  u = 2^-24, X = max |entry|
  // of the records, Rm = max |rotation entry| (both measured on upload), S3 as above, T1 = max |t1_j|:
  //   p_j  = fma(c0_j,u, fma(c1_j,v, t3_j)):    |p_j| <= S3,  error <= 5u S3   (5 input / result roundings)
  //   q_i  = fma(R_i0,p_0, ... + t2_i):          |q_i| <= Q = 3 Rm S3 + X,
  //          error <= 3 Rm (u S3 + 5u S3) + u X + 3u Q <= 10u Q
  //   e_i  = q_i - t1_i (or - p_i of the frame):  error Ee32 = u(10 Q + 2 (Q + T)) = 12u (Q + T),  T = T1 or X
  // and the reference's fp64 e_i is within Ee64 = 16 u64 (Q + 2X + T) of exact: Ee = Ee32 + Ee64 per component.
  // Thresholds: prepare_f32 below (valid for any Ee since r04).
  enum { SPF = 16 };  // c0(3) c1(3) t3(3) t1(3) tin tout 0 0
  // k_scan_us_f32: fp32 scalars fetched per hypothesis, index of tin (tout follows), fp32 record fields
  enum { NF32 = 14, TIN = 12, NFLD = SINGLE ? 14 : 17 };
  static LSQR_HD void prepare_f32(const double *sp, const ModelConsts &c, float *f) {
    const double X = c.absmax, Rm = c.absmax_rot, u = 5.9604644775390625e-08,
                 u64 = 1.1102230246251565e-16, d2 = c.delta_sq;
    double S3 = 0.0, T = SINGLE ? 0.0 : X;
    bool finite = true;
    for (int j = 0; j < 3; j++) {
      double s = (fabs(sp[T3C + j]) + fabs(sp[T3C + 3 + j])) * X + fabs(sp[T3T + j]);
      S3 = s > S3 ? s : S3;
      if (SINGLE) T = fabs(sp[j]) > T ? fabs(sp[j]) : T;
      f[j] = (float)sp[T3C + j];
      f[3 + j] = (float)sp[T3C + 3 + j];
      f[6 + j] = (float)sp[T3T + j];
      f[9 + j] = SINGLE ? (float)sp[j] : 0.0f;
    }
    for (int j = 0; j < P; j++) finite = finite && sp[j] == sp[j];
    const double Q = 3.0 * Rm * S3 + X;
    const double Ee = 12.0 * u * (Q + T) + 16.0 * u64 * (Q + 2.0 * X + T);
    // r04: thresholds that hold for ANY Ee.  The fp32 error vector is within Ee of the reference's in every component,
    // so | |e32| - |e_ref| | <= sqrt3 Ee, and the fp32 sum of squares carries <= 3 roundings (32 u covers them, the
    // fp64 sum's 3 u64 and delta_sq = fl(delta^2) generously):
    //   reference agrees (|e_ref|^2 < delta^2)  =>  v <= (delta + sqrt3 Ee)^2 (1 + 32u):  v >= tout  =>  does not agree
    //   reference does not                      =>  v >= (delta - sqrt3 Ee)^2 (1 - 32u):  v <  tin   =>  agrees
    // (tin = -inf when sqrt3 Ee >= delta: no certain inliers).  r03 priced the error at |e| <= 2 delta (band
    // 4 sqrt3 delta Ee + ...) and REQUIRED Ee <= delta / 8 and E <= delta^2 / 4 -- a hypothesis from a subset with an
    // outlier frame has scale factors in the thousands (S3 ~ 1e5 and beyond) and failed them: 114 of the bench's 4096
    // hypotheses sent every one of the 1 M frames to the exact fp64 predicate, 1.7 M of the scan's 1.8 M exact
    // evaluations (and 11 GB of its fabric reads).
    const double r3 = 1.7320508075688774 * Ee * (1.0 + 1e-9);
    const double hi = c.delta + r3, lo = c.delta - r3;
    const bool ok = finite && X <= 1e15 && X >= 1e-10 && Rm <= 1e15 && S3 <= 1e15 && T <= 1e15 && d2 > 1e-30 &&
                    d2 <= 1e30 && c.delta > 0.0;
    f[12] = (ok && lo > 0.0) ? PlaneModel<3>::round_down_f32(lo * lo * (1.0 - 32.0 * u)) : -INFINITY;
    f[13] = ok ? PlaneModel<3>::round_up_f32(hi * hi * (1.0 + 32.0 * u)) : INFINITY;
    f[14] = f[15] = 0.0f;
    if (!finite) f[12] = f[13] = __builtin_nanf("");  // NaN model: never agrees
  }
#if defined(__HIPCC__)
  // xs: 14 (17) packed fields of two frames: R 0..8, t2 9..11, u 12, v 13, (p 14..16); f as above (scalars)
  static __device__ inline v2f filter_value_f32(const v2f *xs, const float *f) {
    v2f p[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      v2f t = {f[6 + j], f[6 + j]};
      t = __builtin_elementwise_fma(xs[13], (v2f){f[3 + j], f[3 + j]}, t);
      p[j] = __builtin_elementwise_fma(xs[12], (v2f){f[j], f[j]}, t);
    }
    v2f d2 = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      v2f q = __builtin_elementwise_fma(xs[3 * i + 2], p[2], xs[9 + i]);
      q = __builtin_elementwise_fma(xs[3 * i + 1], p[1], q);
      q = __builtin_elementwise_fma(xs[3 * i], p[0], q);
      v2f e = SINGLE ? q - (v2f){f[9 + i], f[9 + i]} : q - xs[14 + i];
      d2 = i == 0 ? e * e : __builtin_elementwise_fma(e, e, d2);
    }
    return d2;
  }
#endif

  // one row (j = 0..2) of the analytic system [u*R2 v*R2 R2 (-I)] x = rhs
  // (...Estimator.cxx:137-190 / :800-836)
  static LSQR_HD double row(const double *x, int j, double *a) {
    const double u = x[13], v = x[14];
    for (int k = 0; k < 3; k++) {
      a[k] = x[3 * j + k] * u;
      a[3 + k] = x[3 * j + k] * v;
      a[6 + k] = x[3 * j + k];
    }
    if (SINGLE) {
      a[9] = a[10] = a[11] = 0.0;
      a[9 + j] = -1.0;
      return -x[9 + j];
    }
    return x[15 + j] - x[9 + j];
  }

  static LSQR_HD void accumulate(const double *x, const double *, double *m) {
    m[0] += 1.0;
    for (int j = 0; j < 3; j++) {
      double a[NC];
      double b = row(x, j, a);
      int k = 1;
      for (int p = 0; p < NC; p++)
        for (int q = p; q < NC; q++, k++) m[k] = fma(a[p], a[q], m[k]);
      for (int p = 0; p < NC; p++, k++) m[k] = fma(a[p], b, m[k]);
    }
  }

  // scale factors, closest rotation, Euler angles -> parameter vector
  // (...Estimator.cxx:204-269 / :851-916)
  static LSQR_HD void finish(const double *x, double *par) {
    double r1[3], r2[3], r3[3], R3[9], s[3], V[9];
    for (int j = 0; j < 3; j++) {
      r1[j] = x[j];
      r2[j] = x[3 + j];
    }
    double m_x = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
    double m_y = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    for (int j = 0; j < 3; j++) {
      r1[j] /= m_x;
      r2[j] /= m_y;
    }
    r3[0] = r1[1] * r2[2] - r1[2] * r2[1];
    r3[1] = r1[2] * r2[0] - r1[0] * r2[2];
    r3[2] = r1[0] * r2[1] - r1[1] * r2[0];
    for (int j = 0; j < 3; j++) {
      R3[3 * j + 0] = r1[j];
      R3[3 * j + 1] = r2[j];
      R3[3 * j + 2] = r3[j];
    }
    svd_jacobi(3, 3, R3, 3, s, V);  // R3 <- U
    double Rn[9];
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) {
        double t = 0;
        for (int l = 0; l < 3; l++) t += R3[3 * j + l] * V[3 * k + l];
        Rn[3 * j + k] = t;
      }
    const double smallAngle = 0.008726535498373935, halfPI = 1.5707963267948966192313216916398;
    double omega_z, omega_x;
    double omega_y = atan2(-Rn[6], sqrt(Rn[0] * Rn[0] + Rn[3] * Rn[3]));
    if (fabs(omega_y - halfPI) > smallAngle && fabs(omega_y + halfPI) > smallAngle) {
      double cy = cos(omega_y);
      omega_z = atan2(Rn[3] / cy, Rn[0] / cy);
      omega_x = atan2(Rn[7] / cy, Rn[8] / cy);
    } else {
      omega_z = 0;
      omega_x = atan2(Rn[1], Rn[4]);
    }
    int k = 0;
    if (SINGLE) {
      par[k++] = x[9];
      par[k++] = x[10];
      par[k++] = x[11];
    }
    par[k++] = x[6];
    par[k++] = x[7];
    par[k++] = x[8];
    par[k++] = omega_z;
    par[k++] = omega_y;
    par[k++] = omega_x;
    par[k++] = m_x;
    par[k++] = m_y;
    par[k++] = m_x * Rn[0];
    par[k++] = m_x * Rn[3];
    par[k++] = m_x * Rn[6];
    par[k++] = m_y * Rn[1];
    par[k++] = m_y * Rn[4];
    par[k++] = m_y * Rn[7];
    par[k++] = Rn[2];
    par[k++] = Rn[5];
    par[k++] = Rn[8];
  }

  // analytic least squares from the normal equations (...Estimator.cxx:120-270 / :775-917).
  // The reference thresholds the singular values of A at FLT_EPSILON; on the normal equations
  // rank deficiency shows as eigenvalues at the rounding floor of the (scaled) Gram matrix.
  // ws: 2*NC*NC + 5*NC doubles (LDS on the device: the eigen solver indexes its matrices dynamically)
  static LSQR_HD bool solve_ws(const double *m, const double *, const ModelConsts &, double *par,
                               double *ws) {
    if (m[0] < (double)K) return false;
    double *G = ws, *rhs = G + NC * NC, *x = rhs + NC, *work = x + NC;  // work: NC*NC + 3*NC ... see below
    int k = 1;
    for (int p = 0; p < NC; p++)
      for (int q = p; q < NC; q++, k++) G[p * NC + q] = G[q * NC + p] = m[k];
    for (int p = 0; p < NC; p++, k++) rhs[p] = m[k];
    // well-conditioned normal equations: Cholesky (a few hundred flops on this one lane); near the
    // rank decision the eigen decomposition that makes it (0.8 ms on one lane for 12 x 12)
    if (!spd_solve_chol(NC, G, rhs, x, work)) {
      int rank = spd_solve_eig(NC, G, rhs, 1e-13, x, work);
      if (rank < NC) return false;
    }
    finish(x, par);
    return true;
  }
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &c, double *par) {
    double ws[3 * NC * NC + 5 * NC];
    return solve_ws(m, org, c, par, ws);
  }

  // f (...Estimator.cxx:415-509 / :1059-1146) and gradf (:512-658 / :1149-1286) of one frame,
  // reduced to {sum f^2, J^T J upper, J^T f}
  static LSQR_HD void accumulate_lm(const double *rec, const double *xk, double *m) {
    const int o = SINGLE ? 3 : 0;
    double t_1x = 0, t_1y = 0, t_1z = 0;
    if (SINGLE) {
      t_1x = xk[0];
      t_1y = xk[1];
      t_1z = xk[2];
    }
    const double t_3x = xk[o + 0], t_3y = xk[o + 1], t_3z = xk[o + 2];
    const double sz = sin(xk[o + 3]), cz = cos(xk[o + 3]);
    const double sy = sin(xk[o + 4]), cy = cos(xk[o + 4]);
    const double sx = sin(xk[o + 5]), cx = cos(xk[o + 5]);
    const double m_x = xk[o + 6], m_y = xk[o + 7];
    const double R3_11 = cz * cy, R3_21 = sz * cy, R3_31 = -sy;
    const double R3_12 = cz * sy * sx - sz * cx, R3_22 = sz * sy * sx + cz * cx, R3_32 = cy * sx;
    const double *R2 = rec, *t2 = rec + 9;
    const double u = rec[13], v = rec[14];
    double A1[9], A2[9], A3[9];  // A_i1..A_i9 of the reference, per row i
    for (int k = 0; k < 3; k++) {
      A1[k] = u * R2[k];     A1[3 + k] = v * R2[k];     A1[6 + k] = R2[k];
      A2[k] = u * R2[3 + k]; A2[3 + k] = v * R2[3 + k]; A2[6 + k] = R2[3 + k];
      A3[k] = u * R2[6 + k]; A3[3 + k] = v * R2[6 + k]; A3[6 + k] = R2[6 + k];
    }
    double b_1, b_2, b_3;
    if (SINGLE) {
      b_1 = -t2[0]; b_2 = -t2[1]; b_3 = -t2[2];
    } else {
      b_1 = rec[15] - t2[0]; b_2 = rec[16] - t2[1]; b_3 = rec[17] - t2[2];
    }
#define LSQR_US_EXPR(A, t1, b)                                                                  \
  (A[0] * m_x * R3_11 + A[1] * m_x * R3_21 + A[2] * m_x * R3_31 + A[3] * m_y * R3_12 +           \
   A[4] * m_y * R3_22 + A[5] * m_y * R3_32 + A[6] * t_3x + A[7] * t_3y + A[8] * t_3z - (t1) - (b))
    const double expr1 = LSQR_US_EXPR(A1, t_1x, b_1);
    const double expr2 = LSQR_US_EXPR(A2, t_1y, b_2);
    const double expr3 = LSQR_US_EXPR(A3, t_1z, b_3);
#undef LSQR_US_EXPR
    const double delta_i = sqrt(expr1 * expr1 + expr2 * expr2 + expr3 * expr3);
    double J[NLM];
    if (SINGLE) {
      J[0] = -expr1 / delta_i;
      J[1] = -expr2 / delta_i;
      J[2] = -expr3 / delta_i;
    }
    J[o + 0] = (A1[6] * expr1 + A2[6] * expr2 + A3[6] * expr3) / delta_i;
    J[o + 1] = (A1[7] * expr1 + A2[7] * expr2 + A3[7] * expr3) / delta_i;
    J[o + 2] = (A1[8] * expr1 + A2[8] * expr2 + A3[8] * expr3) / delta_i;
    double v1 = -m_x * sz * cy, v2 = m_x * cz * cy, v3 = -m_y * (sz * sy * sx + cz * cx),
           v4 = m_y * (cz * sy * sx - sz * cx), v5, v6;
    J[o + 3] = ((A1[0] * v1 + A1[1] * v2 + A1[3] * v3 + A1[4] * v4) * expr1 +
                (A2[0] * v1 + A2[1] * v2 + A2[3] * v3 + A2[4] * v4) * expr2 +
                (A3[0] * v1 + A3[1] * v2 + A3[3] * v3 + A3[4] * v4) * expr3) / delta_i;
    v1 = -m_x * sy * cz;
    v2 = -m_x * sy * sz;
    v3 = -m_x * cy;
    v4 = m_y * sx * cy * cz;
    v5 = m_y * sx * cy * sz;
    v6 = -m_y * sx * sy;
    J[o + 4] = ((A1[0] * v1 + A1[1] * v2 + A1[2] * v3 + A1[3] * v4 + A1[4] * v5 + A1[5] * v6) * expr1 +
                (A2[0] * v1 + A2[1] * v2 + A2[2] * v3 + A2[3] * v4 + A2[4] * v5 + A2[5] * v6) * expr2 +
                (A3[0] * v1 + A3[1] * v2 + A3[2] * v3 + A3[3] * v4 + A3[4] * v5 + A3[5] * v6) * expr3) /
               delta_i;
    v1 = m_y * (cz * sy * cx + sz * sx);
    v2 = m_y * (sz * sy * cx - cz * sx);
    v3 = m_y * cy * cx;
    J[o + 5] = ((A1[3] * v1 + A1[4] * v2 + A1[5] * v3) * expr1 +
                (A2[3] * v1 + A2[4] * v2 + A2[5] * v3) * expr2 +
                (A3[3] * v1 + A3[4] * v2 + A3[5] * v3) * expr3) / delta_i;
    J[o + 6] = ((A1[0] * R3_11 + A1[1] * R3_21 + A1[2] * R3_31) * expr1 +
                (A2[0] * R3_11 + A2[1] * R3_21 + A2[2] * R3_31) * expr2 +
                (A3[0] * R3_11 + A3[1] * R3_21 + A3[2] * R3_31) * expr3) / delta_i;
    J[o + 7] = ((A1[3] * R3_12 + A1[4] * R3_22 + A1[5] * R3_32) * expr1 +
                (A2[3] * R3_12 + A2[4] * R3_22 + A2[5] * R3_32) * expr2 +
                (A3[3] * R3_12 + A3[4] * R3_22 + A3[5] * R3_32) * expr3) / delta_i;
    m[0] = fma(delta_i, delta_i, m[0]);
    int k = 1;
    for (int p = 0; p < NLM; p++)
      for (int q = p; q < NLM; q++, k++) m[k] = fma(J[p], J[q], m[k]);
    for (int p = 0; p < NLM; p++, k++) m[k] = fma(J[p], delta_i, m[k]);
  }

  // ---- one Levenberg-Marquardt evaluation, re-associated (k_lm_pass) ------------------------------------------
  // accumulate_lm above follows the reference's f / gradf term by term (11 divisions by the residual, a square
  // root and ~600 flops per frame: the pass was bound by the vector ALU, 55 us per 1 M frames).  The same
  // sums in a cheaper association: with q = u c0 + v c1 + t3 (c0 = m_x R3(:,1), c1 = m_y R3(:,2)),
  //   e = R2 q + t2 - t1      (= [expr1, expr2, expr3]),   f = |e|,   J_p = e . de/dp / |e| = g_p / |e|,
  // so   sum f^2 = sum e.e,   J^T f = sum g,   J^T J = sum g g^T / (e.e)   -- one division per frame, no root --
  // and every probe-side derivative is  de/dp = R2 (u a_p + v b_p)  with per-evaluation constant 3-vectors
  // a_p, b_p, hence  g_p = (R2^T e) . (u a_p + v b_p).  Rounding differs from the literal form in the last
  // bits; a frame that fits exactly gives 0 * inf = NaN exactly as the reference's 0 / 0.
  struct LmCoef {
    double c0[3], c1[3], t3[3], t1[3];
    double a[5][3], b[5][3];  // wz, wy, wx, m_x, m_y
  };
  static LSQR_HD void lm_coef(const double *xk, LmCoef &k) {
    const int o = SINGLE ? 3 : 0;
    for (int i = 0; i < 3; i++) k.t1[i] = SINGLE ? xk[i] : 0.0, k.t3[i] = xk[o + i];
    // lsqr_sincos: the same bits on the host and in the persistent kernel (small_linalg.h)
    double sz, cz, sy, cy, sx, cx;
    lsqr_sincos(xk[o + 3], &sz, &cz);
    lsqr_sincos(xk[o + 4], &sy, &cy);
    lsqr_sincos(xk[o + 5], &sx, &cx);
    const double m_x = xk[o + 6], m_y = xk[o + 7];
    const double r1[3] = {cz * cy, sz * cy, -sy};                                       // R3(:,1)
    const double r2[3] = {cz * sy * sx - sz * cx, sz * sy * sx + cz * cx, cy * sx};     // R3(:,2)
    for (int i = 0; i < 3; i++) k.c0[i] = m_x * r1[i], k.c1[i] = m_y * r2[i];
    // d/dwz (...Estimator.cxx:605-617), d/dwy (:619-635), d/dwx (:637-645), d/dm_x, d/dm_y (:647-656)
    const double az[3] = {-m_x * sz * cy, m_x * cz * cy, 0.0};
    const double bz[3] = {-m_y * (sz * sy * sx + cz * cx), m_y * (cz * sy * sx - sz * cx), 0.0};
    const double ay[3] = {-m_x * sy * cz, -m_x * sy * sz, -m_x * cy};
    const double by[3] = {m_y * sx * cy * cz, m_y * sx * cy * sz, -m_y * sx * sy};
    const double bx[3] = {m_y * (cz * sy * cx + sz * sx), m_y * (sz * sy * cx - cz * sx), m_y * cy * cx};
    for (int i = 0; i < 3; i++) {
      k.a[0][i] = az[i], k.b[0][i] = bz[i];
      k.a[1][i] = ay[i], k.b[1][i] = by[i];
      k.a[2][i] = 0.0, k.b[2][i] = bx[i];
      k.a[3][i] = r1[i], k.b[3][i] = 0.0;
      k.a[4][i] = 0.0, k.b[4][i] = r2[i];
    }
  }
  // one row of the iteration's (J | f) matrix: z[0..NLM) = J_p = g_p / |e|, z[NLM] = f = |e| -- what k_lm_pass_mfma
  // feeds to the matrix cores (sum z z^T holds J^T J, J^T f and sum f^2 at once); one root and one division
  static LSQR_HD void lm_row(const double *rec, const LmCoef &k, double *z) {
    const int o = SINGLE ? 3 : 0;
    const double u = rec[13], v = rec[14];
    double q[3], e[3], h[3];
    for (int i = 0; i < 3; i++) q[i] = fma(u, k.c0[i], fma(v, k.c1[i], k.t3[i]));
    for (int i = 0; i < 3; i++) {
      double s = rec[9 + i] - (SINGLE ? k.t1[i] : rec[15 + i]);
      s = fma(rec[3 * i + 2], q[2], s);
      s = fma(rec[3 * i + 1], q[1], s);
      e[i] = fma(rec[3 * i], q[0], s);
    }
    for (int j = 0; j < 3; j++) h[j] = fma(rec[j], e[0], fma(rec[3 + j], e[1], rec[6 + j] * e[2]));  // R2^T e
    const double ee = fma(e[0], e[0], fma(e[1], e[1], e[2] * e[2]));
    const double f = sqrt(ee), rs = 1.0 / f;
    if (SINGLE)
      for (int i = 0; i < 3; i++) z[i] = -e[i] * rs;
    for (int i = 0; i < 3; i++) z[o + i] = h[i] * rs;
    for (int p = 0; p < 5; p++) {
      const double ha = fma(h[0], k.a[p][0], fma(h[1], k.a[p][1], h[2] * k.a[p][2]));
      const double hb = fma(h[0], k.b[p][0], fma(h[1], k.b[p][1], h[2] * k.b[p][2]));
      z[o + 3 + p] = fma(u, ha, v * hb) * rs;
    }
    z[NLM] = f;
  }
  static LSQR_HD void accumulate_lm_fast(const double *rec, const LmCoef &k, double *m) {
    const int o = SINGLE ? 3 : 0;
    const double u = rec[13], v = rec[14];
    double q[3], e[3], h[3], g[NLM];
    for (int i = 0; i < 3; i++) q[i] = fma(u, k.c0[i], fma(v, k.c1[i], k.t3[i]));
    for (int i = 0; i < 3; i++) {
      double s = rec[9 + i] - (SINGLE ? k.t1[i] : rec[15 + i]);
      s = fma(rec[3 * i + 2], q[2], s);
      s = fma(rec[3 * i + 1], q[1], s);
      e[i] = fma(rec[3 * i], q[0], s);
    }
    for (int j = 0; j < 3; j++) h[j] = fma(rec[j], e[0], fma(rec[3 + j], e[1], rec[6 + j] * e[2]));  // R2^T e
    if (SINGLE)
      for (int i = 0; i < 3; i++) g[i] = -e[i];
    for (int i = 0; i < 3; i++) g[o + i] = h[i];
    for (int p = 0; p < 5; p++) {
      const double ha = fma(h[0], k.a[p][0], fma(h[1], k.a[p][1], h[2] * k.a[p][2]));
      const double hb = fma(h[0], k.b[p][0], fma(h[1], k.b[p][1], h[2] * k.b[p][2]));
      g[o + 3 + p] = fma(u, ha, v * hb);
    }
    const double ee = fma(e[0], e[0], fma(e[1], e[1], e[2] * e[2]));
    const double w = 1.0 / ee;
    m[0] += ee;
    int idx = 1;
    for (int p = 0; p < NLM; p++) {
      const double gw = g[p] * w;
      for (int r = p; r < NLM; r++, idx++) m[idx] = fma(gw, g[r], m[idx]);
    }
    for (int p = 0; p < NLM; p++, idx++) m[idx] += g[p];
  }

  // ...Estimator.cxx:300-327 / :944-971: append the rotation products to the LM solution
  static LSQR_HD int lm_finalize(const double *x, double *par) {
    const int o = SINGLE ? 3 : 0;
    for (int i = 0; i < NLM; i++) par[i] = x[i];
    double sz, cz, sy, cy, sx, cx;
    lsqr_sincos(x[o + 3], &sz, &cz);
    lsqr_sincos(x[o + 4], &sy, &cy);
    lsqr_sincos(x[o + 5], &sx, &cx);
    const double mx = x[o + 6], my = x[o + 7];
    int k = NLM;
    par[k++] = mx * cz * cy;
    par[k++] = mx * sz * cy;
    par[k++] = -mx * sy;
    par[k++] = my * (cz * sy * sx - sz * cx);
    par[k++] = my * (sz * sy * sx + cz * cx);
    par[k++] = my * cy * sx;
    par[k++] = cz * sy * cx + sz * sx;
    par[k++] = sz * sy * cx - cz * sx;
    par[k++] = cy * cx;
    return k;
  }
};

}  // namespace lsqr
