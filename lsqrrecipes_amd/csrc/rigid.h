// rigid.h -- the small geometric estimators of SURVEY.md section 8(f):
//   AbsOrModel   AbsoluteOrientationParametersEstimator      (parametersEstimators/AbsoluteOrientation...cxx)
//   PivotModel   PivotCalibrationEstimator                   (parametersEstimators/PivotCalibration...cxx)
//   RayModel     RayIntersectionParametersEstimator          (parametersEstimators/RayIntersection...cxx)
//   Line2DModel  Line2DParametersEstimator                   (parametersEstimators/Line2DParametersEstimator.cxx)
// Same contract as models.h: estimate() / agree() follow the reference's operation order (this TU is
// compiled with -ffp-contract=off); the final fits reduce shifted moments / normal equations in one
// pass and solve them with the small dense kernels.  Frame arithmetic restated from common/Frame.cxx.
#pragma once
#include "models.h"

namespace lsqr {

// common/Frame.cxx:750-771 (setRotationQuaternion) -- R row-major
LSQR_HD void frame_from_quaternion(double s, double qx, double qy, double qz, bool normalize,
                                   double *R) {
  if (normalize) {
    double norm = sqrt(s * s + qx * qx + qy * qy + qz * qz);
    s /= norm;
    qx /= norm;
    qy /= norm;
    qz /= norm;
  }
  R[0] = 1 - 2 * (qy * qy + qz * qz);
  R[1] = 2 * (qx * qy - s * qz);
  R[2] = 2 * (qx * qz + s * qy);
  R[3] = 2 * (qx * qy + s * qz);
  R[4] = 1 - 2 * (qx * qx + qz * qz);
  R[5] = 2 * (qy * qz - s * qx);
  R[6] = 2 * (qx * qz - s * qy);
  R[7] = 2 * (qy * qz + s * qx);
  R[8] = 1 - 2 * (qx * qx + qy * qy);
}

// common/Frame.cxx:952-991 (getRotationQuaternion)
LSQR_HD void frame_quaternion(const double *R, double *q) {
  const double smallAngle = 0.008726535498373935, halfPI = 3.14159265358979323846 / 2.0;
  const double startSingularRange = halfPI - smallAngle, endSingularRange = halfPI + smallAngle;
  q[0] = (0.5 * sqrt(R[0] + R[4] + R[8] + 1));
  double halfTheta = acos(q[0]);
  if (!(halfTheta > startSingularRange && halfTheta < endSingularRange)) {
    double denom = 4 * q[0];
    q[1] = (R[7] - R[5]) / denom;
    q[2] = (R[2] - R[6]) / denom;
    q[3] = (R[3] - R[1]) / denom;
  } else {
    int i = 0;
    if (R[4] > R[4 * i]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double w = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1);
    q[i + 1] = w / 2.0;
    q[j + 1] = (R[3 * i + j] + R[3 * j + i]) / (2 * w);
    q[k + 1] = (R[3 * i + k] + R[3 * k + i]) / (2 * w);
  }
}

// ------------------------------------------------------------------------ absolute orientation
// record: std::pair<Point3D,Point3D> = [first(3), second(3)]; parameters [s,qx,qy,qz,tx,ty,tz];
// scan parameters: the 7 + the 9 rotation entries agree() would rebuild per datum
// (Frame ctor, Frame.cxx:174-198, no normalisation).
struct AbsOrModel {
  // REC = 7: slot 6 is the pair's weight -- read from the record when ls_type == 2 (weighted fit,
  // AbsoluteOrientation...cxx:208-291; records of 7 doubles), 1.0 otherwise (records of 6 doubles)
  enum { ND = 6, K = 3, P = 7, SP = 16, REC = 7, PPL = 4, IS_DENSE = 0, IS_US = 0, ORIGIN_FIRST = 1 };
  enum { NMOM = 1 + 3 + 3 + 9 };
  static LSQR_HD void load(const double *p, const ModelConsts &c, double *rec) {
    for (int i = 0; i < 6; i++) rec[i] = p[i];
    rec[6] = c.ls_type == 2 ? p[6] : 1.0;
  }
  // vnl_vector::normalize(): multiply by 1/sqrt(sum of squares) unless the sum is zero
  static LSQR_HD void normalize3(double *v) {
    double tmp = 0;
    for (int i = 0; i < 3; i++) tmp += v[i] * v[i];
    if (tmp != 0) {
      tmp = 1.0 / sqrt(tmp);
      for (int i = 0; i < 3; i++) v[i] = tmp * v[i];
    }
  }
  // AbsoluteOrientation...cxx:25-50 / :57-79; R columns = x, y, z axes (row-major storage)
  static LSQR_HD bool triad(const double *p0, const double *p1, const double *p2, double *mean,
                            double *R) {
    double x[3], y[3], z[3], d;
    for (int i = 0; i < 3; i++) mean[i] = (p0[i] + p1[i] + p2[i]) / 3.0;
    for (int i = 0; i < 3; i++) x[i] = p0[i] - mean[i];
    normalize3(x);
    for (int i = 0; i < 3; i++) y[i] = p1[i] - mean[i];
    d = 0;
    for (int i = 0; i < 3; i++) d += y[i] * x[i];
    for (int i = 0; i < 3; i++) y[i] = y[i] - d * x[i];
    normalize3(y);
    z[0] = x[1] * y[2] - x[2] * y[1];
    z[1] = x[2] * y[0] - x[0] * y[2];
    z[2] = x[0] * y[1] - x[1] * y[0];
    d = 0;
    for (int i = 0; i < 3; i++) d += z[i] * z[i];
    if (sqrt(d) < kEPS) return false;  // collinear (:48, :78)
    for (int i = 0; i < 3; i++) {
      R[3 * i] = x[i];
      R[3 * i + 1] = y[i];
      R[3 * i + 2] = z[i];
    }
    return true;
  }
  // AbsoluteOrientation...cxx:14-105
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &, double *par) {
    double R1[9], R2[9], R[9], m1[3], m2[3];
    if (!triad(r[0], r[1], r[2], m1, R1)) return false;
    if (!triad(r[0] + 3, r[1] + 3, r[2] + 3, m2, R2)) return false;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {  // secondR * firstR^T (:86), running sum from 0
        double sum = 0;
        for (int k = 0; k < 3; k++) sum += R2[3 * i + k] * R1[3 * j + k];
        R[3 * i + j] = sum;
      }
    frame_quaternion(R, par);
    for (int i = 0; i < 3; i++) {  // t = meanSecond - R*meanFirst (:88)
      double sum = 0;
      for (int k = 0; k < 3; k++) sum += R[3 * i + k] * m1[k];
      par[4 + i] = m2[i] - sum;
    }
    return true;
  }
  static LSQR_HD void prepare(double *sp, const ModelConsts &) {
    frame_from_quaternion(sp[0], sp[1], sp[2], sp[3], false, sp + 7);
  }
  // AbsoluteOrientation...cxx:316-327 (Frame::apply, Frame.cxx:229-247)
  static LSQR_HD double dist_sq(const double *sp, const double *x) {
    const double *R = sp + 7;
    double px = R[0] * x[0] + R[1] * x[1] + R[2] * x[2] + sp[4];
    double py = R[3] * x[0] + R[4] * x[1] + R[5] * x[2] + sp[5];
    double pz = R[6] * x[0] + R[7] * x[1] + R[8] * x[2] + sp[6];
    double dx = px - x[3], dy = py - x[4], dz = pz - x[5];
    return (dx * dx + dy * dy + dz * dz);
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return dist_sq(sp, x) < c.delta_sq;
  }
  static LSQR_HD double residual(const double *par, const double *x, const ModelConsts &) {
    double sp[SP];
    for (int i = 0; i < P; i++) sp[i] = par[i];
    frame_from_quaternion(sp[0], sp[1], sp[2], sp[3], false, sp + 7);
    return sqrt(dist_sq(sp, x));
  }
  // moments about org = (first, second) of the first datum: {sum w, sum w l', sum w r', sum w l' r'^T}
  // (AbsoluteOrientation...cxx:133-168 accumulates the same sums un-shifted with w = 1, :223-258 with the
  // caller's weights; multiplying by w = 1.0 is exact, so the unweighted block is unchanged)
  static LSQR_HD void accumulate(const double *x, const double *org, double *m) {
    double l[3], r[3];
    const double w = x[6];
    for (int i = 0; i < 3; i++) {
      l[i] = x[i] - org[i];
      r[i] = x[3 + i] - org[3 + i];
    }
    m[0] += w;
    for (int i = 0; i < 3; i++) {
      const double wl = w * l[i];
      m[1 + i] += wl;
      m[4 + i] += w * r[i];
      for (int j = 0; j < 3; j++) m[7 + 3 * i + j] = fma(wl, r[j], m[7 + 3 * i + j]);
    }
  }
  // Horn's closed form (AbsoluteOrientation...cxx:170-198): largest eigenvector of the 4x4 N
  // (weighted: n = sum of the weights; the caller has checked the pair count, :213-216)
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &c, double *par) {
    const double n = m[0];
    if (c.ls_type == 2 ? !(n > 0.0) : n < 3.0) return false;  // :129
    double ml[3], mr[3], M[9], Nm[16], w[4], V[16], R[9];
    for (int i = 0; i < 3; i++) {
      ml[i] = m[1 + i] / n;
      mr[i] = m[4 + i] / n;
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) M[3 * i + j] = m[7 + 3 * i + j] - n * ml[i] * mr[j];
    const double tr = M[0] + M[4] + M[8];
    const double A12 = M[5] - M[7], A20 = M[6] - M[2], A01 = M[1] - M[3];
    Nm[0] = tr, Nm[1] = A12, Nm[2] = A20, Nm[3] = A01;
    Nm[4] = A12, Nm[8] = A20, Nm[12] = A01;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        Nm[(i + 1) * 4 + (j + 1)] = (i == j ? -tr : 0.0) + (M[3 * i + j] + M[3 * j + i]);
    sym_eig(4, Nm, w, V);
    for (int i = 0; i < 4; i++) par[i] = V[i * 4 + 3];
    frame_from_quaternion(par[0], par[1], par[2], par[3], true, R);
    for (int i = 0; i < 3; i++) {
      double rot = 0;
      for (int k = 0; k < 3; k++) rot += R[3 * i + k] * (ml[k] + org[k]);
      par[4 + i] = (mr[i] + org[3 + i]) - rot;
    }
    return true;
  }
};

// ------------------------------------------------------------------------ pivot calibration
// record: Frame = rotation[3][3] (slots 0..8), translation (9..11), int outputFormat + pad (12);
// parameters [DRF^t(3), W^t(3)].
struct PivotModel {
  enum { ND = 13, K = 3, P = 6, SP = 6, REC = 12, PPL = 2, IS_DENSE = 0, IS_US = 0, ORIGIN_FIRST = 1 };
  // normal equations of rows [R_i, -I], rhs -t_i:
  //   {N, sum R^T R (upper 6), sum R (9), sum R^T t (3), sum t (3)}
  enum { NMOM = 1 + 6 + 9 + 3 + 3 };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < 12; i++) rec[i] = p[i];
  }
  // PivotCalibration...cxx:9-50: 9x6 pseudo-inverse, singular values <= EPS zeroed, rank < 6 -> empty
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &, double *par) {
    double A[9 * 6], b[9], s[6], v[36];
    for (int i = 0; i < 3; i++)
      for (int a = 0; a < 3; a++) {
        for (int c = 0; c < 3; c++) A[(3 * i + a) * 6 + c] = r[i][3 * a + c];
        for (int c = 0; c < 3; c++) A[(3 * i + a) * 6 + 3 + c] = (a == c) ? -1.0 : 0.0;
        b[3 * i + a] = -r[i][9 + a];
      }
    return pinv_solve(9, 6, A, 6, b, kEPS, par, s, v) == 6;
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}
  // PivotCalibration...cxx:109-123 (Frame::apply in place, Frame.cxx:208-227; l2Norm)
  static LSQR_HD double dist(const double *sp, const double *f) {
    double x = f[0] * sp[0] + f[1] * sp[1] + f[2] * sp[2] + f[9];
    double y = f[3] * sp[0] + f[4] * sp[1] + f[5] * sp[2] + f[10];
    double z = f[6] * sp[0] + f[7] * sp[1] + f[8] * sp[2] + f[11];
    double dx = x - sp[3], dy = y - sp[4], dz = z - sp[5];
    return sqrt(dx * dx + dy * dy + dz * dz);
  }
  static LSQR_HD bool agree(const double *sp, const double *f, const ModelConsts &c) {
    return dist(sp, f) < c.delta;
  }
  static LSQR_HD double residual(const double *par, const double *f, const ModelConsts &) {
    return dist(par, f);
  }
  static LSQR_HD void accumulate(const double *f, const double *, double *m) {
    m[0] += 1.0;
    int q = 1;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++, q++) {  // (R^T R)_ab = sum_k R_ka R_kb
        double t = f[a] * f[b];
        t = fma(f[3 + a], f[3 + b], t);
        t = fma(f[6 + a], f[6 + b], t);
        m[q] += t;
      }
    for (int i = 0; i < 9; i++) m[7 + i] += f[i];
    for (int a = 0; a < 3; a++) {  // (R^T t)_a
      double t = f[a] * f[9];
      t = fma(f[3 + a], f[10], t);
      t = fma(f[6 + a], f[11], t);
      m[16 + a] += t;
    }
    for (int a = 0; a < 3; a++) m[19 + a] += f[9 + a];
  }
  // PivotCalibration...cxx:63-96 through the 6x6 normal equations  A^T A x = A^T b:
  //   A^T A = [[sum R^T R, -sum R^T], [-sum R, N I]],  A^T b = [-sum R^T t, sum t]
  static LSQR_HD bool solve(const double *m, const double *, const ModelConsts &, double *par) {
    const double n = m[0];
    if (n < 3.0) return false;  // :69
    double G[36], rhs[6], work[2 * 36 + 18];
    int q = 1;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++, q++) G[a * 6 + b] = G[b * 6 + a] = m[q];
    for (int k = 0; k < 3; k++)
      for (int a = 0; a < 3; a++) {  // -(sum R)_{k a}: block (3+k, a) and its transpose
        G[(3 + k) * 6 + a] = -m[7 + 3 * k + a];
        G[a * 6 + 3 + k] = -m[7 + 3 * k + a];
      }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) G[(3 + a) * 6 + 3 + b] = (a == b) ? n : 0.0;
    for (int a = 0; a < 3; a++) {
      rhs[a] = -m[16 + a];
      rhs[3 + a] = m[19 + a];
    }
    // the reference declares rank deficiency for singular values <= 2.2e-16 (absolute, :88-91);
    // on the normal equations that is sigma^2: a relative 1e-13 as for the dense system
    return spd_solve_eig(6, G, rhs, 1e-13, par, work) == 6;
  }
};

// ------------------------------------------------------------------------ 2-D line (normal form)
// Line2DParametersEstimator: parameters [n_x, n_y, a_x, a_y]; agree() is the 2-D hyperplane test
// (.cxx:117-121 == PlaneParametersEstimator.hxx:196-203), so the scan -- exhaustive kernels, fp32 filter,
// two-level cell scan -- is PlaneModel<2>'s; estimate() and the closed-form fit are its own.
struct Line2DModel : PlaneModel<2> {
  // .cxx:9-27: normal (y1 - y0, x0 - x1); "too close" when its squared length is below delta^2
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &c, double *par) {
    double nx = r[1][1] - r[0][1];
    double ny = r[0][0] - r[1][0];
    double normSquared = nx * nx + ny * ny;
    if (normSquared < c.delta_sq) return false;
    double norm = sqrt(nx * nx + ny * ny);
    par[0] = nx / norm;
    par[1] = ny / norm;
    par[2] = r[0][0];
    par[3] = r[0][1];
    return par[0] == par[0] && par[1] == par[1];  // NaN / inf points give no model
  }
  // .cxx:44-100 on shifted moments {N, sum x', sum x'x'^T}: closed-form eigenvector of the 2x2 covariance,
  // with the reference's 1e-12 thresholds on the diagonal entries
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &, double *par) {
    const double N = m[0];
    if (N < 2.0) return false;
    const double mx = m[1] / N, my = m[2] / N;
    const double c11 = m[3] - N * mx * mx, c12 = m[4] - N * mx * my, c22 = m[5] - N * my * my;
    double nx, ny;
    if (c11 < 1e-12) {
      nx = 1.0;
      ny = 0.0;
      if (c22 < 1e-12) return false;  // all the points are the "same" point
    } else {
      double lambda1 = (c11 + c22 + sqrt((c11 - c22) * (c11 - c22) + 4 * c12 * c12)) / 2.0;
      nx = -c12;
      ny = lambda1 - c22;
      double norm = sqrt(nx * nx + ny * ny);
      nx /= norm;
      ny /= norm;
    }
    par[0] = nx;
    par[1] = ny;
    par[2] = mx + org[0];
    par[3] = my + org[1];
    return true;
  }
};

// ------------------------------------------------------------------------ ray intersection
// record: Ray3D = [p(3), n(3)] (common/Ray3D.h:23-24), r(t) = p + t n, t >= 0, |n| = 1 assumed by the
// reference; parameters [x, y, z].
struct RayModel {
  enum { ND = 6, K = 2, P = 3, SP = 3, REC = 6, PPL = 4, IS_DENSE = 0, IS_US = 0, ORIGIN_FIRST = 1 };
  // {N, sum n n^T (upper 6), sum (p - (n.p) n) (3)}
  enum { NMOM = 1 + 6 + 3 };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < 6; i++) rec[i] = p[i];
  }
  // RayIntersection...Estimator.cxx:23-72 (Goldman, Graphics Gems p.304): mid-point of the common
  // perpendicular; empty for (nearly) parallel rays (:51) and when a line parameter is negative (:63)
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &c, double *par) {
    const double *p1 = r[0], *n1 = r[0] + 3, *p2 = r[1], *n2 = r[1] + 3;
    double p21[3], x[3];
    p21[0] = p2[0] - p1[0];
    p21[1] = p2[1] - p1[1];
    p21[2] = p2[2] - p1[2];
    x[0] = n1[1] * n2[2] - n1[2] * n2[1];
    x[1] = n1[2] * n2[0] - n1[0] * n2[2];
    x[2] = n1[0] * n2[1] - n1[1] * n2[0];
    double denominator = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    if (denominator < c.aux) return false;
    double t1 = (x[0] * (p21[1] * n2[2] - p21[2] * n2[1]) - x[1] * (p21[0] * n2[2] - p21[2] * n2[0]) +
                 x[2] * (p21[0] * n2[1] - p21[1] * n2[0])) / denominator;
    double t2 = (x[0] * (p21[1] * n1[2] - p21[2] * n1[1]) - x[1] * (p21[0] * n1[2] - p21[2] * n1[0]) +
                 x[2] * (p21[0] * n1[1] - p21[1] * n1[0])) / denominator;
    if (t1 < 0 || t2 < 0) return false;
    par[0] = (p1[0] + t1 * n1[0] + p2[0] + t2 * n2[0]) / 2.0;
    par[1] = (p1[1] + t1 * n1[1] + p2[1] + t2 * n2[1]) / 2.0;
    par[2] = (p1[2] + t1 * n1[2] + p2[2] + t2 * n2[2]) / 2.0;
    return par[0] == par[0] && par[1] == par[1] && par[2] == par[2];  // NaN rays give no model
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}
  // RayIntersection...Estimator.cxx:163-177
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    const double *p = x, *n = x + 3;
    double t = n[0] * (sp[0] - p[0]) + n[1] * (sp[1] - p[1]) + n[2] * (sp[2] - p[2]);
    double dx = sp[0] - p[0] - t * n[0];
    double dy = sp[1] - p[1] - t * n[1];
    double dz = sp[2] - p[2] - t * n[2];
    return t >= 0 && (dx * dx + dy * dy + dz * dz < c.delta_sq);
  }
  // distance of the point from the ray's line (Ray3D::distance)
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &) {
    const double *p = x, *n = x + 3;
    double t = n[0] * (sp[0] - p[0]) + n[1] * (sp[1] - p[1]) + n[2] * (sp[2] - p[2]);
    double dx = sp[0] - p[0] - t * n[0], dy = sp[1] - p[1] - t * n[1], dz = sp[2] - p[2] - t * n[2];
    return sqrt(dx * dx + dy * dy + dz * dz);
  }
  // RayIntersection...Estimator.cxx:103-127, ray origins taken about org (first ray's origin):
  //   [N I - sum n n^T] (x - org) = sum (p' - (n.p') n),  p' = p - org
  static LSQR_HD void accumulate(const double *x, const double *org, double *m) {
    const double *n = x + 3;
    double q[3] = {x[0] - org[0], x[1] - org[1], x[2] - org[2]};
    m[0] += 1.0;
    int k = 1;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++, k++) m[k] = fma(n[a], n[b], m[k]);
    double s = n[0] * q[0] + n[1] * q[1] + n[2] * q[2];
    for (int a = 0; a < 3; a++) m[7 + a] += q[a] - s * n[a];
  }
  // :129-143: pseudo-inverse of the 3x3 system, rank < 3 (all rays parallel) -> empty.  The reference
  // needs >= 0 rays only; an empty or single-ray set is rank deficient and comes out empty as well.
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &, double *par) {
    const double N = m[0];
    double A[9], w[3], V[9], y[3];
    int k = 1;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++, k++) A[a * 3 + b] = A[b * 3 + a] = (a == b ? N : 0.0) - m[k];
    double amax = 0.0;
    for (int i = 0; i < 9; i++) amax = fabs(A[i]) > amax ? fabs(A[i]) : amax;
    sym_eig(3, A, w, V);
    // singular values of the symmetric A are |w|; zero_out_absolute(EPS) drops those <= EPS
    for (int j = 0; j < 3; j++)
      if (!(fabs(w[j]) > kEPS) || !(fabs(w[j]) > 1e-14 * amax)) return false;
    for (int j = 0; j < 3; j++) {
      double t = 0;
      for (int i = 0; i < 3; i++) t += V[i * 3 + j] * m[7 + i];
      y[j] = t / w[j];
    }
    for (int i = 0; i < 3; i++) {
      double t = 0;
      for (int j = 0; j < 3; j++) t += V[i * 3 + j] * y[j];
      par[i] = t + org[i];
    }
    return true;
  }
};

}  // namespace lsqr
