/*
 * estimators.c -- oracle (TEST INFRASTRUCTURE ONLY, see lsqr_oracle.h): the five hot-path
 * estimators of /root/reference/parametersEstimators (and, SURVEY.md section 8f, the
 * AbsoluteOrientation and PivotCalibration estimators) restated in plain C, fp64, operation
 * order preserved (compile with -ffp-contract=off, no -ffast-math).
 */
#include "lsqr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double EPS = 2.220446049250313e-016; /* common/Epsilon.h:19 */
static const double SPHERE_EPS = 1e-9;            /* SphereParametersEstimator.hxx:11 */
static const double US_SV_EPS = 1.192092896e-07;  /* SinglePointTarget...cxx:196,843 */

/* US record slots (doubles): Frame::rotation[3][3] 0..8, translation 9..11, (int outputFormat
 * + padding) 12, Point2D q 13..14, Point3D p 15..17 (common/Frame.h:30-31,41;
 * SinglePointTarget...h:45-48,335-339) */
enum { US_T2 = 9, US_Q = 13, US_P = 15 };

int orc_min_subset(const orc_cfg *c) {
  switch (c->model) {
    case ORC_PLANE: return c->dim;      /* PlaneParametersEstimator.hxx:14 */
    case ORC_SPHERE: return c->dim + 1; /* SphereParametersEstimator.hxx:15 */
    case ORC_LINE: return 2;            /* LineParametersEstimator.hxx:14 */
    case ORC_DENSE: return c->dim;      /* DenseLinear...hxx:11 */
    case ORC_US_SINGLE: return 4;       /* SinglePointTarget...cxx:11 */
    case ORC_US_POINTER: return 3;      /* SinglePointTarget...cxx:665 */
    case ORC_ABSOR: return 3;           /* AbsoluteOrientation...cxx:9 */
    case ORC_PIVOT: return 3;           /* PivotCalibration...cxx:7 */
    case ORC_RAY: return 2;             /* RayIntersection...cxx:11 */
    case ORC_LINE2D: return 2;          /* Line2DParametersEstimator.cxx:6 */
    case ORC_PHANTOM: return 31;        /* PlanePhantomUSCalibration...cxx:10 */
  }
  return 0;
}
int orc_num_params(const orc_cfg *c) {
  switch (c->model) {
    case ORC_PLANE: return 2 * c->dim;
    case ORC_SPHERE: return c->dim + 1;
    case ORC_LINE: return 2 * c->dim;
    case ORC_DENSE: return c->dim;
    case ORC_US_SINGLE: return 20;
    case ORC_US_POINTER: return 17;
    case ORC_ABSOR: return 7; /* [s,qx,qy,qz,tx,ty,tz] */
    case ORC_PIVOT: return 6; /* [DRF^t, W^t] */
    case ORC_RAY: return 3;   /* [x,y,z] */
    case ORC_LINE2D: return 4; /* [n_x,n_y,a_x,a_y] */
    case ORC_PHANTOM: return 41;
  }
  return 0;
}
int orc_record_doubles(const orc_cfg *c) {
  switch (c->model) {
    case ORC_PLANE:
    case ORC_SPHERE:
    case ORC_LINE: return c->dim;
    case ORC_DENSE: return c->dim + 1;
    case ORC_US_SINGLE: return 15;
    case ORC_US_POINTER: return 18;
    case ORC_ABSOR: return 6;  /* std::pair<Point3D,Point3D> */
    case ORC_PIVOT: return 13; /* Frame: rotation 9, translation 3, int outputFormat + pad */
    case ORC_RAY: return 6;    /* Ray3D: Point3D p, Vector3D n (common/Ray3D.h:23-24) */
    case ORC_LINE2D: return 2; /* Point2D */
    case ORC_PHANTOM: return 15; /* Frame + Point2D, as ORC_US_SINGLE */
  }
  return 0;
}

/* ================================================================== plane */
/* PlaneParametersEstimator.hxx:36-109 */
static int plane_estimate(int d, const double *const *p, size_t n, double *out) {
  int i, j;
  if (d == 0 || n < (size_t)d) return 0;
  if (d == 3) {
    double v1[3], v2[3], nx, ny, nz, norm;
    v1[0] = p[1][0] - p[0][0];
    v1[1] = p[1][1] - p[0][1];
    v1[2] = p[1][2] - p[0][2];
    v2[0] = p[2][0] - p[0][0];
    v2[1] = p[2][1] - p[0][1];
    v2[2] = p[2][2] - p[0][2];
    nx = v1[1] * v2[2] - v1[2] * v2[1];
    ny = v1[2] * v2[0] - v1[0] * v2[2];
    nz = v1[0] * v2[1] - v1[1] * v2[0];
    norm = sqrt(nx * nx + ny * ny + nz * nz);
    if (norm < EPS) return 0;
    out[0] = nx / norm;
    out[1] = ny / norm;
    out[2] = nz / norm;
  } else {
    /* :70-104 null vector of the d x (d+1) matrix [p,-1] via SVD, rank decided on the singular values at
     * EPS (zero_out_absolute).  The matrix is padded with a zero row to (d+1) x (d+1): same null space,
     * same non-zero singular values, and d+1 singular values in all -- what vnl_svd reports for a
     * d x (d+1) matrix (LINPACK dsvdc yields min(m+1, n) of them).  nullvector() is the right singular
     * vector of the smallest one.  (VNL absent: bit-level parity of this branch is unpinned.) */
    int m = d + 1;
    double *A = (double *)calloc((size_t)m * m, sizeof(double));
    double *U = (double *)malloc(sizeof(double) * m * m);
    double *w = (double *)malloc(sizeof(double) * m);
    double *V = (double *)malloc(sizeof(double) * m * m);
    double norm = 0;
    int rank = 0;
    for (i = 0; i < d; i++) {
      for (j = 0; j < d; j++) A[i * m + j] = p[i][j];
      A[i * m + d] = -1;
    }
    orc_svd(m, m, A, U, w, V); /* singular values descending */
    for (i = 0; i < m; i++)
      if (w[i] > EPS) rank++;
    if (rank >= d)
      for (i = 0; i < d; i++) {
        out[i] = V[i * m + (m - 1)];
        norm += out[i] * out[i];
      }
    free(A);
    free(U);
    free(w);
    free(V);
    if (rank < d || !(norm > 0)) return 0;
    norm = 1.0 / sqrt(norm);
    for (i = 0; i < d; i++) out[i] *= norm;
  }
  for (i = 0; i < d; i++) out[d + i] = p[0][i]; /* :107-108 */
  return 2 * d;
}

/* PlaneParametersEstimator.hxx:196-203 */
static int plane_agree(int d, double delta_sq, const double *par, const double *x) {
  double s = 0;
  int i;
  for (i = 0; i < d; i++) s += par[i] * (x[i] - par[d + i]);
  return (s * s) < delta_sq;
}

/* shared by plane (:129-172, smallest eigenvector) and line (LineParametersEstimator.hxx:
 * 68-111, largest eigenvector) */
static int cov_ls(int d, const double *const *p, size_t n, int min_n, int largest,
                  double *out) {
  double *mean, *meanMat, *cov, *w, *V, sqrtN;
  size_t i;
  int j, k;
  if (n < (size_t)min_n) return 0;
  mean = (double *)calloc(d, sizeof(double));
  meanMat = (double *)calloc((size_t)d * d, sizeof(double));
  cov = (double *)calloc((size_t)d * d, sizeof(double));
  w = (double *)malloc(sizeof(double) * d);
  V = (double *)malloc(sizeof(double) * d * d);
  sqrtN = sqrt((double)n);
  for (i = 0; i < n; i++)
    for (j = 0; j < d; j++) mean[j] += p[i][j];
  for (j = 0; j < d; j++) mean[j] /= sqrtN;
  for (j = 0; j < d; j++)
    for (k = j; k < d; k++) meanMat[j * d + k] = meanMat[k * d + j] = mean[j] * mean[k];
  for (i = 0; i < n; i++)
    for (j = 0; j < d; j++)
      for (k = j; k < d; k++) cov[j * d + k] += p[i][j] * p[i][k];
  for (j = 0; j < d; j++)
    for (k = j + 1; k < d; k++) cov[k * d + j] = cov[j * d + k];
  for (j = 0; j < d * d; j++) cov[j] -= meanMat[j];
  orc_sym_eig(d, cov, w, V);
  for (j = 0; j < d; j++) out[j] = V[j * d + (largest ? d - 1 : 0)];
  for (j = 0; j < d; j++) out[d + j] = mean[j] / sqrtN;
  free(mean);
  free(meanMat);
  free(cov);
  free(w);
  free(V);
  return 2 * d;
}

/* ================================================================== line */
/* LineParametersEstimator.hxx:23-48 */
static int line_estimate(int d, double delta_sq, const double *const *p, size_t n,
                         double *out) {
  double dist = 0, dirNorm = 0;
  int i;
  if (n < 2) return 0;
  /* Point::distanceSquared = squared_magnitude of the difference (common/Point.h:97-99) */
  for (i = 0; i < d; i++) dist += (p[0][i] - p[1][i]) * (p[0][i] - p[1][i]);
  if (dist < delta_sq) return 0;
  for (i = 0; i < d; i++) {
    out[i] = p[0][i] - p[1][i];
    dirNorm += out[i] * out[i];
    out[d + i] = p[0][i];
  }
  dirNorm = sqrt(dirNorm);
  for (i = 0; i < d; i++) out[i] /= dirNorm;
  return 2 * d;
}

/* LineParametersEstimator.hxx:135-150 */
static int line_agree(int d, double delta_sq, const double *par, const double *x) {
  double v[64], vDotN = 0, dist = 0;
  int i;
  for (i = 0; i < d; i++) {
    v[i] = x[i] - par[d + i];
    vDotN += v[i] * par[i];
  }
  for (i = 0; i < d; i++) dist += (v[i] - vDotN * par[i]) * (v[i] - vDotN * par[i]);
  return dist < delta_sq;
}

/* ================================================================== sphere */
/* SphereParametersEstimator.hxx:80-109 */
static int sphere_estimate2d(const double *const *p, double *out) {
  const double *p0 = p[0], *p1 = p[1], *p2 = p[2];
  double A00 = p0[0] - p1[0], A01 = p0[1] - p1[1];
  double A10 = p0[0] - p2[0], A11 = p0[1] - p2[1];
  double detA = (A00 * A11 - A01 * A10), b0, b1;
  if (fabs(detA) < SPHERE_EPS) return 0;
  detA *= 2.0;
  b0 = A00 * (p0[0] + p1[0]) + A01 * (p0[1] + p1[1]);
  b1 = A10 * (p0[0] + p2[0]) + A11 * (p0[1] + p2[1]);
  out[0] = (A11 * b0 - A01 * b1) / detA;
  out[1] = (A00 * b1 - A10 * b0) / detA;
  out[2] = sqrt((p0[0] - out[0]) * (p0[0] - out[0]) + (p0[1] - out[1]) * (p0[1] - out[1]));
  return 3;
}

/* SphereParametersEstimator.hxx:115-163 */
static int sphere_estimate3d(const double *const *p, double *out) {
  const double *p0 = p[0], *p1 = p[1], *p2 = p[2], *p3 = p[3];
  double A00, A01, A02, A10, A11, A12, A20, A21, A22;
  double CT00, CT01, CT02, CT10, CT11, CT12, CT20, CT21, CT22, b0, b1, b2, detA;
  A00 = p0[0] - p1[0]; A01 = p0[1] - p1[1]; A02 = p0[2] - p1[2];
  A10 = p0[0] - p2[0]; A11 = p0[1] - p2[1]; A12 = p0[2] - p2[2];
  A20 = p0[0] - p3[0]; A21 = p0[1] - p3[1]; A22 = p0[2] - p3[2];
  CT00 = A11 * A22 - A12 * A21;
  CT10 = A12 * A20 - A10 * A22;
  CT20 = A10 * A21 - A11 * A20;
  detA = A00 * CT00 + A01 * CT10 + A02 * CT20;
  if (fabs(detA) < SPHERE_EPS) return 0;
  detA *= 2;
  CT01 = A02 * A21 - A01 * A22;
  CT11 = A00 * A22 - A02 * A20;
  CT21 = A01 * A20 - A00 * A21;
  CT02 = A01 * A12 - A02 * A11;
  CT12 = A02 * A10 - A00 * A12;
  CT22 = A00 * A11 - A01 * A10;
  b0 = A00 * (p0[0] + p1[0]) + A01 * (p0[1] + p1[1]) + A02 * (p0[2] + p1[2]);
  b1 = A10 * (p0[0] + p2[0]) + A11 * (p0[1] + p2[1]) + A12 * (p0[2] + p2[2]);
  b2 = A20 * (p0[0] + p3[0]) + A21 * (p0[1] + p3[1]) + A22 * (p0[2] + p3[2]);
  out[0] = (CT00 * b0 + CT01 * b1 + CT02 * b2) / detA;
  out[1] = (CT10 * b0 + CT11 * b1 + CT12 * b2) / detA;
  out[2] = (CT20 * b0 + CT21 * b1 + CT22 * b2) / detA;
  out[3] = sqrt(((p0[0] - out[0]) * (p0[0] - out[0])) + ((p0[1] - out[1]) * (p0[1] - out[1])) +
                ((p0[2] - out[2]) * (p0[2] - out[2])));
  return 4;
}

/* SphereParametersEstimator.hxx:169-202 */
static int sphere_estimateNd(int d, const double *const *p, double *out) {
  double *A = (double *)malloc(sizeof(double) * d * d);
  double *b = (double *)calloc(d, sizeof(double));
  double *x = (double *)malloc(sizeof(double) * d), r2 = 0;
  int i, j, rank;
  for (i = 0; i < d; i++)
    for (j = 0; j < d; j++) {
      A[i * d + j] = p[0][j] - p[i + 1][j];
      b[i] += A[i * d + j] * (p[0][j] + p[i + 1][j]);
    }
  rank = orc_pinv_solve(d, d, A, b, EPS, x);
  if (rank == d)
    for (i = 0; i < d; i++) {
      out[i] = x[i] * 0.5; /* x = Ainv*b*0.5 */
      r2 += (p[0][i] - out[i]) * (p[0][i] - out[i]);
    }
  free(A);
  free(b);
  free(x);
  if (rank < d) return 0;
  out[d] = sqrt(r2);
  return d + 1;
}

/* SphereParametersEstimator.hxx:255-264 */
static int sphere_agree(int d, double delta, const double *par, const double *x) {
  double s = 0;
  int i;
  for (i = 0; i < d; i++) s += ((x[i] - par[i]) * (x[i] - par[i]));
  s = fabs(sqrt(s) - par[d]);
  return s < delta;
}

/* SphereParametersEstimator.hxx:267-307 */
int orc_sphere_algebraic(int d, const double *const *p, size_t n, double *out) {
  int m = d + 1, j, rank;
  size_t i;
  double *A, *b, *x, r2;
  if (n < (size_t)m) return 0;
  A = (double *)malloc(sizeof(double) * n * m);
  b = (double *)calloc(n, sizeof(double));
  x = (double *)malloc(sizeof(double) * m);
  for (i = 0; i < n; i++) {
    for (j = 0; j < d; j++) {
      A[i * m + j] = -2 * p[i][j];
      b[i] += -(p[i][j] * p[i][j]);
    }
    A[i * m + d] = 1;
  }
  rank = orc_pinv_solve((int)n, m, A, b, EPS, x);
  free(A);
  free(b);
  if (rank < m) {
    free(x);
    return 0;
  }
  r2 = -x[d];
  for (j = 0; j < d; j++) {
    out[j] = x[j];
    r2 += x[j] * x[j];
  }
  free(x);
  if (!(r2 > 0)) return 0;
  out[d] = sqrt(r2);
  return m;
}

typedef struct { int d; const double *const *p; } sphere_lm_ctx;

/* f: SphereParametersEstimator.hxx:394-409; gradf: :413-431 */
static void sphere_lm_fcn(void *vctx, int m, int n, const double *x, double *fvec,
                          double *fjac, int iflag) {
  sphere_lm_ctx *c = (sphere_lm_ctx *)vctx;
  int d = c->d, i, j;
  for (i = 0; i < m; i++) {
    const double *pt = c->p[i];
    double sq = 0.0;
    for (j = 0; j < d; j++) sq += (pt[j] - x[j]) * (pt[j] - x[j]);
    if (iflag == 1)
      fvec[i] = sqrt(sq) - x[d];
    else {
      double s = sqrt(sq);
      for (j = 0; j < d; j++) fjac[(size_t)i * n + j] = (x[j] - pt[j]) / s;
      fjac[(size_t)i * n + d] = -1;
    }
  }
}

/* SphereParametersEstimator.hxx:310-338.  vnl_levenberg_marquardt defaults not overridden there:
 * ftol = 1e-8*0.01 = 1e-10 (vnl_nonlinear_minimizer), factor 100, mode 1. */
int orc_sphere_geometric(int d, const double *const *p, size_t n, const double *init,
                         double *out, int *info_out, int *nfev_out) {
  sphere_lm_ctx c;
  double x[65];
  int info, nfev, j;
  c.d = d;
  c.p = p;
  for (j = 0; j <= d; j++) x[j] = init[j];
  info = orc_lmder(sphere_lm_fcn, &c, (int)n, d + 1, x, 1e-10, 10e-16, 10e-16, 500, 100.0,
                   &nfev, NULL, NULL);
  if (info_out) *info_out = info;
  if (nfev_out) *nfev_out = nfev;
  if (info < 1 || info > 4) return 0; /* vnl_levenberg_marquardt: ok for info in 1..4 */
  for (j = 0; j <= d; j++) out[j] = x[j];
  return d + 1;
}

/* SphereParametersEstimator.hxx:216-236 */
static int sphere_ls(const orc_cfg *c, const double *const *p, size_t n, double *out) {
  double init[65];
  if (n < (size_t)(c->dim + 1)) return 0;
  if (c->ls_type == ORC_LS_ALGEBRAIC) return orc_sphere_algebraic(c->dim, p, n, out);
  if (!orc_sphere_algebraic(c->dim, p, n, init)) return 0;
  return orc_sphere_geometric(c->dim, p, n, init, out, NULL, NULL);
}

/* ================================================================== dense Ax=b */
/* DenseLinear...hxx:17-49 (n rows) and :64-96 (m rows): x = pinv(A) b, rank < n -> empty */
static int dense_solve(int n, const double *const *rows, size_t m, double *out) {
  double *A, *b;
  size_t i;
  int rank;
  if (m < (size_t)n) return 0;
  A = (double *)malloc(sizeof(double) * m * n);
  b = (double *)malloc(sizeof(double) * m);
  for (i = 0; i < m; i++) {
    memcpy(A + i * n, rows[i], sizeof(double) * n);
    b[i] = rows[i][n];
  }
  rank = orc_pinv_solve((int)m, n, A, b, EPS, out);
  free(A);
  free(b);
  return rank < n ? 0 : n;
}

/* DenseLinear...hxx:111-119 */
static int dense_agree(int n, double delta, const double *par, const double *row) {
  double sum = 0.0;
  int i;
  for (i = 0; i < n; i++) sum += row[i] * par[i];
  sum -= row[n];
  return fabs(sum) < delta;
}


/* ================================================================== absolute orientation */
/* vnl_vector<double>::normalize(): scale by 1/sqrt(sum of squares) unless the sum is zero
 * (VNL absent from /root/reference: restated from its published source, parity unpinned) */
static void vnl_normalize3(double *v) {
  double tmp = 0;
  int i;
  for (i = 0; i < 3; i++) tmp += v[i] * v[i];
  if (tmp != 0) {
    tmp = 1.0 / sqrt(tmp);
    for (i = 0; i < 3; i++) v[i] = tmp * v[i];
  }
}
/* AbsoluteOrientationParametersEstimator.cxx:25-50 / :57-79: orthonormal triad of three points.
 * Returns 0 when the points are collinear (:48, :78). */
static int absor_triad(const double *p0, const double *p1, const double *p2, double mean[3],
                       double R[3][3]) {
  double x[3], y[3], z[3], d;
  int i;
  for (i = 0; i < 3; i++) mean[i] = (p0[i] + p1[i] + p2[i]) / 3.0;
  for (i = 0; i < 3; i++) x[i] = p0[i] - mean[i];
  vnl_normalize3(x);
  for (i = 0; i < 3; i++) y[i] = p1[i] - mean[i];
  d = 0;
  for (i = 0; i < 3; i++) d += y[i] * x[i];
  for (i = 0; i < 3; i++) y[i] = y[i] - d * x[i];
  vnl_normalize3(y);
  z[0] = x[1] * y[2] - x[2] * y[1];
  z[1] = x[2] * y[0] - x[0] * y[2];
  z[2] = x[0] * y[1] - x[1] * y[0];
  d = 0;
  for (i = 0; i < 3; i++) d += z[i] * z[i];
  if (sqrt(d) < EPS) return 0;
  for (i = 0; i < 3; i++) {
    R[i][0] = x[i];
    R[i][1] = y[i];
    R[i][2] = z[i];
  }
  return 1;
}
/* common/Frame.cxx:952-991 getRotationQuaternion */
static void frame_quaternion(double R[3][3], double q[4]) {
  const double smallAngle = 0.008726535498373935, halfPI = 3.14159265358979323846 / 2.0;
  double startSingularRange = halfPI - smallAngle, endSingularRange = halfPI + smallAngle;
  double halfTheta;
  q[0] = (0.5 * sqrt(R[0][0] + R[1][1] + R[2][2] + 1));
  halfTheta = acos(q[0]);
  if (!(halfTheta > startSingularRange && halfTheta < endSingularRange)) {
    double denom = 4 * q[0];
    q[1] = (R[2][1] - R[1][2]) / denom;
    q[2] = (R[0][2] - R[2][0]) / denom;
    q[3] = (R[1][0] - R[0][1]) / denom;
  } else {
    int i = 0, j, k;
    double w;
    if (R[1][1] > R[i][i]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    j = (i + 1) % 3;
    k = (j + 1) % 3;
    w = sqrt(R[i][i] - R[j][j] - R[k][k] + 1);
    q[i + 1] = w / 2.0;
    q[j + 1] = (R[i][j] + R[j][i]) / (2 * w);
    q[k + 1] = (R[i][k] + R[k][i]) / (2 * w);
  }
}
/* common/Frame.cxx:750-771 setRotationQuaternion */
static void frame_from_quaternion(double s, double qx, double qy, double qz, int normalize,
                                  double R[3][3]) {
  if (normalize) {
    double norm = sqrt(s * s + qx * qx + qy * qy + qz * qz);
    s /= norm;
    qx /= norm;
    qy /= norm;
    qz /= norm;
  }
  R[0][0] = 1 - 2 * (qy * qy + qz * qz);
  R[0][1] = 2 * (qx * qy - s * qz);
  R[0][2] = 2 * (qx * qz + s * qy);
  R[1][0] = 2 * (qx * qy + s * qz);
  R[1][1] = 1 - 2 * (qx * qx + qz * qz);
  R[1][2] = 2 * (qy * qz - s * qx);
  R[2][0] = 2 * (qx * qz - s * qy);
  R[2][1] = 2 * (qy * qz + s * qx);
  R[2][2] = 1 - 2 * (qx * qx + qy * qy);
}
/* AbsoluteOrientationParametersEstimator.cxx:14-105; record = [first(3), second(3)] */
static int absor_estimate(const double *const *p, size_t n, double *out) {
  double R1[3][3], R2[3][3], R[3][3], m1[3], m2[3], t[3], q[4];
  int i, j, k;
  if (n < 3) return 0;
  if (!absor_triad(p[0], p[1], p[2], m1, R1)) return 0;
  if (!absor_triad(p[0] + 3, p[1] + 3, p[2] + 3, m2, R2)) return 0;
  for (i = 0; i < 3; i++) /* R = secondR * firstR^T (:86), vnl product = running sum from 0 */
    for (j = 0; j < 3; j++) {
      double sum = 0;
      for (k = 0; k < 3; k++) sum += R2[i][k] * R1[j][k];
      R[i][j] = sum;
    }
  for (i = 0; i < 3; i++) { /* t = meanSecond - R*meanFirst (:88) */
    double sum = 0;
    for (k = 0; k < 3; k++) sum += R[i][k] * m1[k];
    t[i] = m2[i] - sum;
  }
  frame_quaternion(R, q);
  for (i = 0; i < 4; i++) out[i] = q[i];
  for (i = 0; i < 3; i++) out[4 + i] = t[i];
  return 7;
}
/* AbsoluteOrientationParametersEstimator.cxx:316-327 (Frame ctor Frame.cxx:174-198 without
 * normalisation, apply() Frame.cxx:229-247) */
static int absor_agree(double delta_sq, const double *par, const double *rec) {
  double R[3][3], x, y, z, dx, dy, dz;
  frame_from_quaternion(par[0], par[1], par[2], par[3], 0, R);
  x = R[0][0] * rec[0] + R[0][1] * rec[1] + R[0][2] * rec[2] + par[4];
  y = R[1][0] * rec[0] + R[1][1] * rec[1] + R[1][2] * rec[2] + par[5];
  z = R[2][0] * rec[0] + R[2][1] * rec[1] + R[2][2] * rec[2] + par[6];
  dx = x - rec[3];
  dy = y - rec[4];
  dz = z - rec[5];
  return ((dx * dx + dy * dy + dz * dz) < delta_sq);
}
/* AbsoluteOrientationParametersEstimator.cxx:123-198 (Horn) */
static int absor_ls(const double *const *p, size_t n, double *out) {
  double m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0}, M[3][3], N[16], w[4], V[16], R[3][3], q[4];
  double traceM, A12, A20, A01, mf[3];
  size_t i;
  int a, b;
  if (n < 3) return 0;
  for (i = 0; i < n; i++)
    for (a = 0; a < 3; a++) {
      m1[a] += p[i][a];
      m2[a] += p[i][3 + a];
    }
  for (a = 0; a < 3; a++) {
    m1[a] /= (double)(unsigned int)n;
    m2[a] /= (double)(unsigned int)n;
  }
  memset(M, 0, sizeof M);
  for (i = 0; i < n; i++)
    for (a = 0; a < 3; a++)
      for (b = 0; b < 3; b++) M[a][b] += p[i][a] * p[i][3 + b];
  for (a = 0; a < 3; a++)
    for (b = 0; b < 3; b++) M[a][b] += (m1[a] * m2[b]) * (double)(-((int)n)); /* :168 */
  traceM = 0.0;
  for (a = 0; a < 3; a++) traceM += M[a][a];
  A12 = M[1][2] - M[2][1];
  A20 = M[2][0] - M[0][2];
  A01 = M[0][1] - M[1][0];
  N[0] = traceM; N[1] = A12; N[2] = A20; N[3] = A01;
  N[4] = A12; N[8] = A20; N[12] = A01;
  for (a = 0; a < 3; a++)
    for (b = 0; b < 3; b++)
      N[(a + 1) * 4 + (b + 1)] = (a == b ? -traceM : 0.0) + (M[a][b] + M[b][a]);
  orc_sym_eig(4, N, w, V); /* ascending: column 3 = largest (:187-195) */
  for (a = 0; a < 4; a++) q[a] = V[a * 4 + 3];
  for (a = 0; a < 4; a++) out[a] = q[a];
  frame_from_quaternion(q[0], q[1], q[2], q[3], 1, R);
  for (a = 0; a < 3; a++) mf[a] = R[a][0] * m1[0] + R[a][1] * m1[1] + R[a][2] * m1[2] + 0.0;
  for (a = 0; a < 3; a++) out[4 + a] = m2[a] - mf[a];
  return 7;
}

/* AbsoluteOrientationParametersEstimator.cxx:208-291 weightedLeastSquaresEstimate: Horn's closed form with
 * weighted means and a weighted correlation matrix.  Records: [first(3), second(3)]; w: one weight each. */
int orc_absor_weighted_ls(const double *const *p, const double *wt, size_t n, double *out) {
  double m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0}, M[3][3], N[16], w[4], V[16], R[3][3], q[4];
  double traceM, A12, A20, A01, mf[3], sumWeights = 0.0;
  size_t i;
  int a, b;
  if (n < 3) return 0;
  for (i = 0; i < n; i++) sumWeights += wt[i];
  for (i = 0; i < n; i++)
    for (a = 0; a < 3; a++) {
      m1[a] += p[i][a] * wt[i];
      m2[a] += p[i][3 + a] * wt[i];
    }
  for (a = 0; a < 3; a++) {
    m1[a] /= sumWeights;
    m2[a] /= sumWeights;
  }
  memset(M, 0, sizeof M);
  for (i = 0; i < n; i++)
    for (a = 0; a < 3; a++)
      for (b = 0; b < 3; b++) M[a][b] += (p[i][a] * p[i][3 + b]) * wt[i];
  for (a = 0; a < 3; a++)
    for (b = 0; b < 3; b++) M[a][b] += (m1[a] * m2[b]) * (-sumWeights);
  traceM = 0.0;
  for (a = 0; a < 3; a++) traceM += M[a][a];
  A12 = M[1][2] - M[2][1];
  A20 = M[2][0] - M[0][2];
  A01 = M[0][1] - M[1][0];
  N[0] = traceM; N[1] = A12; N[2] = A20; N[3] = A01;
  N[4] = A12; N[8] = A20; N[12] = A01;
  for (a = 0; a < 3; a++)
    for (b = 0; b < 3; b++)
      N[(a + 1) * 4 + (b + 1)] = (a == b ? -traceM : 0.0) + (M[a][b] + M[b][a]);
  orc_sym_eig(4, N, w, V);
  for (a = 0; a < 4; a++) q[a] = V[a * 4 + 3];
  for (a = 0; a < 4; a++) out[a] = q[a];
  frame_from_quaternion(q[0], q[1], q[2], q[3], 1, R);
  for (a = 0; a < 3; a++) mf[a] = R[a][0] * m1[0] + R[a][1] * m1[1] + R[a][2] * m1[2] + 0.0;
  for (a = 0; a < 3; a++) out[4 + a] = m2[a] - mf[a];
  return 7;
}

/* ================================================================== pivot calibration */
/* record = Frame: rotation[3][3] slots 0..8, translation 9..11 (common/Frame.h:30-31) */
/* PivotCalibrationParametersEstimator.cxx:9-50 (3 frames) and :63-96 (n frames): rows [R_i, -I],
 * rhs -t_i, pseudo-inverse with singular values <= EPS zeroed, rank < 6 -> empty */
static int pivot_solve(const double *const *f, size_t n, double *out) {
  double *A, *b;
  size_t i;
  int r, c, rank;
  if (n < 3) return 0;
  A = (double *)calloc(3 * n * 6, sizeof(double));
  b = (double *)calloc(3 * n, sizeof(double));
  for (i = 0; i < n; i++)
    for (r = 0; r < 3; r++) {
      for (c = 0; c < 3; c++) A[(3 * i + r) * 6 + c] = f[i][3 * r + c];
      A[(3 * i + r) * 6 + 3 + r] = -1.0;
      b[3 * i + r] = -f[i][9 + r];
    }
  rank = orc_pinv_solve((int)(3 * n), 6, A, b, EPS, out);
  free(A);
  free(b);
  return rank < 6 ? 0 : 6;
}
/* PivotCalibrationParametersEstimator.cxx:109-123: ||R*tDRF + t - tW|| < delta
 * (Frame::apply Frame.cxx:208-227, Vector l2Norm = sqrt of the sum of squares) */
static int pivot_agree(double delta, const double *par, const double *f) {
  double x, y, z, dx, dy, dz;
  x = f[0] * par[0] + f[1] * par[1] + f[2] * par[2] + f[9];
  y = f[3] * par[0] + f[4] * par[1] + f[5] * par[2] + f[10];
  z = f[6] * par[0] + f[7] * par[1] + f[8] * par[2] + f[11];
  dx = x - par[3];
  dy = y - par[4];
  dz = z - par[5];
  return sqrt(dx * dx + dy * dy + dz * dz) < delta;
}




/* ================================================================== plane-phantom US calibration */
/* PlanePhantomUSCalibrationParametersEstimator.{h,cxx}; record = US single record (Frame T2 slots
 * 0..11, Point2D q slots 13..14); parameters: 11 minimal [omega1_y, omega1_x, t1_z, t3(3), omega3_z,
 * omega3_y, omega3_x, m_x, m_y] + 30 derived products (.cxx:325-354) = 41. */
/* the data row a_i of the homogeneous system (.cxx:163-193); f_i = a_i . e(parameters) */
static void phantom_row(const double *rec, double a[31]) {
  double u = rec[US_Q], v = rec[US_Q + 1];
  int j;
  for (j = 0; j < 9; j++) {
    a[j] = rec[j] * u;
    a[9 + j] = rec[j] * v;
    a[18 + j] = rec[j];
  }
  a[27] = rec[US_T2];
  a[28] = rec[US_T2 + 1];
  a[29] = rec[US_T2 + 2];
  a[30] = 1;
}
/* .cxx:73-135: the 31-term sum in the reference's order (u*R2*p, left to right) */
static double phantom_err(const double *par, const double *rec) {
  double u = rec[US_Q], v = rec[US_Q + 1], err;
  const double *R2 = rec, *t2 = rec + US_T2;
  err = u * R2[0] * par[11] + u * R2[1] * par[12] + u * R2[2] * par[13] + u * R2[3] * par[14] +
        u * R2[4] * par[15] + u * R2[5] * par[16] + u * R2[6] * par[17] + u * R2[7] * par[18] +
        u * R2[8] * par[19] + v * R2[0] * par[20] + v * R2[1] * par[21] + v * R2[2] * par[22] +
        v * R2[3] * par[23] + v * R2[4] * par[24] + v * R2[5] * par[25] + v * R2[6] * par[26] +
        v * R2[7] * par[27] + v * R2[8] * par[28] + R2[0] * par[29] + R2[1] * par[30] +
        R2[2] * par[31] + R2[3] * par[32] + R2[4] * par[33] + R2[5] * par[34] + R2[6] * par[35] +
        R2[7] * par[36] + R2[8] * par[37] + t2[0] * par[38] + t2[1] * par[39] + t2[2] * par[40] +
        par[2];
  return err;
}
static int phantom_agree(double delta_sq, const double *par, const double *rec) {
  double err = phantom_err(par, rec);
  return (err * err < delta_sq);
}
/* the 30 derived entries from the 11 minimal ones (.cxx:383-452) */
static void phantom_expand(double *p) {
  double cy = cos(p[0]), sy = sin(p[0]), cx = cos(p[1]), sx = sin(p[1]);
  double R1[3], R3[9], cz, sz, mx = p[9], my = p[10];
  int k = 11, a, j;
  R1[0] = -sy;
  R1[1] = cy * sx;
  R1[2] = cy * cx;
  cz = cos(p[6]); sz = sin(p[6]);
  cy = cos(p[7]); sy = sin(p[7]);
  cx = cos(p[8]); sx = sin(p[8]);
  R3[0] = cz * cy; R3[1] = cz * sy * sx - sz * cx; R3[2] = cz * sy * cx + sz * sx;
  R3[3] = sz * cy; R3[4] = sz * sy * sx + cz * cx; R3[5] = sz * sy * cx - cz * sx;
  R3[6] = -sy;     R3[7] = cy * sx;                R3[8] = cy * cx;
  for (a = 0; a < 3; a++)
    for (j = 0; j < 3; j++) p[k++] = mx * R3[3 * j + 0] * R1[a];
  for (a = 0; a < 3; a++)
    for (j = 0; j < 3; j++) p[k++] = my * R3[3 * j + 1] * R1[a];
  for (a = 0; a < 3; a++)
    for (j = 0; j < 3; j++) p[k++] = p[3 + j] * R1[a];
  for (a = 0; a < 3; a++) p[k++] = R1[a];
}
/* .cxx:137-355 */
int orc_phantom_analytic(const double *const *recs, size_t n, double *out) {
  const double smallAngle = 0.008726535498373935, halfPI = 1.5707963267948966192313216916398;
  double *A, *U, s[31], V[31 * 31], x[31], denominator, scaleFactor, t1_z, R1_31, R1_32, R1_33;
  double omega1_y, omega1_x, t3[3], r1[3], r2[3], r3[3], m_x, m_y, R3[9], U3[9], s3[3], V3[9];
  double omega3_z, omega3_y, omega3_x, nrm;
  size_t i;
  int j, k, rank = 0;
  if (n < 31) return 0;
  A = (double *)malloc(n * 31 * sizeof(double));
  U = (double *)malloc(n * 31 * sizeof(double));
  for (i = 0; i < n; i++) phantom_row(recs[i], A + 31 * i);
  orc_svd((int)n, 31, A, U, s, V); /* singular values descending: column 30 = smallest (:203) */
  free(A);
  free(U);
  for (j = 0; j < 31; j++) rank += (s[j] > 0.0); /* vnl_svd default: only exact zeros are dropped */
  if (rank < 31) return 0;
  for (j = 0; j < 31; j++) x[j] = V[j * 31 + 30];
  denominator = sqrt(x[27] * x[27] + x[28] * x[28] + x[29] * x[29]);
  if (denominator < EPS) return 0;
  scaleFactor = 1 / denominator;
  for (j = 0; j < 31; j++) x[j] *= scaleFactor;
  t1_z = x[30];
  R1_31 = x[27];
  R1_32 = x[28];
  R1_33 = x[29];
  omega1_y = atan2(-R1_31, sqrt(R1_32 * R1_32 + R1_33 * R1_33));
  if (fabs(omega1_y - halfPI) > smallAngle && fabs(omega1_y + halfPI) > smallAngle) {
    double cy = cos(omega1_y);
    omega1_x = atan2(R1_32 / cy, R1_33 / cy);
  } else {
    omega1_x = 0.0;
  }
  for (j = 0; j < 3; j++) t3[j] = (x[18 + j] / R1_31 + x[21 + j] / R1_32 + x[24 + j] / R1_33) / 3.0;
  for (j = 0; j < 3; j++) {
    r1[j] = (x[j] / R1_31 + x[3 + j] / R1_32 + x[6 + j] / R1_33) / 3.0;
    r2[j] = (x[9 + j] / R1_31 + x[12 + j] / R1_32 + x[15 + j] / R1_33) / 3.0;
  }
  m_x = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  nrm = r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2];
  if (nrm != 0) for (j = 0; j < 3; j++) r1[j] = (1.0 / sqrt(nrm)) * r1[j];
  m_y = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
  nrm = r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2];
  if (nrm != 0) for (j = 0; j < 3; j++) r2[j] = (1.0 / sqrt(nrm)) * r2[j];
  r3[0] = r1[1] * r2[2] - r1[2] * r2[1];
  r3[1] = r1[2] * r2[0] - r1[0] * r2[2];
  r3[2] = r1[0] * r2[1] - r1[1] * r2[0];
  for (j = 0; j < 3; j++) {
    R3[3 * j + 0] = r1[j];
    R3[3 * j + 1] = r2[j];
    R3[3 * j + 2] = r3[j];
  }
  orc_svd(3, 3, R3, U3, s3, V3);
  for (j = 0; j < 3; j++)
    for (k = 0; k < 3; k++) {
      double sum = 0;
      int l;
      for (l = 0; l < 3; l++) sum += U3[3 * j + l] * V3[3 * k + l];
      R3[3 * j + k] = sum;
    }
  omega3_y = atan2(-R3[6], sqrt(R3[0] * R3[0] + R3[3] * R3[3]));
  if (fabs(omega3_y - halfPI) > smallAngle && fabs(omega3_y + halfPI) > smallAngle) {
    double cy = cos(omega3_y);
    omega3_z = atan2(R3[3] / cy, R3[0] / cy);
    omega3_x = atan2(R3[7] / cy, R3[8] / cy);
  } else {
    omega3_z = 0;
    omega3_x = atan2(R3[1], R3[4]);
  }
  out[0] = omega1_y; out[1] = omega1_x; out[2] = t1_z;
  out[3] = t3[0]; out[4] = t3[1]; out[5] = t3[2];
  out[6] = omega3_z; out[7] = omega3_y; out[8] = omega3_x;
  out[9] = m_x; out[10] = m_y;
  k = 11;
  {
    double R1[3];
    int a;
    R1[0] = R1_31; R1[1] = R1_32; R1[2] = R1_33;
    for (a = 0; a < 3; a++)
      for (j = 0; j < 3; j++) out[k++] = m_x * R3[3 * j + 0] * R1[a];
    for (a = 0; a < 3; a++)
      for (j = 0; j < 3; j++) out[k++] = m_y * R3[3 * j + 1] * R1[a];
    for (a = 0; a < 3; a++)
      for (j = 0; j < 3; j++) out[k++] = t3[j] * R1[a];
    for (a = 0; a < 3; a++) out[k++] = R1[a];
  }
  return 41;
}
typedef struct { const double *const *recs; } phantom_lm_ctx;
/* .cxx:564-670 (f); the reference runs vnl_levenberg_marquardt WITHOUT a gradient (:552), i.e.
 * MINPACK lmdif: the Jacobian comes from forward differences with step sqrt(eps) |x_j| */
static void phantom_fvec(const phantom_lm_ctx *c, int m, const double *x, double *fvec) {
  double p[41];
  int i;
  for (i = 0; i < 11; i++) p[i] = x[i];
  phantom_expand(p);
  for (i = 0; i < m; i++) fvec[i] = phantom_err(p, c->recs[i]);
}
static void phantom_lm_fcn(void *vctx, int m, int n, const double *x, double *fvec, double *fjac,
                           int iflag) {
  const phantom_lm_ctx *c = (const phantom_lm_ctx *)vctx;
  if (iflag == 1) {
    phantom_fvec(c, m, x, fvec);
  } else { /* fdjac2 */
    const double eps = sqrt(2.220446049250313e-16);
    double *f0 = (double *)malloc(2 * (size_t)m * sizeof(double)), *f1 = f0 + m, xx[11];
    int i, j;
    phantom_fvec(c, m, x, f0);
    for (j = 0; j < n; j++) {
      double h = eps * fabs(x[j]);
      if (h == 0.0) h = eps;
      for (i = 0; i < n; i++) xx[i] = x[i];
      xx[j] = x[j] + h;
      phantom_fvec(c, m, xx, f1);
      for (i = 0; i < m; i++) fjac[(size_t)i * n + j] = (f1[i] - f0[i]) / h;
    }
    free(f0);
  }
}
/* .cxx:357-453 */
int orc_phantom_iterative(const double *const *recs, size_t n, const double *init, double *out,
                          int *info_out, int *nfev_out) {
  double x[11];
  int info, nfev, i;
  phantom_lm_ctx c;
  c.recs = recs;
  for (i = 0; i < 11; i++) x[i] = init[i];
  info = orc_lmder(phantom_lm_fcn, &c, (int)n, 11, x, 10e-16, 10e-16, 10e-16, 5000, 100.0, &nfev,
                   NULL, NULL);
  if (info_out) *info_out = info;
  if (nfev_out) *nfev_out = nfev;
  for (i = 0; i < 11; i++) out[i] = x[i];
  phantom_expand(out);
  if (info < 1 || info > 4) return 0;
  return 41;
}
/* .cxx:36-57 */
static int phantom_ls(const orc_cfg *c, const double *const *recs, size_t n, double *out) {
  double init[41];
  if (n < 31) return 0;
  if (c->ls_type == ORC_LS_ALGEBRAIC) return orc_phantom_analytic(recs, n, out);
  if (!orc_phantom_analytic(recs, n, init)) return 0;
  return orc_phantom_iterative(recs, n, init, out, NULL, NULL);
}

/* ================================================================== 2-D line, normal form */
/* Line2DParametersEstimator.cxx:9-27 */
static int line2d_estimate(double delta_sq, const double *const *p, size_t n, double *out) {
  double nx, ny, normSquared, norm;
  if (n < 2) return 0;
  nx = p[1][1] - p[0][1];
  ny = p[0][0] - p[1][0];
  normSquared = nx * nx + ny * ny;
  if (normSquared < delta_sq) return 0;
  norm = sqrt(nx * nx + ny * ny);
  out[0] = nx / norm;
  out[1] = ny / norm;
  out[2] = p[0][0];
  out[3] = p[0][1];
  if (out[0] != out[0] || out[1] != out[1]) return 0;
  return 4;
}
/* Line2DParametersEstimator.cxx:44-100 */
static int line2d_ls(const double *const *p, size_t n, double *out) {
  double meanX = 0, meanY = 0, nx, ny, norm, c11 = 0, c12 = 0, c22 = 0;
  int i, dataSize = (int)n;
  if (n < 2) return 0;
  for (i = 0; i < dataSize; i++) {
    meanX += p[i][0];
    meanY += p[i][1];
    c11 += p[i][0] * p[i][0];
    c12 += p[i][0] * p[i][1];
    c22 += p[i][1] * p[i][1];
  }
  meanX /= dataSize;
  meanY /= dataSize;
  c11 -= dataSize * meanX * meanX;
  c12 -= dataSize * meanX * meanY;
  c22 -= dataSize * meanY * meanY;
  if (c11 < 1e-12) {
    nx = 1.0;
    ny = 0.0;
    if (c22 < 1e-12) return 0;
  } else {
    double lambda1 = (c11 + c22 + sqrt((c11 - c22) * (c11 - c22) + 4 * c12 * c12)) / 2.0;
    nx = -c12;
    ny = lambda1 - c22;
    norm = sqrt(nx * nx + ny * ny);
    nx /= norm;
    ny /= norm;
  }
  out[0] = nx;
  out[1] = ny;
  out[2] = meanX;
  out[3] = meanY;
  return 4;
}

/* ================================================================== ray intersection */
/* RayIntersectionParametersEstimator.cxx:23-72; record = [p(3), n(3)] */
static int ray_estimate(double cross_eps, const double *const *r, size_t n, double *out) {
  const double *p1, *n1, *p2, *n2;
  double p21[3], x[3], denominator, t1, t2;
  if (n < 2) return 0;
  p1 = r[0]; n1 = r[0] + 3; p2 = r[1]; n2 = r[1] + 3;
  p21[0] = p2[0] - p1[0];
  p21[1] = p2[1] - p1[1];
  p21[2] = p2[2] - p1[2];
  x[0] = n1[1] * n2[2] - n1[2] * n2[1];
  x[1] = n1[2] * n2[0] - n1[0] * n2[2];
  x[2] = n1[0] * n2[1] - n1[1] * n2[0];
  denominator = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  if (denominator < cross_eps) return 0;
  t1 = (x[0] * (p21[1] * n2[2] - p21[2] * n2[1]) - x[1] * (p21[0] * n2[2] - p21[2] * n2[0]) +
        x[2] * (p21[0] * n2[1] - p21[1] * n2[0])) / denominator;
  t2 = (x[0] * (p21[1] * n1[2] - p21[2] * n1[1]) - x[1] * (p21[0] * n1[2] - p21[2] * n1[0]) +
        x[2] * (p21[0] * n1[1] - p21[1] * n1[0])) / denominator;
  if (t1 < 0 || t2 < 0) return 0;
  out[0] = (p1[0] + t1 * n1[0] + p2[0] + t2 * n2[0]) / 2.0;
  out[1] = (p1[1] + t1 * n1[1] + p2[1] + t2 * n2[1]) / 2.0;
  out[2] = (p1[2] + t1 * n1[2] + p2[2] + t2 * n2[2]) / 2.0;
  if (out[0] != out[0] || out[1] != out[1] || out[2] != out[2]) return 0;
  return 3;
}
/* RayIntersectionParametersEstimator.cxx:163-177 */
static int ray_agree(double delta_sq, const double *par, const double *r) {
  const double *p = r, *n = r + 3;
  double t = n[0] * (par[0] - p[0]) + n[1] * (par[1] - p[1]) + n[2] * (par[2] - p[2]);
  double dx = par[0] - p[0] - t * n[0];
  double dy = par[1] - p[1] - t * n[1];
  double dz = par[2] - p[2] - t * n[2];
  return t >= 0 && (dx * dx + dy * dy + dz * dz < delta_sq);
}
/* RayIntersectionParametersEstimator.cxx:95-143 */
static int ray_ls(const double *const *r, size_t n, double *out) {
  double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  size_t i;
  int rank;
  for (i = 0; i < n; i++) {
    const double *p = r[i], *d = r[i] + 3;
    double s;
    A[0] += -(d[0] * d[0]);
    A[1] += -(d[0] * d[1]);
    A[2] += -(d[0] * d[2]);
    A[4] += -(d[1] * d[1]);
    A[5] += -(d[1] * d[2]);
    A[8] += -(d[2] * d[2]);
    s = d[0] * p[0] + d[1] * p[1] + d[2] * p[2];
    b[0] += p[0] - s * d[0];
    b[1] += p[1] - s * d[1];
    b[2] += p[2] - s * d[2];
  }
  A[0] += (double)n;
  A[3] = A[1];
  A[4] += (double)n;
  A[6] = A[2];
  A[7] = A[5];
  A[8] += (double)n;
  rank = orc_pinv_solve(3, 3, A, b, EPS, out);
  return rank < 3 ? 0 : 3;
}

/* ================================================================== US calibration */
static void us_T3(int model, const double *par, double T3[3][4]) {
  /* SinglePointTarget...cxx:90-93 (single) / :745-748 (pointer) */
  int c0 = (model == ORC_US_SINGLE) ? 11 : 8, t = (model == ORC_US_SINGLE) ? 3 : 0, i;
  for (i = 0; i < 3; i++) {
    T3[i][0] = par[c0 + i];
    T3[i][1] = par[c0 + 3 + i];
    T3[i][2] = par[c0 + 6 + i];
    T3[i][3] = par[t + i];
  }
}

/* q' = T2*T3*[u,v,0,1] with vnl_matrix operator* semantics: each entry a running sum from 0
 * over the inner index (SinglePointTarget...cxx:100).  Bottom rows are (0,0,0,1); the products
 * with those exact 0/1 entries are kept so the rounding sequence is the reference's. */
static void us_map(const double *rec, double T3[3][4], double out[3]) {
  double M[4][4], T2[4][4], T3f[4][4], q[4], r[4];
  int i, j, k;
  for (i = 0; i < 3; i++) {
    for (j = 0; j < 3; j++) T2[i][j] = rec[3 * i + j];
    T2[i][3] = rec[US_T2 + i];
    for (j = 0; j < 4; j++) T3f[i][j] = T3[i][j];
  }
  T2[3][0] = T2[3][1] = T2[3][2] = 0.0;
  T2[3][3] = 1;
  T3f[3][0] = T3f[3][1] = T3f[3][2] = 0.0;
  T3f[3][3] = 1;
  for (i = 0; i < 4; i++)
    for (k = 0; k < 4; k++) {
      double sum = 0;
      for (j = 0; j < 4; j++) sum += T2[i][j] * T3f[j][k];
      M[i][k] = sum;
    }
  q[0] = rec[US_Q];
  q[1] = rec[US_Q + 1];
  q[2] = 0.0;
  q[3] = 1.0;
  for (i = 0; i < 4; i++) {
    double sum = 0;
    for (j = 0; j < 4; j++) sum += M[i][j] * q[j];
    r[i] = sum;
  }
  out[0] = r[0];
  out[1] = r[1];
  out[2] = r[2];
}

/* SinglePointTarget...cxx:74-107 (single), :728-766 (pointer) */
static int us_agree(int model, double delta_sq, const double *par, const double *rec) {
  double T3[3][4], q[3], ex, ey, ez;
  us_T3(model, par, T3);
  us_map(rec, T3, q);
  if (model == ORC_US_SINGLE) {
    ex = q[0] - par[0];
    ey = q[1] - par[1];
    ez = q[2] - par[2];
  } else {
    ex = q[0] - rec[US_P];
    ey = q[1] - rec[US_P + 1];
    ez = q[2] - rec[US_P + 2];
  }
  return (ex * ex + ey * ey + ez * ez < delta_sq);
}

static void us_distance(int model, const double *par, const double *rec, double *dist) {
  double T3[3][4], q[3], ex, ey, ez;
  us_T3(model, par, T3);
  us_map(rec, T3, q);
  if (model == ORC_US_SINGLE) {
    ex = q[0] - par[0]; ey = q[1] - par[1]; ez = q[2] - par[2];
  } else {
    ex = q[0] - rec[US_P]; ey = q[1] - rec[US_P + 1]; ez = q[2] - rec[US_P + 2];
  }
  *dist = sqrt(ex * ex + ey * ey + ez * ez);
}

/* SinglePointTarget...cxx:120-270 (single, 3N x 12) and :775-917 (pointer, 3N x 9) */
int orc_us_analytic(int model, const double *const *recs, size_t n, double *out) {
  int single = (model == ORC_US_SINGLE), nc = single ? 12 : 9, minN = single ? 4 : 3;
  double *A, *b, x[12], r1[3], r2[3], r3[3], R3[9], U[9], s[3], V[9], m_x, m_y;
  double omega_z, omega_y, omega_x, nr;
  const double smallAngle = 0.008726535498373935, halfPI = 1.5707963267948966192313216916398;
  size_t i;
  int j, k, rank;
  if (n < (size_t)minN) return 0;
  A = (double *)calloc(3 * n * nc, sizeof(double));
  b = (double *)malloc(sizeof(double) * 3 * n);
  for (i = 0; i < n; i++) {
    const double *rec = recs[i];
    double ui = rec[US_Q], vi = rec[US_Q + 1];
    for (j = 0; j < 3; j++) {
      double *row = A + (3 * i + j) * nc;
      for (k = 0; k < 3; k++) {
        row[k] = rec[3 * j + k] * ui;
        row[3 + k] = rec[3 * j + k] * vi;
        row[6 + k] = rec[3 * j + k];
      }
      if (single) {
        row[9 + j] = -1.0;
        b[3 * i + j] = -rec[US_T2 + j];
      } else
        b[3 * i + j] = rec[US_P + j] - rec[US_T2 + j];
    }
  }
  rank = orc_pinv_solve((int)(3 * n), nc, A, b, US_SV_EPS, x);
  free(A);
  free(b);
  if (rank < nc) return 0;
  for (j = 0; j < 3; j++) {
    r1[j] = x[j];
    r2[j] = x[3 + j];
  }
  m_x = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
  for (j = 0; j < 3; j++) r1[j] /= m_x;
  m_y = sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
  for (j = 0; j < 3; j++) r2[j] /= m_y;
  r3[0] = r1[1] * r2[2] - r1[2] * r2[1];
  r3[1] = r1[2] * r2[0] - r1[0] * r2[2];
  r3[2] = r1[0] * r2[1] - r1[1] * r2[0];
  for (j = 0; j < 3; j++) {
    R3[3 * j + 0] = r1[j];
    R3[3 * j + 1] = r2[j];
    R3[3 * j + 2] = r3[j];
  }
  orc_svd(3, 3, R3, U, s, V);
  for (j = 0; j < 3; j++)
    for (k = 0; k < 3; k++) {
      double sum = 0;
      int l;
      for (l = 0; l < 3; l++) sum += U[3 * j + l] * V[3 * k + l];
      R3[3 * j + k] = sum;
    }
  (void)nr;
  omega_y = atan2(-R3[6], sqrt(R3[0] * R3[0] + R3[3] * R3[3]));
  if (fabs(omega_y - halfPI) > smallAngle && fabs(omega_y + halfPI) > smallAngle) {
    double cy = cos(omega_y);
    omega_z = atan2(R3[3] / cy, R3[0] / cy);
    omega_x = atan2(R3[7] / cy, R3[8] / cy);
  } else {
    omega_z = 0;
    omega_x = atan2(R3[1], R3[4]);
  }
  k = 0;
  if (single) {
    out[k++] = x[9]; out[k++] = x[10]; out[k++] = x[11];
  }
  out[k++] = x[6]; out[k++] = x[7]; out[k++] = x[8];
  out[k++] = omega_z; out[k++] = omega_y; out[k++] = omega_x;
  out[k++] = m_x; out[k++] = m_y;
  out[k++] = m_x * R3[0]; out[k++] = m_x * R3[3]; out[k++] = m_x * R3[6];
  out[k++] = m_y * R3[1]; out[k++] = m_y * R3[4]; out[k++] = m_y * R3[7];
  out[k++] = R3[2]; out[k++] = R3[5]; out[k++] = R3[8];
  return k;
}

typedef struct { int model; const double *const *recs; } us_lm_ctx;

/* f: SinglePointTarget...cxx:415-509 / :1059-1146; gradf: :512-658 / :1149-1286 */
static void us_lm_fcn(void *vctx, int m, int n, const double *x, double *fvec, double *fjac,
                      int iflag) {
  us_lm_ctx *c = (us_lm_ctx *)vctx;
  int single = (c->model == ORC_US_SINGLE), o = single ? 3 : 0, i;
  double t_1x = 0, t_1y = 0, t_1z = 0, t_3x, t_3y, t_3z, sz, cz, sy, cy, sx, cx, m_x, m_y;
  double R3_11, R3_21, R3_31, R3_12, R3_22, R3_32;
  if (single) {
    t_1x = x[0]; t_1y = x[1]; t_1z = x[2];
  }
  t_3x = x[o + 0]; t_3y = x[o + 1]; t_3z = x[o + 2];
  sz = sin(x[o + 3]); cz = cos(x[o + 3]);
  sy = sin(x[o + 4]); cy = cos(x[o + 4]);
  sx = sin(x[o + 5]); cx = cos(x[o + 5]);
  m_x = x[o + 6]; m_y = x[o + 7];
  R3_11 = cz * cy;
  R3_21 = sz * cy;
  R3_31 = -sy;
  R3_12 = cz * sy * sx - sz * cx;
  R3_22 = sz * sy * sx + cz * cx;
  R3_32 = cy * sx;
  for (i = 0; i < m; i++) {
    const double *rec = c->recs[i];
    const double *R2 = rec, *t2 = rec + US_T2;
    double u = rec[US_Q], v = rec[US_Q + 1];
    double A_11 = u * R2[0], A_12 = u * R2[1], A_13 = u * R2[2];
    double A_21 = u * R2[3], A_22 = u * R2[4], A_23 = u * R2[5];
    double A_31 = u * R2[6], A_32 = u * R2[7], A_33 = u * R2[8];
    double A_14 = v * R2[0], A_15 = v * R2[1], A_16 = v * R2[2];
    double A_24 = v * R2[3], A_25 = v * R2[4], A_26 = v * R2[5];
    double A_34 = v * R2[6], A_35 = v * R2[7], A_36 = v * R2[8];
    double A_17 = R2[0], A_18 = R2[1], A_19 = R2[2];
    double A_27 = R2[3], A_28 = R2[4], A_29 = R2[5];
    double A_37 = R2[6], A_38 = R2[7], A_39 = R2[8];
    double b_1, b_2, b_3, expr1, expr2, expr3, delta_i;
    if (single) {
      b_1 = -t2[0]; b_2 = -t2[1]; b_3 = -t2[2];
      expr1 = A_11 * m_x * R3_11 + A_12 * m_x * R3_21 + A_13 * m_x * R3_31 + A_14 * m_y * R3_12 +
              A_15 * m_y * R3_22 + A_16 * m_y * R3_32 + A_17 * t_3x + A_18 * t_3y + A_19 * t_3z -
              t_1x - b_1;
      expr2 = A_21 * m_x * R3_11 + A_22 * m_x * R3_21 + A_23 * m_x * R3_31 + A_24 * m_y * R3_12 +
              A_25 * m_y * R3_22 + A_26 * m_y * R3_32 + A_27 * t_3x + A_28 * t_3y + A_29 * t_3z -
              t_1y - b_2;
      expr3 = A_31 * m_x * R3_11 + A_32 * m_x * R3_21 + A_33 * m_x * R3_31 + A_34 * m_y * R3_12 +
              A_35 * m_y * R3_22 + A_36 * m_y * R3_32 + A_37 * t_3x + A_38 * t_3y + A_39 * t_3z -
              t_1z - b_3;
    } else {
      b_1 = rec[US_P] - t2[0]; b_2 = rec[US_P + 1] - t2[1]; b_3 = rec[US_P + 2] - t2[2];
      expr1 = A_11 * m_x * R3_11 + A_12 * m_x * R3_21 + A_13 * m_x * R3_31 + A_14 * m_y * R3_12 +
              A_15 * m_y * R3_22 + A_16 * m_y * R3_32 + A_17 * t_3x + A_18 * t_3y + A_19 * t_3z -
              b_1;
      expr2 = A_21 * m_x * R3_11 + A_22 * m_x * R3_21 + A_23 * m_x * R3_31 + A_24 * m_y * R3_12 +
              A_25 * m_y * R3_22 + A_26 * m_y * R3_32 + A_27 * t_3x + A_28 * t_3y + A_29 * t_3z -
              b_2;
      expr3 = A_31 * m_x * R3_11 + A_32 * m_x * R3_21 + A_33 * m_x * R3_31 + A_34 * m_y * R3_12 +
              A_35 * m_y * R3_22 + A_36 * m_y * R3_32 + A_37 * t_3x + A_38 * t_3y + A_39 * t_3z -
              b_3;
    }
    delta_i = sqrt(expr1 * expr1 + expr2 * expr2 + expr3 * expr3);
    if (iflag == 1) {
      fvec[i] = delta_i;
    } else {
      double *J = fjac + (size_t)i * n;
      double v1, v2, v3, v4, v5, v6;
      if (single) {
        J[0] = -expr1 / delta_i;
        J[1] = -expr2 / delta_i;
        J[2] = -expr3 / delta_i;
      }
      J[o + 0] = (A_17 * expr1 + A_27 * expr2 + A_37 * expr3) / delta_i;
      J[o + 1] = (A_18 * expr1 + A_28 * expr2 + A_38 * expr3) / delta_i;
      J[o + 2] = (A_19 * expr1 + A_29 * expr2 + A_39 * expr3) / delta_i;
      v1 = -m_x * sz * cy;
      v2 = m_x * cz * cy;
      v3 = -m_y * (sz * sy * sx + cz * cx);
      v4 = m_y * (cz * sy * sx - sz * cx);
      J[o + 3] = ((A_11 * v1 + A_12 * v2 + A_14 * v3 + A_15 * v4) * expr1 +
                  (A_21 * v1 + A_22 * v2 + A_24 * v3 + A_25 * v4) * expr2 +
                  (A_31 * v1 + A_32 * v2 + A_34 * v3 + A_35 * v4) * expr3) / delta_i;
      v1 = -m_x * sy * cz;
      v2 = -m_x * sy * sz;
      v3 = -m_x * cy;
      v4 = m_y * sx * cy * cz;
      v5 = m_y * sx * cy * sz;
      v6 = -m_y * sx * sy;
      J[o + 4] = ((A_11 * v1 + A_12 * v2 + A_13 * v3 + A_14 * v4 + A_15 * v5 + A_16 * v6) * expr1 +
                  (A_21 * v1 + A_22 * v2 + A_23 * v3 + A_24 * v4 + A_25 * v5 + A_26 * v6) * expr2 +
                  (A_31 * v1 + A_32 * v2 + A_33 * v3 + A_34 * v4 + A_35 * v5 + A_36 * v6) * expr3) /
                 delta_i;
      v1 = m_y * (cz * sy * cx + sz * sx);
      v2 = m_y * (sz * sy * cx - cz * sx);
      v3 = m_y * cy * cx;
      J[o + 5] = ((A_14 * v1 + A_15 * v2 + A_16 * v3) * expr1 +
                  (A_24 * v1 + A_25 * v2 + A_26 * v3) * expr2 +
                  (A_34 * v1 + A_35 * v2 + A_36 * v3) * expr3) / delta_i;
      J[o + 6] = ((A_11 * R3_11 + A_12 * R3_21 + A_13 * R3_31) * expr1 +
                  (A_21 * R3_11 + A_22 * R3_21 + A_23 * R3_31) * expr2 +
                  (A_31 * R3_11 + A_32 * R3_21 + A_33 * R3_31) * expr3) / delta_i;
      J[o + 7] = ((A_14 * R3_12 + A_15 * R3_22 + A_16 * R3_32) * expr1 +
                  (A_24 * R3_12 + A_25 * R3_22 + A_26 * R3_32) * expr2 +
                  (A_34 * R3_12 + A_35 * R3_22 + A_36 * R3_32) * expr3) / delta_i;
    }
  }
}

/* the residual vector (iflag 1 -> fvec[n]) or the Jacobian (iflag 2 -> fjac[n x np], row-major) of the US
 * calibration at x, for callers that drive their own minimiser (tests/golden/make_golden.py hands them to
 * SciPy's MINPACK to pin lmder's behaviour at the reference's tolerances) */
void orc_us_fcn(int model, const double *const *recs, size_t n, const double *x, double *fvec, double *fjac,
                int iflag) {
  us_lm_ctx c;
  c.model = model;
  c.recs = recs;
  us_lm_fcn(&c, (int)n, model == ORC_US_SINGLE ? 11 : 8, x, fvec, fjac, iflag);
}

/* SinglePointTarget...cxx:272-329 (tolerances 10e-16, 5000 evals) and :919-973 (10e-8) */
int orc_us_iterative(int model, const double *const *recs, size_t n, const double *init,
                     double *out, int *info_out, int *nfev_out) {
  int single = (model == ORC_US_SINGLE), np = single ? 11 : 8, o = single ? 3 : 0, info, nfev, i,
      k;
  double tol = single ? 10e-16 : 10e-8, x[11], cz, sz, cy, sy, cx, sx, mx, my;
  us_lm_ctx c;
  c.model = model;
  c.recs = recs;
  for (i = 0; i < np; i++) x[i] = init[i];
  info = orc_lmder(us_lm_fcn, &c, (int)n, np, x, tol, tol, tol, 5000, 100.0, &nfev, NULL, NULL);
  if (info_out) *info_out = info;
  if (nfev_out) *nfev_out = nfev;
  /* `out` always receives the last iterate (diagnostics); the return value carries the
   * reference's ok flag: 0 == empty vector unless MINPACK info is 1..4 */
  for (i = 0; i < np; i++) out[i] = x[i];
  cz = cos(x[o + 3]); sz = sin(x[o + 3]);
  cy = cos(x[o + 4]); sy = sin(x[o + 4]);
  cx = cos(x[o + 5]); sx = sin(x[o + 5]);
  mx = x[o + 6];
  my = x[o + 7];
  k = np;
  out[k++] = mx * cz * cy;
  out[k++] = mx * sz * cy;
  out[k++] = -mx * sy;
  out[k++] = my * (cz * sy * sx - sz * cx);
  out[k++] = my * (sz * sy * sx + cz * cx);
  out[k++] = my * cy * sx;
  out[k++] = cz * sy * cx + sz * sx;
  out[k++] = sz * sy * cx - cz * sx;
  out[k++] = cy * cx;
  if (info < 1 || info > 4) return 0;
  return k;
}

/* SinglePointTarget...cxx:37-58 / :690-712 */
static int us_ls(const orc_cfg *c, const double *const *recs, size_t n, double *out) {
  double init[20];
  if (n < (size_t)orc_min_subset(c)) return 0;
  if (c->ls_type == 0) return orc_us_analytic(c->model, recs, n, out);
  if (!orc_us_analytic(c->model, recs, n, init)) return 0;
  return orc_us_iterative(c->model, recs, n, init, out, NULL, NULL);
}

/* ================================================================== dispatch */
int orc_estimate(const orc_cfg *c, const double *const *recs, size_t n, double *params) {
  switch (c->model) {
    case ORC_PLANE: return plane_estimate(c->dim, recs, n, params);
    case ORC_SPHERE: /* SphereParametersEstimator.hxx:56-73 */
      if (n < (size_t)(c->dim + 1)) return 0;
      if (c->dim == 2) return sphere_estimate2d(recs, params);
      if (c->dim == 3) return sphere_estimate3d(recs, params);
      return sphere_estimateNd(c->dim, recs, params);
    case ORC_LINE: return line_estimate(c->dim, c->delta * c->delta, recs, n, params);
    case ORC_DENSE: return dense_solve(c->dim, recs, n, params);
    case ORC_ABSOR: return absor_estimate(recs, n, params);
    case ORC_PIVOT: return pivot_solve(recs, n < 3 ? n : 3, params);
    case ORC_RAY: return ray_estimate(sin(c->aux) * sin(c->aux), recs, n, params);
    case ORC_LINE2D: return line2d_estimate(c->delta * c->delta, recs, n, params);
    case ORC_PHANTOM: /* .cxx:16-24: exactly 31 elements */
      if (n != 31) return 0;
      return orc_phantom_analytic(recs, n, params);
    case ORC_US_SINGLE: /* :17-25: exactly minForEstimate elements */
    case ORC_US_POINTER:
      if (n != (size_t)orc_min_subset(c)) return 0;
      return orc_us_analytic(c->model, recs, n, params);
  }
  return 0;
}

int orc_agree(const orc_cfg *c, const double *params, const double *rec) {
  switch (c->model) {
    case ORC_PLANE: return plane_agree(c->dim, c->delta * c->delta, params, rec);
    case ORC_SPHERE: return sphere_agree(c->dim, c->delta, params, rec);
    case ORC_LINE: return line_agree(c->dim, c->delta * c->delta, params, rec);
    case ORC_DENSE: return dense_agree(c->dim, c->delta, params, rec);
    case ORC_ABSOR: return absor_agree(c->delta * c->delta, params, rec);
    case ORC_PIVOT: return pivot_agree(c->delta, params, rec);
    case ORC_RAY: return ray_agree(c->delta * c->delta, params, rec);
    case ORC_LINE2D: return plane_agree(2, c->delta * c->delta, params, rec); /* .cxx:117-121 */
    case ORC_PHANTOM: return phantom_agree(c->delta * c->delta, params, rec);
    case ORC_US_SINGLE:
    case ORC_US_POINTER: return us_agree(c->model, c->delta * c->delta, params, rec);
  }
  return 0;
}

int orc_ls(const orc_cfg *c, const double *const *recs, size_t n, double *params) {
  switch (c->model) {
    case ORC_PLANE: return cov_ls(c->dim, recs, n, c->dim, 0, params);
    case ORC_LINE: return cov_ls(c->dim, recs, n, 2, 1, params);
    case ORC_SPHERE: return sphere_ls(c, recs, n, params);
    case ORC_DENSE: return dense_solve(c->dim, recs, n, params);
    case ORC_ABSOR: return absor_ls(recs, n, params);
    case ORC_PIVOT: return pivot_solve(recs, n, params);
    case ORC_RAY: return ray_ls(recs, n, params);
    case ORC_LINE2D: return line2d_ls(recs, n, params);
    case ORC_PHANTOM: return phantom_ls(c, recs, n, params);
    case ORC_US_SINGLE:
    case ORC_US_POINTER: return us_ls(c, recs, n, params);
  }
  return 0;
}

int orc_ls_masked(const orc_cfg *c, const double *data, size_t n, size_t stride,
                  const uint8_t *mask, double *params) {
  const double **ptrs = (const double **)malloc(sizeof(double *) * (n ? n : 1));
  size_t i, m = 0;
  int r;
  for (i = 0; i < n; i++)
    if (!mask || mask[i]) ptrs[m++] = data + i * stride;
  r = orc_ls(c, ptrs, m, params);
  free(ptrs);
  return r;
}

size_t orc_scan(const orc_cfg *c, const double *params, const double *data, size_t n,
                size_t stride, uint8_t *mask) {
  size_t i, cnt = 0;
  for (i = 0; i < n; i++) {
    int a = orc_agree(c, params, data + i * stride);
    if (mask) mask[i] = (uint8_t)a;
    cnt += (size_t)a;
  }
  return cnt;
}

/* getDistanceStatistics: SphereParametersEstimator.hxx:341-377, SinglePointTarget...cxx:331-401.
 * For plane / line / dense (no such method in the reference) the residual is the quantity
 * agree() thresholds: |n.(p-a)|, point-line distance, |a.x-b|. */
int orc_stats(const orc_cfg *c, const double *params, const double *data, size_t n,
              size_t stride, const uint8_t *mask, double out[4]) {
  double mn = 0, mx = 0, sum = 0, sumsq = 0;
  size_t i, cnt = 0;
  int d = c->dim, j;
  for (i = 0; i < n; i++) {
    const double *x = data + i * stride;
    double dist = 0;
    if (mask && !mask[i]) continue;
    switch (c->model) {
      case ORC_PLANE:
        for (j = 0; j < d; j++) dist += params[j] * (x[j] - params[d + j]);
        dist = fabs(dist);
        break;
      case ORC_SPHERE:
        for (j = 0; j < d; j++) dist += (x[j] - params[j]) * (x[j] - params[j]);
        dist = fabs(sqrt(dist) - params[d]);
        break;
      case ORC_LINE: {
        double v[64], vn = 0;
        for (j = 0; j < d; j++) {
          v[j] = x[j] - params[d + j];
          vn += v[j] * params[j];
        }
        for (j = 0; j < d; j++) dist += (v[j] - vn * params[j]) * (v[j] - vn * params[j]);
        dist = sqrt(dist);
        break;
      }
      case ORC_DENSE:
        for (j = 0; j < d; j++) dist += x[j] * params[j];
        dist = fabs(dist - x[d]);
        break;
      case ORC_ABSOR: { /* ||T*first - second|| */
        double R[3][3], e[3];
        frame_from_quaternion(params[0], params[1], params[2], params[3], 0, R);
        for (j = 0; j < 3; j++)
          e[j] = R[j][0] * x[0] + R[j][1] * x[1] + R[j][2] * x[2] + params[4 + j] - x[3 + j];
        dist = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        break;
      }
      case ORC_PIVOT: { /* ||R*tDRF + t - tW|| */
        double e[3];
        for (j = 0; j < 3; j++)
          e[j] = x[3 * j] * params[0] + x[3 * j + 1] * params[1] + x[3 * j + 2] * params[2] +
                 x[9 + j] - params[3 + j];
        dist = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        break;
      }
      case ORC_LINE2D:
        dist = fabs(params[0] * (x[0] - params[2]) + params[1] * (x[1] - params[3]));
        break;
      case ORC_PHANTOM: dist = fabs(phantom_err(params, x)); break; /* .cxx:455-549 */
      case ORC_RAY: { /* distance of the point from the ray's line */
        double t = x[3] * (params[0] - x[0]) + x[4] * (params[1] - x[1]) + x[5] * (params[2] - x[2]);
        double e[3];
        for (j = 0; j < 3; j++) e[j] = params[j] - x[j] - t * x[3 + j];
        dist = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        break;
      }
      default: us_distance(c->model, params, x, &dist);
    }
    if (cnt == 0) mn = mx = dist;
    if (dist > mx) mx = dist;
    else if (dist < mn) mn = dist;
    sum += dist;
    sumsq += dist * dist;
    cnt++;
  }
  out[0] = mn;
  out[1] = mx;
  out[2] = cnt ? sum / (double)cnt : 0.0;
  out[3] = sumsq;
  return cnt > 0;
}
