// phantom.h -- PlanePhantomUSCalibrationParametersEstimator on the device
// (parametersEstimators/PlanePhantomUSCalibrationParametersEstimator.{h,cxx}).
//
// Record = the US single-target record (Frame T2 in slots 0..11, slot 12 = int outputFormat + padding,
// Point2D q in 13..14).  Parameters: 11 minimal [omega1_y, omega1_x, t1_z, t3(3), omega3_z, omega3_y,
// omega3_x, m_x, m_y] + 30 derived products = 41 (.cxx:325-354).
//
// Every residual is LINEAR in a 31-vector e(parameters):  f_i = a_i . e  with the data row
// a_i = [u R2 (9), v R2 (9), R2 (9), t2 (3), 1] (.cxx:163-193, :633-668).  Hence with the Gram matrix
// G = sum a_i a_i^T (31 x 31, one pass over the frames on the matrix cores):
//   * the analytic fit (.cxx:137-355, right singular vector of the smallest singular value of A) is
//     the eigenvector of the smallest eigenvalue of G,
//   * the iterative fit (.cxx:357-453, Levenberg-Marquardt on 11 parameters) needs no further pass:
//     sum f^2 = e^T G e,  J^T J = E^T G E,  J^T f = E^T G e  with E = de/dx (forward-mode derivatives),
//     fed to the same MINPACK control flow (lm_core.h) as the other LM fits.
// The minimal solve (exactly 31 frames, .cxx:16-24) takes the null vector of the 31 x 31 system by a
// workgroup-parallel one-sided Jacobi SVD in LDS.  The sign of a singular vector is arbitrary: the
// reference fixes the scale factor's sign arbitrarily too (.cxx:215-218), it flips T1 and leaves
// agree() -- a squared quantity -- unchanged.
#pragma once
#include <hip/hip_runtime.h>

#include "models.h"
#include "wave_linalg.h"

namespace lsqr {

struct PhantomModel {
  enum { ND = 15, K = 31, P = 41, SP = 41, REC = 15, PPL = 2, IS_DENSE = 0, IS_US = 0, IS_PHANTOM = 1 };
  enum { NMOM = 1, NE = 31, NX = 11 };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
#pragma unroll
    for (int i = 0; i < ND; i++) rec[i] = (i == 12) ? 0.0 : p[i];
  }
  // entry c of the data row a_i (.cxx:163-193)
  static LSQR_HD double row_entry(const double *x, int c) {
    if (c < 9) return x[c] * x[13];
    if (c < 18) return x[c - 9] * x[14];
    if (c < 27) return x[c - 18];
    if (c < 30) return x[9 + c - 27];
    return 1.0;
  }
  // .cxx:73-135: the reference's 31-term sum, terms (u*R2)*p in its order
  static LSQR_HD double err(const double *par, const double *x) {
    const double u = x[13], v = x[14];
    double e = u * x[0] * par[11];
#pragma unroll
    for (int j = 1; j < 9; j++) e += u * x[j] * par[11 + j];
#pragma unroll
    for (int j = 0; j < 9; j++) e += v * x[j] * par[20 + j];
#pragma unroll
    for (int j = 0; j < 9; j++) e += x[j] * par[29 + j];
    e += x[9] * par[38];
    e += x[10] * par[39];
    e += x[11] * par[40];
    e += par[2];
    return e;
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    const double e = err(sp, x);
    return e * e < c.delta_sq;
  }
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &) {
    return fabs(err(sp, x));
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}
  // the generic moment / solve templates are instantiated for every model; the phantom fit never
  // goes through them (run_fit branches on IS_PHANTOM)
  static LSQR_HD void accumulate(const double *, const double *, double *) {}
  static LSQR_HD bool solve(const double *, const double *, const ModelConsts &, double *) {
    return false;
  }

  // 30 derived entries from the 11 minimal ones (.cxx:383-452)
  static LSQR_HD void expand(double *p) {
    double cy = cos(p[0]), sy = sin(p[0]), cx = cos(p[1]), sx = sin(p[1]);
    double R1[3] = {-sy, cy * sx, cy * cx}, R3[9];
    const double mx = p[9], my = p[10];
    double cz = cos(p[6]), sz = sin(p[6]);
    cy = cos(p[7]), sy = sin(p[7]);
    cx = cos(p[8]), sx = sin(p[8]);
    R3[0] = cz * cy, R3[1] = cz * sy * sx - sz * cx, R3[2] = cz * sy * cx + sz * sx;
    R3[3] = sz * cy, R3[4] = sz * sy * sx + cz * cx, R3[5] = sz * sy * cx - cz * sx;
    R3[6] = -sy, R3[7] = cy * sx, R3[8] = cy * cx;
    int k = 11;
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 3; j++) p[k++] = mx * R3[3 * j] * R1[a];
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 3; j++) p[k++] = my * R3[3 * j + 1] * R1[a];
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 3; j++) p[k++] = p[3 + j] * R1[a];
    for (int a = 0; a < 3; a++) p[k++] = R1[a];
  }

  // .cxx:205-354: calibration parameters from the (unit) null vector x of the homogeneous system
  static LSQR_HD bool finish(const double *xin, double *par) {
    const double smallAngle = 0.008726535498373935, halfPI = 1.5707963267948966192313216916398;
    double x[31];
    for (int j = 0; j < 31; j++) x[j] = xin[j];
    const double denominator = sqrt(x[27] * x[27] + x[28] * x[28] + x[29] * x[29]);
    if (!(denominator >= kEPS)) return false;
    const double scaleFactor = 1 / denominator;
    for (int j = 0; j < 31; j++) x[j] *= scaleFactor;
    const double R1[3] = {x[27], x[28], x[29]};
    const double omega1_y = atan2(-R1[0], sqrt(R1[1] * R1[1] + R1[2] * R1[2]));
    double omega1_x = 0.0;
    if (fabs(omega1_y - halfPI) > smallAngle && fabs(omega1_y + halfPI) > smallAngle) {
      double cy = cos(omega1_y);
      omega1_x = atan2(R1[1] / cy, R1[2] / cy);
    }
    double t3[3], r1[3], r2[3], r3[3];
    for (int j = 0; j < 3; j++) {
      t3[j] = (x[18 + j] / R1[0] + x[21 + j] / R1[1] + x[24 + j] / R1[2]) / 3.0;
      r1[j] = (x[j] / R1[0] + x[3 + j] / R1[1] + x[6 + j] / R1[2]) / 3.0;
      r2[j] = (x[9 + j] / R1[0] + x[12 + j] / R1[1] + x[15 + j] / R1[2]) / 3.0;
    }
    double n1 = r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2];
    double n2 = r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2];
    const double m_x = sqrt(n1), m_y = sqrt(n2);
    if (n1 != 0)
      for (int j = 0; j < 3; j++) r1[j] = (1.0 / sqrt(n1)) * r1[j];
    if (n2 != 0)
      for (int j = 0; j < 3; j++) r2[j] = (1.0 / sqrt(n2)) * r2[j];
    r3[0] = r1[1] * r2[2] - r1[2] * r2[1];
    r3[1] = r1[2] * r2[0] - r1[0] * r2[2];
    r3[2] = r1[0] * r2[1] - r1[1] * r2[0];
    // closest rotation (Frobenius norm) to [r1 r2 r3]: U V^T of its SVD (:273-280)
    double Mq[9], s[3], V[9], R3[9];
    for (int j = 0; j < 3; j++) Mq[3 * j] = r1[j], Mq[3 * j + 1] = r2[j], Mq[3 * j + 2] = r3[j];
    svd_jacobi(3, 3, Mq, 3, s, V);  // Mq <- U
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double t = 0;
        for (int l = 0; l < 3; l++) t += Mq[3 * i + l] * V[3 * j + l];
        R3[3 * i + j] = t;
      }
    const double omega3_y = atan2(-R3[6], sqrt(R3[0] * R3[0] + R3[3] * R3[3]));
    double omega3_z, omega3_x;
    if (fabs(omega3_y - halfPI) > smallAngle && fabs(omega3_y + halfPI) > smallAngle) {
      double cy = cos(omega3_y);
      omega3_z = atan2(R3[3] / cy, R3[0] / cy);
      omega3_x = atan2(R3[7] / cy, R3[8] / cy);
    } else {
      omega3_z = 0;
      omega3_x = atan2(R3[1], R3[4]);
    }
    par[0] = omega1_y, par[1] = omega1_x, par[2] = x[30];
    par[3] = t3[0], par[4] = t3[1], par[5] = t3[2];
    par[6] = omega3_z, par[7] = omega3_y, par[8] = omega3_x;
    par[9] = m_x, par[10] = m_y;
    int k = 11;
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 3; j++) par[k++] = m_x * R3[3 * j] * R1[a];
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 3; j++) par[k++] = m_y * R3[3 * j + 1] * R1[a];
    for (int a = 0; a < 3; a++)
      for (int j = 0; j < 3; j++) par[k++] = t3[j] * R1[a];
    for (int a = 0; a < 3; a++) par[k++] = R1[a];
    bool fin = true;
    for (int j = 0; j < 41; j++) fin = fin && par[j] == par[j] && fabs(par[j]) < 1e300;
    return fin;
  }
};

#if defined(__HIPCC__)
// K1: one 256-thread workgroup per hypothesis; the 31 x 31 system and V in LDS (2 x 31 x 31 x 8 B)
__global__ __launch_bounds__(256) void k_estimate_phantom(const double *__restrict__ data, size_t stride,
                                                          size_t nobs,
                                                          const uint32_t *__restrict__ subsets,
                                                          uint32_t H, double *__restrict__ hparams,
                                                          uint8_t *__restrict__ valid) {
  typedef PhantomModel M;
  constexpr int N = 31, LDA = 31;
  __shared__ double A[N * LDA], V[N * LDA], recs[N][M::ND], x[N];
  __shared__ int s_bad, s_npos;
  const int tid = threadIdx.x;
  const uint32_t h = blockIdx.x;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int idx = tid; idx < N * M::ND; idx += 256) {
    int l = idx / M::ND, c = idx % M::ND;
    size_t i = subsets[(size_t)h * N + l];
    if (i >= nobs) {
      s_bad = 1;
      i = 0;
    }
    recs[l][c] = (c == 12) ? 0.0 : data[i * stride + c];
  }
  __syncthreads();
  for (int idx = tid; idx < N * N; idx += 256) {
    int l = idx / N, c = idx % N;  // row (frame) l, column c
    A[c * LDA + l] = M::row_entry(recs[l], c);
  }
  __syncthreads();
  int npos = 0;
  block_null_vector<256>(N, N, A, LDA, V, LDA, x, &npos);
  if (tid == 0) s_npos = npos;
  __syncthreads();
  if (tid == 0) {
    double par[M::P];
    bool ok = !s_bad && s_npos == N && M::finish(x, par);
    const double qnan = __builtin_nan("");
    for (int j = 0; j < M::P; j++) hparams[(size_t)h * M::SP + j] = ok ? par[j] : qnan;
    valid[h] = ok ? 1 : 0;
  }
}

// the data rows a_i as a dense m x 32 matrix (column 31 = 0) for the matrix-core SYRK of the dense model
__global__ __launch_bounds__(256) void k_phantom_rows(const double *__restrict__ data, size_t stride,
                                                      size_t n, double *__restrict__ rows) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // one thread per (frame, 4 columns)
  const size_t i = idx >> 3;
  const int c0 = (int)(idx & 7) * 4;
  if (i >= n) return;
  double x[PhantomModel::ND];
  PhantomModel::load(data + i * stride, ModelConsts(), x);
#pragma unroll
  for (int k = 0; k < 4; k++) rows[i * 32 + c0 + k] = c0 + k < 31 ? PhantomModel::row_entry(x, c0 + k) : 0.0;
}
#endif

// ---- host side of the fits (31-dimensional: like the LM control flow and the RANSAC replay) -----------
struct PhDual {  // value + derivatives with respect to the 11 minimal parameters
  double v, d[11];
};
inline PhDual ph_const(double c) {
  PhDual r;
  r.v = c;
  for (int i = 0; i < 11; i++) r.d[i] = 0.0;
  return r;
}
inline PhDual ph_var(double c, int k) {
  PhDual r = ph_const(c);
  r.d[k] = 1.0;
  return r;
}
inline PhDual operator*(const PhDual &a, const PhDual &b) {
  PhDual r;
  r.v = a.v * b.v;
  for (int i = 0; i < 11; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
inline PhDual operator+(const PhDual &a, const PhDual &b) {
  PhDual r;
  r.v = a.v + b.v;
  for (int i = 0; i < 11; i++) r.d[i] = a.d[i] + b.d[i];
  return r;
}
inline PhDual operator-(const PhDual &a, const PhDual &b) {
  PhDual r;
  r.v = a.v - b.v;
  for (int i = 0; i < 11; i++) r.d[i] = a.d[i] - b.d[i];
  return r;
}
inline PhDual operator-(const PhDual &a) { return ph_const(0.0) - a; }
inline PhDual ph_sin(const PhDual &a) {
  PhDual r;
  r.v = sin(a.v);
  for (int i = 0; i < 11; i++) r.d[i] = cos(a.v) * a.d[i];
  return r;
}
inline PhDual ph_cos(const PhDual &a) {
  PhDual r;
  r.v = cos(a.v);
  for (int i = 0; i < 11; i++) r.d[i] = -sin(a.v) * a.d[i];
  return r;
}
// e(x): the 31 coefficients the data rows multiply (.cxx:566-631), with derivatives
inline void phantom_e(const double *x, PhDual e[31]) {
  PhDual p[11];
  for (int i = 0; i < 11; i++) p[i] = ph_var(x[i], i);
  PhDual cy = ph_cos(p[0]), sy = ph_sin(p[0]), cx = ph_cos(p[1]), sx = ph_sin(p[1]);
  PhDual R1[3] = {-sy, cy * sx, cy * cx};
  PhDual cz = ph_cos(p[6]), sz = ph_sin(p[6]);
  cy = ph_cos(p[7]), sy = ph_sin(p[7]);
  cx = ph_cos(p[8]), sx = ph_sin(p[8]);
  PhDual c1[3] = {cz * cy, sz * cy, -sy};                                 // first column of R3
  PhDual c2[3] = {cz * sy * sx - sz * cx, sz * sy * sx + cz * cx, cy * sx};  // second column
  int k = 0;
  for (int a = 0; a < 3; a++)
    for (int j = 0; j < 3; j++) e[k++] = p[9] * c1[j] * R1[a];
  for (int a = 0; a < 3; a++)
    for (int j = 0; j < 3; j++) e[k++] = p[10] * c2[j] * R1[a];
  for (int a = 0; a < 3; a++)
    for (int j = 0; j < 3; j++) e[k++] = p[3 + j] * R1[a];
  for (int a = 0; a < 3; a++) e[k++] = R1[a];
  e[k++] = p[2];
}
// LM block {sum f^2, J^T J (upper, row-major), J^T f} at x from the Gram matrix G (31 x 31, full)
inline void phantom_lm_block(const double *G, const double *x, double *blk) {
  PhDual e[31];
  phantom_e(x, e);
  double Ge[31], GE[31][11];
  for (int i = 0; i < 31; i++) {
    double t = 0;
    for (int j = 0; j < 31; j++) t += G[i * 31 + j] * e[j].v;
    Ge[i] = t;
    for (int q = 0; q < 11; q++) {
      double u = 0;
      for (int j = 0; j < 31; j++) u += G[i * 31 + j] * e[j].d[q];
      GE[i][q] = u;
    }
  }
  double cost = 0;
  for (int i = 0; i < 31; i++) cost += e[i].v * Ge[i];
  int k = 0;
  blk[k++] = cost > 0 ? cost : 0.0;
  for (int p = 0; p < 11; p++)
    for (int q = p; q < 11; q++) {
      double t = 0;
      for (int i = 0; i < 31; i++) t += e[i].d[p] * GE[i][q];
      blk[k++] = t;
    }
  for (int p = 0; p < 11; p++) {
    double t = 0;
    for (int i = 0; i < 31; i++) t += e[i].d[p] * Ge[i];
    blk[k++] = t;
  }
}

}  // namespace lsqr
