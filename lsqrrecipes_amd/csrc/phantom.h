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
// one-sided Jacobi SVD in LDS (one wave per hypothesis).  The sign of a singular vector is arbitrary: the
// reference fixes the scale factor's sign arbitrarily too (.cxx:215-218), it flips T1 and leaves
// agree() -- a squared quantity -- unchanged.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif
#include <string.h>

#include "lm_core.h"
#include "models.h"
#if defined(__HIPCC__)
#include "wave_linalg.h"
#endif

namespace lsqr {

struct PhantomModel {
  enum { ND = 15, K = 31, P = 41, SP = 41, REC = 15, PPL = 2, IS_DENSE = 0, IS_US = 0, IS_PHANTOM = 1 };
  enum { NMOM = 1, NE = 31, NX = 11 };
  // packed fp32 pre-filter of the scan (us_kernels.h: k_scan_us_f32): c0(3) c1(3) t3(3) R1(3) t1_z 0 tin tout
  enum { SPF = 16, NF32 = 16, TIN = 14, NFLD = 14 };
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

  // ---- packed fp32 pre-filter ---------------------------------------------------------------------------
  // The 30 products are m_x R3[:,0] R1[a], m_y R3[:,1] R1[a], t3 R1[a] (.cxx:325-354), so the error factors
  // as  err = R1 . (R2 p + t2) + t1_z,  p = u c0 + v c1 + t3  with c0 = m_x R3[:,0], c1 = m_y R3[:,1]:
  // 18 fused fp32 operations per frame instead of 62 fp64 ones.  c0, c1 are taken from the products of the
  // largest |R1[a]| (>= 1/sqrt 3) and the factorisation is CHECKED against all 30 products; a parameter
  // vector that does not factor (relative 1e-11) gets tin = -inf / tout = +inf, i.e. the exact predicate for
  // every frame.  Bound: with X >= |u|, |v|, |t2|, Rm >= |R2 entries| (k_absmax),
  //   s_j = (|c0_j| + |c1_j|) X + |t3_j|,  Q = Rm (s_0 + s_1 + s_2) + X,  W = (|R1_0|+|R1_1|+|R1_2|) Q + |t1_z|
  // bounds every partial sum of either evaluation; the fp32 evaluation (inputs rounded once, 18 fma in
  // chains of depth <= 9) is within 20 u32 W of the exact value, the reference's fp64 31-term sum (each
  // term two products of rounded factors) within 64 u64 W, the refactoring (one division, one product per
  // coefficient) within 8 u64 W.  |err| is compared with T = c.thr (|s| < T <=> fl(s s) < delta^2,
  // models.h square_threshold):  v < T - E => agrees,  v >= T + E => does not.
  static LSQR_HD void prepare_f32(const double *sp, const ModelConsts &c, float *f) {
    const double X = c.absmax, Rm = c.absmax_rot, u32 = 5.9604644775390625e-08, u64 = 1.1102230246251565e-16;
    const double *R1 = sp + 38;
    int a = fabs(R1[0]) >= fabs(R1[1]) ? 0 : 1;
    if (fabs(R1[2]) > fabs(R1[a])) a = 2;
    double c0[3], c1[3], t3[3], scale = 0.0, dev = 0.0;
    bool finite = true;
    for (int j = 0; j < P; j++) finite = finite && sp[j] == sp[j];
    for (int j = 0; j < 3; j++) {
      c0[j] = sp[11 + 3 * a + j] / R1[a];
      c1[j] = sp[20 + 3 * a + j] / R1[a];
      t3[j] = sp[29 + 3 * a + j] / R1[a];
    }
    for (int b = 0; b < 3; b++)
      for (int j = 0; j < 3; j++) {
        const double q0 = c0[j] * R1[b], q1 = c1[j] * R1[b], q2 = t3[j] * R1[b];
        const double d0 = fabs(sp[11 + 3 * b + j] - q0), d1 = fabs(sp[20 + 3 * b + j] - q1),
                     d2 = fabs(sp[29 + 3 * b + j] - q2);
        // deviation weighted as the term enters the sum (u, v <= X; 1)
        dev = fmax(dev, fmax(fmax(d0, d1) * X, d2));
        scale = fmax(scale, fmax(fmax(fabs(q0), fabs(q1)) * X, fabs(q2)));
      }
    double S = 0.0;
    for (int j = 0; j < 3; j++) S += (fabs(c0[j]) + fabs(c1[j])) * X + fabs(t3[j]);
    const double Q = Rm * S + X;
    const double W = (fabs(R1[0]) + fabs(R1[1]) + fabs(R1[2])) * Q + fabs(sp[2]);
    const double E = 1.01 * ((20.0 * u32 + 72.0 * u64) * W);
    const double T = c.thr;
    // (r04: no smallness condition on E -- |v - |err_ref|| <= E makes both implications above hold for ANY E; with E >=
    // T there simply are no certain inliers.  r03 switched the filter off for E > T / 4, which sends every frame of a
    // hypothesis with large scale factors to the exact predicate: the US scan lost a third of its time that way.)
    const bool ok = finite && X <= 1e15 && X >= 1e-10 && Rm <= 1e15 && W <= 1e30 && fabs(R1[a]) >= 0.5 &&
                    dev <= 1e-11 * scale && T > 1e-15 && T <= 1e15;
    for (int j = 0; j < 3; j++) {
      f[j] = (float)c0[j];
      f[3 + j] = (float)c1[j];
      f[6 + j] = (float)t3[j];
      f[9 + j] = (float)R1[j];
    }
    f[12] = (float)sp[2];
    f[13] = 0.0f;
    f[14] = (ok && T - E > 0.0) ? PlaneModel<3>::round_down_f32(T - E) : -INFINITY;
    f[15] = ok ? PlaneModel<3>::round_up_f32(T + E) : INFINITY;
    if (!finite) f[14] = f[15] = __builtin_nanf("");  // NaN model: never agrees
  }
#if defined(__HIPCC__)
  // xs: 14 packed fields of two frames (R2 0..8, t2 9..11, u 12, v 13); f as above (scalars) -> |err|
  static __device__ inline v2f filter_value_f32(const v2f *xs, const float *f) {
    v2f p[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      v2f t = {f[6 + j], f[6 + j]};
      t = __builtin_elementwise_fma(xs[13], (v2f){f[3 + j], f[3 + j]}, t);
      p[j] = __builtin_elementwise_fma(xs[12], (v2f){f[j], f[j]}, t);
    }
    v2f e = {f[12], f[12]};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      v2f q = __builtin_elementwise_fma(xs[3 * i + 2], p[2], xs[9 + i]);
      q = __builtin_elementwise_fma(xs[3 * i + 1], p[1], q);
      q = __builtin_elementwise_fma(xs[3 * i], p[0], q);
      e = __builtin_elementwise_fma(q, (v2f){f[9 + i], f[9 + i]}, e);
    }
    return __builtin_elementwise_abs(e);
  }
#endif
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
// K1: one workgroup of T threads (default: one wave) per hypothesis; the 31 x 31 system and V in LDS
// (2 x 31 x 31 x 8 B)
template <int T>
__global__ __launch_bounds__(T) void k_estimate_phantom(const double *__restrict__ data, size_t stride,
                                                          size_t nobs,
                                                          const uint32_t *__restrict__ subsets,
                                                          uint32_t H, double *__restrict__ hparams,
                                                          uint8_t *__restrict__ valid,
                                                          const uint8_t *__restrict__ only = nullptr) {
  typedef PhantomModel M;
  constexpr int N = 31, LDA = 31;
  __shared__ double A[N * LDA], V[N * LDA], recs[N][M::ND], x[N];
  __shared__ int s_bad, s_npos;
  const int tid = threadIdx.x;
  // behind k_estimate_phantom_lu (`only`): a few hundred workgroups walk the batch and solve what the fast path refused
  // -- one workgroup per hypothesis, 4096 of them returning at once, took 63 us of the step
  for (uint32_t h = blockIdx.x; h < H; h += gridDim.x) {
  if (only && !only[h]) continue;  // (workgroup-uniform)
  __syncthreads();  // the previous hypothesis of this workgroup is written
  if (tid == 0) s_bad = 0;
  __syncthreads();
  for (int idx = tid; idx < N * M::ND; idx += T) {
    int l = idx / M::ND, c = idx % M::ND;
    size_t i = subsets[(size_t)h * N + l];
    if (i >= nobs) {
      s_bad = 1;
      i = 0;
    }
    recs[l][c] = (c == 12) ? 0.0 : data[i * stride + c];
  }
  __syncthreads();
  for (int idx = tid; idx < N * N; idx += T) {
    int l = idx / N, c = idx % N;  // row (frame) l, column c
    A[c * LDA + l] = M::row_entry(recs[l], c);
  }
  __syncthreads();
  int npos = 0;
  block_null_vector<T, N, N>(N, N, A, LDA, V, LDA, x, &npos);
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
}

// K1, fast path (r05): the null vector by LU + inverse iteration, one wave per hypothesis, four per workgroup.
//
// k_estimate_phantom above takes the right singular vector of the smallest singular value of the 31 x 31 system from a
// one-sided Jacobi SVD: ~295 rounds of a dependent chain through LDS per hypothesis, 1.57 ms per 4096 -- half of a
// plane-phantom step.  Only ONE singular vector is wanted, and on the bench's frames sigma_31 / sigma_30 is
// 0.2 in the median, 0.6 at the 90th percentile (all-inlier subsets: 1e-5), so inverse iteration on A^T A converges
// by (sigma_31 / sigma_30)^2 per step: 6 steps in the median, ~20 at the 90th percentile.  A^T A is never formed
// (sigma_1 / sigma_30 reaches 6e5: its square would cost the vector ten digits): with P A = L U,
//     (A^T A)^-1 v = U^-1 L^-1 L^-T U^-T v                 (the permutation cancels)
// four triangular solves per step.  Lane k holds row k of the factors (pivot order) AND column k (a transposed copy,
// through LDS once): every solve is 31 steps of {broadcast one lane's value with v_readlane, one fma on the lanes
// behind it}, nothing but registers.  The elimination is wave_gepp_solve_reg64's (rows keep their lanes, positions are
// exchanged).  A pivot below 1e-14 max|A| (an exactly dependent subset: the reference's "rank < 31"), a non-finite
// entry or a system that has not converged to 1e-9 after `max_iter` steps is marked in `refused` and goes through the
// Jacobi kernel behind this one, which then returns at once for everything else.  The vector is the same vector --
// 1e-9 against the SVD's on the bench's subsets (tests/test_gpu_phantom_estimate.py: against the oracle's SVD at 1e-6,
// and the Jacobi kernel's); votes are counted on the device's own models either way.
__global__ __launch_bounds__(256) void k_estimate_phantom_lu(const double *__restrict__ data, size_t stride, size_t nobs,
                                                             const uint32_t *__restrict__ subsets, uint32_t H,
                                                             double *__restrict__ hparams, uint8_t *__restrict__ valid,
                                                             uint8_t *__restrict__ refused, int max_iter,
                                                             unsigned long long *__restrict__ dbg = nullptr) {
  typedef PhantomModel M;
  constexpr int N = 31, PT = 33;
  const unsigned long long tdbg0 = wall_clock64();  // odd pitch: the transposition's column reads hit distinct banks
  __shared__ double s_t[4][N * PT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t h = blockIdx.x * 4 + wave;
  if (h >= H) return;  // (whole waves: no workgroup barrier below)
  double *T = s_t[wave];
  const bool row = lane < N;
  bool bad = false;
  double a[N];
  {
    size_t i = row ? subsets[(size_t)h * N + lane] : 0;
    if (i >= nobs) {
      bad = true;
      i = 0;
    }
    double rec[M::ND];
#pragma unroll
    for (int c = 0; c < M::ND; c++) rec[c] = (c == 12) ? 0.0 : data[i * stride + c];
#pragma unroll
    for (int c = 0; c < N; c++) a[c] = row ? M::row_entry(rec, c) : 0.0;
  }
  bad = __ballot(bad && row) != 0;
  double am = 0.0;
#pragma unroll
  for (int c = 0; c < N; c++) {
    const double v = fabs(a[c]);
    am = v > am ? v : (v == v ? am : INFINITY);
  }
  am = wave_max(am);
  const double tiny = 1e-14 * am;
  bool refuse = !(am <= 1e150) || !(am > 0.0);
  // ---- P A = L U, rows on their lanes, positions exchanged (wave_linalg.h: wave_gepp_solve_reg64) --------------------
  int pos = row ? lane : 64 + lane;
#pragma unroll
  for (int k = 0; k < N; k++) {
    double v = (row && pos >= k) ? fabs(a[k]) : -1.0;
    int idx = pos;
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o);
      const int oi = __shfl_xor(idx, o);
      if (ov > v || (ov == v && oi < idx)) v = ov, idx = oi;
    }
    const int p = __builtin_amdgcn_readfirstlane(idx);
    refuse = refuse || !(v > tiny);
    const int lp = __builtin_ctzll(__ballot(pos == p) | (1ull << 63));
    const double rpiv = 1.0 / wave_readlane_f64(a[k], lp);
    if (p != k) pos = pos == p ? k : pos == k ? p : pos;
    const bool below = row && pos > k;
    double mult = 0.0;
    if (below) {
      mult = a[k] * rpiv;
      a[k] = mult;
    }
#pragma unroll
    for (int j = k + 1; j < N; j++) {
      const double akj = wave_readlane_f64(a[j], lp);
      if (below) a[j] = fma(-mult, akj, a[j]);
      if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // rows into pivot order (lane k <- the row at position k), then the transposed copy through LDS
  {
    int src = lane;
#pragma unroll
    for (int k = 0; k < N; k++) {
      const int lk = __builtin_ctzll(__ballot(pos == k) | (1ull << 63));
      if (lane == k) src = lk;
    }
#pragma unroll
    for (int c = 0; c < N; c++) a[c] = __shfl(a[c], src);
  }
  double t[N];
  if (row) {
#pragma unroll
    for (int c = 0; c < N; c++) T[lane * PT + c] = a[c];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < N; c++) t[c] = row ? T[c * PT + lane] : 0.0;
  __builtin_amdgcn_wave_barrier();
  // 1 / U_kk on lane k
  double dk = 1.0;
#pragma unroll
  for (int c = 0; c < N; c++)
    if (lane == c) dk = a[c];
  const double rd = 1.0 / dk;
  // ---- inverse iteration --------------------------------------------------------------------------------------------
  const unsigned long long tdbg1 = wall_clock64();
  // v -> (A^T A)^-1 v = U^-1 L^-1 L^-T U^-T v: four triangular solves, lane i = component i
  auto inv_apply = [&](double r) -> double {
    // U^T s = v      (forward; U_ji on lane i is t[j])
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double sj = wave_readlane_f64(r * rd, j);
      r = lane == j ? sj : (lane > j ? fma(-t[j], sj, r) : r);
    }
    // L^T u = s      (backward, unit diagonal; L_ji on lane i is t[j])
#pragma unroll
    for (int j = N - 1; j > 0; j--) {
      const double uj = wave_readlane_f64(r, j);
      r = lane < j ? fma(-t[j], uj, r) : r;
    }
    // L c = u        (forward; L_ij on lane i is a[j])
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      const double cj = wave_readlane_f64(r, j);
      r = (row && lane > j) ? fma(-a[j], cj, r) : r;
    }
    // U v' = c       (backward; U_ij on lane i is a[j])
#pragma unroll
    for (int j = N - 1; j >= 0; j--) {
      const double vj = wave_readlane_f64(r * rd, j);
      r = lane == j ? vj : (lane < j ? fma(-a[j], vj, r) : r);
    }
    return row ? r : 0.0;
  };
  // v -> A^T A v = U^T L^T L U v through the same factors (the permutation cancels): four triangular PRODUCTS -- every
  // step reads the same input vector, no dependent chain as in the solves
  auto fwd_apply = [&](double x) -> double {
    double y = 0.0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double xj = wave_readlane_f64(x, j);
      y = lane <= j ? fma(a[j], xj, y) : y;  // U x
    }
    x = row ? y : 0.0;
#pragma unroll
    for (int j = 0; j < N - 1; j++) {
      const double xj = wave_readlane_f64(x, j);
      y = (row && lane > j) ? fma(a[j], xj, y) : y;  // L x (unit diagonal: y starts as x)
    }
    x = row ? y : 0.0;
#pragma unroll
    for (int j = 1; j < N; j++) {
      const double xj = wave_readlane_f64(x, j);
      y = lane < j ? fma(t[j], xj, y) : y;  // L^T x
    }
    x = row ? y : 0.0;
    y = 0.0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const double xj = wave_readlane_f64(x, j);
      y = lane >= j ? fma(t[j], xj, y) : y;  // U^T x
    }
    return row ? y : 0.0;
  };
  double r = row ? 1.0 / sqrt((double)N) * (1.0 + 0.01 * lane) : 0.0;  // start: no symmetry the null vector could be orthogonal to
  bool conv = false;
  int it = 0, since = 0, pcg = 0;
  double pp = r, dprev = 0.0;
  if (!refuse) {
    for (; it < max_iter; it++) {
      const double prev = r;
      r = inv_apply(r);
      // unit length; (A^T A)^-1 is positive definite, so consecutive iterates do not flip their sign
      const double n2 = wave_sum(row ? r * r : 0.0);
      r = row ? r / sqrt(n2) : 0.0;
      double d = fabs(r - prev);
      d = wave_max(d);
      if (!(n2 > 0.0) || !(n2 < INFINITY)) break;  // overflow / breakdown: the Jacobi kernel decides
      // PHASE 1 only has to bring the vector close enough for the correction below to be linear (its square below
      // 1e-14): the distance to the limit is d rho / (1 - rho) with rho the decay per step, estimated from two steps
      const double rho_hat = dprev > 0.0 ? fmin(d / dprev, 0.999) : 0.999;
      if (it >= 1 && d < 1e-7 && d * rho_hat / (1.0 - rho_hat) < 1e-7) {
        conv = true;
        break;
      }
      dprev = d;
      // Aitken's extrapolation of the vector sequence: with sigma_31 / sigma_30 near one the error is one mode decaying
      // by rho = (sigma_31 / sigma_30)^2 per step; three clean iterates give rho and the limit x + rho / (1 - rho) dx.
      // On the bench's subsets: plain iteration 99 % within 61 steps, 15 of 3000 beyond 80, the slowest beyond 300;
      // with the extrapolation 99 % within 20, the slowest 79 (one wave at the limit used to BE the kernel's time).
      if (++since >= 3) {
        const double d1 = prev - pp, d2 = r - prev;
        const double den = wave_sum(row ? d1 * d1 : 0.0), num = wave_sum(row ? d2 * d1 : 0.0);
        const double rho = num / den;
        if (den > 0.0 && rho > 0.3 && rho < 0.999) {
          r = fma(rho / (1.0 - rho), d2, r);
          const double m2 = wave_sum(row ? r * r : 0.0);
          r = row ? r / sqrt(m2) : 0.0;
          since = 0;
          dprev = 0.0;
        }
      }
      pp = prev;
    }
  }
  // ---- PHASE 2: one correction from an ACCURATE residual ------------------------------------------------------------
  // The fixed point of the iteration above is the singular vector of a matrix eps ||A|| away from A: eps sigma_1 /
  // sigma_30 from the wanted vector -- 1e-11 in the median and up to 6e-10 on rescaled uploads (sigma_1 / sigma_30 to
  // 3e6), and M::finish divides by the vector's SMALL components (R1's entries are 1e-3 of its largest): 1e-9 in
  // the vector was 1e-6 .. 1e-4 in the 41 parameters for one hypothesis in 800 (soak against the oracle, r05), where
  // the Jacobi SVD -- accurate relative to each column's scale -- stays at 1e-11.  So: g = A^T (A v) from the ORIGINAL
  // rows in double-double arithmetic (error-free products through fma, two-sum accumulation: A v must be good to
  // 1e-13 sigma_30 beside terms of size sigma_1), the Rayleigh quotient lambda = v.g, and the correction equation
  //     P (A^T A - lambda) P e = P (g - lambda v),      P = I - v v^T,          v <- (v - e) / |v - e|
  // solved by conjugate gradients preconditioned with P (A^T A)^-1 P through the factors (the operator through them
  // too: e is small, the factors' eps is relative to e).  The preconditioned spectrum is 1 - (sigma_31 / sigma_j)^2:
  // one cluster at 1 and a few values below it -- 1 step for 45 % of the subsets, 2 for 47 %, never more than 4 in a
  // numpy model of this loop (1200 subsets over random uploads), after which the vector is within 2e-13 and the
  // parameters within 5e-10 of a long-double reference (LAPACK's own SVD: 9e-13 / 2e-9).
  if (!refuse && conv) {
    conv = false;
    // the original rows again (the elimination overwrote them): lane i -> T[i][*]
    {
      size_t i = row ? subsets[(size_t)h * N + lane] : 0;
      if (i >= nobs) i = 0;  // (such a subset is marked bad above and never reaches M::finish)
      double rec[M::ND];
#pragma unroll
      for (int c = 0; c < M::ND; c++) rec[c] = (c == 12) ? 0.0 : data[i * stride + c];
      if (row) {
#pragma unroll
        for (int c = 0; c < N; c++) T[lane * PT + c] = M::row_entry(rec, c);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // r1 = A v, lane = row, double-double
    double r1h = 0.0, r1l = 0.0;
#pragma unroll
    for (int c = 0; c < N; c++) {
      const double ac = row ? T[lane * PT + c] : 0.0, vc = wave_readlane_f64(r, c);
      const double p = ac * vc, e = fma(ac, vc, -p);
      const double s = r1h + p, z = s - r1h;
      r1l += ((r1h - (s - z)) + (p - z)) + e;
      r1h = s;
    }
    {
      const double s = r1h + r1l;
      r1l = r1l - (s - r1h);
      r1h = s;
    }
    // g = A^T r1, lane = column, double-double
    double gh = 0.0, gl = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const double ac = row ? T[i * PT + lane] : 0.0;
      const double bh = wave_readlane_f64(r1h, i), bl = wave_readlane_f64(r1l, i);
      const double p = ac * bh, e = fma(ac, bh, -p) + ac * bl;
      const double s = gh + p, z = s - gh;
      gl += ((gh - (s - z)) + (p - z)) + e;
      gh = s;
    }
    const double g = row ? gh + gl : 0.0;
    const double lam = wave_sum(g * r);
    auto proj = [&](double x) -> double { return fma(-wave_sum(x * r), r, x); };
    double rr = proj(fma(-lam, r, g));
    double z = proj(inv_apply(rr));
    double pdir = z, e = 0.0;
    double rz = wave_sum(rr * z);
    const double rz0 = rz;
    bool fine = rz0 == rz0 && rz0 < INFINITY;
    if (fine && rz0 > 0.0) {
      fine = false;
      for (; pcg < 8; pcg++) {
        const double ap = proj(fma(-lam, pdir, fwd_apply(pdir)));
        const double pap = wave_sum(pdir * ap);
        if (!(pap > 0.0)) break;  // not positive definite on v's complement: lambda is not the smallest -- Jacobi decides
        const double alpha = rz / pap;
        e = fma(alpha, pdir, e);
        rr = fma(-alpha, ap, rr);
        z = proj(inv_apply(rr));
        const double rz2 = wave_sum(rr * z);
        if (!(rz2 == rz2)) break;
        if (rz2 <= 1e-10 * rz0) {  // the preconditioned residual 1e-5 of the start's
          fine = true;
          pcg++;
          break;
        }
        pdir = fma(rz2 / rz, pdir, z);
        rz = rz2;
      }
    }
    if (fine) {
      r = r - e;
      const double m2 = wave_sum(row ? r * r : 0.0);
      r = row ? r / sqrt(m2) : 0.0;
      // e beyond what phase 1 left (1e-5 with the margin) means the linearisation does not hold
      conv = wave_max(fabs(e)) < 1e-5 && m2 > 0.0 && m2 < INFINITY;
    }
  }
  const bool ok_vec = !refuse && conv;
  const unsigned long long tdbg2 = wall_clock64();
  if (dbg && lane == 0) {
    dbg[(size_t)h * 4 + 0] = (unsigned long long)it | ((unsigned long long)(refuse ? 1 : 0) << 32) |
                             ((unsigned long long)(conv ? 1 : 0) << 33) | ((unsigned long long)pcg << 40);
    dbg[(size_t)h * 4 + 1] = tdbg1 - tdbg0;
    dbg[(size_t)h * 4 + 2] = tdbg2 - tdbg1;
  }
  if (!ok_vec) {  // the Jacobi kernel behind this one takes the hypothesis
    if (lane == 0) refused[h] = bad ? 0 : 1;
    if (bad && lane == 0) {
      const double qnan = __builtin_nan("");
      for (int j = 0; j < M::P; j++) hparams[(size_t)h * M::SP + j] = qnan;
      valid[h] = 0;
    }
    return;
  }
  if (row) T[lane] = r;
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    double x[N], par[M::P];
    for (int j = 0; j < N; j++) x[j] = T[j];
    const bool ok = !bad && M::finish(x, par);
    const double qnan = __builtin_nan("");
    for (int j = 0; j < M::P; j++) hparams[(size_t)h * M::SP + j] = ok ? par[j] : qnan;
    valid[h] = ok ? 1 : 0;
    refused[h] = 0;
    if (dbg) dbg[(size_t)h * 4 + 3] = wall_clock64() - tdbg2;
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
// LM block {sum f^2, J^T J (upper, row-major), J^T f} at x from the Gram matrix G (31 x 31, full).
// The reference minimises WITHOUT a gradient (.cxx:552: vnl_levenberg_marquardt::minimize_without_gradient, i.e.
// MINPACK lmdif): its Jacobian is the forward difference  J(:, j) = (f(x + h_j e_j) - f(x)) / h_j,
// h_j = sqrt(eps) |x_j| (sqrt(eps) when x_j = 0; fdjac2 with vnl's epsfcn = xtol * 0.001 < eps).  Every residual is
// a_i . e(x), so that Jacobian is  A dE  with  dE(:, j) = (e(x + h_j e_j) - e(x)) / h_j  -- the same forward
// difference taken on the 31 coefficients instead of on the N residuals: same truncation error term by term, and the
// iterates follow the reference's (fd = true, the default of the fit).  fd = false: exact derivatives (r01 - r04).
inline void phantom_lm_block(const double *G, const double *x, double *blk, bool fd = true) {
  PhDual e[31];
  phantom_e(x, e);
  if (fd) {
    const double eps = 1.4901161193847656e-08;  // sqrt(2^-52)
    for (int q = 0; q < 11; q++) {
      double xx[11];
      for (int i = 0; i < 11; i++) xx[i] = x[i];
      double h = eps * fabs(x[q]);
      if (h == 0.0) h = eps;
      xx[q] = x[q] + h;
      PhDual eh[31];
      phantom_e(xx, eh);
      for (int i = 0; i < 31; i++) e[i].d[q] = (eh[i].v - e[i].v) / h;
    }
  }
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


struct PhantomFit {
  int ok, lm_info, lm_nfev;
  double cost;  // sum of squared residuals at the result
  double params[41];
};

inline void phantom_unpack(const double *blk, double *G) {  // upper triangle, row-major -> full 31 x 31
  for (int i = 0; i < 31; i++)
    for (int j = i; j < 31; j++) G[i * 31 + j] = G[j * 31 + i] = blk[i * 31 - i * (i - 1) / 2 + (j - i)];
}

// PlanePhantom...Estimator.cxx:357-453: Levenberg-Marquardt on the 11 minimal parameters, every
// evaluation a function of the Gram matrix G (31 x 31, full); s: initialised by lm_init
inline void phantom_lm(const double *G, LmState &s, PhantomFit *out) {
  double blk78[LM_MOM_MAX], par[41];
  for (;;) {
    phantom_lm_block(G, s.xtrial, blk78);
    if (!lm_advance(s, blk78)) break;
  }
  const bool ok = s.info >= 1 && s.info <= 4;  // vnl_levenberg_marquardt::minimize -> true
  for (int j = 0; j < 11; j++) par[j] = s.x[j];
  PhantomModel::expand(par);
  out->ok = ok ? 1 : 0;
  out->lm_info = s.info;
  out->lm_nfev = s.nfev;
  out->cost = s.fnorm * s.fnorm;
  for (int j = 0; j < 41; j++) out->params[j] = par[j];
}

// leastSquaresEstimate() from the Gram block {upper triangle of G (496), number of frames}
inline void phantom_fit_block(const double *blk, bool iterative, PhantomFit *out) {
  memset(out, 0, sizeof *out);
  const int N = 31;
  if (!(blk[496] >= 31.0)) return;  // .cxx:139-141: fewer than 31 frames
  double G[N * N], a[N * N], w[N], v[N * N], x[N], par[41];
  phantom_unpack(blk, G);
  for (int i = 0; i < N * N; i++) {
    if (!(fabs(G[i]) <= 1e300)) return;  // non-finite data: no estimate
    a[i] = G[i];
  }
  sym_eig(N, a, w, v);  // ascending: column 0 belongs to the smallest singular value of the row matrix
  for (int j = 0; j < N; j++) x[j] = v[j * N];
  if (!PhantomModel::finish(x, par)) return;
  if (!iterative) {
    // cost at the RETURNED parameters (their derived products are rebuilt from the extracted angles and
    // scales, so they are not the scaled singular vector)
    double e[N], cost = 0;
    for (int j = 0; j < 30; j++) e[j] = par[11 + j];
    e[30] = par[2];
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) cost += e[i] * G[i * N + j] * e[j];
    out->ok = 1;
    out->cost = cost > 0 ? cost : 0.0;
    for (int j = 0; j < 41; j++) out->params[j] = par[j];
    return;
  }
  LmState s;
  lm_init(s, 11, par, 10e-16, 10e-16, 10e-16, 5000, 100.0);
  phantom_lm(G, s, out);
}

}  // namespace lsqr
