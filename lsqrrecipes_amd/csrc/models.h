// models.h -- per-model arithmetic of the hot path, written once as LSQR_HD so the device
// kernels (lsqr_hip.hip) and the host-compiled unit tests (tests/host_math/) share it.
//
// Bit-exactness contract: estimate() and agree() of plane / sphere / line follow the reference's
// operation order literally (file:line cited at each function) and this translation unit is
// compiled with -ffp-contract=off, so on gfx950 (IEEE fp64 add/mul/div/sqrt, correctly rounded)
// they produce the same bits as the reference's x86-64 build.  The final fits only owe 1e-6
// relative agreement and use the most accurate formulation available (shifted moments, fma).
#pragma once
#include <math.h>
#include <stdint.h>

#include "lm_core.h"
#include "small_linalg.h"

// The one-workgroup selection kernels (k_pick_* in cells.h / axis.h, k_ee_* in earlyexit.h) give each of their 1024
// threads 8 consecutive hypotheses and keep an alive bitmap of 256 words: H <= kSelCap.  k_plane_order sorts the batch
// in LDS: H <= kOrderCap.  The host only launches them inside these limits (static_asserts in lsqr_hip.hip tie
// kEeCap / kScanChunk to them) and the kernels return at once beyond them, so that a change of one constant cannot
// turn into an out-of-bounds write.
constexpr uint32_t kSelCap = 1024 * 8;
constexpr uint32_t kOrderCap = 4096;

namespace lsqr {

typedef float v2f __attribute__((ext_vector_type(2)));  // two observations per packed fp32 op

struct ModelConsts {
  double delta;     // constructor argument
  double delta_sq;  // delta*delta as the reference stores it (PlaneParametersEstimator.hxx:16)
  int dim;
  int ls_type;
  double thr;     // plane: smallest T >= 0 with fl(T*T) >= delta_sq, so that s*s < delta_sq <=> |s| < T
  double absmax;  // max |coordinate| over the uploaded observations (error band of the fp32 filter)
  double absmax_rot;  // US records: max |entry| of the rotation slots 0..8 (else = absmax)
  double aux;         // RAY: sin(minimalAngularDeviation)^2 (RayIntersection...Estimator.cxx:13-14)
};

// Exact restatement of "x*x < q" as "|x| < T" (fl(x*x) is monotone in |x|).
LSQR_HD double square_threshold(double q) {
  if (!(q > 0.0)) return 0.0;  // x*x < q is never true for q <= 0 (and for NaN)
  // smallest non-negative double T with fl(T*T) >= q: bisection over the bit patterns of the
  // non-negative doubles (their order is the order of their bit patterns)
  unsigned long long lo = 0ULL, hi = 0x7FEFFFFFFFFFFFFFULL;  // 0.0 (false) .. DBL_MAX
  double th;
  __builtin_memcpy(&th, &hi, 8);
  if (!(th * th >= q)) return INFINITY;  // unreachable for finite q
  while (hi - lo > 1) {
    unsigned long long mid = lo + (hi - lo) / 2;
    double t;
    __builtin_memcpy(&t, &mid, 8);
    if (t * t >= q) hi = mid;
    else lo = mid;
  }
  __builtin_memcpy(&th, &hi, 8);
  return th;
}

// Bisection over the bit patterns of the non-negative doubles for a monotone predicate.
LSQR_HD double bits_to_double(unsigned long long b) {
  double d;
  __builtin_memcpy(&d, &b, 8);
  return d;
}
// smallest non-negative finite double with pred true (pred: false ... false true ... true);
// returns +inf when pred(DBL_MAX) is false
template <class Pred>
LSQR_HD double first_true(const Pred &pred) {
  unsigned long long lo = 0ULL, hi = 0x7FEFFFFFFFFFFFFFULL;
  if (pred(0.0)) return 0.0;
  if (!pred(bits_to_double(hi))) return INFINITY;
  while (hi - lo > 1) {
    unsigned long long mid = lo + (hi - lo) / 2;
    if (pred(bits_to_double(mid))) hi = mid;
    else lo = mid;
  }
  return bits_to_double(hi);
}
// largest non-negative finite double with pred true (pred: true ... true false ... false);
// returns -1 when pred(0) is false
template <class Pred>
LSQR_HD double last_true(const Pred &pred) {
  unsigned long long lo = 0ULL, hi = 0x7FEFFFFFFFFFFFFFULL;
  if (!pred(0.0)) return -1.0;
  if (pred(bits_to_double(hi))) return bits_to_double(hi);
  while (hi - lo > 1) {
    unsigned long long mid = lo + (hi - lo) / 2;
    if (pred(bits_to_double(mid))) lo = mid;
    else hi = mid;
  }
  return bits_to_double(lo);
}
struct PredSubLt {  // fl(s - r) < d
  double r, d;
  LSQR_HD bool operator()(double s) const { return (s - r) < d; }
};
struct PredSubGt {  // fl(s - r) > d
  double r, d;
  LSQR_HD bool operator()(double s) const { return (s - r) > d; }
};
struct PredSqrtLe {  // fl(sqrt(x)) <= v
  double v;
  LSQR_HD bool operator()(double x) const { return sqrt(x) <= v; }
};
struct PredSqrtGe {  // fl(sqrt(x)) >= v
  double v;
  LSQR_HD bool operator()(double x) const { return sqrt(x) >= v; }
};

static const double kEPS = 2.220446049250313e-016;  // common/Epsilon.h:19
static const double kSphereEPS = 1e-9;              // SphereParametersEstimator.hxx:11

enum { MOM_MAX = 96 };  // largest moment block handled by the generic reduction

// ------------------------------------------------------------------------------------ plane
template <int D>
struct PlaneModel {
  enum { ND = D, K = D, P = 2 * D, SP = 2 * D, REC = D, PPL = 4, IS_DENSE = 0, IS_US = 0 };
  enum { NMOM = 1 + D + D * (D + 1) / 2 };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < D; i++) rec[i] = p[i];
  }

  // PlaneParametersEstimator.hxx:36-109 (D == 3: :48-69; point a = first drawn datum, :107-108).
  // D == 2 takes the reference's SVD null-vector branch (:70-104), restated in closed form
  // (normal = unit perpendicular of p1-p0; sign arbitrary as with any null vector).
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &, double *par) {
    if constexpr (D == 3) {
      double v1[3], v2[3];
      v1[0] = r[1][0] - r[0][0];
      v1[1] = r[1][1] - r[0][1];
      v1[2] = r[1][2] - r[0][2];
      v2[0] = r[2][0] - r[0][0];
      v2[1] = r[2][1] - r[0][1];
      v2[2] = r[2][2] - r[0][2];
      double nx = v1[1] * v2[2] - v1[2] * v2[1];
      double ny = v1[2] * v2[0] - v1[0] * v2[2];
      double nz = v1[0] * v2[1] - v1[1] * v2[0];
      double norm = sqrt(nx * nx + ny * ny + nz * nz);
      if (norm < kEPS) return false;
      par[0] = nx / norm;
      par[1] = ny / norm;
      par[2] = nz / norm;
    } else {
      double vx = r[1][0] - r[0][0], vy = r[1][1] - r[0][1];
      double norm = sqrt(vx * vx + vy * vy);
      if (norm < kEPS) return false;
      par[0] = -vy / norm;
      par[1] = vx / norm;
    }
    for (int i = 0; i < D; i++) par[D + i] = r[0][i];
    return true;
  }

  // PlaneParametersEstimator.hxx:196-203
  // The reference accumulates from 0.0 (0.0 + x == x) and tests s*s < delta^2; |s| < c.thr is the
  // same predicate (square_threshold) with two fp64 operations fewer per observation.
  static LSQR_HD double signed_dist(const double *sp, const double *x) {
    double s = sp[0] * (x[0] - sp[D]);
    for (int i = 1; i < D; i++) s += sp[i] * (x[i] - sp[D + i]);
    return s;
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return fabs(signed_dist(sp, x)) < c.thr;
  }
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &) {
    return fabs(signed_dist(sp, x));
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}

  // ---- fp32 pre-filter (k_scan_plane_f32) ------------------------------------------------------
  // The filter evaluates s32 = fma(x0,n0, fma(x1,n1, fma(x2,n2, -c))) with c = n.a, all inputs rounded
  // to fp32 (u = 2^-24, every fp32 operation correctly rounded, X = max |coordinate| over the
  // observations, so |x_i| <= X).  Per hypothesis:
  //   inputs   |x32_i n32_i - x_i n_i| <= 2u|n_i|X (+u^2),  |c32 - c| <= u|c|
  //   roundings of the three fma results: u(X|n2| + |c|), u(X(|n1|+|n2|) + |c|), u|s32| (negligible)
  //   =>  |s32 - n.(x-a)| <= u [ X (2 sum|n_i| + |n1| + 2|n2|) + 3|c| ] =: B
  // and the reference's fp64 s is within 1e-14 X of n.(x-a).  With E = 1.01 B + 1e-12 X:
  //     |s32| <  T - E  =>  |s| < T   (certainly agrees)
  //     |s32| >= T + E  =>  |s| >= T  (certainly does not)
  // and only observations in the band in between are re-evaluated with the exact fp64 formula.
  enum { NF = 4, SPF = 12, FGRAN = 0 };  // (n0,n0) (n1,n1) (n2,n2) (-c,-c) tin tout 0 0: pairs feed v_pk_* directly
  static LSQR_HD float round_down_f32(double v) {
    float f = (float)v;
    if ((double)f > v) f = nextafterf(f, -INFINITY);
    return f;
  }
  static LSQR_HD float round_up_f32(double v) {
    float f = (float)v;
    if ((double)f < v) f = nextafterf(f, INFINITY);
    return f;
  }
  static LSQR_HD void prepare_f32(const double *sp, const ModelConsts &c, float *f) {
    const double X = c.absmax, u = 5.9604644775390625e-08;
    double cc = 0.0, s1 = 0.0;
    for (int i = 0; i < 12; i++) f[i] = 0.0f;
    bool ok = X >= 1e-10 && X <= 1e15;
    for (int i = 0; i < D; i++) {
      ok = ok && fabs(sp[i]) <= 1.0000001 && fabs(sp[D + i]) <= X;
      f[2 * i] = f[2 * i + 1] = (float)sp[i];
      cc += sp[i] * sp[D + i];
      s1 += fabs(sp[i]);
    }
    const double n1 = fabs(sp[1]), n2 = D == 3 ? fabs(sp[D - 1]) : 0.0;
    const double B = u * (X * (2.0 * s1 + (D == 3 ? n1 + 2.0 * n2 : n1)) + 3.0 * fabs(cc));
    const double E = 1.01 * B + 1e-12 * X;
    ok = ok && c.thr - E > 1e-30;
    f[6] = f[7] = -(float)cc;
    f[8] = ok ? round_down_f32(c.thr - E) : -INFINITY;  // |s32| below: certain inlier
    f[9] = ok ? round_up_f32(c.thr + E) : INFINITY;     // |s32| at or above: certain outlier
    if (!(sp[0] == sp[0])) f[8] = f[9] = __builtin_nanf("");  // NaN model: nothing agrees
  }
#if defined(__HIPCC__)
  // filter measure of two observations (xs[d] = coordinate d of both): |value| is compared with
  // f[4].x (certain inlier below) and f[4].y (certain outlier at or above)
  static __device__ inline v2f filter_value(const v2f *xs, const v2f *f) {
    v2f s = f[3];
    if (D == 3) s = __builtin_elementwise_fma(xs[2], f[2], s);
    s = __builtin_elementwise_fma(xs[1], f[1], s);
    s = __builtin_elementwise_fma(xs[0], f[0], s);
    return s;
  }
#endif

  // moments about `org`: {N, sum x', sum x'x'^T (upper)}  (PlaneParametersEstimator.hxx:141-154
  // accumulates the same sums un-shifted; shifting removes the cancellation at :160)
  static LSQR_HD void accumulate(const double *x, const double *org, double *m) {
    double d[D];
    for (int i = 0; i < D; i++) d[i] = x[i] - org[i];
    m[0] += 1.0;
    int q = 1 + D;
    for (int i = 0; i < D; i++) {
      m[1 + i] += d[i];
      for (int j = i; j < D; j++, q++) m[q] = fma(d[i], d[j], m[q]);
    }
  }

  // covariance eigenvector: smallest (plane, :163-171) or largest (line) eigenvalue
  static LSQR_HD bool solve_cov(const double *m, const double *org, bool largest, int min_n,
                                double *par) {
    double N = m[0];
    if (N < (double)min_n) return false;
    double mean[D], cov[D * D], w[D], v[D * D];
    for (int i = 0; i < D; i++) mean[i] = m[1 + i] / N;
    int q = 1 + D;
    for (int i = 0; i < D; i++)
      for (int j = i; j < D; j++) {
        double cij = m[q++] - N * mean[i] * mean[j];
        cov[i * D + j] = cov[j * D + i] = cij;
      }
    sym_eig(D, cov, w, v);
    int col = largest ? D - 1 : 0;
    for (int i = 0; i < D; i++) par[i] = v[i * D + col];
    for (int i = 0; i < D; i++) par[D + i] = mean[i] + org[i];
    return true;
  }
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &, double *par) {
    return solve_cov(m, org, false, D, par);  // :133 needs >= D points
  }
};

// ------------------------------------------------------------------------------------ line
template <int D>
struct LineModel {
  enum { ND = D, K = 2, P = 2 * D, SP = 2 * D, REC = D, PPL = 4, IS_DENSE = 0, IS_US = 0 };
  enum { NMOM = PlaneModel<D>::NMOM };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < D; i++) rec[i] = p[i];
  }

  // LineParametersEstimator.hxx:23-48
  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &c, double *par) {
    // Point::distanceSquared = squared_magnitude of the difference (common/Point.h:97-99)
    double dist = 0;
    for (int i = 0; i < D; i++) dist += (r[0][i] - r[1][i]) * (r[0][i] - r[1][i]);
    if (dist < c.delta_sq) return false;
    double dirNorm = 0.0;
    for (int i = 0; i < D; i++) {
      par[i] = r[0][i] - r[1][i];
      dirNorm += par[i] * par[i];
      par[D + i] = r[0][i];
    }
    dirNorm = sqrt(dirNorm);
    for (int i = 0; i < D; i++) par[i] /= dirNorm;
    return true;
  }

  // LineParametersEstimator.hxx:135-150
  static LSQR_HD double dist_sq(const double *sp, const double *x) {
    double v[D], vDotN = 0.0;
    for (int i = 0; i < D; i++) {
      v[i] = x[i] - sp[D + i];
      vDotN += v[i] * sp[i];
    }
    double d = 0.0;
    for (int i = 0; i < D; i++) d += (v[i] - vDotN * sp[i]) * (v[i] - vDotN * sp[i]);
    return d;
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return dist_sq(sp, x) < c.delta_sq;
  }
  static LSQR_HD void prepare(double *, const ModelConsts &) {}
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &) { return sqrt(dist_sq(sp, x)); }

  // ---- fp32 pre-filter ------------------------------------------------------------------------
  // filter_value() = |(x - a) x n|^2 in packed fp32 (the cross-product form has no cancellation of
  // the along-line component).  Bound, with u = 2^-24, u64 = 2^-53, X = max |coordinate|,
  // A = max |a_i|, W = X + A (bounds every |x_i - a_i|) and cap = 4 delta^2; for observations whose
  // exact squared distance D* = |(x-a) x n|^2 is <= cap:
  //   v32_i is within 2uW(1+u) of x_i - a_i and n32_i within u of n_i, so every cross-product
  //   component is within ec = 6uW + u sqrt(cap) of the exact one (3 sqrt2 uW from the operands, uW
  //   from the rounded product, u|c| from the fma), and the fp32 sum of squares within
  //   E32 = 2 sqrt3 sqrt(cap) ec + 3 ec^2 + 4 u cap of D*;
  //   the reference's fp64 expression |v - (v.n) n|^2 is within Eref = 2 sqrt3 sqrt(cap) ew + 3 ew^2 +
  //   4 u64 cap, ew = 12 u64 W, of its exact value, which differs from D* by at most
  //   Enn = 6 W^2 (| |n|^2 - 1 | + 4 u64)  (n is a unit vector only to rounding).
  // With E = 1.01 (E32 + Eref + Enn):  s32 < delta^2 - E  =>  agrees;  s32 >= delta^2 + E  =>  does
  // not (observations beyond cap evaluate above 3 delta^2 in fp32 as long as 6 sqrt3 uW <= delta/4,
  // and E <= delta^2/4 is required); in between the exact fp64 predicate decides.
  enum { NF = 6, SPF = 16, FGRAN = 1 };  // (-a_i,-a_i) x3, (n_i,n_i) x3, tin, tout, 0, 0
  static LSQR_HD void prepare_f32(const double *sp, const ModelConsts &c, float *f) {
    const double X = c.absmax, u = 5.9604644775390625e-08, u64 = 1.1102230246251565e-16;
    for (int i = 0; i < SPF; i++) f[i] = 0.0f;
    double A = 0.0, nn = 0.0;
    bool ok = X >= 1e-10 && X <= 1e15;
    for (int i = 0; i < D; i++) {
      f[2 * i] = f[2 * i + 1] = -(float)sp[D + i];
      f[6 + 2 * i] = f[7 + 2 * i] = (float)sp[i];
      A = fabs(sp[D + i]) > A ? fabs(sp[D + i]) : A;
      nn += sp[i] * sp[i];
      ok = ok && fabs(sp[i]) <= 1.0000001;
    }
    ok = ok && A <= 1e15 && fabs(nn - 1.0) <= 1e-6;
    const double W = X + A, cap = 4.0 * c.delta_sq, rc = 2.0 * c.delta, s3 = 1.7320508075688774;
    const double ec = 6.0 * u * W * (1.0 + 4.0 * u) + u * rc;
    const double E32 = 2.0 * s3 * rc * ec + 3.0 * ec * ec + 4.0 * u * cap;
    const double ew = 12.0 * u64 * W;
    const double Eref = 2.0 * s3 * rc * ew + 3.0 * ew * ew + 4.0 * u64 * cap;
    const double Enn = 6.0 * W * W * (fabs(nn - 1.0) + 4.0 * u64);
    const double E = 1.01 * (E32 + Eref + Enn);
    ok = ok && 6.0 * s3 * u * W <= 0.25 * c.delta && E <= 0.25 * c.delta_sq && c.delta_sq > 1e-30 &&
         c.delta_sq <= 1e30;
    f[12] = ok ? PlaneModel<3>::round_down_f32(c.delta_sq - E) : -INFINITY;
    f[13] = ok ? PlaneModel<3>::round_up_f32(c.delta_sq + E) : INFINITY;
    // reach of the model for the cell test of the two-level scan (cells.h: LineCell): an observation
    // whose fp32 distance from the line exceeds rho + (radius of its cell) cannot agree
    f[14] = ok ? PlaneModel<3>::round_up_f32(c.delta * (1.0 + 1e-6) + sqrt(Enn + Eref) +
                                             1.01 * 24.0 * u * W)
               : INFINITY;
    f[15] = ok ? PlaneModel<3>::round_up_f32(1.01 * (Eref + Enn)) : INFINITY;
    if (!(sp[0] == sp[0])) f[12] = f[13] = f[14] = __builtin_nanf("");  // NaN model: nothing agrees
  }
#if defined(__HIPCC__)
  static __device__ inline v2f filter_value(const v2f *xs, const v2f *f) {
    v2f v0 = xs[0] + f[0], v1 = xs[1] + f[1];
    if constexpr (D == 3) {
      v2f v2 = xs[2] + f[2];
      v2f c0 = __builtin_elementwise_fma(v1, f[5], -(v2 * f[4]));
      v2f c1 = __builtin_elementwise_fma(v2, f[3], -(v0 * f[5]));
      v2f c2 = __builtin_elementwise_fma(v0, f[4], -(v1 * f[3]));
      v2f s = c0 * c0;
      s = __builtin_elementwise_fma(c1, c1, s);
      return __builtin_elementwise_fma(c2, c2, s);
    } else {
      v2f c = __builtin_elementwise_fma(v0, f[4], -(v1 * f[3]));
      return c * c;
    }
  }
#endif

  static LSQR_HD void accumulate(const double *x, const double *org, double *m) {
    PlaneModel<D>::accumulate(x, org, m);
  }
  // LineParametersEstimator.hxx:68-111: largest eigenvector, needs >= 2 points
  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &, double *par) {
    return PlaneModel<D>::solve_cov(m, org, true, 2, par);
  }
};

// ------------------------------------------------------------------------------------ sphere
template <int D>
struct SphereModel {
  enum { ND = D, K = D + 1, P = D + 1, SP = D + 3, REC = D, PPL = 4, IS_DENSE = 0, IS_US = 0 };
  static LSQR_HD void load(const double *p, const ModelConsts &, double *rec) {
    for (int i = 0; i < D; i++) rec[i] = p[i];
  }
  // algebraic phase: {N, sum x', sum x'x'^T upper, sum x'|x'|^2, sum |x'|^2, sum |x'|^4}
  enum { NMOM = 1 + D + D * (D + 1) / 2 + D + 2 };
  enum { NLM = D + 1, NMOM_LM = 1 + (D + 1) * (D + 2) / 2 + (D + 1) };

  static LSQR_HD bool estimate(const double (*r)[ND], const ModelConsts &, double *par) {
    if constexpr (D == 2) {  // SphereParametersEstimator.hxx:80-109
      const double *p0 = r[0], *p1 = r[1], *p2 = r[2];
      double A00 = p0[0] - p1[0], A01 = p0[1] - p1[1];
      double A10 = p0[0] - p2[0], A11 = p0[1] - p2[1];
      double detA = (A00 * A11 - A01 * A10);
      if (fabs(detA) < kSphereEPS) return false;
      detA *= 2.0;
      double b0 = A00 * (p0[0] + p1[0]) + A01 * (p0[1] + p1[1]);
      double b1 = A10 * (p0[0] + p2[0]) + A11 * (p0[1] + p2[1]);
      par[0] = (A11 * b0 - A01 * b1) / detA;
      par[1] = (A00 * b1 - A10 * b0) / detA;
      par[2] = sqrt((p0[0] - par[0]) * (p0[0] - par[0]) + (p0[1] - par[1]) * (p0[1] - par[1]));
      return true;
    } else {  // SphereParametersEstimator.hxx:115-163
      const double *p0 = r[0], *p1 = r[1], *p2 = r[2], *p3 = r[3];
      double A00 = p0[0] - p1[0], A01 = p0[1] - p1[1], A02 = p0[2] - p1[2];
      double A10 = p0[0] - p2[0], A11 = p0[1] - p2[1], A12 = p0[2] - p2[2];
      double A20 = p0[0] - p3[0], A21 = p0[1] - p3[1], A22 = p0[2] - p3[2];
      double CT00 = A11 * A22 - A12 * A21;
      double CT10 = A12 * A20 - A10 * A22;
      double CT20 = A10 * A21 - A11 * A20;
      double detA = A00 * CT00 + A01 * CT10 + A02 * CT20;
      if (fabs(detA) < kSphereEPS) return false;
      detA *= 2;
      double CT01 = A02 * A21 - A01 * A22;
      double CT11 = A00 * A22 - A02 * A20;
      double CT21 = A01 * A20 - A00 * A21;
      double CT02 = A01 * A12 - A02 * A11;
      double CT12 = A02 * A10 - A00 * A12;
      double CT22 = A00 * A11 - A01 * A10;
      double b0 = A00 * (p0[0] + p1[0]) + A01 * (p0[1] + p1[1]) + A02 * (p0[2] + p1[2]);
      double b1 = A10 * (p0[0] + p2[0]) + A11 * (p0[1] + p2[1]) + A12 * (p0[2] + p2[2]);
      double b2 = A20 * (p0[0] + p3[0]) + A21 * (p0[1] + p3[1]) + A22 * (p0[2] + p3[2]);
      par[0] = (CT00 * b0 + CT01 * b1 + CT02 * b2) / detA;
      par[1] = (CT10 * b0 + CT11 * b1 + CT12 * b2) / detA;
      par[2] = (CT20 * b0 + CT21 * b1 + CT22 * b2) / detA;
      par[3] = sqrt(((p0[0] - par[0]) * (p0[0] - par[0])) + ((p0[1] - par[1]) * (p0[1] - par[1])) +
                    ((p0[2] - par[2]) * (p0[2] - par[2])));
      return true;
    }
  }

  // SphereParametersEstimator.hxx:255-264 (distance, not squared, against delta)
  static LSQR_HD double dist_sq(const double *sp, const double *x) {
    double s = ((x[0] - sp[0]) * (x[0] - sp[0]));
    for (int i = 1; i < D; i++) s += ((x[i] - sp[i]) * (x[i] - sp[i]));
    return s;
  }
  static LSQR_HD double residual(const double *sp, const double *x, const ModelConsts &) {
    return fabs(sqrt(dist_sq(sp, x)) - sp[D]);
  }
  static LSQR_HD bool agree_literal(const double *sp, const double *x, const ModelConsts &c) {
    return residual(sp, x, c) < c.delta;
  }
  // |fl(fl(sqrt(d2)) - r)| < delta holds exactly for d2 in a closed interval [sp[D+1], sp[D+2]] of
  // doubles (sqrt and s - r are monotone and correctly rounded): prepare() finds its end points
  // with the same sqrt and subtraction, agree() then needs no sqrt.  sp[D+1] < 0 marks "interval
  // search did not settle" and selects the literal formula.
  static LSQR_HD void prepare(double *sp, const ModelConsts &c) {
    const double r = sp[D], delta = c.delta;
    sp[D + 1] = -1.0;
    sp[D + 2] = 0.0;
    if (!(r == r)) {  // NaN model: never agrees
      sp[D + 1] = r;
      sp[D + 2] = r;
      return;
    }
    if (!(delta > 0.0) || !(r >= 0.0) || !(r < 1e300)) return;  // literal formula
    PredSubLt below = {r, delta};
    PredSubGt above = {r, -delta};
    const double shi = last_true(below);   // largest s with fl(s-r) <  delta
    const double slo = first_true(above);  // smallest s with fl(s-r) > -delta
    if (shi < 0.0 || !(slo <= shi)) {      // empty set: a NaN interval never agrees
      sp[D + 1] = sp[D + 2] = __builtin_nan("");
      return;
    }
    PredSqrtLe le = {shi};
    PredSqrtGe ge = {slo};
    const double dhi = last_true(le);   // largest d2 with fl(sqrt(d2)) <= shi
    const double dlo = first_true(ge);  // smallest d2 with fl(sqrt(d2)) >= slo
    if (dhi < 0.0 || !(dlo <= dhi)) {
      sp[D + 1] = sp[D + 2] = __builtin_nan("");
      return;
    }
    sp[D + 1] = dlo;
    sp[D + 2] = dhi;
  }
  // ---- fp32 pre-filter (k_scan_f32) --------------------------------------------------------------
  // d32 = sum (x32_i - c32_i)^2 (fma chain), value t = d32 - mid with [Dlo, Dhi] = mid -+ half the exact
  // squared-radius interval of prepare().  u = 2^-24, X = max |coordinate| of the observations,
  // C = max |c_i|, D* = sum (x_i - c_i)^2:
  //   |d32_i - (x_i - c_i)| <= e := 2u(X + C)(1 + u)        (two input roundings + the subtraction)
  //   |d32 - D*| <= 2 sqrt(3) e sqrt(D*) + 3 e^2 + 3u D* =: E(D*)   (squares + three roundings)
  // E is increasing; for every observation with D* <= Dcap := 2 Dhi + 1 the error is <= E(Dcap), and
  // (when E(Dcap) <= Dhi/4, checked) an observation with D* > Dcap has d32 > Dhi + E(Dcap), i.e. lies
  // outside the candidate band anyway.  Ecap = 1.01 E(Dcap) + 3u Dcap (the subtraction of mid and
  // its rounding) + 1e-12 Dcap (fp64 vs exact):
  //   |t| <  half - Ecap  =>  D_ref in [Dlo, Dhi]  (certainly agrees)
  //   |t| >= half + Ecap  =>  certainly does not;   in between: exact fp64 predicate.
  // f[12..19]: centre (3 doubles) and mid (1 double) as pairs of 32-bit words, f[20] = half, f[21] = 1 when the
  // filter is usable: the cell-relative evaluation of the two-level scan (cells.h: SphereCell)
  enum { NF = 4, SPF = 24, FGRAN = 0 };
  static LSQR_HD void prepare_f32(const double *sp, const ModelConsts &c, float *f) {
    const double X = c.absmax, u = 5.9604644775390625e-08;
    for (int i = 0; i < 24; i++) f[i] = 0.0f;
    double C = 0.0;
    for (int i = 0; i < D; i++) {
      f[2 * i] = f[2 * i + 1] = -(float)sp[i];  // x + (-c)
      C = fabs(sp[i]) > C ? fabs(sp[i]) : C;
    }
    const double dlo = sp[D + 1], dhi = sp[D + 2];
    bool ok = X >= 1e-10 && X <= 1e15 && C <= 1e15 && dlo >= 0.0 && dhi >= dlo && dhi <= 1e30;
    const double dcap = 2.0 * dhi + 1.0;
    const double e = 2.0 * u * (X + C) * (1.0 + u);
    const double E = 2.0 * 1.7320508075688774 * e * sqrt(dcap) + 3.0 * e * e + 3.0 * u * dcap;
    const double Ecap = 1.01 * E + 3.0 * u * dcap + 1e-12 * dcap;
    ok = ok && Ecap <= 0.25 * dhi;
    const double mid = 0.5 * (dlo + dhi), half = 0.5 * (dhi - dlo);
    f[6] = f[7] = -(float)mid;
    const double slack = Ecap + 2.0 * u * mid;  // mid itself is rounded to fp32
    f[8] = (ok && half - slack > 0.0) ? PlaneModel<3>::round_down_f32(half - slack) : -INFINITY;
    f[9] = ok ? PlaneModel<3>::round_up_f32(half + slack) : INFINITY;
    // squared-radius interval rounded outwards for the cell test of the two-level scan (cells.h)
    f[10] = ok ? PlaneModel<3>::round_down_f32(dlo * (1.0 - 1e-9)) : 0.0f;
    f[11] = ok ? PlaneModel<3>::round_up_f32(dhi * (1.0 + 1e-9)) : INFINITY;
    if (!(sp[0] == sp[0]) || (dlo != dlo)) f[8] = f[9] = f[10] = f[11] = __builtin_nanf("");  // never agrees
    {
      double w[4] = {D > 0 ? sp[0] : 0.0, D > 1 ? sp[1] : 0.0, D > 2 ? sp[D > 2 ? 2 : 0] : 0.0, mid};
      for (int i = 0; i < 4; i++) {
        unsigned long long bits;
        __builtin_memcpy(&bits, &w[i], 8);
        const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
        __builtin_memcpy(&f[12 + 2 * i], &lo, 4);
        __builtin_memcpy(&f[13 + 2 * i], &hi, 4);
      }
      f[20] = (float)half;  // rounding is covered by the cell model's slack
      f[21] = ok ? 1.0f : 0.0f;
      if (!(sp[0] == sp[0]) || (dlo != dlo)) f[21] = __builtin_nanf("");
    }
  }
#if defined(__HIPCC__)
  static __device__ inline v2f filter_value(const v2f *xs, const v2f *f) {
    v2f d0 = xs[0] + f[0], d1 = xs[1] + f[1];
    v2f s = d0 * d0;
    s = __builtin_elementwise_fma(d1, d1, s);
    if (D == 3) {
      v2f d2 = xs[2] + f[2];
      s = __builtin_elementwise_fma(d2, d2, s);
    }
    return s + f[3];
  }
#endif

  static LSQR_HD bool use_literal(const double *sp) { return sp[D + 1] < 0.0; }  // per hypothesis
  static LSQR_HD bool agree_interval(const double *sp, const double *x) {
    double d2 = dist_sq(sp, x);
    return (d2 >= sp[D + 1]) & (d2 <= sp[D + 2]);  // no short circuit: both are one v_cmp
  }
  static LSQR_HD bool agree(const double *sp, const double *x, const ModelConsts &c) {
    return use_literal(sp) ? agree_literal(sp, x, c) : agree_interval(sp, x);
  }

  // algebraic fit (SphereParametersEstimator.hxx:267-307): rows [-2x, 1], rhs -|x|^2, as
  // (D+1)x(D+1) normal equations in coordinates shifted by `org` (the fit is translation
  // equivariant; shifting keeps the normal equations well conditioned).
  static LSQR_HD void accumulate(const double *x, const double *org, double *m) {
    double d[D], q = 0;
    for (int i = 0; i < D; i++) {
      d[i] = x[i] - org[i];
      q = fma(d[i], d[i], q);
    }
    m[0] += 1.0;
    int k = 1 + D;
    for (int i = 0; i < D; i++) {
      m[1 + i] += d[i];
      for (int j = i; j < D; j++, k++) m[k] = fma(d[i], d[j], m[k]);
    }
    for (int i = 0; i < D; i++, k++) m[k] = fma(d[i], q, m[k]);
    m[k] += q;
    m[k + 1] = fma(q, q, m[k + 1]);
  }

  static LSQR_HD bool solve(const double *m, const double *org, const ModelConsts &, double *par) {
    const int n = D + 1;
    if (m[0] < (double)n) return false;  // :271
    // unknown y = [c', rho] with rho = |c'|^2 - r^2:  A = [-2x', 1], b = -|x'|^2
    double G[n * n], rhs[n], x[n], work[2 * n * n + 3 * n];
    int k = 1 + D;
    for (int i = 0; i < D; i++)
      for (int j = i; j < D; j++) {
        double v = 4.0 * m[k++];
        G[i * n + j] = G[j * n + i] = v;
      }
    for (int i = 0; i < D; i++) {
      G[i * n + D] = G[D * n + i] = -2.0 * m[1 + i];
      rhs[i] = 2.0 * m[k++];
    }
    G[D * n + D] = m[0];
    rhs[D] = -m[k];
    int rank = spd_solve_eig(n, G, rhs, 1e-14, x, work);
    if (rank < n) return false;  // :295-296
    double r2 = -x[D];
    for (int i = 0; i < D; i++) r2 += x[i] * x[i];
    if (!(r2 > 0)) return false;  // :303-306
    for (int i = 0; i < D; i++) par[i] = x[i] + org[i];
    par[D] = sqrt(r2);
    return true;
  }

  static LSQR_HD int lm_finalize(const double *x, double *par) {
    for (int i = 0; i <= D; i++) par[i] = x[i];
    return D + 1;
  }

  // one row of (J | f) at xk = [c, r] (f: SphereParametersEstimator.hxx:394-409, gradf: :413-431) for the
  // matrix-core pass k_lm_pass_mfma: z[0..D) = (c - x) / |x - c|, z[D] = -1, z[D + 1] = |x - c| - r
  struct LmCoef {
    double x[D + 1];
  };
  static LSQR_HD void lm_coef(const double *xk, LmCoef &k) {
    for (int i = 0; i <= D; i++) k.x[i] = xk[i];
  }
  static LSQR_HD void lm_row(const double *x, const LmCoef &k, double *z) {
    double sq = 0.0;
    for (int j = 0; j < D; j++) sq += (x[j] - k.x[j]) * (x[j] - k.x[j]);
    const double s = sqrt(sq);
    for (int j = 0; j < D; j++) z[j] = (k.x[j] - x[j]) / s;
    z[D] = -1.0;
    z[D + 1] = s - k.x[D];
  }
  // geometric fit pass (f: SphereParametersEstimator.hxx:394-409, gradf: :413-431):
  // {sum f^2, J^T J upper, J^T f} at xk = [c, r]
  static LSQR_HD void accumulate_lm(const double *x, const double *xk, double *m) {
    double J[D + 1], sq = 0.0;
    for (int j = 0; j < D; j++) sq += (x[j] - xk[j]) * (x[j] - xk[j]);
    double s = sqrt(sq);
    double f = s - xk[D];
    for (int j = 0; j < D; j++) J[j] = (xk[j] - x[j]) / s;
    J[D] = -1;
    m[0] = fma(f, f, m[0]);
    int k = 1;
    for (int i = 0; i <= D; i++)
      for (int j = i; j <= D; j++, k++) m[k] = fma(J[i], J[j], m[k]);
    for (int i = 0; i <= D; i++, k++) m[k] = fma(J[i], f, m[k]);
  }
};

}  // namespace lsqr
