// estimatorTests -- the assertions of the reference's ctest programs for the hot-path estimators
// (testing/PlaneParametersEstimatorTest.cxx, SphereParametersEstimatorTest.cxx,
// LineParametersEstimatorTest.cxx, DenseLinearEquationSystemParametersEstimatorTest.cxx,
// SinglePointTargetUSCalibrationParametersEstimatorTest.cxx, and -- SURVEY.md section 8f --
// AbsoluteOrientationParametersEstimatorTest.cxx, PivotCalibrationParametersEstimatorTest.cxx,
// PlanePhantomUSCalibrationParametersEstimatorTest.cxx)
// re-written against the drop-in
// headers, plus RANSAC-level checks (the reference has none).  Runs on the GPU through the C++
// API exactly as a user of the reference would call it.  Exit code 0 == all passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "AbsoluteOrientationParametersEstimator.h"
#include "DenseLinearEquationSystemParametersEstimator.h"
#include "Frame.h"
#include "Line2DParametersEstimator.h"
#include "LineParametersEstimator.h"
#include "PivotCalibrationParametersEstimator.h"
#include "PlanePhantomUSCalibrationParametersEstimator.h"
#include "PlaneParametersEstimator.h"
#include "RayIntersectionParametersEstimator.h"
#include "RANSAC.h"
#include "SinglePointTargetUSCalibrationParametersEstimator.h"
#include "SphereParametersEstimator.h"

using namespace lsqrRecipes;

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) {                                                         \
      std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);        \
      failures++;                                                          \
    }                                                                      \
  } while (0)

static std::mt19937_64 gen(12345);
static double U(double a, double b) { return std::uniform_real_distribution<double>(a, b)(gen); }
static double N(double s) { return std::normal_distribution<double>(0.0, s)(gen); }
static const double COS5 = 0.99619469809174553229501040247389;

static void planeTest() {  // testing/PlaneParametersEstimatorTest.cxx:71-158
  typedef Point<double, 3> P;
  const double delta = 0.5;
  PlaneParametersEstimator<3> est(delta);
  std::vector<P> minimal(3), noisy;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) minimal[i][j] = U(-1000, 1000);
  double v1[3], v2[3], n[3];
  for (int j = 0; j < 3; j++) {
    v1[j] = minimal[1][j] - minimal[0][j];
    v2[j] = minimal[2][j] - minimal[0][j];
  }
  n[0] = v1[1] * v2[2] - v1[2] * v2[1];
  n[1] = v1[2] * v2[0] - v1[0] * v2[2];
  n[2] = v1[0] * v2[1] - v1[1] * v2[0];
  double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  for (int j = 0; j < 3; j++) n[j] /= nn;
  std::vector<double> truth(n, n + 3), params;
  for (int j = 0; j < 3; j++) truth.push_back(minimal[0][j]);
  P on = minimal[1], off = minimal[1];
  for (int j = 0; j < 3; j++) off[j] += 2 * delta * n[j];
  CHECK(est.agree(truth, on));
  CHECK(!est.agree(truth, off));
  est.estimate(minimal, params);
  CHECK(params.size() == 6);
  if (params.size() == 6) {
    double dot = params[0] * n[0] + params[1] * n[1] + params[2] * n[2], d = 0;
    for (int j = 0; j < 3; j++) d += (params[3 + j] - minimal[0][j]) * n[j];
    CHECK(std::fabs(dot) > COS5);
    CHECK(std::fabs(d) < delta);
  }
  for (int i = 0; i < 20; i++) {  // barycentric combinations + noise
    double w[3] = {U(0, 1), U(0, 1), U(0, 1)}, s = w[0] + w[1] + w[2];
    P p;
    for (int j = 0; j < 3; j++)
      p[j] = (w[0] * minimal[0][j] + w[1] * minimal[1][j] + w[2] * minimal[2][j]) / s + N(0.1);
    noisy.push_back(p);
  }
  est.leastSquaresEstimate(noisy, params);
  CHECK(params.size() == 6);
  if (params.size() == 6) {
    double dot = params[0] * n[0] + params[1] * n[1] + params[2] * n[2], d = 0;
    for (int j = 0; j < 3; j++) d += (params[3 + j] - minimal[0][j]) * n[j];
    CHECK(std::fabs(dot) > COS5);
    CHECK(std::fabs(d) < delta);
  }
  std::vector<P> collinear(3);  // degenerate minimal set -> empty
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) collinear[i][j] = i;
  est.estimate(collinear, params);
  CHECK(params.empty());
  std::vector<P *> ptrs;  // pointer overloads
  for (size_t i = 0; i < noisy.size(); i++) ptrs.push_back(&noisy[i]);
  est.leastSquaresEstimate(ptrs, params);
  CHECK(params.size() == 6);
}

static void sphereTest() {  // testing/SphereParametersEstimatorTest.cxx:280-296,504-506
  typedef Point<double, 2> P2;
  SphereParametersEstimator<2> circ(0.5);
  std::vector<double> truth = {0.0, 0.0, 2.0}, params;
  P2 p;
  p[1] = 0;
  p[0] = 2.0; CHECK(circ.agree(truth, p));
  p[0] = 2.4; CHECK(circ.agree(truth, p));
  p[0] = 2.6; CHECK(!circ.agree(truth, p));
  p[0] = 1.4; CHECK(!circ.agree(truth, p));
  // Gander, Golub, Strebel circle (test :313-324): geometric fit (4.7398, 2.9835, 4.7142)
  double g[6][2] = {{1, 7}, {2, 6}, {5, 8}, {7, 7}, {9, 5}, {3, 7}};
  std::vector<P2> pts;
  for (auto &r : g) {
    P2 q;
    q[0] = r[0];
    q[1] = r[1];
    pts.push_back(q);
  }
  circ.setLeastSquaresType(SphereParametersEstimator<2>::GEOMETRIC);
  circ.leastSquaresEstimate(pts, params);
  CHECK(params.size() == 3);
  if (params.size() == 3) {
    CHECK(std::fabs(params[0] - 4.7398) < 1e-4);
    CHECK(std::fabs(params[1] - 2.9835) < 1e-4);
    CHECK(std::fabs(params[2] - 4.7142) < 1e-4);
  }
  typedef Point<double, 3> P3;
  SphereParametersEstimator<3> sph(0.5);
  double c[3] = {U(-50, 50), U(-50, 50), U(-50, 50)}, r = U(10, 50), sigma = 1.0;
  std::vector<P3> data;
  for (int i = 0; i < 20; i++) {
    double u[3] = {U(-1, 1), U(-1, 1), U(-1, 1)}, nu = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    P3 q;
    for (int j = 0; j < 3; j++) q[j] = c[j] + r * u[j] / nu + N(sigma) / 3;
    data.push_back(q);
  }
  std::vector<P3 *> ptrs;
  for (auto &q : data) ptrs.push_back(&q);
  std::vector<double> alg, geo;
  sph.algebraicLeastSquaresEstimate(ptrs, alg);
  CHECK(alg.size() == 4);
  sph.geometricLeastSquaresEstimate(ptrs, alg, geo);
  CHECK(geo.size() == 4);
  for (auto *est : {&alg, &geo})
    if (est->size() == 4) {
      double dc = 0;
      for (int j = 0; j < 3; j++) dc += ((*est)[j] - c[j]) * ((*est)[j] - c[j]);
      CHECK(std::sqrt(dc) <= 3 * sigma);
      CHECK(std::fabs((*est)[3] - r) <= 3 * sigma);
    }
  bool threw = false;
  try {
    SphereParametersEstimator<3> bad(0.5, (SphereParametersEstimator<3>::LeastSquaresType)7);
  } catch (std::exception &) {
    threw = true;
  }
  CHECK(threw);  // SphereParametersEstimator.hxx:17-18
}

static void lineTest() {  // testing/LineParametersEstimatorTest.cxx:98-213
  typedef Point<double, 2> P;
  LineParametersEstimator<2> est(0.5);
  std::vector<P> two(2), noisy;
  two[0][0] = 10; two[0][1] = -4; two[1][0] = -70; two[1][1] = 55;
  std::vector<double> params;
  est.estimate(two, params);
  CHECK(params.size() == 4);
  double d[2] = {two[0][0] - two[1][0], two[0][1] - two[1][1]}, nd = std::sqrt(d[0] * d[0] + d[1] * d[1]);
  d[0] /= nd; d[1] /= nd;
  if (params.size() == 4) CHECK(std::fabs(params[0] * d[0] + params[1] * d[1]) > COS5);
  P off = two[1];
  off[0] += -d[1];
  off[1] += d[0];
  CHECK(est.agree(params, two[1]));
  CHECK(!est.agree(params, off));
  for (int i = 0; i < 20; i++) {
    double t = U(-100, 100);
    P p;
    p[0] = two[0][0] + t * d[0] + N(0.2);
    p[1] = two[0][1] + t * d[1] + N(0.2);
    noisy.push_back(p);
  }
  est.leastSquaresEstimate(noisy, params);
  CHECK(params.size() == 4 && std::fabs(params[0] * d[0] + params[1] * d[1]) > COS5);
  // the same assertions for Line2DParametersEstimator (normal form): testing/Line...Test.cxx:121-213
  Line2DParametersEstimator est2(0.5);
  const double nrm[2] = {-d[1], d[0]};
  est2.estimate(two, params);
  CHECK(params.size() == 4);
  if (params.size() == 4) {
    CHECK(std::fabs(params[0] * nrm[0] + params[1] * nrm[1]) > COS5);
    CHECK(est2.agree(params, two[1]));
    CHECK(!est2.agree(params, off));
  }
  est2.leastSquaresEstimate(noisy, params);
  CHECK(params.size() == 4 && std::fabs(params[0] * nrm[0] + params[1] * nrm[1]) > COS5);
}

static void denseTest(const char *matrixFile) {  // testing/DenseLinear...Test.cxx:72-209
  const unsigned int n = 5;
  typedef AugmentedRow<double, n> Row;
  DenseLinearEquationSystemParametersEstimator<double, n> est(0.1);
  std::vector<double> x(n), params;
  for (auto &v : x) v = U(-1, 1);
  std::vector<Row> exact, noisy;
  for (unsigned i = 0; i < 200; i++) {
    double a[n + 1], b = 0;
    for (unsigned j = 0; j < n; j++) {
      a[j] = U(-1, 1);
      b += a[j] * x[j];
    }
    a[n] = b;
    if (i < n) exact.push_back(Row(a));
    a[n] = b * (1 + U(-0.05, 0.05));
    noisy.push_back(Row(a));
  }
  est.estimate(exact, params);
  CHECK(params.size() == n);
  for (unsigned j = 0; j < params.size(); j++) CHECK(std::fabs(params[j] - x[j]) < 1e-10);
  est.leastSquaresEstimate(noisy, params);
  CHECK(params.size() == n);
  for (unsigned j = 0; j < params.size(); j++) CHECK(std::fabs(params[j] - x[j]) < 0.1);
  CHECK(est.agree(x, exact[0]));
  if (matrixFile) {  // known-answer fixture, test :162-164, tolerance :183
    const double known[6] = {-1.777985584409468e+001, 1.111302171667757e+000, -1.568653413096010e+002,
                             1.469013927556186e+002,  -6.296891425314718e+001, -1.042139650090033e+003};
    typedef AugmentedRow<double, 6> Row6;
    std::ifstream in(matrixFile);
    std::vector<Row6> rows;
    double v[7];
    while (in >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5] >> v[6]) rows.push_back(Row6(v));
    CHECK(rows.size() == 1443);
    DenseLinearEquationSystemParametersEstimator<double, 6> est6(0.5);
    est6.leastSquaresEstimate(rows, params);
    CHECK(params.size() == 6);
    for (unsigned j = 0; j < params.size(); j++) CHECK(std::fabs(params[j] - known[j]) < 0.5);
    for (unsigned j = 0; j < params.size(); j++) CHECK(std::fabs(params[j] - known[j]) < 1e-6 * 1042);
  }
}

static void usTest() {  // testing/SinglePointTargetUSCalibration...Test.cxx:179-220,466-552
  typedef SingleUnknownPointTargetUSCalibrationParametersEstimator Est;
  typedef Est::DataType D;
  const double PI = 3.14159265358979323846, mx = 0.143, my = 0.139;
  Frame T3;
  double w3[3] = {U(0, PI), U(0, PI), U(0, PI)}, t3[3] = {U(-100, 100), U(-100, 100), U(-100, 100)};
  T3.setRotationEulerAngles(w3[2], w3[1], w3[0]);
  T3.setTranslation(t3);
  double t1[3] = {U(-100, 100), U(-100, 100), U(-100, 100)};
  std::vector<D> clean, noisy;
  for (int i = 0; i < 50; i++) {
    D d;
    double u = U(0, 640), v = U(0, 480), q[3] = {mx * u, my * v, 0}, q3[3], rq[3];
    d.T2.setRotationEulerAngles(U(0, PI), U(0, PI), U(0, PI));
    T3.apply(q, q3);
    d.T2.apply(q3, rq);
    d.T2.setTranslation(t1[0] - rq[0], t1[1] - rq[1], t1[2] - rq[2]);
    d.q[0] = u;
    d.q[1] = v;
    clean.push_back(d);
    d.q[0] += N(1.0);
    d.q[1] += N(1.0);
    noisy.push_back(d);
  }
  Est est(3.0);
  std::vector<D> minimal(clean.begin(), clean.begin() + 4);
  std::vector<double> params;
  est.estimate(minimal, params);
  CHECK(params.size() == 20);
  if (params.size() == 20) CHECK(est.agree(params, clean[0]));
  std::vector<D> five(clean.begin(), clean.begin() + 5);
  est.estimate(five, params);
  CHECK(params.empty());  // exactly four elements required
  est.setLeastSquaresType(Est::ANALYTIC);
  est.leastSquaresEstimate(noisy, params);
  CHECK(params.size() == 20);
  if (params.size() == 20) {
    double dt = 0;
    for (int j = 0; j < 3; j++) dt += (params[3 + j] - t3[j]) * (params[3 + j] - t3[j]);
    CHECK(std::sqrt(dt) < 1.0);
    CHECK(std::fabs(params[9] - mx) < 1.0 && std::fabs(params[10] - my) < 1.0);
  }
  est.setLeastSquaresType(Est::ITERATIVE);
  est.leastSquaresEstimate(noisy, params);  // "iterative LS on noisy data must pass" (:213-220)
  CHECK(params.size() == 20);
  double mn, mxd, mean;
  if (params.size() == 20) {
    Est::getDistanceStatistics(params, noisy, mn, mxd, mean);
    CHECK(mean < 3.0);
  }
}

// testing/PlanePhantomUSCalibrationParametersEstimatorTest.cxx:277-379: only T3 is compared (3 mm, 5
// degrees on one of the two Euler solutions, scale 1.0)
static bool phantomClose(const std::vector<double> &e, const double t3[3], const double w3[3],
                         double mx, double my) {
  if (e.size() != 41) return false;
  const double ANG = 0.08726646259971647884618453842445, small = 0.008726535498373935,
               halfPI = 1.5707963267948966192313216916398;
  double r1[3], r2[3], R[3][3];
  for (int j = 0; j < 3; j++) {
    r1[j] = e[11 + j] / (e[9] * e[38]);
    r2[j] = e[20 + j] / (e[10] * e[38]);
  }
  double r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
  for (int j = 0; j < 3; j++) R[j][0] = r1[j], R[j][1] = r2[j], R[j][2] = r3[j];
  const double h = std::sqrt(R[0][0] * R[0][0] + R[1][0] * R[1][0]);
  const double y1 = std::atan2(-R[2][0], h), y2 = std::atan2(-R[2][0], -h);
  double z1, z2, x1, x2;
  if (std::fabs(y1 - halfPI) > small && std::fabs(y1 + halfPI) > small) {
    double c1 = std::cos(y1), c2 = std::cos(y2);
    z1 = std::atan2(R[1][0] / c1, R[0][0] / c1), x1 = std::atan2(R[2][1] / c1, R[2][2] / c1);
    z2 = std::atan2(R[1][0] / c2, R[0][0] / c2), x2 = std::atan2(R[2][1] / c2, R[2][2] / c2);
  } else {
    z1 = z2 = 0;
    x1 = x2 = std::atan2(R[0][1], R[1][1]);
  }
  // w3 = {x, y, z}
  // angles compared modulo 2 pi (the reference compares raw differences, which fails spuriously when
  // a true angle sits at the +-pi seam)
  auto ad = [](double a, double b) { return std::fabs(std::remainder(a - b, 2 * 3.14159265358979323846)); };
  bool ang = (ad(z1, w3[2]) < ANG && ad(y1, w3[1]) < ANG && ad(x1, w3[0]) < ANG) ||
             (ad(z2, w3[2]) < ANG && ad(y2, w3[1]) < ANG && ad(x2, w3[0]) < ANG);
  bool tr = std::fabs(e[3] - t3[0]) < 3.0 && std::fabs(e[4] - t3[1]) < 3.0 && std::fabs(e[5] - t3[2]) < 3.0;
  const bool ok = ang && tr && std::fabs(e[9] - mx) < 1.0 && std::fabs(e[10] - my) < 1.0;
  if (!ok) {
    std::printf("  phantom estimate t3 %.6g %.6g %.6g  w3(z,y,x) %.6g %.6g %.6g | %.6g %.6g %.6g  m %.6g %.6g\n", e[3],
                e[4], e[5], z1, y1, x1, z2, y2, x2, e[9], e[10]);
    std::printf("  expected         t3 %.6g %.6g %.6g  w3(z,y,x) %.6g %.6g %.6g  m %.6g %.6g\n", t3[0], t3[1],
                t3[2], w3[2], w3[1], w3[0], mx, my);
  }
  return ok;
}

static void phantomTest() {  // testing/PlanePhantomUSCalibration...Test.cxx:131-187, data :382-548
  typedef PlanePhantomUSCalibrationParametersEstimator Est;
  typedef Est::DataType D;
  const double PI = 3.14159265358979323846, mx = 0.143, my = 0.139;
  Frame T3, T1;
  double w3[3] = {U(0, PI), U(0, PI), U(0, PI)}, t3[3] = {U(-100, 100), U(-100, 100), U(-100, 100)};
  T3.setRotationEulerAngles(w3[0], w3[1], w3[2]);
  T3.setTranslation(t3);
  double t1[3] = {U(-100, 100), U(-100, 100), U(-100, 100)}, R1[3][3];
  T1.setRotationEulerAngles(U(0, PI), U(0, PI), U(0, PI));
  T1.getRotationMatrix(R1);
  std::vector<D> clean, noisy;
  for (int i = 0; i < 50; i++) {
    D d;
    double u = U(0, 640), v = U(0, 480), q[3] = {mx * u, my * v, 0}, q3[3], rq[3];
    double onPlane[3] = {U(-100, 100), U(-100, 100), 0.0}, inTracker[3];
    for (int r = 0; r < 3; r++) {
      inTracker[r] = 0;
      for (int c = 0; c < 3; c++) inTracker[r] += R1[c][r] * (onPlane[c] - t1[c]);
    }
    d.T2.setRotationEulerAngles(U(0, PI), U(0, PI), U(0, PI));
    T3.apply(q, q3);
    d.T2.apply(q3, rq);
    d.T2.setTranslation(inTracker[0] - rq[0], inTracker[1] - rq[1], inTracker[2] - rq[2]);
    d.q[0] = u;
    d.q[1] = v;
    clean.push_back(d);
    d.q[0] += N(1.0);
    d.q[1] += N(1.0);
    noisy.push_back(d);
  }
  Est est(3.0);
  std::vector<D> minimal(clean.begin(), clean.begin() + 31);
  std::vector<double> params;
  est.estimate(minimal, params);
  CHECK(phantomClose(params, t3, w3, mx, my));
  if (params.size() == 41) CHECK(est.agree(params, clean[0]));
  std::vector<D> tooMany(clean.begin(), clean.begin() + 32);
  est.estimate(tooMany, params);
  CHECK(params.empty());  // exactly 31 elements required (.cxx:19)
  est.setLeastSquaresType(Est::ANALYTIC);
  est.leastSquaresEstimate(noisy, params);
  CHECK(phantomClose(params, t3, w3, mx, my));
  std::vector<double> analytic(params), distances;
  est.setLeastSquaresType(Est::ITERATIVE);
  est.leastSquaresEstimate(noisy, params);
  CHECK(phantomClose(params, t3, w3, mx, my));
  double mn, mxd, mean, sseIt = 0, sseAn = 0;
  if (params.size() == 41 && analytic.size() == 41) {
    Est::getDistanceStatistics(params, noisy, distances, mn, mxd, mean);
    CHECK(distances.size() == noisy.size());
    double s = 0;
    for (size_t i = 0; i < distances.size(); i++) {
      sseIt += distances[i] * distances[i];
      s += distances[i];
      CHECK(distances[i] >= mn && distances[i] <= mxd);
    }
    CHECK(std::fabs(s / distances.size() - mean) < 1e-9);
    Est::getDistanceStatistics(analytic, noisy, distances, mn, mxd, mean);
    for (size_t i = 0; i < distances.size(); i++) sseAn += distances[i] * distances[i];
    CHECK(sseIt <= sseAn + 1e-9);
    // refinement from a caller-supplied start reaches the same minimum
    std::vector<D *> ptrs;
    for (size_t i = 0; i < noisy.size(); i++) ptrs.push_back(&noisy[i]);
    std::vector<double> refined;
    est.iterativeLeastSquaresEstimate(ptrs, analytic, refined);
    CHECK(refined.size() == 41);
    if (refined.size() == 41)
      for (int j = 3; j < 11; j++) CHECK(std::fabs(refined[j] - params[j]) < 1e-6 * (1 + std::fabs(params[j])));
  }
}

static void ransacTest() {
  typedef Point<double, 3> P;
  std::vector<P> data;
  double n[3] = {0.6, 0.0, 0.8}, a[3] = {10, 20, 30};
  for (int i = 0; i < 2000; i++) {
    P p;
    double d = 0;
    for (int j = 0; j < 3; j++) {
      p[j] = U(-1000, 1000);
      d += (p[j] - a[j]) * n[j];
    }
    if (i % 2 == 0)
      for (int j = 0; j < 3; j++) p[j] += -d * n[j] + N(0.2);
    data.push_back(p);
  }
  PlaneParametersEstimator<3> est(0.5);
  std::vector<double> params(3, 42.0);
  std::vector<bool> consensus;
  // invalid input: returns 0 and leaves parameters untouched (RANSAC.hxx:16-19)
  CHECK((RANSAC<P, double>::compute(params, &est, data, 1.0) == 0));
  CHECK(params.size() == 3 && params[0] == 42.0);
  std::vector<P> tooFew(data.begin(), data.begin() + 2);
  CHECK((RANSAC<P, double>::compute(params, &est, tooFew, 0.9) == 0));
  CHECK(params.size() == 3);
  CHECK((RANSAC<P, double>::compute(params, &est, tooFew) == 0));  // exhaustive: cleared
  CHECK(params.empty());
  double frac = RANSAC<P, double>::compute(params, &est, data, 0.999, &consensus);
  CHECK(params.size() == 6 && consensus.size() == data.size());
  CHECK(frac > 0.35 && frac < 0.55);
  if (params.size() == 6) CHECK(std::fabs(std::fabs(params[0] * n[0] + params[1] * n[1] + params[2] * n[2]) - 1) < 1e-6);
  size_t cnt = 0;
  for (size_t i = 0; i < consensus.size(); i++) cnt += consensus[i];
  CHECK(std::fabs((double)cnt / data.size() - frac) < 1e-12);
  // reproducible for a fixed seed, different stream for another
  std::vector<double> again;
  double frac2 = RANSAC<P, double>::compute(again, &est, data, 0.999);
  CHECK(frac2 == frac && again == params);
  // exhaustive overload on a small set
  std::vector<P> small(data.begin(), data.begin() + 16);
  double fe = RANSAC<P, double>::compute(params, &est, small, &consensus);
  CHECK(fe >= 3.0 / 16 && params.size() == 6);
  CHECK((RANSAC<P, double>::lastInfo().iterations == 560));  // C(16,3)
}

// ---- plugin path: a user-defined estimator (the reference's advertised use, readme.txt:40-72) --------------
// The readme's example: a 2-D line [n, a] estimator written by the user against ParametersEstimator<T,S>.
// It has no device model, so RANSAC<T,S>::compute() drives its virtuals with the serial loop; because that
// loop shares the subset stream and the stopping rule with the device path, it must reproduce what the
// built-in Line2DParametersEstimator computes on the device for the same seed.
struct UserPoint2D {
  double x, y;
};
class UserLine2D : public ParametersEstimator<UserPoint2D, double> {
 public:
  UserLine2D(double delta) : ParametersEstimator<UserPoint2D, double>(2), d2(delta * delta), calls(0) {}
  virtual void estimate(std::vector<UserPoint2D *> &data, std::vector<double> &p) {
    p.clear();
    if (data.size() < 2) return;
    double nx = data[1]->y - data[0]->y, ny = data[0]->x - data[1]->x;
    double norm = std::sqrt(nx * nx + ny * ny);
    if (norm < 2.220446049250313e-16) return;
    p.push_back(nx / norm);
    p.push_back(ny / norm);
    p.push_back(data[0]->x);
    p.push_back(data[0]->y);
  }
  virtual void estimate(std::vector<UserPoint2D> &data, std::vector<double> &p) {
    std::vector<UserPoint2D *> q;
    for (size_t i = 0; i < data.size(); i++) q.push_back(&data[i]);
    estimate(q, p);
  }
  virtual void leastSquaresEstimate(std::vector<UserPoint2D *> &data, std::vector<double> &p) {
    p.clear();
    if (data.size() < 2) return;
    double mx = 0, my = 0, sxx = 0, sxy = 0, syy = 0;
    for (size_t i = 0; i < data.size(); i++) mx += data[i]->x, my += data[i]->y;
    mx /= data.size(), my /= data.size();
    for (size_t i = 0; i < data.size(); i++) {
      double dx = data[i]->x - mx, dy = data[i]->y - my;
      sxx += dx * dx, sxy += dx * dy, syy += dy * dy;
    }
    double th = 0.5 * std::atan2(2 * sxy, sxx - syy);  // direction of largest spread
    p.push_back(-std::sin(th));
    p.push_back(std::cos(th));
    p.push_back(mx);
    p.push_back(my);
  }
  virtual void leastSquaresEstimate(std::vector<UserPoint2D> &data, std::vector<double> &p) {
    std::vector<UserPoint2D *> q;
    for (size_t i = 0; i < data.size(); i++) q.push_back(&data[i]);
    leastSquaresEstimate(q, p);
  }
  virtual bool agree(std::vector<double> &p, UserPoint2D &d) {
    calls++;
    double s = p[0] * (d.x - p[2]) + p[1] * (d.y - p[3]);
    return s * s < d2;
  }
  double d2;
  size_t calls;
};

static void pluginTest() {
  std::vector<UserPoint2D> data;
  std::vector<Point2D> same;
  const double n[2] = {0.6, 0.8}, a[2] = {5, -7};
  for (int i = 0; i < 3000; i++) {
    double x = U(-500, 500), y = U(-500, 500);
    if (i % 5 < 3) {  // 60 % inliers
      double d = (x - a[0]) * n[0] + (y - a[1]) * n[1];
      x += -d * n[0] + N(0.2), y += -d * n[1] + N(0.2);
    }
    UserPoint2D u = {x, y};
    Point2D q;
    q[0] = x, q[1] = y;
    data.push_back(u);
    same.push_back(q);
  }
  UserLine2D user(0.5);
  lsqr_model_cfg none;
  CHECK(!user.deviceModel(none));
  std::vector<double> pu, pd;
  std::vector<bool> cu, cd;
  RANSAC<UserPoint2D, double>::seed() = 77;
  double fu = RANSAC<UserPoint2D, double>::compute(pu, &user, data, 0.999, &cu);
  lsqr_ransac_info iu = RANSAC<UserPoint2D, double>::lastInfo();
  CHECK(pu.size() == 4 && cu.size() == data.size());
  CHECK(fu > 0.55 && fu < 0.65);
  CHECK(user.calls > 0);
  if (pu.size() == 4) CHECK(std::fabs(std::fabs(pu[0] * n[0] + pu[1] * n[1]) - 1) < 1e-5);
  // the built-in estimator of the same model on the device, same stream
  Line2DParametersEstimator builtin(0.5);
  RANSAC<Point2D, double>::seed() = 77;
  double fd = RANSAC<Point2D, double>::compute(pd, &builtin, same, 0.999, &cd);
  lsqr_ransac_info id = RANSAC<Point2D, double>::lastInfo();
  CHECK(fd == fu && cd == cu);
  CHECK(id.iterations == iu.iterations && id.best_index == iu.best_index && id.best_votes == iu.best_votes);
  if (pd.size() == 4 && pu.size() == 4) {
    double dot = pd[0] * pu[0] + pd[1] * pu[1];
    CHECK(std::fabs(std::fabs(dot) - 1) < 1e-9);
  }
  // a built-in estimator forced through the plugin loop (virtuals evaluated one call at a time on the device)
  // gives what the batched device path gives: small set, or the per-datum calls take minutes
  typedef Point<double, 3> P;
  std::vector<P> pts;
  for (int i = 0; i < 60; i++) {
    P p;
    for (int j = 0; j < 3; j++) p[j] = U(-100, 100);
    if (i % 3) p[2] = 0.25 * p[0] - 0.5 * p[1] + 3 + N(0.05);
    pts.push_back(p);
  }
  PlaneParametersEstimator<3> plane(0.3);
  std::vector<double> a1, a2;
  std::vector<bool> c1, c2;
  RANSAC<P, double>::seed() = 5;
  double f1 = RANSAC<P, double>::compute(a1, &plane, pts, 0.99, &c1);
  lsqr_ransac_info i1 = RANSAC<P, double>::lastInfo();
  RANSAC<P, double>::forceHostLoop() = true;
  double f2 = RANSAC<P, double>::compute(a2, &plane, pts, 0.99, &c2);
  lsqr_ransac_info i2 = RANSAC<P, double>::lastInfo();
  CHECK(f1 == f2 && c1 == c2 && a1.size() == 6 && a2.size() == 6);
  CHECK(i1.iterations == i2.iterations && i1.best_index == i2.best_index);
  // exhaustive overload through the plugin loop == device
  std::vector<P> few(pts.begin(), pts.begin() + 9);
  double e2 = RANSAC<P, double>::compute(a2, &plane, few, &c2);
  RANSAC<P, double>::forceHostLoop() = false;
  double e1 = RANSAC<P, double>::compute(a1, &plane, few, &c1);
  CHECK((e1 == e2 && c1 == c2 && RANSAC<P, double>::lastInfo().iterations == 84));  // C(9,3)
  RANSAC<P, double>::seed() = 1;
  // degenerate user estimator (never produces a model): no consensus, parameters empty, return 0
  struct Mute : public UserLine2D {
    Mute() : UserLine2D(0.5) {}
    void estimate(std::vector<UserPoint2D *> &, std::vector<double> &p) { p.clear(); }
  } mute;
  std::vector<UserPoint2D> tiny(data.begin(), data.begin() + 6);
  std::vector<double> pm(2, 1.0);
  CHECK((RANSAC<UserPoint2D, double>::compute(pm, &mute, tiny, 0.9) == 0));
  CHECK(pm.empty());
}

// ---- dimensions above 3 (the reference's templates take any dimension; its sphere test runs 4-D) -----------
template <unsigned int D>
static void sphereNdTest() {  // testing/SphereParametersEstimatorTest.cxx:379-431 (testnD), data :432-468
  typedef Point<double, D> P;
  const double sigma = 1.0;
  std::vector<double> truth, params;
  for (unsigned i = 0; i < D; i++) truth.push_back(U(-1000, 1000));
  truth.push_back(U(0, 1000));
  std::vector<P> data, clean;
  for (unsigned i = 0; i < 10 * (D + 1); i++) {
    double t[D], nn = 0;
    for (unsigned j = 0; j < D; j++) t[j] = U(-1, 1), nn += t[j] * t[j];
    nn = std::sqrt(nn);
    P c, p;
    for (unsigned j = 0; j < D; j++) {
      c[j] = truth[j] + truth[D] * t[j] / nn;
      p[j] = c[j] + N(sigma);
    }
    clean.push_back(c);
    data.push_back(p);
  }
  SphereParametersEstimator<D> est(0.5);
  auto close = [&](const std::vector<double> &e) {
    if (e.size() != D + 1) return false;
    double d = 0;
    for (unsigned j = 0; j < D; j++) d += (e[j] - truth[j]) * (e[j] - truth[j]);
    return std::sqrt(d) <= 3 * sigma && (e[D] - truth[D]) <= 3 * sigma;
  };
  est.estimate(clean, params);
  CHECK(close(params));
  est.setLeastSquaresType(SphereParametersEstimator<D>::ALGEBRAIC);
  est.leastSquaresEstimate(data, params);
  CHECK(close(params));
  est.setLeastSquaresType(SphereParametersEstimator<D>::GEOMETRIC);
  est.leastSquaresEstimate(data, params);
  CHECK(close(params));
  CHECK(est.agree(truth, clean[3]));
  P off = clean[3];
  off[0] += 2.0;
  CHECK(!est.agree(truth, off) || std::fabs(off[0] - truth[0]) < 1.0);
  // distances vector of getDistanceStatistics (SphereParametersEstimator.h:156-159): appended
  std::vector<double> dist(2, -1.0);
  double mn, mx, mean;
  SphereParametersEstimator<D>::getDistanceStatistics(truth, clean, dist, mn, mx, mean);
  CHECK(dist.size() == 2 + clean.size() && dist[0] == -1.0 && dist[1] == -1.0);
  double sum = 0, big = 0;
  for (size_t i = 2; i < dist.size(); i++) sum += dist[i], big = std::max(big, dist[i]);
  CHECK(mx == big && std::fabs(mean - sum / clean.size()) < 1e-12 && mx < 1e-9);
  // RANSAC in D dimensions with gross outliers
  std::vector<P> mixed(data);
  for (unsigned i = 0; i < 4 * (D + 1); i++) {
    P p;
    for (unsigned j = 0; j < D; j++) p[j] = U(-2000, 2000);
    mixed.push_back(p);
  }
  SphereParametersEstimator<D> rest(4.0);
  std::vector<bool> cons;
  double frac = RANSAC<P, double>::compute(params, &rest, mixed, 0.99, &cons);
  CHECK(params.size() == D + 1 && frac > 0.5);
  CHECK(close(params));
}

template <unsigned int D>
static void planeLineNdTest() {
  typedef Point<double, D> P;
  double n[D], a[D], nn = 0;
  for (unsigned j = 0; j < D; j++) n[j] = U(-1, 1), a[j] = U(-100, 100), nn += n[j] * n[j];
  for (unsigned j = 0; j < D; j++) n[j] /= std::sqrt(nn);
  std::vector<P> onPlane, onLine;
  for (unsigned i = 0; i < 40 * D; i++) {
    P p, q;
    double d = 0, t = U(-300, 300);
    for (unsigned j = 0; j < D; j++) p[j] = U(-500, 500), d += (p[j] - a[j]) * n[j];
    for (unsigned j = 0; j < D; j++) {
      p[j] += -d * n[j] + N(0.05);
      q[j] = a[j] + t * n[j] + N(0.05);
    }
    if (i % 4 == 3)
      for (unsigned j = 0; j < D; j++) p[j] = U(-500, 500), q[j] = U(-500, 500);
    onPlane.push_back(p);
    onLine.push_back(q);
  }
  std::vector<double> params;
  std::vector<bool> cons;
  PlaneParametersEstimator<D> pe(0.3);
  std::vector<P> minimal;
  for (unsigned i = 0; minimal.size() < D; i++)
    if (i % 4 != 3) minimal.push_back(onPlane[i]);
  pe.estimate(minimal, params);   // SVD null-vector branch (PlaneParametersEstimator.hxx:70-104)
  CHECK(params.size() == 2 * D);
  if (params.size() == 2 * D) {
    double dot = 0;
    for (unsigned j = 0; j < D; j++) dot += params[j] * n[j];
    CHECK(std::fabs(dot) > 0.999);
    for (unsigned j = 0; j < D; j++) CHECK(params[D + j] == minimal[0][j]);
  }
  std::vector<P> repeated(D, minimal[0]);  // rank-deficient minimal set -> empty
  pe.estimate(repeated, params);
  CHECK(params.empty());
  double frac = RANSAC<P, double>::compute(params, &pe, onPlane, 0.99, &cons);
  CHECK(params.size() == 2 * D && frac > 0.7 && frac < 0.8);
  if (params.size() == 2 * D) {
    double dot = 0, off = 0;
    for (unsigned j = 0; j < D; j++) dot += params[j] * n[j], off += (params[D + j] - a[j]) * n[j];
    CHECK(std::fabs(std::fabs(dot) - 1) < 1e-6 && std::fabs(off) < 0.1);
    CHECK(pe.agree(params, onPlane[0]));
  }
  LineParametersEstimator<D> le(0.5);
  frac = RANSAC<P, double>::compute(params, &le, onLine, 0.99, &cons);
  CHECK(params.size() == 2 * D && frac > 0.7 && frac < 0.8);
  if (params.size() == 2 * D) {
    double dot = 0;
    for (unsigned j = 0; j < D; j++) dot += params[j] * n[j];
    CHECK(std::fabs(std::fabs(dot) - 1) < 1e-6);
  }
}

static void absoluteOrientationTest() {  // testing/AbsoluteOrientationParametersEstimatorTest.cxx:19-118
  typedef std::pair<Point3D, Point3D> DataType;
  const int pairNum = 10;
  const double bounds = 100.0, maxTranslation = 1000.0, noiseSigma = 5.0 / 3.0;
  double qx = U(0.0, 1.0), qy = U(0.0, std::sqrt(1.0 - qx * qx));
  double qz = U(0.0, std::sqrt(1.0 - qx * qx - qy * qy));
  std::vector<double> known;
  known.push_back(std::sqrt(1.0 - qx * qx - qy * qy - qz * qz));
  known.push_back(qx); known.push_back(qy); known.push_back(qz);
  for (int i = 0; i < 3; i++) known.push_back(U(-maxTranslation, maxTranslation));
  Frame T(known[4], known[5], known[6], known[0], known[1], known[2], known[3]);
  std::vector<DataType> clean, noisy, targets;
  DataType pr, outlier;
  for (int i = 0; i < pairNum; i++) {
    for (int k = 0; k < 3; k++) pr.first[k] = U(-bounds, bounds);
    T.apply(pr.first, pr.second);
    clean.push_back(pr);
    for (int k = 0; k < 3; k++) pr.second[k] += N(noiseSigma);
    noisy.push_back(pr);
    for (int k = 0; k < 3; k++) pr.first[k] = U(-bounds, bounds);
    T.apply(pr.first, pr.second);
    targets.push_back(pr);
  }
  for (int k = 0; k < 3; k++) outlier.first[k] = U(-bounds, bounds);
  T.apply(outlier.first, outlier.second);
  outlier.second[0] += 10 * noiseSigma;
  AbsoluteOrientationParametersEstimator est(1.0);
  const double distanceThreshold = 3.0 * noiseSigma;
  std::vector<double> par;
  for (int pass = 0; pass < 2; pass++) {  // exact from three clean pairs, LS from the noisy ones
    if (pass == 0) est.estimate(clean, par);
    else est.leastSquaresEstimate(noisy, par);
    CHECK(par.size() == 7);
    if (par.size() != 7) continue;
    Frame E(par[4], par[5], par[6], par[0], par[1], par[2], par[3], true);
    double worst = 0;
    for (size_t i = 0; i < targets.size(); i++) {
      Point3D q;
      E.apply(targets[i].first, q);
      worst = std::max(worst, std::sqrt(q.distanceSquared(targets[i].second)));
    }
    CHECK(worst < distanceThreshold);
  }
  CHECK(est.agree(known, clean[0]));
  CHECK(!est.agree(known, outlier));
}

static void rayIntersectionTest() {  // testing/RayIntersectionParametersTest.cxx:15-117
  const unsigned NUM_RAYS = 10;
  const double NOISE_SIGMA = 20.0, maxRange = 1000.0, maxDistanceToRay = 0.5;
  Point3D known;
  for (int k = 0; k < 3; k++) known[k] = U(-maxRange, maxRange);
  std::vector<Ray3D> rayData, noNoise;
  Ray3D ray;
  for (unsigned i = 0; i < NUM_RAYS + 2; i++) {
    for (int k = 0; k < 3; k++) ray.p[k] = U(-maxRange, maxRange);
    for (int k = 0; k < 3; k++) ray.n[k] = known[k] + (i < NUM_RAYS ? N(NOISE_SIGMA) : 0.0) - ray.p[k];
    ray.n.normalize();
    (i < NUM_RAYS ? rayData : noNoise).push_back(ray);
  }
  RayIntersectionParametersEstimator est(maxDistanceToRay);
  std::vector<double> pt(3);
  for (int k = 0; k < 3; k++) pt[k] = known[k];
  CHECK(est.agree(pt, noNoise[0]));
  est.estimate(noNoise, pt);
  CHECK(pt.size() == 3);
  if (pt.size() == 3) {
    Point3D tmp;
    for (int k = 0; k < 3; k++) tmp[k] = pt[k];
    CHECK(std::sqrt(tmp.distanceSquared(known)) <= maxDistanceToRay);
  }
  est.leastSquaresEstimate(rayData, pt);
  CHECK(pt.size() == 3);
}

static void pivotTest(const char *file) {  // testing/PivotCalibrationParametersEstimatorTest.cxx:19-119
  if (!file) return;
  std::ifstream in(file);
  CHECK(in.is_open());
  std::vector<Frame> poses;
  Frame f;
  double x, y, z, qx, qy, qz, qs;
  while (in >> x >> y >> z >> qx >> qy >> qz >> qs) {
    f.setRotationQuaternion(qs, qx, qy, qz);
    f.setTranslation(x, y, z);
    poses.push_back(f);
  }
  CHECK(!poses.empty());
  if (poses.empty()) return;
  const double maxError = 1.0;
  PivotCalibrationEstimator pivot(maxError);
  const double knownExact[] = {-18.586, 1.98134, -157.439, 146.965, -62.0497, -1042.87};
  const double knownLS[] = {-17.7799, 1.1113, -156.865, 146.901, -62.9689, -1042.14};
  std::vector<Frame> mins(3);
  mins[0] = poses[0];
  mins[1] = poses[(unsigned)(poses.size() / 2.0)];
  mins[2] = poses[poses.size() - 1];
  std::vector<double> est;
  pivot.estimate(mins, est);
  CHECK(est.size() == 6);
  for (size_t i = 0; i < est.size(); i++) CHECK(std::fabs(est[i] - knownExact[i]) < maxError);
  for (size_t i = 0; i < mins.size() && est.size() == 6; i++) CHECK(pivot.agree(est, mins[i]));
  pivot.leastSquaresEstimate(poses, est);
  CHECK(est.size() == 6);
  for (size_t i = 0; i < est.size(); i++) CHECK(std::fabs(est[i] - knownLS[i]) < maxError);
}

static void weightedAbsoluteOrientationTest() {  // AbsoluteOrientationParametersEstimator.h:86, .cxx:208-291
  typedef std::pair<Point3D, Point3D> Pair;
  // ground truth: rotation about (1,2,3)/|.| by 0.7 rad, translation (5,-3,2)
  const double ax[3] = {1 / std::sqrt(14.0), 2 / std::sqrt(14.0), 3 / std::sqrt(14.0)}, ang = 0.7;
  const double q[4] = {std::cos(ang / 2), ax[0] * std::sin(ang / 2), ax[1] * std::sin(ang / 2),
                       ax[2] * std::sin(ang / 2)};
  Frame f(5, -3, 2, q[0], q[1], q[2], q[3]);
  std::vector<Pair> pairs;
  for (int i = 0; i < 12; i++) {
    Pair pr;
    for (int j = 0; j < 3; j++) pr.first[j] = U(-100, 100);
    f.apply(pr.first, pr.second);
    pairs.push_back(pr);
  }
  for (int j = 0; j < 3; j++) pairs[2].second[j] += 40;   // two corrupted pairs ...
  for (int j = 0; j < 3; j++) pairs[7].second[j] -= 25;
  std::vector<Pair *> ptrs;
  for (size_t i = 0; i < pairs.size(); i++) ptrs.push_back(&pairs[i]);
  std::vector<double> w(pairs.size(), 1.0), pw, pu, pz;
  AbsoluteOrientationParametersEstimator est(0.5);
  est.weightedLeastSquaresEstimate(ptrs, w, pw);   // unit weights == the plain fit
  est.leastSquaresEstimate(ptrs, pu);
  CHECK(pw.size() == 7 && pu.size() == 7);
  for (size_t i = 0; i < 7 && pw.size() == 7 && pu.size() == 7; i++) CHECK(std::fabs(pw[i] - pu[i]) < 1e-12);
  w[2] = w[7] = 0.0;                                // ... weighted out: the exact transformation
  est.weightedLeastSquaresEstimate(ptrs, w, pz);
  CHECK(pz.size() == 7);
  if (pz.size() == 7) {
    double s = pz[0] * q[0] < 0 ? -1.0 : 1.0;
    for (int i = 0; i < 4; i++) CHECK(std::fabs(s * pz[i] - q[i]) < 1e-9);
    CHECK(std::fabs(pz[4] - 5) < 1e-7 && std::fabs(pz[5] + 3) < 1e-7 && std::fabs(pz[6] - 2) < 1e-7);
  }
  if (pu.size() == 7) CHECK(std::fabs(pu[4] - 5) > 0.5 || std::fabs(pu[5] + 3) > 0.5 || std::fabs(pu[6] - 2) > 0.5);
  for (size_t i = 0; i < w.size(); i++) w[i] = U(0.1, 3.0);   // non-uniform weights halved: same answer
  std::vector<double> h(w), p1, p2;
  for (size_t i = 0; i < h.size(); i++) h[i] *= 0.5;
  est.weightedLeastSquaresEstimate(ptrs, w, p1);
  est.weightedLeastSquaresEstimate(ptrs, h, p2);
  CHECK(p1.size() == 7 && p2.size() == 7);
  for (size_t i = 0; i < 7 && p1.size() == 7 && p2.size() == 7; i++) CHECK(std::fabs(p1[i] - p2[i]) < 1e-9);
  std::vector<Pair *> two(ptrs.begin(), ptrs.begin() + 2);
  est.weightedLeastSquaresEstimate(two, w, p1);
  CHECK(p1.empty());
}

// records resident across compute() calls (lsqrRecipes::ResidentData): same results as the vector overload,
// whatever the order of the calls; agree() / estimate() on the host give what the device gives
static void residentTest() {
  typedef Point<double, 3> P;
  std::vector<P> pts;
  const double n[3] = {0.6, -0.48, 0.64}, a[3] = {10, 20, -30};
  for (int i = 0; i < 200000; i++) {
    P p;
    for (int j = 0; j < 3; j++) p[j] = U(-300, 300);
    if (i % 2) {
      double d = 0;
      for (int j = 0; j < 3; j++) d += (p[j] - a[j]) * n[j];
      for (int j = 0; j < 3; j++) p[j] += -d * n[j] + N(0.3);
    }
    pts.push_back(p);
  }
  ResidentData<P> resident(pts);
  const double deltas[3] = {0.5, 1.0, 0.25};
  for (int rep = 0; rep < 3; rep++) {
    PlaneParametersEstimator<3> est(deltas[rep]);
    std::vector<double> pv, pr;
    std::vector<bool> cv, cr;
    RANSAC<P, double>::seed() = 5 + rep;
    double fv = RANSAC<P, double>::compute(pv, &est, pts, 0.999, &cv);
    lsqr_ransac_info iv = RANSAC<P, double>::lastInfo();
    RANSAC<P, double>::seed() = 5 + rep;
    double fr = RANSAC<P, double>::compute(pr, &est, resident, 0.999, &cr);
    lsqr_ransac_info ir = RANSAC<P, double>::lastInfo();
    CHECK(fv == fr && cv == cr && pv == pr && pv.size() == 6);
    CHECK(iv.iterations == ir.iterations && iv.best_index == ir.best_index && iv.best_votes == ir.best_votes);
    CHECK(fr > 0.2);
    // agree() per datum (host) against the consensus set of the device run, estimate() of a subset (host) against
    // the device's minimal solve
    if (pr.size() == 6 && cr.size() == pts.size()) {
      std::vector<double> winner;
      std::vector<P *> sub;
      // (the winner's own parameters are not returned by compute(): check agree() with the fitted model's mask
      //  computed by the device)
      int bad = 0;
      lsqr_model_cfg cfg;
      est.deviceModel(cfg);
      lsqr_ctx *ctx = resident.attach(cfg);
      std::vector<uint8_t> m(pts.size());
      resident.check(lsqr_mask(ctx, &pr[0], 0, pts.size(), &m[0], 0));
      for (size_t i = 0; i < pts.size(); i += 97) bad += (est.agree(pr, pts[i]) != (m[i] != 0));
      CHECK(bad == 0);
      sub.push_back(&pts[3]), sub.push_back(&pts[1001]), sub.push_back(&pts[77777]);
      est.estimate(sub, winner);
      uint32_t idx[3] = {3, 1001, 77777};
      double hp[6];
      uint8_t valid = 0;
      resident.check(lsqr_hypotheses_from_subsets(ctx, idx, 1));
      resident.check(lsqr_get_hypothesis(ctx, 0, hp, &valid));
      CHECK(valid && winner.size() == 6);
      if (winner.size() == 6)
        for (int j = 0; j < 6; j++) CHECK(winner[j] == hp[j]);
    }
  }
  // another estimator of the same record type on the same resident records
  SphereParametersEstimator<3> sph(0.5);
  std::vector<double> ps;
  RANSAC<P, double>::compute(ps, &sph, resident, 0.9);
  CHECK(ps.empty() || ps.size() == 4);
}

int main(int argc, char *argv[]) {
  try {
    planeTest();
    residentTest();
    sphereTest();
    lineTest();
    denseTest(argc > 1 ? argv[1] : 0);
    usTest();
    phantomTest();
    absoluteOrientationTest();
    rayIntersectionTest();
    pivotTest(argc > 2 ? argv[2] : 0);
    ransacTest();
    pluginTest();
    sphereNdTest<4>();
    sphereNdTest<6>();
    planeLineNdTest<4>();
    planeLineNdTest<5>();
    planeLineNdTest<8>();
    weightedAbsoluteOrientationTest();
  } catch (std::exception &e) {
    std::printf("EXCEPTION: %s\n", e.what());
    return EXIT_FAILURE;
  }
  std::printf(failures ? "%d check(s) FAILED\n" : "all checks passed (%d failures)\n", failures);
  return failures ? EXIT_FAILURE : EXIT_SUCCESS;
}
