// crosswireUSCalibration -- counterpart of the reference's examples/crosswireUSCalibration.cxx
// (:30-85): load tracker transformations (rows "R00 R01 R02 t0" x 3 per frame) and the 2D image
// points of the cross-wire, calibrate analytically / iteratively and robustly with RANSAC.
// usage: crosswireUSCalibration transformationsFile pointsFile     (without arguments: simulated)
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "RANSAC.h"
#include "SinglePointTargetUSCalibrationParametersEstimator.h"
#include "common.h"

typedef lsqrRecipes::SingleUnknownPointTargetUSCalibrationParametersEstimator Estimator;
typedef Estimator::DataType DataType;

static bool load(const char *tf, const char *pf, std::vector<DataType> &data) {
  std::ifstream t(tf), p(pf);
  if (!t.is_open() || !p.is_open()) return false;
  double R[3][3], tr[3], u, v;
  while (p >> u >> v) {
    for (int i = 0; i < 3; i++)
      if (!(t >> R[i][0] >> R[i][1] >> R[i][2] >> tr[i])) return !data.empty();
    DataType d;
    d.T2.setRotationMatrix(R);
    d.T2.setTranslation(tr);
    d.q[0] = u;
    d.q[1] = v;
    data.push_back(d);
  }
  return !data.empty();
}

static void simulate(std::vector<DataType> &data) {
  Rng rng(5);
  const double mx = 0.143, my = 0.139, PI = 3.14159265358979323846;
  lsqrRecipes::Frame T3;
  T3.setRotationEulerAngles(rng.uniform(0, PI), rng.uniform(0, PI), rng.uniform(0, PI));
  T3.setTranslation(rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-100, 100));
  double t1[3] = {rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-100, 100)};
  for (int i = 0; i < 60; i++) {
    DataType d;
    double u = rng.uniform(0, 640), v = rng.uniform(0, 480);
    d.T2.setRotationEulerAngles(rng.uniform(0, PI), rng.uniform(0, PI), rng.uniform(0, PI));
    double q[3] = {mx * u, my * v, 0}, q3[3], rq[3];
    T3.apply(q, q3);
    d.T2.setTranslation(0, 0, 0);
    d.T2.apply(q3, rq);
    if (i % 6 == 5)  // outlier frame
      d.T2.setTranslation(rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-100, 100));
    else
      d.T2.setTranslation(t1[0] - rq[0], t1[1] - rq[1], t1[2] - rq[2]);
    d.q[0] = u + rng.normal(0.5);
    d.q[1] = v + rng.normal(0.5);
    data.push_back(d);
  }
}

int main(int argc, char *argv[]) {
  std::vector<DataType> data;
  if (argc == 3) {
    if (!load(argv[1], argv[2], data)) {
      std::cerr << "Failed to load data files.\n";
      return EXIT_FAILURE;
    }
  } else
    simulate(data);
  std::cout << data.size() << " frames\n";
  std::vector<double> params;
  Estimator estimator(3.0, Estimator::ANALYTIC);
  estimator.leastSquaresEstimate(data, params);
  printVec("Analytic least squares [t1, t3, wz, wy, wx, mx, my, ...]", params);
  double mn, mx_, mean;
  if (!params.empty()) {
    Estimator::getDistanceStatistics(params, data, mn, mx_, mean);
    std::cout << "\tdistance to target: min " << mn << " max " << mx_ << " mean " << mean << "\n\n";
  }
  std::vector<bool> consensus;
  double used = lsqrRecipes::RANSAC<DataType, double>::compute(params, &estimator, data, 0.999, &consensus);
  if (params.empty()) {
    std::cout << "RANSAC calibration failed\n";
    return EXIT_FAILURE;
  }
  printVec("RANSAC + analytic least squares", params);
  std::vector<DataType> inl;
  for (size_t i = 0; i < data.size(); i++)
    if (consensus[i]) inl.push_back(data[i]);
  Estimator::getDistanceStatistics(params, inl, mn, mx_, mean);
  std::cout << "\tPercentage of frames used: " << used << "\n";
  std::cout << "\tdistance to target over the consensus set: min " << mn << " max " << mx_
            << " mean " << mean << "\n";
  return (used > 0.5 && mx_ < 3.0) ? EXIT_SUCCESS : EXIT_FAILURE;
}
