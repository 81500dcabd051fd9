// sphereEstimation -- counterpart of the reference's examples/sphereEstimation.cxx (:60-184):
// points near a random sphere plus outliers; algebraic and geometric least squares, then RANSAC.
#include <cstdlib>
#include <iostream>

#include "RANSAC.h"
#include "SphereParametersEstimator.h"
#include "common.h"
#include "oiv.h"

// scene files go to $LSQR_OIV_DIR (default: the working directory, as the reference's example does)
static std::string oivPath(const char *name) {
  const char *dir = std::getenv("LSQR_OIV_DIR");
  return std::string(dir ? dir : ".") + "/" + name;
}

int main() {
  const unsigned int DIM = 3;
  typedef lsqrRecipes::Point<double, DIM> P;
  typedef lsqrRecipes::SphereParametersEstimator<DIM> Est;
  Rng rng(7);
  double c[DIM], r = rng.uniform(100, 1000);
  for (unsigned i = 0; i < DIM; i++) c[i] = rng.uniform(-1000, 1000);
  std::vector<P> data;
  for (unsigned i = 0; i < 90; i++) {
    double u[DIM], nu = 0;
    for (unsigned j = 0; j < DIM; j++) {
      u[j] = rng.uniform(-1, 1);
      nu += u[j] * u[j];
    }
    P p;
    for (unsigned j = 0; j < DIM; j++) p[j] = c[j] + r * u[j] / std::sqrt(nu) + rng.normal(0.4);
    data.push_back(p);
  }
  while (data.size() < 100) {
    P p;
    double d = 0;
    for (unsigned j = 0; j < DIM; j++) {
      p[j] = rng.uniform(-1000, 1000);
      d += (p[j] - c[j]) * (p[j] - c[j]);
    }
    if (std::fabs(std::sqrt(d) - r) >= 20.0) data.push_back(p);
  }
  std::vector<double> truth(c, c + DIM), params;
  truth.push_back(r);
  printVec("Known (hyper)sphere parameters [c,r]", truth);
  Est estimator(0.5, Est::ALGEBRAIC);
  estimator.leastSquaresEstimate(data, params);
  printVec("Algebraic least squares parameters [c,r] (all data, outliers included)", params);
  estimator.setLeastSquaresType(Est::GEOMETRIC);
  estimator.leastSquaresEstimate(data, params);
  printVec("Geometric least squares parameters [c,r] (all data, outliers included)", params);
  if (!params.empty()) {
    OivScene scene(oivPath("leastSquaresSphereEstimation.iv"));
    scene.observations(data, classify(estimator, params, data), 15.0);
    scene.sphere(params);
  }
  std::vector<bool> consensus;
  double used = lsqrRecipes::RANSAC<P, double>::compute(params, &estimator, data, 0.999, &consensus);
  if (params.empty()) {
    std::cout << "RANSAC estimate failed\n";
    return EXIT_FAILURE;
  }
  printVec("RANSAC parameters [c,r]", params);
  double dc = 0;
  for (unsigned i = 0; i < DIM; i++) dc += (params[i] - c[i]) * (params[i] - c[i]);
  double mn, mx, mean;
  Est::getDistanceStatistics(params, data, mn, mx, mean);
  std::cout << "\tDistance between real and computed centers: " << std::sqrt(dc) << "\n";
  std::cout << "\tDifference between real and computed radius: " << params[DIM] - r << "\n";
  std::cout << "\tPercentage of points which were used for final estimate: " << used << "\n";
  std::cout << "\tResidual over all data: min " << mn << " max " << mx << " mean " << mean << "\n";
  OivScene scene(oivPath("RANSACSphereEstimation.iv"));
  scene.observations(data, consensus, 15.0);
  scene.sphere(params);
  return (std::sqrt(dc) < 1.0 && std::fabs(params[DIM] - r) < 1.0 && used > 0.5) ? EXIT_SUCCESS
                                                                                  : EXIT_FAILURE;
}
