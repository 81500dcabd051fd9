// lineEstimation -- counterpart of the reference's examples/lineEstimation.cxx; both fits are also
// saved as Open Inventor scenes (leastSquaresLineEstimation.iv, RANSACLineEstimation.iv).
#include <cstdlib>
#include <iostream>

#include "LineParametersEstimator.h"
#include "RANSAC.h"
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
  Rng rng(11);
  double d[DIM], a[DIM], nd = 0;
  for (unsigned i = 0; i < DIM; i++) {
    d[i] = rng.uniform();
    a[i] = rng.uniform(-1000, 1000);
    nd += d[i] * d[i];
  }
  for (unsigned i = 0; i < DIM; i++) d[i] /= std::sqrt(nd);
  std::vector<P> data;
  for (unsigned i = 0; i < 90; i++) {
    double t = rng.uniform(-1000, 1000);
    P p;
    for (unsigned j = 0; j < DIM; j++) p[j] = a[j] + t * d[j] + rng.normal(0.4);
    data.push_back(p);
  }
  for (unsigned i = 0; i < 10; i++) {
    P p;
    for (unsigned j = 0; j < DIM; j++) p[j] = rng.uniform(-1000, 1000);
    data.push_back(p);
  }
  std::vector<double> params;
  lsqrRecipes::LineParametersEstimator<DIM> estimator(0.5);
  estimator.leastSquaresEstimate(data, params);
  printVec("Least squares line parameters [direction,a]", params);
  if (!params.empty()) {
    OivScene scene(oivPath("leastSquaresLineEstimation.iv"));
    scene.observations(data, classify(estimator, params, data), 50.0);
    scene.line(params);
  }
  std::vector<bool> consensus;
  double used = lsqrRecipes::RANSAC<P, double>::compute(params, &estimator, data, 0.999, &consensus);
  if (params.empty()) return EXIT_FAILURE;
  printVec("RANSAC line parameters [direction,a]", params);
  double dot = 0;
  for (unsigned i = 0; i < DIM; i++) dot += params[i] * d[i];
  std::cout << "\tDot product of real and computed directions[+-1=correct]: " << dot << "\n";
  std::cout << "\tPercentage of points which were used for final estimate: " << used << "\n";
  OivScene scene(oivPath("RANSACLineEstimation.iv"));
  scene.observations(data, consensus, 50.0);
  scene.line(params);
  return std::fabs(std::fabs(dot) - 1.0) < 1e-5 ? EXIT_SUCCESS : EXIT_FAILURE;
}
