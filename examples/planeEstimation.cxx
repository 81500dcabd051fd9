// planeEstimation -- counterpart of the reference's examples/planeEstimation.cxx (main :60-148,
// generateData :152-200) on the MI355X drop-in: synthesise points near a random plane plus outliers,
// fit by plain least squares, then robustly with RANSAC; print the same quantities.  Returns non-zero
// if the robust fit misses the known plane.  Both fits are also saved as Open Inventor scenes
// (leastSquaresPlaneEstimation.iv, RANSACPlaneEstimation.iv; reference :65-66,:110,:145).
// usage: planeEstimation [inliers outliers]
#include <cstdlib>
#include <iostream>

#include "PlaneParametersEstimator.h"
#include "RANSAC.h"
#include "common.h"
#include "oiv.h"

// scene files go to $LSQR_OIV_DIR (default: the working directory, as the reference's example does)
static std::string oivPath(const char *name) {
  const char *dir = std::getenv("LSQR_OIV_DIR");
  return std::string(dir ? dir : ".") + "/" + name;
}

int main(int argc, char *argv[]) {
  const unsigned int DIM = 3;
  unsigned int inliers = argc > 2 ? std::atoi(argv[1]) : 90, outliers = argc > 2 ? std::atoi(argv[2]) : 10;
  typedef lsqrRecipes::Point<double, DIM> P;
  Rng rng(20261003);
  double n[DIM], a[DIM], nn = 0;
  for (unsigned i = 0; i < DIM; i++) {
    n[i] = rng.uniform();
    a[i] = rng.uniform(-1000, 1000);
    nn += n[i] * n[i];
  }
  for (unsigned i = 0; i < DIM; i++) n[i] /= std::sqrt(nn);
  std::vector<P> data;
  for (unsigned i = 0; i < inliers; i++) {  // random point projected on the plane + N(0, 0.4^2)
    P p;
    double d = 0;
    for (unsigned j = 0; j < DIM; j++) {
      p[j] = rng.uniform(-1000, 1000);
      d += (p[j] - a[j]) * n[j];
    }
    for (unsigned j = 0; j < DIM; j++) p[j] += -d * n[j] + rng.normal(0.4);
    data.push_back(p);
  }
  while (data.size() < inliers + outliers) {  // outliers at least 20 away from the plane
    P p;
    double d = 0;
    for (unsigned j = 0; j < DIM; j++) {
      p[j] = rng.uniform(-1000, 1000);
      d += (p[j] - a[j]) * n[j];
    }
    if (std::fabs(d) >= 20.0) data.push_back(p);
  }
  std::vector<double> truth(n, n + DIM), params;
  truth.insert(truth.end(), a, a + DIM);
  printVec("Known (hyper)plane parameters [n,a]", truth);

  lsqrRecipes::PlaneParametersEstimator<DIM> estimator(0.5);
  estimator.leastSquaresEstimate(data, params);
  if (params.empty()) std::cout << "Least squares estimate failed, degenerate configuration?\n";
  else {
    printVec("Least squares hyper(plane) parameters: [n,a]", params);
    double dot = 0, off = 0;
    for (unsigned i = 0; i < DIM; i++) {
      dot += params[i] * n[i];
      off += (params[DIM + i] - a[i]) * n[i];
    }
    std::cout << "\tDot product of real and computed normals[+-1=correct]: " << dot << "\n";
    std::cout << "\tCheck if computed point is on known plane [0=correct]: " << off << "\n\n";
    OivScene scene(oivPath("leastSquaresPlaneEstimation.iv"));
    scene.observations(data, classify(estimator, params, data), 50.0);
    scene.plane(params);
  }
  std::vector<bool> consensus;
  double used = lsqrRecipes::RANSAC<P, double>::compute(params, &estimator, data, 0.999, &consensus);
  if (params.empty()) {
    std::cout << "RANSAC estimate failed, degenerate configuration?\n";
    return EXIT_FAILURE;
  }
  printVec("RANSAC hyper(plane) parameters: [n,a]", params);
  double dot = 0, off = 0;
  for (unsigned i = 0; i < DIM; i++) {
    dot += params[i] * n[i];
    off += (params[DIM + i] - a[i]) * n[i];
  }
  std::cout << "\tDot product of real and computed normals[+-1=correct]: " << dot << "\n";
  std::cout << "\tCheck if computed point is on known plane [0=correct]: " << off << "\n\n";
  std::cout << "\tPercentage of points which were used for final estimate: " << used << "\n\n";
  OivScene scene(oivPath("RANSACPlaneEstimation.iv"));
  scene.observations(data, consensus, 50.0);
  scene.plane(params);
  return (std::fabs(std::fabs(dot) - 1.0) < 1e-5 && std::fabs(off) < 0.5) ? EXIT_SUCCESS : EXIT_FAILURE;
}
