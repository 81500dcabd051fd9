// rayIntersectionEstimation -- counterpart of the reference's examples/rayIntersectionEstimation.cxx:
// rays that (approximately) meet in one point plus rays that do not; plain least squares, then RANSAC.
#include <cmath>
#include <cstdlib>
#include <iostream>

#include "RANSAC.h"
#include "RayIntersectionParametersEstimator.h"
#include "common.h"

int main() {
  const unsigned inliers = 90, outliers = 10;
  const double maxRange = 1000.0, noiseSigma = 0.3;
  Rng rng(404);
  lsqrRecipes::Point3D target;
  for (int k = 0; k < 3; k++) target[k] = rng.uniform(-maxRange, maxRange);
  std::vector<lsqrRecipes::Ray3D> rays;
  lsqrRecipes::Ray3D ray;
  for (unsigned i = 0; i < inliers + outliers; i++) {
    for (int k = 0; k < 3; k++) ray.p[k] = rng.uniform(-maxRange, maxRange);
    for (int k = 0; k < 3; k++)
      ray.n[k] = (i < inliers ? target[k] + rng.normal(noiseSigma) : rng.uniform(-maxRange, maxRange)) - ray.p[k];
    ray.n.normalize();
    rays.push_back(ray);
  }
  std::vector<double> params;
  lsqrRecipes::RayIntersectionParametersEstimator estimator(1.0);
  estimator.leastSquaresEstimate(rays, params);
  if (params.empty()) return EXIT_FAILURE;
  printVec("Least squares intersection point [x,y,z]", params);
  double used = lsqrRecipes::RANSAC<lsqrRecipes::Ray3D, double>::compute(params, &estimator, rays, 0.999);
  if (params.empty()) return EXIT_FAILURE;
  printVec("RANSAC intersection point [x,y,z]", params);
  double err = 0;
  for (int k = 0; k < 3; k++) err += (params[k] - target[k]) * (params[k] - target[k]);
  std::cout << "\tDistance to the known intersection point: " << std::sqrt(err) << "\n";
  std::cout << "\tPercentage of rays used for the final estimate: " << used * 100 << "\n";
  return std::sqrt(err) < 1.0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
