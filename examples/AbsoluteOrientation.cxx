// AbsoluteOrientation -- counterpart of the reference's examples/AbsoluteOrientation.cxx: paired-point
// rigid registration of a handful of fiducials, one of them an outlier; plain least squares, then the
// EXHAUSTIVE RANSAC overload (all C(N,3) subsets, RANSAC.h:111-113) on the device.
#include <cmath>
#include <cstdlib>
#include <iostream>

#include "AbsoluteOrientationParametersEstimator.h"
#include "Frame.h"
#include "RANSAC.h"
#include "common.h"

typedef std::pair<lsqrRecipes::Point3D, lsqrRecipes::Point3D> DataType;

static double maxTargetError(const std::vector<double> &par, const std::vector<DataType> &targets) {
  lsqrRecipes::Frame f(par[4], par[5], par[6], par[0], par[1], par[2], par[3], true);
  double worst = 0;
  for (size_t i = 0; i < targets.size(); i++) {
    lsqrRecipes::Point3D q;
    f.apply(targets[i].first, q);
    double e = 0;
    for (int k = 0; k < 3; k++) e += (q[k] - targets[i].second[k]) * (q[k] - targets[i].second[k]);
    worst = std::max(worst, std::sqrt(e));
  }
  return worst;
}

int main() {
  const int inliers = 5, outliers = 1;
  const double bounds = 100.0, maxTranslation = 1000.0, noiseSigma = 0.5;
  Rng rng(2026);
  double qx = rng.uniform(0.0, 1.0), qy = rng.uniform(0.0, std::sqrt(1.0 - qx * qx));
  double qz = rng.uniform(0.0, std::sqrt(1.0 - qx * qx - qy * qy));
  double qs = std::sqrt(1.0 - qx * qx - qy * qy - qz * qz);
  lsqrRecipes::Frame known(rng.uniform(-maxTranslation, maxTranslation),
                           rng.uniform(-maxTranslation, maxTranslation),
                           rng.uniform(-maxTranslation, maxTranslation), qs, qx, qy, qz);
  std::vector<DataType> data, targets;
  DataType pr;
  for (int i = 0; i < inliers + outliers; i++) {
    for (int k = 0; k < 3; k++) pr.first[k] = rng.uniform(-bounds, bounds);
    known.apply(pr.first, pr.second);
    for (int k = 0; k < 3; k++) pr.second[k] += rng.normal(noiseSigma);
    if (i >= inliers) pr.second[2] += 5.0;  // the outlier
    data.push_back(pr);
    for (int k = 0; k < 3; k++) pr.first[k] = rng.uniform(-bounds, bounds);
    known.apply(pr.first, pr.second);
    targets.push_back(pr);
  }
  std::vector<double> params;
  lsqrRecipes::AbsoluteOrientationParametersEstimator estimator(2 * noiseSigma);
  estimator.leastSquaresEstimate(data, params);
  if (params.empty()) return EXIT_FAILURE;
  printVec("Least squares transformation [s,qx,qy,qz,tx,ty,tz]", params);
  const double lsErr = maxTargetError(params, targets);
  std::cout << "\tMaximal target registration error: " << lsErr << "\n\n";

  std::vector<bool> consensus;
  double used = lsqrRecipes::RANSAC<DataType, double>::compute(params, &estimator, data, &consensus);
  if (params.empty()) return EXIT_FAILURE;
  printVec("Exhaustive search transformation [s,qx,qy,qz,tx,ty,tz]", params);
  const double rErr = maxTargetError(params, targets);
  std::cout << "\tMaximal target registration error: " << rErr << "\n";
  std::cout << "\tFiducials used in final estimate: ";
  for (size_t i = 0; i < consensus.size(); i++) std::cout << consensus[i] << " ";
  std::cout << "(" << used * 100 << "%)\n";
  // the deliberately displaced fiducial must not be part of the consensus set
  return (!consensus.empty() && !consensus.back() && used >= 0.5) ? EXIT_SUCCESS : EXIT_FAILURE;
}
