// pivotCalibration -- counterpart of the reference's examples/pivotCalibration.cxx: tracked poses
// "x y z qx qy qz qs" per row (the reference's examples/Data/pivotCalibrationDataWithOutliers.txt:
// 2/3 inliers, 1/3 outliers); algebraic least squares breaks down, the RANSAC-wrapped estimate does
// not.  Without a file argument a synthetic data set of the same structure is generated.
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "PivotCalibrationParametersEstimator.h"
#include "RANSAC.h"
#include "common.h"

int main(int argc, char *argv[]) {
  std::vector<lsqrRecipes::Frame> poses;
  lsqrRecipes::Frame f;
  double truth[6] = {-17.0, 1.0, -157.0, 147.0, -63.0, -1042.0};
  if (argc == 2) {
    std::ifstream in(argv[1]);
    if (!in.is_open()) return EXIT_FAILURE;
    double x, y, z, qx, qy, qz, qs;
    while (in >> x >> y >> z >> qx >> qy >> qz >> qs) {
      f.setRotationQuaternion(qs, qx, qy, qz);
      f.setTranslation(x, y, z);
      poses.push_back(f);
    }
  } else {
    Rng rng(77);
    for (int i = 0; i < 480; i++) {
      // pivoting: random orientation, translation such that R*tip + t = pivot (+ noise / outliers)
      f.setRotationEulerAngles(rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(-3.1, 3.1));
      double R[3][3], t[3];
      f.getRotationMatrix(R);
      for (int a = 0; a < 3; a++) {
        t[a] = truth[3 + a] - (R[a][0] * truth[0] + R[a][1] * truth[1] + R[a][2] * truth[2]) +
               rng.normal(0.15);
        if (i % 3 == 2) t[a] += rng.uniform(-40.0, 40.0);  // a third of the poses are outliers
      }
      f.setTranslation(t);
      poses.push_back(f);
    }
  }
  if (poses.empty()) return EXIT_FAILURE;
  std::vector<double> ls, robust;
  lsqrRecipes::PivotCalibrationEstimator pivot(1.0);  // at most 1 mm between the two points
  pivot.leastSquaresEstimate(poses, ls);
  if (ls.empty()) return EXIT_FAILURE;
  printVec("Least squares translations [DRF^t, W^t]", ls);
  double used = lsqrRecipes::RANSAC<lsqrRecipes::Frame, double>::compute(robust, &pivot, poses, 0.999);
  if (robust.empty()) return EXIT_FAILURE;
  printVec("RANSAC translations [DRF^t, W^t]", robust);
  std::cout << "\tPercentage of poses used for the final estimate: " << used * 100 << "\n";
  double worst = 0;
  for (int i = 0; i < 6; i++) worst = std::max(worst, std::fabs(robust[i] - truth[i]));
  std::cout << "\tLargest deviation from the expected translations: " << worst << "\n";
  return worst < 2.0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
