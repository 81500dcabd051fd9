// linearEquationSystemSolver -- counterpart of the reference's
// examples/linearEquationSystemSolver.cxx (:42-220): (a) simulated 200 x 5 system with 5 % of the
// right-hand sides scaled by 20, (b) optionally a whitespace-separated augmented matrix file with 7
// columns (examples/Data/augmentedMatrixWithOutliers.txt).  usage: linearEquationSystemSolver [file]
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "DenseLinearEquationSystemParametersEstimator.h"
#include "RANSAC.h"
#include "common.h"

int main(int argc, char *argv[]) {
  const unsigned int N = 5;
  typedef lsqrRecipes::AugmentedRow<double, N> Row;
  Rng rng(3);
  std::vector<double> x(N), params;
  for (unsigned i = 0; i < N; i++) x[i] = rng.uniform(-1, 1);
  std::vector<Row> rows;
  for (unsigned i = 0; i < 200; i++) {
    double a[N + 1], b = 0;
    for (unsigned j = 0; j < N; j++) {
      a[j] = rng.uniform(-1, 1);
      b += a[j] * x[j];
    }
    a[N] = b * (1.0 + rng.uniform(-0.01, 0.01));
    if (i % 20 == 0) a[N] *= 20.0;  // outlier equations
    rows.push_back(Row(a));
  }
  printVec("Known solution [x_0,...,x_{n-1}]", x);
  lsqrRecipes::DenseLinearEquationSystemParametersEstimator<double, N> solver(0.2);
  solver.leastSquaresEstimate(rows, params);
  printVec("Least squares solution (outliers included)", params);
  double used = lsqrRecipes::RANSAC<Row, double>::compute(params, &solver, rows, 0.999);
  if (params.empty()) return EXIT_FAILURE;
  printVec("RANSAC solution", params);
  std::cout << "\tPercentage of equations used for final estimate: " << used << "\n\n";
  double err = 0;
  for (unsigned i = 0; i < N; i++) err = std::max(err, std::fabs(params[i] - x[i]));
  int rc = err < 0.05 ? EXIT_SUCCESS : EXIT_FAILURE;

  if (argc > 1) {  // experimental data: rows of 6 coefficients + right-hand side
    const unsigned int M = 6;
    typedef lsqrRecipes::AugmentedRow<double, M> Row6;
    std::ifstream in(argv[1]);
    std::vector<Row6> rows6;
    double v[M + 1];
    while (in >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5] >> v[6]) rows6.push_back(Row6(v));
    if (rows6.empty()) {
      std::cerr << "Failed to load augmented matrix file.\n";
      return EXIT_FAILURE;
    }
    lsqrRecipes::DenseLinearEquationSystemParametersEstimator<double, M> solver6(std::sqrt(1.0 / 3.0));
    std::vector<double> p6;
    solver6.leastSquaresEstimate(rows6, p6);
    printVec("Experimental data, least squares solution", p6);
    used = lsqrRecipes::RANSAC<Row6, double>::compute(p6, &solver6, rows6, 0.999);
    printVec("Experimental data, RANSAC solution (approximately -17, 1, -157, 147, -63, -1042)", p6);
    std::cout << "\tPercentage of equations used for final estimate: " << used << "\n";
    if (p6.size() != M || std::fabs(p6[5] + 1042) > 2.0) rc = EXIT_FAILURE;
  }
  return rc;
}
