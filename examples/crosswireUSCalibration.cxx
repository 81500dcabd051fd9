// crosswireUSCalibration -- counterpart of the reference's examples/crosswireUSCalibration.cxx
// (:30-85): load tracker transformations (rows "R00 R01 R02 t0" x 3 per frame) and the 2D image
// points of the cross-wire, calibrate analytically / iteratively and robustly with RANSAC.
// usage: crosswireUSCalibration [transformationsFile pointsFile [outputXMLFile]]
//        (without arguments: simulated data).  With an output file name the calibration T3 is
//        written as an IGSTK "precomputed_transform" XML document, the wire format of the
//        reference's example (:181-210).
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iomanip>
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

// T3 = [m_x r1, m_y r2, r3 | t3] from the 20-vector [t1, t3, wz, wy, wx, m_x, m_y, m_x r1, m_y r2, r3]
static bool writeIgstkXml(const char *fileName, const std::vector<double> &par, double meanError) {
  std::ofstream out(fileName);
  if (!out.is_open()) return false;
  char stamp[64];
  std::time_t now = std::time(0);
  std::strftime(stamp, sizeof stamp, "%Y %b %d %H:%M:%S", std::localtime(&now));
  out << std::fixed << std::setprecision(10);
  out << "<?xml version=\"1.0\" encoding=\"ISO-8859-1\"?>\n\n";
  out << "<precomputed_transform>\n\n";
  out << "\t<description>\n\tUS calibration - Crosswire Phantom\n\t</description>\n\n";
  out << "\t<computation_date>\n\t" << stamp << "\n\t</computation_date>\n\n";
  out << "\t<transformation estimation_error=\"" << meanError << "\">\n";
  for (int row = 0; row < 3; row++)
    out << "\t" << par[11 + row] << "\t" << par[14 + row] << "\t" << par[17 + row] << "\t"
        << par[3 + row] << "\n";
  out << "\t</transformation>\n\n</precomputed_transform>\n";
  return out.good();
}

int main(int argc, char *argv[]) {
  std::vector<DataType> data;
  if (argc == 3 || argc == 4) {
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
  if (argc == 4 && !writeIgstkXml(argv[3], params, mean)) {
    std::cerr << "Failed to write " << argv[3] << "\n";
    return EXIT_FAILURE;
  }
  return (used > 0.5 && mx_ < 3.0) ? EXIT_SUCCESS : EXIT_FAILURE;
}
