// pointerUSCalibration -- counterpart of the reference's examples/pointerUSCalibration.cxx (:30-112): US-probe
// transformations (rows "R00 R01 R02 t0" x 3 per frame), the 2D image coordinates and the 3D tracked-pointer
// coordinates of the target -> calibration with the CalibratedPointerTarget estimator inside RANSAC (threshold 2 mm,
// p = 0.999, as the reference's main), result on the console and, with an output file name, as an IGSTK
// "precomputed_transform" XML document (the reference's wire format, :181-210 of the cross-wire example).
// usage: pointerUSCalibration [transformationsFile 2DPointsFile 3DPointsFile [outputXMLFile]]
//        (without arguments: simulated data)
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "RANSAC.h"
#include "SinglePointTargetUSCalibrationParametersEstimator.h"
#include "common.h"

typedef lsqrRecipes::CalibratedPointerTargetUSCalibrationParametersEstimator Estimator;
typedef Estimator::DataType DataType;

static bool load(const char *tf, const char *qf, const char *pf, std::vector<DataType> &data) {
  std::ifstream t(tf), q(qf), p(pf);
  if (!t.is_open() || !q.is_open() || !p.is_open()) return false;
  double R[3][3], tr[3], u, v, x, y, z;
  while ((q >> u >> v) && (p >> x >> y >> z)) {
    for (int i = 0; i < 3; i++)
      if (!(t >> R[i][0] >> R[i][1] >> R[i][2] >> tr[i])) return !data.empty();
    DataType d;
    d.T2.setRotationMatrix(R);
    d.T2.setTranslation(tr);
    d.q[0] = u, d.q[1] = v;
    d.p[0] = x, d.p[1] = y, d.p[2] = z;
    data.push_back(d);
  }
  return !data.empty();
}

// frames of a probe that images a tracked pointer tip: p_i = T2_i T3 [m_x u, m_y v, 0, 1]
static void simulate(std::vector<DataType> &data) {
  Rng rng(11);
  const double mx = 0.143, my = 0.139, PI = 3.14159265358979323846;
  lsqrRecipes::Frame T3;
  T3.setRotationEulerAngles(rng.uniform(0, PI), rng.uniform(0, PI), rng.uniform(0, PI));
  T3.setTranslation(rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-100, 100));
  for (int i = 0; i < 60; i++) {
    DataType d;
    const double u = rng.uniform(0, 640), v = rng.uniform(0, 480);
    d.T2.setRotationEulerAngles(rng.uniform(0, PI), rng.uniform(0, PI), rng.uniform(0, PI));
    d.T2.setTranslation(rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-100, 100));
    double q[3] = {mx * u, my * v, 0}, q3[3], tip[3];
    T3.apply(q, q3);
    d.T2.apply(q3, tip);
    for (int j = 0; j < 3; j++) d.p[j] = tip[j] + rng.normal(0.2);
    if (i % 6 == 5)  // outlier: the pointer was somewhere else
      for (int j = 0; j < 3; j++) d.p[j] = rng.uniform(-150, 150);
    d.q[0] = u + rng.normal(0.5);
    d.q[1] = v + rng.normal(0.5);
    data.push_back(d);
  }
}

// T3 = [m_x r1, m_y r2, r3 | t3] from the 17-vector [t3, wz, wy, wx, m_x, m_y, m_x r1, m_y r2, r3]
static bool writeIgstkXml(const char *fileName, const std::vector<double> &par, double meanError) {
  std::ofstream out(fileName);
  if (!out.is_open()) return false;
  char stamp[64];
  std::time_t now = std::time(0);
  std::strftime(stamp, sizeof stamp, "%Y %b %d %H:%M:%S", std::localtime(&now));
  out << std::fixed << std::setprecision(10);
  out << "<?xml version=\"1.0\" encoding=\"ISO-8859-1\"?>\n\n";
  out << "<precomputed_transform>\n\n";
  out << "\t<description>\n\tUS calibration - Calibrated Pointer\n\t</description>\n\n";
  out << "\t<computation_date>\n\t" << stamp << "\n\t</computation_date>\n\n";
  out << "\t<transformation estimation_error=\"" << meanError << "\">\n";
  for (int row = 0; row < 3; row++)
    out << "\t" << par[8 + row] << "\t" << par[11 + row] << "\t" << par[14 + row] << "\t" << par[row] << "\n";
  out << "\t</transformation>\n\n</precomputed_transform>\n";
  return out.good();
}

int main(int argc, char *argv[]) {
  std::vector<DataType> data;
  if (argc == 4 || argc == 5) {
    if (!load(argv[1], argv[2], argv[3], data)) {
      std::cerr << "Failed to load data files.\n";
      return EXIT_FAILURE;
    }
  } else if (argc == 1) {
    simulate(data);
  } else {
    std::cerr << "Usage: \n\t" << argv[0] << " transformationsFileName 2DPointsFileName 3DPointsFileName [outputFileName]\n";
    return EXIT_FAILURE;
  }
  std::cout << data.size() << " frames\n";
  const double maxDistanceBetweenPoints = 2.0;
  Estimator usCalibration(maxDistanceBetweenPoints);
  std::vector<double> params;
  std::vector<bool> consensus;
  double used = lsqrRecipes::RANSAC<DataType, double>::compute(params, &usCalibration, data, 0.999, &consensus);
  if (params.empty()) {
    std::cout << "FAILED CALIBRATION, possibly degenerate configuration\n\n\n";
    return EXIT_FAILURE;
  }
  printVec("RANSAC calibration [t3, wz, wy, wx, mx, my, mx r1, my r2, r3]", params);
  std::vector<DataType> inl;
  for (size_t i = 0; i < data.size(); i++)
    if (consensus[i]) inl.push_back(data[i]);
  double mn, mx_, mean;
  Estimator::getDistanceStatistics(params, inl, mn, mx_, mean);
  std::cout << "\tPercentage of frames used: " << used << "\n";
  std::cout << "\tdistance to the pointer tip over the consensus set: min " << mn << " max " << mx_ << " mean " << mean
            << "\n";
  if (argc == 5 && !writeIgstkXml(argv[4], params, mean)) {
    std::cerr << "Failed to write " << argv[4] << "\n";
    return EXIT_FAILURE;
  }
  return (used > 0.5 && mx_ < maxDistanceBetweenPoints) ? EXIT_SUCCESS : EXIT_FAILURE;
}
