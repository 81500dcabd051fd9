// planeUSCalibration -- counterpart of the reference's examples/planeUSCalibration.cxx (:30-85):
// ultrasound calibration with a planar phantom, robustly with RANSAC (k = 31 frames per hypothesis,
// delta 2 mm, p = 0.999) followed by the iterative least squares fit on the consensus set.
// usage: planeUSCalibration [transformationsFile pointsFile [outputXMLFile]]
//        (without arguments: simulated frames).  With an output file name the calibration T3 is
//        written as an IGSTK "precomputed_transform" XML document, the wire format of the
//        reference's example (:181-222).
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>

#include "PlanePhantomUSCalibrationParametersEstimator.h"
#include "RANSAC.h"
#include "common.h"

typedef lsqrRecipes::PlanePhantomUSCalibrationParametersEstimator Estimator;
typedef Estimator::DataType DataType;

static bool load(const char *tf, const char *pf, std::vector<DataType> &data) {
  std::ifstream t(tf), p(pf);
  if (!t.is_open() || !p.is_open()) return false;
  double R[3][3], tr[3], u, v;
  for (;;) {
    bool row = true;
    for (int i = 0; i < 3 && row; i++) row = bool(t >> R[i][0] >> R[i][1] >> R[i][2] >> tr[i]);
    if (!row || !(p >> u >> v)) break;
    DataType d;
    d.T2.setRotationMatrix(R);
    d.T2.setTranslation(tr);
    d.q[0] = u;
    d.q[1] = v;
    data.push_back(d);
  }
  return !data.empty();
}

// frames whose image point, mapped through T2 T3, lies on the phantom plane z = 0 of T1; every
// eighth frame is off the plane by 2-10 cm
static void simulate(std::vector<DataType> &data) {
  Rng rng(11);
  const double mx = 0.143, my = 0.139, PI = 3.14159265358979323846;
  lsqrRecipes::Frame T3, T1;
  T3.setRotationEulerAngles(rng.uniform(0, PI), rng.uniform(0, PI), rng.uniform(0, PI));
  T3.setTranslation(rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-100, 100));
  T1.setRotationEulerAngles(rng.uniform(0, PI), rng.uniform(0, PI), rng.uniform(0, PI));
  double R1[3][3], t1[3] = {rng.uniform(-100, 100), rng.uniform(-100, 100), rng.uniform(-100, 100)};
  T1.getRotationMatrix(R1);
  for (int i = 0; i < 400; i++) {
    DataType d;
    const double u = rng.uniform(0, 640), v = rng.uniform(0, 480);
    double onPlane[3] = {rng.uniform(-100, 100), rng.uniform(-100, 100), 0.0};
    if (i % 8 == 7) onPlane[2] = (rng.uniform() < 0.5 ? -1 : 1) * rng.uniform(20, 100);
    double inTracker[3], q[3] = {mx * u, my * v, 0}, q3[3], rq[3];
    for (int r = 0; r < 3; r++) {  // T1^-1
      inTracker[r] = 0;
      for (int c = 0; c < 3; c++) inTracker[r] += R1[c][r] * (onPlane[c] - t1[c]);
    }
    d.T2.setRotationEulerAngles(rng.uniform(0, PI), rng.uniform(0, PI), rng.uniform(0, PI));
    d.T2.setTranslation(0, 0, 0);
    T3.apply(q, q3);
    d.T2.apply(q3, rq);
    d.T2.setTranslation(inTracker[0] - rq[0], inTracker[1] - rq[1], inTracker[2] - rq[2]);
    d.q[0] = u;
    d.q[1] = v;
    data.push_back(d);
  }
}

// T3 from the minimal parameters [.., t3 (3..5), omega3_z, omega3_y, omega3_x, m_x, m_y]
static bool writeIgstkXml(const char *fileName, const std::vector<double> &p, double meanError) {
  std::ofstream out(fileName);
  if (!out.is_open()) return false;
  char stamp[64];
  std::time_t now = std::time(0);
  std::strftime(stamp, sizeof stamp, "%Y %b %d %H:%M:%S", std::localtime(&now));
  const double cz = std::cos(p[6]), sz = std::sin(p[6]), cy = std::cos(p[7]), sy = std::sin(p[7]),
               cx = std::cos(p[8]), sx = std::sin(p[8]), mx = p[9], my = p[10];
  const double T[3][4] = {{mx * cz * cy, my * (cz * sy * sx - sz * cx), cz * sy * cx + sz * sx, p[3]},
                          {mx * sz * cy, my * (sz * sy * sx + cz * cx), sz * sy * cx - cz * sx, p[4]},
                          {-mx * sy, my * cy * sx, cy * cx, p[5]}};
  out << std::fixed << std::setprecision(10);
  out << "<?xml version=\"1.0\" encoding=\"ISO-8859-1\"?>\n\n";
  out << "<precomputed_transform>\n\n";
  out << "\t<description>\n\tUS calibration - Plane Phantom\n\t</description>\n\n";
  out << "\t<computation_date>\n\t" << stamp << "\n\t</computation_date>\n\n";
  out << "\t<transformation estimation_error=\"" << meanError << "\">\n";
  for (int r = 0; r < 3; r++)
    out << "\t" << T[r][0] << "\t" << T[r][1] << "\t" << T[r][2] << "\t" << T[r][3] << "\n";
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
  Estimator estimator(2.0);
  std::vector<double> params, errors;
  std::vector<bool> consensus;
  double used = lsqrRecipes::RANSAC<DataType, double>::compute(params, &estimator, data, 0.999, &consensus);
  if (params.empty()) {
    std::cout << "RANSAC calibration failed\n";
    return EXIT_FAILURE;
  }
  printVec("RANSAC + iterative least squares [w1y, w1x, t1z, t3, w3z, w3y, w3x, mx, my, ...]", params);
  std::vector<DataType> inl;
  for (size_t i = 0; i < data.size(); i++)
    if (consensus[i]) inl.push_back(data[i]);
  double mn, mx_, mean, sse = 0;
  Estimator::getDistanceStatistics(params, inl, errors, mn, mx_, mean);
  for (size_t i = 0; i < errors.size(); i++) sse += errors[i] * errors[i];
  std::cout << "\tPercentage of data used in estimate: " << used << "\n";
  std::cout << "\tsum of squared errors: " << sse << "\n";
  std::cout << "\tmax, min, mean error: " << mx_ << ", " << mn << ", " << mean << "\n";
  if (argc == 4 && !writeIgstkXml(argv[3], params, mean)) {
    std::cerr << "Failed to write " << argv[3] << "\n";
    return EXIT_FAILURE;
  }
  return (used > 0.5 && mx_ < 2.0) ? EXIT_SUCCESS : EXIT_FAILURE;
}
