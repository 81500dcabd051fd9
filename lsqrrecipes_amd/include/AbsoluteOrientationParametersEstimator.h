// AbsoluteOrientationParametersEstimator.h -- drop-in for
// parametersEstimators/AbsoluteOrientationParametersEstimator.{h,cxx}: rigid transformation
// [s, qx, qy, qz, tx, ty, tz] with data[i].second = T * data[i].first.  Same constructor and
// virtuals; every method runs on the device through the C ABI (LSQR_MODEL_ABSOR):
//   estimate()              triads of the first three pairs           (.cxx:14-105)
//   leastSquaresEstimate()  Horn's closed form on device-reduced sums (.cxx:123-198)
//   agree()                 ||second - T first||^2 < delta^2         (.cxx:316-327)
// The quaternion of a least squares estimate is defined up to sign (eigenvector of the 4x4 N).
#ifndef _ABSOLUTE_ORIENTATION_PARAMETERS_ESTIMATOR_H_
#define _ABSOLUTE_ORIENTATION_PARAMETERS_ESTIMATOR_H_

#include <utility>

#include "LsqrDevice.h"
#include "ParametersEstimator.h"
#include "Point3D.h"

namespace lsqrRecipes {

class AbsoluteOrientationParametersEstimator
    : public ParametersEstimator<std::pair<Point3D, Point3D>, double> {
  typedef std::pair<Point3D, Point3D> DataT;
  static_assert(sizeof(DataT) == 6 * sizeof(double), "pair<Point3D,Point3D> must be 6 doubles");

 public:
  AbsoluteOrientationParametersEstimator(double delta)
      : ParametersEstimator<DataT, double>(3), delta(delta) {}

  virtual void estimate(std::vector<DataT *> &data, std::vector<double> &parameters) {
    std::vector<DataT> tmp;
    detail::gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<DataT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::exactFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<DataT *> &data, std::vector<double> &parameters) {
    std::vector<DataT> tmp;
    detail::gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<DataT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::lsFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<double> &parameters, DataT &data) {
    return detail::agreeOne(cfg(), parameters, data);
  }
  void setDelta(double d) { this->delta = d; }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg();
    return true;
  }

 private:
  lsqr_model_cfg cfg() const {
    lsqr_model_cfg c = {LSQR_MODEL_ABSOR, 3, delta, 0, 0};
    return c;
  }
  double delta;
};

}  // namespace lsqrRecipes
#endif
