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
  // AbsoluteOrientationParametersEstimator.h:86 / .cxx:208-291: Horn's closed form with per-pair weights
  // (`weights` holds at least data.size() non-negative entries).  The device reduces the weighted sums
  // {sum w, sum w l, sum w r, sum w l r^T} of records [first, second, weight].
  void weightedLeastSquaresEstimate(std::vector<DataT *> &data, std::vector<double> &weights,
                                    std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    if (weights.size() < data.size()) throw std::out_of_range("lsqrRecipes: fewer weights than point pairs");
    std::vector<double> rec(7 * data.size());
    for (size_t i = 0; i < data.size(); i++) {
      for (int j = 0; j < 3; j++) {
        rec[7 * i + j] = data[i]->first[j];
        rec[7 * i + 3 + j] = data[i]->second[j];
      }
      rec[7 * i + 6] = weights[i];
    }
    lsqr_model_cfg c = cfg();
    c.ls_type = 2;  // records [first, second, weight]
    detail::lsFitRaw(c, &rec[0], data.size(), 7 * sizeof(double), parameters);
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
