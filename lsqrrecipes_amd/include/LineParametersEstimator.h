// LineParametersEstimator.h -- drop-in for parametersEstimators/LineParametersEstimator.{h,hxx}:
// line [direction, a] through a.  Same constructor / setDelta / virtuals; every method
// runs on the device through the C ABI.  Device models exist for dimensions 2 to 8.
#ifndef _LINE_PARAMETERS_ESTIMATOR_H_
#define _LINE_PARAMETERS_ESTIMATOR_H_

#include "LsqrDevice.h"
#include "ParametersEstimator.h"
#include "Point.h"

namespace lsqrRecipes {

template <unsigned int dimension>
class LineParametersEstimator : public ParametersEstimator<Point<double, dimension>, double> {
  typedef Point<double, dimension> PointT;

 public:
  LineParametersEstimator(double delta)
      : ParametersEstimator<PointT, double>(2), delta(delta) {}

  virtual void estimate(std::vector<PointT *> &data, std::vector<double> &parameters) {
    std::vector<PointT> tmp;
    detail::gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<PointT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (this->minForEstimate == 0 || data.size() < this->minForEstimate) return;
    detail::exactFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<PointT *> &data, std::vector<double> &parameters) {
    std::vector<PointT> tmp;
    detail::gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<PointT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::lsFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<double> &parameters, PointT &data) {
    return detail::agreeOne(cfg(), parameters, data);
  }
  void setDelta(double d) { this->delta = d; }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg();
    return dimension >= 2 && dimension <= 8;  // device models: 2, 3 (models.h) and 4..8 (models_nd.h)
  }

 private:
  lsqr_model_cfg cfg() const {
    lsqr_model_cfg c = {LSQR_MODEL_LINE, (int32_t)dimension, delta, 0, 0};
    return c;
  }
  double delta;  // the reference stores delta*delta; the device layer squares it the same way
};

}  // namespace lsqrRecipes
#endif
