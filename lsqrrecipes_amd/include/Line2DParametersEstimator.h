// Line2DParametersEstimator.h -- drop-in for parametersEstimators/Line2DParametersEstimator.{h,cxx}:
// 2-D line in normal form [n_x, n_y, a_x, a_y].  Same constructor and virtuals; every method runs on the
// device (LSQR_MODEL_LINE2D): estimate() .cxx:9-27, closed-form least squares .cxx:44-100, agree()
// .cxx:117-121 (the scan is the 2-D hyperplane's, including the two-level cell scan for large uploads).
#ifndef _LINE2D_PARAMETERS_ESTIMATOR_H_
#define _LINE2D_PARAMETERS_ESTIMATOR_H_

#include "LsqrDevice.h"
#include "ParametersEstimator.h"
#include "Point2D.h"

namespace lsqrRecipes {

class Line2DParametersEstimator : public ParametersEstimator<Point2D, double> {
 public:
  Line2DParametersEstimator(double delta) : ParametersEstimator<Point2D, double>(2), delta(delta) {}

  virtual void estimate(std::vector<Point2D *> &data, std::vector<double> &parameters) {
    std::vector<Point2D> tmp;
    detail::gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<Point2D> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::exactFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<Point2D *> &data, std::vector<double> &parameters) {
    std::vector<Point2D> tmp;
    detail::gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<Point2D> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::lsFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<double> &parameters, Point2D &data) {
    return detail::agreeOne(cfg(), parameters, data);
  }
  void setDelta(double d) { this->delta = d; }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg();
    return true;
  }

 private:
  lsqr_model_cfg cfg() const {
    lsqr_model_cfg c = {LSQR_MODEL_LINE2D, 2, delta, 0, 0, 0.0};
    return c;
  }
  double delta;
};

}  // namespace lsqrRecipes
#endif
