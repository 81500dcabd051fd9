// SphereParametersEstimator.h -- drop-in for parametersEstimators/SphereParametersEstimator.{h,hxx}:
// (hyper)sphere [c, r].  Same constructor (delta, LeastSquaresType = GEOMETRIC), setters and public
// helpers (algebraic / geometric least squares, getDistanceStatistics); device models for
// dimensions 2 (circle), 3 (sphere) and 4 to 8 (hyperspheres).
#ifndef _SPHERE_PARAMETERS_ESTIMATOR_H_
#define _SPHERE_PARAMETERS_ESTIMATOR_H_

#include <exception>

#include "LsqrDevice.h"
#include "ParametersEstimator.h"
#include "Point.h"

namespace lsqrRecipes {

template <unsigned int dimension>
class SphereParametersEstimator : public ParametersEstimator<Point<double, dimension>, double> {
  typedef Point<double, dimension> PointT;

 public:
  enum LeastSquaresType { ALGEBRAIC = 0, GEOMETRIC };
  enum { CIRCLE = 2, SPHERE = 3 };

  SphereParametersEstimator(double delta, LeastSquaresType lsType = GEOMETRIC)
      : ParametersEstimator<PointT, double>(dimension + 1), delta(delta), lsType(lsType) {
    if (lsType != ALGEBRAIC && lsType != GEOMETRIC) throw std::exception();
  }

  virtual void estimate(std::vector<PointT *> &data, std::vector<double> &parameters) {
    std::vector<PointT> tmp;
    detail::gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<PointT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::exactFit(cfg(lsType), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<PointT *> &data, std::vector<double> &parameters) {
    std::vector<PointT> tmp;
    detail::gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<PointT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::lsFit(cfg(lsType), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<double> &parameters, PointT &data) {
    return detail::agreeOne(cfg(lsType), parameters, data);
  }

  void setDelta(double d) { this->delta = d; }
  void setLeastSquaresType(LeastSquaresType t) {
    if (t != ALGEBRAIC && t != GEOMETRIC) throw std::exception();
    this->lsType = t;
  }

  void algebraicLeastSquaresEstimate(std::vector<PointT *> &data, std::vector<double> &parameters) {
    std::vector<PointT> tmp;
    detail::gather(data, tmp);
    parameters.clear();
    if (tmp.size() < this->minForEstimate) return;
    detail::lsFit(cfg(ALGEBRAIC), &tmp[0], tmp.size(), parameters);
  }
  void geometricLeastSquaresEstimate(std::vector<PointT *> &data,
                                     std::vector<double> &initialParameters,
                                     std::vector<double> &finalParameters) {
    std::vector<PointT> tmp;
    detail::gather(data, tmp);
    finalParameters.clear();
    if (tmp.empty() || initialParameters.size() < dimension + 1) return;
    detail::lmFit(cfg(GEOMETRIC), &tmp[0], tmp.size(), initialParameters, finalParameters);
  }
  // min / max / mean of | ||p - c|| - r |
  static void getDistanceStatistics(std::vector<double> &parameters, std::vector<PointT> &data,
                                    double &min, double &max, double &mean) {
    if (parameters.size() < dimension + 1) throw std::exception();
    lsqr_model_cfg c = {LSQR_MODEL_SPHERE, (int32_t)dimension, 1.0, GEOMETRIC, 0};
    detail::distanceStats(c, parameters, &data[0], data.size(), min, max, mean);
  }

  // the reference's signature (SphereParametersEstimator.h:156-159, .hxx:341-377): the distance of every
  // point is appended to `distances` (push_back: earlier entries stay, as in the reference)
  static void getDistanceStatistics(std::vector<double> &parameters, std::vector<PointT> &data,
                                    std::vector<double> &distances, double &min, double &max,
                                    double &mean) {
    if (parameters.size() < dimension + 1) throw std::exception();
    lsqr_model_cfg c = {LSQR_MODEL_SPHERE, (int32_t)dimension, 1.0, GEOMETRIC, 0};
    detail::distanceStats(c, parameters, &data[0], data.size(), min, max, mean);
    const size_t at = distances.size();
    distances.resize(at + data.size());
    detail::Device &d = detail::Device::instance();
    d.check(lsqr_residuals(d.ctx(), &parameters[0], 0, data.size(), &distances[at]));
  }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg(lsType);
    return dimension >= 2 && dimension <= 8;  // device models: 2, 3 (models.h) and 4..8 (models_nd.h)
  }

 private:
  lsqr_model_cfg cfg(int ls) const {
    lsqr_model_cfg c = {LSQR_MODEL_SPHERE, (int32_t)dimension, delta, ls, 0};
    return c;
  }
  double delta;
  LeastSquaresType lsType;
};

}  // namespace lsqrRecipes
#endif
