// RayIntersectionParametersEstimator.h -- drop-in for
// parametersEstimators/RayIntersectionParametersEstimator.{h,cxx}: the point [x, y, z] where a set of
// rays (approximately) meet.  Same constructor (delta, minimalAngularDeviation = 1 degree) and
// virtuals; every method runs on the device through the C ABI (LSQR_MODEL_RAY):
//   estimate()              mid-point of the common perpendicular of two rays   (.cxx:23-72)
//   leastSquaresEstimate()  [N I - sum n n^T] x = sum (p - (n.p) n)             (.cxx:95-143)
//   agree()                 closest point on the ray (t >= 0) within delta       (.cxx:163-177)
#ifndef _RAY_INTERSECTION_PARAMETERS_ESTIMATOR_H_
#define _RAY_INTERSECTION_PARAMETERS_ESTIMATOR_H_

#include "LsqrDevice.h"
#include "ParametersEstimator.h"
#include "Ray3D.h"

namespace lsqrRecipes {

class RayIntersectionParametersEstimator : public ParametersEstimator<Ray3D, double> {
  static_assert(sizeof(Ray3D) == 6 * sizeof(double), "Ray3D must be 6 doubles (common/Ray3D.h:23-24)");

 public:
  RayIntersectionParametersEstimator(double delta,
                                     double minimalAngularDeviation = 0.017453292519943295769236907684886)
      : ParametersEstimator<Ray3D, double>(2), delta(delta), minAngle(minimalAngularDeviation) {}

  virtual void estimate(std::vector<Ray3D *> &data, std::vector<double> &parameters) {
    std::vector<Ray3D> tmp;
    detail::gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<Ray3D> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::exactFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<Ray3D *> &data, std::vector<double> &parameters) {
    std::vector<Ray3D> tmp;
    detail::gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<Ray3D> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.empty()) return;
    detail::lsFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<double> &parameters, Ray3D &data) {
    return detail::agreeOne(cfg(), parameters, data);
  }
  void setDelta(double d) { this->delta = d; }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg();
    return true;
  }

 private:
  lsqr_model_cfg cfg() const {
    lsqr_model_cfg c = {LSQR_MODEL_RAY, 3, delta, 0, 0, minAngle};
    return c;
  }
  double delta, minAngle;
};

}  // namespace lsqrRecipes
#endif
