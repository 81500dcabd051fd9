// PivotCalibrationParametersEstimator.h -- drop-in for
// parametersEstimators/PivotCalibrationParametersEstimator.{h,cxx}: tool tip [DRF^t] and pivot
// point [W^t] from tracked poses, R_i DRF^t + t_i = W^t.  Same class name (PivotCalibrationEstimator),
// constructor, setDelta and virtuals; every method runs on the device (LSQR_MODEL_PIVOT):
//   estimate()              9x6 pseudo-inverse of three poses         (.cxx:9-50)
//   leastSquaresEstimate()  6x6 normal equations of all poses         (.cxx:63-96)
//   agree()                 ||R DRF^t + t - W^t|| < delta             (.cxx:109-123)
#ifndef _PIVOT_CALIBRATION_PARAMETERS_ESTIMATOR_H_
#define _PIVOT_CALIBRATION_PARAMETERS_ESTIMATOR_H_

#include "Frame.h"
#include "LsqrDevice.h"
#include "ParametersEstimator.h"

namespace lsqrRecipes {

class PivotCalibrationEstimator : public ParametersEstimator<Frame, double> {
  static_assert(sizeof(Frame) == 13 * sizeof(double), "Frame must be 104 bytes (common/Frame.h:30-31,41)");

 public:
  PivotCalibrationEstimator(double delta) : ParametersEstimator<Frame, double>(3), delta(delta) {}

  virtual void estimate(std::vector<Frame *> &data, std::vector<double> &parameters) {
    std::vector<Frame> tmp;
    detail::gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<Frame> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::exactFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<Frame *> &data, std::vector<double> &parameters) {
    std::vector<Frame> tmp;
    detail::gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<Frame> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    detail::lsFit(cfg(), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<double> &parameters, Frame &data) {
    return detail::agreeOne(cfg(), parameters, data);
  }
  void setDelta(double d) { this->delta = d; }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg();
    return true;
  }

 private:
  lsqr_model_cfg cfg() const {
    lsqr_model_cfg c = {LSQR_MODEL_PIVOT, 3, delta, 0, 0};
    return c;
  }
  double delta;
};

}  // namespace lsqrRecipes
#endif
