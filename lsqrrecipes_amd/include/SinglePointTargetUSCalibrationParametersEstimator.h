// SinglePointTargetUSCalibrationParametersEstimator.h -- drop-in for the reference's header of
// the same name: the cross-wire (single unknown point target) and calibrated-pointer ultrasound
// calibration estimators.  Data types, constructors (delta, LeastSquaresType = ITERATIVE),
// parameter vector layouts (20 / 17 entries) and public helpers as in the reference.
#ifndef _SINGLE_POINT_TARGET_US_CALIBRATION_PARAMETERS_ESTIMATOR_H_
#define _SINGLE_POINT_TARGET_US_CALIBRATION_PARAMETERS_ESTIMATOR_H_

#include <exception>

#include "Frame.h"
#include "LsqrDevice.h"
#include "ParametersEstimator.h"
#include "Point2D.h"
#include "Point3D.h"

namespace lsqrRecipes {

struct SingleUnknownPointTargetUSCalibrationParametersEstimatorDataType {
  Frame T2;   // US reference frame -> tracker
  Point2D q;  // pixel coordinates of the target
};

struct CalibratedPointerTargetUSCalibrationParametersEstimatorDataType {
  Frame T2;
  Point2D q;
  Point3D p;  // pointer tip in the tracker frame
};

static_assert(sizeof(SingleUnknownPointTargetUSCalibrationParametersEstimatorDataType) == 120,
              "record layout must match the reference (Frame 104 B + Point2D)");
static_assert(sizeof(CalibratedPointerTargetUSCalibrationParametersEstimatorDataType) == 144,
              "record layout must match the reference (Frame 104 B + Point2D + Point3D)");

namespace detail {

template <class DataT, int MODEL, unsigned int MIN>
class USEstimatorBase : public ParametersEstimator<DataT, double> {
 public:
  typedef DataT DataType;
  enum LeastSquaresType { ANALYTIC = 0, ITERATIVE };

  USEstimatorBase(double delta, LeastSquaresType lsType)
      : ParametersEstimator<DataT, double>(MIN), delta(delta), lsType(lsType) {}

  // requires exactly minForEstimate elements (reference .cxx:21 / :675)
  virtual void estimate(std::vector<DataT *> &data, std::vector<double> &parameters) {
    std::vector<DataT> tmp;
    gather(data, tmp);
    estimate(tmp, parameters);
  }
  virtual void estimate(std::vector<DataT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() != this->minForEstimate) return;
    exactFit(cfg(lsType), &data[0], data.size(), parameters);
  }
  virtual void leastSquaresEstimate(std::vector<DataT *> &data, std::vector<double> &parameters) {
    std::vector<DataT> tmp;
    gather(data, tmp);
    leastSquaresEstimate(tmp, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<DataT> &data, std::vector<double> &parameters) {
    parameters.clear();
    if (data.size() < this->minForEstimate) return;
    lsFit(cfg(lsType), &data[0], data.size(), parameters);
  }
  virtual bool agree(std::vector<double> &parameters, DataT &data) {
    return agreeOne(cfg(lsType), parameters, data);
  }

  void setDelta(double d) { this->delta = d; }
  void setLeastSquaresType(LeastSquaresType t) { this->lsType = t; }

  void analyticLeastSquaresEstimate(std::vector<DataT *> &data, std::vector<double> &parameters) {
    std::vector<DataT> tmp;
    gather(data, tmp);
    parameters.clear();
    if (tmp.size() < this->minForEstimate) return;
    lsFit(cfg(ANALYTIC), &tmp[0], tmp.size(), parameters);
  }
  void iterativeLeastSquaresEstimate(std::vector<DataT *> &data,
                                     std::vector<double> &initialParameters,
                                     std::vector<double> &finalParameters) {
    std::vector<DataT> tmp;
    gather(data, tmp);
    finalParameters.clear();
    if (tmp.empty()) return;
    lmFit(cfg(ITERATIVE), &tmp[0], tmp.size(), initialParameters, finalParameters);
  }
  // min / max / mean distance between the mapped target and its expected location
  static void getDistanceStatistics(const std::vector<double> &parameters,
                                    const std::vector<DataT> &data, double &min, double &max,
                                    double &mean) {
    lsqr_model_cfg c = {MODEL, 0, 1.0, ITERATIVE, 0};
    if ((int)parameters.size() < lsqr_num_params(&c)) throw std::exception();
    distanceStats(c, parameters, &data[0], data.size(), min, max, mean);
  }

  virtual bool deviceModel(lsqr_model_cfg &c) const {
    c = cfg(lsType);
    return true;
  }

 private:
  lsqr_model_cfg cfg(int ls) const {
    lsqr_model_cfg c = {MODEL, 0, delta, ls, 0};
    return c;
  }
  double delta;
  LeastSquaresType lsType;
};

}  // namespace detail

class SingleUnknownPointTargetUSCalibrationParametersEstimator
    : public detail::USEstimatorBase<SingleUnknownPointTargetUSCalibrationParametersEstimatorDataType,
                                     LSQR_MODEL_US_SINGLE, 4> {
  typedef detail::USEstimatorBase<SingleUnknownPointTargetUSCalibrationParametersEstimatorDataType,
                                  LSQR_MODEL_US_SINGLE, 4> Base;

 public:
  SingleUnknownPointTargetUSCalibrationParametersEstimator(double delta,
                                                           LeastSquaresType lsType = ITERATIVE)
      : Base(delta, lsType) {}
};

class CalibratedPointerTargetUSCalibrationParametersEstimator
    : public detail::USEstimatorBase<CalibratedPointerTargetUSCalibrationParametersEstimatorDataType,
                                     LSQR_MODEL_US_POINTER, 3> {
  typedef detail::USEstimatorBase<CalibratedPointerTargetUSCalibrationParametersEstimatorDataType,
                                  LSQR_MODEL_US_POINTER, 3> Base;

 public:
  CalibratedPointerTargetUSCalibrationParametersEstimator(double delta,
                                                          LeastSquaresType lsType = ITERATIVE)
      : Base(delta, lsType) {}
};

}  // namespace lsqrRecipes
#endif
