// PlanePhantomUSCalibrationParametersEstimator.h -- drop-in for the reference's header of the same
// name: ultrasound calibration with a planar phantom.  Data type, constructor (delta,
// LeastSquaresType = ITERATIVE), the 41-entry parameter vector [omega1_y, omega1_x, t1_z, t3(3),
// omega3_z, omega3_y, omega3_x, m_x, m_y, 30 derived products] and the public helpers as in the
// reference (parametersEstimators/PlanePhantomUSCalibrationParametersEstimator.h:140-263).
#ifndef _PLANE_PHANTOM_US_CALIBRATION_PARAMETERS_ESTIMATOR_H_
#define _PLANE_PHANTOM_US_CALIBRATION_PARAMETERS_ESTIMATOR_H_

#include <exception>

#include "SinglePointTargetUSCalibrationParametersEstimator.h"

namespace lsqrRecipes {

struct PlanePhantomUSCalibrationParametersEstimatorDataType {
  Frame T2;   // US reference frame -> tracker
  Point2D q;  // pixel on the line the phantom plane makes in the image
};
static_assert(sizeof(PlanePhantomUSCalibrationParametersEstimatorDataType) == 120,
              "record layout must match the reference (Frame 104 B + Point2D)");

class PlanePhantomUSCalibrationParametersEstimator
    : public detail::USEstimatorBase<PlanePhantomUSCalibrationParametersEstimatorDataType,
                                     LSQR_MODEL_PHANTOM, 31> {
  typedef detail::USEstimatorBase<PlanePhantomUSCalibrationParametersEstimatorDataType,
                                  LSQR_MODEL_PHANTOM, 31> Base;

 public:
  PlanePhantomUSCalibrationParametersEstimator(double delta, LeastSquaresType lsType = ITERATIVE)
      : Base(delta, lsType) {}

  // |distance of the mapped pixel from the phantom plane| per frame and its min / max / mean
  // (reference .cxx:455-549)
  static void getDistanceStatistics(const std::vector<double> &parameters,
                                    const std::vector<DataType> &data,
                                    std::vector<double> &distances, double &min, double &max,
                                    double &mean) {
    lsqr_model_cfg c = {LSQR_MODEL_PHANTOM, 0, 1.0, ITERATIVE, 0};
    if ((int)parameters.size() < lsqr_num_params(&c)) throw std::exception();
    distances.assign(data.size(), 0.0);
    if (data.empty()) return;
    detail::distanceStats(c, parameters, &data[0], data.size(), min, max, mean);
    detail::Device &d = detail::Device::instance();
    d.check(lsqr_residuals(d.ctx(), &parameters[0], 0, data.size(), &distances[0]));
  }
};

}  // namespace lsqrRecipes
#endif
