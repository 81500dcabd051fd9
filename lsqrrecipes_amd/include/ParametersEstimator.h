// ParametersEstimator.h -- the reference's plugin interface
// (parametersEstimators/ParametersEstimator.h:26-64), unchanged in names, argument meaning and the
// "empty parameters vector == failure" convention, plus ONE addition: deviceModel(), through which
// an estimator tells RANSAC<T,S>::compute() which device model evaluates it.  The five hot-path
// estimators of this drop-in implement it; the hot path has no CPU fallback, so handing an
// estimator without a device model to RANSAC::compute() throws (see RANSAC.h).
#ifndef _PARAMETERS_ESTIMATOR_H_
#define _PARAMETERS_ESTIMATOR_H_

#include <vector>

#include "lsqr_hip.h"

namespace lsqrRecipes {

template <class T, class S>
class ParametersEstimator {
 public:
  ParametersEstimator(unsigned int minElements) : minForEstimate(minElements) {}
  virtual ~ParametersEstimator() {}

  virtual void estimate(std::vector<T *> &data, std::vector<S> &parameters) = 0;
  virtual void estimate(std::vector<T> &data, std::vector<S> &parameters) = 0;
  virtual void leastSquaresEstimate(std::vector<T *> &data, std::vector<S> &parameters) = 0;
  virtual void leastSquaresEstimate(std::vector<T> &data, std::vector<S> &parameters) = 0;
  virtual bool agree(std::vector<S> &parameters, T &data) = 0;

  unsigned int numForEstimate() { return this->minForEstimate; }

  // false: no device implementation (RANSAC::compute will refuse it)
  virtual bool deviceModel(lsqr_model_cfg &) const { return false; }

 protected:
  unsigned int minForEstimate;
};

}  // namespace lsqrRecipes
#endif
