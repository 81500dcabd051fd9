// ParametersEstimator.h -- the reference's plugin interface
// (parametersEstimators/ParametersEstimator.h:26-64), unchanged in names, argument meaning and the
// "empty parameters vector == failure" convention, plus ONE addition: deviceModel(), through which
// an estimator tells RANSAC<T,S>::compute() which device model evaluates it.  Every estimator this
// library ships implements it and runs on the MI355X only.  A user-defined subclass keeps the default
// (false): RANSAC<T,S>::compute() then drives its virtuals with the reference's serial loop (RANSAC.h,
// "plugin path"), as the reference's readme.txt:40-72 promises for user estimators.
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

  // false: no device implementation (RANSAC::compute drives the virtuals on the host: plugin path)
  virtual bool deviceModel(lsqr_model_cfg &) const { return false; }

 protected:
  unsigned int minForEstimate;
};

}  // namespace lsqrRecipes
#endif
