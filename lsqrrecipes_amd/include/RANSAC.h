// RANSAC.h -- drop-in for the reference's parametersEstimators/RANSAC.h(.hxx): the same two static
// compute() overloads, argument meaning, return value (fraction of the data in the winning
// consensus set) and failure conventions, with the work done on an MI355X through the C ABI of
// include/lsqr_hip.h (hypothesis batches -> minimal solves -> agree() scan -> first-max winner ->
// consensus mask -> leastSquaresEstimate, and a host replay of the serial adaptive loop so that the
// result equals the serial algorithm's for the same subset stream).
//
// Differences a maintainer should know about:
//  * the subset stream comes from a seeded counter-based sampler instead of srand(time(NULL))/rand()
//    (RANSAC.hxx:44,59): runs are reproducible; RANSAC<T,S>::seed() sets the stream;
//  * the estimator must expose a device model (ParametersEstimator::deviceModel); there is no CPU
//    path -- an estimator without one makes compute() throw std::runtime_error.
#ifndef _RANSAC_H_
#define _RANSAC_H_

#include <algorithm>
#include <stdexcept>
#include <vector>

#include "LsqrDevice.h"
#include "ParametersEstimator.h"

namespace lsqrRecipes {

template <class T, class S>
class RANSAC {
 public:
  // probabilistic search, reference RANSAC.h:75-79
  static double compute(std::vector<S> &parameters, ParametersEstimator<T, S> *paramEstimator,
                        std::vector<T> &data, double desiredProbabilityForNoOutliers,
                        std::vector<bool> *consensusSet = NULL) {
    lsqr_model_cfg cfg;
    requireDevice(paramEstimator, cfg);
    // RANSAC.hxx:16-19: invalid input returns 0 and leaves `parameters` untouched
    if (data.size() < paramEstimator->numForEstimate() || desiredProbabilityForNoOutliers >= 1.0 ||
        desiredProbabilityForNoOutliers <= 0.0)
      return 0;
    detail::Device &d = detail::Device::instance();
    d.model(cfg);
    d.check(lsqr_upload(d.ctx(), &data[0], data.size(), sizeof(T)));
    std::vector<double> p(64);
    std::vector<uint8_t> cons(consensusSet ? data.size() : 0);
    lsqr_ransac_info info;
    parameters.clear();  // RANSAC.hxx:43
    bool ok = d.check(lsqr_ransac(d.ctx(), desiredProbabilityForNoOutliers, seed(), NULL, 0, &p[0],
                                  consensusSet ? &cons[0] : NULL, &info));
    lastInfo() = info;
    return finish(ok, info, p, cons, parameters, consensusSet);
  }

  // exhaustive search over all subsets, reference RANSAC.h:111-113
  static double compute(std::vector<S> &parameters, ParametersEstimator<T, S> *paramEstimator,
                        std::vector<T> &data, std::vector<bool> *consensusSet = NULL) {
    lsqr_model_cfg cfg;
    requireDevice(paramEstimator, cfg);
    parameters.clear();  // RANSAC.hxx:165 clears before the size check
    if (data.size() < paramEstimator->numForEstimate()) return 0;
    detail::Device &d = detail::Device::instance();
    d.model(cfg);
    d.check(lsqr_upload(d.ctx(), &data[0], data.size(), sizeof(T)));
    std::vector<double> p(64);
    std::vector<uint8_t> cons(consensusSet ? data.size() : 0);
    lsqr_ransac_info info;
    bool ok = d.check(lsqr_ransac_exhaustive(d.ctx(), &p[0], consensusSet ? &cons[0] : NULL, &info));
    lastInfo() = info;
    return finish(ok, info, p, cons, parameters, consensusSet);
  }

  // sampler stream of the probabilistic overload (default 1); set it to vary the hypotheses
  static uint64_t &seed() {
    static thread_local uint64_t s = 1;
    return s;
  }
  // diagnostics of the last compute() on this thread (iterations, hypotheses scanned, LM info)
  static lsqr_ransac_info &lastInfo() {
    static thread_local lsqr_ransac_info i;
    return i;
  }

 private:
  static void requireDevice(ParametersEstimator<T, S> *est, lsqr_model_cfg &cfg) {
    if (!est || !est->deviceModel(cfg))
      throw std::runtime_error(
          "lsqrRecipes::RANSAC: this estimator has no device model; the MI355X drop-in covers "
          "Plane/Sphere/Line/DenseLinearEquationSystem/SinglePointTargetUSCalibration/"
          "AbsoluteOrientation/PivotCalibration only");
  }
  static double finish(bool ok, const lsqr_ransac_info &info, const std::vector<double> &p,
                       const std::vector<uint8_t> &cons, std::vector<S> &parameters,
                       std::vector<bool> *consensusSet) {
    if (info.best_votes > 0 && consensusSet) {  // RANSAC.hxx:129-137: only when a set was found
      consensusSet->clear();
      consensusSet->insert(consensusSet->begin(), cons.begin(), cons.end());
    }
    if (ok) parameters.assign(p.begin(), p.begin() + info.n_params);
    return info.fraction;
  }
};

}  // namespace lsqrRecipes
#endif
