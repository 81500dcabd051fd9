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
//  * estimators that expose a device model (ParametersEstimator::deviceModel -- every estimator this
//    library ships) run on the MI355X and nowhere else: without a usable device compute() throws.
//    A USER-DEFINED ParametersEstimator subclass (the plugin use the reference advertises,
//    readme.txt:40-72) has no device model; for it compute() runs the reference's serial loop
//    (RANSAC.hxx:49-139) over the estimator's own virtuals on the host, fed by the same counter-based
//    subset stream and the same replay of the adaptive stopping rule (lsqr_sample_subsets /
//    lsqr_replay), so a user estimator that restates a built-in one reaches the same iteration count,
//    winner and consensus set as the device path.
#ifndef _RANSAC_H_
#define _RANSAC_H_

#include <algorithm>
#include <set>
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
    if (!paramEstimator) throw std::invalid_argument("lsqrRecipes::RANSAC: null estimator");
    // RANSAC.hxx:16-19: invalid input returns 0 and leaves `parameters` untouched
    if (data.size() < paramEstimator->numForEstimate() || desiredProbabilityForNoOutliers >= 1.0 ||
        desiredProbabilityForNoOutliers <= 0.0)
      return 0;
    if (!paramEstimator->deviceModel(cfg) || forceHostLoop())
      return pluginCompute(parameters, paramEstimator, data, desiredProbabilityForNoOutliers, consensusSet);
    detail::Device &d = detail::Device::instance();
    std::vector<double> p(64);
    std::vector<uint8_t> cons(consensusSet ? data.size() : 0);
    lsqr_ransac_info info;
    bool ok;
    if (lsqr_multi *m = d.multi()) {  // LSQR_DEVICES lists several devices: batches sharded over them
      d.checkMulti(lsqr_multi_set_model(m, &cfg));
      d.checkMulti(lsqr_multi_upload(m, &data[0], data.size(), sizeof(T)));
      parameters.clear();
      ok = d.checkMulti(lsqr_multi_ransac(m, desiredProbabilityForNoOutliers, seed(), &p[0],
                                          consensusSet ? &cons[0] : NULL, &info));
      lastInfo() = info;
      return finish(ok, info, p, cons, parameters, consensusSet);
    }
    d.model(cfg);
    d.check(lsqr_upload(d.ctx(), &data[0], data.size(), sizeof(T)));
    parameters.clear();  // RANSAC.hxx:43
    ok = d.check(lsqr_ransac(d.ctx(), desiredProbabilityForNoOutliers, seed(), NULL, 0, &p[0],
                             consensusSet ? &cons[0] : NULL, &info));
    lastInfo() = info;
    return finish(ok, info, p, cons, parameters, consensusSet);
  }

  // the same on records that are already resident on the device (lsqrRecipes::ResidentData, LsqrDevice.h): no
  // upload; repeat with other thresholds / probabilities / seeds / estimators of the same record type
  static double compute(std::vector<S> &parameters, ParametersEstimator<T, S> *paramEstimator,
                        ResidentData<T> &data, double desiredProbabilityForNoOutliers,
                        std::vector<bool> *consensusSet = NULL) {
    lsqr_model_cfg cfg;
    if (!paramEstimator) throw std::invalid_argument("lsqrRecipes::RANSAC: null estimator");
    if (data.size() < paramEstimator->numForEstimate() || desiredProbabilityForNoOutliers >= 1.0 ||
        desiredProbabilityForNoOutliers <= 0.0)
      return 0;
    if (!paramEstimator->deviceModel(cfg))
      throw std::invalid_argument("lsqrRecipes::RANSAC: ResidentData needs an estimator with a device model");
    lsqr_ctx *ctx = data.attach(cfg);
    std::vector<double> p(64);
    std::vector<uint8_t> cons(consensusSet ? data.size() : 0);
    lsqr_ransac_info info;
    parameters.clear();
    bool ok = data.check(lsqr_ransac(ctx, desiredProbabilityForNoOutliers, seed(), NULL, 0, &p[0],
                                     consensusSet ? &cons[0] : NULL, &info));
    lastInfo() = info;
    return finish(ok, info, p, cons, parameters, consensusSet);
  }

  // exhaustive search over all subsets, reference RANSAC.h:111-113
  static double compute(std::vector<S> &parameters, ParametersEstimator<T, S> *paramEstimator,
                        std::vector<T> &data, std::vector<bool> *consensusSet = NULL) {
    lsqr_model_cfg cfg;
    if (!paramEstimator) throw std::invalid_argument("lsqrRecipes::RANSAC: null estimator");
    parameters.clear();  // RANSAC.hxx:165 clears before the size check
    if (data.size() < paramEstimator->numForEstimate()) return 0;
    if (!paramEstimator->deviceModel(cfg) || forceHostLoop())
      return pluginComputeExhaustive(parameters, paramEstimator, data, consensusSet);
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

 // route an estimator that HAS a device model through the plugin loop as well (tests: both paths must
  // agree on iterations, winner and consensus set); default false
  static bool &forceHostLoop() {
    static thread_local bool f = false;
    return f;
  }

 private:
  // ---- plugin path: estimators without a device model -------------------------------------------------
  // One hypothesis of the serial loop (RANSAC.hxx:84-99): exact fit of the subset, then the agree() pass
  // with the reference's early exit -- a hypothesis that can no longer overtake the best one is
  // abandoned, which never changes the winner (its final count would stay below the best).
  static bool tryHypothesis(ParametersEstimator<T, S> *est, std::vector<T> &data,
                            const std::vector<uint32_t> &subset, uint64_t bestVotes,
                            std::vector<S> &model, std::vector<char> &agrees, uint32_t &votes,
                            bool earlyExit) {
    std::vector<T *> minimal(subset.size());
    for (size_t l = 0; l < subset.size(); l++) minimal[l] = &data[subset[l]];  // draw order, RANSAC.hxx:65
    est->estimate(minimal, model);
    votes = 0;
    if (model.empty()) return false;  // degenerate subset, RANSAC.hxx:87-88
    const long long N = (long long)data.size();
    std::fill(agrees.begin(), agrees.end(), 0);
    for (long long m = 0; m < N; m++) {
      if (earlyExit && (long long)bestVotes - (long long)votes >= N - m + 1) break;  // RANSAC.hxx:94
      if (est->agree(model, data[(size_t)m])) {
        agrees[(size_t)m] = 1;
        votes++;
      }
    }
    return true;
  }

  static double pluginFinish(std::vector<S> &parameters, ParametersEstimator<T, S> *est,
                             std::vector<T> &data, const std::vector<char> &best, uint64_t bestVotes,
                             uint64_t iterations, uint64_t bestIndex, std::vector<bool> *consensusSet) {
    lsqr_ransac_info &info = lastInfo();
    info = lsqr_ransac_info();
    info.iterations = iterations;
    info.evaluated = iterations;
    info.best_index = bestIndex;
    info.best_votes = (uint32_t)bestVotes;
    info.fraction = (double)bestVotes / (double)data.size();
    if (bestVotes > 0) {  // RANSAC.hxx:129-139
      std::vector<T *> inliers;
      inliers.reserve((size_t)bestVotes);
      for (size_t m = 0; m < data.size(); m++)
        if (best[m]) inliers.push_back(&data[m]);
      if (consensusSet) consensusSet->assign(best.begin(), best.end());
      est->leastSquaresEstimate(inliers, parameters);
      info.n_params = (int32_t)parameters.size();
      info.fit.n_params = info.n_params;
      info.fit.n_used = bestVotes;
    }
    return info.fraction;
  }

  static double pluginCompute(std::vector<S> &parameters, ParametersEstimator<T, S> *est,
                              std::vector<T> &data, double p, std::vector<bool> *consensusSet) {
    const size_t N = data.size();
    const int k = (int)est->numForEstimate();
    parameters.clear();  // RANSAC.hxx:43
    uint64_t st[6];      // {i, numTries, best votes, best index, has best, done}: lsqr_replay's state
    lsqr_replay_init(N, k, p, st);
    std::set<std::vector<uint32_t> > drawn;  // sorted index tuples already tried, RANSAC.hxx:33-34,79
    std::vector<uint32_t> subset((size_t)k), key((size_t)k);
    std::vector<S> model;
    std::vector<char> cur(N, 0), best(N, 0);
    for (uint64_t it = 0; !st[5]; it++) {
      if (lsqr_sample_subsets(seed(), it, 1, N, k, &subset[0]) != LSQR_OK)
        throw std::runtime_error("lsqrRecipes::RANSAC: subset sampler failed");
      key = subset;
      std::sort(key.begin(), key.end());
      uint8_t valid = 0;
      uint32_t votes = 0;
      if (drawn.insert(key).second)  // a repeated subset still consumes the iteration, RANSAC.hxx:114-116
        valid = tryHypothesis(est, data, subset, st[2], model, cur, votes, true) ? 1 : 0;
      const uint64_t hadBest = st[4], bestBefore = st[3];
      // strict '>' update and the adaptive bound on numTries (RANSAC.hxx:100-111), shared with the device path
      if (lsqr_replay(N, k, p, &subset[0], &valid, &votes, 1, it, NULL, st) == 0) break;
      if (st[4] && (!hadBest || st[3] != bestBefore)) best.swap(cur);
    }
    return pluginFinish(parameters, est, data, best, st[4] ? st[2] : 0, st[0], st[3], consensusSet);
  }

  // exhaustive overload (RANSAC.hxx:150-249): every k-subset in lexicographic order, full agree() passes,
  // first maximum wins
  static double pluginComputeExhaustive(std::vector<S> &parameters, ParametersEstimator<T, S> *est,
                                        std::vector<T> &data, std::vector<bool> *consensusSet) {
    const size_t N = data.size();
    const int k = (int)est->numForEstimate();
    std::vector<uint32_t> subset((size_t)k);
    for (int l = 0; l < k; l++) subset[(size_t)l] = (uint32_t)l;
    std::vector<S> model;
    std::vector<char> cur(N, 0), best(N, 0);
    uint64_t bestVotes = 0, bestIndex = 0, index = 0;
    if (k == 0) return pluginFinish(parameters, est, data, best, 0, 0, 0, consensusSet);
    for (bool more = true; more; index++) {
      uint32_t votes = 0;
      if (tryHypothesis(est, data, subset, bestVotes, model, cur, votes, false) && votes > bestVotes) {
        bestVotes = votes;  // RANSAC.hxx:245 strict
        bestIndex = index;
        best.swap(cur);
      }
      int l = k - 1;  // next combination
      while (l >= 0 && subset[(size_t)l] == (uint32_t)(N - (size_t)k + (size_t)l)) l--;
      if (l < 0) more = false;
      else {
        subset[(size_t)l]++;
        for (int j = l + 1; j < k; j++) subset[(size_t)j] = subset[(size_t)j - 1] + 1;
      }
    }
    return pluginFinish(parameters, est, data, best, bestVotes, index, bestIndex, consensusSet);
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
