/*
 * ref_driver.cxx -- builds the REAL reference RANSAC driver into oracle/_ref/libref_ransac.so.
 *
 * TEST INFRASTRUCTURE ONLY.  /root/reference/parametersEstimators/RANSAC.h(.hxx) and
 * ParametersEstimator.h are compiled unmodified from where they lie (include path only; no
 * reference source is copied into this repository).  They are VNL-free.  The estimator
 * classes of the reference are NOT (they need VNL, which the image lacks), so the plugin handed
 * to the reference driver is an adapter over the oracle's C restatement (estimators.c).
 *
 * Determinism: RANSAC.hxx seeds with srand(time(NULL)) and draws with rand() (:44,:59).  This
 * library is linked with -Bsymbolic-functions and defines its own rand()/srand(), so the
 * reference's calls bind to the deterministic LCG below (the same LCG as orc_lcg_rand).
 *
 * Used (a) to pin oracle/ransac.c against the real RANSAC.hxx, (b) as bench.py's
 * cpu_baseline "reference" leg.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "RANSAC.h" /* from /root/reference/parametersEstimators */
#include "lsqr_oracle.h"

static uint64_t g_lcg = 0;
static uint64_t g_rand_calls = 0;
extern "C" int rand(void) {
  g_lcg = g_lcg * 6364136223846793005ULL + 1442695040888963407ULL;
  g_rand_calls++;
  return (int)((g_lcg >> 33) & 0x7fffffff);
}
extern "C" void srand(unsigned) {}

namespace {

template <int ND>
struct Rec {
  double v[ND];
};

struct Counters {
  uint64_t estimate_calls, agree_calls, ls_calls;
  std::vector<uint32_t> subsets; /* draw order, one tuple per estimate() call */
};

template <int ND>
class Adapter : public lsqrRecipes::ParametersEstimator<Rec<ND>, double> {
 public:
  Adapter(const orc_cfg &c, const Rec<ND> *base, Counters *cnt, bool record)
      : lsqrRecipes::ParametersEstimator<Rec<ND>, double>(orc_min_subset(&c)),
        cfg(c), base(base), cnt(cnt), record(record) {}

  virtual void estimate(std::vector<Rec<ND> *> &data, std::vector<double> &parameters) {
    double out[72];
    const double *ptrs[64];
    parameters.clear();
    for (size_t i = 0; i < data.size() && i < 64; i++) {
      ptrs[i] = data[i]->v;
      if (record) cnt->subsets.push_back((uint32_t)(data[i] - base));
    }
    cnt->estimate_calls++;
    int np = orc_estimate(&cfg, ptrs, data.size(), out);
    parameters.assign(out, out + np);
  }
  virtual void estimate(std::vector<Rec<ND> > &data, std::vector<double> &parameters) {
    std::vector<Rec<ND> *> p;
    for (size_t i = 0; i < data.size(); i++) p.push_back(&data[i]);
    estimate(p, parameters);
  }
  virtual void leastSquaresEstimate(std::vector<Rec<ND> *> &data,
                                    std::vector<double> &parameters) {
    double out[72];
    std::vector<const double *> ptrs(data.size());
    for (size_t i = 0; i < data.size(); i++) ptrs[i] = data[i]->v;
    cnt->ls_calls++;
    int np = orc_ls(&cfg, ptrs.data(), ptrs.size(), out);
    parameters.assign(out, out + np);
  }
  virtual void leastSquaresEstimate(std::vector<Rec<ND> > &data,
                                    std::vector<double> &parameters) {
    std::vector<Rec<ND> *> p;
    for (size_t i = 0; i < data.size(); i++) p.push_back(&data[i]);
    leastSquaresEstimate(p, parameters);
  }
  virtual bool agree(std::vector<double> &parameters, Rec<ND> &data) {
    cnt->agree_calls++;
    return orc_agree(&cfg, &parameters[0], data.v) != 0;
  }

 private:
  orc_cfg cfg;
  const Rec<ND> *base;
  Counters *cnt;
  bool record;
};

template <int ND>
double run(const orc_cfg *c, const double *data, size_t n, double p, int exhaustive,
           double *params, int *nparams, uint8_t *consensus, Counters *cnt, bool record) {
  std::vector<Rec<ND> > v(n);
  if (n) std::memcpy(&v[0], data, n * sizeof(Rec<ND>));
  Adapter<ND> est(*c, n ? &v[0] : 0, cnt, record);
  std::vector<double> out;
  if (*nparams > 0) out.assign(params, params + *nparams); /* to observe "untouched" */
  std::vector<bool> cons;
  double r;
  if (exhaustive)
    r = lsqrRecipes::RANSAC<Rec<ND>, double>::compute(out, &est, v, &cons);
  else
    r = lsqrRecipes::RANSAC<Rec<ND>, double>::compute(out, &est, v, p, &cons);
  *nparams = (int)out.size();
  for (size_t i = 0; i < out.size(); i++) params[i] = out[i];
  if (consensus)
    for (size_t i = 0; i < cons.size() && i < n; i++) consensus[i] = cons[i] ? 1 : 0;
  return r;
}

}  // namespace

extern "C" {

/* Runs the reference's RANSAC<T,S>::compute on tightly packed records (stride ==
 * orc_record_doubles(cfg)).  *nparams on entry = number of doubles already in params (lets the
 * caller observe the "parameters untouched on invalid input" convention, RANSAC.hxx:16-19).
 * stats_out = {estimate calls, agree calls, ls calls, rand calls}.  subsets_out (may be NULL)
 * receives up to subsets_cap tuples of k indices, one per estimate() call. */
double ref_ransac(const orc_cfg *c, const double *data, size_t n, double p, int exhaustive,
                  uint64_t lcg_seed, double *params, int *nparams, uint8_t *consensus,
                  uint64_t stats_out[4], uint32_t *subsets_out, size_t subsets_cap) {
  Counters cnt;
  cnt.estimate_calls = cnt.agree_calls = cnt.ls_calls = 0;
  g_lcg = lcg_seed;
  g_rand_calls = 0;
  bool record = subsets_out != 0;
  double r = -1;
  switch (orc_record_doubles(c)) {
#define CASE(ND) \
  case ND: r = run<ND>(c, data, n, p, exhaustive, params, nparams, consensus, &cnt, record); break;
    CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(9) CASE(15) CASE(18) CASE(17) CASE(33)
    CASE(65)
#undef CASE
    default: return -1;
  }
  if (stats_out) {
    stats_out[0] = cnt.estimate_calls;
    stats_out[1] = cnt.agree_calls;
    stats_out[2] = cnt.ls_calls;
    stats_out[3] = g_rand_calls;
  }
  if (subsets_out) {
    size_t k = (size_t)orc_min_subset(c);
    size_t ncopy = std::min(cnt.subsets.size(), subsets_cap * k);
    if (ncopy) std::memcpy(subsets_out, &cnt.subsets[0], ncopy * sizeof(uint32_t));
  }
  return r;
}

}  // extern "C"
