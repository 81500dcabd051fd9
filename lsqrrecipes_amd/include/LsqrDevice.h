// LsqrDevice.h -- C++ glue between the header-level API of the drop-in and the C ABI
// (include/lsqr_hip.h).  One lsqr_ctx per host thread (the reference is single threaded and not
// thread safe, RANSAC.hxx:44,59; here each thread simply gets its own context and stream).
// Errors of the device layer are never swallowed: anything other than LSQR_OK / LSQR_EMPTY throws
// std::runtime_error; there is no CPU path to fall back to.
#ifndef _LSQR_DEVICE_H_
#define _LSQR_DEVICE_H_

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "lsqr_hip.h"

namespace lsqrRecipes {
namespace detail {

class Device {
 public:
  static Device &instance() {
    static thread_local Device d;
    return d;
  }
  lsqr_ctx *ctx() {
    if (!h) {
      int dev = 0;
      if (const char *e = std::getenv("LSQR_DEVICE")) dev = std::atoi(e);
      int st = lsqr_ctx_create(dev, &h);
      if (st != LSQR_OK)
        throw std::runtime_error(std::string("lsqrRecipes: cannot create a device context: ") +
                                 lsqr_status_string(st));
    }
    return h;
  }
  // returns true for LSQR_OK, false for LSQR_EMPTY, throws otherwise
  bool check(int st) {
    if (st == LSQR_OK) return true;
    if (st == LSQR_EMPTY) return false;
    throw std::runtime_error(std::string("lsqrRecipes device error: ") + lsqr_status_string(st) +
                             " (" + (h ? lsqr_last_error(h) : "") + ")");
  }
  void model(const lsqr_model_cfg &cfg) { check(lsqr_set_model(ctx(), &cfg)); }
  // LSQR_DEVICES="0,1,2,3": RANSAC<T,S>::compute() shards its hypothesis batches over these devices
  // (lsqr_multi_*); unset or a single device: the one context above.  NULL when not requested.
  lsqr_multi *multi() {
    if (!multiTried) {
      multiTried = true;
      if (const char *e = std::getenv("LSQR_DEVICES")) {
        std::vector<int> devs;
        for (const char *p = e; *p;) {
          char *end = 0;
          long v = std::strtol(p, &end, 10);
          if (end == p) break;
          devs.push_back((int)v);
          p = (*end == ',') ? end + 1 : end;
        }
        if (devs.size() > 1) {
          int st = lsqr_multi_create(&devs[0], (int)devs.size(), &hm);
          if (st != LSQR_OK)
            throw std::runtime_error(std::string("lsqrRecipes: LSQR_DEVICES=") + e + ": " + lsqr_status_string(st));
        }
      }
    }
    return hm;
  }
  bool checkMulti(int st) {
    if (st == LSQR_OK) return true;
    if (st == LSQR_EMPTY) return false;
    throw std::runtime_error(std::string("lsqrRecipes multi-device error: ") + lsqr_status_string(st) + " (" +
                             (hm ? lsqr_multi_last_error(hm) : "") + ")");
  }
  ~Device() {
    if (hm) lsqr_multi_destroy(hm);
    if (h) lsqr_ctx_destroy(h);
  }

 private:
  Device() : h(0), hm(0), multiTried(false) {}
  lsqr_ctx *h;
  lsqr_multi *hm;
  bool multiTried;
};

// contiguous copy of the records behind a vector of pointers (the reference passes
// std::vector<T*> to estimate() / leastSquaresEstimate())
template <class T>
inline void gather(const std::vector<T *> &ptrs, std::vector<T> &out) {
  out.clear();
  out.reserve(ptrs.size());
  for (size_t i = 0; i < ptrs.size(); i++) out.push_back(*ptrs[i]);
}

// estimate(): minimal-subset solve of exactly k records, in the given (draw) order.  Closed-form models: the
// library's per-model code evaluated on the host (lsqr_estimate_host: the code the kernels run, same bits) -- no
// upload, no launch; the dense / US / phantom minimal solves are device kernels.
template <class T>
inline void exactFit(const lsqr_model_cfg &cfg, const T *recs, size_t count,
                     std::vector<double> &parameters) {
  parameters.clear();
  const int k = lsqr_min_subset(&cfg), P = lsqr_num_params(&cfg);
  if (count < (size_t)k) return;
  {
    std::vector<double> hp((size_t)(P > 0 ? P : 1));
    int np = 0;
    const int st = lsqr_estimate_host(&cfg, recs, count, sizeof(T), &hp[0], &np);
    if (st == LSQR_OK) {
      parameters.assign(hp.begin(), hp.begin() + np);
      return;
    }
    if (st == LSQR_EMPTY) return;  // degenerate subset: empty vector, as the reference
  }
  Device &d = Device::instance();
  d.model(cfg);
  d.check(lsqr_upload(d.ctx(), recs, count, sizeof(T)));
  std::vector<uint32_t> idx((size_t)k);
  for (int i = 0; i < k; i++) idx[i] = (uint32_t)i;
  d.check(lsqr_hypotheses_from_subsets(d.ctx(), &idx[0], 1));
  std::vector<double> p((size_t)P);
  uint8_t valid = 0;
  d.check(lsqr_get_hypothesis(d.ctx(), 0, &p[0], &valid));
  if (valid) parameters.assign(p.begin(), p.end());
}

// leastSquaresEstimate() over all given records
template <class T>
inline void lsFit(const lsqr_model_cfg &cfg, const T *recs, size_t count,
                  std::vector<double> &parameters, lsqr_fit_info *info = 0) {
  parameters.clear();
  if (count == 0) return;
  Device &d = Device::instance();
  d.model(cfg);
  d.check(lsqr_upload(d.ctx(), recs, count, sizeof(T)));
  std::vector<double> p(64);
  lsqr_fit_info fi;
  if (d.check(lsqr_ls_fit(d.ctx(), 0, &p[0], &fi))) parameters.assign(p.begin(), p.begin() + fi.n_params);
  if (info) *info = fi;
}

// leastSquaresEstimate() over records the caller has laid out itself (pointer, count, stride in bytes)
inline void lsFitRaw(const lsqr_model_cfg &cfg, const void *recs, size_t count, size_t stride_bytes,
                     std::vector<double> &parameters) {
  parameters.clear();
  if (count == 0) return;
  Device &d = Device::instance();
  d.model(cfg);
  d.check(lsqr_upload(d.ctx(), recs, count, stride_bytes));
  std::vector<double> p(64);
  lsqr_fit_info fi;
  if (d.check(lsqr_ls_fit(d.ctx(), 0, &p[0], &fi))) parameters.assign(p.begin(), p.begin() + fi.n_params);
}

// agree(parameters, datum): on the host (lsqr_agree_host: the kernels' own predicate, same bits) -- the reference's
// agree() is an inline that callers may use in a loop
template <class T>
inline bool agreeOne(const lsqr_model_cfg &cfg, const std::vector<double> &parameters, const T &rec) {
  if ((int)parameters.size() < lsqr_num_params(&cfg))
    throw std::out_of_range("lsqrRecipes: parameters vector too short for agree()");
  int a = 0;
  if (lsqr_agree_host(&cfg, &parameters[0], &rec, &a) == LSQR_OK) return a != 0;
  Device &d = Device::instance();
  d.model(cfg);
  d.check(lsqr_upload(d.ctx(), &rec, 1, sizeof(T)));
  uint8_t m = 0;
  d.check(lsqr_mask(d.ctx(), &parameters[0], 0, 1, &m, 0));
  return m != 0;
}

// getDistanceStatistics(): min / max / mean of the model's residual over the data
template <class T>
inline void distanceStats(const lsqr_model_cfg &cfg, const std::vector<double> &parameters,
                          const T *recs, size_t count, double &mn, double &mx, double &mean) {
  Device &d = Device::instance();
  d.model(cfg);
  d.check(lsqr_upload(d.ctx(), recs, count, sizeof(T)));
  double out[4];
  d.check(lsqr_stats(d.ctx(), &parameters[0], 0, out));
  mn = out[0];
  mx = out[1];
  mean = out[2];
}

// iterative refinement from a caller-supplied start (Sphere::geometricLeastSquaresEstimate,
// US::iterativeLeastSquaresEstimate): MINPACK control flow on the device, one pass per evaluation
template <class T>
inline void lmFit(const lsqr_model_cfg &cfg, const T *recs, size_t count,
                  const std::vector<double> &initial, std::vector<double> &final_) {
  final_.clear();
  Device &d = Device::instance();
  d.model(cfg);
  d.check(lsqr_upload(d.ctx(), recs, count, sizeof(T)));
  const int nmom = lsqr_moments_len(&cfg, 1);
  std::vector<double> block((size_t)nmom), xt(64, 0.0), x0(64, 0.0), out(64, 0.0);
  for (size_t i = 0; i < initial.size() && i < 64; i++) x0[i] = initial[i];
  d.check(lsqr_lm_begin(d.ctx(), &x0[0], &xt[0]));
  for (;;) {
    int cont = 0;
    lsqr_fit_info fi;
    d.check(lsqr_moments(d.ctx(), 0, 0, count, 1, &xt[0], &block[0]));
    bool ok = d.check(lsqr_lm_step(d.ctx(), &block[0], &xt[0], &cont, &out[0], &fi));
    if (!cont) {
      if (ok) final_.assign(out.begin(), out.begin() + fi.n_params);
      return;
    }
  }
}

}  // namespace detail

// Records kept on the device across several RANSAC<T,S>::compute() calls: the vector is uploaded once (its own
// context, so that other calls of the thread do not disturb it) and compute(parameters, estimator, resident, p,
// consensus) may be repeated with another threshold, probability, seed or least squares type -- or another estimator
// of the same record type -- without the PCIe copy that dominates a cold call (4.3 of 5.3 ms at 10 M points).  What
// was derived from the records alone (bounds, spatial index) is kept while the estimator type stays the same.
template <class T>
class ResidentData {
 public:
  explicit ResidentData(const std::vector<T> &data, int device = -1) : h(0), n(data.size()) {
    int dev = device;
    if (dev < 0) {
      dev = 0;
      if (const char *e = std::getenv("LSQR_DEVICE")) dev = std::atoi(e);
    }
    int st = lsqr_ctx_create(dev, &h);
    if (st != LSQR_OK)
      throw std::runtime_error(std::string("lsqrRecipes::ResidentData: cannot create a device context: ") +
                               lsqr_status_string(st));
    host = n ? &data[0] : 0;
    uploaded = false;
  }
  ~ResidentData() {
    if (h) lsqr_ctx_destroy(h);
  }
  size_t size() const { return n; }
  // the context with `cfg` as its model and the records resident (uploaded on first use: the record layout check
  // of lsqr_upload needs a model)
  lsqr_ctx *attach(const lsqr_model_cfg &cfg) {
    int st = lsqr_set_model(h, &cfg);
    if (st != LSQR_OK) fail(st);
    if (!uploaded || lsqr_count(h) != n) {
      if (n && (st = lsqr_upload(h, host, n, sizeof(T))) != LSQR_OK) fail(st);
      uploaded = true;
    }
    return h;
  }
  bool check(int st) {
    if (st == LSQR_OK) return true;
    if (st == LSQR_EMPTY) return false;
    fail(st);
    return false;
  }

 private:
  ResidentData(const ResidentData &);
  ResidentData &operator=(const ResidentData &);
  void fail(int st) {
    throw std::runtime_error(std::string("lsqrRecipes::ResidentData device error: ") + lsqr_status_string(st) + " (" +
                             lsqr_last_error(h) + ")");
  }
  lsqr_ctx *h;
  size_t n;
  const T *host;   // the caller's vector must outlive the first compute() (it is read once, then never again)
  bool uploaded;
};

}  // namespace lsqrRecipes
#endif
