// host_entry.h -- everything of the C ABI that needs no device: the model dispatch, the description of a model
// (minimal subset, parameters, record), the counter-based subset stream on the host (lsqr_sample_subsets), the replay of
// RANSAC.hxx:49-117 over a batch (lsqr_replay*: duplicate set, strict '>' update, adaptive numTries with the reference's
// saturating choose), and the single-datum host calls (lsqr_agree_host / lsqr_estimate_host: the kernels' own per-model
// code compiled for the host).  Plain C++ -- no HIP runtime -- so that the SAME source is (a) compiled into
// liblsqr_hip.so by hipcc (lsqr_hip.hip wraps every host_* function in its extern "C" entry point) and (b) compiled
// alone by g++ -fsanitize=address,undefined in the CPU suite (tests/test_sanitizers.py, tests/sanitize/driver.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>

#include "../../include/lsqr_hip.h"
#include "dense_model.h"
#include "models.h"
#include "models_nd.h"
#include "phantom.h"
#include "rigid.h"
#include "sampler.h"
#include "us.h"

namespace lsqr {

template <class M>
struct Tag {
  typedef M type;
};

// model dispatch: f(Tag<Model>{}) -> int
template <class F>
inline int dispatch(const lsqr_model_cfg &cfg, F &&f) {
#ifdef LSQR_DEV_SUBSET  // development builds (make DEV=1): the five BASELINE workloads only, a third of the compile time
  switch (cfg.model) {
    case LSQR_MODEL_PLANE: if (cfg.dim == 3) return f(Tag<PlaneModel<3>>{}); break;
    case LSQR_MODEL_SPHERE: if (cfg.dim == 3) return f(Tag<SphereModel<3>>{}); break;
    case LSQR_MODEL_LINE: if (cfg.dim == 3) return f(Tag<LineModel<3>>{}); break;
    case LSQR_MODEL_US_SINGLE: return f(Tag<USModel<true>>{});
    case LSQR_MODEL_DENSE: if (cfg.dim <= 64 && cfg.dim > 32) return f(Tag<DenseModel<64>>{}); break;
    default: break;
  }
  return LSQR_ERR_INVALID;
#else
  switch (cfg.model) {
    case LSQR_MODEL_PLANE:
      if (cfg.dim == 3) return f(Tag<PlaneModel<3>>{});
      if (cfg.dim == 2) return f(Tag<PlaneModel<2>>{});
      if (cfg.dim == 4) return f(Tag<PlaneModelN<4>>{});  // general dimension: models_nd.h
      if (cfg.dim == 5) return f(Tag<PlaneModelN<5>>{});
      if (cfg.dim == 6) return f(Tag<PlaneModelN<6>>{});
      if (cfg.dim == 7) return f(Tag<PlaneModelN<7>>{});
      if (cfg.dim == 8) return f(Tag<PlaneModelN<8>>{});
      break;
    case LSQR_MODEL_SPHERE:
      if (cfg.dim == 3) return f(Tag<SphereModel<3>>{});
      if (cfg.dim == 2) return f(Tag<SphereModel<2>>{});
      if (cfg.dim == 4) return f(Tag<SphereModelN<4>>{});
      if (cfg.dim == 5) return f(Tag<SphereModelN<5>>{});
      if (cfg.dim == 6) return f(Tag<SphereModelN<6>>{});
      if (cfg.dim == 7) return f(Tag<SphereModelN<7>>{});
      if (cfg.dim == 8) return f(Tag<SphereModelN<8>>{});
      break;
    case LSQR_MODEL_LINE:
      if (cfg.dim == 3) return f(Tag<LineModel<3>>{});
      if (cfg.dim == 2) return f(Tag<LineModel<2>>{});
      if (cfg.dim == 4) return f(Tag<LineModelN<4>>{});
      if (cfg.dim == 5) return f(Tag<LineModelN<5>>{});
      if (cfg.dim == 6) return f(Tag<LineModelN<6>>{});
      if (cfg.dim == 7) return f(Tag<LineModelN<7>>{});
      if (cfg.dim == 8) return f(Tag<LineModelN<8>>{});
      break;
    case LSQR_MODEL_US_SINGLE: return f(Tag<USModel<true>>{});
    case LSQR_MODEL_US_POINTER: return f(Tag<USModel<false>>{});
    case LSQR_MODEL_ABSOR: return f(Tag<AbsOrModel>{});
    case LSQR_MODEL_PIVOT: return f(Tag<PivotModel>{});
    case LSQR_MODEL_RAY: return f(Tag<RayModel>{});
    case LSQR_MODEL_LINE2D: return f(Tag<Line2DModel>{});
    case LSQR_MODEL_PHANTOM: return f(Tag<PhantomModel>{});
    case LSQR_MODEL_DENSE:
      if (cfg.dim >= 1 && cfg.dim <= 8) return f(Tag<DenseModel<8>>{});
      if (cfg.dim <= 16 && cfg.dim > 8) return f(Tag<DenseModel<16>>{});
      if (cfg.dim <= 32 && cfg.dim > 16) return f(Tag<DenseModel<32>>{});
      if (cfg.dim <= 64 && cfg.dim > 32) return f(Tag<DenseModel<64>>{});
      break;
    default: break;
  }
  return LSQR_ERR_INVALID;
#endif
}

// the per-model constants a configuration implies (what the reference's constructors / setters store)
inline void model_consts(const lsqr_model_cfg &cfg, ModelConsts *mc) {
  mc->delta = cfg.delta;
  mc->delta_sq = cfg.delta * cfg.delta;
  mc->dim = cfg.dim;
  mc->ls_type = cfg.ls_type;
  mc->thr = square_threshold(mc->delta_sq);
  const double ce = sin(cfg.aux);  // RayIntersectionParametersEstimator.cxx:13-14
  mc->aux = ce * ce;
  mc->absmax = 0.0;
  mc->absmax_rot = 0.0;
}

inline bool cfg_supported(const lsqr_model_cfg &cfg) {
  return dispatch(cfg, [](auto) { return (int)LSQR_OK; }) == LSQR_OK;
}

// ---- RANSAC.hxx replay --------------------------------------------------------------------------
struct TupleLess {
  bool operator()(const std::vector<uint32_t> &a, const std::vector<uint32_t> &b) const {
    return a < b;  // lexicographic == RANSAC.h:135-149 SubSetIndexComparator
  }
};
// The subsets already drawn (RANSAC.hxx:79: a std::set of sorted index tuples).  A set of heap-allocated vectors costs
// ~250 ns per hypothesis -- 1 ms per batch of 4096, as long as the batch's scan on the device --, so tuples of up to
// four indices (every point model, the rigid and ray estimators) live in an open-addressing table of 16-byte keys;
// longer tuples (dense system, plane phantom) keep the ordered set.  Only membership is ever asked.
struct DedupSet {
  struct Key {
    uint64_t a, b;  // four 32-bit indices (+1, sorted), zero padded: (0, 0) never occurs as a key
  };
  std::vector<Key> tab;
  size_t used = 0;
  std::set<std::vector<uint32_t>, TupleLess> big;
  static uint64_t mix(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
  }
  void grow() {
    std::vector<Key> old;
    old.swap(tab);
    tab.assign(old.empty() ? 8192 : old.size() * 2, Key{0, 0});
    used = 0;
    for (const Key &k : old)
      if (k.a | k.b) put(k);
  }
  bool put(const Key &k) {  // true: was not present
    const size_t mask = tab.size() - 1;
    size_t i = (size_t)(mix(k.a) ^ mix(k.b + 0x9e3779b97f4a7c15ULL)) & mask;
    for (;; i = (i + 1) & mask) {
      Key &s = tab[i];
      if (!(s.a | s.b)) {
        s = k;
        used++;
        return true;
      }
      if (s.a == k.a && s.b == k.b) return false;
    }
  }
  // the k <= 4 indices + 1, sorted ascending
  bool insert_small(const uint32_t *key, int k) {
    if (tab.empty() || 2 * (used + 1) > tab.size()) grow();
    Key q{0, 0};
    for (int l = 0; l < k; l++) {
      if (l < 2) q.a |= (uint64_t)key[l] << (32 * l);
      else q.b |= (uint64_t)key[l] << (32 * (l - 2));
    }
    return put(q);
  }
  bool insert_sorted(const std::vector<uint32_t> &key) {
    if (key.size() > 4) return big.insert(key).second;
    return insert_small(key.data(), (int)key.size());
  }
};

// C(n, m), the cap on numTries (RANSAC.hxx:41,110).  The reference evaluates it in double (:254-280): the
// running product over the shorter of the two factor ranges, ascending, one division, saturation to UINT_MAX
// when a product overflowed or the quotient does not fit.  The cap has to come out identical, so the same
// products are formed in the same order here.
inline unsigned int choose_sat(unsigned int n, unsigned int m) {
  if (m > n) return 0;
  const unsigned int shorter = std::min(m, n - m);
  double top = 1.0, bottom = 1.0;
  for (double f = (double)(n - shorter) + 1.0; f <= (double)n; f += 1.0) top *= f;
  for (double f = 1.0; f <= (double)shorter; f += 1.0) bottom *= f;
  const double c = top / bottom;
  const bool saturated = std::isinf(top) || std::isinf(bottom) || !(c <= 4294967295.0);
  return saturated ? 0xFFFFFFFFu : (unsigned int)c;
}

// (int) cast at RANSAC.hxx:108 made explicit: out-of-range / NaN -> 0x80000000 as cvttsd2si does
inline unsigned int cast_tries(double x) {
  if (!(x > -2147483649.0 && x < 2147483648.0)) return 0x80000000u;
  return (unsigned int)(int)x;
}

enum { RS_I = 0, RS_TRIES = 1, RS_BEST = 2, RS_BEST_IDX = 3, RS_HAS = 4, RS_DONE = 5 };
inline bool has_any_best(const uint64_t *rs) { return rs[RS_HAS] != 0; }

inline int host_min_subset(const lsqr_model_cfg *cfg) {
  if (!cfg) return 0;
  switch (cfg->model) {
    case LSQR_MODEL_PLANE: return cfg->dim;
    case LSQR_MODEL_SPHERE: return cfg->dim + 1;
    case LSQR_MODEL_LINE: return 2;
    case LSQR_MODEL_DENSE: return cfg->dim;
    case LSQR_MODEL_US_SINGLE: return 4;
    case LSQR_MODEL_US_POINTER: return 3;
    case LSQR_MODEL_ABSOR: return 3;
    case LSQR_MODEL_PIVOT: return 3;
    case LSQR_MODEL_RAY: return 2;
    case LSQR_MODEL_LINE2D: return 2;
    case LSQR_MODEL_PHANTOM: return 31;
  }
  return 0;
}
inline int host_num_params(const lsqr_model_cfg *cfg) {
  if (!cfg) return 0;
  switch (cfg->model) {
    case LSQR_MODEL_PLANE:
    case LSQR_MODEL_LINE: return 2 * cfg->dim;
    case LSQR_MODEL_SPHERE: return cfg->dim + 1;
    case LSQR_MODEL_DENSE: return cfg->dim;
    case LSQR_MODEL_US_SINGLE: return 20;
    case LSQR_MODEL_US_POINTER: return 17;
    case LSQR_MODEL_ABSOR: return 7;
    case LSQR_MODEL_PIVOT: return 6;
    case LSQR_MODEL_RAY: return 3;
    case LSQR_MODEL_LINE2D: return 4;
    case LSQR_MODEL_PHANTOM: return 41;
  }
  return 0;
}
inline int host_record_doubles(const lsqr_model_cfg *cfg) {
  if (!cfg) return 0;
  switch (cfg->model) {
    case LSQR_MODEL_PLANE:
    case LSQR_MODEL_SPHERE:
    case LSQR_MODEL_LINE: return cfg->dim;
    case LSQR_MODEL_DENSE: return cfg->dim + 1;
    case LSQR_MODEL_US_SINGLE: return 15;
    case LSQR_MODEL_US_POINTER: return 18;
    case LSQR_MODEL_ABSOR: return cfg->ls_type == 2 ? 7 : 6;  // weighted fit: [first, second, weight]
    case LSQR_MODEL_PIVOT: return 13;
    case LSQR_MODEL_RAY: return 6;
    case LSQR_MODEL_LINE2D: return 2;
    case LSQR_MODEL_PHANTOM: return 15;
  }
  return 0;
}

inline int host_sample_subsets(uint64_t seed, uint64_t first, size_t H, uint64_t n, int k, uint32_t *out) {
  if (!out || k < 1 || k > 64 || n < (uint64_t)k || n > 0xFFFFFFF0ull) return LSQR_ERR_INVALID;
  uint32_t sorted[64];
  for (size_t h = 0; h < H; h++) ctr_subset(seed, first + h, n, k, out + h * (size_t)k, sorted);
  return LSQR_OK;
}

// ---- replay of the serial loop ------------------------------------------------------------------------
inline void *host_dedup_create(int) { return new DedupSet(); }
inline void host_dedup_destroy(void *s) { delete (DedupSet *)s; }

inline int host_replay_init(size_t n, int k, double, uint64_t st[6]) {
  if (!st) return LSQR_ERR_INVALID;
  st[RS_I] = 0;
  st[RS_TRIES] = choose_sat((unsigned int)n, (unsigned int)k);  // RANSAC.hxx:41,47
  st[RS_BEST] = 0;
  st[RS_BEST_IDX] = 0;
  st[RS_HAS] = 0;
  st[RS_DONE] = (st[RS_TRIES] == 0);
  return LSQR_OK;
}

// RANSAC.hxx:49-117 over the entries of one batch; base_index = loop index of entry 0.
inline size_t host_replay(size_t n, int k, double p, const uint32_t *subsets, const uint8_t *valid,
                   const uint32_t *votes, size_t H, uint64_t base_index, void *dedup,
                   uint64_t st[6]) {
  DedupSet *set = (DedupSet *)dedup;
  const unsigned int N = (unsigned int)n;
  const unsigned int allTries = choose_sat(N, (unsigned int)k);
  const double numerator = log(1.0 - p);
  size_t e = 0;
  std::vector<uint32_t> key((size_t)k);
  for (; e < H && !st[RS_DONE]; e++) {
    uint64_t i = base_index + e;
    if (i >= st[RS_TRIES]) {
      st[RS_DONE] = 1;
      break;
    }
    st[RS_I] = i + 1;
    bool fresh = true;
    if (set && k <= 4) {
      uint32_t kk[4] = {0, 0, 0, 0};
      for (int l = 0; l < k; l++) {  // :71-76, insertion sort of at most four
        uint32_t v = subsets[e * k + l] + 1;
        int j = l;
        for (; j > 0 && kk[j - 1] > v; j--) kk[j] = kk[j - 1];
        kk[j] = v;
      }
      fresh = set->insert_small(kk, k);  // :79
    } else if (set) {
      for (int l = 0; l < k; l++) key[l] = subsets[e * k + l] + 1;  // :71-76
      std::sort(key.begin(), key.end());
      fresh = set->insert_sorted(key);  // :79
    }
    if (fresh && valid[e]) {            // :84-88
      unsigned int cur = votes[e];
      if (cur > st[RS_BEST]) {          // :100 strict
        st[RS_BEST] = cur;
        st[RS_BEST_IDX] = i;
        st[RS_HAS] = 1;
        if (cur == N) {                 // :104-105
          st[RS_DONE] = 1;
          e++;
          break;
        }
        double denominator = log(1.0 - pow((double)cur / (double)N, (double)k));
        unsigned int t = cast_tries(numerator / denominator + 0.5);  // :108
        st[RS_TRIES] = t < allTries ? t : allTries;                   // :110
      }
    }
    if (i + 1 >= st[RS_TRIES]) {
      st[RS_DONE] = 1;
      e++;
      break;
    }
  }
  return e;
}

// ---- single-datum calls on the HOST ------------------------------------------------------------------------
// ParametersEstimator::agree(parameters, datum) is a ten-flop inline in the reference (PlaneParametersEstimator
// .hxx:196-203) that user code may call in a loop, and estimate() of a minimal subset a closed form: an upload and a
// kernel launch per call would cost microseconds each.  These two evaluate the SAME per-model code the kernels run
// (models.h, models_nd.h, rigid.h, us.h -- LSQR_HD, compiled here for the host with -ffp-contract=off; tests/
// test_host_math.py and tests/test_host_calls.py hold host and device to the same bits).  No context, no device.
// LSQR_ERR_INVALID: the model has no host form for that call (minimal solves that are wave kernels: dense, US,
// phantom) -- the caller then takes the device path.
inline int host_agree_host(const lsqr_model_cfg *cfg, const double *params, const void *record, int *agree_out) {
  if (!cfg || !params || !record || !agree_out || !cfg_supported(*cfg)) return LSQR_ERR_INVALID;
  ModelConsts mc;
  model_consts(*cfg, &mc);
  return dispatch(*cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    double sp[128];
    for (int j = 0; j < 128; j++) sp[j] = 0.0;
    // the caller's vector holds host_num_params(cfg) doubles: for the dense model that is cfg.dim, NOT the padded
    // width M::P of DenseModel<8/16/32/64> (the zero fill above is the padding)
    const int np = host_num_params(cfg) < (int)M::P ? host_num_params(cfg) : (int)M::P;
    for (int j = 0; j < np; j++) sp[j] = params[j];
    M::prepare(sp, mc);
    double x[M::REC > 0 ? M::REC : 1];
    M::load((const double *)record, mc, x);
    *agree_out = M::agree(sp, x, mc) ? 1 : 0;
    return LSQR_OK;
  });
}

inline int host_estimate_host(const lsqr_model_cfg *cfg, const void *records, size_t count, size_t stride_bytes,
                       double *params_out, int *n_params_out) {
  if (!cfg || !records || !params_out || !n_params_out || !cfg_supported(*cfg) || stride_bytes % sizeof(double))
    return LSQR_ERR_INVALID;
  ModelConsts mc;
  model_consts(*cfg, &mc);
  const size_t stride = stride_bytes / sizeof(double);
  return dispatch(*cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (M::IS_DENSE || M::IS_US || requires { M::IS_PHANTOM; }) {
      return LSQR_ERR_INVALID;  // wave / workgroup kernels: device only
    } else {
      if (count < (size_t)M::K || stride < (size_t)M::ND) return LSQR_ERR_INVALID;
      double r[M::K][M::ND];
      for (int l = 0; l < (int)M::K; l++)
        for (int j = 0; j < (int)M::ND; j++) r[l][j] = ((const double *)records)[l * stride + j];
      double par[M::P];
      const bool ok = M::estimate(r, mc, par);
      *n_params_out = ok ? (int)M::P : 0;
      for (int j = 0; j < (int)M::P; j++) params_out[j] = ok ? par[j] : 0.0;
      return ok ? LSQR_OK : LSQR_EMPTY;
    }
  });
}


}  // namespace lsqr
