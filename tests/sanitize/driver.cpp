// CPU sanitizer leg (tests/test_sanitizers.py): the product's host-side code -- csrc/host_entry.h (subset stream, replay of
// RANSAC.hxx:49-117, duplicate set, single-datum agree / estimate for every model), csrc/lm_core.h + the per-model
// arithmetic of csrc/{models,models_nd,rigid,us,phantom,dense_model}.h, and the plugin loop of
// lsqrrecipes_amd/include/RANSAC.h -- compiled by g++ with -fsanitize=address,undefined -fno-sanitize-recover and driven
// with buffers of exactly the documented sizes (heap blocks: a read or write one element past them stops the run),
// dense dimensions that are not powers of two, duplicate-ridden batches and degenerate subsets.  The extern "C"
// definitions below are the library's own wrappers (lsqr_hip.hip) over the same host_* functions; being defined in the
// executable they take precedence over liblsqr_hip.so's, so RANSAC.h's plugin path runs on the sanitized copies.
// The oracle's C sources are linked in (same flags) and used as the checker.  Test-only.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <vector>

#include "host_entry.h"
#include "lsqr_oracle.h"

using namespace lsqr;

extern "C" {
LSQR_API int lsqr_min_subset(const lsqr_model_cfg *cfg) { return host_min_subset(cfg); }
LSQR_API int lsqr_num_params(const lsqr_model_cfg *cfg) { return host_num_params(cfg); }
LSQR_API int lsqr_record_doubles(const lsqr_model_cfg *cfg) { return host_record_doubles(cfg); }
LSQR_API int lsqr_sample_subsets(uint64_t seed, uint64_t first, size_t H, uint64_t n, int k, uint32_t *out) {
  return host_sample_subsets(seed, first, H, n, k, out);
}
LSQR_API void *lsqr_dedup_create(int k) { return host_dedup_create(k); }
LSQR_API void lsqr_dedup_destroy(void *s) { host_dedup_destroy(s); }
LSQR_API int lsqr_replay_init(size_t n, int k, double p, uint64_t st[6]) { return host_replay_init(n, k, p, st); }
LSQR_API size_t lsqr_replay(size_t n, int k, double p, const uint32_t *subsets, const uint8_t *valid,
                            const uint32_t *votes, size_t H, uint64_t base_index, void *dedup, uint64_t st[6]) {
  return host_replay(n, k, p, subsets, valid, votes, H, base_index, dedup, st);
}
LSQR_API int lsqr_agree_host(const lsqr_model_cfg *cfg, const double *params, const void *record, int *agree_out) {
  return host_agree_host(cfg, params, record, agree_out);
}
LSQR_API int lsqr_estimate_host(const lsqr_model_cfg *cfg, const void *records, size_t count, size_t stride_bytes,
                                double *params_out, int *n_params_out) {
  return host_estimate_host(cfg, records, count, stride_bytes, params_out, n_params_out);
}
}

#include "RANSAC.h"

static int failures = 0;
#define CHECK(...)                                                  \
  do {                                                              \
    if (!(__VA_ARGS__)) {                                                   \
      printf("CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #__VA_ARGS__);   \
      failures++;                                                   \
    }                                                               \
  } while (0)

static std::mt19937_64 rng(20261005);
static double U(double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); }
static double N(double s) { return std::normal_distribution<double>(0.0, s)(rng); }

// exact-size heap blocks: ASan's red zones sit right behind the last element
template <class T>
static std::unique_ptr<T[]> block(size_t n) { return std::unique_ptr<T[]>(new T[n ? n : 1]); }

static void samplerTest() {
  const int ks[] = {1, 2, 3, 4, 5, 31, 64};
  for (int k : ks) {
    const uint64_t ns[] = {(uint64_t)k, (uint64_t)k + 1, 1000, 4000000000ull};
    for (uint64_t n : ns) {
      const size_t H = 37;
      auto out = block<uint32_t>(H * (size_t)k);
      CHECK(lsqr_sample_subsets(5, 123456789, H, n, k, out.get()) == LSQR_OK);
      for (size_t h = 0; h < H; h++) {
        uint32_t want[64];
        orc_ctr_subset(5, 123456789 + h, (size_t)n, k, want);  // the oracle's restatement of the same stream
        for (int l = 0; l < k; l++) {
          CHECK(out[h * k + l] < n);
          if (n <= 0xFFFFFFFFull) CHECK(out[h * k + l] == want[l]);
          for (int m = 0; m < l; m++) CHECK(out[h * k + l] != out[h * k + m]);
        }
      }
    }
  }
  uint32_t one;
  CHECK(lsqr_sample_subsets(1, 0, 1, 3, 4, &one) == LSQR_ERR_INVALID);   // n < k
  CHECK(lsqr_sample_subsets(1, 0, 1, 10, 65, &one) == LSQR_ERR_INVALID);  // k > 64
  CHECK(lsqr_sample_subsets(1, 0, 0, 10, 1, &one) == LSQR_OK);            // empty batch writes nothing
}

// the replay against a literal restatement of RANSAC.hxx:49-117 on the same batch
static void replayTest() {
  const int ks[] = {2, 3, 4, 7, 64};
  for (int k : ks) {
    const size_t n = k == 64 ? 200 : 50, H = 600;
    auto subs = block<uint32_t>(H * (size_t)k);
    auto valid = block<uint8_t>(H);
    auto votes = block<uint32_t>(H);
    for (size_t h = 0; h < H; h++) {
      if (h % 7 == 3 && h > 10) {  // a repeated subset (in another order: the key is the sorted tuple)
        for (int l = 0; l < k; l++) subs[h * k + l] = subs[(h - 5) * k + (k - 1 - l)];
      } else {
        lsqr_sample_subsets(9, h, 1, n, k, subs.get() + h * k);
      }
      valid[h] = (h % 11) != 0;
      votes[h] = (uint32_t)(rng() % (n / 3 + h / 40));
    }
    for (int with_set = 0; with_set < 2; with_set++) {
      uint64_t st[6];
      CHECK(lsqr_replay_init(n, k, 0.99, st) == LSQR_OK);
      void *set = with_set ? lsqr_dedup_create(k) : nullptr;
      // in two pieces, as lsqr_ransac feeds its batches
      size_t used = lsqr_replay(n, k, 0.99, subs.get(), valid.get(), votes.get(), 100, 0, set, st);
      if (used == 100 && !st[5])
        used += lsqr_replay(n, k, 0.99, subs.get() + 100 * (size_t)k, valid.get() + 100, votes.get() + 100, H - 100, 100,
                            set, st);
      if (set) lsqr_dedup_destroy(set);
      // literal loop
      std::set<std::vector<uint32_t>> seen;
      uint64_t tries = orc_choose((unsigned)n, (unsigned)k), best = 0, bestIdx = 0, i = 0;
      bool has = false;
      const double numerator = std::log(1.0 - 0.99);
      for (; i < H && i < tries; i++) {
        std::vector<uint32_t> key(subs.get() + i * k, subs.get() + (i + 1) * k);
        std::sort(key.begin(), key.end());
        const bool fresh = !with_set || seen.insert(key).second;
        if (fresh && valid[i] && votes[i] > best) {
          best = votes[i], bestIdx = i, has = true;
          if (best == n) { i++; break; }
          const double den = std::log(1.0 - std::pow((double)best / (double)n, (double)k));
          const double t = numerator / den + 0.5;
          const unsigned int ti = (t > -2147483649.0 && t < 2147483648.0) ? (unsigned int)(int)t : 0x80000000u;
          tries = std::min<uint64_t>(ti, orc_choose((unsigned)n, (unsigned)k));
        }
      }
      CHECK(st[0] == i);
      CHECK(st[2] == best && (!has || st[3] == bestIdx) && (st[4] != 0) == has);
      CHECK(used <= H);
    }
  }
  CHECK(orc_choose(10000000, 3) == choose_sat(10000000, 3));
  CHECK(choose_sat(5, 7) == 0 && choose_sat(64, 32) == 0xFFFFFFFFu && choose_sat(10, 3) == 120);
}

static orc_cfg ocfg(const lsqr_model_cfg &c) {
  orc_cfg o;
  o.model = c.model;  // the enumerations agree (tests/test_abi.py pins the product's, oracle/lsqr_oracle.h the oracle's)
  o.dim = c.dim;
  o.delta = c.delta;
  o.ls_type = c.ls_type;
  o.aux = c.aux;
  return o;
}

// single-datum host calls with exact-size buffers against the oracle
static void hostCallsTest() {
  struct Case { int model, dim; };
  std::vector<Case> cases;
  for (int d = 2; d <= 8; d++) cases.push_back({LSQR_MODEL_PLANE, d}), cases.push_back({LSQR_MODEL_SPHERE, d}),
                               cases.push_back({LSQR_MODEL_LINE, d});
  for (int d : {1, 2, 5, 8, 9, 13, 16, 17, 31, 33, 63, 64}) cases.push_back({LSQR_MODEL_DENSE, d});
  cases.push_back({LSQR_MODEL_LINE2D, 2});
  cases.push_back({LSQR_MODEL_RAY, 3});
  cases.push_back({LSQR_MODEL_US_SINGLE, 0});
  cases.push_back({LSQR_MODEL_US_POINTER, 0});
  cases.push_back({LSQR_MODEL_PHANTOM, 0});
  for (const Case &cs : cases) {
    lsqr_model_cfg cfg{};
    cfg.model = cs.model;
    cfg.dim = cs.dim;
    cfg.delta = 0.75;
    cfg.ls_type = 0;
    cfg.aux = 0.05;
    const orc_cfg oc = ocfg(cfg);
    const int P = lsqr_num_params(&cfg), ND = lsqr_record_doubles(&cfg), K = lsqr_min_subset(&cfg);
    CHECK(P == orc_num_params(&oc) && ND == orc_record_doubles(&oc) && K == orc_min_subset(&oc));
    if (P <= 0 || ND <= 0) continue;
    for (int rep = 0; rep < 40; rep++) {
      auto par = block<double>((size_t)P);
      auto rec = block<double>((size_t)ND);
      for (int j = 0; j < P; j++) par[j] = U(-2, 2);
      for (int j = 0; j < ND; j++) rec[j] = U(-2, 2);
      if (cs.model == LSQR_MODEL_PLANE || cs.model == LSQR_MODEL_LINE || cs.model == LSQR_MODEL_LINE2D) {
        double nn = 0;  // unit direction first, as estimate() produces it
        const int d = cs.model == LSQR_MODEL_LINE2D ? 2 : cs.dim;
        for (int j = 0; j < d; j++) nn += par[j] * par[j];
        for (int j = 0; j < d; j++) par[j] /= std::sqrt(nn);
      }
      int got = -1;
      const int st = lsqr_agree_host(&cfg, par.get(), rec.get(), &got);
      CHECK(st == LSQR_OK);
      if (st == LSQR_OK) CHECK(got == orc_agree(&oc, par.get(), rec.get()));
    }
    // minimal solves (closed forms only: the others answer LSQR_ERR_INVALID), records at a padded stride
    if (K > 0) {
      const size_t stride = (size_t)ND + 3;
      auto recs = block<double>((size_t)K * stride - 3);  // the last record ends the block
      for (size_t j = 0; j < (size_t)K * stride - 3; j++) recs[j] = U(-50, 50);
      auto out = block<double>((size_t)std::max(P, 1));
      int np = -1;
      const int st = lsqr_estimate_host(&cfg, recs.get(), (size_t)K, stride * sizeof(double), out.get(), &np);
      const bool closed = cs.model != LSQR_MODEL_DENSE && cs.model != LSQR_MODEL_US_SINGLE &&
                          cs.model != LSQR_MODEL_US_POINTER && cs.model != LSQR_MODEL_PHANTOM;
      CHECK(closed ? (st == LSQR_OK || st == LSQR_EMPTY) : st == LSQR_ERR_INVALID);
      if (closed && st == LSQR_OK) {
        std::vector<const double *> ptr;
        for (int l = 0; l < K; l++) ptr.push_back(recs.get() + (size_t)l * stride);
        std::vector<double> want(64);
        const int nw = orc_estimate(&oc, ptr.data(), (size_t)K, want.data());
        CHECK(nw == np);
        // closed forms are bit-exact up to the sign of a null / eigen vector (dimensions > 3 go through an SVD)
        // plane: only d = 3 is a closed form (PlaneParametersEstimator.hxx:48-69; any other d takes the SVD null vector)
        const bool exact = cs.model == LSQR_MODEL_PLANE ? cs.dim == 3
                           : cs.model == LSQR_MODEL_SPHERE ? cs.dim <= 3 : cs.model == LSQR_MODEL_LINE;
        for (int j = 0; j < np && j < nw; j++) {
          if (exact) CHECK(out[j] == want[j] || out[j] == -want[j]);
          else CHECK(std::fabs(std::fabs(out[j]) - std::fabs(want[j])) <= 1e-9 * (1.0 + std::fabs(want[j])));
          if (exact && !(out[j] == want[j] || out[j] == -want[j])) printf("  model %d dim %d param %d: %.17g vs %.17g\n", cs.model, cs.dim, j, out[j], want[j]);
        }
      }
      CHECK(lsqr_estimate_host(&cfg, recs.get(), (size_t)K - 1, stride * sizeof(double), out.get(), &np) != LSQR_OK || !closed);
    }
  }
  lsqr_model_cfg bad{};
  bad.model = LSQR_MODEL_DENSE;
  bad.dim = 65;
  int a;
  double x[1] = {0};
  CHECK(lsqr_agree_host(&bad, x, x, &a) == LSQR_ERR_INVALID);
  CHECK(lsqr_agree_host(nullptr, x, x, &a) == LSQR_ERR_INVALID);
}

// lm_core.h on a small geometric sphere fit (the host side of every iterative fit) against the oracle's lmder
static void lmTest() {
  typedef SphereModel<3> M;
  const size_t n = 500;
  auto pts = block<double>(n * 3);
  std::vector<const double *> ptr;
  for (size_t i = 0; i < n; i++) {
    double u[3] = {N(1), N(1), N(1)}, nn = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (int j = 0; j < 3; j++) pts[3 * i + j] = 10.0 * (j + 1) + 40.0 * u[j] / nn + N(0.3);
    ptr.push_back(pts.get() + 3 * i);
  }
  double x0[4] = {11, 19, 31, 38};
  LmState st;
  lm_init(st, 4, x0, 1e-10, 1e-15, 1e-15, 500, 100.0);
  for (;;) {
    double m[MOM_MAX] = {0};
    for (size_t i = 0; i < n; i++) M::accumulate_lm(pts.get() + 3 * i, st.xtrial, m);
    if (!lm_advance(st, m)) break;
  }
  double want[4];
  int info = 0, nfev = 0;
  orc_sphere_geometric(3, ptr.data(), n, x0, want, &info, &nfev);
  CHECK(st.info >= 1 && st.info <= 4);
  for (int j = 0; j < 4; j++) CHECK(std::fabs(st.x[j] - want[j]) <= 1e-6 * std::fabs(want[j]));
  // the portable sine / cosine of the persistent fit's coefficients: |error| against libm below 2 ulp
  for (int i = 0; i < 100000; i++) {
    const double a = U(-700, 700);
    double s, c;
    lsqr_sincos(a, &s, &c);
    CHECK(std::fabs(s - std::sin(a)) < 4.5e-16 && std::fabs(c - std::cos(a)) < 4.5e-16);
  }
}

// a user-written estimator through RANSAC.h's plugin loop (the reference's advertised extension point), and the
// same data through the oracle's restatement of RANSAC.hxx on the same subset stream
struct UserPoint2D { double x, y; };
class UserLine2D : public lsqrRecipes::ParametersEstimator<UserPoint2D, double> {
 public:
  explicit UserLine2D(double delta) : lsqrRecipes::ParametersEstimator<UserPoint2D, double>(2), d2(delta * delta) {}
  virtual void estimate(std::vector<UserPoint2D *> &d, std::vector<double> &p) {
    p.clear();
    if (d.size() < 2) return;
    const double nx = d[1]->y - d[0]->y, ny = d[0]->x - d[1]->x, nn = std::sqrt(nx * nx + ny * ny);
    if (nn < 1e-12) return;
    p = {nx / nn, ny / nn, d[0]->x, d[0]->y};
  }
  virtual void estimate(std::vector<UserPoint2D> &d, std::vector<double> &p) {
    std::vector<UserPoint2D *> q;
    for (size_t i = 0; i < d.size(); i++) q.push_back(&d[i]);
    estimate(q, p);
  }
  virtual void leastSquaresEstimate(std::vector<UserPoint2D *> &d, std::vector<double> &p) {
    p.clear();
    if (d.size() < 2) return;
    double mx = 0, my = 0, sxx = 0, sxy = 0, syy = 0;
    for (UserPoint2D *q : d) mx += q->x, my += q->y;
    mx /= d.size(), my /= d.size();
    for (UserPoint2D *q : d) sxx += (q->x - mx) * (q->x - mx), sxy += (q->x - mx) * (q->y - my), syy += (q->y - my) * (q->y - my);
    const double th = 0.5 * std::atan2(2 * sxy, sxx - syy);  // direction of largest spread
    p = {-std::sin(th), std::cos(th), mx, my};
  }
  virtual void leastSquaresEstimate(std::vector<UserPoint2D> &d, std::vector<double> &p) {
    std::vector<UserPoint2D *> q;
    for (size_t i = 0; i < d.size(); i++) q.push_back(&d[i]);
    leastSquaresEstimate(q, p);
  }
  virtual bool agree(std::vector<double> &p, UserPoint2D &d) {
    const double s = p[0] * (d.x - p[2]) + p[1] * (d.y - p[3]);
    return s * s < d2;
  }
  double d2;
};

static void pluginTest() {
  using lsqrRecipes::RANSAC;
  std::vector<UserPoint2D> data;
  const double n[2] = {0.6, 0.8}, a[2] = {5, -7};
  for (int i = 0; i < 700; i++) {
    double x = U(-500, 500), y = U(-500, 500);
    if (i % 5 < 3) {
      const double d = (x - a[0]) * n[0] + (y - a[1]) * n[1];
      x += -d * n[0] + N(0.2), y += -d * n[1] + N(0.2);
    }
    data.push_back({x, y});
  }
  UserLine2D user(0.5);
  std::vector<double> p;
  std::vector<bool> cons;
  RANSAC<UserPoint2D, double>::seed() = 77;
  const double f = RANSAC<UserPoint2D, double>::compute(p, &user, data, 0.999, &cons);
  const lsqr_ransac_info info = RANSAC<UserPoint2D, double>::lastInfo();
  CHECK(p.size() == 4 && cons.size() == data.size() && f > 0.55 && f < 0.65);
  if (p.size() == 4) CHECK(std::fabs(std::fabs(p[0] * n[0] + p[1] * n[1]) - 1) < 1e-4);
  // the oracle's RANSAC.hxx on the same records and the same counter-based stream: the Line2D model restates the
  // user's estimator (same estimate / agree arithmetic up to rounding of the normal), so iterations and winner agree
  orc_cfg oc{ORC_LINE2D, 2, 0.5, 0, 0.0};
  orc_ctr_sampler smp{77, 0};
  std::vector<double> flat;
  for (const UserPoint2D &q : data) flat.push_back(q.x), flat.push_back(q.y);
  std::vector<uint8_t> oc_cons(data.size());
  double op[8];
  int onp = 0;
  orc_trace tr{};
  const double of = orc_ransac(&oc, flat.data(), data.size(), 2, 0.999, orc_ctr_sampler_next, &smp, 0, op, &onp,
                               oc_cons.data(), &tr);
  CHECK(std::fabs(of - f) < 0.01);
  CHECK(tr.iters == info.iterations);
  // invalid input returns 0 and leaves the parameters alone (RANSAC.hxx:16-19)
  std::vector<double> keep = {1, 2, 3};
  std::vector<UserPoint2D> one(1);
  CHECK(RANSAC<UserPoint2D, double>::compute(keep, &user, one, 0.999) == 0 && keep.size() == 3);
  CHECK(RANSAC<UserPoint2D, double>::compute(keep, &user, data, 1.0) == 0 && keep.size() == 3);
  // exhaustive overload on a small set, every subset degenerate for part of it
  std::vector<UserPoint2D> few(data.begin(), data.begin() + 14);
  few[3] = few[2];
  std::vector<bool> c2;
  const double fe = RANSAC<UserPoint2D, double>::compute(p, &user, few, &c2);
  CHECK(fe > 0 && c2.size() == few.size());
}

int main() {
  samplerTest();
  replayTest();
  hostCallsTest();
  lmTest();
  pluginTest();
  if (failures) {
    printf("%d checks failed\n", failures);
    return 1;
  }
  printf("sanitizer driver ok\n");
  return 0;
}
