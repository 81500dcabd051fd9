// Host-compiled unit-test driver for the PRODUCT's LSQR_HD math headers
// (lsqrrecipes_amd/csrc/{models,lm_core,small_linalg,sampler,us,phantom}.h).  Built by tests/test_host_math.py
// with g++ -ffp-contract=off; lets the CPU suite check the product's per-model arithmetic, small
// solvers and LM state machine against the oracle before anything runs on a GPU.  Test-only: the
// product never executes this path.
#include <cstdint>
#include <cstring>
#include <vector>

#include "models.h"
#include "phantom.h"
#include "sampler.h"
#include "us.h"

using namespace lsqr;

template <class M>
static int t_estimate(const double *recs, const ModelConsts &mc, double *par) {
  if constexpr (M::IS_US) {
    return -1;  // K1 of the US models is a wave kernel (device only)
  } else {
  double r[M::K][M::ND];
  for (int l = 0; l < M::K; l++)
    for (int j = 0; j < M::ND; j++) r[l][j] = recs[l * M::ND + j];
  return M::estimate(r, mc, par) ? M::P : 0;
  }
}

template <class M>
static int t_ls(const double *data, size_t n, const uint8_t *mask, const double *org,
                const ModelConsts &mc, double *par, double *mom_out) {
  double m[MOM_MAX] = {0};
  for (size_t i = 0; i < n; i++)
    if (!mask || mask[i]) M::accumulate(data + i * M::ND, org, m);
  if (mom_out) memcpy(mom_out, m, sizeof(double) * M::NMOM);
  return M::solve(m, org, mc, par) ? M::P : 0;
}

template <class M>
static int t_lm(const double *data, size_t n, const double *x0, double ftol, double xtol,
                double gtol, int maxfev, double *x, int *info, int *nfev) {
  LmState st;
  lm_init(st, M::NLM, x0, ftol, xtol, gtol, maxfev, 100.0);
  for (;;) {
    double m[MOM_MAX] = {0};
    for (size_t i = 0; i < n; i++) M::accumulate_lm(data + i * M::ND, st.xtrial, m);
    if (!lm_advance(st, m)) break;
  }
  for (int j = 0; j < M::NLM; j++) x[j] = st.x[j];
  *info = st.info;
  *nfev = st.nfev;
  return (st.info >= 1 && st.info <= 4) ? M::NLM : 0;
}

#define DISPATCH(model, dim, CALL)                                   \
  switch (model * 10 + dim) {                                        \
    case 13: { typedef PlaneModel<3> M; CALL; } break;               \
    case 12: { typedef PlaneModel<2> M; CALL; } break;               \
    case 23: { typedef SphereModel<3> M; CALL; } break;              \
    case 22: { typedef SphereModel<2> M; CALL; } break;              \
    case 33: { typedef LineModel<3> M; CALL; } break;                \
    case 32: { typedef LineModel<2> M; CALL; } break;                \
    case 50: { typedef USModel<true> M; CALL; } break;               \
    case 60: { typedef USModel<false> M; CALL; } break;              \
    default: break;                                                  \
  }

static ModelConsts consts(int dim, double delta, int ls) {
  ModelConsts mc;
  mc.delta = delta;
  mc.delta_sq = delta * delta;
  mc.dim = dim;
  mc.ls_type = ls;
  mc.thr = square_threshold(mc.delta_sq);
  mc.absmax = 0.0;
  return mc;
}

extern "C" {

int hm_estimate(int model, int dim, double delta, const double *recs, double *par) {
  ModelConsts mc = consts(dim, delta, 0);
  int r = -1;
  DISPATCH(model, dim, r = t_estimate<M>(recs, mc, par));
  return r;
}

int hm_agree(int model, int dim, double delta, const double *par, const double *data, size_t n,
             uint8_t *mask) {
  ModelConsts mc = consts(dim, delta, 0);
  int ok = 0;
  DISPATCH(model, dim, {
    ok = 1;
    double sp[M::SP];
    for (int j = 0; j < M::SP; j++) sp[j] = j < M::P ? par[j] : 0.0;
    M::prepare(sp, mc);
    for (size_t i = 0; i < n; i++) mask[i] = M::agree(sp, data + i * M::ND, mc) ? 1 : 0;
  });
  return ok;
}

int hm_ls(int model, int dim, double delta, const double *data, size_t n, const uint8_t *mask,
          const double *org, double *par, double *mom_out) {
  ModelConsts mc = consts(dim, delta, 0);
  int r = -1;
  DISPATCH(model, dim, r = t_ls<M>(data, n, mask, org, mc, par, mom_out));
  return r;
}

int hm_us_lm(int single, const double *data, size_t n, const double *x0, double tol, int maxfev,
             double *par, int *info, int *nfev) {
  double x[11];
  int r;
  if (single) {
    r = t_lm<USModel<true>>(data, n, x0, tol, tol, tol, maxfev, x, info, nfev);
    if (r) r = USModel<true>::lm_finalize(x, par);
  } else {
    r = t_lm<USModel<false>>(data, n, x0, tol, tol, tol, maxfev, x, info, nfev);
    if (r) r = USModel<false>::lm_finalize(x, par);
  }
  return r;
}

int hm_sphere_lm(int dim, const double *data, size_t n, const double *x0, double ftol, double xtol,
                 double gtol, int maxfev, double *x, int *info, int *nfev) {
  if (dim == 3) return t_lm<SphereModel<3>>(data, n, x0, ftol, xtol, gtol, maxfev, x, info, nfev);
  if (dim == 2) return t_lm<SphereModel<2>>(data, n, x0, ftol, xtol, gtol, maxfev, x, info, nfev);
  return -1;
}

double hm_square_threshold(double q) { return square_threshold(q); }

// sphere interval thresholds: out = {dlo, dhi}
void hm_sphere_prepare(int dim, double r, double delta, double *out) {
  ModelConsts mc = consts(dim, delta, 0);
  double sp[8] = {0};
  sp[dim] = r;
  if (dim == 3) SphereModel<3>::prepare(sp, mc); else SphereModel<2>::prepare(sp, mc);
  out[0] = sp[dim + 1];
  out[1] = sp[dim + 2];
}

void hm_ctr_subset(uint64_t seed, uint64_t h, uint64_t n, int k, uint32_t *idx) {
  uint32_t sorted[64];
  ctr_subset(seed, h, n, k, idx, sorted);
}

void hm_sym_eig(int n, double *a, double *w, double *v) { sym_eig(n, a, w, v); }

int hm_pinv_solve(int m, int n, double *a, const double *b, double tol, double *x) {
  std::vector<double> s(n), v((size_t)n * n);
  return pinv_solve(m, n, a, n, b, tol, x, s.data(), v.data());
}


// ---- plane phantom (phantom.h): rows, agree(), parameter extraction, both fits from the Gram matrix ----
void hm_phantom_rows(const double *rec, size_t n, double *rows31) {
  for (size_t i = 0; i < n; i++) {
    double x[PhantomModel::ND];
    PhantomModel::load(rec + i * 15, ModelConsts(), x);
    for (int c = 0; c < 31; c++) rows31[i * 31 + c] = PhantomModel::row_entry(x, c);
  }
}
void hm_phantom_agree(const double *par, double delta, const double *rec, size_t n, uint8_t *mask,
                      double *resid) {
  ModelConsts mc = consts(0, delta, 1);
  for (size_t i = 0; i < n; i++) {
    double x[PhantomModel::ND];
    PhantomModel::load(rec + i * 15, mc, x);
    mask[i] = PhantomModel::agree(par, x, mc) ? 1 : 0;
    resid[i] = PhantomModel::residual(par, x, mc);
  }
}
int hm_phantom_finish(const double *x31, double *par41) { return PhantomModel::finish(x31, par41) ? 41 : 0; }
// block: upper triangle of sum a a^T (496) + count; -> 41 parameters or 0
int hm_phantom_fit(const double *block, int iterative, double *par41, int *info, int *nfev, double *cost) {
  PhantomFit f;
  phantom_fit_block(block, iterative != 0, &f);
  *info = f.lm_info;
  *nfev = f.lm_nfev;
  *cost = f.cost;
  for (int j = 0; j < 41; j++) par41[j] = f.params[j];
  return f.ok ? 41 : 0;
}
// fp32 filter block of a hypothesis: f[16] = c0 c1 t3 R1 t1_z 0 tin tout (phantom.h prepare_f32)
void hm_phantom_prepare_f32(const double *par41, double delta, double absmax, double absmax_rot, float *f16) {
  ModelConsts mc = consts(0, delta, 1);
  mc.absmax = absmax;
  mc.absmax_rot = absmax_rot;
  PhantomModel::prepare_f32(par41, mc, f16);
}
// LM block {cost, J^T J upper, J^T f} at x from the full Gram matrix
void hm_phantom_lm_block(const double *G, const double *x11, double *blk78) { phantom_lm_block(G, x11, blk78); }

}  // extern "C"
