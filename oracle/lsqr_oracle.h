/*
 * lsqr_oracle.h -- CPU restatement of the LSQRRecipes RANSAC + least-squares hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (lsqrrecipes_amd/) never
 * links, imports or calls anything in this directory.
 *
 * Every function cites the reference file:line (relative to /root/reference) whose
 * arithmetic it restates.  Reference-owned arithmetic (agree(), the closed-form minimal
 * solves, the LS accumulations, RANSAC.hxx) is restated operation-for-operation in fp64
 * with no FMA contraction, so it is a bit-level oracle for those pieces.
 *
 * PARITY PINNING.  RANSAC.hxx itself is pinned against the real reference: oracle/_ref
 * compiles /root/reference/parametersEstimators/RANSAC.hxx unmodified (ref_driver.cxx) and
 * tests/test_oracle_ransac.py checks this restatement against it on identical rand()
 * streams.  The estimators' third-party numerics live in VNL (VXL/ITK, version unpinned
 * by the reference's CMakeLists.txt:36,58 and absent from /root/reference):
 * vnl_svd / vnl_matrix_inverse (LINPACK dsvdc), vnl_symmetric_eigensystem (EISPACK rs),
 * vnl_levenberg_marquardt (MINPACK lmder).  Those published algorithms are restated here
 * (Jacobi eigen / one-sided Jacobi SVD give the same factorisations up to rounding; lmder
 * is restated step by step) and pinned by the reference's own known-answer vector
 * (testing/DenseLinearEquationSystemParametersEstimatorTest.cxx:162-164), the literature
 * values it quotes (testing/SphereParametersEstimatorTest.cxx:302-308), its tolerance
 * tests, and cross-checks against NumPy/SciPy (scipy.optimize.leastsq wraps the same
 * MINPACK lmder) recorded in tests/golden/.  Bit-level outputs of the VNL routines are
 * therefore "parity unpinned"; everything else is pinned.
 */
#ifndef LSQR_ORACLE_H
#define LSQR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  ORC_PLANE = 1,   /* PlaneParametersEstimator<dim>                     params [n(dim), a(dim)]      */
  ORC_SPHERE = 2,  /* SphereParametersEstimator<dim>                    params [c(dim), r]           */
  ORC_LINE = 3,    /* LineParametersEstimator<dim>                      params [dir(dim), a(dim)]    */
  ORC_DENSE = 4,   /* DenseLinearEquationSystemParametersEstimator<double,n>  params x(n), dim = n  */
  ORC_US_SINGLE = 5, /* SingleUnknownPointTargetUSCalibrationParametersEstimator  20 params         */
  ORC_US_POINTER = 6, /* CalibratedPointerTargetUSCalibrationParametersEstimator  17 params         */
  ORC_ABSOR = 7,   /* AbsoluteOrientationParametersEstimator  params [s,qx,qy,qz,tx,ty,tz]           */
  ORC_PIVOT = 8,   /* PivotCalibrationEstimator               params [DRF^t(3), W^t(3)]              */
  ORC_RAY = 9,     /* RayIntersectionParametersEstimator      params [x,y,z]; record Ray3D = [p(3), n(3)] */
  ORC_LINE2D = 10, /* Line2DParametersEstimator               params [n_x,n_y,a_x,a_y]; record Point2D    */
  ORC_PHANTOM = 11 /* PlanePhantomUSCalibrationParametersEstimator  41 params; record as ORC_US_SINGLE     */
};

enum { ORC_LS_ALGEBRAIC = 0, ORC_LS_GEOMETRIC = 1 }; /* sphere; US: 0 = ANALYTIC, 1 = ITERATIVE */

typedef struct {
  int model;
  int dim;      /* point dimension, or n for ORC_DENSE; ignored for US models */
  double delta; /* constructor argument (NOT squared) */
  int ls_type;
  double aux;   /* ORC_RAY: minimalAngularDeviation (radians) */
} orc_cfg;

/* record layout: every datum is an array of doubles (Point<double,d>: d; AugmentedRow<double,n>:
 * n+1; US single: Frame(12 doubles + int + pad = 13 slots) + Point2D = 15 slots (120 B);
 * US pointer: 18 slots (144 B); absolute orientation: pair<Point3D,Point3D> = 6; pivot: Frame = 13
 * slots (104 B)).  "stride" arguments are in doubles. */
int orc_min_subset(const orc_cfg *c);
int orc_num_params(const orc_cfg *c);
int orc_record_doubles(const orc_cfg *c);

/* The three virtuals of ParametersEstimator.h:41-55.  Return value of estimate/ls = number of
 * parameters written (0 == reference's "empty vector" failure convention). */
int orc_estimate(const orc_cfg *c, const double *const *recs, size_t n, double *params);
int orc_agree(const orc_cfg *c, const double *params, const double *rec);
int orc_ls(const orc_cfg *c, const double *const *recs, size_t n, double *params);
/* convenience: contiguous data, optional mask (NULL = all) */
int orc_ls_masked(const orc_cfg *c, const double *data, size_t n, size_t stride,
                  const uint8_t *mask, double *params);
/* sphere / US helpers exposed like the reference's public methods */
int orc_sphere_algebraic(int dim, const double *const *recs, size_t n, double *params);
int orc_sphere_geometric(int dim, const double *const *recs, size_t n, const double *init,
                         double *params, int *info, int *nfev);
int orc_us_analytic(int model, const double *const *recs, size_t n, double *params);
int orc_us_iterative(int model, const double *const *recs, size_t n, const double *init,
                     double *params, int *info, int *nfev);
int orc_phantom_analytic(const double *const *recs, size_t n, double *params);
int orc_phantom_iterative(const double *const *recs, size_t n, const double *init, double *params,
                          int *info, int *nfev);
/* residual statistics as getDistanceStatistics(): out = {min, max, mean, sumsq} */
int orc_stats(const orc_cfg *c, const double *params, const double *data, size_t n,
              size_t stride, const uint8_t *mask, double out[4]);
/* full scan: mask[i] = agree(params, data[i]); returns the count */
size_t orc_scan(const orc_cfg *c, const double *params, const double *data, size_t n,
                size_t stride, uint8_t *mask);

/* ---- RANSAC.hxx restatement -------------------------------------------------------- */
typedef int (*orc_subset_fn)(void *ctx, size_t n, int k, uint32_t *idx_draw_order);

typedef struct {
  size_t cap;           /* capacity of the per-iteration arrays below (may be 0) */
  size_t iters;         /* loop iterations consumed (RANSAC.hxx:49 "i") */
  size_t evaluated;     /* hypotheses that reached the agree scan */
  uint32_t *votes;      /* per iteration; partial for early-exited losers unless full_scan */
  uint8_t *status;      /* 0 scanned, 1 duplicate subset, 2 degenerate */
  uint32_t *num_tries;  /* numTries after the iteration */
  uint32_t *subsets;    /* cap*k, draw order */
  size_t best_iter;     /* iteration index of the winner */
  uint32_t best_votes;
} orc_trace;

/* probabilistic compute(), RANSAC.hxx:4-145.  subsets come from `next` (return 0 = exhausted,
 * which ends the loop).  full_scan=1 disables the early exit at :94 (never changes the result). */
double orc_ransac(const orc_cfg *c, const double *data, size_t n, size_t stride, double p,
                  orc_subset_fn next, void *next_ctx, int full_scan, double *params,
                  int *nparams, uint8_t *consensus, orc_trace *tr);
/* exhaustive compute(), RANSAC.hxx:150-249 */
double orc_ransac_exhaustive(const orc_cfg *c, const double *data, size_t n, size_t stride,
                             double *params, int *nparams, uint8_t *consensus);
unsigned int orc_choose(unsigned int n, unsigned int m); /* RANSAC.hxx:254-280 */

/* subset providers */
typedef struct { int (*rand_fn)(void *); void *rand_ctx; uint8_t *not_chosen; } orc_ref_sampler;
int orc_ref_sampler_next(void *ctx, size_t n, int k, uint32_t *idx); /* RANSAC.hxx:51-68 */
typedef struct { const uint32_t *subsets; size_t count, pos; } orc_list_sampler;
int orc_list_sampler_next(void *ctx, size_t n, int k, uint32_t *idx);
typedef struct { uint64_t seed, next_index; } orc_ctr_sampler;
int orc_ctr_sampler_next(void *ctx, size_t n, int k, uint32_t *idx);
/* the product's counter-based sampler, restated (lsqrrecipes_amd/csrc/sampler.h) */
void orc_ctr_subset(uint64_t seed, uint64_t hyp_index, size_t n, int k, uint32_t *idx);
/* LCG used to feed rand() deterministically in tests (same code in ref_driver.cxx) */
typedef struct { uint64_t s; } orc_lcg;
int orc_lcg_rand(void *ctx);

/* ---- small dense linear algebra (restating what the reference takes from VNL) -------- */
/* symmetric eigen: A (n*n row-major, destroyed), w ascending, V columns = eigenvectors */
/* AbsoluteOrientation...cxx:208-291; p[i] = [first(3), second(3)], wt[i] >= 0 */
int orc_absor_weighted_ls(const double *const *p, const double *wt, size_t n, double *out);
void orc_us_fcn(int model, const double *const *recs, size_t n, const double *x, double *fvec, double *fjac,
                int iflag);
void orc_sym_eig(int n, double *A, double *w, double *V);
/* thin SVD by one-sided Jacobi: A m*n row-major (m>=n) -> U m*n, s n (descending), V n*n */
void orc_svd(int m, int n, const double *A, double *U, double *s, double *V);
/* x = pinv(A) b with singular values <= tol zeroed (vnl_matrix_inverse + zero_out_absolute);
 * returns rank */
int orc_pinv_solve(int m, int n, const double *A, const double *b, double tol, double *x);
/* MINPACK lmder restated; fcn(iflag=1 -> fvec, iflag=2 -> fjac m*n row-major) */
typedef void (*orc_lm_fcn)(void *ctx, int m, int n, const double *x, double *fvec,
                           double *fjac, int iflag);
int orc_lmder(orc_lm_fcn fcn, void *ctx, int m, int n, double *x, double ftol, double xtol,
              double gtol, int maxfev, double factor, int *nfev, int *njev, double *fnorm_out);

#ifdef __cplusplus
}
#endif
#endif
