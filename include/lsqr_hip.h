/*
 * lsqr_hip.h -- C ABI of the MI355X-native RANSAC + least-squares hot path
 * (liblsqr_hip.so, built from lsqrrecipes_amd/csrc/ with hipcc --offload-arch=gfx950).
 *
 * This is the drop-in boundary for the one hot path of zivy/LSQRRecipes:
 *     RANSAC<T,S>::compute()                      parametersEstimators/RANSAC.h:75-79,111-113
 *       -> ParametersEstimator<T,S>::estimate()   parametersEstimators/ParametersEstimator.h:41-43
 *       -> ParametersEstimator<T,S>::agree()      parametersEstimators/ParametersEstimator.h:55
 *       -> ...::leastSquaresEstimate()            parametersEstimators/ParametersEstimator.h:51-53
 * for the Plane / Sphere / Line / DenseLinearEquationSystem / SinglePointTargetUSCalibration
 * estimators (and AbsoluteOrientation / PivotCalibration).  Plain C types only: no STL, no exceptions, no torch types cross this ABI.  Every
 * entry point returns an lsqr_status; LSQR_OK == 0.  The reference's "empty parameters vector"
 * failure convention (RANSAC.h:54-63) maps to LSQR_EMPTY (a normal outcome, not an error).
 *
 * The C++ header shim (lsqrrecipes_amd/include/RANSAC.h, ...Estimator.h) and the Python mirror
 * (lsqrrecipes_amd/ python modules) are thin callers of exactly these functions.  There is no CPU
 * fallback behind this ABI: without a usable HIP device lsqr_ctx_create fails with
 * LSQR_ERR_NO_DEVICE.
 *
 * All records are fp64, array-of-structures, exactly as the reference lays them out:
 *   Point<double,d>            d doubles                         common/Point.h:127
 *   AugmentedRow<double,n>     n+1 doubles (aValues[n], bValue)  .../DenseLinear...Estimator.h:133-134
 *   SingleUnknown DataType     15 slots = Frame{rotation[3][3], translation[3], int outputFormat
 *                              (+4 B pad)} + Point2D             .../SinglePointTarget...h:45-48,
 *                                                                common/Frame.h:30-31,41
 *   CalibratedPointer DataType 18 slots (+ Point3D p)            .../SinglePointTarget...h:335-339
 *   pair<Point3D,Point3D>      6 doubles (first, second)         .../AbsoluteOrientation...h:14-15
 *                              (ls_type 2 = weightedLeastSquaresEstimate, .h:86: 7 doubles, slot 6 = weight)
 *   Frame                      13 slots (104 B)                  common/Frame.h:30-31,41
 *   Ray3D                      6 doubles (Point3D p, Vector3D n) common/Ray3D.h:23-24
 * A caller's std::vector<T> is passed as (pointer, count, stride in bytes) without repacking.
 */
#ifndef LSQR_HIP_H
#define LSQR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define LSQR_API __attribute__((visibility("default")))
#else
#define LSQR_API
#endif

typedef enum {
  LSQR_OK = 0,
  LSQR_EMPTY = 1,          /* reference's empty-vector outcome: degenerate data / LS failed */
  LSQR_ERR_INVALID = 2,    /* bad argument */
  LSQR_ERR_NO_DEVICE = 3,  /* no HIP device / runtime unusable */
  LSQR_ERR_HIP = 4,        /* a HIP call failed; see lsqr_last_error */
  LSQR_ERR_STATE = 5       /* call order (no data uploaded, no hypotheses, ...) */
} lsqr_status;

typedef enum {
  LSQR_MODEL_PLANE = 1,      /* PlaneParametersEstimator<dim>       params [n(dim), a(dim)]   */
  LSQR_MODEL_SPHERE = 2,     /* SphereParametersEstimator<dim>      params [c(dim), r]        */
  LSQR_MODEL_LINE = 3,       /* LineParametersEstimator<dim>        params [dir(dim), a(dim)] */
  LSQR_MODEL_DENSE = 4,      /* DenseLinearEquationSystemParametersEstimator<double,n>, dim=n */
  LSQR_MODEL_US_SINGLE = 5,  /* SingleUnknownPointTargetUSCalibrationParametersEstimator      */
  LSQR_MODEL_US_POINTER = 6, /* CalibratedPointerTargetUSCalibrationParametersEstimator       */
  LSQR_MODEL_ABSOR = 7,      /* AbsoluteOrientationParametersEstimator  params [s,qx,qy,qz,t(3)] */
  LSQR_MODEL_PIVOT = 8,      /* PivotCalibrationEstimator               params [DRF^t(3), W^t(3)] */
  LSQR_MODEL_RAY = 9,        /* RayIntersectionParametersEstimator      params [x, y, z]          */
  LSQR_MODEL_LINE2D = 10,    /* Line2DParametersEstimator               params [n_x, n_y, a_x, a_y] */
  LSQR_MODEL_PHANTOM = 11    /* PlanePhantomUSCalibrationParametersEstimator  31 frames, 41 params */
} lsqr_model;

/* SphereParametersEstimator::LeastSquaresType (SphereParametersEstimator.h:28) and the US
 * estimators' {ANALYTIC, ITERATIVE} (SinglePointTarget...h:57) */
enum { LSQR_LS_ALGEBRAIC = 0, LSQR_LS_GEOMETRIC = 1, LSQR_LS_ANALYTIC = 0, LSQR_LS_ITERATIVE = 1 };

/* What the reference passes to an estimator's constructor / setters. */
typedef struct {
  int32_t model;   /* lsqr_model */
  int32_t dim;     /* point dimension (plane/sphere/line: 2 or 3), n for DENSE (1..64) */
  double delta;    /* constructor argument, NOT squared (PlaneParametersEstimator.hxx:13-17) */
  int32_t ls_type; /* sphere / US; ABSOR: 2 = records carry a weight in slot 6 (weighted fit) */
  int32_t reserved;
  double aux;      /* RAY: minimalAngularDeviation in radians (RayIntersection...Estimator.h:34-35);
                      unused by the other models */
} lsqr_model_cfg;

typedef struct lsqr_ctx lsqr_ctx; /* owns a device, a stream and all device buffers */

/* Result block of a final fit (leastSquaresEstimate). */
typedef struct {
  int32_t n_params;     /* 0 when status is LSQR_EMPTY */
  int32_t lm_info;      /* MINPACK info code of the LM run (0 when no LM) */
  int32_t lm_nfev;      /* LM function evaluations (= device passes) */
  int32_t reserved;     /* LM runs: the evaluation after which the cost never again fell by more than 1e-7
                           relative (diagnostics: where a run that ends at the evaluation limit stopped gaining).
                           Dense system: 1 = the elimination on the normal equations was refused and the fit was taken
                           from the rows in double-double (the reference's SVD route, csrc/dense.h); 2 = only the summed
                           block was at hand (lsqr_solve_moments, lsqr_step_finish*) and a pivot fell below 1e-6 max|G|:
                           the result carries eps cond(A)^2 -- a caller that holds the rows should fit again from them
                           (lsqr_mask + lsqr_ls_fit; lsqr_multi_* and distributed.ShardedRansac do, so that an N-GPU
                           fit equals the one-GPU fit on ill-conditioned systems too) */
  uint64_t n_used;      /* observations that entered the fit */
  double cost;          /* final sum of squared residuals where the model defines one, else 0 */
} lsqr_fit_info;

/* Outcome of RANSAC<T,S>::compute(). */
typedef struct {
  double fraction;        /* return value: |consensus| / N (RANSAC.hxx:144) */
  uint64_t iterations;    /* loop iterations consumed, duplicates and degenerates included */
  uint64_t evaluated;     /* hypotheses scanned on the device (>= the serial count) */
  uint64_t best_index;    /* iteration index of the winner */
  uint32_t best_votes;
  int32_t n_params;       /* 0 -> parameters empty */
  lsqr_fit_info fit;
} lsqr_ransac_info;

/* ---- library / device -------------------------------------------------------------------- */
LSQR_API const char *lsqr_version(void);
LSQR_API int lsqr_device_count(int *count);
LSQR_API const char *lsqr_status_string(int status);

/* ---- context ------------------------------------------------------------------------------ */
LSQR_API int lsqr_ctx_create(int device, lsqr_ctx **out);
LSQR_API void lsqr_ctx_destroy(lsqr_ctx *ctx);
LSQR_API const char *lsqr_last_error(const lsqr_ctx *ctx);
LSQR_API int lsqr_synchronize(lsqr_ctx *ctx);

/* ---- model description (host only) -------------------------------------------------------- */
LSQR_API int lsqr_min_subset(const lsqr_model_cfg *cfg);     /* numForEstimate() */
LSQR_API int lsqr_num_params(const lsqr_model_cfg *cfg);     /* length of the parameters vector */
LSQR_API int lsqr_record_doubles(const lsqr_model_cfg *cfg); /* tight record size in doubles */
LSQR_API int lsqr_set_model(lsqr_ctx *ctx, const lsqr_model_cfg *cfg);

/* ---- observations -------------------------------------------------------------------------- */
/* Copies `count` host records (std::vector<T>::data(), stride sizeof(T)) to the device.
 * Replaces: the `std::vector<T> &data` argument of RANSAC.h:75-79. */
LSQR_API int lsqr_upload(lsqr_ctx *ctx, const void *host_records, size_t count,
                         size_t stride_bytes);
/* Adopts records already resident in device memory (no copy; caller keeps ownership). */
LSQR_API int lsqr_attach(lsqr_ctx *ctx, const void *device_records, size_t count,
                         size_t stride_bytes);
LSQR_API size_t lsqr_count(const lsqr_ctx *ctx);

/* ---- hypotheses: minimal-subset solve (estimate(), ParametersEstimator.h:41) ---------------- */
/* Explicit subsets: H tuples of k record indices in DRAW order (RANSAC.hxx:65). */
LSQR_API int lsqr_hypotheses_from_subsets(lsqr_ctx *ctx, const uint32_t *subsets, size_t H);
/* Counter-based device sampler: hypothesis i uses stream element first_index + i of `seed`
 * (replaces RANSAC.hxx:51-68; same "rank-th not yet chosen" selection rule, O(k^2) not O(N)).
 * subsets_out (nullable) receives the H*k indices. */
LSQR_API int lsqr_hypotheses_sample(lsqr_ctx *ctx, uint64_t seed, uint64_t first_index, size_t H,
                                    uint32_t *subsets_out);
/* The same sampler stream evaluated on the HOST (no context, no device): subsets_out receives hypotheses
 * [first_index, first_index + H) of stream `seed` for n observations and subsets of k (1..64) indices, draw
 * order.  Used by RANSAC<T,S>::compute() for user-defined estimators without a device model (the plugin path
 * of lsqrrecipes_amd/include/RANSAC.h), so that host and device paths walk one subset stream. */
LSQR_API int lsqr_sample_subsets(uint64_t seed, uint64_t first_index, size_t H, uint64_t n, int k,
                                 uint32_t *subsets_out);
/* ---- single-datum calls, evaluated on the HOST (no context, no device) --------------------------------
 * ParametersEstimator<T,S>::agree(parameters, datum) (ParametersEstimator.h:55; a ten-flop inline in the reference,
 * PlaneParametersEstimator.hxx:196-203) and estimate() of one minimal subset (ParametersEstimator.h:41-43) for the
 * closed-form models: the library's own per-model code (the code the kernels run, compiled for the host; bit-identical
 * results) without an upload and a kernel launch per call.  record(s): the caller's record(s) as laid out for
 * lsqr_upload.  lsqr_estimate_host: `count` >= lsqr_min_subset records in draw order; returns LSQR_EMPTY
 * (*n_params_out = 0) for a degenerate subset.  Both return LSQR_ERR_INVALID when the model has no host form for
 * the call (minimal solves of the dense system, the US calibrations and the plane phantom are device kernels). */
LSQR_API int lsqr_agree_host(const lsqr_model_cfg *cfg, const double *params, const void *record, int *agree_out);
LSQR_API int lsqr_estimate_host(const lsqr_model_cfg *cfg, const void *records, size_t count, size_t stride_bytes,
                                double *params_out, int *n_params_out);

/* ---- agree() scan over all observations (RANSAC.hxx:94-99, without the early exit) ---------- */
LSQR_API int lsqr_scan(lsqr_ctx *ctx);
/* Results of the current batch: any pointer may be NULL.  params: H*P doubles; valid: H bytes
 * (0 = degenerate subset, estimate() returned an empty vector); votes: H counts. */
LSQR_API int lsqr_get_hypotheses(lsqr_ctx *ctx, double *params, uint8_t *valid, uint32_t *votes);
LSQR_API size_t lsqr_num_hypotheses(const lsqr_ctx *ctx);
/* parameters (P doubles) and validity of one hypothesis of the current batch */
LSQR_API int lsqr_get_hypothesis(lsqr_ctx *ctx, size_t hypothesis, double *params, uint8_t *valid);
/* First-max winner of the current batch packed as (votes << 32) | (0xFFFFFFFF - index), so that a
 * max-reduction over ranks (RCCL all-reduce) keeps the earliest best hypothesis (RANSAC.hxx:100). */
LSQR_API int lsqr_best(lsqr_ctx *ctx, uint64_t *packed);

/* ---- consensus mask (RANSAC.hxx:129-137) ----------------------------------------------------- */
/* mask[i] = agree(params, data[i]) for i in [begin, end); kept on the device for lsqr_ls_fit.
 * mask_out (nullable) receives end-begin bytes.  count_out (nullable) the number of inliers. */
LSQR_API int lsqr_mask(lsqr_ctx *ctx, const double *params, size_t begin, size_t end,
                       uint8_t *mask_out, uint64_t *count_out);
LSQR_API int lsqr_mask_from_hypothesis(lsqr_ctx *ctx, size_t hypothesis, uint8_t *mask_out,
                                       uint64_t *count_out);
LSQR_API int lsqr_set_mask(lsqr_ctx *ctx, const uint8_t *mask); /* N bytes from the host */

/* ---- final fit (leastSquaresEstimate(), ParametersEstimator.h:51) ---------------------------- */
/* use_mask = 0 fits all observations, 1 only those in the device mask.  LSQR_EMPTY = the reference's empty
 * vector (info->n_params == 0); after a failed Levenberg-Marquardt run (info->lm_info outside 1..4) params_out
 * still receives the last iterate, for diagnostics only. */
LSQR_API int lsqr_ls_fit(lsqr_ctx *ctx, int use_mask, double *params_out, lsqr_fit_info *info);
/* Building blocks for multi-GPU fits: the normal-equation / moment block of [begin,end) and the
 * small solve from a (summed) block.  lsqr_moments_len gives the block length in doubles. */
LSQR_API int lsqr_moments_len(const lsqr_model_cfg *cfg, int phase);
/* phase 0: the model's linear LS block about the origin x (ND doubles: any point near the model,
 * identical on every rank); phase 1: the LM block {sum f^2, J^T J, J^T f} at the trial point x. */
LSQR_API int lsqr_moments(lsqr_ctx *ctx, int use_mask, size_t begin, size_t end, int phase,
                          const double *x, double *block_out);
/* Small solve (on the device) from a summed phase-0 block taken about `origin`. */
LSQR_API int lsqr_solve_moments(lsqr_ctx *ctx, const double *block, const double *origin,
                                double *params_out, lsqr_fit_info *info);
/* Multi-GPU step, per rank, one host synchronisation: hypothesis `stream_index` of sampler stream
 * `seed` (the global winner, re-derived locally -- the sampler is stateless) -> consensus mask over
 * [begin, end) -> phase-0 block of that slice about the model's own point (plane / line: a, sphere: c,
 * else zeros).  params_out: lsqr_num_params doubles; origin_out: 32 doubles (pass it to
 * lsqr_solve_moments together with the summed block); block_out: lsqr_moments_len(cfg, 0) doubles;
 * count_out: inliers of the slice.  LSQR_EMPTY if the subset is degenerate. */
LSQR_API int lsqr_winner_moments(lsqr_ctx *ctx, uint64_t seed, uint64_t stream_index, size_t begin,
                                 size_t end, double *params_out, double *origin_out,
                                 double *block_out, uint64_t *count_out);
/* lsqr_moments with the block left in device memory and no synchronisation (the caller all-reduces it in
 * place on the context's stream, see lsqr_set_stream, and reads it back once).  x is staged in one of four
 * pinned slots, each guarded by an event: calls may be issued back to back. */
LSQR_API int lsqr_moments_dev(lsqr_ctx *ctx, int use_mask, size_t begin, size_t end, int phase,
                              const double *x, double *block_dev);
/* Levenberg-Marquardt over summed phase-1 blocks (MINPACK lmder control flow on the device):
 *   lsqr_lm_begin(x0) -> x_trial;  repeat { block = sum over ranks of lsqr_moments(phase 1,
 *   x_trial);  lsqr_lm_step(block) -> cont, x_trial }  until !cont.  On the last step params_out
 *   receives the full parameter vector (status LSQR_EMPTY if LM did not converge).
 *   LSQR_MODEL_PHANTOM: every residual is linear in 31 functions of the parameters, so the phase-1
 *   block is the phase-0 Gram block (independent of x_trial) and the first lsqr_lm_step runs the whole
 *   minimisation from x0 on it (cont = 0). */
LSQR_API int lsqr_lm_begin(lsqr_ctx *ctx, const double *x0, double *x_trial_out);
LSQR_API int lsqr_lm_step(lsqr_ctx *ctx, const double *block, double *x_trial_out, int *cont,
                          double *params_out, lsqr_fit_info *info);
/* residual statistics as getDistanceStatistics() (SphereParametersEstimator.hxx:341-377):
 * out = {min, max, mean, sum of squares} */
LSQR_API int lsqr_stats(lsqr_ctx *ctx, const double *params, int use_mask, double out[4]);
/* the residual of every record in [begin, end) under `params`, in record order: the `distances`
 * vector of PlanePhantomUSCalibrationParametersEstimator::getDistanceStatistics (.cxx:455-549);
 * defined for every model (the quantity lsqr_stats summarises).  out: end - begin doubles (host). */
LSQR_API int lsqr_residuals(lsqr_ctx *ctx, const double *params, size_t begin, size_t end,
                            double *out);

/* ---- whole path: RANSAC<T,S>::compute() ------------------------------------------------------ */
/* Probabilistic overload (RANSAC.h:75-79).  Subsets come from `subsets` (n_subsets tuples, draw
 * order) when non-NULL, else from the device sampler stream `seed`.  Hypotheses are evaluated in
 * batches on the device and the serial loop of RANSAC.hxx:49-117 (duplicate filter, adaptive
 * numTries, strict-> best update) is replayed over the batch results, so the outcome equals
 * the serial reference's for the same subset stream.  params_out: lsqr_num_params doubles;
 * consensus_out (nullable): N bytes.  Invalid input (N < k, p outside (0,1)) returns
 * LSQR_ERR_INVALID with info->fraction = 0 and params_out untouched (RANSAC.hxx:16-19). */
LSQR_API int lsqr_ransac(lsqr_ctx *ctx, double p, uint64_t seed, const uint32_t *subsets,
                         size_t n_subsets, double *params_out, uint8_t *consensus_out,
                         lsqr_ransac_info *info);
/* One fixed-size batch of the same loop without the adaptive stopping rule: hypotheses
 * [first_index, first_index + H) of the sampler stream `seed` are solved and scanned, the first
 * hypothesis with the maximal vote count wins (the strict '>' of RANSAC.hxx:100), its consensus set
 * (RANSAC.hxx:129-137) is fitted (leastSquaresEstimate, :138).  All device work is chained on the
 * context's stream; the host synchronises once.  info->iterations = H, info->best_index is the
 * stream index of the winner.  Returns LSQR_EMPTY when no hypothesis was valid or the fit failed. */
LSQR_API int lsqr_batch_fit(lsqr_ctx *ctx, uint64_t seed, uint64_t first_index, size_t H,
                            double *params_out, uint8_t *consensus_out, lsqr_ransac_info *info);
/* lsqr_batch_fit split in two for pipelining: _enqueue chains the whole batch (sample .. fit and the copies
 * of the results into a pinned slot) and returns without waiting; _wait blocks on that slot's event and returns
 * what lsqr_batch_fit returns (no consensus copy: read it with lsqr_mask_* before the slot's lane gets its next
 * batch if it is needed).  The next batches can be enqueued before the previous ones are read: the host's latency
 * between steps disappears behind the device's work.
 * Slots and lanes: the context runs L = option "batch_lanes" (1..4, default 4) LANES -- HIP streams with their own
 * hypothesis / vote / mask buffers and their own spatial index, all reading the context's records -- and has 2 L
 * slots: slot s is batch depth s / L (0 or 1) of lane s % L.  Batches on one lane execute in order; batches on
 * different lanes OVERLAP on the device, which fills the time the dozen one-workgroup kernels of a batch (selection,
 * winner, solve) and the tails of the large ones leave the chip idle: plane, 10 M points, 4096 hypotheses per
 * batch: 0.81 ms per batch on one lane, 0.64 on two, 0.56 on four.  A caller that only uses slots 0 and 1 gets two
 * lanes.  Results do not depend on the number of lanes (tests/test_gpu_parity.py::test_batch_lanes_*).  Replacing the
 * records or the model (lsqr_upload / lsqr_attach / lsqr_set_model) waits for the lanes and voids unread slots.
 * Closed-form fits only (LSQR_ERR_INVALID for the iterative fits and the plane phantom, whose final fit keeps the
 * host in the loop). */
LSQR_API int lsqr_batch_fit_enqueue(lsqr_ctx *ctx, uint64_t seed, uint64_t first_index, size_t H, int slot);
LSQR_API int lsqr_batch_fit_wait(lsqr_ctx *ctx, int slot, double *params_out, lsqr_ransac_info *info);
/* ---- multi-GPU step with device-resident exchange buffers ------------------------------------------
 * One rank's part of a step of `world` x H hypotheses, all device work chained on the context's stream and
 * ONE host synchronisation (lsqr_step_finish); the two exchanges are collectives on the caller's device
 * buffers between the calls (RCCL: all-reduce MAX of packed_dev[0] as int64, all-reduce SUM of block_dev):
 *   lsqr_step_scan    hypotheses [first, first + H) of the stream: sample, solve, scan; packed_dev[0] =
 *                     (votes << 32) | (0xFFFFFFFF - (index_base + index)) of the first best one, 0 if none
 *   lsqr_step_winner  the winner (stream index batch_first + in-batch index taken from packed_dev) is
 *                     re-derived on the device, its consensus mask taken over observations [begin, end)
 *                     and block_dev[0 .. len) = phase-0 moment block of the slice, block_dev[len] = its
 *                     inlier count (len = lsqr_moments_len(cfg, 0))
 *   lsqr_step_finish  final fit from the (summed) block; winner_out: the winner's lsqr_num_params
 *                     parameters; info->best_votes / best_index (in-batch index) from packed_dev,
 *                     info->evaluated = 1 when the batch had a valid hypothesis (else 0),
 *                     info->fit.n_used = the summed count.  LSQR_EMPTY: no valid hypothesis / fit failed.
 * Host synchronisations of a step: lsqr_step_finish / _wait is the one every model has.  The dense system and the
 * US / plane-phantom calibrations add ONE inside lsqr_step_scan: their matrix-core filters decide the band of every
 * batch from per-workgroup worklists, and the host reads the fullest segment's fill right after the scan so that an
 * overflow (never seen) re-runs it on the exact kernels before the winner is packed -- the packed winner feeds a
 * collective and cannot be taken back afterwards.  lsqr_batch_fit_enqueue, whose results stay local until
 * lsqr_batch_fit_wait, defers that check to the wait and never blocks (r05).
 * lsqr_set_stream(external = 1) makes the context enqueue on the caller's HIP stream (the stream the
 * collectives are ordered with, e.g. torch's current stream; NULL = the default stream); external = 0
 * restores the context's own (non-blocking) stream. */
LSQR_API int lsqr_set_stream(lsqr_ctx *ctx, void *hip_stream, int external);
LSQR_API int lsqr_step_scan(lsqr_ctx *ctx, uint64_t seed, uint64_t first, size_t H, uint32_t index_base,
                            uint64_t *packed_dev);
LSQR_API int lsqr_step_winner(lsqr_ctx *ctx, uint64_t seed, uint64_t batch_first,
                              const uint64_t *packed_dev, size_t begin, size_t end, double *block_dev);
LSQR_API int lsqr_step_finish(lsqr_ctx *ctx, const uint64_t *packed_dev, const double *block_dev,
                              double *winner_out, double *params_out, lsqr_ransac_info *info);
/* lsqr_step_finish in two halves (as lsqr_batch_fit_enqueue / _wait): the solve and the copies of the step's
 * results into pinned slot `slot` (0 or 1) are enqueued, the next step can be enqueued on the same exchange
 * buffers (stream order keeps this step's copies ahead of the next step's writes), and _wait blocks on the
 * slot's event.  The phantom's host-side solve runs in _wait and uses one staging area: do not pipeline it. */
LSQR_API int lsqr_step_finish_enqueue(lsqr_ctx *ctx, const uint64_t *packed_dev, const double *block_dev,
                                      int slot);
LSQR_API int lsqr_step_finish_wait(lsqr_ctx *ctx, int slot, double *winner_out, double *params_out,
                                   lsqr_ransac_info *info);
/* ---- several devices from one process ------------------------------------------------------------------
 * north_star: "hypothesis batches shard trivially across the 8 GPUs of one node".  One lsqr_ctx per entry of
 * `devices` (an entry may repeat: several contexts on one device -- the tests run two on the box's one GPU).
 * Observations are uploaded to the first device once and replicated device-to-device (peer copies: xGMI inside a
 * node); a batch of n * H hypotheses is scanned in n contiguous slices; the earliest best hypothesis is picked by
 * a max over the packed (votes, ~index) words and its consensus mask + moment block are taken slice-wise and
 * summed in rank order -- both exchanges are peer copies into the first device's gather area (8 B and <= 17 KB:
 * latency-bound either way) followed by a reduction kernel there; between PROCESSES (bench.py --gpus N, one rank
 * per GPU) the same two exchanges are RCCL all-reduces.  Results: winner, consensus set and iteration count equal
 * the single-device entry points bit for bit; the final fit agrees to rounding (different summation tree). */
typedef struct lsqr_multi lsqr_multi;
LSQR_API int lsqr_multi_create(const int *devices, int n, lsqr_multi **out);
LSQR_API void lsqr_multi_destroy(lsqr_multi *m);
/* The transport of the handle's two exchanges: "peer-copy" (default: hipMemcpyPeerAsync into the first device's gather
 * area + a fixed-order reduction kernel there) or "rccl" (environment LSQR_MULTI_TRANSPORT=rccl at lsqr_multi_create:
 * one RCCL communicator per device -- ncclCommInitAll; librccl is dlopen'ed --, the winner as ncclAllReduce MAX of the
 * packed 64-bit word, the moment blocks as ncclAllReduce SUM on fixed buffers, each on its context's stream; needs
 * DISTINCT devices, lsqr_multi_create fails with LSQR_ERR_INVALID otherwise).  *bringup_seconds (nullable): what
 * creating and proving the communicators took. */
LSQR_API const char *lsqr_multi_transport(const lsqr_multi *m, double *bringup_seconds);
LSQR_API int lsqr_multi_size(const lsqr_multi *m);
LSQR_API lsqr_ctx *lsqr_multi_ctx(lsqr_multi *m, int rank); /* per-device options / profiling */
LSQR_API const char *lsqr_multi_last_error(const lsqr_multi *m);
LSQR_API int lsqr_multi_set_model(lsqr_multi *m, const lsqr_model_cfg *cfg);
LSQR_API int lsqr_multi_upload(lsqr_multi *m, const void *host_records, size_t count, size_t stride_bytes);
/* lsqr_batch_fit over n devices: hypotheses [first, first + n * H_per_device) of the stream; info->best_index is
 * the stream index of the winner */
LSQR_API int lsqr_multi_batch_fit(lsqr_multi *m, uint64_t seed, uint64_t first_index, size_t H_per_device,
                                  double *params_out, uint8_t *consensus_out, lsqr_ransac_info *info);
/* lsqr_ransac (RANSAC<T,S>::compute, probabilistic overload) over n devices, device sampler stream `seed` */
LSQR_API int lsqr_multi_ransac(lsqr_multi *m, double p, uint64_t seed, double *params_out,
                               uint8_t *consensus_out, lsqr_ransac_info *info);

/* Exhaustive overload (RANSAC.h:111-113): all C(N,k) subsets in lexicographic order. */
LSQR_API int lsqr_ransac_exhaustive(lsqr_ctx *ctx, double *params_out, uint8_t *consensus_out,
                                    lsqr_ransac_info *info);
/* Host-side replay of RANSAC.hxx:79-112 over batch results (exposed for tests and for the
 * multi-GPU driver).  state is 6 uint64: {i, numTries, best_votes, best_index, has_best, done};
 * initialise with lsqr_replay_init.  Returns the number of batch entries consumed. */
LSQR_API int lsqr_replay_init(size_t n, int k, double p, uint64_t state[6]);
LSQR_API size_t lsqr_replay(size_t n, int k, double p, const uint32_t *subsets,
                            const uint8_t *valid, const uint32_t *votes, size_t H,
                            uint64_t base_index, void *dedup_set, uint64_t state[6]);
LSQR_API void *lsqr_dedup_create(int k);
LSQR_API void lsqr_dedup_destroy(void *set);

/* ---- tuning knobs (A/B measurements inside one process; defaults are the tuned values) --------- */
/* "scan_ppl": observations per lane in k_scan (2, 4 or 8; 0 = model default);
 * "scan_filter": 1 = cheap pre-filter with a proven error band + exact fp64 re-evaluation of the
 *                ambiguous observations (bit-identical votes): packed fp32 for plane / sphere / line
 *                and the US estimators, fp64 MFMA for the dense system; 0 = plain fp64 scan; 2 / 3 =
 *                as 1 with the re-evaluation forced per packed pair / per tile (US: 2 = the fused
 *                fp64 filter);
 * "upload_threads": host threads lsqr_upload uses to stage a large pageable buffer through pinned chunks
 *                while earlier chunks are already in flight (0 = one plain hipMemcpy: the default, measured
 *                faster on the MI355X box -- 56 GB/s; -1 = LSQR_UPLOAD_THREADS or 0);
 * "max_iterations": stop lsqr_ransac after this many loop iterations even if the adaptive bound
 *                asks for more (0 = the reference's behaviour: up to C(N,k));
 * "lm_persist": 1 (default) = an iterative US fit (SinglePointTarget...Estimator.cxx:272-329, :926-971) is ONE launch
 *               when this context is the only one of the process that has been fitting on the device lately: the
 *               workgroups stay resident, every evaluation's pass over the compacted consensus set, the sum of the block
 *               partials and the hand-over of the next trial point happen inside it (csrc/lm_persist.h), MINPACK's step
 *               runs on the host between tagged granules in pinned memory (measured r05: 26 against 30 us per
 *               evaluation); with several contexts fitting at once the launch path is taken (their passes overlap each
 *               other's host round trips: the better aggregate); 3 = the persistent kernel always; 2 = persistent with
 *               the step on the device too (no host in the loop at all: 230 us per step on one lane -- measured, kept
 *               for the record); 0 = two launches per evaluation (r02-r04).  Same iterates, lm_info and lm_nfev every way.  "lm_persist_wgs": resident workgroups (0 = the device's compute units divided
 *               by the contexts alive on it, at most four ways); "lm_persist_timeout_ms": bound of every wait inside
 *               the kernel (2000) -- when it expires the fit is run again on the launch path;
 * "lm_host": 1 (default) = the Levenberg-Marquardt control flow between device passes runs on the
 *                host, 0 = in a single-lane device kernel (same lm_core.h code either way);
 * "scan_index":  two-level scan of the point models over a spatial index of the observations
 *                (Morton-sorted copy, re-partitioned as a k-d tree inside runs of 8192 records, + one fp32 bounding
 *                box per cell of 256 / 512 observations, built on the device once per upload): 1 (default) = built
 *                when it pays -- the upload holds >= 65536 observations and >= 768 hypotheses have been scanned on
 *                it or are still announced by the adaptive bound (the build costs about as much as 700 exhaustive
 *                hypothesis scans) --,
 *                0 = never, 2 = always.  Votes are
 *                bit-identical either way;  "scan_cell": observations per cell (128, 256 or 512; 0 = the
 *                model's default), "scan_cpt": cells per wave tile (1, 2 or 4; 0 = default), "scan_block":
 *                workgroup size (256 / 1024), "scan_hsplit": hypothesis segments per tile (A/B knobs);
 * "scan_bound":  1 (default) = the batch entry points (lsqr_batch_fit*, lsqr_step_scan, lsqr_ransac) over an indexed
 *                upload count only hypotheses that can still become the running maximum (an upper bound on every
 *                hypothesis' votes from the cell boxes, a few early candidates counted first); the others
 *                report 0 votes -- winner, consensus set and iteration count are unchanged (RANSAC.hxx:94 abandons
 *                exactly such hypotheses).  Dense system (n > 32) and US calibrations: the observations are scanned in
 *                chunks and a hypothesis stops being counted once it cannot win any more (lsqr_scan_work; it reports
 *                its partial count).  0 = every hypothesis is counted.  lsqr_scan always counts all;
 * "scan_refine": 1 (default) = the index build re-partitions every run of 8192 Morton-ordered records as a k-d tree
 *                (median splits along the widest extent; csrc/cells.h: k_refine_runs) so that cells are compact:
 *                15 % (plane) to 40 % (sphere) fewer (hypothesis, cell) pairs reach the second level of the scan;
 *                0 = plain Morton runs (A/B knob).  "scan_presorted": 1 = cells are runs of the UPLOAD order (experiments
 *                with other spatial orders: tools/ab_order_kd.py).  Votes are identical whatever the order;
 * "scan_kd_levels": k-d levels ABOVE those runs, each one radix sort of (segment, coordinate) pairs (default 7: runs of
 *               1 M records; 0 = the Morton order above the runs, as until r05).  5 - 12 % fewer surviving pairs for
 *               ~0.6 ms per level and 10 M records; built by the first scan after the upload has been asked to scan
 * "scan_kd_after" hypotheses (default 16384; 0 = with the first index) -- the first index of an upload keeps the
 *               cheaper order.  Votes do not depend on either.
 * "batch_lanes": streams (1..4, default 4) the slots of lsqr_batch_fit_enqueue / _wait are spread over (see there);
 * "scan_pairs":  plain (unbounded) scans of an indexed upload: 0 (default) = the model's measured choice (plane, batches
 *                of >= 1024: the statically balanced kernel of the bounded scan, k_scan_pairs; else k_scan_cells),
 *                1 = always k_scan_pairs, 2 = always k_scan_cells (A/B knob); "scan_pairs_waves": workgroups
 *                per CU of that kernel (0 = what fits); "scan_bound_merge": cells per box of the vote bounds
 *                (0 = default: 4 while >= 4096 boxes remain, 1 = the cells themselves, 2 / 4 / 8);
 * "dense_f32":   dense scan filter at n > 32: 1 (default) = fp32 matrix cores, hypothesis fragments prefetched through
 *                an LDS ring (global_load_lds) and the next tile of rows in registers, 0 = fp64 matrix cores.  Votes are
 *                identical (the band is decided exactly);
 * "dense_fast_solve": 1 (default) = the n x n minimal solves of the dense system use elimination with
 *                partial pivoting and only fall back to the SVD pseudo-inverse near the rank decision,
 *                0 = always the SVD pseudo-inverse (and, for the dense least-squares fit, always the double-double
 *                route below);
 * "dense_dd":    dense least-squares fit (leastSquaresEstimate, DenseLinear...Estimator.hxx:64-96): 1 (default) = when
 *                the elimination on the normal equations A^T A meets a pivot below 1e-6 max|A^T A| (cond(A) beyond
 *                ~1e3) the system is solved again FROM THE ROWS: augmented Gram matrix in double-double, its Cholesky
 *                factor (= R and Q^T b of A's QR decomposition) in double-double, x = pinv(R) z by a Jacobi SVD with
 *                the reference's ABSOLUTE rank threshold 2.2e-16 -- the singular values, rank decision and solution
 *                of the reference's SVD pseudo-inverse of A, to 1e-6 up to cond(A) ~ 1e10 (tests/test_gpu_dense_cond.py);
 *                0 = the pseudo-inverse of the Gram block with a relative rank test 1e-13 (what entry points that only
 *                hold the block -- lsqr_solve_moments, the multi-GPU sum -- always do);
 * "scan_axis":   plane in 3-D over an indexed upload: 1 (default) = every cell also gets the direction of least spread
 *                of its observations, which are re-sorted inside the cell along it, and the bounded scan ("scan_bound")
 *                takes upper AND lower vote bounds of its candidates by rank from that order (csrc/axis.h: no
 *                observation is evaluated; what is counted exactly shrinks from ~500 to ~170 of 4096 hypotheses at
 *                10 M points); 0 = box-population bounds and pilots only.  Results are identical either way;
 * "scan_hyp_order": plane in 3-D, full counts (scan_bound 0 / lsqr_scan) of 1024..4096 hypotheses over an indexed
 *                upload: 1 (default) = the batch is walked in the order of a Morton key of (normal direction, offset)
 *                so that the 64 hypotheses of a group miss the same cells (csrc/cells.h: k_plane_order); 0 = sampling
 *                order.  Votes are identical;
 * "dense_mask_ring": LDS tile buffers per wave of the dense final fit's fused mask + normal-equations pass: 4 (default at
 *                n = 64) = one workgroup per CU with three tiles in flight, 2 = two workgroups per CU (A/B knob);
 * "mom_chunk":   records per workgroup of the mask / moment passes in units of 256 (0 = default: 4, wide US / phantom
 *                records 16; A/B knob -- the fixed-order sums, and with them the last bits of a fit, depend on it). */
LSQR_API int lsqr_set_option(lsqr_ctx *ctx, const char *name, int value);

/* State of the spatial index of the current upload ("scan_index"): out = {built (0/1), indexed
 * (finite) observations, cells, observations per cell}. */
LSQR_API int lsqr_index_info(const lsqr_ctx *ctx, uint64_t out[4]);

/* Work of the two-level scan for the CURRENT batch of hypotheses over the indexed upload: runs level 1 (the
 * cell-box test) alone.  out[0..7] = {surviving (hypothesis, cell) pairs of ALL hypotheses, (64-hypothesis group,
 * cell) level-1 evaluations of one pass over all hypotheses, cells, observations per cell, 1 if the batch was last
 * scanned by the bounded scan ("scan_bound"), its pilots, its second-pass hypotheses, surviving pairs of the
 * hypotheses it actually counted (= out[0] when not bounded)}; bound_out (nullable, H entries) receives per
 * hypothesis the summed population of its surviving cells: an upper bound on its votes.  LSQR_ERR_STATE when the
 * upload has no index.  Used by bench.py to price the scan against the instruction-issue roof. */
LSQR_API int lsqr_scan_workload(lsqr_ctx *ctx, uint32_t *bound_out, uint64_t out[8]);

/* Work of the LAST scan of the current batch for the models without a spatial index (dense system, US calibrations,
 * plane phantom).  The batch entry points (lsqr_batch_fit*, lsqr_step_scan, lsqr_ransac) scan the observations in
 * chunks and stop counting a hypothesis once  votes so far + observations still to come  cannot exceed a lower bound
 * of the running maximum at its index (option "scan_bound" 1, the batched form of RANSAC.hxx:94; csrc/earlyexit.h) --
 * winner, consensus set and iteration count are unchanged, an abandoned hypothesis reports its partial count.
 * out[0..5] = {1 if that path ran (0: every pair was evaluated), (observation, hypothesis) pairs handed to the scan
 * kernels, pairs of a full scan = H * N, candidates counted to the end first, hypotheses abandoned at the first
 * selection, hypotheses still alive after the last one}.  Used by bench.py to price the scan. */
LSQR_API int lsqr_scan_work(lsqr_ctx *ctx, uint64_t out[6]);

/* The LAST persistent Levenberg-Marquardt fit of this context (option "lm_persist"; csrc/lm_persist.h):
 * out[0..7] = {mode (1 / 3: MINPACK's step on the host between granules in pinned memory, 2: on the device), resident
 * workgroups, evaluations, end state (2: finished, 3: a bounded wait expired and the launch path took over), kernel
 * time in microseconds by the device's 100 MHz clock, fits of this context that fell back to the launch path, host
 * nanoseconds spent waiting for moment blocks, host nanoseconds spent in MINPACK's step (mode 1)}.
 * trace (nullable, 4 * trace_cap entries): per evaluation (the first 64) workgroup 0's clock in 10 ns ticks at
 * {own pass done, every workgroup arrived, partial blocks summed, step done / the host's reply in};
 * *trace_n = evaluations written.  Used by bench.py and tools/lm_persist_time.py. */
LSQR_API int lsqr_lm_persist_info(const lsqr_ctx *ctx, uint64_t out[8], uint64_t *trace, uint32_t trace_cap,
                                  uint32_t *trace_n);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Per-kernel HIP-event timing on the context's stream.  kernel ids: 0 sample, 1 estimate,
 * 2 scan, 3 mask, 4 moments, 5 solve, 6 spatial-index build (once per upload), 7 the max-|coordinate| pass
 * of the fp32 filters (once per upload). */
LSQR_API int lsqr_profile_enable(lsqr_ctx *ctx, int on);
LSQR_API int lsqr_profile_get(lsqr_ctx *ctx, int kernel_id, uint64_t *launches, double *total_ms);
LSQR_API int lsqr_profile_reset(lsqr_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* LSQR_HIP_H */
