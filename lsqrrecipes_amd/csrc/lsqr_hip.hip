// lsqr_hip.hip -- C ABI (include/lsqr_hip.h) over the HIP kernels in kernels.h.
// Host code here is plumbing: buffer ownership, launches on the context's stream, the serial
// replay of RANSAC.hxx's adaptive loop over batch results.  No CPU implementation of the scan
// or the fits exists in this library: without a device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <set>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lsqr_hip.h"
#include "kernels.h"
#include "models_nd.h"
#include "dense.h"
#include "dense_h16.h"
#include "us_kernels.h"
#include "us_h16.h"
#include "phantom_h16.h"
#include "cells.h"
#include "earlyexit.h"
#include "axis.h"
#include "sort.h"
#include "rigid.h"
#include "phantom.h"
#include "lm_persist.h"
#include "host_entry.h"

#include <condition_variable>
#include <mutex>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl is dlopen'ed when LSQR_MULTI_TRANSPORT=rccl asks for it

using namespace lsqr;

// ------------------------------------------------------------------------------------------------
struct lsqr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // the stream all work is enqueued on
  hipStream_t own_stream = nullptr;  // created with the context (stream == own_stream unless lsqr_set_stream)
  lsqr_model_cfg cfg{};
  ModelConsts mc{};
  bool has_model = false;
  int K = 0, P = 0, ND = 0, HS = 0;  // HS: row stride (doubles) of d_hparams

  const double *d_data = nullptr;  // observations (owned or attached)
  double *d_data_owned = nullptr;
  size_t data_cap = 0;  // doubles
  size_t n = 0, stride = 0;  // stride in doubles

  size_t H = 0, H_cap = 0;
  uint32_t *d_subsets = nullptr;
  double *d_hparams = nullptr;
  float *d_hparams_f32 = nullptr;
  unsigned long long *d_amb = nullptr;  // worklist of the dense MFMA filter
  bool absmax_valid = false;
  // spatial index of the point models (cells.h): Morton-sorted copy + one fp32 box per 128 records
  double *d_sorted = nullptr;
  uint32_t *d_queues = nullptr;  // work-queue counters of k_scan_cells
  CellBox *d_boxes = nullptr;   // [n_cells cell boxes | merged boxes of the bounds pass (k_super_boxes)]
  size_t sorted_cap = 0, boxes_cap = 0;  // doubles / boxes allocated
  uint32_t super_merge = 0;     // cells per merged box the second part currently holds (0: not built)
  // Selection counts of the bounded scans come back through pinned memory and steer the NEXT batches' launch sequence
  // (merged boxes or cells; pilot pass or not).  Two records, written alternately; a record is folded into
  // `bsel_known` only at a point where the host HAS synchronised with the batch that wrote it -- a ..._wait of its
  // slot, the end of a blocking entry point, or right before the record is reused (two bounded scans later) -- so the
  // launch sequence of a batch follows from the call sequence, never from host / device timing.
  BoundSel *h_bsel = nullptr;   // pinned [2], 64 B apart
  hipEvent_t bsel_ev[2] = {nullptr, nullptr};
  bool bsel_pending[2] = {false, false};
  uint64_t bsel_seq_of[2] = {0, 0};
  uint32_t bsel_H_of[2] = {0, 0};
  uint64_t bsel_seq = 0;        // bounded scans issued on this context
  int bsel_last_rec = -1;       // record the last bounded scan writes
  BoundSel bsel_known{};        // counts of the latest bounded scan the host has synchronised with ...
  uint32_t bsel_known_H = 0;    // ... its batch size (0: nothing known on this upload / model) ...
  uint64_t bsel_known_seq = 0;  // ... and its sequence number
  int slot_bsel_rec[2] = {-1, -1};   // lsqr_batch_fit_enqueue: the record of the slot's scan
  uint64_t slot_bsel_seq[2] = {0, 0};
  int step_bsel_rec[2] = {-1, -1};   // lsqr_step_finish_enqueue: the record of the step's scan (lsqr_step_scan)
  uint64_t step_bsel_seq[2] = {0, 0};
  bool merge_off = false;       // this upload: merged boxes let too many hypotheses through, bounds stay on the cells
  int opt_bound_merge = 0;      // 0: the cell model's default, 1: bounds on the cells themselves, 2 / 4 / 8
  size_t n_sorted = 0;      // finite records (non-finite ones never agree and are left out)
  uint32_t n_cells = 0, cell_pts = 0;
  bool index_valid = false;
  bool index_failed = false;  // build failed on this upload: stay on the exhaustive kernels
  // k_bounds of the current upload (min / max per dimension, max |coordinate|, non-finite count): serves both
  // the fp32 filters' absmax and the Morton grid of the index
  BoundsRow h_bounds{};
  bool bounds_valid = false;
  void *d_idx_scratch = nullptr;  // keys / permutation / radix-sort temporaries of the index build (kept)
  size_t idx_scratch_cap = 0;
  uint64_t hyp_since_upload = 0;  // hypotheses scanned on this upload (index build heuristic)
  uint64_t hyp_expected = 0;      // hypotheses the caller still expects to scan on this upload (lsqr_ransac: numTries)
  unsigned dense_amb_max = 0;  // fullest worklist segment of the last fp32 dense scan (diagnostics)
  int opt_dense_f32 = 2;  // dense scan filter at n = 64: 2 = fp16 matrix cores on two-way splits of the rows and the
                          // unknowns (dense_h16.h; the rows' fragments are built once per upload), 1 = fp32 matrix
                          // cores (hypothesis fragments through an LDS ring + next tile in registers), 0 = fp64 matrix
                          // cores; every filter sends its band to the exact fp64 re-check
  // dense_h16.h: the rows as fp16 fragment pairs (8 KiB per 32 rows) + the scaled right-hand sides, once per upload
  uint4 *d_h16 = nullptr;
  float *d_h16_bs = nullptr;
  size_t h16_tiles_cap = 0;
  bool h16_valid = false, h16_attr = false, h16_nomem = false;  // nomem: no room for the fragments on this upload
  int h16_unit = 0;       // 0: this device's matrix unit not probed yet, 1: keeps dense_h16.h's accumulation assumption, -1: not
  double h16_unit_dev = 0.0;  // the probe's worst deviation (u of the sum of magnitudes)
  double h16_pa = 1.0;
  float *d_h16_thr = nullptr;  // (-a, band, -ph, 0) per hypothesis of the batch
  // us_h16.h: the frames as fp16 fragment pairs (6 KiB per 32 frames), once per upload; the batch's hypothesis fragments
  uint4 *d_us16 = nullptr, *d_us16_x = nullptr;
  size_t us16_tiles_cap = 0;
  bool us16_valid = false, us16_attr = false, us16_nomem = false;
  Us16Scales us16_sc{};
  int opt_us_h16 = 1;  // US calibrations: 1 = agree() scan on the fp16 matrix cores (us_h16.h), 0 = packed fp32 filter
  int opt_dense_fast = 1;  // minimal solves: elimination first, SVD when near the rank decision
  int opt_us_fast = 1;     // US calibrations' minimal solves likewise (k_estimate_us)
  uint8_t *d_refused = nullptr;  // k_estimate_phantom_lu: hypotheses left to the Jacobi kernel
  size_t refused_cap = 0;
  int opt_phantom_fast = 1;  // plane phantom's minimal solves: LU + inverse iteration first (> 1: its iteration limit), Jacobi SVD for what it refuses; 0: Jacobi only
  int opt_lm_tiles = 1;      // matrix-core LM pass: compacted consensus set in field-major tiles, next tile in flight
  int opt_us_mask_mfma = 1;  // US calibrations: mask + analytic moment block on the fp64 matrix cores (kernels.h)
  int opt_refine = 1;      // index build: k-d refinement of the Morton order inside runs of 8192 records (cells.h)
  int opt_kd_levels = 7;   // ... and this many k-d levels above the runs by segmented sorts (runs of 8192 << levels),
  int opt_kd_after = 16384;  // built once the upload has been asked to scan this many hypotheses (the first index of an
                             // upload keeps the Morton order above the runs: 1.3 instead of 5.8 ms per 10 M records)
  int kd_build_levels = 0, index_kd_levels = 0;  // levels of the build in progress / of the index in place
  int opt_presorted = 0;   // index build: cells = runs of the UPLOAD order (experiments with other spatial orders)
  int opt_dense_wave = 3;  // dense minimal solves: 3 = elimination by one wave per system IN REGISTERS (n = 64; else as 1),
                           // 1 / 2 = in the wave's LDS area, four / two systems per workgroup, 0 = one workgroup per system
  int opt_dense_dd = 1;    // dense fit: systems the elimination refuses are solved again from the rows in double-double
  double *d_ddpart = nullptr;  // partial double-double Gram blocks of k_gram_dd_dense (allocated on first use)
  int opt_index = 1, opt_cpt = 0, opt_cell = 0, opt_block = 0, opt_hsplit = 0, opt_pairs = 0, opt_pairs_waves = 0;  // 0 off, 1 auto, 2 always; cells per wave tile, cell size
  uint8_t *d_valid = nullptr;
  uint32_t *d_votes = nullptr;
  uint32_t *d_ub = nullptr;  // per-hypothesis vote bound of the two-level scan's first level (k_cells_bounds)
  // axis-sorted cells (axis.h; plane, 3-D): per-cell axis, the cells' sorted projections
  CellAxis *d_axis = nullptr;
  float *d_cellT = nullptr;
  size_t axis_cap = 0, cellT_cap = 0;
  bool axis_valid = false;
  int opt_mom_chunk = 0;          // chunk of the mask / moment passes in units of kBlock records (0 = default; A/B)
  int opt_hyp_order = 1;          // 1: a full count of the plane walks the batch in key order (k_plane_order)
  int opt_axis = 1;               // 1: the bounded scan of the plane takes its vote bounds by rank (k_bound_axis)
  uint32_t *d_ub2 = nullptr;  // rank bounds: [upper | lower] of the candidates (compact order) | lower per hypothesis
  uint8_t *d_paircnt = nullptr;   // k_scan_pairs: survivors per (cell, group of 64 hypotheses)
  uint32_t *d_paircost = nullptr; // [n_cells cell costs | chunk sums]
  uint32_t *d_vpart = nullptr;    // per-workgroup partial votes of k_scan_pairs
  size_t paircnt_cap = 0, paircost_cap = 0, vpart_cap = 0;
  // bounded scan (cells.h: k_pick_*): the selected hypotheses as a compact batch
  uint32_t *d_sel = nullptr;        // [kPilots pilots | H_cap rest]
  BoundSel *d_bsel = nullptr;
  double *d_hparams2 = nullptr;
  float *d_hparams2_f32 = nullptr;
  uint32_t *d_votes2 = nullptr;     // [kPilots | H_cap]
  int opt_bound = 1;                // 1: batch entry points skip hypotheses that cannot win (exact winner), 0: count all
  uint32_t best_before = 0;         // exact votes of the best hypothesis of earlier batches (lsqr_ransac)
  bool allow_bound = false;         // set by the batch entry points around run_scan (lsqr_scan always counts all)
  uint64_t last_bound[4] = {0, 0, 0, 0};  // diagnostics of the last bounded scan: {used, pilots, rest, H}
  // chunked early exit of the dense / US scans (earlyexit.h)
  void *d_ee = nullptr;             // [EeState | sel_c | sel_o x 2 | compact thresholds | compact fp32 rows]
  EeState *h_ee = nullptr;          // pinned copy of the last early-exit scan's state (diagnostics: read after a synchronisation, lsqr_scan_work)
  bool ee_last = false;             // the last scan of this context took the early-exit path
  uint64_t ee_H = 0, ee_n = 0;      // its batch size and observation count
  bool scanned = false;
  bool external_stream = false;
  // lsqr_batch_fit_enqueue / _wait run on LANES: independent contexts on the same device, each with its own stream
  // and buffers, attached to this context's records.  Batches of different lanes overlap on the device: the dozen
  // one-workgroup kernels of a batch (selection, winner, solve) and the tails of the big ones no longer leave the
  // chip idle (measured, plane 10 M x 4096: 0.72 ms per batch on one lane, 0.56 on two, 0.51 on three).
  // Lane 0 is this context itself.
  static constexpr int kMaxLanes = 4;
  lsqr_ctx *lanes[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
  int opt_lanes = 4;
  bool is_lane = false;
  uint64_t data_epoch = 1;   // bumped whenever the records or the model change
  uint64_t lane_epoch = 0;   // (lanes) the epoch of the parent this lane is attached to
  std::vector<std::pair<std::string, int>> opt_log;  // options set on this context, replayed on new lanes
  hipEvent_t slot_ev[2] = {nullptr, nullptr};  // lsqr_batch_fit_enqueue / _wait
  // The matrix-core filters of the dense / US / phantom scans decide their band from per-workgroup worklists; a segment
  // that overflowed (never seen) means "scan again without the filter".  The blocking entry points read the fill right
  // after the scan (one host synchronisation); lsqr_batch_fit_enqueue must not wait, so there the scan only notes the
  // segment size (ovf_cap), the fill travels with the slot's pinned record and lsqr_batch_fit_wait runs the batch
  // again without the filter if it ever exceeds it.
  bool defer_ovf = false;        // set around run_scan_batch by lsqr_batch_fit_enqueue
  uint32_t ovf_cap = 0;          // segment size the deferred check compares against (0: this scan has no worklist)
  uint32_t slot_ovf_cap[2] = {0, 0};
  uint64_t slot_seed[2] = {0, 0};
  int opt_test_overflow = 0;     // tests: lsqr_batch_fit_wait treats the slot's worklist as overflowed
  uint64_t ovf_reruns = 0;       // batches run again because a worklist segment overflowed (diagnostics)
  uint64_t slot_first[2] = {0, 0}, slot_H[2] = {0, 0};
  bool slot_busy[2] = {false, false};
  hipEvent_t mdev_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // lsqr_moments_dev: staging slots of x
  unsigned mdev_next = 0;
  hipEvent_t step_ev[2] = {nullptr, nullptr};  // lsqr_step_finish_enqueue / _wait
  bool step_busy[2] = {false, false};

  double *d_rows = nullptr;  // plane phantom: the data rows a_i as an n x 32 matrix (phantom.h)
  size_t rows_cap = 0;
  bool rows_valid = false;

  uint8_t *d_mask = nullptr;
  size_t mask_cap = 0;
  bool mask_valid = false;

  double *d_partials = nullptr;  // kMaxPartials * MOM_MAX
  double *d_mom = nullptr;       // MOM_MAX
  double *d_vec = nullptr;       // 32 doubles: origin / parameter vector handed to kernels
  double *d_par = nullptr;       // 32 doubles: model parameters for mask/stats
  double *d_best = nullptr;      // scan-parameter row of the best hypothesis so far (lsqr_ransac*)
  LmState *d_lm = nullptr;
  SolveOut *d_out = nullptr;
  unsigned long long *d_counter = nullptr;
  bool origin_valid = false;
  int opt_ppl = 0, opt_filter = 1, opt_lm_host = 1, opt_syrk_diag = 0;
  int opt_fuse_mask = 1;  // winner's mask + moment block in one pass (0: two kernels, for A/B runs)
  int opt_mask_ring = 4;  // k_mask_syrk_dense: tile buffers per wave (2: two workgroups per CU; 4: one, three tiles in flight)
  int opt_mask_diag = 0;  // timing diagnostics of k_mask_syrk_dense: 1 = no matrix instructions, 2 = no row evaluation
  int opt_mask_band = 0;  // tests: scale factor of the dense fused mask's band (forces its serial re-evaluation path)
  long long opt_max_iter = 0;  // 0 = the reference's bound (numTries <= C(N,k))
  LmState h_lm;  // host copy of the LM state (opt_lm_host)
  double *d_lmrec = nullptr;  // consensus set copied tight and in order for the iterative fits (k_compact_*)
  size_t lmrec_cap = 0;
  unsigned long long *h_lmres = nullptr;  // pinned, device-visible: tagged granules written by k_lm_publish
  uint32_t lm_seq = 0;        // sequence number of the last evaluation (the tag the host polls for)
  int opt_lm_mfma = 1;        // 1: the LM pass accumulates (J | f)^T (J | f) on the matrix cores; 0: per-lane sums
  int opt_lm_fused = 1;       // 1: one launch per LM evaluation, result polled in pinned memory; 0: r01 path
  // lm_persist.h: a whole matrix-core LM fit in one launch.  0: the launch path (two launches per evaluation);
  // 1: persistent kernel with MINPACK's step on the host between tagged granules in pinned memory WHEN this context is
  // the only one of the process fitting on the device, the launch path otherwise; 3: the same kernel always; 2: persistent
  // kernel with the step on the device (workgroup 0).  Same iterates every way.
  int opt_lm_persist = 1;
  int opt_lm_persist_wgs = 0;          // resident workgroups G (0: 256 / 128 / 64 by the number of contexts on the device)
  int opt_lm_persist_timeout_ms = 2000;  // bound of every wait inside the kernel
  int opt_lm_persist_resident = 0;       // 1: a fit alone on the device keeps its tiles in registers (k_lm_persist<M, 4>).  Built,
                                         // bit-identical, measured no faster (r05: 12.4 us against 11.1 us per pass + broadcast at 245
                                         // workgroups -- with 1.9 waves per SIMD the pass is bound by its dependent chains, not by HBM)
  int opt_lm_persist_test_abort = 0;     // tests: workgroup 0 gives up at this evaluation (the fallback path)
  LmpCtl *d_lmp = nullptr;
  unsigned long long *h_lmcmd = nullptr;  // pinned, device-visible: the host's reply (coefficient granules + command)
  bool counted = false;  // this context counts among the device's root contexts (lmp_pool: the share of a persistent fit)
  // diagnostics of the last persistent fit: {mode, G, evaluations, status, kernel us, fallbacks, host ns waiting, host ns stepping}
  uint64_t lmp_last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t lmp_trace[LMP_TRACE][4];
  uint32_t lmp_trace_n = 0;

  // staged upload (lsqr_upload of large pageable buffers): ring of pinned chunks filled by a few host threads
  // while earlier chunks are in flight to the device
  static constexpr int kUpSlots = 8;
  static constexpr size_t kUpChunk = (size_t)8 << 20;
  void *h_up[kUpSlots] = {nullptr};
  hipEvent_t up_ev[kUpSlots] = {nullptr};
  int opt_upload_threads = -1;  // -1: LSQR_UPLOAD_THREADS or 4; 0: one plain hipMemcpy
  double last_upload_ms = 0.0;
  void *h_pin = nullptr;  // pinned staging (64 KiB)
  void *h_batch = nullptr;  // pinned results of a lsqr_ransac batch (grown on demand)
  size_t batch_pin_cap = 0;

  bool prof = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // HIP-event pairs of profiled launches; resolved lazily (lsqr_profile_get) so that profiling
  // adds no host synchronisation between kernels
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<int, int>> ev_pending;  // (kernel id, pair index)
  int ev_next = 0;
  uint64_t launches[8] = {0};
  double ms[8] = {0};

  char err[512] = {0};
};

namespace {

enum { KID_SAMPLE = 0, KID_ESTIMATE = 1, KID_SCAN = 2, KID_MASK = 3, KID_MOMENTS = 4, KID_SOLVE = 5,
       KID_INDEX = 6, KID_ABSMAX = 7 };

int fail(lsqr_ctx *c, int status, const char *fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof c->err, fmt, ap);
    va_end(ap);
  }
  return status;
}

#define HIPCHK(c, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail((c), LSQR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                    \
  } while (0)

void prof_flush(lsqr_ctx *c) {
  for (auto &pe : c->ev_pending) {
    float t = 0;
    hipEvent_t e0 = c->ev_pool[2 * pe.second], e1 = c->ev_pool[2 * pe.second + 1];
    if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&t, e0, e1) == hipSuccess) {
      c->launches[pe.first]++;
      c->ms[pe.first] += t;
    }
  }
  c->ev_pending.clear();
  c->ev_next = 0;
}

struct ProfScope {
  lsqr_ctx *c;
  int id, pair = -1;
  ProfScope(lsqr_ctx *c, int id) : c(c), id(id) {
    if (!c->prof) return;
    constexpr int kPairs = 512;
    if (c->ev_next >= kPairs) prof_flush(c);
    if ((int)c->ev_pool.size() < 2 * (c->ev_next + 1)) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      c->ev_pool.push_back(a);
      c->ev_pool.push_back(b);
    }
    pair = c->ev_next++;
    (void)hipEventRecord(c->ev_pool[2 * pair], c->stream);
  }
  ~ProfScope() {
    if (pair < 0) return;
    (void)hipEventRecord(c->ev_pool[2 * pair + 1], c->stream);
    c->ev_pending.push_back({id, pair});
  }
};

size_t dense_lds_bytes(int n) { return sizeof(double) * ((size_t)2 * n * (n | 1) + 3 * n); }
int dense_ne(int n) { return (n + 1) * (n + 2) / 2; }
constexpr int kDenseBlocks = 256;
int dense_pstride(int n) { return (dense_ne(n) + 1 + 7) & ~7; }

template <class T>
int ensure(lsqr_ctx *c, T **p, size_t *cap, size_t need) {
  if (need <= *cap && *p) return LSQR_OK;
  if (*p) HIPCHK(c, hipFree(*p));
  *p = nullptr;
  size_t ncap = std::max(need, *cap * 2);
  HIPCHK(c, hipMalloc((void **)p, ncap * sizeof(T)));
  *cap = ncap;
  return LSQR_OK;
}

int ensure_hyp(lsqr_ctx *c, size_t H) {
  if (H <= c->H_cap) return LSQR_OK;
  size_t cap = std::max<size_t>(H, 4096);
  if (c->d_subsets) (void)hipFree(c->d_subsets);
  if (c->d_hparams) (void)hipFree(c->d_hparams);
  if (c->d_hparams_f32) (void)hipFree(c->d_hparams_f32);
  c->d_hparams_f32 = nullptr;
  if (c->d_valid) (void)hipFree(c->d_valid);
  if (c->d_votes) (void)hipFree(c->d_votes);
  if (c->d_ub) (void)hipFree(c->d_ub);
  c->d_ub = nullptr;
  if (c->d_ub2) (void)hipFree(c->d_ub2);
  c->d_ub2 = nullptr;  // (allocated by the first bounded scan with rank bounds)
  for (void *b : {(void *)c->d_sel, (void *)c->d_bsel, (void *)c->d_hparams2, (void *)c->d_hparams2_f32, (void *)c->d_votes2})
    if (b) (void)hipFree(b);
  c->d_sel = nullptr; c->d_bsel = nullptr; c->d_hparams2 = nullptr; c->d_hparams2_f32 = nullptr; c->d_votes2 = nullptr;
  c->d_subsets = nullptr; c->d_hparams = nullptr; c->d_valid = nullptr; c->d_votes = nullptr;
  c->H_cap = 0;
  HIPCHK(c, hipMalloc((void **)&c->d_subsets, cap * 64 * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc((void **)&c->d_hparams, cap * 64 * sizeof(double)));
  HIPCHK(c, hipMalloc((void **)&c->d_hparams_f32, cap * 32 * sizeof(float)));  // M::SPF <= 32
  HIPCHK(c, hipMalloc((void **)&c->d_valid, cap));
  HIPCHK(c, hipMalloc((void **)&c->d_votes, cap * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc((void **)&c->d_ub, cap * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc((void **)&c->d_sel, (cap + kPilots) * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc((void **)&c->d_bsel, sizeof(BoundSel)));
  HIPCHK(c, hipMalloc((void **)&c->d_hparams2, (cap + kPilots) * 64 * sizeof(double)));
  HIPCHK(c, hipMalloc((void **)&c->d_hparams2_f32, (cap + kPilots) * 32 * sizeof(float)));
  HIPCHK(c, hipMalloc((void **)&c->d_votes2, (cap + kPilots) * sizeof(uint32_t)));
  c->H_cap = cap;
  return LSQR_OK;
}

int need_ready(lsqr_ctx *c, bool need_data) {
  if (!c) return LSQR_ERR_INVALID;
  if (!c->has_model) return fail(c, LSQR_ERR_STATE, "lsqr_set_model has not been called");
  if (need_data && (!c->d_data || c->n == 0))
    return fail(c, LSQR_ERR_STATE, "no observations uploaded");
  HIPCHK(c, hipSetDevice(c->device));
  return LSQR_OK;
}

// smallest chunk per block of the mask / moment passes (kernels.h: kMomChunk)
static inline int mom_chunk(const lsqr_ctx *c) {
  if (c->opt_mom_chunk > 0) return c->opt_mom_chunk * kBlock;
  const int m = c->cfg.model;
  return (m == LSQR_MODEL_US_SINGLE || m == LSQR_MODEL_US_POINTER || m == LSQR_MODEL_PHANTOM) ? kMomChunkWide : kMomChunk;
}
int grid_for(size_t items, int per_block, int max_blocks) {
  size_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (size_t)max_blocks) b = max_blocks;
  return (int)b;
}

// ---- hypotheses ---------------------------------------------------------------------------------
template <int D>
int run_bounds(lsqr_ctx *c);

// ---- selection feedback of the bounded scan (see lsqr_ctx::h_bsel) ----------------------------------------------
static BoundSel *bsel_rec(lsqr_ctx *c, int r) { return (BoundSel *)((char *)c->h_bsel + 64 * r); }
// the host has synchronised with the batch that wrote record r
static void bsel_fold(lsqr_ctx *c, int r) {
  if (r < 0 || !c->h_bsel || !c->bsel_pending[r]) return;
  c->bsel_pending[r] = false;
  if (c->bsel_known_H && c->bsel_seq_of[r] < c->bsel_known_seq) return;  // an older batch than the one known
  c->bsel_known = *bsel_rec(c, r);
  c->bsel_known_H = c->bsel_H_of[r];
  c->bsel_known_seq = c->bsel_seq_of[r];
}
// the host has synchronised with everything enqueued on the context's stream
static void bsel_host_synced(lsqr_ctx *c) {
  if (!c->h_bsel) return;
  const int first = c->bsel_seq_of[0] <= c->bsel_seq_of[1] ? 0 : 1;
  bsel_fold(c, first);
  bsel_fold(c, first ^ 1);
}
// new records / new model: nothing is known; copies still in flight are waited for, not abandoned
static void bsel_reset(lsqr_ctx *c) {
  for (int r = 0; r < 2; r++)
    if (c->bsel_pending[r] && c->bsel_ev[r]) {
      (void)hipEventSynchronize(c->bsel_ev[r]);
      c->bsel_pending[r] = false;
    }
  c->bsel_known = BoundSel{};
  c->bsel_known_H = 0;
  c->slot_bsel_rec[0] = c->slot_bsel_rec[1] = -1;
  c->step_bsel_rec[0] = c->step_bsel_rec[1] = -1;
  c->bsel_last_rec = -1;
}

// the host waits for everything enqueued on the context's stream: also the point where the selection feedback folds
static hipError_t sync_stream(lsqr_ctx *c) {
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) bsel_host_synced(c);
  return e;
}

int ensure_absmax(lsqr_ctx *c) {
  if (c->absmax_valid) return LSQR_OK;
  const int m = c->cfg.model;
  if ((m == LSQR_MODEL_PLANE || m == LSQR_MODEL_SPHERE || m == LSQR_MODEL_LINE || m == LSQR_MODEL_LINE2D) &&
      c->ND <= 3) {
    // point models: one pass gives max |coordinate| AND the bounds the index build needs (cells.h: k_bounds; the
    // profile scope is around the kernel itself in run_bounds: r03 timed the final reduction, the copy back and the
    // synchronisation with it and reported 67 us for a 45 us pass)
    int st = c->ND == 3 ? run_bounds<3>(c) : run_bounds<2>(c);
    if (st != LSQR_OK) return st;
    double am;
    memcpy(&am, &c->h_bounds.amax, sizeof am);
    if (c->h_bounds.nonfinite) am = std::numeric_limits<double>::infinity();  // as k_absmax: switches the filters off
    c->mc.absmax = am;
    c->mc.absmax_rot = am;
    c->absmax_valid = true;
    return LSQR_OK;
  }
  HIPCHK(c, hipMemsetAsync(c->d_counter + 5, 0, 2 * sizeof(unsigned long long), c->stream));
  int grid = grid_for(c->n, kBlock * 16, 1024);
  const bool us = c->cfg.model == LSQR_MODEL_US_SINGLE || c->cfg.model == LSQR_MODEL_US_POINTER ||
                  c->cfg.model == LSQR_MODEL_PHANTOM;  // Frame records: int slot 12, rotation first
  {
    ProfScope ps(c, KID_ABSMAX);
    hipLaunchKernelGGL(k_absmax, dim3(grid), dim3(kBlock), 0, c->stream, c->d_data, c->stride, c->n,
                       c->ND, us ? 12 : -1, us ? 9 : (c->cfg.model == LSQR_MODEL_DENSE ? c->ND - 1 : c->ND),
                       c->d_counter + 5);  // dense: the coefficient columns separately from the right-hand side
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter + 5, 2 * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  memcpy(&c->mc.absmax, c->h_pin, sizeof(double));
  memcpy(&c->mc.absmax_rot, (char *)c->h_pin + 8, sizeof(double));
  c->absmax_valid = true;
  return LSQR_OK;
}

int run_estimate(lsqr_ctx *c) {
  return dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (requires { M::SPF; } || M::IS_US) {
      int st = ensure_absmax(c);
      if (st != LSQR_OK) return st;
    }
    ProfScope ps(c, KID_ESTIMATE);
    if constexpr (requires { M::IS_PHANTOM; }) {
      // one wave per hypothesis (measured 1.76 ms per 4096 against 1.96 / 2.53 ms with 128 / 256 threads:
      // the sweeps are bound by instruction issue, and a single-wave workgroup lands on any SIMD)
      if (c->opt_phantom_fast) {
        // r05: LU + inverse iteration (phantom.h: k_estimate_phantom_lu), the Jacobi SVD for what it refuses
        int st = ensure(c, &c->d_refused, &c->refused_cap, std::max<size_t>(c->H, 1));
        if (st != LSQR_OK) return st;
        uint8_t *d_ref = c->d_refused;
        static const bool dbg_on = getenv("LSQR_PHANTOM_DEBUG") != nullptr;
        unsigned long long *d_dbg = nullptr;
        if (dbg_on) (void)hipMalloc((void **)&d_dbg, sizeof(unsigned long long) * 4 * c->H);
        hipLaunchKernelGGL(k_estimate_phantom_lu, dim3((unsigned)((c->H + 3) / 4)), dim3(256), 0, c->stream, c->d_data,
                           c->stride, c->n, c->d_subsets, (uint32_t)c->H, c->d_hparams, c->d_valid, d_ref,
                           c->opt_phantom_fast > 1 ? c->opt_phantom_fast : 96, d_dbg);
        if (d_dbg) {  // diagnostics (LSQR_PHANTOM_DEBUG): iterations and 10-ns ticks per phase and hypothesis
          std::vector<unsigned long long> hd(4 * c->H);
          (void)hipStreamSynchronize(c->stream);
          (void)hipMemcpy(hd.data(), d_dbg, hd.size() * 8, hipMemcpyDeviceToHost);
          (void)hipFree(d_dbg);
          double sit = 0, slu = 0, sitr = 0, sfin = 0, mit = 0, mitr = 0, spcg = 0, mpcg = 0;
          size_t nref = 0, nconv = 0;
          for (size_t h = 0; h < c->H; h++) {
            const double it = (double)(hd[4 * h] & 0xFFFFFFFFu);
            sit += it, slu += (double)hd[4 * h + 1], sitr += (double)hd[4 * h + 2], sfin += (double)hd[4 * h + 3];
            mit = std::max(mit, it), mitr = std::max(mitr, (double)hd[4 * h + 2]);
            spcg += (double)((hd[4 * h] >> 40) & 0xFF), mpcg = std::max(mpcg, (double)((hd[4 * h] >> 40) & 0xFF));
            nref += (hd[4 * h] >> 32) & 1, nconv += (hd[4 * h] >> 33) & 1;
          }
          fprintf(stderr, "phantom_lu: H %zu mean it %.1f max %.0f, correction steps mean %.2f max %.0f, refused %zu converged %zu; mean us: setup+LU %.1f iter %.1f (max %.1f) finish %.1f\n",
                  c->H, sit / c->H, mit, spcg / c->H, mpcg, nref, nconv, slu / c->H / 100, sitr / c->H / 100, mitr / 100, sfin / c->H / 100);
        }
        hipLaunchKernelGGL(k_estimate_phantom<64>, dim3((unsigned)std::min<size_t>(c->H, 512)), dim3(64), 0, c->stream,
                           c->d_data, c->stride, c->n, c->d_subsets, (uint32_t)c->H, c->d_hparams, c->d_valid,
                           (const uint8_t *)d_ref);
      } else if (c->opt_block == 256)
        hipLaunchKernelGGL(k_estimate_phantom<256>, dim3((unsigned)c->H), dim3(256), 0, c->stream, c->d_data,
                           c->stride, c->n, c->d_subsets, (uint32_t)c->H, c->d_hparams, c->d_valid, (const uint8_t *)nullptr);
      else
        hipLaunchKernelGGL(k_estimate_phantom<64>, dim3((unsigned)c->H), dim3(64), 0, c->stream, c->d_data,
                           c->stride, c->n, c->d_subsets, (uint32_t)c->H, c->d_hparams, c->d_valid, (const uint8_t *)nullptr);
      hipLaunchKernelGGL((k_prepare_f32_us<M>), dim3((unsigned)((c->H + 255) / 256)), dim3(256), 0,
                         c->stream, c->d_hparams, (uint32_t)c->H, c->mc, c->d_hparams_f32);
    } else if constexpr (M::IS_DENSE) {
      const int n = (int)c->cfg.dim;
      if (c->opt_dense_fast && c->opt_dense_wave) {
        // fast path: four hypotheses per workgroup, one wave each (dense.h: k_estimate_dense_w4); what its elimination
        // refuses is marked and taken through the SVD pseudo-inverse by the workgroup kernel behind it
        const int wpb = c->opt_dense_wave == 2 ? 2 : 4;   // systems per workgroup (2: 68 KB of LDS at n = 64, A/B)
        const size_t lds = sizeof(double) * wpb * ((size_t)n * (n | 1) + 2 * n);
        // (unconditionally: the attribute belongs to the (kernel, device) pair, and a flag per process or per context
        // misses the second device / the second kernel; the call is cheap)
        (void)hipFuncSetAttribute((const void *)k_estimate_dense_w4, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(sizeof(double) * 4 * (64 * 65 + 128)));
        if (n == 64 && c->opt_dense_wave == 3)  // the system in registers, one wave per hypothesis (k_estimate_dense_r64)
          hipLaunchKernelGGL(k_estimate_dense_r64, dim3((unsigned)c->H), dim3(64), 0, c->stream, c->d_data, c->stride, c->n,
                             c->d_subsets, (uint32_t)c->H, (int)M::SP, c->d_hparams, c->d_valid);
        else
        hipLaunchKernelGGL(k_estimate_dense_w4, dim3((unsigned)((c->H + wpb - 1) / wpb)), dim3(64 * wpb), lds, c->stream, c->d_data,
                           c->stride, c->n, c->d_subsets, (uint32_t)c->H, n, (int)M::SP, c->d_hparams, c->d_valid);
        hipLaunchKernelGGL(k_estimate_dense, dim3((unsigned)c->H), dim3(256), dense_lds_bytes(n), c->stream, c->d_data,
                           c->stride, c->n, c->d_subsets, (uint32_t)c->H, n, (int)M::SP, 0, c->d_hparams, c->d_valid, 1);
      } else {
        hipLaunchKernelGGL(k_estimate_dense, dim3((unsigned)c->H), dim3(256), dense_lds_bytes(n), c->stream, c->d_data,
                           c->stride, c->n, c->d_subsets, (uint32_t)c->H, n, (int)M::SP, c->opt_dense_fast,
                           c->d_hparams, c->d_valid, 0);
      }
    } else if constexpr (M::IS_US) {
      hipLaunchKernelGGL((k_estimate_us<(M::K == 4)>), dim3((unsigned)c->H), dim3(64), 0,
                         c->stream, c->d_data, c->stride, c->n, c->d_subsets, (uint32_t)c->H,
                         c->mc, c->d_hparams, c->d_valid, c->opt_us_fast);
      hipLaunchKernelGGL((k_prepare_f32_us<M>), dim3((unsigned)((c->H + 255) / 256)),
                         dim3(256), 0, c->stream, c->d_hparams, (uint32_t)c->H, c->mc,
                         c->d_hparams_f32);
    } else {
      int grid = (int)((c->H + kBlock - 1) / kBlock);
      hipLaunchKernelGGL((k_estimate<M>), dim3(grid), dim3(kBlock), 0, c->stream, c->d_data,
                         c->stride, c->n, c->d_subsets, (uint32_t)c->H, c->mc, c->d_hparams,
                         c->d_hparams_f32, c->d_valid);
    }
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  });
}

constexpr uint32_t kScanChunk = 8192;  // hypotheses per scan launch (LDS counters: 32 KiB)

template <class M, int PPL>
int run_scan_ppl(lsqr_ctx *c) {
  HIPCHK(c, hipMemsetAsync(c->d_votes, 0, c->H * sizeof(uint32_t), c->stream));
  size_t tiles = (c->n + (size_t)kBlock * PPL - 1) / ((size_t)kBlock * PPL);
  for (size_t h0 = 0; h0 < c->H; h0 += kScanChunk) {
    uint32_t hc = (uint32_t)std::min<size_t>(kScanChunk, c->H - h0);
    size_t lds = (size_t)hc * sizeof(uint32_t);
    int per_cu = (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
    if (per_cu < 1) per_cu = 1;
    size_t max_blocks = (size_t)256 * per_cu;
    size_t tpb = (tiles + max_blocks - 1) / max_blocks;
    int grid = (int)((tiles + tpb - 1) / tpb);
    ProfScope ps(c, KID_SCAN);
    if (c->opt_filter)
      hipLaunchKernelGGL((k_scan<M, PPL, true>), dim3(grid), dim3(kBlock), lds, c->stream, c->d_data,
                         c->stride, c->n, c->d_hparams + h0 * M::SP, hc, c->mc, c->d_votes + h0);
    else
      hipLaunchKernelGGL((k_scan<M, PPL, false>), dim3(grid), dim3(kBlock), lds, c->stream,
                         c->d_data, c->stride, c->n, c->d_hparams + h0 * M::SP, hc, c->mc,
                         c->d_votes + h0);
    HIPCHK(c, hipGetLastError());
  }
  return LSQR_OK;
}

template <class M, int PPL, int GRAN = 0>
int run_scan_f32(lsqr_ctx *c) {
  HIPCHK(c, hipMemsetAsync(c->d_votes, 0, c->H * sizeof(uint32_t), c->stream));
  size_t tiles = (c->n + (size_t)kBlock * PPL - 1) / ((size_t)kBlock * PPL);
  for (size_t h0 = 0; h0 < c->H; h0 += kScanChunk) {
    uint32_t hc = (uint32_t)std::min<size_t>(kScanChunk, c->H - h0);
    size_t lds = (size_t)hc * sizeof(uint32_t);
    int per_cu = (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
    if (per_cu < 1) per_cu = 1;
    size_t max_blocks = (size_t)256 * per_cu;
    size_t tpb = (tiles + max_blocks - 1) / max_blocks;
    int grid = (int)((tiles + tpb - 1) / tpb);
    ProfScope ps(c, KID_SCAN);
    hipLaunchKernelGGL((k_scan_f32<M, PPL, GRAN>), dim3(grid), dim3(kBlock), lds, c->stream,
                       c->d_data, c->stride, c->n, c->d_hparams + h0 * M::SP,
                       c->d_hparams_f32 + h0 * M::SPF, hc, c->mc, c->d_votes + h0);
    HIPCHK(c, hipGetLastError());
  }
  return LSQR_OK;
}


// cell model of a point model (cells.h); models without one keep the exhaustive kernels
template <class M>
struct CellOf {};
template <int D>
struct CellOf<PlaneModel<D>> {
  typedef PlaneCell<D> type;
};
template <>
struct CellOf<Line2DModel> {  // same agree() as the 2-D hyperplane
  typedef PlaneCell<2> type;
};
template <int D>
struct CellOf<SphereModel<D>> {
  typedef SphereCell<D> type;
};
template <int D>
struct CellOf<LineModel<D>> {
  typedef LineCell<D> type;
};

// ---- spatial index (cells.h) ------------------------------------------------------------------------
// invalidates the index; its buffers are kept (a later upload of similar size rebuilds into them: hipMalloc /
// hipFree of a few hundred MB cost milliseconds, several times the build's 0.65 ms of kernel time)
// ---- lanes (lsqr_batch_fit_enqueue / _wait) -------------------------------------------------------------------
// nothing of any lane may still be reading this context's records
void lanes_quiesce(lsqr_ctx *c) {
  for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
    if (c->lanes[i]) {
      (void)hipStreamSynchronize(c->lanes[i]->stream);
      for (int s = 0; s < 2; s++) c->lanes[i]->slot_busy[s] = false;  // results of a replaced upload are void
    }
}

void drop_index(lsqr_ctx *c) {
  c->h16_valid = false;  // (derived from the records alone, like the index)
  c->us16_valid = false;
  c->h16_nomem = c->us16_nomem = false;
  c->n_sorted = 0;
  c->n_cells = 0;
  c->index_valid = false;
  c->index_kd_levels = 0;
  c->axis_valid = false;
}
void free_index(lsqr_ctx *c) {
  if (c->d_sorted) (void)hipFree(c->d_sorted);
  if (c->d_boxes) (void)hipFree(c->d_boxes);
  c->d_sorted = nullptr;
  c->d_boxes = nullptr;
  c->sorted_cap = c->boxes_cap = 0;
  drop_index(c);
}

// k_bounds over the current upload (point models): min / max per dimension, max |coordinate|, non-finite count
template <int D>
int run_bounds(lsqr_ctx *c) {
  if (c->bounds_valid) return LSQR_OK;
  const int nb = grid_for(c->n, 256 * 16, 1024);
  BoundsRow *rows = (BoundsRow *)c->d_partials;  // scratch: 1024 rows x 64 B (d_partials holds 4.4 MB)
  static_assert(sizeof(BoundsRow) == 64, "BoundsRow is 8 words");
  {
    ProfScope ps(c, KID_ABSMAX);
    hipLaunchKernelGGL((k_bounds<D>), dim3(nb), dim3(256), 0, c->stream, c->d_data, c->stride, c->n, rows);
  }
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(256), 0, c->stream, rows, nb, rows + 1024);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pin, rows + 1024, sizeof(BoundsRow), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  memcpy(&c->h_bounds, c->h_pin, sizeof(BoundsRow));
  c->bounds_valid = true;
  return LSQR_OK;
}

template <int D>
int build_index(lsqr_ctx *c, uint32_t cell_pts) {
  drop_index(c);
  c->cell_pts = cell_pts;
  if (getenv("LSQR_TEST_FAIL_INDEX"))  // test hook: behave as if the sorted copy could not be allocated
    return fail(c, LSQR_ERR_HIP, "index build failure requested by LSQR_TEST_FAIL_INDEX");
  const size_t n = c->n;
#define IDXCHK(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      drop_index(c);                                                                          \
      return fail(c, LSQR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                  __FILE__, __LINE__);                                                        \
    }                                                                                         \
  } while (0)
  int st = run_bounds<D>(c);  // (usually already there: the fp32 filters asked for max |x| when the hypotheses came)
  if (st != LSQR_OK) return st;
  const BoundsRow &hb = c->h_bounds;
  if (hb.mn[0] == ~0ULL) {  // no finite record at all: nothing can agree
    c->index_valid = true;
    return LSQR_OK;
  }
  IndexGrid g;
  memset(&g, 0, sizeof g);
  int total = 0;
  while (total < 24 && ((size_t)4 << (total + 1)) <= n) total++;  // about 4 records per bin
  int bits = total / D;
  if (bits < 1) bits = 1;
  if (D == 3 && bits > 8) bits = 8;
  if (D == 2 && bits > 12) bits = 12;
  g.bits = (uint32_t)bits;
  g.nbins = 1u << (bits * D);
  for (int d = 0; d < D; d++) {
    double lo = ord_u64_inv(hb.mn[d]), hi = ord_u64_inv(hb.mx[d]);
    double sc = hi > lo ? (double)(1u << bits) / (hi - lo) : 0.0;
    if (!(sc >= 0.0) || !(sc <= 1.7976931348623157e308)) sc = 0.0;
    g.lo[d] = lo;
    g.scale[d] = sc;
  }
  if (hb.nonfinite > n) return fail(c, LSQR_ERR_HIP, "index build: inconsistent bounds pass");
  // scratch: keys, permutation (in / out) + the radix sort's temporaries; kept with the context
  const size_t words = (n + 63) & ~(size_t)63;
  size_t tmp_bytes = 0;
  IDXCHK(sort_pairs_u32(nullptr, &tmp_bytes, nullptr, nullptr, nullptr, nullptr, n, (unsigned)(bits * D + 1),
                        c->stream));
  {  // the k-d levels above the runs sort (segment, coordinate) keys: up to 16 + 12 bits
    size_t tb2 = 0;
    IDXCHK(sort_pairs_u32(nullptr, &tb2, nullptr, nullptr, nullptr, nullptr, n, 28u, c->stream));
    tmp_bytes = std::max(tmp_bytes, tb2);
  }
  constexpr size_t kSegExtBytes = 4096 * sizeof(SegExtent);
  // (+ one more index array and the compact fp32 copy of the coordinates for the k-d levels: 16 B per record)
  const size_t need = (5 + D) * words * sizeof(uint32_t) + tmp_bytes + 256 + kSegExtBytes;
  if (need > c->idx_scratch_cap) {
    if (c->d_idx_scratch) (void)hipFree(c->d_idx_scratch);
    c->d_idx_scratch = nullptr;
    c->idx_scratch_cap = 0;
    IDXCHK(hipMalloc(&c->d_idx_scratch, need));
    c->idx_scratch_cap = need;
  }
  if ((st = ensure(c, &c->d_sorted, &c->sorted_cap, std::max<size_t>(n, 1) * D)) != LSQR_OK) return st;
  const size_t n_sorted = n - (size_t)hb.nonfinite;  // the non-finite records carry the largest key: they sort to the tail
  const uint32_t n_cells = (uint32_t)((n_sorted + cell_pts - 1) / cell_pts);
  if ((st = ensure(c, &c->d_boxes, &c->boxes_cap, std::max<size_t>((size_t)n_cells + n_cells / 2 + 2, 1))) != LSQR_OK)
    return st;
  c->super_merge = 0;
  // every buffer is in place (kept across uploads): from here on the build is device work only
  ProfScope ps(c, KID_INDEX);
  uint32_t *k_in = (uint32_t *)c->d_idx_scratch, *v_in = k_in + words, *k_out = v_in + words,
           *v_out = k_out + words;
  uint32_t *q_b = v_out + words;
  float *xyz = (float *)(q_b + words);
  void *tmp = (void *)(xyz + (size_t)D * words);
  uint32_t *perm = v_out;  // the order the cells are cut from
  const unsigned gn = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL((k_keys<D>), dim3(gn), dim3(256), 0, c->stream, c->d_data, c->stride, n, g, k_in, v_in,
                     c->opt_presorted);
  IDXCHK(hipGetLastError());
  IDXCHK(sort_pairs_u32(tmp, &tmp_bytes, k_in, k_out, v_in, v_out, n, (unsigned)(bits * D + 1), c->stream));
  c->n_sorted = n_sorted;
  c->n_cells = n_cells;
  // k-d levels above the runs (cells.h: k_seg_extent / k_seg_keys + one radix sort per level), then the local k-d
  // refinement of every run (k_refine_runs): compact cells, fewer surviving pairs
  if (c->opt_refine && !c->opt_presorted && n_sorted > cell_pts && cell_pts >= 128 && cell_pts < kRunPts &&
      (cell_pts & (cell_pts - 1)) == 0) {
    SegExtent *d_ext = (SegExtent *)((char *)tmp + ((tmp_bytes + 255) & ~(size_t)255));
    uint32_t *q_a = v_in;  // (k_in / v_in are free after the Morton sort) positions into the Morton-ordered copy
    uint32_t top_shift = 13 + (uint32_t)std::max(0, std::min(c->kd_build_levels, 12));  // S = 8192 << levels
    while (top_shift > 14 && ((size_t)1 << (top_shift - 1)) >= n_sorted) top_shift--;  // one segment holds everything
    bool any = false;
    for (uint32_t sh = top_shift; sh > 13; sh--) {
      const uint32_t nseg = (uint32_t)((n_sorted + ((size_t)1 << sh) - 1) >> sh);
      if (nseg + 1 > 4096) continue;  // (more segments than the extent table: this level is left to the Morton order)
      if (!any) {
        hipLaunchKernelGGL((k_seg_gather<D>), dim3(gn), dim3(256), 0, c->stream, c->d_data, c->stride, v_out, n, n_sorted,
                           xyz, q_a);
        any = true;
      }
      unsigned segbits = 1;
      while ((1u << segbits) < nseg + 1) segbits++;
      hipLaunchKernelGGL(k_seg_extent_init, dim3((nseg + 255) / 256), dim3(256), 0, c->stream, d_ext, nseg);
      hipLaunchKernelGGL((k_seg_extent<D>), dim3((unsigned)((n_sorted + 1023) / 1024)), dim3(256), 0, c->stream, xyz, q_a,
                         n_sorted, sh, d_ext);
      hipLaunchKernelGGL((k_seg_keys<D>), dim3(gn), dim3(256), 0, c->stream, xyz, q_a, n, n_sorted, sh, nseg, d_ext, k_in,
                         q_b);
      IDXCHK(hipGetLastError());
      // sort (k_in, q_b) -> (k_out, q_a): the old order's buffer takes the new one
      IDXCHK(sort_pairs_u32(tmp, &tmp_bytes, k_in, k_out, q_b, q_a, n, 16u + segbits, c->stream));
    }
    if (any) {  // positions -> record indices
      hipLaunchKernelGGL(k_seg_compose, dim3(gn), dim3(256), 0, c->stream, v_out, q_a, n, q_b);
      IDXCHK(hipGetLastError());
      perm = q_b;
    }
    (void)hipFuncSetAttribute((const void *)k_refine_runs<D>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)refine_lds_bytes(D));
    const unsigned runs = (unsigned)((n_sorted + kRunPts - 1) / kRunPts);
    hipLaunchKernelGGL((k_refine_runs<D>), dim3(runs), dim3(1024), refine_lds_bytes(D), c->stream, c->d_data, c->stride,
                       perm, n_sorted, cell_pts);
    IDXCHK(hipGetLastError());
  }
  if (c->n_cells) {
    hipLaunchKernelGGL((k_gather_boxes<D>), dim3((c->n_cells + 3) / 4), dim3(256), 0, c->stream, c->d_data,
                       c->stride, perm, c->n_sorted, c->n_cells, cell_pts, c->d_sorted, c->d_boxes);
    IDXCHK(hipGetLastError());
  }
  // axis-sorted cells (axis.h): plane in 3-D, cells of 256 / 512 records
  c->axis_valid = false;
  if (D == 3 && c->opt_axis && c->cfg.model == LSQR_MODEL_PLANE && c->n_cells && (cell_pts == 512 || cell_pts == 256)) {
    if ((st = ensure(c, &c->d_axis, &c->axis_cap, (size_t)c->n_cells)) != LSQR_OK) return st;
    if ((st = ensure(c, &c->d_cellT, &c->cellT_cap, (size_t)c->n_cells * cell_pts)) != LSQR_OK) return st;
    double *d_cmom = (double *)c->d_idx_scratch;  // 80 B per cell: the sort's scratch is free again (16 B per record)
    if ((size_t)c->n_cells * 10 * sizeof(double) <= c->idx_scratch_cap) {
      const unsigned gc = (c->n_cells + 3) / 4;
      if (cell_pts == 512) {
        hipLaunchKernelGGL((k_cell_moments<8>), dim3(gc), dim3(256), 0, c->stream, c->d_sorted, c->n_sorted, c->n_cells,
                           c->d_boxes, d_cmom);
      } else {
        hipLaunchKernelGGL((k_cell_moments<4>), dim3(gc), dim3(256), 0, c->stream, c->d_sorted, c->n_sorted, c->n_cells,
                           c->d_boxes, d_cmom);
      }
      const unsigned ngrp = (c->n_cells + kAxisGroup - 1) / kAxisGroup;
      hipLaunchKernelGGL(k_cell_axes, dim3((ngrp + 255) / 256), dim3(256), 0, c->stream, d_cmom, c->d_boxes, c->n_cells,
                         c->d_axis);
      if (cell_pts == 512)
        hipLaunchKernelGGL((k_cell_sort<8>), dim3(gc), dim3(256), 0, c->stream, c->d_sorted, c->n_sorted, c->n_cells,
                           c->d_boxes, c->d_axis, c->d_cellT);
      else
        hipLaunchKernelGGL((k_cell_sort<4>), dim3(gc), dim3(256), 0, c->stream, c->d_sorted, c->n_sorted, c->n_cells,
                           c->d_boxes, c->d_axis, c->d_cellT);
      IDXCHK(hipGetLastError());
      c->axis_valid = true;
    }
  }
#undef IDXCHK
  c->index_kd_levels = c->kd_build_levels;
  c->index_valid = true;
  return LSQR_OK;
}

// a batch of hypotheses handed to the two-level scan: the context's current batch, or a compacted selection of it
struct ScanBatch {
  const double *sp;       // fp64 scan parameters, M::SP per hypothesis
  const float *spf;       // fp32 block, M::SPF per hypothesis
  size_t H;               // hypotheses (capacity when h_dev is set)
  uint32_t *votes;        // zeroed by the caller when h_dev is set
  const uint32_t *h_dev;  // device-side count (bounded scan) or null
  uint32_t h_off = 0;     // the batch starts at hypothesis h_off of the device-side selection (k_scan_pairs chunks)
};

template <class CM, int PP, int CPT, int BS, bool LDSB = false>
int launch_scan_cells(lsqr_ctx *c, const ScanBatch &b) {
  typedef typename CM::M M;
  const CellConsts cc = cell_consts((const CM *)nullptr, c->mc);
  HIPCHK(c, hipMemsetAsync(b.votes, 0, b.H * sizeof(uint32_t), c->stream));
  if (c->n_cells == 0) return LSQR_OK;
  const size_t wtiles = ((size_t)c->n_cells + CPT - 1) / CPT;
  constexpr int wpb = BS / 64;  // waves per workgroup
  if (!c->d_queues) HIPCHK(c, hipMalloc((void **)&c->d_queues, kQueues * kQueuePitch * sizeof(uint32_t)));
  uint32_t *d_next = c->d_queues;
  for (size_t h0 = 0; h0 < b.H; h0 += kScanChunk) {
    uint32_t hc = (uint32_t)std::min<size_t>(kScanChunk, b.H - h0);
    size_t lds = (size_t)((hc + 3) & ~3u) * sizeof(uint32_t) + (LDSB ? (size_t)wpb * 2048 : 0);
    int per_cu = (int)std::min<size_t>(32 / wpb, (160 * 1024) / std::max<size_t>(lds, 1));
    {  // persistent waves pulling tiles from the queues: launch exactly what is resident at once
      int occ = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_scan_cells<CM, PP, CPT, BS, LDSB>, BS, lds) ==
              hipSuccess && occ >= 1)
        per_cu = std::min(per_cu, occ);
      else
        (void)hipGetLastError();
    }
    if (per_cu < 1) per_cu = 1;
    size_t blocks = std::min<size_t>((wtiles + wpb - 1) / wpb, (size_t)256 * per_cu);
    // a unit is a whole tile by default: splitting the hypothesis range of a tile into segments
    // (scan_hsplit) evens the load but every extra unit start measured ~20 us of exposed memory
    // waiting (2.4 ms instead of 1.6 ms at 4 segments)
    const size_t waves = blocks * wpb, groups = (hc + 63) / 64;
    uint32_t hsplit = 1;
    if (c->opt_hsplit > 0) hsplit = (uint32_t)std::min<size_t>(groups, (size_t)c->opt_hsplit);
    const size_t units = wtiles * hsplit;
    const uint32_t grab = (uint32_t)std::min<size_t>(8, std::max<size_t>(1, units / (32 * waves)));
    ProfScope ps(c, KID_SCAN);
    HIPCHK(c, hipMemsetAsync(d_next, 0, kQueues * kQueuePitch * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL((k_scan_cells<CM, PP, CPT, BS, LDSB>), dim3((unsigned)blocks), dim3(BS), lds,
                       c->stream, c->d_sorted, c->n_sorted, c->d_boxes, c->n_cells,
                       b.sp + h0 * M::SP,
                       CM::ROW_F32 ? b.spf + h0 * M::SPF : (const float *)(b.sp + h0 * M::SP),
                       b.spf + h0 * M::SPF, hc, c->mc, cc, b.votes + h0, d_next, grab, hsplit, b.h_dev);
    HIPCHK(c, hipGetLastError());
  }
  return LSQR_OK;
}
template <class CM, int PP, int CPT>
int run_scan_cells(lsqr_ctx *c, const ScanBatch &b) {
  // hypothesis broadcast to the survivors: v_readlane, or (scan_block 257 / the model's choice)
  // uniform-address LDS reads.  (1024-thread workgroups measured 3-5 % slower and are no longer built.)
  if (c->opt_block == 257 || (c->opt_block == 0 && CM::LDS_BROADCAST))
    return launch_scan_cells<CM, PP, CPT, 256, true>(c, b);
  return launch_scan_cells<CM, PP, CPT, 256>(c, b);
}
template <class CM, int PP, int CPT>
int run_scan_cells(lsqr_ctx *c) {
  const ScanBatch b = {c->d_hparams, c->d_hparams_f32, c->H, c->d_votes, nullptr};
  return run_scan_cells<CM, PP, CPT>(c, b);
}

// cell size -> packed pairs per lane (PP = cell / 128); 1024-point cells only where the cell model is built for them
template <class CM, class F>
int with_pp(uint32_t cell_pts, F &&f) {
  if constexpr (requires { CM::MAX_PP; }) {
    if (cell_pts == 1024) return f(std::integral_constant<int, 8>{});
  }
  if (cell_pts == 512) return f(std::integral_constant<int, 4>{});
  return f(std::integral_constant<int, 2>{});
}

// level 1 of the two-level scan alone over the current batch: d_ub[h] = vote bound, d_counter[4] = surviving pairs
template <class CM, int PP>
int run_cells_bounds(lsqr_ctx *c, uint32_t *d_ub, uint32_t *d_nc = nullptr) {
  typedef typename CM::M M;
  const CellConsts cc = cell_consts((const CM *)nullptr, c->mc);
  HIPCHK(c, hipMemsetAsync(d_ub, 0, c->H * sizeof(uint32_t), c->stream));
  if (d_nc) HIPCHK(c, hipMemsetAsync(d_nc, 0, c->H * sizeof(uint32_t), c->stream));
  if (d_nc) HIPCHK(c, hipMemsetAsync(c->d_counter + 4, 0, sizeof(unsigned long long), c->stream));
  if (c->n_cells == 0) return LSQR_OK;
  // The bounds of the bounded scan are taken on merged boxes (a quarter of the tests; the bound stays valid);
  // the diagnostics (d_nc: lsqr_scan_workload) count the surviving CELLS and stay on the cells.
  // ... as long as a few thousand boxes remain: on a small upload coarse boxes cost the selection its teeth (300 k
  // points in 147 boxes: 261 of 2048 hypotheses skipped instead of 1800)
  // ... and as long as the looser bound still prunes: a random plane cuts ~13 % of the cells but ~25 % of the merged
  // boxes, and with few inliers (80 % outliers and up) that population exceeds the best model's votes -- every
  // hypothesis would be counted.  The selection counts of earlier batches come back through pinned memory
  // (bsel_known: the latest batch the host has synchronised with; an older value only delays the switch): once a
  // second pass held more than a third of its batch, the bounds of this upload stay on the cells.
  uint32_t merge = c->opt_bound_merge ? (uint32_t)c->opt_bound_merge : (uint32_t)CM::BOUND_MERGE;
  if (!c->opt_bound_merge) {
    while (merge > 1 && c->n_cells / merge < 4096) merge /= 2;
    if (c->bsel_known_H && (uint64_t)c->bsel_known.n_rest * 3 > c->bsel_known_H) c->merge_off = true;
    if (c->merge_off) merge = 1;
  }
  if (d_nc || merge < 2 || c->n_cells < 4 * merge) merge = 1;
  const CellBox *boxes = c->d_boxes;
  uint32_t nbox = c->n_cells;
  if (merge > 1) {
    nbox = (c->n_cells + merge - 1) / merge;
    if (c->super_merge != merge) {
      hipLaunchKernelGGL((k_super_boxes<M::ND>), dim3((nbox + 255) / 256), dim3(256), 0, c->stream, c->d_boxes,
                         c->n_cells, c->d_boxes + c->n_cells, nbox, merge);
      HIPCHK(c, hipGetLastError());
      c->super_merge = merge;
    }
    boxes = c->d_boxes + c->n_cells;
  }
  const unsigned gy = (unsigned)((c->H + 255) / 256);
  unsigned gx = std::max(1u, std::min<unsigned>(nbox, 2048u / gy));
  const uint32_t per = (nbox + gx - 1) / gx;
  gx = (nbox + per - 1) / per;
  hipLaunchKernelGGL((k_cells_bounds<CM, PP>), dim3(gx, gy), dim3(256), 0, c->stream, boxes, nbox, c->n_sorted,
                     CM::ROW_F32 ? c->d_hparams_f32 : (const float *)c->d_hparams, c->d_hparams_f32,
                     (uint32_t)c->H, cc, per, d_ub, d_nc ? c->d_counter + 4 : (unsigned long long *)nullptr, d_nc,
                     (uint8_t *)nullptr, 0u, (const uint32_t *)nullptr, 0u,  // (pair total: diagnostics only -- 8192
                     (uint32_t)(128 * PP) * merge);                          // atomics on one address are ~100 us)
  HIPCHK(c, hipGetLastError());
  return LSQR_OK;
}

// Level 2 of a (compacted) batch in statically balanced pieces (cells.h, "statically balanced level 2"): count the
// survivors per (cell, group), sum them per cell and per chunk, then k_scan_pairs.  Chained on the stream.
template <class CM, int PP>
int run_scan_pairs(lsqr_ctx *c, const ScanBatch &b0) {
  typedef typename CM::M M;
  // k_scan_pairs holds a cell's groups in the 64 lanes of a wave: batches of more than 4096 hypotheses are counted
  // 4096 at a time (rare: the bounded scan's second pass is sized for the whole batch but holds ~1/8 of it, so the
  // later launches find an empty cost table and return at once)
  constexpr size_t kPairsChunk = 4096;  // 64 groups of 64
  if (b0.H > kPairsChunk) {
    for (size_t h0 = 0; h0 < b0.H; h0 += kPairsChunk) {
      ScanBatch sub = {b0.sp + h0 * M::SP, b0.spf + h0 * M::SPF, std::min<size_t>(kPairsChunk, b0.H - h0),
                       b0.votes + h0, b0.h_dev, (uint32_t)(b0.h_off + h0)};
      int st = run_scan_pairs<CM, PP>(c, sub);
      if (st != LSQR_OK) return st;
    }
    return LSQR_OK;
  }
  const ScanBatch &b = b0;
  const CellConsts cc = cell_consts((const CM *)nullptr, c->mc);
  if (c->n_cells == 0 || b.H == 0) {
    HIPCHK(c, hipMemsetAsync(b.votes, 0, b.H * sizeof(uint32_t), c->stream));
    return LSQR_OK;
  }
  const uint32_t Hc = (uint32_t)b.H, groups = (Hc + 63) / 64;
  const uint32_t gstride = groups <= 1 ? 1u : (groups + 15) / 16 * 16;  // 16-byte rows for k_tile_costs
  const uint32_t nchunks = (c->n_cells + kChunkCells - 1) / kChunkCells;
  int st;
  if ((st = ensure(c, &c->d_paircnt, &c->paircnt_cap, (size_t)c->n_cells * gstride)) != LSQR_OK) return st;
  if ((st = ensure(c, &c->d_paircost, &c->paircost_cap, (size_t)c->n_cells + nchunks)) != LSQR_OK) return st;
  uint32_t *d_cost = c->d_paircost, *d_csum = c->d_paircost + c->n_cells;
  const float *rows = CM::ROW_F32 ? b.spf : (const float *)b.sp;
  {  // counting pass: waves past the device-side H leave at once, so the grid is cut finely in x
    const unsigned gy = (Hc + 255) / 256;
    const uint32_t per = std::max<uint32_t>(8, (c->n_cells + 1023) / 1024);
    const unsigned gx = (c->n_cells + per - 1) / per;
    hipLaunchKernelGGL((k_cells_bounds<CM, PP>), dim3(gx, gy), dim3(256), 0, c->stream, c->d_boxes, c->n_cells,
                       c->n_sorted, rows, b.spf, Hc, cc, per, (uint32_t *)nullptr, (unsigned long long *)nullptr,
                       (uint32_t *)nullptr, c->d_paircnt, gstride, b.h_dev, b.h_off, (uint32_t)(128 * PP));
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(k_tile_costs, dim3(nchunks), dim3(kChunkCells), 0, c->stream, c->d_paircnt, gstride, Hc, b.h_dev,
                     c->n_cells, d_cost, d_csum, b.votes, b.h_off);  // (zeroes the batch's votes)
  HIPCHK(c, hipGetLastError());
  bool ldsb_default = CM::LDS_BROADCAST;
  if constexpr (requires { CM::LDS_BROADCAST_PAIRS; }) ldsb_default = CM::LDS_BROADCAST_PAIRS;
  const bool ldsb = c->opt_block == 257 || (c->opt_block == 0 && ldsb_default);
  constexpr int BS = 256, wpb = BS / 64;
  const size_t lds = (size_t)((Hc + 3) & ~3u) * sizeof(uint32_t) + (ldsb ? (size_t)wpb * 2048 : 0);
  auto launch = [&](auto kern) -> int {
    int per_cu = (int)std::min<size_t>(32 / wpb, (160 * 1024) / std::max<size_t>(lds, 1)), occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, BS, lds) == hipSuccess && occ >= 1)
      per_cu = std::min(per_cu, occ);
    else
      (void)hipGetLastError();
    if (per_cu < 1) per_cu = 1;
    if (c->opt_pairs_waves > 0) per_cu = std::min(per_cu, c->opt_pairs_waves);
    const unsigned blocks = (unsigned)std::min<size_t>(((size_t)c->n_cells + wpb - 1) / wpb, (size_t)256 * per_cu);
    int st2 = ensure(c, &c->d_vpart, &c->vpart_cap, (size_t)blocks * Hc);
    if (st2 != LSQR_OK) return st2;
    ProfScope ps(c, KID_SCAN);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BS), lds, c->stream, c->d_sorted, c->n_sorted, c->d_boxes, c->n_cells,
                       b.sp, rows, b.spf, Hc, c->mc, cc, c->d_vpart, Hc, b.h_dev, (const uint8_t *)c->d_paircnt, gstride,
                       (const uint32_t *)d_cost, (const uint32_t *)d_csum, nchunks, b.h_off);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_votes_reduce, dim3((Hc + 63) / 64, 48), dim3(256), 0, c->stream,
                       (const uint32_t *)c->d_vpart, Hc, (uint32_t)blocks, Hc, b.h_dev, b.votes, b.h_off);
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  };
  return ldsb ? launch(k_scan_pairs<CM, PP, BS, true>) : launch(k_scan_pairs<CM, PP, BS, false>);
}

// Vote bounds by rank (axis.h; plane in 3-D over axis-sorted cells) on top of the box-population bounds in d_ub:
// candidates -> their upper AND lower vote bounds by rank (k_bound_axis: no observation is evaluated) -> d_ub refined
// in place, *lo_out = per-hypothesis lower bounds for the selections of the bounded scan.
template <int PP>
int run_rank_bounds(lsqr_ctx *c, const uint32_t **lo_out) {
  typedef PlaneCell<3> CM;
  typedef typename CM::M M;
  const uint32_t H = (uint32_t)c->H;
  int st;
  if (!c->d_ub2) HIPCHK(c, hipMalloc((void **)&c->d_ub2, 3 * c->H_cap * sizeof(uint32_t)));  // [ub2 | lb2 | lo]
  uint32_t *ub2 = c->d_ub2, *lb2 = c->d_ub2 + c->H_cap, *lo = c->d_ub2 + 2 * c->H_cap;
  const CellConsts cc = cell_consts((const CM *)nullptr, c->mc);
  // (the candidate list and their rows use the second pass's areas: free until the pilots have been counted)
  uint32_t *cand = c->d_sel + kPilots;
  double *sp_b = c->d_hparams2 + (size_t)kPilots * M::SP;
  float *spf_b = c->d_hparams2_f32 + (size_t)kPilots * M::SPF;
  hipLaunchKernelGGL(k_pick_cands, dim3(1), dim3(1024), 0, c->stream, c->d_ub, c->d_valid, H, c->best_before, cand,
                     c->d_bsel, c->d_votes, ub2, lb2, lo);
  hipLaunchKernelGGL(k_gather_rows, dim3((H + 3) / 4), dim3(256), 0, c->stream, cand, &c->d_bsel->n_cand, H,
                     c->d_hparams, (int)M::SP, c->d_hparams_f32, (int)M::SPF, sp_b, spf_b);
  HIPCHK(c, hipGetLastError());
  const unsigned gy = (H + 511) / 512;  // (workgroups past the device-side candidate count return at once)
  const uint32_t per = std::max<uint32_t>(8, ((c->n_cells + 3071) / 3072 + 7) / 8 * 8);
  const unsigned gx = (c->n_cells + per - 1) / per;
  float thr_up = (float)c->mc.thr, thr_dn = thr_up;
  if ((double)thr_up < c->mc.thr) thr_up = nextafterf(thr_up, INFINITY);
  if ((double)thr_dn > c->mc.thr) thr_dn = nextafterf(thr_dn, -INFINITY);
  float xabs = (float)c->mc.absmax;
  if ((double)xabs < c->mc.absmax) xabs = nextafterf(xabs, INFINITY);
  hipLaunchKernelGGL((k_bound_axis<PP>), dim3(gx, gy), dim3(512), 0, c->stream, c->d_boxes, c->d_axis, c->d_cellT,
                     c->n_sorted, c->n_cells, (const float *)sp_b, H, &c->d_bsel->n_cand, cc, thr_up, thr_dn, xabs, per,
                     ub2, lb2);
  hipLaunchKernelGGL(k_refine_bounds, dim3((H + 255) / 256), dim3(256), 0, c->stream, cand, c->d_bsel, ub2, lb2,
                     c->d_ub, lo);
  HIPCHK(c, hipGetLastError());
  (void)st;
  *lo_out = lo;
  return LSQR_OK;
}

// The bounded scan of the current batch over the index (cells.h, "bounded scan"): bounds, pilots counted exactly,
// then only the hypotheses that can still become the running maximum.  Everything is chained on the stream.
template <class CM, int PP>
int run_scan_bounded(lsqr_ctx *c) {
  typedef typename CM::M M;
  // profiling: ONE scope over the whole scan phase (bounds, selections, both counting launches)
  ProfScope whole(c, KID_SCAN);
  struct Mute {
    lsqr_ctx *c;
    bool was;
    ~Mute() { c->prof = was; }
  } mute{c, c->prof};
  c->prof = false;
  // the record this scan will write: whatever it still holds (the scan two back) has to have landed before it is
  // reused, and is folded here at the latest
  if (!c->h_bsel) {
    HIPCHK(c, hipHostMalloc((void **)&c->h_bsel, 128));
    memset(c->h_bsel, 0, 128);
    for (int r = 0; r < 2; r++) HIPCHK(c, hipEventCreateWithFlags(&c->bsel_ev[r], hipEventDisableTiming));
  }
  const int rec = (int)(c->bsel_seq & 1);
  if (c->bsel_pending[rec]) {
    HIPCHK(c, hipEventSynchronize(c->bsel_ev[rec]));
    bsel_fold(c, rec);
  }
  int st = run_cells_bounds<CM, PP>(c, c->d_ub);
  if (st != LSQR_OK) return st;
  const uint32_t H = (uint32_t)c->H;
  uint32_t *sel_a = c->d_sel, *sel_b = c->d_sel + kPilots;
  uint32_t *votes_a = c->d_votes2, *votes_b = c->d_votes2 + kPilots;
  double *sp_a = c->d_hparams2, *sp_b = c->d_hparams2 + (size_t)kPilots * M::SP;
  float *spf_a = c->d_hparams2_f32, *spf_b = c->d_hparams2_f32 + (size_t)kPilots * M::SPF;
  const uint32_t *lo = nullptr;  // lower vote bounds (rank bounds of the plane), or none
  if constexpr (std::is_same<CM, PlaneCell<3>>::value && (PP == 4 || PP == 2)) {
    if (c->axis_valid && c->opt_axis && c->cell_pts == (uint32_t)(128 * PP))
      if ((st = run_rank_bounds<PP>(c, &lo)) != LSQR_OK) return st;
  }
  // The pilot pass is five launches that find nothing to do when a lower bound of the running maximum is known
  // without pilots (rank bounds, or the best of earlier batches).  Whether it was is reported back through pinned
  // memory (bsel_known: the latest batch the host has synchronised with -- one or two batches back): after a batch of
  // this upload and model that needed no pilots the pass is not launched -- k_pick_pilots then selects none; if that
  // batch would have needed them after all, its second pass counts more hypotheses (never wrongly: L[h] is a lower
  // bound either way) and a later batch gets its pilots back.
  const bool no_pilots = c->bsel_known_H == H && c->bsel_known.known != 0;
  hipLaunchKernelGGL(k_pick_pilots, dim3(1), dim3(1024), 0, c->stream, c->d_ub, c->d_valid, H, sel_a, c->d_bsel,
                     c->d_votes, c->best_before, lo, no_pilots ? 1 : 0);  // (also zeroes the batch's votes)
  if (!no_pilots) {
    hipLaunchKernelGGL(k_gather_rows, dim3(kPilots / 4), dim3(256), 0, c->stream, sel_a, &c->d_bsel->n_pilot,
                       (uint32_t)kPilots, c->d_hparams, (int)M::SP, c->d_hparams_f32, (int)M::SPF, sp_a, spf_a);
    HIPCHK(c, hipGetLastError());
    const ScanBatch pa = {sp_a, spf_a, (size_t)kPilots, votes_a, &c->d_bsel->n_pilot};
    if ((st = run_scan_pairs<CM, PP>(c, pa)) != LSQR_OK) return st;
  }

  hipLaunchKernelGGL(k_pick_rest, dim3(1), dim3(1024), 0, c->stream, c->d_ub, c->d_valid, H, sel_a, votes_a,
                     c->best_before, sel_b, c->d_bsel, lo);
  hipLaunchKernelGGL(k_gather_rows, dim3((H + 3) / 4), dim3(256), 0, c->stream, sel_b, &c->d_bsel->n_rest, H,
                     c->d_hparams, (int)M::SP, c->d_hparams_f32, (int)M::SPF, sp_b, spf_b);
  HIPCHK(c, hipGetLastError());
  const ScanBatch pb = {sp_b, spf_b, (size_t)H, votes_b, &c->d_bsel->n_rest};
  if ((st = run_scan_pairs<CM, PP>(c, pb)) != LSQR_OK) return st;

  hipLaunchKernelGGL(k_scatter_votes, dim3((H + 255) / 256 + 1), dim3(256), 0, c->stream, sel_a, &c->d_bsel->n_pilot,
                     votes_a, sel_b, &c->d_bsel->n_rest, votes_b, c->d_votes);
  HIPCHK(c, hipGetLastError());
  c->last_bound[0] = 1;
  c->last_bound[3] = H;
  HIPCHK(c, hipMemcpyAsync(bsel_rec(c, rec), c->d_bsel, sizeof(BoundSel), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipEventRecord(c->bsel_ev[rec], c->stream));
  c->bsel_pending[rec] = true;
  c->bsel_seq_of[rec] = ++c->bsel_seq;
  c->bsel_H_of[rec] = H;
  c->bsel_last_rec = rec;
  return LSQR_OK;
}

// ---- chunked early exit (earlyexit.h) ---------------------------------------------------------------------
constexpr size_t kEeCap = kSelCap;  // hypotheses per batch the selection kernels handle (models.h)
static_assert(kEeCap <= kSelCap && kScanChunk <= kSelCap, "selection kernels: 1024 threads x 8 hypotheses");
struct EeBuf {
  EeState *st;
  uint32_t *sel_c, *sel_o[2];
  float *thr_c, *rows_c;
};
int ee_buffers(lsqr_ctx *c, EeBuf *b) {
  const size_t bytes = 256 + 3 * kEeCap * sizeof(uint32_t) + (4 + 64) * (kEeCap + 64) * sizeof(float);
  if (!c->d_ee) HIPCHK(c, hipMalloc(&c->d_ee, bytes));
  if (!c->h_ee) {
    HIPCHK(c, hipHostMalloc((void **)&c->h_ee, 64));
    memset(c->h_ee, 0, 64);
  }
  char *p = (char *)c->d_ee;
  b->st = (EeState *)p;
  b->sel_c = (uint32_t *)(p + 256);
  b->sel_o[0] = b->sel_c + kEeCap;
  b->sel_o[1] = b->sel_o[0] + kEeCap;
  b->thr_c = (float *)(b->sel_o[1] + kEeCap);
  b->rows_c = b->thr_c + 4 * (kEeCap + 64);  // (the fp16 filter keeps 4 floats per hypothesis, the fp32 filter 2)
  return LSQR_OK;
}
// scan(row_begin, row_end, range_dev, h_dev, sel): sel == null -> the context's whole batch over [row_begin, row_end),
// else the compact selection the last gather(sel, n_dev) produced, over the device-side range range_dev (launch sized
// for [row_begin, row_end)).  Everything is chained on the stream; no host round trip.
template <class Scan, class Gather>
int run_early_exit(lsqr_ctx *c, size_t align, const EeBuf &b, Scan &&scan, Gather &&gather) {
  const size_t n = c->n;
  const uint32_t H = (uint32_t)c->H;
  size_t b1 = (n / 16 + align - 1) / align * align;
  if (b1 > n) b1 = n;
  int st;
  HIPCHK(c, hipMemsetAsync(c->d_votes, 0, H * sizeof(uint32_t), c->stream));
  // A: the first sixteenth, every hypothesis
  if ((st = scan(0, b1, (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr)) != LSQR_OK)
    return st;
  hipLaunchKernelGGL(k_ee_split, dim3(1), dim3(1024), 0, c->stream, c->d_votes, c->d_valid, H, b.sel_c, b.sel_o[0], b.st,
                     (uint32_t)b1, (uint32_t)n);
  HIPCHK(c, hipGetLastError());
  if (b1 < n) {
    // B: the candidates to the end
    if ((st = gather(b.sel_c, &b.st->n_cand)) != LSQR_OK) return st;
    if ((st = scan(b1, n, &b.st->rng[0][0], &b.st->n_cand, b.sel_c)) != LSQR_OK) return st;
    hipLaunchKernelGGL(k_ee_plan, dim3(1), dim3(1024), 0, c->stream, c->d_votes, b.sel_c, b.st, c->best_before,
                       (uint32_t)b1, (uint32_t)n, (uint32_t)align);
    HIPCHK(c, hipGetLastError());
    // C: the others over the planned ranges
    int cur = 0;
    for (int k = 1; k <= kEeChunksC; k++) {
      if ((st = gather(b.sel_o[cur], &b.st->n_alive)) != LSQR_OK) return st;
      if ((st = scan(b1, n, &b.st->rng[k][0], &b.st->n_alive, b.sel_o[cur])) != LSQR_OK) return st;
      if (k < kEeChunksC) {
        hipLaunchKernelGGL(k_ee_select, dim3(1), dim3(1024), 0, c->stream, c->d_votes, c->d_valid, H, (uint32_t)n, k,
                           c->best_before, b.sel_o[cur], b.sel_o[cur ^ 1], b.st);
        HIPCHK(c, hipGetLastError());
        cur ^= 1;
      }
    }
  }
  HIPCHK(c, hipMemcpyAsync(c->h_ee, b.st, sizeof(EeState), hipMemcpyDeviceToHost, c->stream));
  c->ee_last = true;
  c->ee_H = H;
  c->ee_n = n;
  return LSQR_OK;
}

// once per context: does this device's matrix unit align the products of one instruction as the fp16 filters'
// thresholds assume (dense_h16.h: k_dense_h16_probe)?  c->h16_unit = 1 / -1.
int h16_probe_unit(lsqr_ctx *c) {
  if (c->h16_unit != 0) return LSQR_OK;
  if (!c->d_h16_thr) HIPCHK(c, hipMalloc((void **)&c->d_h16_thr, sizeof(float) * 4 * 8192));
  float *d_probe = c->d_h16_thr;  // (the batch's threshold block: not in use yet)
  hipLaunchKernelGGL(k_dense_h16_probe, dim3(1), dim3(64), 0, c->stream, d_probe);
  // ... and 64 instructions of random operands (65 536 sums of 16 products and an addend against fp64)
  hipLaunchKernelGGL(k_dense_h16_probe_random, dim3(1), dim3(64), 0, c->stream, d_probe + kH16ProbeVariants, 64);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pin, d_probe, sizeof(float) * (kH16ProbeVariants + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  c->h16_unit_dev = dense_h16_probe_worst((const float *)c->h_pin);
  {
    const double rnd = (double)((const float *)c->h_pin)[kH16ProbeVariants];
    if (!(rnd <= c->h16_unit_dev)) c->h16_unit_dev = rnd;  // (a NaN is the worst)
  }
  c->h16_unit = c->h16_unit_dev <= kH16ProbeLimit ? 1 : -1;
  if (c->h16_unit < 0)
    (void)fail(c, LSQR_OK, "fp16 matrix-core filters: this device's matrix unit loses %.1f u per instruction (limit %.1f): fp32 filter used",
               c->h16_unit_dev, kH16ProbeLimit);
  return LSQR_OK;
}

// US calibrations: the frames as fp16 fragment pairs for us_h16.h's filter, once per upload (1 M frames: 192 MB).
// *ok = false when the filter cannot be used (magnitudes, the device's matrix unit): the packed fp32 filter scans.
template <class M>
constexpr bool kHasUsH16 = M::IS_US || requires { M::IS_PHANTOM; };
template <class M>
int ensure_us_h16(lsqr_ctx *c, bool *ok) {
  *ok = false;
  constexpr bool PH = requires { M::IS_PHANTOM; };
  constexpr bool SINGLE = PH || M::K == 4;
  const double X = c->mc.absmax, Rm = c->mc.absmax_rot;
  if (!c->absmax_valid || !(X > 0.0) || !(X < 1e15) || !(Rm > 0.0) || !(Rm < 1e15) || c->us16_nomem) return LSQR_OK;
  int st = h16_probe_unit(c);
  if (st != LSQR_OK) return st;
  if (c->h16_unit < 0) return LSQR_OK;
  const Us16Scales sc = us16_scales<SINGLE>(X, Rm);
  if (!c->us16_valid || memcmp(&sc, &c->us16_sc, sizeof sc) != 0) {
    const size_t n_tiles = (c->n + 31) / 32 + 2;
    if (n_tiles > c->us16_tiles_cap) {
      if (c->d_us16) (void)hipFree(c->d_us16);
      c->d_us16 = nullptr, c->us16_tiles_cap = 0;
      if (hipMalloc((void **)&c->d_us16, n_tiles * (size_t)kUs16FrameTile) != hipSuccess) {
        (void)hipGetLastError();  // (no room for the fragments: not an error -- the packed fp32 filter scans the records)
        c->d_us16 = nullptr;
        c->us16_nomem = true;
        return LSQR_OK;
      }
      c->us16_tiles_cap = n_tiles;
    }
    if constexpr (PH)  // (4 KiB per 32 frames of the 6 KiB the allocation holds)
      hipLaunchKernelGGL(k_phantom_rows_h16, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, c->stream, c->d_data, c->stride,
                         c->n, sc, c->d_us16, n_tiles);
    else
      hipLaunchKernelGGL((k_us_rows_h16<SINGLE>), dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, c->stream, c->d_data,
                         c->stride, c->n, sc, c->d_us16, n_tiles);
    HIPCHK(c, hipGetLastError());
    c->us16_valid = true;
    c->us16_sc = sc;
  }
  if (!c->d_us16_x) HIPCHK(c, hipMalloc((void **)&c->d_us16_x, (size_t)(8192 / 32) * 4096));
  if (!c->d_h16_thr) HIPCHK(c, hipMalloc((void **)&c->d_h16_thr, sizeof(float) * 4 * 8192));
  if (!c->d_amb) HIPCHK(c, hipMalloc((void **)&c->d_amb, sizeof(unsigned long long) * kAmbCap));
  if constexpr (PH)  // (every time: one flag per context served two kernels and missed the second)
    (void)hipFuncSetAttribute((const void *)k_scan_phantom_h16, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)phantom_h16_lds(kPh16HypChunk));
  else
    (void)hipFuncSetAttribute((const void *)k_scan_us_h16<SINGLE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)us_h16_lds(kUs16HypChunk));
  *ok = true;
  return LSQR_OK;
}
// frames [rb, re) against the H hypotheses whose scan parameters are `sp` (the batch or a compact selection): split of
// the unknowns, the filter in launches of 1024 hypotheses, the exact decision of the band
template <class M>
int launch_us_h16(lsqr_ctx *c, size_t rb, size_t re, const double *sp, uint32_t H, unsigned int *d_segcnt, uint32_t seg_cap,
                  const uint32_t *h_dev, const uint32_t *sel, const uint32_t *range_dev) {
  constexpr bool PH = requires { M::IS_PHANTOM; };
  constexpr bool SINGLE = PH || M::K == 4;
  if constexpr (PH)
    hipLaunchKernelGGL(k_phantom_prep_h16, dim3((H + 31 + 255) / 256), dim3(256), 0, c->stream, sp, (int)M::SP, H,
                       sqrt(c->mc.delta_sq), c->mc.absmax, c->mc.absmax_rot, c->us16_sc, c->d_us16_x, c->d_h16_thr);
  else
    hipLaunchKernelGGL((k_us_prep_h16<SINGLE>), dim3((H + 31 + 255) / 256), dim3(256), 0, c->stream, sp, (int)M::SP, H,
                       c->mc.delta_sq, c->mc.absmax, c->mc.absmax_rot, c->us16_sc, c->d_us16_x, c->d_h16_thr);
  HIPCHK(c, hipGetLastError());
  const size_t passes = (re - rb + kUs16Wg - 1) / kUs16Wg;
  const unsigned nblk = (unsigned)std::min<size_t>(passes, 256);  // one workgroup (eight waves) per CU
  for (size_t h0 = 0; h0 < H; h0 += kUs16HypChunk) {
    const uint32_t hc = (uint32_t)std::min<size_t>(kUs16HypChunk, H - h0);
    if constexpr (PH)
      hipLaunchKernelGGL(k_scan_phantom_h16, dim3(nblk), dim3(kPh16Wg), phantom_h16_lds(hc), c->stream, c->d_us16, c->n, rb, re,
                         c->d_us16_x + (h0 / 32) * 256, c->d_h16_thr + 4 * h0, hc, c->d_votes, c->d_amb, d_segcnt, seg_cap,
                         (uint32_t)h0, h_dev, sel, range_dev);
    else
      hipLaunchKernelGGL((k_scan_us_h16<SINGLE>), dim3(nblk), dim3(kUs16Wg), us_h16_lds(hc), c->stream, c->d_us16, c->n, rb,
                         re, c->d_us16_x + (h0 / 32) * 128, c->d_h16_thr + 4 * h0, hc, c->d_votes, c->d_amb, d_segcnt, seg_cap,
                         (uint32_t)h0, h_dev, sel, range_dev);
    HIPCHK(c, hipGetLastError());
    // the exact decision of the band: the phantom after every launch (its 31-term sums leave ~3e-4 of the pairs there,
    // a workgroup's segment holds one launch's share, not four), the calibrations once
    if (PH || h0 + kUs16HypChunk >= H) {
      hipLaunchKernelGGL((k_us_recheck_seg<M>), dim3(256), dim3(1024), 0, c->stream, c->d_data, c->stride, c->d_hparams,
                         (int)M::SP, c->mc, c->d_amb, d_segcnt, seg_cap, c->d_votes, (unsigned int *)(c->d_counter + 3));
      HIPCHK(c, hipGetLastError());
    }
  }
  return LSQR_OK;
}

// dense system, n = 64: the rows as fp16 fragment pairs for dense_h16.h's filter, once per upload (2 M x 64: 0.5 GB,
// ~1.9 ms).  *ok = false when the magnitudes do not fit the filter's scaling (the fp32 filter takes the scan).
int ensure_dense_h16(lsqr_ctx *c, bool *ok) {
  *ok = false;
  const double amax = c->mc.absmax_rot, bmax = c->mc.absmax;
  if (!(amax > 0.0) || !(amax < 1e15) || !(bmax < 1e15)) return LSQR_OK;
  const double pa = 32768.0 / amax;
  if (!(pa < 1e30) || !(bmax * pa < 1e18) || c->h16_nomem) return LSQR_OK;
  int stp = h16_probe_unit(c);
  if (stp != LSQR_OK) return stp;
  if (c->h16_unit < 0) return LSQR_OK;
  if (!c->h16_valid || c->h16_pa != pa) {
    const size_t n_tiles = (c->n + 31) / 32 + 8;  // a workgroup pass reads up to 255 rows past its last one
    if (n_tiles > c->h16_tiles_cap) {
      if (c->d_h16) (void)hipFree(c->d_h16);
      if (c->d_h16_bs) (void)hipFree(c->d_h16_bs);
      c->d_h16 = nullptr, c->d_h16_bs = nullptr, c->h16_tiles_cap = 0;
      // (no room for the second copy of the rows: not an error -- the fp32 filter scans the fp64 rows)
      if (hipMalloc((void **)&c->d_h16, n_tiles * (size_t)kH16TileBytes) != hipSuccess ||
          hipMalloc((void **)&c->d_h16_bs, n_tiles * 32 * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        if (c->d_h16) (void)hipFree(c->d_h16);
        c->d_h16 = nullptr, c->d_h16_bs = nullptr;
        c->h16_nomem = true;
        return LSQR_OK;
      }
      c->h16_tiles_cap = n_tiles;
    }
    hipLaunchKernelGGL(k_dense_rows_h16, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, c->stream, c->d_data,
                       c->stride, c->n, (int)c->cfg.dim, pa, c->d_h16, c->d_h16_bs, n_tiles);
    HIPCHK(c, hipGetLastError());
    c->h16_valid = true;
    c->h16_pa = pa;
  }
  if (!c->d_h16_thr) HIPCHK(c, hipMalloc((void **)&c->d_h16_thr, sizeof(float) * 4 * 8192));
  if (!c->h16_attr) {  // (per context: the attribute belongs to the device the context lives on)
    c->h16_attr = true;
    (void)hipFuncSetAttribute((const void *)k_scan_dense_h16<64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)dense_h16_lds(1024));
  }
  *ok = true;
  return LSQR_OK;
}
// rows [rb, re) against hypotheses of `xh` / `thr4` (the batch itself or a compact selection), in launches of 1024
int launch_dense_h16(lsqr_ctx *c, size_t rb, size_t re, const _Float16 *xh, const float *thr4, uint32_t H,
                     unsigned int *d_segcnt, uint32_t seg_cap, const uint32_t *h_dev, const uint32_t *sel,
                     const uint32_t *range_dev, size_t *nblk_out) {
  const size_t passes = (re - rb + kH16RowsPerWg - 1) / kH16RowsPerWg;
  const size_t nb = std::min<size_t>(passes, 512);  // two workgroups per CU
  const size_t rpb = (passes + nb - 1) / nb * kH16RowsPerWg;
  const size_t nblk = (re - rb + rpb - 1) / rpb;
  if (nblk_out) *nblk_out = nblk;
  for (size_t h0 = 0; h0 < H; h0 += 1024) {
    const uint32_t hc = (uint32_t)std::min<size_t>(1024, H - h0);
    hipLaunchKernelGGL((k_scan_dense_h16<64>), dim3((unsigned)nblk), dim3(256), dense_h16_lds(hc), c->stream, c->d_h16,
                       c->d_h16_bs, rb, re, rpb, xh + h0 * 128, thr4 + 4 * h0, hc, c->d_votes, c->d_amb, d_segcnt,
                       seg_cap, (uint32_t)h0, h_dev, sel, range_dev);
    HIPCHK(c, hipGetLastError());
  }
  return LSQR_OK;
}

// diagnostics (LSQR_DENSE_DEBUG): how full the dense scan's worklist got (the exact kernel has emptied the segments by
// now: the fullest one is what it recorded; a launch's total is about that times the filter's workgroups)
static void dense_worklist_debug(lsqr_ctx *c, const unsigned int *, uint32_t seg_cap) {
  static const bool dbg_on = getenv("LSQR_DENSE_DEBUG") != nullptr;
  if (dbg_on)
    fprintf(stderr, "dense scan: fullest worklist segment %u of %u (%zu rows x %zu hypotheses)\n", c->dense_amb_max, seg_cap,
            (size_t)c->n, (size_t)c->H);
}

// dense system, n > 32: the fp32 matrix-core filter (dense.h: k_scan_dense_mfma32r) over row chunks and compacted
// selections; the band of every chunk is decided exactly (k_dense_recheck_seg) before the next selection looks at the
// votes.  Returns LSQR_OK with *done = false when a worklist segment overflowed (the caller counts everything with
// the fp64 filter instead).
int run_scan_dense_ee(lsqr_ctx *c, bool *done) {
  *done = false;
  EeBuf b;
  int st = ee_buffers(c, &b);
  if (st != LSQR_OK) return st;
  const uint32_t H = (uint32_t)c->H;
  const uint32_t seg_cap = kAmbCap / 1024;
  float *d_thr32 = (float *)c->d_partials;
  float *d_sp32 = (float *)c->d_partials + 2 * 8192;
  unsigned int *d_segcnt = (unsigned int *)((float *)c->d_partials + 2 * 8192 + 64 * 8192);
  ProfScope whole(c, KID_SCAN);
  struct Mute {
    lsqr_ctx *c;
    bool was;
    ~Mute() { c->prof = was; }
  } mute{c, c->prof};
  c->prof = false;
  HIPCHK(c, hipMemsetAsync(d_segcnt, 0, 1024 * sizeof(unsigned int), c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), c->stream));
  bool h16 = false;
  if (c->opt_dense_f32 == 2 && (st = ensure_dense_h16(c, &h16)) != LSQR_OK) return st;
  if (h16)  // the fp16 filter's rows live where the fp32 filter keeps its own (256 B per hypothesis either way)
    hipLaunchKernelGGL(k_dense_prep_h16, dim3((H + 3) / 4), dim3(256), 0, c->stream, c->d_hparams, H, (int)c->cfg.dim,
                       64, c->mc.delta, c->mc.absmax_rot, c->mc.absmax, c->h16_pa, (_Float16 *)d_sp32, c->d_h16_thr);
  else
    hipLaunchKernelGGL(k_dense_thresholds32, dim3((H + 255) / 256), dim3(256), 0, c->stream, c->d_hparams, H,
                       (int)c->cfg.dim, 64, c->mc.delta, c->mc.absmax_rot, c->mc.absmax, d_thr32, d_sp32);
  HIPCHK(c, hipGetLastError());
  auto scan = [&](size_t rb, size_t re, const uint32_t *range_dev, const uint32_t *h_dev, const uint32_t *sel) -> int {
    if (rb >= re) return LSQR_OK;
    if (h16) {
      int s2 = launch_dense_h16(c, rb, re, (const _Float16 *)(sel ? b.rows_c : d_sp32), sel ? b.thr_c : c->d_h16_thr, H,
                                d_segcnt, seg_cap, h_dev, sel, range_dev, nullptr);
      if (s2 != LSQR_OK) return s2;
      hipLaunchKernelGGL((k_dense_recheck_seg<64>), dim3(512), dim3(256), 0, c->stream, c->d_data, c->stride,
                         c->d_hparams, c->mc, c->d_amb, d_segcnt, seg_cap, c->d_votes, (unsigned int *)(c->d_counter + 3));
      HIPCHK(c, hipGetLastError());
      return LSQR_OK;
    }
    const size_t tiles = (re - rb + 63) / 64;
    const size_t nb2 = std::min<size_t>(tiles, 512);  // two workgroups per CU
    const size_t rpb = (tiles + nb2 - 1) / nb2 * 64;
    const size_t nblk = (re - rb + rpb - 1) / rpb;
    const float *rows = sel ? b.rows_c : d_sp32, *thr = sel ? b.thr_c : d_thr32;
    constexpr size_t kRingChunk = 1024;  // 62.7 KiB of LDS per workgroup: two per CU
    for (size_t h0 = 0; h0 < H; h0 += kRingChunk) {
      const uint32_t hc = (uint32_t)std::min<size_t>(kRingChunk, H - h0);
      const uint32_t nhb2 = (((hc + 63) / 64) + 1) & ~1u;
      const size_t lds = sizeof(float) * (8192 + 64 * kDmPitch32 + 64 + 128 * nhb2) + sizeof(uint32_t) * (hc + 1);
      hipLaunchKernelGGL((k_scan_dense_mfma32r<64>), dim3((unsigned)nblk), dim3(256), lds, c->stream, c->d_data,
                         c->stride, rb, re, rpb, rows + h0 * 64, thr + 2 * h0, hc, (int)c->cfg.dim, c->d_votes,
                         c->d_amb, d_segcnt, seg_cap, (uint32_t)h0, h_dev, sel, range_dev);
      HIPCHK(c, hipGetLastError());
    }
    hipLaunchKernelGGL((k_dense_recheck_seg<64>), dim3(512), dim3(256), 0, c->stream, c->d_data, c->stride,
                       c->d_hparams, c->mc, c->d_amb, d_segcnt, seg_cap, c->d_votes, (unsigned int *)(c->d_counter + 3));
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  };
  auto gather = [&](const uint32_t *sel, const uint32_t *n_dev) -> int {
    if (h16)  // (positions past the selection are never read: the scan takes its size from *n_dev)
      hipLaunchKernelGGL(k_ee_gather_f32, dim3((H + 3) / 4), dim3(256), 0, c->stream, sel, n_dev, H,
                         (const float *)d_sp32, 64, b.rows_c, (const float *)c->d_h16_thr, 4, b.thr_c, 0.0f);
    else
    hipLaunchKernelGGL(k_ee_gather_f32, dim3((H + 3) / 4), dim3(256), 0, c->stream, sel, n_dev, H,
                       (const float *)d_sp32, 64, b.rows_c, (const float *)d_thr32, 2, b.thr_c, -1.0f);
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  };
  if ((st = run_early_exit(c, 64, b, scan, gather)) != LSQR_OK) return st;
  if (c->defer_ovf) {  // lsqr_batch_fit_enqueue: no host synchronisation here, lsqr_batch_fit_wait checks the fill
    c->ovf_cap = seg_cap;
    *done = true;
    return LSQR_OK;
  }
  HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  c->dense_amb_max = *(unsigned int *)c->h_pin;
  dense_worklist_debug(c, d_segcnt, seg_cap);
  if (c->dense_amb_max <= seg_cap) {
    *done = true;
    return LSQR_OK;
  }
  c->ee_last = false;
  (void)fail(c, LSQR_OK, "dense fp16 / fp32 filter: worklist segment overflow (fill %u > %u), fp64 filter used",
             c->dense_amb_max, seg_cap);
  return LSQR_OK;
}

// US calibrations / plane phantom: k_scan_us_f32 over frame chunks and compacted selections
template <class M>
int run_scan_us_ee(lsqr_ctx *c) {
  EeBuf b;
  int st = ee_buffers(c, &b);
  if (st != LSQR_OK) return st;
  const uint32_t H = (uint32_t)c->H;
  const int np = c->opt_ppl == 2 ? 1 : 2;
  const size_t tile = (size_t)kBlock * 2 * np;
  ProfScope whole(c, KID_SCAN);
  struct Mute {
    lsqr_ctx *c;
    bool was;
    ~Mute() { c->prof = was; }
  } mute{c, c->prof};
  c->prof = false;
  bool h16 = false;
  unsigned int *d_segcnt = (unsigned int *)((float *)c->d_partials + 2 * 8192 + 64 * 8192);
  const uint32_t seg_cap = kAmbCap / 1024;
  if constexpr (kHasUsH16<M>) {
    if (c->opt_us_h16 && H <= 8192) {
      if ((st = ensure_us_h16<M>(c, &h16)) != LSQR_OK) return st;
      if (h16) {
        HIPCHK(c, hipMemsetAsync(d_segcnt, 0, 1024 * sizeof(unsigned int), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), c->stream));
      }
    }
  }
  auto scan = [&](size_t rb, size_t re, const uint32_t *range_dev, const uint32_t *h_dev, const uint32_t *sel) -> int {
    if (rb >= re) return LSQR_OK;
    if constexpr (kHasUsH16<M>) {
      if (h16)  // fp16 matrix cores (us_h16.h); the band of every chunk is decided exactly before the next selection
        return launch_us_h16<M>(c, rb, re, sel ? c->d_hparams2 : c->d_hparams, H, d_segcnt, seg_cap, h_dev, sel, range_dev);
    }
    const size_t tiles = (re - rb + tile - 1) / tile;
    const size_t lds = (size_t)H * sizeof(uint32_t);
    int per_cu = (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
    if (per_cu < 1) per_cu = 1;
    const size_t max_blocks = (size_t)256 * per_cu;
    const size_t tpb = (tiles + max_blocks - 1) / max_blocks;
    const int grid = (int)((tiles + tpb - 1) / tpb);
    unsigned ysplit = (unsigned)std::min<size_t>(std::max<size_t>(1, (size_t)256 * 5 / (size_t)grid),
                                                 std::max<size_t>(1, H / 256));
    if (c->opt_hsplit > 0) ysplit = (unsigned)c->opt_hsplit;
    const double *sp = sel ? c->d_hparams2 : c->d_hparams;
    const float *spf = sel ? c->d_hparams2_f32 : c->d_hparams_f32;
    if (np == 1)
      hipLaunchKernelGGL((k_scan_us_f32<M, 1>), dim3(grid, ysplit), dim3(kBlock), lds, c->stream, c->d_data, c->stride,
                         rb, re, sp, spf, H, c->mc, c->d_votes, h_dev, sel, range_dev);
    else
      hipLaunchKernelGGL((k_scan_us_f32<M, 2>), dim3(grid, ysplit), dim3(kBlock), lds, c->stream, c->d_data, c->stride,
                         rb, re, sp, spf, H, c->mc, c->d_votes, h_dev, sel, range_dev);
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  };
  auto gather = [&](const uint32_t *sel, const uint32_t *n_dev) -> int {
    hipLaunchKernelGGL(k_gather_rows, dim3((H + 3) / 4), dim3(256), 0, c->stream, sel, n_dev, H, c->d_hparams,
                       (int)M::SP, c->d_hparams_f32, (int)M::SPF, c->d_hparams2, c->d_hparams2_f32);
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  };
  if ((st = run_early_exit(c, tile, b, scan, gather)) != LSQR_OK) return st;
  if (h16 && c->defer_ovf) {  // lsqr_batch_fit_enqueue: lsqr_batch_fit_wait checks the fill (no synchronisation here)
    c->ovf_cap = seg_cap;
    return LSQR_OK;
  }
  if (h16) {  // a worklist segment that overflowed (not seen): everything again with the packed fp32 filter
    HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c));
    if (*(unsigned int *)c->h_pin > seg_cap) {
      (void)fail(c, LSQR_OK, "US fp16 filter: worklist segment overflow (fill %u > %u), fp32 filter used",
                 *(unsigned int *)c->h_pin, seg_cap);
      const int keep = c->opt_us_h16;
      c->opt_us_h16 = 0;
      c->prof = mute.was;
      st = run_scan_us_ee<M>(c);
      c->opt_us_h16 = keep;
      return st;
    }
  }
  return LSQR_OK;
}

int run_scan(lsqr_ctx *c) {
  c->hyp_since_upload += c->H;
  c->ee_last = false;
  return dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (M::IS_DENSE) {  // default: MFMA filter + exact recheck of ambiguous pairs
      if (c->opt_filter) {
        int st = ensure_absmax(c);
        if (st != LSQR_OK) return st;
        if (!c->d_amb) HIPCHK(c, hipMalloc((void **)&c->d_amb, sizeof(unsigned long long) * kAmbCap));
        if (c->mc.absmax <= 1e100) {  // finite, sane magnitudes: the filter's bound applies
          if constexpr (M::NR == 64) {
            // batch entry points: chunked early exit (earlyexit.h) -- hypotheses that can no longer become the running
            // maximum stop being counted
            if (c->allow_bound && c->opt_bound && c->opt_dense_f32 && c->H >= 128 &&
                c->H <= kEeCap && c->n >= 65536 && c->mc.absmax < 1e15) {
              bool done = false;
              if ((st = run_scan_dense_ee(c, &done)) != LSQR_OK) return st;
              if (done) return LSQR_OK;
            }
          }
          HIPCHK(c, hipMemsetAsync(c->d_votes, 0, c->H * sizeof(uint32_t), c->stream));
          HIPCHK(c, hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), c->stream));
          double *d_thr = c->d_partials;  // scratch: 2 doubles per hypothesis (H <= 2^20 checked)
          if (c->H * 2 > (size_t)kDenseBlocks * 2160) return fail(c, LSQR_ERR_INVALID, "batch too large");
          size_t tiles = (c->n + 63) / 64;
          size_t nblk = std::min<size_t>(tiles, 768);  // 3 workgroups per CU
          size_t rpb = (tiles + nblk - 1) / nblk * 64;
          nblk = (c->n + rpb - 1) / rpb;
          if constexpr (M::NR == 64) {
            // default at n > 32: the filter in fp32 on the matrix cores (twice the fp64 MFMA rate); its band holds
            // ~1e-4 of the pairs, decided exactly from a per-workgroup worklist.  A segment overflow (not seen)
            // falls through to the fp64 filter below.
            if (c->opt_dense_f32 && c->H <= 8192 && c->mc.absmax < 1e15) {
              const uint32_t seg_cap = kAmbCap / 1024;  // <= 512 segments
              {  // two waves per SIMD (the A fragments live in registers): exactly two workgroups per CU
                size_t nb2 = std::min<size_t>(tiles, 512);
                rpb = (tiles + nb2 - 1) / nb2 * 64;
                nblk = (c->n + rpb - 1) / rpb;
              }
              float *d_thr32 = (float *)c->d_partials;                        // 2 floats per hypothesis
              float *d_sp32 = (float *)c->d_partials + 2 * 8192;              // 64 floats per hypothesis (2 MB)
              unsigned int *d_segcnt = (unsigned int *)((float *)c->d_partials + 2 * 8192 + 64 * 8192);  // 1024 words
              HIPCHK(c, hipMemsetAsync(d_segcnt, 0, 1024 * sizeof(unsigned int), c->stream));
              bool h16 = false;
              if (c->opt_dense_f32 == 2 && (st = ensure_dense_h16(c, &h16)) != LSQR_OK) return st;
              if (h16) {  // fp16 matrix cores on two-way splits (dense_h16.h)
                ProfScope ps(c, KID_SCAN);
                hipLaunchKernelGGL(k_dense_prep_h16, dim3((unsigned)((c->H + 3) / 4)), dim3(256), 0, c->stream,
                                   c->d_hparams, (uint32_t)c->H, (int)c->cfg.dim, 64, c->mc.delta, c->mc.absmax_rot,
                                   c->mc.absmax, c->h16_pa, (_Float16 *)d_sp32, c->d_h16_thr);
                HIPCHK(c, hipGetLastError());
                size_t nb16 = 0;
                if ((st = launch_dense_h16(c, 0, c->n, (const _Float16 *)d_sp32, c->d_h16_thr, (uint32_t)c->H, d_segcnt,
                                           seg_cap, nullptr, nullptr, nullptr, &nb16)) != LSQR_OK)
                  return st;
                hipLaunchKernelGGL((k_dense_recheck_seg<64>), dim3((unsigned)nb16), dim3(256), 0, c->stream, c->d_data,
                                   c->stride, c->d_hparams, c->mc, c->d_amb, d_segcnt, seg_cap, c->d_votes,
                                   (unsigned int *)(c->d_counter + 3));
                HIPCHK(c, hipGetLastError());
              } else {
                ProfScope ps(c, KID_SCAN);
                hipLaunchKernelGGL(k_dense_thresholds32, dim3((unsigned)((c->H + 255) / 256)), dim3(256), 0, c->stream,
                                   c->d_hparams, (uint32_t)c->H, (int)c->cfg.dim, 64, c->mc.delta, c->mc.absmax_rot,
                                   c->mc.absmax, d_thr32, d_sp32);
                HIPCHK(c, hipGetLastError());
                {  // hypothesis fragments prefetched through an LDS ring (dense.h)
                  constexpr size_t kRingChunk = 1024;  // 62.7 KiB of LDS per workgroup: two per CU
                  for (size_t h0 = 0; h0 < c->H; h0 += kRingChunk) {
                    uint32_t hc = (uint32_t)std::min<size_t>(kRingChunk, c->H - h0);
                    const uint32_t nhb2 = (((hc + 63) / 64) + 1) & ~1u;
                    size_t lds = sizeof(float) * (8192 + 64 * kDmPitch32 + 64 + 128 * nhb2) + sizeof(uint32_t) * (hc + 1);
                    hipLaunchKernelGGL((k_scan_dense_mfma32r<64>), dim3((unsigned)nblk), dim3(256), lds, c->stream,
                                       c->d_data, c->stride, (size_t)0, c->n, rpb, d_sp32 + h0 * 64, d_thr32 + 2 * h0, hc,
                                       (int)c->cfg.dim, c->d_votes, c->d_amb, d_segcnt, seg_cap, (uint32_t)h0,
                                       (const uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
                    HIPCHK(c, hipGetLastError());
                  }
                }
                hipLaunchKernelGGL((k_dense_recheck_seg<64>), dim3((unsigned)nblk), dim3(256), 0, c->stream, c->d_data,
                                   c->stride, c->d_hparams, c->mc, c->d_amb, d_segcnt, seg_cap, c->d_votes,
                                   (unsigned int *)(c->d_counter + 3));
                HIPCHK(c, hipGetLastError());
              }
              if (c->defer_ovf) {
                c->ovf_cap = seg_cap;
                return LSQR_OK;
              }
              HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                       c->stream));
              HIPCHK(c, sync_stream(c));
              c->dense_amb_max = *(unsigned int *)c->h_pin;
              dense_worklist_debug(c, d_segcnt, seg_cap);
              if (c->dense_amb_max <= seg_cap) return LSQR_OK;
              (void)fail(c, LSQR_OK, "dense fp16 / fp32 filter: worklist segment overflow (fill %u > %u), fp64 filter used",
                         c->dense_amb_max, seg_cap);
              HIPCHK(c, hipMemsetAsync(c->d_votes, 0, c->H * sizeof(uint32_t), c->stream));   // overflow: fp64 filter
              HIPCHK(c, hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), c->stream));
            }
          }
          {
            ProfScope ps(c, KID_SCAN);
            hipLaunchKernelGGL(k_dense_thresholds, dim3((unsigned)((c->H + 255) / 256)), dim3(256), 0,
                               c->stream, c->d_hparams, (uint32_t)c->H, (int)c->cfg.dim, (int)M::NR,
                               c->mc.delta, c->mc.absmax, d_thr);
            HIPCHK(c, hipGetLastError());
            for (size_t h0 = 0; h0 < c->H; h0 += kDmHypChunk) {
              uint32_t hc = (uint32_t)std::min<size_t>(kDmHypChunk, c->H - h0);
              if constexpr (M::NR == 64) {
                {  // n = 64: B fragments in registers, no barriers per block
                  size_t lds2 = sizeof(double) * (64 * kDmPitch + 64) + sizeof(uint32_t) * hc;
                  hipLaunchKernelGGL((k_scan_dense_mfma2<64>), dim3((unsigned)nblk), dim3(256), lds2,
                                     c->stream, c->d_data, c->stride, c->n, rpb,
                                     c->d_hparams + h0 * M::NR, d_thr + 2 * h0, hc, (int)c->cfg.dim,
                                     c->d_votes + h0, c->d_amb, (unsigned int *)(c->d_counter + 3),
                                     (uint32_t)h0);
                  HIPCHK(c, hipGetLastError());
                  continue;
                }
              }
              size_t lds = sizeof(double) * (2 * 64 * kDmPitch + 3 * 64) + sizeof(uint32_t) * hc;
              hipLaunchKernelGGL((k_scan_dense_mfma<M::NR>), dim3((unsigned)nblk), dim3(256), lds,
                                 c->stream, c->d_data, c->stride, c->n, rpb,
                                 c->d_hparams + h0 * M::NR, d_thr + 2 * h0, hc, (int)c->cfg.dim,
                                 c->d_votes + h0, c->d_amb, (unsigned int *)(c->d_counter + 3),
                                 (uint32_t)h0);
              HIPCHK(c, hipGetLastError());
            }
            hipLaunchKernelGGL((k_dense_recheck<M::NR>), dim3(64), dim3(256), 0, c->stream, c->d_data,
                               c->stride, c->d_hparams, c->mc, c->d_amb,
                               (const unsigned int *)(c->d_counter + 3), c->d_votes);
            HIPCHK(c, hipGetLastError());
          }
          // worklist overflow (never seen: ~1e-13 of the pairs are ambiguous) -> exact kernel
          if (c->defer_ovf) {
            c->ovf_cap = kAmbCap;
            return LSQR_OK;
          }
          HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter + 3, sizeof(unsigned long long),
                                   hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, sync_stream(c));
          if (*(unsigned int *)c->h_pin <= kAmbCap) return LSQR_OK;
        }
      }
    }
    if constexpr (requires { M::NF32; }) {  // packed fp32 pre-filter (scan_filter 1); US: 2 = the fused fp64 filter
      if (c->opt_filter == 1 && c->absmax_valid && c->mc.absmax <= 1e15) {
        c->ee_last = false;
        if (c->allow_bound && c->opt_bound && c->H >= 256 && c->H <= kEeCap && c->n >= 65536)
          return run_scan_us_ee<M>(c);  // batch entry points: chunked early exit (earlyexit.h)
        const int np = c->opt_ppl == 2 ? 1 : 2;  // pairs of frames per lane (scan_ppl 2 / 4)
        HIPCHK(c, hipMemsetAsync(c->d_votes, 0, c->H * sizeof(uint32_t), c->stream));
        if constexpr (kHasUsH16<M>) {
          if (c->opt_us_h16 && c->H >= 32 && c->H <= 8192 && c->n >= 4096) {  // fp16 matrix cores (us_h16.h)
            bool h16 = false;
            int st = ensure_us_h16<M>(c, &h16);
            if (st != LSQR_OK) return st;
            if (h16) {
              unsigned int *d_segcnt = (unsigned int *)((float *)c->d_partials + 2 * 8192 + 64 * 8192);
              const uint32_t seg_cap = kAmbCap / 1024;
              HIPCHK(c, hipMemsetAsync(d_segcnt, 0, 1024 * sizeof(unsigned int), c->stream));
              HIPCHK(c, hipMemsetAsync(c->d_counter + 3, 0, sizeof(unsigned long long), c->stream));
              {
                ProfScope ps(c, KID_SCAN);
                if ((st = launch_us_h16<M>(c, 0, c->n, c->d_hparams, (uint32_t)c->H, d_segcnt, seg_cap, nullptr, nullptr,
                                           nullptr)) != LSQR_OK)
                  return st;
              }
              if (c->defer_ovf) {
                c->ovf_cap = seg_cap;
                return LSQR_OK;
              }
              HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                       c->stream));
              HIPCHK(c, sync_stream(c));
              if (*(unsigned int *)c->h_pin <= seg_cap) return LSQR_OK;
              (void)fail(c, LSQR_OK, "US fp16 filter: worklist segment overflow (fill %u > %u), fp32 filter used",
                         *(unsigned int *)c->h_pin, seg_cap);
              HIPCHK(c, hipMemsetAsync(c->d_votes, 0, c->H * sizeof(uint32_t), c->stream));
            }
          }
        }
        size_t tiles = (c->n + (size_t)kBlock * 2 * np - 1) / ((size_t)kBlock * 2 * np);
        for (size_t h0 = 0; h0 < c->H; h0 += kScanChunk) {
          uint32_t hc = (uint32_t)std::min<size_t>(kScanChunk, c->H - h0);
          size_t lds = (size_t)hc * sizeof(uint32_t);
          int per_cu = (int)std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1));
          if (per_cu < 1) per_cu = 1;
          size_t max_blocks = (size_t)256 * per_cu;
          size_t tpb = (tiles + max_blocks - 1) / max_blocks;
          int grid = (int)((tiles + tpb - 1) / tpb);
          // few tiles (1 M frames = 977): split the hypothesis range over blockIdx.y to fill the chip
          unsigned ysplit = (unsigned)std::min<size_t>(std::max<size_t>(1, (size_t)256 * 5 / (size_t)grid),
                                                       std::max<size_t>(1, hc / 256));
          if (c->opt_hsplit > 0) ysplit = (unsigned)c->opt_hsplit;
          ProfScope ps(c, KID_SCAN);
          if (np == 1)
            hipLaunchKernelGGL((k_scan_us_f32<M, 1>), dim3(grid, ysplit), dim3(kBlock), lds, c->stream,
                               c->d_data, c->stride, (size_t)0, c->n, c->d_hparams + h0 * M::SP,
                               c->d_hparams_f32 + h0 * M::SPF, hc, c->mc, c->d_votes + h0, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr, (const uint32_t *)nullptr);
          else
            hipLaunchKernelGGL((k_scan_us_f32<M, 2>), dim3(grid, ysplit), dim3(kBlock), lds, c->stream,
                               c->d_data, c->stride, (size_t)0, c->n, c->d_hparams + h0 * M::SP,
                               c->d_hparams_f32 + h0 * M::SPF, hc, c->mc, c->d_votes + h0, (const uint32_t *)nullptr,
                               (const uint32_t *)nullptr, (const uint32_t *)nullptr);
          HIPCHK(c, hipGetLastError());
        }
        return LSQR_OK;
      }
    } else if constexpr (requires { M::SPF; }) {  // plane, sphere, line: fp32 pre-filter + exact re-evaluation
      // magnitudes the fp32 copies cannot hold (or NaN): the plain fp64 kernel below
      const bool f32_ok = c->absmax_valid && c->mc.absmax <= 1e15;
      if constexpr (requires { typename CellOf<M>::type; }) {
        typedef typename CellOf<M>::type CM;
        // two-level scan over the spatial index; auto: built once an upload has seen enough
        // hypotheses to pay for the build (a few HBM passes)
        const bool tuned_defaults = c->opt_filter == 1 && c->opt_ppl == 0;  // A/B knobs untouched
        // Cost model of the build (auto mode).  Per (hypothesis, observation) the exhaustive filter kernel costs
        // ~1.9e-13 s and the two-level scan ~0.2e-13 s (10 M points x 4096 hypotheses: 7.6 ms against 0.8 ms);
        // the build costs ~1.2e-10 s per observation (radix sort + k-d refinement + gather; r03 without the
        // refinement: 0.7e-10).  It pays for itself once
        //   hypotheses still to come  >  1.2e-10 / 1.7e-13  ~  700,
        // and "still to come" is estimated by the larger of what the caller announced (lsqr_ransac: the current
        // numTries bound) and what this upload has been asked to scan so far, this batch included.
        constexpr uint64_t kIndexPaysAfter = 768;
        const uint64_t to_come = std::max<uint64_t>(c->hyp_expected, c->hyp_since_upload);
        const bool want = c->opt_filter && f32_ok && c->mc.absmax >= 1e-10 && !c->index_failed &&
                          (c->opt_index == 2 ||
                           (c->opt_index == 1 && tuned_defaults &&
                            (c->index_valid || (c->n >= 65536 && to_come >= kIndexPaysAfter))));
        if (want) {
          uint32_t cell_pts = c->opt_cell ? (uint32_t)c->opt_cell : (uint32_t)CM::DEFAULT_CELL;
          if constexpr (!requires { CM::MAX_PP; }) cell_pts = cell_pts > 512 ? 512 : cell_pts;
          bool usable = true;
          // the k-d levels above the runs: 5 - 12 % fewer surviving pairs for 4.5 ms more per build (10 M records) -- worth
          // it once an upload has been scanned by "scan_kd_after" hypotheses (what the caller announced does not count
          // here: the adaptive bound of a RANSAC run starts in the millions and collapses within a few batches)
          int kd_now = c->opt_refine && c->hyp_since_upload >= (uint64_t)c->opt_kd_after ? c->opt_kd_levels : 0;
          if constexpr (std::is_same<CM, PlaneCell<3>>::value) {
            // ... except for the plane's BOUNDED scan: its axis bound (axis.h: k_bound_axis) is tighter on the flat
            // Morton runs than on the more cubical k-d regions (166 against 207 of 4096 hypotheses reach the exact
            // count: 0.30 against 0.32 ms), while the full count gains 8 % from the levels.  So a bounded batch never
            // asks for them (an index that has them keeps them: no rebuilding back and forth), a counting one does.
            if (CM::USE_BOUND && c->allow_bound && c->opt_bound) kd_now = c->index_kd_levels;
          }
          if (!c->index_valid || c->cell_pts != cell_pts || kd_now > c->index_kd_levels) {
            c->kd_build_levels = kd_now;
            // the index is an accelerator: if it cannot be built (typically no memory for the sorted
            // copy) this upload keeps the exhaustive kernels instead of failing the scan
            if (build_index<M::ND>(c, cell_pts) != LSQR_OK) {
              (void)hipGetLastError();
              c->index_failed = true;
              usable = false;
            }
          }
          // one cell per wave tile; several cells per tile and 128-record cells were measured dead ends
          //  and are no longer instantiated
          if (usable) {
            c->last_bound[0] = 0;
            // batch entry points: hypotheses that cannot become the running maximum are not counted (the extra
            // launches only pay for batches of >= 1024; the selection kernels handle <= 8192)
            if (CM::USE_BOUND && c->allow_bound && c->opt_bound && c->H >= 1024 && c->H <= kSelCap && c->n_cells > 0) {
              return with_pp<CM>(cell_pts, [&](auto pp) { return run_scan_bounded<CM, decltype(pp)::value>(c); });
            }
            // plain scans of a large batch: the statically balanced kernel where it measured faster (plane, 10 M x
            // 4096: 1.13 against 1.24 ms; sphere and line are faster with tiles handed out dynamically);
            // "scan_pairs" 1 / 2 force one or the other (A/B)
            bool full_pairs = false;
            if constexpr (requires { CM::FULL_COUNT_PAIRS; }) full_pairs = c->opt_pairs == 0 && c->H >= 1024;
            if (c->opt_pairs == 1 || full_pairs) {
              if constexpr (std::is_same<CM, PlaneCell<3>>::value) {
                // the batch in key order (cells.h: k_plane_order): similar planes share a 64-group
                if (c->opt_hyp_order && c->H >= 1024 && c->H <= kOrderCap) {
                  const uint32_t H = (uint32_t)c->H;
                  uint32_t *perm = c->d_sel + kPilots, *cnt = (uint32_t *)(c->d_counter + 7);
                  double *sp_b = c->d_hparams2 + (size_t)kPilots * M::SP;
                  float *spf_b = c->d_hparams2_f32 + (size_t)kPilots * M::SPF;
                  uint32_t *votes_b = c->d_votes2 + kPilots;
                  hipLaunchKernelGGL((k_plane_order<M::SP>), dim3(1), dim3(1024), 0, c->stream, c->d_hparams, H,
                                     c->mc.absmax, perm, cnt);
                  hipLaunchKernelGGL(k_gather_rows, dim3((H + 3) / 4), dim3(256), 0, c->stream, perm, cnt, H,
                                     c->d_hparams, (int)M::SP, c->d_hparams_f32, (int)M::SPF, sp_b, spf_b);
                  HIPCHK(c, hipGetLastError());
                  const ScanBatch pb = {sp_b, spf_b, (size_t)H, votes_b, nullptr};
                  int st2 = with_pp<CM>(cell_pts,
                                        [&](auto pp) { return run_scan_pairs<CM, decltype(pp)::value>(c, pb); });
                  if (st2 != LSQR_OK) return st2;
                  hipLaunchKernelGGL(k_scatter_perm, dim3((H + 255) / 256), dim3(256), 0, c->stream, perm, H, votes_b,
                                     c->d_votes);
                  HIPCHK(c, hipGetLastError());
                  return LSQR_OK;
                }
              }
              const ScanBatch b = {c->d_hparams, c->d_hparams_f32, c->H, c->d_votes, nullptr};
              return with_pp<CM>(cell_pts, [&](auto pp) { return run_scan_pairs<CM, decltype(pp)::value>(c, b); });
            }
            return with_pp<CM>(cell_pts, [&](auto pp) { return run_scan_cells<CM, decltype(pp)::value, 1>(c); });
          }
        }
      }
      if (c->opt_filter && f32_ok) {
        int ppl = c->opt_ppl ? c->opt_ppl : 4;  // measured best (tools/ab_scan.py)
        // re-check granularity: per packed pair (line: wide band, ambiguous tiles are common) or per
        // tile; scan_filter 2 / 3 force one or the other for A/B runs
        const bool pair = c->opt_filter == 2 || (c->opt_filter == 1 && M::FGRAN == 1);
        if (pair) {
          if (ppl == 8) return run_scan_f32<M, 8, 1>(c);
          return run_scan_f32<M, 4, 1>(c);
        }
        if (ppl == 8) return run_scan_f32<M, 8>(c);
        if (ppl == 16) return run_scan_f32<M, 16>(c);
        return run_scan_f32<M, 4>(c);
      }
    }
    if constexpr (M::REC <= 3) {  // point models: PPL is tunable
      int ppl = c->opt_ppl ? c->opt_ppl : 8;
      if (ppl == 2) return run_scan_ppl<M, 2>(c);
      if (ppl == 8) return run_scan_ppl<M, 8>(c);
      return run_scan_ppl<M, 4>(c);
    } else {
      return run_scan_ppl<M, M::PPL>(c);
    }
  });
}

// run_scan for the entry points that only need the first-max winner (and, for lsqr_ransac's replay, exact votes
// of the hypotheses that become the running maximum): the bounded scan may leave the rest uncounted
int run_scan_batch(lsqr_ctx *c, uint32_t best_before) {
  c->allow_bound = true;
  c->best_before = best_before;
  int st = run_scan(c);
  c->allow_bound = false;
  c->best_before = 0;
  return st;
}

// ---- moments / solves ---------------------------------------------------------------------------
// phase 0: the model's LS moment block about d_vec (origin); phase 1: LM block at d_vec (x trial)
// sum z z^T over rows [begin, end) of a row-major matrix whose rows hold n + 1 entries z
int launch_syrk(lsqr_ctx *c, const double *data, size_t stride, int n, int use_mask, size_t begin,
                size_t end, int *nmom) {
  const int ne = dense_ne(n), ps = dense_pstride(n);
  size_t cnt = end - begin;
  int nb = grid_for(cnt, kSyrkTile * 8, kDenseBlocks);
  size_t chunk = (cnt + nb - 1) / nb;
  chunk = (chunk + kSyrkTile - 1) / kSyrkTile * kSyrkTile;
  nb = (int)((cnt + chunk - 1) / chunk);
  if (nb < 1) nb = 1;
  *nmom = ne + 1;
  {
    ProfScope ps_(c, KID_MOMENTS);
    if (c->opt_filter) {  // default: matrix-core SYRK
      if (c->opt_syrk_diag == 1)  // diagnostics (tools/syrk_ab.py): loads only / MFMAs only
        hipLaunchKernelGGL(k_syrk_mfma<1>, dim3(nb), dim3(256), 0, c->stream, data, stride,
                           begin, end, chunk, n, c->d_mask, use_mask, ps, c->d_partials);
      else if (c->opt_syrk_diag == 2)
        hipLaunchKernelGGL(k_syrk_mfma<2>, dim3(nb), dim3(256), 0, c->stream, data, stride,
                           begin, end, chunk, n, c->d_mask, use_mask, ps, c->d_partials);
      else if (n + 1 <= 32)  // the plane phantom's 31 + 1 columns: two of the five column blocks
        hipLaunchKernelGGL((k_syrk_mfma<0, 2, 10>), dim3(nb), dim3(256), 0, c->stream, data, stride,
                           begin, end, chunk, n, c->d_mask, use_mask, ps, c->d_partials);
      else
        hipLaunchKernelGGL(k_syrk_mfma<0>, dim3(nb), dim3(256), 0, c->stream, data, stride,
                           begin, end, chunk, n, c->d_mask, use_mask, ps, c->d_partials);
    } else {
      size_t lds = sizeof(double) * kSyrkTile * ((n + 1) | 1) + kSyrkTile;
      hipLaunchKernelGGL(k_syrk_dense, dim3(nb), dim3(256), lds, c->stream, data, stride,
                         begin, end, chunk, n, c->d_mask, use_mask, ps, c->d_partials);
    }
    HIPCHK(c, hipGetLastError());
  }
  {
    ProfScope ps_(c, KID_SOLVE);
    hipLaunchKernelGGL(k_reduce, dim3(*nmom), dim3(64), 0, c->stream, c->d_partials, nb, ps, *nmom,
                       c->d_mom);
    HIPCHK(c, hipGetLastError());
  }
  return LSQR_OK;
}

int launch_moments_dense(lsqr_ctx *c, int use_mask, size_t begin, size_t end, int *nmom) {
  return launch_syrk(c, c->d_data, c->stride, (int)c->cfg.dim, use_mask, begin, end, nmom);
}

void phantom_to_out(const PhantomFit &f, SolveOut *out) {
  memset(out, 0, sizeof *out);
  out->ok = f.ok;
  out->n_params = f.ok ? PhantomModel::P : 0;
  out->lm_info = f.lm_info;
  out->lm_nfev = f.lm_nfev;
  out->cost = f.cost;
  for (int j = 0; j < PhantomModel::P; j++) out->params[j] = f.params[j];
}

// plane phantom: the Gram matrix of the data rows (upper triangle, 496 sums) + the row count.  The rows
// are materialised once per upload (256 B per frame) and summed on the matrix cores by the dense SYRK.
int launch_moments_phantom(lsqr_ctx *c, int use_mask, size_t begin, size_t end, int *nmom) {
  if (!c->rows_valid) {
    int st = ensure(c, &c->d_rows, &c->rows_cap, std::max<size_t>(c->n, 1) * 32);
    if (st != LSQR_OK) return st;
    ProfScope ps(c, KID_MOMENTS);
    hipLaunchKernelGGL(k_phantom_rows, dim3((unsigned)((c->n * 8 + 255) / 256)), dim3(256), 0, c->stream,
                       c->d_data, c->stride, c->n, c->d_rows);
    HIPCHK(c, hipGetLastError());
    c->rows_valid = true;
  }
  return launch_syrk(c, c->d_rows, 32, 30, use_mask, begin, end, nmom);
}

// plane phantom: both fits from the Gram block on the host (phantom.h)
void phantom_solve_block(const lsqr_model_cfg &cfg, const double *blk, SolveOut *out) {
  PhantomFit f;
  phantom_fit_block(blk, cfg.ls_type == LSQR_LS_ITERATIVE, &f);
  phantom_to_out(f, out);
}

// rows: the records the block in d_mom was summed over are at hand -- rows [begin, end) of the upload, those of the
// mask when use_mask -- so a system the elimination refuses is solved again from them in double-double (dense.h:
// k_gram_dd_dense / k_dense_dd_solve: the reference's SVD pseudo-inverse of A with its absolute rank test).  Without
// rows (lsqr_solve_moments, the multi-GPU sum of blocks) the pseudo-inverse of the Gram block decides.
int launch_solve_dense(lsqr_ctx *c, bool rows = false, int use_mask = 0, size_t begin = 0, size_t end = 0) {
  ProfScope ps(c, KID_SOLVE);
  const int n = (int)c->cfg.dim;
  int *flag = (int *)(c->d_counter + 7) + 1;  // (the low word of the slot is k_plane_order's count: another model's)
  rows = rows && c->opt_dense_dd && end > begin;
  if (rows) HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_solve_dense, dim3(1), dim3(256), dense_lds_bytes(n), c->stream, c->d_mom, n,
                     c->opt_dense_fast, c->d_out, rows ? flag : (int *)nullptr);
  HIPCHK(c, hipGetLastError());
  if (!rows) return LSQR_OK;
  if (!c->d_ddpart) HIPCHK(c, hipMalloc((void **)&c->d_ddpart, sizeof(double) * 2 * kDdNe * kDdBlocks));
  const int nb = (int)std::min<size_t>(kDdBlocks, (end - begin + 31) / 32);
  hipLaunchKernelGGL((k_gram_dd_dense<32>), dim3(nb), dim3(256), 0, c->stream, c->d_data, c->stride, begin, end, n,
                     use_mask ? c->d_mask : (const uint8_t *)nullptr, flag, c->d_ddpart);
  HIPCHK(c, hipGetLastError());
  (void)hipFuncSetAttribute((const void *)k_dense_dd_solve, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)dense_dd_lds(64));
  hipLaunchKernelGGL(k_dense_dd_solve, dim3(1), dim3(256), dense_dd_lds(n), c->stream, c->d_ddpart, nb, n, c->d_mom,
                     flag, c->d_out);
  HIPCHK(c, hipGetLastError());
  return LSQR_OK;
}

template <class M>
int launch_moments(lsqr_ctx *c, int use_mask, size_t begin, size_t end, int phase, int *nmom) {
  if constexpr (requires { M::IS_PHANTOM; }) {
    (void)phase;  // the LM block at any x is a function of the Gram block
    return launch_moments_phantom(c, use_mask, begin, end, nmom);
  } else if constexpr (M::IS_DENSE) {
    if (phase != 0) return fail(c, LSQR_ERR_INVALID, "dense model has no iterative phase");
    return launch_moments_dense(c, use_mask, begin, end, nmom);
  } else {
  size_t cnt = end - begin;
  int nb = grid_for(cnt, mom_chunk(c), kMaxPartials);
  size_t chunk = (cnt + nb - 1) / nb;
  chunk = (chunk + kBlock - 1) / kBlock * kBlock;
  nb = (int)((cnt + chunk - 1) / chunk);
  if (nb < 1) nb = 1;
  {
    ProfScope ps(c, KID_MOMENTS);
    if (phase == 0) {
      *nmom = M::NMOM;
      if (use_mask)
        hipLaunchKernelGGL((k_moments<M, AccLs<M>, true>), dim3(nb), dim3(kBlock), 0, c->stream,
                           c->d_data, c->stride, begin, end, chunk, c->d_mask, c->d_vec, c->mc,
                           c->d_partials);
      else
        hipLaunchKernelGGL((k_moments<M, AccLs<M>, false>), dim3(nb), dim3(kBlock), 0, c->stream,
                           c->d_data, c->stride, begin, end, chunk, c->d_mask, c->d_vec, c->mc,
                           c->d_partials);
    } else {
      if constexpr (requires { M::NMOM_LM; }) {
        *nmom = M::NMOM_LM;
        if (use_mask)
          hipLaunchKernelGGL((k_moments<M, AccLm<M>, true>), dim3(nb), dim3(kBlock), 0, c->stream,
                             c->d_data, c->stride, begin, end, chunk, c->d_mask, c->d_vec, c->mc,
                             c->d_partials);
        else
          hipLaunchKernelGGL((k_moments<M, AccLm<M>, false>), dim3(nb), dim3(kBlock), 0,
                             c->stream, c->d_data, c->stride, begin, end, chunk, c->d_mask,
                             c->d_vec, c->mc, c->d_partials);
      } else {
        return fail(c, LSQR_ERR_INVALID, "model has no iterative phase");
      }
    }
    HIPCHK(c, hipGetLastError());
  }
  {
    ProfScope ps(c, KID_SOLVE);
    hipLaunchKernelGGL(k_reduce, dim3(*nmom), dim3(64), 0, c->stream, c->d_partials, nb,
                       (int)MOM_MAX, *nmom, c->d_mom);
    HIPCHK(c, hipGetLastError());
  }
  return LSQR_OK;
  }
}

int read_out(lsqr_ctx *c, SolveOut *o) {
  HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_out, sizeof(SolveOut), hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, sync_stream(c));
  memcpy(o, c->h_pin, sizeof(SolveOut));
  return LSQR_OK;
}

bool wants_lm(const lsqr_model_cfg &cfg) {
  return (cfg.model == LSQR_MODEL_SPHERE && cfg.ls_type == LSQR_LS_GEOMETRIC) ||
         ((cfg.model == LSQR_MODEL_US_SINGLE || cfg.model == LSQR_MODEL_US_POINTER) &&
          cfg.ls_type == LSQR_LS_ITERATIVE) ||
         (cfg.model == LSQR_MODEL_PHANTOM && cfg.ls_type == LSQR_LS_ITERATIVE);
}

void lm_settings(const lsqr_model_cfg &cfg, int *n, double *ftol, double *xtol, double *gtol,
                 int *maxfev) {
  // SphereParametersEstimator.hxx:323-329: x and g tolerances 10e-16, 500 evaluations; ftol is
  // vnl_nonlinear_minimizer's default xtol*0.01 = 1e-10.
  *n = cfg.dim + 1;
  *ftol = 1e-10;
  *xtol = 10e-16;
  *gtol = 10e-16;
  *maxfev = 500;
  if (cfg.model == LSQR_MODEL_US_SINGLE) {  // SinglePointTarget...Estimator.cxx:287-295
    *n = 11;
    *ftol = *xtol = *gtol = 10e-16;
    *maxfev = 5000;
  } else if (cfg.model == LSQR_MODEL_PHANTOM) {  // PlanePhantom...Estimator.cxx:368-376
    *n = 11;
    *ftol = *xtol = *gtol = 10e-16;
    *maxfev = 5000;
  } else if (cfg.model == LSQR_MODEL_US_POINTER) {  // :931-939
    *n = 8;
    *ftol = *xtol = *gtol = 10e-8;
    *maxfev = 5000;
  }
}

// the closed-form part of run_fit without the read-back: moments + solve chained on the stream, result in
// d_out.  Models whose fit needs the host in the loop (LM, the phantom's Gram solve) are refused.
int enqueue_fit(lsqr_ctx *c, int use_mask, bool have_moments = false) {
  if (wants_lm(c->cfg) || c->cfg.model == LSQR_MODEL_PHANTOM)
    return fail(c, LSQR_ERR_INVALID, "this fit needs the host between device passes");
  return dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    int nmom = 0, st;
    if constexpr (M::IS_DENSE) {
      if (!have_moments && (st = launch_moments_dense(c, use_mask, 0, c->n, &nmom)) != LSQR_OK) return st;
      return launch_solve_dense(c, true, use_mask, 0, c->n);
    } else {
      if (!have_moments) {
        bool first_datum = !c->origin_valid;
        if constexpr (requires { M::ORIGIN_FIRST; }) first_datum = true;
        if (first_datum) {
          HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_data, sizeof(double) * M::ND, hipMemcpyDeviceToDevice,
                                   c->stream));
        } else {
          HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_par + (c->cfg.model == LSQR_MODEL_SPHERE ? 0 : M::ND),
                                   sizeof(double) * M::ND, hipMemcpyDeviceToDevice, c->stream));
        }
        if ((st = launch_moments<M>(c, use_mask, 0, c->n, 0, &nmom)) != LSQR_OK) return st;
      }
      ProfScope ps(c, KID_SOLVE);
      hipLaunchKernelGGL((k_solve<M>), dim3(1), dim3(64), 0, c->stream, c->d_mom, c->d_vec, c->mc,
                         c->d_out);
      HIPCHK(c, hipGetLastError());
      return LSQR_OK;
    }
  });
}

// ---- persistent LM fit (lm_persist.h) -------------------------------------------------------------------------------
// Compute-unit tokens: the persistent kernels of this process never ask for more workgroups than the device holds
// (one per compute unit: the kernel's LDS request), so none of them can wait for a workgroup that another one keeps
// from becoming resident.  A fit draws G tokens before it launches and returns them when its kernel has ended.
struct LmpPool {
  std::mutex mu;
  std::condition_variable cv;
  int cus[16] = {0}, free_[16] = {0}, live[16] = {0};  // per device: compute units, free tokens, root contexts alive
  // contexts that started a persistent fit lately: how many ways the device is shared (a process with four contexts
  // of which one is fitting gives that one the whole device)
  std::vector<std::pair<const void *, double>> recent[16];
};
LmpPool &lmp_pool() {
  static LmpPool p;
  return p;
}
void lmp_ctx_count(int device, int d, const void *who = nullptr) {
  if (device < 0 || device >= 16) return;
  LmpPool &p = lmp_pool();
  std::lock_guard<std::mutex> lk(p.mu);
  p.live[device] += d;
  if (who) {  // a context that is gone is not fitting any more
    auto &rec = p.recent[device];
    for (auto it = rec.begin(); it != rec.end();) it = it->first == who ? rec.erase(it) : it + 1;
  }
}
int lmp_acquire(int device, const void *who, int forced, int need_max, bool only_if_alone) {
  LmpPool &p = lmp_pool();
  std::unique_lock<std::mutex> lk(p.mu);
  if (device < 0 || device >= 16) return 0;
  if (!p.cus[device]) {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cu <= 0) return 0;
    p.cus[device] = p.free_[device] = cu;
  }
  int G = forced;
  {  // the device shared evenly by the contexts that have been fitting in the last half second (up to eight)
    const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    auto &rec = p.recent[device];
    bool mine = false;
    int active = 0;
    for (auto it = rec.begin(); it != rec.end();) {
      if (it->first == who) {
        it->second = now;
        mine = true;
      }
      if (now - it->second > 0.5) {
        it = rec.erase(it);
        continue;
      }
      active++;
      ++it;
    }
    if (!mine) {
      rec.push_back({who, now});
      active++;
    }
    // "lm_persist" 1 (the default): several fits in flight are served better by the launch path -- their passes overlap
    // each other's host round trips and the aggregate is HBM-bound either way (r05: 70 k hypotheses/s on eight streams
    // of launches against 36 - 60 k on four to eight persistent fits) --, a fit alone on the device by the persistent
    // kernel (31 k against 26 - 27 k)
    if (only_if_alone && active > 1) return 0;
    if (G <= 0) {
      const int share = std::max(1, std::min(8, active));
      G = p.cus[device] / share;
      int pw = 1;
      while (2 * pw <= G) pw *= 2;
      G = pw;
    }
  }
  G = std::max(1, std::min({G, p.cus[device], need_max}));
  p.cv.wait(lk, [&] { return p.free_[device] >= G; });
  p.free_[device] -= G;
  return G;
}
void lmp_release(int device, int G) {
  LmpPool &p = lmp_pool();
  {
    std::lock_guard<std::mutex> lk(p.mu);
    p.free_[device] += G;
  }
  p.cv.notify_all();
}

// *done = true: finished, `s` holds the final state; false: not run or given up (the caller takes the launch path)
template <class M>
int lm_persist_fit(lsqr_ctx *c, const double *tiles, size_t cnt, int nb, LmState &s, const double *x0, int n,
                   double ftol, double xtol, double gtol, int maxfev, bool *done) {
  *done = false;
  static const bool shared_gpu = getenv("LSQR_SHARE_GPU") != nullptr;  // several processes on one device: no tokens across them
  if (!c->opt_lm_persist || shared_gpu || c->is_lane) return LSQR_OK;
  typedef typename M::LmCoef Coef;
  constexpr int NCW = (int)(sizeof(Coef) / 8), NMOM = (int)M::NMOM_LM;
  if (!c->d_lmp) {
    if (hipMalloc((void **)&c->d_lmp, sizeof(LmpCtl)) != hipSuccess) return LSQR_OK;
    if (hipHostMalloc((void **)&c->h_lmcmd, sizeof(unsigned long long) * 256, hipHostMallocCoherent) != hipSuccess) {
      (void)hipFree(c->d_lmp);
      c->d_lmp = nullptr;
      return LSQR_OK;
    }
    memset(c->h_lmcmd, 0, sizeof(unsigned long long) * 256);
  }
  if (c->lm_seq > 0xF0000000u) {  // tags would wrap: start again from clean granules (nothing is in flight here)
    HIPCHK(c, sync_stream(c));
    memset(c->h_lmcmd, 0, sizeof(unsigned long long) * 256);
    memset(c->h_lmres, 0, sizeof(unsigned long long) * 256);
    c->lm_seq = 0;
  }
  const int host_step = c->opt_lm_persist != 2;
  // one round of virtual blocks when the device is ours alone (512-thread workgroups: two blocks at a time each)
  const int G = lmp_acquire(c->device, c, c->opt_lm_persist_wgs, (nb + 1) / 2, c->opt_lm_persist == 1);  // tokens held
  if (G <= 0) return LSQR_OK;
  // workgroups of eight waves (two virtual blocks at a time) when the tokens cover all blocks in one round -- the pass
  // then spreads over twice the compute units --, else sixteen waves (four blocks at a time)
  const int threads = G * 2 >= nb ? 512 : 1024;
  const int Gl = threads == 512 ? G : std::min(G, (nb + 3) / 4);  // workgroups launched
  const size_t lds = std::max<size_t>((size_t)(threads / 64) * 64 * 17 * sizeof(double), (size_t)96 << 10);
  // a fit that has the device to itself in one round keeps its tiles in registers (k_lm_persist<M, 4>: 512 threads)
  const bool resident = host_step && threads == 512 && (size_t)Gl * 2 >= (size_t)nb && c->opt_lm_persist_resident;
  if (hipFuncSetAttribute((const void *)k_lm_persist<M, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10) !=
          hipSuccess ||
      hipFuncSetAttribute((const void *)k_lm_persist<M, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 << 10) !=
          hipSuccess) {
    lmp_release(c->device, G);
    return LSQR_OK;
  }
  LmpInit init{};
  init.n = n;
  init.maxfev = maxfev;
  init.ftol = ftol;
  init.xtol = xtol;
  init.gtol = gtol;
  init.factor = 100.0;
  for (int j = 0; j < LM_NMAX; j++) init.x0[j] = j < n ? x0[j] : 0.0;
  const uint32_t seq0 = c->lm_seq;
  c->lm_seq += (uint32_t)maxfev + 8;
  const unsigned long long timeout = (unsigned long long)std::max(1, c->opt_lm_persist_timeout_ms) * 100000ULL;  // 100 MHz
  auto give_up = [&](int rc) {
    lmp_release(c->device, G);
    return rc;
  };
  if (hipMemsetAsync(c->d_lmp, 0, sizeof(LmpCtl), c->stream) != hipSuccess) return give_up(LSQR_OK);
  {
    ProfScope ps(c, KID_MOMENTS);
    if (resident)
      hipLaunchKernelGGL((k_lm_persist<M, 4>), dim3(Gl), dim3(threads), lds, c->stream, tiles, cnt, nb, c->d_lmp,
                         c->d_partials, c->h_lmres, c->h_lmcmd, c->d_lm, c->d_out, init, seq0, host_step, timeout,
                         (uint32_t)c->opt_lm_persist_test_abort);
    else
      hipLaunchKernelGGL((k_lm_persist<M, 0>), dim3(Gl), dim3(threads), lds, c->stream, tiles, cnt, nb, c->d_lmp,
                         c->d_partials, c->h_lmres, c->h_lmcmd, c->d_lm, c->d_out, init, seq0, host_step, timeout,
                         (uint32_t)c->opt_lm_persist_test_abort);
  }
  if (hipGetLastError() != hipSuccess) return give_up(LSQR_OK);
  bool gave_up = false;
  uint64_t host_wait_ns = 0, host_step_ns = 0;
  if (host_step) {
    lm_init(s, n, x0, ftol, xtol, gtol, maxfev, 100.0);
    volatile unsigned long long *res = c->h_lmres;
    volatile unsigned long long *cmd = c->h_lmcmd;
    for (uint32_t e = 1; !gave_up; e++) {
      const uint32_t tag = seq0 + e;
      unsigned long long spins = 0;
      double blk[LM_MOM_MAX];
      const auto tw0 = std::chrono::steady_clock::now();
      for (int j = 0; j < NMOM && !gave_up; j++) {
        unsigned long long g0, g1;
        while ((uint32_t)(g0 = res[2 * j]) != tag || (uint32_t)(g1 = res[2 * j + 1]) != tag) {
          if ((++spins & 0xFFFF) == 0) {  // every 65 k polls: is the kernel still there?
            hipError_t q = hipStreamQuery(c->stream);
            if (q != hipSuccess && q != hipErrorNotReady) {
              lmp_release(c->device, G);
              return fail(c, LSQR_ERR_HIP, "persistent LM kernel failed: %s", hipGetErrorString(q));
            }
            if (q == hipSuccess && ((uint32_t)res[2 * j] != tag || (uint32_t)res[2 * j + 1] != tag)) {
              gave_up = true;  // the kernel left (a bounded wait expired) without this evaluation
              break;
            }
          }
        }
        if (gave_up) break;
        const unsigned long long bits = (g0 & 0xFFFFFFFF00000000ULL) | (g1 >> 32);
        memcpy(&blk[j], &bits, 8);
      }
      if (gave_up) break;
      const auto tw1 = std::chrono::steady_clock::now();
      const bool cont = lm_advance(s, blk);
      uint32_t words[2 * LMP_MAXCOEF] = {0};
      if (cont) {
        Coef k;
        M::lm_coef(s.xtrial, k);
        memcpy(words, &k, sizeof(Coef));
      }
      for (int i = 0; i < 2 * NCW; i++) cmd[i] = ((unsigned long long)words[i ^ 1] << 32) | tag;
      std::atomic_thread_fence(std::memory_order_release);  // the command granule is what the device polls: written last
      cmd[2 * NCW] = ((unsigned long long)(cont ? LMP_EVAL : LMP_FIN) << 32) | tag;
      host_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(tw1 - tw0).count();
      host_step_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tw1).count();
      if (!cont) break;
    }
  }
  hipError_t se = sync_stream(c);
  lmp_release(c->device, G);
  if (se != hipSuccess) return fail(c, LSQR_ERR_HIP, "persistent LM kernel: %s", hipGetErrorString(se));
  LmpCtl *hc = (LmpCtl *)((char *)c->h_pin + 32768);  // (8192: the mask count finish_ransac reads after the fit; 16384, 49152+: other stagings)
  HIPCHK(c, hipMemcpy(hc, c->d_lmp, sizeof(LmpCtl), hipMemcpyDeviceToHost));
  c->lmp_last[0] = (uint64_t)c->opt_lm_persist;
  c->lmp_last[1] = (uint64_t)Gl;
  c->lmp_last[2] = hc->evals;
  c->lmp_last[3] = hc->status;
  c->lmp_last[4] = (hc->t_end - hc->t_begin) / 100;  // us
  c->lmp_last[6] = host_wait_ns;
  c->lmp_last[7] = host_step_ns;
  c->lmp_trace_n = std::min<uint32_t>(hc->evals, LMP_TRACE);
  memcpy(c->lmp_trace, hc->trace, sizeof(hc->trace));
  if (gave_up || hc->status != LMP_FIN) {
    c->lmp_last[5]++;
    return LSQR_OK;
  }
  if (!host_step) HIPCHK(c, hipMemcpy(&s, c->d_lm, sizeof(LmState), hipMemcpyDeviceToHost));
  *done = true;
  return LSQR_OK;
}

// leastSquaresEstimate over [0,n) (single device).  Leaves the result in d_out.
// have_moments: d_mom already holds the phase-0 block about d_vec (launch_mask_moments)
int run_fit(lsqr_ctx *c, int use_mask, SolveOut *out, bool have_moments = false) {
  return dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    int nmom = 0, st;
    if constexpr (requires { M::IS_PHANTOM; }) {
      if ((st = launch_moments_phantom(c, use_mask, 0, c->n, &nmom)) != LSQR_OK) return st;
      HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_mom, sizeof(double) * nmom, hipMemcpyDeviceToHost,
                               c->stream));
      HIPCHK(c, sync_stream(c));
      phantom_solve_block(c->cfg, (const double *)c->h_pin, out);
      return LSQR_OK;
    } else if constexpr (M::IS_DENSE) {
      if (!have_moments && (st = launch_moments_dense(c, use_mask, 0, c->n, &nmom)) != LSQR_OK) return st;
      if ((st = launch_solve_dense(c, true, use_mask, 0, c->n)) != LSQR_OK) return st;
      return read_out(c, out);
    } else {
    if (!have_moments) {
    bool first_datum = !c->origin_valid;  // default origin: the first observation
    if constexpr (requires { M::ORIGIN_FIRST; }) first_datum = true;  // parameters hold no point
    if (first_datum) {
      HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_data, sizeof(double) * M::ND,
                               hipMemcpyDeviceToDevice, c->stream));
    } else {
      HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_par + (c->cfg.model == LSQR_MODEL_SPHERE ? 0 : M::ND),
                               sizeof(double) * M::ND, hipMemcpyDeviceToDevice, c->stream));
    }
    if ((st = launch_moments<M>(c, use_mask, 0, c->n, 0, &nmom)) != LSQR_OK) return st;
    }
    {
      ProfScope ps(c, KID_SOLVE);
      hipLaunchKernelGGL((k_solve<M>), dim3(1), dim3(64), 0, c->stream, c->d_mom, c->d_vec, c->mc,
                         c->d_out);
      HIPCHK(c, hipGetLastError());
    }
    if (!wants_lm(c->cfg)) return read_out(c, out);
    if ((st = read_out(c, out)) != LSQR_OK) return st;
    if (!out->ok) return LSQR_OK;  // algebraic initialiser failed -> empty (Sphere...hxx:231-232)
    int n;
    double ftol, xtol, gtol;
    int maxfev;
    lm_settings(c->cfg, &n, &ftol, &xtol, &gtol, &maxfev);
    if constexpr (requires { M::NMOM_LM; }) {
      if (c->opt_lm_host) {
        // MINPACK's control flow between device passes runs on the host (like the RANSAC replay):
        // a few hundred flops per evaluation; every N-scale operation stays a device pass.
        LmState &s = c->h_lm;
        lm_init(s, n, out->params, ftol, xtol, gtol, maxfev, 100.0);
        double *pin = (double *)c->h_pin;
        if (c->opt_lm_fused) {
          // one launch per evaluation: trial point by value, block sums + final sum in the same kernel, the
          // result lands in pinned host memory and the host polls its sequence flag (no stream synchronisation,
          // no staging copies): per evaluation = the pass + one launch latency + a few hundred host flops
          // the consensus set, tight and in order (every evaluation then streams n_in records, all lanes busy)
          // Compacting costs about as much as six evaluations through the mask save (10 M records, 38 % inliers:
          // count + scan + host read-back + write = 130 us; a pass over the compacted set is 23 us against 43 us), and
          // the sphere's geometric fit from the algebraic start typically needs three: the per-lane pass starts
          // THROUGH THE MASK and the set is compacted once `kCompactAfter` evaluations have been spent.  The
          // matrix-core pass (US: thousands of evaluations) reads compacted records only and compacts at once.
          constexpr int kCompactAfter = 8;
          const bool mfma_pass = c->opt_lm_mfma && M::NLM >= 8;
          const bool tile_layout = mfma_pass && c->opt_lm_tiles;   // compacted set in tiles of 64 records, field-major
          bool tiles = false;                                      // ... and it has been written that way
          const double *lm_data = c->d_data;
          size_t lm_stride = c->stride, cnt = c->n;
          bool through_mask = use_mask;
          auto compact = [&]() -> int {
            int cb = grid_for(c->n, kBlock * 8, 1024);
            size_t cchunk = (c->n + cb - 1) / cb;
            cchunk = (cchunk + kBlock - 1) / kBlock * kBlock;
            cb = (int)((c->n + cchunk - 1) / cchunk);
            uint32_t *d_cnt = (uint32_t *)c->d_partials, *d_off = d_cnt + 1024;  // scratch (4.4 MB buffer)
            hipLaunchKernelGGL(k_compact_count, dim3(cb), dim3(kBlock), 0, c->stream, c->d_mask, c->n, cchunk, d_cnt);
            hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, c->stream, d_cnt, cb, d_off);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(pin + 200, d_off + cb, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, sync_stream(c));
            cnt = *(const uint32_t *)(pin + 200);
            int st2 = ensure(c, &c->d_lmrec, &c->lmrec_cap, ((std::max<size_t>(cnt, 1) + 63) & ~(size_t)63) * M::ND);
            if (st2 != LSQR_OK) return st2;
            if (tile_layout)  // the matrix-core pass reads tiles of 64 records stored field-major (kernels.h)
              hipLaunchKernelGGL((k_compact_write_tiles<M::ND>), dim3(cb), dim3(kBlock), 0, c->stream, c->d_data,
                                 c->stride, c->d_mask, c->n, cchunk, d_off, c->d_lmrec);
            else
              hipLaunchKernelGGL((k_compact_write<M::ND>), dim3(cb), dim3(kBlock), 0, c->stream, c->d_data, c->stride,
                                 c->d_mask, c->n, cchunk, d_off, c->d_lmrec);
            HIPCHK(c, hipGetLastError());
            lm_data = c->d_lmrec;
            lm_stride = M::ND;
            through_mask = false;
            tiles = tile_layout;
            return LSQR_OK;
          };
          int nb = 1;
          size_t chunk = 0;
          auto shape = [&]() {
            if (tiles) {   // four tiles per wave and more: one in flight behind the one being evaluated
              nb = (int)std::min<size_t>(512, std::max<size_t>(1, (cnt + 1023) / 1024));
              chunk = 0;
              return;
            }
            nb = grid_for(cnt, kBlock * (mfma_pass ? 2 : 8), kMaxPartials);
            chunk = (cnt + nb - 1) / nb;
            chunk = (chunk + kBlock - 1) / kBlock * kBlock;
            nb = (int)((cnt + chunk - 1) / chunk);
            if (nb < 1) nb = 1;
          };
          if (use_mask && mfma_pass && (st = compact()) != LSQR_OK) return st;
          shape();
          bool persist_done = false;
          if constexpr (requires { typename M::LmCoef; }) {
            if (mfma_pass && tiles) {  // the whole fit in one launch (lm_persist.h); falls through when it gives up
              if ((st = lm_persist_fit<M>(c, lm_data, cnt, nb, s, out->params, n, ftol, xtol, gtol, maxfev,
                                          &persist_done)) != LSQR_OK)
                return st;
              if (!persist_done) lm_init(s, n, out->params, ftol, xtol, gtol, maxfev, 100.0);
            }
          }
          volatile unsigned long long *res = c->h_lmres;
          while (!persist_done) {
            if (through_mask && s.nfev >= kCompactAfter) {
              if ((st = compact()) != LSQR_OK) return st;
              shape();
            }
            LmX xk;
            for (int j = 0; j < LM_NMAX; j++) xk.x[j] = j < n ? s.xtrial[j] : 0.0;
            uint32_t seq = ++c->lm_seq;
            if (seq == 0) seq = c->lm_seq = 1;  // (the zero-initialised granules carry tag 0)
            // profiling: every 16th evaluation carries event pairs (four event records per evaluation would cost
            // more host time than the evaluation's own launches); lsqr_profile_get's averages are unaffected
            const bool timed = c->prof && (s.nfev & 15) == 0;
            const bool prof_saved = c->prof;
            c->prof = timed;
            {
              ProfScope ps(c, KID_MOMENTS);
              if constexpr (requires { typename M::LmCoef; }) {
                // the matrix-core pass pays when the (J | f) rows are wide (US: 12 / 9 columns, 78 / 45 sums); for the
                // sphere's 5 columns the 16 x 16 tile is mostly padding and the instruction time alone (25 us at 3.8 M
                // points) exceeds the per-lane version's whole pass
                if (mfma_pass && tiles) {
                  typename M::LmCoef coef;
                  M::lm_coef(xk.x, coef);
                  hipLaunchKernelGGL((k_lm_pass_mfma_t<M>), dim3(nb), dim3(kBlock), 0, c->stream, lm_data, cnt, coef,
                                     c->d_partials);
                } else if (mfma_pass) {
                  typename M::LmCoef coef;
                  M::lm_coef(xk.x, coef);
                  hipLaunchKernelGGL((k_lm_pass_mfma<M>), dim3(nb), dim3(kBlock), 0, c->stream, lm_data, lm_stride,
                                     cnt, coef, c->mc, c->d_partials);
                } else if (through_mask)
                  hipLaunchKernelGGL((k_lm_pass<M, true>), dim3(nb), dim3(kBlock), 0, c->stream, lm_data, lm_stride,
                                     (size_t)0, cnt, chunk, (const uint8_t *)c->d_mask, xk, c->mc, c->d_partials);
                else
                  hipLaunchKernelGGL((k_lm_pass<M, false>), dim3(nb), dim3(kBlock), 0, c->stream, lm_data, lm_stride,
                                     (size_t)0, cnt, chunk, (const uint8_t *)nullptr, xk, c->mc, c->d_partials);
              } else if (through_mask) {
                hipLaunchKernelGGL((k_lm_pass<M, true>), dim3(nb), dim3(kBlock), 0, c->stream, lm_data, lm_stride,
                                   (size_t)0, cnt, chunk, (const uint8_t *)c->d_mask, xk, c->mc, c->d_partials);
              } else {
                hipLaunchKernelGGL((k_lm_pass<M, false>), dim3(nb), dim3(kBlock), 0, c->stream, lm_data, lm_stride,
                                   (size_t)0, cnt, chunk, (const uint8_t *)nullptr, xk, c->mc, c->d_partials);
              }
              HIPCHK(c, hipGetLastError());
            }
            {
              ProfScope ps(c, KID_SOLVE);
              hipLaunchKernelGGL(k_lm_publish, dim3((unsigned)M::NMOM_LM), dim3(64), 0, c->stream, c->d_partials, nb,
                                 (int)M::NMOM_LM, c->h_lmres, seq);
              HIPCHK(c, hipGetLastError());
            }
            c->prof = prof_saved;
            // poll the granules in order: each one is valid as soon as its tag is this evaluation's
            unsigned long long spins = 0;
            double blk[LM_MOM_MAX];
            for (int j = 0; j < (int)M::NMOM_LM; j++) {
              unsigned long long g0, g1;
              while ((uint32_t)(g0 = res[2 * j]) != seq || (uint32_t)(g1 = res[2 * j + 1]) != seq) {
                if ((++spins & 0xFFFF) == 0) {  // every 65 k polls: is the stream still alive?
                  hipError_t q = hipStreamQuery(c->stream);
                  if (q != hipSuccess && q != hipErrorNotReady)
                    return fail(c, LSQR_ERR_HIP, "LM pass failed: %s", hipGetErrorString(q));
                  if (q == hipSuccess && ((uint32_t)res[2 * j] != seq || (uint32_t)res[2 * j + 1] != seq))
                    return fail(c, LSQR_ERR_HIP, "LM pass finished without publishing its result");
                }
              }
              const unsigned long long bits = (g0 & 0xFFFFFFFF00000000ULL) | (g1 >> 32);
              memcpy(&blk[j], &bits, 8);
            }
            if (!lm_advance(s, blk)) break;
          }
          HIPCHK(c, sync_stream(c));
        } else
        for (;;) {
          for (int j = 0; j < n; j++) pin[j] = s.xtrial[j];
          HIPCHK(c, hipMemcpyAsync(c->d_vec, pin, sizeof(double) * n, hipMemcpyHostToDevice,
                                   c->stream));
          if ((st = launch_moments<M>(c, use_mask, 0, c->n, 1, &nmom)) != LSQR_OK) return st;
          HIPCHK(c, hipMemcpyAsync(pin + 64, c->d_mom, sizeof(double) * nmom,
                                   hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, sync_stream(c));
          if (!lm_advance(s, pin + 64)) break;
        }
        bool ok = s.info >= 1 && s.info <= 4;  // vnl_levenberg_marquardt::minimize -> true
        {
          static const bool dbg_on = getenv("LSQR_LM_DEBUG") != nullptr;
          if (dbg_on)
            fprintf(stderr, "lm: info %d nfev %d outer iterations %d (accepted steps %d) cost %.17g stall %d par %.3g delta %.3g\n",
                    s.info, s.nfev, s.iter, s.iter - 1, s.fnorm * s.fnorm, s.stall, s.par, s.delta);
        }
        out->ok = ok ? 1 : 0;
        out->cont = 0;
        out->lm_info = s.info;
        out->lm_nfev = s.nfev;
        out->pad = s.stall;
        out->cost = s.fnorm * s.fnorm;
        int np = M::lm_finalize(s.x, out->params);
        out->n_params = ok ? np : 0;
        return LSQR_OK;
      }
    }
    hipLaunchKernelGGL(k_lm_init, dim3(1), dim3(64), 0, c->stream, c->d_lm, c->d_out, n, ftol,
                       xtol, gtol, maxfev, 100.0);
    HIPCHK(c, hipGetLastError());
    for (;;) {
      HIPCHK(c, hipMemcpyAsync(c->d_vec, (const char *)c->d_lm + offsetof(LmState, xtrial),
                               sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
      if ((st = launch_moments<M>(c, use_mask, 0, c->n, 1, &nmom)) != LSQR_OK) return st;
      {
        ProfScope ps(c, KID_SOLVE);
        if constexpr (requires { M::NMOM_LM; })
          hipLaunchKernelGGL((k_lm_advance<M>), dim3(1), dim3(64), 0, c->stream, c->d_lm, c->d_mom,
                             c->d_out);
        HIPCHK(c, hipGetLastError());
      }
      if ((st = read_out(c, out)) != LSQR_OK) return st;
      if (!out->cont) break;
    }
    return LSQR_OK;
    }
  });
}

int launch_mask(lsqr_ctx *c, size_t begin, size_t end) {
  int st = ensure(c, &c->d_mask, &c->mask_cap, c->n);
  if (st != LSQR_OK) return st;
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    HIPCHK(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
    if constexpr (M::SP > M::P && !M::IS_DENSE) {
      hipLaunchKernelGGL((k_prepare<M>), dim3(1), dim3(64), 0, c->stream, c->d_par, c->mc);
      HIPCHK(c, hipGetLastError());
    }
    ProfScope ps(c, KID_MASK);
    if constexpr (M::IS_DENSE) {  // wide rows: LDS-staged, coalesced
      const int n = c->cfg.dim;
      const size_t lds = sizeof(double) * 4 * 32 * ((n + 1) | 1);
      int grid = grid_for(end - begin, 4 * 32 * 4, 256 * 4);
      hipLaunchKernelGGL((k_mask_dense<M::NR>), dim3(grid), dim3(256), lds, c->stream, c->d_data,
                         c->stride, begin, end, n, c->d_par, c->mc.delta, c->d_mask, c->d_counter);
      HIPCHK(c, hipGetLastError());
      return LSQR_OK;
    }
    int grid = grid_for(end - begin, kBlock * 8, 256 * 8);
    hipLaunchKernelGGL((k_mask<M>), dim3(grid), dim3(kBlock), 0, c->stream, c->d_data, c->stride,
                       begin, end, c->d_par, c->mc, c->d_mask, c->d_counter);
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  });
  if (st != LSQR_OK) return st;
  c->mask_valid = true;
  c->origin_valid = true;
  return LSQR_OK;
}

int launch_mask_moments(lsqr_ctx *c, size_t begin, size_t end, int *nmom, bool *fused);
int run_mask(lsqr_ctx *c, size_t begin, size_t end, uint8_t *mask_out, uint64_t *count_out) {
  int st, nm = 0;
  bool fused = false;
  // dense system: the one-pass kernel (mask + block of sums, dense.h: k_mask_syrk_dense) is the faster mask as well
  if (c->cfg.model == LSQR_MODEL_DENSE && (st = launch_mask_moments(c, begin, end, &nm, &fused)) != LSQR_OK) return st;
  if (!fused && (st = launch_mask(c, begin, end)) != LSQR_OK) return st;
  HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter, sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, c->stream));
  if (mask_out)
    HIPCHK(c, hipMemcpyAsync(mask_out, c->d_mask + begin, end - begin, hipMemcpyDeviceToHost,
                             c->stream));
  HIPCHK(c, sync_stream(c));
  if (count_out) *count_out = *(unsigned long long *)c->h_pin;
  return LSQR_OK;
}

// mask of d_par over [begin, end) and the phase-0 moment block of the agreeing records about d_vec (set by the
// caller) in ONE pass; -> d_mask, d_counter[0], d_mom.  *fused = false (nothing launched) for the models
// whose moments are not a per-record accumulate (dense: SYRK on the matrix cores; phantom: Gram of the rows).
int launch_mask_moments(lsqr_ctx *c, size_t begin, size_t end, int *nmom, bool *fused) {
  *fused = false;
  if (c->cfg.model == LSQR_MODEL_PHANTOM || !c->opt_fuse_mask) return LSQR_OK;
  int st = ensure(c, &c->d_mask, &c->mask_cap, c->n);
  if (st != LSQR_OK) return st;
  if (c->cfg.model == LSQR_MODEL_DENSE) {
    // dense system: mask + sum z z^T of the agreeing rows in one pass (dense.h: k_mask_syrk_dense); tight records
    // and the matrix-core path only -- anything else keeps the two kernels
    const int n = c->cfg.dim, nz = n + 1;
    if (c->stride != (size_t)nz || !c->opt_filter || c->opt_syrk_diag || end <= begin ||
        ((uintptr_t)c->d_data & 15) != 0 || ((begin * (size_t)nz) & 1) != 0)  // 16-byte pieces from the first row on
      return LSQR_OK;
    if ((st = ensure_absmax(c)) != LSQR_OK) return st;  // magnitudes for the band of the four-chain evaluation
    HIPCHK(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
    const size_t cnt = end - begin;
    const int nbuf = (c->opt_mask_ring == 4 && n == 64) ? 4 : 2;
    int nb = grid_for(cnt, 64 * 4, nbuf == 2 ? 512 : 256);   // two workgroups per CU / one
    size_t chunk = (cnt + nb - 1) / nb;
    chunk = (chunk + 63) / 64 * 64;                 // whole rounds of the four waves' 16-row tiles
    nb = (int)((cnt + chunk - 1) / chunk);
    const int ps = dense_pstride(n), na16 = (n + 15) / 16;
    // ring + model + waiting rows while streaming; the fold area afterwards
    const size_t lds = std::max<size_t>(sizeof(double) * (4 * nbuf * 16 * nz + 64 + 4 * 3 * nz),
                                        sizeof(double) * (10 * 256 + 80));
    if ((size_t)nb * ps > (size_t)2 * kDenseBlocks * 2160) return LSQR_OK;  // (partials area: 512 x 2160 doubles)
    *nmom = dense_ne(n) + 1;
    {
      ProfScope ps_(c, KID_MASK);
      auto launch = [&](auto kern) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, c->stream, c->d_data, begin, end, chunk, n, c->d_par,
                           c->mc.delta, c->d_mask, c->d_counter, ps, c->d_partials, c->mc.absmax_rot, c->mc.absmax,
                           c->opt_mask_band > 0 ? (double)c->opt_mask_band : 1.0, c->opt_mask_diag);
      };
      if (nbuf == 4) {
        launch(k_mask_syrk_dense<4, 4>);   // n = 64: one workgroup per CU, three tiles in flight per wave
      } else
      switch (na16) {
        case 1: launch(k_mask_syrk_dense<1, 2>); break;
        case 2: launch(k_mask_syrk_dense<2, 2>); break;
        case 3: launch(k_mask_syrk_dense<3, 2>); break;
        default: launch(k_mask_syrk_dense<4, 2>); break;
      }
      HIPCHK(c, hipGetLastError());
    }
    {
      ProfScope ps_(c, KID_SOLVE);
      hipLaunchKernelGGL(k_reduce, dim3(*nmom), dim3(64), 0, c->stream, c->d_partials, nb, ps, *nmom, c->d_mom);
      HIPCHK(c, hipGetLastError());
    }
    c->mask_valid = true;
    c->origin_valid = true;
    *fused = true;
    return LSQR_OK;
  }
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (M::IS_DENSE || requires { M::IS_PHANTOM; }) {
      return LSQR_ERR_INVALID;
    } else {
      HIPCHK(c, hipMemsetAsync(c->d_counter, 0, sizeof(unsigned long long), c->stream));
      if constexpr (M::SP > M::P) {
        hipLaunchKernelGGL((k_prepare<M>), dim3(1), dim3(64), 0, c->stream, c->d_par, c->mc);
        HIPCHK(c, hipGetLastError());
      }
      size_t cnt = end - begin;
      int nb = grid_for(cnt, mom_chunk(c), kMaxPartials);  // the chunking of launch_moments
      size_t chunk = (cnt + nb - 1) / nb;
      chunk = (chunk + kBlock - 1) / kBlock * kBlock;
      nb = (int)((cnt + chunk - 1) / chunk);
      if (nb < 1) nb = 1;
      *nmom = M::NMOM;
      bool done = false;
      if constexpr (M::IS_US) {
        // US calibrations with tight records: rows of the agreeing frames to the fp64 matrix cores (kernels.h)
        if (c->opt_us_mask_mfma && c->stride == (size_t)M::REC && cnt > 0) {
          nb = (int)std::min<size_t>(1024, (cnt + 4 * 64 * 2 - 1) / (4 * 64 * 2));  // >= two tiles per wave: one in flight
          if (nb < 1) nb = 1;
          ProfScope ps(c, KID_MASK);
          hipLaunchKernelGGL((k_mask_moments_us_mfma<M>), dim3(nb), dim3(kBlock), 0, c->stream, c->d_data, begin, end,
                             c->d_par, c->mc, c->d_mask, c->d_counter, c->d_partials);
          HIPCHK(c, hipGetLastError());
          done = true;
        }
      }
      if (!done) {
        ProfScope ps(c, KID_MASK);
        hipLaunchKernelGGL((k_mask_moments<M>), dim3(nb), dim3(kBlock), 0, c->stream, c->d_data, c->stride,
                           begin, end, chunk, c->d_par, c->d_vec, c->mc, c->d_mask, c->d_counter,
                           c->d_partials);
        HIPCHK(c, hipGetLastError());
      }
      ProfScope ps(c, KID_SOLVE);
      hipLaunchKernelGGL(k_reduce, dim3(*nmom), dim3(64), 0, c->stream, c->d_partials, nb, (int)MOM_MAX,
                         *nmom, c->d_mom);
      HIPCHK(c, hipGetLastError());
      return LSQR_OK;
    }
  });
  if (st != LSQR_OK) return st;
  c->mask_valid = true;
  c->origin_valid = true;
  *fused = true;
  return LSQR_OK;
}

// origin of a masked fit as run_fit chooses it: the model's own point (d_par) once a mask exists, the first
// record for the models whose parameters hold no point
int set_fit_origin(lsqr_ctx *c, bool from_model) {
  return dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (M::IS_DENSE || requires { M::IS_PHANTOM; }) {
      return LSQR_OK;
    } else {
      bool first_datum = !from_model;
      if constexpr (requires { M::ORIGIN_FIRST; }) first_datum = true;
      if (first_datum) {
        HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_data, sizeof(double) * M::ND, hipMemcpyDeviceToDevice,
                                 c->stream));
      } else {
        HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_par + (c->cfg.model == LSQR_MODEL_SPHERE ? 0 : M::ND),
                                 sizeof(double) * M::ND, hipMemcpyDeviceToDevice, c->stream));
      }
      return LSQR_OK;
    }
  });
}

// the winner's scan parameters -> d_par, on the device (no host round trip between scan and mask)
__global__ void k_take_best(const unsigned long long *__restrict__ packed,
                            const double *__restrict__ hparams, int hs, double *__restrict__ par) {
  const unsigned long long pk = *packed;
  const int t = threadIdx.x;
  for (int k = hs + t; k < 128; k += blockDim.x) par[k] = 0.0;  // (the block holds 128 doubles)
  if (t >= hs) return;
  if (pk == 0) {
    par[t] = __builtin_nan("");  // no valid hypothesis: nothing agrees
    return;
  }
  const size_t idx = 0xFFFFFFFFull - (pk & 0xFFFFFFFFull);
  par[t] = hparams[idx * (size_t)hs + t];
}

}  // namespace

// =================================================================================================
extern "C" {

const char *lsqr_version(void) {
#ifdef LSQR_DEV_SUBSET
  return "lsqrrecipes_amd 0.2 (gfx950) DEVELOPMENT SUBSET BUILD";
#else
  return "lsqrrecipes_amd 0.2 (gfx950)";
#endif
}

const char *lsqr_status_string(int s) {
  switch (s) {
    case LSQR_OK: return "ok";
    case LSQR_EMPTY: return "empty result (degenerate data or failed fit)";
    case LSQR_ERR_INVALID: return "invalid argument";
    case LSQR_ERR_NO_DEVICE: return "no usable HIP device";
    case LSQR_ERR_HIP: return "HIP runtime error";
    case LSQR_ERR_STATE: return "call order error";
  }
  return "unknown status";
}

int lsqr_device_count(int *count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (count) *count = (e == hipSuccess) ? n : 0;
  return e == hipSuccess ? LSQR_OK : LSQR_ERR_NO_DEVICE;
}

int lsqr_ctx_create(int device, lsqr_ctx **out) {
  if (!out) return LSQR_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
    return LSQR_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return LSQR_ERR_NO_DEVICE;
  lsqr_ctx *c = new lsqr_ctx();
  c->device = device;
  bool ok = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess &&
            hipMalloc((void **)&c->d_partials, sizeof(double) * 2 * kDenseBlocks * 2160) == hipSuccess &&
            hipMalloc((void **)&c->d_mom, sizeof(double) * 4096) == hipSuccess &&
            hipMalloc((void **)&c->d_vec, sizeof(double) * 128) == hipSuccess &&
            hipMalloc((void **)&c->d_par, sizeof(double) * 128) == hipSuccess &&
            hipMalloc((void **)&c->d_best, sizeof(double) * 128) == hipSuccess &&
            hipMalloc((void **)&c->d_lm, sizeof(LmState)) == hipSuccess &&
            hipMalloc((void **)&c->d_out, sizeof(SolveOut)) == hipSuccess &&
            hipMalloc((void **)&c->d_counter, 64) == hipSuccess &&
            hipHostMalloc(&c->h_pin, 1 << 16) == hipSuccess &&
            hipHostMalloc((void **)&c->h_lmres, sizeof(unsigned long long) * 256, hipHostMallocCoherent) == hipSuccess;
  if (ok) {
    memset(c->h_lmres, 0, sizeof(unsigned long long) * 256);
    ok = hipMemsetAsync(c->d_counter, 0, 64, c->own_stream) == hipSuccess;
  }
  c->stream = c->own_stream;
  if (!ok) {
    lsqr_ctx_destroy(c);
    return LSQR_ERR_HIP;
  }
  *out = c;
  lmp_ctx_count(device, +1);
  c->counted = true;
  return LSQR_OK;
}

void lsqr_ctx_destroy(lsqr_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  lmp_ctx_count(c->device, c->counted ? -1 : 0, c);
  for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
    if (c->lanes[i]) {
      lsqr_ctx_destroy(c->lanes[i]);
      c->lanes[i] = nullptr;
    }
  (void)hipStreamSynchronize(c->stream);
  free_index(c);
  void *bufs[] = {c->d_refused, c->d_us16, c->d_us16_x, c->d_h16, c->d_h16_bs, c->d_h16_thr, c->d_ddpart, c->d_ub2, c->d_axis, c->d_cellT, c->d_vpart, c->d_paircnt, c->d_paircost, c->d_sel, c->d_bsel, c->d_hparams2, c->d_hparams2_f32, c->d_votes2, c->d_lmrec, c->d_idx_scratch, c->d_ub, c->d_queues, c->d_data_owned, c->d_subsets, c->d_hparams, c->d_hparams_f32, c->d_amb, c->d_valid, c->d_votes, c->d_mask, c->d_rows,
                  c->d_partials, c->d_mom, c->d_vec, c->d_par, c->d_best, c->d_lm, c->d_out, c->d_counter};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  if (c->h_bsel) (void)hipHostFree(c->h_bsel);
  for (int r = 0; r < 2; r++)
    if (c->bsel_ev[r]) (void)hipEventDestroy(c->bsel_ev[r]);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->h_lmres) (void)hipHostFree(c->h_lmres);
  if (c->h_lmcmd) (void)hipHostFree(c->h_lmcmd);
  if (c->d_lmp) (void)hipFree(c->d_lmp);
  if (c->h_batch) (void)hipHostFree(c->h_batch);
  for (int i = 0; i < lsqr_ctx::kUpSlots; i++) {
    if (c->h_up[i]) (void)hipHostFree(c->h_up[i]);
    if (c->up_ev[i]) (void)hipEventDestroy(c->up_ev[i]);
  }
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t e : c->slot_ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->step_ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->mdev_ev)
    if (e) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char *lsqr_last_error(const lsqr_ctx *c) { return c ? c->err : "null context"; }

int lsqr_synchronize(lsqr_ctx *c) {
  if (!c) return LSQR_ERR_INVALID;
  HIPCHK(c, sync_stream(c));
  for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
    if (c->lanes[i]) HIPCHK(c, hipStreamSynchronize(c->lanes[i]->stream));
  return LSQR_OK;
}

// ---- model description ---------------------------------------------------------------------------
// ---- entry points that need no device: host_entry.h -------------------------------------------------------------------
int lsqr_min_subset(const lsqr_model_cfg *cfg) { return host_min_subset(cfg); }
int lsqr_num_params(const lsqr_model_cfg *cfg) { return host_num_params(cfg); }
int lsqr_record_doubles(const lsqr_model_cfg *cfg) { return host_record_doubles(cfg); }
int lsqr_sample_subsets(uint64_t seed, uint64_t first, size_t H, uint64_t n, int k, uint32_t *out) {
  return host_sample_subsets(seed, first, H, n, k, out);
}
void *lsqr_dedup_create(int k) { return host_dedup_create(k); }
void lsqr_dedup_destroy(void *s) { host_dedup_destroy(s); }
int lsqr_replay_init(size_t n, int k, double p, uint64_t st[6]) { return host_replay_init(n, k, p, st); }
size_t lsqr_replay(size_t n, int k, double p, const uint32_t *subsets, const uint8_t *valid, const uint32_t *votes,
                   size_t H, uint64_t base_index, void *dedup, uint64_t st[6]) {
  return host_replay(n, k, p, subsets, valid, votes, H, base_index, dedup, st);
}
int lsqr_agree_host(const lsqr_model_cfg *cfg, const double *params, const void *record, int *agree_out) {
  return host_agree_host(cfg, params, record, agree_out);
}
int lsqr_estimate_host(const lsqr_model_cfg *cfg, const void *records, size_t count, size_t stride_bytes,
                       double *params_out, int *n_params_out) {
  return host_estimate_host(cfg, records, count, stride_bytes, params_out, n_params_out);
}

int lsqr_set_model(lsqr_ctx *c, const lsqr_model_cfg *cfg) {
  if (!c || !cfg) return LSQR_ERR_INVALID;
  if (!cfg_supported(*cfg))
    return fail(c, LSQR_ERR_INVALID, "unsupported model %d / dim %d", cfg->model, cfg->dim);
  if (cfg->model == LSQR_MODEL_SPHERE && cfg->ls_type != LSQR_LS_ALGEBRAIC &&
      cfg->ls_type != LSQR_LS_GEOMETRIC)  // SphereParametersEstimator.hxx:17-18 throws
    return fail(c, LSQR_ERR_INVALID, "invalid sphere least squares type %d", cfg->ls_type);
  lanes_quiesce(c);
  c->data_epoch++;
  // The same estimator type with another threshold / fit type (RANSAC<T,S>::compute() again on resident records,
  // lsqrRecipes::ResidentData): what was derived from the records alone -- their bounds and magnitudes, the spatial
  // index -- stays valid.
  const bool same_shape = c->has_model && c->cfg.model == cfg->model && c->cfg.dim == cfg->dim;
  const double absmax_keep = c->mc.absmax, absrot_keep = c->mc.absmax_rot;
  c->cfg = *cfg;
  model_consts(*cfg, &c->mc);
  c->K = lsqr_min_subset(cfg);
  c->P = lsqr_num_params(cfg);
  c->ND = lsqr_record_doubles(cfg);
  c->HS = dispatch(*cfg, [](auto tag) { return (int)decltype(tag)::type::SP; });
  c->has_model = true;
  if (c->n && c->stride < (size_t)c->ND) c->n = 0;  // the resident records cannot be read as this model's
  if (same_shape && c->n) {
    c->mc.absmax = absmax_keep;
    c->mc.absmax_rot = absrot_keep;
  } else {
    drop_index(c);
    c->absmax_valid = false;
    c->bounds_valid = false;
  }
  c->rows_valid = false;
  c->H = 0;
  c->scanned = false;
  c->mask_valid = false;
  c->origin_valid = false;
  c->merge_off = false;   // selection feedback belongs to one model (threshold included) on one upload
  bsel_reset(c);
  return LSQR_OK;
}

// ---- observations ----------------------------------------------------------------------------------
static int set_data_common(lsqr_ctx *c, size_t count, size_t stride_bytes) {
  if (stride_bytes % sizeof(double) != 0 || stride_bytes / sizeof(double) < (size_t)c->ND)
    return fail(c, LSQR_ERR_INVALID, "record stride %zu B does not hold %d doubles", stride_bytes,
                c->ND);
  if (count > 0xFFFFFFF0ull) return fail(c, LSQR_ERR_INVALID, "too many observations");
  lanes_quiesce(c);  // lanes read the records this call is about to replace
  c->data_epoch++;
  c->merge_off = false;
  bsel_reset(c);
  c->n = count;
  c->absmax_valid = false;
  c->bounds_valid = false;
  c->rows_valid = false;
  drop_index(c);
  c->index_failed = false;
  c->hyp_since_upload = 0;
  c->hyp_expected = 0;
  c->stride = stride_bytes / sizeof(double);
  c->H = 0;
  c->scanned = false;
  c->mask_valid = false;
  c->origin_valid = false;
  return LSQR_OK;
}

// Host -> device copy of a large pageable buffer.  A plain hipMemcpy stages pageable memory through the
// runtime's own bounce buffers with one host thread (measured ~5 GB/s: 50 ms for the 240 MB of 10 M points);
// here a few threads copy 8 MiB chunks into a ring of pinned slots and enqueue each slot's DMA as soon as it is
// filled, so the host copies and the PCIe transfers overlap.
static int staged_upload(lsqr_ctx *c, void *dst, const void *src, size_t bytes, int threads) {
  for (int i = 0; i < lsqr_ctx::kUpSlots; i++) {
    if (!c->h_up[i]) HIPCHK(c, hipHostMalloc(&c->h_up[i], lsqr_ctx::kUpChunk));
    if (!c->up_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->up_ev[i], hipEventDisableTiming));
  }
  const size_t chunk = lsqr_ctx::kUpChunk, nchunks = (bytes + chunk - 1) / chunk;
  std::atomic<size_t> next{0};
  std::atomic<int> err{(int)hipSuccess};
  // chunk i uses slot i % kUpSlots; it may be overwritten once chunk i - kUpSlots has left it (its event)
  std::vector<std::atomic<int>> issued(nchunks);
  for (auto &f : issued) f.store(0);
  auto work = [&]() {
    if (hipSetDevice(c->device) != hipSuccess) {
      err.store((int)hipErrorInvalidDevice);
      return;
    }
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= nchunks || err.load() != (int)hipSuccess) return;
      const int slot = (int)(i % lsqr_ctx::kUpSlots);
      if (i >= (size_t)lsqr_ctx::kUpSlots) {
        const size_t prev = i - lsqr_ctx::kUpSlots;
        while (!issued[prev].load(std::memory_order_acquire)) std::this_thread::yield();
        hipError_t e = hipEventSynchronize(c->up_ev[slot]);
        if (e != hipSuccess) {
          err.store((int)e);
          issued[i].store(1, std::memory_order_release);
          return;
        }
      }
      const size_t off = i * chunk, len = std::min(chunk, bytes - off);
      memcpy(c->h_up[slot], (const char *)src + off, len);
      hipError_t e = hipMemcpyAsync((char *)dst + off, c->h_up[slot], len, hipMemcpyHostToDevice, c->stream);
      if (e == hipSuccess) e = hipEventRecord(c->up_ev[slot], c->stream);
      if (e != hipSuccess) err.store((int)e);
      issued[i].store(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  const int nt = (int)std::min<size_t>((size_t)threads, nchunks);
  for (int t = 1; t < nt; t++) pool.emplace_back(work);
  work();
  for (auto &t : pool) t.join();
  if (err.load() != (int)hipSuccess)
    return fail(c, LSQR_ERR_HIP, "staged upload failed: %s", hipGetErrorString((hipError_t)err.load()));
  return LSQR_OK;
}

int lsqr_upload(lsqr_ctx *c, const void *host, size_t count, size_t stride_bytes) {
  int st = need_ready(c, false);
  if (st != LSQR_OK) return st;
  if (!host && count) return fail(c, LSQR_ERR_INVALID, "null records");
  if ((st = set_data_common(c, count, stride_bytes)) != LSQR_OK) return st;
  size_t doubles = std::max<size_t>(count * c->stride, 1);
  if ((st = ensure(c, &c->d_data_owned, &c->data_cap, doubles)) != LSQR_OK) return st;
  const size_t bytes = count * stride_bytes;
  // measured on the MI355X box (bench.py cold_call, 240 MB from a pageable numpy buffer): one plain
  // hipMemcpy 4.3 ms = 56 GB/s, the staged ring with 4 threads 5.1 ms -- so the plain copy is the default and
  // the staged path stays an option for hosts whose runtime stages pageable memory slowly
  int threads = c->opt_upload_threads;
  if (threads < 0) {
    const char *e = getenv("LSQR_UPLOAD_THREADS");
    threads = e ? atoi(e) : 0;
  }
  threads = std::max(0, std::min(threads, 16));
  const auto t0 = std::chrono::steady_clock::now();
  if (bytes >= ((size_t)32 << 20) && threads > 0) {
    if ((st = staged_upload(c, c->d_data_owned, host, bytes, threads)) != LSQR_OK) return st;
  } else if (count) {
    HIPCHK(c, hipMemcpyAsync(c->d_data_owned, host, bytes, hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, sync_stream(c));
  c->last_upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  c->d_data = c->d_data_owned;
  return LSQR_OK;
}

int lsqr_attach(lsqr_ctx *c, const void *dev, size_t count, size_t stride_bytes) {
  int st = need_ready(c, false);
  if (st != LSQR_OK) return st;
  if (!dev && count) return fail(c, LSQR_ERR_INVALID, "null records");
  if ((st = set_data_common(c, count, stride_bytes)) != LSQR_OK) return st;
  c->d_data = (const double *)dev;
  return LSQR_OK;
}

size_t lsqr_count(const lsqr_ctx *c) { return c ? c->n : 0; }

// ---- hypotheses ------------------------------------------------------------------------------------
int lsqr_hypotheses_from_subsets(lsqr_ctx *c, const uint32_t *subsets, size_t H) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!subsets || H == 0 || H > (1u << 20)) return fail(c, LSQR_ERR_INVALID, "bad subset batch");
  if (c->n < (size_t)c->K) return fail(c, LSQR_ERR_INVALID, "fewer observations than a subset");
  if ((st = ensure_hyp(c, H)) != LSQR_OK) return st;
  c->H = H;
  c->scanned = false;
  HIPCHK(c, hipMemcpyAsync(c->d_subsets, subsets, H * c->K * sizeof(uint32_t),
                           hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, sync_stream(c));  // caller may free `subsets` on return
  return run_estimate(c);
}

int lsqr_hypotheses_sample(lsqr_ctx *c, uint64_t seed, uint64_t first, size_t H,
                           uint32_t *subsets_out) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (H == 0 || H > (1u << 20)) return fail(c, LSQR_ERR_INVALID, "bad batch size");
  if (c->n < (size_t)c->K) return fail(c, LSQR_ERR_INVALID, "fewer observations than a subset");
  if ((st = ensure_hyp(c, H)) != LSQR_OK) return st;
  c->H = H;
  c->scanned = false;
  {
    ProfScope ps(c, KID_SAMPLE);
    int grid = (int)((H + kBlock - 1) / kBlock);
    if (c->K <= 4)
      hipLaunchKernelGGL((k_sample<4>), dim3(grid), dim3(kBlock), 0, c->stream, seed, first,
                         (uint32_t)H, (uint64_t)c->n, c->K, c->d_subsets);
    else  // large subsets: one wave per hypothesis
      hipLaunchKernelGGL(k_sample_wave, dim3((unsigned)H), dim3(64), 0, c->stream, seed, first,
                         (uint32_t)H, (uint64_t)c->n, c->K, c->d_subsets);
    HIPCHK(c, hipGetLastError());
  }
  if ((st = run_estimate(c)) != LSQR_OK) return st;
  if (subsets_out) {
    HIPCHK(c, hipMemcpyAsync(subsets_out, c->d_subsets, H * c->K * sizeof(uint32_t),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c));
  }
  return LSQR_OK;
}

int lsqr_scan(lsqr_ctx *c) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (c->H == 0) return fail(c, LSQR_ERR_STATE, "no hypotheses to scan");
  if ((st = run_scan(c)) != LSQR_OK) return st;
  c->scanned = true;
  return LSQR_OK;
}

size_t lsqr_num_hypotheses(const lsqr_ctx *c) { return c ? c->H : 0; }

int lsqr_get_hypotheses(lsqr_ctx *c, double *params, uint8_t *valid, uint32_t *votes) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (c->H == 0) return fail(c, LSQR_ERR_STATE, "no hypotheses");
  if (votes && !c->scanned) return fail(c, LSQR_ERR_STATE, "lsqr_scan has not run");
  if (params)
    HIPCHK(c, hipMemcpy2DAsync(params, c->P * sizeof(double), c->d_hparams,
                               c->HS * sizeof(double), c->P * sizeof(double), c->H,
                               hipMemcpyDeviceToHost, c->stream));
  if (valid)
    HIPCHK(c, hipMemcpyAsync(valid, c->d_valid, c->H, hipMemcpyDeviceToHost, c->stream));
  if (votes)
    HIPCHK(c, hipMemcpyAsync(votes, c->d_votes, c->H * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             c->stream));
  HIPCHK(c, sync_stream(c));
  return LSQR_OK;
}

int lsqr_get_hypothesis(lsqr_ctx *c, size_t h, double *params, uint8_t *valid) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (h >= c->H) return fail(c, LSQR_ERR_INVALID, "hypothesis index out of range");
  double *hp = (double *)c->h_pin;
  HIPCHK(c, hipMemcpyAsync(hp, c->d_hparams + h * c->HS, sizeof(double) * c->P,
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(hp + 64, c->d_valid + h, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  if (params) memcpy(params, hp, sizeof(double) * c->P);
  if (valid) *valid = *(uint8_t *)(hp + 64);
  return LSQR_OK;
}

int lsqr_best(lsqr_ctx *c, uint64_t *packed) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!packed || !c->scanned) return fail(c, LSQR_ERR_STATE, "lsqr_scan has not run");
  hipLaunchKernelGGL(k_best, dim3(1), dim3(kBlock), 0, c->stream, c->d_votes, c->d_valid,
                     (uint32_t)c->H, c->d_counter + 1);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_counter + 1, sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  *packed = *(unsigned long long *)c->h_pin;
  return LSQR_OK;
}

// ---- mask ----------------------------------------------------------------------------------------
int lsqr_mask(lsqr_ctx *c, const double *params, size_t begin, size_t end, uint8_t *mask_out,
              uint64_t *count_out) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!params || begin > end || end > c->n) return fail(c, LSQR_ERR_INVALID, "bad mask range");
  HIPCHK(c, hipMemsetAsync(c->d_par, 0, sizeof(double) * 128, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_par, params, sizeof(double) * c->P, hipMemcpyHostToDevice,
                           c->stream));
  HIPCHK(c, sync_stream(c));
  return run_mask(c, begin, end, mask_out, count_out);
}

int lsqr_mask_from_hypothesis(lsqr_ctx *c, size_t h, uint8_t *mask_out, uint64_t *count_out) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (h >= c->H) return fail(c, LSQR_ERR_INVALID, "hypothesis index out of range");
  HIPCHK(c, hipMemcpyAsync(c->d_par, c->d_hparams + h * c->HS, sizeof(double) * c->HS,
                           hipMemcpyDeviceToDevice, c->stream));
  return run_mask(c, 0, c->n, mask_out, count_out);
}

int lsqr_set_mask(lsqr_ctx *c, const uint8_t *mask) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!mask) return fail(c, LSQR_ERR_INVALID, "null mask");
  if ((st = ensure(c, &c->d_mask, &c->mask_cap, c->n)) != LSQR_OK) return st;
  HIPCHK(c, hipMemcpyAsync(c->d_mask, mask, c->n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, sync_stream(c));
  c->mask_valid = true;
  c->origin_valid = false;
  return LSQR_OK;
}

// ---- final fit -------------------------------------------------------------------------------------
int lsqr_ls_fit(lsqr_ctx *c, int use_mask, double *params_out, lsqr_fit_info *info) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (use_mask && !c->mask_valid) return fail(c, LSQR_ERR_STATE, "no mask on the device");
  if (!use_mask) c->origin_valid = false;
  SolveOut out;
  memset(&out, 0, sizeof out);
  if ((st = run_fit(c, use_mask, &out)) != LSQR_OK) return st;
  if (info) {
    memset(info, 0, sizeof *info);
    info->n_params = out.n_params;
    info->lm_info = out.lm_info;
    info->lm_nfev = out.lm_nfev;
    info->reserved = out.pad;  // LM: stall evaluation; dense: 1 = the double-double route produced the result
    info->cost = out.cost;
  }
  if (!out.ok) {
    // a Levenberg-Marquardt run that MINPACK reports as failed (the reference then returns an empty vector):
    // the last iterate is still handed out for diagnostics -- status LSQR_EMPTY and info->n_params == 0 say
    // that it is not a result
    if (params_out && out.lm_info != 0)
      for (int j = 0; j < c->P; j++) params_out[j] = out.params[j];
    return LSQR_EMPTY;
  }
  if (params_out)
    for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
  return LSQR_OK;
}

int lsqr_moments_len(const lsqr_model_cfg *cfg, int phase) {
  if (!cfg) return 0;
  return dispatch(*cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (requires { M::IS_PHANTOM; }) {
      return phase == 0 || phase == 1 ? dense_ne(30) + 1 : 0;  // phase 1: the same block (lsqr_hip.h)
    } else if constexpr (M::IS_DENSE) {
      return phase == 0 ? dense_ne(cfg->dim) + 1 : 0;
    } else {
      if (phase == 0) return (int)M::NMOM;
      if constexpr (requires { M::NMOM_LM; }) return (int)M::NMOM_LM;
      return 0;
    }
  });
}

int lsqr_moments(lsqr_ctx *c, int use_mask, size_t begin, size_t end, int phase, const double *x,
                 double *block_out) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (begin > end || end > c->n || !block_out) return fail(c, LSQR_ERR_INVALID, "bad range");
  if (use_mask && !c->mask_valid) return fail(c, LSQR_ERR_STATE, "no mask on the device");
  if (!x) return fail(c, LSQR_ERR_INVALID, "moments need an origin / evaluation point");
  HIPCHK(c, hipMemcpyAsync(c->d_vec, x, sizeof(double) * 32, hipMemcpyHostToDevice, c->stream));
  int nmom = 0;
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    return launch_moments<M>(c, use_mask, begin, end, phase, &nmom);
  });
  if (st != LSQR_OK) return st;
  HIPCHK(c, hipMemcpyAsync(block_out, c->d_mom, sizeof(double) * nmom, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, sync_stream(c));
  return LSQR_OK;
}

// lsqr_moments with the block left in DEVICE memory (no synchronisation): the multi-GPU LM loop all-reduces
// it in place and reads it back once
int lsqr_moments_dev(lsqr_ctx *c, int use_mask, size_t begin, size_t end, int phase, const double *x,
                     double *block_dev) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (begin > end || end > c->n || !block_dev) return fail(c, LSQR_ERR_INVALID, "bad range");
  if (use_mask && !c->mask_valid) return fail(c, LSQR_ERR_STATE, "no mask on the device");
  if (!x) return fail(c, LSQR_ERR_INVALID, "moments need an origin / evaluation point");
  // x staged in pinned memory (the copy stays asynchronous): four slots in rotation, each guarded by an event, so
  // that a call issued before an earlier call's copy has executed cannot overwrite that copy's source
  const int slot = c->mdev_next++ & 3;
  if (!c->mdev_ev[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->mdev_ev[slot], hipEventDisableTiming));
  else HIPCHK(c, hipEventSynchronize(c->mdev_ev[slot]));
  double *pin = (double *)((char *)c->h_pin + 57344 + slot * 512);
  memcpy(pin, x, sizeof(double) * 32);
  HIPCHK(c, hipMemcpyAsync(c->d_vec, pin, sizeof(double) * 32, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->mdev_ev[slot], c->stream));
  int nmom = 0;
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    return launch_moments<M>(c, use_mask, begin, end, phase, &nmom);
  });
  if (st != LSQR_OK) return st;
  HIPCHK(c, hipMemcpyAsync(block_dev, c->d_mom, sizeof(double) * nmom, hipMemcpyDeviceToDevice, c->stream));
  return LSQR_OK;
}

static void fill_info(const SolveOut &out, lsqr_fit_info *info) {
  if (!info) return;
  memset(info, 0, sizeof *info);
  info->n_params = out.n_params;
  info->lm_info = out.lm_info;
  info->lm_nfev = out.lm_nfev;
  info->reserved = out.pad;  // LM: the evaluation after which the cost never again fell by more than 1e-7 relative
  info->cost = out.cost;
}

int lsqr_solve_moments(lsqr_ctx *c, const double *block, const double *origin, double *params_out,
                       lsqr_fit_info *info) {
  int st = need_ready(c, false);
  if (st != LSQR_OK) return st;
  if (!block || !origin) return fail(c, LSQR_ERR_INVALID, "null argument");
  int nmom = lsqr_moments_len(&c->cfg, 0);
  if (c->cfg.model == LSQR_MODEL_PHANTOM) {  // solved on the host from the Gram block
    SolveOut out;
    phantom_solve_block(c->cfg, block, &out);
    fill_info(out, info);
    if (!out.ok) return LSQR_EMPTY;
    if (params_out)
      for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
    return LSQR_OK;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_mom, block, sizeof(double) * nmom, hipMemcpyHostToDevice,
                           c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_vec, origin, sizeof(double) * std::min(c->ND, 32),
                           hipMemcpyHostToDevice, c->stream));
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (M::IS_DENSE) {
      return launch_solve_dense(c);
    } else {
      ProfScope ps(c, KID_SOLVE);
      hipLaunchKernelGGL((k_solve<M>), dim3(1), dim3(64), 0, c->stream, c->d_mom, c->d_vec, c->mc,
                         c->d_out);
      HIPCHK(c, hipGetLastError());
      return LSQR_OK;
    }
  });
  if (st != LSQR_OK) return st;
  SolveOut out;
  if ((st = read_out(c, &out)) != LSQR_OK) return st;
  fill_info(out, info);
  if (!out.ok) return LSQR_EMPTY;
  if (params_out)
    for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
  return LSQR_OK;
}

int lsqr_lm_begin(lsqr_ctx *c, const double *x0, double *x_trial_out) {
  int st = need_ready(c, false);
  if (st != LSQR_OK) return st;
  if (!wants_lm(c->cfg)) return fail(c, LSQR_ERR_INVALID, "model/ls_type has no iterative fit");
  if (!x0) return fail(c, LSQR_ERR_INVALID, "null x0");
  int n, maxfev;
  double ftol, xtol, gtol;
  lm_settings(c->cfg, &n, &ftol, &xtol, &gtol, &maxfev);
  if (c->opt_lm_host || c->cfg.model == LSQR_MODEL_PHANTOM) {
    lm_init(c->h_lm, n, x0, ftol, xtol, gtol, maxfev, 100.0);
    if (x_trial_out)
      for (int j = 0; j < n; j++) x_trial_out[j] = x0[j];
    return LSQR_OK;
  }
  SolveOut seed;
  memset(&seed, 0, sizeof seed);
  for (int j = 0; j < n; j++) seed.params[j] = x0[j];
  memcpy(c->h_pin, &seed, sizeof seed);
  HIPCHK(c, hipMemcpyAsync(c->d_out, c->h_pin, sizeof seed, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_lm_init, dim3(1), dim3(64), 0, c->stream, c->d_lm, c->d_out, n, ftol, xtol,
                     gtol, maxfev, 100.0);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, sync_stream(c));
  if (x_trial_out)
    for (int j = 0; j < n; j++) x_trial_out[j] = x0[j];
  return LSQR_OK;
}

int lsqr_lm_step(lsqr_ctx *c, const double *block, double *x_trial_out, int *cont,
                 double *params_out, lsqr_fit_info *info) {
  int st = need_ready(c, false);
  if (st != LSQR_OK) return st;
  if (!block || !cont) return fail(c, LSQR_ERR_INVALID, "null argument");
  int nmom = lsqr_moments_len(&c->cfg, 1);
  if (nmom <= 0) return fail(c, LSQR_ERR_INVALID, "model has no iterative phase");
  if (c->cfg.model == LSQR_MODEL_PHANTOM) {  // the whole minimisation on the (summed) Gram block
    double G[31 * 31];
    SolveOut out;
    PhantomFit f;
    phantom_unpack(block, G);
    phantom_lm(G, c->h_lm, &f);
    phantom_to_out(f, &out);
    *cont = 0;
    fill_info(out, info);
    if (!out.ok) return LSQR_EMPTY;
    if (params_out)
      for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
    return LSQR_OK;
  }
  if (c->opt_lm_host) {
    LmState &s = c->h_lm;
    bool go = lm_advance(s, block);
    *cont = go ? 1 : 0;
    lsqr_fit_info fi;
    memset(&fi, 0, sizeof fi);
    fi.lm_info = s.info;
    fi.lm_nfev = s.nfev;
    fi.reserved = s.stall;
    fi.cost = s.fnorm * s.fnorm;
    if (go) {
      if (x_trial_out)
        for (int j = 0; j < s.n; j++) x_trial_out[j] = s.xtrial[j];
      if (info) *info = fi;
      return LSQR_OK;
    }
    bool ok = s.info >= 1 && s.info <= 4;
    double par[64];
    int np = dispatch(c->cfg, [&](auto tag) -> int {
      typedef typename decltype(tag)::type M;
      if constexpr (requires { M::NMOM_LM; }) return M::lm_finalize(s.x, par);
      else return 0;
    });
    fi.n_params = ok ? np : 0;
    if (info) *info = fi;
    if (!ok) return LSQR_EMPTY;
    if (params_out)
      for (int j = 0; j < np; j++) params_out[j] = par[j];
    return LSQR_OK;
  }
  HIPCHK(c, hipMemcpyAsync(c->d_mom, block, sizeof(double) * nmom, hipMemcpyHostToDevice,
                           c->stream));
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (requires { M::NMOM_LM; }) {
      ProfScope ps(c, KID_SOLVE);
      hipLaunchKernelGGL((k_lm_advance<M>), dim3(1), dim3(64), 0, c->stream, c->d_lm, c->d_mom,
                         c->d_out);
      HIPCHK(c, hipGetLastError());
      return LSQR_OK;
    } else {
      return LSQR_ERR_INVALID;
    }
  });
  if (st != LSQR_OK) return st;
  SolveOut out;
  if ((st = read_out(c, &out)) != LSQR_OK) return st;
  *cont = out.cont;
  fill_info(out, info);
  if (out.cont) {
    if (x_trial_out) {
      HIPCHK(c, hipMemcpyAsync(c->h_pin, (const char *)c->d_lm + offsetof(LmState, xtrial),
                               sizeof(double) * LM_NMAX, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, sync_stream(c));
      memcpy(x_trial_out, c->h_pin, sizeof(double) * c->P);
    }
    return LSQR_OK;
  }
  if (!out.ok) return LSQR_EMPTY;
  if (params_out)
    for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
  return LSQR_OK;
}

int lsqr_stats(lsqr_ctx *c, const double *params, int use_mask, double out[4]) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!params || !out) return fail(c, LSQR_ERR_INVALID, "null argument");
  if (use_mask && !c->mask_valid) return fail(c, LSQR_ERR_STATE, "no mask on the device");
  HIPCHK(c, hipMemsetAsync(c->d_par, 0, sizeof(double) * 128, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_par, params, sizeof(double) * c->P, hipMemcpyHostToDevice,
                           c->stream));
  int nb = grid_for(c->n, mom_chunk(c), kMaxPartials);
  size_t chunk = (c->n + nb - 1) / nb;
  chunk = (chunk + kBlock - 1) / kBlock * kBlock;
  nb = (int)((c->n + chunk - 1) / chunk);
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if (use_mask)
      hipLaunchKernelGGL((k_stats<M, true>), dim3(nb), dim3(kBlock), 0, c->stream, c->d_data,
                         c->stride, c->n, chunk, c->d_mask, c->d_par, c->mc, c->d_partials);
    else
      hipLaunchKernelGGL((k_stats<M, false>), dim3(nb), dim3(kBlock), 0, c->stream, c->d_data,
                         c->stride, c->n, chunk, c->d_mask, c->d_par, c->mc, c->d_partials);
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  });
  if (st != LSQR_OK) return st;
  std::vector<double> part((size_t)nb * 8);
  HIPCHK(c, hipMemcpyAsync(part.data(), c->d_partials, part.size() * sizeof(double),
                           hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  double mn = INFINITY, mx = -INFINITY, sum = 0, sq = 0, cnt = 0;
  for (int b = 0; b < nb; b++) {  // fixed order
    mn = std::min(mn, part[b * 8 + 0]);
    mx = std::max(mx, part[b * 8 + 1]);
    sum += part[b * 8 + 2];
    sq += part[b * 8 + 3];
    cnt += part[b * 8 + 4];
  }
  if (cnt == 0) return LSQR_EMPTY;
  out[0] = mn;
  out[1] = mx;
  out[2] = sum / cnt;
  out[3] = sq;
  return LSQR_OK;
}

int lsqr_residuals(lsqr_ctx *c, const double *params, size_t begin, size_t end, double *out) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!params || !out || begin > end || end > c->n) return fail(c, LSQR_ERR_INVALID, "bad argument");
  if (begin == end) return LSQR_OK;
  // staged through d_rows (the phantom's row matrix is rebuilt on the next fit)
  c->rows_valid = false;
  if ((st = ensure(c, &c->d_rows, &c->rows_cap, end - begin)) != LSQR_OK) return st;
  HIPCHK(c, hipMemsetAsync(c->d_par, 0, sizeof(double) * 128, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_par, params, sizeof(double) * c->P, hipMemcpyHostToDevice, c->stream));
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    int grid = grid_for(end - begin, kBlock, 256 * 8);
    hipLaunchKernelGGL((k_residuals<M>), dim3(grid), dim3(kBlock), 0, c->stream, c->d_data, c->stride,
                       begin, end, c->d_par, c->mc, c->d_rows);
    HIPCHK(c, hipGetLastError());
    return LSQR_OK;
  });
  if (st != LSQR_OK) return st;
  HIPCHK(c, hipMemcpyAsync(out, c->d_rows, sizeof(double) * (end - begin), hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, sync_stream(c));
  return LSQR_OK;
}

// ---- RANSAC<T,S>::compute ------------------------------------------------------------------------------
// RANSAC.hxx:129-139 for the winner stashed in d_best (its full scan-parameter row): consensus mask + the
// moment block of its least squares fit in one pass, the fit, one host synchronisation (LM fits: one more per
// evaluation).  The mask count must reproduce the scan's vote count.
static int finish_ransac(lsqr_ctx *c, bool has_best, uint32_t best_votes, double *params_out,
                         uint8_t *consensus_out, lsqr_ransac_info *info) {
  info->best_votes = best_votes;
  info->fraction = (double)best_votes / (double)c->n;
  info->n_params = 0;
  if (!has_best || best_votes == 0) return LSQR_EMPTY;  // RANSAC.hxx:129: nothing written
  int st;
  HIPCHK(c, hipMemsetAsync(c->d_par, 0, sizeof(double) * 128, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_par, c->d_best, sizeof(double) * c->HS, hipMemcpyDeviceToDevice, c->stream));
  bool fused = false;
  int nm = 0;
  if ((st = set_fit_origin(c, true)) != LSQR_OK) return st;
  if ((st = launch_mask_moments(c, 0, c->n, &nm, &fused)) != LSQR_OK) return st;
  if (!fused && (st = launch_mask(c, 0, c->n)) != LSQR_OK) return st;
  unsigned long long *pin2 = (unsigned long long *)((char *)c->h_pin + 8192);
  HIPCHK(c, hipMemcpyAsync(pin2, c->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  if (consensus_out)
    HIPCHK(c, hipMemcpyAsync(consensus_out, c->d_mask, c->n, hipMemcpyDeviceToHost, c->stream));
  SolveOut out;
  memset(&out, 0, sizeof out);
  if ((st = run_fit(c, 1, &out, fused)) != LSQR_OK) return st;  // synchronises the stream
  const unsigned long long cnt = pin2[0];
  if (cnt != best_votes)
    return fail(c, LSQR_ERR_HIP, "consensus mask count %llu != scan votes %u", cnt, best_votes);
  info->fit.n_params = out.ok ? out.n_params : 0;
  info->fit.lm_info = out.lm_info;
  info->fit.lm_nfev = out.lm_nfev;
  info->fit.reserved = out.pad;
  info->fit.cost = out.cost;
  info->fit.n_used = cnt;
  if (!out.ok) return LSQR_EMPTY;
  info->n_params = out.n_params;
  if (params_out)
    for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
  return LSQR_OK;
}

int lsqr_ransac(lsqr_ctx *c, double p, uint64_t seed, const uint32_t *subsets, size_t n_subsets,
                double *params_out, uint8_t *consensus_out, lsqr_ransac_info *info) {
  lsqr_ransac_info local;
  if (!info) info = &local;
  memset(info, 0, sizeof *info);
  int st = need_ready(c, false);
  if (st != LSQR_OK) return st;
  const int k = c->K;
  // RANSAC.hxx:16-19: return 0, parameters untouched
  if (c->n < (size_t)k || !(p < 1.0) || !(p > 0.0)) return LSQR_ERR_INVALID;
  if ((st = need_ready(c, true)) != LSQR_OK) return st;
  uint64_t rs[6];
  lsqr_replay_init(c->n, k, p, rs);
  DedupSet dedup;
  uint64_t base = 0;
  size_t batch = 256;
  while (!rs[RS_DONE]) {
    size_t H = batch;
    if (subsets) {
      if (base >= n_subsets) break;  // caller's stream exhausted
      H = std::min<size_t>(H, n_subsets - base);
    }
    uint64_t remaining = rs[RS_TRIES] - base;
    if (remaining < H) H = (size_t)remaining;
    if (H == 0) break;
    // batch results land in pinned memory with one synchronisation: {votes[H], subsets[H*k], valid[H]}
    const size_t need = H * sizeof(uint32_t) * (1 + (size_t)k) + H;
    if (need > c->batch_pin_cap) {
      if (c->h_batch) (void)hipHostFree(c->h_batch);
      c->h_batch = nullptr;
      c->batch_pin_cap = 0;
      HIPCHK(c, hipHostMalloc(&c->h_batch, std::max<size_t>(need, 1 << 16)));
      c->batch_pin_cap = std::max<size_t>(need, 1 << 16);
    }
    uint32_t *p_votes = (uint32_t *)c->h_batch, *p_sub = p_votes + H;
    uint8_t *p_valid = (uint8_t *)(p_sub + H * k);
    if (subsets) {
      memcpy(p_sub, subsets + base * k, H * k * sizeof(uint32_t));
      st = lsqr_hypotheses_from_subsets(c, p_sub, H);
    } else {
      st = lsqr_hypotheses_sample(c, seed, base, H, nullptr);
    }
    if (st != LSQR_OK) return st;
    // what the adaptive bound still asks for (saturated: a poor start leaves it at C(N,k)); the index build
    // heuristic weighs it against the cost of the build
    c->hyp_expected = has_any_best(rs) ? std::min<uint64_t>(rs[RS_TRIES] - base, 1u << 20) : 0;
    if ((st = run_scan_batch(c, has_any_best(rs) ? (uint32_t)rs[RS_BEST] : 0u)) != LSQR_OK) return st;
    c->scanned = true;
    if (!subsets)
      HIPCHK(c, hipMemcpyAsync(p_sub, c->d_subsets, H * k * sizeof(uint32_t), hipMemcpyDeviceToHost,
                               c->stream));
    if ((st = lsqr_get_hypotheses(c, nullptr, p_valid, p_votes)) != LSQR_OK) return st;  // synchronises
    uint64_t prev_best_idx = rs[RS_BEST_IDX];
    bool had = rs[RS_HAS] != 0;
    size_t used = lsqr_replay(c->n, k, p, p_sub, p_valid, p_votes, H, base, &dedup, rs);
    info->evaluated += H;
    if (rs[RS_HAS] && (!had || rs[RS_BEST_IDX] != prev_best_idx)) {
      size_t e = (size_t)(rs[RS_BEST_IDX] - base);
      // the winner's row stays on the device (the next batch overwrites d_hparams): no host round trip
      HIPCHK(c, hipMemcpyAsync(c->d_best, c->d_hparams + e * c->HS, sizeof(double) * c->HS,
                               hipMemcpyDeviceToDevice, c->stream));
    }
    base += used;
    if (used < H) break;
    batch = std::min<size_t>(batch * 4, 4096);
    // Safety stop (deviation from the reference, which would keep drawing up to C(N,k) subsets):
    // if 2^22 consecutive iterations produced no model at all the data are degenerate.
    if (!rs[RS_HAS] && base >= (1ull << 22)) break;
    if (c->opt_max_iter > 0 && base >= (uint64_t)c->opt_max_iter) break;  // caller's budget
  }
  c->hyp_expected = 0;
  info->iterations = rs[RS_I];
  info->best_index = rs[RS_BEST_IDX];
  return finish_ransac(c, rs[RS_HAS] != 0, (uint32_t)rs[RS_BEST], params_out,
                       consensus_out, info);
}

int lsqr_ransac_exhaustive(lsqr_ctx *c, double *params_out, uint8_t *consensus_out,
                           lsqr_ransac_info *info) {
  lsqr_ransac_info local;
  if (!info) info = &local;
  memset(info, 0, sizeof *info);
  int st = need_ready(c, false);
  if (st != LSQR_OK) return st;
  const int k = c->K;
  if (c->n < (size_t)k) return LSQR_EMPTY;  // RANSAC.hxx:165-169: cleared, returns 0
  if ((st = need_ready(c, true)) != LSQR_OK) return st;
  // all C(N,k) tuples in lexicographic order (RANSAC.hxx:197-213), in batches
  std::vector<uint32_t> comb((size_t)k), sub, votes;
  std::vector<uint8_t> valid;
  for (int l = 0; l < k; l++) comb[l] = (uint32_t)l;
  bool more = true, has = false;
  uint32_t best = 0;
  uint64_t index = 0, best_idx = 0;
  const size_t B = 4096;
  while (more) {
    sub.clear();
    size_t H = 0;
    while (more && H < B) {
      sub.insert(sub.end(), comb.begin(), comb.end());
      H++;
      int l = k - 1;
      while (l >= 0 && comb[l] == (uint32_t)(c->n - k + l)) l--;
      if (l < 0) more = false;
      else {
        comb[l]++;
        for (int j = l + 1; j < k; j++) comb[j] = comb[j - 1] + 1;
      }
    }
    if ((st = lsqr_hypotheses_from_subsets(c, sub.data(), H)) != LSQR_OK) return st;
    if ((st = lsqr_scan(c)) != LSQR_OK) return st;
    votes.resize(H);
    valid.resize(H);
    if ((st = lsqr_get_hypotheses(c, nullptr, valid.data(), votes.data())) != LSQR_OK) return st;
    long winner = -1;
    for (size_t e = 0; e < H; e++)
      if (valid[e] && votes[e] > best) {  // :245 strict
        best = votes[e];
        winner = (long)e;
        best_idx = index + e;
        has = true;
      }
    if (winner >= 0) {
      HIPCHK(c, hipMemcpyAsync(c->d_best, c->d_hparams + (size_t)winner * c->HS, sizeof(double) * c->HS,
                               hipMemcpyDeviceToDevice, c->stream));
    }
    index += H;
    info->evaluated += H;
  }
  info->iterations = index;
  info->best_index = best_idx;
  return finish_ransac(c, has, best, params_out, consensus_out, info);
}

// One fixed-size batch end to end, everything chained on the stream: sample -> solve -> scan ->
// first-max winner -> consensus mask -> final fit; the host synchronises once (LM fits: once per
// evaluation).
int lsqr_batch_fit(lsqr_ctx *c, uint64_t seed, uint64_t first, size_t H, double *params_out,
                   uint8_t *consensus_out, lsqr_ransac_info *info) {
  int st = lsqr_hypotheses_sample(c, seed, first, H, nullptr);
  if (st != LSQR_OK) return st;
  if ((st = run_scan_batch(c, 0)) != LSQR_OK) return st;
  c->scanned = true;
  hipLaunchKernelGGL(k_best, dim3(1), dim3(kBlock), 0, c->stream, c->d_votes, c->d_valid,
                     (uint32_t)c->H, c->d_counter + 1);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_take_best, dim3(1), dim3(64), 0, c->stream, c->d_counter + 1, c->d_hparams,
                     c->HS, c->d_par);
  HIPCHK(c, hipGetLastError());
  bool fused = false;
  int nm = 0;
  if ((st = set_fit_origin(c, true)) != LSQR_OK) return st;
  if ((st = launch_mask_moments(c, 0, c->n, &nm, &fused)) != LSQR_OK) return st;
  if (!fused && (st = launch_mask(c, 0, c->n)) != LSQR_OK) return st;
  unsigned long long *pin2 = (unsigned long long *)((char *)c->h_pin + 8192);
  HIPCHK(c, hipMemcpyAsync(pin2, c->d_counter, 2 * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, c->stream));  // {inliers, packed winner}
  if (consensus_out)
    HIPCHK(c, hipMemcpyAsync(consensus_out, c->d_mask, c->n, hipMemcpyDeviceToHost, c->stream));
  SolveOut out;
  memset(&out, 0, sizeof out);
  if ((st = run_fit(c, 1, &out, fused)) != LSQR_OK) return st;  // synchronises the stream
  const unsigned long long cnt = pin2[0], pk = pin2[1];
  if (info) {
    memset(info, 0, sizeof *info);
    info->fraction = c->n ? (double)cnt / (double)c->n : 0.0;
    info->iterations = H;
    info->evaluated = H;
    info->best_votes = (uint32_t)(pk >> 32);
    info->best_index = pk ? first + (0xFFFFFFFFull - (pk & 0xFFFFFFFFull)) : 0;
    info->n_params = out.ok ? out.n_params : 0;
    info->fit.n_params = info->n_params;
    info->fit.lm_info = out.lm_info;
    info->fit.lm_nfev = out.lm_nfev;
    info->fit.reserved = out.pad;
    info->fit.n_used = cnt;
    info->fit.cost = out.cost;
  }
  if (pk == 0 || !out.ok) return LSQR_EMPTY;
  if (params_out)
    for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
  return LSQR_OK;
}

// Second half of a multi-GPU step on one rank, chained on the stream with one synchronisation: the
// winner is re-derived from its index in the (stateless) sampler stream, its consensus mask is taken
// over this rank's observation slice [begin, end) and the slice's phase-0 moment block is reduced about
// the model's own point (plane / line: a, sphere: c; zeros for the other models).
int lsqr_winner_moments(lsqr_ctx *c, uint64_t seed, uint64_t stream_index, size_t begin, size_t end,
                        double *params_out, double *origin_out, double *block_out,
                        uint64_t *count_out) {
  int st = lsqr_hypotheses_sample(c, seed, stream_index, 1, nullptr);
  if (st != LSQR_OK) return st;
  if (begin > end || end > c->n || !block_out) return fail(c, LSQR_ERR_INVALID, "bad range");
  HIPCHK(c, hipMemsetAsync(c->d_par, 0, sizeof(double) * 128, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_par, c->d_hparams, sizeof(double) * c->HS, hipMemcpyDeviceToDevice,
                           c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_vec, 0, sizeof(double) * 32, c->stream));
  const int m = c->cfg.model;
  if (m == LSQR_MODEL_PLANE || m == LSQR_MODEL_LINE || m == LSQR_MODEL_SPHERE)
    HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_par + (m == LSQR_MODEL_SPHERE ? 0 : c->ND),
                             sizeof(double) * c->ND, hipMemcpyDeviceToDevice, c->stream));
  int nmom = 0;
  bool fused = false;
  if ((st = launch_mask_moments(c, begin, end, &nmom, &fused)) != LSQR_OK) return st;
  if (!fused) {
    if ((st = launch_mask(c, begin, end)) != LSQR_OK) return st;
    st = dispatch(c->cfg, [&](auto tag) -> int {
      typedef typename decltype(tag)::type M;
      return launch_moments<M>(c, 1, begin, end, 0, &nmom);
    });
    if (st != LSQR_OK) return st;
  }
  double *pin = (double *)((char *)c->h_pin + 16384);  // {valid, count, params[64], origin[32]}
  HIPCHK(c, hipMemcpyAsync(pin, c->d_valid, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(pin + 1, c->d_counter, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, hipMemcpyAsync(pin + 2, c->d_hparams, sizeof(double) * c->P, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, hipMemcpyAsync(pin + 66, c->d_vec, sizeof(double) * 32, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, hipMemcpyAsync(block_out, c->d_mom, sizeof(double) * nmom, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, sync_stream(c));
  if (*(const uint8_t *)pin == 0) return LSQR_EMPTY;  // degenerate subset: not a possible winner
  if (count_out) memcpy(count_out, pin + 1, sizeof(uint64_t));
  if (params_out) memcpy(params_out, pin + 2, sizeof(double) * c->P);
  if (origin_out) memcpy(origin_out, pin + 66, sizeof(double) * 32);
  return LSQR_OK;
}

// ---- pipelined batches: enqueue now, read later ------------------------------------------------------------
static char *slot_pin(lsqr_ctx *c, int slot) { return (char *)c->h_pin + 49152 + slot * 2048; }

// slot s of a context with L lanes: lane s % L, that lane's own slot s / L (0 or 1)
static int lane_count(const lsqr_ctx *c) {
  return c->is_lane ? 1 : std::max(1, std::min(c->opt_lanes, (int)lsqr_ctx::kMaxLanes));
}
// the lane context, created on first use and (re-)attached to the parent's model, options and records
static int lane_get(lsqr_ctx *c, int li, lsqr_ctx **out) {
  if (li == 0) {
    *out = c;
    return LSQR_OK;
  }
  lsqr_ctx *&l = c->lanes[li];
  if (!l) {
    int st = lsqr_ctx_create(c->device, &l);
    if (st != LSQR_OK) return fail(c, st, "cannot create lane %d", li);
    l->is_lane = true;
    if (l->counted) {  // lanes never run a persistent fit: they do not take a share of the device
      lmp_ctx_count(l->device, -1);
      l->counted = false;
    }
    l->prof = c->prof;
  }
  if (l->lane_epoch != c->data_epoch) {
    if (!c->has_model || !c->d_data) return fail(c, LSQR_ERR_STATE, "no model / records");
    HIPCHK(c, hipStreamSynchronize(l->stream));
    int st = lsqr_set_model(l, &c->cfg);
    for (size_t k = 0; st == LSQR_OK && k < c->opt_log.size(); k++)
      st = lsqr_set_option(l, c->opt_log[k].first.c_str(), c->opt_log[k].second);
    if (st == LSQR_OK) st = lsqr_attach(l, c->d_data, c->n, c->stride * sizeof(double));
    if (st != LSQR_OK) return fail(c, st, "lane %d: %s", li, lsqr_last_error(l));
    HIPCHK(c, sync_stream(c));  // an upload still in flight on the parent's stream
    l->lane_epoch = c->data_epoch;
  }
  *out = l;
  return LSQR_OK;
}

int lsqr_batch_fit_enqueue(lsqr_ctx *c, uint64_t seed, uint64_t first, size_t H, int slot) {
  if (!c) return LSQR_ERR_INVALID;
  if (!c->is_lane) {
    const int L = lane_count(c);
    if (slot < 0 || slot >= 2 * L) return fail(c, LSQR_ERR_INVALID, "slot %d of %d", slot, 2 * L);
    if (slot % L != 0 || slot / L != slot) {
      lsqr_ctx *l = nullptr;
      int st = lane_get(c, slot % L, &l);
      if (st != LSQR_OK) return st;
      if (l != c) {
        st = lsqr_batch_fit_enqueue(l, seed, first, H, slot / L);
        return st == LSQR_OK ? st : fail(c, st, "lane %d: %s", slot % L, lsqr_last_error(l));
      }
      slot = slot / L;
    }
  }
  if (slot < 0 || slot > 1) return LSQR_ERR_INVALID;
  if (c->slot_busy[slot]) return fail(c, LSQR_ERR_STATE, "slot %d holds an unread result", slot);
  if (c->has_model && (wants_lm(c->cfg) || c->cfg.model == LSQR_MODEL_PHANTOM))
    return fail(c, LSQR_ERR_INVALID, "this model's fit needs the host between device passes");
  int st = lsqr_hypotheses_sample(c, seed, first, H, nullptr);
  if (st != LSQR_OK) return st;
  const uint64_t seq_before = c->bsel_seq;
  c->defer_ovf = true;  // no host synchronisation inside the scan: the worklist fill is checked by lsqr_batch_fit_wait
  c->ovf_cap = 0;
  st = run_scan_batch(c, 0);
  c->defer_ovf = false;
  if (st != LSQR_OK) return st;
  c->slot_ovf_cap[slot] = c->ovf_cap;
  c->slot_seed[slot] = seed;
  if (c->ovf_cap)  // {fullest worklist segment of this batch's scan} beside the slot's results
    HIPCHK(c, hipMemcpyAsync(slot_pin(c, slot) + 32, c->d_counter + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                             c->stream));
  c->slot_bsel_rec[slot] = c->bsel_seq != seq_before ? c->bsel_last_rec : -1;  // this batch's selection record
  c->slot_bsel_seq[slot] = c->bsel_seq;
  c->scanned = true;
  hipLaunchKernelGGL(k_best, dim3(1), dim3(kBlock), 0, c->stream, c->d_votes, c->d_valid,
                     (uint32_t)c->H, c->d_counter + 1, 0u);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_take_best, dim3(1), dim3(64), 0, c->stream, c->d_counter + 1, c->d_hparams,
                     c->HS, c->d_par);
  HIPCHK(c, hipGetLastError());
  bool fused = false;
  int nm = 0;
  if ((st = set_fit_origin(c, true)) != LSQR_OK) return st;
  if ((st = launch_mask_moments(c, 0, c->n, &nm, &fused)) != LSQR_OK) return st;
  if (!fused && (st = launch_mask(c, 0, c->n)) != LSQR_OK) return st;
  char *pin = slot_pin(c, slot);
  HIPCHK(c, hipMemcpyAsync(pin, c->d_counter, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           c->stream));  // {inliers, packed winner}
  if ((st = enqueue_fit(c, 1, fused)) != LSQR_OK) return st;
  HIPCHK(c, hipMemcpyAsync(pin + 64, c->d_out, sizeof(SolveOut), hipMemcpyDeviceToHost, c->stream));
  if (!c->slot_ev[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->slot_ev[slot], hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->slot_ev[slot], c->stream));
  c->slot_first[slot] = first;
  c->slot_H[slot] = H;
  c->slot_busy[slot] = true;
  return LSQR_OK;
}

int lsqr_batch_fit_wait(lsqr_ctx *c, int slot, double *params_out, lsqr_ransac_info *info) {
  if (!c) return LSQR_ERR_INVALID;
  if (!c->is_lane) {
    const int L = lane_count(c);
    if (slot < 0 || slot >= 2 * L) return fail(c, LSQR_ERR_INVALID, "slot %d of %d", slot, 2 * L);
    const int li = slot % L;
    slot = slot / L;
    if (li != 0) {
      lsqr_ctx *l = c->lanes[li];
      if (!l) return fail(c, LSQR_ERR_STATE, "slot has nothing in flight");
      int st = lsqr_batch_fit_wait(l, slot, params_out, info);
      return (st == LSQR_OK || st == LSQR_EMPTY) ? st : fail(c, st, "lane %d: %s", li, lsqr_last_error(l));
    }
  }
  if (slot < 0 || slot > 1) return LSQR_ERR_INVALID;
  if (!c->slot_busy[slot]) return fail(c, LSQR_ERR_STATE, "slot %d has nothing in flight", slot);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(c->slot_ev[slot]));
  c->slot_busy[slot] = false;
  if (c->slot_bsel_rec[slot] >= 0 && c->bsel_seq_of[c->slot_bsel_rec[slot]] == c->slot_bsel_seq[slot])
    bsel_fold(c, c->slot_bsel_rec[slot]);  // the host has synchronised with this batch: its selection counts count
  c->slot_bsel_rec[slot] = -1;
  const char *pin = slot_pin(c, slot);
  if (c->slot_ovf_cap[slot]) {
    unsigned int fill = 0;
    memcpy(&fill, pin + 32, sizeof fill);
    if (fill > c->slot_ovf_cap[slot] || c->opt_test_overflow) {
      // a worklist segment of the matrix-core filter overflowed (never seen outside the tests): the batch again, blocking,
      // on the exact kernels
      (void)fail(c, LSQR_OK, "worklist segment overflow in slot %d (fill %u > %u): batch run again without the filter",
                 slot, fill, c->slot_ovf_cap[slot]);
      c->ovf_reruns++;
      const int keep = c->opt_filter;
      c->opt_filter = 0;
      const int st2 = lsqr_batch_fit(c, c->slot_seed[slot], c->slot_first[slot], (size_t)c->slot_H[slot], params_out,
                                     nullptr, info);
      c->opt_filter = keep;
      return st2;
    }
  }
  unsigned long long head[2];
  SolveOut out;
  memcpy(head, pin, sizeof head);
  memcpy(&out, pin + 64, sizeof out);
  const unsigned long long cnt = head[0], pk = head[1];
  if (info) {
    memset(info, 0, sizeof *info);
    info->fraction = c->n ? (double)cnt / (double)c->n : 0.0;
    info->iterations = c->slot_H[slot];
    info->evaluated = c->slot_H[slot];
    info->best_votes = (uint32_t)(pk >> 32);
    info->best_index = pk ? c->slot_first[slot] + (0xFFFFFFFFull - (pk & 0xFFFFFFFFull)) : 0;
    info->n_params = out.ok ? out.n_params : 0;
    info->fit.n_params = info->n_params;
    info->fit.n_used = cnt;
    info->fit.cost = out.cost;
  }
  if (pk == 0 || !out.ok) return LSQR_EMPTY;
  if (params_out)
    for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
  return LSQR_OK;
}

// ---- multi-GPU step with device-resident exchange buffers (lsqr_hip.h) ----------------------------------
int lsqr_set_stream(lsqr_ctx *c, void *hip_stream, int external) {
  if (!c) return LSQR_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, sync_stream(c));  // nothing of ours may still be in flight on the old stream
  // profiling events recorded on the old stream stay valid (events are not bound to a stream)
  // external: the caller's stream as given -- NULL is the (legacy) default stream, which torch uses unless
  // told otherwise
  c->stream = external ? (hipStream_t)hip_stream : c->own_stream;
  c->external_stream = external != 0;
  return LSQR_OK;
}

int lsqr_step_scan(lsqr_ctx *c, uint64_t seed, uint64_t first, size_t H, uint32_t index_base,
                   uint64_t *packed_dev) {
  if (!packed_dev) return c ? fail(c, LSQR_ERR_INVALID, "null exchange buffer") : LSQR_ERR_INVALID;
  int st = lsqr_hypotheses_sample(c, seed, first, H, nullptr);
  if (st != LSQR_OK) return st;
  if ((st = run_scan_batch(c, 0)) != LSQR_OK) return st;
  c->scanned = true;
  hipLaunchKernelGGL(k_best, dim3(1), dim3(kBlock), 0, c->stream, c->d_votes, c->d_valid,
                     (uint32_t)c->H, (unsigned long long *)packed_dev, index_base);
  HIPCHK(c, hipGetLastError());
  return LSQR_OK;
}

int lsqr_step_winner(lsqr_ctx *c, uint64_t seed, uint64_t batch_first, const uint64_t *packed_dev,
                     size_t begin, size_t end, double *block_dev) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!packed_dev || !block_dev || begin > end || end > c->n) return fail(c, LSQR_ERR_INVALID, "bad argument");
  if ((st = ensure_hyp(c, 1)) != LSQR_OK) return st;
  c->H = 1;
  c->scanned = false;
  {
    ProfScope ps(c, KID_SAMPLE);
    hipLaunchKernelGGL(k_winner_subset, dim3(1), dim3(64), 0, c->stream, seed, batch_first,
                       (const unsigned long long *)packed_dev, (uint64_t)c->n, c->K, c->d_subsets);
    HIPCHK(c, hipGetLastError());
  }
  if ((st = run_estimate(c)) != LSQR_OK) return st;
  HIPCHK(c, hipMemsetAsync(c->d_par, 0, sizeof(double) * 128, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_par, c->d_hparams, sizeof(double) * c->HS, hipMemcpyDeviceToDevice,
                           c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_vec, 0, sizeof(double) * 32, c->stream));
  const int m = c->cfg.model;
  if (m == LSQR_MODEL_PLANE || m == LSQR_MODEL_LINE || m == LSQR_MODEL_SPHERE)
    HIPCHK(c, hipMemcpyAsync(c->d_vec, c->d_par + (m == LSQR_MODEL_SPHERE ? 0 : c->ND),
                             sizeof(double) * c->ND, hipMemcpyDeviceToDevice, c->stream));
  int nmom = 0;
  bool fused = false;
  if ((st = launch_mask_moments(c, begin, end, &nmom, &fused)) != LSQR_OK) return st;
  if (!fused) {
    if ((st = launch_mask(c, begin, end)) != LSQR_OK) return st;
    st = dispatch(c->cfg, [&](auto tag) -> int {
      typedef typename decltype(tag)::type M;
      return launch_moments<M>(c, 1, begin, end, 0, &nmom);
    });
    if (st != LSQR_OK) return st;
  }
  hipLaunchKernelGGL(k_pack_block, dim3(1), dim3(256), 0, c->stream, c->d_mom, nmom, c->d_counter,
                     block_dev);
  HIPCHK(c, hipGetLastError());
  return LSQR_OK;
}

// results of a step are staged per slot: {packed, count, valid, winner params[64], SolveOut, phantom block}
static char *step_pin(lsqr_ctx *c, int slot) { return (char *)c->h_pin + 53248 + slot * 2048; }

int lsqr_step_finish_enqueue(lsqr_ctx *c, const uint64_t *packed_dev, const double *block_dev, int slot) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!packed_dev || !block_dev || slot < 0 || slot > 1) return fail(c, LSQR_ERR_INVALID, "bad argument");
  if (c->step_busy[slot]) return fail(c, LSQR_ERR_STATE, "step slot %d holds an unread result", slot);
  const int nmom = lsqr_moments_len(&c->cfg, 0);
  char *pin = step_pin(c, slot);
  HIPCHK(c, hipMemcpyAsync(pin, packed_dev, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(pin + 8, block_dev + nmom, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(pin + 16, c->d_valid, 1, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(pin + 64, c->d_hparams, sizeof(double) * c->P, hipMemcpyDeviceToHost, c->stream));
  if (c->cfg.model == LSQR_MODEL_PHANTOM) {  // solved on the host from the Gram block (in _wait)
    double *blk = (double *)((char *)c->h_pin + 40960);  // one block area: phantom steps are not pipelined
    HIPCHK(c, hipMemcpyAsync(blk, block_dev, sizeof(double) * nmom, hipMemcpyDeviceToHost, c->stream));
  } else {
    HIPCHK(c, hipMemcpyAsync(c->d_mom, block_dev, sizeof(double) * nmom, hipMemcpyDeviceToDevice,
                             c->stream));
    st = dispatch(c->cfg, [&](auto tag) -> int {
      typedef typename decltype(tag)::type M;
      if constexpr (M::IS_DENSE) {
        return launch_solve_dense(c);
      } else {
        ProfScope ps(c, KID_SOLVE);
        hipLaunchKernelGGL((k_solve<M>), dim3(1), dim3(64), 0, c->stream, c->d_mom, c->d_vec, c->mc,
                           c->d_out);
        HIPCHK(c, hipGetLastError());
        return LSQR_OK;
      }
    });
    if (st != LSQR_OK) return st;
    HIPCHK(c, hipMemcpyAsync(pin + 1024, c->d_out, sizeof(SolveOut), hipMemcpyDeviceToHost, c->stream));
  }
  if (!c->step_ev[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->step_ev[slot], hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->step_ev[slot], c->stream));
  c->step_bsel_rec[slot] = c->bsel_last_rec;   // the step's scan (lsqr_step_scan) came before on this stream
  c->step_bsel_seq[slot] = c->bsel_last_rec >= 0 ? c->bsel_seq_of[c->bsel_last_rec] : 0;
  c->step_busy[slot] = true;
  return LSQR_OK;
}

int lsqr_step_finish_wait(lsqr_ctx *c, int slot, double *winner_out, double *params_out,
                          lsqr_ransac_info *info) {
  if (!c || slot < 0 || slot > 1) return LSQR_ERR_INVALID;
  if (!c->step_busy[slot]) return fail(c, LSQR_ERR_STATE, "step slot %d has nothing in flight", slot);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(c->step_ev[slot]));  // the one synchronisation of the step
  c->step_busy[slot] = false;
  if (c->step_bsel_rec[slot] >= 0 && c->bsel_seq_of[c->step_bsel_rec[slot]] == c->step_bsel_seq[slot])
    bsel_fold(c, c->step_bsel_rec[slot]);
  c->step_bsel_rec[slot] = -1;
  const char *pin = step_pin(c, slot);
  unsigned long long pk;
  double count;
  memcpy(&pk, pin, 8);
  memcpy(&count, pin + 8, 8);
  const uint8_t valid = *(const uint8_t *)(pin + 16);
  SolveOut out;
  memset(&out, 0, sizeof out);
  if (c->cfg.model == LSQR_MODEL_PHANTOM)
    phantom_solve_block(c->cfg, (const double *)((const char *)c->h_pin + 40960), &out);
  else
    memcpy(&out, pin + 1024, sizeof out);
  if (info) {
    memset(info, 0, sizeof *info);
    info->evaluated = pk != 0;  // a winner exists
    info->best_votes = (uint32_t)(pk >> 32);
    info->best_index = pk ? 0xFFFFFFFFull - (pk & 0xFFFFFFFFull) : 0;
    info->n_params = out.ok ? out.n_params : 0;
    info->fit.n_params = info->n_params;
    info->fit.lm_info = out.lm_info;
    info->fit.lm_nfev = out.lm_nfev;
    info->fit.reserved = out.pad;
    info->fit.cost = out.cost;
    info->fit.n_used = (uint64_t)(count + 0.5);
    info->fraction = c->n ? count / (double)c->n : 0.0;
  }
  if (pk == 0 || !valid) return LSQR_EMPTY;
  if (winner_out) memcpy(winner_out, pin + 64, sizeof(double) * c->P);
  if (!out.ok) return LSQR_EMPTY;
  if (params_out)
    for (int j = 0; j < out.n_params; j++) params_out[j] = out.params[j];
  return LSQR_OK;
}

int lsqr_step_finish(lsqr_ctx *c, const uint64_t *packed_dev, const double *block_dev,
                     double *winner_out, double *params_out, lsqr_ransac_info *info) {
  if (c && c->step_busy[0]) return fail(c, LSQR_ERR_STATE, "step slot 0 holds an unread result");
  int st = lsqr_step_finish_enqueue(c, packed_dev, block_dev, 0);
  if (st != LSQR_OK) return st;
  return lsqr_step_finish_wait(c, 0, winner_out, params_out, info);
}

// ---- several devices from ONE process (lsqr_hip.h: lsqr_multi_*) ------------------------------------------------
// The hypothesis stream is sharded over n contexts, one per listed device, exactly as bench.py's ranks shard it
// (lsqr_step_scan / _winner / _finish per context); the two exchanges of a step -- 8 B winner, <= 17 KB moment
// block -- are peer copies into rank 0's gather area (hipMemcpyPeerAsync: xGMI between GPUs of a node), a
// fixed-order reduction kernel there, and a peer copy of the winner back.  Streams are chained with events; the
// host synchronises once per step.  (Between PROCESSES the same exchanges are RCCL all-reduces: distributed.py.)
}  // extern "C"

namespace {
constexpr int kMultiBlk = 2304;  // >= dense_ne(64) + 1 + count, in doubles

__global__ void k_multi_max(const unsigned long long *__restrict__ g, int n, unsigned long long *__restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long b = 0;
  for (int r = 0; r < n; r++) b = g[r] > b ? g[r] : b;
  *out = b;
}
__global__ void k_multi_sum(const double *__restrict__ g, int n, int len, int pitch, double *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= len) return;
  double t = 0.0;
  for (int r = 0; r < n; r++) t += g[(size_t)r * pitch + k];  // rank order: deterministic
  out[k] = t;
}
}  // namespace

struct lsqr_multi {
  int n = 0;
  std::vector<lsqr_ctx *> ctx;
  std::vector<unsigned long long *> packed;  // per rank: the step's packed winner (device)
  std::vector<double *> block;               // per rank: moment block + count (device)
  unsigned long long *g_packed = nullptr;    // rank 0: gather areas
  double *g_block = nullptr;
  std::vector<hipEvent_t> ev;                // per rank: "my contribution has been sent"
  hipEvent_t ev_root = nullptr;              // rank 0: "the reduced value is on its way back"
  // LSQR_MULTI_TRANSPORT=rccl: the two exchanges as RCCL all-reduces over xGMI (one communicator per device,
  // ncclCommInitAll) instead of peer copies into rank 0 + a reduction kernel
  bool rccl = false;
  std::vector<ncclComm_t> comm;
  double rccl_bringup_s = 0.0;
  char err[512] = {0};
};

namespace {
int mfail(lsqr_multi *m, int st, const char *what, lsqr_ctx *c = nullptr) {
  if (m) snprintf(m->err, sizeof m->err, "%s%s%s", what, c ? ": " : "", c ? c->err : "");
  return st;
}
#define MHIP(m, call)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      snprintf((m)->err, sizeof(m)->err, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
               __LINE__);                                                                                 \
      return LSQR_ERR_HIP;                                                                                \
    }                                                                                                     \
  } while (0)

// librccl, loaded on first use and only when asked for (a process that also imports PyTorch carries PyTorch's own copy
// of the library: RTLD_LOCAL keeps the two apart)
struct RcclApi {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  bool ok = false;
};
RcclApi &rccl_api() {
  static RcclApi a;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {getenv("LSQR_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names) {
      if (!nm || !*nm) continue;
      a.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (a.lib) break;
    }
    if (!a.lib) return;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.lib, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(a.lib, "ncclAllReduce");
    a.GroupStart = (decltype(a.GroupStart))dlsym(a.lib, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.lib, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
    a.GetVersion = (decltype(a.GetVersion))dlsym(a.lib, "ncclGetVersion");
    a.ok = a.CommInitAll && a.CommDestroy && a.AllReduce && a.GroupStart && a.GroupEnd && a.GetErrorString;
  });
  return a;
}
#define MNCCL(m, call)                                                                                       \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess) {                                                                                 \
      snprintf((m)->err, sizeof(m)->err, "%s failed: %s (%s:%d)", #call, rccl_api().GetErrorString(r_), __FILE__, \
               __LINE__);                                                                                    \
      return LSQR_ERR_HIP;                                                                                   \
    }                                                                                                        \
  } while (0)

// RCCL transport: buf[r] (count elements on rank r's device) <- op over all ranks, in place, each on its own stream
int multi_allreduce(lsqr_multi *m, const std::vector<void *> &buf, size_t count, ncclDataType_t type, ncclRedOp_t op) {
  RcclApi &a = rccl_api();
  MNCCL(m, a.GroupStart());
  for (int r = 0; r < m->n; r++) {
    ncclResult_t e = a.AllReduce(buf[r], buf[r], count, type, op, m->comm[r], m->ctx[r]->stream);
    if (e != ncclSuccess) {
      (void)a.GroupEnd();
      snprintf(m->err, sizeof m->err, "ncclAllReduce failed on rank %d: %s", r, a.GetErrorString(e));
      return LSQR_ERR_HIP;
    }
  }
  MNCCL(m, a.GroupEnd());
  MHIP(m, hipSetDevice(m->ctx[0]->device));
  return LSQR_OK;
}

// every rank's `src[r]` (bytes) -> rank 0's gather area at r * pitch; rank 0's stream waits for all of them
int multi_gather(lsqr_multi *m, const std::vector<const void *> &src, void *gather, size_t bytes, size_t pitch) {
  lsqr_ctx *c0 = m->ctx[0];
  for (int r = 0; r < m->n; r++) {
    lsqr_ctx *c = m->ctx[r];
    MHIP(m, hipSetDevice(c->device));
    MHIP(m, hipMemcpyPeerAsync((char *)gather + (size_t)r * pitch, c0->device, src[r], c->device, bytes, c->stream));
    if (r) {
      MHIP(m, hipEventRecord(m->ev[r], c->stream));
      MHIP(m, hipSetDevice(c0->device));
      MHIP(m, hipStreamWaitEvent(c0->stream, m->ev[r], 0));
    }
  }
  MHIP(m, hipSetDevice(c0->device));
  return LSQR_OK;
}

// rank 0's `src` (bytes) -> every other rank's dst[r]; their streams wait for it
int multi_bcast(lsqr_multi *m, const void *src, const std::vector<void *> &dst, size_t bytes) {
  lsqr_ctx *c0 = m->ctx[0];
  MHIP(m, hipSetDevice(c0->device));
  for (int r = 1; r < m->n; r++)
    MHIP(m, hipMemcpyPeerAsync(dst[r], m->ctx[r]->device, src, c0->device, bytes, c0->stream));
  MHIP(m, hipEventRecord(m->ev_root, c0->stream));
  for (int r = 1; r < m->n; r++) {
    MHIP(m, hipSetDevice(m->ctx[r]->device));
    MHIP(m, hipStreamWaitEvent(m->ctx[r]->stream, m->ev_root, 0));
  }
  MHIP(m, hipSetDevice(c0->device));
  return LSQR_OK;
}

void slice_of(size_t n, int r, int w, size_t *lo, size_t *hi) {
  *lo = n * (size_t)r / (size_t)w;
  *hi = n * (size_t)(r + 1) / (size_t)w;
}

// winner known on every rank (packed[r] holds it, in-batch index relative to batch_first): masks of the slices,
// summed moment block, final fit on rank 0 (LM fits: one summed block per evaluation), consensus slices to host
int multi_finish(lsqr_multi *m, uint64_t seed, uint64_t batch_first, double *params_out, uint8_t *consensus_out,
                 lsqr_ransac_info *info) {
  const int n = m->n;
  lsqr_ctx *c0 = m->ctx[0];
  const int len = lsqr_moments_len(&c0->cfg, 0);
  if (len + 1 > kMultiBlk) return mfail(m, LSQR_ERR_INVALID, "moment block too large");
  std::vector<size_t> lo(n), hi(n);
  std::vector<const void *> src(n);
  int st;
  for (int r = 0; r < n; r++) {
    slice_of(c0->n, r, n, &lo[r], &hi[r]);
    if ((st = lsqr_step_winner(m->ctx[r], seed, batch_first, (const uint64_t *)m->packed[r], lo[r], hi[r],
                               m->block[r])) != LSQR_OK)
      return mfail(m, st, "lsqr_step_winner", m->ctx[r]);
    src[r] = m->block[r];
  }
  if (m->rccl) {  // one all-reduce SUM of [moment block, inlier count]: every rank ends up with the summed block
    std::vector<void *> buf(n);
    for (int r = 0; r < n; r++) buf[r] = m->block[r];
    if ((st = multi_allreduce(m, buf, (size_t)len + 1, ncclDouble, ncclSum)) != LSQR_OK) return st;
  } else {
  if ((st = multi_gather(m, src, m->g_block, sizeof(double) * (len + 1), sizeof(double) * kMultiBlk)) != LSQR_OK)
    return st;
  hipLaunchKernelGGL(k_multi_sum, dim3((len + 1 + 255) / 256), dim3(256), 0, c0->stream, m->g_block, n, len + 1,
                     kMultiBlk, m->block[0]);
  MHIP(m, hipGetLastError());
  }
  double winner[64], fit[64];
  lsqr_ransac_info local;
  if (!info) info = &local;
  st = lsqr_step_finish(c0, (const uint64_t *)m->packed[0], m->block[0], winner, fit, info);  // synchronises rank 0
  if (st != LSQR_OK && st != LSQR_EMPTY) return mfail(m, st, "lsqr_step_finish", c0);
  info->fraction = c0->n ? (double)info->fit.n_used / (double)c0->n : 0.0;
  if (info->evaluated == 0) return LSQR_EMPTY;  // no valid hypothesis
  if (consensus_out)
    for (int r = 0; r < n; r++) {
      lsqr_ctx *c = m->ctx[r];
      MHIP(m, hipSetDevice(c->device));
      if (hi[r] > lo[r])
        MHIP(m, hipMemcpyAsync(consensus_out + lo[r], c->d_mask + lo[r], hi[r] - lo[r], hipMemcpyDeviceToHost,
                               c->stream));
    }
  if (c0->cfg.model == LSQR_MODEL_DENSE && info->fit.reserved == 2 && c0->opt_dense_dd) {
    // the summed Gram block is ill-conditioned (a pivot below 1e-6 max|G|): its solution carries eps cond(A)^2, and a
    // one-device fit of the same rows would have taken the double-double route.  The records are replicated, so rank 0
    // masks the whole upload with the winner and fits from the rows -- the N-device fit IS the one-device fit then.
    MHIP(m, hipSetDevice(c0->device));
    uint64_t cnt = 0;
    int st2 = lsqr_mask(c0, winner, 0, c0->n, nullptr, &cnt);
    lsqr_fit_info fi;
    if (st2 == LSQR_OK) st2 = lsqr_ls_fit(c0, 1, fit, &fi);
    if (st2 != LSQR_OK && st2 != LSQR_EMPTY) return mfail(m, st2, "refit of an ill-conditioned dense system from the rows", c0);
    const uint64_t used = info->fit.n_used;
    info->fit = fi;
    info->fit.n_used = used;
    info->n_params = st2 == LSQR_OK ? fi.n_params : 0;
    st = st2;
  }
  if (st == LSQR_OK && wants_lm(c0->cfg) && c0->cfg.model != LSQR_MODEL_PHANTOM) {
    // Levenberg-Marquardt over the sharded consensus set: per evaluation every rank reduces its slice at the trial
    // point, the blocks are summed on rank 0, MINPACK's control flow runs there (lsqr_lm_begin / lsqr_lm_step)
    const int n1 = lsqr_moments_len(&c0->cfg, 1);
    double xt[64] = {0}, x0[64] = {0}, blk[LM_MOM_MAX + 8];
    for (int j = 0; j < info->n_params && j < 64; j++) x0[j] = fit[j];
    if ((st = lsqr_lm_begin(c0, x0, xt)) != LSQR_OK) return mfail(m, st, "lsqr_lm_begin", c0);
    for (;;) {
      for (int r = 0; r < n; r++) {
        if ((st = lsqr_moments_dev(m->ctx[r], 1, lo[r], hi[r], 1, xt, m->block[r])) != LSQR_OK)
          return mfail(m, st, "lsqr_moments_dev", m->ctx[r]);
        src[r] = m->block[r];
      }
      if (m->rccl) {  // one all-reduce SUM of the {sum f^2, J^T J, J^T f} block per evaluation
        std::vector<void *> buf(n);
        for (int r = 0; r < n; r++) buf[r] = m->block[r];
        if ((st = multi_allreduce(m, buf, (size_t)n1, ncclDouble, ncclSum)) != LSQR_OK) return st;
      } else {
      if ((st = multi_gather(m, src, m->g_block, sizeof(double) * n1, sizeof(double) * kMultiBlk)) != LSQR_OK)
        return st;
      hipLaunchKernelGGL(k_multi_sum, dim3((n1 + 255) / 256), dim3(256), 0, c0->stream, m->g_block, n, n1,
                         kMultiBlk, m->block[0]);
      MHIP(m, hipGetLastError());
      }
      MHIP(m, hipMemcpyAsync(c0->h_pin, m->block[0], sizeof(double) * n1, hipMemcpyDeviceToHost, c0->stream));
      for (int r = 0; r < n; r++) {  // every rank's x staging must be consumed before the next trial point
        MHIP(m, hipSetDevice(m->ctx[r]->device));
        MHIP(m, hipStreamSynchronize(m->ctx[r]->stream));
      }
      memcpy(blk, c0->h_pin, sizeof(double) * n1);
      int cont = 0;
      lsqr_fit_info fi;
      st = lsqr_lm_step(c0, blk, xt, &cont, fit, &fi);
      if (st != LSQR_OK && st != LSQR_EMPTY) return mfail(m, st, "lsqr_lm_step", c0);
      if (!cont) {
        const uint64_t used = info->fit.n_used;
        info->fit = fi;
        info->fit.n_used = used;
        info->n_params = fi.n_params;
        break;
      }
    }
  }
  for (int r = 0; r < n; r++) {
    MHIP(m, hipSetDevice(m->ctx[r]->device));
    MHIP(m, hipStreamSynchronize(m->ctx[r]->stream));
  }
  MHIP(m, hipSetDevice(c0->device));
  if (st != LSQR_OK) return st;
  if (params_out)
    for (int j = 0; j < info->n_params; j++) params_out[j] = fit[j];
  return LSQR_OK;
}
}  // namespace

extern "C" {

int lsqr_multi_create(const int *devices, int n, lsqr_multi **out) {
  if (!out || !devices || n < 1 || n > 64) return LSQR_ERR_INVALID;
  *out = nullptr;
  lsqr_multi *m = new lsqr_multi();
  m->n = n;
  int st = LSQR_OK;
  for (int r = 0; r < n && st == LSQR_OK; r++) {
    lsqr_ctx *c = nullptr;
    st = lsqr_ctx_create(devices[r], &c);
    if (st != LSQR_OK) break;
    m->ctx.push_back(c);
    unsigned long long *p = nullptr;
    double *b = nullptr;
    hipEvent_t e = nullptr;
    if (hipMalloc((void **)&p, 64) != hipSuccess || hipMalloc((void **)&b, sizeof(double) * kMultiBlk) != hipSuccess ||
        hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
      st = LSQR_ERR_HIP;
    m->packed.push_back(p);
    m->block.push_back(b);
    m->ev.push_back(e);
    for (int q = 0; q < r && st == LSQR_OK; q++)  // peer access both ways where the devices differ
      if (devices[q] != devices[r]) {
        int can = 0;
        (void)hipDeviceCanAccessPeer(&can, devices[r], devices[q]);
        if (can) {
          (void)hipSetDevice(devices[r]);
          (void)hipDeviceEnablePeerAccess(devices[q], 0);
          (void)hipSetDevice(devices[q]);
          (void)hipDeviceEnablePeerAccess(devices[r], 0);
          (void)hipGetLastError();  // "already enabled" is fine
        }
      }
  }
  if (st == LSQR_OK) {
    (void)hipSetDevice(devices[0]);
    if (hipMalloc((void **)&m->g_packed, sizeof(unsigned long long) * 64) != hipSuccess ||
        hipMalloc((void **)&m->g_block, sizeof(double) * kMultiBlk * n) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_root, hipEventDisableTiming) != hipSuccess)
      st = LSQR_ERR_HIP;
  }
  const char *tr = getenv("LSQR_MULTI_TRANSPORT");
  if (st == LSQR_OK && tr && !strcmp(tr, "rccl")) {
    // RCCL wants one communicator per DISTINCT device of the process; anything else keeps the peer copies
    bool distinct = true;
    for (int r = 0; r < n; r++)
      for (int q = 0; q < r; q++) distinct = distinct && devices[q] != devices[r];
    RcclApi &a = rccl_api();
    if (!a.ok || !distinct) {
      snprintf(m->err, sizeof m->err, "LSQR_MULTI_TRANSPORT=rccl: %s", !a.ok ? "librccl could not be loaded" : "the same device is listed twice");
      lsqr_multi_destroy(m);
      return LSQR_ERR_INVALID;
    }
    const auto t0 = std::chrono::steady_clock::now();
    m->comm.assign((size_t)n, nullptr);
    ncclResult_t e = a.CommInitAll(m->comm.data(), n, devices);
    if (e != ncclSuccess) {
      m->comm.clear();
      lsqr_multi_destroy(m);
      return LSQR_ERR_HIP;
    }
    m->rccl = true;
    // prove the communicators with one small all-reduce before anything is timed
    std::vector<void *> buf((size_t)n);
    for (int r = 0; r < n; r++) {
      buf[(size_t)r] = m->packed[(size_t)r];
      (void)hipSetDevice(devices[r]);
      (void)hipMemsetAsync(m->packed[(size_t)r], 0, 8, m->ctx[(size_t)r]->stream);
    }
    st = multi_allreduce(m, buf, 1, ncclUint64, ncclMax);
    for (int r = 0; r < n && st == LSQR_OK; r++) {
      (void)hipSetDevice(devices[r]);
      if (hipStreamSynchronize(m->ctx[(size_t)r]->stream) != hipSuccess) st = LSQR_ERR_HIP;
    }
    (void)hipSetDevice(devices[0]);
    m->rccl_bringup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  if (st != LSQR_OK) {
    lsqr_multi_destroy(m);
    return st;
  }
  *out = m;
  return LSQR_OK;
}

const char *lsqr_multi_transport(const lsqr_multi *m, double *bringup_seconds) {
  if (bringup_seconds) *bringup_seconds = m ? m->rccl_bringup_s : 0.0;
  return !m ? "none" : m->rccl ? "rccl" : "peer-copy";
}

void lsqr_multi_destroy(lsqr_multi *m) {
  if (!m) return;
  for (ncclComm_t cm : m->comm)
    if (cm) (void)rccl_api().CommDestroy(cm);
  m->comm.clear();
  for (size_t r = 0; r < m->ctx.size(); r++) {
    (void)hipSetDevice(m->ctx[r]->device);
    (void)hipStreamSynchronize(m->ctx[r]->stream);
    if (r < m->packed.size() && m->packed[r]) (void)hipFree(m->packed[r]);
    if (r < m->block.size() && m->block[r]) (void)hipFree(m->block[r]);
    if (r < m->ev.size() && m->ev[r]) (void)hipEventDestroy(m->ev[r]);
  }
  if (!m->ctx.empty()) (void)hipSetDevice(m->ctx[0]->device);
  if (m->g_packed) (void)hipFree(m->g_packed);
  if (m->g_block) (void)hipFree(m->g_block);
  if (m->ev_root) (void)hipEventDestroy(m->ev_root);
  for (lsqr_ctx *c : m->ctx) lsqr_ctx_destroy(c);
  delete m;
}

int lsqr_multi_size(const lsqr_multi *m) { return m ? m->n : 0; }
lsqr_ctx *lsqr_multi_ctx(lsqr_multi *m, int rank) { return (m && rank >= 0 && rank < m->n) ? m->ctx[rank] : nullptr; }
const char *lsqr_multi_last_error(const lsqr_multi *m) { return m ? m->err : "null handle"; }

int lsqr_multi_set_model(lsqr_multi *m, const lsqr_model_cfg *cfg) {
  if (!m || !cfg) return LSQR_ERR_INVALID;
  for (lsqr_ctx *c : m->ctx) {
    int st = lsqr_set_model(c, cfg);
    if (st != LSQR_OK) return mfail(m, st, "lsqr_set_model", c);
  }
  return LSQR_OK;
}

int lsqr_multi_upload(lsqr_multi *m, const void *host, size_t count, size_t stride_bytes) {
  if (!m) return LSQR_ERR_INVALID;
  lsqr_ctx *c0 = m->ctx[0];
  int st = lsqr_upload(c0, host, count, stride_bytes);  // ONE host -> device transfer
  if (st != LSQR_OK) return mfail(m, st, "lsqr_upload", c0);
  for (int r = 1; r < m->n; r++) {  // replicas: device to device
    lsqr_ctx *c = m->ctx[r];
    if ((st = need_ready(c, false)) != LSQR_OK) return mfail(m, st, "context not ready", c);
    if ((st = set_data_common(c, count, stride_bytes)) != LSQR_OK) return mfail(m, st, "bad records", c);
    if ((st = ensure(c, &c->d_data_owned, &c->data_cap, std::max<size_t>(count * c->stride, 1))) != LSQR_OK)
      return mfail(m, st, "replica allocation", c);
    if (count) MHIP(m, hipMemcpyPeerAsync(c->d_data_owned, c->device, c0->d_data, c0->device, count * stride_bytes,
                                           c->stream));
    c->d_data = c->d_data_owned;
  }
  for (int r = 1; r < m->n; r++) {
    MHIP(m, hipSetDevice(m->ctx[r]->device));
    MHIP(m, hipStreamSynchronize(m->ctx[r]->stream));
  }
  MHIP(m, hipSetDevice(c0->device));
  return LSQR_OK;
}

int lsqr_multi_batch_fit(lsqr_multi *m, uint64_t seed, uint64_t first, size_t H, double *params_out,
                         uint8_t *consensus_out, lsqr_ransac_info *info) {
  if (!m || H == 0 || H * (size_t)m->n > 0xFFFFFFF0ull) return LSQR_ERR_INVALID;
  const int n = m->n;
  lsqr_ctx *c0 = m->ctx[0];
  int st;
  std::vector<const void *> src(n);
  std::vector<void *> dst(n);
  for (int r = 0; r < n; r++) {
    if ((st = lsqr_step_scan(m->ctx[r], seed, first + (uint64_t)r * H, H, (uint32_t)((size_t)r * H),
                             (uint64_t *)m->packed[r])) != LSQR_OK)
      return mfail(m, st, "lsqr_step_scan", m->ctx[r]);
    src[r] = m->packed[r];
    dst[r] = m->packed[r];
  }
  if (m->rccl) {  // all-reduce MAX of the packed (votes << 32) | ~index: the earliest best hypothesis, on every rank
    if ((st = multi_allreduce(m, dst, 1, ncclUint64, ncclMax)) != LSQR_OK) return st;
  } else {
  if ((st = multi_gather(m, src, m->g_packed, 8, 8)) != LSQR_OK) return st;
  hipLaunchKernelGGL(k_multi_max, dim3(1), dim3(64), 0, c0->stream, m->g_packed, n, m->packed[0]);
  MHIP(m, hipGetLastError());
  if ((st = multi_bcast(m, m->packed[0], dst, 8)) != LSQR_OK) return st;
  }
  lsqr_ransac_info local;
  if (!info) info = &local;
  st = multi_finish(m, seed, first, params_out, consensus_out, info);
  info->iterations = H * (uint64_t)n;
  info->evaluated = info->evaluated ? H * (uint64_t)n : 0;
  if (info->best_votes) info->best_index += first;  // stream index of the winner
  return st;
}

// RANSAC<T,S>::compute() over several devices: every batch of the adaptive loop is sharded (rank r scans the r-th
// contiguous part), the replay runs on the host over the batch in stream order -- same winner, iteration count and
// consensus set as lsqr_ransac on one device.
int lsqr_multi_ransac(lsqr_multi *m, double p, uint64_t seed, double *params_out, uint8_t *consensus_out,
                      lsqr_ransac_info *info) {
  lsqr_ransac_info local;
  if (!info) info = &local;
  memset(info, 0, sizeof *info);
  if (!m) return LSQR_ERR_INVALID;
  const int n = m->n;
  lsqr_ctx *c0 = m->ctx[0];
  int st = need_ready(c0, false);
  if (st != LSQR_OK) return mfail(m, st, "context not ready", c0);
  const int k = c0->K;
  if (c0->n < (size_t)k || !(p < 1.0) || !(p > 0.0)) return LSQR_ERR_INVALID;  // RANSAC.hxx:16-19
  uint64_t rs[6];
  lsqr_replay_init(c0->n, k, p, rs);
  DedupSet dedup;
  uint64_t base = 0;
  size_t per = 256;  // hypotheses per device per batch
  std::vector<uint32_t> votes, subs;
  std::vector<uint8_t> valid;
  while (!rs[RS_DONE]) {
    uint64_t remaining = rs[RS_TRIES] - base;
    size_t total = (size_t)std::min<uint64_t>(remaining, (uint64_t)per * n);
    if (total == 0) break;
    votes.resize(total);
    valid.resize(total);
    subs.resize(total * k);
    std::vector<size_t> off(n + 1, 0);
    for (int r = 0; r < n; r++) off[r + 1] = std::min(total, off[r] + per);
    for (int r = 0; r < n; r++) {
      const size_t h = off[r + 1] - off[r];
      if (!h) continue;
      lsqr_ctx *c = m->ctx[r];
      c->hyp_expected = rs[RS_HAS] ? std::min<uint64_t>(remaining / n, 1u << 20) : 0;
      if ((st = lsqr_hypotheses_sample(c, seed, base + off[r], h, nullptr)) != LSQR_OK ||
          (st = lsqr_scan(c)) != LSQR_OK)
        return mfail(m, st, "batch scan", c);
    }
    for (int r = 0; r < n; r++) {
      const size_t h = off[r + 1] - off[r];
      if (!h) continue;
      lsqr_ctx *c = m->ctx[r];
      MHIP(m, hipSetDevice(c->device));
      MHIP(m, hipMemcpyAsync(subs.data() + off[r] * k, c->d_subsets, h * k * sizeof(uint32_t), hipMemcpyDeviceToHost,
                             c->stream));
      if ((st = lsqr_get_hypotheses(c, nullptr, valid.data() + off[r], votes.data() + off[r])) != LSQR_OK)
        return mfail(m, st, "lsqr_get_hypotheses", c);
    }
    size_t used = lsqr_replay(c0->n, k, p, subs.data(), valid.data(), votes.data(), total, base, &dedup, rs);
    info->evaluated += total;
    base += used;
    if (used < total) break;
    per = std::min<size_t>(per * 4, 4096);
    if (!rs[RS_HAS] && base >= (1ull << 22)) break;
    if (c0->opt_max_iter > 0 && base >= (uint64_t)c0->opt_max_iter) break;
  }
  info->iterations = rs[RS_I];
  info->best_index = rs[RS_BEST_IDX];
  info->best_votes = (uint32_t)rs[RS_BEST];
  info->fraction = (double)rs[RS_BEST] / (double)c0->n;
  if (!rs[RS_HAS] || rs[RS_BEST] == 0) return LSQR_EMPTY;  // RANSAC.hxx:129: nothing written
  // the winner, as a packed value relative to its own stream index, on every rank
  const unsigned long long pk = ((unsigned long long)rs[RS_BEST] << 32) | 0xFFFFFFFFull;
  for (int r = 0; r < n; r++) {
    lsqr_ctx *c = m->ctx[r];
    MHIP(m, hipSetDevice(c->device));
    MHIP(m, hipMemcpyAsync(m->packed[r], &pk, 8, hipMemcpyHostToDevice, c->stream));
    MHIP(m, hipStreamSynchronize(c->stream));  // pk lives on this stack frame
  }
  const uint64_t iters = info->iterations, eval = info->evaluated, bidx = info->best_index;
  const uint32_t bv = info->best_votes;
  st = multi_finish(m, seed, rs[RS_BEST_IDX], params_out, consensus_out, info);
  info->iterations = iters;
  info->evaluated = eval;
  info->best_index = bidx;
  info->best_votes = bv;
  if ((st == LSQR_OK || st == LSQR_EMPTY) && info->fit.n_used != bv)
    return mfail(m, LSQR_ERR_HIP, "consensus count of the slices differs from the winner's votes");
  return st;
}

static int set_option_one(lsqr_ctx *c, const char *name, int value);
int lsqr_set_option(lsqr_ctx *c, const char *name, int value) {
  if (!c || !name) return LSQR_ERR_INVALID;
  if (!strcmp(name, "batch_lanes")) {  // streams the pipelined batch entry points spread their slots over (1..4)
    if (value < 1 || value > lsqr_ctx::kMaxLanes) return fail(c, LSQR_ERR_INVALID, "batch_lanes must be 1..4");
    for (int s = 0; s < 2; s++)
      if (c->slot_busy[s]) return fail(c, LSQR_ERR_STATE, "batches in flight");
    for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
      if (c->lanes[i])
        for (int s = 0; s < 2; s++)
          if (c->lanes[i]->slot_busy[s]) return fail(c, LSQR_ERR_STATE, "batches in flight");
    c->opt_lanes = value;
    return LSQR_OK;
  }
  int st = set_option_one(c, name, value);
  if (st != LSQR_OK || c->is_lane) return st;
  bool found = false;
  for (auto &kv : c->opt_log)
    if (kv.first == name) {
      kv.second = value;
      found = true;
    }
  if (!found) c->opt_log.emplace_back(name, value);
  for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
    if (c->lanes[i]) (void)set_option_one(c->lanes[i], name, value);
  return LSQR_OK;
}
static int set_option_one(lsqr_ctx *c, const char *name, int value) {
  if (!strcmp(name, "scan_ppl")) {
    if (value != 0 && value != 2 && value != 4 && value != 8 && value != 16)
      return fail(c, LSQR_ERR_INVALID, "scan_ppl must be 0, 2, 4, 8 or 16");
    c->opt_ppl = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_filter")) {
    c->opt_filter = value < 0 ? 0 : (value > 3 ? 1 : value);  // 2 / 3: force pair / tile re-check
    return LSQR_OK;
  }
  if (!strcmp(name, "upload_threads")) {  // host threads of the staged upload (0: one plain hipMemcpy, -1: default)
    c->opt_upload_threads = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "max_iterations")) {  // budget for lsqr_ransac (0 = reference behaviour)
    c->opt_max_iter = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "syrk_diag")) {  // 1: loads only, 2: MFMAs only (timing diagnostics, wrong sums)
    c->opt_syrk_diag = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_bound")) {  // 1 (default): batch entry points skip hypotheses that cannot win; 0: count all
    c->opt_bound = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_mfma")) {  // 1 (default): LM pass on the matrix cores; 0: per-lane accumulators + shuffle trees
    c->opt_lm_mfma = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_fused")) {  // 1 (default): one launch per LM evaluation, result polled in pinned memory
    c->opt_lm_fused = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_host")) {
    c->opt_lm_host = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_index")) {  // 0 off, 1 auto, 2 always (point models)
    if (value < 0 || value > 2) return fail(c, LSQR_ERR_INVALID, "scan_index must be 0, 1 or 2");
    c->opt_index = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_cpt")) {  // cells per wave tile of the two-level scan
    if (value != 0 && value != 1) return fail(c, LSQR_ERR_INVALID, "scan_cpt must be 0 or 1");
    c->opt_cpt = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_bound_merge")) {  // cells per box of the bounds pass (0 = the cell model's default)
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
      return fail(c, LSQR_ERR_INVALID, "scan_bound_merge must be 0, 1, 2, 4 or 8");
    c->opt_bound_merge = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_pairs")) {  // 0: default, 1: k_scan_pairs for plain scans too (A/B)
    c->opt_pairs = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_axis")) {  // 1 (default): axis-sorted cells + vote bounds by rank (plane, 3-D); 0: off
    c->opt_axis = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_hyp_order")) {  // 1 (default): full counts of the plane in key order (similar planes share a group)
    c->opt_hyp_order = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "mom_chunk")) {
    c->opt_mom_chunk = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "us_mfma")) {  // 1 (default): US calibrations' scan on the fp16 matrix cores; 0: packed fp32 filter
    c->opt_us_h16 = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_pairs_mfma")) {  // r04's level 2 of the plane on the fp16 matrix cores: exact, measured 25 % slower;
    // out of the product since r05 (the kernel is kept under tools/attic/cells_h16.h)
    return value == 0 ? LSQR_OK : fail(c, LSQR_ERR_INVALID, "scan_pairs_mfma: removed (measured slower; tools/attic/cells_h16.h)");
  }
  if (!strcmp(name, "scan_pairs_waves")) {  // workgroups per CU of k_scan_pairs (0 = what fits)
    c->opt_pairs_waves = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_hsplit")) {  // hypothesis segments per tile of the two-level scan (0 = auto)
    if (value < 0 || value > 128) return fail(c, LSQR_ERR_INVALID, "scan_hsplit must be 0..128");
    c->opt_hsplit = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "dense_mask_ring")) {  // A/B: 2 (default) or 4 tile buffers per wave
    c->opt_mask_ring = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "dense_mask_diag")) {  // timing diagnostics (wrong results): 1 no MFMA, 2 no row evaluation
    c->opt_mask_diag = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "dense_mask_band")) {  // tests: widen the band of k_mask_syrk_dense's four-chain evaluation
    c->opt_mask_band = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "fuse_mask")) {  // winner's mask + moment block in one pass (default) or two kernels
    c->opt_fuse_mask = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_block")) {  // workgroup size of the two-level scan
    if (value != 0 && value != 256 && value != 257)
      return fail(c, LSQR_ERR_INVALID, "scan_block must be 0, 256 or 257 (256 + LDS broadcast)");
    c->opt_block = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_cell")) {  // observations per cell of the spatial index
    if (value != 0 && value != 256 && value != 512)
      return fail(c, LSQR_ERR_INVALID, "scan_cell must be 0, 256 or 512");
    c->opt_cell = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "dense_f32")) {  // 2 (default): dense scan filter on the fp16 matrix cores (two-way splits);
    c->opt_dense_f32 = value < 0 ? 0 : value > 2 ? 2 : value;  // 1: fp32 matrix cores; 0: fp64 matrix cores
    return LSQR_OK;
  }
  if (!strcmp(name, "us_fast_solve")) {  // 0: every minimal solve of the US calibrations through the SVD pseudo-inverse
    c->opt_us_fast = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "dense_fast_solve")) {  // 0: every minimal solve through the SVD pseudo-inverse
    c->opt_dense_fast = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_persist")) {  // 0: two launches per LM evaluation; 1 (alone on the device) / 3 (always): one persistent launch per fit, step on the host; 2: step on the device
    if (value < 0 || value > 3) return LSQR_ERR_INVALID;
    c->opt_lm_persist = (int)value;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_persist_wgs")) {  // resident workgroups of the persistent fit (0: by the contexts on the device)
    if (value < 0 || value > 1024) return LSQR_ERR_INVALID;
    c->opt_lm_persist_wgs = (int)value;
    return LSQR_OK;
  }
  if (!strcmp(name, "phantom_fast_solve")) {  // 1 (default): null vectors by LU + inverse iteration, Jacobi SVD as the fallback; 0: Jacobi only; n > 1: iteration limit
    c->opt_phantom_fast = (int)std::max<long long>(0, value);
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_test_overflow")) {  // tests: the deferred worklist check of lsqr_batch_fit_wait fires
    c->opt_test_overflow = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_persist_resident")) {
    c->opt_lm_persist_resident = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_persist_test_abort")) {
    c->opt_lm_persist_test_abort = (int)std::max<long long>(0, value);
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_persist_timeout_ms")) {
    if (value < 1 || value > 60000) return LSQR_ERR_INVALID;
    c->opt_lm_persist_timeout_ms = (int)value;
    return LSQR_OK;
  }
  if (!strcmp(name, "lm_tiles")) {  // 0: the r03 pass over the record-major compacted set (A/B knob)
    c->opt_lm_tiles = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "us_mask_mfma")) {  // 0: per-lane accumulators (k_mask_moments<US>; r03, A/B knob)
    c->opt_us_mask_mfma = value != 0;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_kd_after")) {  // hypotheses an upload is scanned by before the k-d levels above the runs are built
    if (value < 0) return fail(c, LSQR_ERR_INVALID, "scan_kd_after: >= 0");
    c->opt_kd_after = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_kd_levels")) {  // k-d levels above the 8192-record runs (0: Morton order above them, as until r05)
    if (value < 0 || value > 12) return fail(c, LSQR_ERR_INVALID, "scan_kd_levels: 0 .. 12");
    c->opt_kd_levels = value;
    drop_index(c);
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_refine")) {  // 0: cells are plain runs of the Morton order (r03; A/B knob)
    c->opt_refine = value != 0;
    drop_index(c);
    return LSQR_OK;
  }
  if (!strcmp(name, "scan_presorted")) {  // cells are runs of the upload order (set before the index is built)
    c->opt_presorted = value != 0;
    drop_index(c);
    return LSQR_OK;
  }
  if (!strcmp(name, "dense_wave_solve")) {  // 0: one workgroup per minimal solve (r03; bit-identical results, A/B)
    c->opt_dense_wave = value;
    return LSQR_OK;
  }
  if (!strcmp(name, "dense_dd")) {  // 0: an ill-conditioned dense fit stays on the Gram block (r03 behaviour; A/B)
    c->opt_dense_dd = value != 0;
    return LSQR_OK;
  }
  return fail(c, LSQR_ERR_INVALID, "unknown option %s", name);
}

int lsqr_scan_workload(lsqr_ctx *c, uint32_t *bound_out, uint64_t out[8]) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!out) return fail(c, LSQR_ERR_INVALID, "null argument");
  if (c->H == 0) return fail(c, LSQR_ERR_STATE, "no hypotheses");
  if (!c->index_valid) return fail(c, LSQR_ERR_STATE, "this upload has no spatial index (scan_index)");
  uint32_t *d_nc = c->d_votes2 + kPilots;  // scratch: per-hypothesis surviving cells (the batch has been read)
  const bool bounded = c->last_bound[0] != 0 && c->last_bound[3] == c->H;
  uint32_t h_sel[2] = {0, 0};
  if (bounded) {  // the selection of the bounded scan that just ran is still on the device
    HIPCHK(c, hipMemcpyAsync(c->h_pin, c->d_bsel, sizeof(BoundSel), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_stream(c));
    memcpy(h_sel, c->h_pin, sizeof h_sel);
  }
  st = dispatch(c->cfg, [&](auto tag) -> int {
    typedef typename decltype(tag)::type M;
    if constexpr (requires { typename CellOf<M>::type; }) {
      typedef typename CellOf<M>::type CM;
      return with_pp<CM>(c->cell_pts,
                         [&](auto pp) { return run_cells_bounds<CM, decltype(pp)::value>(c, c->d_ub, d_nc); });
    } else {
      return fail(c, LSQR_ERR_INVALID, "model has no two-level scan");
    }
  });
  if (st != LSQR_OK) return st;
  HIPCHK(c, hipMemsetAsync(c->d_counter + 2, 0, sizeof(unsigned long long), c->stream));
  if (bounded) {
    hipLaunchKernelGGL(k_sum_selected, dim3(1), dim3(256), 0, c->stream, c->d_sel, &c->d_bsel->n_pilot, d_nc,
                       c->d_counter + 2);
    hipLaunchKernelGGL(k_sum_selected, dim3(8), dim3(256), 0, c->stream, c->d_sel + kPilots, &c->d_bsel->n_rest, d_nc,
                       c->d_counter + 2);
    HIPCHK(c, hipGetLastError());
  }
  unsigned long long *pin = (unsigned long long *)c->h_pin;
  HIPCHK(c, hipMemcpyAsync(pin, c->d_counter + 4, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(pin + 1, c->d_counter + 2, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  if (bound_out)
    HIPCHK(c, hipMemcpyAsync(bound_out, c->d_ub, c->H * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, sync_stream(c));
  out[0] = pin[0];
  out[1] = (uint64_t)c->n_cells * ((c->H + 63) / 64);
  out[2] = c->n_cells;
  out[3] = c->cell_pts;
  out[4] = bounded ? 1 : 0;
  out[5] = h_sel[0];
  out[6] = h_sel[1];
  out[7] = bounded ? pin[1] : pin[0];
  return LSQR_OK;
}

int lsqr_lm_persist_info(const lsqr_ctx *c, uint64_t out[8], uint64_t *trace, uint32_t trace_cap, uint32_t *trace_n) {
  if (!c || !out) return LSQR_ERR_INVALID;
  for (int i = 0; i < 8; i++) out[i] = c->lmp_last[i];
  uint32_t nt = 0;
  if (trace) {
    nt = std::min(trace_cap, c->lmp_trace_n);
    memcpy(trace, c->lmp_trace, sizeof(uint64_t) * 4 * nt);
  }
  if (trace_n) *trace_n = nt;
  return LSQR_OK;
}

int lsqr_scan_work(lsqr_ctx *c, uint64_t out[6]) {
  int st = need_ready(c, true);
  if (st != LSQR_OK) return st;
  if (!out) return fail(c, LSQR_ERR_INVALID, "null argument");
  if (c->H == 0 || !c->scanned) return fail(c, LSQR_ERR_STATE, "no scanned batch");
  HIPCHK(c, sync_stream(c));  // the state of the last early-exit scan has landed in h_ee
  const uint64_t all = (uint64_t)c->H * (uint64_t)c->n;
  memset(out, 0, 6 * sizeof(uint64_t));
  out[2] = all;
  if (c->ee_last && c->h_ee && c->ee_H == c->H && c->ee_n == c->n) {
    out[0] = 1;
    out[1] = c->h_ee->work;
    out[3] = c->h_ee->n_cand;
    out[4] = c->h_ee->n_drop_first;
    out[5] = c->h_ee->n_alive;
  } else {
    out[1] = all;
    bsel_host_synced(c);
    if (c->last_bound[0] && c->last_bound[3] == c->H && c->bsel_known_H == c->H)
      out[3] = c->bsel_known.n_cand;  // bounded plane scan with rank bounds: the hypotheses whose bounds were refined
  }
  return LSQR_OK;
}

int lsqr_index_info(const lsqr_ctx *c, uint64_t out[4]) {
  if (!c || !out) return LSQR_ERR_INVALID;
  out[0] = c->index_valid ? 1 : 0;
  out[1] = c->n_sorted;
  out[2] = c->n_cells;
  out[3] = c->cell_pts;
  return LSQR_OK;
}

// ---- measurement ------------------------------------------------------------------------------------------
// (the lanes of lsqr_batch_fit_enqueue are profiled with their context: launches and times add up)
int lsqr_profile_enable(lsqr_ctx *c, int on) {
  if (!c) return LSQR_ERR_INVALID;
  prof_flush(c);
  c->prof = on != 0;
  for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
    if (c->lanes[i]) lsqr_profile_enable(c->lanes[i], on);
  return LSQR_OK;
}
int lsqr_profile_get(lsqr_ctx *c, int id, uint64_t *launches, double *total_ms) {
  if (!c || id < 0 || id >= 8) return LSQR_ERR_INVALID;
  prof_flush(c);
  uint64_t n = c->launches[id];
  double ms = c->ms[id];
  for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
    if (c->lanes[i]) {
      uint64_t ln = 0;
      double lms = 0.0;
      lsqr_profile_get(c->lanes[i], id, &ln, &lms);
      n += ln;
      ms += lms;
    }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  return LSQR_OK;
}
int lsqr_profile_reset(lsqr_ctx *c) {
  if (!c) return LSQR_ERR_INVALID;
  prof_flush(c);
  memset(c->launches, 0, sizeof c->launches);
  memset(c->ms, 0, sizeof c->ms);
  for (int i = 1; i < lsqr_ctx::kMaxLanes; i++)
    if (c->lanes[i]) lsqr_profile_reset(c->lanes[i]);
  return LSQR_OK;
}

}  // extern "C"
