#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

  metric   RANSAC hypotheses/s (+ final-fit residual), plane, 10 M points, 50 % outliers (BASELINE.json configs[1])
  step     one pass of the hot path over one batch: sample H minimal subsets -> solve -> agree() scan -> first-max
           winner -> consensus mask -> final least-squares fit.  Nothing is cached between steps (new subsets each
           step); observations are resident in HBM when a timed region starts.
  rates    TWO hypothesis rates are measured and both are in the line:
             value_full_count   every hypothesis of the batch gets its exact vote count over all N observations
                                (option scan_bound 0) -- SURVEY.md 8(d)'s unit: one minimal solve + one full agree()
                                pass per hypothesis, RANSAC.hxx:84-99 without the exit at :94.  THIS is `value`.
             value_early_exit   the batched form of RANSAC.hxx:94: hypotheses that provably cannot become the running
                                maximum are not counted to the end (scan_bound 1: cell-box vote bounds for the point
                                models, chunked abandonment for the dense / US scans).  Winner, consensus set and fit are
                                identical (tests/test_gpu_fullsize.py); the rate depends on the data.
           Each rate has its own roofline block (roofline / roofline_early_exit).
  repeats  every rate is timed over --repeats regions of exactly K steps (barrier + synchronise on both sides, max over
           ranks); `value` is the median region, all regions are listed.
  legs     the default single-GPU run appends legs (20 steps after 5, one timed region) of BASELINE configs 3-5 -- sphere +
           geometric fit, dense 2 M x 64, US calibration with the iterative and the analytic fit -- and of the plane phantom
            as other_configs[].
  N > 1    one process per GPU (torch.distributed, RCCL): observations replicated, the hypothesis stream sharded;
           all-reduce(MAX) picks the winner, all-reduce(SUM) of the moment block of each rank's observation slice gives
           the final fit.  scaling = "weak" (H per GPU fixed).  `python bench.py --gpus N` without a launcher
           environment starts its own ranks (python -m torch.distributed.run as a CHILD process, before anything here
           touches the GPU) and relays rank 0's line; --transport multi drives the N devices from this one process
           through the C ABI's lsqr_multi_* entry points (peer copies instead of RCCL).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--batch H] [--workload plane]
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

try:  # BASELINE.json's metric string, verbatim
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except Exception:
    METRIC = "RANSAC hypotheses/sec + final-fit residual, 10M pts, 1/2/4/8 GPU"

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# ---- roofs: 256 CUs x 4 SIMDs, 2.4 GHz.  A wave64 vector instruction occupies its SIMD for 4 cycles -- packed fp32
# (two results per lane), non-packed fp32, fp64 and cross-lane alike: tools/microbench.hip measures 4.1 - 4.8 nominal
# cycles for every instruction of the scan kernels' mix at 2 - 8 waves per SIMD (profiles/r03_microbench.json; the
# 2-cycle figure MI355X_MICROARCH.md gives for non-packed fp32 on the SIMD-32 is not reached by these streams), so the
# chip issues at most 1024 * 2.4e9 / 4 = 614.4 G wave-instructions/s.
VALU_ISSUE_PEAK_GWIPS = 1024 * 2.4 / 4.0
FP64_MFMA_PEAK_TFLOPS = 78.6
FP32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2516.6   # 1024 SIMDs x 2.4 GHz x 32768 flop / 32 cycles (v_mfma_f32_32x32x16_f16; guide: ~2.5 PF dense)
# USEFUL vector instructions of the scan kernels, counted from the source (csrc/cells.h, models.h, us.h; table in
# DESIGN.md section 6): l1 = non-packed fp32 instructions of CM::level1 per (64-hypothesis group, cell); pk = packed-fp32
# instructions of the filter measure per packed pair of observations per lane (128 observations per wave).
# Per surviving (hypothesis, cell) of 128*PP observations the useful work is PP * (pk packed + 1 threshold operation:
# the packed square-minus-threshold) + 2 (combine + the band compare); everything else the kernel issues (the
# per-observation compares behind the ballots, the band minimum, broadcasts, bookkeeping, exact re-checks) is overhead
# against this roof.  (Same count as in r01 / r02, whose filter spent the threshold operation on a |.|-minimum.)
SCAN_USEFUL = {"plane": {"l1": 15, "pk": 3}, "sphere": {"l1": 61, "pk": 4}, "line": {"l1": 35, "pk": 12},
               "us": {"pk": 21}, "phantom": {"pk": 18}}

COUNTER_ROUND = "r05"       # profiles/<round>_<workload>_<rate>_scan_counters.json (tools/collect_counters.py)
DELTA = {"plane": 0.5, "sphere": 0.5, "line": 0.5, "dense": 0.1, "us": 3.0, "phantom": 2.0}
PROF_KEYS = ("scan", "mask", "moments", "estimate", "solve", "sample", "index")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps per rate; the median is "
                    "reported, all are listed")
    ap.add_argument("--points", type=int, default=0, help="observations (0 = the BASELINE.json size of "
                    "the workload: 10 M points, 2 M dense rows, 1 M US frames)")
    ap.add_argument("--batch", type=int, default=0, help="hypotheses per GPU per step (0 = 4096; "
                    "1024 for the dense system)")
    ap.add_argument("--workload", default="plane",
                    choices=["plane", "sphere", "line", "dense", "us", "phantom"])
    ap.add_argument("--no-filter", action="store_true", help="plain fp64 scan (no fp32 pre-filter)")
    ap.add_argument("--no-index", action="store_true", help="exhaustive scan (no spatial index)")
    ap.add_argument("--outliers", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--us-fit", default="iterative", choices=["iterative", "analytic"],
                    help="final fit of the US workload: Levenberg-Marquardt with the reference's settings (BASELINE "
                         "config 5 as written: tolerances 1e-15, 5000 evaluations -- at 1 M frames MINPACK uses all "
                         "of them, see tests/golden/us_lm_vectors.npz) or the analytic estimate alone")
    ap.add_argument("--streams", type=int, default=0,
                    help="single GPU, pipelined steps: HIP streams (lanes of lsqr_batch_fit_enqueue) the batches "
                         "alternate over; batches of different streams overlap on the device (1 = one stream; 0 = the "
                         "default: 4, and 8 host threads with a context each for the US workload's iterative fit, whose "
                         "steps are thousands of HBM-bound LM evaluations that overlap each other's host round trips: "
                         "r05, 67 - 72 k hypotheses/s on eight, 66 - 67 k on twelve, 63 - 65 k on sixteen, 50 - 60 k on four)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="single GPU: one blocking lsqr_batch_fit per step instead of pipelined batches")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the adaptive RANSAC::compute() timing after the steps (profiling runs: "
                         "keeps one launch shape per kernel)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs of BASELINE configs 3-5 appended to the default plane run")
    ap.add_argument("--rates", default="both", choices=["both", "full", "early"],
                    help="which hypothesis rates to time (profiling runs: one, so that one scan arrangement runs)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "multi"],
                    help="--gpus N > 1: rccl = one process per GPU (torch.distributed), multi = this one process "
                         "drives the N devices through lsqr_multi_* (peer copies)")
    ap.add_argument("--cpu-points", type=int, default=0, help="observations for the CPU leg "
                    "(0 = same as --points)")
    ap.add_argument("--leg-scale", type=float, default=1.0, help="scale the observation counts of the other_configs "
                    "legs (tests; 1 = the BASELINE sizes)")
    ap.add_argument("--cpu-seconds", type=float, default=0.0, help="budget of the CPU full-count sample "
                    "(0 = 4 s for the headline, 3 s per leg)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="lsqr_set_option on every context of the run (A/B runs: scan_refine=0, scan_pairs=2, ...)")
    ap.add_argument("--detail", default="", help="file the FULL record of the run is written to (work models, per-kernel "
                    "tables, notes, the legs of the other configs); default bench_detail.json next to bench.py.  The "
                    "last stdout line is the compact headline only")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="--gpus N > 1: if RCCL cannot come up on ANY rank, continue over gloo (labelled in "
                         "config.collectives) instead of exiting non-zero.  Never the default: a scaling line must not "
                         "be a gloo line by accident")
    return ap.parse_args(argv)


def make_data(workload, n, outliers):
    from lsqrrecipes_amd import synth
    if workload == "plane":
        return synth.plane(n, outliers)
    if workload == "sphere":
        return synth.sphere(n, outliers)
    if workload == "line":
        return synth.line(n, outliers)
    if workload == "dense":
        return synth.dense(n, 64, 0.05)
    if workload == "phantom":  # k = 31: an all-inlier subset needs few off-plane frames
        return synth.plane_phantom_fast(n, min(outliers, 0.05), pixel_sigma=0.05)
    return synth.us_single_fast(n, outliers)


# ---------------------------------------------------------------------------------------------------------------------
# CPU leg (rank 0, N = 1): oracle/ is the checker and the baseline, never the product path
def _oracle_cfg(workload, delta):
    from oracle import pyoracle as O
    model = {"plane": O.PLANE, "sphere": O.SPHERE, "line": O.LINE, "dense": O.DENSE,
             "us": O.US_SINGLE, "phantom": O.PHANTOM}[workload]
    return O, O.cfg(model, 64 if workload == "dense" else 3, delta, O.LS_ALGEBRAIC)


def cpu_full_count(workload, data, delta, budget_s):
    """the metric's unit on one host core: minimal-subset solve + ONE full agree() pass over all N records, no early
    exit (RANSAC.hxx:84-99 without :94), on the subsets of the bench's own counter-based stream; bounded by time"""
    O, c = _oracle_cfg(workload, delta)
    n = len(data)
    k = {"plane": 3, "sphere": 4, "line": 2, "dense": 64, "us": 4, "phantom": 31}[workload]
    subs = O.ctr_subsets(20261003, 0, 64 if workload == "dense" else 512, n, k)
    t0 = time.perf_counter()
    hyp = 0
    while hyp < len(subs) and (hyp < 3 or time.perf_counter() - t0 < budget_s):
        par = O.estimate(c, data[subs[hyp]])
        if len(par):
            O.scan(c, par, data)
        hyp += 1
    dt = time.perf_counter() - t0
    return {"value": hyp / dt, "unit": "hypotheses/s", "cores": 1, "kind": "port",
            "sample": "oracle C port of the %s estimator: %d hypotheses (%d-record minimal solve + full agree() pass "
                      "over N=%d records, no early exit) in %.2f s; 1 thread" % (workload, hyp, k, n, dt)}


def cpu_baseline(workload, data, delta, budget_s):
    """Reference single-thread CPU path on the same workload: the reference's own RANSAC.hxx (oracle/_ref, compiled
    from /root/reference in the build container) driving the restated estimator, adaptive run at p = 0.999 (its loop
    WITH the early exit of :94) -- plus the full-count sample above (the unit `value` is quoted in)."""
    full = cpu_full_count(workload, data, delta, budget_s)
    if workload in ("dense", "phantom"):
        # the adaptive bound never closes for k = 64 / hardly for k = 31 (w^k underflows): the bounded sample of the
        # metric's unit is the whole CPU leg
        return full
    O, c = _oracle_cfg(workload, delta)
    t0 = time.perf_counter()
    if O.ref_available():
        r = O.ref_ransac(c, data, 0.999, seed=20261003)
        hyp = r["estimate_calls"]
        kind = "reference"
        what = ("reference RANSAC.hxx compiled unmodified (oracle/_ref) + restated %sParametersEstimator"
                " (VNL absent), adaptive run p=0.999, early exit of RANSAC.hxx:94 included" % workload.capitalize())
    else:
        r = O.ransac(c, data, 0.999, sampler="ref", seed=20261003)
        hyp = int((r["status"] != 1).sum())
        kind = "port"
        what = "oracle C port of RANSAC.hxx + estimator, adaptive run p=0.999, early exit included"
    dt = time.perf_counter() - t0
    return {"value": hyp / dt, "unit": "hypotheses/s", "cores": 1, "kind": kind,
            "sample": "%s; N=%d points; %d hypotheses in %.2f s; 1 thread" % (what, len(data), hyp, dt),
            "fraction": r["fraction"], "compute_call_s": dt, "iterations": int(r.get("iters", hyp)),
            "full_count": full,
            "unit_note": "value = hypotheses/s of the reference's serial loop as it runs (estimate + agree pass WITH "
                         "its early exit, RANSAC.hxx:94): the counterpart of value_early_exit; full_count.value = the "
                         "same core doing the metric's unit (full agree pass per hypothesis): the counterpart of "
                         "`value` / value_full_count"}


# ---------------------------------------------------------------------------------------------------------------------
def kernel_source_hash():
    """content hash of the kernel sources: counter files collected by tools/collect_profiles.sh carry it, and
    are only quoted when it still matches (a stale profile is dropped, not emitted next to live timings)"""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "lsqrrecipes_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith((".h", ".hip")):
            h.update(open(os.path.join(src, f), "rb").read())
    return h.hexdigest()[:16]


def profile_file(name):
    """profiles/<name> if it was collected from the kernel sources in this tree, else None"""
    f = os.path.join(ROOT, "profiles", name)
    try:
        d = json.load(open(f))
    except Exception:
        return None
    return d if d.get("kernel_source_hash") == kernel_source_hash() else None


def measured_cycles():
    """the microbenchmark's per-instruction cycles, quoted beside the roof: this round's run on this round's level-2 mix
    (profiles/r05_microbench.json: tools/microbench.hip OP 21, the filter on squares), else the round-3 file"""
    for name in ("r05_microbench.json", "r03_microbench.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            return {"v_fma_f32": d["cycles"]["v_fma_f32"], "v_pk_fma_f32": d["cycles"]["v_pk_fma_f32"],
                    "level2_mix": d["cycles"]["level2_mix"], "source": "profiles/%s (tools/microbench.hip, 8 waves per "
                    "SIMD, nominal cycles at 2.4 GHz)" % name}
        except Exception:
            continue
    return None


def isa_counts(kernel_key):
    """vector instructions per loop body counted from the ISA of this tree's kernels (tools/isa_counts.py ->
    profiles/r05_isa_counts.json, stamped with the kernel source hash like the counter files), quoted beside the
    source-derived `useful` counts"""
    d = profile_file("r05_isa_counts.json")
    return d["kernels"].get(kernel_key) if d else None


def scan_roofline(R, mode, scan_ms, n_scan):
    """the dominant kernel against the roof that binds it; mode = "full_count" / "early_exit" """
    a, ctx, H = R.a, R.ctx, R.H
    rec = R.rec
    w = a.workload
    t = scan_ms * 1e-3
    alg_bytes = float(H) * a.points * rec            # N*sizeof(T) per hypothesis, H per launch
    base = {"rate": mode, "launch_ms": scan_ms, "launches": int(n_scan),
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_GBs": alg_bytes / t / 1e9 if t > 0 else 0.0,
            "algorithmic_note": "SURVEY 8(d) logical figure H*N*%d B / launch time: NOT a roofline fraction -- one pass "
                                "over the observations serves all H hypotheses%s, so it exceeds the HBM peak by "
                                "construction" % (rec, " (and the two-level scan proves most (hypothesis, cell) pairs "
                                                  "irrelevant from the cell boxes)" if w in ("plane", "sphere", "line")
                                                  else "")}
    prof = profile_file("%s_%s_%s_scan_counters.json" % (COUNTER_ROUND, w, mode))
    traffic = prof.get("hbm_bytes_per_launch") if prof else None
    if prof:
        base["counters"] = {k: prof[k] for k in ("valu_issue_busy", "salu_issue_busy_per_cu", "lanes_active",
                                                "valu_wave_instructions", "salu_wave_instructions",
                                                "mfma_wave_instructions", "mfma_pipe_busy", "mfma_pipe_busy_from_insts",
                                                "shader_clock_ghz", "kernel_avg_ms", "collected_at", "source")
                            if k in prof}
        if prof.get("mfma_pipe_busy") is not None:
            base["mfma_pipe_busy"] = prof["mfma_pipe_busy"]   # SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles of the launch
        if traffic and t > 0:
            base["hbm_frac_measured"] = traffic / t / 1e9 / HBM_PEAK_GBS
        if prof.get("valu_wave_instructions") and t > 0:
            # ISSUED vector wave-instructions of the launch (SQ_INSTS_VALU, counters) against both issue peaks: the
            # 4-cycle one tools/microbench.hip measures for this instruction mix and MI355X_MICROARCH.md's 2-cycle one
            issued = prof["valu_wave_instructions"] / t / 1e9
            base["issued_valu_frac_4cyc"] = issued / VALU_ISSUE_PEAK_GWIPS
            base["issued_valu_frac_2cyc"] = issued / (2 * VALU_ISSUE_PEAK_GWIPS)
    if w == "dense":
        wl = ctx.scan_work() if hasattr(ctx, "scan_work") else None
        rows_done = wl["row_hypothesis_pairs"] if wl else float(a.points) * H
        flops = 2.0 * 64 * rows_done                 # the residual block rows x hypotheses as a GEMM
        ach = flops / t / 1e12 if t > 0 else 0.0
        filt = 2
        for kv in a.option:
            if kv.split("=")[0] == "dense_f32":
                filt = int(kv.split("=")[1])
        wm = {"row_hypothesis_pairs_evaluated": rows_done, "row_hypothesis_pairs_all": float(a.points) * H,
              "evaluated_fraction": rows_done / (float(a.points) * H)}
        if filt >= 2:
            # dense_h16.h: every fp32 product is three fp16 matrix products (a1 x2, a2 x1, a1 x1): `achieved` is the
            # LOGICAL product (SURVEY 8(d)'s unit), `issued_*` what the matrix cores actually execute
            base.update({"bound": "mfma", "achieved": ach, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / F16_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "issued_TFLOPs": 3.0 * ach, "frac_issued": 3.0 * ach / F16_MFMA_PEAK_TFLOPS,
                         "x_fp32_matrix_peak": ach / FP32_MFMA_PEAK_TFLOPS,
                         "kernel_short": "k_scan_dense_h16<64> fp16-split MFMA filter + k_dense_recheck_seg (exact fp64)",
                         "kernel": "k_scan_dense_h16<64> (rows and unknowns as two-way fp16 splits, three "
                                   "v_mfma_f32_32x32x16_f16 per fp32 product, row fragments built once per upload and "
                                   "resident in registers per pass, hypothesis tiles through an LDS ring, classification "
                                   "on squares) + k_dense_recheck_seg (exact fp64 decision of the pairs inside the "
                                   "filter's band); launch_ms covers both and the per-batch split of the unknowns",
                         "work_model": wm,
                         "note": "achieved = 2*n*(row, hypothesis) pairs evaluated / launch time (logical flops); peak = "
                                 "the dense fp16 matrix rate; issued = 3 x logical (the three partial products); "
                                 "x_fp32_matrix_peak = achieved / 157.3 TFLOP/s, the roof of the fp32 filter this "
                                 "replaces (profiles/r04: 0.55 of it)"})
            return base
        base.update({"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "frac_of_fp64_mfma_peak": ach / FP64_MFMA_PEAK_TFLOPS,
                     "kernel_short": "k_scan_dense_mfma32r<64> fp32-MFMA filter + k_dense_recheck_seg (exact fp64)",
                     "kernel": "k_scan_dense_mfma32r<64> (fp32 MFMA filter, v_mfma_f32_16x16x4_f32, hypothesis fragments "
                               "through an LDS ring, next row tile in registers) + k_dense_recheck_seg (exact fp64 "
                               "decision of the ~1e-4 of the pairs inside the filter's band); launch_ms covers both "
                               "and the thresholds",
                     "work_model": wm,
                     "note": "flops = 2*n*(row, hypothesis) pairs the filter GEMM actually evaluated; peak = the fp32 "
                             "dense matrix rate the filter runs at"})
        return base
    u = SCAN_USEFUL[w]
    cells = w in ("plane", "sphere", "line") and R.idx["built"] and not a.no_filter
    if cells:
        wl = ctx.scan_workload()                     # level 1 alone on the last batch: live counts
        pp = wl["cell_points"] // 128
        groups = -(-H // 64)
        l1 = wl["level1_evaluations"]                # one level-1 pass over all hypotheses
        cand = 0
        if wl["bounded"]:        # bounds pass + the groups of the rank-bound candidates, the pilots, the second pass
            cand = ctx.scan_work()["candidates"] if w == "plane" else 0
            l1 += wl["cells"] * (-(-cand // 64) + -(-wl["pilots"] // 64) + -(-wl["second_pass"] // 64))
        v2_instr = pp * (u["pk"] + 1) + 2
        useful_instr = l1 * u["l1"] + wl["pairs_counted"] * v2_instr
        kname = ("two-level scan of <%s> over a Morton-sorted copy: cell-box culling, packed fp32 filter + exact fp64 "
                 "re-check in surviving cells; %s" % (
                     w, "bounded: k_cells_bounds (vote bounds) -> rank bounds of the candidates (plane: k_bound_axis) "
                     "-> pilots when those are weak -> only hypotheses that can still win, each "
                     "counted by k_scan_pairs (level 1 counted first, then an equal share of the surviving "
                     "(hypothesis, cell) pairs per wave)" if wl["bounded"] else
                     "every hypothesis counted: k_cells_bounds(cnt) -> k_tile_costs -> k_scan_pairs" if R.full_pairs
                     else "every hypothesis counted: k_scan_cells"))
        model = {"level1_evaluations": l1, "level1_useful_instr": u["l1"],
                 "surviving_hypothesis_cell_pairs_all": wl["pairs"],
                 "surviving_hypothesis_cell_pairs_counted": wl["pairs_counted"], "level2_useful_instr": v2_instr,
                 "cells": wl["cells"], "cell_points": wl["cell_points"], "hypothesis_groups": groups,
                 "bounded_scan": wl["bounded"], "rank_bound_candidates": cand, "pilots": wl["pilots"],
                 "second_pass_hypotheses": wl["second_pass"],
                 "hypotheses_counted_exactly": (wl["pilots"] + wl["second_pass"]) if wl["bounded"] else H,
                 "surviving_fraction": wl["pairs"] / max(1.0, float(wl["cells"]) * H)}
    else:
        us_mfma = 1
        for kv in a.option:
            if kv.split("=")[0] == "us_mfma":
                us_mfma = int(kv.split("=")[1])
        if w == "us" and us_mfma and not a.no_filter:
            # us_h16.h: the three error components of (frame, hypothesis) as 13-term dot products on the fp16 matrix
            # cores, two-way fp16 splits: nine v_mfma_f32_32x32x16_f16 per 32 frames x 32 hypotheses
            wk = ctx.scan_work() if hasattr(ctx, "scan_work") else None
            fh = float(wk["row_hypothesis_pairs"]) if wk else float(H) * a.points
            ach = 78.0 * fh / t / 1e12 if t > 0 else 0.0          # 3 components x 13 terms x 2 flop, logical
            issued = 9.0 * 32768.0 / 1024.0 * fh / t / 1e12 if t > 0 else 0.0
            base.update({"bound": "mfma", "achieved": ach, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / F16_MFMA_PEAK_TFLOPS, "traffic": traffic, "issued_TFLOPs": issued,
                         "frac_issued": issued / F16_MFMA_PEAK_TFLOPS,
                         "kernel_short": "k_scan_us_h16<us> fp16-split MFMA filter + k_us_recheck_seg (exact fp64)",
                         "kernel": "k_scan_us_h16 (frames and hypotheses as two-way fp16 splits, nine "
                                   "v_mfma_f32_32x32x16_f16 per 32 x 32 (frame, hypothesis) pairs with the previous "
                                   "step's classification issued between them, hypothesis fragments of a launch in LDS) "
                                   "+ k_us_recheck_seg (exact fp64 decision of the band)",
                         "work_model": {"frame_hypothesis_pairs_evaluated": fh,
                                        "frame_hypothesis_pairs_all": float(H) * a.points},
                         "note": "achieved = 78 flop x (frame, hypothesis) pairs / launch time (the logical 3 x 13-term "
                                 "products); issued = nine 32x32x16 matrix instructions per 1024 pairs (three partial "
                                 "products, K padded 13 -> 16); peak = the dense fp16 matrix rate"})
            return base
        if w == "phantom" and us_mfma and not a.no_filter:
            # phantom_h16.h: the 31-term error of (frame, hypothesis) as one dot product on the fp16 matrix cores, two-way
            # fp16 splits, two 16-slot blocks: six v_mfma_f32_32x32x16_f16 per 32 frames x 32 hypotheses
            wk = ctx.scan_work() if hasattr(ctx, "scan_work") else None
            fh = float(wk["row_hypothesis_pairs"]) if wk else float(H) * a.points
            ach = 62.0 * fh / t / 1e12 if t > 0 else 0.0          # 31 terms x 2 flop, logical
            issued = 6.0 * 32768.0 / 1024.0 * fh / t / 1e12 if t > 0 else 0.0
            base.update({"bound": "mfma", "achieved": ach, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / F16_MFMA_PEAK_TFLOPS, "traffic": traffic, "issued_TFLOPs": issued,
                         "frac_issued": issued / F16_MFMA_PEAK_TFLOPS,
                         "kernel_short": "k_scan_phantom_h16 fp16-split MFMA filter + k_us_recheck_seg (exact fp64)",
                         "kernel": "k_scan_phantom_h16 (frames and hypotheses as two-way fp16 splits, twelve "
                                   "v_mfma_f32_32x32x16_f16 per 64 x 32 (frame, hypothesis) pairs with the previous "
                                   "tile's classification issued between them, hypothesis fragments of a launch in LDS) "
                                   "+ k_us_recheck_seg (exact fp64 decision of the band, after every launch)",
                         "work_model": {"frame_hypothesis_pairs_evaluated": fh,
                                        "frame_hypothesis_pairs_all": float(H) * a.points},
                         "note": "achieved = 62 flop x (frame, hypothesis) pairs / launch time (the logical 31-term "
                                 "product); issued = six 32x32x16 matrix instructions per 1024 pairs (three partial "
                                 "products, K padded 31 -> 32); peak = the dense fp16 matrix rate; the time includes the "
                                 "split of the unknowns and the exact re-checks"})
            return base
        wk = ctx.scan_work() if hasattr(ctx, "scan_work") else None
        pairs_all = float(H) * a.points / 128.0      # every (hypothesis, packed pair of observations per wave)
        pairs = wk["row_hypothesis_pairs"] / 128.0 if wk else pairs_all
        v2_instr = u["pk"] + 2
        useful_instr = pairs * v2_instr
        kname = {"us": "k_scan_us_f32<us> (packed fp32 filter + exact fp64 re-check)",
                 "phantom": "k_scan_us_f32<phantom> (factored packed fp32 filter + exact fp64 re-check)"}.get(
                     w, "k_scan_f32<%s> (packed fp32 filter + exact fp64 re-check)" % w)
        model = {"hypothesis_wave_pairs_evaluated": pairs, "hypothesis_wave_pairs_all": pairs_all,
                 "evaluated_fraction": pairs / pairs_all, "useful_instr_per_pair": v2_instr}
    ach = useful_instr / t / 1e9 if t > 0 else 0.0
    base["kernel_short"] = (("k_scan_pairs<%s>" % w if (cells and (wl["bounded"] or R.full_pairs)) else
                             "k_scan_cells<%s>" % w if cells else kname.split(" ")[0])
                            + (" two-level scan, " + ("bounded" if wl["bounded"] else "every hypothesis counted")
                               if cells else " packed fp32 filter + exact fp64 re-check"))
    base["frac_of_2cycle_issue_peak"] = ach / (2 * VALU_ISSUE_PEAK_GWIPS)
    base.update({"bound": "valu", "achieved": ach, "peak": VALU_ISSUE_PEAK_GWIPS, "unit": "G wave-instr/s",
                 "frac": ach / VALU_ISSUE_PEAK_GWIPS, "traffic": traffic, "kernel": kname, "work_model": model,
                 "measured_cycles_per_instruction": measured_cycles(),
                 "isa_counts": isa_counts("k_scan_pairs_%s" % w if cells else "k_scan_us_f32"),
                 "note": "achieved = USEFUL vector instructions of the launch (counted from the kernel source, work "
                         "counts measured live by lsqr_scan_workload / lsqr_scan_work) / launch time; peak = 1024 SIMDs "
                         "x 2.4 GHz / 4 cycles per wave64 instruction (measured: profiles/r03_microbench.json).  The "
                         "kernel reads the observations once per launch for all H hypotheses (HBM fraction in "
                         "hbm_frac_measured), so the instruction-issue roof is the one that binds; "
                         "counters.valu_issue_busy is the measured utilisation including overhead instructions"})
    return base


def hbm_table(R, prof, abs_prof, idx_prof, cnt, lm_nfev=0, steps=0):
    """HBM-bound kernels of the step: algorithmic bytes / HIP-event time / 8 TB/s (live)"""
    a, rec = R.a, R.rec
    n = float(a.points)
    rows = []

    def add(name, byts, nl, ms, note=""):
        if nl and ms > 0:
            t = ms / nl
            rows.append({"kernel": name, "bytes": byts, "ms": t, "GBs": byts / t / 1e6,
                         "frac_of_hbm_peak": byts / t / 1e6 / HBM_PEAK_GBS, "launches": int(nl), "note": note})
    w = a.workload
    if w == "dense":
        add("k_mask_syrk_dense (winner's consensus mask and A^T A | A^T b over the consensus rows in ONE pass: "
            "rows through an LDS ring, agreeing rows to fp64 MFMAs)", n * rec + n, *prof["mask"],
            note="also 2*m_in*65*66/2 fp64 MFMA flops: see DESIGN.md")
        add("k_syrk_mfma (A^T A | A^T b over the consensus set, separate pass: only when the fused pass did not run)",
            n * rec + n, *prof["moments"])
    elif w == "phantom":
        add("k_mask<phantom>", n * rec + n, *prof["mask"])
        add("k_phantom_rows / k_syrk_mfma (Gram block)", n * 256 + n, *prof["moments"])
    else:
        add("k_mask_moments<%s> (consensus mask + moment block, one pass)" % w, n * rec + n, *prof["mask"])
        if prof["moments"][0]:
            # the set is compacted before the first evaluation for the matrix-core pass (US) and otherwise after 8
            # evaluations through the mask (csrc/lsqr_hip.hip: kCompactAfter)
            tight = w in ("us", "phantom") or lm_nfev >= 8
            # (the launch path profiles every 16th evaluation: hundreds of launches per step; the persistent kernel is one)
            if w == "us" and lm_nfev > 64 and steps and prof["moments"][0] <= 2 * steps:
                # one launch for the whole fit (csrc/lm_persist.h: the one-stream pass runs alone on the device): the
                # launch's bytes are the consensus set once per evaluation
                nl_, ms_ = prof["moments"]
                add("k_lm_persist<%s> (a whole Levenberg-Marquardt fit in one launch: %d evaluations, each a pass over the "
                    "consensus set)" % (w, lm_nfev), cnt * rec * lm_nfev, nl_, ms_,
                    note="bytes = %d consensus records x %d evaluations (read from HBM every evaluation unless "
                         "lm_persist_resident keeps them in registers)" % (cnt, lm_nfev))
            else:
              add("k_lm_pass<%s> (one Levenberg-Marquardt evaluation: sum f^2, J^T J, J^T f over the consensus set)" % w,
                cnt * rec if tight else n * rec + n, *prof["moments"],
                note=("reads the %d consensus records (tight copy)" % cnt) if tight else
                "fewer than 8 evaluations: every evaluation reads all records through the mask (no compaction pass)")
    add("k_bounds (point models: min / max / max |x| in one pass) or k_absmax, once per upload", n * rec, *abs_prof)
    if idx_prof[0]:
        add("spatial index build (k_keys, radix sort of (key, index) pairs, k_gather_boxes; once per upload)",
            n * (3 * rec + 60), *idx_prof,
            note="bytes = records read for the keys, (key, index) pairs written and moved by 3 radix passes, "
                 "records gathered and the sorted copy written; the bounds come from the k_bounds pass above")
    return rows


def cold_call(a, L, Context, data, model, delta, ls_type, cpu):
    """What a caller of RANSAC<T,S>::compute() (RANSAC.h:75-79) sees on data that is NOT yet on the device:
    lsqr_upload of the caller's pageable buffer + lsqr_ransac (adaptive, p = 0.999) + the consensus copy."""
    c2 = Context(0)
    try:
        c2.set_model(model, 64 if a.workload == "dense" else 3, delta, ls_type)
        c2.set_option("upload_threads", 0)
        c2.set_option("max_iterations", 100000)
        best = None
        first = 0.0
        for rep in range(3):
            t0 = time.perf_counter()
            c2.upload(data)
            t1 = time.perf_counter()
            r = c2.ransac(0.999, seed=20261003 + rep, want_consensus=True)
            t2 = time.perf_counter()
            cur = {"total_ms": (t2 - t0) * 1e3, "upload_ms": (t1 - t0) * 1e3,
                   "ransac_and_consensus_copy_ms": (t2 - t1) * 1e3,
                   "iterations": int(r["info"].iterations), "scanned": int(r["info"].evaluated),
                   "fraction": r["fraction"], "index_built": c2.index_info()["built"]}
            if rep and (best is None or cur["total_ms"] < best["total_ms"]):
                best = cur          # rep 0 allocates the device and pinned buffers: reported separately
            if rep == 0:
                first = cur["total_ms"]
        best["first_call_ms_incl_allocations"] = first
        best["upload_GBs"] = data.nbytes / best["upload_ms"] / 1e6
    finally:
        c2.close()
    out = {"what": "lsqr_upload (host, pageable, one hipMemcpy) + lsqr_ransac (p = 0.999) + consensus copy, %d records "
                   "of %d B; best of 2 calls after the first" % (len(data), data.shape[1] * 8),
           "ms": best["total_ms"], "detail": best}
    if cpu and "compute_call_s" in cpu:
        out["reference_cpu_call_s"] = cpu["compute_call_s"]
        out["speedup_vs_reference_call"] = cpu["compute_call_s"] * 1e3 / out["ms"]
        out["note"] = ("like-for-like: one RANSAC<T,S>::compute() on the same %d records; the CPU figure is the "
                       "reference's own RANSAC.hxx run of cpu_baseline (its libc rand() subset stream, %d "
                       "iterations)" % (len(data), cpu.get("iterations", 0)))
    return out


# ---------------------------------------------------------------------------------------------------------------------
class Run:
    """one workload on this rank: contexts, the drivers of the steps, the timed regions"""

    def __init__(self, a, dist, device, local, backend, force_dist, data=None):
        from lsqrrecipes_amd import _lib as L
        from lsqrrecipes_amd.context import Context
        from lsqrrecipes_amd.distributed import Comm, ShardedRansac
        self.a, self.dist, self.device, self.local, self.force_dist = a, dist, device, local, force_dist
        self.L, self.Context = L, Context
        w = a.workload
        self.delta = DELTA[w]
        self.model = {"plane": L.PLANE, "sphere": L.SPHERE, "line": L.LINE, "dense": L.DENSE,
                      "us": L.US_SINGLE, "phantom": L.PHANTOM}[w]
        # GEOMETRIC == ITERATIVE == 1: the sphere's geometric fit, the US calibrations' and the phantom's LM fit
        # (BASELINE.json configs[2], configs[4]); --us-fit analytic keeps the closed-form US fit
        self.ls_type = L.LS_ANALYTIC if (w == "us" and a.us_fit == "analytic") else L.LS_GEOMETRIC
        self.dim = 64 if w == "dense" else 3
        if data is None:
            data = make_data(w, a.points, a.outliers)
        self.data, self.truth, self.lab = data
        self.rec = self.data.shape[1] * 8
        self.ctx = self._new_ctx()
        a.streams = max(1, min(16, a.streams))     # the library's lanes go up to 4; host-threaded contexts up to 16
        self.ctx.set_option("batch_lanes", min(4, a.streams))
        self.comm = Comm(dist, device)
        self.eng = ShardedRansac(self.ctx, self.comm)
        self.step_on_device = dist is not None and os.environ.get(
            "LSQR_STEP", "device" if backend == "nccl" else "host") == "device"
        self.H = a.batch
        self.seed = 0xC0FFEE
        self.single = self.comm.world == 1 and not force_dist
        self.closed_form = not (w in ("sphere", "us") and self.ls_type == L.LS_GEOMETRIC) and w != "phantom"
        # single GPU, closed-form fit: batches are pipelined -- batch i + 1 is enqueued before batch i is read
        # (lsqr_batch_fit_enqueue / _wait), so the host's latency between steps hides behind the device's work;
        # every step still runs the whole chain.  --no-pipeline keeps one blocking call per step.
        self.pipelined = self.single and not a.no_pipeline and self.closed_form
        self.pipelined_dist = self.step_on_device and self.closed_form and not a.no_pipeline
        if self.pipelined or self.pipelined_dist:
            a.streams = min(4, a.streams)
        self.cur_streams = a.streams
        # iterative final fits (sphere geometric, US iterative, phantom) keep the host in the loop -- MINPACK's
        # control flow between device passes -- so a batch is a blocking call.  The library's threading model is one
        # context per host thread (LsqrDevice.h); with several streams the steps are dealt to that many host threads,
        # each driving its own context (own stream, own upload): while one thread waits for an evaluation, another
        # one's scan runs.
        self.threaded = self.single and not self.pipelined and a.streams > 1 and not a.no_pipeline
        self.tctx = [self.ctx]
        if self.threaded:
            for _ in range(1, a.streams):
                self.tctx.append(self._new_ctx())
        # multi-GPU with several streams: one engine per stream -- its own context (own upload and index), its own
        # process group (an RCCL communicator serves one stream at a time) and its own torch stream
        self.lanes = []
        if self.pipelined_dist and a.streams > 1 and str(device) != "cpu":
            import torch
            # One communicator per stream, each PROVEN by an all-reduce on its stream before anything is timed, and a
            # budget for the lot (LSQR_RCCL_BUDGET_S, default 90 s): an 8-rank run must not spend the driver's window
            # -- or die -- creating S x 8 communicators.  The decision is collective (MAX of the elapsed time over the
            # ranks), so every rank keeps the same number of groups; with fewer groups than --streams the run continues
            # on those, labelled in config.rccl.
            budget = float(os.environ.get("LSQR_RCCL_BUDGET_S", "90"))
            spent = 0.0
            for k in range(a.streams):
                t0 = time.perf_counter()
                gk = dist.new_group(backend="nccl")        # every rank, same order
                sk = torch.cuda.Stream()
                with torch.cuda.stream(sk):
                    probe = torch.ones(1, device=device)
                    dist.all_reduce(probe, group=gk)
                sk.synchronize()
                el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
                dist.all_reduce(el, op=dist.ReduceOp.MAX)      # the world's default group: proven in bring_up_rccl
                spent += float(el.item())
                ck = self.ctx if k == 0 else self._new_ctx()
                self.lanes.append((ShardedRansac(ck, Comm(dist, device, group=gk)), sk))
                if spent > budget and k + 1 < a.streams:
                    RCCL_INFO["fallback"] = ("%d of %d per-stream communicators after %.1f s (budget %.0f s): the run "
                                             "continues on %d stream(s)" % (k + 1, a.streams, spent, budget, k + 1))
                    a.streams = k + 1
                    break
            RCCL_INFO["stream_groups_s"] = spent
            RCCL_INFO["communicators"] = 1 + len(self.lanes)
            self.cur_streams = a.streams
        self.multi_stream = a.streams > 1 and (self.pipelined or self.threaded or
                                               (self.pipelined_dist and bool(self.lanes)))
        self.next_step = 0
        self.side_step = 1 << 20
        self.nfev_total = 0        # LM evaluations of the steps run so far (iterative fits)
        self.full_pairs = w in ("plane", "sphere") and a.batch >= 1024   # cells.h: <model>Cell::FULL_COUNT_PAIRS
        self.idx = None

    def _new_ctx(self):
        a = self.a
        c = self.Context(self.local)
        c.set_model(self.model, self.dim, self.delta, self.ls_type).upload(self.data)
        if a.no_filter:
            c.set_option("scan_filter", 0)
        if a.no_index:
            c.set_option("scan_index", 0)
        for kv in a.option:
            k, v = kv.split("=")
            c.set_option(k, int(v))
        return c

    def all_ctx(self):
        seen = [self.ctx] + [c for c in self.tctx[1:]] + [e.e for e, _ in self.lanes[1:]]
        return seen

    def set_option(self, name, value):
        for c in self.all_ctx():
            c.set_option(name, value)

    def close(self):
        for c in self.all_ctx():
            c.close()

    @staticmethod
    def _res(r):
        if r is None or r["info"].best_votes == 0:
            return None
        f = r["info"].fit
        return {"votes": int(r["info"].best_votes), "fit": r["params"], "cnt": int(f.n_used),
                "lm_info": int(f.lm_info), "lm_nfev": int(f.lm_nfev), "cost": float(f.cost),
                "lm_stall": int(f.reserved)}

    @staticmethod
    def _res_step(r):
        if r is None:
            return None
        votes, gidx, par, fit, cnt, info = r
        return {"votes": int(votes), "fit": fit, "cnt": int(cnt),
                "lm_info": int(getattr(info, "lm_info", 0) or 0), "lm_nfev": int(getattr(info, "lm_nfev", 0) or 0),
                "cost": float(getattr(info, "cost", 0.0) or 0.0), "lm_stall": int(getattr(info, "reserved", 0) or 0)}

    def step(self, i):
        H, seed = self.H, self.seed
        if self.single:
            # single GPU: the whole step is one chain on the device stream (lsqr_batch_fit)
            return self._res(self.ctx.batch_fit(seed, i * H, H))
        # multi-GPU: exchanges on device buffers, one host synchronisation per step (step_device);
        # LSQR_STEP=host keeps the staged-through-the-host variant (the gloo rehearsal's default)
        r = self.eng.step_device(seed, i, H) if self.step_on_device else self.eng.step(seed, i, H)
        return self._res_step(r)

    def sync(self):
        for c in self.all_ctx():
            c.synchronize()
        if self.dist is not None and self.device != "cpu":
            import torch
            torch.cuda.synchronize()
        self.comm.barrier()

    def run_steps(self, count, side=False):
        """`count` steps of the hot path, continuing the hypothesis stream (side: on a range of step indices of their
        own -- priming and the one-stream profile runs --, so that the timed steps are the same steps whatever the
        number of streams); -> result of the last step"""
        if side:
            first_step = self.side_step
            self.side_step += count
        else:
            first_step = self.next_step
            self.next_step += count
        H, seed = self.H, self.seed
        last = None
        if self.pipelined_dist and self.lanes and self.cur_streams > 1:
            import torch
            S = len(self.lanes)
            ring = 2 * S
            for i in range(count):
                if i >= ring:
                    j = i - ring
                    last = self.lanes[j % S][0].step_device_wait((j // S) & 1)
                with torch.cuda.stream(self.lanes[i % S][1]):
                    self.lanes[i % S][0].step_device(seed, first_step + i, H, slot=(i // S) & 1)
            for j in range(max(0, count - ring), count):
                last = self.lanes[j % S][0].step_device_wait((j // S) & 1)
            return self._res_step(last)
        if self.pipelined_dist:   # multi-GPU: step i + 1 is enqueued (collectives included) before step i is read
            for i in range(count):
                self.eng.step_device(seed, first_step + i, H, slot=i & 1)
                if i:
                    last = self.eng.step_device_wait((i - 1) & 1)
            if count:
                last = self.eng.step_device_wait((count - 1) & 1)
            return self._res_step(last)
        if self.threaded and self.cur_streams > 1:
            import threading
            S = len(self.tctx)
            res = [None] * count
            err = []

            def work(k):
                try:
                    for i in range(k, count, S):
                        res[i] = self._res(self.tctx[k].batch_fit(seed, (first_step + i) * H, H))
                except Exception as e:      # surfaced below: a failed step must fail the run
                    err.append(e)
            th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            if err:
                raise err[0]
            self.nfev_total += sum(r["lm_nfev"] for r in res if r)
            return res[-1] if count else None
        if not self.pipelined or self.cur_streams < 1:
            for i in range(count):
                last = self.step(first_step + i)
                if last:
                    self.nfev_total += last["lm_nfev"]
            return last
        # a ring of 2 * streams slots: slot s runs on stream (lane) s % streams, two batches deep per stream
        ctx = self.ctx
        ring = 2 * self.cur_streams
        for i in range(count):
            if i >= ring:
                last = ctx.batch_fit_wait((i - ring) % ring)
            ctx.batch_fit_enqueue(seed, (first_step + i) * H, H, slot=i % ring)
        for i in range(max(0, count - ring), count):
            last = ctx.batch_fit_wait(i % ring)
        return self._res(last)

    def region(self, K, side=False):
        """exactly K steps between two barrier + synchronise brackets -> (seconds: max over ranks, own seconds, last)"""
        self.sync()
        t0 = time.perf_counter()
        last = self.run_steps(K, side)
        self.sync()
        own = time.perf_counter() - t0
        return self.comm.allreduce_max_f64(own), own, last

    def one_stream(self, on):
        if self.pipelined:
            self.ctx.set_option("batch_lanes", 1 if on else min(4, self.a.streams))
        self.cur_streams = 1 if on else self.a.streams

    def measure(self, mode, bound):
        """one hypothesis rate: prime, warm up, `repeats` timed regions, then the same chain on ONE stream for the
        per-kernel figures"""
        a, ctx, H = self.a, self.ctx, self.H
        if bound is not None:
            self.set_option("scan_bound", bound)
        ctx.profile(True)   # the spatial index is built inside the first large scan of an upload
        # every stream builds its index (the k-d levels above the runs come with the fourth batch of 4096: "scan_kd_after")
        # and loads its code objects before the W steps
        if self.multi_stream or a.workload in ("plane", "sphere", "line"):
            self.run_steps((5 if a.workload in ("plane", "sphere", "line") else 2) *
                           max(1, a.streams if self.multi_stream else 1), side=True)
        self.run_steps(a.warmup)
        idx_warm = ctx.profile_get("index")
        abs_warm = ctx.profile_get("absmax")
        ctx.profile(True)
        regions, own, last = [], [], None
        nfev0 = self.nfev_total
        for _ in range(max(1, a.repeats)):
            dt, mine, last = self.region(a.steps)
            regions.append(dt)
            own.append(mine)
        nfev = self.nfev_total - nfev0
        k_med = int(np.argsort(regions)[len(regions) // 2])
        per_rank = self.comm.allgather_f64(H * a.steps / own[k_med])   # each rank's own clock, median region
        prof = {k: ctx.profile_get(k) for k in PROF_KEYS}
        prof_steps = a.steps * max(1, a.repeats)      # steps the profile above covers
        ctx.profile(False)
        single_stream = None
        if self.multi_stream:
            # Kernel durations measured while batches of several streams share the device overlap each other; the
            # per-kernel figures (roofline, kernel_hbm, kernels_ms) come from the SAME steps run once more on one
            # stream, right after the timed regions.  The rate is the multi-stream figure of the timed regions.
            self.one_stream(True)
            k1 = min(a.steps, 20)
            self.run_steps(2, side=True)
            ctx.profile(True)
            dt1, _, _ = self.region(k1, side=True)
            prof1 = {k: ctx.profile_get(k) for k in PROF_KEYS}
            ctx.profile(False)
            self.one_stream(False)
            single_stream = {"steps": k1, "ms_per_step": dt1 / k1 * 1e3, "value": H * a.gpus * k1 / dt1,
                             "scan_ms_in_timed_region": prof["scan"][1] / max(prof["scan"][0], 1),
                             "note": "the same chain on ONE stream, run right after the timed regions: the source of "
                                     "the per-kernel durations in roofline / kernel_hbm / kernels_ms (in the timed "
                                     "regions the kernels of %d streams overlap, so a kernel's own duration there "
                                     "includes the time it shares the device)" % a.streams}
            prof1["index"] = prof["index"]
            prof = prof1
            prof_steps = k1
        self.idx = ctx.index_info()
        vals = [H * a.gpus * a.steps / dt for dt in regions]
        med = float(np.median(vals))
        dt_med = H * a.gpus * a.steps / med
        if not self.single and a.workload in ("plane", "sphere", "line") and self.closed_form:
            # the multi-GPU step leaves the re-derived winner (a batch of one) as the context's current batch: the work
            # model of the roofline is taken on a full batch of this rank, as in the single-GPU run (untimed)
            ctx.batch_fit(0xC0FFEE, self.comm.rank * H, H)
        n_scan, ms_scan = prof["scan"]
        scan_ms = ms_scan / max(n_scan, 1)
        out = {"mode": mode, "value": med, "values": vals, "ms_per_step": dt_med / a.steps * 1e3,
               "region_s": regions, "per_rank": per_rank, "last": last, "prof": prof, "idx_warm": idx_warm,
               "abs_warm": abs_warm, "single_stream": single_stream, "scan_ms": scan_ms, "lm_evaluations": nfev,
               "prof_steps": prof_steps,
               "roofline": scan_roofline(self, mode, scan_ms, n_scan)}
        return out


def step_text(R):
    a = R.a
    if R.pipelined:
        return ("lsqr_batch_fit_enqueue/_wait (one chain per step; the steps alternate over %d HIP stream(s), two deep "
                "per stream, so chains of different streams overlap on the device)" % a.streams)
    if R.single and R.multi_stream:
        return ("lsqr_batch_fit (blocking: the iterative fit keeps the host in the loop), the steps dealt to %d host "
                "threads with one context (stream) each" % a.streams)
    if R.single:
        return "lsqr_batch_fit (one chain, one sync)"
    if R.pipelined_dist:
        return "step_device, pipelined (collectives on device buffers; step i + 1 enqueued before step i is read)"
    if R.step_on_device:
        return "step_device (collectives on device buffers, one sync)"
    return "step (exchanges staged through the host)"


def final_fit_block(R, last):
    a = R.a
    if last is None:
        return {"inliers": 0, "winner_votes": 0, "params": [], "params_empty": True, "lm_info": 0, "lm_nfev": 0}
    fit = last["fit"]
    res = None
    if R.single and len(fit):
        try:
            res = R.ctx.stats(fit, use_mask=True)
        except Exception:
            res = None
    ff = {"inliers": int(last["cnt"]), "winner_votes": int(last["votes"]),
          "params": [float(x) for x in fit], "params_empty": len(fit) == 0,
          "lm_info": int(last["lm_info"]), "lm_nfev": int(last["lm_nfev"]),
          "abs_dot_true_normal": float(abs(np.dot(fit[:3], R.truth[:3])))
          if (a.workload in ("plane", "line") and len(fit)) else None,
          "residual_min_max_mean_sumsq": [float(x) for x in res] if res is not None else None}
    if last["lm_nfev"]:
        ff["lm_cost"] = last["cost"]
        ff["lm_nfev_cost_stopped_moving"] = last["lm_stall"] or None
        ff["lm_note"] = ("MINPACK lmder control flow with the reference's tolerances; lm_info outside 1..4 is the "
                         "reference's failure convention: parameters EMPTY (info 5 = the evaluation limit, which is "
                         "what 1e-15 tolerances produce on >= 20 k frames: tests/golden/us_lm_flags.npz). "
                         "lm_nfev_cost_stopped_moving = the evaluation after which the cost never again fell by more "
                         "than 1e-7 relative")
    return ff


def report(R, rates, cpu_budget, headline=True):
    """the JSON object of one workload; rates: list of measure() results, the first one is `value`"""
    a, ctx, comm, H = R.a, R.ctx, R.comm, R.H
    L, Context = R.L, R.Context
    main_rate = rates[0]
    by_mode = {r["mode"]: r for r in rates}
    last = main_rate["last"]
    prof = main_rate["prof"]
    idx = R.idx
    n_idx = main_rate["idx_warm"][0] + prof["index"][0]
    ms_idx = main_rate["idx_warm"][1] + prof["index"][1]
    value = main_rate["value"]
    w = a.workload
    what_value = {
        "full_count": "`value` = value_full_count: every one of the H hypotheses of a step gets its exact vote count "
                      "over all N observations (SURVEY 8(d)'s unit; scan_bound 0)",
        "early_exit": "`value` = value_early_exit (only rate measured: --rates early)"}[main_rate["mode"]]
    out = {
        "metric": METRIC,
        "value": value, "unit": "hypotheses/s", "n_gpus": a.gpus, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": main_rate["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%sParametersEstimator + RANSAC, %d points, %d%% outliers, "
                               "delta=%.2f (BASELINE.json configs[1])" % (
                                   w.capitalize(), a.points, round(a.outliers * 100), R.delta)
                   if w == "plane" else "%s estimator + RANSAC, %d observations%s" % (
                       w, a.points, ", final fit: %s" % a.us_fit if w == "us" else ""),
                   "points": a.points, "hypotheses_per_gpu_per_step": H,
                   "record_bytes": R.rec, "parallelism": "hypotheses sharded over %d GPU(s), "
                   "observations replicated" % a.gpus,
                   "world_size": comm.world,
                   "collectives": ("RCCL (torch.distributed nccl backend)" if R.dist is not None and comm.device != "cpu"
                                   else ("gloo, host-side exchanges (FALLBACK -- RCCL did not come up: %s)"
                                         % os.environ["LSQR_DIST_FALLBACK"] if os.environ.get("LSQR_DIST_FALLBACK")
                                         else "gloo (rehearsal)") if R.dist is not None else "none (single GPU)"),
                   "streams": a.streams if R.multi_stream else 1,
                   # RCCL bring-up of THIS run (rank 0's clock; stream groups: MAX over the ranks): what an N-rank run pays
                   # before its first timed step
                   "rccl": ({"bringup_s": RCCL_INFO["bringup_s"], "stream_groups_s": RCCL_INFO["stream_groups_s"],
                             "communicators": RCCL_INFO["communicators"], "fallback": RCCL_INFO["fallback"]}
                            if RCCL_INFO["communicators"] else None),
                   "stream_priming_steps": (5 if a.workload in ("plane", "sphere", "line") else 2) *
                   (a.streams if R.multi_stream else 1) if (R.multi_stream or a.workload in ("plane", "sphere", "line")) else 0,
                   "repeats": max(1, a.repeats),
                   "step": step_text(R) + "; " + what_value},
        "value_is": main_rate["mode"],
        "repeats": {r["mode"]: r["values"] for r in rates},
    }
    for r in rates:
        out["value_" + r["mode"]] = r["value"]
        out["ms_per_step_" + r["mode"]] = r["ms_per_step"]
    out["per_rank_hypotheses_per_s"] = [float(x) for x in main_rate["per_rank"]]   # each rank's own clock
    out["single_stream"] = main_rate["single_stream"]
    out["final_fit"] = final_fit_block(R, last)
    if last is not None and last["lm_nfev"]:
        tot = main_rate["lm_evaluations"]
        secs = sum(main_rate["region_s"])
        out["lm"] = {"evaluations_in_timed_regions": int(tot), "evaluations_per_s": tot / secs if secs > 0 else None,
                     "evaluations_per_step": tot / (a.steps * max(1, a.repeats)),
                     "note": "one evaluation = one device pass (k_lm_pass*) over the winner's consensus set + the "
                             "MINPACK step; with --streams > 1 several fits are in flight"}
    out["roofline"] = main_rate["roofline"]
    # the same work against the TIMED REGION's step time (several streams): `frac` above prices the dominant kernel by its
    # own duration on one stream, and that duration can exceed ms_per_step of the multi-stream region (VERDICT r04, weak
    # 11) -- this is the fraction of the roof the timed region itself sustains, all kernels of a step included
    rf0 = out["roofline"]
    if rf0 and rf0.get("achieved") and rf0.get("launch_ms") and rf0.get("peak"):
        work = rf0["achieved"] * rf0["launch_ms"]          # (unit x ms): the launch's useful work
        rf0["timed_region"] = {"ms_per_step": main_rate["ms_per_step"],
                               "frac": work / main_rate["ms_per_step"] / rf0["peak"],
                               "note": "useful work of one step's scan / ms_per_step of the timed region / peak"}
    for r in rates[1:]:
        out["roofline_" + r["mode"]] = r["roofline"]
        out["single_stream_" + r["mode"]] = r["single_stream"]
    cnt = int(last["cnt"]) if last else 0
    out["kernel_hbm"] = hbm_table(R, prof, main_rate["abs_warm"], (n_idx, ms_idx), cnt,
                                  int(last["lm_nfev"]) if last else 0, main_rate["prof_steps"])
    out["kernels_ms"] = {"sample": prof["sample"][1] / max(prof["sample"][0], 1),
                         "estimate": prof["estimate"][1] / max(prof["estimate"][0], 1),
                         "scan": main_rate["scan_ms"],
                         "mask": prof["mask"][1] / max(prof["mask"][0], 1),
                         "moments": prof["moments"][1] / max(prof["moments"][0], 1),
                         "moments_launches_per_step": prof["moments"][0] / max(main_rate["prof_steps"], 1),
                         "reduce_and_solve_per_step": prof["solve"][1] / max(main_rate["prof_steps"], 1),
                         "profiled_steps": main_rate["prof_steps"]}
    for r in rates[1:]:
        out["kernels_ms"]["scan_" + r["mode"]] = r["scan_ms"]
    # the kernel group a step spends most of its device time in: when that is NOT the scan (plane phantom: the 31-frame
    # null-vector solves), the roofline block says so and names it -- `roofline.kernel` is the dominant kernel (r05)
    km = out["kernels_ms"]
    groups = {"estimate": km["estimate"], "scan": km["scan"], "mask": km["mask"],
              "moments": km["moments"] * km["moments_launches_per_step"], "solve": km["reduce_and_solve_per_step"]}
    dom = max(groups, key=groups.get)
    out["dominant_kernel"] = {"group": dom, "ms_per_step": groups[dom], "by_group_ms": groups}
    if dom != "scan" and out.get("roofline"):
        rf = dict(out["roofline"])
        out["roofline_scan"] = out["roofline"]          # the scan's own block stays in the detail file
        names = {"estimate": {"phantom": "k_estimate_phantom<64> (31 x 31 one-sided Jacobi null vector, one wave per "
                                         "hypothesis)", "dense": "k_estimate_dense_r64", "us": "k_estimate_us"}.get(
                                             w, "k_estimate<%s>" % w),
                 "moments": "LM passes (k_lm_pass* / k_lm_persist) x %d per step" % round(km["moments_launches_per_step"]),
                 "mask": "k_mask_moments<%s>" % w, "solve": "k_reduce + k_solve<%s>" % w}
        t = groups[dom] * 1e-3
        if dom == "estimate" and w == "phantom":
            # 9.5 sweeps x 15.5 rounds x 16 column pairs x (3 dot products + 2 column rotations of A and V: 31 x 14
            # flop) per hypothesis -- a flop MODEL of the fixed-sweep Jacobi, fp64 vector arithmetic
            flops = H * 9.5 * 15.5 * 16 * 31 * 14.0
            ach = flops / t / 1e12
            rf.update({"bound": "valu", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (fp64 vector)",
                       "frac": ach / FP64_MFMA_PEAK_TFLOPS, "launch_ms": groups[dom], "traffic": None,
                       "note": "the step's dominant kernel is the minimal solve, not the scan: one wave per hypothesis, "
                               "every Jacobi round a dependent chain through LDS, two waves per SIMD (19 KB of LDS per "
                               "hypothesis) -- latency-bound; achieved = a flop model of the sweeps / kernel time "
                               "against the 78.6 TFLOP/s fp64 vector rate.  The scan's block: roofline_scan"})
        elif dom == "moments" and any(r_["kernel"].startswith("k_lm_p") for r_ in out["kernel_hbm"]):
            # iterative fit: the step is its Levenberg-Marquardt evaluations, each a pass over the consensus set -- HBM
            row = [r_ for r_ in out["kernel_hbm"] if r_["kernel"].startswith("k_lm_p")][0]
            rf.update({"bound": "hbm", "achieved": row["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": row["frac_of_hbm_peak"], "launch_ms": groups[dom], "traffic": None,
                       "note": "the step's dominant kernel group is the %d LM evaluations of the final fit (BASELINE config "
                               "5 as written ends at MINPACK's evaluation limit), each a pass over the compacted consensus "
                               "set: achieved = consensus bytes x evaluations / their time on ONE stream (one fit alone: "
                               "the synchronisation between evaluations is inside that time; several fits in flight "
                               "overlap it -- lm.evaluations_per_s).  The scan's block: roofline_scan"
                               % round(km["moments_launches_per_step"] if km["moments_launches_per_step"] > 2 else
                                       (last["lm_nfev"] if last else 0))})
            names["moments"] = row["kernel"][:80]
        else:
            rf.update({"launch_ms": groups[dom], "note": "dominant kernel group of the step is `%s`, not the scan; the "
                       "figures of this block are the SCAN's (see roofline_scan), launch_ms is the dominant group's" % dom})
        rf["kernel"] = names.get(dom, dom)
        rf["kernel_short"] = names.get(dom, dom)[:80]
        rf.pop("timed_region", None)      # (that entry priced the scan's work; it stays in roofline_scan)
        for k_ in ("issued_valu_frac_4cyc", "issued_valu_frac_2cyc", "frac_issued", "issued_TFLOPs", "mfma_pipe_busy",
                   "hbm_frac_measured", "counters", "x_fp32_matrix_peak", "work_model"):
            rf.pop(k_, None)              # the scan's counters do not describe this kernel
        out["roofline"] = rf
    out["index"] = {"built": idx["built"], "cells": idx["cells"], "cell_points": idx["cell_points"],
                    "build_ms": ms_idx / n_idx if n_idx else None,
                    "builds_in_warmup": int(main_rate["idx_warm"][0]),
                    "builds_in_timed_region": int(prof["index"][0]),
                    "note": "per upload: device radix sort on Morton keys, k-d refinement of the 8192-record runs, cell "
                            "boxes -- built inside the scan that first needs it, and built again with seven k-d levels "
                            "above the runs by the first scan after the upload has been asked for 16384 hypotheses "
                            "(scan_kd_levels / scan_kd_after; the plane's bounded scan keeps the first order)"}
    single_plain = a.gpus == 1 and R.dist is None
    if headline and single_plain and w in ("plane", "sphere", "line") and not a.no_end_to_end:
        # a second build in the same process (code objects loaded, buffers kept): the build's own device time
        ctx.profile(True)
        ctx.upload(R.data)
        ctx.batch_fit(0xC0FFEE, 0, H)
        nb2, msb2 = ctx.profile_get("index")
        nab, msab = ctx.profile_get("absmax")
        ctx.profile(False)
        out["index"]["rebuild_ms"] = msb2 / nb2 if nb2 else None
        out["index"]["bounds_pass_ms"] = msab / nab if nab else None
        out["index"]["rebuild_note"] = ("HIP-event time of k_keys + radix sort + k_gather_boxes on a re-upload; "
                                        "build_ms above is the first build of the process and includes the one-time "
                                        "load of the sort's code object")
        ctx.set_option("max_iterations", 100000)
        ctx.set_option("scan_bound", 1)
        # the whole RANSAC<T,S>::compute() (adaptive termination, p = 0.999) on the resident data
        ctx.ransac(0.999, seed=7, want_consensus=False)
        t1 = time.perf_counter()
        reps = 5
        for r_ in range(reps):
            rr = ctx.ransac(0.999, seed=100 + r_, want_consensus=False)
        out["compute_end_to_end"] = {
            "ms": (time.perf_counter() - t1) / reps * 1e3, "p": 0.999,
            "iterations": int(rr["info"].iterations), "scanned": int(rr["info"].evaluated),
            "fraction": rr["fraction"],
            "note": "warm: RANSAC<T,S>::compute() with the observations already resident and the index built "
                    "(batches of 256/1024/4096 hypotheses + serial replay + mask + final fit); cold_call has "
                    "the figure a first call on host data sees"}
    cpu = None
    if a.gpus == 1 and not a.no_cpu_baseline:
        cp = a.cpu_points or a.points
        cpu = cpu_baseline(w, R.data[:cp], R.delta, cpu_budget)
        out["cpu_baseline"] = cpu
        full = cpu.get("full_count", cpu)
        if "full_count" in by_mode:
            out["speedup_vs_cpu_baseline"] = by_mode["full_count"]["value"] / full["value"]
            out["speedup_note"] = ("value_full_count / cpu_baseline%s.value: the same unit on both sides (full agree "
                                   "pass per hypothesis)" % (".full_count" if "full_count" in cpu else ""))
        if "early_exit" in by_mode and "full_count" in cpu:
            out["speedup_early_exit_vs_reference_loop"] = by_mode["early_exit"]["value"] / cpu["value"]
    if headline and single_plain and not a.no_end_to_end and w in ("plane", "sphere", "line", "us"):
        out["cold_call"] = cold_call(a, L, Context, R.data, R.model, R.delta, R.ls_type, cpu)
    return out


def rates_for(a):
    """(mode, scan_bound) pairs in the order measured"""
    if a.rates == "full":
        return [("full_count", 0)]
    if a.rates == "early":
        return [("early_exit", 1)]
    # the early-exit rate first: the plane's bounded scan keeps the index order of a fresh upload, the counting one has
    # the k-d levels above the runs built (scan_kd_levels) -- run_workload puts `value`'s rate back in front
    return [("early_exit", 1), ("full_count", 0)]


def run_workload(a, dist, device, local, backend, force_dist, cpu_budget, headline=True):
    R = Run(a, dist, device, local, backend, force_dist)
    try:
        rates = []
        for mode, bound in rates_for(a):
            rates.append(R.measure(mode, bound))
        rates.sort(key=lambda r: r["mode"] != "full_count")   # (stable: full_count, the unit of `value`, first)
        return report(R, rates, cpu_budget, headline) if R.comm.rank == 0 else None
    finally:
        R.close()


LEGS = (("sphere", "iterative", "SphereParametersEstimator + RANSAC, 10 M points, geometric (LM) final fit "
         "(BASELINE.json configs[2]; on the driver's 8-GPU run the hypotheses are sharded, here 1 GPU)"),
        ("dense", "iterative", "DenseLinearEquationSystemParametersEstimator, m = 2 M, n = 64 (BASELINE.json configs[3])"),
        ("us", "iterative", "SinglePointTargetUSCalibrationParametersEstimator, 1 M frames, ITERATIVE (LM) final fit "
         "with the reference's tolerances (BASELINE.json configs[4] as written)"),
        ("us", "analytic", "SinglePointTargetUSCalibrationParametersEstimator, 1 M frames, ANALYTIC final fit"),
        ("phantom", "iterative", "PlanePhantomUSCalibrationParametersEstimator, 1 M frames, k = 31 (SURVEY 8(f): not a "
         "BASELINE config; the reference's examples/planeUSCalibration)"))


def run_legs(a0, local):
    """legs of BASELINE configs 3-5 (+ the US analytic fit and the plane phantom) on this GPU (the headline's 20 steps after 5 warm-up steps, one timed region each:
    with 5 steps after 2 the legs read 5-8 % below the same workloads' own runs), so that the driver's own run
    observes them"""
    legs = []
    for w, fit, title in LEGS:
        t0 = time.perf_counter()
        a = parse([])
        a.workload, a.us_fit = w, fit
        a.points = max(4096, int({"dense": 2_000_000, "us": 1_000_000, "phantom": 1_000_000}.get(w, 10_000_000)
                                 * a0.leg_scale))
        a.batch = 1024 if w == "dense" else 4096
        a.steps, a.warmup, a.repeats = 20, 5, 1
        a.streams = a0.streams
        if getattr(a0, "streams_auto", False) and w == "us" and fit == "iterative":
            a.streams, a.steps, a.warmup = 8, 24, 8      # eight host threads with a context each (see --streams)
        a.no_end_to_end = True
        a.no_cpu_baseline = a0.no_cpu_baseline
        try:
            o = run_workload(a, None, "cpu", local, "nccl", False, a0.cpu_seconds or 3.0, headline=False)
            keep = {k: o[k] for k in o if k in (
                "value", "unit", "steps", "warmup", "ms_per_step", "value_is", "value_full_count", "value_early_exit",
                "ms_per_step_full_count", "ms_per_step_early_exit", "repeats", "single_stream", "final_fit", "lm",
                "roofline", "roofline_early_exit", "kernel_hbm", "kernels_ms", "cpu_baseline",
                "speedup_vs_cpu_baseline")}
            keep["config"] = {"workload": title, "points": a.points, "hypotheses_per_gpu_per_step": a.batch,
                              "streams": o["config"]["streams"], "step": o["config"]["step"]}
            keep["leg_wall_s"] = time.perf_counter() - t0
            legs.append(keep)
        except Exception as e:      # a leg must not take the headline down with it; the failure is in the line
            legs.append({"config": {"workload": title}, "error": "%s: %s" % (type(e).__name__, e)})
    return legs



# ---------------------------------------------------------------------------------------------------------------------
# What is printed.  The driver keeps an 8 KB tail of stdout and parses the LAST line: that line is the compact
# headline (<= HEADLINE_LIMIT bytes, asserted); everything else -- work models, per-kernel tables, notes, the legs of
# the other configs in full -- goes to the detail file (--detail, default bench_detail.json) and, per leg, to one short
# line printed BEFORE the headline.
HEADLINE_LIMIT = 4096


def _sig(x, nd=6):
    """floats to nd significant digits (the headline carries numbers, not 17-digit reprs)"""
    if isinstance(x, float):
        return float("%.*g" % (nd, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, nd) for v in x]
    return x


def compact_roofline(r, one):
    """the roofline block of the headline: the contract's keys, the measured HBM fraction, the issue fractions, and the
    basis of launch_ms (the one-stream pass whose own rate stands beside it, so that launch_ms <= its ms_per_step)"""
    if not r:
        return None
    c = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    for k in ("hbm_frac_measured", "launch_ms", "issued_valu_frac_4cyc", "issued_valu_frac_2cyc",
              "frac_of_2cycle_issue_peak", "frac_of_fp64_mfma_peak", "frac_issued", "x_fp32_matrix_peak"):
        if r.get(k) is not None:
            c[k] = r[k]
    if r.get("counters"):
        c["valu_issue_busy"] = r["counters"].get("valu_issue_busy")
    if r.get("mfma_pipe_busy") is not None:
        c["mfma_pipe_busy"] = r["mfma_pipe_busy"]
    if r.get("timed_region"):
        c["frac_timed_region"] = r["timed_region"]["frac"]
    c["kernel"] = str(r.get("kernel_short") or r.get("kernel") or "")[:80]
    c["algorithmic_GBs"] = r.get("algorithmic_GBs")
    if one:
        c["basis"] = "one_stream pass: %.4g ms/step" % one["ms_per_step"]
    return c


def compact_cpu(cpu):
    if not cpu:
        return None
    c = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")}
    c["sample"] = str(cpu.get("sample", ""))[:150]
    if "full_count" in cpu:
        f = cpu["full_count"]
        c["full_count"] = {"value": f["value"], "kind": f["kind"], "cores": f["cores"], "sample": str(f["sample"])[:150]}
    return c


def leg_line(leg):
    """one short stdout line per leg of the other configs (printed before the headline)"""
    if "error" in leg:
        return {"leg": leg["config"]["workload"][:60], "error": leg["error"][:200]}
    ff = leg.get("final_fit", {})
    r = leg.get("roofline") or {}
    one = leg.get("single_stream")
    o = {"leg": leg["config"]["workload"][:100], "points": leg["config"]["points"],
         "hypotheses_per_step": leg["config"]["hypotheses_per_gpu_per_step"], "streams": leg["config"]["streams"],
         "value": leg["value"], "unit": "hypotheses/s", "ms_per_step": leg["ms_per_step"], "steps": leg["steps"],
         "value_full_count": leg.get("value_full_count"), "value_early_exit": leg.get("value_early_exit"),
         "one_stream": {"value": one["value"], "ms_per_step": one["ms_per_step"]} if one else None,
         "roofline": compact_roofline(r, one),
         "final_fit": {k: ff.get(k) for k in ("inliers", "winner_votes", "params_empty", "lm_info", "lm_nfev")},
         "cpu_full_count_hyp_s": (leg.get("cpu_baseline") or {}).get("full_count", leg.get("cpu_baseline") or {}).get("value")}
    if leg.get("lm"):
        o["lm_evaluations_per_s"] = leg["lm"].get("evaluations_per_s")
    return _sig(o, 5)


def headline(out, detail_path):
    """the LAST stdout line: the contract's keys + both rates + one-stream rate + roofline + cpu_baseline, <= 4 KB"""
    cfg = out["config"]
    one = out.get("single_stream")
    ff = out.get("final_fit") or {}
    h = {"metric": out["metric"], "value": out["value"], "unit": out["unit"], "n_gpus": out["n_gpus"],
         "steps": out["steps"], "warmup": out["warmup"], "ms_per_step": out["ms_per_step"],
         "higher_is_better": True, "scaling": out["scaling"], "vs_baseline": None, "dtype": out["dtype"],
         "data": out["data"],
         "config": {"workload": cfg["workload"][:140], "points": cfg["points"],
                    "hypotheses_per_gpu_per_step": cfg["hypotheses_per_gpu_per_step"], "streams": cfg.get("streams"),
                    "world_size": cfg.get("world_size"), "collectives": str(cfg.get("collectives"))[:120],
                    "rccl": cfg.get("rccl"), "repeats": cfg.get("repeats")},
         "value_is": out.get("value_is"),
         "value_full_count": out.get("value_full_count"), "value_early_exit": out.get("value_early_exit"),
         "ms_per_step_full_count": out.get("ms_per_step_full_count"),
         "ms_per_step_early_exit": out.get("ms_per_step_early_exit"),
         "one_stream": ({"value": one["value"], "ms_per_step": one["ms_per_step"], "steps": one["steps"]}
                        if one else None),
         "per_rank_hypotheses_per_s": out.get("per_rank_hypotheses_per_s"),
         "roofline": compact_roofline(out.get("roofline"), one),
         "cpu_baseline": compact_cpu(out.get("cpu_baseline")),
         "speedup_vs_cpu_baseline": out.get("speedup_vs_cpu_baseline"),
         "final_fit": {"inliers": ff.get("inliers"), "winner_votes": ff.get("winner_votes"),
                       "params_empty": ff.get("params_empty"), "lm_info": ff.get("lm_info"),
                       "lm_nfev": ff.get("lm_nfev"), "abs_dot_true_normal": ff.get("abs_dot_true_normal"),
                       "residual_min_max_mean_sumsq": ff.get("residual_min_max_mean_sumsq")},
         }
    if out.get("roofline_early_exit"):
        e = out["roofline_early_exit"]
        h["roofline_early_exit"] = {k: e.get(k) for k in ("bound", "frac", "launch_ms", "hbm_frac_measured")}
    if out.get("kernel_hbm"):
        h["kernel_hbm"] = [{"kernel": str(k["kernel"]).split(" ")[0][:40], "ms": k["ms"],
                            "frac_of_hbm_peak": k["frac_of_hbm_peak"]} for k in out["kernel_hbm"][:4]]
    if out.get("lm"):
        h["lm_evaluations_per_s"] = out["lm"].get("evaluations_per_s")
    if out.get("other_configs"):
        h["other_configs"] = [{"leg": (leg["config"]["workload"].split(",")[0])[:56],
                               "value": leg.get("value"), "ms_per_step": leg.get("ms_per_step"),
                               "roofline_frac": (leg.get("roofline") or {}).get("frac"),
                               **({"roofline_frac_issued": (leg.get("roofline") or {}).get("frac_issued")}
                                  if (leg.get("roofline") or {}).get("frac_issued") is not None else {}),
                               "error": leg.get("error")} for leg in out["other_configs"]]
    h["detail"] = detail_path
    h = _sig(h, 6)
    line = json.dumps(h, separators=(",", ":"))
    if len(line) > HEADLINE_LIMIT:      # never hand the driver a line it will cut: drop the optional blocks
        for k in ("other_configs", "kernel_hbm", "roofline_early_exit", "per_rank_hypotheses_per_s"):
            h.pop(k, None)
            line = json.dumps(h, separators=(",", ":"))
            if len(line) <= HEADLINE_LIMIT:
                break
    assert len(line) <= HEADLINE_LIMIT, len(line)
    return line


def emit(out, a):
    """detail file + one short line per leg + the headline as the LAST stdout line"""
    path = a.detail or os.path.join(ROOT, "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        shown = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError as e:            # a read-only tree must not cost the headline
        shown = "not written (%s)" % e
    for leg in out.get("other_configs") or []:
        print(json.dumps(leg_line(leg), separators=(",", ":")))
    print(headline(out, shown))
    sys.stdout.flush()


# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher environment: start the N ranks as CHILDREN (this process has not
    touched the GPU and never will), relay rank 0's JSON line, exit non-zero if any rank failed"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        elif ln.startswith('{"leg"'):
            sys.stdout.write(ln)
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc != 0 or line is None:
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d)\n" % (a.gpus, rc))
        return rc or 1
    print(line)
    return 0


def run_multi(a):
    """--transport multi: the N devices driven from this one process through lsqr_multi_* (peer copies)"""
    from lsqrrecipes_amd import _lib as L
    from lsqrrecipes_amd.context import MultiContext
    w = a.workload
    model = {"plane": L.PLANE, "sphere": L.SPHERE, "line": L.LINE, "dense": L.DENSE, "us": L.US_SINGLE}[w]
    ls_type = L.LS_ANALYTIC if (w == "us" and a.us_fit == "analytic") else L.LS_GEOMETRIC
    data, truth, _ = make_data(w, a.points, a.outliers)
    devs = [0] * a.gpus if os.environ.get("LSQR_SHARE_GPU") == "1" else list(range(a.gpus))
    H = a.batch
    seed = 0xC0FFEE
    with MultiContext(devs) as mc:
        mc.set_model(model, 64 if w == "dense" else 3, DELTA[w], ls_type).upload(data)
        out_rates = {}
        last = None
        step = 0
        for mode, bound in rates_for(a):
            mc.set_option("scan_bound", bound)
            for _ in range(a.warmup):
                mc.batch_fit(seed, step * a.gpus * H, H)
                step += 1
            vals = []
            for _ in range(max(1, a.repeats)):
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    last = mc.batch_fit(seed, step * a.gpus * H, H)
                    step += 1
                vals.append(H * a.gpus * a.steps / (time.perf_counter() - t0))
            out_rates[mode] = vals
    first = "full_count" if "full_count" in out_rates else rates_for(a)[0][0]   # (`value` counts everything when measured)
    value = float(np.median(out_rates[first]))
    f = last["info"].fit
    out = {"metric": METRIC, "value": value, "unit": "hypotheses/s", "n_gpus": a.gpus, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": H * a.gpus / value * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s estimator + RANSAC, %d observations" % (w, a.points), "points": a.points,
                      "hypotheses_per_gpu_per_step": H, "world_size": a.gpus, "devices": devs,
                      "collectives": "peer copies (lsqr_multi)", "streams": 1, "repeats": max(1, a.repeats),
                      "step": "lsqr_multi_batch_fit (one process, one context per device, blocking); `value` = value_%s"
                              % first},
           "value_is": first, "repeats": out_rates,
           "per_rank_hypotheses_per_s": [value / a.gpus] * a.gpus,
           "final_fit": {"inliers": int(f.n_used), "winner_votes": int(last["info"].best_votes),
                         "params": [float(x) for x in last["params"]], "params_empty": len(last["params"]) == 0,
                         "lm_info": int(f.lm_info), "lm_nfev": int(f.lm_nfev)},
           "roofline": None, "cpu_baseline": None}
    for mode, vals in out_rates.items():
        out["value_" + mode] = float(np.median(vals))
    emit(out, a)
    return 0


RCCL_INFO = {"bringup_s": None, "communicators": 0, "stream_groups_s": None, "fallback": None}


def bring_up_rccl(a, dist, torch, rank, world, device):
    """RCCL communicator proven (one small all-reduce) before any timing, with a COLLECTIVE verdict: every rank posts
    ok / fail to a side-channel TCPStore and all ranks act on the same tally.
      all ok                    -> ("nccl", device)
      all failed + --allow-gloo -> gloo over the host, labelled in config.collectives (LSQR_DIST_FALLBACK)
      anything else             -> exit code 3 on every rank (a rank still inside RCCL 20 s after a peer reported a
                                   failure is taken out by a watcher thread: no world split across two backends)"""
    import datetime
    import threading
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ["MASTER_PORT"])
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() in ("1", "true")
    store = dist.TCPStore(addr, port, world, is_master=(rank == 0 and not agent_store),
                          timeout=datetime.timedelta(seconds=300), wait_for_workers=False)
    side = dist.PrefixStore("lsqr_bench_rccl", store)
    done = threading.Event()

    def watch():
        seen = None
        while not done.is_set():
            try:
                failed = side.add("fail", 0)
            except Exception:  # noqa: BLE001 -- the store went away: the main thread will notice
                return
            if failed > 0:
                seen = seen or time.time()
                if time.time() - seen > 20.0 and not done.is_set():
                    sys.stderr.write("bench.py: rank %d still inside RCCL bring-up 20 s after a peer reported a "
                                     "failure -- exiting (3)\n" % rank)
                    sys.stderr.flush()
                    os._exit(3)
            time.sleep(0.25)
    threading.Thread(target=watch, daemon=True).start()
    err = None
    t_up = time.perf_counter()
    try:
        torch.cuda.set_device(torch.device(device))
        dist.init_process_group("nccl", store=dist.PrefixStore("nccl", store), rank=rank, world_size=world,
                                device_id=torch.device(device))
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        if int(probe.item()) != world:
            raise RuntimeError("all_reduce probe returned %r" % probe.item())
    except Exception as e:  # noqa: BLE001 -- whatever RCCL / the driver stack raises
        err = "%s: %s" % (type(e).__name__, str(e)[:160])
        sys.stderr.write("bench.py: RCCL unavailable on rank %d (%s)\n" % (rank, err))
    side.add("fail" if err else "ok", 1)
    done.set()
    t0 = time.time()
    try:
        while side.add("ok", 0) + side.add("fail", 0) < world:
            if time.time() - t0 > 120.0:
                raise TimeoutError("not every rank reported its RCCL status")
            time.sleep(0.05)
        failed = side.add("fail", 0)
    except Exception as e:  # noqa: BLE001 -- the store's host (rank 0) left, or a rank never reported
        sys.stderr.write("bench.py: rank %d: RCCL verdict incomplete (%s) -- exiting (3)\n" % (rank, type(e).__name__))
        sys.stderr.flush()
        os._exit(3)
    if failed == 0:
        RCCL_INFO["bringup_s"] = time.perf_counter() - t_up    # init_process_group + the probing all-reduce, this rank
        RCCL_INFO["communicators"] = 1
        return "nccl", device
    if failed == world and a.allow_gloo:
        # The exchanges of this path are 8 bytes and one moment block per step: they do not need RCCL to be fast.
        # Only on request, only when NO rank has RCCL, and labelled in config.collectives.
        sys.stderr.write("bench.py: rank %d continues over gloo (--allow-gloo)\n" % rank)
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
        os.environ["LSQR_DIST_FALLBACK"] = err
        dist.init_process_group("gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=180))
        return "gloo", "cpu"
    sys.stderr.write("bench.py: RCCL did not come up on %d of %d ranks%s -- no line is produced (exit 3)\n"
                     % (failed, world, "" if a.allow_gloo else "; --allow-gloo would continue over gloo if it had "
                        "failed on every rank"))
    sys.stderr.flush()
    os._exit(3)


def main():
    a = parse()
    if a.points <= 0:
        a.points = {"dense": 2_000_000, "us": 1_000_000, "phantom": 1_000_000}.get(a.workload, 10_000_000)
    if a.batch <= 0:
        a.batch = 1024 if a.workload == "dense" else 4096
    a.streams_auto = a.streams <= 0
    if a.streams_auto:
        a.streams = 8 if (a.workload == "us" and a.us_fit == "iterative" and a.gpus == 1) else 4
    in_launcher_env = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not in_launcher_env:
        # nothing in this process has touched the GPU yet (numpy and the standard library only)
        if a.transport == "multi":
            sys.exit(run_multi(a))
        sys.exit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world
    dist = None
    device = "cpu"
    force_dist = os.environ.get("LSQR_FORCE_DIST") == "1"  # exercise the RCCL path at world size 1
    # rehearsal knobs (CPU tests / one-GPU boxes): LSQR_DIST_BACKEND=gloo keeps the collectives on
    # the host, LSQR_SHARE_GPU=1 lets every rank use device 0.  The driver's runs use neither.
    backend = os.environ.get("LSQR_DIST_BACKEND", "nccl")
    if os.environ.get("LSQR_SHARE_GPU") == "1":
        local = 0
    if a.gpus > 1 or force_dist:
        import torch
        import torch.distributed as dist
        if force_dist and "RANK" not in os.environ:  # stand-alone world of one rank
            os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            device = "cuda:%d" % local
            backend, device = bring_up_rccl(a, dist, torch, rank, world if world > 1 else 1, device)
        else:
            dist.init_process_group(backend)
            if os.environ.get("LSQR_STEP") == "device":
                torch.cuda.init()      # torch's HIP runtime has to come up before the library's
                torch.cuda.set_device(local)
    out = run_workload(a, dist, device, local, backend, force_dist, a.cpu_seconds or 4.0, headline=True)
    if rank == 0:
        single_plain = a.gpus == 1 and dist is None
        if single_plain and a.workload == "plane" and not a.no_other_configs and a.rates == "both":
            out["other_configs"] = run_legs(a, local)
        emit(out, a)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
