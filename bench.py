#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

  metric   RANSAC hypotheses/s (+ final-fit residual), plane, 10 M points, 50 % outliers
  step     one pass of the hot path over one batch: sample H minimal subsets -> solve -> agree()
           scan of all H hypotheses over all N observations -> first-max winner -> consensus mask
           -> final least-squares fit.  Nothing is cached between steps (new subsets each step).
  value    whole-job hypotheses/s = H * n_gpus * steps / wall time (observations already resident
           in HBM when the timed region starts).
  N > 1    one process per GPU (torch.distributed, RCCL): observations replicated, the hypothesis
           stream sharded; all-reduce(MAX) picks the winner, all-reduce(SUM) of the moment block of
           each rank's observation slice gives the final fit.  scaling = "weak" (H per GPU fixed).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--points P] [--batch H] [--workload plane]
"""
import argparse
import json
import os
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
FP64_VALU_PEAK_GOPS = 39321.6  # 256 CU * 4 SIMD * 16 lanes/clk * 2.4 GHz: fp64 add/mul issue rate
#                                (= the 78.6 TFLOP/s vector fp64 peak counting an FMA as one op;
#                                the bit-exact agree() may not fuse, so this is its op roof)
FP64_VALU_MEASURED_GOPS = 33000.0  # tools/microbench.hip on this pool: v_add_f64 / v_mul_f64 with
#                                    every SIMD busy (4.8 nominal cycles per wave instruction)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
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
    ap.add_argument("--us-fit", default="analytic", choices=["iterative", "analytic"],
                    help="final fit of the US workload: the analytic estimate (default) or Levenberg-Marquardt "
                         "from it with the reference's 1e-15 tolerances -- at 1 M noisy frames MINPACK then "
                         "wanders inside rounding noise until its 5000-evaluation limit (90 us per evaluation: "
                         "0.45 s per step) and reports failure, exactly as the reference's settings make it")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="single GPU: one blocking lsqr_batch_fit per step instead of pipelined batches")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the adaptive RANSAC::compute() timing after the steps (profiling runs: "
                         "keeps one launch shape per kernel)")
    ap.add_argument("--cpu-points", type=int, default=0, help="observations for the CPU leg "
                    "(0 = same as --points)")
    return ap.parse_args()


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


# fp64 VALU instructions of the exact agree() per (hypothesis, observation) pair: arithmetic + compares
OPS_PER_PAIR = {"plane": 9, "sphere": 10, "line": 20, "dense": 130, "us": 69, "phantom": 64}


def cpu_baseline(workload, data, delta):
    """Reference single-thread CPU path on the same workload, bounded to ~10-30 s: the
    reference's own RANSAC.hxx (oracle/_ref, compiled from /root/reference in the build
    container) driving the restated estimator; falls back to the oracle's C port of the loop."""
    from oracle import pyoracle as O
    model = {"plane": O.PLANE, "sphere": O.SPHERE, "line": O.LINE, "dense": O.DENSE,
             "us": O.US_SINGLE, "phantom": O.PHANTOM}[workload]
    c = O.cfg(model, 64 if workload == "dense" else 3, delta, O.LS_ALGEBRAIC)
    cores = 1
    t0 = time.perf_counter()
    if workload in ("dense", "phantom"):
        # the adaptive bound never closes for k = 64 / hardly for k = 31 (w^k underflows), so the CPU leg is
        # a bounded sample of the metric's unit itself: minimal-subset solve + one full agree() pass
        n = len(data)
        k = 64 if workload == "dense" else 31
        subs = O.ctr_subsets(20261003, 0, 64 if workload == "dense" else 4096, n, k)
        hyp = 0
        while hyp < len(subs) and (hyp < 4 or time.perf_counter() - t0 < 12.0):
            par = O.estimate(c, data[subs[hyp]])
            if len(par):
                O.scan(c, par, data)
            hyp += 1
        dt = time.perf_counter() - t0
        return {"value": hyp / dt, "unit": "hypotheses/s", "cores": cores, "kind": "port",
                "sample": "oracle C port of %s: %d hypotheses "
                          "(%dx%d minimal solve + full agree() pass over N=%d records, no early exit) "
                          "in %.2f s; 1 thread" % ("DenseLinearEquationSystemParametersEstimator" if k == 64 else
                                                   "PlanePhantomUSCalibrationParametersEstimator", hyp, k, k, n, dt)}
    if O.ref_available():
        r = O.ref_ransac(c, data, 0.999, seed=20261003)
        hyp = r["estimate_calls"]
        kind = "reference"
        what = ("reference RANSAC.hxx compiled unmodified (oracle/_ref) + restated %sParametersEstimator"
                " (VNL absent), adaptive run p=0.999" % workload.capitalize())
    else:
        r = O.ransac(c, data, 0.999, sampler="ref", seed=20261003)
        hyp = int((r["status"] != 1).sum())
        kind = "port"
        what = "oracle C port of RANSAC.hxx + estimator, adaptive run p=0.999"
    dt = time.perf_counter() - t0
    return {"value": hyp / dt, "unit": "hypotheses/s", "cores": cores, "kind": kind,
            "sample": "%s; N=%d points; %d hypotheses in %.2f s; 1 thread" % (what, len(data), hyp, dt),
            "fraction": r["fraction"]}


def main():
    a = parse()
    if a.points <= 0:
        a.points = {"dense": 2_000_000, "us": 1_000_000, "phantom": 1_000_000}.get(a.workload, 10_000_000)
    if a.batch <= 0:
        a.batch = 1024 if a.workload == "dense" else 4096
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
            os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.cuda.set_device(local)
            device = "cuda:%d" % local
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
            if os.environ.get("LSQR_STEP") == "device":
                torch.cuda.init()      # torch's HIP runtime has to come up before the library's
                torch.cuda.set_device(local)
    from lsqrrecipes_amd import _lib as L
    from lsqrrecipes_amd.context import Context
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac

    delta = {"plane": 0.5, "sphere": 0.5, "line": 0.5, "dense": 0.1, "us": 3.0, "phantom": 2.0}[a.workload]
    model = {"plane": L.PLANE, "sphere": L.SPHERE, "line": L.LINE, "dense": L.DENSE,
             "us": L.US_SINGLE, "phantom": L.PHANTOM}[a.workload]
    # GEOMETRIC == ITERATIVE == 1: the sphere's geometric fit, the US calibrations' and the phantom's LM fit
    # (BASELINE.json configs[2], configs[4]); --us-fit analytic keeps the closed-form US fit
    ls_type = L.LS_ANALYTIC if (a.workload == "us" and a.us_fit == "analytic") else L.LS_GEOMETRIC
    data, truth, lab = make_data(a.workload, a.points, a.outliers)
    ctx = Context(local)
    ctx.set_model(model, 64 if a.workload == "dense" else 3, delta, ls_type).upload(data)
    if a.no_filter:
        ctx.set_option("scan_filter", 0)
    if a.no_index:
        ctx.set_option("scan_index", 0)
    comm = Comm(dist, device)
    eng = ShardedRansac(ctx, comm)
    step_on_device = dist is not None and os.environ.get(
        "LSQR_STEP", "device" if backend == "nccl" else "host") == "device"
    H = a.batch
    seed = 0xC0FFEE

    def step(i):
        if comm.world == 1 and not force_dist:
            # single GPU: the whole step is one chain on the device stream (lsqr_batch_fit)
            r = ctx.batch_fit(seed, i * H, H)
            if r["info"].best_votes == 0:
                return None
            return int(r["info"].best_votes), r["params"], int(r["info"].fit.n_used)
        # multi-GPU: exchanges on device buffers, one host synchronisation per step (step_device);
        # LSQR_STEP=host keeps the staged-through-the-host variant (the gloo rehearsal's default)
        r = eng.step_device(seed, i, H) if step_on_device else eng.step(seed, i, H)
        if r is None:
            return None
        votes, gidx, par, fit, cnt, info = r
        return votes, fit, cnt

    def sync():
        ctx.synchronize()
        if dist is not None and device != "cpu":
            import torch
            torch.cuda.synchronize()
        comm.barrier()

    # single GPU, closed-form fit: batches are pipelined -- batch i + 1 is enqueued before batch i is read
    # (lsqr_batch_fit_enqueue / _wait), so the host's latency between steps hides behind the device's work;
    # every step still runs the whole chain.  --no-pipeline keeps one blocking call per step.
    pipelined = (comm.world == 1 and not force_dist and not a.no_pipeline
                 and not (a.workload in ("sphere", "us") and ls_type == L.LS_GEOMETRIC)
                 and a.workload != "phantom")

    closed_form = not (a.workload in ("sphere", "us") and ls_type == L.LS_GEOMETRIC) and a.workload != "phantom"
    pipelined_dist = step_on_device and closed_form and not a.no_pipeline

    def run_steps(first_step, count):
        last = None
        if pipelined_dist:   # multi-GPU: step i + 1 is enqueued (collectives included) before step i is read
            for i in range(count):
                eng.step_device(seed, first_step + i, H, slot=i & 1)
                if i:
                    last = eng.step_device_wait((i - 1) & 1)
            if count:
                last = eng.step_device_wait((count - 1) & 1)
            if last is None:
                return None
            return last[0], last[3], last[4]
        if not pipelined:
            for i in range(count):
                last = step(first_step + i)
            return last
        for i in range(count):
            ctx.batch_fit_enqueue(seed, (first_step + i) * H, H, slot=i & 1)
            if i:
                last = ctx.batch_fit_wait((i - 1) & 1)
        if count:
            last = ctx.batch_fit_wait((count - 1) & 1)
        if last is None or last["info"].best_votes == 0:
            return None
        return int(last["info"].best_votes), last["params"], int(last["info"].fit.n_used)

    ctx.profile(True)   # the spatial index is built inside the first large scan of an upload
    run_steps(0, a.warmup)
    n_idx, ms_idx = ctx.profile_get("index")
    ctx.profile(True)
    sync()
    t0 = time.perf_counter()
    last = run_steps(a.warmup, a.steps)
    sync()
    dt = time.perf_counter() - t0
    dt = comm.allreduce_max_f64(dt)
    n_scan, ms_scan = ctx.profile_get("scan")
    n_mask, ms_mask = ctx.profile_get("mask")
    n_mom, ms_mom = ctx.profile_get("moments")
    n_est, ms_est = ctx.profile_get("estimate")
    n_sol, ms_sol = ctx.profile_get("solve")
    n_smp, ms_smp = ctx.profile_get("sample")
    n_idx2, ms_idx2 = ctx.profile_get("index")
    ctx.profile(False)
    idx = ctx.index_info()

    if rank == 0:
        total_hyp = H * a.gpus * a.steps
        value = total_hyp / dt
        votes, fit, cnt = last
        res = ctx.stats(fit, use_mask=True) if (comm.world == 1 and not force_dist) else None
        rec = data.shape[1] * 8
        scan_ms = ms_scan / max(n_scan, 1)
        alg_bytes = float(H) * a.points * rec          # N*sizeof(T) per hypothesis
        achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
        pairs_per_s = float(H) * a.points / (scan_ms * 1e-3) if scan_ms > 0 else 0.0
        filtered = not a.no_filter
        if idx["built"] and not a.no_filter:
            kname = ("k_scan_cells<%s> (two-level: fp32 cell-box culling over a Morton-sorted copy, "
                     "packed fp32 filter + exact fp64 re-check in surviving cells)" % a.workload)
        else:
            kname = {"plane": "k_scan_f32<plane> (fp32 pre-filter + exact fp64 re-check)",
                     "sphere": "k_scan_f32<sphere> (fp32 pre-filter + exact fp64 re-check)",
                     "line": "k_scan_f32<line> (fp32 pre-filter + exact fp64 re-check)",
                     "us": "k_scan_us_f32<us> (packed fp32 pre-filter + exact fp64 re-check)",
                     "phantom": "k_scan_us_f32<phantom> (factored packed fp32 pre-filter + exact fp64 re-check)",
                     "dense": "k_scan_dense_mfma2 (fp64 MFMA filter + exact re-check worklist)"}.get(
                         a.workload)
        eq_gops = pairs_per_s * OPS_PER_PAIR[a.workload] / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tfile):
            try:
                t = json.load(open(tfile))
                key = "%s_%d_%d%s" % (a.workload, a.points, H, "_cells" if idx["built"] else "")
                traffic = t.get(key)
            except Exception:
                traffic = None
        issue = None   # measured instruction-issue utilisation of the scan kernel (profiles/, SQ counters)
        sfile = os.path.join(ROOT, "profiles", "r01h_plane_scan_sq_counters.json")
        if a.workload == "plane" and a.points == 10_000_000 and H == 4096 and idx["built"] and os.path.exists(sfile):
            try:
                d = json.load(open(sfile))["derived"]
                issue = {"valu_issue_busy": d["valu_issue_busy"], "salu_issue_busy_per_cu": d["salu_issue_busy_per_cu"],
                         "lanes_active": d["lanes_active"],
                         "source": "profiles/r01h_plane_scan_sq_counters.json (rocprofv3 --pmc SQ_* passes of this kernel "
                                   "and shape; not collected live)"}
            except Exception:
                issue = None
        out = {
            "metric": METRIC,
            "value": value, "unit": "hypotheses/s", "n_gpus": a.gpus, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%sParametersEstimator + RANSAC, %d points, %d%% outliers, "
                                   "delta=%.2f (BASELINE.json configs[1])" % (
                                       a.workload.capitalize(), a.points, round(a.outliers * 100), delta)
                       if a.workload == "plane" else "%s estimator + RANSAC, %d observations" % (
                           a.workload, a.points),
                       "points": a.points, "hypotheses_per_gpu_per_step": H,
                       "record_bytes": rec, "parallelism": "hypotheses sharded over %d GPU(s), "
                       "observations replicated" % a.gpus,
                       "step": ("lsqr_batch_fit_enqueue/_wait (one chain per step, next step enqueued before "
                                "the previous one is read)" if pipelined
                                else "lsqr_batch_fit (one chain, one sync)" if comm.world == 1 and not force_dist
                                else "step_device, pipelined (collectives on device buffers; step i + 1 enqueued "
                                "before step i is read)" if pipelined_dist
                                else "step_device (collectives on device buffers, one sync)" if step_on_device
                                else "step (exchanges staged through the host)")},
            "final_fit": {"inliers": int(cnt), "winner_votes": int(votes),
                          "params": [float(x) for x in fit],
                          "abs_dot_true_normal": float(abs(np.dot(fit[:3], truth[:3])))
                          if a.workload in ("plane", "line") else None,
                          "residual_min_max_mean_sumsq": [float(x) for x in res] if res is not None else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kname if filtered else "k_scan<%s> (exact fp64)" % a.workload,
                         "launch_ms": scan_ms, "launches": int(n_scan),
                         "note": "achieved = algorithmic bytes (H*N*%d B per launch, SURVEY 8d) / "
                                 "launch time; the batched scan reads the observations once per "
                                 "launch for all H hypotheses (and the two-level scan proves most "
                                 "(hypothesis, cell) pairs irrelevant without touching their "
                                 "observations), so it is bound by VALU/SALU issue, not HBM: see valu "
                                 "and traffic (measured HBM bytes per launch)" % rec,
                         "valu": {"pairs_per_s": pairs_per_s,
                                  "exact_fp64_ops_per_pair": OPS_PER_PAIR[a.workload],
                                  "exact_equivalent_gops": eq_gops,
                                  "fp64_issue_peak_gops": FP64_VALU_PEAK_GOPS,
                                  "fp64_issue_measured_gops": FP64_VALU_MEASURED_GOPS,
                                  "frac_of_peak": eq_gops / FP64_VALU_PEAK_GOPS,
                                  "frac_of_measured_issue_rate": eq_gops / FP64_VALU_MEASURED_GOPS,
                                  "issue_utilisation": issue,
                                  "note": ("culling and the pre-filter decide most pairs without "
                                           "the exact fp64 formula, so the exact-equivalent rate "
                                           "exceeds the fp64 issue roof") if filtered else
                                          "exact fp64 path: fraction of the fp64 add/mul issue rate"}},
            "kernels_ms": {"sample": ms_smp / max(n_smp, 1), "estimate": ms_est / max(n_est, 1),
                           "scan": scan_ms,
                           "mask": ms_mask / max(n_mask, 1), "moments": ms_mom / max(n_mom, 1),
                           "reduce_and_solve_per_step": ms_sol / max(a.steps, 1)},
            "index": {"built": idx["built"], "cells": idx["cells"], "cell_points": idx["cell_points"],
                      "build_ms": (ms_idx + ms_idx2) / max(n_idx + n_idx2, 1) if (n_idx + n_idx2) else None,
                      "builds_in_warmup": int(n_idx), "builds_in_timed_region": int(n_idx2),
                      "note": "one-time per upload (device counting sort on Morton keys + cell boxes); "
                              "built inside the first scan once the upload has seen >= 2048 hypotheses"},
        }
        if a.gpus == 1 and dist is None and a.workload in ("plane", "sphere", "line") and not a.no_end_to_end:
            ctx.set_option("max_iterations", 100000)
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
                "note": "RANSAC<T,S>::compute(): batches of 256/1024/4096 hypotheses + serial replay + "
                        "mask + final fit, observations resident"}
        if a.gpus == 1 and not a.no_cpu_baseline:
            cp = a.cpu_points or a.points
            out["cpu_baseline"] = cpu_baseline(a.workload, data[:cp], delta)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
