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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
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
    ap.add_argument("--streams", type=int, default=4,
                    help="single GPU, pipelined steps: HIP streams (lanes of lsqr_batch_fit_enqueue) the batches "
                         "alternate over; batches of different streams overlap on the device (1 = one stream)")
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

# ---- roofs (MI355X_MICROARCH.md): 256 CUs x 4 SIMDs, 2.4 GHz; a wave64 VALU instruction occupies its 16-lane
# SIMD for 4 cycles (packed fp32: two results per lane in the same slot), so the chip issues at most
# 1024 * 2.4e9 / 4 = 614.4 G wave-instructions/s; fp64 MFMA dense peak 78.6 TFLOP/s.
VALU_ISSUE_PEAK_GWIPS = 1024 * 2.4 / 4.0
FP64_MFMA_PEAK_TFLOPS = 78.6
FP32_MFMA_PEAK_TFLOPS = 157.3
# USEFUL vector instructions of the scan kernels, counted from the source (csrc/cells.h, models.h, us.h; table in
# DESIGN.md section 6): l1 = arithmetic of CM::level1 per (64-hypothesis group, cell); pk = packed-fp32
# instructions of the filter measure per packed pair of observations per lane (128 observations per wave).
# Per surviving (hypothesis, cell) of 128*PP observations the useful count is PP * (pk + 1 |.|-min) + 2
# (min3 combine + the candidate compare); everything else the kernel issues (v_readlane broadcasts, the
# two-threshold ballots, bookkeeping, exact re-checks) is overhead against this roof.
SCAN_USEFUL = {"plane": {"l1": 15, "pk": 3}, "sphere": {"l1": 61, "pk": 6}, "line": {"l1": 35, "pk": 12},
               "us": {"pk": 21}, "phantom": {"pk": 18}}


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
            "fraction": r["fraction"], "compute_call_s": dt, "iterations": int(r.get("iters", hyp))}


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
    a.streams = max(1, min(8, a.streams))      # the library's lanes go up to 4; host-threaded contexts up to 8
    ctx.set_option("batch_lanes", min(4, a.streams))
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

    if pipelined or pipelined_dist:
        a.streams = min(4, a.streams)
    cur_streams = [a.streams]

    # iterative final fits (sphere geometric, US iterative, phantom) keep the host in the loop -- MINPACK's control
    # flow between device passes -- so a batch is a blocking call.  The library's threading model is one context per
    # host thread (LsqrDevice.h); with several streams the steps are dealt to that many host threads, each driving
    # its own context (own stream, own upload): while one thread waits for an evaluation, another one's scan runs.
    threaded = (comm.world == 1 and not force_dist and not pipelined and a.streams > 1 and not a.no_pipeline)
    tctx = [ctx]
    if threaded:
        for k in range(1, a.streams):
            ck = Context(local)
            ck.set_model(model, 64 if a.workload == "dense" else 3, delta, ls_type).upload(data)
            if a.no_filter:
                ck.set_option("scan_filter", 0)
            if a.no_index:
                ck.set_option("scan_index", 0)
            tctx.append(ck)

    # multi-GPU with several streams: one engine per stream -- its own context (own upload and index), its own
    # process group (an RCCL communicator serves one stream at a time) and its own torch stream
    lanes = []
    if pipelined_dist and a.streams > 1 and str(device) != "cpu":
        import torch
        for k in range(a.streams):
            ck = ctx if k == 0 else Context(local)
            if k:
                ck.set_model(model, 64 if a.workload == "dense" else 3, delta, ls_type).upload(data)
                if a.no_filter:
                    ck.set_option("scan_filter", 0)
                if a.no_index:
                    ck.set_option("scan_index", 0)
            gk = dist.new_group(backend="nccl")        # every rank, same order
            lanes.append((ShardedRansac(ck, Comm(dist, device, group=gk)), torch.cuda.Stream()))

    def run_steps(first_step, count):
        last = None
        if pipelined_dist and lanes and cur_streams[0] > 1:
            import torch
            S = len(lanes)
            ring = 2 * S
            for i in range(count):
                if i >= ring:
                    j = i - ring
                    last = lanes[j % S][0].step_device_wait((j // S) & 1)
                with torch.cuda.stream(lanes[i % S][1]):
                    lanes[i % S][0].step_device(seed, first_step + i, H, slot=(i // S) & 1)
            for j in range(max(0, count - ring), count):
                last = lanes[j % S][0].step_device_wait((j // S) & 1)
            if last is None:
                return None
            return last[0], last[3], last[4]
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
        if threaded and cur_streams[0] > 1:
            import threading
            S = len(tctx)
            res = [None] * count
            err = []

            def work(k):
                try:
                    for i in range(k, count, S):
                        r = tctx[k].batch_fit(seed, (first_step + i) * H, H)
                        res[i] = (None if r["info"].best_votes == 0 else
                                  (int(r["info"].best_votes), r["params"], int(r["info"].fit.n_used)))
                except Exception as e:      # surfaced below: a failed step must fail the run
                    err.append(e)
            th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            if err:
                raise err[0]
            return res[-1] if count else None
        if not pipelined:
            for i in range(count):
                last = step(first_step + i)
            return last
        # a ring of 2 * streams slots: slot s runs on stream (lane) s % streams, two batches deep per stream
        ring = 2 * cur_streams[0]
        for i in range(count):
            if i >= ring:
                last = ctx.batch_fit_wait((i - ring) % ring)
            ctx.batch_fit_enqueue(seed, (first_step + i) * H, H, slot=i % ring)
        for i in range(max(0, count - ring), count):
            last = ctx.batch_fit_wait(i % ring)
        if last is None or last["info"].best_votes == 0:
            return None
        return int(last["info"].best_votes), last["params"], int(last["info"].fit.n_used)

    ctx.profile(True)   # the spatial index is built inside the first large scan of an upload
    multi_stream = a.streams > 1 and (pipelined or threaded or (pipelined_dist and bool(lanes)))
    if multi_stream:
        run_steps(0, 2 * a.streams)   # every stream builds its index / loads its code objects before the W steps
    run_steps(0, a.warmup)
    n_idx, ms_idx = ctx.profile_get("index")
    n_abs, ms_abs = ctx.profile_get("absmax")
    ctx.profile(True)
    sync()
    t0 = time.perf_counter()
    last = run_steps(a.warmup, a.steps)
    sync()
    dt = time.perf_counter() - t0
    dt = comm.allreduce_max_f64(dt)
    prof = {k: ctx.profile_get(k) for k in ("scan", "mask", "moments", "estimate", "solve", "sample", "index")}
    ctx.profile(False)
    single_stream = None
    if multi_stream:
        # Kernel durations measured while batches of several streams share the device overlap each other; the
        # per-kernel figures (roofline, kernel_hbm, kernels_ms) come from the SAME steps run once more on one
        # stream, right after the timed region.  `value` is the multi-stream figure of the timed region above.
        if pipelined:
            ctx.set_option("batch_lanes", 1)
        cur_streams[0] = 1
        k1 = min(a.steps, 20)
        run_steps(a.warmup + a.steps, 2)
        ctx.profile(True)
        sync()
        t1 = time.perf_counter()
        run_steps(a.warmup + a.steps + 2, k1)
        sync()
        dt1 = comm.allreduce_max_f64(time.perf_counter() - t1)
        prof1 = {k: ctx.profile_get(k) for k in ("scan", "mask", "moments", "estimate", "solve", "sample", "index")}
        ctx.profile(False)
        if pipelined:
            ctx.set_option("batch_lanes", min(4, a.streams))
        cur_streams[0] = a.streams
        single_stream = {"steps": k1, "ms_per_step": dt1 / k1 * 1e3, "value": H * a.gpus * k1 / dt1,
                         "scan_ms_in_timed_region": prof["scan"][1] / max(prof["scan"][0], 1),
                         "note": "the same chain on ONE stream, run right after the timed region: the source of the "
                                 "per-kernel durations in roofline / kernel_hbm / kernels_ms (in the timed region "
                                 "the kernels of %d streams overlap, so a kernel's own duration there includes the "
                                 "time it shares the device)" % a.streams}
        prof1["index"] = prof["index"]
        prof = prof1
    idx = ctx.index_info()

    if rank == 0:
        out = report(a, ctx, comm, data, truth, delta, ls_type, H, dt, last, prof, idx,
                     (n_idx, ms_idx), (n_abs, ms_abs), pipelined, pipelined_dist, step_on_device, force_dist,
                     dist, model, single_stream)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


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


def scan_roofline(a, ctx, idx, H, scan_ms, n_scan, rec):
    """the dominant kernel against the roof that binds it"""
    w = a.workload
    t = scan_ms * 1e-3
    alg_bytes = float(H) * a.points * rec            # N*sizeof(T) per hypothesis, H per launch
    base = {"launch_ms": scan_ms, "launches": int(n_scan),
            "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_GBs": alg_bytes / t / 1e9 if t > 0 else 0.0,
            "algorithmic_note": "SURVEY 8(d) logical figure H*N*%d B / launch time: NOT a roofline fraction -- one pass "
                                "over the observations serves all H hypotheses, so it exceeds the HBM peak by "
                                "construction" % rec}
    prof = profile_file("r02_%s_scan_counters.json" % w)
    traffic = prof.get("hbm_bytes_per_launch") if prof else None
    if prof:
        base["counters"] = {k: prof[k] for k in ("valu_issue_busy", "salu_issue_busy_per_cu", "lanes_active",
                                                "valu_wave_instructions", "salu_wave_instructions",
                                                "kernel_avg_ms", "collected_at", "source") if k in prof}
        if traffic and t > 0:
            base["hbm_frac_measured"] = traffic / t / 1e9 / HBM_PEAK_GBS
    if w == "dense":
        flops = 2.0 * a.points * 64 * H              # the residual block rows x hypotheses as a GEMM
        ach = flops / t / 1e12 if t > 0 else 0.0
        base.update({"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "frac_of_fp64_mfma_peak": ach / FP64_MFMA_PEAK_TFLOPS,
                     "kernel": "k_scan_dense_mfma32<64> (fp32 MFMA filter, v_mfma_f32_16x16x4_f32, A fragments in "
                               "registers, two accumulator sets) + k_dense_recheck_seg (exact fp64 decision of the "
                               "~1e-4 of the pairs inside the filter's band); launch_ms covers both and the thresholds",
                     "note": "flops = 2*m*n*H of the filter GEMM per launch; peak = the fp32 dense matrix rate the "
                             "filter runs at (the fp64 MFMA filter of round 1 reached 0.65 of the fp64 rate = 51 TFLOP/s)"})
        return base
    u = SCAN_USEFUL[w]
    cells = w in ("plane", "sphere", "line") and idx["built"] and not a.no_filter
    if cells:
        wl = ctx.scan_workload()                     # level 1 alone on the last batch: live counts
        pp = wl["cell_points"] // 128
        v2 = pp * (u["pk"] + 1) + 2
        groups = -(-H // 64)
        l1 = wl["level1_evaluations"]                # one level-1 pass over all hypotheses
        if wl["bounded"]:                            # bounds pass + the pilots' group + the second pass' groups
            l1 += wl["cells"] * (1 + -(-wl["second_pass"] // 64))
        useful = l1 * u["l1"] + wl["pairs_counted"] * v2
        kname = ("two-level scan of <%s> over a Morton-sorted copy: cell-box culling, packed fp32 filter + exact fp64 "
                 "re-check in surviving cells; %s" % (
                     w, "bounded: k_cells_bounds (vote bounds) -> pilots -> only hypotheses that can still win, each "
                     "counted by k_scan_pairs (level 1 counted first, then an equal share of the surviving "
                     "(hypothesis, cell) pairs per wave)" if wl["bounded"] else "k_scan_cells"))
        model = {"level1_evaluations": l1, "level1_useful_instr": u["l1"],
                 "surviving_hypothesis_cell_pairs_all": wl["pairs"],
                 "surviving_hypothesis_cell_pairs_counted": wl["pairs_counted"], "level2_useful_instr": v2,
                 "cells": wl["cells"], "cell_points": wl["cell_points"], "hypothesis_groups": groups,
                 "bounded_scan": wl["bounded"], "pilots": wl["pilots"], "second_pass_hypotheses": wl["second_pass"],
                 "surviving_fraction": wl["pairs"] / max(1.0, float(wl["cells"]) * H)}
    else:
        pairs = float(H) * a.points / 128.0          # every (hypothesis, packed pair of observations per wave)
        v2 = u["pk"] + 2
        useful = pairs * v2
        kname = {"us": "k_scan_us_f32<us> (packed fp32 filter + exact fp64 re-check)",
                 "phantom": "k_scan_us_f32<phantom> (factored packed fp32 filter + exact fp64 re-check)"}.get(
                     w, "k_scan_f32<%s> (packed fp32 filter + exact fp64 re-check)" % w)
        model = {"hypothesis_wave_pairs": pairs, "useful_instr_per_pair": v2}
    ach = useful / t / 1e9 if t > 0 else 0.0
    base.update({"bound": "valu", "achieved": ach, "peak": VALU_ISSUE_PEAK_GWIPS, "unit": "G wave-instr/s",
                 "frac": ach / VALU_ISSUE_PEAK_GWIPS, "traffic": traffic, "kernel": kname, "work_model": model,
                 "note": "achieved = USEFUL vector instructions of the launch (counted from the kernel source, "
                         "work counts measured live by lsqr_scan_workload) / launch time; peak = 1024 SIMDs x "
                         "2.4 GHz / 4 cycles per wave64 instruction.  The kernel reads the observations once "
                         "per launch for all H hypotheses (HBM fraction in hbm_frac_measured), so the "
                         "instruction-issue roof is the one that binds; counters.valu_issue_busy is the "
                         "measured utilisation including overhead instructions"})
    return base


def hbm_table(a, H, rec, prof, abs_prof, idx_prof, cnt):
    """HBM-bound kernels of the step: algorithmic bytes / HIP-event time / 8 TB/s (live)"""
    n = float(a.points)
    rows = []

    def add(name, byts, nl, ms, note=""):
        if nl and ms > 0:
            t = ms / nl
            rows.append({"kernel": name, "bytes": byts, "ms": t, "GBs": byts / t / 1e6,
                         "frac_of_hbm_peak": byts / t / 1e6 / HBM_PEAK_GBS, "launches": int(nl), "note": note})
    w = a.workload
    if w == "dense":
        add("k_mask_dense (winner's consensus mask)", n * rec + n, *prof["mask"])
        add("k_syrk_mfma (A^T A | A^T b over the consensus set)", n * rec + n, *prof["moments"],
            note="also 2*m*65*66/2*... fp64 MFMA flops: see DESIGN.md")
    elif w == "phantom":
        add("k_mask<phantom>", n * rec + n, *prof["mask"])
        add("k_phantom_rows / k_syrk_mfma (Gram block)", n * 256 + n, *prof["moments"])
    else:
        add("k_mask_moments<%s> (consensus mask + moment block, one pass)" % w, n * rec + n, *prof["mask"])
        if prof["moments"][0]:
            add("k_moments<%s, LM> (one Levenberg-Marquardt evaluation: sum f^2, J^T J, J^T f)" % w,
                n + cnt * rec, *prof["moments"],
                note="reads the mask (N B) and the %d consensus records" % cnt)
    add("k_bounds (point models: min / max / max |x| in one pass) or k_absmax, once per upload", n * rec, *abs_prof)
    if idx_prof[0]:
        add("spatial index build (k_keys, radix sort of (key, index) pairs, k_gather_boxes; once per upload)",
            n * (3 * rec + 60), *idx_prof,
            note="bytes = records read for the keys, (key, index) pairs written and moved by 3 radix passes, "
                 "records gathered and the sorted copy written; the bounds come from the k_bounds pass above")
    return rows


def closed_form_fit(a, ls_type):
    from lsqrrecipes_amd import _lib as L
    return not (a.workload in ("sphere", "us") and ls_type == L.LS_GEOMETRIC) and a.workload != "phantom"


def cold_call(a, L, Context, data, model, delta, ls_type, cpu):
    """What a caller of RANSAC<T,S>::compute() (RANSAC.h:75-79) sees on data that is NOT yet on the device:
    lsqr_upload of the caller's pageable buffer + lsqr_ransac (adaptive, p = 0.999) + the consensus copy."""
    res = {}
    for label, threads in (("plain_hipMemcpy", 0), ("staged_upload_4_threads", 4)):
        c2 = Context(0)
        try:
            c2.set_model(model, 64 if a.workload == "dense" else 3, delta, ls_type)
            c2.set_option("upload_threads", threads)
            c2.set_option("max_iterations", 100000)
            best = None
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
            res[label] = best
        finally:
            c2.close()
    out = {"what": "lsqr_upload (host, pageable) + lsqr_ransac (p = 0.999) + consensus copy, %d records of %d B; "
                   "best of 2 calls after the first" % (len(data), data.shape[1] * 8),
           "ms": res["plain_hipMemcpy"]["total_ms"], "detail": res}
    if cpu and "compute_call_s" in cpu:
        out["reference_cpu_call_s"] = cpu["compute_call_s"]
        out["speedup_vs_reference_call"] = cpu["compute_call_s"] * 1e3 / out["ms"]
        out["note"] = ("like-for-like: one RANSAC<T,S>::compute() on the same %d records; the CPU figure is the "
                       "reference's own RANSAC.hxx run of cpu_baseline (its libc rand() subset stream, %d "
                       "iterations)" % (len(data), cpu.get("iterations", 0)))
    return out


def report(a, ctx, comm, data, truth, delta, ls_type, H, dt, last, prof, idx, idx_warm, abs_prof, pipelined,
           pipelined_dist, step_on_device, force_dist, dist, model, single_stream=None):
    from lsqrrecipes_amd import _lib as L
    from lsqrrecipes_amd.context import Context
    total_hyp = H * a.gpus * a.steps
    value = total_hyp / dt
    votes, fit, cnt = last
    single = comm.world == 1 and not force_dist
    res = ctx.stats(fit, use_mask=True) if single else None
    rec = data.shape[1] * 8
    n_scan, ms_scan = prof["scan"]
    scan_ms = ms_scan / max(n_scan, 1)
    if not single and a.workload in ("plane", "sphere", "line") and closed_form_fit(a, ls_type):
        # the multi-GPU step leaves the re-derived winner (a batch of one) as the context's current batch: the work
        # model of the roofline is taken on a full batch of this rank, as in the single-GPU run (untimed)
        ctx.batch_fit(0xC0FFEE, comm.rank * H, H)
    n_idx = idx_warm[0] + prof["index"][0]
    ms_idx = idx_warm[1] + prof["index"][1]
    out = {
        "metric": METRIC,
        "value": value, "unit": "hypotheses/s", "n_gpus": a.gpus, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%sParametersEstimator + RANSAC, %d points, %d%% outliers, "
                               "delta=%.2f (BASELINE.json configs[1])" % (
                                   a.workload.capitalize(), a.points, round(a.outliers * 100), delta)
                   if a.workload == "plane" else "%s estimator + RANSAC, %d observations%s" % (
                       a.workload, a.points,
                       ", final fit: %s" % a.us_fit if a.workload == "us" else ""),
                   "points": a.points, "hypotheses_per_gpu_per_step": H,
                   "record_bytes": rec, "parallelism": "hypotheses sharded over %d GPU(s), "
                   "observations replicated" % a.gpus,
                   "world_size": comm.world,
                   "collectives": ("RCCL (torch.distributed nccl backend)" if dist is not None and comm.device != "cpu"
                                   else ("gloo (rehearsal)" if dist is not None else "none (single GPU)")),
                   "streams": a.streams if single_stream is not None else 1,
                   "stream_priming_steps": 2 * a.streams if single_stream is not None else 0,
                   "step": ("lsqr_batch_fit_enqueue/_wait (one chain per step; the steps alternate over %d HIP "
                            "stream(s), two deep per stream, so chains of different streams overlap on the "
                            "device)" % a.streams if pipelined
                            else "lsqr_batch_fit (blocking: the iterative fit keeps the host in the loop), the steps "
                            "dealt to %d host threads with one context (stream) each" % a.streams
                            if (single and single_stream is not None)
                            else "lsqr_batch_fit (one chain, one sync)" if single
                            else "step_device, pipelined (collectives on device buffers; step i + 1 enqueued "
                            "before step i is read)" if pipelined_dist
                            else "step_device (collectives on device buffers, one sync)" if step_on_device
                            else "step (exchanges staged through the host)")},
        "per_rank_hypotheses_per_s": value / a.gpus,
        "single_stream": single_stream,
        "final_fit": {"inliers": int(cnt), "winner_votes": int(votes),
                      "params": [float(x) for x in fit],
                      "abs_dot_true_normal": float(abs(np.dot(fit[:3], truth[:3])))
                      if a.workload in ("plane", "line") else None,
                      "residual_min_max_mean_sumsq": [float(x) for x in res] if res is not None else None},
        "roofline": scan_roofline(a, ctx, idx, H, scan_ms, n_scan, rec),
        "kernel_hbm": hbm_table(a, H, rec, prof, abs_prof, (n_idx, ms_idx), int(cnt)),
        "kernels_ms": {"sample": prof["sample"][1] / max(prof["sample"][0], 1),
                       "estimate": prof["estimate"][1] / max(prof["estimate"][0], 1),
                       "scan": scan_ms,
                       "mask": prof["mask"][1] / max(prof["mask"][0], 1),
                       "moments": prof["moments"][1] / max(prof["moments"][0], 1),
                       "moments_launches_per_step": prof["moments"][0] / max(a.steps, 1),
                       "reduce_and_solve_per_step": prof["solve"][1] / max(a.steps, 1)},
        "index": {"built": idx["built"], "cells": idx["cells"], "cell_points": idx["cell_points"],
                  "build_ms": ms_idx / n_idx if n_idx else None,
                  "builds_in_warmup": int(idx_warm[0]), "builds_in_timed_region": int(prof["index"][0]),
                  "note": "one-time per upload (device counting sort on Morton keys + cell boxes), built inside "
                          "the scan that first needs it"},
    }
    if a.gpus == 1 and dist is None and a.workload in ("plane", "sphere", "line") and not a.no_end_to_end:
        # a second build in the same process (code objects loaded, buffers kept): the build's own device time
        ctx.profile(True)
        ctx.upload(data)
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
        cpu = cpu_baseline(a.workload, data[:cp], delta)
        out["cpu_baseline"] = cpu
        out["cpu_baseline"]["unit_note"] = (
            "hypotheses/s of the reference's serial loop (estimate + agree pass WITH its early exit, "
            "RANSAC.hxx:94); `value` counts full agree passes, so the ratio below overstates the "
            "like-for-like gain -- cold_call.speedup_vs_reference_call is the end-to-end comparison")
        out["speedup_vs_cpu_baseline"] = value / cpu["value"]
    if a.gpus == 1 and dist is None and not a.no_end_to_end and a.workload in ("plane", "sphere", "line", "us"):
        out["cold_call"] = cold_call(a, L, Context, data, model, delta, ls_type, cpu)
    return out


if __name__ == "__main__":
    main()
