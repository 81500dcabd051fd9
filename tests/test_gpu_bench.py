"""bench.py end to end on a small workload (run with -m gpu): the JSON contract of the line it prints, on every driver
of the steps -- lanes of one context (closed-form fits), host threads with a context each (iterative fits), one
stream, the RCCL path at world size 1 with one engine, process group and torch stream per stream, the self-launched
two-rank run (no launcher environment) and the one-process multi-device transport."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                "TORCHELASTIC_RUN_ID")


def _run(args, env=None, extra=("--no-cpu-baseline", "--no-end-to-end", "--no-other-configs"), rc=0, points="200000"):
    """-> the FULL record (the --detail file), after checking the stdout contract: the LAST line is the compact
    headline (< 4 KB, the driver parses it), any earlier JSON lines are the short per-leg lines, the whole stdout fits
    the driver's 8 KB tail, and the headline's numbers are the detail's"""
    e = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV}
    if env:
        e.update(env)
    with tempfile.TemporaryDirectory() as td:
        detail = os.path.join(td, "detail.json")
        r = subprocess.run([sys.executable, "bench.py", "--points", points, "--steps", "6", "--warmup", "2",
                            "--repeats", "2", "--detail", detail] + list(extra) + args, cwd=ROOT, env=e,
                           capture_output=True, text=True, timeout=900)
        if rc:
            assert r.returncode != 0, (r.returncode, r.stderr[-3000:])     # (torchrun maps a rank's exit 3 to 1)
            assert "RCCL did not come up" in r.stderr, r.stderr[-3000:]
            assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
            return r
        assert r.returncode == 0, r.stderr[-3000:]
        j = json.load(open(detail))
    raw = r.stdout.strip().splitlines()
    assert raw[-1].startswith('{"metric"'), r.stdout[-2000:]      # the driver parses the LAST line
    lines = [ln for ln in raw if ln.startswith("{")]               # (RCCL prints its version banner to stdout)
    assert len(r.stdout) < 8000                                    # the driver keeps an 8 KB tail
    assert len(lines[-1]) < 4096 and lines[-1].startswith('{"metric"')
    assert all(ln.startswith('{"leg"') for ln in lines[:-1])
    h = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in h, k
    for k in ("value", "ms_per_step", "value_full_count", "value_early_exit"):
        if j.get(k) is not None:
            assert abs(h[k] - j[k]) <= 1e-5 * abs(j[k]), k
    assert h["n_gpus"] == j["n_gpus"] and h["steps"] == j["steps"] and h["config"]["points"] == j["config"]["points"]
    if j.get("roofline"):
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "kernel"):
            assert k in h["roofline"], k
        assert len(h["roofline"]["kernel"]) <= 80 and abs(h["roofline"]["frac"] - j["roofline"]["frac"]) < 1e-5
        if j.get("single_stream"):
            # the per-kernel figures come from the one-stream pass, whose own rate is in the line beside them: the
            # dominant kernel's duration cannot exceed that pass's time per step
            assert h["roofline"]["launch_ms"] <= h["one_stream"]["ms_per_step"] * 1.001
    j["_headline"], j["_legs"] = h, [json.loads(ln) for ln in lines[:-1]]
    return j


def _check(j, streams, n_gpus=1):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "value_is", "repeats", "final_fit"):
        assert k in j, k
    assert j["n_gpus"] == n_gpus and j["steps"] == 6 and j["warmup"] == 2 and j["unit"] == "hypotheses/s"
    assert j["value"] > 0 and abs(j["value"] - n_gpus * j["config"]["hypotheses_per_gpu_per_step"] * 6 /
                                  (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]
    # `value` is the full-count rate (SURVEY 8d's unit) and says so; the early-exit rate rides beside it
    assert j["value_is"] == "full_count" and j["value"] == j["value_full_count"]
    assert "value_full_count" in j["config"]["step"]
    assert j["value_early_exit"] > 0 and len(j["repeats"]["full_count"]) == 2 and len(j["repeats"]["early_exit"]) == 2
    assert abs(sum(j["repeats"]["full_count"]) / 2 - j["value"]) < 1e-9 * j["value"]    # the median of the regions
    for key in ("roofline", "roofline_early_exit"):
        r = j[key]
        assert r["bound"] in ("valu", "mfma", "hbm") and 0 < r["frac"] <= 1.0, (key, r["frac"])
    assert j["roofline"]["rate"] == "full_count" and j["roofline_early_exit"]["rate"] == "early_exit"
    assert j["config"]["streams"] == streams
    if streams > 1:
        assert j["single_stream"]["value"] > 0
    ff = j["final_fit"]
    assert ff["winner_votes"] > 0
    for k in ("lm_info", "lm_nfev", "params_empty"):
        assert k in ff, k
    assert len(j["per_rank_hypotheses_per_s"]) == j["config"]["world_size"]


def test_bench_lanes_and_single_stream():
    a = _run(["--workload", "plane"])
    _check(a, 4)
    b = _run(["--workload", "plane", "--streams", "1"])
    _check(b, 1)
    # the same steps: the last step's winner and fit do not depend on the number of streams
    assert a["final_fit"]["winner_votes"] == b["final_fit"]["winner_votes"]
    assert a["final_fit"]["params"] == b["final_fit"]["params"]
    # counting everything and the bounded scan see different work, and say so
    wf, we = a["roofline"]["work_model"], a["roofline_early_exit"]["work_model"]
    assert not wf["bounded_scan"] and wf["hypotheses_counted_exactly"] == 4096
    assert we["bounded_scan"] and we["hypotheses_counted_exactly"] < 4096


def test_bench_host_threads_for_the_iterative_fit():
    a = _run(["--workload", "sphere", "--streams", "3"])
    _check(a, 3)
    b = _run(["--workload", "sphere", "--streams", "1"])
    assert a["final_fit"]["winner_votes"] == b["final_fit"]["winner_votes"]
    assert a["final_fit"]["lm_info"] in (1, 2, 3, 4) and a["final_fit"]["lm_nfev"] > 0 and not a["final_fit"]["params_empty"]
    assert a["lm"]["evaluations_per_s"] > 0


def test_bench_us_iterative_line_says_what_the_fit_did():
    """BASELINE config 5 as written ends at MINPACK's evaluation limit on large frame counts (info 5 = the
    reference's EMPTY vector): whatever happens, the line must say it"""
    j = _run(["--workload", "us", "--points", "30000", "--rates", "full"])
    ff = j["final_fit"]
    assert ff["lm_nfev"] > 0 and ff["lm_info"] >= 1
    assert ff["params_empty"] == (ff["lm_info"] not in (1, 2, 3, 4)) == (len(ff["params"]) == 0)
    assert j["lm"]["evaluations_per_step"] >= 1 and j["lm"]["evaluations_per_s"] > 0
    assert "lm_nfev_cost_stopped_moving" in ff


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_rccl_path_with_one_engine_per_stream():
    j = _run(["--workload", "plane", "--streams", "2"],
             env={"LSQR_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    _check(j, 2)
    assert j["config"]["world_size"] == 1 and "RCCL" in j["config"]["collectives"]
    # the line says what bringing RCCL up cost: the world's communicator + one per stream, each proven by an all-reduce
    r = j["config"]["rccl"]
    assert r["communicators"] == 3 and r["bringup_s"] > 0 and r["stream_groups_s"] > 0 and r["fallback"] is None
    print("RCCL bring-up at world 1: world group %.2f s, two stream groups %.2f s" % (r["bringup_s"], r["stream_groups_s"]))


def test_bench_rccl_stream_groups_stop_at_their_budget():
    """a budget of zero seconds: the first per-stream communicator is kept, the others are not created, the run goes on
    with one stream and says so -- what keeps an 8-rank run inside the driver's window if communicator creation is slow"""
    j = _run(["--workload", "plane", "--streams", "4"],
             env={"LSQR_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                  "LSQR_RCCL_BUDGET_S": "0"})
    r = j["config"]["rccl"]
    assert r["communicators"] == 2 and "1 of 4" in r["fallback"] and j["config"]["streams"] == 1
    assert j["value"] > 0


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher environment: the parent starts the ranks as children and relays
    rank 0's line.  One-GPU box: both ranks share device 0 and the collectives run over gloo (RCCL refuses two ranks
    on one device); the driver's 8-GPU run uses RCCL."""
    for step in ("host", "device"):
        j = _run(["--workload", "plane", "--gpus", "2", "--streams", "1"],
                 env={"LSQR_SHARE_GPU": "1", "LSQR_DIST_BACKEND": "gloo", "LSQR_STEP": step})
        _check(j, 1, n_gpus=2)
        assert j["config"]["world_size"] == 2 and j["n_gpus"] == 2
        assert len(j["per_rank_hypotheses_per_s"]) == 2 and min(j["per_rank_hypotheses_per_s"]) > 0


def test_bench_four_ranks_equal_one_rank_on_the_same_stream():
    """rehearsal of the driver's multi-GPU run with several ranks on the box's one card (its guard admits 6 processes
    on the GPU at once; the test runner and whatever an earlier test left exiting count too, so four ranks; the
    8-rank shape runs on CPU in tests/test_distributed.py): the ranks share device 0, exchanges over gloo on device
    buffers.  The last step's winner, consensus count and fit are those of ONE rank scanning the same 4 x H hypotheses
    per step."""
    env = {"LSQR_SHARE_GPU": "1", "LSQR_DIST_BACKEND": "gloo", "LSQR_STEP": "device"}
    a = _run(["--workload", "plane", "--gpus", "4", "--streams", "1", "--batch", "512"], env=env)
    _check(a, 1, n_gpus=4)
    assert a["config"]["world_size"] == 4 and len(a["per_rank_hypotheses_per_s"]) == 4
    assert min(a["per_rank_hypotheses_per_s"]) > 0
    assert abs(sum(a["per_rank_hypotheses_per_s"]) - a["value"]) < 0.35 * a["value"]   # each rank's own clock
    b = _run(["--workload", "plane", "--streams", "1", "--batch", "2048"])
    fa, fb = a["final_fit"], b["final_fit"]
    assert fa["winner_votes"] == fb["winner_votes"] and fa["inliers"] == fb["inliers"]
    pa, pb = fa["params"], fb["params"]
    assert len(pa) == len(pb) == 6 and max(abs(x - y) for x, y in zip(pa, pb)) < 1e-9 * max(1.0, max(map(abs, pb)))


def test_bench_gloo_fallback_is_opt_in():
    """two ranks on ONE device with the default (nccl) backend: RCCL refuses the duplicate device on every rank.
    Default: NO line, exit non-zero (a scaling record can never be a gloo line by accident).  --allow-gloo: every rank
    moves to gloo together and the line says so in config.collectives."""
    _run(["--workload", "plane", "--gpus", "2", "--streams", "1"], env={"LSQR_SHARE_GPU": "1"}, rc=3)
    j = _run(["--workload", "plane", "--gpus", "2", "--streams", "1", "--allow-gloo"], env={"LSQR_SHARE_GPU": "1"})
    _check(j, 1, n_gpus=2)
    assert j["config"]["world_size"] == 2 and "FALLBACK" in j["config"]["collectives"], j["config"]["collectives"]
    assert "FALLBACK" in j["_headline"]["config"]["collectives"]
    assert len(j["per_rank_hypotheses_per_s"]) == 2 and min(j["per_rank_hypotheses_per_s"]) > 0


def test_bench_gpus_2_one_process_multi_transport():
    j = _run(["--workload", "plane", "--gpus", "2", "--transport", "multi"], env={"LSQR_SHARE_GPU": "1"})
    assert j["config"]["world_size"] == 2 and j["config"]["collectives"] == "peer copies (lsqr_multi)"
    assert j["value"] > 0 and j["value_is"] == "full_count" and j["value_early_exit"] > 0
    assert j["final_fit"]["winner_votes"] > 0


def test_bench_default_run_carries_the_other_configs():
    """the default plane run appends short legs of BASELINE configs 3-5 (scaled down here)"""
    j = _run(["--workload", "plane", "--leg-scale", "0.02", "--cpu-seconds", "0.3"], extra=("--no-end-to-end",))
    legs = j["other_configs"]
    assert len(legs) == 5 and not any("error" in leg for leg in legs), legs
    names = [leg["config"]["workload"] for leg in legs]
    assert "Sphere" in names[0] and "Dense" in names[1] and "ITERATIVE" in names[2] and "ANALYTIC" in names[3]
    assert "PlanePhantom" in names[4]
    for leg in legs:
        assert leg["value"] > 0 and leg["ms_per_step"] > 0 and 0 < leg["roofline"]["frac"] <= 1
        assert leg["cpu_baseline"]["value"] > 0 and leg["cpu_baseline"]["cores"] == 1
        for k in ("lm_info", "lm_nfev", "params_empty"):
            assert k in leg["final_fit"]
    assert legs[2]["final_fit"]["lm_nfev"] > 0 and legs[2]["lm"]["evaluations_per_s"] > 0
    assert len(j["_legs"]) == 5 and [leg["value"] > 0 for leg in j["_legs"]]
    assert len(j["_headline"]["other_configs"]) == 5
    hb = j["_headline"]["cpu_baseline"]
    assert hb["cores"] == 1 and hb["kind"] in ("reference", "port") and hb["full_count"]["value"] > 0
    cb = j["cpu_baseline"]
    assert cb["full_count"]["value"] > 0 and "full agree" in cb["unit_note"]
    assert abs(j["speedup_vs_cpu_baseline"] - j["value_full_count"] / cb["full_count"]["value"]) < 1e-6 * j["speedup_vs_cpu_baseline"]
