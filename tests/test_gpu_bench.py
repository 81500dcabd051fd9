"""bench.py end to end on a small workload (run with -m gpu): the JSON contract of the line it prints, on every driver
of the steps -- lanes of one context (closed-form fits), host threads with a context each (iterative fits), one
stream, and the RCCL path at world size 1 with one engine, process group and torch stream per stream."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env=None):
    e = dict(os.environ)
    if env:
        e.update(env)
    r = subprocess.run([sys.executable, "bench.py", "--points", "200000", "--steps", "6", "--warmup", "2",
                        "--no-cpu-baseline", "--no-end-to-end"] + args, cwd=ROOT, env=e, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line
    return json.loads(lines[0])


def _check(j, streams):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 2 and j["unit"] == "hypotheses/s"
    assert j["value"] > 0 and abs(j["value"] - j["config"]["hypotheses_per_gpu_per_step"] * 6 /
                                  (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] in ("valu", "mfma", "hbm") and 0 < r["frac"] <= 1.0
    assert j["config"]["streams"] == streams
    if streams > 1:
        assert j["single_stream"]["value"] > 0
    assert j["final_fit"]["winner_votes"] > 0


def test_bench_lanes_and_single_stream():
    a = _run(["--workload", "plane"])
    _check(a, 4)
    b = _run(["--workload", "plane", "--streams", "1"])
    _check(b, 1)
    # the same steps: the last step's winner and fit do not depend on the number of streams
    assert a["final_fit"]["winner_votes"] == b["final_fit"]["winner_votes"]
    assert a["final_fit"]["params"] == b["final_fit"]["params"]


def test_bench_host_threads_for_the_iterative_fit():
    a = _run(["--workload", "sphere", "--streams", "3"])
    _check(a, 3)
    b = _run(["--workload", "sphere", "--streams", "1"])
    assert a["final_fit"]["winner_votes"] == b["final_fit"]["winner_votes"]


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
