"""Multi-GPU step logic (lsqrrecipes_amd/distributed.py) on CPU: world_size 2 over gloo.

The collective pattern (all-reduce MAX of the packed winner, all-reduce SUM of zero-padded
parameters and of the moment blocks) is exercised with a stand-in engine that answers the Context
calls from the CPU oracle -- test infrastructure only; on the GPU the engine is the real Context
(covered by bench.py --gpus N on the driver's 8-GPU node)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleEngine:
    """Implements the subset of the Context interface ShardedRansac uses, on the CPU oracle."""

    def __init__(self, data, delta=0.5):
        from lsqrrecipes_amd import _lib as L
        from oracle import pyoracle as O
        self.O, self.L = O, L
        self.data = data
        self.n = len(data)
        self.oc = O.cfg(O.PLANE, 3, delta)
        self.cfg = L.ModelCfg(L.PLANE, 3, delta, 0, 0)
        self.K, self.P, self.ND = 3, 6, 3
        self._mask = np.zeros(self.n, dtype=np.uint8)

    def hypotheses_sample(self, seed, first, H):
        O = self.O
        self.subs = O.ctr_subsets(seed, first, H, self.n, self.K)
        self.par = [O.estimate(self.oc, self.data[s]) for s in self.subs]

    def scan(self):
        self.votes = np.array([self.O.scan(self.oc, p, self.data)[0] if len(p) else 0
                               for p in self.par], dtype=np.int64)

    def best(self):
        if self.votes.max() == 0:
            return 0, 0, 0
        i = int(np.argmax(self.votes))
        return (int(self.votes[i]) << 32) | (0xFFFFFFFF - i), int(self.votes[i]), i

    def hypothesis(self, h):
        return self.par[h], True

    def mask(self, params, begin, end, want_mask=False):
        cnt, m = self.O.scan(self.oc, params, self.data[begin:end])
        self._mask[begin:end] = m
        return None, cnt

    def moments(self, origin, begin, end, phase=0, use_mask=True):
        x = self.data[begin:end][self._mask[begin:end].astype(bool)] - origin
        blk = [float(len(x))] + list(x.sum(0))
        for i in range(3):
            for j in range(i, 3):
                blk.append(float((x[:, i] * x[:, j]).sum()))
        return np.array(blk)

    def solve_moments(self, block, origin):
        N = block[0]
        mean = block[1:4] / N
        C = np.zeros((3, 3))
        k = 4
        for i in range(3):
            for j in range(i, 3):
                C[i, j] = C[j, i] = block[k] - N * mean[i] * mean[j]
                k += 1
        w, V = np.linalg.eigh(C)
        return np.concatenate([V[:, 0], mean + origin]), None


def _worker(rank, world, port, data, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac
    eng = OracleEngine(data)
    sr = ShardedRansac(eng, Comm(dist, "cpu"))
    votes, gidx, par = sr.batch(seed=5, batch_index=0, H=24)
    fit, cnt, _ = sr.fit(par)
    votes2, gidx2, par2 = sr.batch(seed=5, batch_index=1, H=24)
    st = sr.step(seed=5, batch_index=0, H=24)  # the one-call form (falls back to batch() + fit() here)
    assert (st[0], st[1]) == (votes, gidx) and np.array_equal(st[2], par) and st[4] == cnt
    assert np.allclose(st[3], fit, rtol=0, atol=1e-12)
    if rank == 0:
        out.put((votes, gidx, par, fit, cnt, votes2, gidx2))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_ransac_world2_matches_single_process():
    from lsqrrecipes_amd import synth
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac, slice_bounds
    from oracle import pyoracle as O
    data = synth.plane(1500, 0.4, seed=321)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    votes, gidx, par, fit, cnt, votes2, gidx2 = res
    # single process over the same 48-hypothesis global batch: same first-max winner
    eng = OracleEngine(data)
    sr = ShardedRansac(eng, Comm(None))
    v1, g1, p1 = sr.batch(seed=5, batch_index=0, H=48)
    assert (votes, gidx) == (v1, g1) and np.array_equal(par, p1)
    oc = O.cfg(O.PLANE, 3, 0.5)
    wcnt, wmask = O.scan(oc, par, data)
    assert cnt == wcnt == votes
    want = O.ls(oc, data, wmask)
    assert abs(abs(fit[:3] @ want[:3]) - 1) < 1e-9
    assert abs((fit[3:] - want[3:]) @ want[:3]) < 1e-7
    # second batch continues the global stream at index 48
    v2, g2, _ = sr.batch(seed=5, batch_index=1, H=48)
    assert (votes2, gidx2) == (v2, g2) and 48 <= gidx2 < 96
    # observation slices tile [0, n) exactly
    b = [slice_bounds(1501, r, 4) for r in range(4)]
    assert b[0][0] == 0 and b[-1][1] == 1501 and all(b[i][1] == b[i + 1][0] for i in range(3))


def test_sharded_ransac_world8_matches_single_process():
    """the driver's N = 8 shape rehearsed on CPU: eight gloo ranks, hypothesis stream sharded eight ways, all-reduce MAX
    of the packed winner, all-reduce SUM of the slices' moment blocks -- the winner is the one a single process finds
    in the same 8 x H stream, the consensus count and the fit are the whole upload's"""
    from lsqrrecipes_amd import synth
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac
    from oracle import pyoracle as O
    data = synth.plane(1200, 0.4, seed=654)[0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, data, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    votes, gidx, par, fit, cnt, votes2, gidx2 = res
    sr = ShardedRansac(OracleEngine(data), Comm(None))
    v1, g1, p1 = sr.batch(seed=5, batch_index=0, H=8 * 24)
    assert (votes, gidx) == (v1, g1) and np.array_equal(par, p1)
    oc = O.cfg(O.PLANE, 3, 0.5)
    wcnt, wmask = O.scan(oc, par, data)
    assert cnt == wcnt == votes
    want = O.ls(oc, data, wmask)
    assert abs(abs(fit[:3] @ want[:3]) - 1) < 1e-9
    v2, g2, _ = sr.batch(seed=5, batch_index=1, H=8 * 24)
    assert (votes2, gidx2) == (v2, g2) and 8 * 24 <= gidx2 < 2 * 8 * 24


def test_comm_world1_is_identity():
    from lsqrrecipes_amd.distributed import Comm
    c = Comm(None)
    assert c.allreduce_max_i64(7) == 7 and c.world == 1
    assert np.array_equal(c.allreduce_sum_f64([1.0, 2.0]), [1.0, 2.0])


def _bringup(world, allow, mode):
    """`world` ranks of tests/dist_bringup.py (bench.bring_up_rccl with no GPU: RCCL cannot come up) -> [(rc, stdout)]"""
    import subprocess
    port = str(_free_port())
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_bringup.py"), str(r), str(world), port,
                            "1" if allow else "0", mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for r in range(world)]
    return [(p.wait(timeout=180), p.communicate()[0]) for p in ps]


def test_rccl_verdict_is_collective_and_gloo_is_opt_in():
    """bench.py --gpus N: a SCALE line can never be a gloo line by accident.  RCCL failing on every rank exits 3 on every
    rank by default; with --allow-gloo all ranks move to gloo together and say so."""
    res = _bringup(2, allow=False, mode="all")
    assert [rc for rc, _ in res] == [3, 3], res
    res = _bringup(2, allow=True, mode="all")
    assert [rc for rc, _ in res] == [0, 0], res
    for _, out in res:
        assert "RESULT gloo cpu 2" in out and "No HIP GPUs" in out, out


def test_rccl_partial_failure_never_splits_the_world():
    """one rank stuck inside RCCL bring-up while its peer failed: nobody falls back alone -- every rank exits non-zero
    within the watcher's 20 s, --allow-gloo or not"""
    import time
    t0 = time.time()
    res = _bringup(2, allow=True, mode="some")
    assert all(rc != 0 for rc, _ in res), res
    assert not any("RESULT" in out for _, out in res)
    assert time.time() - t0 < 90


def test_bench_headline_fits_the_drivers_tail():
    """the LAST stdout line of bench.py must stay under 4 KB whatever the run produced (BENCH_r03 was unparsable: one
    42 KB line); checked here on the committed full record of a real default run"""
    import json
    import bench
    rec = os.path.join(ROOT, "profiles", "r03_bench_plane.json")
    d = json.load(open(rec))
    line = bench.headline(d, "bench_detail.json")
    assert len(line) <= bench.HEADLINE_LIMIT and "\n" not in line
    h = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "value_full_count",
              "value_early_exit"):
        assert k in h, k
    assert abs(h["value"] - d["value"]) < 1e-5 * d["value"] and h["config"]["points"] == 10_000_000
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms", "kernel"):
        assert k in h["roofline"], k
    assert len(h["roofline"]["kernel"]) <= 80
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in h["cpu_baseline"], k
    legs = [json.dumps(bench.leg_line(leg), separators=(",", ":")) for leg in d["other_configs"]]
    assert all(len(x) < 1200 for x in legs)
    assert len(line) + sum(len(x) + 1 for x in legs) < 8000      # the whole stdout fits the driver's 8 KB tail
    # a pathological record (huge strings everywhere) still yields a line under the limit
    d["config"]["workload"] = "x" * 5000
    d["cpu_baseline"]["sample"] = "y" * 5000
    d["roofline"]["kernel"] = "z" * 5000
    assert len(bench.headline(d, "bench_detail.json")) <= bench.HEADLINE_LIMIT


def test_sharded_dense_fit_is_taken_from_the_rows_when_the_block_is_ill_conditioned():
    """ShardedRansac._refine (r05): a dense final fit that only had the summed Gram block and saw a pivot below 1e-6
    max|G| (lsqr_fit_info.reserved == 2) is taken again from the replicated rows -- the winner masks the WHOLE upload,
    lsqr_ls_fit runs the double-double route -- on every rank alike; any other fit is left alone"""
    from types import SimpleNamespace
    from lsqrrecipes_amd import _lib as L
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac

    class Eng:
        def __init__(self):
            self.cfg = L.ModelCfg(L.DENSE, 8, 0.1, 0, 0)
            self.n, self.P, self.calls = 1000, 8, []

        def mask(self, params, begin, end, want_mask=False):
            self.calls.append(("mask", begin, end, tuple(params)))
            return None, 700

        def ls_fit(self, use_mask=False):
            self.calls.append(("ls_fit", use_mask))
            return np.arange(8.0), SimpleNamespace(n_params=8, reserved=1, n_used=0, lm_info=0, lm_nfev=0)

    e = Eng()
    s = ShardedRansac(e, Comm(None, "cpu"))
    winner = np.linspace(1, 2, 8)
    block_fit = np.ones(8)
    fit, info = s._refine(block_fit, SimpleNamespace(n_params=8, reserved=2, n_used=640), 0, 500, winner)
    assert e.calls == [("mask", 0, 1000, tuple(winner)), ("ls_fit", True)]
    assert np.array_equal(fit, np.arange(8.0)) and info.reserved == 1 and info.n_used == 640
    e.calls.clear()
    fit, info = s._refine(block_fit, SimpleNamespace(n_params=8, reserved=0, n_used=640), 0, 500, winner)
    assert e.calls == [] and fit is block_fit and info.reserved == 0
