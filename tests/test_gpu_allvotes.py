"""Every vote of a whole batch against the oracle (run with -m gpu): 4096 hypotheses x 1 M observations per model, the
full count of the bench path (lsqr_batch_fit with scan_bound 0 -- bench.py's `value`) AND the bounded scan, each vote
compared with oracle/ (orc_scan: the serial agree() loop of RANSAC.hxx:94-99 without its exit, on 16 host threads).
tests/test_gpu_fullsize.py checks ~40 hypotheses per config at 10 M; here nothing is sampled.

Plus BASELINE.json configs[0] AS WRITTEN (plane, 10 k points, 30 % outliers, p = 0.999): the device against the
restated loop on the product's own subset stream, and against the REFERENCE's RANSAC.hxx run on the same records
(tests/golden/config1_ref_vectors.npz: its draws replayed through the device)."""
import os

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
SEED = 0xC0FFEE
H = 4096
N = 1_000_000


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _all_votes(ctx, model, omodel, data, delta, ls_type, exact_models, dim=3):
    oc = O.cfg(omodel, dim, delta, ls_type)
    k = O.lib().orc_min_subset(oc)
    n = len(data)
    ctx.set_model(model, dim, delta, ls_type).upload(data)
    ctx.set_option("scan_bound", 0)
    r0 = ctx.batch_fit(SEED, 0, H, want_consensus=True)          # bench.py's step 0, every hypothesis counted
    par, valid, votes0 = ctx.hypotheses()
    assert r0["status"] == L.OK
    if exact_models:    # closed-form minimal solves: the device's models ARE the oracle's, bit for bit, all 4096
        subs = O.ctr_subsets(SEED, 0, H, n, k)
        for h in range(H):
            want = O.estimate(oc, data[subs[h]])
            assert bool(valid[h]) == (len(want) > 0), h
            if valid[h]:
                assert np.array_equal(par[h][:len(want)], want), h
    want_votes = O.scan_many(oc, par, valid, data)               # the oracle's count for EVERY hypothesis
    bad = np.flatnonzero(np.where(valid > 0, votes0, 0) != want_votes)
    assert len(bad) == 0, "full count differs from the oracle at %d of %d hypotheses, first h=%d: %d vs %d" % (
        len(bad), H, bad[0], votes0[bad[0]], want_votes[bad[0]])
    bi = int(np.argmax(want_votes))                                   # first max: RANSAC.hxx:100's strict '>'
    assert int(r0["info"].best_index) == bi and r0["info"].best_votes == want_votes[bi]
    wcnt, wmask = O.scan(oc, par[bi], data)
    assert wcnt == want_votes[bi] and np.array_equal(r0["consensus"], wmask)
    # the bounded scan / early exit on the same batch: a vote is exact or the hypothesis provably could not win
    ctx.set_option("scan_bound", 1)
    r1 = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    _, valid1, votes1 = ctx.hypotheses(params=False)
    assert np.array_equal(valid1, valid)
    run_max = np.maximum.accumulate(np.concatenate([[0], want_votes[:-1]]))   # best count before h
    differs = votes1 != want_votes
    assert np.all(votes1[differs] <= run_max[differs]) and np.all(want_votes[differs] <= run_max[differs])
    newmax = want_votes > run_max                                      # every update of the serial loop: exact
    assert np.array_equal(votes1[newmax], want_votes[newmax])
    assert int(r1["info"].best_index) == bi and np.array_equal(r1["consensus"], wmask)
    assert np.array_equal(r1["params"], r0["params"])
    return r0, oc, wmask


def test_all_4096_plane_votes_at_1M_equal_the_oracle(ctx):
    data = synth.plane(N, 0.5, seed=41)[0]
    r, oc, wmask = _all_votes(ctx, L.PLANE, O.PLANE, data, 0.5, 0, True)
    assert ctx.index_info()["built"]
    want = O.ls(oc, data, wmask)
    assert abs(abs(r["params"][:3] @ want[:3]) - 1) < 1e-6


def test_all_4096_sphere_votes_at_1M_equal_the_oracle(ctx):
    data = synth.sphere(N, 0.5, seed=42)[0]
    r, oc, wmask = _all_votes(ctx, L.SPHERE, O.SPHERE, data, 0.5, L.LS_ALGEBRAIC, True)
    assert ctx.index_info()["built"]
    assert np.allclose(r["params"], O.ls(oc, data, wmask), rtol=1e-6, atol=1e-6)


def test_all_4096_line_votes_at_1M_equal_the_oracle(ctx):
    data = synth.line(N, 0.5, seed=43)[0]
    _all_votes(ctx, L.LINE, O.LINE, data, 0.5, 0, True)
    assert ctx.index_info()["built"]


def test_all_4096_us_votes_at_1M_frames_equal_the_oracle(ctx):
    """the 12 x 12 pseudo-inverse solves differ from the oracle's in the last bits (tests/test_gpu_flips.py), so the
    votes are compared with the oracle's agree() on the DEVICE's own models -- all 4096 of them"""
    data = synth.us_single_fast(N, 0.5, seed=44)[0]
    _all_votes(ctx, L.US_SINGLE, O.US_SINGLE, data, 3.0, L.LS_ANALYTIC, False, dim=0)


# ---- BASELINE.json configs[0] as written -----------------------------------------------------------------------------
def _plane_close(got, want):
    assert len(got) == len(want) == 6
    s = np.sign(got[:3] @ want[:3])
    assert np.allclose(s * got[:3], want[:3], rtol=1e-6, atol=1e-6)
    assert abs((got[3:] - want[3:]) @ want[:3]) < 1e-6 * max(1.0, np.abs(want[3:]).max())
    assert np.allclose(got[3:], want[3:], rtol=1e-6, atol=1e-6)


def test_config1_plane_10k_30pct_as_written(ctx, golden_dir):
    data = synth.plane(10_000, 0.3)[0]                      # BASELINE configs[0]; delta 0.5, p 0.999
    oc = O.cfg(O.PLANE, 3, 0.5)
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    # (a) the product's own subset stream: device vs the restated serial loop -- iterations, consensus, parameters
    for seed in (1, 2, 3, 20261003):
        for index in (0, 2):                                # exhaustive kernels / two-level scan
            ctx.set_option("scan_index", index)
            r = ctx.ransac(0.999, seed=seed)
            w = O.ransac(oc, data, 0.999, sampler="ctr", seed=seed)
            assert r["status"] == L.OK and r["info"].iterations == w["iters"]
            assert r["fraction"] == w["fraction"] and np.array_equal(r["consensus"], w["consensus"])
            assert r["info"].best_votes == w["best_votes"]
            _plane_close(r["params"], w["params"])
    ctx.set_option("scan_index", 1)
    # (b) the plumbing leg: the REFERENCE's RANSAC.hxx ran on these records (golden vectors); its draws replayed
    # through the device select the same winner -> same fraction, same consensus set bit for bit, same fit
    g = np.load(os.path.join(golden_dir, "config1_ref_vectors.npz"))
    for seed in (21, 22, 23):
        key = "s%d_" % seed
        r = ctx.ransac(0.999, subsets=g[key + "subsets"])
        assert r["fraction"] == g[key + "fraction"][0]
        assert np.array_equal(r["consensus"], np.unpackbits(g[key + "consensus_bits"])[:len(data)])
        _plane_close(r["params"], g[key + "params"])
        if O.ref_available():                               # and live, when the compiled reference travelled
            live = O.ref_ransac(oc, data, 0.999, seed=seed, subsets_cap=4096)
            assert live["fraction"] == r["fraction"] and np.array_equal(live["consensus"], r["consensus"])


# ---- the refined index (k_refine_runs) on awkward uploads ---------------------------------------------------------------
@pytest.mark.parametrize("n", [300, 511, 8191, 8192, 8193, 12_289, 70_001, 131_072])
@pytest.mark.parametrize("kind", ["plane", "duplicates", "one_point", "collinear"])
def test_refined_index_gives_the_exhaustive_kernels_votes(ctx, n, kind):
    """run boundaries (8192 records per refined run), partial runs, partial cells; many equal coordinates (the radix
    select's ties are split by position), every record the same point, every record on one line: the two-level scan
    over the refined order counts exactly what the exhaustive fp64 kernel counts"""
    g = np.random.default_rng(n)
    if kind == "plane":
        data = synth.plane(n, 0.5, seed=n)[0]
    elif kind == "duplicates":          # 37 distinct points, each repeated ~n / 37 times, plus a few outliers
        base = g.uniform(-50, 50, (37, 3))
        data = base[g.integers(0, 37, n)]
        data[:: 97] = g.uniform(-500, 500, (len(data[:: 97]), 3))
    elif kind == "one_point":
        data = np.tile(np.array([[3.0, -7.0, 11.0]]), (n, 1))
        data[: n // 3] += g.normal(0, 0.2, (n // 3, 3))
    else:
        t = g.uniform(-1000, 1000, n)
        data = np.outer(t, [0.6, 0.0, 0.8]) + [1.0, 2.0, 3.0]
        data[::5] += g.normal(0, 0.3, (len(data[::5]), 3))
    data = np.ascontiguousarray(data)
    Hn = 512
    for model, ls in ((L.PLANE, 0), (L.SPHERE, L.LS_ALGEBRAIC), (L.LINE, 0)):
        ctx.set_model(model, 3, 0.5, ls).upload(data)
        ctx.hypotheses_sample(5, 0, Hn)
        out = {}
        for index, refine in ((0, 1), (2, 1), (2, 0)):
            ctx.set_option("scan_refine", refine)
            ctx.set_option("scan_index", index)
            ctx.set_option("scan_filter", 1 if index else 0)
            ctx.scan()
            out[(index, refine)] = ctx.hypotheses(params=False)[2].copy()
        ctx.set_option("scan_refine", 1)
        ctx.set_option("scan_index", 1)
        ctx.set_option("scan_filter", 1)
        assert np.array_equal(out[(2, 1)], out[(0, 1)]), (model, n, kind)
        assert np.array_equal(out[(2, 0)], out[(0, 1)]), (model, n, kind)
        assert out[(0, 1)].max() > 0
