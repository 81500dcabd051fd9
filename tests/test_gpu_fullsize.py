"""Full-size parity on the exact path bench.py times (run with -m gpu on an MI355X).

One test per BASELINE.json config that runs on a GPU (configs[1..4]): the workload bench.py generates
for it (same generator, size, threshold, batch size and entry point -- lsqr_batch_fit, i.e. sample ->
solve -> scan -> first-max winner -> consensus mask -> final fit, with the spatial index armed exactly as
the bench's warm-up arms it), then compared with the CPU oracle at full size:
  * a few dozen hypotheses of the batch (the winner, the first, the last, the best-voted ones and a
    random rest): minimal-subset model and vote count BIT-EXACT,
  * the winner's consensus mask BIT-EXACT,
  * the final fit within 1e-6 relative (BASELINE.json north_star).
RANSAC.hxx:94-99 (agree loop), :129-138 (consensus set + leastSquaresEstimate).
"""
import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L
from lsqrrecipes_amd import synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu

REL = 1e-6
SEED = 0xC0FFEE  # bench.py's sampler stream


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


def _pick(votes, valid, bi, n_good, n_total, seed):
    """indices to compare with the oracle: winner, first, last, the n_good best-voted, random rest"""
    H = len(votes)
    vv = np.where(valid > 0, votes.astype(np.int64), -1)
    good = list(np.argsort(-vv, kind="stable")[:n_good])
    rest = list(np.random.default_rng(seed).choice(H, size=n_total, replace=False))
    out = []
    for h in [bi, 0, H - 1] + good + rest:
        if int(h) not in out:
            out.append(int(h))
    return out[:max(n_total, 3 + n_good)]


def _early_exit_equals_full(ctx, r_full, valid, votes, H, n):
    """the same batch through the chunked early exit (scan_bound 1, what bench.py times as value_early_exit): winner,
    consensus set and fit identical; an abandoned hypothesis reports a partial count that is inert in the replay"""
    ctx.set_option("scan_bound", 1)
    r = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    w = ctx.scan_work()
    _, valid1, v1 = ctx.hypotheses(params=False)
    assert w["early_exit"] and w["row_hypothesis_pairs"] < H * n
    assert r["info"].best_index == r_full["info"].best_index and r["info"].best_votes == r_full["info"].best_votes
    assert np.array_equal(r["consensus"], r_full["consensus"]) and np.array_equal(r["params"], r_full["params"])
    assert np.array_equal(valid1, valid) and np.all(v1 <= votes)
    run = np.maximum.accumulate(np.where(valid > 0, votes, 0))
    prev = np.concatenate([[0], run[:-1]])
    rec = (valid > 0) & (votes > prev)                     # the serial loop's new maxima (strict '>')
    assert np.array_equal(v1[rec], votes[rec])
    part = v1 != votes
    assert np.all(votes[part] <= prev[part])
    return w


def _point_model_fullsize(ctx, model, omodel, gen, ls_type, expect_cell):
    n, H = 10_000_000, 4096
    data, truth, lab = gen(n, 0.5)                       # bench.py: make_data(workload, 10 M, 0.5)
    oc = O.cfg(omodel, 3, 0.5, ls_type)
    k = O.lib().orc_min_subset(oc)
    ctx.set_model(model, 3, 0.5, ls_type).upload(data)    # scan_index left at its default
    r = ctx.batch_fit(SEED, 0, H, want_consensus=True)    # bench.py step 0
    idx = ctx.index_info()
    assert idx["built"] and idx["cell_points"] == expect_cell, idx   # the two-level scan ran (k_scan_cells)
    assert r["status"] == L.OK
    info = r["info"]
    par, valid, votes = ctx.hypotheses()
    subs = O.ctr_subsets(SEED, 0, H, n, k)
    bi = int(info.best_index)
    vv = np.where(valid > 0, votes, 0)
    assert info.best_votes == vv.max() and bi == int(np.argmax(vv))      # first max (RANSAC.hxx:100)
    checked = _pick(votes, valid, bi, 8, 40, 11)
    assert len(checked) >= 32
    skipped = 0
    for h in checked:
        want = O.estimate(oc, data[subs[h]])
        assert bool(valid[h]) == (len(want) > 0), h
        if not valid[h]:
            assert votes[h] == 0
            continue
        assert np.array_equal(par[h], want), "minimal-subset model differs at h=%d" % h
        exact = O.scan(oc, want, data)[0]
        if votes[h] != exact:
            # the bounded scan (scan_bound, default for batches): a hypothesis that cannot become the running
            # maximum is not counted and reports 0 -- RANSAC.hxx:94 abandons exactly these
            assert votes[h] == 0 and h > 0 and exact <= votes[:h].max(), "vote count differs at h=%d" % h
            skipped += 1
    # the winner and the best-voted ones were counted exactly (with the rank bounds of the plane most of a random
    # sample is never counted: 30 of 40 here)
    assert len(checked) - skipped >= 8
    wcnt, wmask = O.scan(oc, par[bi], data)
    assert wcnt == info.best_votes == info.fit.n_used
    assert np.array_equal(r["consensus"], wmask), "winner's consensus mask differs from the oracle"
    # the same batch with every hypothesis counted (scan_bound 0: bench.py's `value` = value_full_count): every vote
    # of the sample exact, winner / consensus / fit identical
    ctx.set_option("scan_bound", 0)
    r0 = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    _, valid0, votes0 = ctx.hypotheses(params=False)
    ctx.set_option("scan_bound", 1)
    assert np.array_equal(valid0, valid)
    for h in checked:
        if valid[h]:
            assert votes0[h] == O.scan(oc, par[h], data)[0], "full count differs at h=%d" % h
    # r04: not a sample -- EVERY one of the 4096 full counts at 10 M observations against the oracle (orc_scan on the
    # host's threads: 4096 x 22 ms of one core)
    all_want = O.scan_many(oc, par, valid, data)
    bad = np.flatnonzero(np.where(valid0 > 0, votes0, 0) != all_want)
    assert len(bad) == 0, "full count differs from the oracle at %d of %d hypotheses (first h=%d)" % (len(bad), H, bad[0])
    assert np.all(votes <= votes0) and np.all((votes == votes0) | (votes == 0))
    assert r0["info"].best_index == info.best_index and r0["info"].best_votes == info.best_votes
    assert np.array_equal(r0["consensus"], r["consensus"]) and np.array_equal(r0["params"], r["params"])
    want = O.ls(oc, data, wmask)
    assert len(want) == len(r["params"]) > 0
    return r, want, truth, lab, wmask


def test_config2_plane_10M_batch_on_bench_path(ctx):
    """BASELINE configs[1]: plane, 10 M points, 50 % outliers, 4096 hypotheses through k_scan_cells."""
    r, want, truth, lab, wmask = _point_model_fullsize(ctx, L.PLANE, O.PLANE, synth.plane, 0, 512)
    got = r["params"]
    assert abs(abs(got[:3] @ want[:3]) - 1.0) < REL
    s = np.sign(got[:3] @ want[:3])
    assert np.allclose(s * got[:3], want[:3], rtol=REL, atol=REL)
    assert abs((got[3:] - want[3:]) @ want[:3]) < REL * max(1.0, np.abs(want[3:]).max())
    assert np.allclose(got[3:], want[3:], rtol=REL, atol=1e-6)
    assert abs(abs(got[:3] @ truth[:3]) - 1) < 1e-7
    # the next bench step (different subsets, index already built): winner votes and mask again
    r2 = ctx.batch_fit(SEED, 4096, 4096, want_consensus=True)
    par, valid, votes = ctx.hypotheses()
    bi = int(r2["info"].best_index) - 4096
    oc = O.cfg(O.PLANE, 3, 0.5)
    data = synth.plane(10_000_000, 0.5)[0]
    wcnt, wm = O.scan(oc, par[bi], data)
    assert wcnt == votes[bi] == r2["info"].best_votes and np.array_equal(r2["consensus"], wm)
    # bench.py's timed region drives the steps through lsqr_batch_fit_enqueue / _wait on FOUR lanes (streams with
    # their own buffers and index): at full size, both rates, every lane's step must be the blocking call's step --
    # whose winner mask is compared with the oracle
    for bound in (1, 0):
        ctx.set_option("scan_bound", bound)
        ctx.set_option("batch_lanes", 4)
        for sl in range(4):
            ctx.batch_fit_enqueue(SEED, sl * 4096, 4096, slot=sl)
        lanes = [ctx.batch_fit_wait(sl) for sl in range(4)]
        for sl in range(4):
            rb = ctx.batch_fit(SEED, sl * 4096, 4096, want_consensus=True)
            parb, _, votesb = ctx.hypotheses()
            bib = int(rb["info"].best_index) - sl * 4096
            cnt, mask = O.scan(oc, parb[bib], data)
            assert cnt == rb["info"].best_votes and np.array_equal(rb["consensus"], mask), (bound, sl)
            la = lanes[sl]["info"]
            assert (la.best_index, la.best_votes, la.fit.n_used) == (
                rb["info"].best_index, rb["info"].best_votes, rb["info"].fit.n_used), (bound, sl)
            assert np.array_equal(lanes[sl]["params"], rb["params"]), (bound, sl)
    ctx.set_option("scan_bound", 1)


def test_config3_sphere_10M_geometric_on_bench_path(ctx):
    """BASELINE configs[2] (one GPU's share): sphere, 10 M points, geometric (Levenberg-Marquardt) fit."""
    r, want, truth, lab, wmask = _point_model_fullsize(ctx, L.SPHERE, O.SPHERE, synth.sphere,
                                                       L.LS_GEOMETRIC, 512)
    assert 1 <= r["info"].fit.lm_info <= 4
    assert np.allclose(r["params"], want, rtol=REL, atol=1e-6)
    assert np.allclose(r["params"], truth, rtol=1e-4, atol=1e-2)


def test_line_10M_batch_on_bench_path(ctx):
    """north_star's third point model at the BASELINE size (not a BASELINE config: r04 compared it at 1 M only):
    line, 10 M points, 50 % outliers, 4096 hypotheses -- sampled minimal solves bit-exact, EVERY full count against
    the oracle, winner's consensus set bit for bit, the fit (LineParametersEstimator.hxx:68-111) to 1e-6."""
    r, want, truth, lab, wmask = _point_model_fullsize(ctx, L.LINE, O.LINE, synth.line, 0, 256)
    got = r["params"]
    assert abs(abs(got[:3] @ want[:3]) - 1.0) < REL                       # direction, modulo sign (eigenvector)
    d = got[3:] - want[3:]                                                  # the two points lie on the same line
    assert np.linalg.norm(d - (d @ want[:3]) * want[:3]) < REL * max(1.0, np.abs(want[3:]).max())
    assert abs(abs(got[:3] @ truth[:3]) - 1) < 1e-7


def test_config4_dense_2Mx64_on_bench_path(ctx):
    """BASELINE configs[3]: dense Ax ~ b, m = 2 M, n = 64, 1024 hypotheses (MFMA filter scan + worklist),
    consensus mask, MFMA SYRK + 64 x 64 solve.  Votes and mask bit-exact against the oracle evaluated on
    the device's minimal-solve models (the n x n SVD pseudo-inverse itself is VNL, unpinned: the models are
    compared at 1e-6).  The full-size least squares reference is LAPACK's SVD solve (numpy.linalg.lstsq: the
    same x = pinv(A) b the reference computes, DenseLinear...hxx:85-92); the oracle's own Jacobi-SVD
    restatement needs ~12 minutes at this size and is compared on a 100 000-row subset of the consensus."""
    m, ncol, H = 2_000_000, 64, 1024
    rows, x_true, lab = synth.dense(m, ncol, 0.05)               # bench.py: make_data("dense", 2 M, .)
    oc = O.cfg(O.DENSE, ncol, 0.1)
    ctx.set_model(L.DENSE, ncol, 0.1, L.LS_GEOMETRIC).upload(rows)
    ctx.set_option("scan_bound", 0)                               # bench.py's value_full_count: every vote exact
    r = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    assert r["status"] == L.OK
    info = r["info"]
    par, valid, votes = ctx.hypotheses()
    subs = O.ctr_subsets(SEED, 0, H, m, ncol)
    bi = int(info.best_index)
    vv = np.where(valid > 0, votes, 0)
    assert info.best_votes == vv.max() and bi == int(np.argmax(vv))
    checked = _pick(votes, valid, bi, 4, 12, 5)
    assert len(checked) >= 8
    for h in checked:
        assert valid[h]
        assert votes[h] == O.scan(oc, par[h], rows)[0], "vote count differs at h=%d" % h
    all_want = O.scan_many(oc, par, valid, rows)       # r04: all 1024 full counts over the 2 M rows, not a sample
    assert np.array_equal(np.where(valid > 0, votes, 0), all_want)
    for h in checked[:3]:   # the 64 x 64 minimal solve against the oracle's SVD pseudo-inverse
        want = O.estimate(oc, rows[subs[h]])
        assert len(want) == ncol
        assert np.allclose(par[h], want, rtol=REL, atol=REL * max(1.0, np.abs(want).max()))
    wcnt, wmask = O.scan(oc, par[bi], rows)
    assert wcnt == info.best_votes == info.fit.n_used
    assert np.array_equal(r["consensus"], wmask)
    sel = rows[wmask.astype(bool)]
    want = np.linalg.lstsq(sel[:, :ncol], sel[:, ncol], rcond=None)[0]
    scale = max(1.0, np.abs(want).max())
    assert np.abs(r["params"] - want).max() <= REL * scale
    _early_exit_equals_full(ctx, r, valid, votes, H, m)           # bench.py's value_early_exit on the same batch
    # the oracle's restatement on a subset of the consensus set, device fit of the same subset
    sub = np.ascontiguousarray(sel[:100_000])
    ctx.upload(sub)
    got_sub, _ = ctx.ls_fit()
    want_sub = O.ls(oc, sub)
    assert np.abs(got_sub - want_sub).max() <= REL * max(1.0, np.abs(want_sub).max())


def test_config5_us_1M_frames_on_bench_path(ctx):
    """BASELINE configs[4] shape: single-point-target US calibration, 1 M frames: 4096 hypotheses
    (12 x 12 analytic minimal solves), packed-fp32-filter scan, consensus mask, analytic fit of the
    consensus set; the iterative (Levenberg-Marquardt) fit of the same set is covered by
    test_config5_us_iterative_fit_1M."""
    n, H = 1_000_000, 4096
    rec, truth, lab = synth.us_single_fast(n, 0.5)              # bench.py: make_data("us", 1 M, 0.5)
    oc = O.cfg(O.US_SINGLE, 0, 3.0, 0)
    ctx.set_model(L.US_SINGLE, 3, 3.0, L.LS_ANALYTIC).upload(rec)
    ctx.set_option("scan_bound", 0)                               # bench.py's value_full_count: every vote exact
    r = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    assert r["status"] == L.OK
    info = r["info"]
    par, valid, votes = ctx.hypotheses()
    subs = O.ctr_subsets(SEED, 0, H, n, 4)
    bi = int(info.best_index)
    vv = np.where(valid > 0, votes, 0)
    assert info.best_votes == vv.max() and bi == int(np.argmax(vv))
    checked = _pick(votes, valid, bi, 8, 24, 3)
    for h in checked:
        want = O.estimate(oc, rec[subs[h]])
        assert bool(valid[h]) == (len(want) > 0), h
        if not valid[h]:
            continue
        assert np.allclose(par[h], want, rtol=REL, atol=REL * np.abs(want).max())
        assert votes[h] == O.scan(oc, par[h], rec)[0], "vote count differs at h=%d" % h
    all_want = O.scan_many(oc, par, valid, rec)        # r04: all 4096 full counts over the 1 M frames, not a sample
    assert np.array_equal(np.where(valid > 0, votes, 0), all_want)
    wcnt, wmask = O.scan(oc, par[bi], rec)
    assert wcnt == info.best_votes == info.fit.n_used
    assert np.array_equal(r["consensus"], wmask)
    want = O.ls(oc, rec, wmask)
    assert len(want) == len(r["params"]) == 20
    assert np.allclose(r["params"], want, rtol=REL, atol=REL * np.abs(want).max())
    assert (wmask.astype(bool) & ~lab).sum() <= 0.02 * n
    w = _early_exit_equals_full(ctx, r, valid, votes, H, n)       # bench.py's value_early_exit on the same batch
    assert w["row_hypothesis_pairs"] < 0.7 * H * n                # 50 % inliers: the wrong models go at half time


def test_config5_us_iterative_fit_against_minpack_fixtures(ctx, golden_dir):
    """BASELINE configs[4] as written: Levenberg-Marquardt with the reference's settings
    (SinglePointTarget...Estimator.cxx:287-295: tolerances 1e-15, 5000 evaluations).  tests/golden/us_lm_vectors.npz
    holds what MINPACK itself (SciPy's lmder) does on 1 k / 20 k / 100 k frames from the analytic start: the cost
    drops to its final 7 digits within a few dozen evaluations and then creeps along an ill-conditioned valley
    (scale factors ~0.14 against translations ~100), still gaining in the 9th digit -- relative gradient 5e-4
    after 60, 1.6e-5 after 5000 evaluations -- until either the trust region collapses (info 1/2, thousands of
    evaluations) or the 5000-evaluation limit fires (info 5 -> the reference returns an EMPTY vector); at 20 k
    frames and above the limit fired.  tests/golden/us_lm_flags.npz shows that the flag is not reproducible
    even between two MINPACKs fed the same f and J.  So: the cost the device reaches must be as low as SciPy's
    (1e-7 relative); where the device reports success its parameters match SciPy's iterate to 1e-6; where SciPy
    exhausted the limit the device must have spent thousands of evaluations too (no early 'success' by a
    looser rule)."""
    import os
    d = np.load(os.path.join(golden_dir, "us_lm_vectors.npz"))
    for m in (1000, 20000, 100000):
        key = "us_lm_%d_" % m
        rec = synth.us_single_fast(m, 0.0, seed=int(d[key + "seed"][0]))[0]
        ctx.set_model(L.US_SINGLE, 3, 3.0, L.LS_ITERATIVE).upload(rec)
        got, info = ctx.ls_fit()
        nfev_s, ier_s = [int(v) for v in d[key + "scipy_nfev_ier"]]
        cost_s = float(d[key + "scipy_cost"][0])
        assert cost_s * (1 - 1e-6) <= info.cost <= cost_s * (1 + 1e-7), (m, info.cost, cost_s)
        assert (len(got) > 0) == (1 <= info.lm_info <= 4)
        if len(got):
            x = d[key + "scipy_x"]
            assert np.allclose(got[:11], x, rtol=0, atol=REL * np.abs(x).max()), m
        if ier_s == 5:
            assert info.lm_nfev >= 2000, (m, info.lm_nfev, info.lm_info)
        assert 30 <= info.lm_nfev <= 5000


def test_config5_us_iterative_fit_1M(ctx):
    """BASELINE configs[4] at full size with the reference's default ITERATIVE fit: one bench batch (4096
    hypotheses over 1 M frames, 50 % outlier frames), Levenberg-Marquardt over the ~500 k frames of the winner's
    consensus set with the reference's 1e-15 tolerances.  MINPACK at this size exhausts its 5000 evaluations
    (fixtures, previous test) while still creeping along the valley at a relative gradient of ~1e-5, so what is
    checked is the iterate itself: the relative gradient |J_p . f| / (|J_p| |f|) on the whole consensus set (through
    the device's literal f / J pass, which is itself compared with the oracle's on a subset) is at MINPACK's own
    level, the cost is no larger than the analytic fit's, and the reference's success / failure convention holds."""
    n, H = 1_000_000, 4096
    rec, truth, lab = synth.us_single_fast(n, 0.5)
    ctx.set_model(L.US_SINGLE, 3, 3.0, L.LS_ITERATIVE).upload(rec)
    r = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    info = r["info"]
    assert info.best_votes > 0.4 * n
    wmask = r["consensus"].astype(bool)
    ctx.set_mask(r["consensus"])
    got, fi = ctx.ls_fit(use_mask=True)
    assert fi.lm_nfev >= 30 and fi.lm_info in (1, 2, 3, 4, 5)
    assert (len(got) > 0) == (1 <= fi.lm_info <= 4)
    # the batch's own fit ran the same minimisation from the same analytic start -- to rounding: since r04 the batch
    # takes the start's moment block from the matrix-core pass (k_mask_moments_us_mfma: another summation order), and
    # 5000 evaluations along the valley carry a last-bit difference of the start into the 9th digit of the cost
    assert (fi.lm_info, fi.lm_nfev) == (info.fit.lm_info, info.fit.lm_nfev)
    assert abs(fi.cost - info.fit.cost) <= 1e-7 * info.fit.cost
    ctx.set_option("us_mask_mfma", 0)   # ... and with the per-lane pass, bit for bit (fixed-order sums: deterministic)
    r_lane = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    ctx.set_option("us_mask_mfma", 1)
    assert r_lane["info"].fit.cost == fi.cost and np.array_equal(r_lane["consensus"], r["consensus"])
    assert (r["status"] == L.OK) == (len(got) > 0)
    x = ctx.last_iterate[:11]
    nlm = 11
    ntri = nlm * (nlm + 1) // 2
    # first-order optimality on the whole consensus set, through the device's own LM pass at x
    blk = ctx.moments(x, phase=1, use_mask=True)
    gdev = blk[1 + ntri:1 + ntri + nlm]
    jcol = np.sqrt(np.array([blk[1 + sum(nlm - q for q in range(p))] for p in range(nlm)]))   # sqrt(diag J^T J)
    assert np.all(np.abs(gdev) <= 2e-4 * jcol * np.sqrt(blk[0])), gdev / (jcol * np.sqrt(blk[0]))
    assert abs(blk[0] - fi.cost) <= 1e-9 * fi.cost
    # ... and that pass against the oracle's f / J (SinglePointTarget...cxx:415-658) on a 100 k-frame subset
    sel = np.ascontiguousarray(rec[wmask][:100_000])
    F = O.UsFunction(O.US_SINGLE, sel)
    f, J = F.f(x), F.jac(x)
    ctx.upload(sel)
    bs = ctx.moments(x, phase=1)
    JtJ = J.T @ J
    want = np.concatenate([[f @ f], JtJ[np.triu_indices(nlm)], J.T @ f])
    scale = np.concatenate([[f @ f], np.sqrt(np.outer(np.diag(JtJ), np.diag(JtJ)))[np.triu_indices(nlm)],
                            np.sqrt(np.diag(JtJ) * (f @ f))])
    assert np.all(np.abs(bs - want) <= 1e-9 * scale), np.abs((bs - want) / scale).max()
    ctx.set_model(L.US_SINGLE, 3, 3.0, L.LS_ANALYTIC).upload(rec)
    ctx.set_mask(r["consensus"])
    ana, _ = ctx.ls_fit(use_mask=True)
    cost_ana = ctx.stats(ana, use_mask=True)[3]
    assert fi.cost <= cost_ana * (1 + 1e-12)


def test_plane_phantom_1M_frames_on_bench_path(ctx):
    """PlanePhantomUSCalibration at the size `bench.py --workload phantom` (and the fifth leg of the default run)
    times -- 1 M frames, 4096 hypotheses (31-frame null-vector solves), the agree() scan on the fp16 matrix cores
    (csrc/phantom_h16.h; reference PlanePhantomUSCalibrationParametersEstimator.cxx:73-135): EVERY one of the 4096 full
    counts against the oracle, the winner's consensus mask bit-exact, the early exit against the full count."""
    n, H = 1_000_000, 4096
    rec = synth.plane_phantom_fast(n, 0.05, pixel_sigma=0.05)[0]    # bench.py: make_data("phantom", 1 M, .)
    oc = O.cfg(O.PHANTOM, 0, 2.0, 0)
    ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ANALYTIC).upload(rec)
    ctx.set_option("scan_bound", 0)
    r = ctx.batch_fit(SEED, 0, H, want_consensus=True)
    assert r["status"] == L.OK
    assert b"fp32 filter used" not in ctx._lib.lsqr_last_error(ctx._h)   # the fp16 filter ran, no worklist overflow
    info = r["info"]
    par, valid, votes = ctx.hypotheses()
    bi = int(info.best_index)
    vv = np.where(valid > 0, votes, 0)
    assert valid.mean() > 0.99 and info.best_votes == vv.max() and bi == int(np.argmax(vv))
    all_want = O.scan_many(oc, par, valid, rec)
    bad = np.flatnonzero(vv != all_want)
    assert len(bad) == 0, "full count differs from the oracle at %d of %d hypotheses (first h=%d)" % (len(bad), H, bad[0])
    wcnt, wmask = O.scan(oc, par[bi], rec)
    assert wcnt == info.best_votes and np.array_equal(r["consensus"], wmask)
    assert wcnt > 0.9 * n                                            # 5 % off-plane frames
    w = _early_exit_equals_full(ctx, r, valid, votes, H, n)
    assert w["row_hypothesis_pairs"] < H * n
    ctx.set_option("scan_bound", 1)
