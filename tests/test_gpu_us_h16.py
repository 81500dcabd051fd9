"""The US calibrations' agree() scan on the fp16 matrix cores (lsqrrecipes_amd/csrc/us_h16.h; reference:
SinglePointTargetUSCalibrationParametersEstimator.cxx:74-107 / :728-766).  A filter: whatever it cannot decide goes through
the exact fp64 predicate, so every vote must equal the packed fp32 filter's, the exact kernel's and the oracle's."""
import os
import re
import subprocess

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu


def _votes(ctx, model, data, delta, seed, H, mfma, filt=1):
    ctx.set_option("us_mfma", mfma)
    ctx.set_option("scan_filter", filt)
    ctx.set_model(model, 0, delta, L.LS_ANALYTIC).upload(data)
    ctx.hypotheses_sample(seed, 0, H)
    ctx.scan()
    return ctx.hypotheses()


@pytest.mark.parametrize("kind", ["single", "pointer"])
def test_us_h16_votes_equal_the_other_paths_and_the_oracle(kind):
    """ragged sizes (frames not a multiple of 512, hypotheses not a multiple of 32), thresholds from far below the
    noise to far above the data, minimal solves on subsets with outlier frames (scale factors in the thousands)"""
    if kind == "single":
        data = synth.us_single_fast(70_013, 0.3, seed=5)
        model, omodel = L.US_SINGLE, O.US_SINGLE
    else:
        data = synth.us_pointer(20_013, 0.3, seed=6)[0]
        model, omodel = L.US_POINTER, O.US_POINTER
    if isinstance(data, tuple):
        data = data[0]
    H = 333
    with Context(0) as ctx:
        for delta in (3.0, 1e-3, 1e4):
            par, valid, v16 = _votes(ctx, model, data, delta, 77, H, 1)
            assert b"fp32 filter used" not in ctx._lib.lsqr_last_error(ctx._h)
            _, v2, v32 = _votes(ctx, model, data, delta, 77, H, 0)
            _, v3, vex = _votes(ctx, model, data, delta, 77, H, 0, filt=0)
            assert np.array_equal(valid, v2) and np.array_equal(valid, v3)
            assert np.array_equal(v16, v32), (kind, delta)
            assert np.array_equal(v16, vex), (kind, delta)
            oc = O.cfg(omodel, 0, delta, 0)
            for h in (0, 1, 2, 100, 332):
                if valid[h]:
                    assert v16[h] == O.scan(oc, par[h], data)[0], (kind, delta, h)
        ctx.set_option("us_mfma", 1)
        ctx.set_option("scan_filter", 1)


def test_us_h16_delta_on_a_frames_own_distance():
    """delta^2 placed exactly on one (frame, hypothesis) pair's reference squared distance, and one ulp either side:
    strict '<' through the worklist"""
    data = synth.us_single(20_000, 0.2, seed=9)[0]
    with Context(0) as ctx:
        par, valid, votes = _votes(ctx, L.US_SINGLE, data, 3.0, 5, 64, 1)
        h = int(np.argmax(votes))
        res = float(ctx.residuals(par[h], 777, 778)[0])      # sqrt of the reference's squared distance of frame 777
        assert res > 0
        for delta in (res, np.nextafter(res, np.inf), np.nextafter(res, 0.0), res * (1 + 1e-7), res * (1 - 1e-7)):
            _, _, a = _votes(ctx, L.US_SINGLE, data, float(delta), 5, 64, 1)
            _, _, b = _votes(ctx, L.US_SINGLE, data, float(delta), 5, 64, 0, filt=0)
            assert np.array_equal(a, b), delta
        ctx.set_option("us_mfma", 1)
        ctx.set_option("scan_filter", 1)


def test_phantom_h16_votes_equal_the_other_paths_and_the_oracle():
    """plane phantom (lsqrrecipes_amd/csrc/phantom_h16.h; reference: PlanePhantomUSCalibrationParametersEstimator.cxx:73-135):
    the 31-term error as one matrix product; ragged sizes, thresholds from far below the noise to above every frame,
    the early-exit driver (scan_bound) and the plain one"""
    data = synth.plane_phantom_fast(70_013, 0.05, seed=15, pixel_sigma=0.05)[0]
    H = 333
    with Context(0) as ctx:
        for delta in (2.0, 1e-3, 1e4):
            par, valid, v16 = _votes(ctx, L.PHANTOM, data, delta, 78, H, 1)
            assert b"fp32 filter used" not in ctx._lib.lsqr_last_error(ctx._h)
            _, v2, v32 = _votes(ctx, L.PHANTOM, data, delta, 78, H, 0)
            _, v3, vex = _votes(ctx, L.PHANTOM, data, delta, 78, H, 0, filt=0)
            assert np.array_equal(valid, v2) and np.array_equal(valid, v3)
            assert np.array_equal(v16, v32), delta
            assert np.array_equal(v16, vex), delta
            oc = O.cfg(O.PHANTOM, 0, delta, 0)
            for h in (0, 1, 100, 332):
                if valid[h]:
                    assert v16[h] == O.scan(oc, par[h], data)[0], (delta, h)
        ctx.set_option("us_mfma", 1)
        ctx.set_option("scan_filter", 1)


def test_phantom_h16_delta_on_a_frames_own_error():
    """delta placed exactly on one (frame, hypothesis) pair's |err|, and one ulp either side"""
    data = synth.plane_phantom_fast(20_000, 0.05, seed=19, pixel_sigma=0.05)[0]
    with Context(0) as ctx:
        par, valid, votes = _votes(ctx, L.PHANTOM, data, 2.0, 5, 64, 1)
        h = int(np.argmax(votes))
        res = float(ctx.residuals(par[h], 777, 778)[0])
        assert res > 0
        for delta in (res, np.nextafter(res, np.inf), np.nextafter(res, 0.0), res * (1 + 1e-7), res * (1 - 1e-7)):
            _, _, a = _votes(ctx, L.PHANTOM, data, float(delta), 5, 64, 1)
            _, _, b = _votes(ctx, L.PHANTOM, data, float(delta), 5, 64, 0, filt=0)
            assert np.array_equal(a, b), delta
        ctx.set_option("us_mfma", 1)
        ctx.set_option("scan_filter", 1)


def test_phantom_h16_standalone_check():
    """tools/ph16_bench (built by __graft_entry__.build()): every vote of 1001 hypotheses x 99 937 synthetic frames (ragged:
    the last pass, the last hypothesis tile) against a brute-force count with the reference's predicate"""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ph16_bench")
    if not os.path.exists(exe):
        pytest.skip("tools/ph16_bench not built")
    out = subprocess.run([exe, "99937", "1001", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(r"votes: 0 of 1001 hypotheses differ", out.stdout), out.stdout


@pytest.mark.parametrize("kind", ["single", "phantom"])
def test_refused_subsets_are_never_counted_and_do_not_flood_the_worklist(kind):
    """150 of 333 minimal subsets are one frame K times: the solve is refused, the parameters are NaN, the reference's
    comparison with a NaN is false for every frame.  The fp16 filters mark such a hypothesis 'never counted' instead of
    sending every one of its frames to the exact path (10 M worklist entries here: the segments hold 1 M)"""
    if kind == "single":
        data = synth.us_single_fast(70_013, 0.3, seed=25)
        model, K, delta = L.US_SINGLE, 4, 3.0
    else:
        data = synth.plane_phantom_fast(70_013, 0.05, seed=26, pixel_sigma=0.05)[0]
        model, K, delta = L.PHANTOM, 31, 2.0
    data = data[0] if isinstance(data, tuple) else data
    H = 333
    subs = O.ctr_subsets(5, 0, H, len(data), K).copy()
    subs[:150, 1:] = subs[:150, :1]      # one frame K times
    with Context(0) as ctx:
        def votes(mfma, filt):
            ctx.set_option("us_mfma", mfma)
            ctx.set_option("scan_filter", filt)
            ctx.set_model(model, 0, delta, L.LS_ANALYTIC).upload(data)
            ctx.hypotheses_from_subsets(subs)
            ctx.scan()
            return ctx.hypotheses()
        par, valid, v16 = votes(1, 1)
        msg = ctx._lib.lsqr_last_error(ctx._h)
        assert b"overflow" not in msg and b"fp32 filter used" not in msg, msg
        assert int((valid[:150] == 0).sum()) >= 100 and not np.isfinite(par[:150][valid[:150] == 0]).any()
        assert not v16[:150][valid[:150] == 0].any()
        _, v2, vex = votes(0, 0)
        assert np.array_equal(valid, v2) and np.array_equal(v16, vex)
        ctx.set_option("us_mfma", 1)
        ctx.set_option("scan_filter", 1)


@pytest.mark.parametrize("kind", ["single", "pointer"])
def test_us_minimal_solves_elimination_against_svd(kind):
    """k_estimate_us (reference: SinglePointTargetUSCalibrationParametersEstimator.cxx:137-201, SVD pseudo-inverse with
    the rank threshold FLT_EPSILON): the elimination fast path (`us_fast_solve` 1, default) against the SVD path (0) --
    same hypotheses accepted and refused (degenerate subsets included), parameters equal to 1e-9 relative"""
    if kind == "single":
        data = synth.us_single_fast(50_000, 0.3, seed=31)
        model, K = L.US_SINGLE, 4
    else:
        data = synth.us_pointer(20_000, 0.3, seed=32)[0]
        model, K = L.US_POINTER, 3
    data = data[0] if isinstance(data, tuple) else data
    H = 2048
    subs = O.ctr_subsets(9, 0, H, len(data), K).copy()
    subs[:40, 1] = subs[:40, 0]            # a frame twice: rank-deficient, refused by both
    with Context(0) as ctx:
        res = []
        for fast in (1, 0):
            ctx.set_option("us_fast_solve", fast)
            ctx.set_model(model, 0, 3.0, L.LS_ANALYTIC).upload(data)
            ctx.hypotheses_from_subsets(subs)
            ctx.scan()
            res.append(ctx.hypotheses())
        ctx.set_option("us_fast_solve", 1)
    (p1, v1, c1), (p0, v0, c0) = res
    assert np.array_equal(v1, v0) and not v1[:40].any() and v1[40:].mean() > 0.9
    ok = v1.astype(bool)
    scale = np.maximum(np.abs(p0[ok]).max(axis=0), 1e-300)
    assert (np.abs(p1[ok] - p0[ok]) / scale).max() < 1e-9
    assert np.abs(c1.astype(np.int64) - c0.astype(np.int64)).max() <= 2   # (a frame within 1e-10 of delta may flip)


@pytest.mark.parametrize("kind", ["single", "dense"])
def test_votes_of_a_batch_larger_than_the_rechecks_vote_table(kind):
    """the exact re-check adds its votes up per workgroup in a direct-mapped LDS table keyed by the hypothesis index
    (4096 slots for the calibrations, 1024 for the dense system: r05); with more hypotheses than slots keys collide and
    the colliding ones go to the global counters -- every vote must still equal the exact kernel's"""
    with Context(0) as ctx:
        if kind == "single":
            data = synth.us_single_fast(40_000, 0.3, seed=15)[0]
            _, valid, v16 = _votes(ctx, L.US_SINGLE, data, 3.0, 99, 9000, 1)
            _, v2, vex = _votes(ctx, L.US_SINGLE, data, 3.0, 99, 9000, 0, filt=0)
            ctx.set_option("us_mfma", 1)
            ctx.set_option("scan_filter", 1)
        else:
            data = synth.dense(30_000, 64, 0.05, seed=16)[0]
            out = []
            for f32, filt in ((2, 1), (0, 0)):
                ctx.set_option("dense_f32", f32)
                ctx.set_option("scan_filter", filt)
                ctx.set_model(L.DENSE, 64, 0.1, L.LS_ALGEBRAIC).upload(data)
                ctx.hypotheses_sample(98, 0, 2500)
                ctx.scan()
                out.append(ctx.hypotheses())
            (_, valid, v16), (_, v2, vex) = out
            ctx.set_option("dense_f32", 2)
            ctx.set_option("scan_filter", 1)
        assert np.array_equal(valid, v2) and valid.sum() > 0
        assert np.array_equal(v16, vex)
