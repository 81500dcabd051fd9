"""The iterative US fits as ONE persistent launch (lsqrrecipes_amd/csrc/lm_persist.h; reference:
SinglePointTargetUSCalibrationParametersEstimator.cxx:272-329 / :926-971).  Three drivers of the same minimisation --
two launches per evaluation with the host's step between them (lm_persist 0, r02-r04), the persistent kernel with the
host's step (1) and with the step on the device (2) -- must walk through the SAME iterates: the block sums are cut and
ordered identically whatever the number of resident workgroups, MINPACK's control flow is one source (lm_core.h), and
the trial points' sines and cosines are lsqr_sincos on both sides (small_linalg.h)."""
import ctypes as C

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

pytestmark = pytest.mark.gpu


def _info(ctx):
    out = (C.c_uint64 * 8)()
    tr = (C.c_uint64 * (4 * 64))()
    n = C.c_uint32(0)
    assert ctx._lib.lsqr_lm_persist_info(ctx._h, out, tr, 64, C.byref(n)) == L.OK
    return dict(mode=out[0], wgs=out[1], evals=out[2], status=out[3], kernel_us=out[4], fallbacks=out[5],
                trace=np.array(tr[:4 * n.value], dtype=np.uint64).reshape(-1, 4))


def _fit(ctx, model, data, mode, wgs=0, use_mask=False, **opts):
    ctx.set_option("lm_persist", mode)
    ctx.set_option("lm_persist_wgs", wgs)
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.set_model(model, 0, 3.0, L.LS_ITERATIVE).upload(data)
    # the persistent kernel (like the matrix-core launch path) reads the COMPACTED consensus set: always through a mask
    ctx.set_mask(np.ones(len(data), np.uint8) if use_mask is False else use_mask)
    fit, info = ctx.ls_fit(True)
    return fit, ctx.last_iterate.copy(), info.lm_info, info.lm_nfev, info.cost


@pytest.mark.parametrize("kind", ["single", "pointer"])
def test_three_drivers_one_minimisation(kind):
    """bit for bit: last iterate, info, nfev, cost -- at sizes of one tile, a ragged last tile, fewer blocks than
    workgroups, and several rounds of blocks per workgroup"""
    for m, seed in ((50, 21), (3_001, 22), (70_013, 23), (300_000, 24)):
        if kind == "single":
            data = synth.us_single_fast(m, 0.0, seed=seed)[0]
            model = L.US_SINGLE
        else:
            data = synth.us_pointer(min(m, 70_013), 0.0, seed=seed)[0]
            model = L.US_POINTER
        with Context(0) as ctx:
            ref = _fit(ctx, model, data, 0)
            for mode in (3, 2):          # 3: the persistent kernel with the host's step, always; 2: with the device's step
                for wgs in (0, 64, 7, 1):
                    got = _fit(ctx, model, data, mode, wgs)
                    inf = _info(ctx)
                    assert inf["mode"] == mode and inf["status"] == 2 and inf["evals"] == ref[3], (m, mode, wgs, inf)
                    assert got[2] == ref[2] and got[3] == ref[3], (m, mode, wgs, got[2:], ref[2:])
                    assert np.array_equal(got[1], ref[1]), (m, mode, wgs)
                    assert np.array_equal(got[0], ref[0]) and got[4] == ref[4], (m, mode, wgs)
            assert _info(ctx)["fallbacks"] == 0
            # the default (1): this context is the only one fitting on the device -> the persistent kernel
            got = _fit(ctx, model, data, 1)
            assert np.array_equal(got[1], ref[1]) and got[2:4] == ref[2:4]
            assert _info(ctx)["mode"] == 1 and _info(ctx)["status"] == 2


def test_persistent_fit_through_the_mask():
    """the consensus set of a model (mask) compacted into tiles, then the fit: equal to the launch path, and the
    minimiser it stops at is the calibration the frames were made from (the existing fixtures pin the launch path's
    iterates against MINPACK: tests/test_gpu_parity.py, us_lm_vectors.npz)"""
    data, truth, lab = synth.us_single_fast(20_000, 0.3, seed=31)
    mask = lab.astype(np.uint8)
    with Context(0) as ctx:
        ref = _fit(ctx, L.US_SINGLE, data, 0, use_mask=mask)
        for mode in (3, 2):
            got = _fit(ctx, L.US_SINGLE, data, mode, use_mask=mask)
            assert got[2:4] == ref[2:4] and np.array_equal(got[1], ref[1])
    assert np.allclose(ref[1][:3], truth[:3], atol=0.5) and np.allclose(ref[1][9:11], truth[9:11], atol=2e-3)


def test_a_wait_that_expires_falls_back_to_the_launch_path():
    data = synth.us_single_fast(40_000, 0.0, seed=41)[0]
    with Context(0) as ctx:
        ref = _fit(ctx, L.US_SINGLE, data, 0)
        for mode in (3, 2):
            before = _info(ctx)["fallbacks"]
            got = _fit(ctx, L.US_SINGLE, data, mode, lm_persist_test_abort=17)
            inf = _info(ctx)
            assert inf["status"] == 3 and inf["fallbacks"] == before + 1 and inf["evals"] == 17
            assert got[2:4] == ref[2:4] and np.array_equal(got[1], ref[1])
        ctx.set_option("lm_persist_test_abort", 0)
        got = _fit(ctx, L.US_SINGLE, data, 3)
        assert _info(ctx)["status"] == 2 and np.array_equal(got[1], ref[1])


def test_four_contexts_fit_at_the_same_time():
    """four host threads with a context each (bench.py's C5 leg): the device's compute units are shared by tokens, every
    fit equals its own launch-path run"""
    import threading
    sets = [synth.us_single_fast(120_000 + 1000 * k, 0.0, seed=50 + k)[0] for k in range(4)]
    ctxs = [Context(0) for _ in range(4)]
    try:
        refs = [_fit(c, L.US_SINGLE, d, 0) for c, d in zip(ctxs, sets)]
        for mode in (3, 2):
            res, err = [None] * 4, []

            def work(k):
                try:
                    res[k] = (_fit(ctxs[k], L.US_SINGLE, sets[k], mode), _info(ctxs[k]))
                except Exception as e_:
                    err.append(e_)
            th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not err, err
            for k in range(4):
                got, inf = res[k]
                assert inf["status"] == 2 and inf["wgs"] <= 128, inf
                assert got[2:4] == refs[k][2:4] and np.array_equal(got[1], refs[k][1]), (mode, k)
        # the default (1): with four contexts fitting at once the launch path serves them (the better aggregate); same result
        res = [None] * 4
        th = [threading.Thread(target=lambda k=k: res.__setitem__(k, _fit(ctxs[k], L.US_SINGLE, sets[k], 1))) for k in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        for k in range(4):
            assert res[k][2:4] == refs[k][2:4] and np.array_equal(res[k][1], refs[k][1]), k
    finally:
        for c in ctxs:
            c.close()


def test_lsqr_sincos_is_the_hosts_on_the_device():
    """lm_finalize's rotation products (sines and cosines of the fitted angles) formed by the device step equal the
    host's to the bit -- the parameters 11..19 of the two persistent modes"""
    data = synth.us_single_fast(5_000, 0.0, seed=61)[0]
    with Context(0) as ctx:
        a = _fit(ctx, L.US_SINGLE, data, 3)
        b = _fit(ctx, L.US_SINGLE, data, 2)
        assert np.array_equal(a[1], b[1])
        if len(a[0]):
            assert len(a[0]) == 20 and np.array_equal(a[0], b[0])
