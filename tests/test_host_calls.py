"""lsqr_agree_host / lsqr_estimate_host (include/lsqr_hip.h): ParametersEstimator::agree(parameters, datum) and
estimate() of one minimal subset evaluated on the HOST by the library's own per-model code (what the C++ drop-in's
agree() / estimate() call instead of an upload + launch per datum).  No GPU needed: compared with the CPU oracle
here (bit-exact where the reference's arithmetic is restated literally), and with the device in test_gpu_parity."""
import ctypes as C

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from oracle import pyoracle as O


def _agree(cfg, par, rec):
    a = C.c_int(-1)
    p = np.ascontiguousarray(par, dtype=np.float64)
    r = np.ascontiguousarray(rec, dtype=np.float64)
    st = L.load().lsqr_agree_host(C.byref(cfg), L.ptr(p), L.ptr(r), C.byref(a))
    return st, a.value


def _estimate(cfg, recs):
    r = np.ascontiguousarray(recs, dtype=np.float64)
    out = np.zeros(64)
    n = C.c_int(-1)
    st = L.load().lsqr_estimate_host(C.byref(cfg), L.ptr(r), r.shape[0], r.shape[1] * 8, L.ptr(out), C.byref(n))
    return st, out[:max(n.value, 0)].copy()


@pytest.mark.parametrize("model,omodel,gen,dim,exact", [
    (L.PLANE, O.PLANE, synth.plane, 3, True), (L.PLANE, O.PLANE, synth.plane, 2, False),
    (L.SPHERE, O.SPHERE, synth.sphere, 3, True), (L.SPHERE, O.SPHERE, synth.sphere, 2, True),
    (L.LINE, O.LINE, synth.line, 3, True), (L.LINE, O.LINE, synth.line, 2, True),
    (L.PLANE, O.PLANE, synth.plane, 5, False), (L.SPHERE, O.SPHERE, synth.sphere, 4, False)])
def test_host_estimate_and_agree_equal_the_oracle(model, omodel, gen, dim, exact):
    data = gen(4000, 0.4, seed=17, dim=dim)[0]
    cfg = L.ModelCfg(model, dim, 0.5, L.LS_GEOMETRIC, 0, 0.0)
    oc = O.cfg(omodel, dim, 0.5)
    k = L.load().lsqr_min_subset(C.byref(cfg))
    rng = np.random.default_rng(3)
    checked = 0
    for t in range(40):
        sub = data[rng.choice(len(data), k, replace=False)]
        st, par = _estimate(cfg, sub)
        want = O.estimate(oc, sub)
        assert (st == L.OK) == (len(want) > 0)
        if not len(want):
            continue
        if exact:
            assert np.array_equal(par, want)
        else:   # N-D branches go through a pseudo-inverse / null vector: sign and rounding may differ
            if model == L.PLANE:
                s = np.sign(par[:dim] @ want[:dim])
                assert np.allclose(s * par[:dim], want[:dim], rtol=1e-9, atol=1e-9)
            else:
                assert np.allclose(par, want, rtol=1e-8, atol=1e-8)
        for i in range(0, len(data), 37):
            st, a = _agree(cfg, want, data[i])
            assert st == L.OK and a == int(O.agree(oc, want, data[i]))
            checked += 1
    assert checked > 1000
    # a degenerate subset: empty vector (LSQR_EMPTY), as the reference's estimate()
    st, par = _estimate(cfg, np.repeat(data[:1], k, axis=0))
    assert st == L.EMPTY and len(par) == 0


def test_host_agree_for_records_with_wide_layouts():
    """dense rows and US frames: agree() on the host; their minimal solves are device kernels (LSQR_ERR_INVALID)"""
    rows, x_true, _ = synth.dense(300, 6, 0.2, seed=2)
    cfg = L.ModelCfg(L.DENSE, 6, 0.1, 0, 0, 0.0)
    oc = O.cfg(O.DENSE, 6, 0.1)
    for i in range(len(rows)):
        st, a = _agree(cfg, x_true, rows[i])
        assert st == L.OK and a == int(O.agree(oc, x_true, rows[i]))
    st, _ = _estimate(cfg, rows[:6])
    assert st == L.ERR_INVALID
    rec, truth, _ = synth.us_single_fast(400, 0.3, seed=4)
    cfg = L.ModelCfg(L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC, 0, 0.0)
    oc = O.cfg(O.US_SINGLE, 0, 3.0, 0)
    par = O.estimate(oc, rec[:4])
    assert len(par) == 20
    for i in range(len(rec)):
        st, a = _agree(cfg, par, rec[i])
        assert st == L.OK and a == int(O.agree(oc, par, rec[i]))
    st, _ = _estimate(cfg, rec[:4])
    assert st == L.ERR_INVALID


@pytest.mark.parametrize("dim", [5, 6, 9, 17, 33, 63])
def test_host_agree_dense_reads_only_dim_parameters(dim):
    """the dense model is compiled for padded widths 8/16/32/64; the caller's parameter vector has exactly `dim`
    entries.  Poison (NaN) right behind them must not be read: 0 * NaN in the padded dot product would turn every
    agree() into false (heap over-read in lsqr_agree_host)."""
    rows, x_true, _ = synth.dense(200, dim, 0.2, seed=dim)
    cfg = L.ModelCfg(L.DENSE, dim, 0.1, 0, 0, 0.0)
    oc = O.cfg(O.DENSE, dim, 0.1)
    buf = np.full(dim + 64, np.nan)
    buf[:dim] = x_true
    lib = L.load()
    got, want = [], []
    for i in range(len(rows)):
        a = C.c_int(-1)
        r = np.ascontiguousarray(rows[i])
        assert lib.lsqr_agree_host(C.byref(cfg), L.ptr(buf), L.ptr(r), C.byref(a)) == L.OK
        got.append(a.value)
        want.append(int(O.agree(oc, x_true, rows[i])))
    assert got == want and 0 < sum(want) < len(want)
