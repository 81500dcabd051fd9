"""The plane phantom's minimal solves (PlanePhantomUSCalibrationParametersEstimator.cxx:137-355: exactly 31 frames, the
null vector of the 31 x 31 system, then the parameter extraction) by LU + inverse iteration (csrc/phantom.h:
k_estimate_phantom_lu, r05) against the one-sided Jacobi SVD it replaces as the default and against the oracle's SVD:
the same hypotheses valid, the same 41 parameters (T1's entries and the products derived from them up to the null
vector's common sign), and the Jacobi kernel still decides whatever the fast path refuses."""
import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu


def _align(p, q):
    """the null vector's sign is arbitrary: the products 11..40 and par[2] flip together, the row R1 of the flipped
    vector is -R1 and its two Euler angles (-omega_y, omega_x +- pi) -- unless omega_y is within the reference's
    small angle of +-pi/2 (.cxx:240-260), where omega_x is SET to zero whatever the sign"""
    q = q.copy()
    blk = list(range(11, 41)) + [2]
    if np.dot(p[blk], q[blk]) < 0:
        q[blk] = -q[blk]
        if abs(abs(q[0]) - np.pi / 2) > 0.008726535498373935:
            q[1] = q[1] - np.pi if abs(q[1] - np.pi - p[1]) < abs(q[1] + np.pi - p[1]) else q[1] + np.pi
        q[0] = -q[0]
    return q


def _dist(p, q):
    """largest difference of the 41 parameters relative to max(|p_i|, 1e-3 max |p|); the five angles modulo 2 pi"""
    q = _align(p, q)
    d = np.abs(q - p)
    ang = [0, 1, 6, 7, 8]
    d[ang] = np.minimum(d[ang], np.abs(2 * np.pi - d[ang]))
    return float(np.max(d / np.maximum(np.abs(p), 1e-3 * np.abs(p).max())))


def _estimate(ctx, data, H, fast, seed=0xC0FFEE):
    ctx.set_option("phantom_fast_solve", fast)
    ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ANALYTIC).upload(data)
    ctx.hypotheses_sample(seed, 0, H)
    par, valid, _ = ctx.hypotheses(votes=False)
    return par, valid


def test_fast_null_vectors_equal_the_jacobi_svd_and_the_oracle():
    data, truth, lab = synth.plane_phantom_fast(100_000, 0.05, seed=11, pixel_sigma=0.05)
    H = 2048
    with Context(0) as ctx:
        pj, vj = _estimate(ctx, data, H, 0)
        pf, vf = _estimate(ctx, data, H, 1)
        pl, vl = _estimate(ctx, data, H, 3)      # iteration limit 3: most hypotheses go to the Jacobi kernel behind it
    assert np.array_equal(vj, vf) and np.array_equal(vj, vl) and vj.sum() > 0.95 * H
    worst = 0.0
    for h in np.flatnonzero(vj):
        for other in (pf, pl):
            worst = max(worst, _dist(pj[h], other[h]))
    assert worst < 1e-7, worst     # (r05: the fast path's vector is corrected from a double-double residual)
    # the oracle's SVD on a sample of the same subsets
    oc = O.cfg(O.PHANTOM, 0, 2.0, 0)
    subs = O.ctr_subsets(0xC0FFEE, 0, H, len(data), 31)
    for h in list(np.flatnonzero(vj)[:40]):
        want = O.estimate(oc, data[subs[h]])
        assert len(want) == 41
        assert _dist(want, pf[h]) < 1e-7, h


def test_fast_path_votes_and_step_equal_the_jacobi_path():
    """a whole step either way: the winner's votes and the consensus set are counted on the device's own models, which
    agree to rounding -- same winner, same set on this upload"""
    data = synth.plane_phantom_fast(200_000, 0.05, seed=12, pixel_sigma=0.05)[0]
    res = []
    with Context(0) as ctx:
        for fast in (0, 1):
            ctx.set_option("phantom_fast_solve", fast)
            ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ANALYTIC).upload(data)
            res.append(ctx.batch_fit(5, 0, 1024, want_consensus=True))
    a, b = res
    assert a["info"].best_index == b["info"].best_index and abs(int(a["info"].best_votes) - int(b["info"].best_votes)) <= 2
    assert np.count_nonzero(a["consensus"] != b["consensus"]) <= 2


def test_degenerate_subsets_take_the_same_decision():
    """a subset with a duplicated frame (an exactly dependent row) and one with an index out of range: valid flags equal"""
    data = synth.plane_phantom_fast(5_000, 0.0, seed=13, pixel_sigma=0.05)[0]
    subs = O.ctr_subsets(3, 0, 64, len(data), 31).astype(np.uint32)
    subs[5, 7] = subs[5, 3]            # duplicate frame
    data2 = data.copy()
    data2[subs[9, 1]] = data2[subs[9, 0]]   # two different indices, identical records
    out = []
    with Context(0) as ctx:
        for fast in (0, 1):
            ctx.set_option("phantom_fast_solve", fast)
            ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ANALYTIC).upload(data2)
            ctx.hypotheses_from_subsets(subs)
            out.append(ctx.hypotheses(votes=False))
    (pj, vj, _), (pf, vf, _) = out
    assert np.array_equal(vj, vf)
    for h in np.flatnonzero(vj):
        if h in (5, 9):
            continue   # whatever vector an (almost) rank-deficient system yields, both paths call it valid or not alike
        assert _dist(pj[h], pf[h]) < 1e-6, h
