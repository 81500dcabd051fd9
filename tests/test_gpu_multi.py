"""Several devices from one process through the C ABI (lsqr_multi_*, include/lsqr_hip.h): two contexts on the
box's one GPU must reproduce the single-context entry points -- winner, stream index, consensus set and
iteration count bit for bit, the final fit to rounding (the moment block is summed slice-wise in rank order).
The exchanges are the ones a node with several GPUs performs (peer copies into rank 0's gather area + a
reduction kernel); only the devices coincide here."""
import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L
from lsqrrecipes_amd import synth
from lsqrrecipes_amd.context import Context, MultiContext

pytestmark = pytest.mark.gpu

CASES = [(L.PLANE, 3, 0, 0.5), (L.SPHERE, 3, L.LS_GEOMETRIC, 0.5), (L.SPHERE, 3, L.LS_ALGEBRAIC, 0.5),
         (L.LINE, 2, 0, 0.5), (L.DENSE, 8, 0, 0.1), (L.US_SINGLE, 0, L.LS_ANALYTIC, 3.0),
         (L.US_POINTER, 0, L.LS_ITERATIVE, 3.0), (L.PHANTOM, 0, L.LS_ITERATIVE, 2.0)]


def _data(model, dim):
    if model == L.DENSE:
        return synth.dense(40_000, 8, 0.3, seed=5)[0]
    if model == L.US_SINGLE:
        return synth.us_single_fast(40_000, 0.3, seed=5)[0]
    if model == L.US_POINTER:
        return synth.us_pointer(20_000, 0.3, seed=5)[0]
    if model == L.PHANTOM:
        return synth.plane_phantom_fast(20_000, 0.05, seed=5, pixel_sigma=0.02)[0]
    gen = {L.PLANE: synth.plane, L.SPHERE: synth.sphere, L.LINE: synth.line}[model]
    return gen(150_000, 0.5, seed=77, dim=dim)[0]


@pytest.mark.parametrize("model,dim,ls,delta", CASES)
@pytest.mark.parametrize("world", [2, 3])
def test_multi_batch_fit_equals_single_context(model, dim, ls, delta, world):
    data = _data(model, dim)
    H = 128 if model in (L.DENSE, L.PHANTOM) else 512
    with Context(0) as c1, MultiContext([0] * world) as m:
        c1.set_model(model, dim, delta, ls).upload(data)
        m.set_model(model, dim, delta, ls).upload(data)
        for b in range(2):
            want = c1.batch_fit(11, b * world * H, world * H, want_consensus=True)
            got = m.batch_fit(11, b * world * H, H, want_consensus=True)
            wi, gi = want["info"], got["info"]
            assert got["status"] == want["status"]
            assert (gi.best_votes, gi.best_index, gi.fit.n_used) == (wi.best_votes, wi.best_index, wi.fit.n_used)
            assert np.array_equal(got["consensus"], want["consensus"])
            assert len(got["params"]) == len(want["params"])
            if len(want["params"]):
                tol = 1e-6 if model in (L.PHANTOM, L.US_POINTER) else 1e-9   # LM paths: a different summation tree
                assert np.allclose(got["params"], want["params"], rtol=tol, atol=tol * max(1.0, np.abs(want["params"]).max()))


@pytest.mark.parametrize("model,dim,ls,delta", [(L.PLANE, 3, 0, 0.5), (L.SPHERE, 3, L.LS_GEOMETRIC, 0.5),
                                                (L.US_SINGLE, 0, L.LS_ANALYTIC, 3.0)])
def test_multi_ransac_equals_single_context(model, dim, ls, delta):
    data = _data(model, dim)
    with Context(0) as c1, MultiContext([0, 0]) as m:
        c1.set_model(model, dim, delta, ls).upload(data)
        m.set_model(model, dim, delta, ls).upload(data)
        for seed in (3, 4):
            want = c1.ransac(0.999, seed=seed)
            got = m.ransac(0.999, seed=seed)
            wi, gi = want["info"], got["info"]
            assert got["status"] == want["status"] == L.OK
            assert (gi.iterations, gi.best_index, gi.best_votes) == (wi.iterations, wi.best_index, wi.best_votes)
            assert np.array_equal(got["consensus"], want["consensus"])
            assert np.allclose(got["params"], want["params"], rtol=1e-9, atol=1e-9 * max(1.0, np.abs(want["params"]).max()))
        # invalid input: untouched, status INVALID (RANSAC.hxx:16-19)
        assert m.ransac(1.0)["status"] == L.ERR_INVALID


def test_multi_upload_replicates_device_to_device():
    data = synth.plane(70_000, 0.4, seed=9)[0]
    with MultiContext([0, 0, 0]) as m:
        m.set_model(L.PLANE, 3, 0.5).upload(data)
        r1 = m.batch_fit(5, 0, 256, want_consensus=True)
        m.upload(data[::-1].copy())            # a second upload replaces every replica
        r2 = m.batch_fit(5, 0, 256, want_consensus=True)
        assert r1["info"].best_votes > 0 and r2["info"].best_votes > 0
        with Context(0) as c1:
            c1.set_model(L.PLANE, 3, 0.5).upload(data[::-1].copy())
            w2 = c1.batch_fit(5, 0, 768, want_consensus=True)
        assert np.array_equal(r2["consensus"], w2["consensus"])


@pytest.mark.parametrize("model,dim,ls,delta", [(L.PLANE, 3, 0, 0.5), (L.SPHERE, 3, L.LS_GEOMETRIC, 0.5),
                                                (L.DENSE, 8, 0, 0.1), (L.US_POINTER, 0, L.LS_ITERATIVE, 3.0)])
def test_rccl_transport_equals_the_peer_copies(model, dim, ls, delta, monkeypatch):
    """LSQR_MULTI_TRANSPORT=rccl (one communicator per device: ncclCommInitAll; winner = ncclAllReduce MAX, moment
    blocks = ncclAllReduce SUM) at the ONE device this box has: librccl loads, the communicator comes up, is proved by an
    all-reduce, and every result equals the default transport's (a world of one leaves nothing to re-associate).
    More than one device has never run this code: no hardware scaling curve exists."""
    data = _data(model, dim)
    H = 128 if model == L.DENSE else 512
    res = {}
    for tr in ("peer-copy", "rccl"):
        if tr == "rccl":
            monkeypatch.setenv("LSQR_MULTI_TRANSPORT", "rccl")
        else:
            monkeypatch.delenv("LSQR_MULTI_TRANSPORT", raising=False)
        with MultiContext([0]) as m:
            name, secs = m.transport()
            assert name == tr and (secs > 0) == (tr == "rccl")
            m.set_model(model, dim, delta, ls).upload(data)
            b = m.batch_fit(0xBEEF, 1000, H, want_consensus=True)
            r = m.ransac(0.999, seed=5)
            res[tr] = (b, r, secs)
    b0, r0, _ = res["peer-copy"]
    b1, r1, secs = res["rccl"]
    print("rccl bring-up at one device: %.3f s" % secs)
    assert b0["info"].best_votes == b1["info"].best_votes and b0["info"].best_index == b1["info"].best_index
    assert np.array_equal(b0["consensus"], b1["consensus"]) and np.array_equal(b0["params"], b1["params"])
    assert r0["info"].iterations == r1["info"].iterations and np.array_equal(r0["consensus"], r1["consensus"])
    assert np.array_equal(r0["params"], r1["params"])
    assert b0["info"].fit.lm_nfev == b1["info"].fit.lm_nfev


def test_rccl_transport_refuses_a_device_listed_twice(monkeypatch):
    monkeypatch.setenv("LSQR_MULTI_TRANSPORT", "rccl")
    with pytest.raises(L.LsqrError):
        MultiContext([0, 0])


def test_sharded_dense_fit_on_an_ill_conditioned_system_equals_the_single_device_fit():
    """with the rows at hand a one-GPU fit of an ill-conditioned system takes the double-double route
    (1e-6 against the reference's SVD to cond 1e10); the sharded paths only hold the summed Gram block.  They now see
    the refused pivot (lsqr_fit_info.reserved == 2) and fit again from the replicated rows: cond(A) = 1e7, two and three
    contexts against one -- parameters equal, and within 1e-6 of the oracle's SVD pseudo-inverse over the same set."""
    from oracle import pyoracle as O
    from tests.test_gpu_dense_cond import system, rel
    rows, x_true = system(20_000, 16, 1e7, 77, 1e-4)
    g = np.random.default_rng(9)
    out = g.choice(len(rows), 4000, replace=False)
    rows[out, 16] += g.uniform(1.0, 50.0, 4000) * g.choice([-1.0, 1.0], 4000)
    with Context(0) as c1:
        c1.set_model(L.DENSE, 16, 0.01).upload(rows)
        r1 = c1.batch_fit(0xD15E, 0, 512, want_consensus=True)
        assert r1["status"] == L.OK and r1["info"].fit.reserved == 1          # the double-double route ran
        want = O.ls(O.cfg(O.DENSE, 16, 0.01), rows, r1["consensus"])
        assert rel(r1["params"], want) < 1e-6
        # the block alone: flagged, and visibly worse than the bar
        win, _ = c1.hypothesis(int(r1["info"].best_index))
        c1.mask(win, want_mask=False)
        blk = c1.moments(np.zeros(3), phase=0, use_mask=True)
        fit_blk, info_blk = c1.solve_moments(blk, np.zeros(3))
        assert info_blk.reserved == 2
    for world in (2, 3):
        with MultiContext([0] * world) as m:
            m.set_model(L.DENSE, 16, 0.01).upload(rows)
            # (the batch is world x 512 hypotheses: not the single context's batch -- compare the fits through the
            # oracle of each consensus set, and the flag)
            rm = m.batch_fit(0xD15E, 0, 512, want_consensus=True)
            assert rm["status"] == L.OK and rm["info"].fit.reserved == 1, rm["info"].fit.reserved
            wantm = O.ls(O.cfg(O.DENSE, 16, 0.01), rows, rm["consensus"])
            assert rel(rm["params"], wantm) < 1e-6
