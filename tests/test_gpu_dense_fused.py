"""Dense system: consensus mask + normal-equation block in ONE pass (csrc/dense.h: k_mask_syrk_dense; run with -m gpu).
The mask must be bit-identical to the reference's running sum (DenseLinear...Estimator.hxx:111-119) -- the kernel
evaluates four interleaved chains and re-walks a tile serially when a row sits inside the proven band; the block of
sums must equal the separate matrix-core SYRK's up to rounding (different summation tree)."""
import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m", [(64, 70_001), (64, 5_000), (33, 40_000), (16, 30_000), (5, 20_000), (64, 63)])
def test_fused_mask_and_block_equal_the_two_kernel_path(n, m):
    rows, x_true, _ = synth.dense(m, n, 0.1, seed=3)
    # a model near the truth: about a third of the rows agree, many residuals close to delta
    x = x_true * (1 + 1e-3 * np.random.default_rng(1).standard_normal(n))
    oc = O.cfg(O.DENSE, n, 0.1)
    want_cnt, want_mask = O.scan(oc, x, rows)
    res = {}
    for name, opts in (("fused", {}), ("serial_forced", {"dense_mask_band": 2_000_000_000}), ("two_kernels", {"fuse_mask": 0})):
        with Context(0) as c:
            c.set_model(L.DENSE, n, 0.1, 0).upload(rows)
            for k, v in opts.items():
                c.set_option(k, v)
            mask, cnt = c.mask(x)
            fit, info = c.ls_fit(use_mask=True)
            res[name] = (mask.copy(), cnt, fit.copy(), info.n_used)
    for name, (mask, cnt, fit, used) in res.items():
        assert cnt == want_cnt and np.array_equal(mask, want_mask), name
    # whole path (lsqr_batch_fit takes the fused kernel): winner mask and fit against the two-kernel run
    outs = []
    for opts in ({}, {"fuse_mask": 0}, {"dense_mask_band": 2_000_000_000}):
        with Context(0) as c:
            c.set_model(L.DENSE, n, 0.1, 0).upload(rows)
            for k, v in opts.items():
                c.set_option(k, v)
            if m < n:
                continue
            r = c.batch_fit(7, 0, 64, want_consensus=True)
            outs.append(r)
    for r in outs[1:]:
        assert r["info"].best_index == outs[0]["info"].best_index and r["info"].best_votes == outs[0]["info"].best_votes
        assert np.array_equal(r["consensus"], outs[0]["consensus"])
        assert (len(r["params"]) == 0) == (len(outs[0]["params"]) == 0)
        if len(r["params"]):
            scale = max(1.0, np.abs(outs[0]["params"]).max())
            assert np.abs(r["params"] - outs[0]["params"]).max() <= 1e-9 * scale
    if outs and len(outs[0]["params"]) and outs[0]["info"].best_votes >= 4 * n:
        sel = rows[outs[0]["consensus"].astype(bool)]
        want = np.linalg.lstsq(sel[:, :n], sel[:, n], rcond=None)[0]
        assert np.abs(outs[0]["params"] - want).max() <= 1e-6 * max(1.0, np.abs(want).max())


def test_fused_mask_rows_on_the_threshold():
    """residuals pushed onto delta to a few ulp: the band must send them to the serial evaluation"""
    n, m = 64, 20_000
    rng = np.random.default_rng(5)
    A = rng.uniform(-1, 1, (m, n))
    x = rng.uniform(-1, 1, n)
    oc = O.cfg(O.DENSE, n, 0.1)
    b = np.empty(m)
    for i in range(m):
        s = 0.0
        for k in range(n):               # the reference's running sum
            s += A[i, k] * x[k]
        off = 0.1 * (1 + (rng.integers(-4, 5)) * 2.2e-16) * (1 if i % 2 else -1)
        b[i] = s - off if i % 3 else s - 0.5 * off
    rows = np.ascontiguousarray(np.concatenate([A, b[:, None]], axis=1))
    want_cnt, want_mask = O.scan(oc, x, rows)
    assert 0 < want_cnt < m
    with Context(0) as c:
        c.set_model(L.DENSE, n, 0.1, 0).upload(rows)
        mask, cnt = c.mask(x)
        assert cnt == want_cnt and np.array_equal(mask, want_mask)


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("ncol,m,H", [(5, 4000, 300), (8, 20_000, 512), (33, 30_000, 257), (64, 60_000, 1024)])
def test_dense_minimal_solves_by_one_wave_are_bit_identical(ctx, ncol, m, H):
    """r04: k_estimate_dense_w4 (four hypotheses per workgroup, elimination by one wave each: wave_gepp_solve) against
    the workgroup kernel it replaces on the fast path -- every hypothesis bit for bit, validity included; subsets with
    a repeated row (singular: the elimination refuses, the SVD pseudo-inverse decides) take the same path as before;
    and the final fit's elimination (k_solve_dense) by one wave gives the same bits as well"""
    rows = synth.dense(m, ncol, 0.2, seed=40 + ncol)[0]
    ctx.set_model(L.DENSE, ncol, 0.1).upload(rows)
    subs = O.ctr_subsets(77, 0, H, m, ncol)
    subs[3, 1] = subs[3, 0]                       # a repeated row: rank deficient -> invalid
    subs[H - 1, ncol - 1] = subs[H - 1, 0]
    out = {}
    for wave in (3, 1, 0):   # 3: the 64 x 64 system in registers (k_estimate_dense_r64; else as 1), 1: in LDS, 0: workgroup
        ctx.set_option("dense_wave_solve", wave)
        ctx.hypotheses_from_subsets(subs)
        par, valid, _ = ctx.hypotheses(votes=False)
        ctx.scan()
        votes = ctx.hypotheses(params=False)[2].copy()
        r = ctx.batch_fit(5, 0, min(H, 256), want_consensus=True)
        out[wave] = (par.copy(), valid.copy(), votes, r["params"].copy(), r["info"].best_index)
    ctx.set_option("dense_wave_solve", 3)
    (p3, v3, k3, f3, b3) = out[3]
    (p1, v1, k1, f1, b1), (p0, v0, k0, f0, b0) = out[1], out[0]
    assert np.array_equal(v3, v0) and np.array_equal(p3[v3 > 0], p0[v0 > 0])  # the register elimination: bit for bit
    assert np.array_equal(k3, k0) and b3 == b0 and np.array_equal(f3, f0)
    assert np.array_equal(v1, v0) and not v1[3] and not v1[H - 1] and v1.sum() >= H - 4
    assert np.array_equal(p1[v1 > 0], p0[v0 > 0])            # bit for bit
    assert np.array_equal(k1, k0) and b1 == b0 and np.array_equal(f1, f0)
