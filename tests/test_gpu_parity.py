"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.  Bars (BASELINE.json north_star): consensus
masks, vote counts, minimal-subset models and sampler output BIT-EXACT; final-fit parameters within
1e-6 relative (plane / line normals modulo sign, as the reference's own tests compare |dot|)."""
import os

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L
from lsqrrecipes_amd import synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu

REL = 1e-6  # tolerance stated by BASELINE.json north_star for estimated parameters

CASES = [(L.PLANE, 3), (L.PLANE, 2), (L.SPHERE, 3), (L.SPHERE, 2), (L.LINE, 3), (L.LINE, 2)]


def _data(model, dim, n, seed, outliers=0.4):
    if model == L.PLANE:
        return synth.plane(n, outliers, seed=seed, dim=dim)[0]
    if model == L.SPHERE:
        return synth.sphere(n, outliers, seed=seed, dim=dim)[0]
    return synth.line(n, outliers, seed=seed, dim=dim)[0]


def _params_close(model, dim, got, want):
    assert len(got) == len(want) and len(want) > 0
    if model in (L.PLANE, L.LINE):
        assert abs(abs(np.dot(got[:dim], want[:dim])) - 1.0) < REL
        s = np.sign(np.dot(got[:dim], want[:dim]))
        assert np.allclose(s * got[:dim], want[:dim], rtol=REL, atol=REL)
        if model == L.PLANE:
            # the point is only defined up to motion inside the plane: compare its offset along n
            assert abs(np.dot(got[dim:] - want[dim:], want[:dim])) < REL * max(1.0, np.abs(want[dim:]).max())
            assert np.allclose(got[dim:], want[dim:], rtol=REL, atol=1e-6)
        else:
            assert np.allclose(got[dim:], want[dim:], rtol=REL, atol=1e-6)
    else:
        assert np.allclose(got, want, rtol=REL, atol=1e-6)


@pytest.fixture(scope="module")
def ctx():
    c = Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("model,dim", CASES)
@pytest.mark.parametrize("n", [1, 63, 1000, 4097, 50_003])
def test_hypotheses_and_scan_bit_exact(ctx, model, dim, n):
    oc = O.cfg(model, dim, 0.5)
    k = O.lib().orc_min_subset(oc)
    if n < k:
        n = k
    data = _data(model, dim, n, 1000 + n + model)
    H = 96
    subs = O.ctr_subsets(5, 0, H, n, k)
    ctx.set_model(model, dim, 0.5).upload(data)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in range(H):
        want = O.estimate(oc, data[subs[h]])
        assert bool(valid[h]) == (len(want) > 0)
        if not valid[h]:
            assert votes[h] == 0 and np.all(np.isnan(par[h]))
            continue
        if model == L.PLANE and dim == 2:   # SVD null-vector branch: sign arbitrary (unpinned)
            assert abs(abs(par[h][:2] @ want[:2]) - 1) < 1e-12
            want = par[h]
        else:
            assert np.array_equal(par[h], want), "minimal-subset model differs at h=%d" % h
        cnt, _ = O.scan(oc, want, data)
        assert votes[h] == cnt, "vote count differs at h=%d" % h
    # winner (first max) and its mask
    packed, bv, bi = ctx.best()
    vv = np.where(valid > 0, votes, 0)
    assert bv == vv.max() and bi == int(np.argmax(vv))
    m, cnt = ctx.mask_from_hypothesis(bi)
    wcnt, wmask = O.scan(oc, par[bi], data)
    assert cnt == wcnt == bv and np.array_equal(m, wmask)


@pytest.mark.parametrize("n,k_model", [(10, (L.PLANE, 3)), (1000, (L.SPHERE, 3)),
                                       (10_000_000, (L.PLANE, 3)), (77, (L.LINE, 2))])
def test_device_sampler_bit_exact(ctx, n, k_model):
    model, dim = k_model
    data = np.zeros((min(n, 1000), dim))
    ctx.set_model(model, dim, 0.5)
    if n > 1000:
        data = np.zeros((n, dim))
    ctx.upload(data)
    subs = ctx.hypotheses_sample(1234, 10_000, 300, want_subsets=True)
    want = O.ctr_subsets(1234, 10_000, 300, n, ctx.K)
    assert np.array_equal(subs, want)


@pytest.mark.parametrize("model,dim", CASES)
def test_mask_and_fit_against_oracle(ctx, model, dim):
    for ls_type in ((L.LS_ALGEBRAIC, L.LS_GEOMETRIC) if model == L.SPHERE else (0,)):
        oc = O.cfg(model, dim, 0.5, ls_type)
        n = 20_011
        data = _data(model, dim, n, 555 + model + dim)
        k = O.lib().orc_min_subset(oc)
        ctx.set_model(model, dim, 0.5, ls_type).upload(data)
        subs = O.ctr_subsets(9, 0, 64, n, k)
        ctx.hypotheses_from_subsets(subs)
        ctx.scan()
        par, valid, votes = ctx.hypotheses()
        _, bv, bi = ctx.best()
        m, cnt = ctx.mask(par[bi])
        wcnt, wmask = O.scan(oc, par[bi], data)
        assert np.array_equal(m, wmask) and cnt == wcnt
        got, info = ctx.ls_fit(use_mask=True)
        want = O.ls(oc, data, wmask)
        _params_close(model, dim, got, want)
        # all-data fit (estimator.leastSquaresEstimate(data) without RANSAC)
        clean = _data(model, dim, 5000, 99 + model, outliers=0.0)
        ctx.upload(clean)
        got, info = ctx.ls_fit(use_mask=False)
        _params_close(model, dim, got, O.ls(oc, clean))
        # residual statistics
        st = ctx.stats(got)
        wst = O.stats(oc, got, clean)
        assert np.allclose(st, wst, rtol=1e-9, atol=1e-12)


def test_sphere_lm_info_and_cost(ctx):
    pts = synth.sphere(30_000, 0.0, seed=4242)[0]
    ctx.set_model(L.SPHERE, 3, 0.5, L.LS_GEOMETRIC).upload(pts)
    got, info = ctx.ls_fit()
    init = O.sphere_algebraic(3, pts)
    want, winfo, wnfev = O.sphere_geometric(3, pts, init)
    assert 1 <= info.lm_info <= 4 and 1 <= winfo <= 4
    assert abs(info.lm_nfev - wnfev) <= 3
    assert np.allclose(got, want, rtol=1e-9, atol=1e-8)
    res = np.linalg.norm(pts - want[:3], axis=1) - want[3]
    assert np.isclose(info.cost, (res ** 2).sum(), rtol=1e-9)


@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.SPHERE, 3), (L.LINE, 3), (L.SPHERE, 2)])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ransac_equals_serial_reference_loop(ctx, model, dim, seed):
    """RANSAC<T,S>::compute through the C ABI == the serial loop (oracle, pinned against the
    reference's RANSAC.hxx) for the same subset stream: iterations, winner, mask, fraction."""
    ls_type = L.LS_GEOMETRIC
    oc = O.cfg(model, dim, 0.5, ls_type)
    n = 6000 + seed
    data = _data(model, dim, n, 70 + seed + model, outliers=0.5)
    ctx.set_model(model, dim, 0.5, ls_type).upload(data)
    # (a) device sampler stream vs the oracle's restatement of it
    r = ctx.ransac(0.999, seed=seed)
    w = O.ransac(oc, data, 0.999, sampler="ctr", seed=seed, first=0)
    assert r["info"].iterations == w["iters"]
    assert r["info"].best_index == w["best_iter"] and r["info"].best_votes == w["best_votes"]
    assert r["fraction"] == w["fraction"]
    assert np.array_equal(r["consensus"], w["consensus"])
    _params_close(model, dim, r["params"], w["params"])
    # (b) an explicit subset stream with duplicates
    subs = O.ctr_subsets(seed + 50, 0, 3000, n, ctx.K)
    subs[5] = subs[1]
    subs[9] = subs[4][::-1]
    r = ctx.ransac(0.99, subsets=subs)
    w = O.ransac(oc, data, 0.99, sampler="list", subsets=subs)
    assert r["info"].iterations == w["iters"] and r["info"].best_index == w["best_iter"]
    assert np.array_equal(r["consensus"], w["consensus"])
    _params_close(model, dim, r["params"], w["params"])


@pytest.mark.parametrize("name", ["plane", "sphere", "circle", "line"])
def test_against_reference_golden_vectors(ctx, golden_dir, name):
    """Outputs of the REFERENCE's RANSAC.hxx (tests/golden/ransac_ref_vectors.npz).  The recorded
    subset list holds the non-duplicate draws of the reference run, in order; replaying it selects
    the same first-max winner, so the mask must be bit-identical and the LS parameters within tol."""
    rv = np.load(os.path.join(golden_dir, "ransac_ref_vectors.npz"))
    cfgv = rv[name + "_cfg"]
    model, dim, delta, ls_type = int(cfgv[0]), int(cfgv[1]), float(cfgv[2]), int(cfgv[3])
    data = rv[name + "_data"]
    ctx.set_model(model, dim, delta, ls_type).upload(data)
    for seed in (11, 12, 13):
        key = "%s_s%d_" % (name, seed)
        r = ctx.ransac(0.999, subsets=rv[key + "subsets"])
        assert r["fraction"] == rv[key + "fraction"][0]
        assert np.array_equal(r["consensus"], rv[key + "consensus"])
        _params_close(model, dim, r["params"], rv[key + "params"])


def test_exhaustive_against_reference_golden(ctx, golden_dir):
    rv = np.load(os.path.join(golden_dir, "ransac_ref_vectors.npz"))
    ctx.set_model(L.PLANE, 3, 0.5).upload(rv["exh_data"])
    r = ctx.ransac_exhaustive()
    assert r["fraction"] == rv["exh_fraction"][0]
    assert np.array_equal(r["consensus"], rv["exh_consensus"])
    _params_close(L.PLANE, 3, r["params"], rv["exh_params"])
    assert r["info"].iterations == 364


def test_error_conventions(ctx):
    data = synth.plane(100, 0.2)[0]
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    for p in (0.0, 1.0, -1.0, 2.0, float("nan")):
        r = ctx.ransac(p)
        assert r["status"] == L.ERR_INVALID and r["fraction"] == 0.0 and r["params"] is None
    ctx.upload(data[:2])                       # fewer observations than a minimal subset
    assert ctx.ransac(0.9)["status"] == L.ERR_INVALID
    r = ctx.ransac_exhaustive()                # exhaustive overload: cleared, returns 0
    assert r["status"] == L.EMPTY and len(r["params"]) == 0
    # every subset degenerate -> no consensus, parameters empty, consensus not written
    same = np.tile(np.array([[1.0, 2.0, 3.0]]), (20, 1))
    ctx.upload(same)
    r = ctx.ransac(0.9, subsets=O.ctr_subsets(1, 0, 50, 20, 3))
    assert r["status"] == L.EMPTY and r["fraction"] == 0.0 and r["consensus"] is None
    # too few points for a fit -> empty
    ctx.upload(data[:2])
    got, _ = ctx.ls_fit()
    assert len(got) == 0
    # unsupported model / stride errors are reported, not ignored
    with pytest.raises(L.LsqrError):
        ctx.set_model(L.PLANE, 9, 0.5)          # device models exist for dimensions 2..8
    with pytest.raises(L.LsqrError):
        ctx.set_model(L.SPHERE, 3, 0.5, ls_type=5)


def test_strided_records(ctx):
    """std::vector<T> with sizeof(T) larger than the payload (stride in bytes is honoured)."""
    data = synth.plane(3000, 0.3, seed=8)[0]
    wide = np.zeros((3000, 5))
    wide[:, :3] = data
    wide[:, 3:] = 1e300
    oc = O.cfg(O.PLANE, 3, 0.5)
    subs = O.ctr_subsets(2, 0, 32, 3000, 3)
    ctx.set_model(L.PLANE, 3, 0.5).upload(wide)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in range(32):
        assert votes[h] == O.scan(oc, par[h], data)[0]


def test_sphere_boundary_sqrt(ctx):
    """agree() compares sqrt(d2)-r with delta (SphereParametersEstimator.hxx:261-263): points
    within a few ulp of the decision boundary must classify exactly as on the CPU."""
    g = np.random.default_rng(3)
    c0 = np.array([10.5, -3.25, 7.125])
    r, delta = 123.456, 0.5
    u = g.normal(size=(40_000, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    rad = np.where(g.random(40_000) < 0.5, r + delta, r - delta) * (1 + g.integers(-8, 9, 40_000) * 2.2e-16)
    pts = c0 + u * rad[:, None]
    oc = O.cfg(O.SPHERE, 3, delta)
    par = np.concatenate([c0, [r]])
    ctx.set_model(L.SPHERE, 3, delta).upload(pts)
    m, cnt = ctx.mask(par)
    wcnt, wm = O.scan(oc, par, pts)
    assert 0.2 < wcnt / len(pts) < 0.8
    assert np.array_equal(m, wm)


def test_full_size_plane_properties(ctx):
    """BASELINE config 2 size (10M points, 50% outliers): size-independent properties plus a
    bit-exact check of a few hypotheses against the oracle."""
    n = 10_000_000
    data, truth, lab = synth.plane(n, 0.5)
    oc = O.cfg(O.PLANE, 3, 0.5)
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    subs = ctx.hypotheses_sample(7, 0, 512, want_subsets=True)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    _, bv, bi = ctx.best()
    assert bv == votes.max() and votes[bi] == bv and np.all(votes[:bi] < bv)
    m, cnt = ctx.mask_from_hypothesis(bi)
    assert cnt == bv == int(m.sum())                       # mask count == scan votes
    for h in (0, bi, 511):                                 # oracle, full size, bit-exact
        assert votes[h] == O.scan(oc, par[h], data)[0]
    assert np.array_equal(m, O.scan(oc, par[bi], data)[1])
    # a wider band can only gain votes
    ctx.set_model(L.PLANE, 3, 1.0).upload(data)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    _, _, votes2 = ctx.hypotheses()
    assert np.all(votes2 >= votes)
    # end to end: the consensus is (almost) the true inlier set and the fit recovers the plane
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    r = ctx.ransac(0.999, seed=3)
    assert abs(abs(r["params"][:3] @ truth[:3]) - 1) < 1e-7
    assert abs((r["params"][3:] - truth[3:]) @ truth[:3]) < 0.1   # band is centred on the hypothesis
    assert (r["consensus"].astype(bool) & ~lab).sum() < 0.01 * n
    want = O.ls(oc, data, r["consensus"])
    _params_close(L.PLANE, 3, r["params"], want)


# ------------------------------------------------------------------------------------- dense Ax=b
def _dense_close(got, want, tol=REL):
    assert len(got) == len(want) and len(want) > 0
    scale = max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= tol * scale


@pytest.mark.parametrize("ncol,m", [(5, 200), (6, 1443), (17, 900), (64, 3000)])
def test_dense_hypotheses_scan_mask(ctx, ncol, m):
    """Minimal n x n solves go through a different SVD (one-sided Jacobi, wave-parallel) than the
    oracle's, so models agree to rounding (1e-9), not bitwise; the agree() scan is then checked
    BIT-EXACT against the oracle evaluated on the device's own models."""
    rows = synth.dense(m, ncol, 0.1, seed=40 + ncol)[0]
    oc = O.cfg(O.DENSE, ncol, 0.1)
    ctx.set_model(L.DENSE, ncol, 0.1).upload(rows)
    H = 48
    subs = O.ctr_subsets(3, 0, H, m, ncol)
    subs[5] = np.roll(subs[4], 1)
    subs[7][1] = subs[7][0]          # repeated row -> singular system -> invalid
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in range(H):
        want = O.estimate(oc, rows[subs[h]])
        assert bool(valid[h]) == (len(want) > 0), h
        if not valid[h]:
            assert votes[h] == 0
            continue
        _dense_close(par[h], want, 1e-8)
        cnt, wm = O.scan(oc, par[h], rows)
        assert votes[h] == cnt
    assert not valid[7]
    _, bv, bi = ctx.best()
    m_dev, cnt = ctx.mask_from_hypothesis(bi)
    wcnt, wmask = O.scan(oc, par[bi], rows)
    assert cnt == wcnt == bv and np.array_equal(m_dev, wmask)
    got, info = ctx.ls_fit(use_mask=True)
    _dense_close(got, O.ls(oc, rows, wmask))
    st = ctx.stats(got, use_mask=True)
    assert np.allclose(st, O.stats(oc, got, rows, wmask), rtol=1e-9, atol=1e-12)


def test_dense_known_answer_fixture(ctx, golden_dir):
    """testing/Data/augmentedMatrix.txt (1443 x 7) against the solution hard-coded at
    testing/DenseLinearEquationSystemParametersEstimatorTest.cxx:162-164 (tolerance 0.5 there)."""
    M = np.loadtxt(os.path.join(golden_dir, "ref_data", "augmentedMatrix.txt"))
    known = np.array([-1.777985584409468e+001, 1.111302171667757e+000, -1.568653413096010e+002,
                      1.469013927556186e+002, -6.296891425314718e+001, -1.042139650090033e+003])
    ctx.set_model(L.DENSE, 6, 0.5).upload(M)
    got, info = ctx.ls_fit()
    assert np.abs(got - known).max() < 1e-6 * 1042
    # examples/linearEquationSystemSolver.cxx:160-200: the outlier-contaminated file, delta sqrt(1/3)
    W = np.loadtxt(os.path.join(golden_dir, "ref_data", "augmentedMatrixWithOutliers.txt"))
    ctx.set_model(L.DENSE, 6, np.sqrt(1.0 / 3.0)).upload(W)
    r = ctx.ransac(0.999, seed=5)
    approx = np.array([-17, 1, -157, 147, -63, -1042.0])   # examples/linear...cxx:180-181
    assert np.abs(r["params"] - approx).max() < 1.5
    oc = O.cfg(O.DENSE, 6, np.sqrt(1.0 / 3.0))
    _dense_close(r["params"], O.ls(oc, W, r["consensus"]))


def test_dense_rank_deficient_fit_is_empty(ctx):
    """rank(A) < n -> empty parameters (DenseLinear...Estimator.hxx:90-91).  The reference's test is
    absolute (sigma <= 2.2e-16), which only fires for exactly singular systems; the device's normal
    equations use a relative threshold (documented in DESIGN.md), so a numerically rank-deficient
    system is also reported empty instead of returning a 1e14-sized solution."""
    g = np.random.default_rng(0)
    A = g.uniform(-1, 1, (100, 5))
    A[:, 4] = 0.0                     # zero column: sigma_min == 0 exactly in any SVD
    rows = np.hstack([A, g.uniform(-1, 1, (100, 1))])
    ctx.set_model(L.DENSE, 5, 0.1).upload(rows)
    got, _ = ctx.ls_fit()
    assert len(got) == 0
    assert len(O.ls(O.cfg(O.DENSE, 5, 0.1), rows)) == 0
    A[:, 4] = A[:, 0] + A[:, 1]       # numerically rank 4 (sigma_min ~ 1e-16 relative)
    rows = np.hstack([A, g.uniform(-1, 1, (100, 1))])
    ctx.upload(rows)
    got, _ = ctx.ls_fit()
    assert len(got) == 0


@pytest.mark.parametrize("ncol,m", [(5, 2000), (64, 20_000)])
def test_dense_ransac_end_to_end(ctx, ncol, m):
    rows, x_true, lab = synth.dense(m, ncol, 0.05 if ncol == 64 else 0.2, seed=7 + ncol,
                                    noise=0.0005)
    delta = 0.02
    oc = O.cfg(O.DENSE, ncol, delta)
    ctx.set_model(L.DENSE, ncol, delta).upload(rows)
    subs = O.ctr_subsets(21, 0, 600, m, ncol)
    r = ctx.ransac(0.99, subsets=subs)
    assert r["status"] == L.OK
    info = r["info"]
    # winner: mask and votes consistent with the oracle evaluated on the device's winner model
    ctx.hypotheses_from_subsets(subs[info.best_index:info.best_index + 1])
    wpar, _ = ctx.hypothesis(0)
    wcnt, wmask = O.scan(oc, wpar, rows)
    assert wcnt == info.best_votes and np.array_equal(r["consensus"], wmask)
    _dense_close(r["params"], O.ls(oc, rows, wmask))
    assert np.abs(r["params"] - x_true).max() < 1e-2
    # serial oracle on the same subset stream: same iteration count and winner
    w = O.ransac(oc, rows, 0.99, sampler="list", subsets=subs)
    assert info.iterations == w["iters"] and info.best_index == w["best_iter"]
    assert abs(int(info.best_votes) - int(w["best_votes"])) <= 2


def test_dense_sampler_k64(ctx):
    rows = np.zeros((500, 65))
    ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
    subs = ctx.hypotheses_sample(77, 5, 40, want_subsets=True)
    assert np.array_equal(subs, O.ctr_subsets(77, 5, 40, 500, 64))


# --------------------------------------------------------------------- ultrasound calibration
US = [(L.US_SINGLE, synth.us_single, 4), (L.US_POINTER, synth.us_pointer, 3)]


@pytest.mark.parametrize("model,gen,k", US)
def test_us_hypotheses_scan_mask(ctx, model, gen, k):
    """K1 = analytic solve on exactly k frames (12x12 / 9x9 pinv + 3x3 SVD + Euler angles): agrees
    with the oracle to rounding (VNL SVD unpinned, libm vs device trig), then the agree() scan and
    mask are BIT-EXACT against the oracle evaluated on the device's models."""
    rec = gen(3001, 0.3, seed=90 + model, pixel_sigma=1.0)[0]
    oc = O.cfg(model, 0, 3.0, 1)
    ctx.set_model(model, 0, 3.0, L.LS_ITERATIVE).upload(rec)
    H = 64
    subs = O.ctr_subsets(8, 0, H, len(rec), k)
    subs[3][1] = subs[3][0]     # repeated frame -> rank deficient -> invalid
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    nvalid = 0
    for h in range(H):
        want = O.estimate(oc, rec[subs[h]])
        assert bool(valid[h]) == (len(want) > 0), h
        if not valid[h]:
            assert votes[h] == 0
            continue
        nvalid += 1
        assert np.allclose(par[h], want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
        assert votes[h] == O.scan(oc, par[h], rec)[0]
    assert nvalid >= H - 2 and not valid[3]
    _, bv, bi = ctx.best()
    m, cnt = ctx.mask_from_hypothesis(bi)
    wcnt, wmask = O.scan(oc, par[bi], rec)
    assert cnt == wcnt == bv and np.array_equal(m, wmask)
    st = ctx.stats(par[bi], use_mask=True)
    assert np.allclose(st, O.stats(oc, par[bi], rec, wmask), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("model,gen,k", US)
def test_us_least_squares(ctx, model, gen, k):
    rec, truth, lab = gen(4000, 0.25, seed=31 + model, pixel_sigma=1.0)
    mask = lab.astype(np.uint8)
    # ANALYTIC: 3N x 12 (9) linear system; device solves the normal equations
    ctx.set_model(model, 0, 3.0, L.LS_ANALYTIC).upload(rec)
    ctx.set_mask(mask)
    got, info = ctx.ls_fit(use_mask=True)
    want = O.ls(O.cfg(model, 0, 3.0, 0), rec, mask)
    assert len(got) == len(want) > 0
    assert np.allclose(got, want, rtol=REL, atol=REL * np.abs(want).max())
    # ITERATIVE on a small, quickly converging set (with the reference's 1e-15
    # tolerances MINPACK's termination flag is decided inside rounding noise on large sets)
    small = gen(50, 0.0, seed=21, pixel_sigma=1.0)[0]
    ctx.set_model(model, 0, 3.0, L.LS_ITERATIVE).upload(small)
    got, info = ctx.ls_fit()
    oc = O.cfg(model, 0, 3.0, 1)
    want = O.ls(oc, small)
    assert len(want) > 0 and len(got) == len(want) and 1 <= info.lm_info <= 4
    assert np.allclose(got, want, rtol=REL, atol=REL * np.abs(want).max())
    assert ctx.stats(got)[3] <= O.stats(oc, O.us_analytic(model, small), small)[3] * (1 + 1e-9)


def test_us_pointer_iterative_large(ctx):
    """tolerances 1e-7 (calibrated pointer, ...Estimator.cxx:931-939): robust termination."""
    rec, truth, lab = synth.us_pointer(20_000, 0.2, seed=77, pixel_sigma=1.0)
    mask = lab.astype(np.uint8)
    ctx.set_model(L.US_POINTER, 0, 3.0, L.LS_ITERATIVE).upload(rec)
    ctx.set_mask(mask)
    got, info = ctx.ls_fit(use_mask=True)
    inl = np.ascontiguousarray(rec[lab])
    init = O.us_analytic(O.US_POINTER, inl)
    want, winfo, wnfev = O.us_iterative(O.US_POINTER, inl, init)
    # both succeed; the evaluation counts may differ (the 1e-7 stopping rules fire inside the rounding
    # differences of the two analytic initial estimates), the minimiser may not
    assert 1 <= winfo <= 4 and 1 <= info.lm_info <= 4 and 2 <= info.lm_nfev <= wnfev + 3
    assert np.allclose(got, want, rtol=REL, atol=REL * np.abs(want).max())
    assert np.linalg.norm(got[0:3] - truth[0:3]) < 1.0


@pytest.mark.parametrize("model,gen,k", US)
def test_us_ransac_end_to_end(ctx, model, gen, k):
    rec, truth, lab = gen(5000, 0.3, seed=55 + model, pixel_sigma=0.5)
    oc = O.cfg(model, 0, 3.0, 0)
    ctx.set_model(model, 0, 3.0, L.LS_ANALYTIC).upload(rec)
    subs = O.ctr_subsets(13, 0, 2000, len(rec), k)
    r = ctx.ransac(0.99, subsets=subs)
    assert r["status"] == L.OK
    info = r["info"]
    ctx.hypotheses_from_subsets(subs[info.best_index:info.best_index + 1])
    wpar, _ = ctx.hypothesis(0)
    wcnt, wmask = O.scan(oc, wpar, rec)
    assert wcnt == info.best_votes and np.array_equal(r["consensus"], wmask)
    want = O.ls(oc, rec, wmask)
    assert np.allclose(r["params"], want, rtol=REL, atol=REL * np.abs(want).max())
    w = O.ransac(oc, rec, 0.99, sampler="list", subsets=subs)
    assert abs(int(info.best_votes) - int(w["best_votes"])) <= 3
    assert (r["consensus"].astype(bool) & ~lab).sum() <= 0.02 * len(rec)


def test_crosswire_experimental_data_on_device(ctx, golden_dir):
    """testing/Data/crossWirePhantom*.txt (54 frames), fed as
    testing/SinglePointTargetUSCalibrationParametersEstimatorTest.cxx:115-166 does."""
    T = np.loadtxt(os.path.join(golden_dir, "ref_data", "crossWirePhantomTransformations.txt"))
    q = np.loadtxt(os.path.join(golden_dir, "ref_data", "crossWirePhantom2DPoints.txt"))
    m = q.shape[0]
    rec = np.zeros((m, 15))
    T = T.reshape(m, 3, 4)
    rec[:, 0:9] = T[:, :, :3].reshape(m, 9)
    rec[:, 9:12] = T[:, :, 3]
    rec[:, 13:15] = q
    ctx.set_model(L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC).upload(rec)
    got, _ = ctx.ls_fit()
    want = O.us_analytic(O.US_SINGLE, rec)
    assert np.allclose(got, want, rtol=REL, atol=REL * np.abs(want).max())
    ctx.set_model(L.US_SINGLE, 0, 3.0, L.LS_ITERATIVE).upload(rec)
    got, info = ctx.ls_fit()
    want = O.ls(O.cfg(O.US_SINGLE, 0, 3.0, 1), rec)
    if len(want) and len(got):
        assert np.allclose(got, want, rtol=1e-5, atol=1e-5 * np.abs(want).max())


# ------------------------------------------------------------------------------- edge cases / ABI
def test_large_batch_is_chunked(ctx):
    """H larger than one scan launch's LDS counter block (8192): votes must be identical to
    per-chunk evaluation, and H = 1 works."""
    data = synth.plane(30_000, 0.5, seed=17)[0]
    oc = O.cfg(O.PLANE, 3, 0.5)
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    H = 20_000
    subs = ctx.hypotheses_sample(3, 0, H, want_subsets=True)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in (0, 8191, 8192, 16383, 16384, H - 1):
        assert votes[h] == O.scan(oc, par[h], data)[0]
    _, bv, bi = ctx.best()
    assert bv == votes.max() and bi == int(np.argmax(votes))
    ctx.hypotheses_from_subsets(subs[123:124])
    ctx.scan()
    _, _, v1 = ctx.hypotheses()
    assert v1[0] == votes[123]


@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.SPHERE, 3), (L.LINE, 3)])
def test_filter_and_plain_scan_agree(ctx, model, dim):
    """the fp32 pre-filter (either re-check granularity) and every observations-per-lane variant
    give identical votes"""
    data = _data(model, dim, 200_003, 4321, outliers=0.5)
    ctx.set_model(model, dim, 0.5).upload(data)
    ctx.hypotheses_sample(11, 0, 700)
    ref = None
    for ppl, filt in ((0, 1), (4, 0), (8, 0), (2, 0), (8, 1), (16, 1), (4, 2), (8, 2), (4, 3)):
        ctx.set_option("scan_ppl", ppl)
        ctx.set_option("scan_filter", filt)
        ctx.scan()
        _, _, votes = ctx.hypotheses(params=False)
        if ref is None:
            ref = votes.copy()
        assert np.array_equal(votes, ref), (ppl, filt)
    ctx.set_option("scan_ppl", 0)
    ctx.set_option("scan_filter", 1)
    oc = O.cfg(model, dim, 0.5)
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in range(0, 700, 97):
        assert ref[h] == O.scan(oc, par[h], data)[0]


def test_filter_boundary_stress(ctx):
    """observations placed within a few fp32 ulps of the plane band edge (|s| ~ delta) and huge
    coordinates: the filter must route them to the exact path, votes stay bit-exact."""
    g = np.random.default_rng(5)
    n0 = np.array([0.36, 0.48, 0.8])
    a0 = np.array([900.0, -700.0, 650.0])
    base = g.uniform(-1000, 1000, (60_000, 3))
    base -= ((base - a0) @ n0)[:, None] * n0            # on the plane
    off = np.where(g.random(60_000) < 0.5, 0.5, -0.5) * (1 + g.integers(-40, 41, 60_000) * 1e-7)
    pts = np.ascontiguousarray(base + off[:, None] * n0)
    pts[:3] = [a0, a0 + np.array([1.0, 0, -0.45]) * 300, a0 + np.array([0, 1.0, -0.6]) * 300]
    oc = O.cfg(O.PLANE, 3, 0.5)
    ctx.set_model(L.PLANE, 3, 0.5).upload(pts)
    subs = np.vstack([[0, 1, 2], O.ctr_subsets(4, 0, 63, len(pts), 3)]).astype(np.uint32)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    assert abs(abs(par[0][:3] @ n0) - 1) < 1e-9
    for h in range(64):
        if valid[h]:
            assert votes[h] == O.scan(oc, par[h], pts)[0]
    assert 0.2 < votes[0] / len(pts) < 0.8
    # coordinates beyond the filter's validated range: it must switch itself off, not misclassify
    big = pts * 1e16
    ctx.set_model(L.PLANE, 3, 0.5e16).upload(big)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    ocb = O.cfg(O.PLANE, 3, 0.5e16)
    for h in (0, 5, 9):
        if valid[h]:
            assert votes[h] == O.scan(ocb, par[h], big)[0]


@pytest.mark.parametrize("model,gen,k", US)
def test_us_filter_and_plain_scan_agree(ctx, model, gen, k):
    """the packed fp32 pre-filter (scan_filter 1) and the fused fp64 pre-filter (2) of the US scan give
    the votes of the exact predicate (and the oracle's); the int outputFormat slot of the Frame
    (garbage as a double) must not disturb them."""
    rec = gen(150_001, 0.4, seed=17 + model)[0]
    rec[:, 12] = np.frombuffer(np.full(len(rec), 0x7ff8dead00000001, np.uint64).tobytes(), np.float64)
    oc = O.cfg(model, 0, 3.0, 1)
    ctx.set_model(model, 0, 3.0, L.LS_ANALYTIC).upload(rec)
    ctx.hypotheses_sample(3, 0, 300)
    ref = None
    for filt in (1, 2, 0):
        ctx.set_option("scan_filter", filt)
        ctx.scan()
        _, _, votes = ctx.hypotheses(params=False)
        if ref is None:
            ref = votes.copy()
        assert np.array_equal(votes, ref)
    ctx.set_option("scan_filter", 1)
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in range(0, 300, 37):
        if valid[h]:
            assert ref[h] == O.scan(oc, par[h], rec)[0]
    assert ref.max() > 0.5 * len(rec)


def test_us_filter_boundary_stress(ctx):
    """frames whose mapped point sits within a few ulps of the delta-sphere around the target: the
    fused filter must hand them to the exact predicate; votes stay bit-exact.  Then magnitudes for
    which the error band is not small against delta^2: the filter must switch itself off."""
    g = np.random.default_rng(12)
    rec = synth.us_single(20_000, 0.0, seed=5, pixel_sigma=0.0)[0]
    oc = O.cfg(O.US_SINGLE, 0, 3.0, 1)
    ctx.set_model(L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC).upload(rec)
    subs = O.ctr_subsets(2, 0, 16, 8, 4).astype(np.uint32)   # hypotheses from frames 0..7 only
    ctx.hypotheses_from_subsets(subs)
    par, valid, _ = ctx.hypotheses(votes=False)
    h0 = int(np.flatnonzero(valid)[0])
    P = par[h0]
    R2 = rec[:, 0:9].reshape(-1, 3, 3)
    T3 = np.stack([P[11:14], P[14:17]], axis=1)          # 3x2: the scaled columns of R3
    p3 = rec[:, 13:15] @ T3.T + P[3:6]
    q = np.einsum("mij,mj->mi", R2, p3) + rec[:, 9:12]
    d = g.normal(size=(len(rec), 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    rad = 3.0 * (1 + g.integers(-30, 31, len(rec)) * 2.0 ** -52)
    rec2 = rec.copy()
    rec2[8:, 9:12] += (P[0:3] + rad[:, None] * d - q)[8:]
    ctx.upload(rec2)
    for filt in (1, 2):                                   # packed fp32 filter, fused fp64 filter
        ctx.set_option("scan_filter", filt)
        ctx.hypotheses_from_subsets(subs)
        ctx.scan()
        par2, valid2, votes = ctx.hypotheses()
        assert np.array_equal(par2[h0], P)
        for h in range(16):
            if valid2[h]:
                assert votes[h] == O.scan(oc, par2[h], rec2)[0], (filt, h)
        assert 0.2 < votes[h0] / len(rec2) < 0.8
    ctx.set_option("scan_filter", 1)
    # frames a few fp32 ulps either side of the sphere as well
    rad32 = 3.0 * (1 + g.integers(-30, 31, len(rec)) * 1e-7)
    rec3 = rec.copy()
    rec3[8:, 9:12] += (P[0:3] + rad32[:, None] * d - q)[8:]
    ctx.upload(rec3)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par3_, valid3_, votes_ = ctx.hypotheses()
    for h in range(16):
        if valid3_[h]:
            assert votes_[h] == O.scan(oc, par3_[h], rec3)[0], h
    big = rec2.copy()
    big[:, 9:12] *= 1e9                                  # |t2| ~ 1e11 against delta = 3
    big[:8] = rec2[:8]
    oc2 = O.cfg(O.US_SINGLE, 0, 1e-4, 1)
    ctx.set_model(L.US_SINGLE, 0, 1e-4, L.LS_ANALYTIC).upload(big)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par3, valid3, votes3 = ctx.hypotheses()
    for h in range(16):
        if valid3[h]:
            assert votes3[h] == O.scan(oc2, par3[h], big)[0], h


ATTACH_SCRIPT = r"""
import sys, numpy as np, torch
torch.cuda.init()                      # torch's HIP runtime must come up before the library's
sys.path.insert(0, %r)
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O
data = synth.sphere(50_000, 0.4, seed=9)[0]
t = torch.from_numpy(data).to("cuda:0")
oc = O.cfg(O.SPHERE, 3, 0.5, O.LS_ALGEBRAIC)
with Context(0) as c1, Context(0) as c2:
    c1.set_model(L.SPHERE, 3, 0.5, L.LS_ALGEBRAIC).attach(t.data_ptr(), len(data), 24, keepalive=t)
    c2.set_model(L.PLANE, 3, 0.5).upload(synth.plane(10_000, 0.3, seed=1)[0])
    r1 = c1.ransac(0.99, seed=4)
    r2 = c2.ransac(0.99, seed=4)
    w1 = O.ransac(oc, data, 0.99, sampler="ctr", seed=4)
    assert np.array_equal(r1["consensus"], w1["consensus"]) and r1["info"].iterations == w1["iters"]
    assert len(r2["params"]) == 6
print("attach ok")
"""


def test_attach_device_memory_and_two_contexts():
    """lsqr_attach adopts a device pointer (a torch tensor's storage); contexts are independent.
    Runs in its own process: torch bundles its own HIP runtime, which has to initialise first."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", ATTACH_SCRIPT % root], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "attach ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_call_order_errors(ctx):
    with Context(0) as c:
        with pytest.raises(L.LsqrError) as e:
            c.upload(np.zeros((4, 3)))           # no model yet
        assert e.value.status == L.ERR_STATE
        c.set_model(L.PLANE, 3, 0.5)
        with pytest.raises(L.LsqrError):
            c.n = 0
            c.scan()                              # no data
        c.upload(synth.plane(100, 0.1)[0])
        with pytest.raises(L.LsqrError):
            c.scan()                              # no hypotheses
        with pytest.raises(L.LsqrError):
            c.ls_fit(use_mask=True)               # no mask
        with pytest.raises(L.LsqrError):
            c._chk(c._lib.lsqr_upload(c._h, None, 10, 20))   # stride not a multiple of 8 / too small
        with pytest.raises(L.LsqrError):
            c.set_option("no_such_option", 1)


def test_profile_counters(ctx):
    data = synth.plane(100_000, 0.5, seed=2)[0]
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    ctx.profile(True)
    ctx.hypotheses_sample(1, 0, 256)
    ctx.scan()
    ctx.scan()
    n, ms = ctx.profile_get("scan")
    assert n == 2 and ms > 0
    n, ms = ctx.profile_get("estimate")
    assert n == 1
    ctx.profile(False)


@pytest.mark.parametrize("dim", [2, 3])
def test_line_filter_boundary_stress(ctx, dim):
    """observations within a few fp64 ulps .. fp32 ulps of the delta-cylinder around a line: the
    fp32 cross-product filter must route them to the exact predicate (bit-exact votes); then
    magnitudes where its band is useless: it must switch itself off."""
    g = np.random.default_rng(31 + dim)
    d0 = np.array([0.48, 0.6, 0.64])[:dim]
    d0 /= np.linalg.norm(d0)
    a0 = np.array([700.0, -300.0, 500.0])[:dim]
    m = 80_000
    t = g.uniform(-1000, 1000, m)
    perp = g.normal(size=(m, dim))
    perp -= (perp @ d0)[:, None] * d0
    perp /= np.linalg.norm(perp, axis=1)[:, None]
    scale = np.where(np.arange(m) % 2 == 0, 2.0 ** -52, 1e-7)
    rad = 0.5 * (1 + g.integers(-40, 41, m) * scale)
    pts = np.ascontiguousarray(a0 + t[:, None] * d0 + rad[:, None] * perp)
    pts[0] = a0 - 900 * d0
    pts[1] = a0 + 900 * d0
    oc = O.cfg(O.LINE, dim, 0.5)
    ctx.set_model(L.LINE, dim, 0.5).upload(pts)
    subs = np.vstack([[0, 1], O.ctr_subsets(4, 0, 63, m, 2)]).astype(np.uint32)
    for filt in (1, 3, 0):
        ctx.set_option("scan_filter", filt)
        ctx.hypotheses_from_subsets(subs)
        ctx.scan()
        par, valid, votes = ctx.hypotheses()
        assert abs(abs(par[0][:dim] @ d0) - 1) < 1e-9
        for h in range(64):
            if valid[h]:
                assert votes[h] == O.scan(oc, par[h], pts)[0], (filt, h)
        assert 0.2 < votes[0] / m < 0.8
    ctx.set_option("scan_filter", 1)
    big = pts * 1e9                                      # fp32 band >> delta^2: filter off
    ctx.set_model(L.LINE, dim, 0.5).upload(big)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in (0, 5, 9):
        if valid[h]:
            assert votes[h] == O.scan(oc, par[h], big)[0]


def test_sphere_filter_boundary_stress(ctx):
    """observations within a few fp32 ulps of both edges of the sphere band, scanned through the
    fp32 pre-filter: votes must equal the oracle's exact count for every hypothesis."""
    g = np.random.default_rng(8)
    c0 = np.array([310.5, -420.25, 97.125])
    r, delta = 512.0, 0.5
    u = g.normal(size=(80_000, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    edge = np.where(g.random(80_000) < 0.5, r + delta, r - delta)
    rad = edge * (1 + g.integers(-60, 61, 80_000) * 1e-8)
    pts = np.ascontiguousarray(c0 + u * rad[:, None])
    # four exact points of the sphere first, so that hypothesis 0 is (almost) the true sphere
    pts[:4] = c0 + r * np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0]], float)
    oc = O.cfg(O.SPHERE, 3, delta)
    ctx.set_model(L.SPHERE, 3, delta).upload(pts)
    subs = np.vstack([[0, 1, 2, 3], O.ctr_subsets(6, 0, 63, len(pts), 4)]).astype(np.uint32)
    for filt in (1, 0):
        ctx.set_option("scan_filter", filt)
        ctx.hypotheses_from_subsets(subs)
        ctx.scan()
        par, valid, votes = ctx.hypotheses()
        assert np.allclose(par[0], np.concatenate([c0, [r]]), atol=1e-9)
        for h in range(64):
            if valid[h]:
                assert votes[h] == O.scan(oc, par[h], pts)[0], (filt, h)
        assert 0.2 < votes[0] / len(pts) < 0.8
    ctx.set_option("scan_filter", 1)


def test_lm_host_and_device_stepping_agree(ctx):
    """the LM control flow (lm_core.h) runs on the host by default and in a device kernel with
    lm_host=0: same code, same iterates."""
    pts = synth.sphere(40_000, 0.0, seed=11)[0]
    res = []
    for host in (1, 0):
        ctx.set_option("lm_host", host)
        ctx.set_model(L.SPHERE, 3, 0.5, L.LS_GEOMETRIC).upload(pts)
        fit, info = ctx.ls_fit()
        res.append((fit, info.lm_info, info.lm_nfev))
        small = synth.us_single(50, 0.0, seed=21, pixel_sigma=1.0)[0]
        ctx.set_model(L.US_SINGLE, 0, 3.0, L.LS_ITERATIVE).upload(small)
        fit, info = ctx.ls_fit()
        res.append((fit, info.lm_info, info.lm_nfev))
    ctx.set_option("lm_host", 1)
    assert res[0][1] == res[2][1] and res[0][2] == res[2][2]
    assert np.allclose(res[0][0], res[2][0], rtol=1e-12, atol=1e-12)
    assert len(res[1][0]) == len(res[3][0]) == 20
    assert np.allclose(res[1][0], res[3][0], rtol=1e-6, atol=1e-6)


def test_dense_mfma_filter_is_exact(ctx):
    """the dense scan evaluates residuals with fp64 MFMAs (fused multiply-adds) and only decides
    pairs whose residual is clear of delta by the rigorous band; pairs inside the band go through the
    exact formula.  delta is placed exactly on one row's reference residual (and one ulp above) so
    that this row is forced through the worklist; every variant must give the oracle's votes."""
    ncol, m = 64, 5000
    rows = synth.dense(m, ncol, 0.1, seed=123, noise=0.01)[0]
    subs = O.ctr_subsets(9, 0, 80, m, ncol)
    ctx.set_model(L.DENSE, ncol, 0.1).upload(rows)
    ctx.hypotheses_from_subsets(subs)
    par, valid, _ = ctx.hypotheses(votes=False)
    h = int(np.argmax(valid))
    x = par[h]
    # reference residual of row 777 for hypothesis h, computed in the reference's operation order
    s = 0.0
    for i in range(ncol):
        s += rows[777, i] * x[i]
    rho = abs(s - rows[777, ncol])
    assert rho > 0
    for delta in (rho, np.nextafter(rho, np.inf), np.nextafter(rho, 0.0), 0.1):
        oc = O.cfg(O.DENSE, ncol, delta)
        want = None
        # fp16-split (the default), fp32 (LDS ring) and fp64 matrix-core filters, exact VALU kernel
        for variant in ("h16", "mfma32r", "mfma64", "plain"):
            ctx.set_option("scan_filter", 0 if variant == "plain" else 1)
            ctx.set_option("dense_f32", {"h16": 2, "mfma64": 0}.get(variant, 1))
            ctx.set_model(L.DENSE, ncol, delta).upload(rows)
            ctx.hypotheses_from_subsets(subs)
            ctx.scan()
            p2, v2, votes = ctx.hypotheses()
            assert np.array_equal(p2[valid > 0], par[valid > 0])
            if want is None:
                want = np.array([O.scan(oc, p2[i], rows)[0] if v2[i] else 0 for i in range(80)])
            assert np.array_equal(votes, want), (delta, variant)
        # the boundary row itself: strict '<'
        assert O.agree(oc, x, rows[777]) == (rho < delta)
    ctx.set_option("scan_filter", 1)
    ctx.set_option("dense_f32", 2)


# ---- two-level scan over the spatial index (csrc/cells.h) -------------------------------------------
def _scan_votes(ctx, index, cell=0, cpt=0):
    ctx.set_option("scan_index", index)
    ctx.set_option("scan_cell", cell)
    ctx.set_option("scan_cpt", cpt)
    ctx.scan()
    _, _, v = ctx.hypotheses(params=False)
    ctx.set_option("scan_index", 1)
    ctx.set_option("scan_cell", 0)
    ctx.set_option("scan_cpt", 0)
    return v.copy()


@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.PLANE, 2), (L.SPHERE, 3), (L.SPHERE, 2),
                                       (L.LINE, 3), (L.LINE, 2)])
@pytest.mark.parametrize("n", [4, 127, 128, 129, 5000, 200_003])
def test_cell_scan_matches_exhaustive_and_oracle(ctx, model, dim, n):
    """the culled two-level scan (forced, any size) counts exactly what the exhaustive kernels and
    the oracle count; ragged last cell, fewer observations than one cell, H not a multiple of 64"""
    data = _data(model, dim, n, 777 + n, outliers=0.5)
    H = 200
    ctx.set_model(model, dim, 0.5).upload(data)
    ctx.hypotheses_sample(21, 0, H)
    plain = _scan_votes(ctx, 0)
    for cell, cpt in ((0, 0), (256, 1), (512, 1)):
        assert np.array_equal(_scan_votes(ctx, 2, cell, cpt), plain), (cell, cpt)
    for block in (256, 257):      # v_readlane / LDS broadcast of the hypothesis
        ctx.set_option("scan_block", block)
        v = _scan_votes(ctx, 2)
        ctx.set_option("scan_block", 0)
        assert np.array_equal(v, plain), block
    oc = O.cfg(model, dim, 0.5)
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in range(0, H, 23):
        if valid[h]:
            assert plain[h] == O.scan(oc, par[h], data)[0]


def test_cell_scan_nonfinite_duplicate_and_flat_data(ctx):
    """NaN / inf observations never agree and are left out of the index; identical observations
    and axis-aligned (zero-extent) cells must not break the box test"""
    g = np.random.default_rng(9)
    data = synth.plane(70_001, 0.5, seed=99)[0]
    data[::1000, 0] = np.nan
    data[1::1000, 1] = np.inf
    data[2::1000, 2] = -np.inf
    data[5000:9000] = data[4999]                  # 4000 identical observations
    data[20_000:30_000, 2] = 12.25                # a flat slab: zero half extent in z
    H = 130
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    subs = O.ctr_subsets(8, 0, H, len(data), 3)
    subs[0] = [20_003, 20_004, 20_005]            # the plane z = 12.25 exactly
    subs[1] = [4999, 5000, 5001]                  # degenerate (identical points)
    ctx.hypotheses_from_subsets(subs)
    plain = _scan_votes(ctx, 0)
    assert np.array_equal(_scan_votes(ctx, 2), plain)
    par, valid, _ = ctx.hypotheses(votes=False)
    assert not valid[1] and plain[1] == 0
    assert plain[0] >= 10_000
    oc = O.cfg(O.PLANE, 3, 0.5)
    for h in (0, 2, 50, 129):
        if valid[h]:
            assert plain[h] == O.scan(oc, par[h], data)[0]
    # nothing finite at all
    bad = np.full((300, 3), np.nan)
    bad[:3] = [[0, 0, 0], [1, 0, 0], [0, 1, np.inf]]
    ctx.upload(bad)
    ctx.hypotheses_from_subsets(np.array([[0, 1, 2], [3, 4, 5]], dtype=np.uint32))
    assert np.array_equal(_scan_votes(ctx, 2), _scan_votes(ctx, 0))


def test_cell_scan_boundary_stress(ctx):
    """observations within a few fp32 ulps of the band edge, the model plane cutting through every
    cell, and a tiny delta that switches the fp32 filter off: votes stay bit-exact"""
    g = np.random.default_rng(6)
    n0 = np.array([0.36, 0.48, 0.8])
    a0 = np.array([900.0, -700.0, 650.0])
    base = g.uniform(-1000, 1000, (80_000, 3))
    base -= ((base - a0) @ n0)[:, None] * n0
    off = np.where(g.random(80_000) < 0.5, 0.5, -0.5) * (1 + g.integers(-40, 41, 80_000) * 1e-7)
    pts = np.ascontiguousarray(base + off[:, None] * n0)
    pts[:3] = [a0, a0 + np.array([1.0, 0, -0.45]) * 300, a0 + np.array([0, 1.0, -0.6]) * 300]
    subs = np.vstack([[0, 1, 2], O.ctr_subsets(4, 0, 99, len(pts), 3)]).astype(np.uint32)
    for delta in (0.5, 1e-9):
        oc = O.cfg(O.PLANE, 3, delta)
        ctx.set_model(L.PLANE, 3, delta).upload(pts)
        ctx.hypotheses_from_subsets(subs)
        plain = _scan_votes(ctx, 0)
        assert np.array_equal(_scan_votes(ctx, 2), plain)
        par, valid, _ = ctx.hypotheses(votes=False)
        for h in (0, 1, 37, 99):
            if valid[h]:
                assert plain[h] == O.scan(oc, par[h], pts)[0]
    # magnitudes beyond fp32: the index is bypassed (plain fp64 kernel), not misused
    big = pts * 1e30
    ctx.set_model(L.PLANE, 3, 0.5e30).upload(big)
    ctx.hypotheses_from_subsets(subs)
    v = _scan_votes(ctx, 2)
    ocb = O.cfg(O.PLANE, 3, 0.5e30)
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in (0, 5):
        if valid[h]:
            assert v[h] == O.scan(ocb, par[h], big)[0]


def test_cell_scan_auto_mode_and_ransac(ctx):
    """default options: the index is built once the upload has seen enough hypotheses, and
    RANSAC::compute() gives the same result with and without it"""
    data = synth.plane(300_000, 0.5, seed=31)[0]
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    ctx.profile(True)
    ctx.hypotheses_sample(3, 0, 4096)
    ctx.scan()
    _, _, v_auto = ctx.hypotheses(params=False)
    n_idx, _ = ctx.profile_get("index")
    ctx.profile(False)
    assert n_idx == 1, "index build expected on the first large batch"
    assert np.array_equal(_scan_votes(ctx, 0), v_auto)
    ctx.set_option("max_iterations", 20000)
    r1 = ctx.ransac(0.999, seed=77)
    ctx.set_option("scan_index", 0)
    r0 = ctx.ransac(0.999, seed=77)
    ctx.set_option("scan_index", 1)
    ctx.set_option("max_iterations", 0)
    assert r1["info"].iterations == r0["info"].iterations
    assert r1["info"].best_index == r0["info"].best_index
    assert np.array_equal(r1["consensus"], r0["consensus"])
    assert np.array_equal(r1["params"], r0["params"])


def test_cell_scan_sphere_boundary_stress(ctx):
    """sphere: observations within a few fp32 ulps of both edges of the band, a sphere much larger
    and one much smaller than a cell, a tiny-delta (literal formula) run: the two-level scan counts
    exactly what the oracle counts"""
    g = np.random.default_rng(18)
    c0 = np.array([310.5, -420.25, 97.125])
    r, delta = 512.0, 0.5
    m = 90_000
    u = g.normal(size=(m, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    edge = np.where(g.random(m) < 0.5, r + delta, r - delta)
    rad = edge * (1 + g.integers(-60, 61, m) * 1e-8)
    pts = np.ascontiguousarray(c0 + u * rad[:, None])
    pts[60_000:] = g.uniform(-1000, 1000, (m - 60_000, 3))       # clutter
    pts[:4] = c0 + r * np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0]], float)
    pts[4:8] = pts[70_000] + 0.7 * np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0]], float)
    subs = np.vstack([[0, 1, 2, 3], [4, 5, 6, 7], O.ctr_subsets(6, 0, 98, m, 4)]).astype(np.uint32)
    for d in (delta, 1e-13):
        oc = O.cfg(O.SPHERE, 3, d)
        ctx.set_model(L.SPHERE, 3, d).upload(pts)
        ctx.hypotheses_from_subsets(subs)
        plain = _scan_votes(ctx, 0)
        for cell in (256, 512):
            assert np.array_equal(_scan_votes(ctx, 2, cell, 1), plain), (d, cell)
        par, valid, _ = ctx.hypotheses(votes=False)
        for h in (0, 1, 2, 50, 99):
            if valid[h]:
                assert plain[h] == O.scan(oc, par[h], pts)[0], (d, h)
        if d == delta:
            assert 0.2 < plain[0] / m < 0.8


@pytest.mark.parametrize("dim", [3, 2])
def test_cell_scan_line_boundary_stress(ctx, dim):
    """line: observations within a few fp64 / fp32 ulps of the delta-cylinder, clutter around it:
    the two-level scan counts exactly what the exhaustive kernels and the oracle count"""
    g = np.random.default_rng(131 + dim)
    d0 = np.array([0.48, 0.6, 0.64])[:dim]
    d0 /= np.linalg.norm(d0)
    a0 = np.array([700.0, -300.0, 500.0])[:dim]
    m = 90_000
    t = g.uniform(-1000, 1000, m)
    perp = g.normal(size=(m, dim))
    perp -= (perp @ d0)[:, None] * d0
    perp /= np.linalg.norm(perp, axis=1)[:, None]
    scale = np.where(np.arange(m) % 2 == 0, 2.0 ** -52, 1e-7)
    rad = 0.5 * (1 + g.integers(-40, 41, m) * scale)
    pts = np.ascontiguousarray(a0 + t[:, None] * d0 + rad[:, None] * perp)
    pts[60_000:] = g.uniform(-1000, 1000, (m - 60_000, dim))
    pts[0] = a0 - 900 * d0
    pts[1] = a0 + 900 * d0
    oc = O.cfg(O.LINE, dim, 0.5)
    ctx.set_model(L.LINE, dim, 0.5).upload(pts)
    subs = np.vstack([[0, 1], O.ctr_subsets(4, 0, 99, m, 2)]).astype(np.uint32)
    ctx.hypotheses_from_subsets(subs)
    plain = _scan_votes(ctx, 0)
    for cell in (256, 512):
        assert np.array_equal(_scan_votes(ctx, 2, cell, 1), plain), cell
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in (0, 1, 2, 50, 99):
        if valid[h]:
            assert plain[h] == O.scan(oc, par[h], pts)[0], h
    assert 0.1 < plain[0] / m < 0.6


@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.SPHERE, 3), (L.LINE, 2)])
def test_batch_fit_equals_stepwise_path(ctx, model, dim):
    """lsqr_batch_fit (one chain on the stream) = sample + scan + best + mask + fit step by step"""
    data = _data(model, dim, 120_000, 4242, outliers=0.5)
    ctx.set_model(model, dim, 0.5).upload(data)
    H = 3000
    r = ctx.batch_fit(99, 7000, H, want_consensus=True)
    ctx.hypotheses_sample(99, 7000, H)
    ctx.scan()
    packed, bv, bi = ctx.best()
    m, cnt = ctx.mask_from_hypothesis(bi)
    fit, info = ctx.ls_fit(use_mask=True)
    assert r["info"].best_votes == bv and r["info"].best_index == 7000 + bi
    assert r["info"].fit.n_used == cnt and abs(r["fraction"] - cnt / len(data)) < 1e-15
    assert np.array_equal(r["consensus"], m)
    assert np.array_equal(r["params"], fit)
    # nothing valid: identical observations -> every subset degenerate -> empty
    ctx.upload(np.ones((500, dim)))
    r = ctx.batch_fit(1, 0, 64)
    assert r["status"] == L.EMPTY and r["info"].best_votes == 0 and len(r["params"]) == 0


# ---- AbsoluteOrientation and PivotCalibration on the device -----------------
def _quat_close(got, want, tol=REL):
    s = np.sign(got[:4] @ want[:4])            # q and -q are the same rotation
    assert np.allclose(s * got[:4], want[:4], rtol=tol, atol=tol)
    assert np.allclose(got[4:], want[4:], rtol=tol, atol=tol * max(1.0, np.abs(want[4:]).max()))


@pytest.mark.parametrize("n", [3, 10, 1000, 20_003])
def test_absolute_orientation_device_path(ctx, n):
    """estimate() bit-exact, agree() scan and mask bit-exact, Horn fit within 1e-6 of the oracle"""
    data, truth, lab = synth.absolute_orientation(n, 0.3 if n > 3 else 0.0, seed=500 + n)
    oc = O.cfg(O.ABSOR, 3, 1.0)
    ctx.set_model(L.ABSOR, 3, 1.0).upload(data)
    H = 64
    subs = O.ctr_subsets(9, 0, H, n, 3)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in range(H):
        want = O.estimate(oc, data[subs[h]])
        assert bool(valid[h]) == (len(want) > 0)
        if valid[h]:
            assert np.array_equal(par[h], want), h
            assert votes[h] == O.scan(oc, want, data)[0], h
    packed, bv, bi = ctx.best()
    m, cnt = ctx.mask_from_hypothesis(bi)
    wcnt, wmask = O.scan(oc, par[bi], data)
    assert cnt == wcnt == bv and np.array_equal(m, wmask)
    fit, _ = ctx.ls_fit(use_mask=True)
    _quat_close(fit, O.ls(oc, data, wmask))
    fit_all, _ = ctx.ls_fit(use_mask=False)
    _quat_close(fit_all, O.ls(oc, data))
    st = ctx.stats(fit, use_mask=True)
    assert np.allclose(st, O.stats(oc, fit, data, wmask), rtol=1e-9, atol=1e-9)


def test_absolute_orientation_exhaustive_ransac_matches_oracle(ctx):
    """examples/AbsoluteOrientation.cxx: exhaustive overload over all C(N,3) subsets"""
    data, truth, lab = synth.absolute_orientation(12, 0.25, seed=77, sigma=0.3)
    oc = O.cfg(O.ABSOR, 3, 1.0)
    ctx.set_model(L.ABSOR, 3, 1.0).upload(data)
    r = ctx.ransac_exhaustive()
    w = O.ransac_exhaustive(oc, data)
    assert r["fraction"] == w["fraction"]
    assert np.array_equal(r["consensus"], w["consensus"])
    _quat_close(r["params"], w["params"])
    assert np.array_equal(r["consensus"].astype(bool), lab)
    # collinear fiducials: every subset degenerate -> empty parameters, fraction 0
    col = data.copy()
    col[:, :3] = np.outer(np.arange(12), [1.0, 2.0, 3.0])
    ctx.upload(col)
    r = ctx.ransac_exhaustive()
    assert r["fraction"] == 0 and len(r["params"]) == 0


def test_pivot_calibration_device_path(ctx, golden_dir):
    """the reference's own pivot calibration data: minimal solve, agree() scan, LS known answer"""
    F = synth.frames_from_pose_rows(np.loadtxt(os.path.join(golden_dir, "ref_data",
                                                            "pivotCalibrationData.txt")))
    n = len(F)
    oc = O.cfg(O.PIVOT, 3, 1.0)
    ctx.set_model(L.PIVOT, 3, 1.0).upload(F)
    subs = np.vstack([[0, int(n / 2.0), n - 1], O.ctr_subsets(2, 0, 63, n, 3)]).astype(np.uint32)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    assert valid[0]
    assert np.allclose(par[0], [-18.586, 1.98134, -157.439, 146.965, -62.0497, -1042.87], atol=6e-3)
    for h in range(64):
        want = O.estimate(oc, F[subs[h]])
        assert bool(valid[h]) == (len(want) > 0)
        if valid[h]:
            assert np.allclose(par[h], want, rtol=REL, atol=1e-6)
            assert votes[h] == O.scan(oc, par[h], F)[0]     # scan exact on the device's own model
    fit, _ = ctx.ls_fit(use_mask=False)
    assert np.allclose(fit, [-17.7799, 1.1113, -156.865, 146.901, -62.9689, -1042.14], atol=6e-3)
    assert np.allclose(fit, O.ls(oc, F), rtol=REL, atol=1e-6)
    m, cnt = ctx.mask(fit)
    wcnt, wmask = O.scan(oc, fit, F)
    assert cnt == wcnt and np.array_equal(m, wmask)
    # identical poses: rank deficient -> empty
    ctx.upload(np.repeat(F[:1], 5, axis=0))
    ctx.hypotheses_from_subsets(np.array([[0, 1, 2]], dtype=np.uint32))
    _, valid, _ = ctx.hypotheses(votes=False)
    assert not valid[0]
    fit, _ = ctx.ls_fit(use_mask=False)
    assert len(fit) == 0


def test_pivot_ransac_end_to_end(ctx, golden_dir):
    """examples/pivotCalibration.cxx: the reference's outlier file (1/3 outliers) and a larger
    synthetic set; same loop as the serial oracle on the same subset stream"""
    F = synth.frames_from_pose_rows(np.loadtxt(os.path.join(golden_dir, "ref_data",
                                                            "pivotCalibrationDataWithOutliers.txt")))
    oc = O.cfg(O.PIVOT, 3, 1.0)
    ctx.set_model(L.PIVOT, 3, 1.0).upload(F)
    r = ctx.ransac(0.999, seed=3)
    w = O.ransac(oc, F, 0.999, sampler="ctr", seed=3)
    assert abs(int(r["info"].best_votes) - int(w["best_votes"])) <= 2
    assert np.allclose(r["params"], [-17.78, 1.11, -156.87, 146.90, -62.97, -1042.14], atol=1.0)
    assert np.allclose(r["params"], w["params"], rtol=1e-4, atol=1e-3)
    big, truth, lab = synth.pivot(50_000, 0.4, seed=12)
    ctx.upload(big)
    r = ctx.ransac(0.999, seed=5)
    assert np.allclose(r["params"], truth, atol=0.05)
    assert abs(r["fraction"] - lab.mean()) < 0.02


@pytest.mark.parametrize("model,dim,ls", [(L.PLANE, 3, 0), (L.SPHERE, 3, L.LS_GEOMETRIC), (L.LINE, 3, 0)])
def test_sharded_step_fused_equals_stepwise(ctx, model, dim, ls):
    """ShardedRansac.step() through lsqr_winner_moments (one sync per half) = batch() + fit()"""
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac
    data = _data(model, dim, 150_000, 999, outliers=0.5)
    ctx.set_model(model, dim, 0.5, ls).upload(data)
    sr = ShardedRansac(ctx, Comm(None))
    votes, gidx, par = sr.batch(7, 3, 2500)
    fit, cnt, info = sr.fit(par)
    r = sr.step(7, 3, 2500)
    assert (r[0], r[1]) == (votes, gidx) and np.array_equal(r[2], par)
    assert r[4] == cnt and np.array_equal(r[3], fit)
    # and the single-device chain gives the same winner and fit
    b = ctx.batch_fit(7, 3 * 2500, 2500)
    assert b["info"].best_votes == votes and b["info"].best_index == gidx
    assert b["info"].fit.n_used == cnt
    assert np.allclose(b["params"], fit, rtol=1e-9, atol=1e-9)


def test_cell_scan_large_batches_lattice_data_and_state_changes(ctx):
    """batches beyond one launch chunk (8192 hypotheses), integer-lattice observations (ties, cells
    with zero extent), and the index following uploads / model changes"""
    g = np.random.default_rng(44)
    lat = g.integers(-40, 41, (120_000, 3)).astype(np.float64)      # ~1.5 observations per lattice site
    lat[:30_000, 2] = np.round(0.5 * lat[:30_000, 0] - 0.25 * lat[:30_000, 1])   # a coarse plane
    for model, delta in ((L.PLANE, 0.5), (L.SPHERE, 0.75), (L.LINE, 1.0)):
        oc = O.cfg(model, 3, delta)
        ctx.set_model(model, 3, delta).upload(lat)
        assert not ctx.index_info()["built"]
        ctx.hypotheses_sample(5, 0, 9000)
        auto = _scan_votes(ctx, 1)                 # auto mode: 9000 >= 2048 and N >= 65536 -> index
        assert ctx.index_info()["built"] and ctx.index_info()["observations"] == len(lat)
        assert np.array_equal(_scan_votes(ctx, 0), auto)
        par, valid, _ = ctx.hypotheses(votes=False)
        for h in (0, 4095, 8191, 8192, 8999):
            if valid[h]:
                assert auto[h] == O.scan(oc, par[h], lat)[0], (model, h)
    # a different upload drops the index; a small one never builds it in auto mode
    small = _data(L.PLANE, 3, 5000, 3)
    ctx.set_model(L.PLANE, 3, 0.5).upload(small)
    assert not ctx.index_info()["built"]
    ctx.hypotheses_sample(5, 0, 4096)
    v = _scan_votes(ctx, 1)
    assert not ctx.index_info()["built"]
    assert np.array_equal(_scan_votes(ctx, 2), v) and ctx.index_info()["built"]
    # scan_index = 0 after a build keeps using the exhaustive kernel but does not drop the index
    assert np.array_equal(_scan_votes(ctx, 0), v) and ctx.index_info()["built"]
    ctx.set_model(L.LINE, 3, 0.5)                   # model change: parameters of the boxes differ
    assert not ctx.index_info()["built"]


@pytest.mark.parametrize("ncol", [5, 33, 64])
def test_dense_fast_and_svd_minimal_solves_agree(ctx, ncol):
    """the elimination fast path of the n x n minimal solves and the SVD pseudo-inverse give the
    same models (to rounding) and the same validity; singular, nearly singular and badly scaled
    subsets fall back to the SVD path and get the reference's rank decision"""
    m = 4000
    rows = synth.dense(m, ncol, 0.1, seed=140 + ncol)[0]
    rows[10] = rows[11]                                   # exactly repeated row
    rows[20, :ncol] = rows[21, :ncol] + 1e-9 * np.random.default_rng(1).normal(size=ncol)  # sigma_min ~ 1e-9: SVD path
    rows[30:30 + ncol] *= 1e-12                           # tiny magnitudes: max|A| < 1
    oc = O.cfg(O.DENSE, ncol, 0.1)
    ctx.set_model(L.DENSE, ncol, 0.1).upload(rows)
    H = 40
    subs = O.ctr_subsets(13, 0, H, m, ncol)
    subs[1][:2] = [10, 11]
    subs[2][:2] = [20, 21]
    subs[3] = np.arange(30, 30 + ncol)
    res = []
    for fast in (1, 0):
        ctx.set_option("dense_fast_solve", fast)
        ctx.hypotheses_from_subsets(subs)
        ctx.scan()
        res.append(ctx.hypotheses())
    ctx.set_option("dense_fast_solve", 1)
    (pf, vf, cf), (ps, vs, cs) = res
    assert np.array_equal(vf, vs)
    assert not vf[1]
    for h in range(H):
        want = O.estimate(oc, rows[subs[h]])
        assert bool(vf[h]) == (len(want) > 0), h
        if vf[h]:
            scale = max(1.0, np.abs(ps[h]).max())
            tol = 1e-8 if h not in (2, 3) else 1e-3    # ill-conditioned on purpose: both paths are SVD
            assert np.abs(pf[h] - ps[h]).max() <= tol * scale, h
            assert cf[h] == O.scan(oc, pf[h], rows)[0]


def test_sharded_step_dense_and_us(ctx):
    """ShardedRansac.step() (lsqr_winner_moments) for the models without an origin convention"""
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac
    rows = synth.dense(60_000, 16, 0.05, seed=71)[0]
    ctx.set_model(L.DENSE, 16, 0.1).upload(rows)
    sr = ShardedRansac(ctx, Comm(None))
    votes, gidx, par = sr.batch(3, 1, 300)
    fit, cnt, info = sr.fit(par)
    r = sr.step(3, 1, 300)
    assert (r[0], r[1], r[4]) == (votes, gidx, cnt) and np.array_equal(r[2], par)
    assert np.allclose(r[3], fit, rtol=1e-12, atol=1e-12)
    rec = synth.us_single(30_000, 0.3, seed=8)[0]
    for ls in (L.LS_ANALYTIC, L.LS_ITERATIVE):
        ctx.set_model(L.US_SINGLE, 0, 3.0, ls).upload(rec)
        sr = ShardedRansac(ctx, Comm(None))
        votes, gidx, par = sr.batch(5, 0, 200)
        fit, cnt, info = sr.fit(par)
        r = sr.step(5, 0, 200)
        assert (r[0], r[1], r[4]) == (votes, gidx, cnt) and np.array_equal(r[2], par)
        assert len(r[3]) == len(fit) and np.allclose(r[3], fit, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n", [2, 100, 30_007])
def test_ray_intersection_device_path(ctx, n):
    """RayIntersection: estimate() / agree() bit-exact, least squares within 1e-6, RANSAC = serial loop"""
    aux = 0.017453292519943295769236907684886
    data, target, lab = synth.rays(n, 0.3 if n > 2 else 0.0, seed=600 + n)
    oc = O.cfg(O.RAY, 3, 1.0, aux=aux)
    ctx.set_model(L.RAY, 3, 1.0, aux=aux).upload(data)
    H = 64
    subs = O.ctr_subsets(19, 0, H, n, 2)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in range(H):
        want = O.estimate(oc, data[subs[h]])
        assert bool(valid[h]) == (len(want) > 0), h
        if valid[h]:
            assert np.array_equal(par[h], want), h
            assert votes[h] == O.scan(oc, want, data)[0], h
    if valid.any():
        packed, bv, bi = ctx.best()
        m, cnt = ctx.mask_from_hypothesis(bi)
        wcnt, wmask = O.scan(oc, par[bi], data)
        assert cnt == wcnt == bv and np.array_equal(m, wmask)
        if cnt >= 2:
            fit, _ = ctx.ls_fit(use_mask=True)
            want = O.ls(oc, data, wmask)
            assert len(fit) == len(want)
            if len(want):
                assert np.allclose(fit, want, rtol=REL, atol=1e-6)
    if n > 2:
        r = ctx.ransac(0.999, seed=2)
        w = O.ransac(oc, data, 0.999, sampler="ctr", seed=2)
        assert r["info"].iterations == w["iters"]
        assert np.array_equal(r["consensus"], w["consensus"])
        assert np.allclose(r["params"], w["params"], rtol=REL, atol=1e-6)
        assert np.linalg.norm(r["params"] - target) < 1.0
    # all rays parallel: the least squares system is rank deficient -> empty
    par_rays = np.repeat(data[:1], 5, axis=0)
    par_rays[:, :3] += np.arange(5)[:, None]
    ctx.upload(par_rays)
    fit, _ = ctx.ls_fit(use_mask=False)
    assert len(fit) == 0


@pytest.mark.parametrize("n", [2, 500, 150_000])
def test_line2d_device_path(ctx, n):
    """Line2DParametersEstimator: estimate() bit-exact, the scan (exhaustive and two-level) bit-exact,
    closed-form fit within 1e-6, RANSAC equal to the serial loop"""
    data = synth.line(n, 0.4 if n > 2 else 0.0, seed=700 + n, dim=2)[0] if n > 2 else \
        np.array([[1.0, 2.0], [40.0, -7.5]])
    oc = O.cfg(O.LINE2D, 2, 0.5)
    ctx.set_model(L.LINE2D, 2, 0.5).upload(data)
    H = 100
    subs = O.ctr_subsets(29, 0, H, n, 2)
    ctx.hypotheses_from_subsets(subs)
    plain = _scan_votes(ctx, 0)
    assert np.array_equal(_scan_votes(ctx, 2), plain)
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in range(0, H, 7):
        want = O.estimate(oc, data[subs[h]])
        assert bool(valid[h]) == (len(want) > 0)
        if valid[h]:
            assert np.array_equal(par[h], want)
            assert plain[h] == O.scan(oc, want, data)[0]
    fit, _ = ctx.ls_fit(use_mask=False)
    want = O.ls(oc, data)
    assert len(fit) == len(want) == 4
    assert np.allclose(fit, want, rtol=REL, atol=1e-6)
    if n > 2:
        r = ctx.ransac(0.999, seed=6)
        w = O.ransac(oc, data, 0.999, sampler="ctr", seed=6)
        assert r["info"].iterations == w["iters"] and np.array_equal(r["consensus"], w["consensus"])
        assert np.allclose(r["params"], w["params"], rtol=REL, atol=1e-6)
    ctx.upload(np.repeat(data[:1], 4, axis=0))           # all the same point -> empty
    fit, _ = ctx.ls_fit(use_mask=False)
    assert len(fit) == 0


@pytest.mark.parametrize("model", [L.PLANE, L.SPHERE, L.LINE])
def test_filters_and_cell_scan_across_scales(ctx, model):
    """the error bands of the fp32 filters and of the box tests scale with the data: the same scene at
    magnitudes from 1e-6 to 1e9 (thresholds from 'a few ulps of fp32' to 'most of the scene'), off-centre
    by 50 scene sizes, must give the votes of the plain fp64 kernel in every mode"""
    base = _data(model, 3, 70_000, 2024, outliers=0.5)          # scene size ~ 1000, delta 0.5
    ref_subs = O.ctr_subsets(41, 0, 160, len(base), {L.PLANE: 3, L.SPHERE: 4, L.LINE: 2}[model])
    for scale in (1e-6, 1e-3, 1.0, 1e4, 1e9):
        for rel_delta, shift in ((0.5, 0.0), (1e-4, 0.0), (40.0, 0.0), (0.5, 5e4)):
            data = np.ascontiguousarray((base + shift) * scale)
            delta = rel_delta * scale
            ctx.set_model(model, 3, delta).upload(data)
            ctx.hypotheses_from_subsets(ref_subs)
            ctx.set_option("scan_filter", 0)
            exact = _scan_votes(ctx, 0)
            ctx.set_option("scan_filter", 1)
            assert np.array_equal(_scan_votes(ctx, 0), exact), (scale, rel_delta, shift, "filter")
            assert np.array_equal(_scan_votes(ctx, 2), exact), (scale, rel_delta, shift, "cells")
    oc = O.cfg(model, 3, delta)
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in (0, 77, 159):
        if valid[h]:
            assert exact[h] == O.scan(oc, par[h], data)[0]


def test_dense_mfma_scan_arrangements_agree(ctx):
    """the fp64, the fp32 and the fp16-split matrix-core filter scans and the exact VALU kernel count the same votes; ragged row and
    hypothesis counts"""
    rows = synth.dense(70_013, 64, 0.05, seed=313)[0]
    ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
    ctx.hypotheses_sample(17, 0, 333)
    res = []
    for filt, f32 in ((1, 0), (0, 0), (1, 1), (1, 2)):
        ctx.set_option("scan_filter", filt)
        ctx.set_option("dense_f32", f32)
        ctx.scan()
        res.append(ctx.hypotheses(params=False)[2].copy())
    ctx.set_option("scan_filter", 1)
    ctx.set_option("dense_f32", 2)
    assert np.array_equal(res[0], res[1]) and np.array_equal(res[2], res[1]) and np.array_equal(res[3], res[1])
    assert res[0].max() > 1000


def test_cell_scan_falls_back_when_the_index_cannot_be_built(ctx):
    """the spatial index is an accelerator: a failed build (simulated) must leave the scan on the
    exhaustive kernels with the same votes, for the rest of the upload"""
    data = _data(L.PLANE, 3, 90_000, 77, outliers=0.5)
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    ctx.hypotheses_sample(2, 0, 3000)
    ref = _scan_votes(ctx, 0)
    os.environ["LSQR_TEST_FAIL_INDEX"] = "1"
    try:
        assert np.array_equal(_scan_votes(ctx, 2), ref)
        assert not ctx.index_info()["built"]
    finally:
        del os.environ["LSQR_TEST_FAIL_INDEX"]
    assert np.array_equal(_scan_votes(ctx, 2), ref)      # still disabled for this upload
    assert not ctx.index_info()["built"]
    ctx.upload(data)                                     # a new upload may try again
    ctx.hypotheses_sample(2, 0, 3000)
    assert np.array_equal(_scan_votes(ctx, 2), ref) and ctx.index_info()["built"]


# ---- PlanePhantomUSCalibration ---------------------------------------------------
def _phantom_close(got, want, rtol=REL, atol=1e-6):
    """The sign of a singular vector is arbitrary (PlanePhantom...Estimator.cxx:215-218 says as much of
    its scale factor): T3 (entries 3..10) is unaffected, t1_z and the 30 derived products flip together."""
    assert len(got) == len(want) == 41
    s = 1.0 if np.dot(got[38:41], want[38:41]) >= 0 else -1.0
    assert np.allclose(got[3:11], want[3:11], rtol=rtol, atol=atol)
    assert np.allclose(s * got[11:41], want[11:41], rtol=rtol, atol=atol)
    assert np.isclose(s * got[2], want[2], rtol=rtol, atol=atol)


def test_plane_phantom_device_path(ctx):
    """PlanePhantomUSCalibration: 31-frame null-vector solves against the oracle's SVD, agree() scan
    bit-exact on the device's own models, both least squares fits from the Gram matrix."""
    clean, truth, _ = synth.plane_phantom(80, 0.0, seed=41, pixel_sigma=0.0)
    noisy, _, _ = synth.plane_phantom(80, 0.0, seed=41, pixel_sigma=1.0)
    oc = O.cfg(O.PHANTOM, 0, 3.0, 1)
    ctx.set_model(L.PHANTOM, 0, 3.0, L.LS_ITERATIVE).upload(clean)
    assert (ctx.K, ctx.P, ctx.ND) == (31, 41, 15)
    H = 24
    subs = np.vstack([np.arange(31, dtype=np.uint32)[None, :], O.ctr_subsets(51, 0, H - 1, 80, 31)])
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    assert valid.all()
    assert synth.phantom_check(par[0], truth) and votes[0] == 80
    for h in range(H):
        want = O.estimate(oc, clean[subs[h]])
        _phantom_close(par[h], want, rtol=1e-5, atol=1e-5)   # null vector of an exact system: cond * eps
        assert votes[h] == O.scan(oc, par[h], clean)[0]
    # noisy frames: the minimal models differ, the scan must still match the oracle bit for bit
    ctx.upload(noisy)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    for h in range(H):
        want = O.estimate(oc, noisy[subs[h]])
        assert bool(valid[h]) == (len(want) > 0)
        _phantom_close(par[h], want, rtol=1e-5, atol=1e-5)
        wcnt, wmask = O.scan(oc, par[h], noisy)
        assert votes[h] == wcnt
    m, cnt = ctx.mask(par[3])
    wcnt, wmask = O.scan(oc, par[3], noisy)
    assert cnt == wcnt and np.array_equal(m, wmask)
    # least squares, analytic then iterative (the reference test's sequence, .cxx:167-186)
    ctx.set_model(L.PHANTOM, 0, 3.0, L.LS_ANALYTIC).upload(noisy)
    an, _ = ctx.ls_fit(use_mask=False)
    want = O.ls(O.cfg(O.PHANTOM, 0, 3.0, 0), noisy)
    _phantom_close(an, want)
    assert synth.phantom_check(an, truth)
    ctx.set_model(L.PHANTOM, 0, 3.0, L.LS_ITERATIVE).upload(noisy)
    it, info = ctx.ls_fit(use_mask=False)
    want = O.ls(oc, noisy)
    # the reference differentiates numerically (lmdif); so does the device path, on the 31 coefficients (phantom.h)
    _phantom_close(it, want, rtol=1e-6, atol=1e-6)
    assert synth.phantom_check(it, truth)
    assert 1 <= info.lm_info <= 4
    st_it, st_an = ctx.stats(it), ctx.stats(an)
    assert st_it[3] <= st_an[3] + 1e-9
    assert np.allclose(st_it, O.stats(oc, it, noisy), rtol=1e-9, atol=1e-12)
    res = ctx.residuals(it)
    assert res.shape == (80,) and res.min() == st_it[0] and res.max() == st_it[1]
    assert np.isclose(res.mean(), st_it[2], rtol=1e-12) and np.isclose((res ** 2).sum(), st_it[3], rtol=1e-12)
    assert np.array_equal(ctx.residuals(it, 10, 20), res[10:20])
    # refinement from a caller-supplied start (iterativeLeastSquaresEstimate): lm_begin / one lm_step on
    # the Gram block
    blk = ctx.moments(np.zeros(32), phase=1)
    assert len(blk) == ctx.moments_len(1) == 497 and blk[496] == 80
    ctx.lm_begin(an[:11])
    cont, _, ref, info2 = ctx.lm_step(blk)
    assert not cont and 1 <= info2.lm_info <= 4
    _phantom_close(ref, it, rtol=1e-7, atol=1e-7)
    # masked fit = fit of the subset
    mask = np.zeros(80, dtype=np.uint8)
    mask[5:70] = 1
    ctx.set_mask(mask)
    fit, _ = ctx.ls_fit(use_mask=True)
    _phantom_close(fit, O.ls(oc, noisy, mask), rtol=1e-6, atol=1e-6)
    # fewer than 31 frames: no estimate (.cxx:139-141)
    ctx.upload(noisy[:31])
    ctx.set_mask(np.r_[np.ones(30, np.uint8), np.zeros(1, np.uint8)])
    fit, _ = ctx.ls_fit(use_mask=True)
    assert len(fit) == 0


def test_plane_phantom_ransac_end_to_end(ctx):
    """examples/planeUSCalibration.cxx: RANSAC over the phantom estimator (k = 31, p = 0.999); same loop
    as the serial oracle on the same subset stream, then a large frame set through the batch chain."""
    rec, truth, lab = synth.plane_phantom(150, 0.08, seed=43, pixel_sigma=0.0)
    oc = O.cfg(O.PHANTOM, 0, 2.0, 1)
    ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ITERATIVE).upload(rec)
    r = ctx.ransac(0.999, seed=9)
    w = O.ransac(oc, rec, 0.999, sampler="ctr", seed=9)
    assert r["info"].iterations == w["iters"]
    assert np.array_equal(r["consensus"], w["consensus"])
    assert np.array_equal(r["consensus"].astype(bool), lab)
    assert synth.phantom_check(r["params"], truth)
    _phantom_close(r["params"], w["params"], rtol=1e-4, atol=1e-4)
    big, truth, lab = synth.plane_phantom(60_000, 0.05, seed=44, pixel_sigma=0.0)
    ctx.upload(big)
    b = ctx.batch_fit(3, 0, 512, want_consensus=True)
    assert np.array_equal(b["consensus"].astype(bool), lab)
    assert b["info"].best_votes == lab.sum() == b["info"].fit.n_used
    assert synth.phantom_check(b["params"], truth)


def test_plane_phantom_sharded_step(ctx):
    """multi-GPU step for the phantom: the slice's Gram block is what ranks exchange (497 doubles), the
    solve (analytic + LM) runs from the summed block; world 1 must equal the single-device chain"""
    from lsqrrecipes_amd.distributed import Comm, ShardedRansac
    rec, truth, lab = synth.plane_phantom(20_000, 0.05, seed=45, pixel_sigma=0.0)
    rec[:, 13:15] += np.random.default_rng(1).normal(0, 0.05, (len(rec), 2))   # mild pixel noise
    ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ITERATIVE).upload(rec)
    sr = ShardedRansac(ctx, Comm(None))
    r = sr.step(7, 0, 512)
    b = ctx.batch_fit(7, 0, 512)
    assert r is not None and (r[0], r[1]) == (b["info"].best_votes, b["info"].best_index)
    assert r[4] == b["info"].fit.n_used
    _phantom_close(r[3], b["params"], rtol=1e-9, atol=1e-9)
    assert synth.phantom_check(r[3], truth)
    # two slices summed = one pass (what a world-2 all-reduce produces)
    half = len(rec) // 2
    m, _ = ctx.mask(b["params"])
    blk = ctx.moments(np.zeros(3), 0, half, use_mask=True) + ctx.moments(np.zeros(3), half, len(rec), use_mask=True)
    whole = ctx.moments(np.zeros(3), use_mask=True)
    assert blk[-1] == whole[-1] == m.sum()
    assert np.allclose(blk, whole, rtol=1e-12, atol=1e-9)
    fit, _ = ctx.solve_moments(blk, np.zeros(3))
    _phantom_close(fit, b["params"], rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("n", [31, 1000, 300_001])
def test_plane_phantom_filter_scan_equals_exact_scan(ctx, n):
    """the packed fp32 pre-filter of the phantom scan (factored evaluation + rigorous band + exact re-check)
    against the plain fp64 kernel, with frames pushed onto the agree() threshold to a few ulp"""
    g = np.random.default_rng(7)
    rec, truth, lab = synth.plane_phantom_fast(n, 0.3, seed=90 + n % 7, pixel_sigma=0.3)
    clean = synth.plane_phantom_fast(max(n, 64), 0.0, seed=90 + n % 7, pixel_sigma=0.0)[0]
    delta = 2.0
    oc = O.cfg(O.PHANTOM, 0, delta, 1)
    # hypotheses from clean frames (near the truth) so that many frames sit near the threshold after the push
    ctx.set_model(L.PHANTOM, 0, delta, L.LS_ITERATIVE).upload(clean)
    subs = O.ctr_subsets(5, 0, 64, len(clean), 31)
    ctx.hypotheses_from_subsets(subs)
    par, valid, _ = ctx.hypotheses(votes=False)
    assert valid.all()
    evec = np.r_[par[0][11:41], par[0][2]]
    k = min(n, 2000)
    idx = g.choice(n, k, replace=False)
    uu, vv = rec[idx, 13:14], rec[idx, 14:15]
    rows = np.hstack([uu * rec[idx, :9], vv * rec[idx, :9], rec[idx, :9], rec[idx, 9:12], np.ones((k, 1))])
    target = delta * g.choice([-1.0, 1.0], k) * (1.0 + g.choice([-1, 1], k) * 10.0 ** g.uniform(-15, -5, k))
    rec[idx, 9:12] += np.outer(target - rows @ evec, par[0][38:41])
    # scan the pushed frames with the same 64 models: subsets index the CLEAN upload, so keep the models and
    # swap the data underneath by uploading rec with the clean minimal frames in front
    both = np.vstack([clean[:64], rec]) if n >= 64 else np.vstack([clean, rec])
    ctx.upload(both)
    ctx.hypotheses_from_subsets(subs)
    ctx.set_option("scan_filter", 0)
    ctx.scan()
    par0, _, exact = ctx.hypotheses()
    ctx.set_option("scan_filter", 1)
    ctx.scan()
    par1, _, filt = ctx.hypotheses()
    assert np.array_equal(exact, filt) and np.array_equal(par0, par1)
    for h in (0, 17, 63):
        assert exact[h] == O.scan(oc, par0[h], both)[0]
    # NaN / huge records never agree and do not disturb the filter
    both[64 + 5, 9] = np.nan         # (past the clean frames the subsets index)
    both[64 + 7, 13] = 1e300
    ctx.upload(both)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    _, _, v = ctx.hypotheses()
    for h in (0, 31):
        assert v[h] == O.scan(oc, par0[h], both)[0]


STEP_DEVICE_SCRIPT = r"""
import sys, numpy as np, torch
torch.cuda.init()                      # torch's HIP runtime must come up before the library's
sys.path.insert(0, %r)
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
from lsqrrecipes_amd.distributed import Comm, ShardedRansac
cases = [(L.PLANE, 3, 0), (L.SPHERE, 3, L.LS_GEOMETRIC), (L.LINE, 3, 0), (L.DENSE, 8, 0),
         (L.US_SINGLE, 0, L.LS_ANALYTIC), (L.US_SINGLE, 0, L.LS_ITERATIVE), (L.PHANTOM, 0, L.LS_ITERATIVE)]
for model, dim, ls in cases:
    if model == L.DENSE:
        data, delta = synth.dense(60_000, 8, 0.3, seed=5)[0], 0.1
    elif model == L.US_SINGLE:
        data, delta = synth.us_single_fast(50_000, 0.3, seed=5)[0], 3.0
    elif model == L.PHANTOM:
        data, delta = synth.plane_phantom_fast(30_000, 0.05, seed=5, pixel_sigma=0.02)[0], 2.0
    else:
        gen = {L.PLANE: synth.plane, L.SPHERE: synth.sphere, L.LINE: synth.line}[model]
        data, delta = gen(150_000, 0.5, seed=998, dim=dim)[0], 0.5
    with Context(0) as c1, Context(0) as c2:
        c1.set_model(model, dim, delta, ls).upload(data)
        c2.set_model(model, dim, delta, ls).upload(data)
        H = 256 if model in (L.DENSE, L.PHANTOM) else 2500
        want = ShardedRansac(c1, Comm(None)).step(7, 3, H)
        sr = ShardedRansac(c2, Comm(None))
        for stream in (None, torch.cuda.Stream()):       # torch's default stream, then a side stream
            with torch.cuda.stream(stream):
                got = sr.step_device(7, 3, H)
            assert (got[0], got[1], got[4]) == (want[0], want[1], want[4]), (model, got[:2], want[:2])
            assert np.array_equal(got[2], want[2])
            assert np.allclose(got[3], want[3], rtol=1e-12, atol=1e-12), (model, got[3], want[3])
        if ls == 0 and model != L.PHANTOM:   # closed-form fits: pipelined steps (enqueue i + 1 before reading i)
            blocking = [sr.step_device(7, b, H) for b in range(4)]
            piped = []
            sr.step_device(7, 0, H, slot=0)
            for b in range(1, 4):
                sr.step_device(7, b, H, slot=b & 1)
                piped.append(sr.step_device_wait((b - 1) & 1))
            piped.append(sr.step_device_wait(3 & 1))
            for g, w in zip(piped, blocking):
                assert (g[0], g[1], g[4]) == (w[0], w[1], w[4]) and np.array_equal(g[3], w[3])
        if model == L.PLANE:   # a batch without any valid hypothesis: None, like step()
            c2.upload(np.zeros((1000, 3)))
            assert sr.step_device(7, 0, 64) is None
        c2.set_stream(None)
print("step_device ok")
"""


def test_sharded_step_with_device_exchange_buffers():
    """ShardedRansac.step_device(): packed winner and moment block stay in device tensors between the
    calls (where the collectives act) and the host synchronises once; same result as step() for every
    model family.  Own process: torch's HIP runtime has to initialise before the library's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", STEP_DEVICE_SCRIPT % root], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and "step_device ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_step_device_world2_equals_world1():
    """two ranks (gloo, sharing the box's one GPU) run step_device(): all-reduce MAX / SUM on the device
    exchange tensors; winner, consensus count and fit equal one process scanning the whole batch"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547",
                        os.path.join(root, "tests", "dist_step_device.py")],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "world2 step_device ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("model,dim,ls", [(L.PLANE, 3, 0), (L.SPHERE, 3, L.LS_ALGEBRAIC), (L.LINE, 2, 0),
                                          (L.DENSE, 8, 0), (L.US_SINGLE, 0, L.LS_ANALYTIC), (L.PIVOT, 3, 0)])
def test_pipelined_batches_equal_blocking_batches(ctx, model, dim, ls):
    """lsqr_batch_fit_enqueue / _wait: the next batch is enqueued before the previous one is read; results
    equal lsqr_batch_fit's for the same stream ranges"""
    if model == L.DENSE:
        data, delta = synth.dense(60_000, 8, 0.3, seed=5)[0], 0.1
    elif model == L.US_SINGLE:
        data, delta = synth.us_single_fast(50_000, 0.3, seed=5)[0], 3.0
    elif model == L.PIVOT:
        data, delta = synth.pivot(40_000, 0.3, seed=5)[0], 1.0
    else:
        data, delta = _data(model, dim, 150_000, 997, outliers=0.5), 0.5
    ctx.set_model(model, dim, delta, ls).upload(data)
    H = 512
    want = [ctx.batch_fit(21, i * H, H) for i in range(5)]
    got = []
    ctx.batch_fit_enqueue(21, 0, H, slot=0)
    for i in range(1, 5):
        ctx.batch_fit_enqueue(21, i * H, H, slot=i & 1)
        got.append(ctx.batch_fit_wait((i - 1) & 1))
    got.append(ctx.batch_fit_wait(4 & 1))
    for w, g in zip(want, got):
        assert g["status"] == w["status"]
        assert (g["info"].best_votes, g["info"].best_index, g["info"].fit.n_used) == \
               (w["info"].best_votes, w["info"].best_index, w["info"].fit.n_used)
        assert np.array_equal(g["params"], w["params"])
    # slot discipline
    with pytest.raises(L.LsqrError):
        ctx.batch_fit_wait(0)                         # nothing in flight
    ctx.batch_fit_enqueue(21, 0, H, slot=1)
    with pytest.raises(L.LsqrError):
        ctx.batch_fit_enqueue(21, H, H, slot=1)      # unread result
    ctx.batch_fit_wait(1)


def test_pipelined_batches_refuse_host_in_the_loop_fits(ctx):
    ctx.set_model(L.SPHERE, 3, 0.5, L.LS_GEOMETRIC).upload(_data(L.SPHERE, 3, 5000, 1))
    with pytest.raises(L.LsqrError) as e:
        ctx.batch_fit_enqueue(1, 0, 64)
    assert e.value.status == L.ERR_INVALID


@pytest.mark.parametrize("model,dim,ls", [(L.PLANE, 3, 0), (L.PLANE, 2, 0), (L.SPHERE, 3, L.LS_GEOMETRIC),
                                          (L.LINE, 3, 0), (L.US_SINGLE, 0, L.LS_ANALYTIC), (L.ABSOR, 3, 0),
                                          (L.RAY, 3, 0)])
def test_fused_mask_and_moments_equal_two_kernels(ctx, model, dim, ls):
    """k_mask_moments (the winner's consensus mask and its moment block in one pass over the records) gives
    the same mask, count and bit-identical fit as k_mask followed by k_moments"""
    aux = 0.0
    if model == L.US_SINGLE:
        data, delta = synth.us_single_fast(70_001, 0.3, seed=6)[0], 3.0
    elif model == L.ABSOR:
        data, delta = synth.absolute_orientation(50_003, 0.3, seed=6)[0], 1.0
    elif model == L.RAY:
        data, delta, aux = synth.rays(50_003, 0.3, seed=6)[0], 1.0, 0.017453292519943295
    else:
        data, delta = _data(model, dim, 130_003, 996, outliers=0.5), 0.5
    ctx.set_model(model, dim, delta, ls, aux=aux).upload(data)
    ctx.set_option("fuse_mask", 0)
    two = ctx.batch_fit(31, 0, 700, want_consensus=True)
    ctx.set_option("fuse_mask", 1)
    one = ctx.batch_fit(31, 0, 700, want_consensus=True)
    assert one["status"] == two["status"] == L.OK
    assert np.array_equal(one["consensus"], two["consensus"])
    assert (one["info"].best_votes, one["info"].best_index, one["info"].fit.n_used) == \
           (two["info"].best_votes, two["info"].best_index, two["info"].fit.n_used)
    if model == L.US_SINGLE:
        # r04: the fused pass of the US calibrations accumulates on the fp64 matrix cores (another summation order:
        # the block agrees to rounding); with the per-lane accumulators ("us_mask_mfma" 0) it is bit-identical
        assert np.allclose(one["params"], two["params"], rtol=1e-9, atol=1e-9)
        ctx.set_option("us_mask_mfma", 0)
        lane = ctx.batch_fit(31, 0, 700, want_consensus=True)
        ctx.set_option("us_mask_mfma", 1)
        assert np.array_equal(lane["params"], two["params"]) and np.array_equal(lane["consensus"], two["consensus"])
    else:
        assert np.array_equal(one["params"], two["params"])
    assert one["consensus"].sum() == one["info"].fit.n_used


@pytest.mark.parametrize("model,gen,n", [(L.US_SINGLE, "single", 70_001), (L.US_SINGLE, "single", 63),
                                         (L.US_SINGLE, "single", 1_000_003), (L.US_POINTER, "pointer", 50_017)])
def test_us_mask_and_moments_on_the_matrix_cores(ctx, model, gen, n):
    """k_mask_moments_us_mfma (the fused mask + analytic moment block of the batch entry points) against the per-lane
    kernel and the oracle: consensus mask bit-identical, count equal, the block and the analytic fit equal to rounding
    (another summation order); ragged sizes (partial last tile, fewer frames than one tile) and a slice [begin, end)
    of the upload as the multi-GPU step uses it"""
    data = (synth.us_single_fast if gen == "single" else synth.us_pointer)(n, 0.3, seed=9)[0]
    omodel = O.US_SINGLE if gen == "single" else O.US_POINTER
    oc = O.cfg(omodel, 0, 3.0, 0)
    ctx.set_model(model, 0, 3.0, L.LS_ANALYTIC).upload(data)
    H = 256 if n > 1000 else 64
    res = {}
    for mf in (1, 0):
        ctx.set_option("us_mask_mfma", mf)
        r = ctx.batch_fit(5, 0, H, want_consensus=True)
        assert r["status"] == L.OK
        bi = int(r["info"].best_index)
        slices = []
        if n > 1000:
            for lo, hi in ((0, n), (n // 3 + 7, n - 11), (64, 64 + 129)):
                par, origin, blk, cnt = ctx.winner_moments(5, bi, lo, hi)
                slices.append((par.copy(), blk.copy(), cnt))
        res[mf] = (r, bi, slices)
    ctx.set_option("us_mask_mfma", 1)
    (r1, b1, s1), (r0, b0, s0) = res[1], res[0]
    assert b1 == b0 and np.array_equal(r1["consensus"], r0["consensus"])
    assert r1["info"].fit.n_used == r0["info"].fit.n_used == int(r1["consensus"].sum())
    assert np.allclose(r1["params"], r0["params"], rtol=1e-9, atol=1e-9)
    wpar = ctx.hypothesis(b1)[0] if False else None
    ctx.hypotheses_sample(5, b1, 1)
    wpar, _ = ctx.hypothesis(0)
    wcnt, wmask = O.scan(oc, wpar, data)
    assert wcnt == r1["info"].fit.n_used and np.array_equal(r1["consensus"], wmask)
    want = O.ls(oc, data, wmask)
    assert len(want) and np.allclose(r1["params"], want, rtol=1e-6, atol=1e-6)
    for (p1, k1, c1), (p0, k0, c0) in zip(s1, s0):
        assert c1 == c0 and np.array_equal(p1, p0) and k1[0] == k0[0] == c1
        assert np.allclose(k1, k0, rtol=1e-11, atol=1e-9 * max(1.0, np.abs(k0).max()))


# ---- bounded scan (cells.h: hypotheses that cannot win are not counted) ------------------------------------------
@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.SPHERE, 3), (L.LINE, 3), (L.PLANE, 2)])
@pytest.mark.parametrize("outliers", [0.5, 0.9])
def test_bounded_scan_keeps_winner_and_replay(ctx, model, dim, outliers):
    """scan_bound 1 (default of the batch entry points) against scan_bound 0: every counted hypothesis has its
    exact votes, every skipped one reports 0 and could not have become the running maximum, so the first-max
    winner, its consensus set, and the replay of RANSAC.hxx's adaptive loop over the batch are identical."""
    n, H = 300_000, 2048
    data = _data(model, dim, n, 4242, outliers=outliers)
    ls = L.LS_ALGEBRAIC if model == L.SPHERE else 0
    ctx.set_model(model, dim, 0.5, ls).upload(data)
    ctx.set_option("scan_index", 2)
    res = {}
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        r = ctx.batch_fit(77, 0, H, want_consensus=True)
        _, valid, votes = ctx.hypotheses(params=False)
        subs = O.ctr_subsets(77, 0, H, n, ctx.K)
        res[bound] = (r, votes.copy(), valid.copy(), context_replay(n, ctx.K, 0.999, subs, valid, votes))
    ctx.set_option("scan_bound", 1)
    ctx.set_option("scan_index", 1)
    (r0, v0, ok0, rp0), (r1, v1, ok1, rp1) = res[0], res[1]
    assert np.array_equal(ok0, ok1)
    assert (r0["info"].best_index, r0["info"].best_votes) == (r1["info"].best_index, r1["info"].best_votes)
    assert np.array_equal(r0["consensus"], r1["consensus"]) and np.array_equal(r0["params"], r1["params"])
    runmax = np.maximum.accumulate(np.where(ok0 > 0, v0, 0))
    same = v1 == v0
    skipped = ~same
    assert np.all(v1[skipped] == 0)
    assert np.all(v0[skipped][1:] <= runmax[np.flatnonzero(skipped)[1:] - 1]) if skipped.sum() > 1 else True
    assert not skipped[0] or v0[0] == 0
    assert rp0 == rp1                                   # same adaptive-loop state over the batch
    if outliers == 0.5:
        assert skipped.sum() > 0.5 * H                 # most random hypotheses are never counted


def context_replay(n, k, p, subs, valid, votes):
    from lsqrrecipes_amd.context import replay
    r = replay(n, k, p, subs, valid, votes)
    return (r["used"], r["i"], r["num_tries"], r["best_votes"], r["best_index"], r["has_best"], r["done"])


@pytest.mark.parametrize("case", ["dense_inliers", "small_box", "offset_1e6", "thin_threshold", "two_planes", "axis_off",
                                  "cells_of_256", "batch_8192"])
def test_rank_bounds_of_the_plane_are_exact(ctx, case):
    """axis.h (vote bounds by rank in axis-sorted cells): uploads whose cells along the model ARE flat, so that the
    rank bounds -- not the pilots -- do the pruning, in situations that stress their fp32 margins: a box a tenth of the
    size (cells a tenth of the size against the same threshold), a translation by 1e6 (cell-relative arithmetic), a
    threshold of the noise level (many observations at the boundary), two parallel planes 1.5 thresholds apart (lower
    bounds of one model compete with upper bounds of the other).  scan_bound 1 against counting everything: winner,
    consensus, parameters, replay identical; every counted hypothesis has its exact votes; every other one reports 0
    and could not have become the running maximum."""
    n, H = 2_000_000, (8192 if case == "batch_8192" else 2048)
    kw = dict(seed=77)
    delta = 0.5
    if case == "small_box":
        kw.update(box=100.0, sigma=0.2)
    if case == "thin_threshold":
        delta = 0.4                                   # = sigma of the inlier noise
    data = synth.plane(n, 0.3, **kw)[0]
    if case == "offset_1e6":
        data = data + np.array([1.0e6, -2.0e6, 3.0e6])
    if case == "two_planes":
        second, truth, _ = synth.plane(n // 2, 0.0, seed=77)
        data[: n // 2] = second + 0.75 * truth[:3]    # same plane, shifted by 1.5 delta along its normal
    ctx.set_option("scan_axis", 0 if case == "axis_off" else 1)
    ctx.set_option("scan_cell", 256 if case == "cells_of_256" else 0)
    ctx.set_model(L.PLANE, 3, delta).upload(data)
    ctx.set_option("scan_index", 2)
    res = {}
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        steps = []
        for s in range(3):                            # three consecutive batches: best_before carried as in lsqr_ransac
            r = ctx.batch_fit(91, s * H, H, want_consensus=True)
            _, valid, votes = ctx.hypotheses(params=False)
            steps.append((r, votes.copy(), valid.copy(), ctx.scan_workload() if bound else None,
                          ctx.scan_work() if bound else None))
        res[bound] = steps
    ctx.set_option("scan_bound", 1)
    ctx.set_option("scan_index", 1)
    ctx.set_option("scan_axis", 1)
    ctx.set_option("scan_cell", 0)
    for s in range(3):
        (r0, v0, ok0, _, _), (r1, v1, ok1, wl, wk) = res[0][s], res[1][s]
        assert wl["cell_points"] == (256 if case == "cells_of_256" else 512)
        assert np.array_equal(ok0, ok1)
        assert (r0["info"].best_index, r0["info"].best_votes) == (r1["info"].best_index, r1["info"].best_votes)
        assert np.array_equal(r0["consensus"], r1["consensus"]) and np.array_equal(r0["params"], r1["params"])
        skipped = v1 != v0
        assert np.all(v1[skipped] == 0)
        runmax = np.maximum.accumulate(np.where(ok0 > 0, v0, 0))
        idx = np.flatnonzero(skipped)
        assert np.all(v0[idx[idx > 0]] <= runmax[idx[idx > 0] - 1]) and (not skipped[0] or v0[0] == 0)
        if case != "axis_off":
            assert wk["candidates"] > 0, (wk, wl)      # the rank bounds ran
        # (two_planes: 85 % of the records lie on one of the two planes, most samples are all-inlier ones and have to
        #  be counted -- the selection still may not count everything)
        assert wl["second_pass"] + wl["pilots"] < (0.8 if case == "two_planes" else 0.5) * H, wl


def test_bounded_scan_inside_adaptive_ransac(ctx):
    """lsqr_ransac with batches large enough for the bounded scan (80 % outliers: thousands of iterations), the
    best votes of earlier batches carried as the lower bound: same iterations, winner and consensus as counting all"""
    n = 250_000
    data, truth, lab = synth.plane(n, 0.8, seed=31)
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    ctx.set_option("scan_index", 2)
    out = []
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        r = ctx.ransac(0.999, seed=5)
        out.append((r["info"].iterations, r["info"].best_index, r["info"].best_votes, r["consensus"].copy(), r["params"]))
    ctx.set_option("scan_bound", 1)
    ctx.set_option("scan_index", 1)
    assert out[0][:3] == out[1][:3] and out[0][0] > 1500
    assert np.array_equal(out[0][3], out[1][3]) and np.array_equal(out[0][4], out[1][4])
    w = O.ransac(O.cfg(O.PLANE, 3, 0.5), data, 0.999, sampler="ctr", seed=5)
    assert out[1][0] == w["iters"] and np.array_equal(out[1][3], w["consensus"])


def test_cell_scan_with_the_filter_switched_off_per_hypothesis(ctx):
    """small spheres far from the origin: the fp32 band of most hypotheses exceeds a quarter of the squared radius,
    prepare_f32 switches their filter off (t_in = -inf, t_out = +inf) and the two-level scan has to send every
    observation of every cell down the exact path -- whatever the fp32 measure evaluates to.  Same for a plane model
    whose normal is not a unit vector (|n_i| > 1: E = inf)."""
    g = np.random.default_rng(12)
    n = 70_000
    c0 = np.array([3.0e6, -2.0e6, 1.0e6])
    u = g.normal(size=(n, 3))
    pts = c0 + 2.0 * u / np.linalg.norm(u, axis=1)[:, None] + g.normal(scale=0.01, size=(n, 3))
    pts[::3] = c0 + g.uniform(-6, 6, size=(len(pts[::3]), 3))
    oc = O.cfg(O.SPHERE, 3, 0.05)
    ctx.set_model(L.SPHERE, 3, 0.05).upload(pts)
    ctx.hypotheses_sample(4, 0, 300)
    plain = _scan_votes(ctx, 0)
    assert np.array_equal(_scan_votes(ctx, 2), plain)
    par, valid, _ = ctx.hypotheses(votes=False)
    for h in range(0, 300, 37):
        if valid[h]:
            assert plain[h] == O.scan(oc, par[h], pts)[0]
    assert plain.max() > 0.3 * n


# ---- lanes of the pipelined batch entry points (lsqr_hip.h: lsqr_batch_fit_enqueue / _wait, option batch_lanes) ----
@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.SPHERE, 3)])
def test_batch_lanes_give_the_single_stream_results(ctx, model, dim):
    """batches spread over three streams (each lane: own stream, buffers and index, attached to the context's records)
    return exactly what one stream returns -- winner, votes, consensus size and fit of every batch -- also after the
    records are replaced while lanes exist"""
    n, H, nb = 200_000, 2048, 9
    ls = L.LS_ALGEBRAIC if model == L.SPHERE else 0
    for seed in (11, 12):                       # second round: re-upload with lanes alive
        data = _data(model, dim, n, 900 + seed, outliers=0.6)
        ctx.set_model(model, dim, 0.5, ls).upload(data)
        ctx.set_option("scan_index", 2)
        res = {}
        for lanes in (1, 3):
            ctx.set_option("batch_lanes", lanes)
            ring, out = 2 * lanes, [None] * nb
            for i in range(nb):
                if i >= ring:
                    out[i - ring] = ctx.batch_fit_wait((i - ring) % ring)
                ctx.batch_fit_enqueue(seed, i * H, H, slot=i % ring)
            for i in range(max(0, nb - ring), nb):
                out[i] = ctx.batch_fit_wait(i % ring)
            res[lanes] = [(r["info"].best_index, r["info"].best_votes, r["info"].fit.n_used, tuple(r["params"]))
                          for r in out]
        assert res[1] == res[3]
        one = ctx.batch_fit(seed, 4 * H, H)     # the blocking entry point on the same batch
        assert (one["info"].best_index, one["info"].best_votes, tuple(one["params"])) == \
            (res[3][4][0], res[3][4][1], res[3][4][3])
    ctx.set_option("batch_lanes", 3)
    ctx.set_option("scan_index", 1)
    with pytest.raises(Exception):
        ctx.batch_fit_enqueue(1, 0, H, slot=6)  # three lanes: slots 0..5
    ctx.set_option("batch_lanes", 4)            # the default


def _bounded_equals_full(ctx, model, dim, data, H, seed=77):
    """scan_bound 1 against 0 on one batch: same valid flags, winner, consensus set, fit and replay state; every
    hypothesis the bounded scan counted has its exact votes, every other one reports 0 and could not have become the
    running maximum"""
    n = len(data)
    ls = L.LS_ALGEBRAIC if model == L.SPHERE else 0
    ctx.set_model(model, dim, 0.5, ls).upload(data)
    ctx.set_option("scan_index", 2)
    res = {}
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        r = ctx.batch_fit(seed, 0, H, want_consensus=True)
        _, valid, votes = ctx.hypotheses(params=False)
        subs = O.ctr_subsets(seed, 0, H, n, ctx.K)
        res[bound] = (r, votes.copy(), valid.copy(), context_replay(n, ctx.K, 0.999, subs, valid, votes))
    ctx.set_option("scan_bound", 1)
    ctx.set_option("scan_index", 1)
    (r0, v0, ok0, rp0), (r1, v1, ok1, rp1) = res[0], res[1]
    assert np.array_equal(ok0, ok1)
    assert (r0["info"].best_index, r0["info"].best_votes) == (r1["info"].best_index, r1["info"].best_votes)
    if r0["info"].best_votes:
        assert np.array_equal(r0["consensus"], r1["consensus"]) and np.array_equal(r0["params"], r1["params"])
    runmax = np.maximum.accumulate(np.where(ok0 > 0, v0, 0))
    skipped = v1 != v0
    assert np.all(v1[skipped] == 0)
    idx = np.flatnonzero(skipped)
    assert np.all(v0[idx[idx > 0]] <= runmax[idx[idx > 0] - 1])
    assert not skipped[0] or v0[0] == 0
    assert rp0 == rp1
    return r1, v1, skipped


@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.SPHERE, 3), (L.LINE, 3), (L.LINE, 2), (L.SPHERE, 2)])
@pytest.mark.parametrize("n,H,outliers", [(5_000, 1024, 0.5), (70_001, 1024, 0.0), (70_001, 8192, 0.3),
                                          (130_000, 4096, 1.0), (66_000, 1027, 0.5)])
def test_bounded_scan_shapes(ctx, model, dim, n, H, outliers):
    """the statically balanced second level (k_scan_pairs) on the shapes that stress its bookkeeping: a handful of
    cells, a last partial cell and chunk, every hypothesis a near-model one (no outliers: the second pass holds the
    whole batch), no structure at all (only outliers), the largest batch the selection kernels take, a batch size
    that is not a multiple of 64"""
    data = _data(model, dim, n, 31337 + n + H, outliers=outliers)
    r, votes, skipped = _bounded_equals_full(ctx, model, dim, data, H)
    if r["info"].best_votes:   # the winner's votes against the oracle's scan of the same parameters
        cfg = O.cfg({L.PLANE: O.PLANE, L.SPHERE: O.SPHERE, L.LINE: O.LINE}[model], dim, 0.5, O.LS_ALGEBRAIC)
        subs = O.ctr_subsets(77, 0, H, n, ctx.K)
        par = O.estimate(cfg, data[subs[r["info"].best_index]])
        assert len(par) and O.scan(cfg, par, data)[0] == r["info"].best_votes


def test_bounded_scan_duplicates_and_lattice(ctx):
    """a lattice (thousands of identical coordinates per axis: degenerate cell boxes, equal Morton keys) and records
    that occur twice: the cost table, the equal split and the exact re-checks of k_scan_pairs against counting
    everything (non-finite records switch the filters and with them the two-level scan off: covered by
    test_cell_scan_nonfinite_duplicate_and_flat_data)"""
    rng = np.random.default_rng(5)
    g = rng.integers(0, 40, size=(90_000, 3)).astype(np.float64)           # lattice, many duplicates
    plane_pts = np.column_stack([rng.integers(0, 40, 60_000), rng.integers(0, 40, 60_000)]).astype(np.float64)
    on = np.column_stack([plane_pts, 0.25 * plane_pts[:, 0] + 0.5 * plane_pts[:, 1] + rng.uniform(-0.4, 0.4, 60_000)])
    data = np.vstack([g, on, on[:20_000]])
    rng.shuffle(data)
    r, votes, skipped = _bounded_equals_full(ctx, L.PLANE, 3, data, 2048, seed=9)
    assert r["info"].best_votes > 50_000 and skipped.sum() > 0


def test_bounded_scan_with_filters_switched_off(ctx):
    """the bounded path (batch entry point, 1024 hypotheses) on the data of the test above: most hypotheses have their
    filter off, survive every cell (vote bound = N), and go the exact way through k_scan_pairs; delta 0.05"""
    g = np.random.default_rng(12)
    n = 70_000
    c0 = np.array([3.0e6, -2.0e6, 1.0e6])
    u = g.normal(size=(n, 3))
    pts = c0 + 2.0 * u / np.linalg.norm(u, axis=1)[:, None] + g.normal(scale=0.01, size=(n, 3))
    pts[::3] = c0 + g.uniform(-6, 6, size=(len(pts[::3]), 3))
    ctx.set_model(L.SPHERE, 3, 0.05, L.LS_ALGEBRAIC).upload(pts)
    ctx.set_option("scan_index", 2)
    res = {}
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        r = ctx.batch_fit(4, 0, 1024, want_consensus=True)
        _, valid, votes = ctx.hypotheses(params=False)
        res[bound] = (r["info"].best_index, r["info"].best_votes, r["consensus"].copy(), votes.copy())
    ctx.set_option("scan_bound", 1)
    ctx.set_option("scan_index", 1)
    assert res[0][:2] == res[1][:2] and np.array_equal(res[0][2], res[1][2])
    counted = res[1][3] != 0
    assert np.array_equal(res[1][3][counted], res[0][3][counted]) and res[0][1] > 0.3 * n


@pytest.mark.parametrize("model,dim", [(L.PLANE, 3), (L.SPHERE, 3), (L.LINE, 3), (L.PLANE, 2)])
@pytest.mark.parametrize("merge", [2, 4, 8])
def test_bounded_scan_with_merged_bound_boxes(ctx, model, dim, merge):
    """the vote bounds taken on boxes of 2 / 4 / 8 merged cells (the default only merges on uploads of millions of
    records): looser bounds, the same winner, consensus set, fit and replay state as counting everything"""
    data = _data(model, dim, 150_000, 4711 + merge, outliers=0.5)
    ctx.set_option("scan_bound_merge", merge)
    try:
        r, votes, skipped = _bounded_equals_full(ctx, model, dim, data, 2048)
    finally:
        ctx.set_option("scan_bound_merge", 0)
    assert r["info"].best_votes > 0.2 * len(data)


def test_lanes_and_bounded_scan_random_shapes(ctx):
    """a dozen random (model, records, batch size, outlier share) combinations in a row on ONE context: every batch
    through the lanes equals the blocking entry point with the bound switched off -- winner, votes of the winner,
    consensus size, parameters"""
    rng = np.random.default_rng(20260204)
    models = [(L.PLANE, 3), (L.SPHERE, 3), (L.LINE, 3), (L.PLANE, 2), (L.LINE, 2)]
    ctx.set_option("scan_index", 2)
    for it in range(12):
        model, dim = models[int(rng.integers(len(models)))]
        n = int(rng.integers(3_000, 260_000))
        H = int(rng.choice([1024, 1500, 2048, 4096, 6000]))
        outl = float(rng.choice([0.0, 0.3, 0.5, 0.8]))
        data = _data(model, dim, n, 5000 + it, outliers=outl)
        ls = L.LS_ALGEBRAIC if model == L.SPHERE else 0
        ctx.set_model(model, dim, 0.5, ls).upload(data)
        nb = int(rng.integers(1, 7))
        got = []
        for i in range(nb):
            ctx.batch_fit_enqueue(it, i * H, H, slot=i % 8)
        for i in range(nb):
            r = ctx.batch_fit_wait(i % 8)
            got.append((r["info"].best_index, r["info"].best_votes, r["info"].fit.n_used, tuple(r["params"])))
        ctx.set_option("scan_bound", 0)
        for i in range(nb):
            r = ctx.batch_fit(it, i * H, H)
            assert got[i] == (r["info"].best_index, r["info"].best_votes, r["info"].fit.n_used,
                              tuple(r["params"])), (it, i, model, dim, n, H, outl)
        ctx.set_option("scan_bound", 1)
    ctx.set_option("scan_index", 1)
