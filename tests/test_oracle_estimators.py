"""Oracle estimators against the assertions of the reference's own test programs
(testing/*Test.cxx), restated on seeded data."""
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from lsqrrecipes_amd import synth

COS5 = 0.99619469809174553229501040247389  # testing/PlaneParametersEstimatorTest.cxx:129


def test_plane_like_reference_test():
    """testing/PlaneParametersEstimatorTest.cxx:71-158: delta 0.5, agree on/off plane, exact and
    LS normals within 5 degrees, point-on-plane distance < delta."""
    g = np.random.default_rng(7)
    delta = 0.5
    c = O.cfg(O.PLANE, 3, delta)
    tri = g.uniform(-1000, 1000, (3, 3))
    n = np.cross(tri[1] - tri[0], tri[2] - tri[0])
    n /= np.linalg.norm(n)
    truth = np.concatenate([n, tri[0]])
    exact = O.estimate(c, tri)
    assert len(exact) == 6
    assert abs(exact[:3] @ n) > COS5
    assert abs((exact[3:] - tri[0]) @ n) < delta
    # agree: on the plane, and 2*delta off it
    assert O.agree(c, truth, tri[1])
    assert not O.agree(c, truth, tri[1] + 2 * delta * n)
    # 20 barycentric points + N(0,1) noise
    w = g.uniform(0, 1, (20, 3))
    w /= w.sum(1)[:, None]
    pts = w @ tri + g.normal(0, 1.0, (20, 3))
    ls = O.ls(c, pts)
    assert abs(ls[:3] @ n) > COS5
    assert abs((ls[3:] - tri[0]) @ n) < delta * 4  # noise sigma 1 => looser than the exact case
    # degenerate: collinear points -> empty
    col = np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2]], float)
    assert len(O.estimate(c, col)) == 0
    # too few points
    assert len(O.ls(c, pts[:2])) == 0


@pytest.mark.parametrize("dim", [2, 3, 4])
def test_sphere_like_reference_test(dim):
    """testing/SphereParametersEstimatorTest.cxx:208-234,137-155,432-468,504-506: centre distance
    and radius error <= 3 sigma for exact / algebraic / geometric."""
    sigma = 1.0
    pts, truth, _ = synth.sphere(50, 0.0, seed=100 + dim, dim=dim, sigma=sigma, box=50.0)
    c = O.cfg(O.SPHERE, dim, 0.5, O.LS_GEOMETRIC)
    clean = truth[:dim] + truth[dim] * np.eye(dim + 1, dim)  # axis points
    clean[dim] = truth[:dim] - truth[dim] * np.eye(dim)[0]
    ex = O.estimate(c, clean)
    assert len(ex) == dim + 1
    assert np.linalg.norm(ex[:dim] - truth[:dim]) < 1e-6 * max(1, truth[dim])
    alg = O.sphere_algebraic(dim, pts)
    geo = O.ls(c, pts)
    for est in (alg, geo):
        assert len(est) == dim + 1
        assert np.linalg.norm(est[:dim] - truth[:dim]) <= 3 * sigma
        assert abs(est[dim] - truth[dim]) <= 3 * sigma


def test_sphere_agree_cases():
    """testing/SphereParametersEstimatorTest.cxx:280-296: r = 2, delta = 0.5."""
    c = O.cfg(O.SPHERE, 2, 0.5)
    par = np.array([0.0, 0.0, 2.0])
    assert O.agree(c, par, [2.0, 0.0])
    assert O.agree(c, par, [2.4, 0.0])
    assert O.agree(c, par, [1.6, 0.0])
    assert not O.agree(c, par, [2.6, 0.0])
    assert not O.agree(c, par, [1.4, 0.0])
    assert not O.agree(c, par, [2.5, 0.0])  # strict <


def test_sphere_degenerate():
    c = O.cfg(O.SPHERE, 3, 0.5)
    cop = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], float)  # coplanar
    assert len(O.estimate(c, cop)) == 0
    c2 = O.cfg(O.SPHERE, 2, 0.5)
    assert len(O.estimate(c2, np.array([[0, 0], [1, 1], [2, 2]], float))) == 0


def test_line_like_reference_test():
    """testing/LineParametersEstimatorTest.cxx:98-213: delta 0.5, direction within 5 degrees."""
    g = np.random.default_rng(3)
    delta = 0.5
    c = O.cfg(O.LINE, 2, delta)
    a, b = g.uniform(-100, 100, 2), g.uniform(-100, 100, 2)
    d = (a - b) / np.linalg.norm(a - b)
    ex = O.estimate(c, np.stack([a, b]))
    assert np.allclose(ex[:2], d) and np.allclose(ex[2:], a)
    truth = np.concatenate([d, a])
    nrm = np.array([-d[1], d[0]])
    assert O.agree(c, truth, b)
    assert not O.agree(c, truth, b + 2 * delta * nrm)
    pts = a + g.uniform(-100, 100, 20)[:, None] * d + g.normal(0, 0.3, (20, 2))
    ls = O.ls(c, pts)
    assert abs(ls[:2] @ d) > COS5
    assert abs((ls[2:] - a) @ nrm) < delta
    # points closer than delta -> empty (LineParametersEstimator.hxx:33-35)
    assert len(O.estimate(c, np.stack([a, a + 0.1 * d]))) == 0


def test_dense_like_reference_test():
    """testing/DenseLinear...Test.cxx:72-146: 5x5 exact to 1e-10; 200x5 with <=5% noise to 0.1."""
    g = np.random.default_rng(5)
    c = O.cfg(O.DENSE, 5, 0.1)
    A = g.uniform(-1, 1, (5, 5))
    x = g.uniform(-1, 1, 5)
    rows = np.hstack([A, (A @ x)[:, None]])
    assert np.allclose(O.estimate(c, rows), x, atol=1e-10)
    rows2, x2, _ = synth.dense(200, 5, 0.0, seed=9)
    assert np.allclose(O.ls(c, rows2), x2, atol=0.1)
    # agree
    assert O.agree(c, x, rows[0])
    bad = rows[0].copy()
    bad[5] += 0.2
    assert not O.agree(c, x, bad)
    # singular minimal set -> empty
    sing = rows.copy()
    sing[4] = sing[3]
    assert len(O.estimate(c, sing)) == 0


def _euler_close(a, b, tol):
    d = np.abs((a - b + np.pi) % (2 * np.pi) - np.pi)
    return np.all(d < tol)


def test_us_single_like_reference_test():
    """testing/SinglePointTargetUSCalibration...Test.cxx:179-220,466-552: delta 3; minimal-set
    estimate from 4 clean frames agrees with clean data; iterative LS on noisy data within
    1 mm (t3), 1 degree (one of two Euler solutions), scales within 1.0."""
    c = O.cfg(O.US_SINGLE, 0, 3.0, 1)
    clean, truth, _ = synth.us_single(50, 0.0, seed=21, pixel_sigma=0.0)
    ex = O.estimate(c, clean[:4])
    assert len(ex) == 20
    assert O.agree(c, ex, clean[0]) and O.agree(c, ex, clean[30])
    assert len(O.estimate(c, clean[:5])) == 0  # exactly 4 required (:21)
    noisy, truth, _ = synth.us_single(50, 0.0, seed=21, pixel_sigma=1.0)
    ls = O.ls(c, noisy)
    assert len(ls) == 20
    assert np.linalg.norm(ls[3:6] - truth[3:6]) < 1.0
    wz, wy, wx = truth[6:9]
    alt = np.array([wz + np.pi, np.pi - wy, wx + np.pi])
    deg = np.pi / 180
    assert _euler_close(ls[6:9], truth[6:9], deg) or _euler_close(ls[6:9], alt, deg)
    assert abs(ls[9] - truth[9]) < 1.0 and abs(ls[10] - truth[10]) < 1.0
    # the trailing 9 entries are the rotation products of the 11 (cxx:303-327)
    R3 = synth.euler_zyx(*ls[6:9])
    assert np.allclose(ls[11:14], ls[9] * R3[:, 0])
    assert np.allclose(ls[14:17], ls[10] * R3[:, 1])
    assert np.allclose(ls[17:20], R3[:, 2])


def test_us_pointer_like_reference_test():
    c = O.cfg(O.US_POINTER, 0, 3.0, 1)
    clean, truth, _ = synth.us_pointer(50, 0.0, seed=22, pixel_sigma=0.0)
    ex = O.estimate(c, clean[:3])
    assert len(ex) == 17
    assert O.agree(c, ex, clean[0]) and O.agree(c, ex, clean[40])
    noisy, truth, _ = synth.us_pointer(50, 0.0, seed=22, pixel_sigma=1.0)
    ls = O.ls(c, noisy)
    assert len(ls) == 17
    assert np.linalg.norm(ls[0:3] - truth[0:3]) < 1.0
    assert abs(ls[6] - truth[6]) < 1.0 and abs(ls[7] - truth[7]) < 1.0


def test_crosswire_experimental_data(golden_dir):
    """testing/SinglePointTarget...Test.cxx:115-166 feeds testing/Data/crossWirePhantom*.txt and
    only prints; here: the oracle must produce a calibration whose residuals are sane."""
    T = np.loadtxt(os.path.join(golden_dir, "ref_data", "crossWirePhantomTransformations.txt"))
    q = np.loadtxt(os.path.join(golden_dir, "ref_data", "crossWirePhantom2DPoints.txt"))
    m = q.shape[0]
    rec = np.zeros((m, 15))
    T = T.reshape(m, 3, 4)
    rec[:, 0:9] = T[:, :, :3].reshape(m, 9)
    rec[:, 9:12] = T[:, :, 3]
    rec[:, 13:15] = q
    c = O.cfg(O.US_SINGLE, 0, 3.0, 1)
    ls = O.ls(c, rec)
    assert len(ls) == 20
    st = O.stats(c, ls, rec)
    assert st[2] < 3.0  # mean residual in mm
    an = O.us_analytic(O.US_SINGLE, rec)
    assert O.stats(c, ls, rec)[3] <= O.stats(c, an, rec)[3] + 1e-9  # LM does not worsen


# ---- AbsoluteOrientation and PivotCalibration -------------------------------
def _apply(par, pts):
    R = synth.quat_to_matrix(par[:4] / np.linalg.norm(par[:4]))
    return pts @ R.T + par[4:7]


def test_absolute_orientation_like_reference_test():
    """testing/AbsoluteOrientationParametersEstimatorTest.cxx:19-118: 10 pairs, noise sigma 5/3,
    delta 1; exact estimate from clean pairs and LS from noisy pairs keep the maximal target
    registration error below 3 sigma; agree() accepts a clean pair and rejects the outlier."""
    sigma = 5.0 / 3.0
    clean, truth, _ = synth.absolute_orientation(10, 0.0, seed=91, sigma=0.0)
    noisy = clean.copy()
    g = np.random.default_rng(5)
    noisy[:, 3:] += g.normal(0, sigma, (10, 3))
    targets = g.uniform(-100, 100, (10, 3))
    want = _apply(truth, targets)
    c = O.cfg(O.ABSOR, 3, 1.0)
    exact = O.estimate(c, clean[:3])
    assert len(exact) == 7
    assert np.linalg.norm(_apply(exact, targets) - want, axis=1).max() < 1e-9
    ls = O.ls(c, noisy)
    assert len(ls) == 7
    assert np.linalg.norm(_apply(ls, targets) - want, axis=1).max() < 3 * sigma
    assert O.agree(c, truth, clean[0])
    outlier = clean[1].copy()
    outlier[3] += 10 * sigma
    assert not O.agree(c, truth, outlier)
    # collinear first points -> empty (AbsoluteOrientation...cxx:48)
    col = clean[:3].copy()
    col[:, :3] = np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2]], float)
    assert len(O.estimate(c, col)) == 0
    assert len(O.estimate(c, clean[:2])) == 0 and len(O.ls(c, clean[:2])) == 0


def test_absolute_orientation_ls_matches_svd_solution():
    """Horn's quaternion solution equals the SVD (Kabsch) solution of the same least squares problem"""
    data, truth, _ = synth.absolute_orientation(200, 0.0, seed=17, sigma=1.0)
    ls = O.ls(O.cfg(O.ABSOR, 3, 1.0), data)
    a, b = data[:, :3], data[:, 3:]
    ma, mb = a.mean(0), b.mean(0)
    U, _, Vt = np.linalg.svd((b - mb).T @ (a - ma))
    D = np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    assert np.allclose(synth.quat_to_matrix(ls[:4]), R, atol=1e-10)
    assert np.allclose(ls[4:], mb - R @ ma, atol=1e-8)


def test_pivot_calibration_known_answers_of_the_reference():
    """testing/PivotCalibrationParametersEstimatorTest.cxx:47-119 on its own data file: the exact
    estimate from poses {0, n/2, n-1} and the least squares estimate against the hard-coded vectors
    (tolerance 1.0 there; reproduced here to the printed digits), agree() on the minimal set."""
    rows = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "ref_data",
                                   "pivotCalibrationData.txt"))
    F = synth.frames_from_pose_rows(rows)
    n = len(F)
    assert n == 481
    c = O.cfg(O.PIVOT, 3, 1.0)
    mins = F[[0, int(n / 2.0), n - 1]]
    exact = O.estimate(c, mins)
    assert np.allclose(exact, [-18.586, 1.98134, -157.439, 146.965, -62.0497, -1042.87], atol=6e-3)
    for f in mins:
        assert O.agree(c, exact, f)
    ls = O.ls(c, F)
    assert np.allclose(ls, [-17.7799, 1.1113, -156.865, 146.901, -62.9689, -1042.14], atol=6e-3)
    # rank deficiency: three identical poses -> empty (:44-45)
    assert len(O.estimate(c, np.repeat(F[:1], 3, axis=0))) == 0


def test_pivot_ransac_on_the_reference_outlier_file():
    """examples/pivotCalibration.cxx on examples/Data/pivotCalibrationDataWithOutliers.txt: RANSAC
    recovers the translations the clean file gives (within the 1 mm threshold)"""
    rows = np.loadtxt(os.path.join(os.path.dirname(__file__), "golden", "ref_data",
                                   "pivotCalibrationDataWithOutliers.txt"))
    F = synth.frames_from_pose_rows(rows)
    c = O.cfg(O.PIVOT, 3, 1.0)
    r = O.ransac(c, F, 0.999, sampler="ctr", seed=3)
    assert 0.5 < r["fraction"] < 0.8
    assert np.allclose(r["params"], [-17.78, 1.11, -156.87, 146.90, -62.97, -1042.14], atol=1.0)


def test_ray_intersection_like_reference_test():
    """testing/RayIntersectionParametersTest.cxx:58-117: agree() at the known point, two clean rays
    meet within the distance threshold, least squares over noisy rays succeeds; plus the degenerate
    cases of estimate() (parallel rays, intersection behind an origin)."""
    clean, target, _ = synth.rays(2, 0.0, seed=3, sigma=0.0)
    noisy, target2, _ = synth.rays(10, 0.0, seed=4, sigma=20.0)
    c = O.cfg(O.RAY, 3, 0.5, aux=0.017453292519943295769236907684886)
    assert O.agree(c, target, clean[0])
    est = O.estimate(c, clean)
    assert len(est) == 3 and np.linalg.norm(est - target) <= 0.5
    ls = O.ls(c, noisy)
    assert len(ls) == 3 and np.linalg.norm(ls - target2) < 3 * 20.0
    par = clean.copy()
    par[1, 3:] = par[0, 3:]                       # parallel rays (:51)
    assert len(O.estimate(c, par)) == 0
    back = clean.copy()
    back[1, 3:] *= -1                             # the lines meet, the rays do not (:63)
    assert len(O.estimate(c, back)) == 0
    assert not O.agree(c, target, back[1])        # t < 0 (:177)
    # (all rays parallel: the reference's absolute sigma <= 2.2e-16 rank test (:134-138) only fires when the
    # rounding noise of 1 - |n|^2 happens to vanish; the literal restatement inherits that, the device
    # declares rank deficiency relative to the largest singular value -- tests/test_gpu_parity.py)
    # the least squares point minimises the sum of squared point-line distances
    A = np.zeros((3, 3)); b = np.zeros(3)
    for p_, n_ in zip(noisy[:, :3], noisy[:, 3:]):
        Pm = np.eye(3) - np.outer(n_, n_)
        A += Pm; b += Pm @ p_
    assert np.allclose(ls, np.linalg.solve(A, b), rtol=1e-10, atol=1e-8)


def test_line2d_like_reference_test():
    """testing/LineParametersEstimatorTest.cxx:121-213 for Line2DParametersEstimator: two exact points
    and 20 noisy ones, delta 0.5: normal within 5 degrees of the truth, point on the line within delta;
    agree() on / 2 delta off; the degenerate cases of estimate() and the fit."""
    g = np.random.default_rng(23)
    p0, p1 = g.uniform(-100, 100, 2), g.uniform(-100, 100, 2)
    d = (p1 - p0) / np.linalg.norm(p1 - p0)
    nrm = np.array([-d[1], d[0]])
    c = O.cfg(O.LINE2D, 2, 0.5)
    est = O.estimate(c, np.array([p0, p1]))
    assert len(est) == 4 and abs(est[:2] @ nrm) > COS5 and abs((est[2:] - p0) @ nrm) < 0.5
    assert O.agree(c, est, p1) and not O.agree(c, est, p1 + 1.0 * nrm)
    pts = p0 + np.outer(g.uniform(-100, 100, 20), d) + g.normal(0, 0.2, (20, 2))
    ls = O.ls(c, pts)
    assert len(ls) == 4 and abs(ls[:2] @ nrm) > COS5 and abs((ls[2:] - p0) @ nrm) < 0.5
    assert len(O.estimate(c, np.array([p0, p0 + 0.1]))) == 0            # closer than delta
    assert len(O.ls(c, np.repeat([p0], 5, axis=0))) == 0                 # all the same point
    vert = np.stack([np.full(6, 3.0), np.arange(6.0)], axis=1)           # zero variance in x: n = (1, 0)
    assert np.allclose(O.ls(c, vert)[:2], [1.0, 0.0])
    # agrees with the hyperplane estimator's fit up to the sign of the normal
    pl = O.ls(O.cfg(O.PLANE, 2, 0.5), pts)
    assert abs(abs(pl[:2] @ ls[:2]) - 1) < 1e-9 and np.allclose(pl[2:], ls[2:], atol=1e-9)


# ---- PlanePhantomUSCalibration -------------------------------------------------
def test_plane_phantom_like_reference_test():
    """testing/PlanePhantomUSCalibrationParametersEstimatorTest.cxx:131-187: 50 frames, pixel noise 1,
    delta 3: minimal estimate from the first 31 clean frames, then the analytic and the iterative
    least squares on the noisy frames, each within the test's tolerances (synth.phantom_check)."""
    c = O.cfg(O.PHANTOM, 0, 3.0, 1)
    clean, truth, _ = synth.plane_phantom(50, 0.0, seed=31, pixel_sigma=0.0)
    ex = O.estimate(c, clean[:31])
    assert len(ex) == 41
    assert synth.phantom_check(ex, truth)
    assert O.agree(c, ex, clean[0]) and O.agree(c, ex, clean[45])
    assert len(O.estimate(c, clean[:30])) == 0 and len(O.estimate(c, clean[:32])) == 0  # exactly 31 (.cxx:19)
    noisy, truth, _ = synth.plane_phantom(50, 0.0, seed=31, pixel_sigma=1.0)
    an = O.ls(O.cfg(O.PHANTOM, 0, 3.0, 0), noisy)
    it = O.ls(c, noisy)
    assert len(an) == 41 and len(it) == 41
    assert synth.phantom_check(an, truth) and synth.phantom_check(it, truth)
    sse = lambda p: O.stats(c, p, noisy)[3]
    assert sse(it) <= sse(an) + 1e-9  # LM does not worsen the analytic start
    # the 30 trailing entries are products of the 11 (.cxx:325-354)
    assert np.allclose(it[38:41], [-np.sin(it[0]), np.cos(it[0]) * np.sin(it[1]), np.cos(it[0]) * np.cos(it[1])])
    assert np.allclose(it[29:32], it[3:6] * it[38])


def test_plane_phantom_ransac_rejects_off_plane_frames():
    """RANSAC over the phantom estimator (k = 31): frames whose target point is off the plane are
    not in the consensus set."""
    rec, truth, lab = synth.plane_phantom(120, 0.1, seed=32, pixel_sigma=0.0)  # k = 31: noise-free inliers
    c = O.cfg(O.PHANTOM, 0, 3.0, 1)
    r = O.ransac(c, rec, 0.99, seed=5)
    assert len(r["params"]) == 41
    assert synth.phantom_check(r["params"], truth)
    assert not r["consensus"][~lab].any() and r["consensus"][lab].mean() > 0.9


# ---- round 2 additions -----------------------------------------------------------------------------------
@pytest.mark.parametrize("dim", [2, 4, 5, 8])
def test_plane_nd_minimal_solve_is_the_svd_null_vector(dim):
    """PlaneParametersEstimator.hxx:70-104: normal = null vector of [p_i, -1] (LAPACK SVD as the independent
    reference), point = first datum; a repeated datum is rank deficient -> empty."""
    g = np.random.default_rng(dim)
    oc = O.cfg(O.PLANE, dim, 0.5)
    for _ in range(20):
        p = g.uniform(-1000, 1000, (dim, dim))
        par = O.estimate(oc, p)
        assert len(par) == 2 * dim
        v = np.linalg.svd(np.hstack([p, -np.ones((dim, 1))]))[2][-1]
        n = v[:dim] / np.linalg.norm(v[:dim])
        assert abs(abs(n @ par[:dim]) - 1) < 1e-12
        assert np.array_equal(par[dim:], p[0])
        assert all(O.agree(oc, par, q) for q in p)
    p[1] = p[0]
    assert len(O.estimate(oc, p)) == 0


@pytest.mark.parametrize("dim", [4, 6])
def test_sphere_nd_minimal_solve(dim):
    """SphereParametersEstimator.hxx:169-202: the d+1 points lie on the estimated hypersphere."""
    g = np.random.default_rng(10 + dim)
    oc = O.cfg(O.SPHERE, dim, 0.5)
    c = g.uniform(-1000, 1000, dim)
    r = 321.5
    u = g.normal(size=(dim + 1, dim))
    p = c + r * u / np.linalg.norm(u, axis=1)[:, None]
    par = O.estimate(oc, p)
    assert np.allclose(par, np.concatenate([c, [r]]), rtol=1e-9, atol=1e-7)
    p[2] = p[1]
    assert len(O.estimate(oc, p)) == 0


def test_absor_weighted_ls_against_numpy():
    """AbsoluteOrientationParametersEstimator.cxx:208-291 against an independent weighted Kabsch (SVD) solution."""
    from lsqrrecipes_amd import synth
    pairs, truth, lab = synth.absolute_orientation(60, 0.0, seed=3, sigma=0.3)
    g = np.random.default_rng(8)
    w = g.uniform(0.1, 2.0, len(pairs))
    got = O.absor_weighted_ls(pairs, w)
    assert len(got) == 7
    a, b = pairs[:, :3], pairs[:, 3:]
    ma, mb = (w[:, None] * a).sum(0) / w.sum(), (w[:, None] * b).sum(0) / w.sum()
    Hm = ((a - ma) * w[:, None]).T @ (b - mb)
    U, _, Vt = np.linalg.svd(Hm)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    Rg = synth.quat_to_matrix(got[:4])
    assert np.allclose(Rg, R, atol=1e-9)
    assert np.allclose(got[4:], mb - R @ ma, atol=1e-8)
    assert np.allclose(O.absor_weighted_ls(pairs, np.ones(len(pairs))), O.ls(O.cfg(O.ABSOR, 3, 0.5), pairs), atol=1e-12)
