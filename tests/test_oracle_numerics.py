"""Pins the oracle's restated VNL numerics (oracle/linalg.c) against NumPy/SciPy vectors
(tests/golden/numerics_vectors.npz, made by tests/golden/make_golden.py) and against the
reference's own known answers."""
import os

import numpy as np
import pytest

from oracle import pyoracle as O


@pytest.fixture(scope="module")
def nv(golden_dir):
    return np.load(os.path.join(golden_dir, "numerics_vectors.npz"))


@pytest.mark.parametrize("n", [3, 4, 12, 64])
def test_sym_eig_matches_eigh(nv, n):
    w, V = O.sym_eig(nv["eig%d_A" % n])
    assert np.allclose(w, nv["eig%d_w" % n], rtol=0, atol=1e-12 * n)
    # ascending order, unit columns, sign arbitrary (reference tests use |dot|)
    assert np.all(np.diff(w) >= 0)
    dots = np.abs(np.sum(V * nv["eig%d_V" % n], axis=0))
    assert np.allclose(dots, 1.0, atol=1e-9)


@pytest.mark.parametrize("shape", [(12, 12), (40, 4), (64, 64), (300, 12)])
def test_svd_and_pinv_match_numpy(nv, shape):
    m, n = shape
    A, b = nv["svd%dx%d_A" % shape], nv["svd%dx%d_b" % shape]
    U, s, V = O.svd(A)
    assert np.allclose(s, nv["svd%dx%d_s" % shape], rtol=1e-12, atol=1e-13)
    assert np.allclose((U * s) @ V.T, A, atol=1e-12)
    x, rank = O.pinv_solve(A, b, 2.220446049250313e-16)
    assert rank == n
    assert np.allclose(x, nv["svd%dx%d_x" % shape], rtol=1e-9, atol=1e-10)


def test_pinv_rank_deficient():
    A = np.array([[1.0, 2.0, 3.0], [2.0, 4.0, 6.0], [1.0, 0.0, 1.0], [0.0, 2.0, 2.0]])
    x, rank = O.pinv_solve(A, np.ones(4), 1e-12)
    assert rank == 2


@pytest.mark.parametrize("dim", [3, 2])
def test_lmder_matches_minpack(nv, dim):
    """orc_lmder is MINPACK lmder restated; scipy.optimize.leastsq wraps the real one."""
    x, info, nfev = O.sphere_geometric(dim, nv["lm_sphere%d_pts" % dim], nv["lm_sphere%d_init" % dim])
    ref_nfev, ref_ier = nv["lm_sphere%d_nfev_ier" % dim]
    assert info == ref_ier
    assert nfev == ref_nfev
    assert np.allclose(x, nv["lm_sphere%d_x" % dim], rtol=1e-12, atol=1e-12)


def test_dense_known_answer(golden_dir):
    """testing/DenseLinearEquationSystemParametersEstimatorTest.cxx:162-164 (tolerance 0.5 there)."""
    M = np.loadtxt(os.path.join(golden_dir, "ref_data", "augmentedMatrix.txt"))
    known = np.array([-1.777985584409468e+001, 1.111302171667757e+000, -1.568653413096010e+002,
                      1.469013927556186e+002, -6.296891425314718e+001, -1.042139650090033e+003])
    x = O.ls(O.cfg(O.DENSE, 6, 0.5), M)
    assert np.allclose(x, known, rtol=0, atol=1e-9)


def test_gander_circle():
    """testing/SphereParametersEstimatorTest.cxx:302-324: geometric fit literature value
    (4.7398, 2.9835, 4.7142)."""
    pts = np.array([[1, 7], [2, 6], [5, 8], [7, 7], [9, 5], [3, 7]], float)
    alg = O.sphere_algebraic(2, pts)
    geo, info, _ = O.sphere_geometric(2, pts, alg)
    assert 1 <= info <= 4
    assert np.allclose(geo, [4.7398, 2.9835, 4.7142], atol=5e-5)
    c = O.cfg(O.SPHERE, 2, 0.5, O.LS_GEOMETRIC)
    assert np.array_equal(O.ls(c, pts), geo)
