"""Generates the committed golden fixtures.  Run in the BUILD container only (needs
/root/reference for the data files and for oracle/_ref, and scipy for the MINPACK cross-check):

    python tests/golden/make_golden.py

Outputs (all data, no reference source text):
  ref_data/*.txt            data files the reference's own tests/examples hold
                            (testing/Data/augmentedMatrix.txt, crossWirePhantom*.txt,
                             examples/Data/augmentedMatrixWithOutliers.txt)
  ransac_ref_vectors.npz    inputs + outputs of the REFERENCE RANSAC.hxx (oracle/_ref) on
                            seeded data and seeded rand() streams
  config1_ref_vectors.npz   BASELINE configs[0] as written (plane, 10 k points, 30 % outliers, p = 0.999) run by the
                            REFERENCE RANSAC.hxx: subsets drawn, fraction, consensus set, parameters, call counts
  numerics_vectors.npz      NumPy/SciPy results (eigh, svd, lstsq, MINPACK lmder through
                            scipy.optimize.leastsq) on seeded inputs
  us_lm_vectors.npz         SciPy MINPACK on the US calibration at the reference's 1e-15 tolerances, 1 k / 20 k /
                            100 k frames: stopping code, evaluation count, minimiser (BASELINE config 5)
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as O  # noqa: E402
from lsqrrecipes_amd import synth  # noqa: E402

REF = "/root/reference"


def copy_data():
    for src in ("testing/Data/augmentedMatrix.txt", "testing/Data/crossWirePhantom2DPoints.txt",
                "testing/Data/crossWirePhantomTransformations.txt",
                "examples/Data/augmentedMatrixWithOutliers.txt"):
        shutil.copy(os.path.join(REF, src), os.path.join(HERE, "ref_data", os.path.basename(src)))


def ransac_vectors():
    out = {}
    cases = {
        "plane": (O.cfg(O.PLANE, 3, 0.5), synth.plane(400, 0.3)[0]),
        "sphere": (O.cfg(O.SPHERE, 3, 0.5, O.LS_GEOMETRIC), synth.sphere(400, 0.3)[0]),
        "circle": (O.cfg(O.SPHERE, 2, 0.5, O.LS_ALGEBRAIC), synth.sphere(300, 0.25, dim=2)[0]),
        "line": (O.cfg(O.LINE, 3, 0.5), synth.line(300, 0.3)[0]),
        "dense": (O.cfg(O.DENSE, 5, 0.1), synth.dense(200, 5, 0.1)[0]),
        "us": (O.cfg(O.US_SINGLE, 0, 3.0, 1), synth.us_single(60, 0.2)[0]),
        "usp": (O.cfg(O.US_POINTER, 0, 3.0, 1), synth.us_pointer(60, 0.2)[0]),
    }
    for name, (c, data) in cases.items():
        out[name + "_cfg"] = np.array([c.model, c.dim, c.delta, c.ls_type], dtype=np.float64)
        out[name + "_data"] = data
        for seed in (11, 12, 13):
            r = O.ref_ransac(c, data, 0.99 if name in ("us", "usp", "dense") else 0.999,
                             seed=seed, subsets_cap=1024)
            key = "%s_s%d_" % (name, seed)
            out[key + "fraction"] = np.array([r["fraction"]])
            out[key + "params"] = r["params"]
            out[key + "consensus"] = r["consensus"]
            out[key + "subsets"] = r["subsets"]
            out[key + "counts"] = np.array([r["estimate_calls"], r["agree_calls"], r["ls_calls"],
                                            r["rand_calls"]], dtype=np.int64)
    # exhaustive overload on a tiny set
    c = O.cfg(O.PLANE, 3, 0.5)
    data = synth.plane(14, 0.3, seed=77)[0]
    r = O.ref_ransac(c, data, 0.0, exhaustive=True, subsets_cap=512)
    out["exh_data"] = data
    out["exh_fraction"] = np.array([r["fraction"]])
    out["exh_params"] = r["params"]
    out["exh_consensus"] = r["consensus"]
    out["exh_subsets"] = r["subsets"]
    np.savez_compressed(os.path.join(HERE, "ransac_ref_vectors.npz"), **out)


def config1_vectors():
    """BASELINE.json configs[0] as written: PlaneParametersEstimator + RANSAC, 10 k points, 30 % outliers, p = 0.999,
    run by the REFERENCE's RANSAC.hxx (oracle/_ref).  The records are synth.plane(10_000, 0.3) (seeded: not stored);
    stored: the reference run's subsets (non-duplicate draws in order), fraction, consensus set (packed bits),
    parameters and call counts, for three rand() seeds."""
    c = O.cfg(O.PLANE, 3, 0.5)
    data = synth.plane(10_000, 0.3)[0]
    out = {"data_sha": np.frombuffer(__import__("hashlib").sha256(data.tobytes()).digest(), dtype=np.uint8)}
    for seed in (21, 22, 23):
        r = O.ref_ransac(c, data, 0.999, seed=seed, subsets_cap=4096)
        key = "s%d_" % seed
        out[key + "fraction"] = np.array([r["fraction"]])
        out[key + "params"] = r["params"]
        out[key + "consensus_bits"] = np.packbits(r["consensus"])
        out[key + "subsets"] = r["subsets"]
        out[key + "counts"] = np.array([r["estimate_calls"], r["agree_calls"], r["ls_calls"], r["rand_calls"]],
                                       dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "config1_ref_vectors.npz"), **out)


def numerics_vectors():
    from scipy.optimize import leastsq
    g = np.random.default_rng(1234)
    out = {}
    for n in (3, 4, 12, 64):
        A = g.normal(size=(n, n))
        A = A + A.T
        w, V = np.linalg.eigh(A)
        out["eig%d_A" % n], out["eig%d_w" % n], out["eig%d_V" % n] = A, w, V
    for (m, n) in ((12, 12), (40, 4), (64, 64), (300, 12)):
        A = g.normal(size=(m, n))
        b = g.normal(size=m)
        out["svd%dx%d_A" % (m, n)] = A
        out["svd%dx%d_b" % (m, n)] = b
        out["svd%dx%d_s" % (m, n)] = np.linalg.svd(A, compute_uv=False)
        out["svd%dx%d_x" % (m, n)] = np.linalg.lstsq(A, b, rcond=None)[0]
    # sphere LM through MINPACK lmder (scipy) -- tolerances of SphereParametersEstimator.hxx:323-329
    for dim, seed in ((3, 5), (2, 6)):
        pts = synth.sphere(200, 0.0, seed=seed, dim=dim)[0]
        init = O.sphere_algebraic(dim, pts)

        def f(x, pts=pts, dim=dim):
            return np.sqrt(((pts - x[:dim]) ** 2).sum(1)) - x[dim]

        def J(x, pts=pts, dim=dim):
            d = np.sqrt(((pts - x[:dim]) ** 2).sum(1))
            return np.column_stack([(x[:dim] - pts) / d[:, None], -np.ones(len(pts))])
        r = leastsq(f, init, Dfun=J, ftol=1e-10, xtol=1e-15, gtol=1e-15, maxfev=500,
                    full_output=True)
        out["lm_sphere%d_pts" % dim] = pts
        out["lm_sphere%d_init" % dim] = init
        out["lm_sphere%d_x" % dim] = r[0]
        out["lm_sphere%d_nfev_ier" % dim] = np.array([r[2]["nfev"], r[4]])
    np.savez_compressed(os.path.join(HERE, "numerics_vectors.npz"), **out)


def us_lm_vectors(sizes=(1000, 20000, 100000)):
    """BASELINE config 5 (SingleUnknownPointTarget, Levenberg-Marquardt with the reference's settings,
    SinglePointTarget...Estimator.cxx:287-295: f/x/g tolerances 1e-15, 5000 evaluations): what MINPACK itself
    (scipy.optimize.leastsq = lmder) does on 1 k / 20 k / 100 k noisy frames, started from the analytic estimate --
    the stopping code, the number of function evaluations and the final iterate.  f and J are the restated
    formulae of the oracle (oracle/estimators.c us_lm_fcn, .cxx:415-658).  The frames are NOT stored: they are
    regenerated from (generator, size, seed), which are."""
    from scipy.optimize import leastsq
    out = {}
    for m in sizes:
        seed = 4100 + m
        rec = synth.us_single_fast(m, 0.0, seed=seed)[0]
        init = O.us_analytic(O.US_SINGLE, rec)[:11]
        F = O.UsFunction(O.US_SINGLE, rec)
        r = leastsq(F.f, init, Dfun=F.jac, ftol=1e-15, xtol=1e-15, gtol=1e-15, maxfev=5000, full_output=True)
        w, info, nfev = O.us_iterative(O.US_SINGLE, rec, init)
        key = "us_lm_%d_" % m
        out[key + "seed"] = np.array([seed])
        out[key + "init"] = init
        out[key + "scipy_x"] = r[0]
        out[key + "scipy_nfev_ier"] = np.array([r[2]["nfev"], r[4]])
        out[key + "scipy_cost"] = np.array([(r[2]["fvec"] ** 2).sum()])
        out[key + "oracle_x"] = w[:11]
        out[key + "oracle_nfev_info"] = np.array([nfev, info])
        print(m, "scipy", r[2]["nfev"], r[4], "oracle", nfev, info, np.abs(w[:11] - r[0]).max(), flush=True)
    np.savez_compressed(os.path.join(HERE, "us_lm_vectors.npz"), **out)


def us_lm_flag_table():
    """How reproducible is MINPACK's stopping code at the reference's 1e-15 tolerances?  SciPy's lmder and the
    oracle's literal restatement of lmder (oracle/linalg.c), fed the SAME f and J, on 6 seeds x 3 sizes: the
    evaluation counts differ by tens to hundreds and the success flag itself flips on some inputs -- the
    termination happens inside rounding noise (trust-region radius random-walks until delta <= xtol |x|)."""
    from scipy.optimize import leastsq
    rows = []
    for m in (200, 1000, 3000):
        for seed in range(900, 906):
            rec = synth.us_single_fast(m, 0.0, seed=seed)[0]
            init = O.us_analytic(O.US_SINGLE, rec)[:11]
            F = O.UsFunction(O.US_SINGLE, rec)
            r = leastsq(F.f, init, Dfun=F.jac, ftol=1e-15, xtol=1e-15, gtol=1e-15, maxfev=5000, full_output=True)
            w, info, nfev = O.us_iterative(O.US_SINGLE, rec, init)
            rows.append([m, seed, r[2]["nfev"], r[4], nfev, info, (r[2]["fvec"] ** 2).sum()])
            print(rows[-1], flush=True)
    np.savez_compressed(os.path.join(HERE, "us_lm_flags.npz"),
                        table=np.array(rows, dtype=np.float64),
                        columns=np.array(["frames", "seed", "scipy_nfev", "scipy_ier", "oracle_nfev", "oracle_info",
                                          "scipy_cost"]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "us_lm":
        us_lm_vectors()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "config1":
        config1_vectors()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "us_lm_flags":
        us_lm_flag_table()
        sys.exit(0)
    copy_data()
    ransac_vectors()
    config1_vectors()
    numerics_vectors()
    us_lm_vectors()
    us_lm_flag_table()
    print("golden fixtures written to", HERE)
