"""Pins oracle/ransac.c (restated RANSAC.hxx) against outputs of the REFERENCE's own
RANSAC.hxx: committed vectors (tests/golden/ransac_ref_vectors.npz, produced through
oracle/_ref by tests/golden/make_golden.py) and, when oracle/_ref is present, live runs."""
import os

import numpy as np
import pytest

from oracle import pyoracle as O
from lsqrrecipes_amd import synth

CASES = ["plane", "sphere", "circle", "line", "dense", "us", "usp"]


@pytest.fixture(scope="module")
def rv(golden_dir):
    return np.load(os.path.join(golden_dir, "ransac_ref_vectors.npz"))


def _cfg(v):
    return O.cfg(int(v[0]), int(v[1]), float(v[2]), int(v[3]))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_restatement_matches_reference_vectors(rv, name, seed):
    c = _cfg(rv[name + "_cfg"])
    data = rv[name + "_data"]
    p = 0.99 if name in ("us", "usp", "dense") else 0.999
    key = "%s_s%d_" % (name, seed)
    r = O.ransac(c, data, p, sampler="ref", seed=seed)
    assert r["fraction"] == rv[key + "fraction"][0]
    assert np.array_equal(r["consensus"], rv[key + "consensus"])  # bit-exact mask
    assert np.array_equal(r["params"], rv[key + "params"])        # same code path -> same bits
    counts = rv[key + "counts"]
    k = O.lib().orc_min_subset(c)
    assert r["iters"] * k == counts[3]                             # rand() calls consumed
    not_dup = r["status"] != 1
    assert not_dup.sum() == counts[0]                              # estimate() calls
    assert np.array_equal(r["subsets"][not_dup], rv[key + "subsets"])  # draw order preserved
    # full-scan variant (no early exit at RANSAC.hxx:94) gives the same answer
    r2 = O.ransac(c, data, p, sampler="ref", seed=seed, full_scan=True)
    assert np.array_equal(r2["consensus"], r["consensus"]) and r2["iters"] == r["iters"]
    # replaying the recorded subsets through the list sampler reproduces the run
    r3 = O.ransac(c, data, p, sampler="list", subsets=r["subsets"])
    assert np.array_equal(r3["consensus"], r["consensus"])
    assert np.array_equal(r3["params"], r["params"])


def test_exhaustive_matches_reference_vectors(rv):
    c = O.cfg(O.PLANE, 3, 0.5)
    r = O.ransac_exhaustive(c, rv["exh_data"])
    assert r["fraction"] == rv["exh_fraction"][0]
    assert np.array_equal(r["consensus"], rv["exh_consensus"])
    assert np.array_equal(r["params"], rv["exh_params"])
    assert rv["exh_subsets"].shape[0] == 14 * 13 * 12 // 6  # all C(14,3), lexicographic
    assert np.array_equal(rv["exh_subsets"][0], [0, 1, 2])
    assert np.array_equal(rv["exh_subsets"][-1], [11, 12, 13])


def test_choose():
    L = O.lib()
    assert L.orc_choose(10, 3) == 120
    assert L.orc_choose(100, 3) == 161700
    assert L.orc_choose(10_000_000, 3) == 0xFFFFFFFF  # saturates (RANSAC.hxx:274-277)
    assert L.orc_choose(2000, 3) == 1331334000


def test_invalid_input_conventions():
    """RANSAC.hxx:16-19: returns 0 and leaves `parameters` untouched."""
    c = O.cfg(O.PLANE, 3, 0.5)
    data = synth.plane(50, 0.2)[0]
    for p in (0.0, 1.0, -0.5, 1.5):
        r = O.ransac(c, data, p)
        assert r["fraction"] == 0 and r["iters"] == 0
    r = O.ransac(c, data[:2], 0.99)
    assert r["fraction"] == 0


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("seed", [101, 202, 303, 404])
def test_live_against_reference(seed):
    for c, data, p in (
            (O.cfg(O.PLANE, 3, 0.5), synth.plane(3000, 0.5, seed=seed)[0], 0.999),
            (O.cfg(O.SPHERE, 3, 0.5, O.LS_ALGEBRAIC), synth.sphere(1500, 0.4, seed=seed)[0], 0.99),
            (O.cfg(O.LINE, 2, 0.5), synth.line(500, 0.5, seed=seed, dim=2)[0], 0.999)):
        a = O.ransac(c, data, p, sampler="ref", seed=seed)
        b = O.ref_ransac(c, data, p, seed=seed, subsets_cap=4096)
        assert a["fraction"] == b["fraction"]
        assert np.array_equal(a["consensus"], b["consensus"])
        assert np.array_equal(a["params"], b["params"])
        assert (a["status"] != 1).sum() == b["estimate_calls"]


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (no /root/reference)")
def test_live_reference_invalid_input_untouched():
    c = O.cfg(O.PLANE, 3, 0.5)
    data = synth.plane(50, 0.2)[0]
    r = O.ref_ransac(c, data, 1.0, prefill=[1.0, 2.0, 3.0])
    assert r["fraction"] == 0 and np.array_equal(r["params"], [1.0, 2.0, 3.0])
    r = O.ref_ransac(c, data[:2], 0.9, prefill=[1.0, 2.0, 3.0])
    assert r["fraction"] == 0 and np.array_equal(r["params"], [1.0, 2.0, 3.0])
    # exhaustive overload clears first (RANSAC.hxx:165)
    r = O.ref_ransac(c, data[:2], 0.9, exhaustive=True, prefill=[1.0, 2.0, 3.0])
    assert r["fraction"] == 0 and len(r["params"]) == 0


def test_counter_sampler_properties():
    n, k = 1000, 64
    seen = set()
    for h in range(200):
        s = O.ctr_subset(42, h, n, k)
        assert len(set(s.tolist())) == k and s.max() < n
        seen.add(tuple(s.tolist()))
    assert len(seen) == 200
    assert np.array_equal(O.ctr_subset(42, 7, n, k), O.ctr_subset(42, 7, n, k))
    # n == k: a permutation
    s = O.ctr_subset(1, 0, 8, 8)
    assert sorted(s.tolist()) == list(range(8))
    # uniformity (loose): first index of k=3 over n=10
    cnt = np.zeros(10)
    for h in range(20000):
        for v in O.ctr_subset(5, h, 10, 3):
            cnt[v] += 1
    assert np.all(np.abs(cnt / cnt.sum() - 0.1) < 0.01)


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_config1_as_written_restatement_matches_the_reference(golden_dir, seed):
    """BASELINE.json configs[0]: plane, 10 k points, 30 % outliers, p = 0.999 -- the reference's own RANSAC.hxx run
    (tests/golden/config1_ref_vectors.npz) against the restated loop on the same rand() stream: same draws, same
    call counts, same consensus set and parameters, bit for bit; live against oracle/_ref when it is present"""
    import hashlib
    g = np.load(os.path.join(golden_dir, "config1_ref_vectors.npz"))
    data = synth.plane(10_000, 0.3)[0]
    assert np.array_equal(np.frombuffer(hashlib.sha256(data.tobytes()).digest(), dtype=np.uint8), g["data_sha"])
    c = O.cfg(O.PLANE, 3, 0.5)
    key = "s%d_" % seed
    want_mask = np.unpackbits(g[key + "consensus_bits"])[:len(data)]
    r = O.ransac(c, data, 0.999, sampler="ref", seed=seed)
    assert r["fraction"] == g[key + "fraction"][0]
    assert np.array_equal(r["consensus"], want_mask)
    assert np.array_equal(r["params"], g[key + "params"])
    not_dup = r["status"] != 1
    assert not_dup.sum() == g[key + "counts"][0] and r["iters"] * 3 == g[key + "counts"][3]
    assert np.array_equal(r["subsets"][not_dup], g[key + "subsets"])
    if O.ref_available():
        live = O.ref_ransac(c, data, 0.999, seed=seed, subsets_cap=4096)
        assert live["fraction"] == g[key + "fraction"][0] and np.array_equal(live["consensus"], want_mask)
        assert np.array_equal(live["params"], g[key + "params"]) and np.array_equal(live["subsets"], g[key + "subsets"])
