"""CPU checks of the C-ABI boundary: the library loads without a GPU, exports every symbol
include/lsqr_hip.h declares, the ctypes table covers them all, and the host-only entry points
(model description, replay of RANSAC.hxx's adaptive loop) behave like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L
from lsqrrecipes_amd import context as ctx_mod
from lsqrrecipes_amd import synth
from oracle import pyoracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "lsqr_hip.h")).read()
    return sorted(set(re.findall(r"LSQR_API[^;(]*?\b(lsqr_\w+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = _declared()
    assert len(names) >= 35
    lib = L.load()
    for n in names:
        assert hasattr(lib, n), "liblsqr_hip.so does not export %s" % n
        assert n in L.SIGNATURES, "ctypes table misses %s" % n
    assert sorted(L.SIGNATURES) == names


def test_version_and_status_strings():
    lib = L.load()
    assert b"gfx950" in lib.lsqr_version()
    assert b"SUBSET" not in lib.lsqr_version(), "liblsqr_hip.so is a development subset build (make DEV=1)"
    assert lib.lsqr_status_string(L.EMPTY) != lib.lsqr_status_string(L.OK)


def test_no_device_fails_loudly():
    """No CPU fallback: on a box without a HIP device context creation must raise."""
    if L.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(L.LsqrError) as e:
        ctx_mod.Context(0)
    assert e.value.status == L.ERR_NO_DEVICE


@pytest.mark.parametrize("model,dim,k,P,nd", [
    (L.PLANE, 3, 3, 6, 3), (L.SPHERE, 3, 4, 4, 3), (L.SPHERE, 2, 3, 3, 2), (L.LINE, 3, 2, 6, 3),
    (L.DENSE, 64, 64, 64, 65), (L.US_SINGLE, 0, 4, 20, 15), (L.US_POINTER, 0, 3, 17, 18)])
def test_model_description_matches_reference(model, dim, k, P, nd):
    lib = L.load()
    cfg = L.ModelCfg(model, dim, 0.5, 1, 0)
    assert lib.lsqr_min_subset(C.byref(cfg)) == k
    assert lib.lsqr_num_params(C.byref(cfg)) == P
    assert lib.lsqr_record_doubles(C.byref(cfg)) == nd
    oc = O.cfg(model, dim, 0.5)
    assert O.lib().orc_min_subset(oc) == k and O.lib().orc_num_params(oc) == P


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_replay_equals_serial_ransac(seed):
    """lsqr_replay over full-scan batch results == the serial loop of RANSAC.hxx (oracle restated,
    itself pinned against the reference) on the same subset stream, duplicates included."""
    c = O.cfg(O.PLANE, 3, 0.5)
    n = 600
    data = synth.plane(n, 0.45, seed=900 + seed)[0]
    subs = O.ctr_subsets(seed, 0, 400, n, 3)
    subs[7] = subs[3][::-1]         # duplicate subset in another draw order
    subs[11] = subs[2]              # plain duplicate
    valid = np.zeros(len(subs), dtype=np.uint8)
    votes = np.zeros(len(subs), dtype=np.uint32)
    for i, s in enumerate(subs):
        par = O.estimate(c, data[s])
        if len(par):
            valid[i] = 1
            votes[i] = O.scan(c, par, data)[0]
    serial = O.ransac(c, data, 0.999, sampler="list", subsets=subs)
    r = ctx_mod.replay(n, 3, 0.999, subs, valid, votes)
    assert r["done"]
    assert r["i"] == serial["iters"]
    assert r["best_index"] == serial["best_iter"]
    assert r["best_votes"] == serial["best_votes"]
    assert serial["status"][7] == 1 and serial["status"][11] == 1


def test_replay_all_inliers_stops_at_once():
    n = 50
    subs = O.ctr_subsets(3, 0, 10, n, 3)
    r = ctx_mod.replay(n, 3, 0.99, subs, np.ones(10, np.uint8), np.full(10, n, np.uint32))
    assert r["done"] and r["i"] == 1 and r["best_index"] == 0


def test_replay_first_max_wins():
    n = 1000
    subs = O.ctr_subsets(4, 0, 6, n, 3)
    votes = np.array([10, 500, 500, 499, 500, 3], dtype=np.uint32)
    r = ctx_mod.replay(n, 3, 0.5, subs, np.ones(6, np.uint8), votes)
    assert r["best_index"] == 1 and r["best_votes"] == 500


def test_tries_cap_equals_reference_choose():
    """lsqr_replay_init's initial numTries is C(N, k) as RANSAC.hxx:254-280 evaluates it (double products,
    saturation to UINT_MAX): same value as the oracle's restatement over a grid incl. the saturating cases."""
    lib = L.load()
    st = (C.c_uint64 * 6)()
    for n in (3, 4, 10, 64, 65, 100, 129, 1000, 1625, 1626, 2000, 2954, 2955, 65536, 10_000_000, 0xFFFFFFF0):
        for k in (1, 2, 3, 4, 31, 64):
            if k > n:
                continue
            assert lib.lsqr_replay_init(n, k, 0.99, st) == L.OK
            assert st[1] == O.lib().orc_choose(n, k), (n, k)
