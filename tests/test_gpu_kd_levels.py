"""The k-d levels above the 8192-record runs of the spatial index (csrc/cells.h: k_seg_extent / k_seg_keys + one radix
sort per level; options "scan_kd_levels" / "scan_kd_after", r05).  Only the ORDER of the observations inside the sorted
copy changes: not a single vote may (the reference counts every observation, whatever the order:
PlaneParametersEstimator.hxx:196-203, SphereParametersEstimator.hxx:255-264, LineParametersEstimator.hxx:135-150)."""
import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

pytestmark = pytest.mark.gpu

GEN = {"plane": (synth.plane, L.PLANE), "sphere": (synth.sphere, L.SPHERE), "line": (synth.line, L.LINE)}


def _votes(ctx, model, data, H, levels, index=2):
    ctx.set_model(model, data.shape[1], 0.5, L.LS_ALGEBRAIC)
    ctx.set_option("scan_index", index)
    ctx.set_option("scan_kd_levels", levels)
    ctx.set_option("scan_kd_after", 0)
    ctx.set_option("scan_bound", 0)
    ctx.upload(data)
    ctx.hypotheses_sample(4242, 0, H)
    ctx.scan()
    par, valid, votes = ctx.hypotheses()
    info = ctx.index_info()
    return valid.copy(), votes.copy(), info


@pytest.mark.parametrize("kind", ["plane", "sphere", "line"])
@pytest.mark.parametrize("n", [20_011, 70_001, 300_007, 1_200_003])
def test_votes_do_not_depend_on_the_levels(kind, n):
    """ragged sizes (one run and a bit, a last segment shorter than half a segment, more levels than the upload has
    runs); levels 0 / 3 / 7 / 12 and the exhaustive kernel (no index).  (An upload with non-finite records has no
    index at all: ensure_absmax switches the filters off.)"""
    gen, model = GEN[kind]
    data = gen(n, 0.4, seed=n)[0]
    H = 1500
    with Context(0) as ctx:
        v0, c0, i0 = _votes(ctx, model, data, H, 0)
        assert i0["built"]
        for lv in (3, 7, 12):
            v, c, info = _votes(ctx, model, data, H, lv)
            assert info["built"] and info["observations"] == i0["observations"]
            assert np.array_equal(v, v0) and np.array_equal(c, c0), (kind, n, lv)
        v, c, info = _votes(ctx, model, data, H, 0, index=0)
        assert not info["built"]
        assert np.array_equal(v, v0) and np.array_equal(c, c0)


def test_two_dimensional_models_too():
    data = synth.line(150_001, 0.3, seed=5, dim=2)[0]
    with Context(0) as ctx:
        v0, c0, _ = _votes(ctx, L.LINE, data, 1024, 0)
        v7, c7, _ = _votes(ctx, L.LINE, data, 1024, 7)
        assert np.array_equal(v0, v7) and np.array_equal(c0, c7)


def test_the_levels_come_with_the_fourth_batch_and_leave_fewer_pairs():
    """default options: the first index of an upload keeps the Morton order above the runs; once the upload has been asked
    to scan 16384 hypotheses the index is built again with the k-d levels -- fewer (hypothesis, cell) pairs survive
    level 1, the batch's results do not change"""
    data = synth.sphere(2_000_000, 0.5, seed=3)[0]
    H = 4096
    with Context(0) as ctx:
        ctx.set_model(L.SPHERE, 3, 0.5, L.LS_ALGEBRAIC).upload(data)
        ctx.profile(True)
        first = ctx.batch_fit(99, 0, H, want_consensus=True)
        assert ctx.profile_get("index")[0] == 1
        pairs_before = ctx.scan_workload()["pairs"]
        for s in range(1, 3):
            ctx.batch_fit(99, s * H, H)
        assert ctx.profile_get("index")[0] == 1          # 12288 hypotheses so far
        ctx.batch_fit(99, 3 * H, H)
        assert ctx.profile_get("index")[0] == 2          # 16384: built again
        ctx.batch_fit(99, 4 * H, H)
        assert ctx.profile_get("index")[0] == 2
        again = ctx.batch_fit(99, 0, H, want_consensus=True)
        pairs_after = ctx.scan_workload()["pairs"]
    assert pairs_after < 0.97 * pairs_before, (pairs_before, pairs_after)
    assert again["info"].best_index == first["info"].best_index and again["info"].best_votes == first["info"].best_votes
    assert np.array_equal(again["consensus"], first["consensus"])
    assert np.array_equal(again["params"], first["params"])


def test_the_planes_bounded_scan_keeps_the_order_of_a_fresh_upload():
    """the plane's axis bound is tighter on the Morton runs: bounded batches never ask for the levels, a counting batch
    does, and an index that has them keeps them"""
    data = synth.plane(2_000_000, 0.5, seed=4)[0]
    H = 4096
    with Context(0) as ctx:
        ctx.set_model(L.PLANE, 3, 0.5, L.LS_ALGEBRAIC).upload(data)
        ctx.profile(True)
        for s in range(6):
            ctx.batch_fit(7, s * H, H)                      # bounded (the default): 24576 hypotheses
        assert ctx.profile_get("index")[0] == 1
        ctx.set_option("scan_bound", 0)
        ctx.batch_fit(7, 6 * H, H)
        assert ctx.profile_get("index")[0] == 2             # a counting batch: built again with the levels
        ctx.set_option("scan_bound", 1)
        ctx.batch_fit(7, 7 * H, H)
        assert ctx.profile_get("index")[0] == 2             # ... and kept
