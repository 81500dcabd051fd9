"""lsqr_batch_fit_enqueue must return without waiting for the device (include/lsqr_hip.h): the matrix-core filters of
the dense / US / plane-phantom scans note the size of their worklist segments instead of reading the fill back, the fill
travels with the slot's record, and lsqr_batch_fit_wait runs the batch again on the exact kernels if a segment ever
overflowed (the scan had become blocking).  `scan_test_overflow` makes the wait take that path."""
import time

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["dense", "us"])   # (the plane phantom's fit runs on the host: no enqueue form)
def test_deferred_worklist_check_and_its_rerun(kind):
    if kind == "dense":
        data, model, dim, delta, ls, H = synth.dense(300_000, 64, 0.05, seed=3)[0], L.DENSE, 64, 0.1, 0, 256
    else:
        data, model, dim, delta, ls, H = synth.us_single_fast(200_000, 0.3, seed=3)[0], L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC, 1024
    with Context(0) as ctx:
        ctx.set_model(model, dim, delta, ls).upload(data)
        ctx.set_option("batch_lanes", 2)
        want = [ctx.batch_fit(7, s * H, H) for s in range(4)]
        for forced in (0, 1):
            ctx.set_option("scan_test_overflow", forced)
            for s in range(4):
                ctx.batch_fit_enqueue(7, s * H, H, slot=s)
            got = [ctx.batch_fit_wait(s) for s in range(4)]
            for s in range(4):
                assert got[s]["status"] == want[s]["status"], (kind, forced, s)
                gi, wi = got[s]["info"], want[s]["info"]
                assert (gi.best_votes, gi.best_index, gi.fit.n_used) == (wi.best_votes, wi.best_index, wi.fit.n_used)
                assert np.allclose(got[s]["params"], want[s]["params"], rtol=1e-9, atol=1e-12), (kind, forced, s)
        ctx.set_option("scan_test_overflow", 0)


def test_enqueue_returns_before_the_scan_has_run():
    """four dense batches enqueued back to back: the host is done long before the device (each scan is ~1 ms of
    device time; a blocking enqueue would take at least that long per call)"""
    data = synth.dense(1_000_000, 64, 0.05, seed=4)[0]
    with Context(0) as ctx:
        ctx.set_model(L.DENSE, 64, 0.1).upload(data)
        ctx.set_option("batch_lanes", 1)
        for s in range(2):                       # warm: index-free model, code objects, fragments of the upload
            ctx.batch_fit(9, s * 1024, 1024)
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.batch_fit_enqueue(9, 0, 1024, slot=0)
        ctx.batch_fit_enqueue(9, 1024, 1024, slot=1)
        t_enq = time.perf_counter() - t0
        ctx.batch_fit_wait(0)
        ctx.batch_fit_wait(1)
        t_all = time.perf_counter() - t0
        assert t_enq < 0.6 * t_all, (t_enq, t_all)
