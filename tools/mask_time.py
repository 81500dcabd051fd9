# time of the mask + moments pass (US, 1 M frames) against the chunk size (option mom_chunk, units of 256 records)
import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
rec, truth = synth.us_single(1_000_000, 0.5, seed=5, pixel_sigma=1.0)[:2]
for chunk in (0, 2, 4, 8, 16, 32):
    for staged in (0,):
        ctx = Context(0)
        ctx.set_option('mom_chunk', chunk)
        ctx.set_model(L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC if hasattr(L, 'LS_ANALYTIC') else 0).upload(rec)
        for s in range(3):
            r = ctx.batch_fit(0xC0FFEE, s * 256, 256)
        ctx.profile(True)
        for s in range(10):
            r = ctx.batch_fit(0xC0FFEE, (3 + s) * 256, 256)
        ctx.synchronize()
        n, ms = ctx.profile_get('mask')
        print('chunk', chunk, 'mask+moments us %.1f' % (ms / max(n, 1) * 1e3), 'votes', r['info'].best_votes, flush=True)
        ctx.close()
