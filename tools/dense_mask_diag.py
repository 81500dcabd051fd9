# ablation of the fused dense mask + SYRK kernel: python3 tools/dense_mask_diag.py
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
rows, x_true, _ = synth.dense(2_000_000, 64, 0.7)
ctx = Context(0)
ctx.set_model(L.DENSE, 64, 0.1, 0).upload(rows)
x = x_true * (1 + 1e-3 * np.random.default_rng(1).standard_normal(64))
for ring, diag in ((4, 0), (4, 1), (4, 5), (4, 3), (4, 7)):
    ctx.set_option("dense_mask_ring", ring)
    ctx.set_option('dense_mask_diag', diag)
    ctx.mask(x, want_mask=False)
    ctx.profile(True)
    for _ in range(5):
        m, cnt = ctx.mask(x, want_mask=False)
    n, ms = ctx.profile_get('mask')
    ctx.profile(False)
    print('ring', ring, 'diag', diag, 'mask+syrk %.3f ms' % (ms / n), 'count', cnt, flush=True)
