import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
rows, x, lab = synth.dense(2_000_000, 64, 0.05)
ctx = Context(0); ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
ctx.set_mask(lab.astype(np.uint8))
for name, v in (('full', 0), ('loads only', 1), ('mfma only', 2), ('full', 0)):
    ctx.set_option('syrk_diag', v)
    ts = []
    for r in range(4):
        ctx.profile(True); ctx.moments(np.zeros(3), use_mask=True); n, ms = ctx.profile_get('moments'); ctx.profile(False); ts.append(ms)
    print('%-12s %.3f ms' % (name, min(ts)))
