import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
rows, x, lab = synth.dense(2_000_000, 64, 0.05)
ctx = Context(0); ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
ctx.set_mask(lab.astype(np.uint8))
ctx.ls_fit(use_mask=True); ctx.ls_fit(use_mask=True); ctx.ls_fit(use_mask=False)
