import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
rows = synth.dense(2_000_000, 64, 0.05)[0]
ctx = Context(0); ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
ctx.hypotheses_sample(1, 0, 256)
for t in (0, 1):
    ctx.set_option('dense_transposed', t)
    ctx.scan(); ctx.synchronize()
