import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rows = synth.dense(m, 64, 0.05)[0]
ctx = Context(0); ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
for H in (16, 64, 256, 1024):
    ctx.hypotheses_sample(1, 0, H)
    ts = []
    for r in range(4):
        ctx.profile(True); ctx.scan(); ctx.synchronize()
        n, ms = ctx.profile_get('scan'); ctx.profile(False); ts.append(ms)
    print('dense m=%d H=%d: scan %.3f ms -> %.1f us/hyp, %.2f Tpairs/s' % (m, H, min(ts[1:]), min(ts[1:]) * 1e3 / H, m * H / (min(ts[1:]) * 1e-3) / 1e12))
