# latency of the adaptive RANSAC::compute() path (lsqr_ransac) on small inputs, observations resident
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
ctx = Context(0)
for n in (100, 10_000, 1_000_000):
    data = synth.plane(n, 0.3)[0]
    ctx.set_model(L.PLANE, 3, 0.5).upload(data)
    ctx.ransac(0.999, seed=1)
    t = []
    for s in range(20):
        t0 = time.perf_counter(); r = ctx.ransac(0.999, seed=2 + s); t.append(time.perf_counter() - t0)
    print("N %8d: median %.3f ms  min %.3f ms  iterations %d" % (n, 1e3 * np.median(t), 1e3 * min(t), r["info"].iterations))
