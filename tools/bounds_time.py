# bounds pass of a 10 M-point upload (HIP events around the kernel):  python3 tools/bounds_time.py
import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
data = synth.plane(10_000_000, 0.5)[0]
ctx = Context(0)
ctx.set_model(L.PLANE, 3, 0.5)
for rep in range(3):
    ctx.profile(True)
    for _ in range(5):
        ctx.upload(data)
        ctx.hypotheses_sample(1, 0, 64)     # asks for max |x|: runs the bounds pass
    n, ms = ctx.profile_get("absmax")
    ctx.profile(False)
    print("k_bounds: launches", n, "avg %.1f us" % (ms / n * 1e3), "= %.2f of 8 TB/s" % (240e6 / (ms / n * 1e-3) / 8e12))
