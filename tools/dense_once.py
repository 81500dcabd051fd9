# dense scan launches for timing / rocprof:  python3 tools/dense_once.py [f32=1|0] [H]   (1: fp32 matrix cores, 0: fp64)
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
f32 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rows = synth.dense(2_000_000, 64, 0.05)[0]
ctx = Context(0); ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
ctx.set_option('dense_f32', f32)
ctx.hypotheses_sample(0xC0FFEE, 0, H)
ctx.scan(); ctx.synchronize()
ctx.profile(True)
t0 = time.perf_counter()
for _ in range(3):
    ctx.scan()
ctx.synchronize()
print("f32", f32, "H", H, "wall ms/scan", (time.perf_counter() - t0) / 3 * 1e3, "event", ctx.profile_get("scan"),
      "note:", ctx._lib.lsqr_last_error(ctx._h).decode())
v = ctx.hypotheses(params=False)[2]
print("votes max", v.max(), "sum", int(v.sum()))
