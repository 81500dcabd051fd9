# winner's mask + moments pass at 10 M points (HIP events around the kernel)
import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
for wl, gen, model in (("plane", synth.plane, L.PLANE), ("sphere", synth.sphere, L.SPHERE)):
    data = gen(10_000_000, 0.5)[0]
    ctx = Context(0)
    ctx.set_model(model, 3, 0.5, L.LS_ALGEBRAIC).upload(data)
    for s in range(3):
        ctx.batch_fit(0xC0FFEE, s * 1024, 1024)
    ctx.profile(True)
    for s in range(10):
        ctx.batch_fit(0xC0FFEE, (3 + s) * 1024, 1024)
    n, ms = ctx.profile_get("mask")
    ctx.profile(False)
    print(wl, "k_mask_moments: %.1f us" % (ms / n * 1e3), "= %.2f of 8 TB/s" % (250e6 / (ms / n * 1e-3) / 8e12), flush=True)
    ctx.close()
