# dense minimal solves (1024 x 64 x 64) and the step around them: one wave per system against one workgroup
import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
rows = synth.dense(2_000_000, 64, 0.05)[0]
ctx = Context(0)
ctx.set_model(L.DENSE, 64, 0.1).upload(rows)
for wave in (1, 0, 1, 0):
    ctx.set_option("dense_wave_solve", wave)
    for s in range(2):
        ctx.batch_fit(0xC0FFEE, s * 1024, 1024)
    ctx.profile(True)
    for s in range(6):
        ctx.batch_fit(0xC0FFEE, (2 + s) * 1024, 1024)
    est = ctx.profile_get("estimate"); sol = ctx.profile_get("solve"); scan = ctx.profile_get("scan")
    ctx.profile(False)
    print("dense_wave_solve", wave, "estimate %.3f ms" % (est[1] / est[0]), "solve %.3f ms per step" % (sol[1] / 6), "scan %.3f ms" % (scan[1] / scan[0]), flush=True)
