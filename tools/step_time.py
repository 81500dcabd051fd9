# wall-clock time of bench-shaped steps (lsqr_batch_fit, one batch per step) under a set of context options:
#   python3 tools/step_time.py plane|sphere|line [opt=value ...]      e.g.  scan_hsplit=2
import sys, time
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1]
opts = [a.split('=') for a in sys.argv[2:]]
N, H = 10_000_000, 4096
data = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl](N, 0.5)[0]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
ctx = Context(0)
ctx.set_model(model, 3, 0.5, L.LS_ANALYTIC).upload(data)
ctx.set_option('scan_index', 2)
for k, v in opts:
    ctx.set_option(k, int(v))
for rep in range(3):
    for s in range(3):
        ctx.batch_fit(0xC0FFEE, s * H, H)
    ctx.synchronize()
    t0 = time.perf_counter()
    K = 30
    for s in range(K):
        ctx.batch_fit(0xC0FFEE, (3 + s) * H, H)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(wl, dict(opts), 'ms/step %.4f' % (dt * 1e3), 'hyp/s %.3f M' % (H / dt / 1e6), flush=True)
