# host time of one lsqr_batch_fit_enqueue call (the launches of a whole batch) against the device time per batch:
#   python3 tools/enqueue_cost.py plane|sphere|line
import sys, time
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1] if len(sys.argv) > 1 else 'plane'
N, H = 10_000_000, 4096
data = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl](N, 0.5)[0]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
ctx = Context(0)
ctx.set_model(model, 3, 0.5, L.LS_ANALYTIC).upload(data)
ctx.set_option('scan_index', 2)
ring = 8
for i in range(ring):                      # prime every lane
    ctx.batch_fit_enqueue(1, i * H, H, slot=i)
for i in range(ring):
    ctx.batch_fit_wait(i)
ctx.synchronize()
for rep in range(3):
    t = []
    t0 = time.perf_counter()
    for i in range(ring):
        a = time.perf_counter()
        ctx.batch_fit_enqueue(1, (ring + i) * H, H, slot=i)
        t.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    for i in range(ring):
        ctx.batch_fit_wait(i)
    t2 = time.perf_counter()
    print(wl, 'enqueue us per batch: min %.0f mean %.0f max %.0f | 8 batches: enqueued in %.0f us, done after %.0f us' % (
        min(t) * 1e6, sum(t) / len(t) * 1e6, max(t) * 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6), flush=True)
