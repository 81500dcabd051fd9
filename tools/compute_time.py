# warm RANSAC<T,S>::compute() (lsqr_ransac, p = 0.999) on resident data at several outlier shares:
#   python3 tools/compute_time.py plane|sphere|line
import sys, time
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1] if len(sys.argv) > 1 else 'plane'
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
for outl in (0.5, 0.7, 0.8, 0.9):
    data = gen(10_000_000, outl)[0]
    ctx = Context(0)
    ctx.set_model(model, 3, 0.5, L.LS_ANALYTIC).upload(data)
    ctx.set_option('max_iterations', 1_000_000)
    ctx.ransac(0.999, seed=1, want_consensus=False)
    t = []
    for s in range(3):
        t0 = time.perf_counter()
        r = ctx.ransac(0.999, seed=10 + s, want_consensus=False)
        t.append(time.perf_counter() - t0)
    print(wl, 'outliers %.0f %%' % (outl * 100), 'iterations', int(r['info'].iterations), 'evaluated', int(r['info'].evaluated),
          'ms', ['%.2f' % (x * 1e3) for x in t], 'fraction %.3f' % r['fraction'], flush=True)
