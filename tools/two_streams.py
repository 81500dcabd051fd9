# throughput of bench-shaped steps with the batches alternating between TWO contexts (two streams) on one device,
# against one context:   python3 tools/two_streams.py plane|sphere|line [nctx]
import sys, time
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1]
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, H = 10_000_000, 4096
data = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl](N, 0.5)[0]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
ctxs = []
for k in range(nctx):
    c = Context(0)
    c.set_model(model, 3, 0.5, L.LS_ANALYTIC).upload(data)
    c.set_option('scan_index', 2)
    ctxs.append(c)

def run(first, count):
    q = []
    for i in range(count):
        c = ctxs[i % nctx]
        slot = (i // nctx) & 1
        if len(q) >= 2 * nctx:
            cc, ss = q.pop(0)
            cc.batch_fit_wait(ss)
        c.batch_fit_enqueue(0xC0FFEE, (first + i) * H, H, slot=slot)
        q.append((c, slot))
    for cc, ss in q:
        cc.batch_fit_wait(ss)

for rep in range(3):
    run(0, 8)
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    K = 60
    run(8, K)
    for c in ctxs:
        c.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(wl, 'contexts', nctx, 'ms/step %.4f' % (dt * 1e3), 'hyp/s %.3f M' % (H / dt / 1e6), flush=True)
