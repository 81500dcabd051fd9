# A/B of the two-level (spatial index) scan against the exhaustive fp32-filter scan, one process
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1] if len(sys.argv) > 1 else 'plane'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
data = gen(N, 0.5)[0]
ctx = Context(0); ctx.set_model(model, 3, 0.5).upload(data)
ctx.hypotheses_sample(1, 0, H)
variants = [(2, 0, 1, 0), (2, 128, 1, 256), (2, 256, 1, 256), (2, 512, 1, 256), (2, 256, 1, 257), (2, 512, 1, 257), (2, 512, 1, 1024)]
ref = None
res = {v: [] for v in variants}
ctx.profile(True)
ctx.set_option('scan_index', 2); ctx.scan(); ctx.synchronize()
print('index build: %.3f ms' % ctx.profile_get('index')[1])
for rnd in range(6):
    for v in variants:
        ctx.set_option('scan_index', v[0]); ctx.set_option('scan_cell', v[1]); ctx.set_option('scan_cpt', v[2]); ctx.set_option('scan_block', v[3])
        ctx.profile(True); ctx.scan(); ctx.synchronize()
        n, ms = ctx.profile_get('scan'); ctx.profile(False)
        _, _, votes = ctx.hypotheses(params=False)
        if ref is None: ref = votes.copy()
        assert np.array_equal(votes, ref), 'votes differ for variant %s' % (v,)
        res[v].append(ms)
for v in variants:
    a = np.array(res[v][1:])
    print('%s N=%d H=%d index=%d cell=%d cpt=%d block=%d: median %.3f ms min %.3f ms -> %.0f hyp/s' % (wl, N, H, v[0], v[1], v[2], v[3], np.median(a), a.min(), H / (np.median(a) * 1e-3)))
