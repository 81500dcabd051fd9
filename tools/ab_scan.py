# A/B of scan variants in ONE process (guide rule 24): interleaved rounds, median/min per variant
import sys, time, json, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1] if len(sys.argv) > 1 else 'plane'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
variants = [tuple(map(int, v.split(':'))) for v in (sys.argv[4].split(',') if len(sys.argv) > 4 else ['4:0','8:0','2:0'])]
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
data = gen(N, 0.5)[0]
ctx = Context(0); ctx.set_model(model, 3, 0.5).upload(data)
ctx.hypotheses_sample(1, 0, H)
ref = None
res = {v: [] for v in variants}
for rnd in range(7):
    for v in variants:
        ctx.set_option('scan_ppl', v[0]); ctx.set_option('scan_filter', v[1])
        ctx.profile(True); ctx.scan(); ctx.synchronize()
        n, ms = ctx.profile_get('scan'); ctx.profile(False)
        _, _, votes = ctx.hypotheses(params=False)
        if ref is None: ref = votes.copy()
        assert np.array_equal(votes, ref), 'votes differ for variant %s' % (v,)
        res[v].append(ms)
for v in variants:
    a = np.array(res[v][1:])
    print('%s N=%d H=%d ppl=%d filter=%d: median %.3f ms min %.3f ms -> %.0f hyp/s' % (wl, N, H, v[0], v[1], np.median(a), a.min(), H / (np.median(a) * 1e-3)))
