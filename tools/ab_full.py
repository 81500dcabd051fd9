# A/B of the full-count (every hypothesis counted) two-level scan arrangements, one process, interleaved rounds:
#   python3 tools/ab_full.py plane|sphere|line [N] [H]
# variants: (scan_pairs, scan_block): k_scan_cells (dynamic tiles) vs k_scan_pairs (statically balanced), readlane vs LDS
import sys, numpy as np
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
ctx.hypotheses_sample(0xC0FFEE, 0, H)
ctx.set_option('scan_index', 2); ctx.scan(); ctx.synchronize()
variants = [(0, 0), (0, 256), (0, 257), (1, 0), (1, 256), (1, 257)]
ref = None
res = {v: [] for v in variants}
for rnd in range(7):
    for v in variants:
        ctx.set_option('scan_pairs', v[0]); ctx.set_option('scan_block', v[1])
        ctx.synchronize()
        import time
        t0 = time.perf_counter(); ctx.scan(); ctx.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        _, _, votes = ctx.hypotheses(params=False)
        if ref is None: ref = votes.copy()
        assert np.array_equal(votes, ref), 'votes differ for variant %s' % (v,)
        res[v].append(ms)
for v in variants:
    a = np.array(res[v][1:])
    print('%s N=%d H=%d scan_pairs=%d block=%d: median %.3f ms min %.3f ms -> %.2f M hyp/s' % (wl, N, H, v[0], v[1], np.median(a), a.min(), H / (np.median(a) * 1e3)), flush=True)
