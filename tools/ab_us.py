# A/B of the US scan variants in one process: scan_filter 1 (packed fp32; scan_ppl 2 / 4 = 1 / 2 pairs
# of frames per lane), 2 (fused fp64 filter), 0 (exact fp64)
import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
data = synth.us_single_fast(N, 0.5)[0]
ctx = Context(0); ctx.set_model(L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC).upload(data)
ctx.hypotheses_sample(1, 0, H)
variants = [(1, 4, 0), (1, 4, 1), (1, 4, 2), (1, 4, 4), (1, 2, 0), (2, 0, 0)]
ref = None; res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        ctx.set_option('scan_filter', v[0]); ctx.set_option('scan_ppl', v[1]); ctx.set_option('scan_hsplit', v[2])
        ctx.profile(True); ctx.scan(); ctx.synchronize()
        n, ms = ctx.profile_get('scan'); ctx.profile(False)
        _, _, votes = ctx.hypotheses(params=False)
        if ref is None: ref = votes.copy()
        assert np.array_equal(votes, ref), v
        res[v].append(ms)
for v in variants:
    a = np.array(res[v][1:]); print('us N=%d H=%d filter=%d ppl=%d ysplit=%d: median %.3f ms -> %.0f hyp/s' % (N, H, v[0], v[1], v[2], np.median(a), H / (np.median(a) * 1e-3)))
