# A/B of the key-ordered full count of the plane (cells.h: k_plane_order): python3 tools/ab_order.py [N] [H] [outliers]
import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
out = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
data = synth.plane(N, out)[0]
ctx = Context(0)
ctx.set_model(L.PLANE, 3, 0.5, 0).upload(data)
ctx.set_option('scan_index', 2)
ctx.set_option('scan_bound', 0)
res = {}
for order in (0, 1, 0, 1):
    ctx.set_option('scan_hyp_order', order)
    for s in range(3):
        ctx.batch_fit(0xC0FFEE, s * H, H)
    ctx.profile(True)
    ctx.synchronize()
    vs = []
    for s in range(10):
        r = ctx.batch_fit(0xC0FFEE, (3 + s) * H, H, want_consensus=True)
        vs.append((ctx.hypotheses(params=False)[2].copy(), r['consensus'].copy(), r['params'].copy()))
    ctx.synchronize()
    n, ms = ctx.profile_get('scan')
    ctx.profile(False)
    print('scan_hyp_order', order, 'scan ms %.3f' % (ms / max(n, 1)), flush=True)
    res.setdefault(order, vs)
ok = all(all(np.array_equal(a[i], b[i]) for i in range(3)) for a, b in zip(res[0], res[1]))
print('votes, consensus, parameters equal:', ok)
