# full count through k_scan_cells (dynamic tiles) against k_scan_pairs (counted, statically balanced) on the refined index
import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
H = 4096
for wl in sys.argv[1:] or ['sphere', 'line', 'plane']:
    gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
    model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
    data = gen(10_000_000, 0.5)[0]
    ctx = Context(0)
    ctx.set_model(model, 3, 0.5, L.LS_ALGEBRAIC).upload(data)
    ctx.set_option('scan_index', 2)
    ctx.set_option('scan_bound', 0)
    for pairs in (2, 1, 2, 1):
        ctx.set_option('scan_pairs', pairs)
        for s in range(3):
            ctx.batch_fit(0xC0FFEE, s * H, H)
        ctx.profile(True)
        for s in range(10):
            r = ctx.batch_fit(0xC0FFEE, (3 + s) * H, H)
        n, ms = ctx.profile_get('scan')
        ctx.profile(False)
        print(wl, 'scan_pairs', pairs, '(1 = k_scan_pairs, 2 = k_scan_cells): scan %.3f ms' % (ms / n), 'votes', r['info'].best_votes, flush=True)
    ctx.close()
