# full-count / bounded scan time per cell size with the refined index:  python3 tools/ab_cell_size.py plane|sphere|line
# (r05: with the k-d levels above the runs from the first build, as the bench's timed steps have them)
import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1]
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
data = gen(10_000_000, 0.5)[0]
H = 4096
for cell in (256, 512):
    for refine in (0, 1):
        ctx = Context(0)
        ctx.set_model(model, 3, 0.5, L.LS_ALGEBRAIC)
        ctx.set_option('scan_refine', refine)
        ctx.set_option('scan_cell', cell)
        ctx.set_option('scan_kd_after', 0)
        ctx.upload(data)
        ctx.set_option('scan_index', 2)
        out = []
        for bound in (0, 1):
            ctx.set_option('scan_bound', bound)
            for s in range(3):
                ctx.batch_fit(0xC0FFEE, s * H, H)
            ctx.profile(True)
            for s in range(10):
                r = ctx.batch_fit(0xC0FFEE, (3 + s) * H, H)
            n, ms = ctx.profile_get('scan')
            ctx.profile(False)
            w = ctx.scan_workload()
            out.append('bound %d: scan %.3f ms, pairs %d / counted %d' % (bound, ms / n, w['pairs'], w['pairs_counted']))
        print(wl, 'cell', cell, 'refine', refine, '|', ' | '.join(out), '| votes', r['info'].best_votes, flush=True)
        ctx.close()
