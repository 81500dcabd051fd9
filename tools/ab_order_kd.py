# experiment: how much does the spatial ORDER behind the cells matter?  Upload the bench's plane / sphere / line data
# (a) as is (the library's Morton order), (b) pre-ordered on the host by a k-d partition (median splits along the widest
# extent, cut at multiples of the cell size) with "scan_presorted" 1, and time the scans.
#   python3 tools/ab_order_kd.py plane|sphere|line [points]
import sys, time
sys.path.insert(0, '.')
import numpy as np
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

wl = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
CP = 512 if wl == 'plane' else 256
data = gen(N, 0.5)[0]
H = 4096


def kd_order(pts, cp):
    idx = np.arange(len(pts))
    out = []
    stack = [idx]
    while stack:
        seg = stack.pop()
        if len(seg) <= cp:
            out.append(seg)
            continue
        c = pts[seg]
        ax = int(np.argmax(c.max(0) - c.min(0)))
        cells = -(-len(seg) // cp)
        m = (cells // 2) * cp
        o = np.argpartition(c[:, ax], m)
        stack.append(seg[o[m:]])
        stack.append(seg[o[:m]])
    return np.concatenate(out)


def run(tag, arr, presorted, refine=1):
    ctx = Context(0)
    ctx.set_model(model, 3, 0.5, L.LS_ALGEBRAIC)
    ctx.set_option('scan_presorted', presorted)
    ctx.set_option('scan_refine', refine)
    ctx.upload(arr)
    ctx.set_option('scan_index', 2)
    ctx.profile(True)
    ctx.batch_fit(0xC0FFEE, 0, H)
    ctx.upload(arr)                       # second build in the process: code objects loaded
    ctx.batch_fit(0xC0FFEE, 0, H)
    nb, msb = ctx.profile_get('index')
    ctx.profile(False)
    print(tag, 'index builds', nb, 'last+first ms', round(msb, 3), flush=True)
    res = {}
    for bound in (0, 1):
        ctx.set_option('scan_bound', bound)
        for s in range(3):
            r = ctx.batch_fit(0xC0FFEE, s * H, H)
        ctx.profile(True)
        for s in range(10):
            r = ctx.batch_fit(0xC0FFEE, (3 + s) * H, H)
        n, ms = ctx.profile_get('scan')
        ctx.profile(False)
        wl_ = ctx.scan_workload()
        res[bound] = (ms / n, r['info'].best_votes, wl_['pairs'], wl_['pairs_counted'])
    print(tag, {k: ('scan %.3f ms' % v[0], 'votes', v[1], 'pairs', v[2], v[3]) for k, v in res.items()}, flush=True)
    ctx.close()


run('morton (library, scan_refine 0)', data, 0, 0)
run('morton + k-d inside runs of 8192 (library default)', data, 0, 1)
if len(sys.argv) > 3:
    sys.exit(0)
t0 = time.time()
o = kd_order(data, CP)
print('host k-d order: %.1f s' % (time.time() - t0), flush=True)
run('k-d (host order)', np.ascontiguousarray(data[o]), 1)
