# k-d levels above the 8192-record runs of the index (cells.h: k_seg_extent / k_seg_keys + one radix sort per level,
# option "scan_kd_levels"): surviving (hypothesis, cell) pairs, scan time at both rates, index build time, and the
# votes of a whole batch against level 0 (the order of the observations must not change a single vote).
#   python3 tools/ab_kd_levels.py plane|sphere|line [points] [levels ...]
import sys, time
sys.path.insert(0, '.')
import numpy as np
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

wl = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
levels = [int(x) for x in sys.argv[3:]] or [0, 3, 5, 7]
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
data = gen(N, 0.5)[0]
H = 4096
ref = None
for lv in levels:
    ctx = Context(0)
    ctx.set_model(model, 3, 0.5, L.LS_ALGEBRAIC)
    ctx.set_option('scan_kd_levels', lv)
    ctx.upload(data)
    ctx.set_option('scan_index', 2)
    ctx.profile(True)
    ctx.batch_fit(0xC0FFEE, 0, H)
    ctx.upload(data)                       # second build in the process: code objects loaded
    ctx.batch_fit(0xC0FFEE, 0, H)
    nb, msb = ctx.profile_get('index')
    ctx.profile(False)
    ctx.set_option('scan_bound', 0)
    ctx.hypotheses_sample(0xBEEF, 0, H)
    ctx.scan()
    votes = ctx.hypotheses(params=False)[2].copy()
    if ref is None:
        ref = votes
    same = bool(np.array_equal(votes, ref))
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
        w = ctx.scan_workload()
        res[bound] = (ms / n, w['pairs'])
    print("%s levels %2d: index builds %d, %.3f ms (two builds); full count %.4f ms, %d pairs; early exit %.4f ms; votes equal level %d's: %s"
          % (wl, lv, nb, msb, res[0][0], res[0][1], res[1][0], levels[0], same), flush=True)
    ctx.close()
