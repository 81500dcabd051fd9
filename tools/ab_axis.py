# A/B of the vote bounds by rank over axis-sorted cells (csrc/axis.h): plane N x H, both rates; winner, winner's votes,
# consensus and parameters compared with the index without axes, the counted votes compared with the full count
#   python3 tools/ab_axis.py [N] [H] [outlier fraction]
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
out = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
data = synth.plane(N, out)[0]
res = {}
K = 10
for axis in (0, 1):
    ctx = Context(0)
    ctx.set_option('scan_axis', axis)
    ctx.set_model(L.PLANE, 3, 0.5, 0).upload(data)
    ctx.set_option('scan_index', 2)
    ctx.hypotheses_sample(0xC0FFEE, 0, H)
    ctx.profile(True)
    ctx.scan(); ctx.synchronize()
    print('axis', axis, 'index build ms', ctx.profile_get('index'))
    for bound in (0, 1):
        ctx.set_option('scan_bound', bound)
        for s in range(3):
            r = ctx.batch_fit(0xC0FFEE, s * H, H)
        ctx.profile(True)
        ctx.synchronize()
        t0 = time.perf_counter()
        steps = []
        for s in range(K):
            r = ctx.batch_fit(0xC0FFEE, (3 + s) * H, H, want_consensus=True)
            _, valid, votes = ctx.hypotheses(params=False)
            steps.append((votes.copy(), r['consensus'].copy(), r['params'].copy(), r['info'].best_index,
                          ctx.scan_workload() if bound else None))
        ctx.synchronize()
        n, ms = ctx.profile_get('scan')
        ctx.profile(False)
        res[(axis, bound)] = steps
        w = steps[-1][4]
        print('axis', axis, 'scan_bound', bound, 'scan ms %.3f' % (ms / max(n, 1)), 'votes max', steps[-1][0].max(),
              {k: w[k] for k in ('pilots', 'second_pass', 'pairs_counted')} if w else '', flush=True)
    ctx.close()
ok = True
for s in range(K):
    full = res[(0, 0)][s]
    for key in ((1, 0), (0, 1), (1, 1)):
        b = res[key][s]
        same = (np.array_equal(full[1], b[1]) and np.array_equal(full[2], b[2]) and full[3] == b[3]
                and full[0].max() == b[0].max())
        counted = b[0] != 0
        same = same and np.array_equal(b[0][counted], full[0][counted])
        if not same:
            ok = False
            print('step', s, key, 'DIFFERS: winner', full[3], b[3], 'votes', full[0].max(), b[0].max())
print('all equal' if ok else 'MISMATCH')
