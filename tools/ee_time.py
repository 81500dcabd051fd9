# step time of the dense / US workloads with and without the chunked early exit (csrc/earlyexit.h), bench shapes:
#   python3 tools/ee_time.py dense|us [outlier_fraction]
import sys, time
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1]
fr = float(sys.argv[2]) if len(sys.argv) > 2 else (0.05 if wl == 'dense' else 0.5)
if wl == 'dense':
    N, H, data, model, dim, delta = 2_000_000, 1024, synth.dense(2_000_000, 64, fr)[0], L.DENSE, 64, 0.1
else:
    N, H, data, model, dim, delta = 1_000_000, 4096, synth.us_single_fast(1_000_000, fr)[0], L.US_SINGLE, 3, 3.0
ctx = Context(0)
ctx.set_model(model, dim, delta, L.LS_ANALYTIC).upload(data)
for rep in range(2):
    for bound in (0, 1):
        ctx.set_option('scan_bound', bound)
        for s in range(2):
            ctx.batch_fit(0xC0FFEE, s * H, H)
        ctx.profile(True)
        ctx.synchronize()
        t0 = time.perf_counter()
        K = 10
        for s in range(K):
            r = ctx.batch_fit(0xC0FFEE, (2 + s) * H, H)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / K
        prof = {k: ctx.profile_get(k) for k in ('sample', 'estimate', 'scan', 'mask', 'moments', 'solve')}
        ctx.profile(False)
        w = ctx.scan_work()
        print(wl, fr, 'scan_bound', bound, 'ms/step %.3f' % (dt * 1e3), 'hyp/s %.1f k' % (H / dt / 1e3),
              {k: round(v[1] / max(v[0], 1), 4) for k, v in prof.items()}, 'votes', r['info'].best_votes,
              'evaluated %.3f' % (w['row_hypothesis_pairs'] / w['row_hypothesis_pairs_all']), w, flush=True)
