# Differential soak of the scan arrangements on random uploads: python3 tools/soak.py [seconds] [seed]
# For every drawn configuration (model, dimension, size, outlier fraction, threshold, coordinate offset and scale,
# batch size) the votes of the exhaustive exact fp64 kernel (scan_index 0, scan_filter 0) are the truth; checked
# against it: the exhaustive fp32-filter kernel, the two-level scan in every arrangement (k_scan_cells / k_scan_pairs,
# cells of 256 / 512, readlane / LDS broadcast; r05: 0 / 3 / 7 / 12 k-d levels above the index's runs), and the batch entry point with and without the bounded scan (winner,
# consensus, parameters; counted hypotheses exact, the others 0 and not above the running maximum before them).
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
t_end = time.time() + budget
ctx = Context(0)
done, bad = 0, 0
while time.time() < t_end:
    model, gen = [(L.PLANE, synth.plane), (L.SPHERE, synth.sphere), (L.LINE, synth.line)][int(g.integers(3))]
    dim = int(g.choice([3, 3, 3, 2]))
    n = int(g.choice([70_000, 200_000, 700_000, 2_000_000]))
    out = float(g.choice([0.1, 0.3, 0.5, 0.8]))
    delta = float(g.choice([0.05, 0.5, 2.0]))
    box = float(g.choice([30.0, 1000.0]))
    sigma = float(g.choice([0.1, 0.4]))
    off = float(g.choice([0.0, 1.0e3, 1.0e6]))
    H = int(g.choice([512, 2048, 4096]))
    seed = int(g.integers(1 << 30))
    cfg = dict(model=model, dim=dim, n=n, out=out, delta=delta, box=box, sigma=sigma, off=off, H=H, seed=seed)
    data = gen(n, out, seed=seed, dim=dim, sigma=sigma, box=box)[0] + off
    ls = L.LS_ALGEBRAIC if model == L.SPHERE else 0
    ctx.set_model(model, dim, delta, ls).upload(data)
    kd = int(g.choice([0, 3, 7, 12]))            # k-d levels above the index's runs (r05), built with the first index
    cfg["kd_levels"] = kd
    ctx.set_option("scan_kd_levels", kd)
    ctx.set_option("scan_kd_after", 0)
    ctx.hypotheses_sample(seed, 0, H)

    def votes(index, filt, pairs=0, cell=0, block=0):
        for k, v in (("scan_index", index), ("scan_filter", filt), ("scan_pairs", pairs), ("scan_cell", cell),
                     ("scan_block", block)):
            ctx.set_option(k, v)
        ctx.scan()
        return ctx.hypotheses(params=False)[2].copy()
    truth = votes(0, 0)
    ok = True
    for name, v in (("f32 exhaustive", votes(0, 1)), ("cells", votes(2, 1, 2)), ("pairs", votes(2, 1, 1)),
                    ("cells 256", votes(2, 1, 2, 256)), ("cells 512 lds", votes(2, 1, 2, 512, 257)),
                    ("pairs readlane", votes(2, 1, 1, 0, 256))):
        if not np.array_equal(v, truth):
            d = np.flatnonzero(v != truth)
            print("MISMATCH", name, cfg, len(d), d[:4], v[d[:4]], truth[d[:4]], flush=True)
            ok = False
    for k, v in (("scan_index", 2), ("scan_filter", 1), ("scan_pairs", 0), ("scan_cell", 0), ("scan_block", 0)):
        ctx.set_option(k, v)
    res = []
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        r = ctx.batch_fit(seed, 0, H, want_consensus=True)
        _, valid, v = ctx.hypotheses(params=False)
        res.append((r, v.copy(), valid.copy()))
    ctx.set_option("scan_bound", 1)
    ctx.set_option("scan_index", 1)
    (r0, v0, ok0), (r1, v1, ok1) = res
    if not (np.array_equal(v0, truth) and np.array_equal(r0["consensus"], r1["consensus"])
            and np.array_equal(r0["params"], r1["params"], equal_nan=True)
            and r0["info"].best_index == r1["info"].best_index and r0["info"].best_votes == r1["info"].best_votes):
        print("MISMATCH bounded", cfg, flush=True)
        ok = False
    sk = v1 != v0
    runmax = np.maximum.accumulate(np.where(ok0 > 0, v0, 0))
    idx = np.flatnonzero(sk)
    if not (np.all(v1[sk] == 0) and np.all(v0[idx[idx > 0]] <= runmax[idx[idx > 0] - 1]) and (not sk[0] or v0[0] == 0)):
        print("MISMATCH skipped set", cfg, flush=True)
        ok = False
    done += 1
    bad += 0 if ok else 1
    if done % 10 == 0:
        print("checked", done, "configurations,", bad, "bad", flush=True)
print("soak: %d configurations, %d with a mismatch" % (done, bad))
sys.exit(1 if bad else 0)
