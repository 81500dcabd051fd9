# Differential soak of the chunked early exit (dense system, US calibrations): python3 tools/soak_ee.py [seconds] [seed]
# Random uploads; the batch entry point with scan_bound 0 (every hypothesis counted) against scan_bound 1: winner,
# consensus set and parameters identical; a hypothesis the early exit abandoned reports a partial count that does
# not exceed the running maximum before it (so that the replay of the serial loop is unaffected).
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
t_end = time.time() + budget
ctx = Context(0)
done, bad, engaged = 0, 0, 0
while time.time() < t_end:
    kind = int(g.integers(3))
    seed = int(g.integers(1 << 30))
    out = float(g.choice([0.05, 0.2, 0.5, 0.7]))
    if kind == 0:
        ncol = int(g.choice([33, 48, 64]))
        m = int(g.choice([70_000, 150_000, 400_000, 1_000_000]))
        H = int(g.choice([128, 512, 1024]))
        delta = float(g.choice([0.02, 0.1, 0.5]))
        data = synth.dense(m, ncol, out, seed=seed)[0]
        model, dim, ls = L.DENSE, ncol, 0
    else:
        m = int(g.choice([70_000, 200_000, 500_000]))
        H = int(g.choice([256, 1024, 4096]))
        delta = float(g.choice([1.0, 3.0, 10.0]))
        gen = synth.us_single if kind == 1 else synth.us_pointer
        data = gen(m, out, seed=seed, pixel_sigma=1.0)[0]
        model, dim, ls = (L.US_SINGLE if kind == 1 else L.US_POINTER), 0, L.LS_ALGEBRAIC if hasattr(L, "LS_ALGEBRAIC") else 0
    cfg = dict(kind=kind, m=m, H=H, delta=delta, out=out, seed=seed, dim=dim)
    ctx.set_model(model, dim, delta, 0 if kind == 0 else 0).upload(data)
    res = []
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        r = ctx.batch_fit(seed, 0, H, want_consensus=True)
        _, valid, v = ctx.hypotheses(params=False)
        res.append((r, v.copy(), valid.copy(), ctx.scan_work()))
    ctx.set_option("scan_bound", 1)
    (r0, v0, ok0, _), (r1, v1, ok1, wk) = res
    ok = (np.array_equal(ok0, ok1) and np.array_equal(r0["consensus"], r1["consensus"])
          and np.array_equal(r0["params"], r1["params"], equal_nan=True)
          and r0["info"].best_index == r1["info"].best_index and r0["info"].best_votes == r1["info"].best_votes)
    sk = v1 != v0
    runmax = np.maximum.accumulate(np.where(ok0 > 0, v0, 0))
    idx = np.flatnonzero(sk)
    ok = ok and np.all(v1[sk] <= v0[sk]) and np.all(v0[idx[idx > 0]] <= runmax[idx[idx > 0] - 1]) and not (len(idx) and idx[0] == 0 and v0[0] > 0)
    if not ok:
        print("MISMATCH", cfg, flush=True)
        bad += 1
    engaged += 1 if wk["early_exit"] else 0
    done += 1
    if done % 10 == 0:
        print("checked", done, "configurations (early exit engaged in", engaged, "),", bad, "bad", flush=True)
print("soak_ee: %d configurations (early exit engaged in %d), %d with a mismatch" % (done, engaged, bad))
sys.exit(1 if bad else 0)
