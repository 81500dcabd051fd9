# Differential soak of the fp16 matrix-core filters (dense_h16.h, us_h16.h, phantom_h16.h) on random uploads:
#   python3 tools/soak_h16.py [seconds] [seed]
# Dense: rows / right-hand sides rescaled over twelve orders of magnitude, thresholds from far below the noise to far
# above the data, ragged row and hypothesis counts; truth = the fp64 matrix-core filter + exact re-check (dense_f32 0),
# checked: the fp16 filter (2), the fp32 filter (1).  US single / pointer: truth = the exact fp64 kernel (scan_filter 0),
# checked: the fp16 filter and the packed fp32 filter; the plane phantom likewise, translations and pixel coordinates
# rescaled over six orders of magnitude.  For every configuration also the batch entry point with and
# without the early exit (winner, consensus set, parameters).
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
t_end = time.time() + budget
ctx = Context(0)
done, bad, rich = 0, 0, 0   # rich: configurations whose best hypothesis agrees with > 10 % of the observations
while time.time() < t_end:
    kinds = sys.argv[3].split(",") if len(sys.argv) > 3 else ["dense", "us", "pointer", "phantom"]
    kind = kinds[int(g.integers(len(kinds)))]
    seed = int(g.integers(1 << 30))
    H = int(g.choice([97, 333, 1024, 2048]))
    ok = True
    if kind == "dense":
        n = int(g.choice([70_013, 200_000, 500_001]))
        out = float(g.choice([0.05, 0.3, 0.7]))
        noise = float(g.choice([0.0, 0.01, 0.05]))
        rows = synth.dense(n, 64, out, seed=seed, noise=noise)[0]
        sa, sb = 10.0 ** float(g.integers(-6, 7)), 10.0 ** float(g.integers(-6, 7))
        rows[:, :64] *= sa
        rows[:, 64] *= sa * sb                      # x_true scales by sb
        delta = float(g.choice([1e-4, 0.1, 3.0, 1e3])) * sa * sb
        H = min(H, 1024)
        cfg = dict(kind=kind, n=n, out=out, noise=noise, sa=sa, sb=sb, delta=delta, H=H, seed=seed)
        ctx.set_model(L.DENSE, 64, delta).upload(rows)
        ctx.hypotheses_sample(seed, 0, H)

        def votes(f32):
            ctx.set_option("dense_f32", f32)
            ctx.scan()
            return ctx.hypotheses(params=False)[2].copy()
        truth = votes(0)
        for name, v in (("fp16", votes(2)), ("fp32", votes(1))):
            if not np.array_equal(v, truth):
                d = np.flatnonzero(v != truth)
                print("MISMATCH", name, cfg, len(d), d[:4], v[d[:4]], truth[d[:4]], flush=True)
                ok = False
        ctx.set_option("dense_f32", 2)
        ls = 0
        model = L.DENSE
        dim = 64
    else:
        n = int(g.choice([70_013, 200_000, 500_001]))
        out = float(g.choice([0.0, 0.3, 0.6]))
        sig = float(g.choice([0.5, 2.0]))
        if kind == "us":
            data = synth.us_single_fast(n, out, seed=seed, pixel_sigma=sig)
            model = L.US_SINGLE
        elif kind == "phantom":
            data = synth.plane_phantom_fast(n, min(out, 0.1), seed=seed, pixel_sigma=sig * 0.05)[0].copy()
            st, sp_ = 10.0 ** float(g.integers(-3, 4)), 10.0 ** float(g.integers(-3, 4))
            if g.random() < 0.5:                    # (rescaled frames no longer lie on a plane: every other upload as built)
                st = sp_ = 1.0
            data[:, 9:12] *= st                     # translations (and with them the plane's offset)
            data[:, 13:15] *= sp_                   # pixel coordinates (the scale factors shrink by as much)
            model = L.PHANTOM
            H = min(H, 1024)
        else:
            data = synth.us_pointer(min(n, 70_013), out, seed=seed, pixel_sigma=sig)
            model = L.US_POINTER
        data = data[0] if isinstance(data, tuple) else data
        delta = float(g.choice([0.01, 1.0, 3.0, 50.0, 1e4]))
        if kind == "phantom":
            delta *= st
        cfg = dict(kind=kind, n=len(data), out=out, sig=sig, delta=delta, H=H, seed=seed)
        ls = L.LS_ANALYTIC
        dim = 0
        ctx.set_model(model, dim, delta, ls).upload(data)
        ctx.hypotheses_sample(seed, 0, H)

        def votes(mfma, filt):
            ctx.set_option("us_mfma", mfma)
            ctx.set_option("scan_filter", filt)
            ctx.scan()
            return ctx.hypotheses(params=False)[2].copy()
        truth = votes(0, 0)
        for name, v in (("fp16", votes(1, 1)), ("fp32", votes(0, 1))):
            if not np.array_equal(v, truth):
                d = np.flatnonzero(v != truth)
                print("MISMATCH", name, cfg, len(d), d[:4], v[d[:4]], truth[d[:4]], flush=True)
                ok = False
        ctx.set_option("us_mfma", 1)
        ctx.set_option("scan_filter", 1)
    res = []
    for bound in (0, 1):
        ctx.set_option("scan_bound", bound)
        r = ctx.batch_fit(seed, 0, H, want_consensus=True)
        res.append((r, ctx.hypotheses(params=False)[2].copy()))
    ctx.set_option("scan_bound", 1)
    (r0, v0), (r1, v1) = res
    if not (np.array_equal(v0, truth) and np.array_equal(r0["consensus"], r1["consensus"])
            and np.array_equal(r0["params"], r1["params"], equal_nan=True)
            and r0["info"].best_index == r1["info"].best_index and r0["info"].best_votes == r1["info"].best_votes):
        print("MISMATCH batch / early exit", cfg, flush=True)
        ok = False
    done += 1
    bad += 0 if ok else 1
    rich += 1 if truth.max() * 10 > len(rows if kind == 'dense' else data) else 0
    if done % 10 == 0:
        print("checked", done, "configurations,", bad, "bad", flush=True)
print("soak_h16: %d configurations (%d with a consensus above 10 %%), %d with a mismatch" % (done, rich, bad))
sys.exit(1 if bad else 0)
