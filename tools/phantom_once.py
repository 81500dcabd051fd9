# per-kernel times of the plane-phantom path: N frames, H hypotheses (k = 31) per batch
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
N, H = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sigma = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
data, truth, lab = synth.plane_phantom_fast(N, 0.05, pixel_sigma=sigma)
ctx = Context(0); ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ITERATIVE).upload(data)
if len(sys.argv) > 5: ctx.set_option('scan_block', int(sys.argv[5]))   # 256: four-wave solve
b = ctx.batch_fit(1, 0, H, want_consensus=True)      # warm-up (+ row matrix)
if sigma == 0.0:
    assert np.array_equal(b["consensus"].astype(bool), lab), "consensus != labels"
    assert synth.phantom_check(b["params"], truth)
print("winner votes %d of %d inlier frames" % (b["info"].best_votes, lab.sum()))
ctx.profile(True)
t0 = time.perf_counter()
for i in range(reps):
    b = ctx.batch_fit(1, (i + 1) * H, H)
ctx.synchronize()
dt = (time.perf_counter() - t0) / reps
out = {k: ctx.profile_get(k) for k in ("sample", "estimate", "scan", "mask", "moments", "solve")}
print("N %d H %d: %.3f ms/batch  %.0f hyp/s  lm_nfev %d" % (N, H, dt * 1e3, H / dt, b["info"].fit.lm_nfev))
for k, (n, ms) in out.items():
    print("  %-9s launches %3d  avg %.4f ms" % (k, n, ms / max(n, 1)))
