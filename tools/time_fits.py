# wall time of the final fits at BASELINE sizes (leastSquaresEstimate on resident data)
import sys, time, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
ctx = Context(0)
def run(name, model, dim, delta, ls, data, mask=None):
    ctx.set_model(model, dim, delta, ls).upload(data)
    if mask is not None: ctx.set_mask(mask)
    ctx.ls_fit(use_mask=mask is not None)
    ctx.profile(True)
    t0 = time.perf_counter(); fit, info = ctx.ls_fit(use_mask=mask is not None); dt = time.perf_counter() - t0
    nm, mm = ctx.profile_get('moments'); ns, ms = ctx.profile_get('solve'); ctx.profile(False)
    print('%-28s n=%d: %.3f ms wall, lm_info=%d nfev=%d, moments %d x %.3f ms, solve/reduce %d x %.3f ms, params[:4]=%s' % (
        name, len(data), dt * 1e3, info.lm_info, info.lm_nfev, nm, mm / max(nm, 1), ns, ms / max(ns, 1), np.round(fit[:4], 6)))
d, t, lab = synth.plane(10_000_000, 0.5); run('plane LS (inliers)', L.PLANE, 3, 0.5, 0, d, lab.astype(np.uint8))
d, t, lab = synth.sphere(10_000_000, 0.5); run('sphere algebraic (inliers)', L.SPHERE, 3, 0.5, 0, d, lab.astype(np.uint8)); run('sphere geometric LM', L.SPHERE, 3, 0.5, 1, d, lab.astype(np.uint8))
d, t, lab = synth.dense(2_000_000, 64, 0.05); run('dense 2Mx64 normal eq.', L.DENSE, 64, 0.1, 0, d, lab.astype(np.uint8))
d, t, lab = synth.us_single_fast(1_000_000, 0.3); run('US analytic 1M frames', L.US_SINGLE, 0, 3.0, 0, d, lab.astype(np.uint8)); run('US iterative LM 1M frames', L.US_SINGLE, 0, 3.0, 1, d, lab.astype(np.uint8))
