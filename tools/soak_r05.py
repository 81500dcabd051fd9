"""Differential soak of round 5's new paths (run on the GPU box; exit code 1 on a mismatch):
  * plane phantom minimal solves: LU + inverse iteration (phantom_fast_solve 1) against the Jacobi SVD (0) -- the same
    hypotheses valid, the 41 parameters within 1e-6 (the null vector's common sign aligned, angles modulo 2 pi;
    tests/soak_phantom_lu.py referees anything further apart with the oracle) -- over random uploads (frames,
    off-plane fraction, pixel noise over four orders of magnitude, translations rescaled);
  * iterative US fits: the persistent kernel with the host's step (lm_persist 3) and with the device's step (2, on the
    smaller sets) against the launch path (0) -- last iterate, info, nfev bit for bit -- over random single / pointer
    uploads, sizes, noise levels, consensus masks and workgroup counts.
    python tools/soak_r05.py [seconds] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from lsqrrecipes_amd import _lib as L, synth  # noqa: E402
from lsqrrecipes_amd.context import Context  # noqa: E402

T = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)


def align(p, q):
    """the null vector's sign is arbitrary (tests/soak_phantom_lu.py: align)"""
    q = q.copy()
    blk = list(range(11, 41)) + [2]
    if np.dot(p[blk], q[blk]) < 0:
        q[blk] = -q[blk]
        if abs(abs(q[0]) - np.pi / 2) > 0.008726535498373935:   # else the reference sets omega_x to zero
            q[1] = q[1] - np.pi if abs(q[1] - np.pi - p[1]) < abs(q[1] + np.pi - p[1]) else q[1] + np.pi
        q[0] = -q[0]
    return q


def apart(p, q):
    q = align(p, q)
    d = np.abs(q - p)
    ang = [0, 1, 6, 7, 8]
    d[ang] = np.minimum(d[ang], np.abs(2 * np.pi - d[ang]))
    return float(np.max(d / np.maximum(np.abs(p), 1e-3 * np.abs(p).max())))


def phantom_round(ctx):
    n = int(g.integers(2_000, 200_000))
    sigma = float(10 ** g.uniform(-3, 0.5))
    out = float(g.uniform(0.0, 0.3))
    data = synth.plane_phantom_fast(n, out, seed=int(g.integers(1 << 30)), pixel_sigma=sigma)[0]
    if g.random() < 0.3:
        data[:, 9:12] *= float(10 ** g.uniform(-2, 2))      # translations rescaled
    H = int(g.choice([64, 333, 1024, 2048]))
    seed = int(g.integers(1 << 40))
    res = []
    for fast in (0, 1):
        ctx.set_option("phantom_fast_solve", fast)
        ctx.set_model(L.PHANTOM, 0, 2.0, L.LS_ANALYTIC).upload(data)
        ctx.hypotheses_sample(seed, 0, H)
        res.append(ctx.hypotheses(votes=False))
    (pj, vj, _), (pf, vf, _) = res
    bad = int(np.count_nonzero(vj != vf))
    worst = max([0.0] + [apart(pj[h], pf[h]) for h in np.flatnonzero(vj & vf)])
    return H, bad, worst


def lm_round(ctx):
    single = g.random() < 0.6
    n = int(10 ** g.uniform(2.0, 5.3))
    sigma = float(10 ** g.uniform(-1.5, 0.5))
    if single:
        data, model = synth.us_single_fast(n, 0.0, seed=int(g.integers(1 << 30)), pixel_sigma=sigma)[0], L.US_SINGLE
    else:
        data, model = synth.us_pointer(min(n, 60_000), 0.0, seed=int(g.integers(1 << 30)), pixel_sigma=sigma)[0], L.US_POINTER
    mask = (g.random(len(data)) < g.uniform(0.3, 1.0)).astype(np.uint8)
    mask[:16] = 1
    out = []
    modes = [(0, 0), (3, int(g.choice([0, 1, 5, 64, 200])))]
    if len(data) <= 30_000:
        modes.append((2, int(g.choice([0, 3, 64]))))
    for mode, wgs in modes:
        ctx.set_option("lm_persist", mode)
        ctx.set_option("lm_persist_wgs", wgs)
        ctx.set_model(model, 0, 3.0, L.LS_ITERATIVE).upload(data)
        ctx.set_mask(mask)
        fit, info = ctx.ls_fit(True)
        out.append((ctx.last_iterate.copy(), info.lm_info, info.lm_nfev, len(fit)))
    ok = all(np.array_equal(o[0], out[0][0]) and o[1:] == out[0][1:] for o in out[1:])
    return len(modes), ok, out[0][2]


def main():
    t0 = time.time()
    ph = {"uploads": 0, "hypotheses": 0, "validity_mismatches": 0, "worst_apart": 0.0}
    lm = {"uploads": 0, "fits": 0, "evaluations": 0, "mismatches": 0}
    with Context(0) as ctx:
        while time.time() - t0 < T:
            if g.random() < 0.5:
                H, bad, worst = phantom_round(ctx)
                ph["uploads"] += 1
                ph["hypotheses"] += H
                ph["validity_mismatches"] += bad
                ph["worst_apart"] = max(ph["worst_apart"], worst)
            else:
                k, ok, nfev = lm_round(ctx)
                lm["uploads"] += 1
                lm["fits"] += k
                lm["evaluations"] += nfev * k
                lm["mismatches"] += not ok
    print("phantom LU vs Jacobi:", ph)
    print("persistent LM vs launch path:", lm)
    fail = ph["validity_mismatches"] or ph["worst_apart"] >= 1e-6 or lm["mismatches"]
    print("soak_r05: %s in %.0f s" % ("MISMATCH" if fail else "no difference", time.time() - t0))
    sys.exit(1 if fail else 0)


if __name__ == "__main__":
    main()
