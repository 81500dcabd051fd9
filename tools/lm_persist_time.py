"""Per-evaluation time of the iterative US fit (BASELINE config 5: 1 M frames, the consensus set of the true model)
three ways -- lm_persist 0 (two launches per evaluation), 1 (persistent kernel, host step), 2 (persistent, device
step) -- alone on the device and with four fits in flight (four host threads, a context each, as bench.py's C5 leg).
Prints one JSON object; run on the GPU box:  python tools/lm_persist_time.py [frames] > gpurun_out/lm_persist.json"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsqrrecipes_amd import _lib as L, synth  # noqa: E402
from lsqrrecipes_amd.context import Context  # noqa: E402


def info(ctx):
    out = (C.c_uint64 * 8)()
    tr = (C.c_uint64 * 256)()
    n = C.c_uint32(0)
    ctx._lib.lsqr_lm_persist_info(ctx._h, out, tr, 64, C.byref(n))
    t = np.array(tr[:4 * n.value], dtype=np.int64).reshape(-1, 4)
    d = dict(mode=int(out[0]), wgs=int(out[1]), evals=int(out[2]), status=int(out[3]), kernel_us=int(out[4]),
             fallbacks=int(out[5]), host_wait_us_per_eval=out[6] / 1e3 / max(int(out[2]), 1),
             host_step_us_per_eval=out[7] / 1e3 / max(int(out[2]), 1))
    if len(t) > 8:   # phases of evaluations 4 .. n-1 in microseconds (the clock ticks at 100 MHz)
        tt = t[4:]
        d["phase_us"] = {"wait_for_arrivals": float(np.mean(tt[:, 1] - tt[:, 0]) / 100),
                         "sum_partials": float(np.mean(tt[:, 2] - tt[:, 1]) / 100),
                         "step_or_host_round_trip": float(np.mean(tt[:, 3] - tt[:, 2]) / 100),
                         "broadcast_and_own_pass": float(np.mean(tt[1:, 0] - tt[:-1, 3]) / 100),
                         "evaluation": float(np.mean(tt[1:, 0] - tt[:-1, 0]) / 100)}
    return d


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    data, truth, lab = synth.us_single_fast(m, 0.5)
    mask = lab.astype(np.uint8)
    res = {"frames": m, "consensus": int(mask.sum()), "runs": []}
    ctxs = [Context(0) for _ in range(4)]
    for c in ctxs:
        c.set_model(L.US_SINGLE, 0, 3.0, L.LS_ITERATIVE).upload(data)
        c.set_mask(mask)

    def fit(c, mode, wgs, resident=1):
        c.set_option("lm_persist", mode)
        c.set_option("lm_persist_wgs", wgs)
        c.set_option("lm_persist_resident", resident)
        t0 = time.perf_counter()
        _, fi = c.ls_fit(True)
        return time.perf_counter() - t0, fi.lm_info, fi.lm_nfev, c.last_iterate.copy()

    base = None
    for mode, wgs, resident in ((0, 0, 0), (1, 256, 1), (1, 256, 0), (1, 128, 0), (1, 64, 0), (2, 256, 0)):
        fit(ctxs[0], mode, wgs, resident)                       # warm
        dt, inf, nfev, x = fit(ctxs[0], mode, wgs, resident)
        if base is None:
            base = x
        r = {"fits_in_flight": 1, "lm_persist": mode, "wgs": wgs, "resident_tiles": bool(resident and mode == 1), "seconds": dt, "lm_info": inf, "lm_nfev": nfev,
             "us_per_evaluation": dt / max(nfev, 1) * 1e6, "same_iterate_as_launch_path": bool(np.array_equal(x, base))}
        if mode:
            r["kernel"] = info(ctxs[0])
        res["runs"].append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)
    for nfl, mode, wgs in ((2, 0, 0), (4, 0, 0), (4, 3, 64), (4, 3, 256)):
        out = [None] * nfl

        def work(k):
            out[k] = fit(ctxs[k], mode, wgs)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(nfl)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.perf_counter() - t0
        nfev = sum(o[2] for o in out)
        r = {"fits_in_flight": nfl, "lm_persist": mode, "wgs": wgs, "seconds": dt, "lm_nfev_total": nfev,
             "us_per_evaluation_aggregate": dt / max(nfev, 1) * 1e6,
             "same_iterate_as_launch_path": bool(all(np.array_equal(o[3], base) for o in out))}
        if mode:
            r["kernel0"] = info(ctxs[0])
        res["runs"].append(r)
        print(json.dumps(r), file=sys.stderr, flush=True)
    for c in ctxs:
        c.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
