# scan-phase times of the three matrix-core filters on the bench's uploads, for A/B runs of two builds of the library:
#   python3 tools/scan_ab.py [reps]
import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
def run(name, model, dim, delta, ls, data, H):
    with Context(0) as ctx:
        ctx.set_model(model, dim, delta, ls).upload(data)
        for bound in (0, 1):
            ctx.set_option("scan_bound", bound)
            ctx.batch_fit(1, 0, H)
            ctx.profile(True)
            for i in range(reps):
                ctx.batch_fit(1, (i + 1) * H, H)
            ctx.synchronize()
            n, ms = ctx.profile_get("scan")
            ctx.profile(False)
            print("%-8s %-10s scan %.4f ms (%d launches)" % (name, "early exit" if bound else "full count", ms / max(n, 1), n), flush=True)
run("us", L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC, synth.us_single_fast(1_000_000, 0.5)[0], 4096)
run("phantom", L.PHANTOM, 0, 2.0, L.LS_ANALYTIC, synth.plane_phantom_fast(1_000_000, 0.5)[0], 4096)
run("dense", L.DENSE, 64, 0.1, L.LS_ALGEBRAIC, synth.dense(2_000_000, 64, 0.05)[0], 1024)
