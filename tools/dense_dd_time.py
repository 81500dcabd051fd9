# time of the dense least-squares fit at BASELINE config 4's size (2 M x 64) on a well- and an ill-conditioned system:
# elimination on the Gram block alone against the double-double route from the rows (dense.h)
import sys, time
sys.path.insert(0, '.')
import numpy as np
from lsqrrecipes_amd import _lib as L
from lsqrrecipes_amd.context import Context
m, n = 2_000_000, 64
g = np.random.default_rng(1)
A = g.uniform(-1, 1, (m, n))
x = g.uniform(-1, 1, n)
ctx = Context(0)
for name, scale in (("well-conditioned (uniform entries)", None), ("column scales 1 .. 1e-8 (cond ~ 1e8)", 1e-8)):
    B = A if scale is None else A * np.logspace(0, np.log10(scale), n)[None, :]
    rows = np.ascontiguousarray(np.hstack([B, (B @ x)[:, None]]))
    ctx.set_model(L.DENSE, n, 0.1).upload(rows)
    for dd in (1, 0):
        ctx.set_option("dense_dd", dd)
        ctx.ls_fit(use_mask=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        K = 5
        for _ in range(K):
            got, info = ctx.ls_fit(use_mask=False)
        dt = (time.perf_counter() - t0) / K
        err = float(np.linalg.norm(got - x) / np.linalg.norm(x)) if len(got) else None
        print(name, "| dense_dd", dd, "| ls_fit %.3f ms" % (dt * 1e3), "| dd route used:", bool(info.reserved),
              "| empty:", len(got) == 0, "| error vs the exact solution:", err, flush=True)
