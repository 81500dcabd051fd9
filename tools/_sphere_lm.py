import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
N, H = 10_000_000, 4096
data = synth.sphere(N, 0.5)[0]
ctx = Context(0)
ctx.set_model(L.SPHERE, 3, 0.5, L.LS_GEOMETRIC if hasattr(L, 'LS_GEOMETRIC') else 1).upload(data)
ctx.set_option('scan_index', 2)
for s in range(3):
    ctx.batch_fit(0xC0FFEE, s * H, H)
ctx.synchronize()
