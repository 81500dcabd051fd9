# a few bench-shaped steps of the dense / US workload with the early exit on (for rocprofv3 --kernel-trace --stats):
#   python3 tools/ee_once.py dense|us [scan_bound] [steps]
import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1]
bound = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if wl == 'dense':
    H, data, model, dim, delta = 1024, synth.dense(2_000_000, 64, 0.05)[0], L.DENSE, 64, 0.1
else:
    H, data, model, dim, delta = 4096, synth.us_single_fast(1_000_000, 0.5)[0], L.US_SINGLE, 3, 3.0
ctx = Context(0)
ctx.set_model(model, dim, delta, L.LS_ANALYTIC).upload(data)
ctx.set_option('scan_bound', bound)
for s in range(steps):
    ctx.batch_fit(0xC0FFEE, s * H, H)
ctx.synchronize()
