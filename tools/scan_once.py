# one bench step (lsqr_batch_fit) of a BASELINE workload on the bench's shapes and seed (for rocprofv3 --pmc passes):
#   python3 tools/scan_once.py plane|sphere|line|us|dense|phantom [launches] [scan_bound: 0 = full count, 1 = early exit]
import sys
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bound = int(sys.argv[3]) if len(sys.argv) > 3 else 1
N = {'dense': 2_000_000, 'us': 1_000_000, 'phantom': 1_000_000}.get(wl, 10_000_000)
H = 1024 if wl == 'dense' else 4096
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line, 'us': synth.us_single_fast}
data = (synth.dense(N, 64, 0.05)[0] if wl == 'dense' else
        synth.plane_phantom_fast(N, 0.05, pixel_sigma=0.05)[0] if wl == 'phantom' else gen[wl](N, 0.5)[0])
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE, 'us': L.US_SINGLE, 'dense': L.DENSE, 'phantom': L.PHANTOM}[wl]
delta = {'dense': 0.1, 'us': 3.0, 'phantom': 2.0}.get(wl, 0.5)
ctx = Context(0)
ctx.set_model(model, 64 if wl == 'dense' else 3, delta, L.LS_ANALYTIC).upload(data)
if wl in ('plane', 'sphere', 'line'):
    ctx.set_option('scan_index', 2)
    ctx.set_option('scan_kd_after', 0)   # the order the bench's timed steps run on (there: built with the fourth batch)
ctx.set_option('scan_bound', bound)
for _ in range(reps):
    # one bench step: sample, solve, scan, winner, mask, closed-form fit
    ctx.batch_fit(0xC0FFEE, 0, H)
    ctx.synchronize()
