# one two-level scan launch per requested variant (for rocprofv3 --pmc runs):
#   python tools/cells_once.py plane N H cell:cpt:hsplit,...
import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
wl, N, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
variants = [tuple(map(int, v.split(':'))) for v in sys.argv[4].split(',')]
gen = {'plane': synth.plane, 'sphere': synth.sphere, 'line': synth.line}[wl]
model = {'plane': L.PLANE, 'sphere': L.SPHERE, 'line': L.LINE}[wl]
data = gen(N, 0.5)[0]
ctx = Context(0); ctx.set_model(model, 3, 0.5).upload(data)
ctx.hypotheses_sample(1, 0, H)
ctx.set_option('scan_index', 2)
for v in variants:
    ctx.set_option('scan_cell', v[0]); ctx.set_option('scan_cpt', v[1])
    ctx.set_option('scan_hsplit', v[2] if len(v) > 2 else 0)
    ctx.scan(); ctx.synchronize()
