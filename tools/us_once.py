# one US scan launch per variant (filter:ppl) for rocprofv3 --pmc runs
import sys, numpy as np
sys.path.insert(0, '.')
from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
N, H = int(sys.argv[1]), int(sys.argv[2])
data = synth.us_single_fast(N, 0.5)[0]
ctx = Context(0); ctx.set_model(L.US_SINGLE, 0, 3.0, L.LS_ANALYTIC).upload(data)
ctx.hypotheses_sample(1, 0, H)
for v in sys.argv[3].split(','):
    f, p = map(int, v.split(':'))
    ctx.set_option('scan_filter', f); ctx.set_option('scan_ppl', p)
    ctx.scan(); ctx.synchronize()
