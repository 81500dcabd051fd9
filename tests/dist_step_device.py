"""Helper of test_gpu_parity.py::test_step_device_world2_equals_world1 -- run under torch.distributed.run
with 2 ranks (gloo, both ranks on GPU 0): ShardedRansac.step_device() with the exchanges as collectives on
device tensors must pick the same winner and consensus count as one process scanning the whole batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

torch.cuda.init()  # torch's HIP runtime has to come up before the library's
torch.cuda.set_device(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lsqrrecipes_amd import _lib as L, synth  # noqa: E402
from lsqrrecipes_amd.context import Context  # noqa: E402
from lsqrrecipes_amd.distributed import Comm, ShardedRansac  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ok = True
for model, dim, ls, H in ((L.PLANE, 3, 0, 1024), (L.SPHERE, 3, L.LS_GEOMETRIC, 1024), (L.DENSE, 8, 0, 128),
                          (L.PHANTOM, 0, L.LS_ITERATIVE, 128), (L.US_POINTER, 0, L.LS_ITERATIVE, 256),
                          (L.US_SINGLE, 0, L.LS_ITERATIVE, 256)):
    loose = False
    if model == L.DENSE:
        data, delta = synth.dense(40_000, 8, 0.3, seed=5)[0], 0.1
    elif model == L.US_POINTER:   # BASELINE config 5's sibling: LM with 1e-7 tolerances, one all-reduce per evaluation
        data, delta = synth.us_pointer(6000, 0.3, seed=5)[0], 3.0
    elif model == L.US_SINGLE:    # config 5's model: 1e-15 tolerances -- thousands of evaluations along a flat valley, the
        data, delta = synth.us_single_fast(3000, 0.3, seed=5)[0], 3.0   # iterates of two summation orders part ways
        loose = True
    elif model == L.PHANTOM:
        data, delta = synth.plane_phantom_fast(20_000, 0.05, seed=5, pixel_sigma=0.02)[0], 2.0
    else:
        gen = {L.PLANE: synth.plane, L.SPHERE: synth.sphere}[model]
        data, delta = gen(120_000, 0.5, seed=77, dim=dim)[0], 0.5
    with Context(0) as c:
        c.set_model(model, dim, delta, ls).upload(data)
        got = ShardedRansac(c, Comm(dist, "cpu")).step_device(11, 2, H)
        if rank == 0:
            with Context(0) as c1:
                c1.set_model(model, dim, delta, ls).upload(data)
                want = ShardedRansac(c1, Comm(None)).step(11, 2, H * world)
            same = (got[0], got[1], got[4]) == (want[0], want[1], want[4]) and np.array_equal(got[2], want[2])
            if loose:   # same winner and consensus; the fit: the same cost to 1e-6, LM ran on both sides
                gi, wi = got[5], want[5]
                same = same and gi.lm_nfev > 0 and wi.lm_nfev > 0 and \
                    abs(gi.cost - wi.cost) <= 1e-6 * max(wi.cost, 1e-300) and len(got[3]) == len(want[3])
            else:
                same = same and np.allclose(got[3], want[3], rtol=1e-9, atol=1e-9)
                if model in (L.SPHERE, L.US_POINTER):
                    same = same and got[5].lm_nfev > 0 and 1 <= got[5].lm_info <= 4
            if not same:
                print("MISMATCH", model, got[:2], want[:2], got[4], want[4], got[3], want[3])
            ok = ok and same
        if model == L.PLANE:   # pipelined steps over the collectives
            sr = ShardedRansac(c, Comm(dist, "cpu"))
            blocking = [sr.step_device(11, b, H) for b in range(3)]
            sr.step_device(11, 0, H, slot=0)
            sr.step_device(11, 1, H, slot=1)
            piped = [sr.step_device_wait(0)]
            sr.step_device(11, 2, H, slot=0)
            piped += [sr.step_device_wait(1), sr.step_device_wait(0)]
            for g, w in zip(piped, blocking):
                if not ((g[0], g[1], g[4]) == (w[0], w[1], w[4]) and np.array_equal(g[3], w[3])):
                    print("PIPELINE MISMATCH", g[:2], w[:2])
                    ok = False
        c.set_stream(None)
    dist.barrier()
if rank == 0:
    print("world2 step_device ok" if ok else "world2 step_device FAILED")
dist.destroy_process_group()
