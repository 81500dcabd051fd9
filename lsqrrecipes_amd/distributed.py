"""Multi-GPU RANSAC step: hypothesis ranges sharded over ranks, observations replicated.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  The path shards trivially: every rank scans its own slice of the
hypothesis stream against its replica of the observations; the exchange steps are tiny and
latency-bound:
  C1  all-reduce(MAX) of one int64 = (votes << 32) | (0xFFFFFFFF - index in the global batch):
      picks the earliest best hypothesis exactly as the strict '>' of RANSAC.hxx:100 does.  The
      winner's parameters need no second exchange: the sampler is counter-based, so every rank
      re-derives the winning subset from its stream index and solves it on its own replica of
      the observations (bit-identical on every rank).
  C2  all-reduce(SUM) of [fp64 moment block, inlier count] of each rank's observation slice for
      the final fit (and of the {sum f^2, J^T J, J^T f} block per LM evaluation).
`engine` is a lsqrrecipes_amd.context.Context (tests substitute an object with the same methods).
"""
import numpy as np


class Comm:
    """Minimal collective interface over torch.distributed (or a no-op for world size 1).
    The messages are tiny (8 B .. 17 KB) and latency-bound, so the device and pinned host staging
    buffers are allocated once and every exchange is copy-in, all_reduce, copy-out, one sync."""

    def __init__(self, dist=None, device="cpu", group=None):
        self.dist = dist
        self.device = device
        self.group = group    # process group of the collectives (None: the default group); one per stream when
                              # steps of several streams are in flight: a communicator serves one stream at a time
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self._bufs = {}

    def _staging(self, dtype, n):
        import torch
        key = (dtype, n)
        if key not in self._bufs:
            on_gpu = str(self.device) != "cpu"
            host = torch.zeros(n, dtype=dtype, pin_memory=on_gpu)
            dev = torch.zeros(n, dtype=dtype, device=self.device) if on_gpu else host
            self._bufs[key] = (host, dev)
        return self._bufs[key]

    def _allreduce(self, values, dtype, op):
        import torch
        host, dev = self._staging(dtype, len(values))
        host.copy_(torch.as_tensor(values, dtype=dtype))
        if dev is not host:
            dev.copy_(host, non_blocking=True)
        self.dist.all_reduce(dev, op=op, group=self.group)
        if dev is not host:
            host.copy_(dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return host

    def allreduce_max_i64(self, value):
        if self.dist is None:
            return int(value)
        import torch
        return int(self._allreduce([int(value)], torch.int64, self.dist.ReduceOp.MAX)[0])

    def allreduce_sum_f64(self, arr):
        a = np.asarray(arr, dtype=np.float64)
        if self.dist is None:
            return a
        import torch
        return self._allreduce(a, torch.float64, self.dist.ReduceOp.SUM).numpy().copy()

    def allreduce_max_f64(self, value):
        if self.dist is None:
            return float(value)
        import torch
        return float(self._allreduce([float(value)], torch.float64, self.dist.ReduceOp.MAX)[0])

    def allgather_f64(self, value):
        """every rank's scalar, in rank order (a one-hot sum: the message is 8 B per rank)"""
        if self.dist is None:
            return [float(value)]
        v = np.zeros(self.world)
        v[self.rank] = float(value)
        return [float(x) for x in self.allreduce_sum_f64(v)]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier(group=self.group)


def slice_bounds(n, rank, world):
    """Contiguous observation slice of a rank (fixed split => deterministic reduction order)."""
    return (n * rank) // world, (n * (rank + 1)) // world


class ShardedRansac:
    def __init__(self, engine, comm):
        self.e = engine
        self.c = comm
        self._xbuf = None     # (packed int64[1], block float64[nmom + 1]) device tensors of step_device
        self._stream = None
        self._pending = {}
        self._lmbuf = None

    def batch(self, seed, batch_index, H):
        """One batch of world*H hypotheses: sample+solve+scan own slice, pick the global winner.
        Returns (votes, global_index_in_stream, params) of the earliest best hypothesis, or
        (0, None, None) when no hypothesis of the batch was valid."""
        e, c = self.e, self.c
        first = (batch_index * c.world + c.rank) * H
        e.hypotheses_sample(seed, first, H)
        e.scan()
        packed, votes, idx = e.best()
        mine = 0
        if packed:
            in_batch = c.rank * H + idx
            mine = (int(votes) << 32) | (0xFFFFFFFF - in_batch)
        win = c.allreduce_max_i64(mine)
        if win == 0:
            return 0, None, None
        wvotes = win >> 32
        in_batch = 0xFFFFFFFF - (win & 0xFFFFFFFF)
        gidx = batch_index * c.world * H + in_batch
        if c.world == 1:
            par, _ = e.hypothesis(in_batch)
        else:
            e.hypotheses_sample(seed, gidx, 1)  # stateless sampler: same subset, same bits everywhere
            par, _ = e.hypothesis(0)
        return int(wvotes), gidx, par

    def _check_packable(self):
        """the winner travels as (votes << 32) | ~index in a SIGNED int64 all-reduce(MAX): votes must stay below 2^31
        (the C ABI itself compares the word unsigned and accepts up to 0xFFFFFFF0 observations)"""
        if getattr(self.e, "n", 0) >= 2 ** 31:
            raise ValueError("ShardedRansac: %d observations -- the int64 MAX all-reduce of the packed winner needs "
                             "fewer than 2^31" % self.e.n)

    def step(self, seed, batch_index, H):
        """One whole multi-GPU step: batch() + fit() with the fewest host synchronisations the
        engine offers.  Returns (votes, global_index, params_of_winner, fit, inliers, info) or None
        when no hypothesis of the batch was valid."""
        e, c = self.e, self.c
        self._check_packable()
        if not hasattr(e, "winner_moments"):
            votes, gidx, par = self.batch(seed, batch_index, H)
            if gidx is None:
                return None
            fit, total, info = self.fit(par)
            return votes, gidx, par, fit, total, info
        first = (batch_index * c.world + c.rank) * H
        e.hypotheses_sample(seed, first, H)
        e.scan()
        packed, votes, idx = e.best()
        mine = ((int(votes) << 32) | (0xFFFFFFFF - (c.rank * H + idx))) if packed else 0
        win = c.allreduce_max_i64(mine)
        if win == 0:
            return None
        in_batch = 0xFFFFFFFF - (win & 0xFFFFFFFF)
        gidx = batch_index * c.world * H + in_batch
        lo, hi = slice_bounds(e.n, c.rank, c.world)
        par, origin, blk, cnt = e.winner_moments(seed, gidx, lo, hi)
        blk = c.allreduce_sum_f64(np.concatenate([blk, [float(cnt)]]))
        fit, info = e.solve_moments(blk[:-1], origin)
        fit, info = self._refine(fit, info, lo, hi, par)
        return int(win >> 32), gidx, par, fit, int(round(blk[-1])), info

    def step_device(self, seed, batch_index, H, slot=None):
        """The same step with the two exchanges as collectives on DEVICE buffers and one host
        synchronisation (lsqr_step_scan / _winner / _finish): scan -> k_best into `packed` ->
        all-reduce MAX -> winner re-derived on the device from `packed` -> mask + moment block of the
        rank's slice into `block` -> all-reduce SUM -> solve.  The context enqueues on torch's current
        stream, which the process group orders its collectives with.  Same return value as step().
        torch bundles its own HIP runtime: call torch.cuda.init() before the first Context is created in a
        process that uses both (two runtimes initialised in the other order do not see the devices)."""
        import torch
        e, c = self.e, self.c
        self._check_packable()
        if self._xbuf is None:
            dev = torch.device("cuda", torch.cuda.current_device())
            nmom = e.moments_len(0)
            self._xbuf = (torch.zeros(1, dtype=torch.int64, device=dev),
                          torch.zeros(nmom + 1, dtype=torch.float64, device=dev))
        stream = int(torch.cuda.current_stream().cuda_stream)   # 0: the default stream
        if stream != self._stream:
            e.set_stream(stream)
            self._stream = stream
        packed, block = self._xbuf
        first = (batch_index * c.world + c.rank) * H
        e.step_scan(seed, first, H, c.rank * H, packed.data_ptr())
        if c.dist is not None:
            c.dist.all_reduce(packed, op=c.dist.ReduceOp.MAX, group=c.group)
        lo, hi = slice_bounds(e.n, c.rank, c.world)
        e.step_winner(seed, batch_index * c.world * H, packed.data_ptr(), lo, hi, block.data_ptr())
        if c.dist is not None:
            c.dist.all_reduce(block, op=c.dist.ReduceOp.SUM, group=c.group)
        if slot is not None:     # pipelined: results are fetched later with step_device_wait(slot)
            e.step_finish_enqueue(packed.data_ptr(), block.data_ptr(), slot)
            self._pending[slot] = (batch_index, H, lo, hi)
            return None
        st, par, fit, info = e.step_finish(packed.data_ptr(), block.data_ptr())
        return self._step_result(info, par, fit, batch_index, H, lo, hi)

    def _step_result(self, info, par, fit, batch_index, H, lo, hi):
        if not info.evaluated:   # no valid hypothesis in the whole batch
            return None
        gidx = batch_index * self.c.world * H + int(info.best_index)
        fit, finfo = self._refine(fit, info.fit, lo, hi, par)
        return int(info.best_votes), gidx, par, fit, int(info.fit.n_used), finfo

    def step_device_wait(self, slot):
        """Result of the step enqueued with step_device(..., slot=slot) (closed-form fits only: an LM
        refinement needs further device passes and belongs to the blocking form)."""
        batch_index, H, lo, hi = self._pending.pop(slot)
        st, par, fit, info = self.e.step_finish_wait(slot)
        return self._step_result(info, par, fit, batch_index, H, lo, hi)

    def _refine(self, fit, info, lo, hi, winner=None):
        """LM refinement over the sharded observation range (sphere geometric / US iterative); dense system: a fit from
        the rows when the summed Gram block alone cannot guarantee 1e-6."""
        e, c = self.e, self.c
        from . import _lib as L
        model = e.cfg.model
        if model == L.DENSE and winner is not None and getattr(info, "reserved", 0) == 2 and hasattr(e, "ls_fit"):
            # the summed block is ill-conditioned (a pivot below 1e-6 max|G|, lsqr_hip.h: lsqr_fit_info.reserved == 2):
            # its solution carries eps cond(A)^2 where a one-GPU fit takes the double-double route over the rows.  The
            # records are replicated: every rank masks the whole upload with the winner and fits from the rows --
            # identical on every rank, no exchange, and equal to the one-GPU result
            n_used = info.n_used
            e.mask(winner, 0, e.n, want_mask=False)
            fit, info = e.ls_fit(use_mask=True)
            info.n_used = n_used
            return fit, info
        iterative = (model == L.SPHERE and e.cfg.ls_type == L.LS_GEOMETRIC) or (
            model in (L.US_SINGLE, L.US_POINTER) and e.cfg.ls_type == L.LS_ITERATIVE)
        if len(fit) and iterative and self._stream is not None and hasattr(e, "moments_dev"):
            # device path (step_device): the LM block is all-reduced in device memory, one read-back per
            # evaluation
            import torch
            n1 = e.moments_len(1)
            if self._lmbuf is None or self._lmbuf.numel() != n1:
                self._lmbuf = torch.zeros(n1, dtype=torch.float64, device=self._xbuf[1].device)
            xt = e.lm_begin(fit)
            while True:
                e.moments_dev(xt[:e.P], self._lmbuf.data_ptr(), lo, hi, phase=1, use_mask=True)
                if c.dist is not None:
                    c.dist.all_reduce(self._lmbuf, op=c.dist.ReduceOp.SUM, group=c.group)
                cont, xt, fit, info = e.lm_step(self._lmbuf.cpu().numpy())
                if not cont:
                    break
        elif len(fit) and iterative:
            xt = e.lm_begin(fit)
            while True:
                blk = c.allreduce_sum_f64(e.moments(xt[:e.P], lo, hi, phase=1, use_mask=True))
                cont, xt, fit, info = e.lm_step(blk)
                if not cont:
                    break
        return fit, info

    def fit(self, params):
        """Consensus mask of `params` + final least-squares fit, observation range sharded."""
        e, c = self.e, self.c
        from . import _lib as L
        lo, hi = slice_bounds(e.n, c.rank, c.world)
        _, cnt = e.mask(params, lo, hi, want_mask=False)
        model = e.cfg.model
        if model == L.SPHERE:
            origin = np.asarray(params[:e.ND])
        elif model in (L.PLANE, L.LINE):
            origin = np.asarray(params[e.ND:2 * e.ND])
        else:
            origin = np.zeros(3)  # dense / US / rigid blocks are not taken about a model point
        blk = np.concatenate([e.moments(origin, lo, hi, phase=0, use_mask=True), [float(cnt)]])
        blk = c.allreduce_sum_f64(blk)  # one fused exchange: moment block + inlier count
        total = int(round(blk[-1]))
        fit, info = e.solve_moments(blk[:-1], origin)
        fit, info = self._refine(fit, info, lo, hi, params)
        return fit, total, info
