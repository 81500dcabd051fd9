"""Context: object wrapper over one lsqr_ctx of the C ABI (one device, one stream)."""
import ctypes as C

import numpy as np

from . import _lib as L


class Context:
    def __init__(self, device=0):
        self._lib = L.load()
        h = C.c_void_p()
        st = self._lib.lsqr_ctx_create(int(device), C.byref(h))
        if st != L.OK:
            raise L.LsqrError(st, "lsqr_ctx_create(device=%d): %s" % (
                device, self._lib.lsqr_status_string(st).decode()))
        self._h = h
        self.device = device
        self.cfg = None
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lsqr_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st, allow_empty=False):
        if st == L.OK or (allow_empty and st == L.EMPTY):
            return st
        raise L.LsqrError(st, "%s (%s)" % (self._lib.lsqr_status_string(st).decode(),
                                           self._lib.lsqr_last_error(self._h).decode()))

    # ---- model / data -------------------------------------------------------------------
    def set_model(self, model, dim=3, delta=0.5, ls_type=L.LS_GEOMETRIC, aux=0.0):
        self.cfg = L.ModelCfg(int(model), int(dim), float(delta), int(ls_type), 0, float(aux))
        self._chk(self._lib.lsqr_set_model(self._h, C.byref(self.cfg)))
        self.K = self._lib.lsqr_min_subset(C.byref(self.cfg))
        self.P = self._lib.lsqr_num_params(C.byref(self.cfg))
        self.ND = self._lib.lsqr_record_doubles(C.byref(self.cfg))
        return self

    def upload(self, data):
        a = np.ascontiguousarray(data, dtype=np.float64)
        a = a.reshape(-1, a.shape[-1]) if a.ndim > 1 else a.reshape(-1, self.ND)
        self._chk(self._lib.lsqr_upload(self._h, L.ptr(a), a.shape[0], a.shape[1] * 8))
        self.n = a.shape[0]
        return self

    def attach(self, device_ptr, count, stride_bytes, keepalive=None):
        self._keep = keepalive
        self._chk(self._lib.lsqr_attach(self._h, C.c_void_p(device_ptr), count, stride_bytes))
        self.n = count
        return self

    # ---- hypotheses ---------------------------------------------------------------------
    def hypotheses_from_subsets(self, subsets):
        s = np.ascontiguousarray(subsets, dtype=np.uint32).reshape(-1, self.K)
        self._chk(self._lib.lsqr_hypotheses_from_subsets(self._h, L.ptr(s), s.shape[0]))
        return s.shape[0]

    def hypotheses_sample(self, seed, first, H, want_subsets=False):
        out = np.zeros((H, self.K), dtype=np.uint32) if want_subsets else None
        self._chk(self._lib.lsqr_hypotheses_sample(self._h, seed, first, H, L.ptr(out)))
        return out

    def scan(self):
        self._chk(self._lib.lsqr_scan(self._h))

    def hypotheses(self, params=True, valid=True, votes=True):
        H = self._lib.lsqr_num_hypotheses(self._h)
        p = np.zeros((H, self.P)) if params else None
        v = np.zeros(H, dtype=np.uint8) if valid else None
        c = np.zeros(H, dtype=np.uint32) if votes else None
        self._chk(self._lib.lsqr_get_hypotheses(self._h, L.ptr(p), L.ptr(v), L.ptr(c)))
        return p, v, c

    def hypothesis(self, h):
        p = np.zeros(self.P)
        v = np.zeros(1, dtype=np.uint8)
        self._chk(self._lib.lsqr_get_hypothesis(self._h, int(h), L.ptr(p), L.ptr(v)))
        return p, bool(v[0])

    def best(self):
        packed = C.c_uint64(0)
        self._chk(self._lib.lsqr_best(self._h, C.byref(packed)))
        v = packed.value
        return v, v >> 32, 0xFFFFFFFF - (v & 0xFFFFFFFF)

    # ---- mask / fit ---------------------------------------------------------------------
    def mask(self, params, begin=0, end=None, want_mask=True):
        end = self.n if end is None else end
        p = np.ascontiguousarray(params, dtype=np.float64)
        m = np.zeros(end - begin, dtype=np.uint8) if want_mask else None
        cnt = C.c_uint64(0)
        self._chk(self._lib.lsqr_mask(self._h, L.ptr(p), begin, end, L.ptr(m), C.byref(cnt)))
        return m, cnt.value

    def mask_from_hypothesis(self, h, want_mask=True):
        m = np.zeros(self.n, dtype=np.uint8) if want_mask else None
        cnt = C.c_uint64(0)
        self._chk(self._lib.lsqr_mask_from_hypothesis(self._h, h, L.ptr(m), C.byref(cnt)))
        return m, cnt.value

    def set_mask(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape[0] == self.n
        self._chk(self._lib.lsqr_set_mask(self._h, L.ptr(m)))

    def ls_fit(self, use_mask=False):
        """-> (params ndarray, possibly empty; FitInfo)"""
        out = np.zeros(max(self.P, 32))
        info = L.FitInfo()
        st = self._chk(self._lib.lsqr_ls_fit(self._h, int(use_mask), L.ptr(out), C.byref(info)),
                       allow_empty=True)
        self.last_iterate = out[:self.P].copy()   # LM runs: the last iterate even when the fit is reported failed
        return (out[:info.n_params].copy() if st == L.OK else np.zeros(0)), info

    def moments_len(self, phase):
        return self._lib.lsqr_moments_len(C.byref(self.cfg), phase)

    def moments(self, x, begin=0, end=None, phase=0, use_mask=False):
        end = self.n if end is None else end
        xv = np.zeros(32)
        x = np.asarray(x, dtype=np.float64)
        xv[:len(x)] = x
        blk = np.zeros(self.moments_len(phase))
        self._chk(self._lib.lsqr_moments(self._h, int(use_mask), begin, end, phase, L.ptr(xv),
                                         L.ptr(blk)))
        return blk

    def moments_dev(self, x, block_ptr, begin=0, end=None, phase=0, use_mask=False):
        """moments() with the block left at device address block_ptr, no synchronisation"""
        end = self.n if end is None else end
        xv = np.zeros(32)
        x = np.asarray(x, dtype=np.float64)
        xv[:len(x)] = x
        self._chk(self._lib.lsqr_moments_dev(self._h, int(use_mask), begin, end, phase, L.ptr(xv),
                                             C.c_void_p(block_ptr)))

    def winner_moments(self, seed, stream_index, begin=0, end=None):
        """-> (params, origin(32), block, count): lsqr_winner_moments, one host synchronisation"""
        end = self.n if end is None else end
        par = np.zeros(max(self.P, 1))
        org = np.zeros(32)
        blk = np.zeros(self.moments_len(0))
        cnt = C.c_uint64(0)
        self._chk(self._lib.lsqr_winner_moments(self._h, seed, stream_index, begin, end, L.ptr(par),
                                                L.ptr(org), L.ptr(blk), C.byref(cnt)))
        return par, org, blk, cnt.value

    def solve_moments(self, block, origin):
        b = np.ascontiguousarray(block, dtype=np.float64)
        o = np.zeros(32)
        o[:len(origin)] = origin
        out = np.zeros(max(self.P, 32))
        info = L.FitInfo()
        st = self._chk(self._lib.lsqr_solve_moments(self._h, L.ptr(b), L.ptr(o), L.ptr(out),
                                                    C.byref(info)), allow_empty=True)
        return (out[:info.n_params].copy() if st == L.OK else np.zeros(0)), info

    def lm_begin(self, x0):
        x = np.zeros(32)
        x0 = np.asarray(x0, dtype=np.float64)[:32]
        x[:len(x0)] = x0
        xt = np.zeros(32)
        self._chk(self._lib.lsqr_lm_begin(self._h, L.ptr(x), L.ptr(xt)))
        return xt

    def lm_step(self, block):
        b = np.ascontiguousarray(block, dtype=np.float64)
        xt = np.zeros(32)
        out = np.zeros(max(self.P, 64))
        cont = C.c_int(0)
        info = L.FitInfo()
        st = self._chk(self._lib.lsqr_lm_step(self._h, L.ptr(b), L.ptr(xt), C.byref(cont),
                                              L.ptr(out), C.byref(info)), allow_empty=True)
        return bool(cont.value), xt, (out[:info.n_params].copy() if st == L.OK else np.zeros(0)), info

    def stats(self, params, use_mask=False):
        p = np.ascontiguousarray(params, dtype=np.float64)
        out = np.zeros(4)
        self._chk(self._lib.lsqr_stats(self._h, L.ptr(p), int(use_mask), L.ptr(out)))
        return out

    # ---- multi-GPU step with device-resident exchange buffers ---------------------------
    def set_stream(self, hip_stream):
        """enqueue on the caller's HIP stream (int handle, e.g. torch.cuda.current_stream().cuda_stream;
        0 is the default stream); None restores the context's own stream"""
        if hip_stream is None:
            self._chk(self._lib.lsqr_set_stream(self._h, None, 0))
        else:
            self._chk(self._lib.lsqr_set_stream(self._h, C.c_void_p(int(hip_stream) or None), 1))

    def step_scan(self, seed, first, H, index_base, packed_ptr):
        self._chk(self._lib.lsqr_step_scan(self._h, seed, first, H, index_base, C.c_void_p(packed_ptr)))

    def step_winner(self, seed, batch_first, packed_ptr, begin, end, block_ptr):
        self._chk(self._lib.lsqr_step_winner(self._h, seed, batch_first, C.c_void_p(packed_ptr), begin, end,
                                             C.c_void_p(block_ptr)))

    def step_finish(self, packed_ptr, block_ptr):
        """-> (status, winner params, fitted params (possibly empty), RansacInfo)"""
        win = np.zeros(max(self.P, 1))
        out = np.zeros(max(self.P, 64))
        info = L.RansacInfo()
        st = self._chk(self._lib.lsqr_step_finish(self._h, C.c_void_p(packed_ptr), C.c_void_p(block_ptr),
                                                  L.ptr(win), L.ptr(out), C.byref(info)), allow_empty=True)
        fit = out[:info.n_params].copy() if st == L.OK else np.zeros(0)
        return st, win, fit, info

    def step_finish_enqueue(self, packed_ptr, block_ptr, slot=0):
        self._chk(self._lib.lsqr_step_finish_enqueue(self._h, C.c_void_p(packed_ptr), C.c_void_p(block_ptr), slot))

    def step_finish_wait(self, slot=0):
        win = np.zeros(max(self.P, 1))
        out = np.zeros(max(self.P, 64))
        info = L.RansacInfo()
        st = self._chk(self._lib.lsqr_step_finish_wait(self._h, slot, L.ptr(win), L.ptr(out), C.byref(info)),
                       allow_empty=True)
        fit = out[:info.n_params].copy() if st == L.OK else np.zeros(0)
        return st, win, fit, info

    def residuals(self, params, begin=0, end=None):
        """the model's residual of every record in [begin, end) (lsqr_residuals)"""
        end = self.n if end is None else end
        p = np.ascontiguousarray(params, dtype=np.float64)
        out = np.zeros(end - begin)
        self._chk(self._lib.lsqr_residuals(self._h, L.ptr(p), begin, end, L.ptr(out)))
        return out

    # ---- whole path ---------------------------------------------------------------------
    def ransac(self, p, seed=1, subsets=None, want_consensus=True):
        out = np.zeros(max(self.P, 32))
        cons = np.zeros(max(self.n, 1), dtype=np.uint8) if want_consensus else None
        info = L.RansacInfo()
        s = None
        ns = 0
        if subsets is not None:
            s = np.ascontiguousarray(subsets, dtype=np.uint32).reshape(-1, self.K)
            ns = s.shape[0]
        st = self._lib.lsqr_ransac(self._h, float(p), seed, L.ptr(s), ns, L.ptr(out), L.ptr(cons),
                                   C.byref(info))
        if st == L.ERR_INVALID and info.iterations == 0 and info.fraction == 0:
            return dict(status=st, fraction=0.0, params=None, consensus=None, info=info)
        self._chk(st, allow_empty=True)
        return dict(status=st, fraction=info.fraction,
                    params=out[:info.n_params].copy() if st == L.OK else np.zeros(0),
                    consensus=cons[:self.n] if (cons is not None and info.best_votes > 0) else None,
                    info=info)

    def batch_fit(self, seed, first, H, want_consensus=False):
        """One fixed-size batch end to end on the device (lsqr_batch_fit): winner of hypotheses
        [first, first + H) of the sampler stream, its consensus set, the final fit."""
        out = np.zeros(max(self.P, 32))
        cons = np.zeros(max(self.n, 1), dtype=np.uint8) if want_consensus else None
        info = L.RansacInfo()
        st = self._chk(self._lib.lsqr_batch_fit(self._h, seed, first, H, L.ptr(out), L.ptr(cons),
                                                C.byref(info)), allow_empty=True)
        return dict(status=st, fraction=info.fraction,
                    params=out[:info.n_params].copy() if st == L.OK else np.zeros(0),
                    consensus=cons[:self.n] if cons is not None else None, info=info)

    def batch_fit_enqueue(self, seed, first, H, slot=0):
        """chain a whole batch on the stream and return (lsqr_batch_fit_enqueue); closed-form fits only"""
        self._chk(self._lib.lsqr_batch_fit_enqueue(self._h, seed, first, H, slot))

    def batch_fit_wait(self, slot=0):
        """-> the dict of batch_fit() (without consensus) for the batch enqueued in `slot`"""
        out = np.zeros(max(self.P, 32))
        info = L.RansacInfo()
        st = self._chk(self._lib.lsqr_batch_fit_wait(self._h, slot, L.ptr(out), C.byref(info)),
                       allow_empty=True)
        return dict(status=st, fraction=info.fraction,
                    params=out[:info.n_params].copy() if st == L.OK else np.zeros(0), consensus=None,
                    info=info)

    def ransac_exhaustive(self, want_consensus=True):
        out = np.zeros(max(self.P, 32))
        cons = np.zeros(max(self.n, 1), dtype=np.uint8) if want_consensus else None
        info = L.RansacInfo()
        st = self._chk(self._lib.lsqr_ransac_exhaustive(self._h, L.ptr(out), L.ptr(cons),
                                                        C.byref(info)), allow_empty=True)
        return dict(status=st, fraction=info.fraction,
                    params=out[:info.n_params].copy() if st == L.OK else np.zeros(0),
                    consensus=cons[:self.n] if (cons is not None and info.best_votes > 0) else None,
                    info=info)

    def set_option(self, name, value):
        self._chk(self._lib.lsqr_set_option(self._h, name.encode(), int(value)))

    # ---- measurement --------------------------------------------------------------------
    def profile(self, on=True):
        self._chk(self._lib.lsqr_profile_enable(self._h, int(on)))
        self._chk(self._lib.lsqr_profile_reset(self._h))

    def profile_get(self, name):
        n = C.c_uint64(0)
        ms = C.c_double(0)
        self._chk(self._lib.lsqr_profile_get(self._h, L.KERNEL_IDS[name], C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def index_info(self):
        """-> dict(built, observations, cells, cell_points) of the spatial index (scan_index)"""
        out = (C.c_uint64 * 4)()
        self._chk(self._lib.lsqr_index_info(self._h, out))
        return {"built": bool(out[0]), "observations": int(out[1]), "cells": int(out[2]),
                "cell_points": int(out[3])}

    def scan_workload(self, want_bounds=False):
        """level 1 of the two-level scan alone over the current batch (lsqr_scan_workload) ->
        dict(pairs, level1_evaluations, cells, cell_points[, bounds])"""
        H = self._lib.lsqr_num_hypotheses(self._h)
        ub = np.zeros(H, dtype=np.uint32) if want_bounds else None
        out = (C.c_uint64 * 8)()
        self._chk(self._lib.lsqr_scan_workload(self._h, L.ptr(ub), out))
        r = {"pairs": int(out[0]), "level1_evaluations": int(out[1]), "cells": int(out[2]),
             "cell_points": int(out[3]), "bounded": bool(out[4]), "pilots": int(out[5]),
             "second_pass": int(out[6]), "pairs_counted": int(out[7])}
        if want_bounds:
            r["bounds"] = ub
        return r

    def scan_work(self):
        """what the last scan of the current batch evaluated (lsqr_scan_work): models without a spatial index"""
        out = (C.c_uint64 * 6)()
        self._chk(self._lib.lsqr_scan_work(self._h, out))
        return {"early_exit": bool(out[0]), "row_hypothesis_pairs": int(out[1]), "row_hypothesis_pairs_all": int(out[2]),
                "candidates": int(out[3]), "dropped_first": int(out[4]), "alive_at_end": int(out[5])}

    def synchronize(self):
        self._chk(self._lib.lsqr_synchronize(self._h))


class MultiContext:
    """Several devices from one process (lsqr_multi_*): the hypothesis stream sharded over one lsqr_ctx per entry
    of `devices`, exchanges by peer copies.  devices=[0, 0] puts two contexts on one GPU (tests)."""

    def __init__(self, devices):
        self._lib = L.load()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        st = self._lib.lsqr_multi_create(arr, len(devices), C.byref(h))
        if st != L.OK:
            raise L.LsqrError(st, "lsqr_multi_create(%s): %s" % (list(devices), self._lib.lsqr_status_string(st).decode()))
        self._h = h
        self.size = len(devices)
        self.cfg = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lsqr_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, st, allow_empty=False):
        if st == L.OK or (allow_empty and st == L.EMPTY):
            return st
        raise L.LsqrError(st, "%s (%s)" % (self._lib.lsqr_status_string(st).decode(),
                                           self._lib.lsqr_multi_last_error(self._h).decode()))

    def transport(self):
        """-> ("peer-copy" | "rccl", seconds spent bringing the RCCL communicators up)"""
        t = C.c_double(0.0)
        name = self._lib.lsqr_multi_transport(self._h, C.byref(t))
        return name.decode(), t.value

    def set_model(self, model, dim=3, delta=0.5, ls_type=L.LS_GEOMETRIC, aux=0.0):
        self.cfg = L.ModelCfg(int(model), int(dim), float(delta), int(ls_type), 0, float(aux))
        self._chk(self._lib.lsqr_multi_set_model(self._h, C.byref(self.cfg)))
        self.P = self._lib.lsqr_num_params(C.byref(self.cfg))
        return self

    def upload(self, data):
        a = np.ascontiguousarray(data, dtype=np.float64)
        a = a.reshape(-1, a.shape[-1])
        self._chk(self._lib.lsqr_multi_upload(self._h, L.ptr(a), a.shape[0], a.shape[1] * 8))
        self.n = a.shape[0]
        return self

    def set_option(self, name, value):
        for r in range(self.size):
            c = self._lib.lsqr_multi_ctx(self._h, r)
            st = self._lib.lsqr_set_option(C.c_void_p(c), name.encode(), int(value))
            if st != L.OK:
                raise L.LsqrError(st, "lsqr_set_option(%s)" % name)

    def batch_fit(self, seed, first, H, want_consensus=False):
        out = np.zeros(max(self.P, 64))
        cons = np.zeros(max(self.n, 1), dtype=np.uint8) if want_consensus else None
        info = L.RansacInfo()
        st = self._chk(self._lib.lsqr_multi_batch_fit(self._h, seed, first, H, L.ptr(out), L.ptr(cons),
                                                      C.byref(info)), allow_empty=True)
        return dict(status=st, fraction=info.fraction,
                    params=out[:info.n_params].copy() if st == L.OK else np.zeros(0),
                    consensus=cons[:self.n] if cons is not None else None, info=info)

    def ransac(self, p, seed=1, want_consensus=True):
        out = np.zeros(max(self.P, 64))
        cons = np.zeros(max(self.n, 1), dtype=np.uint8) if want_consensus else None
        info = L.RansacInfo()
        st = self._lib.lsqr_multi_ransac(self._h, float(p), seed, L.ptr(out), L.ptr(cons), C.byref(info))
        if st == L.ERR_INVALID and info.iterations == 0:
            return dict(status=st, fraction=0.0, params=None, consensus=None, info=info)
        self._chk(st, allow_empty=True)
        return dict(status=st, fraction=info.fraction,
                    params=out[:info.n_params].copy() if st == L.OK else np.zeros(0),
                    consensus=cons[:self.n] if (cons is not None and info.best_votes > 0) else None, info=info)


def replay(n, k, p, subsets, valid, votes, dedup=True):
    """Host replay of RANSAC.hxx:49-117 over one batch (exposed for tests)."""
    lib = L.load()
    st = (C.c_uint64 * 6)()
    lib.lsqr_replay_init(n, k, p, st)
    s = np.ascontiguousarray(subsets, dtype=np.uint32)
    v = np.ascontiguousarray(valid, dtype=np.uint8)
    c = np.ascontiguousarray(votes, dtype=np.uint32)
    d = lib.lsqr_dedup_create(k) if dedup else None
    used = lib.lsqr_replay(n, k, p, L.ptr(s), L.ptr(v), L.ptr(c), len(c), 0, d, st)
    if d:
        lib.lsqr_dedup_destroy(d)
    return dict(used=used, i=st[0], num_tries=st[1], best_votes=st[2], best_index=st[3],
                has_best=bool(st[4]), done=bool(st[5]))
