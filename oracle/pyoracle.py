"""ctypes front-end of the CPU oracle (liboracle.so) and of oracle/_ref (the reference's own
RANSAC.hxx).  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by lsqrrecipes_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

PLANE, SPHERE, LINE, DENSE, US_SINGLE, US_POINTER, ABSOR, PIVOT, RAY, LINE2D, PHANTOM = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
LS_ALGEBRAIC, LS_GEOMETRIC = 0, 1


class Cfg(C.Structure):
    _fields_ = [("model", C.c_int), ("dim", C.c_int), ("delta", C.c_double), ("ls_type", C.c_int),
                ("aux", C.c_double)]


class Trace(C.Structure):
    _fields_ = [("cap", C.c_size_t), ("iters", C.c_size_t), ("evaluated", C.c_size_t),
                ("votes", C.POINTER(C.c_uint32)), ("status", C.POINTER(C.c_uint8)),
                ("num_tries", C.POINTER(C.c_uint32)), ("subsets", C.POINTER(C.c_uint32)),
                ("best_iter", C.c_size_t), ("best_votes", C.c_uint32)]


class LcgState(C.Structure):
    _fields_ = [("s", C.c_uint64)]


class RefSampler(C.Structure):
    _fields_ = [("rand_fn", C.c_void_p), ("rand_ctx", C.c_void_p), ("not_chosen", C.c_void_p)]


class ListSampler(C.Structure):
    _fields_ = [("subsets", C.c_void_p), ("count", C.c_size_t), ("pos", C.c_size_t)]


class CtrSampler(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("next_index", C.c_uint64)]


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(so)
            for f in ("linalg.c", "estimators.c", "ransac.c", "lsqr_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "libref_ransac.so")
    if os.path.exists("/root/reference/parametersEstimators/RANSAC.hxx") and (
            force or not os.path.exists(ref)
            or os.path.getmtime(os.path.join(_HERE, "ref_driver.cxx")) > os.path.getmtime(ref)
            or os.path.getmtime(so) > os.path.getmtime(ref)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L = _lib
        L.orc_min_subset.argtypes = [C.POINTER(Cfg)]
        L.orc_num_params.argtypes = [C.POINTER(Cfg)]
        L.orc_record_doubles.argtypes = [C.POINTER(Cfg)]
        L.orc_estimate.argtypes = [C.POINTER(Cfg), C.POINTER(_dp), C.c_size_t, _dp]
        L.orc_agree.argtypes = [C.POINTER(Cfg), _dp, _dp]
        L.orc_ls_masked.argtypes = [C.POINTER(Cfg), _dp, C.c_size_t, C.c_size_t, C.c_void_p, _dp]
        L.orc_scan.argtypes = [C.POINTER(Cfg), _dp, _dp, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_scan.restype = C.c_size_t
        L.orc_stats.argtypes = [C.POINTER(Cfg), _dp, _dp, C.c_size_t, C.c_size_t, C.c_void_p, _dp]
        L.orc_ransac.argtypes = [C.POINTER(Cfg), _dp, C.c_size_t, C.c_size_t, C.c_double,
                                 C.c_void_p, C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int),
                                 C.c_void_p, C.POINTER(Trace)]
        L.orc_ransac.restype = C.c_double
        L.orc_ransac_exhaustive.argtypes = [C.POINTER(Cfg), _dp, C.c_size_t, C.c_size_t, _dp,
                                            C.POINTER(C.c_int), C.c_void_p]
        L.orc_ransac_exhaustive.restype = C.c_double
        L.orc_choose.argtypes = [C.c_uint, C.c_uint]
        L.orc_choose.restype = C.c_uint
        L.orc_ctr_subset.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, C.c_void_p]
        L.orc_sym_eig.argtypes = [C.c_int, _dp, _dp, _dp]
        L.orc_svd.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_pinv_solve.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, _dp]
        L.orc_sphere_algebraic.argtypes = [C.c_int, C.POINTER(_dp), C.c_size_t, _dp]
        L.orc_sphere_geometric.argtypes = [C.c_int, C.POINTER(_dp), C.c_size_t, _dp, _dp,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_us_analytic.argtypes = [C.c_int, C.POINTER(_dp), C.c_size_t, _dp]
        L.orc_us_fcn.argtypes = [C.c_int, C.POINTER(_dp), C.c_size_t, _dp, _dp, _dp, C.c_int]
        L.orc_us_fcn.restype = None
        L.orc_absor_weighted_ls.argtypes = [C.POINTER(_dp), _dp, C.c_size_t, _dp]
        L.orc_us_iterative.argtypes = [C.c_int, C.POINTER(_dp), C.c_size_t, _dp, _dp,
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _lib


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_ransac.so"))


def ref():
    global _ref
    if _ref is None:
        build()
        _ref = C.CDLL(os.path.join(_HERE, "_ref", "libref_ransac.so"))
        _ref.ref_ransac.argtypes = [C.POINTER(Cfg), _dp, C.c_size_t, C.c_double, C.c_int,
                                    C.c_uint64, _dp, C.POINTER(C.c_int), C.c_void_p,
                                    C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t]
        _ref.ref_ransac.restype = C.c_double
    return _ref


def _d(a):
    return a.ctypes.data_as(_dp)


def cfg(model, dim=3, delta=0.5, ls_type=LS_GEOMETRIC, aux=0.0):
    return Cfg(model, dim, float(delta), ls_type, float(aux))


def as_records(c, data):
    a = np.ascontiguousarray(data, dtype=np.float64)
    nd = lib().orc_record_doubles(C.byref(c))
    a = a.reshape(-1, nd)
    return a


def _ptrs(rows):
    arr = (_dp * len(rows))()
    for i, r in enumerate(rows):
        arr[i] = _d(r)
    return arr


def estimate(c, records):
    """records: (k, nd) array in draw order -> params array (len 0 on degenerate)."""
    a = as_records(c, records)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    out = np.zeros(80)
    n = lib().orc_estimate(C.byref(c), _ptrs(rows), len(rows), _d(out))
    return out[:n].copy()


def agree(c, params, record):
    p = np.ascontiguousarray(params, dtype=np.float64)
    r = np.ascontiguousarray(record, dtype=np.float64)
    return bool(lib().orc_agree(C.byref(c), _d(p), _d(r)))


def scan(c, params, data):
    a = as_records(c, data)
    p = np.ascontiguousarray(params, dtype=np.float64)
    mask = np.zeros(a.shape[0], dtype=np.uint8)
    cnt = lib().orc_scan(C.byref(c), _d(p), _d(a), a.shape[0], a.shape[1], mask.ctypes.data)
    return int(cnt), mask


def scan_many(c, params_rows, valid, data, threads=None):
    """vote counts of MANY hypotheses over the same records: orc_scan per hypothesis (the serial agree() loop of
    RANSAC.hxx:94-99 without the exit), dealt to host threads (ctypes releases the GIL; every call is independent).
    params_rows: (H, >= n_params) array; valid: (H,) -- invalid hypotheses get 0.  -> uint32 (H,)"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    a = as_records(c, data)
    P = np.ascontiguousarray(params_rows, dtype=np.float64)
    H = P.shape[0]
    out = np.zeros(H, dtype=np.uint32)
    L = lib()
    threads = threads or max(1, min(32, (os.cpu_count() or 2)))

    def work(k):
        for h in range(k, H, threads):
            if valid[h]:
                out[h] = L.orc_scan(C.byref(c), _d(P[h]), _d(a), a.shape[0], a.shape[1], None)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(threads)))
    return out


def ls(c, data, mask=None):
    a = as_records(c, data)
    out = np.zeros(80)
    m = None
    if mask is not None:
        m = np.ascontiguousarray(mask, dtype=np.uint8)
    n = lib().orc_ls_masked(C.byref(c), _d(a), a.shape[0], a.shape[1],
                            m.ctypes.data if m is not None else None, _d(out))
    return out[:n].copy()


def stats(c, params, data, mask=None):
    a = as_records(c, data)
    p = np.ascontiguousarray(params, dtype=np.float64)
    out = np.zeros(4)
    m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    lib().orc_stats(C.byref(c), _d(p), _d(a), a.shape[0], a.shape[1],
                    m.ctypes.data if m is not None else None, _d(out))
    return out


def ctr_subset(seed, h, n, k):
    out = np.zeros(k, dtype=np.uint32)
    lib().orc_ctr_subset(seed, h, n, k, out.ctypes.data)
    return out


def ctr_subsets(seed, first, count, n, k):
    out = np.zeros((count, k), dtype=np.uint32)
    L = lib()
    for i in range(count):
        L.orc_ctr_subset(seed, first + i, n, k, out[i].ctypes.data)
    return out


def ransac(c, data, p, sampler="ref", seed=1, subsets=None, first=0, full_scan=False,
           trace_cap=4096):
    """Run the restated RANSAC.hxx.  sampler: 'ref' (rand() formula fed by the LCG seeded with
    `seed`), 'list' (explicit subsets, draw order), 'ctr' (product's counter sampler)."""
    L = lib()
    a = as_records(c, data)
    n = a.shape[0]
    k = L.orc_min_subset(C.byref(c))
    params = np.zeros(80)
    nparams = C.c_int(0)
    cons = np.zeros(max(n, 1), dtype=np.uint8)
    votes = np.zeros(trace_cap, dtype=np.uint32)
    status = np.zeros(trace_cap, dtype=np.uint8)
    ntries = np.zeros(trace_cap, dtype=np.uint32)
    subs = np.zeros((trace_cap, k), dtype=np.uint32)
    tr = Trace(trace_cap, 0, 0, votes.ctypes.data_as(C.POINTER(C.c_uint32)),
               status.ctypes.data_as(C.POINTER(C.c_uint8)),
               ntries.ctypes.data_as(C.POINTER(C.c_uint32)),
               subs.ctypes.data_as(C.POINTER(C.c_uint32)), 0, 0)
    keep = []
    if sampler == "ref":
        st = LcgState(seed)
        nc = np.zeros(max(n, 1), dtype=np.uint8)
        s = RefSampler(C.cast(L.orc_lcg_rand, C.c_void_p), C.cast(C.pointer(st), C.c_void_p),
                       nc.ctypes.data)
        fn = C.cast(L.orc_ref_sampler_next, C.c_void_p)
        keep = [st, nc]
    elif sampler == "list":
        sl = np.ascontiguousarray(subsets, dtype=np.uint32).reshape(-1, k)
        s = ListSampler(sl.ctypes.data, sl.shape[0], 0)
        fn = C.cast(L.orc_list_sampler_next, C.c_void_p)
        keep = [sl]
    else:
        s = CtrSampler(seed, first)
        fn = C.cast(L.orc_ctr_sampler_next, C.c_void_p)
    frac = L.orc_ransac(C.byref(c), _d(a), n, a.shape[1], float(p), fn,
                        C.cast(C.pointer(s), C.c_void_p), int(full_scan), _d(params),
                        C.byref(nparams), cons.ctypes.data, C.byref(tr))
    it = min(tr.iters, trace_cap)
    del keep
    return dict(fraction=frac, params=params[:nparams.value].copy(), consensus=cons[:n].copy(),
                iters=tr.iters, evaluated=tr.evaluated, votes=votes[:it].copy(),
                status=status[:it].copy(), num_tries=ntries[:it].copy(),
                subsets=subs[:it].copy(), best_iter=tr.best_iter, best_votes=tr.best_votes)


def ransac_exhaustive(c, data):
    L = lib()
    a = as_records(c, data)
    n = a.shape[0]
    params = np.zeros(80)
    nparams = C.c_int(0)
    cons = np.zeros(max(n, 1), dtype=np.uint8)
    frac = L.orc_ransac_exhaustive(C.byref(c), _d(a), n, a.shape[1], _d(params),
                                   C.byref(nparams), cons.ctypes.data)
    return dict(fraction=frac, params=params[:nparams.value].copy(), consensus=cons[:n].copy())


def ref_ransac(c, data, p, seed=1, exhaustive=False, prefill=None, subsets_cap=0):
    """Run the REFERENCE's RANSAC.hxx (oracle/_ref) with rand() fed by the LCG."""
    R = ref()
    a = as_records(c, data)
    n = a.shape[0]
    k = lib().orc_min_subset(C.byref(c))
    params = np.zeros(80)
    nparams = C.c_int(0)
    if prefill is not None:
        params[:len(prefill)] = prefill
        nparams = C.c_int(len(prefill))
    cons = np.zeros(max(n, 1), dtype=np.uint8)
    st = (C.c_uint64 * 4)()
    subs = np.zeros((max(subsets_cap, 1), k), dtype=np.uint32)
    frac = R.ref_ransac(C.byref(c), _d(a), n, float(p), int(exhaustive), seed, _d(params),
                        C.byref(nparams), cons.ctypes.data, st,
                        subs.ctypes.data if subsets_cap else None, subsets_cap)
    ne = int(st[0])
    return dict(fraction=frac, params=params[:nparams.value].copy(), consensus=cons[:n].copy(),
                estimate_calls=ne, agree_calls=int(st[1]), ls_calls=int(st[2]),
                rand_calls=int(st[3]), subsets=subs[:min(ne, subsets_cap)].copy())


def sym_eig(A):
    A = np.array(A, dtype=np.float64, order="C")
    n = A.shape[0]
    w = np.zeros(n)
    V = np.zeros((n, n))
    lib().orc_sym_eig(n, _d(A), _d(w), _d(V))
    return w, V


def svd(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, n = A.shape
    U = np.zeros((m, n))
    s = np.zeros(n)
    V = np.zeros((n, n))
    lib().orc_svd(m, n, _d(A), _d(U), _d(s), _d(V))
    return U, s, V


def pinv_solve(A, b, tol):
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(A.shape[1])
    rank = lib().orc_pinv_solve(A.shape[0], A.shape[1], _d(A), _d(b), tol, _d(x))
    return x, rank


def sphere_algebraic(dim, pts):
    a = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, dim)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    out = np.zeros(dim + 1)
    n = lib().orc_sphere_algebraic(dim, _ptrs(rows), len(rows), _d(out))
    return out[:n].copy()


def sphere_geometric(dim, pts, init):
    a = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, dim)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    init = np.ascontiguousarray(init, dtype=np.float64)
    out = np.zeros(dim + 1)
    info, nfev = C.c_int(0), C.c_int(0)
    n = lib().orc_sphere_geometric(dim, _ptrs(rows), len(rows), _d(init), _d(out),
                                   C.byref(info), C.byref(nfev))
    return out[:n].copy(), info.value, nfev.value


def absor_weighted_ls(pairs, weights):
    """AbsoluteOrientationParametersEstimator::weightedLeastSquaresEstimate (.cxx:208-291)"""
    a = np.ascontiguousarray(pairs, dtype=np.float64).reshape(-1, 6)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    out = np.zeros(7)
    n = lib().orc_absor_weighted_ls(_ptrs(rows), _d(w), len(rows), _d(out))
    return out[:n].copy()


def us_analytic(model, recs):
    nd = 15 if model == US_SINGLE else 18
    a = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, nd)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    out = np.zeros(20)
    n = lib().orc_us_analytic(model, _ptrs(rows), len(rows), _d(out))
    return out[:n].copy()


def us_iterative(model, recs, init):
    nd = 15 if model == US_SINGLE else 18
    a = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, nd)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    init = np.ascontiguousarray(init, dtype=np.float64)
    out = np.zeros(20)
    info, nfev = C.c_int(0), C.c_int(0)
    lib().orc_us_iterative(model, _ptrs(rows), len(rows), _d(init), _d(out),
                           C.byref(info), C.byref(nfev))
    # the last iterate is returned even when the reference would report failure (info not in 1..4)
    return out[:(20 if model == US_SINGLE else 17)].copy(), info.value, nfev.value


class UsFunction:
    """residual vector f(x) and Jacobian J(x) of the US calibration (SinglePointTarget...cxx:415-658 / :1059-1286)
    over a fixed set of frames, for an external minimiser"""

    def __init__(self, model, recs):
        nd = 15 if model == US_SINGLE else 18
        self.model = model
        self.a = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, nd)
        self.rows = [self.a[i] for i in range(self.a.shape[0])]
        self.ptrs = _ptrs(self.rows)
        self.m = self.a.shape[0]
        self.np = 11 if model == US_SINGLE else 8

    def f(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(self.m)
        lib().orc_us_fcn(self.model, self.ptrs, self.m, _d(x), _d(out), None, 1)
        return out

    def jac(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros((self.m, self.np))
        lib().orc_us_fcn(self.model, self.ptrs, self.m, _d(x), None, _d(out), 2)
        return out


def phantom_analytic(recs):
    a = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, 15)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    out = np.zeros(41)
    n = lib().orc_phantom_analytic(_ptrs(rows), len(rows), _d(out))
    return out[:n].copy()


def phantom_iterative(recs, init):
    a = np.ascontiguousarray(recs, dtype=np.float64).reshape(-1, 15)
    rows = [np.ascontiguousarray(a[i]) for i in range(a.shape[0])]
    init = np.ascontiguousarray(init, dtype=np.float64)
    out = np.zeros(41)
    info, nfev = C.c_int(0), C.c_int(0)
    lib().orc_phantom_iterative(_ptrs(rows), len(rows), _d(init), _d(out), C.byref(info),
                                C.byref(nfev))
    return out.copy(), info.value, nfev.value
