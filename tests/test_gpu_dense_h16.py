"""The dense system's agree() scan on the fp16 matrix cores (lsqrrecipes_amd/csrc/dense_h16.h; reference:
DenseLinearEquationSystemParametersEstimator.hxx:111-119).  The filter works on two-way fp16 splits of the rows and the
unknowns and decides a pair only when its residual is clear of delta by a stated bound; everything else goes through the
exact fp64 formula.  Votes must therefore equal those of every other path -- fp32 / fp64 matrix-core filters, the plain
fp64 kernel, the oracle -- bit for bit."""
import os
import re
import subprocess

import numpy as np
import pytest

from lsqrrecipes_amd import _lib as L, synth
from lsqrrecipes_amd.context import Context
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _votes(ctx, rows, ncol, delta, subs, filt, f32):
    ctx.set_option("scan_filter", filt)
    ctx.set_option("dense_f32", f32)
    ctx.set_model(L.DENSE, ncol, delta).upload(rows)
    ctx.hypotheses_from_subsets(subs)
    ctx.scan()
    return ctx.hypotheses()


def test_h16_votes_equal_every_other_filter_and_the_oracle():
    """ragged sizes (rows not a multiple of 256, hypotheses not a multiple of 32), degenerate subsets (invalid
    hypotheses), subsets of tiny rows (solutions with |x| in the millions: wide bands), and a delta placed on one
    pair's exact residual (strict '<' through the worklist)"""
    ncol, m, H = 64, 150_003, 301
    rows = synth.dense(m, ncol, 0.1, seed=321, noise=0.01)[0]
    rows[5000:5200, :ncol] *= 1e-6           # minimal solves on these rows blow up
    rows[9000:9010] = 0.0                    # all-zero rows (residual 0: agree with everything)
    g = np.random.default_rng(7)
    subs = O.ctr_subsets(11, 0, H, m, ncol).copy()
    subs[3, 1] = subs[3, 0]                  # duplicate row: singular minimal system
    subs[7] = g.choice(np.arange(5000, 5200), ncol, replace=False)
    subs[8, :32] = g.choice(np.arange(5000, 5200), 32, replace=False)
    with Context(0) as ctx:
        par, valid, votes16 = _votes(ctx, rows, ncol, 0.1, subs, 1, 2)
        # the fp16 filter really ran: the library probes the device's matrix unit first and would say so otherwise
        assert b"fp32 filter used" not in ctx._lib.lsqr_last_error(ctx._h)
        assert valid[3] == 0 or not np.all(np.isfinite(par[3]))
        for filt, f32, name in ((1, 1, "fp32 matrix cores"), (1, 0, "fp64 matrix cores"), (0, 2, "plain fp64 kernel")):
            p2, v2, votes = _votes(ctx, rows, ncol, 0.1, subs, filt, f32)
            assert np.array_equal(v2, valid)
            assert np.array_equal(votes, votes16), name
        oc = O.cfg(O.DENSE, ncol, 0.1)
        for h in (0, 1, 7, 8, 100, 300):     # the oracle on a sample (a full pass per hypothesis on the host)
            if valid[h]:
                assert votes16[h] == O.scan(oc, par[h], rows)[0], h
        # delta exactly on one pair's reference residual, one ulp above and below
        h = int(np.argmax(votes16))
        x = par[h]
        s = 0.0
        for i in range(ncol):
            s += rows[777, i] * x[i]
        rho = abs(s - rows[777, ncol])
        assert rho > 0
        for delta in (rho, np.nextafter(rho, np.inf), np.nextafter(rho, 0.0)):
            _, _, a = _votes(ctx, rows, ncol, delta, subs, 1, 2)
            _, _, b = _votes(ctx, rows, ncol, delta, subs, 0, 2)
            assert np.array_equal(a, b), delta
            assert a[h] == O.scan(O.cfg(O.DENSE, ncol, delta), x, rows)[0]


def test_h16_magnitudes_that_do_not_fit_fall_back():
    """right-hand sides 1e25 times the coefficients: the fp16 scaling cannot hold them, the scan must still be exact
    (the fp32 / fp64 filters take it)"""
    ncol, m = 64, 70_000
    rows = synth.dense(m, ncol, 0.1, seed=5, noise=0.01)[0]
    rows[:, :ncol] *= 1e-12
    rows[:, ncol] *= 1e13
    subs = O.ctr_subsets(3, 0, 64, m, ncol)
    with Context(0) as ctx:
        _, valid, a = _votes(ctx, rows, ncol, 1e12, subs, 1, 2)
        _, _, b = _votes(ctx, rows, ncol, 1e12, subs, 0, 2)
        assert np.array_equal(a, b)


def test_h16_standalone_check_and_error_of_the_matrix_unit():
    """tools/h16_bench (built by __graft_entry__.build()): every vote of 1000 hypotheses x 100 000 rows against a
    brute-force fp64 count, incl. hypotheses the filter cannot take; and the deviation of the matrix unit's residual
    from the exact one, which the thresholds ASSUME to be below 83 u S (dense_h16.h header)"""
    exe = os.path.join(ROOT, "tools", "h16_bench")
    if not os.path.exists(exe):
        pytest.skip("tools/h16_bench not built")
    out = subprocess.run([exe, "100000", "1000", "1", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert re.search(r"votes: 0 of 1000 hypotheses differ", out.stdout), out.stdout
    dev = float(re.search(r"probe: largest \|r'' - res''\| = ([0-9.eE+-]+) u S", out.stdout).group(1))
    assert dev < 20.0, dev   # measured 0.8 - 1.5; the bound in use is 83
    # the alignment of one instruction's products: the probe's worst case must stay below half of what is assumed
    offs = [abs(float(m)) for m in re.findall(r"= ([+-][0-9.]+) u of the sum of magnitudes", out.stdout)]
    assert len(offs) == 8 and max(offs) < 8.5, offs   # 17 terms x 0.5 u; measured 5.5 - 7.6
    # the same on random operands (r05): 64 instructions (what the library runs per context) and 4096; measured 4.0 / 6.0
    rnd = [float(x) for x in re.findall(r"random probe: \d+ instructions x 1024 outputs, worst \|result - exact\| = ([0-9.eE+-]+) u",
                                        out.stdout)]
    assert len(rnd) == 2 and max(rnd) < 8.5, rnd


def test_refused_minimal_systems_are_never_counted_and_do_not_flood_the_worklist():
    """60 of 301 minimal systems hold a row twice: refused, unknowns NaN, |residual| < delta false for every row.  The
    fp16 filter marks them 'never counted' instead of sending 60 x 150 003 pairs to the exact path (the worklist holds 4 M)"""
    ncol, m, H = 64, 150_003, 301
    rows = synth.dense(m, ncol, 0.1, seed=322, noise=0.01)[0]
    subs = O.ctr_subsets(12, 0, H, m, ncol).copy()
    subs[:60, 1] = subs[:60, 0]
    with Context(0) as ctx:
        par, valid, v16 = _votes(ctx, rows, ncol, 0.1, subs, 1, 2)
        msg = ctx._lib.lsqr_last_error(ctx._h)
        assert b"overflow" not in msg and b"fp32 filter used" not in msg, msg
        assert not valid[:60].any() and not v16[:60].any()
        _, v2, vex = _votes(ctx, rows, ncol, 0.1, subs, 0, 2)
        assert np.array_equal(valid, v2) and np.array_equal(v16, vex)
        ctx.set_option("scan_filter", 1)
