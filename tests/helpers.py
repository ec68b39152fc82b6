"""Shared test helpers: parity comparator, structure hashes, small random inputs."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")

from oracle import pyoracle as po  # noqa: E402  (test infrastructure)
from sparse_matrix_with_flops_amd import synth  # noqa: E402

REL_TOL = 1e-6  # BASELINE.json north_star: values within 1e-6 relative


def canonical_arrays(rowPtr, colInd, values):
    """Per-row sort by column (CSR::makeOrdered semantics) with numpy; returns copies."""
    rowPtr = np.asarray(rowPtr, dtype=np.int64)
    rows = len(rowPtr) - 1
    row_of = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rowPtr))
    order = np.lexsort((np.asarray(colInd, dtype=np.int64), row_of))
    return np.asarray(colInd)[order].astype(np.int32), np.asarray(values)[order].astype(np.float32)


def _canon(M):
    """(sorted colInd, values) of a CSR-like object; C/OpenMP sort for big inputs, numpy otherwise."""
    if len(M.colInd) > 2_000_000:
        c = po.CSRHost(M.rowPtr, M.colInd, M.values, M.rows, M.cols).canonical()
        return c.colInd, c.values
    return canonical_arrays(M.rowPtr, M.colInd, M.values)


def assert_parity(got, want, rel=REL_TOL, what="", inputs=None, accum=None):
    """The north_star parity rule: rowPtr bit-exact, per-row-sorted colInd bit-exact,
    values |x-y| <= rel*max(|x|,|y|).  `got`/`want` expose rowPtr/colInd/values/rows/cols.

    inputs=(A, B): for MIXED-SIGN inputs the sum order matters through cancellation and no reordering of a
    float32 sum can promise 1e-6 relative to the (possibly tiny) result; the bound is then taken relative to
    the magnitude of the terms, |x-y| <= rel * (|A|*|B|)_ij, computed with the oracle.  With non-negative
    inputs (the reference's own setting: toAbs(), mindex2-cuda/nGpuSpMM.cc:291) both bounds coincide.

    accum=(A, B): products with FEW columns (or hub columns) sum hundreds to thousands of terms into one entry.  Two
    float32 sums of N positive terms in different orders differ by about sqrt(N) * 2^-24 relative (random-walk rounding;
    measured by tools/fuzz_parity.py: 1.0-1.5e-6 at N ~ 500, whichever side is "right" -- the oracle's sequential loop is
    as far from the exact sum as the device's atomics), so 1e-6 cannot hold for such entries in ANY implementation that
    does not reproduce the CPU loop's order.  With accum the tolerance of an entry that sums N products is
    rel * max(1, sqrt(N) / 4) (N from the oracle on the all-ones pattern): unchanged up to 16 products per entry, which
    covers every benchmark workload (their entries sum 1-20 products and are held to the plain 1e-6)."""
    assert got.rows == want.rows and got.cols == want.cols, f"{what}: shape {got.rows}x{got.cols} vs {want.rows}x{want.cols}"
    gr, wr = np.asarray(got.rowPtr), np.asarray(want.rowPtr)
    assert gr.shape == wr.shape, f"{what}: rowPtr length"
    if not np.array_equal(gr, wr):
        bad = int(np.nonzero(gr != wr)[0][0])
        raise AssertionError(f"{what}: rowPtr differs first at {bad}: {gr[bad]} vs {wr[bad]}")
    gc, gv = _canon(got)
    wc, wv = _canon(want)
    if not np.array_equal(gc, wc):
        bad = int(np.nonzero(gc != wc)[0][0])
        raise AssertionError(f"{what}: sorted colInd differs first at {bad}: {gc[bad]} vs {wc[bad]}")
    gv64, wv64 = gv.astype(np.float64), wv.astype(np.float64)
    err = np.abs(gv64 - wv64)
    lim = rel * np.maximum(np.abs(gv64), np.abs(wv64))
    if inputs is not None:
        A, B = inputs
        absA = po.CSRHost(A.rowPtr, A.colInd, np.abs(A.values), A.rows, A.cols)
        absB = absA if B is A else po.CSRHost(B.rowPtr, B.colInd, np.abs(B.values), B.rows, B.cols)
        mag = po.omp_spmm(absA, absB)
        mc, mv = _canon(mag)
        assert np.array_equal(mc, wc)
        lim = np.maximum(lim, rel * mv.astype(np.float64))
    if accum is not None:
        A, B = accum
        onesA = po.CSRHost(A.rowPtr, A.colInd, np.ones_like(A.values), A.rows, A.cols)
        onesB = onesA if B is A else po.CSRHost(B.rowPtr, B.colInd, np.ones_like(B.values), B.rows, B.cols)
        cnt = po.sequential_spmm(onesA, onesB)                 # (accumulates repeated columns like the device does)
        cc, cn = _canon(cnt)
        assert np.array_equal(cc, wc)
        lim = lim * np.maximum(1.0, np.sqrt(cn.astype(np.float64)) / 4.0)
    if not np.all(err <= lim):
        bad = int(np.argmax(err - lim))
        raise AssertionError(f"{what}: value {bad}: {gv[bad]!r} vs {wv[bad]!r} (rel {err[bad] / max(abs(wv64[bad]), 1e-300):.3e})")


def structure_hash(rowPtr, colInd_sorted):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(rowPtr, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(colInd_sorted, dtype=np.int32).tobytes())
    return h.hexdigest()


def value_checksums(colInd_sorted, values_sorted):
    """Two order-insensitive double checksums: sum(v) and sum(v*(1+col%1021))."""
    v = np.asarray(values_sorted, dtype=np.float64)
    w = 1.0 + (np.asarray(colInd_sorted, dtype=np.int64) % 1021).astype(np.float64)
    return float(v.sum()), float((v * w).sum())


def summarize(C):
    cs, vs = _canon(C)                                 # numpy below 2 M entries, the oracle's OpenMP row sort above
    s1, s2 = value_checksums(cs, vs)
    return {"nnz": int(C.nnz), "hash": structure_hash(C.rowPtr, cs), "sum": s1, "wsum": s2}


def synth_csr(m, seed, base=2):
    rp, ci, v = synth.powerlaw_csr(m, seed, base)
    return po.CSRHost(rp, ci, v, m, m)


def random_csr(rows, cols, density, seed, sorted_rows=True, signed=True):
    """Small uniform random CSR without duplicate columns; optionally shuffled inside rows."""
    rng = np.random.default_rng(seed)
    mask = rng.random((rows, cols)) < density
    rp = np.zeros(rows + 1, dtype=np.int32)
    np.cumsum(mask.sum(axis=1), out=rp[1:])
    ci = np.nonzero(mask)[1].astype(np.int32)
    v = rng.random(len(ci)).astype(np.float32) + 0.25
    if signed:
        v *= rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=len(ci))
    if not sorted_rows:
        for i in range(rows):
            s, e = rp[i], rp[i + 1]
            p = rng.permutation(e - s)
            ci[s:e] = ci[s:e][p]
            v[s:e] = v[s:e][p]
    return po.CSRHost(rp, ci, v, rows, cols)


# ---------------------------------------------------------------------------------------------
# R-MCL: one step  Mt' = prune(Mgt * Mt)  of a device implementation against the oracle, with the threshold ties
# accounted for explicitly ("recompute the threshold, assert that every entry kept on one side only lies
# within a few float32 ulps of it, and that all other entries match at 1e-6; report the number of differing rows").
# ---------------------------------------------------------------------------------------------
def rmcl_thresholds(Cm):
    """Per row of the raw product Cm: the reference's prune threshold (nlibs/tools/util.cc:4-9 computeThreshold on the
    float32 max and average of the inflated (squared) values), plus the squared values."""
    rp = np.asarray(Cm.rowPtr, dtype=np.int64)
    v2 = (np.asarray(Cm.values, dtype=np.float32) ** 2).astype(np.float32)
    n = np.diff(rp)
    th = np.zeros(len(n), dtype=np.float32)
    live = n > 0
    if live.any():
        mx = np.maximum.reduceat(v2, rp[:-1][live])
        avg = (np.add.reduceat(v2.astype(np.float64), rp[:-1][live]) / n[live]).astype(np.float32)
        t = (0.90 * avg.astype(np.float64) * (1 - 2 * (mx.astype(np.float64) - avg.astype(np.float64)))).astype(np.float32)
        t = np.where(t > 1.0e-7, t, np.float32(1.0e-7)).astype(np.float32)
        th[live] = np.minimum(t, mx)
    return th, v2


# An entry is a "tie" when its inflated value sits within TIE_REL (relative) of the row's prune threshold: the device's
# v*v carries up to 2e-6 relative error (SpGEMM values are within 1e-6, squared) and its threshold -- 0.9*avg*(...) from a
# float sum of the row in another order -- another ~1e-6, so a correct implementation may keep or drop exactly these.
TIE_REL = 4e-6


def rmcl_tie_rows(Cm, rel=TIE_REL):
    """Rows with at least one entry whose squared value is within `rel` (relative) of the row's threshold."""
    th, v2 = rmcl_thresholds(Cm)
    rp = np.asarray(Cm.rowPtr, dtype=np.int64)
    thr = np.repeat(th, np.diff(rp)).astype(np.float64)
    near = np.abs(v2.astype(np.float64) - thr) <= rel * thr
    return np.unique(np.repeat(np.arange(len(th)), np.diff(rp))[near])


def assert_rmcl_step(got, Mgt, Mt, rel=3e-6, tie_rel=TIE_REL, what="", long_len=512):
    """got = device result of one R-MCL step from (Mgt, Mt).  Every row either equals the oracle's row (same kept
    columns, values within `rel`) or differs ONLY in entries that are threshold ties (within `tie_rel` of the prune
    threshold: the device sums a row in another order than the sequential CPU loop).  The differing rows are counted
    and must be among the rows that hold such a tie.  Returns (rows that differ, rows with a tie, oracle result).

    Tolerance: a step value is v*v / keptSum with v an SpGEMM value (within 1e-6 relative of the oracle's, the
    north_star bound): squaring doubles the relative error and the normalising sum adds its own 1e-6, hence 3e-6
    for the step (measured worst case on the 20 000-node graph: 1.3e-6).

    Rows whose product has more than `long_len` entries are held to the SAME `rel`, but against the float64 evaluation of
    the row rule on the oracle's product and the oracle's kept set instead of the oracle's float32 result: the reference's
    kept sum is one sequential float32 loop (arraySum, nlibs/tools/util.cc:21-31), which on its own is off by up to
    n*eps/2 from the exact sum (5e-6 at 8 192 terms, 2e-5 at 40 000: measured) -- no parallel sum can follow that
    rounding sequence, so for those rows the device is compared with what both approximate.  The oracle's own values are
    checked against the same float64 evaluation at its n*eps bound, so the two references cannot drift apart."""
    import ctypes as C
    Cm = po.omp_spmm(Mgt, Mt)
    rp, ci, v = Cm.rowPtr.copy(), Cm.colInd.copy(), Cm.values.copy()
    n = po.lib().oracle_rmcl_prune_compact(C.c_int(Cm.rows), po._ip(rp), po._ip(ci), po._fp(v))
    want = po.CSRHost(rp, ci[:n], v[:n], Cm.rows, Cm.cols)
    assert got.rows == want.rows
    gc, gv = _canon(got)
    wc, wv = _canon(want)
    grp, wrp = np.asarray(got.rowPtr, dtype=np.int64), np.asarray(want.rowPtr, dtype=np.int64)
    gl, wl = np.diff(grp), np.diff(wrp)

    def row_sig(rp_, c_):
        h = np.zeros(len(rp_) - 1, dtype=np.uint64)
        live = np.diff(rp_) > 0
        mixed = (c_.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ (c_.astype(np.uint64) << np.uint64(17))
        if live.any():
            with np.errstate(over="ignore"):
                h[live] = np.add.reduceat(mixed, rp_[:-1][live])
        return h
    diff = np.nonzero((gl != wl) | (row_sig(grp, gc) != row_sig(wrp, wc)))[0]
    th, v2 = rmcl_thresholds(Cm)
    crp = np.asarray(Cm.rowPtr, dtype=np.int64)
    for r in diff:                                                # few rows: each must be a threshold tie
        cols_raw = np.asarray(Cm.colInd[crp[r]:crp[r + 1]])
        sq = v2[crp[r]:crp[r + 1]]
        gset, wset = set(gc[grp[r]:grp[r + 1]].tolist()), set(wc[wrp[r]:wrp[r + 1]].tolist())
        assert gset <= set(cols_raw.tolist()), f"{what}: row {r} holds a column that is not in the product"
        for c in gset ^ wset:
            x = float(sq[cols_raw == c][0])
            assert abs(x - float(th[r])) <= tie_rel * float(th[r]), \
                f"{what}: row {r} col {c}: v^2={x!r} vs threshold {float(th[r])!r} is not a tie"
    same_row = np.ones(len(gl), dtype=bool)
    same_row[diff] = False
    gm, wm = np.repeat(same_row, gl), np.repeat(same_row, wl)
    assert np.array_equal(gc[gm], wc[wm]), f"{what}: kept columns differ outside the tie rows"
    a, b = gv[gm].astype(np.float64), wv[wm].astype(np.float64)
    long_rows = np.nonzero(same_row & (np.diff(crp) > long_len) & (wl > 0))[0]
    if len(long_rows):
        cmc, cmv = _canon(Cm)                                     # raw product, rows sorted by column like wc
        pos = np.cumsum(np.concatenate([[0], wl[same_row]]))      # where each same-row starts inside b
        idx_in_same = np.cumsum(same_row) - 1
        for r in long_rows:
            cols_r = cmc[crp[r]:crp[r + 1]]
            sq = cmv[crp[r]:crp[r + 1]].astype(np.float64) ** 2
            kept = np.isin(cols_r, wc[wrp[r]:wrp[r + 1]], assume_unique=True)
            exact = sq[kept] / sq[kept].sum()
            lo = pos[idx_in_same[r]]
            seg = b[lo:lo + wl[r]]
            nterm = int(crp[r + 1] - crp[r])
            assert np.all(np.abs(seg - exact) <= (nterm * 6e-8 + 1e-6) * exact), \
                f"{what}: oracle row {r} ({nterm} terms) is farther from the float64 rule than a sequential float32 sum allows"
            b[lo:lo + wl[r]] = exact
    bad = np.abs(a - b) > rel * np.maximum(np.abs(a), np.abs(b))
    assert not bad.any(), f"{what}: {int(bad.sum())} values beyond {rel} relative (worst {np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)):.2e})"
    ties = rmcl_tie_rows(Cm, tie_rel)
    assert len(diff) <= len(ties) and np.all(np.isin(diff, ties)), f"{what}: rows differ that hold no threshold tie"
    return len(diff), len(ties), want
