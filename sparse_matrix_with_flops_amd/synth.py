"""Deterministic synthetic power-law CSR generator.

Integer-only and counter-based, so any language reproduces it bit for bit:

    u64(seed, s, k) = splitmix64(splitmix64(seed ^ s*0x9E3779B97F4A7C15) ^ k)
    degree   t = min(clz64(u64(seed,0,i)), 10); v = base << t; deg_i = v + u64(seed,1,i) % v
    columns  r = u64(seed,2,(i<<20)^e);  r even -> (i + (r>>1)%257 - 128) mod m   (local)
                                          r odd  -> (r>>1) % m                     (global)
             then sort + unique per row
    values   ((u64(seed,3,i*m+col) >> 40) + 1) / 2**24  in (0,1], float32 (positive like toAbs(),
             mindex2-cuda/nGpuSpMM.cc:291)

base=2 gives ~15 nnz/row ("avg 16" configs), base=4 ~30 ("avg 32").
Known instance: seed 42, m=262144, base 2 -> nnzA=3 887 048, P=58 865 303,
nnzC=55 418 390.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """The splitmix64 output function of state x (x may be a uint64 array)."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def u64(seed, s, k):
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed) ^ (np.uint64(s) * _GOLD))
        return splitmix64(base ^ np.asarray(k, dtype=np.uint64))


def _clz_capped(u, cap=10):
    t = np.zeros(u.shape, dtype=np.int64)
    alive = np.ones(u.shape, dtype=bool)
    for k in range(1, cap + 1):
        alive &= (u >> np.uint64(64 - k)) == 0
        t += alive
    return t


def powerlaw_csr(m, seed, base=2):
    """-> (rowPtr int32[m+1], colInd int32[nnz], values float32[nnz]); rows column-sorted, no duplicates."""
    i = np.arange(m, dtype=np.uint64)
    t = _clz_capped(u64(seed, 0, i))
    v = (np.int64(base) << t).astype(np.uint64)
    deg = (v + u64(seed, 1, i) % v).astype(np.int64)
    tot = int(deg.sum())
    start = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(deg, out=start[1:])
    row = np.repeat(np.arange(m, dtype=np.int64), deg)
    e = np.arange(tot, dtype=np.int64) - start[row]
    r = u64(seed, 2, (row.astype(np.uint64) << np.uint64(20)) ^ e.astype(np.uint64))
    h = r >> np.uint64(1)
    local = (row + (h % np.uint64(257)).astype(np.int64) - 128) % m
    glob = (h % np.uint64(m)).astype(np.int64)
    col = np.where((r & np.uint64(1)) == 0, local, glob)
    key = np.unique(row * np.int64(m) + col)
    row_u = key // m
    col_u = key - row_u * m
    rowPtr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(row_u, minlength=m), out=rowPtr[1:])
    val = ((u64(seed, 3, key.astype(np.uint64)) >> np.uint64(40)).astype(np.float64) + 1.0) / float(1 << 24)
    return rowPtr.astype(np.int32), col_u.astype(np.int32), val.astype(np.float32)


# ------------------------------------------------------------------------------------------------
# Web-graph surrogate (BASELINE.json configs[2]: SuiteSparse web-Google, which is in neither container).
# The reference records only totals for that matrix (tools/res.txt:1910: N 916 428, nnzA 5 105 039,
# nnzC 29 710 164, 2P 121 375 672 -> nnzC/P = 0.49, P/nnzA = 11.9, loaded by readSNAPFile(f, false) +
# orderedAndDuplicatesRemoving, mindex2-cuda/nGpuSpMM.cc:285-291).  This generator reproduces that SHAPE, not the
# matrix: same row count, ~5.5 entries per row, skewed in- and out-degree, and -- what the power-law generator
# above lacks -- the COMPRESSIVE regime of real web graphs (about two products per output entry): pages are grouped
# in sites of `W` consecutive ids whose first `N` pages form a densely inter-linked core that the site's other pages
# mostly point to, so the two-hop paths of a row land on the same few core pages again and again.
# "surrogate; reference totals unpinned": parity is checked against the oracle on this same input, the totals of
# the real web-Google stay a known-answer test for whoever supplies the file (tests/test_gpu_parity.py, env-gated).
# Integer-only like powerlaw_csr.
# ------------------------------------------------------------------------------------------------
WEB_M = 916428


def webgraph_csr(m=WEB_M, seed=46, W=128, N=16, cap=10, pnav=200, ploc=28, pnav_core=235, ploc_core=10, core_t=4,
                 gskew=4):
    """-> (rowPtr int32[m+1], colInd int32[nnz], values float32[nnz]); rows column-sorted, no duplicates.

        site(i) = i // W, core page <=> i % W < N
        degree   t = min(clz64(u64(seed,0,i)), cap), core pages t = max(t, core_t); v = 1 << t; deg_i = v + u64(seed,1,i) % v
        entry e  r = u64(seed,2,(i<<20)^e); sel = r & 255; x = (r>>8) & 0xffff; y = (r>>24) & 0xffffff
                 sel < pnav            -> core page  (x*N)>>16 of the row's own site
                 sel < pnav + ploc     -> any page   (x*W)>>16 of the row's own site
                 otherwise             -> core page (x*N)>>16 of site ((y^gskew >> 24(gskew-1)) * nsites) >> 24
                                          (a power of a uniform variate: a few sites collect most outside links)
                 (core rows use pnav_core / ploc_core)
        values   as in powerlaw_csr
    Defaults (m = 916 428, seed 46): nnzA = 5 028 172 (5.49/row), P = 64 697 113, nnz(A*A) = 31 868 431
    (nnzC/P = 0.493), max out-degree 341, max in-degree 5 508.
    """
    i = np.arange(m, dtype=np.uint64)
    core = (i % np.uint64(W)).astype(np.int64) < N
    t = _clz_capped(u64(seed, 0, i), cap)
    t = np.where(core, np.maximum(t, core_t), t)
    v = (np.int64(1) << t).astype(np.uint64)
    deg = (v + u64(seed, 1, i) % v).astype(np.int64)
    tot = int(deg.sum())
    start = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(deg, out=start[1:])
    row = np.repeat(np.arange(m, dtype=np.int64), deg)
    e = np.arange(tot, dtype=np.int64) - start[row]
    r = u64(seed, 2, (row.astype(np.uint64) << np.uint64(20)) ^ e.astype(np.uint64))
    sel = (r & np.uint64(0xff)).astype(np.int64)
    x = ((r >> np.uint64(8)) & np.uint64(0xffff)).astype(np.int64)
    y = ((r >> np.uint64(24)) & np.uint64(0xffffff)).astype(np.int64)
    nsites = (m + W - 1) // W
    nav = (x * N) >> 16
    loc = (x * W) >> 16
    cg = y
    for _ in range(gskew - 1):
        cg = (cg * y) >> 24
    site_g = (cg * nsites) >> 24
    site_l = row // W
    rc = core[row]
    a = np.where(rc, pnav_core, pnav)
    b = a + np.where(rc, ploc_core, ploc)
    col = np.where(sel < a, site_l * W + nav, np.where(sel < b, site_l * W + loc, site_g * W + nav))
    col = np.minimum(col, m - 1)
    key = np.unique(row * np.int64(m) + col)
    row_u = key // m
    col_u = key - row_u * m
    rowPtr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(row_u, minlength=m), out=rowPtr[1:])
    val = ((u64(seed, 3, key.astype(np.uint64)) >> np.uint64(40)).astype(np.float64) + 1.0) / float(1 << 24)
    return rowPtr.astype(np.int32), col_u.astype(np.int32), val.astype(np.float32)


def bytes_alg(m, nnzA, P, nnzC):
    """BYTES_ALG of SURVEY.md §8(d): 8(m+1) + 16 nnzA + 8 P + 8 nnzC."""
    return 8 * (m + 1) + 16 * nnzA + 8 * P + 8 * nnzC
