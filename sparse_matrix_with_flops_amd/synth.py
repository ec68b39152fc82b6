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


def bytes_alg(m, nnzA, P, nnzC):
    """BYTES_ALG of SURVEY.md §8(d): 8(m+1) + 16 nnzA + 8 P + 8 nnzC."""
    return 8 * (m + 1) + 16 * nnzA + 8 * P + 8 * nnzC
