"""Round-3 probe: how much of a wave-per-row kernel's time is per ROW and how much per 64-product ROUND?
Matrices whose rows all have the same shape: E entries per A row, every B row L long (products = E*L per row, all columns
distinct), m rows.  Kernel times of the symbolic / numeric wave-per-row kernels per row as a function of the rounds."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparse_matrix_with_flops_amd import hipspgemm as hs

def make(m, E, L, n, seed):
    rng = np.random.default_rng(seed)
    # B: m rows x n cols, each row L distinct columns (a random start, stride coprime walk keeps them distinct and spread)
    start = rng.integers(0, n, size=m, dtype=np.int64)
    bcols = (start[:, None] + np.arange(L, dtype=np.int64)[None, :] * 7919) % n
    bcols.sort(axis=1)
    Brp = (np.arange(m + 1, dtype=np.int64) * L).astype(np.int32)
    # A: row i points to E distinct B rows
    acols = (rng.integers(0, m, size=m, dtype=np.int64)[:, None] + np.arange(E, dtype=np.int64)[None, :] * 104729) % m
    acols.sort(axis=1)
    Arp = (np.arange(m + 1, dtype=np.int64) * E).astype(np.int32)
    A = hs.CSR.from_arrays(Arp, acols.reshape(-1).astype(np.int32), rng.random(m * E).astype(np.float32) + 0.5, m, m)
    B = hs.CSR.from_arrays(Brp, bcols.reshape(-1).astype(np.int32), rng.random(m * L).astype(np.float32) + 0.5, m, n)
    return A, B

h = hs.Handle(0)
h.set_kernel_timing(0xFFFFF)
m, n = 200000, 1 << 20
print("E entries/row, L per B row -> products/row, rounds; per-row ns (symbolic, numeric) and kernel names")
for E, L in [(8, 9), (8, 16), (8, 24), (8, 32), (16, 8), (16, 16), (4, 32), (4, 64), (32, 8), (8, 48), (8, 64)]:
    A, B = make(m, E, L, n, 1)
    dA, dB = A.toGpuCSR(), B.toGpuCSR()
    best = {}
    for rep in range(4):
        dC = hs.gpuSpMMWrapper(dA, dB, h)
        st = h.stats()
        dC.deviceDispose()
        for k, v in st["ms_kernel"].items():
            if rep: best[k] = min(best.get(k, 1e9), v)
    P = E * L
    sym = {k: v for k, v in best.items() if k.startswith("k_sym") and v > 0.01}
    num = {k: v for k, v in best.items() if k.startswith("k_num") and v > 0.01}
    print(f"E={E:3d} L={L:3d} P/row={P:4d} rounds={(P + 63) // 64}  nnzC/P={st['nnzC'] / st['total_flops']:.3f}  "
          f"sym {sum(sym.values()) * 1e6 / m:7.1f} ns/row {list(sym)}  num {sum(num.values()) * 1e6 / m:7.1f} ns/row {list(num)}")
    dA.deviceDispose(); dB.deviceDispose()
