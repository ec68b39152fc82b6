"""Diagnostic: run the stamps build of the library on one workload and print per-phase cycle shares."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_with_flops_amd import hipspgemm as hs, synth
hs.LIB_PATH = os.path.join(os.path.dirname(hs.LIB_PATH), "libspgemm_hip_stamps.so")
wl = sys.argv[1] if len(sys.argv) > 1 else "256k"
m, seed, base = {"256k": (262144, 42, 2), "1m": (1 << 20, 43, 2), "1m32": (1 << 20, 44, 4)}[wl]
rp, ci, v = synth.powerlaw_csr(m, seed, base)
h = hs.Handle(0)
A = hs.CSR.from_arrays(rp, ci, v, m, m).toGpuCSR()
for it in range(3):
    dC = hs.gpuSpMMWrapper(A, A, h); dC.deviceDispose()
buf = (C.c_ulonglong * 64)()
hs.lib().spgemm_hip_debug_stamps(buf, 1)
dC = hs.gpuSpMMWrapper(A, A, h); dC.deviceDispose()
hs.lib().spgemm_hip_debug_stamps(buf, 0)
a = np.array(buf[:], dtype=np.float64).reshape(4, 16)
names = {0: ("k_num_bighash", ["next_row+meta", "clear", "walk-rest", "spill stream", "compact", "r:search", "r:gather", "r:looptail", "w:stage", "r:insert(+park)", "w:endsync"]) if m > 262144 else ("k_num_big", ["next_row+meta", "bitmap", "prefix", "JC store+sync", "clear acc", "products-rest", "store C", "emit sync-wait", "w:stage", "w:rounds", "w:endsync", "emit bitloops"]),
         1: ("k_num_hash<1>", ["loop", "meta", "clear", "products-rest", "compact", "r:search", "r:gather", "r:looptail", "w:stage", "r:insert", "w:endsync"]),
         2: ("k_num_hash<4>", ["loop", "next_row+meta", "clear", "products-rest", "compact", "r:search", "r:gather", "r:looptail", "w:stage", "r:insert", "w:endsync"]),
         3: ("k_num_hash<8>", ["loop", "next_row+meta", "clear", "products-rest", "compact", "r:search", "r:gather", "r:looptail", "w:stage", "r:insert", "w:endsync"])}
print(h.stats()["ms_kernel"])
for k, (nm, ph) in names.items():
    tot = a[k].sum()
    if tot == 0: continue
    print(nm, "total Mcycles(sum over blocks)", round(tot / 1e6, 1), {p: f"{100 * a[k][i] / tot:.1f}%" for i, p in enumerate(ph) if p})
