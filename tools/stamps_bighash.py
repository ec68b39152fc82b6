"""Diagnostic: where a block of k_num_bighash spends its time (wave 0, s_memtime stamps around the phases of a row).
Needs the stamped build:  make -C sparse_matrix_with_flops_amd/csrc variant VAR_NAME=stamps VAR_FLAGS=-DSMF_STAMPS
    SPGEMM_LIB=.../libspgemm_hip_stamps.so python tools/stamps_bighash.py [workload]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from sparse_matrix_with_flops_amd import hipspgemm as hs, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "synth_1m_16"
m, seed, base = {"synth_1m_16": (1 << 20, 43, 2), "synth_1m_32": (1 << 20, 44, 4)}[wl]
rp, ci, v = synth.powerlaw_csr(m, seed, base)
A = hs.CSR.from_arrays(rp, ci, v, m, m).toGpuCSR()
h = hs.Handle(0)
L = hs.lib()
out = (C.c_ulonglong * 16)()
for _ in range(2):
    hs.gpuSpMMWrapper(A, A, h).deviceDispose()
assert L.spgemm_hip_debug_stamps(out) == 0            # reset
N = 5
for _ in range(N):
    hs.gpuSpMMWrapper(A, A, h).deviceDispose()
assert L.spgemm_hip_debug_stamps(out) == 0
s = [int(x) for x in out]
names = ["dequeue + metadata of the next row", "table clear + barrier", "walk of pass 0 (total)", "  of it: class + insert loop (callback)",
         "  of it: parking of the other classes (callback)", "streaming passes over parked pairs (incl. inserts)", "emission sweep + barrier", "end of row (check + barrier)"]
tot = s[8]
print(f"{wl}: {s[9] // N} blocks, wave 0 of each: {tot / s[9] / 100e6 * 1e3:.3f} ms at the 100 MHz s_memtime clock per launch")
walk_rest = s[2] - s[3] - s[4]
for i, n in enumerate(names):
    print(f"  {n:55s} {100.0 * s[i] / tot:5.1f} %")
print(f"  {'  of it: staging, unit list, claims, gather wait, barriers':55s} {100.0 * walk_rest / tot:5.1f} %")
print(f"  {'unaccounted':55s} {100.0 * (tot - s[0] - s[1] - s[2] - s[5] - s[6] - s[7]) / tot:5.1f} %")
