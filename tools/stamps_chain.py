"""Diagnostic: where a wave of k_wbatch (chain_device.hpp) spends its time: wave 0 of every block, cycle stamps around the phases
of a batch.  Needs the stamped build:  make -C sparse_matrix_with_flops_amd/csrc variant VAR_NAME=stamps VAR_FLAGS=-DSMF_STAMPS
    SPGEMM_PATH=1|2 SPGEMM_LIB=.../libspgemm_hip_stamps.so python tools/stamps_chain.py [workload]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparse_matrix_with_flops_amd import hipspgemm as hs, synth

wl = sys.argv[1] if len(sys.argv) > 1 else "synth_1m_16"
if wl == "web_google_surrogate":
    m = 916428
    rp, ci, v = synth.webgraph_csr(m, 46)
else:
    m, seed, base = {"synth_1m_16": (1 << 20, 43, 2), "synth_1m_32": (1 << 20, 44, 4), "synth_256k_16": (1 << 18, 42, 2)}[wl]
    rp, ci, v = synth.powerlaw_csr(m, seed, base)
A = hs.CSR.from_arrays(rp, ci, v, m, m).toGpuCSR()
h = hs.Handle(0)
L = hs.lib()
out = (C.c_ulonglong * 16)()
for _ in range(2):
    hs.gpuSpMMWrapper(A, A, h).deviceDispose()
assert L.spgemm_hip_debug_chain_stamps(out) == 0            # reset
N = 5
for _ in range(N):
    hs.gpuSpMMWrapper(A, A, h).deviceDispose()
assert L.spgemm_hip_debug_chain_stamps(out) == 0
s = [int(x) for x in out]
names = ["round top: ticket (+2 barriers in chain mode)", "rows -> regions (batchStart, rowFlops, IA loads; scans)",
         "the product walk (staging, gathers, inserts)", "occupancy sweep, step scan, ranks (sym: + IC store, clear)",
         "chain: barrier, publish, look-back", "chain: barrier after the look-back, rowPtr", "emission + clear", "-"]
tot = s[8]
print(f"{wl} SPGEMM_PATH={os.environ.get('SPGEMM_PATH')}: {s[9] // N} block launches; wave 0 of each: "
      f"{tot / max(s[9], 1) / 1e3:.1f} k cycles per launch")
for i, n in enumerate(names):
    if n != "-":
        print(f"  {n:70s} {100.0 * s[i] / tot:5.1f} %")
print(f"  {'unaccounted':70s} {100.0 * (tot - sum(s[:8])) / tot:5.1f} %")
