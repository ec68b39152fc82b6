"""Experiment: order the >4096-product rows largest-first (LPT) for the work queue; compare kernel times."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_matrix_with_flops_amd import hipspgemm as hs, synth
m, seed = (1 << 20, 43)
rp, ci, v = synth.powerlaw_csr(m, seed, 2)[:3]
h = hs.Handle(0)
A = hs.CSR.from_arrays(rp, ci, v, m, m).toGpuCSR()
hv, hvlen, ids, fl, tot = hs.gpuFlopsClassify(A, A, h)
rowIds = hs.d2h(ids, m, np.int32)
lens = np.diff(rp).astype(np.int64)
f = np.zeros(m, np.int64); nz = lens > 0
f[nz] = np.add.reduceat(lens[ci], rp[:-1][nz])
nbig = int((f > 4096).sum())
print("big rows", nbig, "order ok", np.all(f[rowIds[-nbig:]] > 4096))
def run(order_ids, tag):
    fo = f[order_ids]
    dfl = np.zeros(m + 1, np.int32); dfl[1:] = np.cumsum(fo).astype(np.int32)        # int[m+1], wraps like the reference's
    d_ids, d_fl = hs.h2d(order_ids.astype(np.int32)), hs.h2d(dfl)
    best = None
    for it in range(6):
        dC = hs.sgpuSpMMWrapper(A, A, d_ids, hv, d_fl, h); nn = dC.nnz; dC.deviceDispose()
        st = h.stats()
        if it >= 2:
            k = st["ms_kernel"]
            cur = (st["ms_total"], k.get("k_num_bighash"), k.get("k_sym_big"))
            best = cur if best is None or cur[0] < best[0] else best
    print(tag, "nnz", nn, "ms_total %.4f bighash %.4f sym_big %.4f" % best)
    hs.dev_free(d_ids); hs.dev_free(d_fl)
run(rowIds.copy(), "row order   ")
srt = rowIds.copy(); tail = srt[-nbig:]; srt[-nbig:] = tail[np.argsort(-f[tail], kind="stable")]
run(srt, "largest first")
srt2 = rowIds.copy(); srt2[-nbig:] = tail[np.argsort(f[tail], kind="stable")]
run(srt2, "smallest first")
