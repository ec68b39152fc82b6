"""Where an R-MCL iteration spends its time on one GPU (500K-node graph): expansion (hip_gpuSpMM) vs the prune step
(hip_rmcl_prune) vs the copies into the caller's tensors."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs

m = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
A = synth_csr(m, 45, 2)
ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
Mt = po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))
h = hs.Handle(0)
G = hs.CSR.from_arrays(Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols).toGpuCSR()
cur = hs.CSR.from_arrays(Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols).toGpuCSR()
fused = "--two-step" not in sys.argv
for it in range(10):
    t0 = time.perf_counter()
    if fused:
        i_, j_, c_, nn = hs.rmcl_expand_prune_raw(h, G.rowPtr, G.colInd, G.values, G.nnz, cur.rowPtr, cur.colInd, cur.values,
                                                  cur.nnz, G.rows, G.cols, cur.cols)
        t1 = time.perf_counter()
        st = h.stats()
        print(f"iter {it}: nnz(Mt)={cur.nnz:9d} P={st['total_flops']:11d} nnzC={st['nnzC']:10d} fused step {1e3*(t1-t0):7.2f} ms "
              f"(classify {st['ms_classify']:.2f} symbolic {st['ms_symbolic']:.2f} scan {st['ms_scan_alloc']:.2f} numeric {st['ms_numeric']:.2f}) -> nnz {nn}")
        if it: cur.deviceDispose()
        cur = hs.CSR(c_, j_, i_, G.rows, G.cols, nn, True)
        continue
    C_ = hs.gpuSpMMWrapper(G, cur, h)
    t1 = time.perf_counter()
    st = h.stats()
    i_, j_, c_, nn = hs.rmcl_prune_raw(h, C_.rows, C_.rowPtr, C_.colInd, C_.values)
    t2 = time.perf_counter()
    print(f"iter {it}: nnz(Mt)={cur.nnz:9d} P={st['total_flops']:11d} nnzC={C_.nnz:10d} spgemm {1e3*(t1-t0):7.2f} ms (classify {st['ms_classify']:.2f} symbolic {st['ms_symbolic']:.2f} scan {st['ms_scan_alloc']:.2f} numeric {st['ms_numeric']:.2f}) prune {1e3*(t2-t1):6.2f} ms -> nnz {nn}")
    C_.deviceDispose()
    if it: cur.deviceDispose()
    cur = hs.CSR(c_, j_, i_, C_.rows, C_.cols, nn, True)
