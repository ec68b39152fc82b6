"""BASELINE.json configs[4] at 1 GPU: R-MCL (expand A*A + inflate/prune/normalise) on a 500 000-node power-law graph,
10 iterations, through hip_gpuRmclIter.  Prints per-run timing and checks stochasticity of the result; the first two
iterations are compared with the CPU oracle (row lengths / nnz within the threshold-tie tolerance)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs

m = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
A = synth_csr(m, 45, 2)
ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
Mt = po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))      # transpose + self loops + 1/deg
H = hs.CSR.from_arrays(Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols)
hs.gpuRmclIter(1, H, H)                                                        # warm-up (allocator, workspaces)
for iters in (1, 2, 10):
    t0 = time.perf_counter()
    R = hs.gpuRmclIter(iters, H, H)
    dt = time.perf_counter() - t0
    rs = np.add.reduceat(R.values, R.rowPtr[:-1][np.diff(R.rowPtr) > 0])
    print(f"iters={iters:2d} nnz={R.nnz:9d} max_row={np.diff(R.rowPtr).max():5d} wall={dt*1e3:8.1f} ms (incl. H2D/D2H) "
          f"row sums in [{rs.min():.6f},{rs.max():.6f}]")
    if iters == 2:
        W = po.rmcl_iters(Mt, Mt, 2)
        gl, wl = np.diff(R.rowPtr), np.diff(W.rowPtr)
        print(f"   vs CPU oracle after 2 iterations: nnz {R.nnz} vs {W.nnz}; rows with different length: {np.mean(gl != wl):.2e}")
