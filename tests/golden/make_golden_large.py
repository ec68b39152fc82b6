#!/usr/bin/env python3
"""Golden summaries for the two BASELINE configs that are too big to store in full (run in the build container):

  configs[3]  synthetic 1 048 576^2, ~32 nnz/row (seed 44, base 4): summary of A*A made by the REAL reference's
              omp_CSR_SpMM (oracle/_ref): nnz, structure hash, value checksums, P, 8-way flops partition.
  configs[4]  R-MCL on the 500 000-node power-law graph (seed 45): per-iteration summaries (nnz, structure hash, value
              checksums, rows whose kept set sits within float32 rounding of the prune threshold) of 10 iterations.
              Every iteration is  Mt <- prune(Mgt * Mt)  with the product by the reference-pinned OpenMP kernel and the
              reference's prune math (oracle_rmcl_prune_compact == nlibs/tools/util.cc:4-69, pinned in
              tests/test_oracle_vs_ref.py); the state after 10 iterations is cross-checked here against
              RMCL(file, 10, OMP) of the real reference run on the same graph written as a t2.snap-style edge list.

    make -C oracle ref && python tests/golden/make_golden_large.py   ->  tests/golden/golden_large.json
"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import po, summarize, synth_csr  # noqa: E402

assert po.have_ref(), "build oracle/_ref first: make -C oracle ref"


def rmcl_graph(m, seed):
    A = synth_csr(m, seed, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    return A, ri, po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))   # transpose + self loops + 1/deg


def tie_rows(Cm):
    """Rows of the raw product whose prune decision is within float32 rounding of the threshold: entries v with
    |v^2 - thresh| <= 4 ulp(thresh).  A device reduction that sums in another order may legitimately flip those."""
    rp = Cm.rowPtr.astype(np.int64)
    v2 = (Cm.values.astype(np.float32) ** 2).astype(np.float32)
    n = np.diff(rp)
    live = n > 0
    mx = np.maximum.reduceat(v2, rp[:-1][live])
    sm = np.add.reduceat(v2.astype(np.float64), rp[:-1][live])
    avg = (sm / n[live]).astype(np.float32)
    th = (0.90 * avg.astype(np.float64) * (1 - 2 * (mx.astype(np.float64) - avg.astype(np.float64)))).astype(np.float32)
    th = np.where(th > 1.0e-7, th, np.float32(1.0e-7)).astype(np.float32)
    th = np.minimum(th, mx)
    thr = np.repeat(th, n[live])
    near = np.abs(v2.astype(np.float64) - thr.astype(np.float64)) <= 4 * np.spacing(thr).astype(np.float64)
    rows = np.repeat(np.nonzero(live)[0], n[live])
    return np.unique(rows[near])


def main():
    out = {}
    t0 = time.time()
    m, seed, base = 1048576, 44, 4
    A = synth_csr(m, seed, base)
    Cm = po.ref_spmm(A, A, "omp")
    pref = po.ref_row_flops_prefix(A, A)
    s = summarize(Cm)
    s.update({"m": m, "seed": seed, "base": base, "nnzA": A.nnz, "P": int(pref[-1]),
              "max_row_flops": int(np.diff(pref).max()), "partition8": [int(x) for x in po.ref_equal_partition64(pref, 8)]})
    out["synth_1048576_44_4"] = s
    print("configs[3]", s["nnz"], s["P"], f"{time.time() - t0:.0f}s")
    del Cm

    m, seed, iters = 500000, 45, 10
    A, ri, Mt = rmcl_graph(m, seed)
    Mgt, cur = Mt, Mt
    per = []
    for it in range(iters):
        Cm = po.omp_spmm(Mgt, cur)
        ties = tie_rows(Cm)
        rp, ci, v = Cm.rowPtr.copy(), Cm.colInd.copy(), Cm.values.copy()
        n = po.lib().oracle_rmcl_prune_compact(C.c_int(Cm.rows), po._ip(rp), po._ip(ci), po._fp(v))
        cur = po.CSRHost(rp, ci[:n], v[:n], Cm.rows, Cm.cols)
        sm = summarize(cur)
        sm.update({"raw_nnz": int(Cm.nnz), "tie_rows": int(len(ties))})
        per.append(sm)
        print("rmcl iter", it + 1, sm["nnz"], sm["raw_nnz"], sm["tie_rows"], f"{time.time() - t0:.0f}s")
    out["rmcl_500000_45"] = {"m": m, "seed": seed, "iters": iters, "nnz0": int(Mt.nnz), "per_iter": per}

    # the real reference on the same graph, once: RMCL(file, 10, OMP)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "g500k.snap")
        with open(path, "w") as f:
            f.write("# synthetic power-law graph (synth.powerlaw_csr 500000, seed 45)\n")
            f.write(f"{m} {A.nnz}\n")
            np.savetxt(f, np.stack([ri, A.colInd], axis=1), fmt="%d")
        R = po.ref_rmcl(path, iters, 1)
    same = (np.array_equal(R.rowPtr, cur.rowPtr) and np.array_equal(R.colInd, cur.colInd)
            and np.array_equal(R.values.view(np.uint32), cur.values.view(np.uint32)))
    out["rmcl_500000_45"]["reference_RMCL_OMP_10_iterations_bit_identical"] = bool(same)
    print("reference RMCL(file, 10, OMP) identical to the stepwise result:", same, f"{time.time() - t0:.0f}s")
    assert same
    with open(os.path.join(HERE, "golden_large.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
