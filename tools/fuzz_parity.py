"""Randomised parity sweep (round 3): many small and medium A*B with awkward shapes through every entry point of the path,
each against the CPU oracle (the restatement of sequential_CSR_SpMM, nlibs/cpu_csr_kernel.cc:76-119 -- the kernel the
north star names.  With repeated column indices inside rows of B the reference's OWN variants disagree: omp_CSR_SpMM emits
a column once per B-row occurrence pattern, e.g. 4 entries in a row of a 3-column product, sequential_CSR_SpMM accumulates;
the reference's drivers dedupe their inputs first, nGpuSpMM.cc:288.  The HIP path follows the sequential kernel.)  Shapes: empty matrices, rows/columns of size 1, empty rows, one dense row among empty ones,
repeated column indices inside rows (allowed at the boundary: the accumulators key on the column), unsorted rows, hub
columns, products that land in every bin.  Entry points: hip_gpuSpMM (one-shot), hip_spgemm_symbolic/numeric (two-phase),
hip_CSR_SpMM (host arrays), classify + hip_sgpuSpMM, sharded job over 2-3 logical shards, hip_rmcl_expand_prune.
    python tools/fuzz_parity.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.zeros(1, device="cuda")
from helpers import assert_parity, assert_rmcl_step, po
from sparse_matrix_with_flops_amd import hipspgemm as hs

def rand_csr(rng, rows, cols, kind):
    """kind: row-length profile"""
    if rows == 0 or cols == 0:
        return po.CSRHost(np.zeros(rows + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), rows, cols), False
    if kind == "sparse":
        lens = rng.poisson(rng.uniform(0.2, 6), rows)
    elif kind == "skew":
        lens = np.minimum((rng.pareto(1.1, rows) * rng.uniform(0.5, 4)).astype(np.int64), cols * 2)
    elif kind == "onebig":
        lens = np.zeros(rows, np.int64); lens[rng.integers(0, rows)] = min(cols * 2, int(rng.integers(1, 6000)))
    elif kind == "dense":
        lens = rng.integers(0, min(cols, 80) + 1, rows)
    else:
        lens = rng.integers(0, 4, rows)
    lens = np.minimum(lens, 20000).astype(np.int64)
    rp = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=rp[1:])
    nnz = int(rp[-1])
    hub = rng.random() < 0.3
    ci = rng.integers(0, cols, nnz)
    if hub and nnz:
        m = rng.random(nnz) < 0.3
        ci[m] = rng.integers(0, max(1, cols // 50 + 1), int(m.sum()))
    dup_ok = rng.random() < 0.5
    if not dup_ok:                                    # make columns unique inside rows (drop repeats)
        key = np.repeat(np.arange(rows, dtype=np.int64), lens) * cols + ci
        _, first = np.unique(key, return_index=True)
        keep = np.zeros(nnz, bool); keep[first] = True
        row_of = np.repeat(np.arange(rows, dtype=np.int64), lens)[keep]
        ci = ci[keep]
        lens = np.bincount(row_of, minlength=rows)
        rp = np.zeros(rows + 1, np.int64); np.cumsum(lens, out=rp[1:])
        nnz = int(rp[-1])
    v = (rng.random(nnz) + 0.25).astype(np.float32)
    return po.CSRHost(rp.astype(np.int32), ci.astype(np.int32), v, rows, cols), bool(dup_ok)

def to_hs(M):
    return hs.CSR.from_arrays(M.rowPtr, M.colInd, M.values, M.rows, M.cols)

def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    h = hs.Handle(0)
    t0 = time.time()
    nprod = 0
    for case in range(cases):
        m = int(rng.choice([0, 1, 2, 7, 63, 64, 65, 300, 2000, 9000]))
        k = int(rng.choice([1, 2, 5, 64, 500, 3000, 20000]))
        n = int(rng.choice([1, 3, 64, 1000, 70000, 300000, 1 << 20]))
        A, dupA = rand_csr(rng, m, k, str(rng.choice(["sparse", "skew", "onebig", "dense", "tiny"])))
        B, dupB = rand_csr(rng, k, n, str(rng.choice(["sparse", "skew", "onebig", "dense", "tiny"])))
        want = po.sequential_spmm(A, B) if m and A.nnz and B.nnz else po.CSRHost(np.zeros(m + 1, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32), m, n)
        tag = f"case {case} (seed {seed}): A {m}x{k} nnz {A.nnz}, B {k}x{n} nnz {B.nnz}, nnzC {want.nnz}"
        hA, hB = to_hs(A), to_hs(B)
        dA, dB = hA.toGpuCSR(), hB.toGpuCSR()
        try:
            dC = hs.gpuSpMMWrapper(dA, dB, h)                                   # one-shot
            got = dC.toCpuCSR(); dC.deviceDispose()
            assert_parity(got, want, what=tag + " [hip_gpuSpMM]", accum=(A, B))
            nprod += h.stats()["total_flops"]
            if case % 3 == 0:                                                    # host arrays in/out
                assert_parity(hA.hip_spmm(hB), want, what=tag + " [hip_CSR_SpMM]", accum=(A, B))
            if case % 4 == 1:                                                    # classification handed back in
                hv, hv_len, ids, fl, tot = hs.gpuFlopsClassify(dA, dB, h)
                dC = hs.sgpuSpMMWrapper(dA, dB, ids, hv, fl, h)
                got = dC.toCpuCSR(); dC.deviceDispose(); hs.dev_free(ids); hs.dev_free(fl)
                assert_parity(got, want, what=tag + " [classify + hip_sgpuSpMM]", accum=(A, B))
            if case % 5 == 2 and m >= 1:                                         # sharded job, logical shards
                shards = int(rng.integers(2, 4))
                g = hs.Group(shards, devices=[0] * shards, transport=int(rng.choice([hs.XCHG_PEER, hs.XCHG_HOST])))
                job = hs.ShardedSpMM(g, hA, hB)
                job.step(True)
                assert_parity(job.result(shards - 1), want, what=tag + f" [sharded x{shards}]", accum=(A, B))
                job.close(); g.close()
            if case % 6 == 3 and m >= 1 and want.nnz and not (dupA or dupB):     # fused R-MCL step on the same product
                pi, pj, pv, nn = hs.rmcl_expand_prune_raw(h, dA.rowPtr, dA.colInd, dA.values, dA.nnz, dB.rowPtr, dB.colInd,
                                                          dB.values, dB.nnz, m, k, n)
                got = po.CSRHost(hs.d2h(pi, m + 1, np.int32), hs.d2h(pj, nn, np.int32), hs.d2h(pv, nn, np.float32), m, n)
                for p_ in (pi, pj, pv): hs.dev_free(p_)
                assert_rmcl_step(got, A, B, what=tag + " [hip_rmcl_expand_prune]")
        finally:
            dA.deviceDispose(); dB.deviceDispose()
        if case % 25 == 24:
            print(f"{case + 1} cases ok, {nprod} products, {time.time() - t0:.0f}s", flush=True)
    print(f"fuzz ok: {cases} cases, seed {seed}, {nprod} products, {time.time() - t0:.0f}s")

if __name__ == "__main__":
    main()
