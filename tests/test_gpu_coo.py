"""GPU (-m gpu): COO -> CSR on the device (hip_coo_to_csr: sort, duplicate summing, self loops, row normalisation, abs)
against the CPU oracle's restatement of COO::makeOrdered / orderedAndDuplicatesRemoving / toCSR / addSelfLoopIfNeeded /
averAndNormRowQValue and of rmclInit -- bit for bit, floats included (duplicates are summed in input order on both
sides) -- and against the CSRs the real reference loads from its own fixture files (tests/golden/fixtures.npz)."""
import os

import numpy as np
import pytest

from helpers import DATA, GOLDEN, po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs

pytestmark = pytest.mark.gpu
FX = np.load(os.path.join(GOLDEN, "fixtures.npz"))


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    assert hs.device_count() >= 1


def bits_equal(dev, want):
    got = dev.toCpuCSR()
    dev.deviceDispose()
    assert got.rows == want.rows and got.nnz == want.nnz
    assert np.array_equal(got.rowPtr, want.rowPtr) and np.array_equal(got.colInd, want.colInd)
    assert np.array_equal(got.values.view(np.uint32), np.asarray(want.values, np.float32).view(np.uint32))


def random_coo(rows, cols, nnz, seed, dup_frac=0.2, signed=True):
    rng = np.random.default_rng(seed)
    ri = rng.integers(0, rows, size=nnz).astype(np.int32)
    ci = rng.integers(0, cols, size=nnz).astype(np.int32)
    ndup = int(nnz * dup_frac)
    src = rng.integers(0, nnz, size=ndup)                       # repeat some (row,col) pairs, several times for a few
    dst = rng.integers(0, nnz, size=ndup)
    ri[dst], ci[dst] = ri[src], ci[src]
    v = (rng.random(nnz) + 0.25).astype(np.float32)
    if signed:
        v *= rng.choice(np.array([-1.0, 1.0], np.float32), size=nnz)
    return ri, ci, v


@pytest.mark.parametrize("rows,cols,nnz,seed", [(1, 1, 1, 1), (7, 5, 40, 2), (300, 300, 5000, 3), (5000, 70000, 200000, 4),
                                                (200000, 200000, 3000000, 5)])
@pytest.mark.parametrize("dedupe", [True, False])
def test_sort_dedupe_to_csr_matches_oracle(rows, cols, nnz, seed, dedupe):
    ri, ci, v = random_coo(rows, cols, nnz, seed)
    want = po.coo_to_csr(rows, cols, ri, ci, v, dedupe=dedupe)
    bits_equal(hs.coo_to_csr(rows, cols, ri, ci, v, hs.COO_DEDUPE if dedupe else 0), want)


def test_abs_flag_and_empty_input():
    ri, ci, v = random_coo(50, 60, 400, 9)
    bits_equal(hs.coo_to_csr(50, 60, ri, ci, v, hs.COO_DEDUPE | hs.COO_ABS), po.coo_to_csr(50, 60, ri, ci, v, dedupe=True, toAbs=True))
    z = np.zeros(0, np.int32)
    E = hs.coo_to_csr(4, 4, z, z, np.zeros(0, np.float32), hs.COO_DEDUPE)
    e = E.toCpuCSR()
    E.deviceDispose()
    assert e.nnz == 0 and np.array_equal(e.rowPtr, np.zeros(5, np.int32))


@pytest.mark.parametrize("m,seed", [(2000, 11), (60000, 12)])
def test_rmcl_init_on_device_matches_oracle(m, seed):
    """rmclInit = self loops for rows without a diagonal entry + sort (no dedupe) + toCSR + 1/count values."""
    A = synth_csr(m, seed, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    perm = np.random.default_rng(seed).permutation(A.nnz)                  # the loader sees edges in file order
    ri, ci, v = ri[perm], A.colInd[perm], np.ones(A.nnz, np.float32)
    want = po.rmcl_init(A.rows, A.cols, ri, ci, v)
    bits_equal(hs.coo_to_csr(A.rows, A.cols, ri, ci, v, hs.COO_SELF_LOOPS | hs.COO_ROW_NORMALISE), want)


@pytest.mark.parametrize("name", ["test.mtx", "test2.mtx", "own_dups.mtx", "own_sym.mtx", "own_pattern.mtx", "own_graph.snap", "t2.snap"])
def test_reference_fixture_files(name):
    """COO as the loader parses it (oracle text reader) -> device COO->CSR == the CSR the real reference loaded."""
    rows, cols, ri, ci, v = po.read_snap(os.path.join(DATA, name), False)
    key = name.replace(".", "_") + "_load"
    want = po.CSRHost(FX[key + "_rowPtr"], FX[key + "_colInd"], FX[key + "_values"], *[int(x) for x in FX[key + "_shape"]])
    bits_equal(hs.coo_to_csr(rows, cols, ri, ci, v, hs.COO_DEDUPE), want)


def test_out_of_range_entry_is_an_error():
    with pytest.raises(hs.SpgemmError):
        hs.coo_to_csr(3, 3, np.array([0, 5], np.int32), np.array([1, 1], np.int32), np.ones(2, np.float32), hs.COO_DEDUPE)


@pytest.mark.parametrize("m,seed", [(3000, 21), (262144, 42)])
def test_flops_stats_match_oracle(m, seed):
    """hip_flopsStats == the oracle's restatement of flopsStats (nlibs/tools/stats.cc:45-55)."""
    A = synth_csr(m, seed, 2)
    dA = hs.CSR.from_arrays(A.rowPtr, A.colInd, A.values, A.rows, A.cols).toGpuCSR()
    got = hs.flopsStats(dA, dA)
    dA.deviceDispose()
    assert got == [int(x) for x in po.flops_stats(A, A)] and sum(got) == m


def test_row_length_stats_and_per_bin_report():
    """CSR::nnzStats on the device (18 power-of-two buckets of the row lengths) and the per-bin
    GPU-vs-CPU report of resultsComparison: clean on a correct result, and it names the right bin and row when one
    entry of the result is corrupted."""
    A = synth_csr(20000, 33, 2)
    hA = hs.CSR.from_arrays(A.rowPtr, A.colInd, A.values, A.rows, A.cols)
    dA = hA.toGpuCSR()
    got = hs.nnzStats(dA)
    lens = np.diff(A.rowPtr).astype(np.int64)
    want = np.zeros(18, dtype=np.int64)
    for b in range(18):
        lo = 0 if b == 0 else (1 << (b - 1)) + 1
        want[b] = np.sum((lens >= (lo if b else -1)) & (lens <= (1 << b))) if b < 17 else np.sum(lens > (1 << 16))
    want[0] = np.sum(lens <= 1)
    assert got == [int(x) for x in want] and sum(got) == A.rows
    # per-bin report
    hv, hv_len, ids, fl, tot = hs.gpuFlopsClassify(dA, dA)
    queue = hs.d2h(ids, A.rows, np.int32)
    dC = hs.gpuSpMMWrapper(dA, dA)
    hC = dC.toCpuCSR()
    dC.deviceDispose()
    ref = po.omp_spmm(A, A)
    rep = hs.resultsComparison(hC, ref, hv[:hv_len], queue)
    flops = po.row_flops(A, A)
    assert sum(r["rows"] for r in rep) == A.rows and all(r["rows_differ"] == 0 for r in rep)
    assert max(r["max_rel_err"] for r in rep) <= 1e-6
    assert rep[6]["rows"] == int(np.sum((flops > 64) & (flops <= 512)))          # reference bin "65-512 products"
    bad_row = int(np.nonzero((flops > 64) & (flops <= 512))[0][5])
    vals = hC.values.copy()
    vals[hC.rowPtr[bad_row]] *= 1.01
    broken = po.CSRHost(hC.rowPtr, hC.colInd, vals, hC.rows, hC.cols)
    rep2 = hs.resultsComparison(broken, ref, hv[:hv_len], queue)
    assert [r["rows_differ"] for r in rep2] == [0, 0, 0, 0, 0, 0, 1, 0] and rep2[6]["first_bad_row"] == bad_row
    for p in (ids, fl):
        hs.dev_free(p)
    dA.deviceDispose()
