"""GPU (-m gpu): the R-MCL caller of the hot path — device prune step + gpuRmclIter — against the
oracle's seqRmclIter restatement and the golden R-MCL results made from the real reference; plus the C++ mirror
driven by the reference-style test program tests/cpp/testGpuSpMM.cc."""
import os
import subprocess

import numpy as np
import pytest

from helpers import DATA, GOLDEN, ROOT, canonical_arrays, po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs
from test_gpu_parity import FX, SQUARE, to_hs, unpack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    assert hs.device_count() >= 1


def ordered(M):
    c, v = canonical_arrays(M.rowPtr, M.colInd, M.values)
    return np.asarray(M.rowPtr), c, v


@pytest.mark.parametrize("name", SQUARE)
@pytest.mark.parametrize("iters", [1, 3])
def test_rmcl_fixtures_match_reference_goldens(name, iters):
    Mt = po.load(os.path.join(DATA, name), isTrans=True, mode=1)          # RMCL() reads the transpose
    got = hs.gpuRmclIter(iters, to_hs(Mt), to_hs(Mt))
    want = unpack(FX, f"{name.replace('.', '_')}_rmcl{iters}")            # RMCL(file, iters, SEQ) of the reference
    gr, gc, gv = ordered(got)
    wr, wc, wv = ordered(want)
    assert np.array_equal(gr, wr) and np.array_equal(gc, wc)
    assert np.allclose(gv, wv, rtol=1e-5, atol=1e-7)


def test_rmcl_synthetic_graph_three_iterations():
    """Power-law graph, 3 iterations.  The prune threshold comes from float sums whose order differs between the
    CPU loop and the device reduction, so an entry within an ulp of the threshold may be kept on one side only:
    rows may differ in a vanishing fraction, everything else must agree."""
    A = synth_csr(20000, 91, 2)
    ones = po.CSRHost(A.rowPtr, A.colInd, np.ones_like(A.values), A.rows, A.cols)
    rows, cols, ri = A.rows, A.cols, np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    Mt = po.rmcl_init(rows, cols, A.colInd, ri, ones.values)              # transpose + self loops + row-normalise
    got = hs.gpuRmclIter(3, to_hs(Mt), to_hs(Mt))
    want = po.rmcl_iters(Mt, Mt, 3)
    gl, wl = np.diff(got.rowPtr), np.diff(want.rowPtr)
    assert np.mean(gl != wl) < 1e-3
    assert abs(int(got.nnz) - int(want.nnz)) <= max(20, want.nnz // 2000)
    same = np.nonzero(gl == wl)[0][:2000]
    for r in same:
        g0, w0 = got.rowPtr[r], want.rowPtr[r]
        gc = np.sort(got.colInd[g0:g0 + gl[r]])
        wc = np.sort(want.colInd[w0:w0 + wl[r]])
        if np.array_equal(gc, wc):
            gv = got.values[g0:g0 + gl[r]][np.argsort(got.colInd[g0:g0 + gl[r]])]
            wv = want.values[w0:w0 + wl[r]][np.argsort(want.colInd[w0:w0 + wl[r]])]
            assert np.allclose(gv, wv, rtol=1e-4, atol=1e-7)
    rs = np.add.reduceat(got.values, got.rowPtr[:-1][gl > 0])
    assert np.allclose(rs, 1.0, atol=1e-4)                                # every row is a distribution again


def test_sharded_rmcl_single_rank_matches_gpuRmclIter():
    """dist.ShardedRMCL at world size 1 (device-resident loop: hip_gpuSpMM + hip_rmcl_prune per step, torch tensors
    as the replicated Mt) against hip_gpuRmclIter on the same graph."""
    import torch
    from sparse_matrix_with_flops_amd.dist import HipEngine, ShardedRMCL
    A = synth_csr(12000, 57, 2)
    ri = np.repeat(np.arange(A.rows, dtype=np.int32), np.diff(A.rowPtr))
    Mt = po.rmcl_init(A.rows, A.cols, A.colInd, ri, np.ones_like(A.values))
    host = (Mt.rowPtr, Mt.colInd, Mt.values, Mt.rows, Mt.cols)
    job = ShardedRMCL(HipEngine(0), host, host)
    job.iterate(3)
    rp, ci, v = job.result_host()
    torch.cuda.synchronize()
    want = hs.gpuRmclIter(3, to_hs(Mt), to_hs(Mt))
    gl, wl = np.diff(rp), np.diff(want.rowPtr)
    assert np.mean(gl != wl) < 1e-3 and abs(len(ci) - want.nnz) <= max(20, want.nnz // 2000)
    rs = np.add.reduceat(v, rp[:-1][gl > 0])
    assert np.allclose(rs, 1.0, atol=1e-4)
    agree = 0
    for r in np.nonzero(gl == wl)[0][:1000]:          # a threshold tie may swap one entry of a row; most rows are identical
        agree += np.array_equal(np.sort(ci[rp[r]:rp[r + 1]]), np.sort(want.colInd[want.rowPtr[r]:want.rowPtr[r + 1]]))
    assert agree >= 990


def test_cpp_mirror_runs_the_reference_test_protocol():
    """tests/cpp/testGpuSpMM.cc == tests/testGpuSpMM.cc + nrmcl.cc of the reference, built on the C++ mirror."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(ROOT, "tests", "cpp")])
    exe = os.path.join(ROOT, "tests", "cpp", "testGpuSpMM.x")
    for name, extra in (("test2.mtx", []), ("own_graph.snap", ["--rmcl", "3"]), ("t2.snap", ["--rmcl", "3"]),
                        ("own_dups.mtx", [])):
        out = subprocess.run([exe, os.path.join(DATA, name)] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "Differs" not in out.stdout and "Same" in out.stdout
