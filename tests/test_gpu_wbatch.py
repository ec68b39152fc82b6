"""GPU (-m gpu): the wave-per-batch kernels of csrc/chain_device.hpp (round 4: rows dealt ACROSS rows; opt-in, SPGEMM_PATH).

SPGEMM_PATH=1  rows up to ChainCfg::smallMax products through k_wbatch<sym> / k_wbatch<num> (two passes),
SPGEMM_PATH=2  the same rows in ONE pass: no symbolic pass, a chained prefix over groups of batches places them (k_chain),
the bins above on their per-row kernels either way.  Both were measured SLOWER than the per-row kernels (DESIGN.md §8,
profiles/README.md round 4), so the default stays SPGEMM_PATH=0; they are kept as the record of the experiment and have to
stay correct: same parity rule as every other path, on inputs that exercise what is special about them -- the cut of the rows
into batches (windows of table slots, at most 64 rows per batch), rows above smallMax INSIDE a batch (their entries are
skipped over by offset), batches of only such rows, empty rows, the last partial group of a block, the capacity guard of a
compressive product whose previous call undersized C.
"""
import numpy as np
import pytest

from helpers import assert_parity, po, random_csr, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs
from test_gpu_parity import to_hs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    assert hs.device_count() >= 1


def handle_with(monkeypatch, path, cfg=0):
    monkeypatch.setenv("SPGEMM_PATH", str(path))
    monkeypatch.setenv("SPGEMM_CHAIN_CFG", str(cfg))
    return hs.Handle(0)                                # the handle reads its path when it is made


def mul(h, A, B=None):
    dA = to_hs(A).toGpuCSR()
    dB = dA if B is None else to_hs(B).toGpuCSR()
    dC = hs.gpuSpMMWrapper(dA, dB, h)
    got = dC.toCpuCSR()
    dC.deviceDispose()
    dA.deviceDispose()
    if B is not None:
        dB.deviceDispose()
    return got


@pytest.mark.parametrize("path,cfg", [(1, 0), (2, 0), (1, 2), (2, 2), (2, 1), (2, 3), (1, 4)])
def test_power_law_product_every_bin(monkeypatch, path, cfg):
    h = handle_with(monkeypatch, path, cfg)
    A = synth_csr(120000, 77, 2)                       # rows from 2 to > 4096 products: every bin, counted rows inside batches
    want = po.omp_spmm(A, A)
    for _ in range(2):                                 # twice: tickets, chain words and tables start clean every call
        assert_parity(mul(h, A), want, what=f"path {path} cfg {cfg}")
    st = h.stats()
    names = set(st["ms_kernel"])                       # (per-kernel timing is off: nothing recorded, nothing to look at)
    assert st["nnzC"] == want.nnz and st["total_flops"] == int(po.row_flops(A, A).sum()) and isinstance(names, set)
    h.close()


@pytest.mark.parametrize("path", [1, 2])
def test_awkward_shapes(monkeypatch, path):
    h = handle_with(monkeypatch, path)
    rng = np.random.default_rng(5)
    # rectangular, unsorted rows, mixed signs, empty rows at both ends and in runs longer than a batch
    A = random_csr(3000, 700, 0.01, 11, sorted_rows=False)
    B = random_csr(700, 5000, 0.02, 12, sorted_rows=False)
    rp = A.rowPtr.copy()
    keep = np.ones(A.rows, dtype=bool)
    keep[:200] = False
    keep[1000:1300] = False
    keep[-70:] = False
    lens = np.diff(rp) * keep
    sel = np.repeat(keep, np.diff(rp))
    A2 = po.CSRHost(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), A.colInd[sel], A.values[sel], A.rows, A.cols)
    assert_parity(mul(h, A2, B), po.omp_spmm(A2, B), what=f"path {path}: rectangular with empty runs", inputs=(A2, B))
    # only rows ABOVE smallMax (every batch holds nothing but rows that are not accumulated here), then only tiny rows
    dense = random_csr(400, 400, 0.25, 13, signed=False)
    assert_parity(mul(h, dense), po.omp_spmm(dense, dense), what=f"path {path}: every row above smallMax")
    mt = 70000                                         # 0-3 entries per row: rows of 0-9 products, 64-row batches
    deg = rng.integers(0, 4, size=mt)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    ci = np.concatenate([np.sort(rng.choice(mt, size=d, replace=False)) for d in deg]).astype(np.int32)
    tiny = po.CSRHost(rp, ci, (rng.random(len(ci)) + 0.25).astype(np.float32), mt, mt)
    assert_parity(mul(h, tiny), po.omp_spmm(tiny, tiny), what=f"path {path}: rows of 0-9 products")
    # one row, no rows' worth of products, a single batch
    one = random_csr(1, 50, 0.5, 15, signed=False)
    sq = random_csr(50, 50, 0.2, 16, signed=False)
    assert_parity(mul(h, one, sq), po.omp_spmm(one, sq), what=f"path {path}: one row")
    h.close()


def _two_entry_rows(m, same):
    """A: two entries per row (columns 2i, 2i+1 of B's rows); B: four entries per row.  same=True: B rows 2i and 2i+1 hold the
    SAME four columns (nnz(C) = 4 per row, half the products); False: disjoint columns (8 per row).  m, nnz(A), P identical."""
    rpA = (np.arange(m + 1) * 2).astype(np.int32)
    ciA = np.arange(2 * m, dtype=np.int32)
    vA = np.ones(2 * m, dtype=np.float32)
    A = po.CSRHost(rpA, ciA, vA, m, 2 * m)
    rpB = (np.arange(2 * m + 1) * 4).astype(np.int32)
    base = (np.arange(2 * m) // 2 if same else np.arange(2 * m)) * 4
    ciB = (base[:, None] + np.arange(4)[None, :]).astype(np.int32).ravel() % (8 * m)
    vB = np.full(8 * m, 0.5, dtype=np.float32)
    B = po.CSRHost(rpB, ciB, vB, 2 * m, 8 * m)
    return A, B


def test_one_pass_capacity_guard(monkeypatch):
    """A product that compresses gets, from its second call on, exactly the entries its previous call produced.  A later
    product of the same shape and product count with MORE distinct columns must not write past that: the kernels clamp, say
    so, and the call is redone the two-phase way -- the result is right either way."""
    h = handle_with(monkeypatch, 2)
    m = 50000
    A1, B1 = _two_entry_rows(m, same=True)
    A2, B2 = _two_entry_rows(m, same=False)
    want1, want2 = po.omp_spmm(A1, B1), po.omp_spmm(A2, B2)
    assert want1.nnz == 4 * m and want2.nnz == 8 * m
    for _ in range(3):                                 # third call: exact-size C
        assert_parity(mul(h, A1, B1), want1, what="compressive product, one pass")
    assert_parity(mul(h, A2, B2), want2, what="same shape and P, twice the entries: capacity guard + redo")
    assert_parity(mul(h, A1, B1), want1, what="and back")
    h.close()


@pytest.mark.parametrize("path", [1, 2])
def test_more_than_2_pow_30_products_under_the_opt_in_paths(monkeypatch, path):
    """Beyond 2^30 products C is not sized by P: the opt-in paths, which queue only part of the symbolic pass, hand the call
    to the two-phase pipeline (tests/test_gpu_abi.py has the case; here it runs with the path selected)."""
    monkeypatch.setenv("SPGEMM_PATH", str(path))
    import test_gpu_abi
    test_gpu_abi.test_more_than_2_pow_30_products_take_the_two_phase_path()
