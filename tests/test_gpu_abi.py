"""GPU (-m gpu): behaviour of the C ABI around the kernels -- the caching allocator (per device, idle blocks only),
the two-phase protocol (a pending symbolic phase dies with any other use of the handle's workspace) and the
classification scan beyond 2^31 products."""
import ctypes as C

import numpy as np
import pytest

from helpers import po, synth_csr
from sparse_matrix_with_flops_amd import hipspgemm as hs
from test_gpu_parity import to_hs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as ge
    ge.build()
    assert hs.device_count() >= 1


def test_pool_is_per_device_and_trimmed_with_the_last_handle():
    """Blocks freed on a device are cached for THAT device and reused there; destroying the last handle of the device
    gives them back to the driver.  (With one GPU visible the cross-device case cannot run; the keying is by
    hipGetDevice() at allocation and the stored device at release.)"""
    h = hs.Handle(0)
    a = hs.dev_alloc(8 << 20)
    hs.dev_free(a)
    cached = hs.pool_cached_bytes(0)
    assert cached >= (8 << 20)
    b = hs.dev_alloc(8 << 20)                       # comes out of the cache (best fit: this block or an older one)
    assert hs.pool_cached_bytes(0) <= cached - (8 << 20)
    hs.dev_free(b)
    if hs.device_count() > 1:
        assert hs.pool_cached_bytes(1) == 0        # nothing was ever freed on device 1
        h1 = hs.Handle(1)                           # hipSetDevice(1): an allocation now must not get device 0's block
        c = hs.dev_alloc(8 << 20)
        assert c not in (a, b) and hs.pool_cached_bytes(0) >= (8 << 20)
        hs.dev_free(c)
        h1.close()
        hs.lib().spgemm_hip_create(C.byref(C.c_void_p()), 0)   # back to device 0 for the rest of the session
    import gc
    gc.collect()
    others = hs.pool_cached_bytes(0)
    h.close()
    # default handles of other tests may still be alive on device 0; if this was the last one the cache is empty
    assert hs.pool_cached_bytes(0) in (0, others)


def test_entry_points_complete_their_work_before_returning():
    """The pool's invariant: an entry point returns only after its stream work is done, so freeing C right after the
    call and reusing the block cannot race with the kernels that wrote it.  Run SpGEMM, copy C out, free it, let another
    SpGEMM reuse the same blocks, and check the first copy is still the right answer."""
    h = hs.Handle(0)
    A, B = synth_csr(30000, 5, 2), synth_csr(30000, 6, 2)
    dA, dB = to_hs(A).toGpuCSR(), to_hs(B).toGpuCSR()
    c1 = hs.gpuSpMMWrapper(dA, dA, h)
    first = c1.toCpuCSR()
    ptrs = (c1.rowPtr, c1.colInd, c1.values)
    c1.deviceDispose()
    freed = set(ptrs)
    reused = False
    for _ in range(3):                               # the pool hands freed blocks out again (best fit: which ones depends
        c2 = hs.gpuSpMMWrapper(dB, dB, h)            # on what earlier tests left cached; the same request twice must reuse)
        reused = reused or bool({c2.rowPtr, c2.colInd, c2.values} & freed)
        second = c2.toCpuCSR()
        freed |= {c2.rowPtr, c2.colInd, c2.values}
        c2.deviceDispose()
    assert reused, "expected cached blocks to be reused"
    from helpers import assert_parity
    assert_parity(first, po.omp_spmm(A, A), what="first product")
    assert_parity(second, po.omp_spmm(B, B), what="second product")
    dA.deviceDispose(); dB.deviceDispose()


def test_pending_symbolic_phase_is_invalidated_by_other_calls():
    """symbolic -> row_flops on a DIFFERENT matrix -> numeric must fail with SPGEMM_ERR_ARG instead of running the
    numeric kernels on the other matrix's bins."""
    h = hs.Handle(0)
    A, B = synth_csr(20000, 7, 2), synth_csr(20000, 8, 2)
    dA, dB = to_hs(A).toGpuCSR(), to_hs(B).toGpuCSR()
    dIC = hs.dev_alloc(4 * (A.rows + 1))
    nnz = hs.spgemm_symbolic_raw(h, dA.rowPtr, dA.colInd, A.nnz, dA.rowPtr, dA.colInd, A.nnz, A.rows, A.cols, A.cols, dIC)
    tmp = hs.dev_alloc(4 * B.rows)
    hs.row_flops_raw(h, dB.rowPtr, dB.colInd, dB.rowPtr, B.rows, tmp)
    dJC, dC = hs.dev_alloc(4 * max(nnz, 1)), hs.dev_alloc(4 * max(nnz, 1))
    with pytest.raises(hs.SpgemmError, match="status 2"):
        hs.spgemm_numeric_raw(h, dA.rowPtr, dA.colInd, dA.values, A.nnz, dA.rowPtr, dA.colInd, dA.values, A.nnz,
                              A.rows, A.cols, A.cols, dIC, dJC, dC)
    # and the protocol still works afterwards
    nnz = hs.spgemm_symbolic_raw(h, dA.rowPtr, dA.colInd, A.nnz, dA.rowPtr, dA.colInd, A.nnz, A.rows, A.cols, A.cols, dIC)
    hs.spgemm_numeric_raw(h, dA.rowPtr, dA.colInd, dA.values, A.nnz, dA.rowPtr, dA.colInd, dA.values, A.nnz,
                          A.rows, A.cols, A.cols, dIC, dJC, dC)
    for p in (dIC, tmp, dJC, dC):
        hs.dev_free(p)
    dA.deviceDispose(); dB.deviceDispose()


def test_classification_scan_is_consistent_beyond_2_pow_31_products():
    """dflops is an int32 scan like the reference's (flops.cu:119,133) whose consumers take DIFFERENCES: with
    P > 2^31 the prefixes wrap modulo 2^32 -- including the last one, which used to be clamped to INT_MAX and made the
    last row's count wrong.  A fabricated B.rowPtr (row lengths of 5e8..7e8) makes this cheap: the classification never
    touches B's columns."""
    m = 8
    IA = np.arange(m + 1, dtype=np.int32)                     # one entry per row, column = row
    JA = np.arange(m, dtype=np.int32)
    lens = np.array([700_000_000, 3, 600_000_000, 0, 7, 650_000_000, 1, 500_000_001], dtype=np.int64)
    IBfake = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=IBfake[1:])
    assert IBfake[-1] > 2**31 and IBfake[-1] < 2**32
    IB32 = IBfake.astype(np.uint32).view(np.int32)            # B.rowPtr as the int32 bit patterns (differences are exact mod 2^32)
    # row lengths must be < 2^31 individually for the flops kernel: they are
    h = hs.Handle(0)
    dIA, dJA, dIB = hs.h2d(IA), hs.h2d(JA), hs.h2d(IB32)
    ids, fl, hv = C.c_void_p(), C.c_void_p(), (C.c_int * hs.HV_LEN)()
    hvl, tot = C.c_int(0), C.c_longlong(0)
    rc = hs.lib().hip_gpuFlopsClassify(h.ptr, C.c_void_p(dIA), C.c_void_p(dJA), C.c_void_p(dIB), m, m, C.byref(ids),
                                       C.byref(fl), hv, C.byref(hvl), C.byref(tot))
    assert rc == 0, hs.lib().spgemm_hip_last_error()
    rowIds = hs.d2h(ids.value, m, np.int32)
    dfl = hs.d2h(fl.value, m + 1, np.int32).view(np.uint32).astype(np.int64)
    per = (dfl[1:] - dfl[:-1]) % (1 << 32)
    assert tot.value == int(lens.sum())
    assert np.array_equal(per, lens[rowIds]), (per, lens[rowIds])     # every row, the last one included
    for p in (dIA, dJA, dIB, ids.value, fl.value):
        hs.dev_free(p)


def test_more_than_2_pow_30_products_take_the_two_phase_path():
    """hip_gpuSpMM sizes C by P only up to 2^30 products; beyond that it finishes the symbolic phase first and
    allocates exactly nnz(C).  65 536 rows x 130 entries x B rows of 128 entries = 1.09e9 products onto 128 columns
    (values chosen so that every float sum is exact).  The fused R-MCL step falls back to SpGEMM + prune there."""
    import torch
    from sparse_matrix_with_flops_amd.dist import HipEngine, make_matrix
    m, alen, k, blen = 1 << 16, 130, 512, 128
    rng = np.random.default_rng(3)
    rpA = (np.arange(m + 1, dtype=np.int64) * alen).astype(np.int32)
    ciA = np.tile(np.arange(alen, dtype=np.int32), m) + np.repeat(rng.integers(0, k - alen, size=m).astype(np.int32), alen)
    vA = rng.integers(1, 3, size=m * alen).astype(np.float32)
    rpB = (np.arange(k + 1, dtype=np.int64) * blen).astype(np.int32)
    ciB = np.tile(np.arange(blen, dtype=np.int32), k)
    vB = np.ones(k * blen, dtype=np.float32)
    eng = HipEngine(0)
    A = make_matrix(eng, rpA, ciA, vA, m, k)
    B = make_matrix(eng, rpB, ciB, vB, k, blen)
    C_ = eng.spmm(A, B)
    st = eng.stats()
    assert st["total_flops"] == m * alen * blen > (1 << 30)
    rp, ci, v = C_.to_host()
    C_.release()
    assert np.array_equal(rp, np.arange(m + 1, dtype=np.int64) * blen)
    assert np.array_equal(np.sort(ci.reshape(m, blen), axis=1), np.tile(np.arange(blen, dtype=np.int32), (m, 1)))
    want = vA.reshape(m, alen).sum(axis=1)
    assert np.array_equal(v.reshape(m, blen), np.repeat(want[:, None], blen, axis=1))
    prp, pci, pv = eng.expand_prune(A, B)
    torch.cuda.synchronize()
    assert np.array_equal(prp.cpu().numpy(), rp) and pci.numel() == m * blen      # equal values: everything is kept
    assert np.allclose(pv.cpu().numpy(), 1.0 / blen, rtol=1e-6)


def test_nnzC_beyond_int32_is_refused():
    """131 072 rows x 129 entries x B rows of 128 entries with disjoint columns: nnz(C) = P = 2.16e9 > 2^31 - 1.  The
    symbolic phase counts it (64-bit total) and the call fails with SPGEMM_ERR_OVERFLOW instead of wrapping."""
    from sparse_matrix_with_flops_amd.dist import HipEngine, make_matrix
    m, alen, k, blen = 1 << 17, 129, 4096, 128
    rng = np.random.default_rng(4)
    rpA = (np.arange(m + 1, dtype=np.int64) * alen).astype(np.int32)
    ciA = np.tile(np.arange(alen, dtype=np.int32), m) + np.repeat(rng.integers(0, k - alen, size=m).astype(np.int32), alen)
    vA = np.ones(m * alen, dtype=np.float32)
    rpB = (np.arange(k + 1, dtype=np.int64) * blen).astype(np.int32)
    ciB = np.arange(k * blen, dtype=np.int32)                       # row j owns columns [128 j, 128 j + 128)
    vB = np.ones(k * blen, dtype=np.float32)
    eng = HipEngine(0)
    A = make_matrix(eng, rpA, ciA, vA, m, k)
    B = make_matrix(eng, rpB, ciB, vB, k, k * blen)
    with pytest.raises(Exception) as ei:
        eng.spmm(A, B)
    assert "does not fit int32" in str(ei.value)
    # the handle is usable afterwards
    small = synth_csr(4096, 11, 2)
    got = hs.gpuSpMMWrapper(to_hs(small).toGpuCSR(), to_hs(small).toGpuCSR(), eng.handle).toCpuCSR()
    assert got.nnz == po.omp_spmm(small, small).nnz


def test_host_array_entry_point_moves_large_arrays_through_the_copy_lanes():
    """hip_CSR_SpMM on operands and a result far above the 1 MB below which it uses plain copies: the operands go up and
    the malloc()ed result comes down through the 8 threads x 2 pinned slots (spgemm_hip.hip: copy_pageable) -- slices
    that do not divide evenly, a last chunk shorter than a slot.  Full parity against the oracle, twice (slots reused),
    and the phase statistics of the call."""
    A = synth_csr(100003, 23, 2)                                     # odd sizes on purpose
    want = po.omp_spmm(A, A)
    hA = to_hs(A)
    from helpers import assert_parity
    for rep in range(2):
        got = hA.hip_spmm(hA)
        assert_parity(got, want, what=f"host API, run {rep}")
    runs = hs.host_api_timed(hA, hA, reps=2)
    r = runs[-1]
    assert r["nnzC"] == want.nnz
    assert r["bytes_d2h"] == 4 * (A.rows + 1) + 8 * want.nnz and r["bytes_h2d"] == 4 * (A.rows + 1) + 8 * A.nnz
    assert r["ms_total"] >= r["ms_h2d"] + r["ms_device"] + r["ms_d2h"] - 0.5
    assert r["bytes_d2h"] > (64 << 20)                               # really beyond the small-copy path
    print(f"host API: {r['ms_total']:.1f} ms (h2d {r['ms_h2d']:.1f}, device {r['ms_device']:.1f}, d2h {r['ms_d2h']:.1f}; "
          f"{r['bytes_d2h'] / r['ms_d2h'] / 1e6:.1f} GB/s down)")
    # A != B: six arrays go up
    B = synth_csr(100003, 29, 2)
    got = hA.hip_spmm(to_hs(B))
    assert_parity(got, po.omp_spmm(A, B), what="host API, A != B")


def test_compressive_product_is_not_held_at_twice_its_size():
    """hip_gpuSpMM sizes C by the product count P (an upper bound: no host round trip between its phases).  For a product
    that compresses -- web graphs: nnz(C)/P ~ 0.5 -- that is twice the memory; from the second call on the same shape the
    handle takes the two-phase path and allocates nnz(C) entries exactly.  Checked through the pool: the blocks C gives
    back when it is freed."""
    from sparse_matrix_with_flops_amd import synth
    m = 120000
    rp, ci, v = synth.webgraph_csr(m, 7)
    A = po.CSRHost(rp, ci, v, m, m)
    want = po.omp_spmm(A, A)
    P = int(po.row_flops(A, A).sum())
    assert want.nnz < 0.6 * P
    h = hs.Handle(0)
    dA = to_hs(A).toGpuCSR()
    from helpers import assert_parity
    # earlier tests leave idle blocks in the pool and it may hand out one up to 4x a request: start from an empty cache;
    # call 0 is sized by P and its C stays alive, so that the later calls cannot be handed its (larger) blocks either
    hs.pool_trim(0)
    assert hs.pool_cached_bytes(0) == 0
    dC0 = hs.gpuSpMMWrapper(dA, dA, h)
    sizes = []
    for call in (1, 2):
        dC = hs.gpuSpMMWrapper(dA, dA, h)
        mid = hs.pool_cached_bytes(0)
        got = dC.toCpuCSR()
        dC.deviceDispose()
        sizes.append(hs.pool_cached_bytes(0) - mid)    # bytes the three arrays of this C occupied
        assert_parity(got, want, what=f"compressive product, call {call}")
    mid = hs.pool_cached_bytes(0)
    assert_parity(dC0.toCpuCSR(), want, what="compressive product, call 0")
    dC0.deviceDispose()
    size0 = hs.pool_cached_bytes(0) - mid
    assert size0 >= 8 * P                              # first call: sized by P
    assert sizes[0] <= 8 * want.nnz + 4 * (m + 1) + (8 << 20)      # from then on: exact (+ rounding of three blocks)
    # a non-compressive product keeps the one-shot path
    B = synth_csr(60000, 3, 2)
    dB = to_hs(B).toGpuCSR()
    for call in range(2):
        dC = hs.gpuSpMMWrapper(dB, dB, h)
        st = h.stats()
        dC.deviceDispose()
    assert st["nnzC"] > 0.75 * st["total_flops"]
    dA.deviceDispose()
    dB.deviceDispose()
    h.close()
