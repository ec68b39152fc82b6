"""ctypes doorway onto the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (sparse_matrix_with_flops_amd) never imports it.

Two libraries:
  liboracle.so      our C restatement (oracle/oracle.c), built by `make -C oracle`
  _ref/libref.so    the real reference CPU kernels (oracle/ref_shim.cc + /root/reference sources),
                    built by `make -C oracle ref` in the build container only; optional.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_I = C.POINTER(C.c_int)
_F = C.POINTER(C.c_float)
_LL = C.POINTER(C.c_longlong)


def _ip(a):
    return a.ctypes.data_as(_I)


def _fp(a):
    return a.ctypes.data_as(_F)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build(ref=True):
    """Compile liboracle.so (always) and _ref/libref.so (when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE])
    if ref and os.path.isdir("/root/reference/nlibs"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _lib = C.CDLL(path)
        _lib.oracle_compute_threshold.restype = C.c_float
        _lib.oracle_compute_threshold.argtypes = [C.c_float, C.c_float]
        _lib.oracle_gpu_bin_id.argtypes = [C.c_longlong]
        _lib.oracle_free.argtypes = [C.c_void_p]
    return _lib


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))


def ref():
    global _ref
    if _ref is None:
        _ref = C.CDLL(os.path.join(HERE, "_ref", "libref.so"))
        _ref.ref_compute_threshold.restype = C.c_float
        _ref.ref_compute_threshold.argtypes = [C.c_float, C.c_float]
        _ref.ref_free.argtypes = [C.c_void_p]
    return _ref


def _take(ptr, n, dtype, free):
    """Copy a malloc'd C array into numpy and free it."""
    if n > 0:
        out = np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)
    else:
        out = np.zeros(0, dtype=dtype)
    free(C.cast(ptr, C.c_void_p))
    return out


class CSRHost:
    """Plain host CSR triple (numpy), field names as in nlibs/CSR.h:23-38."""

    def __init__(self, rowPtr, colInd, values, rows, cols):
        self.rowPtr = _i32(rowPtr)
        self.colInd = _i32(colInd)
        self.values = _f32(values)
        self.rows = int(rows)
        self.cols = int(cols)
        self.nnz = int(self.rowPtr[-1]) if len(self.rowPtr) else 0

    def canonical(self):
        """Rows sorted by column (CSR::makeOrdered, nlibs/CSR.cc:73-86) — returns a copy."""
        ci = self.colInd.copy()
        v = self.values.copy()
        lib().oracle_csr_make_ordered(C.c_int(self.rows), _ip(self.rowPtr), _ip(ci), _fp(v))
        return CSRHost(self.rowPtr.copy(), ci, v, self.rows, self.cols)


def _spmm_call(fn, free, A, B, *extra):
    ic, jc, cv, nnz = _I(), _I(), _F(), C.c_int()
    rc = fn(*extra, _ip(A.rowPtr), _ip(A.colInd), _fp(A.values), C.c_int(A.nnz),
            _ip(B.rowPtr), _ip(B.colInd), _fp(B.values), C.c_int(B.nnz),
            C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz),
            C.c_int(A.rows), C.c_int(A.cols), C.c_int(B.cols))
    assert rc == 0
    n = nnz.value
    rp = _take(ic, A.rows + 1, np.int32, free)
    ci = _take(jc, n, np.int32, free)
    v = _take(cv, n, np.float32, free)
    return CSRHost(rp, ci, v, A.rows, B.cols)


def sequential_spmm(A, B):
    """oracle_sequential_spmm == sequential_CSR_SpMM (nlibs/cpu_csr_kernel.cc:76-119)."""
    return _spmm_call(lib().oracle_sequential_spmm, lib().oracle_free, A, B)


def omp_spmm(A, B, stride=512):
    """oracle_omp_spmm == omp_CSR_SpMM (nlibs/omp_csr_kernel.cc:238-315)."""
    L = lib()
    ic, jc, cv, nnz = _I(), _I(), _F(), C.c_int()
    rc = L.oracle_omp_spmm(_ip(A.rowPtr), _ip(A.colInd), _fp(A.values), C.c_int(A.nnz),
                           _ip(B.rowPtr), _ip(B.colInd), _fp(B.values), C.c_int(B.nnz),
                           C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz),
                           C.c_int(A.rows), C.c_int(A.cols), C.c_int(B.cols), C.c_int(stride))
    assert rc == 0
    n = nnz.value
    return CSRHost(_take(ic, A.rows + 1, np.int32, L.oracle_free), _take(jc, n, np.int32, L.oracle_free),
                   _take(cv, n, np.float32, L.oracle_free), A.rows, B.cols)


def time_omp_spmm(A, B, use_ref, stride=512):
    """Seconds spent inside the C call alone (no copy-out): the reference's omp_CSR_SpMM when use_ref, else the
    restatement.  The malloc'd outputs are freed unread."""
    import time
    L = ref() if use_ref else lib()
    free = L.ref_free if use_ref else L.oracle_free
    ic, jc, cv, nnz = _I(), _I(), _F(), C.c_int()
    args = [_ip(A.rowPtr), _ip(A.colInd), _fp(A.values), C.c_int(A.nnz), _ip(B.rowPtr), _ip(B.colInd), _fp(B.values),
            C.c_int(B.nnz), C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz), C.c_int(A.rows), C.c_int(A.cols),
            C.c_int(B.cols), C.c_int(stride)]
    t0 = time.perf_counter()
    rc = L.ref_spmm(C.c_int(REF_KINDS["omp"]), *args) if use_ref else L.oracle_omp_spmm(*args)
    dt = time.perf_counter() - t0
    assert rc == 0
    for p_ in (ic, jc, cv):
        free(C.cast(p_, C.c_void_p))
    return dt, nnz.value


REF_KINDS = {"sequential": 0, "omp": 1, "static_omp": 2, "flops_omp": 3, "group": 4, "noindex_somp": 5}


def ref_spmm(A, B, which="sequential", stride=512):
    """The real reference kernels through oracle/_ref/libref.so."""
    R = ref()
    ic, jc, cv, nnz = _I(), _I(), _F(), C.c_int()
    rc = R.ref_spmm(C.c_int(REF_KINDS[which]), _ip(A.rowPtr), _ip(A.colInd), _fp(A.values), C.c_int(A.nnz),
                    _ip(B.rowPtr), _ip(B.colInd), _fp(B.values), C.c_int(B.nnz),
                    C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz),
                    C.c_int(A.rows), C.c_int(A.cols), C.c_int(B.cols), C.c_int(stride))
    assert rc == 0
    n = nnz.value
    return CSRHost(_take(ic, A.rows + 1, np.int32, R.ref_free), _take(jc, n, np.int32, R.ref_free),
                   _take(cv, n, np.float32, R.ref_free), A.rows, B.cols)


def row_flops(A, B):
    out = np.zeros(A.rows + 1, dtype=np.int64)
    lib().oracle_row_flops(_ip(A.rowPtr), _ip(A.colInd), _ip(B.rowPtr), C.c_int(A.rows),
                           out.ctypes.data_as(_LL), C.c_int(0))
    return out[:A.rows]


def ref_row_flops_prefix(A, B, stride=512):
    out = np.zeros(A.rows + 1, dtype=np.int64)  # C long is 64-bit on this ABI
    ref().ref_row_flops_prefix(_ip(A.rowPtr), _ip(A.colInd), _ip(B.rowPtr), _ip(B.colInd),
                               C.c_int(A.rows), C.c_int(B.cols), out.ctypes.data_as(C.POINTER(C.c_long)),
                               C.c_int(stride))
    return out


def equal_partition64(prefix, parts):
    prefix = np.ascontiguousarray(prefix, dtype=np.int64)
    ends = np.zeros(parts + 1, dtype=np.int32)
    lib().oracle_equal_partition64(prefix.ctypes.data_as(_LL), C.c_int(len(prefix) - 1), C.c_int(parts), _ip(ends))
    return ends


def ref_equal_partition64(prefix, parts):
    prefix = np.ascontiguousarray(prefix, dtype=np.int64).copy()
    ends = np.zeros(parts + 1, dtype=np.int32)
    ref().ref_equal_partition64(prefix.ctypes.data_as(C.POINTER(C.c_long)), C.c_int(len(prefix) - 1),
                                C.c_int(parts), _ip(ends))
    return ends


def footprints(A, B, counts=None):
    """Restatement of footPrintsCrowiCount (nlibs/static_omp_csr_kernel.cc:28-62): per row of C = A*B
    (flops + nnzC_row + 32 + nnzA_row) >> 1, 0 for an empty A row; returns the exclusive prefix [m+1] (int64)."""
    fl = row_flops(A, B).astype(np.int64)
    if counts is None:
        counts = np.diff(omp_spmm(A, B).rowPtr)
    na = np.diff(A.rowPtr).astype(np.int64)
    fp = np.where(na > 0, (fl + np.asarray(counts, dtype=np.int64) + 32 + na) >> 1, 0)
    out = np.zeros(A.rows + 1, dtype=np.int64)
    np.cumsum(fp, out=out[1:])
    return out


def ref_footprints(A, B, stride=512):
    """-> (C.rowPtr, footprint prefix) from the real reference (dynamic_omp_CSR_IC_nnzC_footprints)."""
    ic = np.zeros(A.rows + 1, dtype=np.int32)
    fp = np.zeros(A.rows + 1, dtype=np.int32)
    ref().ref_footprints(_ip(A.rowPtr), _ip(A.colInd), _ip(B.rowPtr), _ip(B.colInd), C.c_int(A.rows), C.c_int(B.cols),
                         _ip(ic), _ip(fp), C.c_int(stride))
    return ic, fp


def ref_equal_partition(prefix, parts):
    prefix = np.ascontiguousarray(prefix, dtype=np.int32).copy()
    ends = np.zeros(parts + 1, dtype=np.int32)
    ref().ref_equal_partition(_ip(prefix), C.c_int(len(prefix) - 1), C.c_int(parts), _ip(ends))
    return ends


def group_bins(A, B):
    rf = np.zeros(A.rows + 1, dtype=np.int32)
    groups = np.zeros(A.rows + 1, dtype=np.int32)
    tops = np.zeros(8, dtype=np.int32)
    lib().oracle_group_bins(_ip(A.rowPtr), _ip(A.colInd), _ip(B.rowPtr), C.c_int(A.rows),
                            _ip(rf), _ip(groups), _ip(tops))
    return rf[:A.rows], groups[:A.rows], tops


def ref_group_bins(A, B):
    rf = np.zeros(A.rows + 1, dtype=np.int32)
    groups = np.zeros(A.rows + 1, dtype=np.int32)
    ic = np.zeros(A.rows + 1, dtype=np.int32)
    tops = np.zeros(8, dtype=np.int32)
    ref().ref_group_flops(_ip(A.rowPtr), _ip(A.colInd), _ip(B.rowPtr), _ip(B.colInd), C.c_int(A.rows),
                          C.c_int(B.cols), _ip(ic), _ip(rf), _ip(groups), _ip(tops), C.c_int(512))
    return rf[:A.rows], groups[:A.rows], tops


def gpu_classify(flops):
    """oracle_gpu_classify == gpuFlopsClassify (mindex2-cuda/flops.cu:110-185)."""
    flops = np.ascontiguousarray(flops, dtype=np.int64)
    m = len(flops)
    rowIds = np.zeros(max(m, 1), dtype=np.int32)
    scan = np.zeros(m + 1, dtype=np.int64)
    hv = np.zeros(9, dtype=np.int32)
    n = lib().oracle_gpu_classify(flops.ctypes.data_as(_LL), C.c_int(m), _ip(rowIds),
                                  scan.ctypes.data_as(_LL), _ip(hv))
    return rowIds[:m], scan, hv, n


def _load_common(fn, free, path, *args):
    rows, cols, nnz = C.c_int(), C.c_int(), C.c_int()
    rp, ci, v = _I(), _I(), _F()
    rc = fn(path.encode(), *args, C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(rp), C.byref(ci), C.byref(v))
    assert rc == 0, rc
    return CSRHost(_take(rp, rows.value + 1, np.int32, free), _take(ci, nnz.value, np.int32, free),
                   _take(v, nnz.value, np.float32, free), rows.value, cols.value)


def ref_load(path, isTrans=False, mode=0, toAbs=False):
    """mode 0: readSNAPFile+dedupe+toCSR (nGpuSpMM.cc:285-291); mode 1: readSNAPFile+rmclInit."""
    return _load_common(ref().ref_load, ref().ref_free, path, C.c_int(int(isTrans)), C.c_int(mode), C.c_int(int(toAbs)))


def ref_rmcl(path, iters, opt=0):
    return _load_common(ref().ref_rmcl, ref().ref_free, path, C.c_int(iters), C.c_int(opt))


def ref_mt_rmcl(Mgt, Mt, iters, opt=1, stride=512):
    """The reference's multi-threaded R-MCL loop, mtRmclIter (nlibs/qrmcl.cc:8-84), on in-memory CSRs; opt: 1 OMP, 4 SOMP
    (enum RunOptions).  -> (new Mt, seconds spent inside the reference's loop)."""
    L = ref()
    L.ref_mt_rmcl_iter.restype = C.c_double
    rows, cols, nnz = C.c_int(), C.c_int(), C.c_int()
    rp, ci, v = _I(), _I(), _F()
    dt = L.ref_mt_rmcl_iter(C.c_int(iters), C.c_int(opt), C.c_int(stride), C.c_int(Mgt.rows), C.c_int(Mgt.cols),
                            _ip(Mgt.rowPtr), _ip(Mgt.colInd), _fp(Mgt.values), C.c_int(Mgt.nnz),
                            _ip(Mt.rowPtr), _ip(Mt.colInd), _fp(Mt.values), C.c_int(Mt.nnz),
                            C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(rp), C.byref(ci), C.byref(v))
    out = CSRHost(_take(rp, rows.value + 1, np.int32, L.ref_free), _take(ci, nnz.value, np.int32, L.ref_free),
                  _take(v, nnz.value, np.float32, L.ref_free), rows.value, cols.value)
    return out, float(dt)


def read_snap(path, isTrans=False):
    """-> (rows, cols, ri, ci, v) raw triplets; oracle_read_snap == COO::readSNAPFile."""
    L = lib()
    rows, cols, nnz = C.c_int(), C.c_int(), C.c_int()
    ri, ci, v = _I(), _I(), _F()
    rc = L.oracle_read_snap(path.encode(), C.c_int(int(isTrans)), C.byref(rows), C.byref(cols), C.byref(nnz),
                            C.byref(ri), C.byref(ci), C.byref(v))
    assert rc == 0, rc
    n = nnz.value
    if not ri:
        return rows.value, cols.value, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
    return (rows.value, cols.value, _take(ri, n, np.int32, L.oracle_free), _take(ci, n, np.int32, L.oracle_free),
            _take(v, n, np.float32, L.oracle_free))


def coo_to_csr(rows, cols, ri, ci, v, dedupe=True, toAbs=False):
    """sort (+dedupe) + toCSR: COO::orderedAndDuplicatesRemoving / makeOrdered + toCSR."""
    L = lib()
    ri, ci, v = _i32(ri).copy(), _i32(ci).copy(), _f32(v).copy()
    n = L.oracle_coo_sort(C.c_int(len(ri)), _ip(ri), _ip(ci), _fp(v), C.c_int(int(dedupe)))
    ri, ci, v = ri[:n], ci[:n], v[:n]
    rp = np.zeros(rows + 1, dtype=np.int32)
    L.oracle_coo_to_csr(C.c_int(rows), C.c_int(n), _ip(ri), _ip(rp))
    if toAbs:
        v = np.abs(v)
    return CSRHost(rp, ci, v, rows, cols)


def load(path, isTrans=False, mode=0, toAbs=False):
    """Restatement of ref_load()."""
    rows, cols, ri, ci, v = read_snap(path, isTrans)
    if mode == 0:
        return coo_to_csr(rows, cols, ri, ci, v, dedupe=True, toAbs=toAbs)
    return rmcl_init(rows, cols, ri, ci, v)


def rmcl_init(rows, cols, ri, ci, v):
    """rmclInit (nlibs/qrmcl.cc:126-134): self loops, sort, toCSR, row-normalise."""
    # self loops
    diag = np.zeros(rows, dtype=bool)
    d = ri == ci
    diag[ri[d]] = True
    miss = np.nonzero(~diag)[0].astype(np.int32)
    ri = np.concatenate([_i32(ri), miss])
    ci = np.concatenate([_i32(ci), miss])
    v = np.concatenate([_f32(v), np.ones(len(miss), np.float32)])
    A = coo_to_csr(rows, cols, ri, ci, v, dedupe=False)
    lib().oracle_csr_aver_norm(C.c_int(rows), _ip(A.rowPtr), _fp(A.values))
    return A


def flops_stats(A, B):
    """flopsStats (nlibs/tools/stats.cc:45-55): 13-bucket power-of-two histogram of the per-row flops of A*B."""
    out = np.zeros(13, dtype=np.int32)
    lib().oracle_flops_stats(_ip(A.rowPtr), _ip(A.colInd), _ip(B.rowPtr), C.c_int(A.rows), _ip(out))
    return out


def rmcl_prune_row(cols, vals):
    cols, vals = _i32(cols).copy(), _f32(vals).copy()
    k = lib().oracle_rmcl_prune_row(C.c_int(len(cols)), _ip(cols), _fp(vals))
    return cols[:k], vals[:k]


def rmcl_iters(Mgt, Mt, iters):
    """seqRmclIter (nlibs/qrmcl.cc:86-124) on numpy CSRs; returns the new Mt."""
    cur = Mt
    for _ in range(iters):
        Cm = sequential_spmm(Mgt, cur)
        rp, ci, v = Cm.rowPtr.copy(), Cm.colInd.copy(), Cm.values.copy()
        n = lib().oracle_rmcl_prune_compact(C.c_int(Cm.rows), _ip(rp), _ip(ci), _fp(v))
        cur = CSRHost(rp, ci[:n], v[:n], Cm.rows, Cm.cols)
    return cur
