"""ctypes binding of libspgemm_hip.so (include/spgemm_hip.h) + a thin Python mirror of the
reference's host-side call surface for the SpGEMM path.

Mirrors (reference file:line):
  CSR                         struct CSR                       nlibs/CSR.h:23-50
  CSR.hip_spmm(B)             CSR::spmm / omp_spmm / ...       nlibs/CSR.cc:59-208   (host in, host out)
  CSR.toGpuCSR / toCpuCSR / deviceDispose                      nlibs/CSR.cc:342-379
  gpuSpMMWrapper(dA, dB)      gpuSpMMWrapper                   nlibs/gpus/gpu_csr_kernel.cu:128-173
  gpuFlopsClassify(dA, dB)    gpuFlopsClassify                 mindex2-cuda/flops.cu:110-185
  sgpuSpMMWrapper(...)        sgpuSpMMWrapper                  mindex2-cuda/kernel.cu:311-427
  scudaSpMM(hA, hB)           scudaSpMM                        mindex2-cuda/nGpuSpMM.cc:245-279

There is no CPU fallback: if the library or a HIP device is missing, calls raise SpgemmError.
PyTorch is not needed here; bench.py / dist.py use it only for device memory and torch.distributed.
"""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPGEMM_LIB: diagnostic builds only (make -C csrc ablate); the product library is the default
LIB_PATH = os.environ.get("SPGEMM_LIB") or os.path.join(_HERE, "libspgemm_hip.so")

NBINS = 9
HV_LEN = 9
NKERNELS = 21
_I = C.POINTER(C.c_int)
_F = C.POINTER(C.c_float)

EXPORTS = [
    "spgemm_hip_last_error", "spgemm_hip_device_count", "spgemm_hip_create", "spgemm_hip_destroy",
    "spgemm_hip_get_stats", "spgemm_hip_stream", "spgemm_hip_malloc", "spgemm_hip_free",
    "spgemm_hip_memcpy_h2d", "spgemm_hip_memcpy_d2h", "spgemm_hip_memcpy_d2d", "hip_CSR_SpMM", "hip_gpuSpMM",
    "hip_gpuFlopsClassify", "hip_sgpuSpMM", "hip_csr_sort_rows", "spgemm_hip_selftest",
    "hip_spgemm_symbolic", "hip_spgemm_numeric", "hip_csr_row_flops", "spgemm_hip_kernel_name",
    "hip_rmcl_prune", "hip_gpuRmclIter", "hip_coo_to_csr", "hip_flopsStats", "spgemm_hip_set_kernel_timing",
    "hip_rmcl_prune_n", "hip_rmcl_expand_prune", "hip_gpuRmclIter_device", "spgemm_hip_pool_cached_bytes", "hip_nnzStats", "hip_resultsComparison",
    "spgemm_hip_group_create", "spgemm_hip_unique_id", "spgemm_hip_group_create_rank", "spgemm_hip_group_info",
    "spgemm_hip_group_destroy", "hip_sharded_spmm_create", "hip_sharded_spmm_step", "hip_sharded_spmm_result",
    "hip_sharded_spmm_info", "hip_sharded_spmm_destroy", "hip_gpuRmclIter_sharded", "spgemm_hip_host_api_stats",
    "hip_sharded_spmm_handle", "spgemm_hip_rccl_available", "spgemm_hip_pool_trim",
    "hip_sharded_rmcl_create", "hip_sharded_rmcl_run", "hip_sharded_rmcl_continue", "hip_sharded_rmcl_result", "hip_sharded_rmcl_iter_nnz",
    "hip_sharded_rmcl_info", "hip_sharded_rmcl_destroy", "spgemm_hip_debug_fail_next", "spgemm_hip_rmcl_devices_used", "spgemm_hip_handle_device",
]
XCHG_AUTO, XCHG_RCCL, XCHG_PEER, XCHG_HOST = 0, 1, 2, 3
XCHG_NAMES = {0: "auto", 1: "rccl", 2: "peer", 3: "host"}
UNIQUE_ID_BYTES = 128


class SpgemmError(RuntimeError):
    pass


class BinReport(C.Structure):
    _fields_ = [("rows", C.c_int), ("rows_differ", C.c_int), ("first_bad_row", C.c_int), ("max_rel_err", C.c_double)]


class HostApiStats(C.Structure):
    _fields_ = [("ms_h2d", C.c_float), ("ms_device", C.c_float), ("ms_d2h", C.c_float), ("ms_total", C.c_float),
                ("bytes_h2d", C.c_longlong), ("bytes_d2h", C.c_longlong)]


class Stats(C.Structure):
    _fields_ = [("total_flops", C.c_longlong), ("nnzC", C.c_int), ("bin_rows", C.c_int * NBINS),
                ("ms_classify", C.c_float), ("ms_symbolic", C.c_float), ("ms_scan_alloc", C.c_float),
                ("ms_numeric", C.c_float), ("ms_total", C.c_float), ("ms_kernel", C.c_float * NKERNELS)]

    def as_dict(self):
        return {"total_flops": int(self.total_flops), "nnzC": int(self.nnzC), "bin_rows": [int(x) for x in self.bin_rows],
                "ms_classify": float(self.ms_classify), "ms_symbolic": float(self.ms_symbolic),
                "ms_scan_alloc": float(self.ms_scan_alloc), "ms_numeric": float(self.ms_numeric),
                "ms_total": float(self.ms_total),
                "ms_kernel": {kernel_name(i): float(self.ms_kernel[i]) for i in range(NKERNELS)
                              if kernel_name(i) and self.ms_kernel[i] > 0.0}}


_lib = None


def lib():
    """Load libspgemm_hip.so (built in tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpgemmError(f"{LIB_PATH} is missing: run `make -C sparse_matrix_with_flops_amd/csrc` "
                              "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(LIB_PATH)
        L.spgemm_hip_last_error.restype = C.c_char_p
        L.spgemm_hip_stream.restype = C.c_void_p
        L.spgemm_hip_stream.argtypes = [C.c_void_p]
        L.spgemm_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.spgemm_hip_destroy.argtypes = [C.c_void_p]
        L.spgemm_hip_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.spgemm_hip_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.spgemm_hip_free.argtypes = [C.c_void_p]
        L.spgemm_hip_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.spgemm_hip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.hip_CSR_SpMM.argtypes = [_I, _I, _F, C.c_int, _I, _I, _F, C.c_int, C.POINTER(_I), C.POINTER(_I),
                                   C.POINTER(_F), _I, C.c_int, C.c_int, C.c_int]
        dev_in = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.hip_gpuSpMM.argtypes = [C.c_void_p] + dev_in + dev_in + [C.c_int, C.c_int, C.c_int] + \
            [C.POINTER(C.c_void_p)] * 3 + [_I]
        L.hip_gpuFlopsClassify.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _I,
                                           C.POINTER(C.c_longlong)]
        L.hip_sgpuSpMM.argtypes = [C.c_void_p] + dev_in + dev_in + [C.c_int, C.c_int, C.c_int] + \
            [C.c_void_p, _I, C.c_void_p] + [C.POINTER(C.c_void_p)] * 3 + [_I]
        L.hip_spgemm_symbolic.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p, _I]
        L.hip_spgemm_numeric.argtypes = [C.c_void_p] + dev_in + dev_in + [C.c_int, C.c_int, C.c_int,
                                                                            C.c_void_p, C.c_void_p, C.c_void_p]
        L.hip_csr_row_flops.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.POINTER(C.c_longlong)]
        L.spgemm_hip_kernel_name.restype = C.c_char_p
        L.spgemm_hip_kernel_name.argtypes = [C.c_int]
        L.hip_coo_to_csr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I]
        L.hip_flopsStats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.spgemm_hip_memcpy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.hip_rmcl_prune.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + \
            [C.POINTER(C.c_void_p)] * 3 + [_I]
        L.hip_rmcl_expand_prune.argtypes = [C.c_void_p] + dev_in + dev_in + [C.c_int, C.c_int, C.c_int] + \
            [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(C.c_int)]
        L.hip_gpuRmclIter_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + dev_in + dev_in + \
            [C.POINTER(C.c_void_p)] * 3 + [C.POINTER(C.c_int)]
        L.hip_rmcl_prune_n.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + \
            [C.POINTER(C.c_void_p)] * 3 + [_I]
        L.spgemm_hip_pool_cached_bytes.argtypes = [C.c_int, C.POINTER(C.c_size_t)]
        L.hip_nnzStats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.hip_resultsComparison.argtypes = [C.c_int, C.c_int, _I, _I, _F, _I, _I, _F, _I, C.c_int, _I, C.c_double,
                                            C.POINTER(BinReport)]
        L.hip_gpuRmclIter.argtypes = [C.c_int, C.c_int, C.c_int, _I, _I, _F, C.c_int, _I, _I, _F, C.c_int,
                                      C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), _I]
        L.hip_csr_sort_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.spgemm_hip_selftest.argtypes = [C.c_void_p]
        L.spgemm_hip_set_kernel_timing.argtypes = [C.c_void_p, C.c_uint]
        host_in = [_I, _I, _F, C.c_int]
        L.spgemm_hip_host_api_stats.argtypes = [C.POINTER(HostApiStats)]
        L.spgemm_hip_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, _I, C.c_int]
        L.spgemm_hip_unique_id.argtypes = [C.c_void_p]
        L.spgemm_hip_group_create_rank.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.spgemm_hip_group_info.argtypes = [C.c_void_p, _I, _I, _I]
        L.spgemm_hip_group_destroy.argtypes = [C.c_void_p]
        L.hip_sharded_spmm_create.argtypes = [C.c_void_p] + host_in + host_in + [C.c_int, C.c_int, C.c_int,
                                                                                   C.POINTER(C.c_void_p)]
        L.hip_sharded_spmm_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.hip_sharded_spmm_result.argtypes = [C.c_void_p, C.c_int, C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), _I, _I]
        L.hip_sharded_spmm_info.argtypes = [C.c_void_p, _I, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.hip_sharded_spmm_destroy.argtypes = [C.c_void_p]
        L.hip_sharded_spmm_handle.argtypes = [C.c_void_p, C.c_int]
        L.hip_sharded_spmm_handle.restype = C.c_void_p
        L.hip_gpuRmclIter_sharded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + host_in + host_in + \
            [C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), _I]
        L.hip_sharded_rmcl_create.argtypes = [C.c_void_p, C.c_int, C.c_int] + host_in + host_in + [C.POINTER(C.c_void_p)]
        L.hip_sharded_rmcl_run.argtypes = [C.c_void_p, C.c_int, _I]
        L.hip_sharded_rmcl_continue.argtypes = [C.c_void_p, C.c_int, _I]
        L.hip_sharded_rmcl_result.argtypes = [C.c_void_p, C.c_int, C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), _I]
        L.hip_sharded_rmcl_iter_nnz.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
        L.hip_sharded_rmcl_info.argtypes = [C.c_void_p, _I]
        L.hip_sharded_rmcl_destroy.argtypes = [C.c_void_p]
        L.spgemm_hip_debug_fail_next.argtypes = [C.c_void_p, C.c_int]
        L.spgemm_hip_handle_device.argtypes = [C.c_void_p]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def kernel_name(i):
    return lib().spgemm_hip_kernel_name(int(i)).decode()


def _check(rc, what):
    if rc != 0:
        raise SpgemmError(f"{what} failed with status {rc}: {lib().spgemm_hip_last_error().decode(errors='replace')}")


def device_count():
    n = C.c_int(0)
    rc = lib().spgemm_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


# Objects that own something on the C side, closed in dependency order when the interpreter exits (jobs before the groups
# they were made on, groups before plain handles): left to the garbage collector at shutdown they are finalised in any
# order, after the HIP runtime has started to go away -- seen as a segmentation fault at exit after a failed test.
_LIVE = weakref.WeakSet()


def _close_all():
    live = list(_LIVE)
    for kind in ("ShardedSpMM", "ShardedRmcl", "Group", "Handle"):
        for o in live:
            if type(o).__name__ == kind:
                try:
                    o.close()
                except Exception:
                    pass


import atexit  # noqa: E402
atexit.register(_close_all)


class Handle:
    """spgemm_handle: one HIP stream + workspace on one device."""

    def __init__(self, device=0, _borrowed=None):
        self._own = _borrowed is None
        if _borrowed is not None:                   # a handle owned by someone else (a group's shard): never destroyed here
            self._h = C.c_void_p(_borrowed)
        else:
            self._h = C.c_void_p()
            _check(lib().spgemm_hip_create(C.byref(self._h), int(device)), "spgemm_hip_create")
            _LIVE.add(self)
        self.device = device if _borrowed is None else int(lib().spgemm_hip_handle_device(self._h))

    @property
    def ptr(self):
        return self._h

    def stats(self):
        s = Stats()
        _check(lib().spgemm_hip_get_stats(self._h, C.byref(s)), "spgemm_hip_get_stats")
        return s.as_dict()

    def stream(self):
        return lib().spgemm_hip_stream(self._h)

    def selftest(self):
        _check(lib().spgemm_hip_selftest(self._h), "spgemm_hip_selftest")

    def set_kernel_timing(self, mask):
        """bit i: bracket the launches of kernel i with HIP events (stats()['ms_kernel']); 0 = none (default)."""
        _check(lib().spgemm_hip_set_kernel_timing(self._h, int(mask) & 0xFFFFFFFF), "spgemm_hip_set_kernel_timing")

    def fail_next(self, count=1):
        """test hook: the next `count` symbolic phases / fused R-MCL steps on this handle fail"""
        _check(lib().spgemm_hip_debug_fail_next(self._h, int(count)), "spgemm_hip_debug_fail_next")

    def close(self):
        if self._h and self._own:
            lib().spgemm_hip_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# device buffers
# ---------------------------------------------------------------------------------------------
def dev_alloc(nbytes):
    p = C.c_void_p()
    _check(lib().spgemm_hip_malloc(C.byref(p), max(int(nbytes), 1)), "spgemm_hip_malloc")
    return p.value


def dev_free(ptr):
    if ptr:
        _check(lib().spgemm_hip_free(C.c_void_p(ptr)), "spgemm_hip_free")


def h2d(arr):
    arr = np.ascontiguousarray(arr)
    p = dev_alloc(arr.nbytes)
    if arr.nbytes:
        _check(lib().spgemm_hip_memcpy_h2d(C.c_void_p(p), arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")
    return p


def d2h(ptr, n, dtype):
    out = np.empty(int(n), dtype=dtype)
    if out.nbytes:
        _check(lib().spgemm_hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes), "d2h")
    return out


class CSR:
    """Mirror of the reference's `struct CSR` (nlibs/CSR.h:23-50): three arrays + rows/cols/nnz.
    Host CSRs hold numpy arrays; device CSRs (from toGpuCSR / gpuSpMMWrapper) hold raw device
    pointers in the same fields, exactly as the reference reuses one struct for both."""

    def __init__(self, values=None, colInd=None, rowPtr=None, rows=0, cols=0, nnz=0, on_device=False):
        self.values, self.colInd, self.rowPtr = values, colInd, rowPtr
        self.rows, self.cols, self.nnz = int(rows), int(cols), int(nnz)
        self.on_device = on_device

    @staticmethod
    def from_arrays(rowPtr, colInd, values, rows, cols):
        rowPtr = np.ascontiguousarray(rowPtr, dtype=np.int32)
        return CSR(np.ascontiguousarray(values, dtype=np.float32), np.ascontiguousarray(colInd, dtype=np.int32),
                   rowPtr, rows, cols, int(rowPtr[-1]) if len(rowPtr) else 0)

    # -- nlibs/CSR.cc:342-379 ------------------------------------------------------------------
    def toGpuCSR(self):
        assert not self.on_device
        return CSR(h2d(self.values), h2d(self.colInd), h2d(self.rowPtr), self.rows, self.cols, self.nnz, True)

    def toCpuCSR(self):
        assert self.on_device
        return CSR(d2h(self.values, self.nnz, np.float32), d2h(self.colInd, self.nnz, np.int32),
                   d2h(self.rowPtr, self.rows + 1, np.int32), self.rows, self.cols, self.nnz, False)

    def deviceDispose(self):
        assert self.on_device
        for p in (self.values, self.colInd, self.rowPtr):
            dev_free(p)
        self.values = self.colInd = self.rowPtr = None

    # -- nlibs/CSR.cc:73-86 -------------------------------------------------------------------
    def makeOrdered(self):
        assert not self.on_device
        row_of = np.repeat(np.arange(self.rows, dtype=np.int64), np.diff(self.rowPtr.astype(np.int64)))
        order = np.lexsort((self.colInd, row_of))
        self.colInd = np.ascontiguousarray(self.colInd[order])
        self.values = np.ascontiguousarray(self.values[order])

    # -- the drop-in: same role as CSR::spmm/omp_spmm/... (nlibs/CSR.cc:59-208) ----------------
    def hip_spmm(self, B):
        """C = self * B through hip_CSR_SpMM (host arrays in, malloc'd host arrays out)."""
        assert not self.on_device and not B.on_device
        if self.cols != B.rows:
            raise SpgemmError(f"shape mismatch: A is {self.rows}x{self.cols}, B is {B.rows}x{B.cols}")
        L = lib()
        ic, jc, cv, nnz = _I(), _I(), _F(), C.c_int(0)
        rc = L.hip_CSR_SpMM(self.rowPtr.ctypes.data_as(_I), self.colInd.ctypes.data_as(_I), self.values.ctypes.data_as(_F),
                            self.nnz, B.rowPtr.ctypes.data_as(_I), B.colInd.ctypes.data_as(_I), B.values.ctypes.data_as(_F),
                            B.nnz, C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz), self.rows, self.cols, B.cols)
        _check(rc, "hip_CSR_SpMM")
        n = nnz.value
        rp = np.ctypeslib.as_array(ic, shape=(self.rows + 1,)).copy()
        ci = np.ctypeslib.as_array(jc, shape=(n,)).copy() if n else np.zeros(0, np.int32)
        v = np.ctypeslib.as_array(cv, shape=(n,)).copy() if n else np.zeros(0, np.float32)
        for p in (ic, jc, cv):                      # CSR::dispose() == free() (nlibs/CSR.h:323-327)
            L.free(C.cast(p, C.c_void_p))
        return CSR(v, ci, rp, self.rows, B.cols, n)


def host_api_timed(A, B, reps=3):
    """hip_CSR_SpMM (host arrays in, malloc()ed host arrays out -- what a reference caller of CSR::*spmm gets) timed
    around the C call alone, outputs freed unread.  -> list of per-run dicts (ms_total, ms_h2d, ms_device, ms_d2h, bytes)."""
    import time
    L = lib()
    runs = []
    for _ in range(reps):
        ic, jc, cv, nnz = _I(), _I(), _F(), C.c_int(0)
        t0 = time.perf_counter()
        rc = L.hip_CSR_SpMM(*_host_args(A), *_host_args(B), C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz),
                            A.rows, A.cols, B.cols)
        wall = (time.perf_counter() - t0) * 1e3
        _check(rc, "hip_CSR_SpMM")
        for p in (ic, jc, cv):
            L.free(C.cast(p, C.c_void_p))
        st = HostApiStats()
        _check(L.spgemm_hip_host_api_stats(C.byref(st)), "spgemm_hip_host_api_stats")
        runs.append({"ms_wall": wall, "ms_total": st.ms_total, "ms_h2d": st.ms_h2d, "ms_device": st.ms_device,
                     "ms_d2h": st.ms_d2h, "bytes_h2d": int(st.bytes_h2d), "bytes_d2h": int(st.bytes_d2h), "nnzC": nnz.value})
    return runs


def _dev_args(M):
    return [C.c_void_p(M.rowPtr), C.c_void_p(M.colInd), C.c_void_p(M.values), M.nnz]


def gpuSpMMWrapper(dA, dB, handle=None):
    """CSR gpuSpMMWrapper(const CSR& dA, const CSR& dB) — device CSRs in, device CSR out."""
    assert dA.on_device and dB.on_device
    if dA.cols != dB.rows:
        raise SpgemmError("shape mismatch")
    ic, jc, cv, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    rc = lib().hip_gpuSpMM(handle.ptr if handle else None, *_dev_args(dA), *_dev_args(dB), dA.rows, dA.cols, dB.cols,
                           C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz))
    _check(rc, "hip_gpuSpMM")
    return CSR(cv.value, jc.value, ic.value, dA.rows, dB.cols, nnz.value, True)


def gpu_spmm_raw(handle, IA, JA, VA, nnzA, IB, JB, VB, nnzB, m, k, n):
    """Raw-pointer form for callers that own device memory elsewhere (e.g. torch tensors' data_ptr())."""
    ic, jc, cv, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    rc = lib().hip_gpuSpMM(handle.ptr if handle else None, C.c_void_p(IA), C.c_void_p(JA), C.c_void_p(VA), int(nnzA),
                           C.c_void_p(IB), C.c_void_p(JB), C.c_void_p(VB), int(nnzB), int(m), int(k), int(n),
                           C.byref(ic), C.byref(jc), C.byref(cv), C.byref(nnz))
    _check(rc, "hip_gpuSpMM")
    return ic.value, jc.value, cv.value, nnz.value


def spgemm_symbolic_raw(handle, IA, JA, nnzA, IB, JB, nnzB, m, k, n, IC_out):
    """Phase 1 on raw device pointers: fills IC_out[m+1] (C.rowPtr), returns nnzC."""
    nnz = C.c_int(0)
    _check(lib().hip_spgemm_symbolic(handle.ptr, C.c_void_p(IA), C.c_void_p(JA), int(nnzA), C.c_void_p(IB),
                                     C.c_void_p(JB), int(nnzB), int(m), int(k), int(n), C.c_void_p(IC_out),
                                     C.byref(nnz)), "hip_spgemm_symbolic")
    return nnz.value


def spgemm_numeric_raw(handle, IA, JA, VA, nnzA, IB, JB, VB, nnzB, m, k, n, IC, JC_out, C_out):
    """Phase 2 on raw device pointers: writes JC_out/C_out (caller-owned, nnzC entries each)."""
    _check(lib().hip_spgemm_numeric(handle.ptr, C.c_void_p(IA), C.c_void_p(JA), C.c_void_p(VA), int(nnzA),
                                    C.c_void_p(IB), C.c_void_p(JB), C.c_void_p(VB), int(nnzB), int(m), int(k), int(n),
                                    C.c_void_p(IC), C.c_void_p(JC_out), C.c_void_p(C_out)), "hip_spgemm_numeric")


def d2d(dst, src, nbytes):
    """Device-to-device copy between raw pointers (library pool <-> memory owned elsewhere)."""
    _check(lib().spgemm_hip_memcpy_d2d(C.c_void_p(dst), C.c_void_p(src), int(nbytes)), "spgemm_hip_memcpy_d2d")


def rmcl_prune_raw(handle, m, IC, JC, CV, nnz=None):
    """hip_rmcl_prune on raw device pointers: inflate/prune/normalise the rows of C, compacted into new pool arrays.
    nnz (optional): nnz(C) when the caller knows it (hip_rmcl_prune_n: no device read for it).
    Returns (IN, JN, CN, nnzN); release the three with dev_free."""
    i_, j_, c_, n_ = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    hp = handle.ptr if handle else None
    if nnz is None:
        _check(lib().hip_rmcl_prune(hp, int(m), C.c_void_p(IC), C.c_void_p(JC), C.c_void_p(CV),
                                    C.byref(i_), C.byref(j_), C.byref(c_), C.byref(n_)), "hip_rmcl_prune")
    else:
        _check(lib().hip_rmcl_prune_n(hp, int(m), int(nnz), C.c_void_p(IC), C.c_void_p(JC), C.c_void_p(CV),
                                      C.byref(i_), C.byref(j_), C.byref(c_), C.byref(n_)), "hip_rmcl_prune_n")
    return i_.value, j_.value, c_.value, n_.value


def rmcl_expand_prune_raw(handle, IA, JA, VA, nnzA, IB, JB, VB, nnzB, m, k, n):
    """hip_rmcl_expand_prune on raw device pointers: prune(A*B) of one R-MCL iteration without materialising the product.
    Returns (IN, JN, CN, nnzN) in pool arrays; release the three with dev_free."""
    i_, j_, c_, n_ = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    _check(lib().hip_rmcl_expand_prune(handle.ptr if handle else None, C.c_void_p(IA), C.c_void_p(JA), C.c_void_p(VA),
                                       int(nnzA), C.c_void_p(IB), C.c_void_p(JB), C.c_void_p(VB), int(nnzB),
                                       int(m), int(k), int(n), C.byref(i_), C.byref(j_), C.byref(c_), C.byref(n_)),
           "hip_rmcl_expand_prune")
    return i_.value, j_.value, c_.value, n_.value


def pool_trim(device=0):
    """idle cached blocks of the device go back to the driver"""
    _check(lib().spgemm_hip_pool_trim(int(device)), "spgemm_hip_pool_trim")


def pool_cached_bytes(device=0):
    n = C.c_size_t(0)
    _check(lib().spgemm_hip_pool_cached_bytes(int(device), C.byref(n)), "spgemm_hip_pool_cached_bytes")
    return int(n.value)


COO_DEDUPE, COO_SELF_LOOPS, COO_ROW_NORMALISE, COO_ABS = 1, 2, 4, 8


def flopsStats(dA, dB, handle=None):
    """std::vector<int> flopsStats(...) (nlibs/tools/stats.cc:45-55) for device CSRs: 13 power-of-two buckets."""
    assert dA.on_device and dB.on_device
    out = (C.c_int * 13)()
    _check(lib().hip_flopsStats(handle.ptr if handle else None, C.c_void_p(dA.rowPtr), C.c_void_p(dA.colInd),
                                C.c_void_p(dB.rowPtr), dA.rows, out), "hip_flopsStats")
    return [int(x) for x in out]


def nnzStats(dA, handle=None):
    """CSR::nnzStats (nlibs/CSR.cc:241-248) of a device CSR: 18 power-of-two buckets of the row lengths."""
    out = (C.c_int * 18)()
    _check(lib().hip_nnzStats(handle.ptr if handle else None, C.c_void_p(dA.rowPtr), int(dA.rows), out), "hip_nnzStats")
    return [int(x) for x in out]


def resultsComparison(hC, rC, hv, hqueue, rel=1e-6):
    """resultsComparison / isPartialRawEqual (mindex2-cuda/nGpuSpMM.cc:85-240) on host CSRs: per reference bin
    {rows, rows_differ, first_bad_row, max_rel_err}.  hv/hqueue: outputs of gpuFlopsClassify (queue on the host)."""
    rep = (BinReport * (HV_LEN - 1))()
    hvv = (C.c_int * HV_LEN)(*([int(x) for x in hv] + [int(hv[-1])] * (HV_LEN - len(hv))))
    q = np.ascontiguousarray(hqueue, dtype=np.int32)
    a = [np.ascontiguousarray(x, dtype=t) for x, t in ((hC.rowPtr, np.int32), (hC.colInd, np.int32), (hC.values, np.float32),
                                                        (rC.rowPtr, np.int32), (rC.colInd, np.int32), (rC.values, np.float32))]
    p = lambda arr, ty: arr.ctypes.data_as(ty)
    _check(lib().hip_resultsComparison(int(hC.rows), int(hC.cols), p(a[0], _I), p(a[1], _I), p(a[2], _F), p(a[3], _I),
                                       p(a[4], _I), p(a[5], _F), hvv, len(hv), p(q, _I), float(rel), rep),
           "hip_resultsComparison")
    return [{"rows": r.rows, "rows_differ": r.rows_differ, "first_bad_row": r.first_bad_row, "max_rel_err": r.max_rel_err}
            for r in rep]


def coo_to_csr_raw(handle, rows, cols, nnz, dRow, dCol, dVal, flags):
    """hip_coo_to_csr on raw device pointers -> (dIA, dJA, dA, nnz) from the library pool (release with dev_free)."""
    ia, ja, av, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    _check(lib().hip_coo_to_csr(handle.ptr if handle else None, int(rows), int(cols), int(nnz), C.c_void_p(dRow),
                                C.c_void_p(dCol), C.c_void_p(dVal), int(flags), C.byref(ia), C.byref(ja), C.byref(av),
                                C.byref(n)), "hip_coo_to_csr")
    return ia.value, ja.value, av.value, n.value


def coo_to_csr(rows, cols, ri, ci, v, flags, handle=None):
    """Host COO arrays in, device CSR out (COO::toCSR and friends, on the device): returns a device `CSR`.
    flags: COO_DEDUPE | COO_SELF_LOOPS | COO_ROW_NORMALISE | COO_ABS (rmclInit = SELF_LOOPS | ROW_NORMALISE)."""
    ri = np.ascontiguousarray(ri, dtype=np.int32)
    ci = np.ascontiguousarray(ci, dtype=np.int32)
    v = np.ascontiguousarray(v, dtype=np.float32)
    dr, dc, dv = h2d(ri), h2d(ci), h2d(v)
    try:
        ia, ja, av, n = coo_to_csr_raw(handle, rows, cols, len(ri), dr, dc, dv, flags)
    finally:
        for p in (dr, dc, dv):
            dev_free(p)
    return CSR(av, ja, ia, rows, cols, n, True)


def row_flops_raw(handle, IA, JA, IB, m, out_ptr):
    """Per-row product counts into a device int[m]; returns P."""
    tot = C.c_longlong(0)
    _check(lib().hip_csr_row_flops(handle.ptr if handle else None, C.c_void_p(IA), C.c_void_p(JA), C.c_void_p(IB),
                                   int(m), C.c_void_p(out_ptr), C.byref(tot)), "hip_csr_row_flops")
    return tot.value


def gpuFlopsClassify(dA, dB, handle=None):
    """-> (hv list, drowIds devptr, dflops devptr, total_flops).  hv has the reference's length (max bin + 2)."""
    assert dA.on_device and dB.on_device
    ids, fl = C.c_void_p(), C.c_void_p()
    hv = (C.c_int * HV_LEN)()
    hv_len, tot = C.c_int(0), C.c_longlong(0)
    rc = lib().hip_gpuFlopsClassify(handle.ptr if handle else None, C.c_void_p(dA.rowPtr), C.c_void_p(dA.colInd),
                                    C.c_void_p(dB.rowPtr), dA.rows, dA.cols, C.byref(ids), C.byref(fl), hv,
                                    C.byref(hv_len), C.byref(tot))
    _check(rc, "hip_gpuFlopsClassify")
    return [int(x) for x in hv], hv_len.value, ids.value, fl.value, tot.value


def sgpuSpMMWrapper(dA, dB, drowIds, hv, dflops, handle=None):
    assert dA.on_device and dB.on_device
    hv_arr = (C.c_int * HV_LEN)(*hv)
    ic, jc, cv, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    rc = lib().hip_sgpuSpMM(handle.ptr if handle else None, *_dev_args(dA), *_dev_args(dB), dA.rows, dA.cols, dB.cols,
                            C.c_void_p(drowIds), hv_arr, C.c_void_p(dflops), C.byref(ic), C.byref(jc), C.byref(cv),
                            C.byref(nnz))
    _check(rc, "hip_sgpuSpMM")
    return CSR(cv.value, jc.value, ic.value, dA.rows, dB.cols, nnz.value, True)


def scudaSpMM(hA, hB, handle=None):
    """Host CSRs in, host CSR out via classify + binned SpGEMM (mindex2-cuda/nGpuSpMM.cc:245-279)."""
    dA = hA.toGpuCSR()
    dB = dA if hB is hA else hB.toGpuCSR()
    try:
        hv, hv_len, ids, fl, _ = gpuFlopsClassify(dA, dB, handle)
        try:
            dC = sgpuSpMMWrapper(dA, dB, ids, hv, fl, handle)
        finally:
            dev_free(ids)
            dev_free(fl)
        hC = dC.toCpuCSR()
        dC.deviceDispose()
        return hC
    finally:
        dA.deviceDispose()
        if dB is not dA:
            dB.deviceDispose()


def rmcl_iter_device_raw(handle, maxIter, rows, cols, gI, gJ, gV, gnnz, tI, tJ, tV, tnnz):
    """hip_gpuRmclIter_device on raw device pointers: maxIter iterations Mt <- prune(Mgt * Mt) without packing Mt between
    them.  Returns (I, J, V, nnz) of the final Mt in pool arrays; release the three with dev_free."""
    i_, j_, c_, n_ = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    _check(lib().hip_gpuRmclIter_device(handle.ptr if handle else None, int(maxIter), int(rows), int(cols), C.c_void_p(gI),
                                        C.c_void_p(gJ), C.c_void_p(gV), int(gnnz), C.c_void_p(tI), C.c_void_p(tJ),
                                        C.c_void_p(tV), int(tnnz), C.byref(i_), C.byref(j_), C.byref(c_), C.byref(n_)),
           "hip_gpuRmclIter_device")
    return i_.value, j_.value, c_.value, n_.value


def gpuRmclIter_device(maxIter, Mgt, Mt, handle=None):
    """hip_gpuRmclIter_device on device CSRs: returns the new Mt as a device CSR (the inputs are left alone)."""
    assert Mgt.on_device and Mt.on_device
    i_, j_, c_, n_ = rmcl_iter_device_raw(handle, maxIter, Mgt.rows, Mgt.cols, Mgt.rowPtr, Mgt.colInd, Mgt.values, Mgt.nnz,
                                          Mt.rowPtr, Mt.colInd, Mt.values, Mt.nnz)
    return CSR(c_, j_, i_, Mt.rows, Mt.cols, n_, on_device=True)


def gpuRmclIter(maxIter, Mgt, Mt):
    """void gpuRmclIter(const int maxIter, const CSR Mgt, CSR& Mt): returns the new Mt (host CSR)."""
    assert not Mgt.on_device and not Mt.on_device
    L = lib()
    oi, oj, ov, on = _I(), _I(), _F(), C.c_int(0)
    rc = L.hip_gpuRmclIter(int(maxIter), Mgt.rows, Mgt.cols, Mgt.rowPtr.ctypes.data_as(_I), Mgt.colInd.ctypes.data_as(_I),
                           Mgt.values.ctypes.data_as(_F), Mgt.nnz, Mt.rowPtr.ctypes.data_as(_I),
                           Mt.colInd.ctypes.data_as(_I), Mt.values.ctypes.data_as(_F), Mt.nnz,
                           C.byref(oi), C.byref(oj), C.byref(ov), C.byref(on))
    _check(rc, "hip_gpuRmclIter")
    n = on.value
    rp = np.ctypeslib.as_array(oi, shape=(Mt.rows + 1,)).copy()
    ci = np.ctypeslib.as_array(oj, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    v = np.ctypeslib.as_array(ov, shape=(n,)).copy() if n else np.zeros(0, np.float32)
    for p in (oi, oj, ov):
        L.free(C.cast(p, C.c_void_p))
    return CSR(v, ci, rp, Mt.rows, Mt.cols, n)


# ---------------------------------------------------------------------------------------------
# multi-GPU behind the C ABI (include/spgemm_hip.h "multi-GPU"): groups of shards, sharded SpGEMM, sharded R-MCL
# ---------------------------------------------------------------------------------------------
def rccl_available():
    """True when the library could load librccl into this process"""
    return lib().spgemm_hip_rccl_available() == 0


def unique_id():
    """128-byte RCCL id made by rank 0 of a multi-process job; hand it to every rank (bytes)."""
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _check(lib().spgemm_hip_unique_id(buf), "spgemm_hip_unique_id")
    return buf.raw


class Group:
    """spgemm_group.  Group(nshards, devices=None, transport=XCHG_AUTO): every shard in this process (several may share
    a device).  Group.of_rank(nranks, rank, device, id): one process per GPU, wired with unique_id()."""

    def __init__(self, nshards, devices=None, transport=XCHG_AUTO, _ptr=None):
        self._g = C.c_void_p()
        if _ptr is not None:
            self._g = _ptr
        else:
            dv = None
            if devices is not None:
                dv = (C.c_int * len(devices))(*[int(d) for d in devices])
                assert len(devices) == nshards
            _check(lib().spgemm_hip_group_create(C.byref(self._g), int(nshards), dv, int(transport)), "spgemm_hip_group_create")
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(lib().spgemm_hip_group_info(self._g, C.byref(a), C.byref(b), C.byref(c)), "spgemm_hip_group_info")
        self.nranks, self.nlocal, self.transport = a.value, b.value, c.value
        # jobs made on this group and handles borrowed from it: the C side of a job dereferences the group when it is
        # destroyed, and a borrowed Handle points at a handle the group owns -- close() takes them down first, in that order
        self._jobs = weakref.WeakSet()
        self._borrowed = weakref.WeakSet()
        _LIVE.add(self)

    @staticmethod
    def of_rank(nranks, rank, device, id128):
        g = C.c_void_p()
        buf = C.create_string_buffer(bytes(id128), UNIQUE_ID_BYTES)
        _check(lib().spgemm_hip_group_create_rank(C.byref(g), int(nranks), int(rank), int(device), buf),
               "spgemm_hip_group_create_rank")
        return Group(nranks, _ptr=g)

    @property
    def ptr(self):
        return self._g

    def close(self):
        if self._g:
            for j in list(self._jobs):
                j.close()
            for h in list(self._borrowed):
                h._h = C.c_void_p()                   # the handle dies with the group: a later call fails with "null argument"
            lib().spgemm_hip_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _host_args(M):
    return [M.rowPtr.ctypes.data_as(_I), M.colInd.ctypes.data_as(_I), M.values.ctypes.data_as(_F), int(M.nnz)]


def _take_malloced(L, pi, pj, pv, rows, n):
    rp = np.ctypeslib.as_array(pi, shape=(rows + 1,)).copy()
    ci = np.ctypeslib.as_array(pj, shape=(n,)).copy() if n else np.zeros(0, np.int32)
    v = np.ctypeslib.as_array(pv, shape=(n,)).copy() if n else np.zeros(0, np.float32)
    for p in (pi, pj, pv):
        L.free(C.cast(p, C.c_void_p))
    return rp, ci, v


class ShardedSpMM:
    """spgemm_sharded: C = A*B row-sharded over a Group with the operands resident (host CSRs in; B=None means B=A)."""

    def __init__(self, group, A, B=None):
        assert not A.on_device and (B is None or not B.on_device)
        self.group = group
        self._keep = (A, B)
        self._j = C.c_void_p()
        bargs = _host_args(B) if B is not None else [None, None, None, 0]
        _check(lib().hip_sharded_spmm_create(group.ptr, *_host_args(A), *bargs, A.rows, A.cols,
                                             (B.cols if B is not None else A.cols), C.byref(self._j)), "hip_sharded_spmm_create")
        self.m, self.n = A.rows, (B.cols if B is not None else A.cols)
        group._jobs.add(self)
        _LIVE.add(self)

    def step(self, gather=True):
        """-> (nnzC, total products P)"""
        nz, P = C.c_longlong(0), C.c_longlong(0)
        _check(lib().hip_sharded_spmm_step(self._j, int(bool(gather)), C.byref(nz), C.byref(P)), "hip_sharded_spmm_step")
        return nz.value, P.value

    def result(self, local_shard=0):
        L = lib()
        pi, pj, pv, n, rows = _I(), _I(), _F(), C.c_int(0), C.c_int(0)
        _check(L.hip_sharded_spmm_result(self._j, int(local_shard), C.byref(pi), C.byref(pj), C.byref(pv), C.byref(n),
                                         C.byref(rows)), "hip_sharded_spmm_result")
        rp, ci, v = _take_malloced(L, pi, pj, pv, rows.value, n.value)
        return CSR(v, ci, rp, rows.value, self.n, n.value)

    def handle(self, local_shard=0):
        """the spgemm_handle of a local shard as a (non-owning) Handle: stats(), set_kernel_timing()"""
        p = lib().hip_sharded_spmm_handle(self._j, int(local_shard))
        if not p:
            raise SpgemmError("no such local shard")
        h = Handle(_borrowed=p)
        self.group._borrowed.add(h)
        return h

    def info(self):
        ends = (C.c_int * (self.group.nranks + 1))()
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().hip_sharded_spmm_info(self._j, ends, C.byref(a), C.byref(b)), "hip_sharded_spmm_info")
        return {"ends": [int(x) for x in ends], "ms_compute": a.value, "ms_exchange": b.value}

    def close(self):
        if self._j:
            lib().hip_sharded_spmm_destroy(self._j)
            self._j = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedRmcl:
    """spgemm_sharded_rmcl: the R-MCL loop over a Group with the operands resident (host CSRs in once; every run() starts
    from the initial Mt and leaves the result on the shards)."""

    def __init__(self, group, Mgt, Mt):
        assert not Mgt.on_device and not Mt.on_device
        self.group = group
        self._keep = (Mgt, Mt)
        self.rows, self.cols = Mt.rows, Mt.cols
        self._j = C.c_void_p()
        _check(lib().hip_sharded_rmcl_create(group.ptr, Mgt.rows, Mgt.cols, *_host_args(Mgt), *_host_args(Mt), C.byref(self._j)),
               "hip_sharded_rmcl_create")
        group._jobs.add(self)
        _LIVE.add(self)

    def run(self, maxIter, restart=True):
        """maxIter iterations from the initial Mt (restart=False: from the previous run's result) -> nnz of the final Mt"""
        n = C.c_int(0)
        fn = lib().hip_sharded_rmcl_run if restart else lib().hip_sharded_rmcl_continue
        _check(fn(self._j, int(maxIter), C.byref(n)), "hip_sharded_rmcl_run")
        return n.value

    def iter_nnz(self):
        buf = (C.c_longlong * 256)()
        n = lib().hip_sharded_rmcl_iter_nnz(self._j, buf, 256)
        return [int(buf[i]) for i in range(min(n, 256))]

    def result(self, local_shard=0):
        L = lib()
        pi, pj, pv, n = _I(), _I(), _F(), C.c_int(0)
        _check(L.hip_sharded_rmcl_result(self._j, int(local_shard), C.byref(pi), C.byref(pj), C.byref(pv), C.byref(n)),
               "hip_sharded_rmcl_result")
        rp, ci, v = _take_malloced(L, pi, pj, pv, self.rows, n.value)
        return CSR(v, ci, rp, self.rows, self.cols, n.value)

    def ends(self):
        e = (C.c_int * (self.group.nranks + 1))()
        _check(lib().hip_sharded_rmcl_info(self._j, e), "hip_sharded_rmcl_info")
        return [int(x) for x in e]

    def close(self):
        if self._j:
            lib().hip_sharded_rmcl_destroy(self._j)
            self._j = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gpuRmclIter_sharded(group, maxIter, Mgt, Mt):
    """gpuRmclIter over a Group (hip_gpuRmclIter_sharded): returns the new Mt (host CSR)."""
    assert not Mgt.on_device and not Mt.on_device
    L = lib()
    oi, oj, ov, on = _I(), _I(), _F(), C.c_int(0)
    rc = L.hip_gpuRmclIter_sharded(group.ptr, int(maxIter), Mgt.rows, Mgt.cols, *_host_args(Mgt), *_host_args(Mt),
                                   C.byref(oi), C.byref(oj), C.byref(ov), C.byref(on))
    _check(rc, "hip_gpuRmclIter_sharded")
    rp, ci, v = _take_malloced(L, oi, oj, ov, Mt.rows, on.value)
    return CSR(v, ci, rp, Mt.rows, Mt.cols, on.value)


def sort_rows_device(dC, handle=None):
    _check(lib().hip_csr_sort_rows(handle.ptr if handle else None, dC.rows, C.c_void_p(dC.rowPtr),
                                   C.c_void_p(dC.colInd), C.c_void_p(dC.values)), "hip_csr_sort_rows")
