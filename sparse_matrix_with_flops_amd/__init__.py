"""MI355X-native CSR x CSR SpGEMM (the rMCL expansion step) behind the reference's kernel call surface.

Product code:  csrc/ (HIP kernels + C ABI -> libspgemm_hip.so), hipspgemm.py (ctypes mirror of the
reference's CSR / gpuSpMMWrapper / gpuFlopsClassify surface), dist.py (row-sharded multi-GPU), synth.py
(deterministic synthetic inputs).  The CPU oracle lives in /oracle and is never imported from here.
"""
from . import synth  # noqa: F401
from .hipspgemm import (CSR, Handle, SpgemmError, device_count, gpuFlopsClassify, gpuSpMMWrapper,  # noqa: F401
                        scudaSpMM, sgpuSpMMWrapper)
