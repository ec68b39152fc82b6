// gpu_csr_kernel.h — the GPU call surface the reference's drivers link, re-hosted on libspgemm_hip.so.
//   gpuSpMMWrapper / gpuRmclIter          nlibs/gpus/gpu_csr_kernel.h:5-6 (.cu:128-173, :281-311)
//   gpuFlopsClassify / sgpuSpMMWrapper /
//   scudaSpMM                             mindex2-cuda/nGpuSpMM.cc:19,242,245
//   hip_CSR_SpMM (reference-style refs)   the *_CSR_SpMM family, nlibs/cpu_csr_kernel.h:63-102
// Errors: like HANDLE_ERROR (nlibs/gpus/cuda_handle_error.h:7-15) these wrappers print and exit(EXIT_FAILURE);
// use the C ABI (include/spgemm_hip.h) directly for status codes.
#ifndef SMF_GPU_CSR_KERNEL_H_
#define SMF_GPU_CSR_KERNEL_H_
#include <vector>
#include "../CSR.h"

CSR gpuSpMMWrapper(const CSR& dA, const CSR& dB);
std::vector<int> gpuFlopsClassify(const CSR& dA, const CSR& dB, int** drowIdsp, int** dflopId);
CSR sgpuSpMMWrapper(const CSR& dA, const CSR& dB, int* drowIds, const std::vector<int>& hv, int* dflops);
CSR scudaSpMM(const CSR& hA, const CSR& hB);
void gpuRmclIter(const int maxIter, const CSR Mgt, CSR& Mt);
// Several GPUs (no counterpart in the reference, which is single-device: SURVEY.md section 2.4; the seam is the flops-balanced
// row cut its CPU kernels make for their threads, nlibs/tools/util.cc:123-135).  HOST CSRs in, HOST CSR out: rows of A cut
// into `shards` blocks of equal flops, one shard per device (shards > devices: logical shards share devices), B replicated,
// the row segments of C gathered over RCCL / peer copies (include/spgemm_hip.h "multi-GPU").  shards <= 0: one per device.
CSR gpuShardedSpMM(const CSR& hA, const CSR& hB, int shards = 0);
// gpuRmclIter over an explicit number of shards (gpuRmclIter itself uses one per visible device)
void gpuShardedRmclIter(const int maxIter, const CSR Mgt, CSR& Mt, int shards);
// debug dump of a DEVICE CSR (nlibs/gpus/gpu_csr_kernel.h:7, .cu:15-42: message, shape and device pointers, the triples,
// then the raw rowPtr and colInd/values arrays); the arrays are brought to the host and printed there
void gpuOutputCSRWrapper(const CSR dA, const char* msg);

// bool resultsComparison(CSR& hC, CSR& rC, const vector<int>& hv, const int* hqueue) (mindex2-cuda/nGpuSpMM.cc:138-240):
// hC (result under test) against rC (reference result), whole matrix first, then bin by bin; prints one line per bin
// (rows compared / rows that differ / first such row / largest relative value error) and returns true iff nothing
// differs.  hv and hqueue (HOST copy of the row queue) are the outputs of gpuFlopsClassify.
bool resultsComparison(const CSR& hC, const CSR& rC, const std::vector<int>& hv, const int* hqueue, double rel = 1e-6);

// raw arrays, the signature every CPU kernel of the reference has
void hip_CSR_SpMM(const int IA[], const int JA[], const QValue A[], const int nnzA,
                  const int IB[], const int JB[], const QValue B[], const int nnzB,
                  int*& IC, int*& JC, QValue*& C, int& nnzC,
                  const int m, const int k, const int n, const int stride = 512);
#endif
