// CSR.h — host-side mirror of the reference's `struct CSR` (nlibs/CSR.h:23-50) for the SpGEMM path.
// Same fields, same ownership rules (three malloc()ed arrays on the host, device arrays in the SAME struct after
// toGpuCSR(), dispose()==free(), deviceDispose()==device free), same method names, so a driver written against the
// reference compiles against this header.  The arithmetic lives in libspgemm_hip.so (include/spgemm_hip.h); nothing
// here computes a product on the CPU.
#ifndef SMF_CSR_H_
#define SMF_CSR_H_
#include <vector>
#include "tools/macro.h"

struct CSR {
  QValue* values;   // nnz
  int* colInd;      // nnz, 0-based, unsorted inside a row unless makeOrdered() was called
  int* rowPtr;      // rows + 1
  int rows, cols, nnz;

  CSR() : values(0), colInd(0), rowPtr(0), rows(0), cols(0), nnz(0) {}
  CSR(QValue* v, int* ci, int* rp, int r, int c, int nz) { init(v, ci, rp, r, c, nz); }
  void init(QValue* v, int* ci, int* rp, int r, int c, int nz) { values = v; colInd = ci; rowPtr = rp; rows = r; cols = c; nnz = nz; }

  CSR deepCopy() const;                 // nlibs/CSR.cc:97-106
  void makeOrdered();                   // nlibs/CSR.cc:73-86: sort every row by (col, value)
  void toAbs();                         // nlibs/CSR.h:152-158
  void averAndNormRowQValue();          // nlibs/CSR.cc:88-95
  int rowCount(int r) const { return rowPtr[r + 1] - rowPtr[r]; }
  void dispose();                       // nlibs/CSR.h:323-327: free() x3
  void output(const char* msg) const;   // nlibs/CSR.h:110-130 (zero-based form)

  // comparison.  isEqual keeps the reference's semantics (nlibs/CSR.h:195-245: dims, rowPtr, |dv| <= 1e-7 through a
  // dense row scatter).  isParityEqual is the stricter rule this project is judged on: rowPtr and colInd identical
  // (both operands ordered), values within `rel` relative.
  bool isEqual(const CSR& B) const;
  bool isParityEqual(const CSR& B, double rel = 1e-6) const;

  // device mirror, nlibs/CSR.cc:342-379 (cudaMalloc/cudaMemcpy there, spgemm_hip_malloc/memcpy here)
  CSR toGpuCSR() const;
  CSR toCpuCSR() const;
  void deviceDispose();

  // C = this * B on the MI355X: sits where CSR::spmm / omp_spmm / somp_spmm / flops_spmm / group_spmm sit
  // (nlibs/CSR.cc:59-208).  Host CSRs in, host CSR out (arrays malloc()ed).  `stride` is the CPU scheduling knob of
  // the reference's signatures; accepted and ignored.  Exits on error like the reference's GPU path.
  CSR hip_spmm(const CSR& B, const int stride = 512) const;

  // 2 * (number of intermediate products), the reference's original getSpMMFlops (nlibs/cpu_csr_kernel.cc:39-56)
  long long spMMFlops(const CSR& B) const;

  // 18 power-of-two buckets of the row lengths (nlibs/CSR.cc:241-248); host CSR: counted here, device CSR
  // (after toGpuCSR): hip_nnzStats
  std::vector<int> nnzStats() const;
  std::vector<int> gpuNnzStats() const;
  // how the row lengths moved from *this to B (nlibs/CSR.cc:381-415; the drift report of the R-MCL loop under --stats):
  // counts[k] = rows whose relative change (len_B - len_A) / len_A is below percents[k] (first such k), counts[n] = the
  // rest, then three more: rows that appeared (0 -> >0), rows empty in both, rows unchanged.  n = percents.size().
  std::vector<int> differsStats(const CSR& B, const std::vector<QValue>& percents) const;
};
#endif
