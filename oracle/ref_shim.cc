// oracle/ref_shim.cc — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" doorway onto the *real* reference CPU kernels, compiled from
// the sources where they lie under /root/reference (see oracle/Makefile,
// target `ref`).  Output: oracle/_ref/libref.so (git-ignored).  It is used
//   (1) to pin oracle/oracle.c (our CPU restatement) bit-for-bit, and
//   (2) to generate tests/golden/* (tests/golden/make_golden.py), and
//   (3) optionally as bench.py's cpu_baseline (kind "reference").
// Nothing here is copied from the reference: this file only *calls* it.
//
// Reference entry points bound here:
//   sequential_CSR_SpMM        nlibs/cpu_csr_kernel.cc:76-119
//   omp_CSR_SpMM               nlibs/omp_csr_kernel.cc:295-315
//   static_omp_CSR_SpMM        nlibs/static_omp_csr_kernel.cc:186-206
//   flops_omp_CSR_SpMM         nlibs/flops_csr_kernel.cc:122-142
//   group_CSR_SpMM             nlibs/group_csr_kernel.cc:108-128
//   dynamic_omp_CSR_flops      nlibs/flops_csr_kernel.cc:14-31
//   group_CSR_flops            nlibs/group_csr_kernel.cc:10-52
//   arrayEqualPartition64      nlibs/tools/util.cc:123-135
//   dynamic_omp_CSR_IC_nnzC_footprints / arrayEqualPartition   nlibs/static_omp_csr_kernel.cc:28-95, tools/util.cc:109-121
//   COO::readSNAPFile/orderedAndDuplicatesRemoving/toCSR  nlibs/COO.cc:48-291
//   rmclInit / RMCL            nlibs/qrmcl.cc:126-164
//   mtRmclIter                 nlibs/qrmcl.cc:8-84   (the multi-threaded loop on in-memory CSRs: bench.py's R-MCL cpu_baseline)
//   CSR::makeOrdered / isEqual nlibs/CSR.cc:73-86, nlibs/CSR.h:195-245
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include "CSR.h"
#include "COO.h"
#include "qrmcl.h"
#include "process_args.h"
#include "tools/util.h"

// not declared in any reference header (file-local convention there)
void group_CSR_flops(const int IA[], const int JA[], const int IB[], const int JB[],
                     const int m, const int n, int* IC, int& nnzC, int* rowFlops,
                     int* groups, int* tops, const int stride);

void mtRmclIter(const int maxIter, const CSR Mgt, CSR& Mt, const int stride, const RunOptions runOptions);   // nlibs/qrmcl.cc:8

extern "C" {

// which: 0 sequential, 1 omp, 2 static_omp, 3 flops_omp, 4 group, 5 noindex_somp
int ref_spmm(int which, const int* IA, const int* JA, const float* A, int nnzA,
             const int* IB, const int* JB, const float* B, int nnzB,
             int** IC, int** JC, float** C, int* nnzC, int m, int k, int n, int stride) {
  int *ic = NULL, *jc = NULL; float* c = NULL; int nz = 0;
  switch (which) {
    case 0: sequential_CSR_SpMM(IA, JA, A, nnzA, IB, JB, B, nnzB, ic, jc, c, nz, m, k, n); break;
    case 1: omp_CSR_SpMM(IA, JA, A, nnzA, IB, JB, B, nnzB, ic, jc, c, nz, m, k, n, stride); break;
    case 2: static_omp_CSR_SpMM(IA, JA, A, nnzA, IB, JB, B, nnzB, ic, jc, c, nz, m, k, n, stride); break;
    case 3: flops_omp_CSR_SpMM(IA, JA, A, nnzA, IB, JB, B, nnzB, ic, jc, c, nz, m, k, n, stride); break;
    case 4: group_CSR_SpMM(IA, JA, A, nnzA, IB, JB, B, nnzB, ic, jc, c, nz, m, k, n, stride); break;
    case 5: noindex_somp_CSR_SpMM(IA, JA, A, nnzA, IB, JB, B, nnzB, ic, jc, c, nz, m, k, n, stride); break;
    default: return -1;
  }
  *IC = ic; *JC = jc; *C = c; *nnzC = nz;
  return 0;
}

// IC[m+1] <- C.rowPtr, fp[m+1] <- exclusive prefix of the per-row "footprints" the static scheduler balances
// (static_omp_CSR_SpMM, nlibs/static_omp_csr_kernel.cc:106-130)
void ref_footprints(const int* IA, const int* JA, const int* IB, const int* JB, int m, int n, int* IC, int* fp, int stride) {
  int nnzC = 0;
#pragma omp parallel
  {
    thread_data_t td(n);
    dynamic_omp_CSR_IC_nnzC_footprints(IA, JA, IB, JB, m, n, td, IC, nnzC, fp, stride);
  }
}

// ends[parts+1] <- arrayEqualPartition on an int prefix array (the footprint cut)
void ref_equal_partition(int* prefix, int n, int parts, int* ends) { arrayEqualPartition(prefix, n, parts, ends); }

// rowFlops[m+1] <- exclusive prefix of per-row product counts (long), as the reference leaves it
void ref_row_flops_prefix(const int* IA, const int* JA, const int* IB, const int* JB,
                          int m, int n, long* rowFlops, int stride) {
#pragma omp parallel
  { dynamic_omp_CSR_flops(IA, JA, IB, JB, m, n, rowFlops, stride); }
}

// rowFlops[m] (int), groups[m] (row ids bin by bin), tops[8]; IC[m+1] gets 0/1 for bins 0/1
void ref_group_flops(const int* IA, const int* JA, const int* IB, const int* JB,
                     int m, int n, int* IC, int* rowFlops, int* groups, int* tops, int stride) {
  int nnzC = 0;
  group_CSR_flops(IA, JA, IB, JB, m, n, IC, nnzC, rowFlops, groups, tops, stride);
}

void ref_equal_partition64(long* prefix, int n, int parts, int* ends) {
  arrayEqualPartition64(prefix, n, parts, ends);
}

static void csr_out(CSR& c, int* rows, int* cols, int* nnz, int** rp, int** ci, float** v) {
  *rows = c.rows; *cols = c.cols; *nnz = c.nnz; *rp = c.rowPtr; *ci = c.colInd; *v = c.values;
}

// mode 0: readSNAPFile(f,isTrans) + orderedAndDuplicatesRemoving + toCSR  (mindex2-cuda/nGpuSpMM.cc:285-290)
// mode 1: readSNAPFile(f,isTrans) + rmclInit                               (tests/testGpuSpMM.cc:13-15)
// toAbs != 0 applies CSR::toAbs (nGpuSpMM.cc:291)
int ref_load(const char* fname, int isTrans, int mode, int toAbs,
             int* rows, int* cols, int* nnz, int** rp, int** ci, float** v) {
  COO coo;
  coo.readSNAPFile(fname, isTrans != 0);
  CSR a;
  if (mode == 0) { coo.orderedAndDuplicatesRemoving(); a = coo.toCSR(); }
  else { a = rmclInit(coo); }
  if (toAbs) a.toAbs();
  coo.dispose();
  csr_out(a, rows, cols, nnz, rp, ci, v);
  return 0;
}

// runOption: 0 SEQ, 1 OMP, 4 SOMP (enum RunOptions, nlibs/qrmcl.h:8)
int ref_rmcl(const char* fname, int maxIters, int runOption,
             int* rows, int* cols, int* nnz, int** rp, int** ci, float** v) {
  CSR mt = RMCL(fname, maxIters, (RunOptions)runOption);
  csr_out(mt, rows, cols, nnz, rp, ci, v);
  return 0;
}

// mtRmclIter(maxIter, Mgt, Mt, stride, runOption) on CSRs that are already in memory (no file parsing inside the timed
// call).  Mt is replaced in place by the reference (old arrays disposed, new ones malloc()ed), so both operands are deep
// copies of the caller's arrays.  Returns the seconds spent inside mtRmclIter; the result comes back like ref_rmcl's.
double ref_mt_rmcl_iter(int maxIters, int runOption, int stride, int rows, int cols,
                        const int* gI, const int* gJ, const float* gV, int gnnz,
                        const int* tI, const int* tJ, const float* tV, int tnnz,
                        int* orows, int* ocols, int* onnz, int** rp, int** ci, float** v) {
  CSR g((QValue*)gV, (int*)gJ, (int*)gI, rows, cols, gnnz), t((QValue*)tV, (int*)tJ, (int*)tI, rows, cols, tnnz);
  CSR Mgt = g.deepCopy(), Mt = t.deepCopy();
  const double t0 = omp_get_wtime();
  mtRmclIter(maxIters, Mgt, Mt, stride, (RunOptions)runOption);
  const double dt = omp_get_wtime() - t0;
  Mgt.dispose();
  csr_out(Mt, orows, ocols, onnz, rp, ci, v);
  return dt;
}

void ref_make_ordered(int rows, int cols, int nnz, int* rp, int* ci, float* v) {
  CSR c(v, ci, rp, rows, cols, nnz);
  c.makeOrdered();
}

int ref_is_equal(int rows, int cols, int nnzA, int* rpA, int* ciA, float* vA,
                 int nnzB, int* rpB, int* ciB, float* vB) {
  CSR a(vA, ciA, rpA, rows, cols, nnzA), b(vB, ciB, rpB, rows, cols, nnzB);
  return a.isEqual(b) ? 1 : 0;
}

// R-MCL prune math on one row (nlibs/tools/util.cc:4-69), for pinning the restatement
float ref_compute_threshold(float avg, float mx) { return computeThreshold(avg, mx); }

void ref_free(void* p) { free(p); }
int ref_max_threads(void) { return omp_get_max_threads(); }

}  // extern "C"
