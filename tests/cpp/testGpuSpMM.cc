// tests/cpp/testGpuSpMM.cc — the reference's own GPU test protocol (tests/testGpuSpMM.cc:9-46) against this
// project's C++ mirror:  load -> toGpuCSR -> gpuSpMMWrapper -> toCpuCSR -> makeOrdered -> compare with the CPU
// result.  The CPU result comes from the ORACLE (oracle/liboracle.so: test infrastructure, allowed here, never linked
// into the product).  Also exercises hip_spmm (host in/out), scudaSpMM (classify + binned path) and, with
// "--rmcl N", RMCL(file, N, GPU) vs the oracle's seqRmclIter restatement; the row-sharded forms (gpuShardedSpMM,
// gpuShardedRmclIter) over 2-3 logical shards.
//   usage: testGpuSpMM <file> [--rmcl N]        prints Same / Differs per check, exit code 0 iff all Same
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "COO.h"
#include "CSR.h"
#include "gpus/gpu_csr_kernel.h"
#include "qrmcl.h"
#include "tools/stats.h"
#include "../../include/spgemm_hip.h"

extern "C" {
int oracle_sequential_spmm(const int*, const int*, const float*, int, const int*, const int*, const float*, int,
                           int**, int**, float**, int*, int, int, int);
int oracle_rmcl_iters(int, int, int, const int*, const int*, const float*, int, int**, int**, float**, int*);
}

static CSR oracle_spmm(const CSR& A, const CSR& B) {
  int *IC, *JC, nnzC;
  float* C;
  oracle_sequential_spmm(A.rowPtr, A.colInd, A.values, A.nnz, B.rowPtr, B.colInd, B.values, B.nnz, &IC, &JC, &C, &nnzC,
                         A.rows, A.cols, B.cols);
  return CSR(C, JC, IC, A.rows, B.cols, nnzC);
}

static int report(const char* what, bool same) {
  printf("%-28s %s\n", what, same ? "Same" : "Differs");
  return same ? 0 : 1;
}

int main(int argc, char* argv[]) {
  if (argc < 2) { printf("usage: %s <snap|mtx file> [--rmcl N]\n", argv[0]); return 2; }
  int rmclIters = 0;
  for (int i = 2; i + 1 < argc; ++i) if (!strcmp(argv[i], "--rmcl")) rmclIters = atoi(argv[i + 1]);
  int bad = 0;
  // mindex2-cuda/nGpuSpMM.cc:285-294: readSNAPFile(f,false) + dedupe + toCSR + toAbs, B = A
  COO coo;
  coo.readSNAPFile(argv[1], false);
  coo.orderedAndDuplicatesRemoving();
  CSR A = coo.toCSR();
  A.toAbs();
  {                                               // the loader's sort + dedupe + toCSR + toAbs on the device instead
    COO raw;
    raw.readSNAPFile(argv[1], false);
    CSR dL = raw.toGpuCSR(SPGEMM_COO_DEDUPE | SPGEMM_COO_ABS);
    CSR hL = dL.toCpuCSR();
    dL.deviceDispose();
    const bool bitEq = hL.rows == A.rows && hL.nnz == A.nnz && !memcmp(hL.rowPtr, A.rowPtr, sizeof(int) * ((size_t)A.rows + 1)) &&
                     !memcmp(hL.colInd, A.colInd, sizeof(int) * (size_t)A.nnz) &&
                     !memcmp(hL.values, A.values, sizeof(QValue) * (size_t)A.nnz);
    bad += report("COO::toGpuCSR bit-equal", bitEq);
    hL.dispose(); raw.dispose();
  }
  coo.dispose();
  CSR B = A.deepCopy();
  CSR want = oracle_spmm(A, B);
  want.makeOrdered();

  CSR dA = A.toGpuCSR(), dB = B.toGpuCSR();
  outputStats(gpuFlopsStats(dA, dB));            // tools/stats.cc report: rows by power-of-two flop class
  outputStats(dA.gpuNnzStats());                 // CSR::nnzStats of the input: rows by power-of-two length class
  CSR dC = gpuSpMMWrapper(dA, dB);
  if (A.rows <= 8) gpuOutputCSRWrapper(dC, "C = A*A on the device (gpuOutputCSRWrapper)");   // tiny inputs only
  CSR hC = dC.toCpuCSR();
  dC.deviceDispose();
  {                                               // per-bin report, like resultsComparison (nGpuSpMM.cc:138-240)
    int *drowIds = 0, *dflops = 0;
    std::vector<int> hv = gpuFlopsClassify(dA, dB, &drowIds, &dflops);
    std::vector<int> hqueue((size_t)A.rows + 1);
    if (A.rows) spgemm_hip_memcpy_d2h(hqueue.data(), drowIds, sizeof(int) * (size_t)A.rows);
    spgemm_hip_free(drowIds); spgemm_hip_free(dflops);
    bad += report("resultsComparison per bin", resultsComparison(hC, want, hv, hqueue.data()));
  }
  dA.deviceDispose(); dB.deviceDispose();
  hC.makeOrdered();
  bad += report("gpuSpMMWrapper isEqual", hC.isEqual(want));
  bad += report("gpuSpMMWrapper parity", hC.isParityEqual(want));
  hC.dispose();

  CSR h2 = A.hip_spmm(B);
  h2.makeOrdered();
  bad += report("CSR::hip_spmm parity", h2.isParityEqual(want));
  h2.dispose();

  CSR h3 = scudaSpMM(A, B);
  h3.makeOrdered();
  bad += report("scudaSpMM parity", h3.isParityEqual(want));
  h3.dispose();
  for (int shards = 2; shards <= 3; ++shards) {   // row-sharded over logical shards (every visible device takes its share)
    CSR h4 = gpuShardedSpMM(A, B, shards);
    h4.makeOrdered();
    char what[64];
    snprintf(what, sizeof what, "gpuShardedSpMM(%d) parity", shards);
    bad += report(what, h4.rows == want.rows && h4.isParityEqual(want));
    h4.dispose();
  }
  printf("rows=%d nnzA=%d flops=%lld nnzC=%d\n", A.rows, A.nnz, A.spMMFlops(B), want.nnz);
  want.dispose(); A.dispose(); B.dispose();

  if (rmclIters > 0) {                            // nrmcl.cc:12-37 with GPU in the place of SOMP
    CSR Mt = RMCL(argv[1], rmclIters, GPU);
    COO c2;
    c2.readSNAPFile(argv[1]);
    CSR ref = rmclInit(c2);
    c2.dispose();
    CSR g = ref.deepCopy();
    oracle_rmcl_iters(rmclIters, ref.rows, ref.cols, g.rowPtr, g.colInd, g.values, g.nnz, &ref.rowPtr, &ref.colInd,
                      &ref.values, &ref.nnz);
    Mt.makeOrdered();
    ref.makeOrdered();
    bad += report("RMCL(GPU) vs SEQ isEqual", Mt.isEqual(ref));
    {                                             // the same loop over two logical shards
      COO c3;
      c3.readSNAPFile(argv[1]);
      CSR m2 = rmclInit(c3);
      c3.dispose();
      CSR g2 = m2.deepCopy();
      gpuShardedRmclIter(rmclIters, g2, m2, 2);
      m2.makeOrdered();
      bad += report("gpuShardedRmclIter(2) isEqual", m2.isEqual(ref));
      m2.dispose(); g2.dispose();
    }
    Mt.dispose(); ref.dispose(); g.dispose();
  }
  return bad ? 1 : 0;
}
