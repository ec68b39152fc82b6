#include "qrmcl.h"
#include "gpus/gpu_csr_kernel.h"

#include <cstdio>
#include <cstdlib>

CSR rmclInit(COO& cooAt) {
  cooAt.addSelfLoopIfNeeded();
  cooAt.makeOrdered();
  CSR At = cooAt.toCSR();
  At.averAndNormRowQValue();
  return At;
}

CSR RMCL(const char iname[], int maxIters, RunOptions runOptions) {
  if (runOptions != GPU) {
    printf("This build carries the GPU (HIP) R-MCL path only; run the reference for the CPU options\n");
    exit(-1);
  }
  COO cooAt;
  cooAt.readSNAPFile(iname);                     // isTrans = true: R-MCL works on the transpose
  CSR Mt = rmclInit(cooAt);
  cooAt.dispose();
  CSR Mgt = Mt.deepCopy();
  gpuRmclIter(maxIters, Mgt, Mt);
  Mgt.dispose();
  return Mt;
}
