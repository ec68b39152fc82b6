#include "qrmcl.h"
#include "gpus/gpu_csr_kernel.h"

#include <chrono>
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
  typedef std::chrono::steady_clock clk;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  cooAt.readSNAPFile(iname);                     // isTrans = true: R-MCL works on the transpose
  printf("time pass readSNAPFile (read + parse, %d threads) = %lf\n", COO::lastParseThreads, COO::lastParseMs);
  auto t0 = clk::now();
  CSR Mt = rmclInit(cooAt);
  cooAt.dispose();
  CSR Mgt = Mt.deepCopy();
  printf("time pass rmclInit = %lf\n", ms_since(t0));
  t0 = clk::now();
  gpuRmclIter(maxIters, Mgt, Mt);
  printf("time pass gpuRmclIter (H2D + %d device iterations + D2H) = %lf\n", maxIters, ms_since(t0));
  Mgt.dispose();
  return Mt;
}
