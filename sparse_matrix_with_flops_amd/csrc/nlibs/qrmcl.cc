#include "qrmcl.h"
#include "gpus/gpu_csr_kernel.h"
#include "process_args.h"

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
  if (options.stats) {
    // --stats (nlibs/qrmcl.cc:17-24,65-70): one line per iteration in "percent.stats" with the drift of the row lengths
    // from Mt to the new Mt (CSR::differsStats).  The thresholds are the reference's -- its int array {-30, -20, -5, 0, 5,
    // 20, 30, 100} read as QValue and compared with the FRACTION (len' - len) / len, so in practice rows land in the
    // "< 0" bucket, the "< 5" bucket, or the three special buckets; kept as it is, the file format is the contract.
    // The loop then runs one iteration per call (the matrix comes back to the host every time: this is a report mode).
    static const int cpercents[] = {-30, -20, -5, 0, 5, 20, 30, 100};
    const std::vector<QValue> percents(cpercents, cpercents + sizeof(cpercents) / sizeof(int));
    FILE* fp = fopen("percent.stats", "w");
    if (!fp) { printf("cannot write percent.stats\n"); exit(-1); }
    fprintf(fp, "rows %d\n", Mt.rows);
    fprintf(fp, "percent\t");
    for (size_t i = 0; i < percents.size(); ++i) fprintf(fp, "%lf ", (double)percents[i]);
    fprintf(fp, "\n");
    for (int iter = 0; iter < maxIters; ++iter) {
      CSR old = Mt.deepCopy();
      gpuRmclIter(1, Mgt, Mt);
      const std::vector<int> counts = old.differsStats(Mt, percents);
      fprintf(fp, "%d :\t", iter);
      for (size_t i = 0; i < counts.size(); ++i) fprintf(fp, "%d ", counts[i]);
      fprintf(fp, "\n");
      old.dispose();
    }
    fclose(fp);
  } else {
    gpuRmclIter(maxIters, Mgt, Mt);
  }
  printf("time pass gpuRmclIter (H2D + %d device iterations + D2H) = %lf\n", maxIters, ms_since(t0));
  Mgt.dispose();
  return Mt;
}
