// nrmcl.cc — the R-MCL driver (reference: nrmcl.cc:12-37): parse the reference's flags, run RMCL on the selected path
// and report.  The reference compares its SEQ and SOMP CPU paths; this build carries the GPU (HIP) path, so
//   nrmcl --input g.snap --maxIters 10 --rmclOptions GPU [--stats]
// runs RMCL(input, maxIters, GPU) and prints rows / nnz / time (and, with --stats, the row-length histogram of the
// result in the reference's outputStats format).  tests/cpp/nrmcl_check.cc wraps nrmcl_run() with the Same / Diffs
// comparison against the CPU checker, like the reference's own main does between its two paths.
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "COO.h"
#include "CSR.h"
#include "process_args.h"
#include "qrmcl.h"
#include "tools/stats.h"

// returns 0 and leaves the result in *out (caller disposes), or a non-zero exit code
int nrmcl_run(int argc, char* argv[], CSR* out) {
  process_args(argc, argv);
  print_args();
  if (!options.inputFileName[0]) { printf("no --input file\n"); return 2; }
  const auto t0 = std::chrono::steady_clock::now();
  CSR Mt = RMCL(options.inputFileName, options.maxIters, options.rmclOption);
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  printf("time pass %s rmcl total = %lf\n", runOptionName(options.rmclOption), ms);
  printf("rows=%d cols=%d nnz=%d\n", Mt.rows, Mt.cols, Mt.nnz);
  if (options.stats) outputStats(Mt.nnzStats());
  *out = Mt;
  return 0;
}

#ifndef NRMCL_NO_MAIN
int main(int argc, char* argv[]) {
  CSR Mt;
  const int rc = nrmcl_run(argc, argv, &Mt);
  if (rc == 0) Mt.dispose();
  return rc;
}
#endif
