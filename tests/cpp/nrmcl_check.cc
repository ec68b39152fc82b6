// tests/cpp/nrmcl_check.cc — the product's nrmcl driver (csrc/nlibs/nrmcl.cc, reference flags) followed by the check the
// reference's nrmcl.cc:24-33 makes between its two paths: makeOrdered both, isEqual, print Same / Diffs.  The second
// path here is the CPU checker (oracle/liboracle.so: test infrastructure, never linked into the product).
#define NRMCL_NO_MAIN
#include "nrmcl.cc"

#include <iostream>

extern "C" int oracle_rmcl_iters(int, int, int, const int*, const int*, const float*, int, int**, int**, float**, int*);

int main(int argc, char* argv[]) {
  CSR Mt;
  int rc = nrmcl_run(argc, argv, &Mt);
  if (rc) return rc;
  COO coo;
  coo.readSNAPFile(options.inputFileName);
  CSR M0 = rmclInit(coo);
  coo.dispose();
  int *oI, *oJ, onnz;
  float* oV;
  oracle_rmcl_iters(options.maxIters, M0.rows, M0.cols, M0.rowPtr, M0.colInd, M0.values, M0.nnz, &oI, &oJ, &oV, &onnz);
  CSR want(oV, oJ, oI, M0.rows, M0.cols, onnz);
  Mt.makeOrdered();
  want.makeOrdered();
  const bool isSame = Mt.isEqual(want);
  std::cout << (isSame ? "Same\n" : "Diffs\n");
  want.dispose(); M0.dispose(); Mt.dispose();
  return isSame ? 0 : 1;
}
