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
  CSR Mg = M0.deepCopy();                        // the checker consumes the Mt arrays it is given and returns new ones
  oracle_rmcl_iters(options.maxIters, Mg.rows, Mg.cols, Mg.rowPtr, Mg.colInd, Mg.values, Mg.nnz, &M0.rowPtr, &M0.colInd,
                    &M0.values, &M0.nnz);
  Mt.makeOrdered();
  M0.makeOrdered();
  const bool isSame = Mt.isEqual(M0);
  std::cout << (isSame ? "Same\n" : "Diffs\n");
  Mg.dispose(); M0.dispose(); Mt.dispose();
  return isSame ? 0 : 1;
}
