// parse_check.cc — dumps what COO::readSNAPFile (the multi-threaded loader of the C++ mirror) reads from a file, for
// the CPU test that compares it with the oracle's restatement of the reference loader (nlibs/COO.cc:48-158).
//   parse_check.x <file> <isTrans 0|1> <out.bin>      (thread count: SMF_PARSE_THREADS)
// out.bin: int32 rows, cols, nnz, then rowIndex[nnz], colIndex[nnz] (int32), val[nnz] (float32)
#include <cstdio>
#include <cstdlib>

#include "COO.h"

int main(int argc, char* argv[]) {
  if (argc < 4) { printf("usage: %s file isTrans out.bin\n", argv[0]); return 2; }
  COO coo;
  coo.readSNAPFile(argv[1], atoi(argv[2]) != 0);
  FILE* fp = fopen(argv[3], "wb");
  if (!fp) return 3;
  const int hdr[3] = {coo.rows, coo.cols, coo.nnz};
  fwrite(hdr, sizeof(int), 3, fp);
  if (coo.nnz > 0) {
    fwrite(coo.cooRowIndex, sizeof(int), (size_t)coo.nnz, fp);
    fwrite(coo.cooColIndex, sizeof(int), (size_t)coo.nnz, fp);
    fwrite(coo.cooVal, sizeof(QValue), (size_t)coo.nnz, fp);
  }
  fclose(fp);
  printf("parse: %.2f ms on %d threads, nnz=%d\n", COO::lastParseMs, COO::lastParseThreads, coo.nnz);
  coo.dispose();
  return 0;
}
