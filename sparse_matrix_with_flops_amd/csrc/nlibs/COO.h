// COO.h — mirror of the reference's `class COO` (nlibs/COO.h:6-26): triplet arrays + the text loaders and the
// sort / dedupe / toCSR steps every driver runs before the SpGEMM path.
#ifndef SMF_COO_H_
#define SMF_COO_H_
#include "CSR.h"

class COO {
 public:
  int* cooRowIndex;
  int* cooColIndex;
  QValue* cooVal;
  int rows, cols, nnz;
  COO() : cooRowIndex(0), cooColIndex(0), cooVal(0), rows(0), cols(0), nnz(0) {}
  void dispose();
  // SNAP edge lists and MatrixMarket coordinate files (nlibs/COO.cc:48-158): '#'/'%' comment lines, then a size line
  // "rows nnz" or "rows cols nnz", then "from to [value]" lines; a 5-token '%' banner marks MatrixMarket (1-based
  // indices, "symmetric" expands (i,j)->(j,i)); isTrans reads the transpose (what R-MCL wants).
  // The edge lines are parsed by all host threads (SMF_PARSE_THREADS overrides the count); lastParseMs / lastParseThreads
  // describe the latest call (read + parse, wall clock).
  int readSNAPFile(const char fname[], bool isTrans = true);
  static double lastParseMs;
  static int lastParseThreads;
  void addSelfLoopIfNeeded();           // nlibs/COO.cc:160-188
  void makeOrdered() const;             // nlibs/COO.cc:222-235: sort by (row, col)
  int orderedAndDuplicatesRemoving();   // nlibs/COO.cc:237-266: sort + sum duplicates
  CSR toCSR() const;                    // nlibs/COO.cc:268-291 (input must be ordered)
  // the same three steps on the device in one call (hip_coo_to_csr): upload the triplets in file order, get a
  // DEVICE CSR back.  flags: SPGEMM_COO_DEDUPE | SPGEMM_COO_SELF_LOOPS | SPGEMM_COO_ROW_NORMALISE | SPGEMM_COO_ABS
  CSR toGpuCSR(int flags) const;
};
#endif
