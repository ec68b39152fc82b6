// qrmcl.h — mirror of the R-MCL driver surface (nlibs/qrmcl.h:8-25, nlibs/qrmcl.cc:126-164) for the GPU option.
// Only RunOptions::GPU is implemented in this project (the CPU variants are the reference's own business);
// asking for another option exits with a message, exactly like the reference built without that feature.
#ifndef SMF_QRMCL_H_
#define SMF_QRMCL_H_
#include "COO.h"
#include "CSR.h"

enum RunOptions { SEQ, OMP, GPU, CILK, SOMP, MKL, SFOMP, HYB };

CSR rmclInit(COO& cooAt);                                           // nlibs/qrmcl.cc:126-134
CSR RMCL(const char iname[], int maxIters, RunOptions runOptions);  // nlibs/qrmcl.cc:136-164
#endif
