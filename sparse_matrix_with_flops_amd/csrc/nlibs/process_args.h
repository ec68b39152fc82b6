// process_args.h — command line of the reference's drivers (nlibs/process_args.h:30-45, process_args.cc:12-27):
//   --input/-i FILE   --maxIters/-m N   --stride N   --rmclOptions/-r {SEQ|OMP|GPU|...}   --stats/-s   --calcChange/-c
//   --shared {None|Shared|L1}  (nlibs/process_args.cc:17,60-63; the CUDA cache split of mindex2-cuda/nGpuSpMM.cc:297-299:
//   accepted and carried, no effect on CDNA4)
// Same option names and meaning; the values land in the global `options`, like in the reference.  --stride is a CPU
// scheduling knob there: accepted and carried, the HIP path ignores it.
#ifndef SMF_PROCESS_ARGS_H_
#define SMF_PROCESS_ARGS_H_
#include "qrmcl.h"

enum SharedOption { CachePreferNone, CachePreferShared, CachePreferL1 };   // nlibs/process_args.h:13

struct Options {
  bool calcChange = false;
  bool stats = false;
  int maxIters = 5;
  int stride = 512;
  int ptile = 2;
  int br = 2, bc = 8;
  char inputFileName[200] = "";
  RunOptions rmclOption = GPU;        // the only option this build can run (qrmcl.h)
  SharedOption sharedOption = CachePreferNone;
};

extern Options options;

int process_args(int argc, char** argv);   // 0 on success; unknown --rmclOptions values are reported and kept as GPU
void print_args();
const char* runOptionName(RunOptions r);
const char* sharedOptionName(SharedOption o);
#endif
