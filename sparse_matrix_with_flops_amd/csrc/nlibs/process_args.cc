#include "process_args.h"

#include <getopt.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

Options options;

static const char* const kRunNames[] = {"SEQ", "OMP", "GPU", "CILK", "SOMP", "MKL", "SFOMP", "HYB"};

const char* runOptionName(RunOptions r) { return kRunNames[(int)r]; }

static const char* const kSharedNames[] = {"CachePreferNone", "CachePreferShared", "CachePreferL1"};
const char* sharedOptionName(SharedOption o) { return kSharedNames[(int)o]; }

// --shared None|Shared|L1 (any case in the reference's table: "None"/"none", ...): the L1/shared-memory split
// cudaDeviceSetCacheConfig chooses (mindex2-cuda/nGpuSpMM.cc:297-299).  CDNA4 has no such split (LDS and L1 are separate
// arrays): parsed and carried so that a reference command line runs unchanged, otherwise without effect.
static bool parseSharedOption(const char* s, SharedOption* out) {
  static const char* const keys[] = {"None", "Shared", "L1"};
  for (int i = 0; i < 3; ++i)
    if (!strcasecmp(s, keys[i])) { *out = (SharedOption)i; return true; }
  return false;
}

static bool parseRunOption(const char* s, RunOptions* out) {
  for (int i = 0; i < (int)(sizeof(kRunNames) / sizeof(kRunNames[0])); ++i)
    if (!strcasecmp(s, kRunNames[i])) { *out = (RunOptions)i; return true; }
  return false;
}

int process_args(int argc, char** argv) {
  static const struct option longOpts[] = {
      {"calcChange", no_argument, nullptr, 'c'}, {"input", required_argument, nullptr, 'i'},
      {"rmclOptions", required_argument, nullptr, 'r'}, {"shared", required_argument, nullptr, 'e'},
      {"maxIters", required_argument, nullptr, 'm'},
      {"stride", required_argument, nullptr, 'd'}, {"stats", no_argument, nullptr, 's'},
      {"ptile", required_argument, nullptr, 'p'}, {"br", required_argument, nullptr, 'x'},
      {"bc", required_argument, nullptr, 'y'}, {"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
  optind = 1;
  for (int c; (c = getopt_long(argc, argv, "cr:i:m:sx:y:h", longOpts, nullptr)) != -1;) {
    switch (c) {
      case 'c': options.calcChange = true; break;
      case 's': options.stats = true; break;
      case 'i': snprintf(options.inputFileName, sizeof(options.inputFileName), "%s", optarg); break;
      case 'r':
        if (!parseRunOption(optarg, &options.rmclOption)) printf("unknown --rmclOptions %s (kept %s)\n", optarg, runOptionName(options.rmclOption));
        break;
      case 'e':
        // (the reference's map lookup turns an unknown word into CachePreferNone without a message)
        if (!parseSharedOption(optarg, &options.sharedOption)) options.sharedOption = CachePreferNone;
        break;
      case 'm': options.maxIters = atoi(optarg); break;
      case 'd': options.stride = atoi(optarg); break;
      case 'p': options.ptile = atoi(optarg); break;
      case 'x': options.br = atoi(optarg); break;
      case 'y': options.bc = atoi(optarg); break;
      case 'h':
        printf("usage: %s --input FILE [--maxIters N] [--stride N] [--rmclOptions GPU] [--shared None|Shared|L1] [--stats]\n", argv[0]);
        break;
      default: break;                               // getopt_long has printed its own message
    }
  }
  if (optind < argc) {
    printf("non-option ARGV-elements: ");
    while (optind < argc) printf("%s ", argv[optind++]);
    putchar('\n');
  }
  return 0;
}

void print_args() {
  printf("{\tcalcChange= %s\tstats= %s\tinputFileName= %s\tmaxIters= %d\tstride= %d\tptile= %d\trmclOption= %s\tSharedOption= %s\t}\n",
         options.calcChange ? "true" : "false", options.stats ? "true" : "false", options.inputFileName,
         options.maxIters, options.stride, options.ptile, runOptionName(options.rmclOption), sharedOptionName(options.sharedOption));
}
