// tools/stats.h — mirror of the reference's workload statistics (nlibs/tools/stats.h:8-12, stats.cc:3-55): the
// 13-bucket power-of-two histogram of per-row flops, computed on the device by hip_flopsStats, and the reference's
// report format.
#ifndef SMF_TOOLS_STATS_H_
#define SMF_TOOLS_STATS_H_
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../CSR.h"
#include "../../../../include/spgemm_hip.h"

// std::vector<int> flopsStats(IA, JA, IB, JB, m) for DEVICE CSRs (nlibs/tools/stats.cc:45-55)
inline std::vector<int> gpuFlopsStats(const CSR& dA, const CSR& dB) {
  std::vector<int> stats(SPGEMM_STATS_LEN, 0);
  if (hip_flopsStats(NULL, dA.rowPtr, dA.colInd, dB.rowPtr, dA.rows, stats.data())) {
    printf("%s\n", spgemm_hip_last_error());
    exit(EXIT_FAILURE);
  }
  return stats;
}

// Report in the format of the reference's outputStats (nlibs/tools/stats.cc:14-27): one line per bucket,
// "(lo -> hi)<TAB>count<TAB>share", the last bucket open-ended, preceded by the total.
inline void outputStats(const std::vector<int>& stats) {
  long long total = 0;
  for (int c : stats) total += c;
  printf("Total sum = %lld\n", total);
  const int last = (int)stats.size() - 1;
  for (int b = 0; b <= last; ++b) {
    const long hi = 1l << b, lo = hi / 2 + 1;
    const double share = (double)(QValue)((QValue)stats[b] / total);
    if (b < last) printf("(%ld -> %ld)\t%d\t%.6f\t\n", lo, hi, stats[b], share);
    else printf("(%ld -> INF)\t%d\t%.6f\t\n", lo, stats[b], share);
  }
}
#endif
