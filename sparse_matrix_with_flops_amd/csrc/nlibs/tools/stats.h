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

// void outputStats(const std::vector<int>&), nlibs/tools/stats.cc:14-27: "(lo -> hi)\tcount\tshare"
inline void outputStats(const std::vector<int>& stats) {
  long long sum = 0;
  for (size_t i = 0; i < stats.size(); ++i) sum += stats[i];
  printf("Total sum = %lld\n", sum);
  size_t i = 0;
  for (; i + 1 < stats.size(); ++i) {
    const long bound = 1l << i;
    printf("(%ld -> %ld)\t%d\t%.6f\t\n", bound / 2 + 1, bound, stats[i], (QValue)stats[i] / sum);
  }
  const long bound = 1l << i;
  printf("(%ld -> INF)\t%d\t%.6lf\t\n", bound / 2 + 1, stats[i], (QValue)stats[i] / sum);
}
#endif
