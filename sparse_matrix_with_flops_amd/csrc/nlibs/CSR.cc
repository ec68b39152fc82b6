// CSR.cc — see CSR.h.  Host-side glue only; products are computed by libspgemm_hip.so.
#include "CSR.h"
#include "gpus/gpu_csr_kernel.h"
#include "../../../include/spgemm_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

static void* must_malloc(size_t bytes, const char* what) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p) { printf("out of host memory allocating %s (%zu bytes)\n", what, bytes); exit(EXIT_FAILURE); }
  return p;
}

static void hip_or_die(int rc, const char* what) {
  if (rc != SPGEMM_OK) { printf("%s: %s\n", what, spgemm_hip_last_error()); exit(EXIT_FAILURE); }
}

CSR CSR::deepCopy() const {
  int* rp = (int*)must_malloc(sizeof(int) * ((size_t)rows + 1), "rowPtr");
  int* ci = (int*)must_malloc(sizeof(int) * (size_t)nnz, "colInd");
  QValue* v = (QValue*)must_malloc(sizeof(QValue) * (size_t)nnz, "values");
  memcpy(rp, rowPtr, sizeof(int) * ((size_t)rows + 1));
  memcpy(ci, colInd, sizeof(int) * (size_t)nnz);
  memcpy(v, values, sizeof(QValue) * (size_t)nnz);
  return CSR(v, ci, rp, rows, cols, nnz);
}

void CSR::makeOrdered() {
  std::vector<std::pair<int, QValue> > row;
  for (int i = 0; i < rows; ++i) {
    const int s = rowPtr[i], e = rowPtr[i + 1];
    if (e - s < 2) continue;
    row.resize(e - s);
    for (int p = s; p < e; ++p) row[p - s] = std::make_pair(colInd[p], values[p]);
    std::sort(row.begin(), row.end());
    for (int p = s; p < e; ++p) { colInd[p] = row[p - s].first; values[p] = row[p - s].second; }
  }
}

void CSR::toAbs() { for (int p = 0; p < nnz; ++p) values[p] = std::fabs(values[p]); }

void CSR::averAndNormRowQValue() {
  for (int i = 0; i < rows; ++i) {
    const int count = rowPtr[i + 1] - rowPtr[i];
    for (int p = rowPtr[i]; p < rowPtr[i + 1]; ++p) values[p] = (QValue)(1.0 / count);
  }
}

void CSR::dispose() {
  free(values); values = 0;
  free(colInd); colInd = 0;
  free(rowPtr); rowPtr = 0;
}

void CSR::output(const char* msg) const {
  printf("%s\n", msg);
  for (int i = 0; i < rows; ++i)
    for (int p = rowPtr[i]; p < rowPtr[i + 1]; ++p) printf("%d\t%d\t%.6lf\n", i, colInd[p], (double)values[p]);
}

bool CSR::isEqual(const CSR& B) const {
  bool same = true;
  if (rows != B.rows) { printf("rows = %d\tB_rows = %d\n", rows, B.rows); same = false; }
  if (cols != B.cols) { printf("cols = %d\tB_cols = %d\n", cols, B.cols); same = false; }
  if (nnz != B.nnz) { printf("nnz = %d\tB_nnz = %d\n", nnz, B.nnz); same = false; }
  if (!same) return false;
  for (int i = 0; i <= rows; ++i)
    if (rowPtr[i] != B.rowPtr[i]) { printf("rowPtr[%d] %d\t%d\n", i, rowPtr[i], B.rowPtr[i]); return false; }
  std::vector<double> dense((size_t)cols, 0.0);
  for (int i = 0; i < rows; ++i) {
    for (int p = rowPtr[i]; p < rowPtr[i + 1]; ++p) dense[colInd[p]] = values[p];
    for (int p = B.rowPtr[i]; p < B.rowPtr[i + 1]; ++p) {
      const int c = B.colInd[p];
      if (std::fabs(dense[c] - B.values[p]) > 1e-7) {
        printf("values[%d][%d] %lf\t%lf\n", i, c, dense[c], (double)B.values[p]);
        return false;
      }
      dense[c] = 0.0;
    }
  }
  return true;
}

bool CSR::isParityEqual(const CSR& B, double rel) const {
  if (rows != B.rows || cols != B.cols || nnz != B.nnz) return false;
  if (memcmp(rowPtr, B.rowPtr, sizeof(int) * ((size_t)rows + 1)) != 0) return false;
  if (memcmp(colInd, B.colInd, sizeof(int) * (size_t)nnz) != 0) return false;
  for (int p = 0; p < nnz; ++p) {
    const double x = values[p], y = B.values[p];
    if (std::fabs(x - y) > rel * std::max(std::fabs(x), std::fabs(y))) {
      printf("values[%d] %.9e\t%.9e\n", p, x, y);
      return false;
    }
  }
  return true;
}

CSR CSR::toGpuCSR() const {
  CSR d;
  d.rows = rows; d.cols = cols; d.nnz = nnz;
  hip_or_die(spgemm_hip_malloc((void**)&d.rowPtr, sizeof(int) * ((size_t)rows + 1)), "toGpuCSR");
  hip_or_die(spgemm_hip_memcpy_h2d(d.rowPtr, rowPtr, sizeof(int) * ((size_t)rows + 1)), "toGpuCSR");
  hip_or_die(spgemm_hip_malloc((void**)&d.colInd, sizeof(int) * (size_t)nnz), "toGpuCSR");
  hip_or_die(spgemm_hip_memcpy_h2d(d.colInd, colInd, sizeof(int) * (size_t)nnz), "toGpuCSR");
  hip_or_die(spgemm_hip_malloc((void**)&d.values, sizeof(QValue) * (size_t)nnz), "toGpuCSR");
  hip_or_die(spgemm_hip_memcpy_h2d(d.values, values, sizeof(QValue) * (size_t)nnz), "toGpuCSR");
  return d;
}

CSR CSR::toCpuCSR() const {
  CSR h;
  h.rows = rows; h.cols = cols; h.nnz = nnz;
  h.rowPtr = (int*)must_malloc(sizeof(int) * ((size_t)rows + 1), "rowPtr");
  h.colInd = (int*)must_malloc(sizeof(int) * (size_t)nnz, "colInd");
  h.values = (QValue*)must_malloc(sizeof(QValue) * (size_t)nnz, "values");
  hip_or_die(spgemm_hip_memcpy_d2h(h.rowPtr, rowPtr, sizeof(int) * ((size_t)rows + 1)), "toCpuCSR");
  hip_or_die(spgemm_hip_memcpy_d2h(h.colInd, colInd, sizeof(int) * (size_t)nnz), "toCpuCSR");
  hip_or_die(spgemm_hip_memcpy_d2h(h.values, values, sizeof(QValue) * (size_t)nnz), "toCpuCSR");
  return h;
}

void CSR::deviceDispose() {
  spgemm_hip_free(values); values = 0;
  spgemm_hip_free(colInd); colInd = 0;
  spgemm_hip_free(rowPtr); rowPtr = 0;
}

CSR CSR::hip_spmm(const CSR& B, const int stride) const {
  if (cols != B.rows) { printf("hip_spmm: A is %dx%d but B is %dx%d\n", rows, cols, B.rows, B.cols); exit(EXIT_FAILURE); }
  int *IC = 0, *JC = 0, nnzC = 0;
  QValue* C = 0;
  hip_CSR_SpMM(rowPtr, colInd, values, nnz, B.rowPtr, B.colInd, B.values, B.nnz, IC, JC, C, nnzC, rows, cols, B.cols, stride);
  return CSR(C, JC, IC, rows, B.cols, nnzC);
}

long long CSR::spMMFlops(const CSR& B) const {
  long long p = 0;
  for (int q = 0; q < nnz; ++q) { const int j = colInd[q]; p += B.rowPtr[j + 1] - B.rowPtr[j]; }
  return 2 * p;
}

// ---------------------------------------------------------------------------------------------------------------
// wrapper functions (gpus/gpu_csr_kernel.h)
// ---------------------------------------------------------------------------------------------------------------
void hip_CSR_SpMM(const int IA[], const int JA[], const QValue A[], const int nnzA, const int IB[], const int JB[],
                  const QValue B[], const int nnzB, int*& IC, int*& JC, QValue*& C, int& nnzC, const int m,
                  const int k, const int n, const int stride) {
  (void)stride;
  hip_or_die(::hip_CSR_SpMM(IA, JA, A, nnzA, IB, JB, B, nnzB, &IC, &JC, &C, &nnzC, m, k, n), "hip_CSR_SpMM");
}

CSR gpuSpMMWrapper(const CSR& dA, const CSR& dB) {
  if (dA.cols != dB.rows) { printf("gpuSpMMWrapper: shape mismatch\n"); exit(EXIT_FAILURE); }
  CSR dC;
  hip_or_die(hip_gpuSpMM(0, dA.rowPtr, dA.colInd, dA.values, dA.nnz, dB.rowPtr, dB.colInd, dB.values, dB.nnz, dA.rows,
                         dA.cols, dB.cols, &dC.rowPtr, &dC.colInd, &dC.values, &dC.nnz), "gpuSpMMWrapper");
  dC.rows = dA.rows;
  dC.cols = dB.cols;
  return dC;
}

std::vector<int> gpuFlopsClassify(const CSR& dA, const CSR& dB, int** drowIdsp, int** dflopId) {
  int hv[SPGEMM_HV_LEN], len = 0;
  long long total = 0;
  hip_or_die(hip_gpuFlopsClassify(0, dA.rowPtr, dA.colInd, dB.rowPtr, dA.rows, dA.cols, drowIdsp, dflopId, hv, &len, &total),
             "gpuFlopsClassify");
  return std::vector<int>(hv, hv + len);      // the reference's vector has max bin + 2 entries
}

CSR sgpuSpMMWrapper(const CSR& dA, const CSR& dB, int* drowIds, const std::vector<int>& hv, int* dflops) {
  int full[SPGEMM_HV_LEN];
  for (int i = 0; i < SPGEMM_HV_LEN; ++i) full[i] = i < (int)hv.size() ? hv[i] : (hv.empty() ? 0 : hv.back());
  CSR dC;
  hip_or_die(hip_sgpuSpMM(0, dA.rowPtr, dA.colInd, dA.values, dA.nnz, dB.rowPtr, dB.colInd, dB.values, dB.nnz, dA.rows,
                          dA.cols, dB.cols, drowIds, full, dflops, &dC.rowPtr, &dC.colInd, &dC.values, &dC.nnz),
             "sgpuSpMMWrapper");
  dC.rows = dA.rows;
  dC.cols = dB.cols;
  return dC;
}

CSR scudaSpMM(const CSR& hA, const CSR& hB) {
  CSR dA = hA.toGpuCSR();
  CSR dB = hB.toGpuCSR();
  int *dqueue = 0, *dflops = 0;
  std::vector<int> hv = gpuFlopsClassify(dA, dB, &dqueue, &dflops);
  CSR dC = sgpuSpMMWrapper(dA, dB, dqueue, hv, dflops);
  CSR hC = dC.toCpuCSR();
  dC.deviceDispose();
  spgemm_hip_free(dqueue);
  spgemm_hip_free(dflops);
  dA.deviceDispose();
  dB.deviceDispose();
  return hC;
}

void gpuRmclIter(const int maxIter, const CSR Mgt, CSR& Mt) {
  int *oI = 0, *oJ = 0, on = 0;
  QValue* oA = 0;
  hip_or_die(hip_gpuRmclIter(maxIter, Mt.rows, Mt.cols, Mgt.rowPtr, Mgt.colInd, Mgt.values, Mgt.nnz, Mt.rowPtr, Mt.colInd,
                             Mt.values, Mt.nnz, &oI, &oJ, &oA, &on), "gpuRmclIter");
  Mt.dispose();                                   // nlibs/gpus/gpu_csr_kernel.cu:302-303: old arrays freed, new malloc()ed
  Mt.init(oA, oJ, oI, Mgt.rows, Mgt.cols, on);
}

static spgemm_group* make_group(int shards) {
  int ndev = 0;
  hip_or_die(spgemm_hip_device_count(&ndev), "spgemm_hip_device_count");
  if (shards <= 0) shards = ndev;
  std::vector<int> dev((size_t)shards);
  for (int i = 0; i < shards; ++i) dev[(size_t)i] = i % (ndev > 0 ? ndev : 1);
  spgemm_group* g = 0;
  hip_or_die(spgemm_hip_group_create(&g, shards, dev.data(), SPGEMM_XCHG_AUTO), "spgemm_hip_group_create");
  return g;
}

CSR gpuShardedSpMM(const CSR& hA, const CSR& hB, int shards) {
  spgemm_group* g = make_group(shards);
  spgemm_sharded* job = 0;
  hip_or_die(hip_sharded_spmm_create(g, hA.rowPtr, hA.colInd, hA.values, hA.nnz, hB.rowPtr, hB.colInd, hB.values, hB.nnz,
                                     hA.rows, hA.cols, hB.cols, &job), "hip_sharded_spmm_create");
  long long nnzC = 0, P = 0;
  hip_or_die(hip_sharded_spmm_step(job, 1, &nnzC, &P), "hip_sharded_spmm_step");
  int *IC = 0, *JC = 0, nz = 0, rows = 0;
  QValue* C = 0;
  hip_or_die(hip_sharded_spmm_result(job, 0, &IC, &JC, &C, &nz, &rows), "hip_sharded_spmm_result");
  hip_sharded_spmm_destroy(job);
  spgemm_hip_group_destroy(g);
  return CSR(C, JC, IC, rows, hB.cols, nz);       // malloc()ed like every host result: dispose() == free()
}

void gpuShardedRmclIter(const int maxIter, const CSR Mgt, CSR& Mt, int shards) {
  spgemm_group* g = make_group(shards);
  int *oI = 0, *oJ = 0, on = 0;
  QValue* oA = 0;
  hip_or_die(hip_gpuRmclIter_sharded(g, maxIter, Mt.rows, Mt.cols, Mgt.rowPtr, Mgt.colInd, Mgt.values, Mgt.nnz, Mt.rowPtr,
                                     Mt.colInd, Mt.values, Mt.nnz, &oI, &oJ, &oA, &on), "gpuShardedRmclIter");
  spgemm_hip_group_destroy(g);
  Mt.dispose();
  Mt.init(oA, oJ, oI, Mgt.rows, Mgt.cols, on);
}

void gpuOutputCSRWrapper(const CSR dA, const char* msg) {
  printf("%s\n", msg);
  printf("rows=%d cols=%d nnz=%d rowPtr=%p colInd=%p values=%p\n", dA.rows, dA.cols, dA.nnz, (void*)dA.rowPtr,
         (void*)dA.colInd, (void*)dA.values);
  CSR h = dA.toCpuCSR();
  for (int r = 0; r < h.rows; ++r)
    for (int p = h.rowPtr[r]; p < h.rowPtr[r + 1]; ++p) printf("%d\t%d\t%.6lf\n", r, h.colInd[p], (double)h.values[p]);
  printf("rowPtr= ");
  for (int r = 0; r <= h.rows; ++r) printf("%d ", h.rowPtr[r]);
  printf("\ncolInd, values= ");
  for (int p = 0; p < h.nnz; ++p) printf("%d:%.6lf ", h.colInd[p], (double)h.values[p]);
  printf("\n");
  h.dispose();
}

std::vector<int> CSR::nnzStats() const {
  std::vector<int> stats(SPGEMM_NNZ_STATS_LEN, 0);
  for (int r = 0; r < rows; ++r) {
    const long len = rowPtr[r + 1] - rowPtr[r];
    int b = SPGEMM_NNZ_STATS_LEN - 1;
    for (int q = 0; q < SPGEMM_NNZ_STATS_LEN - 1; ++q) if (len <= (1l << q)) { b = q; break; }
    ++stats[b];
  }
  return stats;
}

std::vector<int> CSR::differsStats(const CSR& B, const std::vector<QValue>& percents) const {
  const size_t n = percents.size();
  std::vector<int> counts(n + 4, 0);                 // [0..n-1] below percents[k], [n] the rest, [n+1] appeared, [n+2] both empty, [n+3] equal
  for (int i = 0; i < rows; ++i) {
    const int a = rowPtr[i + 1] - rowPtr[i], b = B.rowPtr[i + 1] - B.rowPtr[i];
    if (a == 0) { ++counts[b > 0 ? n + 1 : n + 2]; continue; }
    if (a == b) { ++counts[n + 3]; continue; }
    const QValue change = (QValue)(b - a) / (QValue)a;
    size_t k = 0;
    while (k < n && !(change < percents[k])) ++k;
    ++counts[k];
  }
  return counts;
}

std::vector<int> CSR::gpuNnzStats() const {
  std::vector<int> stats(SPGEMM_NNZ_STATS_LEN, 0);
  if (hip_nnzStats(NULL, rowPtr, rows, stats.data())) { printf("%s\n", spgemm_hip_last_error()); exit(EXIT_FAILURE); }
  return stats;
}

bool resultsComparison(const CSR& hC, const CSR& rC, const std::vector<int>& hv, const int* hqueue, double rel) {
  static const char* const kBinNames[SPGEMM_HV_LEN - 1] = {"(bin 0, unused)", "0 products", "1 product", "2-4", "5-16",
                                                           "17-64", "65-512", ">512"};
  if (hC.rows != rC.rows || hC.cols != rC.cols) { printf("resultsComparison: shapes differ\n"); return false; }
  int hvv[SPGEMM_HV_LEN];
  const int len = (int)std::min<size_t>(hv.size(), SPGEMM_HV_LEN);
  for (int i = 0; i < SPGEMM_HV_LEN; ++i) hvv[i] = i < len ? hv[i] : (len ? hv[len - 1] : 0);
  spgemm_bin_report rep[SPGEMM_HV_LEN - 1];
  hip_or_die(hip_resultsComparison(hC.rows, hC.cols, hC.rowPtr, hC.colInd, hC.values, rC.rowPtr, rC.colInd, rC.values, hvv, len,
                                   hqueue, rel, rep), "resultsComparison");
  bool same = hC.nnz == rC.nnz;
  printf("hC compare with rC: nnz %d vs %d\n", hC.nnz, rC.nnz);
  for (int b = 1; b + 1 < len; ++b) {
    printf("Checking %-12s rows= %d differ= %d first= %d max_rel_err= %.3e\n", kBinNames[b], rep[b].rows, rep[b].rows_differ,
           rep[b].first_bad_row, rep[b].max_rel_err);
    same = same && rep[b].rows_differ == 0;
  }
  std::printf("%s\n", same ? "Same" : "Diffs");
  return same;
}
