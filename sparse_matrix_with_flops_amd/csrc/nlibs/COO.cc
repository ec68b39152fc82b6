// COO.cc — see COO.h.  Own implementation of the loader / sort / dedupe / toCSR steps with the reference's
// semantics (nlibs/COO.cc:48-291); plain host code, no device work.
#include "COO.h"
#include "../../../include/spgemm_hip.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct Entry {
  int r, c;
  QValue v;
  bool operator<(const Entry& o) const { return r < o.r || (r == o.r && c < o.c); }
};
template <class T> T* grab(size_t n, const char* what) {
  T* p = (T*)malloc((n ? n : 1) * sizeof(T));
  if (!p) { printf("out of host memory allocating %s\n", what); exit(EXIT_FAILURE); }
  return p;
}
std::string lower(std::string s) { for (auto& ch : s) ch = (char)tolower((unsigned char)ch); return s; }
}  // namespace

void COO::dispose() {
  free(cooRowIndex); cooRowIndex = 0;
  free(cooColIndex); cooColIndex = 0;
  free(cooVal); cooVal = 0;
}

int COO::readSNAPFile(const char fname[], bool isTrans) {
  FILE* fp = fopen(fname, "r");
  if (!fp) { printf("Failed to open file %s\n", fname); exit(-1); }
  char line[1025];
  bool isMtx = false, symmetric = false;
  rows = cols = nnz = 0;
  if (!fgets(line, sizeof line, fp)) { fclose(fp); return 0; }
  if (line[0] == '%') {                      // MatrixMarket banner: 5 tokens, the 5th is the storage scheme
    char t[5][64];
    if (sscanf(line, "%63s %63s %63s %63s %63s", t[0], t[1], t[2], t[3], t[4]) == 5) {
      isMtx = true;
      symmetric = lower(t[4]) == "symmetric";
    }
  }
  while ((line[0] == '#' || line[0] == '%') && !feof(fp))
    if (!fgets(line, sizeof line, fp)) break;
  if (feof(fp)) { fclose(fp); nnz = 0; return 0; }
  int a = 0, b = 0, c = 0;
  const int got = sscanf(line, "%d %d %d", &a, &b, &c);
  int declared;
  if (got == 2) { rows = cols = a; declared = b; }
  else if (got == 3) { rows = a; cols = b; declared = c; }
  else { printf("%s: cannot parse the size line\n", fname); exit(-1); }
  printf("rows=%d cols=%d nnz=%d\n", rows, cols, declared);
  const size_t cap = (size_t)declared * (symmetric ? 2 : 1);
  cooRowIndex = grab<int>(cap, "cooRowIndex");
  cooColIndex = grab<int>(cap, "cooColIndex");
  cooVal = grab<QValue>(cap, "cooVal");
  int top = 0;
  for (int i = 0; i < declared; ++i) {
    int from = 0, to = 0;
    float val = 0.f;
    int ret;
    if (symmetric) ret = fscanf(fp, "%d%d%f", &from, &to, &val);
    else { if (!fgets(line, sizeof line, fp)) break; ret = sscanf(line, "%d%d%f", &from, &to, &val); }
    if (ret < 2) break;
    if (ret == 2) val = 1.0f;
    if (isMtx) { --from; --to; }
    if (symmetric) {
      cooRowIndex[top] = from; cooColIndex[top] = to; cooVal[top++] = val;
      if (from != to) { cooRowIndex[top] = to; cooColIndex[top] = from; cooVal[top++] = val; }
    } else {
      cooRowIndex[top] = isTrans ? to : from;
      cooColIndex[top] = isTrans ? from : to;
      cooVal[top++] = val;
    }
  }
  fclose(fp);
  nnz = top;
  return 0;
}

void COO::addSelfLoopIfNeeded() {
  std::vector<char> has((size_t)rows, 0);
  for (int i = 0; i < nnz; ++i)
    if (cooRowIndex[i] == cooColIndex[i] && cooRowIndex[i] >= 0 && cooRowIndex[i] < rows) has[cooRowIndex[i]] = 1;
  int need = 0;
  for (int i = 0; i < rows; ++i) need += !has[i];
  cooRowIndex = (int*)realloc(cooRowIndex, sizeof(int) * (size_t)(nnz + need + 1));
  cooColIndex = (int*)realloc(cooColIndex, sizeof(int) * (size_t)(nnz + need + 1));
  cooVal = (QValue*)realloc(cooVal, sizeof(QValue) * (size_t)(nnz + need + 1));
  for (int i = 0; i < rows; ++i)
    if (!has[i]) { cooRowIndex[nnz] = i; cooColIndex[nnz] = i; cooVal[nnz++] = 1.0f; }
}

static void sorted_entries(const COO& m, std::vector<Entry>& v) {
  v.resize((size_t)m.nnz);
  for (int i = 0; i < m.nnz; ++i) v[i] = Entry{m.cooRowIndex[i], m.cooColIndex[i], m.cooVal[i]};
  std::stable_sort(v.begin(), v.end());
}

void COO::makeOrdered() const {
  std::vector<Entry> v;
  sorted_entries(*this, v);
  for (int i = 0; i < nnz; ++i) { cooRowIndex[i] = v[i].r; cooColIndex[i] = v[i].c; cooVal[i] = v[i].v; }
}

int COO::orderedAndDuplicatesRemoving() {
  if (nnz <= 0) { nnz = 0; return 0; }
  std::vector<Entry> v;
  sorted_entries(*this, v);
  int j = 0;
  for (int i = 1; i < nnz; ++i) {
    if (v[i].r == v[j].r && v[i].c == v[j].c) v[j].v += v[i].v;
    else v[++j] = v[i];
  }
  nnz = j + 1;
  for (int i = 0; i < nnz; ++i) { cooRowIndex[i] = v[i].r; cooColIndex[i] = v[i].c; cooVal[i] = v[i].v; }
  return nnz;
}

CSR COO::toCSR() const {
  int* rp = grab<int>((size_t)rows + 1, "rowPtr");
  memset(rp, 0, sizeof(int) * ((size_t)rows + 1));
  for (int t = 0; t < nnz; ++t)
    if (cooRowIndex[t] >= 0 && cooRowIndex[t] < rows) ++rp[cooRowIndex[t] + 1];
  for (int i = 0; i < rows; ++i) rp[i + 1] += rp[i];
  int* ci = grab<int>((size_t)nnz, "colInd");
  QValue* v = grab<QValue>((size_t)nnz, "values");
  memcpy(ci, cooColIndex, sizeof(int) * (size_t)nnz);
  memcpy(v, cooVal, sizeof(QValue) * (size_t)nnz);
  return CSR(v, ci, rp, rows, cols, nnz);
}

// COO -> device CSR without the host std::sort: hip_coo_to_csr does sort / duplicate summing / self loops / row
// normalisation / abs on the device (include/spgemm_hip.h).  Exit-on-error like the rest of the mirror.
CSR COO::toGpuCSR(int flags) const {
  int *dR = 0, *dC = 0; QValue* dV = 0;
  CSR d; d.rows = rows; d.cols = cols;
  const size_t nI = sizeof(int) * (size_t)nnz, nV = sizeof(QValue) * (size_t)nnz;
  int rc = spgemm_hip_malloc((void**)&dR, nI) || spgemm_hip_malloc((void**)&dC, nI) || spgemm_hip_malloc((void**)&dV, nV) ||
           spgemm_hip_memcpy_h2d(dR, cooRowIndex, nI) || spgemm_hip_memcpy_h2d(dC, cooColIndex, nI) ||
           spgemm_hip_memcpy_h2d(dV, cooVal, nV) ||
           hip_coo_to_csr(NULL, rows, cols, nnz, dR, dC, dV, flags, &d.rowPtr, &d.colInd, &d.values, &d.nnz);
  spgemm_hip_free(dR); spgemm_hip_free(dC); spgemm_hip_free(dV);
  if (rc) { printf("%s\n", spgemm_hip_last_error()); exit(EXIT_FAILURE); }
  return d;
}
