// COO.cc — see COO.h.  Own implementation of the loader / sort / dedupe / toCSR steps with the reference's
// semantics (nlibs/COO.cc:48-291); plain host code, no device work.
#include "COO.h"
#include "../../../include/spgemm_hip.h"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <thread>
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

// ---- text -> triplets -----------------------------------------------------------------------------------------------
// The reference parses one fgets/sscanf line at a time on one thread (nlibs/COO.cc:130-155); for a 1 M-row edge list
// that is the longest step of the whole run.  Here the file is read into memory once and the
// edge lines are parsed by all host threads: the data region is cut at line ends into one chunk per thread, a first
// sweep counts the lines of every chunk (so that line k of the file lands in slot k of the arrays whatever thread
// parses it), a second sweep converts them with strtol/strtof -- the conversions sscanf("%d%d%f") performs.  Kept from
// the reference: at most `declared` entries are read, reading stops at the first line that does not start with two
// integers, a missing third field means 1.0, MatrixMarket indices are 1-based, isTrans swaps the two.  Symmetric
// MatrixMarket files are a token stream in the reference (fscanf, not lines) and stay on the sequential path.
// (One deliberate difference: fgets cuts lines at 1024 characters; here a line ends at its newline.)
namespace {
struct LineCursor {                         // fgets over a memory buffer
  const char* p; const char* end;
  bool next(std::string& out) {
    if (p >= end) return false;
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* stop = nl ? nl + 1 : end;
    out.assign(p, stop);
    p = stop;
    return true;
  }
};

// parse "from to [val]" from the NUL-terminated line s; returns the number of fields sscanf("%d%d%f") would convert
inline int parse_edge(const char* s, int* from, int* to, float* val) {
  char* e = nullptr;
  const long a = strtol(s, &e, 10);
  if (e == s) return 0;
  const char* s2 = e;
  const long b = strtol(s2, &e, 10);
  if (e == s2) return 1;
  *from = (int)a; *to = (int)b;
  const char* s3 = e;
  const float v = strtof(s3, &e);
  if (e == s3) return 2;
  *val = v;
  return 3;
}

size_t count_lines(const char* b, const char* e) {
  size_t n = 0;
  while (b < e) {
    const char* nl = (const char*)memchr(b, '\n', (size_t)(e - b));
    ++n;
    if (!nl) break;
    b = nl + 1;
  }
  return n;
}
}  // namespace

double COO::lastParseMs = 0.0;
int COO::lastParseThreads = 0;

int COO::readSNAPFile(const char fname[], bool isTrans) {
  const auto t0 = std::chrono::steady_clock::now();
  FILE* fp = fopen(fname, "rb");
  if (!fp) { printf("Failed to open file %s\n", fname); exit(-1); }
  fseek(fp, 0, SEEK_END);
  const long fsize = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  char* buf = grab<char>((size_t)std::max(fsize, 0l) + 1, "file buffer");
  const size_t len = fsize > 0 ? fread(buf, 1, (size_t)fsize, fp) : 0;
  fclose(fp);
  buf[len] = '\0';
  rows = cols = nnz = 0;
  LineCursor cur{buf, buf + len};
  std::string line;
  bool isMtx = false, symmetric = false;
  if (!cur.next(line)) { free(buf); return 0; }
  if (line[0] == '%') {                      // MatrixMarket banner: 5 tokens, the 5th is the storage scheme
    char t[5][64];
    if (sscanf(line.c_str(), "%63s %63s %63s %63s %63s", t[0], t[1], t[2], t[3], t[4]) == 5) {
      isMtx = true;
      symmetric = lower(t[4]) == "symmetric";
    }
  }
  bool more = true;
  while (line[0] == '#' || line[0] == '%') { more = cur.next(line); if (!more) break; }
  if (!more) { free(buf); nnz = 0; return 0; }
  int a = 0, b = 0, c = 0;
  const int got = sscanf(line.c_str(), "%d %d %d", &a, &b, &c);
  int declared;
  if (got == 2) { rows = cols = a; declared = b; }
  else if (got == 3) { rows = a; cols = b; declared = c; }
  else { printf("%s: cannot parse the size line\n", fname); exit(-1); }
  printf("rows=%d cols=%d nnz=%d\n", rows, cols, declared);
  if (declared < 0) declared = 0;
  const size_t cap = (size_t)declared * (symmetric ? 2 : 1);
  cooRowIndex = grab<int>(cap, "cooRowIndex");
  cooColIndex = grab<int>(cap, "cooColIndex");
  cooVal = grab<QValue>(cap, "cooVal");
  int top = 0;
  lastParseThreads = 1;
  if (symmetric) {                           // token stream (the reference uses fscanf here): sequential
    const char* p = cur.p;
    for (int i = 0; i < declared; ++i) {
      int from = 0, to = 0;
      float val = 0.f;
      char* e = nullptr;
      const long x = strtol(p, &e, 10);
      if (e == p) break;
      p = e;
      const long y = strtol(p, &e, 10);
      if (e == p) break;
      p = e;
      from = (int)x; to = (int)y;
      const float v = strtof(p, &e);
      if (e == p) val = 1.0f; else { val = v; p = e; }
      if (isMtx) { --from; --to; }
      cooRowIndex[top] = from; cooColIndex[top] = to; cooVal[top++] = val;
      if (from != to) { cooRowIndex[top] = to; cooColIndex[top] = from; cooVal[top++] = val; }
    }
  } else {
    char* const d0 = buf + (cur.p - buf);
    char* const d1 = buf + len;
    const size_t bytes = (size_t)(d1 - d0);
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = getenv("SMF_PARSE_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    const int T = (int)std::max<size_t>(1, std::min<size_t>(std::min<unsigned>(hw ? hw : 1, 64), bytes / (1 << 16) + 1));
    lastParseThreads = T;
    std::vector<char*> cut((size_t)T + 1);
    cut[0] = d0; cut[T] = d1;
    for (int t = 1; t < T; ++t) {            // chunk t starts right behind a newline
      char* p = d0 + bytes * (size_t)t / (size_t)T;
      if (p < cut[t - 1]) p = cut[t - 1];
      char* nl = (char*)memchr(p, '\n', (size_t)(d1 - p));
      cut[t] = nl ? nl + 1 : d1;
    }
    std::vector<size_t> nlines((size_t)T), first((size_t)T + 1);
    std::vector<long long> badAt((size_t)T, -1);   // file-order index of the chunk's first line that is not an edge
    {
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] { nlines[t] = count_lines(cut[t], cut[t + 1]); });
      for (auto& x : th) x.join();
    }
    first[0] = 0;
    for (int t = 0; t < T; ++t) first[t + 1] = first[t] + nlines[t];
    {
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        char* p = cut[t];
        char* const e = cut[t + 1];
        size_t k = first[t];
        while (p < e && k < (size_t)declared) {
          char* nl = (char*)memchr(p, '\n', (size_t)(e - p));
          char* stop = nl ? nl : e;           // (e == d1 points at the buffer's terminating NUL)
          const char saved = *stop;
          *stop = '\0';
          int from = 0, to = 0;
          float val = 0.f;
          const int ret = parse_edge(p, &from, &to, &val);
          *stop = saved;
          if (ret < 2) { badAt[t] = (long long)k; return; }
          if (ret == 2) val = 1.0f;
          if (isMtx) { --from; --to; }
          cooRowIndex[k] = isTrans ? to : from;
          cooColIndex[k] = isTrans ? from : to;
          cooVal[k] = val;
          ++k;
          p = nl ? nl + 1 : e;
        }
      });
      for (auto& x : th) x.join();
    }
    size_t good = std::min<size_t>(first[T], (size_t)declared);
    for (int t = 0; t < T; ++t)
      if (badAt[t] >= 0) { good = std::min<size_t>(good, (size_t)badAt[t]); break; }
    top = (int)good;
  }
  free(buf);
  nnz = top;
  lastParseMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
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
