/* oracle/oracle.c — CPU restatement of the reference's SpGEMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libspgemm_hip.so) never
 * links, loads or falls back to it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against
 *   (a) the real reference compiled into oracle/_ref/libref.so (tests/test_oracle_vs_ref.py,
 *       runs wherever _ref exists), and
 *   (b) the golden vectors under tests/golden/ that were generated from that library by
 *       tests/golden/make_golden.py (runs everywhere, incl. the GPU box), and
 *   (c) the known answers the reference tree itself holds (test2.mtx A*A,
 *       t2.snap R-MCL result).
 *
 * Plain C11 + OpenMP.  All citations are relative to /root/reference/.
 * QValue is float (nlibs/tools/macro.h:5), indices are int (nlibs/CSR.h:32-38).
 * Built with -ffp-contract=off semantics in mind: no FMA is used so a*b+c rounds twice,
 * exactly as the reference's x86-64 build does.
 */
#define _GNU_SOURCE
#include <limits.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#if defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

typedef float QValue;

static void* xmalloc(size_t n) {
  void* p = malloc(n ? n : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n); exit(EXIT_FAILURE); }
  return p;
}
static void* xcalloc(size_t n, size_t s) {
  void* p = calloc(n ? n : 1, s ? s : 1);
  if (!p) { fprintf(stderr, "oracle: out of memory\n"); exit(EXIT_FAILURE); }
  return p;
}

void oracle_free(void* p) { free(p); }
int oracle_max_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------------------------------
 * Sequential two-phase Gustavson SpGEMM — THE parity oracle.
 * Follows nlibs/cpu_csr_kernel.cc:3-35 (symbolic, rezero_xb) and :76-119 (numeric).
 * Column order inside a C row is first-touch order; structural zeros are kept; the sum for
 * one C entry runs over ascending jp then tp in float32.
 * ------------------------------------------------------------------------------------------ */
int oracle_sequential_spmm(const int* IA, const int* JA, const QValue* A, int nnzA,
                           const int* IB, const int* JB, const QValue* B, int nnzB,
                           int** ICp, int** JCp, QValue** Cp, int* nnzCp, int m, int k, int n) {
  (void)nnzA; (void)nnzB; (void)k;
  int* IC = (int*)xcalloc((size_t)m + 1, sizeof(int));
  unsigned char* xb = (unsigned char*)xcalloc((size_t)n, 1);
  int* iJC = (int*)xcalloc((size_t)n + 1, sizeof(int));
  /* symbolic: cpu_csr_kernel.cc:24-35 */
  IC[0] = 0;
  for (int i = 0; i < m; ++i) {
    int ip = IC[i];
    const int startp = ip;
    for (int vp = IA[i]; vp < IA[i + 1]; ++vp) {
      const int v = JA[vp];
      for (int kp = IB[v]; kp < IB[v + 1]; ++kp) {
        const int c = JB[kp];
        if (!xb[c]) { iJC[ip - startp] = c; ++ip; xb[c] = 1; }
      }
    }
    for (int jp = IC[i]; jp < ip; ++jp) xb[iJC[jp - startp]] = 0;
    IC[i + 1] = ip;
  }
  free(iJC);
  const int nnzC = IC[m];
  int* JC = (int*)xmalloc(sizeof(int) * (size_t)nnzC);
  QValue* C = (QValue*)xmalloc(sizeof(QValue) * (size_t)nnzC);
  QValue* x = (QValue*)xcalloc((size_t)n, sizeof(QValue));
  /* numeric: cpu_csr_kernel.cc:95-116 */
  int ip = 0;
  for (int i = 0; i < m; ++i) {
    for (int jp = IA[i]; jp < IA[i + 1]; ++jp) {
      const int j = JA[jp];
      const QValue a = A[jp];
      for (int tp = IB[j]; tp < IB[j + 1]; ++tp) {
        const int t = JB[tp];
        if (!xb[t]) { JC[ip++] = t; xb[t] = 1; x[t] = a * B[tp]; }
        else        { x[t] += a * B[tp]; }
      }
    }
    for (int vp = IC[i]; vp < ip; ++vp) {
      const int v = JC[vp];
      C[vp] = x[v]; x[v] = 0; xb[v] = 0;
    }
  }
  free(xb); free(x);
  *ICp = IC; *JCp = JC; *Cp = C; *nnzCp = nnzC;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Row kernels shared by the parallel variants.
 * row_count      <- cRowiCount           nlibs/cpu_csr_kernel.h:234-262
 * row_numeric    <- indexProcessCRowI    nlibs/cpu_csr_kernel.h:135-188
 * Both treat the first A entry specially (its B row is taken without a membership test),
 * exactly as the reference does.
 * ------------------------------------------------------------------------------------------ */
static inline int row_count(int i, const int* IA, const int* JA, const int* IB, const int* JB,
                            int* iJC, unsigned char* xb) {
  if (IA[i] == IA[i + 1]) return 0;
  int count = -1;
  {
    const int v = JA[IA[i]];
    for (int kp = IB[v]; kp < IB[v + 1]; ++kp) { const int c = JB[kp]; iJC[++count] = c; xb[c] = 1; }
  }
  for (int vp = IA[i] + 1; vp < IA[i + 1]; ++vp) {
    const int v = JA[vp];
    for (int kp = IB[v]; kp < IB[v + 1]; ++kp) {
      const int c = JB[kp];
      if (!xb[c]) { iJC[++count] = c; xb[c] = 1; }
    }
  }
  ++count;
  for (int jp = 0; jp < count; ++jp) xb[iJC[jp]] = 0;
  return count;
}

static inline int row_numeric(int* restrict index, int iAnnz, const int* iJA, const QValue* iA,
                              const int* IB, const int* JB, const QValue* B,
                              int* restrict iJC, QValue* restrict iC) {
  if (iAnnz == 0) return 0;
  int ip = -1;
  {
    const int j = iJA[0];
    for (int tp = IB[j]; tp < IB[j + 1]; ++tp) {
      const int t = JB[tp];
      iJC[++ip] = t; index[t] = ip; iC[ip] = iA[0] * B[tp];
    }
  }
  for (int jp = 1; jp < iAnnz; ++jp) {
    const int j = iJA[jp];
    for (int tp = IB[j]; tp < IB[j + 1]; ++tp) {
      const int t = JB[tp];
      if (index[t] == -1) { iJC[++ip] = t; index[t] = ip; iC[ip] = iA[jp] * B[tp]; }
      else                { iC[index[t]] += iA[jp] * B[tp]; }
    }
  }
  ++ip;
  for (int vp = 0; vp < ip; ++vp) index[iJC[vp]] = -1;
  return ip;
}

/* exclusive scan of a[0..n) into s[0..n], s[n] = total — noTileOmpPrefixSum semantics
 * (nlibs/tools/prefixSum.cc:31-61); must be called by every thread of a parallel region. */
static void omp_exclusive_scan_int(int* a, int n, int* partial /* nthreads+1 */) {
  const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
  const int chunk = (n + nt - 1) / nt;
  int lo = chunk * tid; if (lo > n) lo = n;
  int hi = lo + chunk; if (hi > n) hi = n;
  int sum = 0;
  for (int i = lo; i < hi; ++i) { const int t = a[i]; a[i] = sum; sum += t; }
  partial[tid + 1] = sum;
#pragma omp barrier
#pragma omp single
  { partial[0] = 0; for (int t = 0; t < nt; ++t) partial[t + 1] += partial[t]; }
  const int off = partial[tid];
  for (int i = lo; i < hi; ++i) a[i] += off;
#pragma omp barrier
#pragma omp single
  { a[n] = partial[nt]; }
}

/* ------------------------------------------------------------------------------------------
 * OpenMP two-phase SpGEMM — the CPU baseline named by BASELINE.md.
 * Follows omp_CSR_SpMM, nlibs/omp_csr_kernel.cc:238-315 (+ omp_CSR_IC_nnzC :103-124,
 * allocateThreadDatas :13-40): per-thread dense xb[n] / index[n], schedule(dynamic) over blocks
 * of `stride` rows, parallel exclusive scan, one malloc of JC/C.  Scratch allocation is inside
 * the call, as in the reference's 4-argument wrapper (:295-315).
 * ------------------------------------------------------------------------------------------ */
int oracle_omp_spmm(const int* IA, const int* JA, const QValue* A, int nnzA,
                    const int* IB, const int* JB, const QValue* B, int nnzB,
                    int** ICp, int** JCp, QValue** Cp, int* nnzCp, int m, int k, int n, int stride) {
  (void)nnzA; (void)nnzB; (void)k;
  if (stride <= 0) stride = 512;
  const int nthreads = omp_get_max_threads();
  int* IC = (int*)xmalloc(((size_t)m + 1) * sizeof(int));
  int* partial = (int*)xcalloc((size_t)nthreads + 2, sizeof(int));
  unsigned char** xbs = (unsigned char**)xcalloc((size_t)nthreads, sizeof(*xbs));
  int** idxs = (int**)xcalloc((size_t)nthreads, sizeof(*idxs));
  for (int t = 0; t < nthreads; ++t) {
    xbs[t] = (unsigned char*)xmalloc((size_t)n + 64);
    idxs[t] = (int*)xmalloc((size_t)n * sizeof(int) + 64);
  }
  int* JC = NULL; QValue* C = NULL; int nnzC = 0;
#pragma omp parallel num_threads(nthreads)
  {
    const int tid = omp_get_thread_num();
    unsigned char* xb = xbs[tid];
    int* index = idxs[tid];
    memset(xb, 0, (size_t)n);
#pragma omp for schedule(dynamic)
    for (int it = 0; it < m; it += stride) {
      const int up = it + stride < m ? it + stride : m;
      for (int i = it; i < up; ++i) IC[i] = row_count(i, IA, JA, IB, JB, index, xb);
    }
    if (m > 0) omp_exclusive_scan_int(IC, m, partial);
#pragma omp barrier
#pragma omp master
    {
      if (m == 0) IC[0] = 0;
      nnzC = IC[m];
      JC = (int*)xmalloc(sizeof(int) * (size_t)nnzC);
      C = (QValue*)xmalloc(sizeof(QValue) * (size_t)nnzC);
    }
    memset(index, -1, (size_t)n * sizeof(int));
#pragma omp barrier
#pragma omp for schedule(dynamic) nowait
    for (int it = 0; it < m; it += stride) {
      const int up = it + stride < m ? it + stride : m;
      for (int i = it; i < up; ++i)
        row_numeric(index, IA[i + 1] - IA[i], JA + IA[i], A + IA[i], IB, JB, B, JC + IC[i], C + IC[i]);
    }
  }
  for (int t = 0; t < nthreads; ++t) { free(xbs[t]); free(idxs[t]); }
  free(xbs); free(idxs); free(partial);
  *ICp = IC; *JCp = JC; *Cp = C; *nnzCp = nnzC;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Per-row product counts ("flops"): rowFlops[i] = sum_{jp in A_i} (IB[JA[jp]+1] - IB[JA[jp]]).
 * Follows dynamic_omp_CSR_flops, nlibs/flops_csr_kernel.cc:14-31 (64-bit) and gcomputeFlops,
 * mindex2-cuda/flops.cu:66-83.  prefix!=0 additionally leaves the reference's exclusive prefix
 * layout: out[i] = sum of rows < i, out[m] = P.
 * ------------------------------------------------------------------------------------------ */
void oracle_row_flops(const int* IA, const int* JA, const int* IB, int m, long long* out, int prefix) {
#pragma omp parallel for schedule(dynamic, 512)
  for (int i = 0; i < m; ++i) {
    long long f = 0;
    for (int jp = IA[i]; jp < IA[i + 1]; ++jp) { const int j = JA[jp]; f += IB[j + 1] - IB[j]; }
    out[i] = f;
  }
  if (prefix) {
    long long s = 0;
    for (int i = 0; i < m; ++i) { const long long t = out[i]; out[i] = s; s += t; }
    out[m] = s;
  }
}

/* arrayEqualPartition64, nlibs/tools/util.cc:123-135: cut rows into `parts` contiguous ranges
 * of ~equal prefix mass; every part gets at least one row while rows remain. */
void oracle_equal_partition64(const long long* prefix, int n, int parts, int* ends) {
  const long long total = prefix[n];
  const long long chunk = (total + parts - 1) / parts;
  ends[0] = 0;
  int now = 0;
  for (int i = 0; i < parts - 1; ++i) {
    long long target = (long long)(i + 1) * chunk;
    if (target > total) target = total;
    /* upper_bound(prefix+now, prefix+n+1, target) */
    int lo = now, hi = n + 1;
    while (lo < hi) { const int mid = lo + (hi - lo) / 2; if (prefix[mid] <= target) lo = mid + 1; else hi = mid; }
    int e = lo - 1;
    if (e < now + 1) e = now + 1;
    if (e > n) e = n;
    ends[i + 1] = e;
    now = e;
  }
  ends[parts] = n;
}

/* CPU binning, group_CSR_flops nlibs/group_csr_kernel.cc:10-52: 7 groups by flops
 * {<=0, <=1, <=2, <=4, <=8, <=16, >16}; stable counting sort of row ids into groups[];
 * tops[8] = group boundaries. */
void oracle_group_bins(const int* IA, const int* JA, const int* IB, int m,
                       int* rowFlops, int* groups, int* tops) {
  int cnt[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < m; ++i) {
    int f = 0;
    for (int jp = IA[i]; jp < IA[i + 1]; ++jp) { const int j = JA[jp]; f += IB[j + 1] - IB[j]; }
    rowFlops[i] = f;
    const int g = f <= 0 ? 0 : f <= 1 ? 1 : f <= 2 ? 2 : f <= 4 ? 3 : f <= 8 ? 4 : f <= 16 ? 5 : 6;
    ++cnt[g];
  }
  tops[0] = 0;
  for (int g = 0; g < 7; ++g) tops[g + 1] = tops[g] + cnt[g];
  int cur[7];
  for (int g = 0; g < 7; ++g) cur[g] = tops[g];
  for (int i = 0; i < m; ++i) {
    const int f = rowFlops[i];
    const int g = f <= 0 ? 0 : f <= 1 ? 1 : f <= 2 ? 2 : f <= 4 ? 3 : f <= 8 ? 4 : f <= 16 ? 5 : 6;
    groups[cur[g]++] = i;
  }
}

/* GPU HEAD bin id, dqueueId mindex2-cuda/flops.cu:39-47:
 * 0->1, 1->2, 2..4->3, 5..16->4, 17..64->5, 65..512->6, >512->7. */
int oracle_gpu_bin_id(long long x) {
  if (x == 0) return 1;
  if (x == 1) return 2;
  if (x > 512) return 7;
  if (x > 64) return 6;
  if (x > 16) return 5;
  if (x > 4) return 4;
  return 3;
}

/* gpuFlopsClassify, mindex2-cuda/flops.cu:110-185, restated without thrust:
 *   rowIds[m]    rows stably sorted by ascending flops            (:131)
 *   flopsScan[m+1] flopsScan[0]=0, flopsScan[1+q] = inclusive scan of the sorted flops (:119,:133)
 *   hv[9]        hv[0]=0, hv[b+1] = #{elements of the (m+1)-long bin array with bin <= b}, where the
 *                array is {dummy flops 0 -> bin 1} followed by the sorted rows (:96-107,:132);
 *                the reference sizes hv to max bin + 2; we always fill 9 entries (trailing = m+1).
 * Returns the number of hv entries the reference would have produced. */
int oracle_gpu_classify(const long long* rowFlops, int m, int* rowIds, long long* flopsScan, int* hv) {
  /* stable counting-free sort: indices by (flops, row) */
  int* tmp = (int*)xmalloc(sizeof(int) * (size_t)(m ? m : 1));
  for (int i = 0; i < m; ++i) rowIds[i] = i;
  /* bottom-up stable merge sort on flops */
  for (int w = 1; w < m; w *= 2) {
    for (int lo = 0; lo < m; lo += 2 * w) {
      int mid = lo + w < m ? lo + w : m, hi = lo + 2 * w < m ? lo + 2 * w : m;
      int a = lo, b = mid, o = lo;
      while (a < mid && b < hi) tmp[o++] = (rowFlops[rowIds[b]] < rowFlops[rowIds[a]]) ? rowIds[b++] : rowIds[a++];
      while (a < mid) tmp[o++] = rowIds[a++];
      while (b < hi) tmp[o++] = rowIds[b++];
    }
    memcpy(rowIds, tmp, sizeof(int) * (size_t)m);
  }
  free(tmp);
  int cnt[9] = {0};
  cnt[1] = 1; /* dummy element 0 has flops 0 -> bin 1 */
  int maxbin = 1;
  flopsScan[0] = 0;
  for (int q = 0; q < m; ++q) {
    const long long f = rowFlops[rowIds[q]];
    flopsScan[q + 1] = flopsScan[q] + f;
    const int b = oracle_gpu_bin_id(f);
    ++cnt[b];
    if (b > maxbin) maxbin = b;
  }
  hv[0] = 0;
  for (int b = 0; b < 8; ++b) hv[b + 1] = hv[b] + cnt[b];
  return maxbin + 2;
}

/* ------------------------------------------------------------------------------------------
 * COO side: text loader, sort/dedupe, CSR conversion.  nlibs/COO.cc:48-291.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int r, c; QValue v; int seq; } Tup;

static int tup_cmp(const void* a, const void* b) {
  const Tup* x = (const Tup*)a; const Tup* y = (const Tup*)b;
  if (x->r != y->r) return x->r < y->r ? -1 : 1;
  if (x->c != y->c) return x->c < y->c ? -1 : 1;
  return x->seq < y->seq ? -1 : (x->seq > y->seq);   /* stable: input order among duplicates */
}

/* readSNAPFile, nlibs/COO.cc:48-158.  Returns 0 and malloc'd triplets.  Semantics kept:
 *  - first line starting with '%' and holding 5 tokens => MatrixMarket banner => 1-based indices,
 *    5th token lower-cased is the storage scheme ("symmetric" expands (i,j)->(j,i), i!=j);
 *  - lines starting with '#' or '%' are skipped; the next line is the size line: 2 ints
 *    "rows nnz" (cols=rows) or 3 ints "rows cols nnz";
 *  - entry lines: "from to [val]" (missing val => 1.0); isTrans swaps row/col (general only). */
int oracle_read_snap(const char* fname, int isTrans, int* rows, int* cols, int* nnz,
                     int** rip, int** cip, QValue** vp) {
  FILE* fp = fopen(fname, "r");
  if (!fp) return -1;
  enum { LMAX = 1025 };
  char line[LMAX], banner[64] = "", mtx[64] = "", crd[64] = "", dtype[64] = "", scheme[64] = "unsymmetric";
  int isMtx = 0;
  *rows = *cols = *nnz = 0; *rip = *cip = NULL; *vp = NULL;
  if (!fgets(line, LMAX, fp)) { fclose(fp); return 0; }
  if (line[0] == '%' && !feof(fp)) {
    char s5[64];
    if (sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, dtype, s5) == 5) {
      for (char* p = s5; *p; ++p) *p = (char)tolower((unsigned char)*p);
      strcpy(scheme, s5);
      isMtx = 1;
    }
  }
  while ((line[0] == '#' || line[0] == '%') && !feof(fp)) { if (!fgets(line, LMAX, fp)) break; }
  if (feof(fp)) { fclose(fp); return 0; }
  int f2 = 0, f3 = 0, r0 = 0;
  const int got = sscanf(line, "%d %d %d", &r0, &f2, &f3);
  int n0;
  if (got == 2) { *rows = r0; *cols = r0; n0 = f2; }
  else if (got == 3) { *rows = r0; *cols = f2; n0 = f3; }
  else { fclose(fp); return -2; }
  const int sym = strcmp(scheme, "symmetric") == 0;
  const size_t cap = (size_t)n0 * (sym ? 2 : 1);
  int* ri = (int*)xmalloc(cap * sizeof(int));
  int* ci = (int*)xmalloc(cap * sizeof(int));
  QValue* v = (QValue*)xmalloc(cap * sizeof(QValue));
  int top = 0;
  for (int i = 0; i < n0; ++i) {
    int from = 0, to = 0; float val = 0;
    int ret;
    if (sym) { ret = fscanf(fp, "%d%d%f", &from, &to, &val); }
    else { if (!fgets(line, LMAX, fp)) break; ret = sscanf(line, "%d%d%f", &from, &to, &val); }
    if (ret < 2) break;
    if (ret == 2) val = 1.0f;
    if (isMtx) { --from; --to; }
    if (sym) {
      ri[top] = from; ci[top] = to; v[top++] = val;
      if (from != to) { ri[top] = to; ci[top] = from; v[top++] = val; }
    } else {
      if (isTrans) { ri[top] = to; ci[top] = from; } else { ri[top] = from; ci[top] = to; }
      v[top++] = val;
    }
  }
  fclose(fp);
  *nnz = top; *rip = ri; *cip = ci; *vp = v;
  return 0;
}

/* COO::makeOrdered (dedupe=0, nlibs/COO.cc:222-235) / orderedAndDuplicatesRemoving (dedupe=1,
 * :237-266): sort by (row, col); duplicates are summed left to right.  The reference uses the
 * unstable std::sort, so with >=3 duplicates of one (row,col) its float sum order is unspecified;
 * we sum in input order.  In place; returns the new nnz. */
int oracle_coo_sort(int nnz, int* ri, int* ci, QValue* v, int dedupe) {
  if (nnz <= 0) return 0;
  Tup* t = (Tup*)xmalloc(sizeof(Tup) * (size_t)nnz);
  for (int i = 0; i < nnz; ++i) { t[i].r = ri[i]; t[i].c = ci[i]; t[i].v = v[i]; t[i].seq = i; }
  qsort(t, (size_t)nnz, sizeof(Tup), tup_cmp);
  int j = 0;
  if (dedupe) {
    for (int i = 1; i < nnz; ++i) {
      if (t[i].r == t[j].r && t[i].c == t[j].c) t[j].v += t[i].v;
      else t[++j] = t[i];
    }
    nnz = j + 1;
  }
  for (int i = 0; i < nnz; ++i) { ri[i] = t[i].r; ci[i] = t[i].c; v[i] = t[i].v; }
  free(t);
  return nnz;
}

/* COO::toCSR, nlibs/COO.cc:268-291 (input must be sorted by row). rowPtr has rows+1 entries. */
void oracle_coo_to_csr(int rows, int nnz, const int* ri, int* rowPtr) {
  memset(rowPtr, 0, sizeof(int) * ((size_t)rows + 1));
  for (int t = 0; t < nnz; ++t) if (ri[t] >= 0 && ri[t] < rows) ++rowPtr[ri[t] + 1];
  for (int i = 0; i < rows; ++i) rowPtr[i + 1] += rowPtr[i];
}

/* COO::addSelfLoopIfNeeded, nlibs/COO.cc:160-188: append (i,i,1.0) for every i with no diagonal
 * entry.  (The reference's count is wrong when a diagonal entry is duplicated; inputs here are
 * assumed free of duplicated diagonals.)  Returns the new nnz; arrays are realloc'd. */
int oracle_add_self_loops(int rows, int nnz, int** rip, int** cip, QValue** vp) {
  unsigned char* u = (unsigned char*)xcalloc((size_t)rows, 1);
  for (int i = 0; i < nnz; ++i) if ((*rip)[i] == (*cip)[i] && (*rip)[i] >= 0 && (*rip)[i] < rows) u[(*rip)[i]] = 1;
  int need = 0;
  for (int i = 0; i < rows; ++i) need += !u[i];
  *rip = (int*)realloc(*rip, sizeof(int) * (size_t)(nnz + need + 1));
  *cip = (int*)realloc(*cip, sizeof(int) * (size_t)(nnz + need + 1));
  *vp = (QValue*)realloc(*vp, sizeof(QValue) * (size_t)(nnz + need + 1));
  int top = nnz;
  for (int i = 0; i < rows; ++i) if (!u[i]) { (*rip)[top] = i; (*cip)[top] = i; (*vp)[top++] = 1.0f; }
  free(u);
  return top;
}

/* CSR::makeOrdered, nlibs/CSR.cc:73-86: sort each row by (col, value). */
typedef struct { int c; QValue v; } CV;
static int cv_cmp(const void* a, const void* b) {
  const CV* x = (const CV*)a; const CV* y = (const CV*)b;
  if (x->c != y->c) return x->c < y->c ? -1 : 1;
  return x->v < y->v ? -1 : (x->v > y->v);
}
void oracle_csr_make_ordered(int rows, const int* rowPtr, int* colInd, QValue* values) {
#pragma omp parallel
  {
    CV* buf = NULL; int cap = 0;
#pragma omp for schedule(dynamic, 256)
    for (int i = 0; i < rows; ++i) {
      const int s = rowPtr[i], e = rowPtr[i + 1], len = e - s;
      if (len < 2) continue;
      if (len > cap) { free(buf); cap = len * 2; buf = (CV*)xmalloc(sizeof(CV) * (size_t)cap); }
      for (int q = 0; q < len; ++q) { buf[q].c = colInd[s + q]; buf[q].v = values[s + q]; }
      qsort(buf, (size_t)len, sizeof(CV), cv_cmp);
      for (int q = 0; q < len; ++q) { colInd[s + q] = buf[q].c; values[s + q] = buf[q].v; }
    }
    free(buf);
  }
}

/* CSR::averAndNormRowQValue, nlibs/CSR.cc:88-95: every entry of row i becomes 1/count(i)
 * (double division, then narrowed to float). */
void oracle_csr_aver_norm(int rows, const int* rowPtr, QValue* values) {
  for (int i = 0; i < rows; ++i) {
    const int count = rowPtr[i + 1] - rowPtr[i];
    for (int j = rowPtr[i]; j < rowPtr[i + 1]; ++j) values[j] = (QValue)(1.0 / count);
  }
}

/* ------------------------------------------------------------------------------------------
 * R-MCL post-step (inflate / prune / normalise) — nlibs/tools/util.cc:4-69, constants
 * nlibs/tools/util.h:11-12, driver nlibs/qrmcl.cc:86-124.  Arithmetic types mirror the
 * reference exactly: MLMCL_PRUNE_A is the double literal 0.90, MLMCL_PRUNE_B the int 2.
 * ------------------------------------------------------------------------------------------ */
QValue oracle_compute_threshold(QValue avg, QValue max) {
  QValue ret = (QValue)(0.90 * avg * (1 - 2 * (max - avg)));
  ret = (QValue)((ret > 1.0e-7) ? ret : 1.0e-7);
  ret = (ret > max) ? max : ret;
  return ret;
}

/* One row, in place on (cols, vals) of length count; returns the kept count.
 * qrmcl.cc:99-111: inflate (square), max (from 0.0), sum, thresh, keep v>=thresh, divide by kept sum. */
int oracle_rmcl_prune_row(int count, int* cols, QValue* vals) {
  QValue rmax = 0.0f, rsum = 0.0f;
  for (int i = 0; i < count; ++i) vals[i] = vals[i] * vals[i];
  for (int i = 0; i < count; ++i) if (rmax < vals[i]) rmax = vals[i];
  for (int i = 0; i < count; ++i) rsum += vals[i];
  const QValue thresh = oracle_compute_threshold(rsum / count, rmax);
  QValue sum = 0; int j = 0;
  for (int i = 0; i < count; ++i) {
    if (vals[i] >= thresh) { sum += vals[i]; cols[j] = cols[i]; vals[j++] = vals[i]; }
  }
  for (int i = 0; i < j; ++i) vals[i] = vals[i] / sum;
  return j;
}

/* The per-iteration tail of seqRmclIter (nlibs/qrmcl.cc:96-117): prune every row of C in place and
 * compact rows to the front; IC is rewritten to the new offsets.  Returns the new nnz. */
int oracle_rmcl_prune_compact(int rows, int* IC, int* JC, QValue* C) {
  int pos = 0;
  for (int i = 0; i < rows; ++i) {
    const int s = IC[i], count = IC[i + 1] - IC[i];
    const int kept = oracle_rmcl_prune_row(count, JC + s, C + s);
    memmove(JC + pos, JC + s, sizeof(int) * (size_t)kept);
    memmove(C + pos, C + s, sizeof(QValue) * (size_t)kept);
    IC[i] = pos; pos += kept;
  }
  IC[rows] = pos;
  return pos;
}

/* seqRmclIter, nlibs/qrmcl.cc:86-124: Mt <- prune(Mgt * Mt), maxIters times.  Mt arrays are
 * consumed (freed) and replaced by malloc'd results. */
int oracle_rmcl_iters(int maxIters, int rows, int cols,
                      const int* gIA, const int* gJA, const QValue* gA, int gnnz,
                      int** tIAp, int** tJAp, QValue** tAp, int* tnnzp) {
  int* IB = *tIAp; int* JB = *tJAp; QValue* B = *tAp; int nnzB = *tnnzp;
  for (int iter = 0; iter < maxIters; ++iter) {
    int *IC, *JC, nnzC; QValue* C;
    oracle_sequential_spmm(gIA, gJA, gA, gnnz, IB, JB, B, nnzB, &IC, &JC, &C, &nnzC, rows, cols, cols);
    const int pos = oracle_rmcl_prune_compact(rows, IC, JC, C);
    free(IB); free(JB); free(B);
    IB = IC; JB = JC; B = C; nnzB = pos;
  }
  *tIAp = IB; *tJAp = JB; *tAp = B; *tnnzp = nnzB;
  return 0;
}

/* rmclInit, nlibs/qrmcl.cc:126-134: self loops, sort (no dedupe), toCSR, row-normalise.
 * Consumes the COO arrays (may realloc them); outputs malloc'd CSR. */
int oracle_rmcl_init(int rows, int nnz, int** rip, int** cip, QValue** vp,
                     int** rowPtrp, int* nnzOut) {
  nnz = oracle_add_self_loops(rows, nnz, rip, cip, vp);
  nnz = oracle_coo_sort(nnz, *rip, *cip, *vp, 0);
  int* rp = (int*)xmalloc(sizeof(int) * ((size_t)rows + 1));
  oracle_coo_to_csr(rows, nnz, *rip, rp);
  oracle_csr_aver_norm(rows, rp, *vp);
  *rowPtrp = rp; *nnzOut = nnz;
  return 0;
}

/* pushToStats + flopsStats, nlibs/tools/stats.cc:3-12,45-55: 13 buckets, bucket i takes the first i with
 * flops <= 2^i, the last bucket the rest. */
void oracle_flops_stats(const int* IA, const int* JA, const int* IB, int m, int* stats /*[13]*/) {
  for (int i = 0; i < 13; ++i) stats[i] = 0;
  for (int i = 0; i < m; ++i) {
    long row_flops = 0;
    for (int jp = IA[i]; jp < IA[i + 1]; ++jp) row_flops += IB[JA[jp] + 1] - IB[JA[jp]];
    int b = 12;
    for (int q = 0; q < 12; ++q) if (row_flops <= (1l << q)) { b = q; break; }
    ++stats[b];
  }
}
