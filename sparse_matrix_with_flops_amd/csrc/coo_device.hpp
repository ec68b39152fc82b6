// COO -> CSR on the device: the step in front of the SpGEMM path.
//
// Replaces, on device arrays,
//   COO::addSelfLoopIfNeeded                       nlibs/COO.cc:160-188
//   COO::makeOrdered / orderedAndDuplicatesRemoving nlibs/COO.cc:222-266   (std::sort on (row,col) tuples; duplicates summed)
//   COO::toCSR                                      nlibs/COO.cc:268-291
//   CSR::averAndNormRowQValue                       nlibs/CSR.cc:88-95      (every entry of row i becomes 1/count(i))
//   CSR::toAbs                                      nlibs/CSR.h:152-158
// i.e. what rmclInit (nlibs/qrmcl.cc:126-134) and the Matrix-Market/SNAP loaders do after parsing the text.
//
// The sort is our own stable LSD radix sort (8 bits per pass: per-block digit histograms in LDS, the library's scan
// kernels over the digit-major histogram, ballot-ranked stable scatter) of the 64-bit key row*cols+col carrying the
// entry's input position, over exactly the bits the shape needs.  Stable order = input order inside a run of equal
// (row,col), so duplicates are summed left to right like the CPU loop (bit-exact floats; a run is summed by one lane,
// which is what keeps the order -- a pair repeated very many times serialises that lane).
// Included at the end of spgemm_hip.hip (uses its pool / error helpers and the k_scan_* kernels).
#pragma once

namespace coo {

constexpr int RS_THREADS = 256, RS_ITEMS = 8, RS_TILE = RS_THREADS * RS_ITEMS, RS_RADIX = 256;

// pass 1 of a digit: how many keys of every digit value each block holds; blockHist is digit-major [256][nblk]
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(int n, const unsigned long long* __restrict__ keys, int shift,
                                                           int nblk, int* __restrict__ blockHist) {
  __shared__ int hist[RS_RADIX];
  const int tid = threadIdx.x;
  hist[tid] = 0;
  __syncthreads();
  const long long base = (long long)blockIdx.x * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const long long idx = base + i * RS_THREADS + tid;
    if (idx < n) atomicAdd(&hist[(int)((keys[idx] >> shift) & 255ull)], 1);
  }
  __syncthreads();
  blockHist[(size_t)tid * nblk + blockIdx.x] = hist[tid];
}

// pass 2: stable scatter.  blockOff = exclusive scan of blockHist (digit-major, block-minor): where the block's keys
// of each digit start.  Inside the block the order is (round, wave, lane) = index order: per round every wave ranks its
// lanes among the lanes with the same digit (8 ballots), one thread per digit turns the per-wave counts into offsets.
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(int n, const unsigned long long* __restrict__ keysIn,
                                                              const int* __restrict__ idxIn,
                                                              unsigned long long* __restrict__ keysOut, int* __restrict__ idxOut,
                                                              int shift, int nblk, const int* __restrict__ blockOff) {
  constexpr int NWV = RS_THREADS / smf::WAVE;
  __shared__ int wcnt[NWV][RS_RADIX];
  __shared__ int woff[NWV][RS_RADIX];
  __shared__ int run[RS_RADIX];
  const int tid = threadIdx.x, w = tid >> 6;
  run[tid] = blockOff[(size_t)tid * nblk + blockIdx.x];
  const long long base = (long long)blockIdx.x * RS_TILE;
  for (int i = 0; i < RS_ITEMS; ++i) {
#pragma unroll
    for (int q = 0; q < NWV; ++q) wcnt[q][tid] = 0;
    __syncthreads();
    const long long idx = base + i * RS_THREADS + tid;
    const bool valid = idx < n;
    unsigned long long key = 0ull;
    int payload = 0;
    if (valid) { key = keysIn[idx]; payload = idxIn[idx]; }
    const int d = (int)((key >> shift) & 255ull);
    unsigned long long peers = smf::ballot64(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long mb = smf::ballot64(valid && ((d >> b) & 1));
      peers &= ((d >> b) & 1) ? mb : ~mb;
    }
    const int rank = smf::mask_rank(peers);
    if (valid && rank == 0) wcnt[w][d] = __popcll(peers);
    __syncthreads();
    {
      int r = run[tid];
#pragma unroll
      for (int q = 0; q < NWV; ++q) { woff[q][tid] = r; r += wcnt[q][tid]; }
      run[tid] = r;
    }
    __syncthreads();
    if (valid) {
      const int dst = woff[w][d] + rank;
      keysOut[dst] = key;
      idxOut[dst] = payload;
    }
  }
}

__global__ void k_check_and_diag(int nnz, int rows, int cols, const int* __restrict__ ri, const int* __restrict__ ci,
                                 unsigned char* __restrict__ hasDiag, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const int r = ri[i], c = ci[i];
  if ((unsigned)r >= (unsigned)rows || (unsigned)c >= (unsigned)cols) { atomicOr(bad, 1); return; }
  if (hasDiag && r == c) hasDiag[r] = 1;
}

__global__ void k_missing_flags(int rows, const unsigned char* __restrict__ hasDiag, int* __restrict__ miss) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) miss[r] = hasDiag[r] ? 0 : 1;
}

// keys of the original entries and of the appended self loops: (row << colBits) | col, same order as row * cols + col
// without a 64-bit division on the way back.  The payload that travels with a key through the (stable) sort is the
// entry's VALUE: the emit step then reads keys and values in order instead of gathering values through an index.
__global__ void k_make_keys(int nnz, int rows, int colBits, const int* __restrict__ ri, const int* __restrict__ ci,
                            const float* __restrict__ val, const int* __restrict__ missPos, const int* __restrict__ miss,
                            unsigned long long* __restrict__ keys, int* __restrict__ payload) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nnz) {
    keys[i] = ((unsigned long long)(unsigned)ri[i] << colBits) | (unsigned)ci[i];
    payload[i] = __float_as_int(val[i]);
  }
  if (miss && i < rows && miss[i]) {            // self loop (i,i,1.0) appended behind the input, in row order
    const int p = nnz + missPos[i];
    keys[p] = ((unsigned long long)(unsigned)i << colBits) | (unsigned)i;
    payload[p] = __float_as_int(1.0f);
  }
}

__global__ void k_heads(int n, const unsigned long long* __restrict__ keys, int dedupe, int* __restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (!dedupe || i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one thread per output entry (= head of a run of equal keys): column and summed value (input order inside a run)
__global__ void k_emit(int n, int colBits, const unsigned long long* __restrict__ keys, const int* __restrict__ payload,
                       const int* __restrict__ head, const int* __restrict__ pos,
                       int dedupe, int useAbs, int* __restrict__ JA, float* __restrict__ A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const unsigned long long k = keys[i];
  float s = __int_as_float(payload[i]);
  if (dedupe)
    for (int j = i + 1; j < n && keys[j] == k; ++j) s += __int_as_float(payload[j]);
  const int o = pos[i];
  JA[o] = (int)(k & ((1ull << colBits) - 1ull));
  A[o] = useAbs ? fabsf(s) : s;
}

// rowPtr straight from the sorted keys: IA[r] = output position of the first key of row r or later (binary search, one
// thread per row; pos[n] = number of output entries).  No per-entry atomics, no scan, empty rows need no special case.
__global__ void k_row_starts(int rows, int n, int colBits, const unsigned long long* __restrict__ keys,
                             const int* __restrict__ pos, int* __restrict__ IA) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > rows) return;
  const unsigned long long want = (unsigned long long)(unsigned)r << colBits;
  int lo = 0, hi = n;                            // first index with keys[idx] >= want
  while (lo < hi) {
    const int mid = (int)(((long long)lo + hi) >> 1);
    if (keys[mid] < want) lo = mid + 1; else hi = mid;
  }
  IA[r] = pos[lo];
}

__global__ void k_row_normalise(int rows, const int* __restrict__ IA, float* __restrict__ A) {
  const int r = blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4), gl = threadIdx.x & 15;
  if (r >= rows) return;
  const int s = IA[r], e = IA[r + 1];
  const float w = (float)(1.0 / (double)(e - s));  // double division, narrowed (nlibs/CSR.cc:88-95)
  for (int p = s + gl; p < e; p += 16) A[p] = w;
}

}  // namespace coo

extern "C" int hip_coo_to_csr(spgemm_handle* h, int rows, int cols, int nnz, const int* dRow, const int* dCol,
                              const float* dVal, int flags, int** dIA, int** dJA, float** dA, int* nnzOut) {
  if (!dIA || !dJA || !dA || !nnzOut) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  *dIA = nullptr; *dJA = nullptr; *dA = nullptr; *nnzOut = 0;
  if (rows < 0 || cols < 0 || nnz < 0) return fail(SPGEMM_ERR_ARG, "negative size");
  if (nnz > 0 && (!dRow || !dCol || !dVal)) return fail(SPGEMM_ERR_ARG, "COO arrays null with nnz=%d", nnz);
  if ((flags & SPGEMM_COO_SELF_LOOPS) && rows != cols) return fail(SPGEMM_ERR_ARG, "self loops need a square matrix");
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  clear_stale_hip_error();
  hipStream_t s = h->stream;
  const int dedupe = (flags & SPGEMM_COO_DEDUPE) ? 1 : 0;
  const bool loops = (flags & SPGEMM_COO_SELF_LOOPS) != 0;
  const long long maxTotal = (long long)nnz + (loops ? rows : 0);
  if (maxTotal > 0x7fffffffLL) return fail(SPGEMM_ERR_OVERFLOW, "nnz does not fit int32");

  unsigned char* hasDiag = nullptr;
  int *bad = nullptr, *miss = nullptr, *missPos = nullptr, *idxA = nullptr, *idxB = nullptr, *head = nullptr, *pos = nullptr;
  unsigned long long *keyA = nullptr, *keyB = nullptr;
  void* tmp = nullptr;
  int* bhist = nullptr;                          // digit-major per-block histograms / offsets of the radix passes
  int *IA = nullptr, *JA = nullptr;
  float* A = nullptr;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)hasDiag, (void*)bad, (void*)miss, (void*)missPos, (void*)idxA, (void*)idxB, (void*)head,
                    (void*)pos, (void*)keyA, (void*)keyB, tmp, (void*)bhist})
      pool().release(p);
    if (rc != SPGEMM_OK) { pool().release(IA); pool().release(JA); pool().release(A); }
    return rc;
  };
#define COO_ALLOC(ptr, bytes) if (pool().alloc((void**)&(ptr), (bytes)) != hipSuccess) return cleanup(fail(SPGEMM_ERR_HIP, "device allocation failed"))
#define COO_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return cleanup(fail(SPGEMM_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_))); } while (0)
  const int T = 256;
  auto grid = [&](long long n) { return dim3((unsigned)std::max<long long>(1, (n + T - 1) / T)); };
  auto bits_of = [](int maxval) { return maxval > 0 ? 32 - __builtin_clz((unsigned)maxval) : 0; };
  const int colBits = std::max(1, bits_of(cols - 1));      // key = (row << colBits) | col
  const int keyBits = std::max(1, bits_of(rows - 1) + colBits);
  // scratch of the scans: tile sums of the longest array scanned below (entries, rows + 1, or 256 x radix blocks)
  const long long nblkMax = (maxTotal + coo::RS_TILE - 1) / coo::RS_TILE + 1;
  const long long longest = std::max<long long>(std::max<long long>(maxTotal, (long long)rows + 1), nblkMax * coo::RS_RADIX) + 1;
  unsigned long long* tile = nullptr;
  COO_ALLOC(tmp, sizeof(unsigned long long) * (size_t)((longest + SCAN_TILE - 1) / SCAN_TILE + 2));
  tile = (unsigned long long*)tmp;
  // in-place exclusive scan of data[0..cnt) with data[cnt] = total (k_scan_* of the SpGEMM path), 64-bit total on the host
  auto scan_inplace = [&](int* data, int cnt, unsigned long long* hostTotal) -> hipError_t {
    const int ntiles = std::max(1, cdiv(cnt, SCAN_TILE));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(SCAN_THREADS), 0, s, cnt, data, tile + 1);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, ntiles, tile + 1, tile);
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(SCAN_THREADS), 0, s, cnt, data, tile + 1, tile, 1);
    if (hostTotal) {
      hipError_t e = hipMemcpyAsync(hostTotal, tile, sizeof(unsigned long long), hipMemcpyDeviceToHost, s);
      if (e != hipSuccess) return e;
      return hipStreamSynchronize(s);
    }
    return hipGetLastError();
  };
  COO_ALLOC(bad, sizeof(int));
  COO_HIP(hipMemsetAsync(bad, 0, sizeof(int), s));
  // (1) validation, diagonal flags, self loops to append
  int extra = 0;
  if (loops && rows > 0) {
    COO_ALLOC(hasDiag, (size_t)rows);
    COO_ALLOC(miss, sizeof(int) * (size_t)rows);
    COO_ALLOC(missPos, sizeof(int) * ((size_t)rows + 1));
    COO_HIP(hipMemsetAsync(hasDiag, 0, (size_t)rows, s));
  }
  if (nnz > 0) hipLaunchKernelGGL(coo::k_check_and_diag, grid(nnz), dim3(T), 0, s, nnz, rows, cols, dRow, dCol, hasDiag, bad);
  if (loops && rows > 0) {
    hipLaunchKernelGGL(coo::k_missing_flags, grid(rows), dim3(T), 0, s, rows, hasDiag, miss);
    COO_HIP(hipMemcpyAsync(missPos, miss, sizeof(int) * (size_t)rows, hipMemcpyDeviceToDevice, s));
    unsigned long long nmiss = 0;
    COO_HIP(scan_inplace(missPos, rows, &nmiss));
    extra = (int)nmiss;
  }
  int hbad = 0;
  COO_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, s));
  COO_HIP(hipStreamSynchronize(s));
  if (hbad) return cleanup(fail(SPGEMM_ERR_INPUT, "COO entry outside the %d x %d matrix", rows, cols));
  const int total = nnz + extra;
  COO_ALLOC(IA, sizeof(int) * ((size_t)rows + 1));
  COO_HIP(hipMemsetAsync(IA, 0, sizeof(int) * ((size_t)rows + 1), s));
  if (total == 0) {
    COO_ALLOC(JA, sizeof(int));
    COO_ALLOC(A, sizeof(float));
    COO_HIP(hipStreamSynchronize(s));
    *dIA = IA; *dJA = JA; *dA = A; *nnzOut = 0;
    return cleanup(SPGEMM_OK);
  }
  // (2) keys + stable sort
  COO_ALLOC(keyA, sizeof(unsigned long long) * (size_t)total);
  COO_ALLOC(keyB, sizeof(unsigned long long) * (size_t)total);
  COO_ALLOC(idxA, sizeof(int) * (size_t)total);
  COO_ALLOC(idxB, sizeof(int) * (size_t)total);
  hipLaunchKernelGGL(coo::k_make_keys, grid(std::max(nnz, loops ? rows : 0)), dim3(T), 0, s, nnz, rows, colBits, dRow, dCol,
                     dVal, missPos, loops ? miss : (int*)nullptr, keyA, idxA);
  {
    const int nblk = cdiv(total, coo::RS_TILE);
    COO_ALLOC(bhist, sizeof(int) * ((size_t)nblk * coo::RS_RADIX + 1));
    for (int shift = 0; shift < keyBits; shift += 8) {     // stable LSD passes over exactly the bits the shape needs
      hipLaunchKernelGGL(coo::k_radix_hist, dim3(nblk), dim3(coo::RS_THREADS), 0, s, total, keyA, shift, nblk, bhist);
      COO_HIP(scan_inplace(bhist, nblk * coo::RS_RADIX, nullptr));
      hipLaunchKernelGGL(coo::k_radix_scatter, dim3(nblk), dim3(coo::RS_THREADS), 0, s, total, keyA, idxA, keyB, idxB, shift,
                         nblk, bhist);
      std::swap(keyA, keyB);
      std::swap(idxA, idxB);
    }
    std::swap(keyA, keyB);                                 // the sorted arrays are called keyB / idxB below
    std::swap(idxA, idxB);
  }
  // (3) heads of runs -> output positions
  COO_ALLOC(head, sizeof(int) * (size_t)total);
  COO_ALLOC(pos, sizeof(int) * ((size_t)total + 1));
  hipLaunchKernelGGL(coo::k_heads, grid(total), dim3(T), 0, s, total, keyB, dedupe, pos);
  COO_HIP(hipMemcpyAsync(head, pos, sizeof(int) * (size_t)total, hipMemcpyDeviceToDevice, s));
  unsigned long long nheads = 0;
  COO_HIP(scan_inplace(pos, total, &nheads));
  const int outN = (int)nheads;
  // (4) emit columns / values / row counts, scan the counts, optional row normalisation
  COO_ALLOC(JA, sizeof(int) * (size_t)std::max(outN, 1));
  COO_ALLOC(A, sizeof(float) * (size_t)std::max(outN, 1));
  hipLaunchKernelGGL(coo::k_emit, grid(total), dim3(T), 0, s, total, colBits, keyB, idxB, head, pos, dedupe,
                     (flags & SPGEMM_COO_ABS) ? 1 : 0, JA, A);
  hipLaunchKernelGGL(coo::k_row_starts, grid((long long)rows + 1), dim3(T), 0, s, rows, total, colBits, keyB, pos, IA);
  if ((flags & SPGEMM_COO_ROW_NORMALISE) && rows > 0)
    hipLaunchKernelGGL(coo::k_row_normalise, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, rows, IA, A);
  COO_HIP(hipGetLastError());
  COO_HIP(hipStreamSynchronize(s));
#undef COO_ALLOC
#undef COO_HIP
  *dIA = IA; *dJA = JA; *dA = A; *nnzOut = outN;
  return cleanup(SPGEMM_OK);
}

// ------------------------------------------------------------------------------------------------
// Workload statistics: the reference's 13-bucket power-of-two histogram of per-row flops
// (pushToStats / flopsStats, nlibs/tools/stats.cc:3-55): bucket i counts rows with 2^(i-1) < flops <= 2^i, bucket 0
// rows with flops <= 1, the last bucket everything above 2^11.  Row flops come from the path's own K1 kernel.
// ------------------------------------------------------------------------------------------------
namespace coo {
// val != nullptr: values themselves; otherwise differences of neighbours of `ptr` (row lengths of a CSR rowPtr)
__global__ void k_pow2_hist(int m, const int* __restrict__ val, const int* __restrict__ ptr, int nb,
                            int* __restrict__ hist /*[nb]*/) {
  __shared__ int sh[32];
  if (threadIdx.x < 32) sh[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) {
    const long long v = val ? (long long)val[i] : (long long)ptr[i + 1] - (long long)ptr[i];
    int b = nb - 1;
    for (int q = 0; q < nb - 1; ++q) if (v <= (1ll << q)) { b = q; break; }
    atomicAdd(&sh[b], 1);
  }
  __syncthreads();
  if (threadIdx.x < nb && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], sh[threadIdx.x]);
}
}  // namespace coo

extern "C" int hip_flopsStats(spgemm_handle* h, const int* dIA, const int* dJA, const int* dIB, int m,
                              int stats[SPGEMM_STATS_LEN]) {
  if (!stats) return fail(SPGEMM_ERR_ARG, "stats is null");
  for (int i = 0; i < SPGEMM_STATS_LEN; ++i) stats[i] = 0;
  if (m < 0 || !dIA) return fail(SPGEMM_ERR_ARG, "bad argument");
  if (m == 0) return SPGEMM_OK;
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  clear_stale_hip_error();
  int *flops = nullptr, *hist = nullptr;
  auto cleanup = [&](int rc) { pool().release(flops); pool().release(hist); return rc; };
  if (pool().alloc((void**)&flops, sizeof(int) * (size_t)m) != hipSuccess ||
      pool().alloc((void**)&hist, sizeof(int) * SPGEMM_STATS_LEN) != hipSuccess)
    return cleanup(fail(SPGEMM_ERR_HIP, "device allocation failed"));
  int rc = hip_csr_row_flops(h, dIA, dJA, dIB, m, flops, nullptr);
  if (rc) return cleanup(rc);
  if (hipMemsetAsync(hist, 0, sizeof(int) * SPGEMM_STATS_LEN, h->stream) != hipSuccess) return cleanup(fail(SPGEMM_ERR_HIP, "memset failed"));
  hipLaunchKernelGGL(coo::k_pow2_hist, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, m, flops, (const int*)nullptr,
                     SPGEMM_STATS_LEN, hist);
  if (hipMemcpyAsync(stats, hist, sizeof(int) * SPGEMM_STATS_LEN, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return cleanup(fail(SPGEMM_ERR_HIP, "flops statistics failed: %s", hipGetErrorString(hipGetLastError())));
  return cleanup(SPGEMM_OK);
}


// vector<int> CSR::nnzStats() (nlibs/CSR.cc:241-248): 18 power-of-two buckets of the ROW LENGTHS of a device CSR
extern "C" int hip_nnzStats(spgemm_handle* h, const int* dIA, int m, int stats[SPGEMM_NNZ_STATS_LEN]) {
  if (!stats) return fail(SPGEMM_ERR_ARG, "stats is null");
  for (int i = 0; i < SPGEMM_NNZ_STATS_LEN; ++i) stats[i] = 0;
  if (m < 0 || !dIA) return fail(SPGEMM_ERR_ARG, "bad argument");
  if (m == 0) return SPGEMM_OK;
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  clear_stale_hip_error();
  int* hist = nullptr;
  auto cleanup = [&](int rc) { pool().release(hist); return rc; };
  if (pool().alloc((void**)&hist, sizeof(int) * SPGEMM_NNZ_STATS_LEN) != hipSuccess) return cleanup(fail(SPGEMM_ERR_HIP, "device allocation failed"));
  if (hipMemsetAsync(hist, 0, sizeof(int) * SPGEMM_NNZ_STATS_LEN, h->stream) != hipSuccess) return cleanup(fail(SPGEMM_ERR_HIP, "memset failed"));
  hipLaunchKernelGGL(coo::k_pow2_hist, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, m, (const int*)nullptr, dIA,
                     SPGEMM_NNZ_STATS_LEN, hist);
  if (hipMemcpyAsync(stats, hist, sizeof(int) * SPGEMM_NNZ_STATS_LEN, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return cleanup(fail(SPGEMM_ERR_HIP, "row-length statistics failed: %s", hipGetErrorString(hipGetLastError())));
  return cleanup(SPGEMM_OK);
}

// ------------------------------------------------------------------------------------------------
// Per-bin correctness report: resultsComparison / isPartialRawEqual (mindex2-cuda/nGpuSpMM.cc:85-240) compare the GPU
// result hC with a CPU result rC bin by bin (rows of reference bin b sit at hqueue[hv[b]-1 .. hv[b+1]-1), flops.cu) so
// that a wrong kernel shows up as "its" bin.  Host arrays in (this is a checking aid, not a device path); rows are
// compared as sets of (column, value): row lengths, columns, values within `rel` relative (|x-y| <= rel*max(|x|,|y|)).
// ------------------------------------------------------------------------------------------------
extern "C" int hip_resultsComparison(int m, int n, const int* hIC, const int* hJC, const float* hC, const int* rIC,
                                     const int* rJC, const float* rC, const int hv[SPGEMM_HV_LEN], int hv_len,
                                     const int* hqueue, double rel, spgemm_bin_report report[SPGEMM_HV_LEN - 1]) {
  if (!hIC || !rIC || !hv || !hqueue || !report) return fail(SPGEMM_ERR_ARG, "null argument");
  if (m < 0 || n < 0 || hv_len < 2 || hv_len > SPGEMM_HV_LEN) return fail(SPGEMM_ERR_ARG, "bad size");
  std::vector<float> val((size_t)std::max(n, 1), 0.f);
  std::vector<int> stamp((size_t)std::max(n, 1), -1);
  for (int b = 0; b < SPGEMM_HV_LEN - 1; ++b) {
    spgemm_bin_report& R = report[b];
    R.rows = 0; R.rows_differ = 0; R.first_bad_row = -1; R.max_rel_err = 0.0;
    if (b + 1 >= hv_len) continue;
    const int lo = std::max(hv[b] - 1, 0), hi = hv[b + 1] - 1;      // element 0 of the (m+1)-long bin array is a dummy
    for (int q = lo; q < hi; ++q) {
      const int row = hqueue[q];
      if ((unsigned)row >= (unsigned)m) return fail(SPGEMM_ERR_INPUT, "queue entry %d is not a row", row);
      ++R.rows;
      bool bad = (hIC[row + 1] - hIC[row]) != (rIC[row + 1] - rIC[row]);
      for (int p = rIC[row]; p < rIC[row + 1]; ++p) {
        if ((unsigned)rJC[p] >= (unsigned)n) return fail(SPGEMM_ERR_INPUT, "column out of range in rC");
        stamp[rJC[p]] = row; val[rJC[p]] = rC[p];
      }
      for (int p = hIC[row]; p < hIC[row + 1]; ++p) {
        const int c = hJC[p];
        if ((unsigned)c >= (unsigned)n) return fail(SPGEMM_ERR_INPUT, "column out of range in hC");
        if (stamp[c] != row) { bad = true; continue; }
        const double x = hC[p], y = val[c];
        const double den = std::max(std::fabs(x), std::fabs(y));
        const double e = den > 0.0 ? std::fabs(x - y) / den : 0.0;
        if (e > R.max_rel_err) R.max_rel_err = e;
        if (e > rel) bad = true;
      }
      if (bad) { ++R.rows_differ; if (R.first_bad_row < 0) R.first_bad_row = row; }
    }
  }
  return SPGEMM_OK;
}
