// spgemm_device.hpp — hand-written HIP kernels for CSR x CSR SpGEMM on MI355X (gfx950, wave64).
//
// Pipeline:   flops+bin  ->  bin scan  ->  scatter rows  ->  symbolic per bin
//                            ->  exclusive scan of IC  ->  numeric per bin
// Reference counterparts (read as text only; nothing is translated line by line):
//   k_row_flops            gcomputeFlops + gcomputeBinId     mindex2-cuda/flops.cu:66-94
//                          dynamic_omp_CSR_flops             nlibs/flops_csr_kernel.cc:14-31
//   k_bin_scan/k_scatter   thrust::stable_sort_by_key + computeHistogram   flops.cu:96-107,131
//                          group_CSR_flops counting sort     nlibs/group_csr_kernel.cc:24-51
//   k_sym_*                sgpu_CSR_IC_nnzC_mid*             mindex2-cuda/tryOutBins.cuh:5-131
//                          gpu_CSR_IC_nnzC                   nlibs/gpus/gpu_csr_kernel.cu:44-82
//   k_num_small/k_num_hash sgpu_SpGEMM_mid / fp1 / fp2 / fpl4 mindex2-cuda/gspgemm.cuh:2-293
//                          hashCASAdd2                       mindex2-cuda/casHash.cuh:34-43
//   k_num_big              sgpu_SpGEMM_olarge (dense map)    "mindex2-cuda/\":143-213
//
// CDNA4 choices: 64-lane ballots / DPP scans instead of __syncthreads()-stepped sub-warp scans;
// products of one C row are flattened over all lanes (B rows of a power-law graph are short: a
// "lanes stride one B row" mapping leaves >90% of a wave idle); LDS tables sized per row; rows with
// more than 4096 products use an LDS column bitmap + popcount ranks (160 KB LDS per CU) instead
// of a global-memory dense map.  No MFMA: this is index/scatter work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smf {

constexpr int WAVE = 64;
constexpr int NBINS = 8;  // {0 | 1 | 2-4 | 5-16 | 17-64 | 65-512 | 513-4096 | >4096}
constexpr int EMPTY_KEY = -1;

__host__ __device__ __forceinline__ int bin_of(unsigned long long f) {
  if (f == 0) return 0;
  if (f == 1) return 1;
  if (f <= 4) return 2;
  if (f <= 16) return 3;
  if (f <= 64) return 4;
  if (f <= 512) return 5;
  if (f <= 4096) return 6;
  return 7;
}

// error flag bits written by kernels into Workspace::d_err
constexpr int ERRF_TABLE_FULL = 1;
constexpr int ERRF_COUNT_MISMATCH = 2;

// ------------------------------------------------------------------------------------------------
// wave64 primitives
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mask_rank(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

#define SMF_DPP(v, ctrl, rowmask, ident) __builtin_amdgcn_update_dpp((ident), (v), (ctrl), (rowmask), 0xf, false)

// inclusive +-scan over the 64 lanes; all lanes must be active
__device__ __forceinline__ int wave_incl_add(int v) {
  v += SMF_DPP(v, 0x111, 0xf, 0);  // row_shr:1
  v += SMF_DPP(v, 0x112, 0xf, 0);  // row_shr:2
  v += SMF_DPP(v, 0x114, 0xf, 0);  // row_shr:4
  v += SMF_DPP(v, 0x118, 0xf, 0);  // row_shr:8
  v += SMF_DPP(v, 0x142, 0xa, 0);  // row_bcast:15 -> rows 1,3
  v += SMF_DPP(v, 0x143, 0xc, 0);  // row_bcast:31 -> rows 2,3
  return v;
}

// inclusive max-scan of non-negative ints over the 64 lanes
__device__ __forceinline__ int wave_incl_max(int v) {
  v = max(v, SMF_DPP(v, 0x111, 0xf, 0));
  v = max(v, SMF_DPP(v, 0x112, 0xf, 0));
  v = max(v, SMF_DPP(v, 0x114, 0xf, 0));
  v = max(v, SMF_DPP(v, 0x118, 0xf, 0));
  v = max(v, SMF_DPP(v, 0x142, 0xa, 0));
  v = max(v, SMF_DPP(v, 0x143, 0xc, 0));
  return v;
}

__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_add(v), 63); }

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// LDS traffic between lanes of ONE wave: the LDS executes a wave's DS ops in order, so only the
// compiler must be kept from reordering.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int lds_load(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int next_pow2_clamped(int x, int lo, int hi) {
  int p = lo;
  while (p < x && p < hi) p <<= 1;
  return p;
}

// ------------------------------------------------------------------------------------------------
// LDS open-addressing hash (keys >= 0, EMPTY_KEY = -1), multiplicative hash, linear probing.
// Returns the slot of `c`; *is_new is set when this call claimed the slot.  `size` is a power of
// two >= 2 * (number of distinct keys), so a probe sequence always terminates; a bounded loop and an
// error flag guard against corrupt inputs instead of hanging the GPU.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int hash_insert(int* keys, int size, int shift, int c, bool* is_new, int* err) {
  unsigned h = ((unsigned)c * 2654435761u) >> shift;
  const unsigned mask = (unsigned)size - 1u;
  *is_new = false;
  for (int probe = 0; probe < size; ++probe) {
    int cur = lds_load(&keys[h]);
    if (cur == c) return (int)h;
    if (cur == EMPTY_KEY) {
      int old = atomicCAS(&keys[h], EMPTY_KEY, c);
      if (old == EMPTY_KEY) { *is_new = true; return (int)h; }
      if (old == c) return (int)h;
    }
    h = (h + 1u) & mask;
  }
  atomicOr(err, ERRF_TABLE_FULL);
  return 0;
}

__device__ __forceinline__ int log2_pow2(int p) { return 31 - __clz(p); }

// ------------------------------------------------------------------------------------------------
// K1  per-row product count + bin id + per-block bin histogram            (mindex2: gcomputeFlops)
// One wave owns 64 consecutive rows and walks their A entries flattened over the lanes, so a row with
// 4095 entries costs the same per entry as a row with 2.  256 threads = 4 waves = 256 rows / block.
// ------------------------------------------------------------------------------------------------
constexpr int K1_THREADS = 256;

__global__ __launch_bounds__(K1_THREADS) void k_row_flops(
    int m, const int* __restrict__ IA, const int* __restrict__ JA, const int* __restrict__ IB,
    int* __restrict__ rowFlops, unsigned char* __restrict__ binId, int* __restrict__ blockHist,
    unsigned long long* __restrict__ totalP, int* __restrict__ IC) {
  __shared__ unsigned long long acc[K1_THREADS];
  __shared__ int hist[NBINS];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  if (tid < NBINS) hist[tid] = 0;
  acc[tid] = 0;
  __syncthreads();
  const int r0 = (blockIdx.x * (K1_THREADS / WAVE) + w) * WAVE;
  const int r = r0 + lane;
  const int rs = IA[min(r, m)];                      // start of my row (IA[m] past the end)
  const int base = __builtin_amdgcn_readfirstlane(rs);
  const int endAll = IA[min(r0 + WAVE, m)];
  unsigned long long* wacc = acc + w * WAVE;
  // uniform trip count: the lane shuffles below need every lane of the wave alive
  const int rounds = (endAll - base + WAVE - 1) / WAVE;
  for (int it = 0; it < rounds; ++it) {
    const int idx = base + it * WAVE + lane;
    const bool valid = idx < endAll;
    int len = 0;
    if (valid) { const int j = JA[idx]; len = IB[j + 1] - IB[j]; }
    // local row = largest l with start_l <= idx  (binary search over the lanes' rs values)
    int lo = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) {
      const int cand = lo + step;
      const int s = __shfl(rs, cand & 63, 64);
      if (cand < WAVE && s <= idx) lo = cand;
    }
    if (valid && len) atomicAdd(&wacc[lo], (unsigned long long)len);
  }
  __syncthreads();
  unsigned long long f = 0;
  int b = -1;
  if (r < m) {
    f = acc[tid];
    rowFlops[r] = f > 0x7fffffffULL ? 0x7fffffff : (int)f;
    b = bin_of(f);
    binId[r] = (unsigned char)b;
    if (b <= 1) IC[r] = b;                           // 0 products -> 0 entries, 1 product -> 1 entry
    atomicAdd(&hist[b], 1);
  }
  const unsigned long long wsum = wave_sum_u64(f);
  if (lane == 0 && wsum) atomicAdd(totalP, wsum);
  __syncthreads();
  if (tid < NBINS) blockHist[blockIdx.x * NBINS + tid] = hist[tid];
}

// ------------------------------------------------------------------------------------------------
// K2  exclusive scan of the per-block histograms in (bin-major, block-minor) order -> where each
//     block writes its rows of each bin; binPtr[NBINS+1].  One 1024-thread block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_bin_scan(int nblk, const int* __restrict__ blockHist,
                                                    int* __restrict__ blockOff, int* __restrict__ binPtr) {
  __shared__ int wsum[16];
  __shared__ int running_s;
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  if (tid == 0) { running_s = 0; binPtr[0] = 0; }
  __syncthreads();
  for (int b = 0; b < NBINS; ++b) {
    for (int t0 = 0; t0 < nblk; t0 += 1024) {
      const int blk = t0 + tid;
      const int v = blk < nblk ? blockHist[blk * NBINS + b] : 0;
      const int incl = wave_incl_add(v);
      if (lane == 63) wsum[w] = incl;
      __syncthreads();
      int woff = 0, tot = 0;
      for (int i = 0; i < 16; ++i) { const int s = wsum[i]; tot += s; if (i < w) woff += s; }
      const int run = running_s;
      if (blk < nblk) blockOff[blk * NBINS + b] = run + woff + incl - v;
      __syncthreads();
      if (tid == 0) running_s = run + tot;
      __syncthreads();
    }
    if (tid == 0) binPtr[b + 1] = running_s;
  }
}

// ------------------------------------------------------------------------------------------------
// K3  stable scatter of row ids into their bins (rows ascend inside a bin).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(K1_THREADS) void k_scatter_rows(int m, const unsigned char* __restrict__ binId,
                                                             const int* __restrict__ blockOff,
                                                             int* __restrict__ rowIds) {
  __shared__ int wcnt[K1_THREADS / WAVE][NBINS];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int r = blockIdx.x * K1_THREADS + tid;
  const int b = r < m ? (int)binId[r] : -1;
  int myrank = 0;
#pragma unroll
  for (int q = 0; q < NBINS; ++q) {
    const unsigned long long mk = __ballot(b == q);
    if (b == q) myrank = mask_rank(mk);
    if (lane == 0) wcnt[w][q] = __popcll(mk);
  }
  __syncthreads();
  if (b >= 0) {
    int off = blockOff[blockIdx.x * NBINS + b];
    for (int i = 0; i < w; ++i) off += wcnt[i][b];
    rowIds[off + myrank] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// Small rows (<= 64 products): a group of G lanes per row, A entries one after the other, the G
// lanes stride the (short, bounded by the bin) B row.  Tables live in LDS, one per group, sized per
// row to the next power of two >= 2*flops.
// ------------------------------------------------------------------------------------------------
template <int G, int TBL>
__global__ __launch_bounds__(256) void k_sym_small(const int* __restrict__ binPtr, int binLo, int binHi,
                                                    const int* __restrict__ rowIds,
                                                    const int* __restrict__ IA, const int* __restrict__ JA,
                                                    const int* __restrict__ IB, const int* __restrict__ JB,
                                                    const int* __restrict__ rowFlops, int* __restrict__ IC,
                                                    int* __restrict__ err) {
  constexpr int GROUPS = 256 / G;
  __shared__ int keys[GROUPS][TBL];
  const int tid = threadIdx.x, g = tid / G, gl = tid % G;
  const int first = binPtr[binLo], count = binPtr[binHi] - first;
  const int iters = (count + GROUPS - 1) / GROUPS;   // uniform trip count per block
  for (int it = blockIdx.x; it < iters; it += gridDim.x) {
    const int q = it * GROUPS + g;
    const bool live = q < count;
    const int row = live ? rowIds[first + q] : 0;
    const int F = live ? rowFlops[row] : 1;
    const int size = next_pow2_clamped(2 * F, 8, TBL);
    const int shift = 32 - log2_pow2(size);
    for (int i = gl; i < size; i += G) keys[g][i] = EMPTY_KEY;
    wave_lds_sync();
    int mine = 0;
    if (live) {
      const int as = IA[row], ae = IA[row + 1];
      for (int ap = as; ap < ae; ++ap) {
        const int j = JA[ap];
        const int bs = IB[j], be = IB[j + 1];
        for (int bp = bs + gl; bp < be; bp += G) {
          bool isnew;
          hash_insert(keys[g], size, shift, JB[bp], &isnew, err);
          mine += isnew ? 1 : 0;
        }
      }
    }
    // group sum (all lanes are back in lock step here)
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (live && gl == 0) IC[row] = mine;
    wave_lds_sync();
  }
}

template <int G, int TBL>
__global__ __launch_bounds__(256) void k_num_small(const int* __restrict__ binPtr, int binLo, int binHi,
                                                    const int* __restrict__ rowIds,
                                                    const int* __restrict__ IA, const int* __restrict__ JA,
                                                    const float* __restrict__ VA,
                                                    const int* __restrict__ IB, const int* __restrict__ JB,
                                                    const float* __restrict__ VB,
                                                    const int* __restrict__ rowFlops,
                                                    const int* __restrict__ IC, int* __restrict__ JC,
                                                    float* __restrict__ C, int* __restrict__ err) {
  constexpr int GROUPS = 256 / G;
  __shared__ int keys[GROUPS][TBL];
  __shared__ float vals[GROUPS][TBL];
  const int tid = threadIdx.x, g = tid / G, gl = tid % G;
  const int first = binPtr[binLo], count = binPtr[binHi] - first;
  const int iters = (count + GROUPS - 1) / GROUPS;
  for (int it = blockIdx.x; it < iters; it += gridDim.x) {
    const int q = it * GROUPS + g;
    const bool live = q < count;
    const int row = live ? rowIds[first + q] : 0;
    const int F = live ? rowFlops[row] : 1;
    const int size = next_pow2_clamped(2 * F, 8, TBL);
    const int shift = 32 - log2_pow2(size);
    for (int i = gl; i < size; i += G) { keys[g][i] = EMPTY_KEY; vals[g][i] = 0.f; }
    wave_lds_sync();
    if (live) {
      const int as = IA[row], ae = IA[row + 1];
      for (int ap = as; ap < ae; ++ap) {
        const int j = JA[ap];
        const float a = VA[ap];
        const int bs = IB[j], be = IB[j + 1];
        for (int bp = bs + gl; bp < be; bp += G) {
          bool isnew;
          const int s = hash_insert(keys[g], size, shift, JB[bp], &isnew, err);
          atomicAdd(&vals[g][s], a * VB[bp]);
        }
      }
    }
    wave_lds_sync();
    // compaction: the G lanes sweep the table; occupied slots are packed in slot order
    const int off = live ? IC[row] : 0;
    const int want = live ? IC[row + 1] - off : 0;
    int written = 0;
    for (int i0 = 0; i0 < size; i0 += G) {
      const int i = i0 + gl;
      const int kx = keys[g][i];
      const bool occ = live && kx != EMPTY_KEY;
      const unsigned long long mk = __ballot(occ);
      const int shiftg = lane_id() - gl;
      const unsigned long long gm = (G == 64) ? mk : ((mk >> shiftg) & ((1ull << G) - 1ull));
      const int rank = __popcll(gm & ((1ull << gl) - 1ull));
      if (occ) { JC[off + written + rank] = kx; C[off + written + rank] = vals[g][i]; }
      written += __popcll(gm);
    }
    if (live && gl == 0 && written != want) atomicOr(err, ERRF_COUNT_MISMATCH);
    wave_lds_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// Flattened product walk of ONE C row by a block of NW waves.
// A entries are staged in LDS in chunks of 64*NW (one per thread) together with the inclusive scan
// of their B-row lengths; then every wave takes rounds of 64 consecutive products.  For a round the
// wave finds the first owning A entry with two 64-ary ballot searches, lets the <=64 entries that start
// inside the round mark their first product, and a DPP max-scan spreads the owner to the products
// that follow.  f(active, jbIndex, aValue) is called in wave-uniform control flow.
// ------------------------------------------------------------------------------------------------
template <int NW>
struct RowStage {
  int incl[WAVE * NW];     // inclusive scan of B-row lengths of the staged A entries
  int off[WAVE * NW];      // IB[j] - exclusive scan: product p of the chunk lives at JB[off + p]
  float aval[WAVE * NW];
  int marks[NW][WAVE];
  int wsum[NW];
};

template <int NW, bool NEED_VAL, class F>
__device__ __forceinline__ void for_each_product(RowStage<NW>& st, int as, int ae,
                                                 const int* __restrict__ JA, const float* __restrict__ VA,
                                                 const int* __restrict__ IB, F&& f) {
  constexpr int K = WAVE * NW;
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  st.marks[w][lane] = 0;
  for (int chunk = as; chunk < ae; chunk += K) {
    // ---- stage
    const int ap = chunk + tid;
    int len = 0, bs = 0;
    float a = 0.f;
    if (ap < ae) {
      const int j = JA[ap];
      bs = IB[j];
      len = IB[j + 1] - bs;
      if (NEED_VAL) a = VA[ap];
    }
    int incl = wave_incl_add(len);
    if (NW > 1) {
      if (lane == 63) st.wsum[w] = incl;
      __syncthreads();
      int woff = 0;
      for (int i = 0; i < w; ++i) woff += st.wsum[i];
      incl += woff;
    }
    st.incl[tid] = incl;
    st.off[tid] = bs - (incl - len);
    if (NEED_VAL) st.aval[tid] = a;
    __syncthreads();
    const int T = st.incl[K - 1];
    const int nk = min(K, ae - chunk);
    // ---- rounds
    for (int base = w * WAVE; base < T; base += K) {
      const int roundEnd = min(base + WAVE, T);
      int grp = 0;
      if (NW > 1) {
        const int v1 = lane < NW ? st.incl[lane * WAVE + 63] : 0x7fffffff;
        grp = __popcll(__ballot(v1 <= base));
      }
      const int v2 = st.incl[grp * WAVE + lane];
      const int k0 = grp * WAVE + __popcll(__ballot(v2 <= base));   // first entry with incl > base
      for (int eb = k0;; eb += WAVE) {
        const int e = eb + lane;
        if (e < nk) {
          const int s = e == 0 ? 0 : st.incl[e - 1];
          const int en = st.incl[e];
          if (en > s && s < roundEnd && en > base) st.marks[w][max(s, base) - base] = e - k0;
        }
        const int lastc = min(eb + WAVE - 1, nk - 1);
        if (st.incl[lastc] >= roundEnd) break;
      }
      wave_lds_sync();
      const int mv = st.marks[w][lane];
      st.marks[w][lane] = 0;
      const int owner = k0 + wave_incl_max(mv);
      const int p = base + lane;
      const bool active = p < T;
      const int oi = active ? owner : k0;
      const int jb = st.off[oi] + p;
      const float av = NEED_VAL ? st.aval[oi] : 0.f;
      f(active, jb, av);
      wave_lds_sync();
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Medium rows (65..4096 products): one block of NW waves per row, LDS key table (symbolic) or
// key+value table with an insertion-ordered slot list (numeric).
// ------------------------------------------------------------------------------------------------
template <int NW, int TBL>
__global__ __launch_bounds__(WAVE * NW) void k_sym_hash(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int* __restrict__ JA,
                                                         const int* __restrict__ IB, const int* __restrict__ JB,
                                                         const int* __restrict__ rowFlops, int* __restrict__ IC,
                                                         int* __restrict__ err) {
  __shared__ int keys[TBL];
  __shared__ RowStage<NW> st;
  __shared__ int cnt_s;
  const int tid = threadIdx.x, lane = lane_id();
  const int first = binPtr[bin], count = binPtr[bin + 1] - first;
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const int row = rowIds[first + q];
    const int size = next_pow2_clamped(2 * rowFlops[row], 64, TBL);
    const int shift = 32 - log2_pow2(size);
    for (int i = tid; i < size; i += WAVE * NW) keys[i] = EMPTY_KEY;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    int mine = 0;
    for_each_product<NW, false>(st, IA[row], IA[row + 1], JA, nullptr, IB, [&](bool active, int jb, float) {
      if (active) {
        bool isnew;
        hash_insert(keys, size, shift, JB[jb], &isnew, err);
        mine += isnew ? 1 : 0;
      }
    });
    const int ws = wave_sum(mine);
    if (lane == 0 && ws) atomicAdd(&cnt_s, ws);
    __syncthreads();
    if (tid == 0) IC[row] = cnt_s;
    __syncthreads();
  }
}

template <int NW, int TBL>
__global__ __launch_bounds__(WAVE * NW) void k_num_hash(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int* __restrict__ JA,
                                                         const float* __restrict__ VA,
                                                         const int* __restrict__ IB, const int* __restrict__ JB,
                                                         const float* __restrict__ VB,
                                                         const int* __restrict__ IC, int* __restrict__ JC,
                                                         float* __restrict__ C, int* __restrict__ err) {
  __shared__ int keys[TBL];
  __shared__ float vals[TBL];
  __shared__ unsigned short slots[TBL / 2];   // slot of the i-th distinct column, in claim order
  __shared__ RowStage<NW> st;
  __shared__ int cnt_s;
  const int tid = threadIdx.x, lane = lane_id();
  const int first = binPtr[bin], count = binPtr[bin + 1] - first;
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const int row = rowIds[first + q];
    const int off = IC[row];
    const int want = IC[row + 1] - off;                 // exact distinct count from the symbolic pass
    const int size = next_pow2_clamped(2 * want, 64, TBL);
    const int shift = 32 - log2_pow2(size);
    for (int i = tid; i < size; i += WAVE * NW) { keys[i] = EMPTY_KEY; vals[i] = 0.f; }
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    for_each_product<NW, true>(st, IA[row], IA[row + 1], JA, VA, IB, [&](bool active, int jb, float a) {
      bool isnew = false;
      int s = 0;
      if (active) {
        s = hash_insert(keys, size, shift, JB[jb], &isnew, err);
        atomicAdd(&vals[s], a * VB[jb]);
      }
      const unsigned long long nm = __ballot(isnew);
      if (nm) {
        int basei = 0;
        if (lane == 0) basei = atomicAdd(&cnt_s, __popcll(nm));
        basei = __builtin_amdgcn_readfirstlane(basei);
        if (isnew) slots[min(basei + mask_rank(nm), TBL / 2 - 1)] = (unsigned short)s;
      }
    });
    __syncthreads();
    const int n = cnt_s;
    if (tid == 0 && n != want) atomicOr(err, ERRF_COUNT_MISMATCH);
    for (int i = tid; i < min(n, want); i += WAVE * NW) {
      const int s = slots[i];
      JC[off + i] = keys[s];
      C[off + i] = vals[s];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Big rows (> 4096 products): one 1024-thread block per row, column window of BIG_WC columns held
// as an LDS bitmap.  Symbolic = popcount.  Numeric = popcount ranks give every column its final
// position, values accumulate in an LDS float array addressed by rank (no probing, sorted output);
// rows with more distinct columns than BIG_CAP take several rank passes, matrices with more than
// BIG_WC columns several column windows.
// ------------------------------------------------------------------------------------------------
constexpr int BIG_NW = 16;
constexpr int BIG_THREADS = BIG_NW * WAVE;
constexpr int BIG_WC = 262144;                 // columns per window (32 KB bitmap)
constexpr int BIG_WORDS = BIG_WC / 32;         // 8192
constexpr int BIG_WPT = BIG_WORDS / BIG_THREADS;  // 8 words per thread
constexpr int BIG_CAP = 18432;                 // float accumulators per rank pass (72 KB)

struct BigSymShared {
  unsigned bitmap[BIG_WORDS];
  RowStage<BIG_NW> st;
  int red[BIG_NW];
};

struct BigNumShared {
  unsigned bitmap[BIG_WORDS];
  int prefix[BIG_WORDS];
  float acc[BIG_CAP];
  RowStage<BIG_NW> st;
  int red[BIG_NW];
};

__device__ __forceinline__ int block_sum_16(int v, int* red) {
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const int ws = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = ws;
  __syncthreads();
  int tot = 0;
  for (int i = 0; i < BIG_NW; ++i) tot += red[i];
  return tot;
}

__global__ __launch_bounds__(BIG_THREADS) void k_sym_big(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int* __restrict__ JA,
                                                         const int* __restrict__ IB, const int* __restrict__ JB,
                                                         int n, int* __restrict__ IC) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BigSymShared& sh = *reinterpret_cast<BigSymShared*>(smem_raw);
  const int tid = threadIdx.x;
  const int first = binPtr[bin], count = binPtr[bin + 1] - first;
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const int row = rowIds[first + q];
    const int as = IA[row], ae = IA[row + 1];
    int total = 0;
    for (int w0 = 0; w0 < n; w0 += BIG_WC) {
      const int wc = min(BIG_WC, n - w0);
      const int words = (wc + 31) >> 5;
      for (int i = tid; i < words; i += BIG_THREADS) sh.bitmap[i] = 0u;
      __syncthreads();
      for_each_product<BIG_NW, false>(sh.st, as, ae, JA, nullptr, IB, [&](bool active, int jb, float) {
        if (active) {
          const int c = JB[jb] - w0;
          if ((unsigned)c < (unsigned)wc) atomicOr(&sh.bitmap[c >> 5], 1u << (c & 31));
        }
      });
      int mine = 0;
      for (int i = tid; i < words; i += BIG_THREADS) mine += __popc(sh.bitmap[i]);
      total += block_sum_16(mine, sh.red);
      __syncthreads();
    }
    if (tid == 0) IC[row] = total;
  }
}

__global__ __launch_bounds__(BIG_THREADS) void k_num_big(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int* __restrict__ JA,
                                                         const float* __restrict__ VA,
                                                         const int* __restrict__ IB, const int* __restrict__ JB,
                                                         const float* __restrict__ VB, int n,
                                                         const int* __restrict__ IC, int* __restrict__ JC,
                                                         float* __restrict__ C, int* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BigNumShared& sh = *reinterpret_cast<BigNumShared*>(smem_raw);
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int first = binPtr[bin], count = binPtr[bin + 1] - first;
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const int row = rowIds[first + q];
    const int as = IA[row], ae = IA[row + 1];
    int outBase = IC[row];
    const int outEnd = IC[row + 1];
    for (int w0 = 0; w0 < n; w0 += BIG_WC) {
      const int wc = min(BIG_WC, n - w0);
      for (int i = tid; i < BIG_WORDS; i += BIG_THREADS) sh.bitmap[i] = 0u;
      __syncthreads();
      // pass 1: which columns of this window occur
      for_each_product<BIG_NW, false>(sh.st, as, ae, JA, nullptr, IB, [&](bool active, int jb, float) {
        if (active) {
          const int c = JB[jb] - w0;
          if ((unsigned)c < (unsigned)wc) atomicOr(&sh.bitmap[c >> 5], 1u << (c & 31));
        }
      });
      // exclusive popcount prefix over the words: thread t owns words [t*WPT, t*WPT+WPT)
      int loc[BIG_WPT];
      int mine = 0;
#pragma unroll
      for (int i = 0; i < BIG_WPT; ++i) { loc[i] = mine; mine += __popc(sh.bitmap[tid * BIG_WPT + i]); }
      const int incl = wave_incl_add(mine);
      __syncthreads();
      if (lane == 63) sh.red[w] = incl;
      __syncthreads();
      int woff = 0, cntw = 0;
      for (int i = 0; i < BIG_NW; ++i) { const int s = sh.red[i]; cntw += s; if (i < w) woff += s; }
      const int texcl = woff + incl - mine;
#pragma unroll
      for (int i = 0; i < BIG_WPT; ++i) sh.prefix[tid * BIG_WPT + i] = texcl + loc[i];
      __syncthreads();
      if (outBase + cntw > outEnd) { if (tid == 0) atomicOr(err, ERRF_COUNT_MISMATCH); cntw = max(0, outEnd - outBase); }
      // column indices, already sorted
#pragma unroll
      for (int i = 0; i < BIG_WPT; ++i) {
        unsigned bits = sh.bitmap[tid * BIG_WPT + i];
        int pos = texcl + loc[i];
        const int cbase = w0 + (tid * BIG_WPT + i) * 32;
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          if (pos < cntw) JC[outBase + pos] = cbase + b;
          ++pos;
        }
      }
      // pass 2..: accumulate values by rank, BIG_CAP ranks at a time
      for (int lo = 0; lo < cntw; lo += BIG_CAP) {
        const int span = min(BIG_CAP, cntw - lo);
        for (int i = tid; i < span; i += BIG_THREADS) sh.acc[i] = 0.f;
        __syncthreads();
        for_each_product<BIG_NW, true>(sh.st, as, ae, JA, VA, IB, [&](bool active, int jb, float a) {
          if (active) {
            const int c = JB[jb] - w0;
            if ((unsigned)c < (unsigned)wc) {
              const int wi = c >> 5;
              const int rk = sh.prefix[wi] + __popc(sh.bitmap[wi] & ((1u << (c & 31)) - 1u)) - lo;
              if ((unsigned)rk < (unsigned)span) atomicAdd(&sh.acc[rk], a * VB[jb]);
            }
          }
        });
        for (int i = tid; i < span; i += BIG_THREADS) C[outBase + lo + i] = sh.acc[i];
        __syncthreads();
      }
      outBase += cntw;
    }
    if (tid == 0 && outBase != outEnd) atomicOr(err, ERRF_COUNT_MISMATCH);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan of IC[0..m) -> offsets, IC[m] = nnzC (64-bit total kept for the overflow check).
// Three small launches: per-block sums, one block scans the sums, per-block rescan with offset.
// (thrust::exclusive_scan in the reference: nlibs/gpus/gpu_csr_kernel.cu:149-150)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_ITEMS = 4;                  // ints per thread
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tile_sums(int m, const int* __restrict__ IC,
                                                                  unsigned long long* __restrict__ tileSum) {
  __shared__ unsigned long long red[16];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int base = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
  unsigned long long s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < m) s += (unsigned)IC[base + i];
  s = wave_sum_u64(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (tid == 0) { unsigned long long t = 0; for (int i = 0; i < 16; ++i) t += red[i]; tileSum[blockIdx.x] = t; }
}

__global__ __launch_bounds__(1024) void k_scan_tiles(int ntiles, unsigned long long* __restrict__ tileSum,
                                                      unsigned long long* __restrict__ total) {
  // serial over tiles in chunks of 1024 (ntiles is m/4096: small)
  __shared__ unsigned long long wsum[16];
  __shared__ unsigned long long running;
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  if (tid == 0) running = 0;
  __syncthreads();
  for (int t0 = 0; t0 < ntiles; t0 += 1024) {
    const int t = t0 + tid;
    const unsigned long long v = t < ntiles ? tileSum[t] : 0ull;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    unsigned long long woff = 0, tot = 0;
    for (int i = 0; i < 16; ++i) { const unsigned long long s = wsum[i]; tot += s; if (i < w) woff += s; }
    const unsigned long long run = running;
    if (t < ntiles) tileSum[t] = run + woff + incl - v;    // exclusive
    __syncthreads();
    if (tid == 0) running = run + tot;
    __syncthreads();
  }
  if (tid == 0) *total = running;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(int m, int* __restrict__ IC,
                                                              const unsigned long long* __restrict__ tileOff,
                                                              const unsigned long long* __restrict__ total) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int base = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { v[i] = base + i < m ? IC[base + i] : 0; s += v[i]; }
  const int incl = wave_incl_add(s);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int woff = 0;
  for (int i = 0; i < w; ++i) woff += wsum[i];
  int run = (int)tileOff[blockIdx.x] + woff + incl - s;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) { if (base + i < m) IC[base + i] = run; run += v[i]; }
  if (blockIdx.x == 0 && tid == 0) {
    const unsigned long long t = *total;
    IC[m] = t > 0x7fffffffULL ? 0x7fffffff : (int)t;
  }
}

// ------------------------------------------------------------------------------------------------
// helpers for the reference-shaped classify outputs and for canonical ordering
// ------------------------------------------------------------------------------------------------
// gathered[q] = rowFlops[rowIds[q]]  (then scanned with the kernels above into dflops[1..m])
__global__ void k_gather_flops(int m, const int* __restrict__ rowIds, const int* __restrict__ rowFlops,
                               int* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < m) out[q] = rowFlops[rowIds[q]];
}

// saturating inclusive scan companion: dst[i+1] = min(INT_MAX, excl[i] + v[i]) is produced on the host side
// by scanning in 64-bit; see spgemm_hip.hip (classify is an API-only path, not part of the hot loop).

// CSR::makeOrdered on the device (nlibs/CSR.cc:73-86): one block per row, bitonic sort in LDS for rows
// of <= SORT_MAX entries.  Longer rows only come out of k_num_big, which emits them sorted already: they
// are checked and, if some caller hands in a long unsorted row, sorted by odd-even transposition in
// global memory (slow, correct).  Used by tests/drivers, not by the timed path.
constexpr int SORT_MAX = 4096;
__global__ __launch_bounds__(256) void k_sort_rows(int m, const int* __restrict__ IC, int* __restrict__ JC,
                                                    float* __restrict__ C) {
  __shared__ int sk[SORT_MAX];
  __shared__ float sv[SORT_MAX];
  __shared__ int unsorted;
  const int tid = threadIdx.x;
  for (int row = blockIdx.x; row < m; row += gridDim.x) {
    const int s = IC[row], len = IC[row + 1] - s;
    if (len < 2) continue;
    if (len <= SORT_MAX) {
      int p2 = 1;
      while (p2 < len) p2 <<= 1;
      for (int i = tid; i < p2; i += 256) { sk[i] = i < len ? JC[s + i] : 0x7fffffff; sv[i] = i < len ? C[s + i] : 0.f; }
      __syncthreads();
      for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = tid; i < p2; i += 256) {
            const int ixj = i ^ j;
            if (ixj > i) {
              const bool up = (i & k) == 0;
              const int a = sk[i], b = sk[ixj];
              if ((a > b) == up) { sk[i] = b; sk[ixj] = a; const float t = sv[i]; sv[i] = sv[ixj]; sv[ixj] = t; }
            }
          }
          __syncthreads();
        }
      }
      for (int i = tid; i < len; i += 256) { JC[s + i] = sk[i]; C[s + i] = sv[i]; }
      __syncthreads();
    } else {
      if (tid == 0) unsorted = 0;
      __syncthreads();
      int bad = 0;
      for (int i = tid; i + 1 < len; i += 256) bad |= JC[s + i] > JC[s + i + 1];
      if (bad) atomicOr(&unsorted, 1);
      __syncthreads();
      const bool need = unsorted != 0;
      __syncthreads();
      if (!need) continue;
      for (int phase = 0; phase < len; ++phase) {
        for (int i = (phase & 1) + 2 * tid; i + 1 < len; i += 512) {
          const int a = JC[s + i], b = JC[s + i + 1];
          if (a > b) { JC[s + i] = b; JC[s + i + 1] = a; const float t = C[s + i]; C[s + i] = C[s + i + 1]; C[s + i + 1] = t; }
        }
        __threadfence();
        __syncthreads();
      }
    }
  }
}

// self-test of the DPP scans / mask ranks against serial results computed by lane 0
__global__ void k_selftest(const int* __restrict__ in, int* __restrict__ bad) {
  __shared__ int buf[WAVE];
  const int lane = lane_id();
  const int v = in[blockIdx.x * WAVE + lane];
  buf[lane] = v;
  const int a = wave_incl_add(v);
  const int mx = wave_incl_max(v & 0xffff);
  const unsigned long long mk = __ballot(v & 1);
  const int rk = mask_rank(mk);
  __syncthreads();
  int ea = 0, em = 0, er = 0;
  for (int i = 0; i <= lane; ++i) { ea += buf[i]; em = max(em, buf[i] & 0xffff); if (i < lane) er += buf[i] & 1; }
  if (a != ea || mx != em || rk != er || wave_sum(v) != __shfl(ea, 63, 64)) atomicAdd(bad, 1);
}

}  // namespace smf
