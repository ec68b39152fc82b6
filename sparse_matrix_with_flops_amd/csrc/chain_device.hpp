// chain_device.hpp — ONE-PASS SpGEMM for the rows of at most `rmax` products (round 4).
//
// The two-pass pipeline (symbolic count -> scan -> numeric) walks every product twice because a row of C can only be
// placed once the sizes of all rows before it are known.  Here a block accumulates a BATCH of consecutive rows in LDS
// -- one pool of (column, value) slots, one region per row, laid out in row order -- and then learns where the batch
// starts in C from a chained prefix over the batches (decoupled look-back: a batch publishes its entry count as soon as
// its accumulation is complete, sums the counts of the batches before it and publishes the inclusive prefix).  Batches
// are handed out in row order from a ticket counter, so a batch only ever waits for blocks that are already running.
// The rows of a batch are consecutive rows of A, so its A entries are ONE contiguous stream: units of 64*U products are
// dealt to the waves from the whole batch (each product carries its row's region), and every per-row step of the old
// kernels (dequeue, row pointers, table clear, count, output range) is done once per batch by parallel lanes.
// No symbolic pass, no scan of IC, no binning for these rows; rowPtr, colInd and values come out of the one kernel.
//
// Rows above `rmax` products (bins 7/8 of the old layout) keep their symbolic + numeric kernels: their exact counts are
// in IC before this kernel starts and enter the chain as known terms, their rowPtr entries are written here.
//
// Reference counterparts: the per-sub-warp tables of sgpu_SpGEMM_mid (mindex2-cuda/gspgemm.cuh:241-293, hbs[warps][HPRIME]:
// several rows per block on small tables), hashCASAdd2 (casHash.cuh:34-43), and the IC scan it replaces
// (thrust::exclusive_scan, "mindex2-cuda/\":532-555).
#pragma once
#include "spgemm_device.hpp"

namespace smf {

constexpr int ERRF_CHAIN_LAYOUT = 4;      // a batch does not fit the pool / a counted row is not the last of its batch
constexpr int ERRF_CHAIN_CAP = 8;         // C (sized from the previous call of a compressive product) is too small: redo

// ------------------------------------------------------------------------------------------------
// The cut: rows -> batches, without a sort and without a compaction.
//   small rows (f <= smallMax products) get a table of 2f slots; consecutive small rows share a batch while the running
//     sum of their slots stays inside one window of poolW slots (the batch then holds < poolW + 2*smallMax <= POOL slots);
//   solo rows (smallMax < f <= rmax) are a batch of their own (table 2f <= POOL);
//   counted rows (f > rmax: not accumulated here) end their batch;
//   a batch never crosses a multiple of rb rows (per-row metadata of a batch lives in LDS).
// id(r) = slotsBefore(r) / poolW + r / rb + specialsBefore(r) + isSolo(r) is non-decreasing in r; batch j = the rows with
// id == j (possibly none).  One 64-bit scan carries both sums: low 40 bits slots, high bits the specials (1 per counted
// row, 2 per solo row).
// ------------------------------------------------------------------------------------------------
struct CutParams { int smallMax, rmax, poolW, rb; };

__device__ __forceinline__ unsigned long long cut_value(int f, const CutParams& p) {
  if (f <= p.smallMax) return (unsigned long long)(2 * f);
  return f <= p.rmax ? (2ull << 40) : (1ull << 40);
}
__device__ __forceinline__ int cut_id(unsigned long long excl, int r, int f, const CutParams& p) {
  const unsigned long long slots = excl & ((1ull << 40) - 1ull);
  return (int)(slots / (unsigned)p.poolW) + r / p.rb + (int)(excl >> 40) + ((f > p.smallMax && f <= p.rmax) ? 1 : 0);
}

__global__ __launch_bounds__(SCAN_THREADS) void k_cut_sums(int m, const int* __restrict__ rowFlops, CutParams p,
                                                            unsigned long long* __restrict__ tileSum) {
  __shared__ unsigned long long red[16];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int base = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
  unsigned long long s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) if (base + i < m) s += cut_value(rowFlops[base + i], p);
  s = wave_sum_u64(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (tid == 0) { unsigned long long t = 0; for (int i = 0; i < 16; ++i) t += red[i]; tileSum[blockIdx.x] = t; }
}

// tileOff = exclusive scan of the tile sums (k_scan_tiles).  Writes batchStart[0..nb], zeroes the chain words of the
// batches, *nBatches = nb.
__global__ __launch_bounds__(SCAN_THREADS) void k_cut_apply(int m, const int* __restrict__ rowFlops, CutParams p,
                                                             const unsigned long long* __restrict__ tileOff,
                                                             int* __restrict__ batchStart,
                                                             unsigned long long* __restrict__ chain,
                                                             int* __restrict__ nBatches) {
  __shared__ unsigned long long wsum[16];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int base = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
  int f[SCAN_ITEMS];
  unsigned long long v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    f[i] = base + i < m ? rowFlops[base + i] : 0;
    v[i] = base + i < m ? cut_value(f[i], p) : 0ull;
    s += v[i];
  }
  const unsigned long long incl = wave_incl_add_u64(s);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  unsigned long long woff = 0;
  for (int i = 0; i < w; ++i) woff += wsum[i];
  unsigned long long run = tileOff[blockIdx.x] + woff + incl - s;      // slots / specials before row `base`
  int prevId = -1;
  if (base > 0 && base < m) {
    const int fp = rowFlops[base - 1];
    prevId = cut_id(run - cut_value(fp, p), base - 1, fp, p);
  }
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const int r = base + i;
    if (r < m) {
      const int id = cut_id(run, r, f[i], p);
      for (int j = prevId + 1; j <= id; ++j) { batchStart[j] = r; chain[j] = 0ull; }
      prevId = id;
      run += v[i];
      if (r == m - 1) { batchStart[id + 1] = m; *nBatches = id + 1; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// inserts into per-row regions of one pool: region = first slot | slots << 16 (per product).  Slot = mulhi(hash, slots),
// linear probing inside the region (any size; load <= 1/2).  Otherwise hash_accum_multi.
// ------------------------------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ void hash_accum_regions(slot_t* pool, const bool (&act)[U], const int (&col)[U],
                                                   const float (&val)[U], const int (&reg)[U], slot_t* dummy, int* err,
                                                   int maxProbe) {
  char* const base = reinterpret_cast<char*>(pool);
  const int dumB = (int)(reinterpret_cast<char*>(dummy) - base);
  int hB[U], loB[U], hiB[U], dupB[U];
  slot_t mine[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned hv = (unsigned)col[u] * 2654435761u;
    const unsigned size = (unsigned)reg[u] >> 16, start = (unsigned)reg[u] & 0xffffu;
    loB[u] = (int)(start * 8u);
    hiB[u] = (int)((start + size) * 8u);
    hB[u] = act[u] ? (int)((start + __umulhi(hv, size)) * 8u) : dumB;
    dupB[u] = dumB;
    mine[u] = make_slot(col[u], val[u]);
  }
  int probe = 0;
  for (;;) {
    slot_t old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) old[u] = atomicCAS(reinterpret_cast<slot_t*>(base + hB[u]), EMPTY_SLOT, mine[u]);
    ++probe;
    int pendBits = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool pend = hB[u] != dumB;
      const bool same = pend && slot_key(old[u]) == col[u];
      const bool adv = pend && old[u] != EMPTY_SLOT && slot_key(old[u]) != col[u];
      dupB[u] = same ? hB[u] : dupB[u];
      int nh = hB[u] + 8;
      nh = nh == hiB[u] ? loB[u] : nh;
      hB[u] = adv ? nh : dumB;
      pendBits |= hB[u] ^ dumB;
    }
    if (ballot64(pendBits != 0) == 0ull) break;
    if (probe >= maxProbe) { atomicOr(err, ERRF_TABLE_FULL); break; }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (dupB[u] != dumB) atomicAdd(reinterpret_cast<float*>(base + dupB[u] + 4), val[u]);
}

// ------------------------------------------------------------------------------------------------
// The product walk of a batch: for_each_product<NW> over the batch's A entries as one stream (the entries of its small
// and solo rows, in row order; the rows are consecutive, so the stream is contiguous in A except where a counted row
// sits), every product tagged with its row's region.
// ------------------------------------------------------------------------------------------------
template <int NW>
struct BatchStage {
  unsigned long long marks[NW][WAVE];
  int4 rec[NW][WAVE];                    // short entries from the bottom {off, a, region, -}; long ones from the top
                                         // {bs, a, len | unit offset << 16, region} (len <= rmax < 65536)
  unsigned char wpre[NW][WAVE];
  int gT[NW], gNS[NW], gNL[NW], gLU[NW];
  int claim;
  slot_t dummy[WAVE];
};

template <int U, class F>
__device__ __forceinline__ void short_trip_b(const char* rec, int T, int ns, int r0, const unsigned long long (&W)[U],
                                             const int (&base)[U], const int* __restrict__ JB,
                                             const float* __restrict__ VB, F&& f) {
  const int lane = lane_id();
  int col[U], reg[U];
  float av[U], vb[U], val[U];
  bool act[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p0 = (r0 + u) * WAVE + lane;
    act[u] = p0 < T;
    const int p = min(p0, T - 1);
    int e = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(W[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)W[u], (unsigned)base[u]));
    e = min(e, ns - 1);
    const int4 ra = *reinterpret_cast<const int4*>(rec + e * 16);
    const int jb = ra.x + p;
    col[u] = JB[jb];
    vb[u] = VB[jb];
    av[u] = __int_as_float(ra.y);
    reg[u] = ra.z;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) val[u] = av[u] * vb[u];
  f(act, col, val, reg);
}

template <int U, class F>
__device__ __forceinline__ void long_trip_b(int kb, int kl, float ka, int region, int s0, const int* __restrict__ JB,
                                            const float* __restrict__ VB, F&& f) {
  const int lane = lane_id();
  int col[U], reg[U];
  float vb[U], val[U];
  bool act[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p0 = s0 + u * WAVE + lane;
    act[u] = p0 < kl;
    const int jb = kb + min(p0, kl - 1);
    col[u] = JB[jb];
    vb[u] = VB[jb];
    reg[u] = region;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) val[u] = ka * vb[u];
  f(act, col, val, reg);
}

// rowE[i]: first A entry of batch row i; rowV[i]: entries of the batch's accumulated rows before row i (rowV[nr] = V);
// rowT[i]: first slot of row i's region (rowT[nr] = slots used).  RBP2: power of two >= rows per batch.
template <int NW, int U, int RBP2, class F>
__device__ __forceinline__ void batch_walk(BatchStage<NW>& st, const int* rowE, const int* rowV, const int* rowT, int nr,
                                           int V, const int2* __restrict__ SBL, const float* __restrict__ VA,
                                           const int* __restrict__ JB, const float* __restrict__ VB, F&& f) {
  static_assert(NW <= 16, "the unit table of a chunk lives in the first 16 lanes");
  constexpr int K = WAVE * NW;
  constexpr int UP = WAVE * U;
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  for (int chunk = 0; chunk < V; chunk += K) {
    {
      const int v = chunk + tid;
      const bool valid = v < V;
      int i = 0;                                           // the row this entry belongs to: last i with rowV[i] <= v
#pragma unroll
      for (int s = RBP2 / 2; s >= 1; s >>= 1) {
        const int c = i + s;
        if (c < nr && rowV[c] <= v) i = c;
      }
      PreA pre{0, 0, 0.f, true};
      int region = 0;
      if (valid) {
        const int e = rowE[i] + (v - rowV[i]);
        const int2 sbl = SBL[e];
        pre.bs = sbl.x; pre.len = sbl.y; pre.a = VA[e];
        const int t0 = rowT[i];
        region = t0 | ((rowT[i + 1] - t0) << 16);
      }
      const GroupLanes g = stage_group<true, 16, LONG_LEN>(st.marks[w], reinterpret_cast<char*>(st.rec[w]), valid ? 0 : 1, 1,
                                                           SBL, VA, pre);
      const bool isLong = (g.lmask >> lane) & 1ull;
      const bool keep = !isLong && g.len > 0;
      const unsigned long long km = ballot64(keep);
      if (keep) reinterpret_cast<int*>(st.rec[w])[mask_rank(km) * 4 + 2] = region;
      const int units = isLong ? (g.len + UP - 1) / UP : 0;
      const int uincl = wave_incl_add(units);
      if (isLong) {
        const int k = mask_rank(g.lmask);
        st.rec[w][WAVE - 1 - k] = make_int4(g.bs, __float_as_int(g.a), g.len | ((uincl - units) << 16), region);
      }
      st.wpre[w][lane] = (unsigned char)g.pexcl;
      if (lane == 0) { st.gT[w] = g.T; st.gNS[w] = g.ns; st.gNL[w] = __popcll(g.lmask); }
      if (lane == 63) st.gLU[w] = uincl;
      if (tid == 0) st.claim = NW;
    }
    __syncthreads();
    const int gT = lane < NW ? st.gT[lane] : 0;
    const int gNS = lane < NW ? st.gNS[lane] : 0;
    const int gNL = lane < NW ? st.gNL[lane] : 0;
    const int lug = lane < NW ? st.gLU[lane] : 0;
    const int ntg = (((gT + WAVE - 1) >> 6) + U - 1) / U;
    const int sIncl = wave_incl_add(ntg), lIncl = wave_incl_add(lug);
    const int S = __builtin_amdgcn_readlane(sIncl, 63);
    const int total = S + __builtin_amdgcn_readlane(lIncl, 63);
    for (int u = __builtin_amdgcn_readfirstlane(w); u < total;) {
      int unext = 0;
      if (lane == 0) unext = atomicAdd(&st.claim, 1);
      if (u < S) {
        const int g = __popcll(ballot64(sIncl <= u));
        const int t = u - (__builtin_amdgcn_readlane(sIncl, g) - __builtin_amdgcn_readlane(ntg, g));
        const int T = __builtin_amdgcn_readlane(gT, g), ns = __builtin_amdgcn_readlane(gNS, g);
        unsigned long long W[U];
        int base[U];
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
          const int r = min(t * U + uu, WAVE - 1);
          W[uu] = st.marks[g][r];
          base[uu] = st.wpre[g][r];
        }
        short_trip_b<U>(reinterpret_cast<const char*>(st.rec[g]), T, ns, t * U, W, base, JB, VB, f);
      } else {
        const int ul = u - S;
        const int g = __popcll(ballot64(lIncl <= ul));
        const int ug = ul - (__builtin_amdgcn_readlane(lIncl, g) - __builtin_amdgcn_readlane(lug, g));
        const int nl = __builtin_amdgcn_readlane(gNL, g);
        const int uex = lane < nl ? (int)((unsigned)st.rec[g][WAVE - 1 - lane].z >> 16) : 0x7fffffff;
        const int k = __popcll(ballot64(uex <= ug)) - 1;
        const int4 r4 = st.rec[g][WAVE - 1 - k];
        long_trip_b<U>(r4.x, r4.z & 0xffff, __int_as_float(r4.y), r4.w, (ug - (int)((unsigned)r4.z >> 16)) * UP, JB, VB, f);
      }
      u = __builtin_amdgcn_readfirstlane(unext);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the callback's return-less LDS atomics have landed
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// The chain.  chain[j]: low 32 bits a count, bits 32-33 its meaning: 0 nothing yet, 1 = entries of batch j,
// 2 = entries of batches 0..j.  One wave looks back 64 batches at a time.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long chain_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// entries of the batches before t (called by one whole wave; wave-uniform result)
__device__ __forceinline__ unsigned chain_lookback(const unsigned long long* chain, int t) {
  const int lane = lane_id();
  unsigned sum = 0;
  int pos = t - 1;
  while (pos >= 0) {
    const int idx = pos - lane;
    const unsigned long long v = idx >= 0 ? chain_load(chain + idx) : (2ull << 32);   // before batch 0: prefix 0
    const unsigned st = (unsigned)(v >> 32);
    const unsigned long long m0 = ballot64(st == 0u), m2 = ballot64(st == 2u);
    const int first2 = m2 ? __builtin_ctzll(m2) : 64;
    const int first0 = m0 ? __builtin_ctzll(m0) : 64;
    if (first0 < first2) { __builtin_amdgcn_s_sleep(4); continue; }         // a batch in between has not counted yet
    const int upto = min(first2, 63);                                       // lanes 0..upto contribute
    const int part = wave_sum(lane <= upto ? (int)(unsigned)v : 0);
    sum += (unsigned)part;
    if (first2 < 64) break;
    pos -= 64;
  }
  return sum;
}

template <int NW, int POOL, int RB>
struct ChainShared {
  slot_t pool[POOL];
  BatchStage<NW> st;
  int rowE[RB + 1], rowV[RB + 1], rowT[RB + 1];
  unsigned long long stepMask[POOL / WAVE];
  int stepPref[POOL / WAVE];
  int ws[3][NW];
  int ticket;
  unsigned base;
  int total;
};

#ifdef SMF_STAMPS
__device__ unsigned long long g_cstamps[16];   // diagnostic build: cycles of wave 0 per phase of k_chain (spgemm_hip_debug_chain_stamps)
#endif
template <int NW, int POOL, int RB, int U>
__global__ __launch_bounds__(WAVE * NW) void k_chain(int m, const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                      const float* __restrict__ VA, const int* __restrict__ JB,
                                                      const float* __restrict__ VB, const int* __restrict__ rowFlops,
                                                      int rmax, const int* __restrict__ batchStart,
                                                      const int* __restrict__ nBatchesPtr, unsigned long long* chain,
                                                      int* ticketCtr, int* IC, int* __restrict__ JC,
                                                      float* __restrict__ C, long long capC,
                                                      unsigned long long* __restrict__ nnzC64, int* __restrict__ err) {
  static_assert(POOL % WAVE == 0 && POOL <= 65535 && RB < WAVE * NW && (RB & (RB - 1)) == 0, "pool / batch geometry");
  static_assert(POOL / WAVE <= 2 * WAVE, "one wave scans the step counts, two per lane");
  constexpr int NT = WAVE * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef ChainShared<NW, POOL, RB> Sh;
  Sh& sh = *reinterpret_cast<Sh*>(smem_raw);
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  for (int i = tid * 2; i < POOL; i += NT * 2) *reinterpret_cast<ulonglong2*>(sh.pool + i) = make_ulonglong2(EMPTY_SLOT, EMPTY_SLOT);
  if (tid < WAVE) sh.st.dummy[tid] = DUMMY_SLOT;
  const int NB = *nBatchesPtr;
  // Tickets are taken ONE at a time, the next one when a batch starts (its row range is then fetched behind the batch's
  // work).  Never two consecutive tickets at once: the second would wait for its holder to finish the first while every
  // later batch waits for it -- the chain would run one batch at a time.
  if (tid == 0) sh.ticket = atomicAdd(ticketCtr, 1);
  __syncthreads();
  int t = sh.ticket;
  int r0 = t < NB ? batchStart[t] : 0, r1 = t < NB ? batchStart[t + 1] : 0;
  __syncthreads();
  STAMP_DECL
  while (t < NB) {
    STAMP(7)
    const int nr = min(r1 - r0, RB);
    // ---- phase 0: the batch's rows -> regions of the pool, the entry stream
    int f = 0, ia0 = 0, ia1 = 0, bigCnt = 0;
    if (tid < nr) { f = rowFlops[r0 + tid]; ia0 = IA[r0 + tid]; ia1 = IA[r0 + tid + 1]; }
    const bool counted = tid < nr && f > rmax;
    if (counted) bigCnt = IC[r0 + tid];                                    // exact count from its symbolic kernel
    const int ts = (tid < nr && !counted) ? 2 * f : 0;
    const int ec = (tid < nr && !counted) ? ia1 - ia0 : 0;
    const int inclT = wave_incl_add(ts), inclV = wave_incl_add(ec), bcw = wave_sum(bigCnt);
    if (lane == 63) { sh.ws[0][w] = inclT; sh.ws[1][w] = inclV; }
    if (lane == 0) sh.ws[2][w] = bcw;
    const bool misplaced = counted && tid != nr - 1;                       // the cut ends a batch after a counted row
    __syncthreads();
    int offT = 0, offV = 0, bigTotal = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      offT += i < w ? sh.ws[0][i] : 0;
      offV += i < w ? sh.ws[1][i] : 0;
      bigTotal += sh.ws[2][i];
    }
    if (tid <= nr) {                                                       // thread nr holds the totals (its own terms are 0)
      sh.rowT[tid] = offT + inclT - ts;
      sh.rowV[tid] = offV + inclV - ec;
      sh.rowE[tid] = ia0;
    }
    __syncthreads();
    const int TT = sh.rowT[nr], V = sh.rowV[nr];
    const bool bad = TT > POOL || r1 - r0 > RB;
    if (misplaced || (bad && tid == 0)) atomicOr(err, ERRF_CHAIN_LAYOUT);
    STAMP(0)
    // ---- phase 1: accumulate
    if (!bad) {
      batch_walk<NW, U, RB>(sh.st, sh.rowE, sh.rowV, sh.rowT, nr, V, SBL, VA, JB, VB,
                            [&](const bool (&act)[U], const int (&col)[U], const float (&val)[U], const int (&reg)[U]) {
        hash_accum_regions<U>(sh.pool, act, col, val, reg, &sh.st.dummy[lane_id()], err, POOL);
      });
    }
    STAMP(1)
    // ---- phase 2: occupancy of the pool in slot order = row order
    const int nsteps = bad ? 0 : (TT + WAVE - 1) >> 6;
    for (int g = w; g < nsteps; g += NW) {
      const slot_t sv = sh.pool[g * WAVE + lane];
      const unsigned long long mk = ballot64(slot_key(sv) != EMPTY_KEY);
      if (lane == 0) sh.stepMask[g] = mk;
    }
    __syncthreads();
    if (w == 0) {
      const int g0 = 2 * lane, g1 = 2 * lane + 1;
      const int c0 = g0 < nsteps ? __popcll(sh.stepMask[g0]) : 0, c1 = g1 < nsteps ? __popcll(sh.stepMask[g1]) : 0;
      const int incl = wave_incl_add(c0 + c1);
      if (g0 < nsteps) sh.stepPref[g0] = incl - c0 - c1;
      if (g1 < nsteps) sh.stepPref[g1] = incl - c1;
      const int total = __builtin_amdgcn_readlane(incl, 63);
      const unsigned agg = (unsigned)(total + bigTotal);
      if (lane == 0) chain_store(chain + t, (1ull << 32) | agg);
      STAMP(2)
      const unsigned base = chain_lookback(chain, t);
      STAMP(3)
      if (lane == 0) {
        chain_store(chain + t, (2ull << 32) | (unsigned long long)(base + agg));
        sh.base = base;
        sh.total = total;
        if (t == NB - 1) {
          const unsigned long long all = (unsigned long long)base + agg;
          *nnzC64 = all;
          IC[m] = (int)min(all, (unsigned long long)capC);               // (capC < 2^31)
        }
      }
    }
    __syncthreads();
    STAMP(4)
    // the next ticket is taken only now, when this batch has its place: a ticket taken earlier would sit unprocessed for
    // as long as this batch waits, and every batch behind it waits for it in turn
    if (tid == 0) sh.ticket = atomicAdd(ticketCtr, 1);
    const unsigned base = sh.base;
    const int total = sh.total;
    // ---- phase 3: rowPtr of the batch's rows, the entries in slot order, the pool left empty
    if (tid < nr) {
      const int s = sh.rowT[tid];
      const int rank = s >= TT || bad ? total : sh.stepPref[s >> 6] + __popcll(sh.stepMask[s >> 6] & ((1ull << (s & 63)) - 1ull));
      // never beyond the capacity of C: the kernels of the counted rows place their entries by these offsets
      IC[r0 + tid] = (int)min((long long)(base + (unsigned)rank), capC);
    }
    bool over = false;
    for (int g = w; g < nsteps; g += NW) {
      const slot_t sv = sh.pool[g * WAVE + lane];
      sh.pool[g * WAVE + lane] = EMPTY_SLOT;
      const unsigned long long mk = sh.stepMask[g];
      const bool occ = slot_key(sv) != EMPTY_KEY;
      const unsigned o = base + (unsigned)(sh.stepPref[g] + mask_rank(mk));
      if (occ) {
        if ((long long)o < capC) { st_out(JC + o, slot_key(sv)); st_out(C + o, slot_val(sv)); }
        else over = true;
      }
    }
    if (over) atomicOr(err, ERRF_CHAIN_CAP);
    __syncthreads();
    STAMP(5)
#ifdef SMF_STAMPS
    st_[10] += 1; st_[11] += (unsigned long long)V;
#endif
    t = sh.ticket;
    r0 = t < NB ? batchStart[t] : 0; r1 = t < NB ? batchStart[t + 1] : 0;
    __syncthreads();
  }
#ifdef SMF_STAMPS
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_cstamps[i], st_[i]);
    atomicAdd(&g_cstamps[8], __builtin_readcyclecounter() - t0_);
    atomicAdd(&g_cstamps[9], 1ull);
    atomicAdd(&g_cstamps[10], st_[10]);
    atomicAdd(&g_cstamps[11], st_[11]);
  }
#endif
}

}  // namespace smf
