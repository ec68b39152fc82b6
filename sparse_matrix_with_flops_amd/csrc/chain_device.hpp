// chain_device.hpp — rows of at most `smallMax` products, dealt ACROSS rows (round 4).
//
// The per-row kernels of spgemm_device.hpp spend a third of their time per ROW, not per product: dequeue, row pointers,
// table clear, staging ~10 A entries on 64 lanes, count, output bookkeeping -- all of it by a wave whose lanes are mostly
// idle (profiles/README.md, "Per row or per round?").  Here a wave owns a BATCH of consecutive rows instead:
//   * the rows of a batch are consecutive rows of A, so their A entries are one contiguous stream: it is staged 64
//     entries at a time whatever row they belong to, and the products of a staged group are flattened over the lanes
//     exactly as in for_each_product -- each product additionally carries its row's REGION of the wave's table;
//   * one table per wave, one region per row, regions laid out in row order: after the accumulation the occupied slots in
//     slot order ARE the batch's part of C in row order -- one sweep gives every row's count (ranks at the region
//     boundaries) and every entry's position (its rank);
//   * every per-row step is done once per batch by parallel lanes (lane i = row i of the batch).
// Three uses of the same walk (template MODE):
//   WB_SYM   counts only                -> IC[row] (then the scan of spgemm_device.hpp)
//   WB_NUM   offsets known (IC scanned) -> colInd / values
//   WB_CHAIN ONE PASS: no symbolic pass at all.  A block of NW waves takes NW consecutive batches with one ticket (they
//            run side by side, one per wave), adds up their counts and learns where its group starts in C from a chained
//            prefix over the groups (decoupled look-back: publish the group's count, sum the counts of the groups
//            before, publish the inclusive prefix).  Tickets are handed out in row order and only when a block is ready
//            to start, so a group only ever waits for groups that are being worked on.
// Rows above smallMax products keep their kernels; in WB_CHAIN their exact counts (their symbolic kernels ran before) enter
// the chain as known terms and their rowPtr entries are written here.
//
// What was measured before this form (profiles/README.md, round 4): the same idea with a whole BLOCK cooperating on one
// batch (units of products dealt to 8-16 waves from one list, block barriers between the phases) is 2x slower than the
// per-row kernels -- five dependent global round trips per batch and 8-16 waves per CU leave the memory system idle.
//
// Reference counterparts: several rows per block on per-sub-warp tables, sgpu_SpGEMM_mid (mindex2-cuda/gspgemm.cuh:241-293,
// hbs[warps][HPRIME]); hashCASAdd2 (casHash.cuh:34-43); the IC scan the chain replaces (thrust::exclusive_scan,
// "mindex2-cuda/\":532-555); the tiny-row kernels that never ran a symbolic pass (gspgemm.cuh:2-59,173-213).
#pragma once
#include "spgemm_device.hpp"

namespace smf {

constexpr int ERRF_CHAIN_LAYOUT = 4;      // a batch does not fit its table
constexpr int ERRF_CHAIN_CAP = 8;         // C (sized from the previous call of a compressive product) is too small: redo

// slots of the region of a row with f products: load <= 2/3 when no two products meet
__host__ __device__ __forceinline__ int wb_slots(int f) { return f > 0 ? f + (f >> 1) + 2 : 0; }

// ------------------------------------------------------------------------------------------------
// The cut: rows -> batches, without a sort and without a compaction.  Rows of at most smallMax products get a region of
// wb_slots(f) slots; consecutive rows share a batch while the running sum of their slots stays inside one window of
// poolW slots (a batch then holds < poolW + wb_slots(smallMax) <= TBL slots); rows above smallMax take no slots (they
// are not accumulated here) and sit inside the batch they fall into; a batch never crosses a multiple of rb rows
// (lane i of the wave = row i of its batch).
// id(r) = slotsBefore(r) / poolW + r / rb is non-decreasing in r; batch j = the rows with id == j (possibly none).
// ------------------------------------------------------------------------------------------------
struct CutParams { int smallMax, poolW, rb; };

__device__ __forceinline__ unsigned long long cut_value(int f, const CutParams& p) {
  return f <= p.smallMax ? (unsigned long long)wb_slots(f) : 0ull;
}
__device__ __forceinline__ int cut_id(unsigned long long excl, int r, const CutParams& p) {
  return (int)(excl / (unsigned)p.poolW) + r / p.rb;
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

// tileOff = exclusive scan of the tile sums (k_scan_tiles).  Writes batchStart[0..nb], zeroes the chain words,
// *nBatches = nb, *ticket = 0.
__global__ __launch_bounds__(SCAN_THREADS) void k_cut_apply(int m, const int* __restrict__ rowFlops, CutParams p,
                                                             const unsigned long long* __restrict__ tileOff,
                                                             int* __restrict__ batchStart,
                                                             unsigned long long* __restrict__ chain,
                                                             int* __restrict__ nBatches, int* __restrict__ ticket) {
  __shared__ unsigned long long wsum[16];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int base = blockIdx.x * SCAN_TILE + tid * SCAN_ITEMS;
  unsigned long long v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = base + i < m ? cut_value(rowFlops[base + i], p) : 0ull;
    s += v[i];
  }
  const unsigned long long incl = wave_incl_add_u64(s);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  unsigned long long woff = 0;
  for (int i = 0; i < w; ++i) woff += wsum[i];
  unsigned long long run = tileOff[blockIdx.x] + woff + incl - s;      // slots before row `base`
  int prevId = -1;
  if (base > 0 && base < m) prevId = cut_id(run - cut_value(rowFlops[base - 1], p), base - 1, p);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    const int r = base + i;
    if (r < m) {
      const int id = cut_id(run, r, p);
      for (int j = prevId + 1; j <= id; ++j) { batchStart[j] = r; chain[j] = 0ull; }
      prevId = id;
      run += v[i];
      if (r == m - 1) { batchStart[id + 1] = m; *nBatches = id + 1; *ticket = 0; }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// inserts into per-row regions of one table: region = first slot | slots << 16 (per product).  Slot = mulhi(hash, slots),
// linear probing inside the region (any size).  Otherwise hash_accum_multi.
// ------------------------------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ void hash_accum_regions(slot_t* tab, const bool (&act)[U], const int (&col)[U],
                                                   const float (&val)[U], const int (&reg)[U], slot_t* dummy, int* err,
                                                   int maxProbe) {
  char* const base = reinterpret_cast<char*>(tab);
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

// U rounds of 64 products of a staged group; rec = {JB offset of product 0 in the group's numbering, a, region, -}
template <int U, bool NEED_VAL, class F>
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
    vb[u] = NEED_VAL ? VB[jb] : 0.f;
    av[u] = __int_as_float(ra.y);
    reg[u] = ra.z;
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) val[u] = av[u] * vb[u];
  f(act, col, val, reg);
}

// ------------------------------------------------------------------------------------------------
// The chain (WB_CHAIN).  chain[j]: low 32 bits a count, bits 32-33 its meaning: 0 nothing yet, 1 = entries of group j,
// 2 = entries of groups 0..j.  One wave looks back 128 groups at a time (two words per lane).
// The frontier of known prefixes advances by at most one look-back width per memory round trip, whatever the number of
// waiting groups: a chain entry has to stand for thousands of products (here: NW batches), not hundreds.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long chain_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// entries of the groups before g (called by one whole wave; wave-uniform result)
__device__ __forceinline__ unsigned chain_lookback(const unsigned long long* chain, int g) {
  const int lane = lane_id();
  unsigned sum = 0;
  int pos = g - 1;                                           // nearest group not summed yet
  while (pos >= 0) {
    // lane l looks at groups pos-2l (a) and pos-2l-1 (b)
    const int ia = pos - 2 * lane, ib = ia - 1;
    const unsigned long long va = ia >= 0 ? chain_load(chain + ia) : (2ull << 32);     // before group 0: prefix 0
    const unsigned long long vb = ib >= 0 ? chain_load(chain + ib) : (2ull << 32);
    const unsigned sa = (unsigned)(va >> 32), sb = (unsigned)(vb >> 32);
    // distance (in groups) from pos to the first word that is not ready / that is a prefix
    const unsigned long long a0 = ballot64(sa == 0u), b0 = ballot64(sb == 0u), a2 = ballot64(sa == 2u), b2 = ballot64(sb == 2u);
    const int f0 = min(a0 ? 2 * (int)__builtin_ctzll(a0) : 128, b0 ? 2 * (int)__builtin_ctzll(b0) + 1 : 128);
    const int f2 = min(a2 ? 2 * (int)__builtin_ctzll(a2) : 128, b2 ? 2 * (int)__builtin_ctzll(b2) + 1 : 128);
    if (f0 < f2) { __builtin_amdgcn_s_sleep(8); continue; }                  // a group in between has not counted yet
    const int upto = min(f2, 127);                                           // words at distance 0..upto contribute
    const int part = wave_sum((2 * lane <= upto ? (int)(unsigned)va : 0) + (2 * lane + 1 <= upto ? (int)(unsigned)vb : 0));
    sum += (unsigned)part;
    if (f2 < 128) break;
    pos -= 128;
  }
  return sum;
}

// ------------------------------------------------------------------------------------------------
// k_wbatch: one wave = one batch of consecutive rows.
// ------------------------------------------------------------------------------------------------
constexpr int WB_SYM = 0, WB_NUM = 1, WB_CHAIN = 2;
constexpr int WB_MAXCOUNTED = WAVE;            // rows above smallMax inside one batch (any number of its rows)

template <int TBL>
struct WaveBatchLds {
  slot_t tab[TBL];
  unsigned long long marks[WAVE];
  int4 rec[WAVE];
  slot_t dummy[WAVE];
  int rowE[WAVE + 1], rowV[WAVE + 1], rowT[WAVE + 1];
  unsigned long long stepMask[TBL / WAVE];
  int stepPref[TBL / WAVE];
  // (the list of the batch's rows above smallMax, {first slot behind the row, its entries}, is written into `rec` once the
  // walk is done with it)
};
static_assert(WB_MAXCOUNTED * sizeof(int2) <= WAVE * sizeof(int4), "the counted-row list lives in the staging records");
template <int NW, int TBL>
struct WaveBatchShared {
  WaveBatchLds<TBL> w[NW];
  unsigned wagg[NW];
  int ticket;
  unsigned base;
};

#ifdef SMF_STAMPS
__device__ unsigned long long g_cstamps[16];   // diagnostic build: cycles of wave 0 per phase (spgemm_hip_debug_chain_stamps)
#endif

// IC: WB_SYM out: counts of the batch rows up to smallMax products.  WB_NUM in: scanned row pointers.  WB_CHAIN: in the
// counts of the rows above smallMax (their symbolic kernels), out the row pointers of ALL rows, IC[m] and *nnzC64.
template <int NW, int TBL, int MODE, int MAXR>
__global__ __launch_bounds__(WAVE * NW) void k_wbatch(int m, const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                       const float* __restrict__ VA, const int* __restrict__ JB,
                                                       const float* __restrict__ VB, const int* __restrict__ rowFlops,
                                                       int smallMax, const int* __restrict__ batchStart,
                                                       const int* __restrict__ nBatchesPtr, unsigned long long* chain,
                                                       int* ticketCtr, int* IC, int* __restrict__ JC,
                                                       float* __restrict__ C, long long capC,
                                                       unsigned long long* __restrict__ nnzC64, int* __restrict__ err) {
  static_assert(TBL % WAVE == 0 && TBL <= 4096, "one lane per 64-slot step; a staged group numbers at most 4096 products");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  typedef WaveBatchShared<NW, TBL> Sh;
  Sh& sh = *reinterpret_cast<Sh*>(smem_raw);
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  WaveBatchLds<TBL>& L = sh.w[w];
  constexpr bool NEED_VAL = MODE != WB_SYM;
  for (int i = lane; i < TBL; i += WAVE) L.tab[i] = EMPTY_SLOT;
  L.dummy[lane] = DUMMY_SLOT;
  const int NB = *nBatchesPtr;
  STAMP_DECL
  for (int round = 0;; ++round) {
    // ---- which batch: WB_CHAIN takes NW consecutive batches per block with one ticket, in row order; the two-pass modes
    // need no order (static schedule)
    int t0;
    if (MODE == WB_CHAIN) {
      __syncthreads();                                               // (everyone is done with sh.base / sh.wagg of the last round)
      if (tid == 0) sh.ticket = atomicAdd(ticketCtr, NW);
      __syncthreads();
      t0 = sh.ticket;
    } else {
      t0 = (round * (int)gridDim.x + (int)blockIdx.x) * NW;
    }
    if (t0 >= NB) break;                                             // block-uniform
    STAMP(0)
    const int t = t0 + w;
    const bool liveBatch = t < NB;
    const int r0 = liveBatch ? batchStart[t] : 0, r1 = liveBatch ? batchStart[t + 1] : 0;
    const int nr = min(r1 - r0, WAVE);
    // ---- lane i = row i of the batch: region of the table, part of the entry stream
    int f = 0, ia0 = 0, ia1 = 0, cnt = 0;
    if (lane < nr) { f = rowFlops[r0 + lane]; ia0 = IA[r0 + lane]; ia1 = IA[r0 + lane + 1]; }
    const bool counted = lane < nr && f > smallMax;                 // not accumulated here
    if (MODE == WB_CHAIN && counted) cnt = IC[r0 + lane];            // exact, from its symbolic kernel
    if (MODE == WB_NUM && counted) cnt = IC[r0 + lane + 1] - IC[r0 + lane];
    const int ts = (lane < nr && !counted) ? wb_slots(f) : 0;
    const int ec = (lane < nr && !counted) ? ia1 - ia0 : 0;
    const int inclT = wave_incl_add(ts), inclV = wave_incl_add(ec), inclC = wave_incl_add(cnt);
    const int TT = __builtin_amdgcn_readlane(inclT, 63), V = __builtin_amdgcn_readlane(inclV, 63);
    const int countedTotal = __builtin_amdgcn_readlane(inclC, 63);
    const int myT = inclT - ts;                                      // first slot of my row's region
    L.rowT[lane] = myT; L.rowV[lane] = inclV - ec; L.rowE[lane] = ia0;
    if (lane == 63) { L.rowT[WAVE] = TT; L.rowV[WAVE] = V; L.rowE[WAVE] = 0; }
    const unsigned long long cmask = ballot64(counted);
    const int nc = MODE == WB_SYM ? 0 : __popcll(cmask);
    const bool bad = TT > TBL || r1 - r0 > WAVE || nc > WB_MAXCOUNTED;
    if (bad && lane == 0) atomicOr(err, ERRF_CHAIN_LAYOUT);
    wave_lds_sync();
    STAMP(1)
    // ---- the products of the batch, 64 A entries at a time, all rounds of a group gathered before the first insert
    if (!bad) {
      for (int g0 = 0; g0 < V; g0 += WAVE) {
        const int v = g0 + lane;
        const bool valid = v < V;
        int i = 0;                                                   // the row of entry v: last i with rowV[i] <= v
#pragma unroll
        for (int s = WAVE / 2; s >= 1; s >>= 1) {
          const int c = i + s;
          if (c < nr && L.rowV[c] <= v) i = c;
        }
        PreA pre{0, 0, 0.f, true};
        int region = 0;
        if (valid) {
          const int e = L.rowE[i] + (v - L.rowV[i]);
          const int2 sbl = SBL[e];
          pre.bs = sbl.x; pre.len = sbl.y;
          if (NEED_VAL) pre.a = VA[e];
          const int t0s = L.rowT[i];
          region = t0s | ((L.rowT[i + 1] - t0s) << 16);
        }
        const GroupLanes g = stage_group<NEED_VAL, 16, 0x40000000>(L.marks, reinterpret_cast<char*>(L.rec), valid ? 0 : 1, 1,
                                                                   SBL, VA, pre);
        const bool keep = g.len > 0;
        const unsigned long long km = ballot64(keep);
        if (keep) reinterpret_cast<int*>(L.rec)[mask_rank(km) * 4 + 2] = region;
        wave_lds_sync();
        if (g.T > WAVE * WAVE) { if (lane == 0) atomicOr(err, ERRF_CHAIN_LAYOUT); continue; }   // (cannot happen: T <= TT)
        const int nrnd = (g.T + WAVE - 1) >> 6;
        const unsigned mlo = (unsigned)g.mw, mhi = (unsigned)(g.mw >> 32);
        for (int q0 = 0; q0 < nrnd; q0 += MAXR) {
          with_rounds(min(MAXR, nrnd - q0), [&](auto rc) {
            constexpr int R = decltype(rc)::value;
            unsigned long long W[R];
            int base[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
              const int r = min(q0 + u, WAVE - 1);
              W[u] = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mlo, r) |
                     ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, r) << 32);
              base[u] = __builtin_amdgcn_readlane(g.pexcl, r);
            }
            short_trip_b<R, NEED_VAL>(reinterpret_cast<const char*>(L.rec), g.T, g.ns, q0, W, base, JB, VB,
                                      [&](const bool (&act)[R], const int (&col)[R], const float (&val)[R], const int (&reg)[R]) {
              hash_accum_regions<R>(L.tab, act, col, val, reg, &L.dummy[lane], err, TBL);
            });
          });
        }
        wave_lds_sync();
      }
    }
    STAMP(2)
    int2* const countedList = reinterpret_cast<int2*>(L.rec);           // the staging records are dead now
    if (MODE != WB_SYM && counted) countedList[mask_rank(cmask)] = make_int2(myT, cnt);
    // ---- occupancy of the table in slot order = row order: counts per 64-slot step, their prefix, ranks at the region
    // boundaries
    const int nsteps = bad ? 0 : (TT + WAVE - 1) >> 6;
    for (int g = 0; g < nsteps; ++g) {
      const unsigned long long mk = ballot64(slot_key(L.tab[g * WAVE + lane]) != EMPTY_KEY);
      if (lane == 0) L.stepMask[g] = mk;
    }
    wave_lds_sync();
    const int myc = lane < nsteps ? __popcll(L.stepMask[lane]) : 0;
    const int incS = wave_incl_add(myc);
    if (lane < nsteps) L.stepPref[lane] = incS - myc;
    const int total = __builtin_amdgcn_readlane(incS, 63);
    wave_lds_sync();
    auto rank_of = [&](int s) {
      return s >= TT ? total : L.stepPref[s >> 6] + __popcll(L.stepMask[s >> 6] & ((1ull << (s & 63)) - 1ull));
    };
    const int myRank = rank_of(myT);
    if (MODE == WB_SYM) {
      if (lane < nr && !counted) IC[r0 + lane] = rank_of(myT + ts) - myRank;
      for (int g = 0; g < nsteps; ++g) L.tab[g * WAVE + lane] = EMPTY_SLOT;
      wave_lds_sync();
      STAMP(3)
      continue;
    }
    // ---- where the batch starts in C
    unsigned base;
    if (MODE == WB_NUM) {
      base = (unsigned)(liveBatch && nr > 0 ? IC[r0] : 0);
    } else {
      if (lane == 0) sh.wagg[w] = (unsigned)(total + countedTotal);
      STAMP(3)
      __syncthreads();
      if (w == 0) {
        unsigned agg = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) agg += sh.wagg[i];
        const int gid = t0 / NW;
        const int ngroups = (NB + NW - 1) / NW;
        if (lane == 0) chain_store(chain + gid, (1ull << 32) | agg);
        const unsigned gbase = chain_lookback(chain, gid);
        if (lane == 0) {
          chain_store(chain + gid, (2ull << 32) | (unsigned long long)(gbase + agg));
          sh.base = gbase;
          if (gid == ngroups - 1) {
            const unsigned long long all = (unsigned long long)gbase + agg;
            *nnzC64 = all;
            IC[m] = (int)min(all, (unsigned long long)capC);           // (capC < 2^31)
          }
        }
      }
      STAMP(4)
      __syncthreads();
      base = sh.base;
#pragma unroll
      for (int i = 0; i < NW; ++i) base += i < w ? sh.wagg[i] : 0u;
      STAMP(5)
      // row pointers of the batch's rows, never beyond the capacity of C (the kernels of the rows above smallMax place
      // their entries by these offsets)
      if (lane < nr) IC[r0 + lane] = (int)min((long long)base + myRank + (inclC - cnt), capC);
    }
    // ---- the entries in slot order; the table is left empty.  Entries behind a row that is not accumulated here skip its range.
    bool over = false;
    for (int g = 0; g < nsteps; ++g) {
      const int s = g * WAVE + lane;
      const slot_t sv = L.tab[s];
      L.tab[s] = EMPTY_SLOT;
      const unsigned long long mk = L.stepMask[g];
      unsigned o = base + (unsigned)(L.stepPref[g] + mask_rank(mk));
      for (int k = 0; k < nc; ++k) { const int2 ck = countedList[k]; o += s >= ck.x ? (unsigned)ck.y : 0u; }
      if (slot_key(sv) != EMPTY_KEY) {
        if ((long long)o < capC) { st_out(JC + o, slot_key(sv)); st_out(C + o, slot_val(sv)); }
        else over = true;
      }
    }
    if (over) atomicOr(err, ERRF_CHAIN_CAP);
    wave_lds_sync();
    STAMP(6)
  }
#ifdef SMF_STAMPS
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_cstamps[i], st_[i]);
    atomicAdd(&g_cstamps[8], __builtin_readcyclecounter() - t0_);
    atomicAdd(&g_cstamps[9], 1ull);
  }
#endif
}

}  // namespace smf
