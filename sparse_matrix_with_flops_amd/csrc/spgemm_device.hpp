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
//   k_num_g16/k_num_hash   sgpu_SpGEMM_mid / fp1 / fp2 / fpl4 mindex2-cuda/gspgemm.cuh:2-293
//                          hashCASAdd2                       mindex2-cuda/casHash.cuh:34-43
//   k_num_big*             sgpu_SpGEMM_olarge (dense map)    "mindex2-cuda/\":143-213
//
// CDNA4 choices: 64-lane ballots / DPP scans instead of __syncthreads()-stepped sub-warp scans;
// products of one C row are flattened over all lanes (B rows of a power-law graph are short: a
// "lanes stride one B row" mapping leaves >90% of a wave idle); several rounds of products are kept
// in flight per wave (the path is latency-bound, not issue-bound); LDS tables sized per row; rows with
// more than 4096 products use either an LDS column bitmap + popcount ranks (n <= 262144: sorted output,
// no probing) or a multi-pass 128 KB LDS hash (any n) — 160 KB LDS per CU is what makes both possible.
// No MFMA: this is index/scatter work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace smf {

#ifdef SMF_ABLATE
// diagnostic build only (make ablate): pieces of the wave-per-row numeric kernel can be switched off at run time
// (SPGEMM_ABLATE bit mask: 1 no insert, 2 no gathers, 4 no compaction, 8 no walk) to see what its time is made of.
// The results are wrong by construction; never the product.
__device__ int g_ablate = 0;
#define ABL(bit) (g_ablate & (bit))
// error flags (int*) are not raised in this build: with pieces switched off every row "fails" and the flag word serialises
__device__ __forceinline__ void smf_or(int*, int) {}
__device__ __forceinline__ void smf_or(unsigned* p, unsigned v) { atomicOr(p, v); }
#define atomicOr(p, v) smf_or((p), (v))
#else
#define ABL(bit) 0
#endif

constexpr int WAVE = 64;

// C is written once and not read again by this call: its stores carry the non-temporal hint so that the output stream
// does not push B out of the caches the gathers live on
template <class T> __device__ __forceinline__ void st_out(T* p, T v) {
#ifdef SMF_PLAIN_STORES
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}
constexpr int NBINS = 9;  // {0 | 1 | 2-4 | 5-16 | 17-64 | 65-512 | 513-2048 | 2049-4096 | >4096}
constexpr int EMPTY_KEY = -1;

__host__ __device__ __forceinline__ int bin_of(unsigned long long f) {
  if (f == 0) return 0;
  if (f == 1) return 1;
  if (f <= 4) return 2;
  if (f <= 16) return 3;
  if (f <= 64) return 4;
  if (f <= 512) return 5;
  if (f <= 2048) return 6;
  if (f <= 4096) return 7;
  return 8;
}

// Layout slots of the row-id array (finer than the bins; a bin is one or more consecutive slots):
//   slots 0..4  = bins 0..4
//   slots 5, 6  = bin 5 split at 256 products: rows up to 256 products need a 512-slot table, so their wave-per-row
//                 kernel fits 24 blocks per CU instead of 16 (measured on the symbolic kernel: 16 -> 24 waves/CU is
//                 17 % faster, 32 is slower again)
//   slots 7, 8  = bin 6 split at 1024 products: a 2048-slot table lets 7 four-wave blocks share a CU instead of 4 (these
//                 kernels are bound by the serial chain of a row times the rows in flight, not by any unit's throughput)
//   slot  9     = bin 7
//   slots 10..15 = the last bin split into NSUB size classes (4097-8191, 8192-16383, ... by powers of two) laid out
//                 LARGEST FIRST, so that the work queue of the block-per-row kernels hands out the heavy rows first
//                 and the kernel does not end on one of them (measured: k_num_bighash 1.58 -> 1.48 ms at 1 M rows)
constexpr int NSUB = 6;
constexpr int SLOT_H1A = 5, SLOT_H1B = 6, SLOT_H4A = 7, SLOT_H4B = 8, SLOT_H8 = 9, SLOT_BIG0 = 10;
constexpr int NSLOTS = SLOT_BIG0 + NSUB;
constexpr int H1A_MAX = 256, H4A_MAX = 1024;
__host__ __device__ __forceinline__ int slot_of(unsigned long long f) {
  const int b = bin_of(f);
  if (b < 5) return b;
  if (b == 5) return f <= (unsigned)H1A_MAX ? SLOT_H1A : SLOT_H1B;
  if (b == 6) return f <= (unsigned)H4A_MAX ? SLOT_H4A : SLOT_H4B;
  if (b == 7) return SLOT_H8;
  int lg = 12;                                        // f >= 4097
  while (lg < 12 + NSUB - 1 && (f >> (lg + 1)) != 0) ++lg;
  return SLOT_BIG0 + (NSUB - 1 - (lg - 12));
}
// last slot of a bin
__host__ __device__ __forceinline__ int last_slot_of_bin(int b) {
  return b < 5 ? b : b == 5 ? SLOT_H1B : b == 6 ? SLOT_H4B : b == 7 ? SLOT_H8 : NSLOTS - 1;
}

// error flag bits written by kernels into Workspace::d_err
constexpr int ERRF_TABLE_FULL = 1;
constexpr int ERRF_COUNT_MISMATCH = 2;

// ------------------------------------------------------------------------------------------------
// wave64 primitives
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// lane mask of a predicate, straight from the compare (HIP's __ballot(int) goes through a 0/1 VGPR and back)
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

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
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // compiler ordering only: no vmcnt drain
  __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int next_pow2_clamped(int x, int lo, int hi) {
  int p = lo;
  while (p < x && p < hi) p <<= 1;
  return p;
}

__device__ __forceinline__ int log2_pow2(int p) { return 31 - __clz(p); }

// power-of-two table for `items` distinct keys at load <= 1/2, clamped to [lo, hi]; `items` may be INT_MAX (saturated flops)
// MULT = 4 (the wave-per-row and multi-wave hash kernels since round 4): a row that leaves room in its kernel's table gets
// four slots per item instead of two.  In the lock-step insert loops every lane pays for the longest probe chain of the
// trip, and the chains shorten with the load faster than the sweep of the larger table grows (measured: hash kernels -1 to
// -4 % each; the 16-lane kernels, which sweep 16 slots at a time, lose with it and stay at 2; R-MCL tables at HALF the
// size: +27 %).
template <int MULT = 2>
__device__ __forceinline__ int table_size(int items, int lo, int hi) {
  return items >= hi / MULT ? hi : next_pow2_clamped(MULT * items, lo, hi);
}

// ------------------------------------------------------------------------------------------------
// LDS open-addressing hash (keys >= 0, EMPTY_KEY = -1), multiplicative hash, linear probing.
// hash_insert: one key per lane (small-row kernels).  Returns the slot of `c`; *is_new is set when this call
// claimed the slot.  `size` is a power of two >= 2 * (number of distinct keys), so a probe sequence always
// terminates; a bounded loop and an error flag guard against corrupt inputs instead of hanging the GPU.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int hash_insert(int* keys, int size, int shift, int c, bool* is_new, int* err) {
  unsigned h = ((unsigned)c * 2654435761u) >> shift;
  const unsigned mask = (unsigned)size - 1u;
  *is_new = false;
  for (int probe = 0; probe < size; ++probe) {
    const int old = atomicCAS(&keys[h], EMPTY_KEY, c);
    if (old == EMPTY_KEY) { *is_new = true; return (int)h; }
    if (old == c) return (int)h;
    h = (h + 1u) & mask;
  }
  atomicOr(err, ERRF_TABLE_FULL);
  return 0;
}

// U keys per lane at once, straight-line code: predication is done through the ADDRESS (a lane with nothing to
// insert aims its CAS / add at a private dummy word), never through a branch, so the U LDS atomics of a probe
// step issue back to back and their latencies overlap; hipcc otherwise wraps every conditional atomic in its
// own exec-mask branch with an s_waitcnt right behind it.  CAS first: most products of a SpGEMM row touch their
// column for the first time, so the claim usually succeeds in one LDS round trip.
// vals == nullptr: symbolic (count only).  `dummy` = one private int per lane.  Returns the slots this lane claimed.
// POW2: size is a power of two (slot = top bits of the multiplicative hash); otherwise any size
// (slot = mulhi(hash, size)), which lets the big-row kernel use every byte of LDS it can get.
// The loop-carried state of the multi-insert loops is integers only: the byte offset of the slot a lane probes next,
// with the lane's private dummy word standing for "done".  (Carried bools live in VGPRs as 0/1 and cost two VALU
// instructions each per round to turn back into lane masks; these kernels are VALU-bound.)  Conditions used inside
// one round stay lane masks in SGPRs and combine on the scalar unit.
// Returns the number of new keys THIS LANE claimed (the caller reduces over the wave once per row).
// The dummy words are initialised to DUMMY_KEY (never EMPTY_KEY, never a column), so a CAS that lands there fails
// and a parked lane needs no masking when the new keys of a round are counted.
constexpr int DUMMY_KEY = (int)0x80000000;

template <bool POW2 = true, int U>
__device__ __forceinline__ int hash_insert_multi(int* keys, int size, int shift, const bool (&act)[U],
                                                 const int (&col)[U], int* dummy, int* err) {
  char* const base = reinterpret_cast<char*>(keys);
  const int dumB = (int)(reinterpret_cast<char*>(dummy) - base);
  const int maskB = size * 4 - 1;
  int hB[U];
  int claimed = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned hv = (unsigned)col[u] * 2654435761u;
    const unsigned h = POW2 ? hv >> shift : __umulhi(hv, (unsigned)size);
    hB[u] = act[u] ? (int)(h * 4u) : dumB;
  }
  int probe = 0;
  // Bookkeeping on the vector side (round 3; these kernels fill ~55 % of the SCALAR issue slots too): `claimed` is a
  // per-LANE count of the keys this lane claimed -- a compare and an add instead of ballot + s_bcnt1 + s_add per product
  // and probe round; the caller sums over the wave once per row.  "Any lane still probing" comes from one OR-ed word and
  // one compare instead of a ballot per product.  (k_sym_hash<4,4096> 0.264 -> 0.249 ms, <1,*> 0.223 -> 0.210 ms.)
  for (;;) {                                        // first round unconditional: nothing between the gathers and their use
    int old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) old[u] = atomicCAS(reinterpret_cast<int*>(base + hB[u]), EMPTY_KEY, col[u]);
    ++probe;
    int pendBits = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      claimed += old[u] == EMPTY_KEY ? 1 : 0;                               // the dummy is never EMPTY_KEY
      const bool adv = hB[u] != dumB && old[u] != EMPTY_KEY && old[u] != col[u];
      // power-of-two tables: triangular steps (+1, +2, +3, ...) visit every slot once and break up probe clusters
      const int nh = POW2 ? ((hB[u] + probe * 4) & maskB) : (hB[u] + 4 == size * 4 ? 0 : hB[u] + 4);
      hB[u] = adv ? nh : dumB;
      pendBits |= hB[u] ^ dumB;
    }
    if (ballot64(pendBits != 0) == 0ull) break;
    if (probe >= size) { atomicOr(err, ERRF_TABLE_FULL); break; }
  }
  return claimed;
}

// ------------------------------------------------------------------------------------------------
// Numeric accumulation.  Measured on gfx950 (tools/micro/lds_atomics.hip): ds_add_f32 costs ~3 cycles PER ACTIVE LANE
// (195 cycles for a full wave, and lanes parked on a dummy address count as active), while ds_cmpst b32/b64 and plain
// reads cost <= 26 cycles per wave instruction.  So the tables of the numeric kernels hold (key, value) PAIRS in one
// 64-bit word and a product that meets an empty slot deposits key AND value with a single 64-bit CAS -- on these
// matrices 9 out of 10 products are the first of their column.  Only a product that finds its key already present
// needs a float add, issued under the EXEC mask (cost = 3 cycles x the few lanes that need it).
// ------------------------------------------------------------------------------------------------
typedef unsigned long long slot_t;                       // low half: key (column), high half: value bits
constexpr slot_t EMPTY_SLOT = 0x00000000FFFFFFFFull;      // {EMPTY_KEY, 0.0f}
constexpr slot_t DUMMY_SLOT = 0x0000000080000000ull;      // {DUMMY_KEY, 0.0f}: what the dummy words hold
__device__ __forceinline__ slot_t make_slot(int col, float v) {
  return (slot_t)(unsigned)col | ((slot_t)__float_as_uint(v) << 32);
}
__device__ __forceinline__ int slot_key(slot_t sl) { return (int)(unsigned)sl; }
__device__ __forceinline__ float slot_val(slot_t sl) { return __uint_as_float((unsigned)(sl >> 32)); }
__device__ __forceinline__ float* slot_val_ptr(slot_t* tab, unsigned i) { return reinterpret_cast<float*>(tab + i) + 1; }

// one product per lane, divergent callers (small-row kernels)
__device__ __forceinline__ void hash_accum(slot_t* tab, int size, int shift, int c, float v, int* err) {
  unsigned h = ((unsigned)c * 2654435761u) >> shift;
  const unsigned mask = (unsigned)size - 1u;
  const slot_t mine = make_slot(c, v);
  for (int probe = 0; probe < size; ++probe) {
    const slot_t old = atomicCAS(&tab[h], EMPTY_SLOT, mine);
    if (old == EMPTY_SLOT) return;
    if (slot_key(old) == c) { atomicAdd(slot_val_ptr(tab, h), v); return; }
    h = (h + 1u) & mask;
  }
  atomicOr(err, ERRF_TABLE_FULL);
}

// U products per lane in lock step, wave-uniform control flow.  Lanes without work aim their CAS at a private
// 8-byte dummy (a CAS costs the same with any lane count; the float add below does not, hence its EXEC mask).
// (Round 3, measured and kept out: adding the repeated columns by a 64-bit CAS on the whole slot {key, value + v} instead of
// ds_add_f32 -- 3 cycles per active lane -- for rounds where most products repeat a column (nnz(C)/P ~ 0.5).  The slot
// content the insert CAS returned has to be carried per product: +2U VGPRs in every numeric kernel.  Web-graph surrogate:
// k_num_hash<1,*> 0.308 -> 0.478 ms; headline matrix 0.293 -> 0.366 ms.  profiles/README.md, round 3.)
template <bool POW2 = true, int U>
__device__ __forceinline__ void hash_accum_multi(slot_t* tab, int size, int shift, const bool (&act)[U],
                                                 const int (&col)[U], const float (&val)[U], slot_t* dummy, int* err) {
  char* const base = reinterpret_cast<char*>(tab);
  const int dumB = (int)(reinterpret_cast<char*>(dummy) - base);
  const int maskB = size * 8 - 1, sizeB = size * 8;
  int hB[U], stepB[U], dupB[U];                   // byte offsets; dumB = "done" / "no repeated column"
  slot_t mine[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const unsigned hv = (unsigned)col[u] * 2654435761u;
    const unsigned h = POW2 ? hv >> shift : __umulhi(hv, (unsigned)size);
    // tables of 1024*k slots (k <= 17): double hashing with a prime step > 17, coprime with every such size
    stepB[u] = POW2 ? 0 : (int)((0x2f2b29251f1d1713ull >> (((hv >> 7) & 7u) * 8u)) & 0xffu) * 8;
    hB[u] = act[u] ? (int)(h * 8u) : dumB;
    dupB[u] = dumB;
    mine[u] = make_slot(col[u], val[u]);
  }
  int probe = 0;
  for (;;) {                                        // first round unconditional: the col and value gathers stay together
    slot_t old[U];
#pragma unroll
    for (int u = 0; u < U; ++u) old[u] = atomicCAS(reinterpret_cast<slot_t*>(base + hB[u]), EMPTY_SLOT, mine[u]);
    ++probe;
    int pendBits = 0;                               // (one OR-ed word and one compare instead of a ballot per product)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool pend = hB[u] != dumB;
      const bool same = pend && slot_key(old[u]) == col[u];
      const bool adv = pend && old[u] != EMPTY_SLOT && slot_key(old[u]) != col[u];
      dupB[u] = same ? hB[u] : dupB[u];
      int nh;
      if (POW2) nh = (hB[u] + probe * 8) & maskB;            // triangular steps
      else { nh = hB[u] + stepB[u]; nh = nh >= sizeB ? nh - sizeB : nh; }
      hB[u] = adv ? nh : dumB;
      pendBits |= hB[u] ^ dumB;
    }
    if (ballot64(pendBits != 0) == 0ull) break;
    if (probe >= size) { atomicOr(err, ERRF_TABLE_FULL); break; }
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (dupB[u] != dumB) atomicAdd(reinterpret_cast<float*>(base + dupB[u] + 4), val[u]);
}

// float add into an LDS word by read + 32-bit CAS (retry on interference): ~2 cheap LDS ops instead of a
// ds_add_f32 over a full wave.  U per lane in lock step; lanes without work aim at a private dummy.
template <int U>
__device__ __forceinline__ void lds_fadd_multi(float* acc, const int (&idx)[U], const bool (&ok)[U], const float (&v)[U],
                                               int* dummy) {
  int* ia = reinterpret_cast<int*>(acc);
  int old[U];
  bool pend[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { pend[u] = ok[u]; old[u] = ia[ok[u] ? idx[u] : 0]; }
  for (;;) {
    int got[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      got[u] = atomicCAS(pend[u] ? &ia[idx[u]] : dummy, old[u], __float_as_int(__int_as_float(old[u]) + v[u]));
    bool more = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pend[u] = pend[u] && got[u] != old[u];
      old[u] = got[u];
      more = more || pend[u];
    }
    if (ballot64(more) == 0ull) break;
  }
}

// The R rounds of a trip (gathered together: one memory round trip) are INSERTED CH rounds at a time: in a lock-step probe
// loop every round of the trip issues its CAS in every iteration until the longest chain among ALL its products has ended
// (R = 8: the longest of 512 chains), so R/CH short loops issue fewer instructions than one long one, and the first of
// them starts as soon as ITS gathers have landed.  Round 4, medians of 4 processes per arm: the wave-per-row kernels (R up
// to 8) with CH = 4 / 2 / 1: symbolic 0.195 -> 0.189 / 0.177 / 0.172 ms, numeric 0.304 -> 0.298 / 0.289 / 0.284 (web
// surrogate numeric 0.296 -> 0.280 at CH = 1); the multi-wave kernels (trips of 2 rounds) LOSE 2-4 % with CH = 1 and
// keep their one loop.  CH <= 0: the whole trip in one loop.
// (trips of 4 rounds inserted 2 at a time in the multi-wave kernels: no change; of 8: slower.)
template <int NW> struct InsChunk { static constexpr int value = NW == 1 ? 1 : 0; };
template <int CH, int R>
__device__ __forceinline__ int hash_insert_chunked(int* keys, int size, int shift, const bool (&act)[R], const int (&col)[R],
                                                   int* dummy, int* err) {
  if constexpr (CH <= 0 || CH >= R) return hash_insert_multi(keys, size, shift, act, col, dummy, err);
  else {
    int claimed = 0;
#pragma unroll
    for (int c0 = 0; c0 < R; c0 += CH) {
      bool a_[CH]; int c_[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) { a_[u] = c0 + u < R ? act[c0 + u < R ? c0 + u : 0] : false; c_[u] = col[c0 + u < R ? c0 + u : 0]; }
      if (CH == 1 && ballot64(a_[0]) == 0ull) continue;          // a round past the end of the group (5 rounds run as 6, 7 as 8)
      claimed += hash_insert_multi(keys, size, shift, a_, c_, dummy, err);
    }
    return claimed;
  }
}
template <int CH, int R>
__device__ __forceinline__ void hash_accum_chunked(slot_t* tab, int size, int shift, const bool (&act)[R], const int (&col)[R],
                                                   const float (&val)[R], slot_t* dummy, int* err) {
  if constexpr (CH <= 0 || CH >= R) hash_accum_multi(tab, size, shift, act, col, val, dummy, err);
  else {
#pragma unroll
    for (int c0 = 0; c0 < R; c0 += CH) {
      bool a_[CH]; int c_[CH]; float v_[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int i = c0 + u < R ? c0 + u : 0;
        a_[u] = c0 + u < R ? act[i] : false; c_[u] = col[i]; v_[u] = val[i];
      }
      if (CH == 1 && ballot64(a_[0]) == 0ull) continue;
      hash_accum_multi(tab, size, shift, a_, c_, v_, dummy, err);
    }
  }
}

// slot-table clear, two slots per lane and instruction
__device__ __forceinline__ void clear_slots(slot_t* tab, int size, int tid, int nthreads) {
  for (int i = tid * 2; i < size; i += nthreads * 2) *reinterpret_cast<ulonglong2*>(tab + i) = make_ulonglong2(EMPTY_SLOT, EMPTY_SLOT);
}

// table clear, four slots per lane and instruction (sizes are multiples of 4; arrays 16-byte aligned)
__device__ __forceinline__ void clear_table(int* keys, float* vals, int size, int tid, int nthreads) {
  const int4 ek = make_int4(EMPTY_KEY, EMPTY_KEY, EMPTY_KEY, EMPTY_KEY);
  const float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid * 4; i < size; i += nthreads * 4) {
    *reinterpret_cast<int4*>(keys + i) = ek;
    if (vals) *reinterpret_cast<float4*>(vals + i) = zv;
  }
}

// One wave emits the occupied slots [base, base+per) of a finished table: slots into registers, one LDS atomic for
// the wave's share of the output range [outLo, outHi), stores from the registers (64 consecutive positions a step).
template <int MAXSTEPS>
__device__ __forceinline__ void emit_claimed(const slot_t* tab, int base, int per, int* emitted,
                                             int outLo, int outHi, int* __restrict__ JC, float* __restrict__ C) {
  const int lane = lane_id();
  slot_t sl[MAXSTEPS];
  int cnt = 0;
#pragma unroll
  for (int sidx = 0; sidx < MAXSTEPS; ++sidx) {
    sl[sidx] = sidx * WAVE < per ? tab[base + sidx * WAVE + lane] : EMPTY_SLOT;
    cnt += __popcll(ballot64(slot_key(sl[sidx]) != EMPTY_KEY));
  }
  int pos = 0;
  if (lane == 0) pos = atomicAdd(emitted, cnt);
  pos = __builtin_amdgcn_readfirstlane(pos);
  // positions relative to the row: one small 32-bit offset serves both output arrays (uniform base + offset stores)
  int* const JCrow = JC + outLo;
  float* const Crow = C + outLo;
  const unsigned lim = (unsigned)(outHi - outLo);
#pragma unroll
  for (int sidx = 0; sidx < MAXSTEPS; ++sidx) {
    const bool occ = slot_key(sl[sidx]) != EMPTY_KEY;
    const unsigned long long mk = ballot64(occ);
    const unsigned o = (unsigned)(pos + mask_rank(mk));
    if (occ && o < lim) { st_out(JCrow + o, slot_key(sl[sidx])); st_out(Crow + o, slot_val(sl[sidx])); }
    pos += __popcll(mk);
  }
}

// ------------------------------------------------------------------------------------------------
// R-MCL row rule (CPU: nlibs/tools/util.cc:4-69; reference GPU: nlibs/gpus/dutil.cuh:8-80): inflate (square), max,
// sum, thresh = clamp(0.9*avg*(1-2(max-avg)), 1e-7, max), keep v >= thresh, divide the kept values by their sum.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rmcl_threshold(float avg, float mx) {
  float ret = (float)(0.90 * avg * (1 - 2 * (mx - avg)));       // same promotions as computeThreshold (util.cc:4-9)
  ret = (float)((ret > 1.0e-7) ? ret : 1.0e-7);
  ret = (ret > mx) ? mx : ret;
  return ret;
}

// all-reduce inside the L lanes that share a row (L = 16: one DPP row; L = 64: the wave).  DPP butterflies (xor 1, xor 2,
// half-row mirror, row mirror): four VALU ops for 16 lanes and no LDS round trip -- a ds_bpermute chain here costs more
// than the whole product walk of a 100-product row.  64 lanes: the four row results are combined in a fixed order.
template <class Op>
__device__ __forceinline__ float row16_allreduce(float v, Op op) {
  v = op(v, __int_as_float(SMF_DPP(__float_as_int(v), 0xB1, 0xf, 0)));    // quad_perm [1,0,3,2]
  v = op(v, __int_as_float(SMF_DPP(__float_as_int(v), 0x4E, 0xf, 0)));    // quad_perm [2,3,0,1]
  v = op(v, __int_as_float(SMF_DPP(__float_as_int(v), 0x141, 0xf, 0)));   // row_half_mirror
  v = op(v, __int_as_float(SMF_DPP(__float_as_int(v), 0x140, 0xf, 0)));   // row_mirror
  return v;
}
template <int L, class Op>
__device__ __forceinline__ float rowL_allreduce(float v, Op op) {
  static_assert(L == 16 || L == 64, "16-lane groups or whole waves");
  v = row16_allreduce(v, op);
  if (L == 64) {
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    v = op(op(r0, r1), op(r2, r3));
  }
  return v;
}
template <int L> __device__ __forceinline__ float rowL_sum(float v) { return rowL_allreduce<L>(v, [](float a, float b) { return a + b; }); }
template <int L> __device__ __forceinline__ float rowL_max(float v) { return rowL_allreduce<L>(v, [](float a, float b) { return fmaxf(a, b); }); }
// this group's 16 (or all 64) bits of a wave-wide lane mask
template <int L>
__device__ __forceinline__ unsigned long long group_mask(unsigned long long mk, int gl) {
  return L == 64 ? mk : ((mk >> (lane_id() - gl)) & 0xffffull);
}

// Fused expansion + prune (hip_rmcl_expand_prune): the row rule applied to a finished row that still sits in LDS, so
// that only the KEPT entries (about a quarter of an R-MCL product) ever reach HBM.  They are written, normalised, to
// the FRONT of the row's range of the scratch arrays, their count to cnt[row]; k_rmcl_move then packs the rows.
// The slots hold either a hash table (occupied = key set) or the plain list of the row's `want` products (expanded rows).
// L lanes share the row (a 16-lane group or a whole wave) and sweep slots [0, per), per a multiple of L.
// want < 0: the number of distinct columns is not known beforehand (no symbolic pass ran), the occupied slots are counted.
template <int L>
__device__ __forceinline__ int prune_emit_group(const slot_t* tab, int per, int want, bool byKey, bool live, int gl,
                                                int* __restrict__ JCrow, float* __restrict__ Crow, int* nocc) {
  float mx = 0.f, sum = 0.f;
  int n = 0;
  for (int i0 = 0; i0 < per; i0 += L) {
    const slot_t sv = tab[i0 + gl];
    const bool occ = live && (byKey ? slot_key(sv) != EMPTY_KEY : i0 + gl < want);
    const float c = slot_val(sv), v = occ ? c * c : 0.f;
    mx = fmaxf(mx, v);
    sum += v;
    n += __popcll(group_mask<L>(ballot64(occ), gl));
  }
  mx = rowL_max<L>(mx);
  sum = rowL_sum<L>(sum);
  *nocc = n;
  const float th = rmcl_threshold(sum / (float)(want >= 0 ? want : n), mx);
  float ks = 0.f;
  for (int i0 = 0; i0 < per; i0 += L) {
    const slot_t sv = tab[i0 + gl];
    const bool occ = live && (byKey ? slot_key(sv) != EMPTY_KEY : i0 + gl < want);
    const float c = slot_val(sv), v = c * c;
    ks += (occ && v >= th) ? v : 0.f;
  }
  ks = rowL_sum<L>(ks);
  const float inv = 1.0f / ks;                        // one division per row; v*inv is within an ulp of v/ks
  int pos = 0;
  for (int i0 = 0; i0 < per; i0 += L) {
    const slot_t sv = tab[i0 + gl];
    const bool occ = live && (byKey ? slot_key(sv) != EMPTY_KEY : i0 + gl < want);
    const float c = slot_val(sv), v = c * c;
    const bool keep = occ && v >= th;
    const unsigned long long gm = group_mask<L>(ballot64(keep), gl);
    if (keep) {
      const int o = pos + __popcll(gm & ((1ull << gl) - 1ull));
      st_out(JCrow + o, slot_key(sv));
      st_out(Crow + o, v * inv);
    }
    pos += __popcll(gm);
  }
  return pos;
}

// the same for a block of NW waves whose wave w owns slots [w*per, w*per+per) of a hash table: slots into registers,
// two block reductions (partials summed in wave order by every thread: the result does not depend on timing)
template <int NW> struct PruneShared { float mx[NW], sum[NW], ks[NW]; int occ[NW], kc[NW]; };

template <int NW, int MAXSTEPS>
__device__ __forceinline__ void prune_emit_block(const slot_t* tab, int per, int want, PruneShared<NW>& ps,
                                                 int* __restrict__ JCrow, float* __restrict__ Crow,
                                                 int* __restrict__ cntRow, int* __restrict__ err) {
  const int lane = lane_id(), w = threadIdx.x >> 6;
  slot_t sl[MAXSTEPS];
  float v[MAXSTEPS];                                  // squared value; -1 marks an empty slot (never >= a threshold)
  float mx = 0.f, sum = 0.f;
  int n = 0;
#pragma unroll
  for (int sidx = 0; sidx < MAXSTEPS; ++sidx) {
    sl[sidx] = sidx * WAVE < per ? tab[w * per + sidx * WAVE + lane] : EMPTY_SLOT;
    const bool occ = slot_key(sl[sidx]) != EMPTY_KEY;
    const float c = slot_val(sl[sidx]);
    v[sidx] = occ ? c * c : -1.f;
    mx = fmaxf(mx, v[sidx]);
    sum += occ ? v[sidx] : 0.f;
    n += __popcll(ballot64(occ));
  }
  mx = rowL_max<64>(mx);
  sum = rowL_sum<64>(sum);
  if (lane == 0) { ps.mx[w] = mx; ps.sum[w] = sum; ps.occ[w] = n; }
  __syncthreads();
  mx = 0.f; sum = 0.f; n = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) { mx = fmaxf(mx, ps.mx[i]); sum += ps.sum[i]; n += ps.occ[i]; }
  const float th = rmcl_threshold(sum / (float)(want >= 0 ? want : n), mx);
  float ks = 0.f;
  int kc = 0;
#pragma unroll
  for (int sidx = 0; sidx < MAXSTEPS; ++sidx) {
    const bool keep = v[sidx] >= th;
    ks += keep ? v[sidx] : 0.f;
    kc += __popcll(ballot64(keep));
  }
  ks = rowL_sum<64>(ks);
  if (lane == 0) { ps.ks[w] = ks; ps.kc[w] = kc; }
  __syncthreads();
  ks = 0.f;
  int pos = 0, total = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) { ks += ps.ks[i]; pos += i < w ? ps.kc[i] : 0; total += ps.kc[i]; }
  const float inv = 1.0f / ks;
#pragma unroll
  for (int sidx = 0; sidx < MAXSTEPS; ++sidx) {
    const bool keep = v[sidx] >= th;
    const unsigned long long mk = ballot64(keep);
    if (keep) {
      const int o = pos + mask_rank(mk);
      st_out(JCrow + o, slot_key(sl[sidx]));
      st_out(Crow + o, v[sidx] * inv);
    }
    pos += __popcll(mk);
  }
  if (threadIdx.x == 0) {
    *cntRow = total;
    if (want >= 0 && n != want) atomicOr(err, ERRF_COUNT_MISMATCH);
  }
}

// ------------------------------------------------------------------------------------------------
// K1  per-row product count + bin id + per-block bin histogram            (mindex2: gcomputeFlops)
// One wave owns 64 consecutive rows.  Phase 1: every lane walks the first FL_SHORT entries of its own
// row (87% of the rows of a power-law matrix end there).  Phase 2: rows that are longer are finished by
// the whole wave, 64 entries per step, so a 4095-entry row costs 64 steps, not 4095.
// ------------------------------------------------------------------------------------------------
constexpr int K1_THREADS = 256;
constexpr int FL_SHORT = 8;

// K1a: per A entry, where its B row starts and how long it is -- ONE coalesced record per entry.  Every later kernel
// stages a row from these records instead of walking the dependent chain JA -> IB (two memory round trips per row in
// every symbolic and numeric kernel, and three random IB gathers per entry over the whole pipeline instead of one).
__global__ __launch_bounds__(256) void k_entry_lens(int nnzA, const int* __restrict__ JA, const int* __restrict__ IB,
                                                     int2* __restrict__ SBL) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p < nnzA; p += gridDim.x * 256) {
    const int j = JA[p];
    const int2 be = make_int2(IB[j], IB[j + 1]);
    SBL[p] = make_int2(be.x, max(be.y - be.x, 0));
  }
}

// inclusive +-scan of 64-bit values over the wave (both halves travel through the same DPP moves)
__device__ __forceinline__ unsigned long long wave_incl_add_u64(unsigned long long v) {
#define SMF_DPP64_STEP(ctrl, rowmask)                                                                          \
  {                                                                                                             \
    const unsigned lo = (unsigned)SMF_DPP((int)(unsigned)v, ctrl, rowmask, 0);                                   \
    const unsigned hi = (unsigned)SMF_DPP((int)(unsigned)(v >> 32), ctrl, rowmask, 0);                           \
    v += ((unsigned long long)hi << 32) | lo;                                                                   \
  }
  SMF_DPP64_STEP(0x111, 0xf) SMF_DPP64_STEP(0x112, 0xf) SMF_DPP64_STEP(0x114, 0xf) SMF_DPP64_STEP(0x118, 0xf)
  SMF_DPP64_STEP(0x142, 0xa) SMF_DPP64_STEP(0x143, 0xc)
#undef SMF_DPP64_STEP
  return v;
}

// K1: one wave owns 64 consecutive rows = ONE contiguous range of A entries.  The wave walks that range 64 entries at a
// time, fully coalesced: entry -> column -> {B-row start, length} (the record of K1a, written here when SBL != nullptr:
// the two kernels are one), an inclusive scan of the 64 lengths, and every row takes the difference of the scan values
// at its two ends (two cross-lane reads).  No per-lane walk of its own row (64 different lines per load) and no serial
// finishing of long rows.  CHK chunks are in flight together (their JA -> IB chains are independent).
// EXTENTS: B is NOT packed -- row j is [IBse[j].x, IBse[j].y) (the pruned matrix of the previous R-MCL
// iteration, left where the epilogues wrote it: hip_gpuRmclIter_device; k_zip_extents makes the pairs so that the
// gather stays ONE 8-byte load per entry).  Everything downstream reads the records.
template <bool EXTENTS>
__global__ __launch_bounds__(K1_THREADS) void k_row_flops(
    int m, const int* __restrict__ IA, const int* __restrict__ JA, const int* __restrict__ IB,
    const int2* __restrict__ IBse, int2* __restrict__ SBL, int sblCap,
    int* __restrict__ rowFlops, unsigned char* __restrict__ binId, int* __restrict__ blockHist,
    unsigned long long* __restrict__ blockP, int* __restrict__ IC) {
  __shared__ int hist[NSLOTS];
  __shared__ unsigned long long psum;
  const int tid = threadIdx.x, lane = lane_id();
  if (tid < NSLOTS) hist[tid] = 0;
  if (tid == 0) psum = 0;
  __syncthreads();
  const int r = blockIdx.x * K1_THREADS + tid;
  const int rs = IA[min(r, m)], re = IA[min(r + 1, m)];      // rows past the end: empty, at the end of the range
  const int s = __builtin_amdgcn_readlane(rs, 0), e = __builtin_amdgcn_readlane(re, 63);
  unsigned long long f = 0;
  constexpr int CHK = 4;
  for (int c0 = s; c0 < e; c0 += CHK * WAVE) {
    int j[CHK];
    int2 be[CHK];
#pragma unroll
    for (int i = 0; i < CHK; ++i) { const int p = c0 + i * WAVE + lane; j[i] = p < e ? JA[p] : -1; }
#pragma unroll
    for (int i = 0; i < CHK; ++i) {
      if (EXTENTS) be[i] = j[i] >= 0 ? IBse[j[i]] : make_int2(0, 0);
      else be[i] = j[i] >= 0 ? make_int2(IB[j[i]], IB[j[i] + 1]) : make_int2(0, 0);
    }
#pragma unroll
    for (int i = 0; i < CHK; ++i) {
      const int cb = c0 + i * WAVE;                           // wave-uniform
      if (cb >= e) break;
      const int len = max(be[i].y - be[i].x, 0);
      if (SBL && j[i] >= 0 && cb + lane < sblCap) SBL[cb + lane] = make_int2(be[i].x, len);
      const unsigned long long incl = wave_incl_add_u64((unsigned long long)(unsigned)len);
      const int a = min(max(rs - cb, 0), WAVE), b = min(max(re - cb, 0), WAVE);   // this row's part of the chunk: [a, b)
      const unsigned long long hiP = __shfl(incl, max(b - 1, 0), WAVE);
      const unsigned long long loP = __shfl(incl, max(a - 1, 0), WAVE);
      if (b > a) f += hiP - (a > 0 ? loP : 0ull);
    }
  }
  int b = -1;
  if (r < m) {
    rowFlops[r] = f > 0x7fffffffULL ? 0x7fffffff : (int)f;
    b = slot_of(f);
    binId[r] = (unsigned char)b;
    if (b <= 1) IC[r] = b;                           // 0 products -> 0 entries, 1 product -> 1 entry
  }
#pragma unroll
  for (int q = 0; q < SLOT_BIG0; ++q) {
    const unsigned long long mk = ballot64(b == q);
    if (lane == 0 && mk) atomicAdd(&hist[q], __popcll(mk));
  }
  if (ballot64(b >= SLOT_BIG0)) {                    // big rows are rare: most waves skip their size classes
    for (int q = SLOT_BIG0; q < NSLOTS; ++q) {
      const unsigned long long mk = ballot64(b == q);
      if (lane == 0 && mk) atomicAdd(&hist[q], __popcll(mk));
    }
  }
  const unsigned long long wsum = wave_sum_u64(f);
  if (lane == 0 && wsum) atomicAdd(&psum, wsum);
  __syncthreads();
  if (tid < NSLOTS) blockHist[(size_t)tid * gridDim.x + blockIdx.x] = hist[tid];   // slot-major
  if (tid == 0) blockP[blockIdx.x] = psum;      // no same-address global atomics: k_bin_scan sums these
}

// ------------------------------------------------------------------------------------------------
// K2  exclusive scan of the per-block histograms in (bin-major, block-minor) order -> where each
//     block writes its rows of each layout slot; binPtr[NBINS+1].  One 1024-thread block.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_bin_scan(int nblk, const int* __restrict__ blockHist,
                                                    int* __restrict__ blockOff, int* __restrict__ binPtr,
                                                    int* __restrict__ slotBase,
                                                    const unsigned long long* __restrict__ blockP,
                                                    unsigned long long* __restrict__ totalP) {
  static_assert(NSLOTS <= 16, "one wave of the 1024-thread block per layout slot");
  __shared__ int slotTot[16];
  __shared__ unsigned long long ptot;
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  if (tid == 0) ptot = 0;
  __syncthreads();
  {
    unsigned long long p = 0;
    for (int b = tid; b < nblk; b += 1024) p += blockP[b];
    p = wave_sum_u64(p);
    if (lane == 0 && p) atomicAdd(&ptot, p);
  }
  // wave w scans the counts of slot w over the blocks (slot-major arrays: coalesced, no barriers inside)
  if (w < NSLOTS) {
    const int* src = blockHist + (size_t)w * nblk;
    int* dst = blockOff + (size_t)w * nblk;
    int run = 0;
    for (int t0 = 0; t0 < nblk; t0 += 4 * WAVE) {
      int v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int blk = t0 + i * WAVE + lane; v[i] = blk < nblk ? src[blk] : 0; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int blk = t0 + i * WAVE + lane;
        const int incl = wave_incl_add(v[i]);
        if (blk < nblk) dst[blk] = run + incl - v[i];
        run += __shfl(incl, 63, 64);
      }
    }
    if (lane == 0) slotTot[w] = run;
  }
  __syncthreads();
  if (tid == 0) {
    *totalP = ptot;
    int run = 0;
    binPtr[0] = 0;
    for (int sl = 0; sl < NSLOTS; ++sl) { slotBase[sl] = run; run += slotTot[sl]; }
    slotBase[NSLOTS] = run;
    for (int b = 0; b < NBINS; ++b) binPtr[b + 1] = slotBase[last_slot_of_bin(b) + 1];
  }
}

// ------------------------------------------------------------------------------------------------
// K3  stable scatter of row ids into their layout slots (rows ascend inside a bin / a size class of the last bin).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(K1_THREADS) void k_scatter_rows(int m, const unsigned char* __restrict__ binId,
                                                             const int* __restrict__ blockOff,
                                                             const int* __restrict__ slotBase,
                                                             int* __restrict__ rowIds) {
  __shared__ int wcnt[K1_THREADS / WAVE][NSLOTS];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int r = blockIdx.x * K1_THREADS + tid;
  const int b = r < m ? (int)binId[r] : -1;
  int myrank = 0;
#pragma unroll
  for (int q = 0; q < NSLOTS; ++q) {
    const unsigned long long mk = ballot64(b == q);
    if (b == q) myrank = mask_rank(mk);
    if (lane == 0) wcnt[w][q] = __popcll(mk);
  }
  __syncthreads();
  if (b >= 0) {
    int off = slotBase[b] + blockOff[(size_t)b * gridDim.x + blockIdx.x];
    for (int i = 0; i < w; ++i) off += wcnt[i][b];
    rowIds[off + myrank] = r;
  }
}

// Launch grids of the statically scheduled kernels are multiples of 8 (host: grid8()).
struct XcdRange { int lo, hi, bi, nb; };
__device__ __forceinline__ XcdRange xcd_range(int count) {
  const int x = blockIdx.x & 7;
  XcdRange r;
  r.bi = blockIdx.x >> 3;
  r.nb = gridDim.x >> 3;
  r.lo = (int)((long long)count * x >> 3);
  r.hi = (int)((long long)count * (x + 1) >> 3);
  return r;
}

// ------------------------------------------------------------------------------------------------
// Rows with 1..64 products: 16 lanes per row (4 rows per wave), products flattened over the 16 lanes.
// The old "A entries one after the other" walk costs three dependent memory round trips per A entry;
// here a row costs three in total (JA -> IB -> JB/VB): the row's A entries are staged 16 at a time with a
// 16-lane DPP scan of their B-row lengths, then 16-product rounds find their entry by a 4-step search.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int row16_incl_add(int v) {   // inclusive scan inside each 16-lane DPP row
  v += SMF_DPP(v, 0x111, 0xf, 0);
  v += SMF_DPP(v, 0x112, 0xf, 0);
  v += SMF_DPP(v, 0x114, 0xf, 0);
  v += SMF_DPP(v, 0x118, 0xf, 0);
  return v;
}

struct G16Stage { int incl[16]; int off[16]; float aval[16]; };

template <int U, bool NEED_VAL, class F>
__device__ __forceinline__ void g16_walk(G16Stage& st, int gl, int as, int ae, const int2* __restrict__ SBL,
                                         const float* __restrict__ VA,
                                         const int* __restrict__ JB, const float* __restrict__ VB, F&& f) {
  int rowBase = 0;                                  // products of the chunks before this one
  for (int chunk = as; chunk < ae; chunk += 16) {
    const int ap = chunk + gl;
    int len = 0, bs = 0;
    float a = 0.f;
    if (ap < ae) {
      const int2 sbl = SBL[ap];
      bs = sbl.x;
      len = sbl.y;
      if (NEED_VAL) a = VA[ap];
    }
    const int incl = row16_incl_add(len);
    st.incl[gl] = incl;
    st.off[gl] = bs - (incl - len);
    if (NEED_VAL) st.aval[gl] = a;
    wave_lds_sync();
    const int T = st.incl[15];
    for (int r0 = 0; r0 * 16 < T; r0 += U) {
      int p[U], e[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { p[u] = min((r0 + u) * 16 + gl, T - 1); e[u] = 0; }   // past the end: shadow the last product
#pragma unroll
      for (int sft = 8; sft >= 1; sft >>= 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int c = e[u] + sft;
          e[u] = st.incl[c - 1] <= p[u] ? c : e[u];
        }
      }
      int col[U];
      float val[U], av[U], vb[U];
      bool act[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {                // straight-line, no branches around the gathers
        act[u] = (r0 + u) * 16 + gl < T;
        const int jb = st.off[e[u]] + p[u];
        col[u] = JB[jb];
        vb[u] = NEED_VAL ? VB[jb] : 0.f;
        av[u] = NEED_VAL ? st.aval[e[u]] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);           // all gathers issued before the first wait
#pragma unroll
      for (int u = 0; u < U; ++u) val[u] = av[u] * vb[u];
#pragma unroll
      for (int u = 0; u < U; ++u) f(act[u], col[u], val[u], rowBase + (r0 + u) * 16 + gl);
    }
    rowBase += T;
    wave_lds_sync();
  }
}

template <int TBL, int U>
__global__ __launch_bounds__(256) void k_sym_g16(const int* __restrict__ binPtr, int bin, int binHi,
                                                  const int* __restrict__ rowIds,
                                                  const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                  const int* __restrict__ JB,
                                                  const int* __restrict__ rowFlops, int* __restrict__ IC,
                                                  int* __restrict__ err) {
  __shared__ int keys[16][TBL];
  __shared__ G16Stage st[16];
  const int tid = threadIdx.x, g = tid >> 4, gl = tid & 15;
  const int first = binPtr[bin], count = binPtr[binHi] - first;
  const int iters = (count + 15) / 16;
  // XCD-contiguous schedule: the blocks that share an XCD (blockIdx % 8, guide "Workgroup dispatch") walk ONE contiguous
  // eighth of the bin, so that the B rows its A rows have in common are fetched into that XCD's L2 once, not into all eight
  const XcdRange xr = xcd_range(iters);
  for (int it = xr.lo + xr.bi; it < xr.hi; it += xr.nb) {
    const int q = it * 16 + g;
    const bool live = q < count;
    const int row = live ? rowIds[first + q] : 0;
    const int F = live ? rowFlops[row] : 1;
    const int size = table_size(F, 16, TBL);
    const int shift = 32 - log2_pow2(size);
    for (int i = gl; i < size; i += 16) keys[g][i] = EMPTY_KEY;
    wave_lds_sync();
    int mine = 0;
    if (live) {
      g16_walk<U, false>(st[g], gl, IA[row], IA[row + 1], SBL, nullptr, JB, nullptr, [&](bool active, int col, float, int) {
        if (active) {
          bool isnew;
          hash_insert(keys[g], size, shift, col, &isnew, err);
          mine += isnew ? 1 : 0;
        }
      });
    }
    mine = row16_incl_add(mine);
    if (live && gl == 15) IC[row] = mine;
    wave_lds_sync();
  }
}

// PRUNE: 0 plain product; 1 fused R-MCL prune, IC = exact row pointers of the product (symbolic pass ran); 2 fused prune
// WITHOUT a symbolic pass: IC = prefix sums of the rows' product counts (an upper bound of every row; rows of bin 8, which
// these kernels do not see, contribute their exact counts), every row is hashed
// in a table sized by its products and the distinct columns are counted by the epilogue itself.
template <int TBL, int U, int PRUNE = 0>
__global__ __launch_bounds__(256) void k_num_g16(const int* __restrict__ binPtr, int bin, int binHi,
                                                  const int* __restrict__ rowIds,
                                                  const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                  const float* __restrict__ VA,
                                                  const int* __restrict__ JB,
                                                  const float* __restrict__ VB,
                                                  const int* __restrict__ IC, int* __restrict__ JC,
                                                  float* __restrict__ C, int* __restrict__ err,
                                                  const int* __restrict__ rowFlops, int* __restrict__ pcnt) {
  __shared__ slot_t tab[16][TBL];              // (column, value) pairs
  __shared__ G16Stage st[16];
  const int tid = threadIdx.x, g = tid >> 4, gl = tid & 15;
  const int first = binPtr[bin], count = binPtr[binHi] - first;
  const int iters = (count + 15) / 16;
  // XCD-contiguous schedule: the blocks that share an XCD (blockIdx % 8, guide "Workgroup dispatch") walk ONE contiguous
  // eighth of the bin, so that the B rows its A rows have in common are fetched into that XCD's L2 once, not into all eight
  const XcdRange xr = xcd_range(iters);
  for (int it = xr.lo + xr.bi; it < xr.hi; it += xr.nb) {
    const int q = it * 16 + g;
    const bool live = q < count;
    const int row = live ? rowIds[first + q] : 0;
    const int off = live ? IC[row] : 0;
    const int want = live ? IC[row + 1] - off : 0;
    // rows whose products all hit different columns (want == flops: 93-99 % of the rows this small on a power-law
    // matrix) are EXPANDED: products go straight to their position in the row, no table, no compaction.  The four
    // rows of a wave take the same path (wave-uniform control flow).
    const bool hashRow = live && (PRUNE == 2 || want != rowFlops[row]);
    if (ballot64(hashRow) == 0ull) {
      if (PRUNE) {                                   // the row's products as a plain list in LDS, then the row rule
        if (live) {
          g16_walk<U, true>(st[g], gl, IA[row], IA[row + 1], SBL, VA, JB, VB, [&](bool active, int col, float v, int pos) {
            if (active && (unsigned)pos < (unsigned)min(want, TBL)) tab[g][pos] = make_slot(col, v);
          });
        }
        wave_lds_sync();
        int nocc;
        const int kept = prune_emit_group<16>(tab[g], (min(want, TBL) + 15) & ~15, min(want, TBL), false, live, gl,
                                              JC + off, C + off, &nocc);
        if (live && gl == 0) { pcnt[row] = kept; if (want > TBL) atomicOr(err, ERRF_TABLE_FULL); }
        wave_lds_sync();
        continue;
      }
      if (live) {
        g16_walk<U, true>(st[g], gl, IA[row], IA[row + 1], SBL, VA, JB, VB, [&](bool active, int col, float v, int pos) {
          if (active && (unsigned)pos < (unsigned)want) { st_out(JC + off + pos, col); st_out(C + off + pos, v); }
        });
      }
      continue;
    }
    const int size = table_size(want, 16, TBL);
    const int shift = 32 - log2_pow2(size);
    for (int i = gl; i < size; i += 16) tab[g][i] = EMPTY_SLOT;
    wave_lds_sync();
    if (live) {
      g16_walk<U, true>(st[g], gl, IA[row], IA[row + 1], SBL, VA, JB, VB, [&](bool active, int col, float v, int) {
        if (active) hash_accum(tab[g], size, shift, col, v, err);
      });
    }
    wave_lds_sync();
    if (PRUNE) {
      int nocc;
      const int kept = prune_emit_group<16>(tab[g], size, PRUNE == 2 ? -1 : want, true, live, gl, JC + off, C + off, &nocc);
      if (live && gl == 0) { pcnt[row] = kept; if (PRUNE == 1 && nocc != want) atomicOr(err, ERRF_COUNT_MISMATCH); }
      wave_lds_sync();
      continue;
    }
    int written = 0;
    for (int i0 = 0; i0 < size; i0 += 16) {
      const slot_t sv = tab[g][i0 + gl];
      const bool occ = live && slot_key(sv) != EMPTY_KEY;
      const unsigned long long mk = ballot64(occ);
      const unsigned gm = (unsigned)(mk >> (lane_id() - gl)) & 0xffffu;
      const int rank = __popc(gm & ((1u << gl) - 1u));
      if (occ) { st_out(JC + off + written + rank, slot_key(sv)); st_out(C + off + written + rank, slot_val(sv)); }
      written += __popc(gm);
    }
    if (live && gl == 0 && written != want) atomicOr(err, ERRF_COUNT_MISMATCH);
    wave_lds_sync();
  }
}

// ------------------------------------------------------------------------------------------------
// Product walk of ONE C row by a block of NW waves.
//
// The row's A entries are staged in groups of 64 (one per lane of the staging wave).  A group splits in two:
//   * entries whose B row has >= LONG_LEN entries ("long": 3 % of the entries, 55-65 % of the products of a power-law
//     matrix) are walked directly: a unit of 64*U consecutive products of ONE B row per wave -- no search, the A value
//     and the B-row base are wave-uniform, the gathers are fully coalesced;
//   * the other entries are flattened: the group's <= 64*63 products are numbered consecutively, a lane finds the entry
//     that owns its product from OWNERSHIP MARKS: bit p of the group's mark words is set when product p is the last one
//     of its entry, so owner(p) = (entries that end before this 64-product round) + popcount(marks below the lane)
//     -- one v_mbcnt pair instead of a six-step binary search through LDS.
// NW > 1: every wave stages one group (no cross-wave scan), one barrier, then the waves take UNITS (a trip of U rounds of
// some group's short products, or 64*U products of a long entry) from one list: the first NW units are dealt, the rest
// claimed from an LDS counter (probe chains make unit times uneven).  f(act[U], col[U], val[U]) is called in wave-uniform
// control flow (val = a*b).
// ------------------------------------------------------------------------------------------------
constexpr int LONG_LEN = WAVE;           // B rows of at least this length are walked by whole waves

// first chunk of A entries fetched ahead of time (wave-per-row kernels prefetch the next row's while they work)
struct PreA { int bs, len; float a; bool valid; };

struct WalkStage1 {                      // one wave per row
  unsigned long long marks[WAVE];        // ownership marks of the staged group (64 entries x <= 63 products)
  int2 rec[WAVE];                        // short entries, compacted: {JB offset of product 0 of the group's numbering, a}
  slot_t dummy[WAVE];                    // one 8-byte word per lane: target of predicated-off atomics
};
template <int NW>
struct WalkStageN {
  unsigned long long marks[NW][WAVE];
  int4 rec[NW][WAVE];                    // short entries from the bottom {off, a, -, -}; long ones from the top {bs, a, len, unit offset}
  unsigned char wpre[NW][WAVE];          // entries of the group that end before round r
  int gT[NW], gNS[NW], gNL[NW], gLU[NW]; // per group: short products, short entries, long entries, long units
  int claim;                             // next unclaimed unit of the chunk
  slot_t dummy[WAVE];                    // shared by the waves: whatever lands there is never read
};
template <int NW> struct WalkSel { typedef WalkStageN<NW> type; };
template <> struct WalkSel<1> { typedef WalkStage1 type; };

struct GroupLanes {                      // what the staging wave keeps in registers about its group
  int bs, len;                           // this lane's entry: B-row start and length
  float a;
  int T, ns;                             // short products / short entries of the group (uniform)
  unsigned long long lmask;              // lanes holding a long entry
  unsigned long long mw;                 // mark word `lane`
  int pexcl;                             // short entries that end before round `lane`
};

template <bool NEED_VAL, int RS, int LONGLEN>
__device__ __forceinline__ GroupLanes stage_group(unsigned long long* marks, char* rec, int ap, int ae,
                                                  const int2* __restrict__ SBL, const float* __restrict__ VA, PreA pre) {
  const int lane = lane_id();
  GroupLanes g;
  g.bs = 0; g.len = 0; g.a = 0.f;
  if (ap < ae) {
    if (pre.valid) { g.bs = pre.bs; g.len = pre.len; g.a = pre.a; }
    else {
      const int2 sbl = SBL[ap];
      g.bs = sbl.x; g.len = sbl.y;
      if (NEED_VAL) g.a = VA[ap];
    }
  }
  const bool isLong = g.len >= LONGLEN;
  const int slen = isLong ? 0 : g.len;
  const bool keep = slen > 0;
  const unsigned long long km = ballot64(keep);
  g.lmask = ballot64(isLong);
  const int incl = wave_incl_add(slen);
  g.T = __builtin_amdgcn_readlane(incl, 63);
  g.ns = __popcll(km);
  marks[lane] = 0ull;
  wave_lds_sync();
  if (keep) {
    *reinterpret_cast<int2*>(rec + mask_rank(km) * RS) = make_int2(g.bs - (incl - slen), __float_as_int(g.a));
    const int q = incl - 1;                              // last product of this entry
    if (q < WAVE * WAVE) atomicOr(reinterpret_cast<unsigned*>(marks) + (q >> 5), 1u << (q & 31));
  }
  wave_lds_sync();
  g.mw = marks[lane];
  const int pc = __popcll(g.mw);
  g.pexcl = wave_incl_add(pc) - pc;
  return g;
}

// U rounds of 64 short products of one staged group.  W[u] / base[u]: mark word and entries-before of round r0+u
// (wave-uniform values, in SGPRs or broadcast VGPRs).
template <int U, bool NEED_VAL, int RS, class F>
__device__ __forceinline__ void short_trip(const char* rec, int T, int ns, int r0, const unsigned long long (&W)[U],
                                           const int (&base)[U], const int* __restrict__ JB,
                                           const float* __restrict__ VB, F&& f, int p0row = -1) {
  const int lane = lane_id();
  int col[U];
  float av[U], vb[U], val[U];
  bool act[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {                          // straight-line, no branches around the gathers
    const int p0 = (r0 + u) * WAVE + lane;
    act[u] = p0 < T;
    const int p = min(p0, T - 1);                        // lanes past the end shadow the last product: valid, nearby reads
    int e = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(W[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)W[u], (unsigned)base[u]));
    e = min(e, ns - 1);
    const int2 ra = *reinterpret_cast<const int2*>(rec + e * RS);
    const int jb = ra.x + p;
    if (ABL(2)) { col[u] = jb; vb[u] = 1.f; }
    else {
    col[u] = JB[jb];
    vb[u] = NEED_VAL ? VB[jb] : 0.f;
    }
    av[u] = __int_as_float(ra.y);
  }
  __builtin_amdgcn_sched_barrier(0);                     // all 2U gathers are issued before the first of them is waited for
#pragma unroll
  for (int u = 0; u < U; ++u) val[u] = av[u] * vb[u];
  f(act, col, val, p0row);                              // p0row: index of the trip's first product in the row's own
}                                                        // numbering (wave-per-row walk), -1 where there is none

// 64*U consecutive products [s0, s0 + 64U) of one long B row: base, length and A value are wave-uniform
template <int U, bool NEED_VAL, class F>
__device__ __forceinline__ void long_trip(int kb, int kl, float ka, int s0, const int* __restrict__ JB,
                                          const float* __restrict__ VB, F&& f, int p0row = -1) {
  const int lane = lane_id();
  int col[U];
  float vb[U], val[U];
  bool act[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p0 = s0 + u * WAVE + lane;
    act[u] = p0 < kl;
    const int jb = kb + min(p0, kl - 1);
    if (ABL(2)) { col[u] = jb; vb[u] = 1.f; }
    else {
    col[u] = JB[jb];
    vb[u] = NEED_VAL ? VB[jb] : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < U; ++u) val[u] = ka * vb[u];
  f(act, col, val, p0row);
}

// wave-uniform R in 1..8 -> compile-time round count (5 runs as 6, 7 as 8): a trip costs what its rounds cost
template <class G>
__device__ __forceinline__ void with_rounds(int R, G&& g) {
  switch (R) {
    case 1: g(std::integral_constant<int, 1>{}); break;
    case 2: g(std::integral_constant<int, 2>{}); break;
    case 3: g(std::integral_constant<int, 3>{}); break;
    case 4: g(std::integral_constant<int, 4>{}); break;
    case 5: case 6: g(std::integral_constant<int, 6>{}); break;
    default: g(std::integral_constant<int, 8>{}); break;
  }
}

// ---- one wave per row (rows of at most 512 products): no long path, the whole group is flattened and ALL its rounds
// are gathered before the first insert -- one memory round trip and one probe sequence per group instead of one per
// pair of rounds (these kernels are bound by the chain of dependent waits of a wave, not by any unit's throughput).
// f is a generic callable: f(act[R], col[R], val[R]) for R in {1,2,3,4,6,8}.
template <int NW, int U, bool NEED_VAL, class F>
__device__ __forceinline__ typename std::enable_if<NW == 1>::type
for_each_product(WalkStage1& st, int as, int ae, const int2* __restrict__ SBL, const float* __restrict__ VA,
                 const int* __restrict__ JB, const float* __restrict__ VB, F&& f,
                 PreA pre = PreA{0, 0, 0.f, false}, int* err = nullptr) {
  const int lane = lane_id();
  int rowBase = 0;                                     // products of the groups before this one
  for (int gb = as; gb < ae; gb += WAVE) {
    const GroupLanes g = stage_group<NEED_VAL, 8, 0x40000000>(st.marks, reinterpret_cast<char*>(st.rec), gb + lane, ae, SBL, VA,
                                                              gb == as ? pre : PreA{0, 0, 0.f, false});
    if (g.T > WAVE * WAVE) {                             // cannot happen for rows of this bin (<= 512 products)
      if (err && lane == 0) atomicOr(err, ERRF_TABLE_FULL);
      continue;
    }
    const int nr = (g.T + WAVE - 1) >> 6;
    const unsigned mlo = (unsigned)g.mw, mhi = (unsigned)(g.mw >> 32);
    for (int r0 = 0; r0 < nr; r0 += 8) {
      with_rounds(min(8, nr - r0), [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        unsigned long long W[R];
        int base[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const int r = min(r0 + u, WAVE - 1);
          W[u] = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mlo, r) |
                 ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)mhi, r) << 32);
          base[u] = __builtin_amdgcn_readlane(g.pexcl, r);
        }
        short_trip<R, NEED_VAL, 8>(reinterpret_cast<const char*>(st.rec), g.T, g.ns, r0, W, base, JB, VB, f, rowBase + r0 * WAVE);
      });
    }
    rowBase += g.T;
  }
}

// ---- NW waves per row
// num != nullptr: every product gets its index in a dense numbering of the ROW (chunk by chunk: the short products of the
// chunk's groups first, then its long entries one after the other) and the callback receives the index of the unit's
// first product -- what the expansion of duplicate-free rows needs to store products without a table.
template <int NW>
struct WalkNumber { int lpro[NW][WAVE]; int gLP[NW]; };   // per long entry: products of the group's earlier long entries; their sum
static_assert(sizeof(WalkNumber<8>) <= 4096 * sizeof(unsigned long long), "lives in the (idle) hash table of an expanded row");

template <int NW, int U, bool NEED_VAL, class F>
__device__ __forceinline__ typename std::enable_if<(NW > 1)>::type
for_each_product(WalkStageN<NW>& st, int as, int ae, const int2* __restrict__ SBL, const float* __restrict__ VA,
                 const int* __restrict__ JB, const float* __restrict__ VB, F&& f,
                 PreA pre = PreA{0, 0, 0.f, false}, int* err = nullptr, WalkNumber<NW>* num = nullptr) {
  (void)err;
  int chunkBase = 0;                                     // products of the chunks before this one (numbering only)
  static_assert(NW <= 16, "the unit table of a chunk lives in the first 16 lanes");
  constexpr int K = WAVE * NW;
  constexpr int UP = WAVE * U;                           // products per unit
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  (void)pre;
  for (int chunk = as; chunk < ae; chunk += K) {
    // ---- wave w stages group w of the chunk
    {
      const GroupLanes g = stage_group<NEED_VAL, 16, LONG_LEN>(st.marks[w], reinterpret_cast<char*>(st.rec[w]), chunk + tid, ae, SBL, VA,
                                                     PreA{0, 0, 0.f, false});
      const bool isLong = (g.lmask >> lane) & 1ull;
      const int units = isLong ? (g.len + UP - 1) / UP : 0;
      const int uincl = wave_incl_add(units);
      if (isLong) st.rec[w][WAVE - 1 - mask_rank(g.lmask)] = make_int4(g.bs, __float_as_int(g.a), g.len, uincl - units);
      if (num) {
        const int lincl = wave_incl_add(isLong ? g.len : 0);
        if (isLong) num->lpro[w][mask_rank(g.lmask)] = lincl - g.len;
        if (lane == 63) num->gLP[w] = lincl;
      }
      st.wpre[w][lane] = (unsigned char)g.pexcl;
      if (lane == 0) { st.gT[w] = g.T; st.gNS[w] = g.ns; st.gNL[w] = __popcll(g.lmask); }
      if (lane == 63) st.gLU[w] = uincl;
      if (tid == 0) st.claim = NW;
    }
    __syncthreads();
    // ---- the chunk's unit list: short trips of the groups first, then the long units (lane g < NW holds group g)
    const int gT = lane < NW ? st.gT[lane] : 0;
    const int gNS = lane < NW ? st.gNS[lane] : 0;
    const int gNL = lane < NW ? st.gNL[lane] : 0;
    const int lug = lane < NW ? st.gLU[lane] : 0;
    const int ntg = (((gT + WAVE - 1) >> 6) + U - 1) / U;
    const int sIncl = wave_incl_add(ntg), lIncl = wave_incl_add(lug);
    const int S = __builtin_amdgcn_readlane(sIncl, 63);
    const int total = S + __builtin_amdgcn_readlane(lIncl, 63);
    int tIncl = 0, lpIncl = 0, gLPv = 0, shortSum = 0;
    if (num) {
      gLPv = lane < NW ? num->gLP[lane] : 0;
      tIncl = wave_incl_add(gT);
      lpIncl = wave_incl_add(gLPv);
      shortSum = __builtin_amdgcn_readlane(tIncl, 63);
    }
    for (int u = __builtin_amdgcn_readfirstlane(w); u < total;) {
      int unext = 0;
      if (lane == 0) unext = atomicAdd(&st.claim, 1);
      if (u < S) {
        const int g = __popcll(ballot64(sIncl <= u));                       // group of this trip (< NW: sIncl[63] = S > u)
        const int t = u - (__builtin_amdgcn_readlane(sIncl, g) - __builtin_amdgcn_readlane(ntg, g));
        const int T = __builtin_amdgcn_readlane(gT, g), ns = __builtin_amdgcn_readlane(gNS, g);
        unsigned long long W[U];
        int base[U];
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
          const int r = min(t * U + uu, WAVE - 1);
          W[uu] = st.marks[g][r];                                          // uniform address: broadcast reads
          base[uu] = st.wpre[g][r];
        }
        const int p0 = num ? chunkBase + (__builtin_amdgcn_readlane(tIncl, g) - T) + t * U * WAVE : -1;
        short_trip<U, NEED_VAL, 16>(reinterpret_cast<const char*>(st.rec[g]), T, ns, t * U, W, base, JB, VB, f, p0);
      } else {
        const int ul = u - S;
        const int g = __popcll(ballot64(lIncl <= ul));
        const int ug = ul - (__builtin_amdgcn_readlane(lIncl, g) - __builtin_amdgcn_readlane(lug, g));
        const int nl = __builtin_amdgcn_readlane(gNL, g);
        const int uex = lane < nl ? st.rec[g][WAVE - 1 - lane].w : 0x7fffffff;   // unit offsets of the group's long entries
        const int k = __popcll(ballot64(uex <= ug)) - 1;
        const int4 r4 = st.rec[g][WAVE - 1 - k];
        const int p0 = num ? chunkBase + shortSum + (__builtin_amdgcn_readlane(lpIncl, g) - __builtin_amdgcn_readlane(gLPv, g)) +
                                 num->lpro[g][k] + (ug - r4.w) * UP
                           : -1;
        long_trip<U, NEED_VAL>(r4.x, r4.z, __int_as_float(r4.y), (ug - r4.w) * UP, JB, VB, f, p0);
      }
      u = __builtin_amdgcn_readfirstlane(unext);
    }
    // the callback's last LDS atomics return nothing (ds_or / ds_add_f32): nothing else waits for them, and a wave
    // that leaves the barrier may read their target before they have landed (seen as lost bitmap bits)
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
    if (num) chunkBase += shortSum + __builtin_amdgcn_readlane(lpIncl, 63);
    __syncthreads();
  }
}

// dynamic row scheduling for block-per-row kernels: rows of a bin differ by up to 8x in work, a static
// round-robin leaves the CUs with the light rows idle.  One agent-scope atomic per row (guide: "dequeue",
// ~0.3-1 us, overlapped with the previous row's tail by fetching one row ahead).
// A single queue word saturates near 88 dequeues/us (guide, "dequeue"): kernels whose rows are short take R
// rows per atomic.
template <int R>
__device__ __forceinline__ int next_row(int* ctr, int* slot, int prev) {
  if (R > 1 && prev >= 0 && (prev + 1) % R != 0) return prev + 1;      // still inside the batch (block-uniform)
  if (threadIdx.x == 0) *slot = atomicAdd(ctr, R);
  __syncthreads();
  const int q = *slot;
  __syncthreads();
  return q;
}
__device__ __forceinline__ int next_row(int* ctr, int* slot) { return next_row<1>(ctr, slot, -1); }

// per-row metadata, fetched one row ahead of use so that the dependent rowIds -> IA/IC loads of the
// next row overlap the current row's work
struct RowMeta { int row, as, ae, x0, x1, x2; };

__device__ __forceinline__ RowMeta load_meta_sym(const int* rows, int q, int count, const int* IA, const int* rowFlops) {
  RowMeta mtd{0, 0, 0, 0, 0, 0};
  if (q < count) { mtd.row = rows[q]; mtd.as = IA[mtd.row]; mtd.ae = IA[mtd.row + 1]; mtd.x0 = rowFlops[mtd.row]; }
  return mtd;
}
__device__ __forceinline__ RowMeta load_meta_num(const int* rows, int q, int count, const int* IA, const int* IC,
                                                 const int* rowFlops = nullptr) {
  RowMeta mtd{0, 0, 0, 0, 0, 0};
  if (q < count) {
    mtd.row = rows[q]; mtd.as = IA[mtd.row]; mtd.ae = IA[mtd.row + 1]; mtd.x0 = IC[mtd.row]; mtd.x1 = IC[mtd.row + 1];
    if (rowFlops) mtd.x2 = rowFlops[mtd.row];
  }
  return mtd;
}

// wave-per-row kernels: this lane's A entry of the row's first chunk, fetched one row ahead
__device__ __forceinline__ PreA load_pre(const RowMeta& mtd, const int2* SBL, const float* VA, bool needVal) {
  PreA p{0, 0, 0.f, true};
  const int ap = mtd.as + lane_id();
  if (ap < mtd.ae) { const int2 sbl = SBL[ap]; p.bs = sbl.x; p.len = sbl.y; if (needVal) p.a = VA[ap]; }
  return p;
}

// ------------------------------------------------------------------------------------------------
// Medium rows (65..4096 products): one block of NW waves per row, LDS key table (symbolic) or
// key+value table compacted by a table sweep (numeric).
// ------------------------------------------------------------------------------------------------
template <int NW, int TBL, int U>
__global__ __launch_bounds__(WAVE * NW) void k_sym_hash(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                         const int* __restrict__ JB,
                                                         const int* __restrict__ rowFlops, int* __restrict__ IC,
                                                         int* __restrict__ err, int* __restrict__ qctr) {
  __shared__ __attribute__((aligned(16))) int keys[TBL];
  __shared__ typename WalkSel<NW>::type st;
  __shared__ int cnt_s;
  __shared__ int qslot;
  const int tid = threadIdx.x, lane = lane_id();
  if (tid < WAVE) st.dummy[tid] = DUMMY_SLOT;    // (ordered before the first insert by the barrier / wave order below)
  int first = binPtr[bin], count = binPtr[bin + 1] - first;
  constexpr int QB = NW >= 8 ? 2 : 4;            // rows per dequeue
  int stride = (int)gridDim.x;
  if (NW == 1) {                                 // static schedule: XCD-contiguous (see xcd_range)
    const XcdRange xr = xcd_range(count);
    first += xr.lo; count = xr.hi - xr.lo; stride = xr.nb;
  }
  const int* rows = rowIds + first;
  int q = NW > 1 ? next_row<QB>(qctr, &qslot, -1) : (int)(blockIdx.x >> 3);
  RowMeta cur = load_meta_sym(rows, q, count, IA, rowFlops);
  // NW == 1: two rows of metadata and one row of A entries are in flight ahead of the row being processed
  RowMeta nxt = NW == 1 ? load_meta_sym(rows, q + stride, count, IA, rowFlops) : RowMeta{0, 0, 0, 0, 0, 0};
  PreA pc = NW == 1 ? load_pre(cur, SBL, nullptr, false) : PreA{0, 0, 0.f, false};
  while (q < count) {
    int qn;
    PreA pn{0, 0, 0.f, false};
    RowMeta nn{0, 0, 0, 0, 0};
    if (NW == 1) {
      qn = q + stride;
      nn = load_meta_sym(rows, qn + stride, count, IA, rowFlops);
      pn = load_pre(nxt, SBL, nullptr, false);
    } else {
      qn = next_row<QB>(qctr, &qslot, q);
      nxt = load_meta_sym(rows, qn, count, IA, rowFlops);
    }
    const int size = table_size<4>(cur.x0, 64, TBL);
    const int shift = 32 - log2_pow2(size);
    clear_table(keys, nullptr, size, tid, WAVE * NW);
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    int mine = 0;
    for_each_product<NW, U, false>(st, cur.as, cur.ae, SBL, nullptr, JB, nullptr,
                                   [&](const auto& act, const auto& col, const auto& val, int) {
      mine += hash_insert_chunked<InsChunk<NW>::value>(keys, size, shift, act, col, reinterpret_cast<int*>(&st.dummy[lane_id()]), err);
    }, pc, err);
    const int ws = wave_sum(mine);                           // hash_insert_multi counts per lane
    if (NW == 1) {
      if (lane == 0) IC[cur.row] = ws;
    } else {
      if (lane == 0 && ws) atomicAdd(&cnt_s, ws);
      __syncthreads();
      if (tid == 0) IC[cur.row] = cnt_s;
    }
    __syncthreads();
    cur = nxt;
    if (NW == 1) { nxt = nn; pc = pn; }
    q = qn;
  }
}

template <int NW, int TBL, int U, int PRUNE = 0>      // PRUNE: see k_num_g16
__global__ __launch_bounds__(WAVE * NW) void k_num_hash(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                         const float* __restrict__ VA,
                                                         const int* __restrict__ JB,
                                                         const float* __restrict__ VB,
                                                         const int* __restrict__ IC, int* __restrict__ JC,
                                                         float* __restrict__ C, int* __restrict__ err,
                                                         int* __restrict__ qctr, const int* __restrict__ rowFlops,
                                                         int* __restrict__ pcnt) {
  __shared__ __attribute__((aligned(16))) slot_t tab[TBL];   // (column, value) pairs
  __shared__ PruneShared<NW> ps;                 // (fused prune, several waves: the block reductions)
  __shared__ typename WalkSel<NW>::type st;
  __shared__ int qslot;
  __shared__ int emitted;                        // output positions handed out so far in this row (NW > 1)
  const int tid = threadIdx.x;
  if (tid < WAVE) st.dummy[tid] = DUMMY_SLOT;
  constexpr int T = WAVE * NW;
  int first = binPtr[bin], count = binPtr[bin + 1] - first;
  constexpr int QB = NW >= 8 ? 2 : 4;            // rows per dequeue
  int stride = (int)gridDim.x;
  if (NW == 1) {                                 // static schedule: XCD-contiguous (see xcd_range)
    const XcdRange xr = xcd_range(count);
    first += xr.lo; count = xr.hi - xr.lo; stride = xr.nb;
  }
  const int* rows = rowIds + first;
  int q = NW > 1 ? next_row<QB>(qctr, &qslot, -1) : (int)(blockIdx.x >> 3);
  const int* const rf = rowFlops;                        // the product count decides hash vs expansion
  RowMeta cur = load_meta_num(rows, q, count, IA, IC, rf);
  RowMeta nxt = NW == 1 ? load_meta_num(rows, q + stride, count, IA, IC, rf) : RowMeta{0, 0, 0, 0, 0, 0};
  PreA pc = NW == 1 ? load_pre(cur, SBL, VA, true) : PreA{0, 0, 0.f, false};
  while (q < count) {
    int qn;
    PreA pn{0, 0, 0.f, false};
    RowMeta nn{0, 0, 0, 0, 0};
    if (NW == 1) {
      qn = q + stride;
      nn = load_meta_num(rows, qn + stride, count, IA, IC, rf);
      pn = load_pre(nxt, SBL, VA, true);
    } else {
      qn = next_row<QB>(qctr, &qslot, q);
      nxt = load_meta_num(rows, qn, count, IA, IC, rf);
    }
    const int off = cur.x0;
    const int want = cur.x1 - off;                      // exact distinct count from the symbolic pass
    if (NW == 1 && PRUNE == 1 && want == cur.x2 && want <= TBL) {
      // fused prune of an expanded row: its products as a plain list in LDS (no clear, no probing), then the row rule
      for_each_product<NW, U, true>(st, cur.as, cur.ae, SBL, VA, JB, VB,
                                    [&](const auto& act, const auto& col, const auto& val, int p0) {
        constexpr int R = (int)(sizeof(col) / sizeof(col[0]));
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const unsigned o = (unsigned)(p0 + u * WAVE + lane_id());
          if (act[u] && o < (unsigned)want) tab[o] = make_slot(col[u], val[u]);
        }
      }, pc, err);
      wave_lds_sync();
      int nocc;
      const int kept = prune_emit_group<64>(tab, (want + 63) & ~63, want, false, true, lane_id(), JC + off, C + off, &nocc);
      if (lane_id() == 0) pcnt[cur.row] = kept;
      wave_lds_sync();
      cur = nxt;
      nxt = nn; pc = pn;
      q = qn;
      continue;
    }
    if (NW == 1 && PRUNE == 0 && want == cur.x2) {
      // As many distinct columns as products: no two products of this row meet, nothing to accumulate.  The row is
      // EXPANDED: every product goes straight to the position it has in the row's own numbering -- coalesced stores,
      // no table, no clear, no compaction sweep (more than half of the products of rows up to 256 products on a
      // power-law matrix sit in such rows).
      int* const JCrow = JC + off;
      float* const Crow = C + off;
      for_each_product<NW, U, true>(st, cur.as, cur.ae, SBL, VA, JB, VB,
                                    [&](const auto& act, const auto& col, const auto& val, int p0) {
        constexpr int R = (int)(sizeof(col) / sizeof(col[0]));
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const unsigned o = (unsigned)(p0 + u * WAVE + lane_id());
          if (act[u] && o < (unsigned)want) { st_out(JCrow + o, col[u]); st_out(Crow + o, val[u]); }
        }
      }, pc, err);
      cur = nxt;
      if (NW == 1) { nxt = nn; pc = pn; }
      q = qn;
      continue;
    }
    if constexpr (NW > 1 && PRUNE == 0) if (want == cur.x2) {
      // several waves per row, no repeated column (77 % of the rows of 513-1024 products on a power-law matrix): expanded
      // like the wave-per-row case, positions from the walk's dense product numbering
      int* const JCrow = JC + off;
      float* const Crow = C + off;
      for_each_product<NW, U, true>(st, cur.as, cur.ae, SBL, VA, JB, VB,
                                    [&](const auto& act, const auto& col, const auto& val, int p0) {
        constexpr int R = (int)(sizeof(col) / sizeof(col[0]));
#pragma unroll
        for (int u = 0; u < R; ++u) {
          const unsigned o = (unsigned)(p0 + u * WAVE + lane_id());
          if (act[u] && o < (unsigned)want) { st_out(JCrow + o, col[u]); st_out(Crow + o, val[u]); }
        }
      }, pc, err, reinterpret_cast<WalkNumber<NW>*>(tab));   // the table is idle in an expanded row: its memory holds the numbering
      cur = nxt;
      q = qn;
      continue;
    }
    const int size = table_size<4>(want, T > 64 ? T : 64, TBL);
    const int shift = 32 - log2_pow2(size);
    clear_slots(tab, size, tid, T);
    if (NW > 1 && tid == 0) emitted = 0;
    __syncthreads();
    if (!ABL(8))
    for_each_product<NW, U, true>(st, cur.as, cur.ae, SBL, VA, JB, VB,
                                  [&](const auto& act, const auto& col, const auto& val, int) {
      if (ABL(1)) {
#pragma unroll
        for (int u = 0; u < (int)(sizeof(col) / sizeof(col[0])); ++u) asm volatile("" :: "v"(col[u]), "v"(val[u]));
      } else
      hash_accum_chunked<InsChunk<NW>::value>(tab, size, shift, act, col, val, &st.dummy[lane_id()], err);
    }, pc, err);
    // compaction: wave w sweeps the contiguous slots [w*per, w*per+per) 64 at a time, so that a wave's stores
    // land on consecutive output positions.  One wave: a single pass.  Several waves: a wave reads its slots into
    // registers, claims its share of the row's output range with ONE LDS atomic (any order inside a row is a valid
    // CSR row) and emits from the registers: no counting sweep, no block scan, no barrier.
    const int per = size / NW;
    const int lane = lane_id(), w = tid >> 6;
    if (ABL(4)) { __syncthreads(); }
    else if (PRUNE && NW == 1) {
      int nocc;
      const int kept = prune_emit_group<64>(tab, per, PRUNE == 2 ? -1 : want, true, true, lane, JC + off, C + off, &nocc);
      if (tid == 0) { pcnt[cur.row] = kept; if (PRUNE == 1 && nocc != want) atomicOr(err, ERRF_COUNT_MISMATCH); }
      __syncthreads();
    } else if (PRUNE) {
      prune_emit_block<NW, TBL / NW / WAVE>(tab, per, PRUNE == 2 ? -1 : want, ps, JC + off, C + off, pcnt + cur.row, err);
      __syncthreads();
    } else if (NW == 1) {
      int pos = 0;                                    // relative to the row: small 32-bit offsets for both stores
      int* const JCrow = JC + off;
      float* const Crow = C + off;
      for (int i0 = 0; i0 < per; i0 += WAVE) {
        const slot_t sv = tab[i0 + lane];
        const bool occ = slot_key(sv) != EMPTY_KEY;
        const unsigned long long mk = ballot64(occ);
        const unsigned o = (unsigned)(pos + mask_rank(mk));
        if (occ && o < (unsigned)want) { st_out(JCrow + o, slot_key(sv)); st_out(Crow + o, slot_val(sv)); }
        pos += __popcll(mk);
      }
      if (tid == 0 && pos != want) atomicOr(err, ERRF_COUNT_MISMATCH);
      __syncthreads();
    } else {
      emit_claimed<TBL / NW / WAVE>(tab, w * per, per, &emitted, off, off + want, JC, C);
      __syncthreads();
      if (tid == 0 && emitted != want) atomicOr(err, ERRF_COUNT_MISMATCH);
    }
    cur = nxt;
    if (NW == 1) { nxt = nn; pc = pn; }
    q = qn;
  }
}

// ------------------------------------------------------------------------------------------------
// Big rows (> 4096 products): one 1024-thread block per row.
//   symbolic : LDS bitmap over a window of SYM_WC columns, popcount           (any n, 1 window up to 1M cols)
//   numeric A: n <= BIG_WC: bitmap (saved by the symbolic pass when the workspace has room, rebuilt
//              otherwise) + popcount ranks give every column its final position; values accumulate in an
//              LDS float array addressed by rank (no probing, sorted output)
//   numeric B: any n: 128 KB LDS hash table; rows with more distinct columns than one table holds take
//              several passes, each pass owning the columns of one hash class
// ------------------------------------------------------------------------------------------------
#ifndef SMF_BIG_NW
#define SMF_BIG_NW 16
#endif
constexpr int BIG_NW = SMF_BIG_NW;   // (SMF_BIG_NW: diagnostic builds with fewer waves per big-row block)
constexpr int BIG_THREADS = BIG_NW * WAVE;
constexpr int BIG_U = 4;                        // rounds in flight per wave: bitmap / rank kernels
#ifndef SMF_BH_U
#define SMF_BH_U 2
#endif
constexpr int BH_U = SMF_BH_U;                         // ... and the big-row hash kernel (probe chains: 2 measured better than 4)
constexpr int SYM_WC = 1 << 20;                // columns per symbolic window (128 KB bitmap)
constexpr int SYM_WORDS = SYM_WC / 32;
constexpr int BIG_WC = 262144;                 // columns covered by the rank kernel (32 KB bitmap)
constexpr int BIG_WORDS = BIG_WC / 32;         // 8192
constexpr int BIG_WPT = BIG_WORDS / BIG_THREADS;  // 8 words per thread
constexpr int BIG_CAP = 16384;                 // float accumulators per rank pass (64 KB)
constexpr int BH_SLOTS = 16384;                // hash kernel: 128 KB of (key, value) pairs (what the walk stage leaves of 160 KB)
constexpr int BH_CAP = 11264;                  // distinct columns per hash pass (load <= 0.6 incl. partition skew; measured
                                               // best of 8704/10240/12800 once later passes stream parked products)
constexpr int BH_CAP_MAX = 12288;              // SPGEMM_BHCAP may raise the cap to this (load 0.74)
constexpr int BH_SPILL = 1 << 18;              // (col,val) pairs a multi-pass row may park in HBM per block (2 MB)
constexpr int BH_MAXCLS = 32;                  // hash classes (passes) with their own parking region

struct BigSymShared {
  unsigned bitmap[SYM_WORDS];
  WalkStageN<BIG_NW> st;
  int red[BIG_NW];
};
constexpr int BIG_GROUPS = BIG_WORDS / WAVE;   // 128 groups of 64 bitmap words
constexpr int BIG_GPW = BIG_GROUPS / BIG_NW;   // 8 groups per wave, interleaved (dense column ranges spread over all waves)
struct BigNumShared {
  unsigned bitmap[BIG_WORDS];
  int prefix[BIG_WORDS];
  float acc[BIG_CAP];
  WalkStageN<BIG_NW> st;
  int red[BIG_NW];
  int gtot[BIG_GROUPS];
  int gbase[BIG_GROUPS];
  int cnt;
};
struct BigHashShared {
  slot_t tab[BH_SLOTS];          // (column, value) pairs
  WalkStageN<BIG_NW> st;
  int red[BIG_NW];
  int spillCnt[BH_MAXCLS];
  int emitted;
  int ovf;                       // a parking region overflowed: redo the row without parking
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

// saveBitmaps: device buffer of saveCap row slots x BIG_WORDS words (may be null / 0); only used when n <= BIG_WC
__global__ __launch_bounds__(BIG_THREADS) void k_sym_big(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                         const int* __restrict__ JB,
                                                         int n, int* __restrict__ IC,
                                                         unsigned* __restrict__ saveBitmaps, int saveCap,
                                                         int* __restrict__ qctr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BigSymShared& sh = *reinterpret_cast<BigSymShared*>(smem_raw);
  const int tid = threadIdx.x;
  const int first = binPtr[bin], count = binPtr[bin + 1] - first;
  // the next row is dequeued and its metadata requested before this row's work starts: the dependent chain
  // queue -> rowIds -> IA (three memory round trips) hides behind the row instead of preceding it
  int q = next_row(qctr, &sh.red[0]);
  RowMeta cur = load_meta_sym(rowIds + first, q, count, IA, IA);
  while (q < count) {
    const int qn = next_row(qctr, &sh.red[0]);
    const RowMeta nxt = load_meta_sym(rowIds + first, qn, count, IA, IA);
    const int row = cur.row;
    const int as = cur.as, ae = cur.ae;
    int total = 0;
    for (int w0 = 0; w0 < n; w0 += SYM_WC) {
      const int wc = min(SYM_WC, n - w0);
      const int words = (wc + 31) >> 5;
      const int words4 = (words + 3) & ~3;             // the bitmap is cleared and counted four words at a time
      for (int i = tid * 4; i < words4; i += BIG_THREADS * 4) *reinterpret_cast<uint4*>(&sh.bitmap[i]) = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
      for_each_product<BIG_NW, BIG_U, false>(sh.st, as, ae, SBL, nullptr, JB, nullptr,
                                             [&](const bool (&act)[BIG_U], const int (&col)[BIG_U], const float (&)[BIG_U], int) {
#pragma unroll
        for (int u = 0; u < BIG_U; ++u) {            // predicated by value: OR-ing 0 changes nothing
          const int c = col[u] - w0;
          const bool ok = act[u] && (unsigned)c < (unsigned)wc;
          atomicOr(ok ? &sh.bitmap[c >> 5] : reinterpret_cast<unsigned*>(&sh.st.dummy[lane_id()]), ok ? 1u << (c & 31) : 0u);
        }
      });
      int mine = 0;
      for (int i = tid * 4; i < words4; i += BIG_THREADS * 4) {
        const uint4 v4 = *reinterpret_cast<const uint4*>(&sh.bitmap[i]);
        mine += __popc(v4.x) + __popc(v4.y) + __popc(v4.z) + __popc(v4.w);
      }
      if (n <= BIG_WC && q < saveCap) {
        unsigned* dst = saveBitmaps + (size_t)q * BIG_WORDS;
        for (int i = tid; i < BIG_WORDS; i += BIG_THREADS) dst[i] = i < words ? sh.bitmap[i] : 0u;
      }
      total += block_sum_16(mine, sh.red);
      __syncthreads();
    }
    if (tid == 0) IC[row] = total;
    cur = nxt;
    q = qn;
  }
}

// numeric A (n <= BIG_WC): rank kernel
__global__ __launch_bounds__(BIG_THREADS) void k_num_big(const int* __restrict__ binPtr, int bin,
                                                         const int* __restrict__ rowIds,
                                                         const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                         const float* __restrict__ VA,
                                                         const int* __restrict__ JB,
                                                         const float* __restrict__ VB, int n,
                                                         const int* __restrict__ IC, int* __restrict__ JC,
                                                         float* __restrict__ C, int* __restrict__ err,
                                                         const unsigned* __restrict__ savedBitmaps, int savedCap,
                                                         int* __restrict__ qctr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BigNumShared& sh = *reinterpret_cast<BigNumShared*>(smem_raw);
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int first = binPtr[bin], count = binPtr[bin + 1] - first;
  for (int q = next_row(qctr, &sh.red[0]); q < count; q = next_row(qctr, &sh.red[0])) {
    const int row = rowIds[first + q];
    const int as = IA[row], ae = IA[row + 1];
    const int outBase = IC[row];
    const int outEnd = IC[row + 1];
    // which columns occur: reload the symbolic pass's bitmap, or rebuild it
    if (q < savedCap) {
      const unsigned* src = savedBitmaps + (size_t)q * BIG_WORDS;
      for (int i = tid; i < BIG_WORDS; i += BIG_THREADS) sh.bitmap[i] = src[i];
      __syncthreads();
    } else {
      for (int i = tid; i < BIG_WORDS; i += BIG_THREADS) sh.bitmap[i] = 0u;
      __syncthreads();
      for_each_product<BIG_NW, BIG_U, false>(sh.st, as, ae, SBL, nullptr, JB, nullptr,
                                             [&](const bool (&act)[BIG_U], const int (&col)[BIG_U], const float (&)[BIG_U], int) {
#pragma unroll
        for (int u = 0; u < BIG_U; ++u) {
          const bool ok = act[u] && (unsigned)col[u] < (unsigned)n;
          atomicOr(&sh.bitmap[ok ? col[u] >> 5 : 0], ok ? 1u << (col[u] & 31) : 0u);
        }
      });
    }
    // exclusive popcount prefix over the words.  Wave w owns the 64-word groups w, w+16, ...: consecutive lanes
    // read consecutive words (no bank conflicts) and a dense column range is shared by all waves.
    int lexcl[BIG_GPW];
#pragma unroll
    for (int si = 0; si < BIG_GPW; ++si) {
      const int g = w + si * BIG_NW;
      const int pc = __popc(sh.bitmap[g * WAVE + lane]);
      const int incl = wave_incl_add(pc);
      lexcl[si] = incl - pc;
      if (lane == 63) sh.gtot[g] = incl;
    }
    __syncthreads();
    if (w == 0) {
      const int s0 = sh.gtot[2 * lane], s1 = sh.gtot[2 * lane + 1];
      const int incl = wave_incl_add(s0 + s1);
      sh.gbase[2 * lane] = incl - s0 - s1;
      sh.gbase[2 * lane + 1] = incl - s1;
      if (lane == 63) sh.cnt = incl;
    }
    __syncthreads();
    int cntw = sh.cnt;
#pragma unroll
    for (int si = 0; si < BIG_GPW; ++si) {
      const int g = w + si * BIG_NW;
      lexcl[si] += sh.gbase[g];
      sh.prefix[g * WAVE + lane] = lexcl[si];
    }
    __syncthreads();
    if (outBase + cntw != outEnd) { if (tid == 0) atomicOr(err, ERRF_COUNT_MISMATCH); cntw = min(cntw, max(0, outEnd - outBase)); }
    // BIG_CAP ranks at a time: column indices first (already sorted; scattered into LDS by rank, then stored
    // coalesced), then the values accumulated by rank
    int* accI = reinterpret_cast<int*>(sh.acc);
    for (int lo = 0; lo < cntw; lo += BIG_CAP) {
      const int span = min(BIG_CAP, cntw - lo);
#pragma unroll
      for (int si = 0; si < BIG_GPW; ++si) {
        const int wd = (w + si * BIG_NW) * WAVE + lane;
        unsigned bits = sh.bitmap[wd];
        int pos = lexcl[si] - lo;
        const int cbase = wd * 32;
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          if ((unsigned)pos < (unsigned)span) accI[pos] = cbase + b;
          ++pos;
        }
      }
      __syncthreads();
      for (int i = tid; i < span; i += BIG_THREADS) st_out(JC + outBase + lo + i, accI[i]);
      __syncthreads();
      for (int i = tid; i < span; i += BIG_THREADS) sh.acc[i] = 0.f;
      __syncthreads();
      for_each_product<BIG_NW, BIG_U, true>(sh.st, as, ae, SBL, VA, JB, VB,
                                            [&](const bool (&act)[BIG_U], const int (&col)[BIG_U], const float (&val)[BIG_U], int) {
        int rk[BIG_U];
#pragma unroll
        for (int u = 0; u < BIG_U; ++u) {          // the U rank lookups (two LDS reads each) overlap
          const int c = act[u] && (unsigned)col[u] < (unsigned)n ? col[u] : 0;
          const int wi = c >> 5;
          rk[u] = sh.prefix[wi] + __popc(sh.bitmap[wi] & ((1u << (c & 31)) - 1u)) - lo;
        }
        bool ok[BIG_U];
#pragma unroll
        for (int u = 0; u < BIG_U; ++u)
          ok[u] = act[u] && (unsigned)col[u] < (unsigned)n && (unsigned)rk[u] < (unsigned)span;
        lds_fadd_multi(sh.acc, rk, ok, val, reinterpret_cast<int*>(&sh.st.dummy[lane_id()]));
      });
      for (int i = tid; i < span; i += BIG_THREADS) st_out(C + outBase + lo + i, sh.acc[i]);
      __syncthreads();
    }
    __syncthreads();
  }
}

// numeric B (any n): multi-pass LDS hash.  A row with more distinct columns than one table holds takes npass passes,
// pass k owning the columns of hash class k.  Only pass 0 walks the products: what belongs to a later class is
// parked as (col, a*b) pairs in a per-block HBM buffer and the later passes stream that buffer (coalesced, no
// search, no gather).  Rows with more products than the buffer holds fall back to walking once per pass.
__device__ __forceinline__ unsigned bh_class(int col, unsigned npass) {
  return ((((unsigned)col * 0x85ebca6bu) >> 16) * npass) >> 16;     // second hash, independent of the slot hash
}

#ifdef SMF_STAMPS
// diagnostic build: cycles of wave 0 of every block, summed per phase of k_num_bighash (read with spgemm_hip_debug_stamps)
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long t_ = __builtin_readcyclecounter(); const unsigned long long t0_ = t_;
#define STAMP(i) { const unsigned long long n_ = __builtin_readcyclecounter(); st_[i] += n_ - t_; t_ = n_; }
#define STAMP_IN(var) const unsigned long long var = __builtin_readcyclecounter();
#define STAMP_OUT(i, var) { st_[i] += __builtin_readcyclecounter() - var; }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_IN(var)
#define STAMP_OUT(i, var)
#endif
__global__ __launch_bounds__(BIG_THREADS) void k_num_bighash(const int* __restrict__ binPtr, int bin,
                                                             const int* __restrict__ rowIds,
                                                             const int* __restrict__ IA, const int2* __restrict__ SBL,
                                                             const float* __restrict__ VA,
                                                             const int* __restrict__ JB,
                                                             const float* __restrict__ VB,
                                                             const int* __restrict__ IC, int* __restrict__ JC,
                                                             float* __restrict__ C, int* __restrict__ err,
                                                             int* __restrict__ qctr, const int* __restrict__ rowFlops,
                                                             int2* __restrict__ spill, int spillCap, int bhCap, int marginPct) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  BigHashShared& sh = *reinterpret_cast<BigHashShared*>(smem_raw);
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int first = binPtr[bin], count = binPtr[bin + 1] - first;
  int2* const park = spill ? spill + (size_t)blockIdx.x * (size_t)spillCap : nullptr;
  if (tid < WAVE) sh.st.dummy[tid] = DUMMY_SLOT;
  STAMP_DECL
  int q = next_row(qctr, &sh.red[0]);                 // one row ahead, as in k_sym_big
  RowMeta cur = load_meta_num(rowIds + first, q, count, IA, IC, rowFlops);
  while (q < count) {
    const int qn = next_row(qctr, &sh.red[0]);
    const RowMeta nxt = load_meta_num(rowIds + first, qn, count, IA, IC, rowFlops);
    const int as = cur.as, ae = cur.ae;
    const int outBase = cur.x0;
    const int outEnd = cur.x1;
    const int want = outEnd - outBase;
    const unsigned npass = (unsigned)((want + bhCap - 1) / bhCap);
    // class c >= 1 parks in its own region of `stride` pairs: the expected class size (the classes are a hash of the
    // column, so products/npass) plus a margin.  A class that outgrows its region raises sh.ovf and the row is redone
    // the slow way (one walk per pass) -- rare, and never wrong.
    const int flops = cur.x2;
    const int stride = (int)min((long long)flops, (long long)flops * marginPct / (100ll * (long long)npass) + 256ll);
    const bool canSpill = npass > 1 && npass <= (unsigned)BH_MAXCLS && park != nullptr &&
                          (long long)(npass - 1) * stride <= (long long)spillCap;
    const int perPass = (want + (int)npass - 1) / (int)npass;
    // any multiple of 1024 slots: FOUR times the distinct columns of a pass when that fits (2x / 3x / 4x: 0.845 / 0.832 /
    // 0.830 ms -- the lock-step probe loop pays for the longest chain of its 128 products, see table_size)
    const int size = min(BH_SLOTS, max(BIG_THREADS, (4 * perPass + BIG_THREADS - 1) / BIG_THREADS * BIG_THREADS));
    const int shift = 0;
    const int per = size / BIG_NW;
    bool useSpill = canSpill;
    STAMP(0)
    for (unsigned pass = 0; pass < npass;) {
      clear_slots(sh.tab, size, tid, BIG_THREADS);
      if (pass == 0 && tid < BH_MAXCLS) sh.spillCnt[tid] = 0;
      if (pass == 0 && tid == 0) { sh.emitted = 0; sh.ovf = 0; }
      __syncthreads();
      STAMP(1)
      if (pass == 0 || !useSpill) {
        for_each_product<BIG_NW, BH_U, true>(sh.st, as, ae, SBL, VA, JB, VB,
                                              [&](const bool (&act)[BH_U], const int (&col)[BH_U], const float (&val)[BH_U], int) {
          STAMP_IN(ti_)
          bool mine[BH_U];
          unsigned cls[BH_U];
#pragma unroll
          for (int u = 0; u < BH_U; ++u) {
            cls[u] = npass == 1 ? 0u : bh_class(col[u], npass);
            mine[u] = act[u] && cls[u] == pass;
          }
          hash_accum_multi<false>(sh.tab, size, shift, mine, col, val, &sh.st.dummy[lane_id()], err);
          STAMP_OUT(3, ti_)
          STAMP_IN(tp_)
          if (useSpill) {                              // block-uniform; here pass == 0
            for (unsigned c = 1; c < npass; ++c) {
              unsigned long long mk[BH_U];
              int total = 0;
#pragma unroll
              for (int u = 0; u < BH_U; ++u) { mk[u] = ballot64(act[u] && cls[u] == c); total += __popcll(mk[u]); }
              if (total) {                             // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(&sh.spillCnt[c], total);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base + total > stride) {           // wave-uniform: region full, the row will be redone
                  if (lane == 0) sh.ovf = 1;
                } else {
                  int2* const dst = park + (size_t)(c - 1) * (size_t)stride;
#pragma unroll
                  for (int u = 0; u < BH_U; ++u) {
                    if (act[u] && cls[u] == c) dst[base + mask_rank(mk[u])] = make_int2(col[u], __float_as_int(val[u]));
                    base += __popcll(mk[u]);
                  }
                }
              }
            }
          }
          STAMP_OUT(4, tp_)
        });
        STAMP(2)
        if (pass == 0 && useSpill && sh.ovf) {        // block-uniform (read after the walk's closing barrier)
          __syncthreads();                             // everyone has seen the flag before it is reset
          useSpill = false;
          continue;                                    // pass 0 again, without parking: table and counters are reset
        }
      } else {
        // stream this class's parked pairs; the next batch is in flight while the current one is inserted
        const int cnt = sh.spillCnt[pass];            // written in pass 0, read-only since its closing barrier
        const int2* const src = park + (size_t)(pass - 1) * (size_t)stride;
        int2 nxt[BH_U];
#pragma unroll
        for (int u = 0; u < BH_U; ++u) { const int idx = u * BIG_THREADS + tid; nxt[u] = src[idx < cnt ? idx : 0]; }
        for (int i0 = 0; i0 < cnt; i0 += BIG_THREADS * BH_U) {
          bool mine[BH_U];
          int col[BH_U];
          float val[BH_U];
#pragma unroll
          for (int u = 0; u < BH_U; ++u) {
            col[u] = nxt[u].x;
            val[u] = __int_as_float(nxt[u].y);
            mine[u] = i0 + u * BIG_THREADS + tid < cnt;
          }
#pragma unroll
          for (int u = 0; u < BH_U; ++u) {
            const int idx = i0 + BIG_THREADS * BH_U + u * BIG_THREADS + tid;
            nxt[u] = src[idx < cnt ? idx : 0];
          }
          hash_accum_multi<false>(sh.tab, size, shift, mine, col, val, &sh.st.dummy[lane_id()], err);
        }
        __syncthreads();
        STAMP(5)
      }
      // compaction: wave w emits the slots [w*per, w*per+per); its share of the row's output range comes from one
      // LDS atomic (the counter runs on across the passes of a row)
      emit_claimed<BH_SLOTS / BIG_NW / WAVE>(sh.tab, w * per, per, &sh.emitted, outBase, outEnd, JC, C);
      __syncthreads();
      STAMP(6)
      ++pass;
    }
    if (tid == 0 && sh.emitted != want) atomicOr(err, ERRF_COUNT_MISMATCH);
    __syncthreads();
    cur = nxt;
    q = qn;
    STAMP(7)
  }
#ifdef SMF_STAMPS
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[i], st_[i]);
    atomicAdd(&g_stamps[8], __builtin_readcyclecounter() - t0_);
    atomicAdd(&g_stamps[9], 1ull);
  }
#endif
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

// clampTotal: IC[m] = min(total, INT_MAX) (C.rowPtr: the host rejects an overflow anyway); otherwise the total wraps
// modulo 2^32 like the interior prefixes, so that differences of neighbours stay exact (the dflops scan of the
// classification API, whose consumers take differences)
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(int m, int* __restrict__ IC,
                                                              const unsigned long long* __restrict__ tileOff,
                                                              const unsigned long long* __restrict__ total, int clampTotal) {
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
    IC[m] = (clampTotal && t > 0x7fffffffULL) ? 0x7fffffff : (int)(unsigned)t;
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

// CSR::makeOrdered on the device (nlibs/CSR.cc:73-86).  Rows of <= SORT_MAX entries: one block per row, bitonic sort in
// LDS (k_sort_rows, which also counts the longer rows that are not sorted yet).  Longer rows (the output of the big-row
// hash kernel: up to tens of thousands of entries): a per-row stable LSD radix sort over the column bits, 8 bits per
// pass, between the row's own segment of the arrays and of a scratch copy (k_sort_long_rows) -- the reference's device
// primitives for this are nlibs/bitonic_sort.cuh:19-87 and mindex2-cuda/radix_sort.cuh:2-62.
// Used by tests/drivers, not by the timed path.
constexpr int SORT_MAX = 4096;
__global__ __launch_bounds__(256) void k_sort_rows(int m, const int* __restrict__ IC, int* __restrict__ JC,
                                                    float* __restrict__ C, int* __restrict__ longUnsorted) {
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
      if (tid == 0 && unsorted) atomicAdd(longUnsorted, 1);
      __syncthreads();
    }
  }
}

// rows longer than SORT_MAX that are not sorted: stable LSD radix sort, 8 bits per pass over `keyBits` column bits,
// ping-pong between (JC, C) and (JS, CS) inside the row's own segment [IC[row], IC[row+1]); an odd number of passes
// ends with a copy back.  Block = 256 threads; a pass = histogram of the row, scan of the 256 counts, then rounds of
// 256 elements ranked with ballots (order inside the block = index order, so every pass is stable).
__global__ __launch_bounds__(256) void k_sort_long_rows(int m, const int* __restrict__ IC, int* __restrict__ JC,
                                                         float* __restrict__ C, int* __restrict__ JS, float* __restrict__ CS,
                                                         int keyBits) {
  constexpr int NWV = 4;
  __shared__ int hist[256];
  __shared__ int run[256];
  __shared__ int wcnt[NWV][256];
  __shared__ int woff[NWV][256];
  __shared__ int unsorted;
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  for (int row = blockIdx.x; row < m; row += gridDim.x) {
    const int s = IC[row], len = IC[row + 1] - s;
    if (len <= SORT_MAX) continue;                       // block-uniform
    if (tid == 0) unsorted = 0;
    __syncthreads();
    int bad = 0;
    for (int i = tid; i + 1 < len; i += 256) bad |= JC[s + i] > JC[s + i + 1];
    if (bad) atomicOr(&unsorted, 1);
    __syncthreads();
    const bool need = unsorted != 0;
    __syncthreads();
    if (!need) continue;
    int* kin = JC + s; float* vin = C + s;
    int* kout = JS + s; float* vout = CS + s;
    int passes = 0;
    for (int shift = 0; shift < keyBits; shift += 8, ++passes) {
      hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < len; i += 256) atomicAdd(&hist[(kin[i] >> shift) & 255], 1);
      __syncthreads();
      if (w == 0) {                                      // exclusive scan of the 256 counts by one wave, 4 per lane
        int c[4], t = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c[q] = hist[lane * 4 + q]; t += c[q]; }
        int ex = wave_incl_add(t) - t;
#pragma unroll
        for (int q = 0; q < 4; ++q) { run[lane * 4 + q] = ex; ex += c[q]; }
      }
      __syncthreads();
      for (int i0 = 0; i0 < len; i0 += 256) {
#pragma unroll
        for (int q = 0; q < NWV; ++q) wcnt[q][tid] = 0;
        __syncthreads();
        const int i = i0 + tid;
        const bool valid = i < len;
        const int key = valid ? kin[i] : 0;
        const float val = valid ? vin[i] : 0.f;
        const int d = (key >> shift) & 255;
        unsigned long long peers = ballot64(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const unsigned long long mb = ballot64(valid && ((d >> b) & 1));
          peers &= ((d >> b) & 1) ? mb : ~mb;
        }
        const int rank = mask_rank(peers);
        if (valid && rank == 0) wcnt[w][d] = __popcll(peers);
        __syncthreads();
        {
          int r = run[tid];
#pragma unroll
          for (int q = 0; q < NWV; ++q) { woff[q][tid] = r; r += wcnt[q][tid]; }
          run[tid] = r;
        }
        __syncthreads();
        if (valid) { const int dst = woff[w][d] + rank; kout[dst] = key; vout[dst] = val; }
      }
      __threadfence_block();
      __syncthreads();
      int* tk = kin; kin = kout; kout = tk;
      float* tv = vin; vin = vout; vout = tv;
    }
    if (passes & 1) {                                    // the sorted row sits in the scratch segment
      for (int i = tid; i < len; i += 256) { kout[i] = kin[i]; vout[i] = vin[i]; }
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------------
// R-MCL post-step (the step right after the SpGEMM in the reference's loop) on a C that is
// already in HBM: the row rule above + compaction.  CPU: nlibs/tools/util.cc:4-69, nlibs/qrmcl.cc:96-117;
// reference GPU: nlibs/gpus/dutil.cuh:8-80 + thrust::remove (gpu_csr_kernel.cu:218-229,265-270).
// 16 lanes per row (4 rows per wave).  Sums are 16-lane tree reductions in float, so a value that sits within an
// ulp of the threshold can fall on the other side than in the sequential CPU loop (the reference's own GPU path
// has the same property).
// ------------------------------------------------------------------------------------------------
// pass 1: inflate (squares, recomputed where needed, never stored), per-row threshold and kept sum, kept count ->
// cnt[row].  L lanes per row: 16 for short rows, a whole wave once rows average ~100 entries (R-MCL products do);
// four loads in flight per lane.
template <int L>
__global__ __launch_bounds__(256) void k_rmcl_stats(int m, const int* __restrict__ IC, const float* __restrict__ C,
                                                     int* __restrict__ cnt, float* __restrict__ thresh,
                                                     float* __restrict__ ksum) {
  constexpr int RPB = 256 / L;                       // rows per block
  const int gl = threadIdx.x & (L - 1);
  const int nrowsL = (m + RPB - 1) / RPB * RPB;
  for (int row = blockIdx.x * RPB + threadIdx.x / L; row < nrowsL; row += gridDim.x * RPB) {
    const bool live = row < m;
    const int s = live ? IC[row] : 0, e = live ? IC[row + 1] : 0;
    float mx = 0.f, sum = 0.f;
    for (int p = s + gl; p < e; p += 4 * L) {
      float c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = p + i * L < e ? C[p + i * L] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float v = c[i] * c[i]; mx = fmaxf(mx, v); sum += v; }   // inflate: v^2
    }
    mx = rowL_max<L>(mx);
    sum = rowL_sum<L>(sum);
    const float th = rmcl_threshold(sum / (float)(e - s), mx);
    float ks = 0.f;
    int kc = 0;
    for (int p = s + gl; p < e; p += 4 * L) {          // the row was just read: L2
      float c[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = p + i * L < e ? C[p + i * L] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float v = c[i] * c[i]; if (p + i * L < e && v >= th) { ks += v; ++kc; } }
    }
    ks = rowL_sum<L>(ks);
    kc = (int)rowL_sum<L>((float)kc);                  // (< 2^24 per lane group: exact)
    if (live && gl == 0) { cnt[row] = kc; thresh[row] = th; ksum[row] = ks; }
  }
}

// pass 2: stable compaction of the kept entries, normalised, into the new arrays at newPtr[row]
template <int L>
__global__ __launch_bounds__(256) void k_rmcl_compact(int m, const int* __restrict__ IC, const int* __restrict__ JC,
                                                       const float* __restrict__ C, const int* __restrict__ newPtr,
                                                       const float* __restrict__ thresh, const float* __restrict__ ksum,
                                                       int* __restrict__ JN, float* __restrict__ CN) {
  constexpr int RPB = 256 / L;
  const int gl = threadIdx.x & (L - 1);
  const int lane = lane_id();
  const int nrowsL = (m + RPB - 1) / RPB * RPB;
  for (int row = blockIdx.x * RPB + threadIdx.x / L; row < nrowsL; row += gridDim.x * RPB) {
    const bool live = row < m;
    const int s = live ? IC[row] : 0, e = live ? IC[row + 1] : 0;
    const float th = live ? thresh[row] : 0.f, ks = live ? ksum[row] : 1.f;
    int out = live ? newPtr[row] : 0;
    for (int p0 = s; p0 < e; p0 += 2 * L) {            // two steps of L entries, their loads issued together
      float c[2];
      int j[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = p0 + i * L + gl;
        c[i] = p < e ? C[p] : 0.f;
        j[i] = p < e ? JC[p] : 0;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float v = c[i] * c[i];
        const bool keep = p0 + i * L + gl < e && v >= th;
        const unsigned long long mk = ballot64(keep);
        const unsigned long long gm = L == 64 ? mk : ((mk >> (lane - gl)) & ((1ull << L) - 1ull));
        if (keep) { const int o = out + __popcll(gm & ((1ull << gl) - 1ull)); JN[o] = j[i]; CN[o] = v / ks; }
        out += __popcll(gm);
      }
    }
  }
}


// fused path, loop form: {start, end} of every row's kept entries in the scratch arrays, one pair per row
__global__ __launch_bounds__(256) void k_zip_extents(int m, const int* __restrict__ IC, const int* __restrict__ cnt,
                                                      int2* __restrict__ se) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < m) { const int s = IC[r]; se[r] = make_int2(s, s + cnt[r]); }
}

// fused path, after the numeric kernels: the kept entries sit at the front of every row's scratch range
// [IC[row], ...), newPtr is the scan of their counts -- pack them into the new arrays
template <int L>
__global__ __launch_bounds__(256) void k_rmcl_move(int m, const int* __restrict__ IC, const int* __restrict__ newPtr,
                                                    const int* __restrict__ JC, const float* __restrict__ C,
                                                    int* __restrict__ JN, float* __restrict__ CN) {
  constexpr int RPB = 256 / L;
  const int gl = threadIdx.x & (L - 1);
  for (int row = blockIdx.x * RPB + threadIdx.x / L; row < m; row += gridDim.x * RPB) {
    const int src = IC[row], dst = newPtr[row], n = newPtr[row + 1] - dst;
    for (int i = gl; i < n; i += L) { JN[dst + i] = JC[src + i]; CN[dst + i] = C[src + i]; }
  }
}

// fused path, rows the numeric kernels wrote out in full (bin 8: their rows pass through LDS in several pieces): the
// row rule from HBM / L2 by one block per row, the kept entries compacted IN PLACE to the front of the row's range.
__global__ __launch_bounds__(256) void k_rmcl_fix_rows(const int* __restrict__ binPtr, int binLo, int binHi,
                                                        const int* __restrict__ rowIds, const int* __restrict__ IC,
                                                        int* __restrict__ JC, float* __restrict__ C,
                                                        int* __restrict__ cnt) {
  __shared__ float fred[2][4];
  __shared__ int wcnt[4];
  const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  const int first = binPtr[binLo], count = binPtr[binHi] - first;
  for (int q = blockIdx.x; q < count; q += gridDim.x) {
    const int row = rowIds[first + q];
    const int s = IC[row], e = IC[row + 1];
    float mx = 0.f, sum = 0.f;
    for (int p = s + tid; p < e; p += 256) { const float c = C[p], v = c * c; mx = fmaxf(mx, v); sum += v; }
    mx = rowL_max<64>(mx);
    sum = rowL_sum<64>(sum);
    if (lane == 0) { fred[0][w] = mx; fred[1][w] = sum; }
    __syncthreads();
    mx = fmaxf(fmaxf(fred[0][0], fred[0][1]), fmaxf(fred[0][2], fred[0][3]));
    sum = ((fred[1][0] + fred[1][1]) + fred[1][2]) + fred[1][3];
    __syncthreads();
    const float th = rmcl_threshold(sum / (float)(e - s), mx);
    float ks = 0.f;
    for (int p = s + tid; p < e; p += 256) { const float c = C[p], v = c * c; ks += v >= th ? v : 0.f; }
    ks = rowL_sum<64>(ks);
    if (lane == 0) fred[0][w] = ks;
    __syncthreads();
    ks = ((fred[0][0] + fred[0][1]) + fred[0][2]) + fred[0][3];
    int out = 0;                                       // kept so far (block-uniform)
    for (int p0 = s; p0 < e; p0 += 256) {
      const int p = p0 + tid;
      const float c = p < e ? C[p] : 0.f;
      const int j = p < e ? JC[p] : 0;
      asm volatile("" :: "v"(c), "v"(j));              // both loads have landed before the barrier: writes below may hit
      const float v = c * c;                           // positions other lanes of this step read
      const bool keep = p < e && v >= th;
      const unsigned long long mk = ballot64(keep);
      if (lane == 0) wcnt[w] = __popcll(mk);
      __syncthreads();
      int base = out, total = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { base += i < w ? wcnt[i] : 0; total += wcnt[i]; }
      if (keep) { const int o = s + base + mask_rank(mk); JC[o] = j; C[o] = v / ks; }
      out += total;
      __syncthreads();
    }
    if (tid == 0) cnt[row] = out;
  }
}

// self-test of the DPP scans / mask ranks against serial results computed by every lane
__global__ void k_selftest(const int* __restrict__ in, int* __restrict__ bad) {
  __shared__ int buf[WAVE];
  const int lane = lane_id();
  const int v = in[blockIdx.x * WAVE + lane];
  buf[lane] = v;
  const int a = wave_incl_add(v);
  const int mx = wave_incl_max(v & 0xffff);
  const unsigned long long mk = ballot64(v & 1);
  const int rk = mask_rank(mk);
  __syncthreads();
  int ea = 0, em = 0, er = 0;
  for (int i = 0; i <= lane; ++i) { ea += buf[i]; em = max(em, buf[i] & 0xffff); if (i < lane) er += buf[i] & 1; }
  if (a != ea || mx != em || rk != er || wave_sum(v) != __shfl(ea, 63, 64)) atomicAdd(bad, 1);
}

}  // namespace smf
