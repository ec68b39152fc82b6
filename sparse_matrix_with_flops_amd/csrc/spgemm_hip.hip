// spgemm_hip.hip — host side of libspgemm_hip.so: the extern "C" boundary declared in
// include/spgemm_hip.h, the workspace/handle, and the launch sequence of the kernels in
// spgemm_device.hpp.  gfx950 only.  There is NO CPU fallback in this library: without a HIP device
// every entry point fails with SPGEMM_ERR_NODEVICE / SPGEMM_ERR_HIP.
#include "spgemm_device.hpp"
#include "chain_device.hpp"
#include "../../include/spgemm_hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sys/mman.h>
#include <thread>
#include <vector>

using namespace smf;

static_assert(NBINS == SPGEMM_NBINS, "bin count of the kernels and of the C ABI differ");

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(expr)                                                                               \
  do {                                                                                             \
    hipError_t e__ = (expr);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return fail(SPGEMM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

#define CHK(expr)                                                                                  \
  do {                                                                                             \
    int rc__ = (expr);                                                                             \
    if (rc__ != SPGEMM_OK) return rc__;                                                            \
  } while (0)

extern "C" const char* spgemm_hip_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// caching device allocator behind spgemm_hip_malloc/free (the role cudaMalloc/cudaFree play in
// CSR::toGpuCSR/deviceDispose, nlibs/CSR.cc:342-379).  Freed blocks are kept (bounded) and reused,
// so the per-call allocation of C costs a map lookup in steady state instead of a hipMalloc.
// ------------------------------------------------------------------------------------------------
namespace {
// Blocks are cached PER DEVICE: a block freed on device 0 is never handed to an allocation made for device 1.
// INVARIANT the reuse relies on: every entry point of this library returns only after the work it queued has completed
// (hipStreamSynchronize on the handle's stream before the function returns), so a block that comes back through
// spgemm_hip_free / release() is idle; callers that queue their own work on such a block must finish it before freeing
// (the same rule cudaFree imposes implicitly by synchronising).  SPGEMM_POOL_CHECK=1 makes release() synchronise the
// device first (debugging aid for foreign streams).
// One entry point queues work and returns before it has completed: an iteration of hip_gpuRmclIter_device that leaves Mt
// unpacked.  The loop keeps the blocks those kernels read out of the pool until a later wait has covered them.
struct DevPool {
  struct Blk { size_t size; int dev; };
  std::mutex mu;
  std::map<void*, Blk> live;                                   // ptr -> rounded size, owning device
  std::map<int, std::multimap<size_t, void*>> free_blocks;     // device -> rounded size -> ptr
  size_t cached_bytes = 0;
  bool check = false;
  static constexpr size_t kMaxCached = size_t(64) << 30;   // 64 GiB of 288 GiB HBM
  DevPool() { const char* e = getenv("SPGEMM_POOL_CHECK"); check = e && e[0] == '1'; }
  static size_t round_up(size_t b) {
    if (b < 512) b = 512;
    if (b <= (size_t(1) << 20)) return (b + 511) & ~size_t(511);
    const size_t g = size_t(2) << 20;            // 2 MiB granules for big blocks
    return (b + g - 1) / g * g;
  }
  hipError_t alloc(void** p, size_t bytes) {
    const size_t r = round_up(bytes);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    {
      std::lock_guard<std::mutex> lk(mu);
      auto& fb = free_blocks[dev];
      auto it = fb.lower_bound(r);
      // best fit, and a cached block up to 4x the request is still better than a hipMalloc (tens of ms for GB-sized
      // blocks; 288 GB of HBM make the slack affordable): R-MCL shrinks its matrices from one iteration to the next
      if (it != fb.end() && it->first <= 4 * r + (size_t(2) << 20)) {
        *p = it->second;
        live[*p] = Blk{it->first, dev};
        cached_bytes -= it->first;
        fb.erase(it);
        return hipSuccess;
      }
    }
    hipError_t e = hipMalloc(p, r);
    if (e != hipSuccess) {                       // give cached memory back and retry once
      trim(dev);
      e = hipMalloc(p, r);
    }
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu); live[*p] = Blk{r, dev}; }
    return e;
  }
  hipError_t release(void* p) {
    if (!p) return hipSuccess;
    if (check) (void)hipDeviceSynchronize();
    Blk blk{0, 0};
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = live.find(p);
      if (it == live.end()) return hipFree(p);   // not ours: plain hipFree
      blk = it->second;
      live.erase(it);
      if (cached_bytes + blk.size <= kMaxCached) {
        free_blocks[blk.dev].emplace(blk.size, p);
        cached_bytes += blk.size;
        return hipSuccess;
      }
    }
    return hipFree(p);
  }
  // give the cached (idle) blocks of one device back to the driver
  void trim(int dev) {
    std::lock_guard<std::mutex> lk(mu);
    auto& fb = free_blocks[dev];
    for (auto& kv : fb) { cached_bytes -= kv.first; (void)hipFree(kv.second); }
    fb.clear();
  }
  size_t cached_on(int dev) {
    std::lock_guard<std::mutex> lk(mu);
    size_t t = 0;
    for (auto& kv : free_blocks[dev]) t += kv.first;
    return t;
  }
};
DevPool& pool() { static DevPool* p = new DevPool(); return *p; }
}  // namespace

// ------------------------------------------------------------------------------------------------
// handle / workspace
// ------------------------------------------------------------------------------------------------
struct HostMirror {                 // one small device block + its pinned host twin, copied once per phase
  int binPtr[NBINS + 1];
  int err;
  int scratch2[4];                  // counts of the 513-2048 / 2049-4096 / 65-256 / 513-1024 rows when a classification is unpacked
  // work-queue heads of the block-per-row kernels (zeroed with the rest per call).  One 128-byte line EACH: the kernels
  // of a phase run side by side and every dequeue is an atomic on its word -- eight words in one line (next to binPtr /
  // slotBase, which every block reads when it starts) made all of them queue up behind one another in the L2.
  alignas(128) int qctr[8 * 32];
  alignas(128) int qpad_;
  int slotBase[NSLOTS + 1];         // first position of every layout slot in rowIds, [NSLOTS] = m (k_bin_scan)
  unsigned long long totalP;
  unsigned long long nnzC64;
  unsigned long long kept64;        // entries that survive the fused R-MCL prune
  unsigned long long cutTotal;      // one-pass path (chain_device.hpp): total of the cut scan (unused), batches, ticket counter
  int nBatches;
  alignas(128) int chainTicket;
  alignas(128) int chainPad_;
};

struct spgemm_handle {
  int device = 0;
  int numCU = 256;
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // device workspace, sized for cap_m rows
  int cap_m = -1;
  int* rowFlops = nullptr;
  unsigned char* binId = nullptr;
  int* blockHist = nullptr;
  int* blockOff = nullptr;
  int* rowIds = nullptr;
  unsigned long long* tileSum = nullptr;
  unsigned long long* blockP = nullptr;     // per-block sums of row flops (reduced by k_bin_scan)
  int2* sbl = nullptr;                       // per A entry {B-row start, B-row length} (k_entry_lens), cap_nnz entries
  int* batchStart = nullptr;                 // one-pass path: first row of every batch (3*cap_m + 64 entries)
  unsigned long long* chainWords = nullptr;  // ... and the batches' words of the chained prefix
  int pathMode = 0;                          // SPGEMM_PATH: 0 per-row kernels for every row (rounds 1-3); 1 rows up to smallMax
                                             // products through the wave-per-batch kernels, two passes; 2 the same in ONE pass
  int chainCfg = 0;                          // SPGEMM_CHAIN_CFG: kernel geometry (experiments)
  long long cap_nnz = -1;
  HostMirror* dsmall = nullptr;
  HostMirror* hsmall = nullptr;
  HostMirror mirror;                 // host copy taken at the end of the symbolic phase
  const int* cur_rowIds = nullptr;   // row list of the pending symbolic phase (workspace or caller's)
  int sym_m = -1;                    // rows of the pending symbolic phase, -1 = none
  hipEvent_t kev[2 * SPGEMM_NKERNELS];
  bool kused[SPGEMM_NKERNELS];
  unsigned ktiming = 0;              // bit i: launches of kernel i are bracketed by events (spgemm_hip_set_kernel_timing)
  hipEvent_t evMid = nullptr;        // "classification is on the host" marker of the one-shot path
  HostMirror* hmid = nullptr;        // pinned copy taken right after the classify kernels
  // side streams: the per-bin kernels of one phase are independent; SPGEMM_CONCURRENT=1 runs them concurrently
  // (default off: the 155 KB-LDS big-row kernel owns the CUs it runs on, overlap measured slower than serial)
  static constexpr int NSIDE = 4;
  hipStream_t side[NSIDE] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t fork_ev = nullptr;
  hipEvent_t join_ev[NSIDE] = {nullptr, nullptr, nullptr, nullptr};
  bool serial = true;
  int U = 2;                         // rounds in flight per wave in the hash kernels (SPGEMM_U=2|4|8)
  // column bitmaps of the big rows, saved by the symbolic pass for the rank kernel (n <= BIG_WC).  Sized from
  // the previous calls (the number of big rows is only known on the host after the symbolic phase).
  unsigned* bigBitmaps = nullptr;
  int bm_cap = 0;
  int2* spill = nullptr;
  int spill_blocks = 0;
  int bhCap = BH_CAP;
  int bhMargin = 125;                // parking region per hash class, % of products/npass (SPGEMM_BHMARGIN; tests force overflows)
  int h1sym = 32;                    // blocks per CU of the wave-per-row symbolic kernel (SPGEMM_H1SYM, experiments)
  int h1num = 24;                    // ... of the wave-per-row numeric kernel on its 512-slot tables (SPGEMM_H1NUM)
  spgemm_stats stats;
  spgemm_host_api_stats host_api = {};   // phases of the latest hip_CSR_SpMM
  int failNext = 0;                  // test hook (spgemm_hip_debug_fail_next): the next symbolic phase / R-MCL step on this handle fails
  int prev_m = -1;                   // shape and sizes of the previous one-shot SpGEMM (allocation policy of the next one)
  unsigned long long prev_P = 0;
  long long prev_nnzC = -1;
};

static std::mutex g_count_mu;
static std::map<int, int> g_handles_on;          // device -> live handles (the pool is trimmed when the last one goes)

static int ws_free(spgemm_handle* h) {
  hipFree(h->rowFlops); hipFree(h->binId); hipFree(h->blockHist); hipFree(h->blockOff);
  hipFree(h->rowIds); hipFree(h->tileSum); hipFree(h->blockP); hipFree(h->batchStart); hipFree(h->chainWords);
  h->batchStart = nullptr; h->chainWords = nullptr;
  h->rowFlops = nullptr; h->binId = nullptr; h->blockHist = nullptr; h->blockOff = nullptr;
  h->rowIds = nullptr; h->tileSum = nullptr; h->blockP = nullptr; h->cap_m = -1;
  return SPGEMM_OK;
}

static int ws_ensure_entries(spgemm_handle* h, long long nnzA) {
  if (nnzA <= h->cap_nnz) return SPGEMM_OK;
  hipFree(h->sbl);
  h->sbl = nullptr; h->cap_nnz = -1;
  const long long cap = nnzA + nnzA / 8 + 1024;
  HIPCHK(hipMalloc((void**)&h->sbl, sizeof(int2) * (size_t)cap));
  h->cap_nnz = cap;
  return SPGEMM_OK;
}

static int ws_ensure(spgemm_handle* h, int m) {
  if (m <= h->cap_m) return SPGEMM_OK;
  ws_free(h);
  const size_t cap = (size_t)m + (size_t)m / 8 + 1024;
  const size_t nblk = (cap + K1_THREADS - 1) / K1_THREADS;
  const size_t ntile = (cap + 1 + SCAN_TILE - 1) / SCAN_TILE + 1;
  HIPCHK(hipMalloc((void**)&h->rowFlops, sizeof(int) * cap));
  HIPCHK(hipMalloc((void**)&h->binId, cap));
  HIPCHK(hipMalloc((void**)&h->blockHist, sizeof(int) * nblk * NSLOTS));
  HIPCHK(hipMalloc((void**)&h->blockOff, sizeof(int) * nblk * NSLOTS));
  HIPCHK(hipMalloc((void**)&h->rowIds, sizeof(int) * cap));
  HIPCHK(hipMalloc((void**)&h->tileSum, sizeof(unsigned long long) * ntile));
  HIPCHK(hipMalloc((void**)&h->blockP, sizeof(unsigned long long) * nblk));
  HIPCHK(hipMalloc((void**)&h->batchStart, sizeof(int) * (3 * cap + 64)));
  HIPCHK(hipMalloc((void**)&h->chainWords, sizeof(unsigned long long) * (3 * cap + 64)));
  h->cap_m = (int)std::min<size_t>(cap, 0x7fffffff);
  return SPGEMM_OK;
}

extern "C" int spgemm_hip_device_count(int* count) {
  if (!count) return fail(SPGEMM_ERR_ARG, "count is null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(SPGEMM_ERR_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = n;
  return SPGEMM_OK;
}

extern "C" int spgemm_hip_create(spgemm_handle** out, int device) {
  if (!out) return fail(SPGEMM_ERR_ARG, "handle out-pointer is null");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(SPGEMM_ERR_NODEVICE, "no HIP device visible: libspgemm_hip has no CPU fallback");
  if (device < 0 || device >= n) return fail(SPGEMM_ERR_ARG, "device %d out of range [0,%d)", device, n);
  HIPCHK(hipSetDevice(device));
  spgemm_handle* h = new spgemm_handle();
  h->device = device;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  h->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  for (auto& e : h->ev) HIPCHK(hipEventCreate(&e));
  for (auto& e : h->kev) HIPCHK(hipEventCreate(&e));
  for (auto& u : h->kused) u = false;
  { const char* e = getenv("SPGEMM_CONCURRENT"); h->serial = !(e && e[0] == '1'); }
#ifdef SMF_ABLATE
  { const char* e = getenv("SPGEMM_ABLATE"); int v = e ? atoi(e) : 0; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(smf::g_ablate), &v, sizeof(int))); }
#endif
  { const char* e = getenv("SPGEMM_H1SYM"); if (e) { const int c = atoi(e); if (c >= 1 && c <= 32) h->h1sym = c; } }
  { const char* e = getenv("SPGEMM_H1NUM"); if (e) { const int c = atoi(e); if (c >= 1 && c <= 32) h->h1num = c; } }
  { const char* e = getenv("SPGEMM_BHMARGIN"); if (e) { const int c = atoi(e); if (c >= 10 && c <= 400) h->bhMargin = c; } }
  { const char* e = getenv("SPGEMM_BHCAP"); if (e) { const int c = atoi(e); if (c >= 1024 && c <= BH_CAP_MAX) h->bhCap = c; } }
  { const char* e = getenv("SPGEMM_PATH"); if (e) { const int v = atoi(e); if (v >= 0 && v <= 2) h->pathMode = v; } }
  { const char* e = getenv("SPGEMM_CHAIN_CFG"); if (e) h->chainCfg = atoi(e); }
  { const char* e = getenv("SPGEMM_U"); if (e) { const int u = atoi(e); if (u == 2 || u == 4 || u == 8) h->U = u; } }
  for (auto& st : h->side) {
    if (h->serial) st = h->stream;
    else HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  }
  HIPCHK(hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
  for (auto& e : h->join_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipMalloc((void**)&h->dsmall, sizeof(HostMirror)));
  HIPCHK(hipHostMalloc((void**)&h->hsmall, sizeof(HostMirror), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&h->hmid, sizeof(HostMirror), hipHostMallocDefault));
  HIPCHK(hipEventCreateWithFlags(&h->evMid, hipEventDisableTiming));
  { const char* e = getenv("SPGEMM_KTIMING"); if (e) h->ktiming = (unsigned)strtoul(e, nullptr, 0); }
  memset(&h->stats, 0, sizeof(h->stats));
  { std::lock_guard<std::mutex> lk(g_count_mu); ++g_handles_on[device]; }
  // the big-row kernels use more than the default 64 KB of dynamic LDS
  HIPCHK(hipFuncSetAttribute((const void*)k_sym_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BigSymShared)));
  HIPCHK(hipFuncSetAttribute((const void*)k_num_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BigNumShared)));
  HIPCHK(hipFuncSetAttribute((const void*)k_num_bighash, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BigHashShared)));
  *out = h;
  return SPGEMM_OK;
}

extern "C" int spgemm_hip_destroy(spgemm_handle* h) {
  if (!h) return SPGEMM_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (auto& st : h->side) if (st && st != h->stream) { hipStreamSynchronize(st); hipStreamDestroy(st); }
  ws_free(h);
  hipFree(h->bigBitmaps);
  hipFree(h->spill);
  hipFree(h->sbl);
  hipFree(h->dsmall);
  hipHostFree(h->hsmall);
  hipHostFree(h->hmid);
  if (h->evMid) hipEventDestroy(h->evMid);
  for (auto& e : h->ev) if (e) hipEventDestroy(e);
  for (auto& e : h->kev) if (e) hipEventDestroy(e);
  if (h->fork_ev) hipEventDestroy(h->fork_ev);
  for (auto& e : h->join_ev) if (e) hipEventDestroy(e);
  if (h->stream) hipStreamDestroy(h->stream);
  bool last = false;
  { std::lock_guard<std::mutex> lk(g_count_mu); last = --g_handles_on[h->device] <= 0; }
  if (last) pool().trim(h->device);              // cached blocks are idle by the pool's invariant
  delete h;
  return SPGEMM_OK;
}

extern "C" int spgemm_hip_get_stats(const spgemm_handle* h, spgemm_stats* out) {
  if (!h || !out) return fail(SPGEMM_ERR_ARG, "null argument");
  *out = h->stats;
  return SPGEMM_OK;
}

extern "C" void* spgemm_hip_stream(spgemm_handle* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int spgemm_hip_handle_device(const spgemm_handle* h) { return h ? h->device : -1; }

// test hook: the next `count` symbolic phases / fused R-MCL steps on this handle return SPGEMM_ERR_INTERNAL before they queue
// anything (how the tests make ONE rank of a multi-rank group fail: every rank must then leave the step with an error)
extern "C" int spgemm_hip_debug_fail_next(spgemm_handle* h, int count) {
  if (!h) return fail(SPGEMM_ERR_ARG, "null handle");
  h->failNext = count;
  return SPGEMM_OK;
}

extern "C" int spgemm_hip_set_kernel_timing(spgemm_handle* h, unsigned mask) {
  if (!h) return fail(SPGEMM_ERR_ARG, "null handle");
  h->ktiming = mask;
  return SPGEMM_OK;
}

static std::mutex g_default_mu;
static spgemm_handle* g_default = nullptr;
static int default_handle(spgemm_handle** out) {
  std::lock_guard<std::mutex> lk(g_default_mu);
  if (!g_default) CHK(spgemm_hip_create(&g_default, 0));
  *out = g_default;
  return SPGEMM_OK;
}

extern "C" int spgemm_hip_malloc(void** dptr, size_t bytes) {
  if (!dptr) return fail(SPGEMM_ERR_ARG, "dptr is null");
  *dptr = nullptr;
  HIPCHK(pool().alloc(dptr, bytes ? bytes : 1));
  return SPGEMM_OK;
}
extern "C" int spgemm_hip_pool_cached_bytes(int device, size_t* bytes) {
  if (!bytes) return fail(SPGEMM_ERR_ARG, "bytes is null");
  *bytes = pool().cached_on(device);
  return SPGEMM_OK;
}
// give the idle cached blocks of `device` back to the driver (blocks in use are not touched)
extern "C" int spgemm_hip_pool_trim(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(SPGEMM_ERR_ARG, "device %d out of range", device);
  int cur = 0;
  const bool had = hipGetDevice(&cur) == hipSuccess;
  HIPCHK(hipSetDevice(device));
  pool().trim(device);
  if (had) (void)hipSetDevice(cur);
  return SPGEMM_OK;
}
extern "C" int spgemm_hip_free(void* dptr) {
  HIPCHK(pool().release(dptr));
  return SPGEMM_OK;
}
// (copies of 8 MB and more between pageable host memory and the device go through the parallel pinned lanes further down:
//  CSR::toGpuCSR / toCpuCSR of a large matrix then run at the link rate instead of the runtime's single staging thread)
struct CopyJob { void* host; void* dev; size_t bytes; };
static int copy_pageable(int dev, const CopyJob* jobs, int njobs, bool toDevice);
static constexpr size_t kLaneCopyMin = size_t(8) << 20;
static int device_of(const void* dptr) {           // the device a pointer lives on (the current one if the runtime cannot say)
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, dptr) == hipSuccess) return a.device;
  (void)hipGetLastError();
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}

extern "C" int spgemm_hip_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return fail(SPGEMM_ERR_ARG, "null pointer in h2d copy");
  if (bytes >= kLaneCopyMin) {
    const CopyJob j{const_cast<void*>(src), dst, bytes};
    return copy_pageable(device_of(dst), &j, 1, true);
  }
  if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return SPGEMM_OK;
}
extern "C" int spgemm_hip_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return fail(SPGEMM_ERR_ARG, "null pointer in d2h copy");
  if (bytes >= kLaneCopyMin) {
    const CopyJob j{dst, const_cast<void*>(src), bytes};
    return copy_pageable(device_of(src), &j, 1, false);
  }
  if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return SPGEMM_OK;
}

extern "C" int spgemm_hip_memcpy_d2d(void* dst, const void* src, size_t bytes) {
  if (bytes && (!dst || !src)) return fail(SPGEMM_ERR_ARG, "null pointer in d2d copy");
  if (bytes) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
  return SPGEMM_OK;
}

// ------------------------------------------------------------------------------------------------
// launch sequence
// ------------------------------------------------------------------------------------------------
// hipGetLastError() after a batch of launches is how launch failures are noticed; the slot it reads is per thread and keeps
// the last error of ANY earlier HIP call, ours or another library's in this process (RCCL probing devices leaves
// "invalid device ordinal" behind: seen as a spurious failure of the next SpGEMM).  Every launch sequence therefore
// starts by emptying the slot.
// What this can hide: a NON-sticky error left by this library's own previous launch on the thread (a bad launch
// configuration) if that launch site did not look at it.  Every launch sequence here ends with HIPCHK(hipGetLastError())
// before its function returns, so there is no such site -- keep it that way: the post-launch check is mandatory wherever
// this call opens a sequence.  Sticky errors (a fault) are not cleared by reading them and still fail the next call.
static inline void clear_stale_hip_error() { (void)hipGetLastError(); }
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int clampi(long long v, int lo, int hi) { return (int)std::max<long long>(lo, std::min<long long>(v, hi)); }
// grid of a statically scheduled kernel: a multiple of 8 (one contiguous eighth of the work per XCD: xcd_range)
static inline int grid8(long long want, int hi) { return (clampi(want, 8, hi) + 7) & ~7; }

static const char* kKernelNames[SPGEMM_NKERNELS] = {
    "k_row_flops", "k_bin_scan", "k_scatter_rows", "k_sym_g16<32,1>", "k_sym_g16", "k_sym_hash<1,1024>",
    "k_sym_hash<4,4096>", "k_sym_hash<8,8192>", "k_sym_big", "k_scan(3 launches)", "k_num_g16<32,1>", "k_num_g16",
    "k_num_hash<1,1024>", "k_num_hash<4,4096>", "k_num_hash<8,8192>", "k_num_big", "k_num_bighash", "k_cut(3 launches)",
    "k_chain", "k_wbatch<sym>", "k_wbatch<num>"};
extern "C" const char* spgemm_hip_kernel_name(int id) { return (id >= 0 && id < SPGEMM_NKERNELS) ? kKernelNames[id] : ""; }

// every launch is bracketed by two events on its stream (per-kernel durations for bench.py's roofline)
// (only for the kernels selected with spgemm_hip_set_kernel_timing: an event record costs a few microseconds of
// stream time, 40 of them per SpGEMM were 3 % of a step)
struct KTimer {
  spgemm_handle* h; int id; hipStream_t s; bool on;
  KTimer(spgemm_handle* h_, int id_, hipStream_t s_ = nullptr)
      : h(h_), id(id_), s(s_ ? s_ : h_->stream), on((h_->ktiming >> id_) & 1u) {
    if (on) { hipEventRecord(h->kev[2 * id], s); h->kused[id] = true; }
  }
  ~KTimer() { if (on) hipEventRecord(h->kev[2 * id + 1], s); }
};

static void collect_kernel_times(spgemm_handle* h, bool reset) {
  for (int i = 0; i < SPGEMM_NKERNELS; ++i) {
    if (reset) h->stats.ms_kernel[i] = 0.f;
    if (h->kused[i]) { float ms = 0.f; if (hipEventElapsedTime(&ms, h->kev[2 * i], h->kev[2 * i + 1]) == hipSuccess) h->stats.ms_kernel[i] = ms; }
    h->kused[i] = false;
  }
}

// fork: the side streams wait for everything queued on the main stream so far
static void fork_streams(spgemm_handle* h) {
  if (h->serial) return;
  hipEventRecord(h->fork_ev, h->stream);
  for (auto& st : h->side) hipStreamWaitEvent(st, h->fork_ev, 0);
}
// join: the main stream waits for the side streams
static void join_streams(spgemm_handle* h) {
  if (h->serial) return;
  for (int i = 0; i < spgemm_handle::NSIDE; ++i) {
    hipEventRecord(h->join_ev[i], h->side[i]);
    hipStreamWaitEvent(h->stream, h->join_ev[i], 0);
  }
}

#define LAUNCH_NUM(NW, TBL, grid, block, st, ...)                                                               \
  do {                                                                                                         \
    if (pcnt && pmode == 2) hipLaunchKernelGGL((k_num_hash<NW, TBL, 2, 2>), grid, block, 0, st, __VA_ARGS__, pcnt); \
    else if (pcnt) hipLaunchKernelGGL((k_num_hash<NW, TBL, 2, 1>), grid, block, 0, st, __VA_ARGS__, pcnt);     \
    else LAUNCH_U(k_num_hash, NW, TBL, grid, block, st, __VA_ARGS__, (int*)nullptr);                           \
  } while (0)

#define LAUNCH_U(KERN, NW, TBL, grid, block, st, ...)                                              \
  do {                                                                                             \
    if (h->U == 8) hipLaunchKernelGGL((KERN<NW, TBL, 8>), grid, block, 0, st, __VA_ARGS__);        \
    else if (h->U == 4) hipLaunchKernelGGL((KERN<NW, TBL, 4>), grid, block, 0, st, __VA_ARGS__);   \
    else hipLaunchKernelGGL((KERN<NW, TBL, 2>), grid, block, 0, st, __VA_ARGS__);                  \
  } while (0)

// flops + bins: K1, K2, K3.  Also presets IC[row] for rows with 0 / 1 products.
// nnzA >= 0: the per-entry records h->sbl are (re)built first and the row sums read them; nnzA < 0 (entry points of
// the C ABI that are not told nnz(A)): the sums gather through JA -> IB directly.
static int launch_classify(spgemm_handle* h, const int* dIA, const int* dJA, const int* dIB, int m, long long nnzA, int* dIC,
                           const int2* dIBse = nullptr) {
  const int nblk = cdiv(m, K1_THREADS);
  clear_stale_hip_error();
  HIPCHK(hipMemsetAsync(h->dsmall, 0, sizeof(HostMirror), h->stream));
  if (nnzA >= 0) CHK(ws_ensure_entries(h, nnzA));
  if (m > 0) {
    { KTimer t(h, SPGEMM_K_ROW_FLOPS);           // also writes the per-entry records h->sbl when nnz(A) is known
      int2* const sbl = nnzA >= 0 ? h->sbl : (int2*)nullptr;
      const int cap = (int)std::max(nnzA, 0ll);
      if (dIBse) hipLaunchKernelGGL(k_row_flops<true>, dim3(nblk), dim3(K1_THREADS), 0, h->stream, m, dIA, dJA, dIB, dIBse,
                                    sbl, cap, h->rowFlops, h->binId, h->blockHist, h->blockP, dIC);
      else hipLaunchKernelGGL(k_row_flops<false>, dim3(nblk), dim3(K1_THREADS), 0, h->stream, m, dIA, dJA, dIB, dIBse,
                              sbl, cap, h->rowFlops, h->binId, h->blockHist, h->blockP, dIC); }
    { KTimer t(h, SPGEMM_K_BIN_SCAN);
      hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, h->stream, nblk, h->blockHist, h->blockOff,
                         h->dsmall->binPtr, h->dsmall->slotBase, h->blockP, &h->dsmall->totalP); }
    { KTimer t(h, SPGEMM_K_SCATTER);
      hipLaunchKernelGGL(k_scatter_rows, dim3(nblk), dim3(K1_THREADS), 0, h->stream, m, h->binId, h->blockOff,
                         h->dsmall->slotBase, h->rowIds); }
  }
  HIPCHK(hipGetLastError());
  return SPGEMM_OK;
}

// symbolic pass over bins 2..8 (bins 0/1 were preset by K1).  Bin sizes are not known on the host
// here (no sync): grids are capped by the CU count and every block strides / dequeues over its bin.
static int launch_symbolic(spgemm_handle* h, const int* dIA, const int* dJB,
                           int m, int n, const int* rowIds, int* dIC, int minBin = 0, bool skipH1A = false) {
  const int2* sbl = h->sbl;
  if (m <= 0) return SPGEMM_OK;
  clear_stale_hip_error();
  const int* bp = h->dsmall->binPtr;
  int* err = &h->dsmall->err;
  int* qc = h->dsmall->qctr;
  const int cu = h->numCU;
  fork_streams(h);
  { hipStream_t st = h->side[3]; KTimer t(h, SPGEMM_K_SYM_BIG, st);
    hipLaunchKernelGGL(k_sym_big, dim3(clampi(m, 1, cu)), dim3(BIG_THREADS), sizeof(BigSymShared), st, bp, 8,
                       rowIds, dIA, sbl, dJB, n, dIC, h->bigBitmaps, h->bm_cap, qc + 0 * 32); }
  if (minBin <= 7) { hipStream_t st = h->side[2]; KTimer t(h, SPGEMM_K_SYM_HASH8, st);
    LAUNCH_U(k_sym_hash, 8, 8192, dim3(clampi(m, 1, cu * 3)), dim3(512), st, bp, 7, rowIds, dIA,
             sbl, dJB, h->rowFlops, dIC, err, qc + 1 * 32); }
  if (minBin <= 6) { hipStream_t st = h->side[2]; KTimer t(h, SPGEMM_K_SYM_HASH4, st);
    // (one launch for both layout slots of bin 6: split like the numeric side it measured 15 % slower)
    LAUNCH_U(k_sym_hash, 4, 4096, dim3(clampi(m, 1, cu * 8)), dim3(256), st, bp, 6, rowIds, dIA,
             sbl, dJB, h->rowFlops, dIC, err, qc + 2 * 32); }
  if (minBin <= 5) { hipStream_t st = h->side[1]; KTimer t(h, SPGEMM_K_SYM_HASH1, st);
    const int* sb = h->dsmall->slotBase;         // bin 5 = two layout slots: table 512 up to 256 products, 1024 above
    if (!skipH1A)
    LAUNCH_U(k_sym_hash, 1, 512, dim3(grid8(m, cu * h->h1sym)), dim3(64), st, sb, SLOT_H1A, rowIds, dIA,
             sbl, dJB, h->rowFlops, dIC, err, qc + 3 * 32);
    LAUNCH_U(k_sym_hash, 1, 1024, dim3(grid8(m, cu * h->h1sym)), dim3(64), st, sb, SLOT_H1B, rowIds, dIA,
             sbl, dJB, h->rowFlops, dIC, err, qc + 3 * 32); }
  if (minBin <= 4) { hipStream_t st = h->side[0]; KTimer t(h, SPGEMM_K_SYM_G16, st);
    hipLaunchKernelGGL((k_sym_g16<128, 4>), dim3(grid8(cdiv(m, 16), cu * 16)), dim3(256), 0, st, bp, 4, 5,
                       rowIds, dIA, sbl, dJB, h->rowFlops, dIC, err); }
  if (minBin <= 3) { KTimer t(h, SPGEMM_K_SYM_SMALL4);           // 2..16 products: the same 16-lane flattened kernel, one round per row
    hipLaunchKernelGGL((k_sym_g16<32, 1>), dim3(grid8(cdiv(m, 16), cu * 16)), dim3(256), 0, h->stream, bp, 2, 4,
                       rowIds, dIA, sbl, dJB, h->rowFlops, dIC, err); }
  join_streams(h);
  HIPCHK(hipGetLastError());
  return SPGEMM_OK;
}

// exclusive scan of cnt[0..m) in place, cnt[m] = total; 64-bit total lands in *dTotal
static int launch_scan(spgemm_handle* h, int* cnt, int m, unsigned long long* dTotal, bool clampTotal = true) {
  const int ntiles = std::max(1, cdiv(m, SCAN_TILE));
  clear_stale_hip_error();
  KTimer t(h, SPGEMM_K_SCAN);
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(SCAN_THREADS), 0, h->stream, m, cnt, h->tileSum);
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, h->stream, ntiles, h->tileSum, dTotal);
  hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(SCAN_THREADS), 0, h->stream, m, cnt, h->tileSum, dTotal, clampTotal ? 1 : 0);
  HIPCHK(hipGetLastError());
  return SPGEMM_OK;
}

static int launch_numeric(spgemm_handle* h, const int* dIA, const float* dA,
                          const int* dJB, const float* dB, int n, const int* rowIds, const int* hostBinPtr,
                          const int* dIC, int* dJC, float* dC, int* pcnt = nullptr, int pmode = 1, int minBin = 0,
                          bool skipH1A = false) {
  // pcnt != nullptr: fused R-MCL prune -- every row leaves only its kept, normalised entries at the front of its range
  // of dJC/dC and their count in pcnt[row] (rows of bin 8 are written in full and fixed up in place right behind).
  // pmode 2: no symbolic pass ran below bin 8: dIC holds the prefix sums of the rows' product counts there, exact counts
  // for the rows of bin 8
  const int2* sbl = h->sbl;
  clear_stale_hip_error();
  const int* bp = h->dsmall->binPtr;
  int* err = &h->dsmall->err;
  int* qc = h->dsmall->qctr;
  const int cu = h->numCU;
  auto rows = [&](int lo, int hi) { return lo < minBin ? 0 : hostBinPtr[hi] - hostBinPtr[lo]; };   // (lo = the lowest bin of the launch)
  fork_streams(h);
  if (rows(8, 9) > 0) {
    hipStream_t st = h->side[3];
    if (n <= BIG_WC) { KTimer t(h, SPGEMM_K_NUM_BIG, st);
      hipLaunchKernelGGL(k_num_big, dim3(clampi(rows(8, 9), 1, cu)), dim3(BIG_THREADS), sizeof(BigNumShared), st,
                         bp, 8, rowIds, dIA, sbl, dA, dJB, dB, n, dIC, dJC, dC, err, h->bigBitmaps, h->bm_cap,
                         qc + 4 * 32);
    } else { KTimer t(h, SPGEMM_K_NUM_BIGHASH, st);
      const int blocks = clampi(rows(8, 9), 1, cu);
      if (blocks > h->spill_blocks) {                  // parking space of multi-pass rows: 1 MB per block, kept
        hipFree(h->spill);
        h->spill = nullptr;
        h->spill_blocks = 0;
        if (hipMalloc((void**)&h->spill, (size_t)blocks * BH_SPILL * sizeof(int2)) == hipSuccess) h->spill_blocks = blocks;
        else (void)hipGetLastError();                  // without it those rows walk once per pass
      }
      hipLaunchKernelGGL(k_num_bighash, dim3(blocks), dim3(BIG_THREADS), sizeof(BigHashShared), st,
                         bp, 8, rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, qc + 4 * 32, h->rowFlops,
                         h->spill_blocks >= blocks ? h->spill : (int2*)nullptr, BH_SPILL, h->bhCap, h->bhMargin); }
    if (pcnt) hipLaunchKernelGGL(k_rmcl_fix_rows, dim3(clampi(rows(8, 9), 1, cu * 8)), dim3(256), 0, st,
                                 bp, 8, 9, rowIds, dIC, dJC, dC, pcnt);
  }
  if (rows(7, 8) > 0) { hipStream_t st = h->side[2]; KTimer t(h, SPGEMM_K_NUM_HASH8, st);
    LAUNCH_NUM(8, 8192, dim3(clampi(rows(7, 8), 1, cu * 2)), dim3(512), st, bp, 7,
             rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, qc + 5 * 32, h->rowFlops); }
  if (rows(6, 7) > 0) { hipStream_t st = h->side[2]; KTimer t(h, SPGEMM_K_NUM_HASH4, st);
    const int* sb = h->dsmall->slotBase;
    const int* hs_ = h->mirror.slotBase;
    const int na = hs_[SLOT_H4A + 1] - hs_[SLOT_H4A], nb = hs_[SLOT_H4B + 1] - hs_[SLOT_H4B];
    if (na > 0) LAUNCH_NUM(4, 2048, dim3(clampi(na, 1, cu * 7)), dim3(256), st, sb, SLOT_H4A,
                         rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, qc + 6 * 32, h->rowFlops);
    if (nb > 0) LAUNCH_NUM(4, 4096, dim3(clampi(nb, 1, cu * 4)), dim3(256), st, sb, SLOT_H4B,
                         rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, qc + 7 * 32, h->rowFlops); }
  if (rows(5, 6) > 0) { hipStream_t st = h->side[1]; KTimer t(h, SPGEMM_K_NUM_HASH1, st);
    const int* sb = h->dsmall->slotBase;
    const int* hs_ = h->mirror.slotBase;
    const int na = hs_[SLOT_H1A + 1] - hs_[SLOT_H1A], nb = hs_[SLOT_H1B + 1] - hs_[SLOT_H1B];
    if (na > 0 && !skipH1A) LAUNCH_NUM(1, 512, dim3(grid8(na, cu * h->h1num)), dim3(64), st, sb, SLOT_H1A,
                         rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, qc + 7 * 32, h->rowFlops);
    if (nb > 0) LAUNCH_NUM(1, 1024, dim3(grid8(nb, cu * 16)), dim3(64), st, sb, SLOT_H1B,
                         rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, qc + 7 * 32, h->rowFlops); }
  if (rows(4, 5) > 0) { hipStream_t st = h->side[0]; KTimer t(h, SPGEMM_K_NUM_G16, st);
    if (pcnt && pmode == 2) hipLaunchKernelGGL((k_num_g16<128, 4, 2>), dim3(grid8(cdiv(rows(4, 5), 16), cu * 16)), dim3(256), 0, st,
                                               bp, 4, 5, rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, h->rowFlops, pcnt);
    else if (pcnt) hipLaunchKernelGGL((k_num_g16<128, 4, 1>), dim3(grid8(cdiv(rows(4, 5), 16), cu * 16)), dim3(256), 0, st,
                                      bp, 4, 5, rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, h->rowFlops, pcnt);
    else hipLaunchKernelGGL((k_num_g16<128, 4>), dim3(grid8(cdiv(rows(4, 5), 16), cu * 16)), dim3(256), 0, st,
                            bp, 4, 5, rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, h->rowFlops, (int*)nullptr); }
  if (minBin <= 1 && rows(1, 4) > 0) { KTimer t(h, SPGEMM_K_NUM_SMALL4);   // 1..16 products: 16-lane flattened kernel, one round per row
    if (pcnt && pmode == 2) hipLaunchKernelGGL((k_num_g16<32, 1, 2>), dim3(grid8(cdiv(rows(1, 4), 16), cu * 16)), dim3(256), 0, h->stream,
                                               bp, 1, 4, rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, h->rowFlops, pcnt);
    else if (pcnt) hipLaunchKernelGGL((k_num_g16<32, 1, 1>), dim3(grid8(cdiv(rows(1, 4), 16), cu * 16)), dim3(256), 0, h->stream,
                                      bp, 1, 4, rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, h->rowFlops, pcnt);
    else hipLaunchKernelGGL((k_num_g16<32, 1>), dim3(grid8(cdiv(rows(1, 4), 16), cu * 16)), dim3(256), 0, h->stream,
                            bp, 1, 4, rowIds, dIA, sbl, dA, dJB, dB, dIC, dJC, dC, err, h->rowFlops, (int*)nullptr); }
  join_streams(h);
  HIPCHK(hipGetLastError());
  return SPGEMM_OK;
}

// ---- rows up to smallMax products dealt across rows (chain_device.hpp): wave-per-batch kernels
// geometry: waves per block, table slots per wave, blocks per CU; rows up to smallMax products go through k_wbatch, the
// bins from minBin on keep their per-row kernels (skipH1A: of bin 5 only the 257-512 slot)
struct ChainCfg { int nw, tbl, bpc, smallMax, minBin; bool skipH1A; };
static const ChainCfg kChainCfgs[] = {
    {4, 1280, 3, 512, 6, false},
    {6, 1280, 2, 512, 6, false},
    {4, 1024, 3, 256, 5, true},
    {4, 2048, 2, 512, 6, false},
    {8, 1280, 1, 512, 6, false},
};
static const ChainCfg& chain_cfg(const spgemm_handle* h) {
  const int n = (int)(sizeof(kChainCfgs) / sizeof(kChainCfgs[0]));
  return kChainCfgs[h->chainCfg >= 0 && h->chainCfg < n ? h->chainCfg : 0];
}
#ifndef SMF_WB_MAXR
#define SMF_WB_MAXR 2          // rounds of 64 products gathered before the first insert (measured 8 / 4 / 2: 1.68 / 1.42 / 1.33 ms)
#endif

static int launch_cut(spgemm_handle* h, int m) {
  const ChainCfg& c = chain_cfg(h);
  const int ntiles = std::max(1, cdiv(m, SCAN_TILE));
  const CutParams p{c.smallMax, c.tbl - wb_slots(c.smallMax), WAVE};
  clear_stale_hip_error();
  KTimer t(h, SPGEMM_K_CUT);
  hipLaunchKernelGGL(k_cut_sums, dim3(ntiles), dim3(SCAN_THREADS), 0, h->stream, m, h->rowFlops, p, h->tileSum);
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, h->stream, ntiles, h->tileSum, &h->dsmall->cutTotal);
  hipLaunchKernelGGL(k_cut_apply, dim3(ntiles), dim3(SCAN_THREADS), 0, h->stream, m, h->rowFlops, p, h->tileSum,
                     h->batchStart, h->chainWords, &h->dsmall->nBatches, &h->dsmall->chainTicket);
  HIPCHK(hipGetLastError());
  return SPGEMM_OK;
}

template <int NW, int TBL, int MODE>
static void launch_wbatch_t(spgemm_handle* h, int blocks, int m, const int* dIA, const float* dA, const int* dJB, const float* dB,
                            int smallMax, int* dIC, int* dJC, float* dC, long long capC) {
  typedef WaveBatchShared<NW, TBL> Sh;
  static bool attr = false;                              // (more than 64 KB of dynamic LDS needs the attribute once)
  if (!attr) { (void)hipFuncSetAttribute((const void*)k_wbatch<NW, TBL, MODE, SMF_WB_MAXR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Sh)); attr = true; }
  hipLaunchKernelGGL((k_wbatch<NW, TBL, MODE, SMF_WB_MAXR>), dim3(blocks), dim3(WAVE * NW), sizeof(Sh), h->stream, m, dIA, h->sbl, dA, dJB, dB,
                     h->rowFlops, smallMax, h->batchStart, &h->dsmall->nBatches, h->chainWords, &h->dsmall->chainTicket, dIC, dJC, dC,
                     capC, &h->dsmall->nnzC64, &h->dsmall->err);
}

template <int MODE>
static int launch_wbatch(spgemm_handle* h, int m, const int* dIA, const float* dA, const int* dJB, const float* dB,
                         int* dIC, int* dJC, float* dC, long long capC) {
  const ChainCfg& c = chain_cfg(h);
  clear_stale_hip_error();
  KTimer t(h, MODE == WB_SYM ? SPGEMM_K_WB_SYM : MODE == WB_NUM ? SPGEMM_K_WB_NUM : SPGEMM_K_CHAIN);
  const int blocks = clampi((long long)h->numCU * c.bpc, 1, 1 << 20);
  switch (h->chainCfg) {
    case 1: launch_wbatch_t<6, 1280, MODE>(h, blocks, m, dIA, dA, dJB, dB, c.smallMax, dIC, dJC, dC, capC); break;
    case 2: launch_wbatch_t<4, 1024, MODE>(h, blocks, m, dIA, dA, dJB, dB, c.smallMax, dIC, dJC, dC, capC); break;
    case 3: launch_wbatch_t<4, 2048, MODE>(h, blocks, m, dIA, dA, dJB, dB, c.smallMax, dIC, dJC, dC, capC); break;
    case 4: launch_wbatch_t<8, 1280, MODE>(h, blocks, m, dIA, dA, dJB, dB, c.smallMax, dIC, dJC, dC, capC); break;
    default: launch_wbatch_t<4, 1280, MODE>(h, blocks, m, dIA, dA, dJB, dB, c.smallMax, dIC, dJC, dC, capC); break;
  }
  HIPCHK(hipGetLastError());
  return SPGEMM_OK;
}

static int check_common(const void* a, const void* b, const void* c, int nnz, const char* who) {
  if (nnz < 0) return fail(SPGEMM_ERR_ARG, "%s: negative nnz", who);
  if (!a) return fail(SPGEMM_ERR_ARG, "%s: rowPtr is null", who);
  if (nnz > 0 && (!b || !c)) return fail(SPGEMM_ERR_ARG, "%s: colInd/values null with nnz=%d", who, nnz);
  return SPGEMM_OK;
}

// `pre` (optional) supplies a classification made earlier by hip_gpuFlopsClassify:
// row ids grouped by bin + dflops + hv.
struct PreClass { const int* drowIds; const int* hv; const int* dflops; };
static int unpack_classification(spgemm_handle* h, int m, const PreClass& pre, int* dIC);

// phase 1: classify + symbolic + scan; one host sync at the end (nnzC, bin sizes, error flags)
static int symbolic_phase(spgemm_handle* h, const int* dIA, const int* dJA, int nnzA, const int* dIB, const int* dJB,
                          int m, int k, int n, const PreClass* pre, int* dIC, int* nnzCp) {
  (void)k;
  if (h->failNext > 0) { --h->failNext; return fail(SPGEMM_ERR_INTERNAL, "forced failure (spgemm_hip_debug_fail_next)"); }
  hipStream_t s = h->stream;
  hipEventRecord(h->ev[0], s);
  h->cur_rowIds = h->rowIds;
  if (pre) {
    CHK(unpack_classification(h, m, *pre, dIC));
    h->cur_rowIds = pre->drowIds;
    CHK(ws_ensure_entries(h, nnzA));                 // the caller's classification carries no per-entry records
    if (nnzA > 0)
      hipLaunchKernelGGL(k_entry_lens, dim3(clampi(cdiv(nnzA, 256), 1, h->numCU * 32)), dim3(256), 0, s, nnzA, dJA, dIB, h->sbl);
  } else {
    CHK(launch_classify(h, dIA, dJA, dIB, m, nnzA, dIC));
  }
  hipEventRecord(h->ev[1], s);
  CHK(launch_symbolic(h, dIA, dJB, m, n, h->cur_rowIds, dIC));
  hipEventRecord(h->ev[2], s);
  if (m > 0) CHK(launch_scan(h, dIC, m, &h->dsmall->nnzC64));
  else HIPCHK(hipMemsetAsync(dIC, 0, sizeof(int), s));
  hipEventRecord(h->ev[3], s);
  if (hipMemcpyAsync(h->hsmall, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return fail(SPGEMM_ERR_HIP, "symbolic phase failed: %s", hipGetErrorString(hipGetLastError()));
  h->mirror = *h->hsmall;
  const HostMirror& hm = h->mirror;
  if (hm.err) return fail(SPGEMM_ERR_INTERNAL, "device invariant broken in symbolic phase (flags=%d)", hm.err);
  if (hm.nnzC64 > 0x7fffffffULL) return fail(SPGEMM_ERR_OVERFLOW, "nnz(C)=%llu does not fit int32 CSR", hm.nnzC64);
  spgemm_stats& st = h->stats;
  st.total_flops = (long long)hm.totalP;
  st.nnzC = (int)hm.nnzC64;
  for (int b = 0; b < NBINS; ++b) st.bin_rows[b] = hm.binPtr[b + 1] - hm.binPtr[b];
  hipEventElapsedTime(&st.ms_classify, h->ev[0], h->ev[1]);
  hipEventElapsedTime(&st.ms_symbolic, h->ev[1], h->ev[2]);
  hipEventElapsedTime(&st.ms_scan_alloc, h->ev[2], h->ev[3]);
  st.ms_numeric = 0.f;
  st.ms_total = st.ms_classify + st.ms_symbolic + st.ms_scan_alloc;
  collect_kernel_times(h, true);
  h->sym_m = m;
  *nnzCp = (int)hm.nnzC64;
  return SPGEMM_OK;
}

// room for the big rows' bitmaps next time (rank kernel only)
static void grow_bitmaps(spgemm_handle* h, int n) {
  const int nbig = h->mirror.binPtr[NBINS] - h->mirror.binPtr[NBINS - 1];
  if (n <= BIG_WC && nbig > h->bm_cap) {
    hipFree(h->bigBitmaps);
    h->bigBitmaps = nullptr;
    const int cap = nbig + nbig / 4 + 16;
    if (hipMalloc((void**)&h->bigBitmaps, (size_t)cap * BIG_WORDS * sizeof(unsigned)) == hipSuccess) h->bm_cap = cap;
    else { h->bm_cap = 0; (void)hipGetLastError(); }
  }
}

// phase 2: numeric into caller-provided dJC/dC; one host sync at the end (error flags)
static int numeric_phase(spgemm_handle* h, const int* dIA, const int* dJA, const float* dA, const int* dIB,
                         const int* dJB, const float* dB, int m, int n, const int* dIC, int* dJC, float* dC) {
  if (h->sym_m != m) return fail(SPGEMM_ERR_ARG, "numeric phase without a matching symbolic phase on this handle");
  hipStream_t s = h->stream;
  const int nnzC = (int)h->mirror.nnzC64;
  hipEventRecord(h->ev[4], s);
  if (m > 0 && nnzC > 0) {
    if (!dJC || !dC) return fail(SPGEMM_ERR_ARG, "output buffers are null");
    CHK(launch_numeric(h, dIA, dA, dJB, dB, n, h->cur_rowIds, h->mirror.binPtr, dIC, dJC, dC));
  }
  hipEventRecord(h->ev[5], s);
  if (hipMemcpyAsync(&h->hsmall->err, &h->dsmall->err, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return fail(SPGEMM_ERR_HIP, "numeric phase failed: %s", hipGetErrorString(hipGetLastError()));
  h->sym_m = -1;
#ifndef SMF_ABLATE
  if (h->hsmall->err) return fail(SPGEMM_ERR_INTERNAL, "device invariant broken in numeric phase (flags=%d)", h->hsmall->err);
#endif
  spgemm_stats& st = h->stats;
  hipEventElapsedTime(&st.ms_numeric, h->ev[4], h->ev[5]);
  st.ms_total += st.ms_numeric;
  collect_kernel_times(h, false);
  grow_bitmaps(h, n);
  return SPGEMM_OK;
}

static int spgemm_device(spgemm_handle* h, const int* dIA, const int* dJA, const float* dA, int nnzA,
                         const int* dIB, const int* dJB, const float* dB, int nnzB, int m, int k, int n,
                         const PreClass* pre, int** dICp, int** dJCp, float** dCp, int* nnzCp) {
  if (!dICp || !dJCp || !dCp || !nnzCp) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  *dICp = nullptr; *dJCp = nullptr; *dCp = nullptr; *nnzCp = 0;
  if (m < 0 || k < 0 || n < 0) return fail(SPGEMM_ERR_ARG, "negative dimension m=%d k=%d n=%d", m, k, n);
  CHK(check_common(dIA, dJA, dA, nnzA, "A"));
  CHK(check_common(dIB, dJB, dB, nnzB, "B"));
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  CHK(ws_ensure(h, m));
  int* dIC = nullptr;
  int* dJC = nullptr;
  float* dC = nullptr;
  auto cleanup = [&](int rc) { pool().release(dIC); pool().release(dJC); pool().release(dC); return rc; };
  HIPCHK(pool().alloc((void**)&dIC, sizeof(int) * ((size_t)m + 1)));
  int nnzC = 0;
  if (!pre && m > 0) {
    // One-shot path, no host round trip between the phases: the classification (bin sizes, P) is copied to the host
    // while the symbolic kernels run; colInd/values are allocated with P entries (an upper bound of nnz(C), 8*P bytes --
    // HBM is 288 GB) so that the numeric kernels can be queued right behind the scan; nnz(C) and the error flags are
    // read once, at the end.  A product with P beyond the bound below takes the two-phase path.
    hipStream_t s = h->stream;
    hipEventRecord(h->ev[0], s);
    h->cur_rowIds = h->rowIds;
    int rc = launch_classify(h, dIA, dJA, dIB, m, nnzA, dIC);
    if (rc) return cleanup(rc);
    // Rows up to ChainCfg::smallMax products are dealt across rows by the wave-per-batch kernels (chain_device.hpp, round 4):
    // pathMode 1 in two passes (counts, scan, entries), pathMode 2 in ONE pass (no symbolic pass and no scan for them: a chained
    // prefix over the batches places them); the bins above keep their per-row kernels.
    const bool chainPath = h->pathMode == 2, wb2 = h->pathMode == 1;
    const int minBin = (chainPath || wb2) ? chain_cfg(h).minBin : 0;
    const bool skipH1A = (chainPath || wb2) && chain_cfg(h).skipH1A;
    if ((chainPath || wb2) && (rc = launch_cut(h, m))) return cleanup(rc);
    hipEventRecord(h->ev[1], s);
    if (hipMemcpyAsync(h->hmid, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipEventRecord(h->evMid, s) != hipSuccess)
      return cleanup(fail(SPGEMM_ERR_HIP, "classification copy failed: %s", hipGetErrorString(hipGetLastError())));
    if (wb2 && (rc = launch_wbatch<WB_SYM>(h, m, dIA, nullptr, dJB, nullptr, dIC, nullptr, nullptr, 0))) return cleanup(rc);
    if ((rc = launch_symbolic(h, dIA, dJB, m, n, h->cur_rowIds, dIC, minBin, skipH1A))) return cleanup(rc);
    hipEventRecord(h->ev[2], s);
    if (!chainPath && (rc = launch_scan(h, dIC, m, &h->dsmall->nnzC64))) return cleanup(rc);
    hipEventRecord(h->ev[3], s);
    if (hipEventSynchronize(h->evMid) != hipSuccess)
      return cleanup(fail(SPGEMM_ERR_HIP, "classification failed: %s", hipGetErrorString(hipGetLastError())));
    const HostMirror mid = *h->hmid;
    const unsigned long long P = mid.totalP;
    // A product that compresses (real web graphs: nnz(C)/P ~ 0.5) would hold twice the memory it needs when sized by P.
    // The handle remembers the previous call: the same shape and product count with nnz(C) < 3/4 P sends this call down
    // the two-phase path (exact allocation, one more host round trip -- cheap next to the halved footprint).
    const bool compressive = h->prev_m == m && h->prev_P == P && h->prev_nnzC >= 0 &&
                             (double)h->prev_nnzC < 0.75 * (double)P;
    // (one-pass path: a compressive product gets exactly the entries its previous, identical-looking call produced; the
    // kernels never write past that and say so if it was not enough -- the call is then redone the two-phase way)
    if (P <= (1ull << 30) && (!compressive || chainPath)) {
      const size_t capC = compressive ? (size_t)std::max<long long>(h->prev_nnzC, 1ll) : (size_t)std::max<unsigned long long>(P, 1ull);
      if (hipSuccess != pool().alloc((void**)&dJC, sizeof(int) * capC) ||
          hipSuccess != pool().alloc((void**)&dC, sizeof(float) * capC))
        return cleanup(fail(SPGEMM_ERR_HIP, "device allocation of C (%llu entries) failed", P));
      hipEventRecord(h->ev[4], s);
      h->mirror = mid;                                 // bin sizes for the numeric launch grids
      if (chainPath && (rc = launch_wbatch<WB_CHAIN>(h, m, dIA, dA, dJB, dB, dIC, dJC, dC, (long long)capC))) return cleanup(rc);
      if (wb2 && P > 0 && (rc = launch_wbatch<WB_NUM>(h, m, dIA, dA, dJB, dB, dIC, dJC, dC, (long long)capC))) return cleanup(rc);
      if (P > 0 && (rc = launch_numeric(h, dIA, dA, dJB, dB, n, h->cur_rowIds, mid.binPtr, dIC, dJC, dC, nullptr, 1, minBin, skipH1A))) return cleanup(rc);
      hipEventRecord(h->ev[5], s);
      if (hipMemcpyAsync(h->hsmall, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess)
        return cleanup(fail(SPGEMM_ERR_HIP, "SpGEMM failed: %s", hipGetErrorString(hipGetLastError())));
      h->mirror = *h->hsmall;
      h->sym_m = -1;
      const HostMirror& hm = h->mirror;
      if (chainPath && (hm.err & ERRF_CHAIN_CAP)) {
        // C sized from the previous call was too small for this product (same shape and product count, more distinct
        // columns): nothing was written out of bounds; forget the guess and compute the product the two-phase way
        pool().release(dJC); pool().release(dC);
        dJC = nullptr; dC = nullptr;
        h->prev_m = -1; h->prev_nnzC = -1;
        rc = symbolic_phase(h, dIA, dJA, nnzA, dIB, dJB, m, k, n, nullptr, dIC, &nnzC);
        if (rc) return cleanup(rc);
        if (hipSuccess != pool().alloc((void**)&dJC, sizeof(int) * (size_t)std::max(nnzC, 1)) ||
            hipSuccess != pool().alloc((void**)&dC, sizeof(float) * (size_t)std::max(nnzC, 1)))
          return cleanup(fail(SPGEMM_ERR_HIP, "device allocation of C (%d entries) failed", nnzC));
        rc = numeric_phase(h, dIA, dJA, dA, dIB, dJB, dB, m, n, dIC, dJC, dC);
        if (rc) return cleanup(rc);
        *dICp = dIC; *dJCp = dJC; *dCp = dC; *nnzCp = nnzC;
        return SPGEMM_OK;
      }
#ifndef SMF_ABLATE
      if (hm.err) return cleanup(fail(SPGEMM_ERR_INTERNAL, "device invariant broken (flags=%d)", hm.err));
#endif
      if (hm.nnzC64 > 0x7fffffffULL) return cleanup(fail(SPGEMM_ERR_OVERFLOW, "nnz(C)=%llu does not fit int32 CSR", hm.nnzC64));
      spgemm_stats& st = h->stats;
      st.total_flops = (long long)hm.totalP;
      st.nnzC = (int)hm.nnzC64;
      for (int b = 0; b < NBINS; ++b) st.bin_rows[b] = hm.binPtr[b + 1] - hm.binPtr[b];
      hipEventElapsedTime(&st.ms_classify, h->ev[0], h->ev[1]);
      hipEventElapsedTime(&st.ms_symbolic, h->ev[1], h->ev[2]);
      hipEventElapsedTime(&st.ms_scan_alloc, h->ev[2], h->ev[4]);
      hipEventElapsedTime(&st.ms_numeric, h->ev[4], h->ev[5]);
      hipEventElapsedTime(&st.ms_total, h->ev[0], h->ev[5]);
      collect_kernel_times(h, true);
      grow_bitmaps(h, n);
      h->prev_m = m; h->prev_P = P; h->prev_nnzC = (long long)hm.nnzC64;
      *dICp = dIC; *dJCp = dJC; *dCp = dC; *nnzCp = (int)hm.nnzC64;
      return SPGEMM_OK;
    }
    // P too large for a speculative allocation: finish the symbolic phase the two-phase way (kernels already queued)
    if (chainPath) {
      // (the one-pass path queued only the symbolic kernels of its counted rows: run the whole two-pass pipeline instead)
      rc = symbolic_phase(h, dIA, dJA, nnzA, dIB, dJB, m, k, n, nullptr, dIC, &nnzC);
      if (rc) return cleanup(rc);
      h->prev_m = m; h->prev_P = P; h->prev_nnzC = nnzC;
      if (hipSuccess != pool().alloc((void**)&dJC, sizeof(int) * (size_t)std::max(nnzC, 1)) ||
          hipSuccess != pool().alloc((void**)&dC, sizeof(float) * (size_t)std::max(nnzC, 1)))
        return cleanup(fail(SPGEMM_ERR_HIP, "device allocation of C (%d entries) failed", nnzC));
      rc = numeric_phase(h, dIA, dJA, dA, dIB, dJB, dB, m, n, dIC, dJC, dC);
      if (rc) return cleanup(rc);
      *dICp = dIC; *dJCp = dJC; *dCp = dC; *nnzCp = nnzC;
      return SPGEMM_OK;
    }
    if (hipMemcpyAsync(h->hsmall, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return cleanup(fail(SPGEMM_ERR_HIP, "symbolic phase failed: %s", hipGetErrorString(hipGetLastError())));
    h->mirror = *h->hsmall;
    if (h->mirror.err) return cleanup(fail(SPGEMM_ERR_INTERNAL, "device invariant broken in symbolic phase (flags=%d)", h->mirror.err));
    if (h->mirror.nnzC64 > 0x7fffffffULL) return cleanup(fail(SPGEMM_ERR_OVERFLOW, "nnz(C)=%llu does not fit int32 CSR", h->mirror.nnzC64));
    h->stats.total_flops = (long long)h->mirror.totalP;
    h->stats.nnzC = (int)h->mirror.nnzC64;
    for (int b = 0; b < NBINS; ++b) h->stats.bin_rows[b] = h->mirror.binPtr[b + 1] - h->mirror.binPtr[b];
    hipEventElapsedTime(&h->stats.ms_classify, h->ev[0], h->ev[1]);
    hipEventElapsedTime(&h->stats.ms_symbolic, h->ev[1], h->ev[2]);
    hipEventElapsedTime(&h->stats.ms_scan_alloc, h->ev[2], h->ev[3]);
    h->stats.ms_total = h->stats.ms_classify + h->stats.ms_symbolic + h->stats.ms_scan_alloc;
    collect_kernel_times(h, true);
    h->sym_m = m;
    nnzC = (int)h->mirror.nnzC64;
    h->prev_m = m; h->prev_P = P; h->prev_nnzC = nnzC;
    if (hipSuccess != pool().alloc((void**)&dJC, sizeof(int) * (size_t)std::max(nnzC, 1)) ||
        hipSuccess != pool().alloc((void**)&dC, sizeof(float) * (size_t)std::max(nnzC, 1)))
      return cleanup(fail(SPGEMM_ERR_HIP, "device allocation of C (%d entries) failed", nnzC));
    rc = numeric_phase(h, dIA, dJA, dA, dIB, dJB, dB, m, n, dIC, dJC, dC);
    if (rc) return cleanup(rc);
    *dICp = dIC; *dJCp = dJC; *dCp = dC; *nnzCp = nnzC;
    return SPGEMM_OK;
  }
  int rc = symbolic_phase(h, dIA, dJA, nnzA, dIB, dJB, m, k, n, pre, dIC, &nnzC);
  if (rc) return cleanup(rc);
  if (hipSuccess != pool().alloc((void**)&dJC, sizeof(int) * (size_t)std::max(nnzC, 1)) ||
      hipSuccess != pool().alloc((void**)&dC, sizeof(float) * (size_t)std::max(nnzC, 1)))
    return cleanup(fail(SPGEMM_ERR_HIP, "device allocation of C (%d entries) failed", nnzC));
  rc = numeric_phase(h, dIA, dJA, dA, dIB, dJB, dB, m, n, dIC, dJC, dC);
  if (rc) return cleanup(rc);
  *dICp = dIC; *dJCp = dJC; *dCp = dC; *nnzCp = nnzC;
  return SPGEMM_OK;
}

extern "C" int hip_gpuSpMM(spgemm_handle* h, const int* dIA, const int* dJA, const float* dA, int nnzA,
                           const int* dIB, const int* dJB, const float* dB, int nnzB, int m, int k, int n,
                           int** dIC, int** dJC, float** dC, int* nnzC) {
  return spgemm_device(h, dIA, dJA, dA, nnzA, dIB, dJB, dB, nnzB, m, k, n, nullptr, dIC, dJC, dC, nnzC);
}

extern "C" int hip_spgemm_symbolic(spgemm_handle* h, const int* dIA, const int* dJA, int nnzA, const int* dIB,
                                   const int* dJB, int nnzB, int m, int k, int n, int* dIC, int* nnzC) {
  if (!h) return fail(SPGEMM_ERR_ARG, "hip_spgemm_symbolic needs a handle");
  if (!dIC || !nnzC) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  if (m < 0 || k < 0 || n < 0) return fail(SPGEMM_ERR_ARG, "negative dimension");
  CHK(check_common(dIA, dJA, dJA, nnzA, "A"));
  CHK(check_common(dIB, dJB, dJB, nnzB, "B"));
  HIPCHK(hipSetDevice(h->device));
  CHK(ws_ensure(h, m));
  return symbolic_phase(h, dIA, dJA, nnzA, dIB, dJB, m, k, n, nullptr, dIC, nnzC);
}

extern "C" int hip_spgemm_numeric(spgemm_handle* h, const int* dIA, const int* dJA, const float* dA, int nnzA,
                                  const int* dIB, const int* dJB, const float* dB, int nnzB, int m, int k, int n,
                                  const int* dIC, int* dJC, float* dC) {
  (void)k;
  if (!h) return fail(SPGEMM_ERR_ARG, "hip_spgemm_numeric needs a handle");
  if (!dIC) return fail(SPGEMM_ERR_ARG, "dIC is null");
  CHK(check_common(dIA, dJA, dA, nnzA, "A"));
  CHK(check_common(dIB, dJB, dB, nnzB, "B"));
  HIPCHK(hipSetDevice(h->device));
  return numeric_phase(h, dIA, dJA, dA, dIB, dJB, dB, m, n, dIC, dJC, dC);
}

extern "C" int hip_csr_row_flops(spgemm_handle* h, const int* dIA, const int* dJA, const int* dIB, int m,
                                 int* dRowFlops, long long* total_flops) {
  if (!h) CHK(default_handle(&h));
  if (m < 0 || !dIA || (m > 0 && !dRowFlops)) return fail(SPGEMM_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(h->device));
  h->sym_m = -1;                                   // the classification workspace is about to be overwritten
  CHK(ws_ensure(h, m));
  int* tmpIC = nullptr;
  HIPCHK(pool().alloc((void**)&tmpIC, sizeof(int) * ((size_t)m + 1)));
  int rc = launch_classify(h, dIA, dJA, dIB, m, -1, tmpIC);
  if (rc == SPGEMM_OK && m > 0 &&
      hipMemcpyAsync(dRowFlops, h->rowFlops, sizeof(int) * (size_t)m, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)
    rc = fail(SPGEMM_ERR_HIP, "copy of row flops failed");
  if (rc == SPGEMM_OK && (hipMemcpyAsync(h->hsmall, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                          hipStreamSynchronize(h->stream) != hipSuccess))
    rc = fail(SPGEMM_ERR_HIP, "row flops failed: %s", hipGetErrorString(hipGetLastError()));
  pool().release(tmpIC);
  if (rc) return rc;
  collect_kernel_times(h, true);
  if (total_flops) *total_flops = (long long)h->hsmall->totalP;
  return SPGEMM_OK;
}

// ------------------------------------------------------------------------------------------------
// classification API (reference-shaped outputs)
// ------------------------------------------------------------------------------------------------
static void hv_from_binptr(const int* binPtr, int hv[SPGEMM_HV_LEN], int* hv_len) {
  // reference bins (dqueueId): 1:{0} 2:{1} 3:{2..4} 4:{5..16} 5:{17..64} 6:{65..512} 7:{>512}; element 0 of the
  // (m+1)-long bin array is a dummy with 0 flops (bin 1).  Internal bins 6,7,8 together are reference bin 7.
  int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  cnt[1] = 1 + (binPtr[1] - binPtr[0]);
  cnt[2] = binPtr[2] - binPtr[1];
  cnt[3] = binPtr[3] - binPtr[2];
  cnt[4] = binPtr[4] - binPtr[3];
  cnt[5] = binPtr[5] - binPtr[4];
  cnt[6] = binPtr[6] - binPtr[5];
  cnt[7] = binPtr[NBINS] - binPtr[6];
  hv[0] = 0;
  int maxbin = 1;
  for (int b = 0; b < 8; ++b) { hv[b + 1] = hv[b] + cnt[b]; if (cnt[b] > 0) maxbin = b; }
  *hv_len = maxbin + 2;
}

extern "C" int hip_gpuFlopsClassify(spgemm_handle* h, const int* dIA, const int* dJA, const int* dIB, int m, int k,
                                    int** drowIds, int** dflops, int hv[SPGEMM_HV_LEN], int* hv_len,
                                    long long* total_flops) {
  (void)k;
  if (!drowIds || !dflops || !hv || !hv_len) return fail(SPGEMM_ERR_ARG, "null output pointer");
  *drowIds = nullptr; *dflops = nullptr;
  if (m < 0) return fail(SPGEMM_ERR_ARG, "negative m");
  if (!dIA) return fail(SPGEMM_ERR_ARG, "A.rowPtr is null");
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  h->sym_m = -1;
  CHK(ws_ensure(h, m));
  int *ids = nullptr, *fl = nullptr, *tmpIC = nullptr;
  auto cleanup = [&](int rc) { pool().release(ids); pool().release(fl); pool().release(tmpIC); return rc; };
  HIPCHK(pool().alloc((void**)&ids, sizeof(int) * (size_t)std::max(m, 1)));
  if (hipSuccess != pool().alloc((void**)&fl, sizeof(int) * ((size_t)m + 1)) ||
      hipSuccess != pool().alloc((void**)&tmpIC, sizeof(int) * ((size_t)m + 1)))
    return cleanup(fail(SPGEMM_ERR_HIP, "device allocation failed"));
  int rc = launch_classify(h, dIA, dJA, dIB, m, -1, tmpIC);
  if (rc) return cleanup(rc);
  hipStream_t s = h->stream;
  if (m > 0) {
    hipMemcpyAsync(ids, h->rowIds, sizeof(int) * (size_t)m, hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(k_gather_flops, dim3(cdiv(m, 256)), dim3(256), 0, s, m, h->rowIds, h->rowFlops, fl);
    if ((rc = launch_scan(h, fl, m, &h->dsmall->nnzC64, false))) return cleanup(rc);   // wraps: differences stay exact
  } else {
    hipMemsetAsync(fl, 0, sizeof(int), s);
  }
  if (hipMemcpyAsync(h->hsmall, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return cleanup(fail(SPGEMM_ERR_HIP, "classify failed: %s", hipGetErrorString(hipGetLastError())));
  hv_from_binptr(h->hsmall->binPtr, hv, hv_len);
  if (total_flops) *total_flops = (long long)h->hsmall->totalP;
  h->stats.total_flops = (long long)h->hsmall->totalP;
  for (int b = 0; b < NBINS; ++b) h->stats.bin_rows[b] = h->hsmall->binPtr[b + 1] - h->hsmall->binPtr[b];
  collect_kernel_times(h, true);
  pool().release(tmpIC);
  *drowIds = ids;
  *dflops = fl;
  return SPGEMM_OK;
}

// rebuild the internal view (rowFlops by row, 9-bin binPtr, IC presets) from a caller-held classification
__global__ void k_unpack_classify(int m, const int* __restrict__ rowIds, const int* __restrict__ dflops,
                                  int* __restrict__ rowFlops, int* __restrict__ IC, int lo6, int* __restrict__ n67) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  int is6 = 0, is7 = 0, is5a = 0, is6a = 0;
  if (q < m) {
    const int r = rowIds[q];
    // dflops is an int scan (like the reference's): differences stay exact modulo 2^32
    const unsigned f = (unsigned)dflops[q + 1] - (unsigned)dflops[q];
    rowFlops[r] = f > 0x7fffffffu ? 0x7fffffff : (int)f;
    if (f <= 1u) IC[r] = (int)f;
    is6 = (q >= lo6 && f <= 2048u) ? 1 : 0;
    is7 = (q >= lo6 && f > 2048u && f <= 4096u) ? 1 : 0;
    is5a = (f > 64u && f <= (unsigned)smf::H1A_MAX) ? 1 : 0;
    is6a = (q >= lo6 && f <= (unsigned)smf::H4A_MAX) ? 1 : 0;
  }
  const unsigned long long m6 = __ballot(is6), m7 = __ballot(is7), m5 = __ballot(is5a), m6a = __ballot(is6a);
  if (smf::lane_id() == 0) {
    if (m6a) atomicAdd(&n67[3], __popcll(m6a));
    if (m6) atomicAdd(&n67[0], __popcll(m6));
    if (m7) atomicAdd(&n67[1], __popcll(m7));
    if (m5) atomicAdd(&n67[2], __popcll(m5));
  }
}

__global__ void k_binptr_from_hv(int* binPtr, int* slotBase, int h2, int h3, int h4, int h5, int h6, int h7, int m,
                                 const int* n67) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    binPtr[0] = 0; binPtr[1] = h2 - 1; binPtr[2] = h3 - 1; binPtr[3] = h4 - 1; binPtr[4] = h5 - 1;
    binPtr[5] = h6 - 1; binPtr[6] = h7 - 1; binPtr[7] = h7 - 1 + n67[0]; binPtr[8] = binPtr[7] + n67[1]; binPtr[9] = m;
    // the two layout slots of bin 5 (the classification came from hip_gpuFlopsClassify: <=256 products first)
    slotBase[smf::SLOT_H1A] = binPtr[5]; slotBase[smf::SLOT_H1B] = binPtr[5] + n67[2]; slotBase[smf::SLOT_H4A] = binPtr[6];
    slotBase[smf::SLOT_H4B] = binPtr[6] + n67[3]; slotBase[smf::SLOT_H8] = binPtr[7];
  }
}

static int unpack_classification(spgemm_handle* h, int m, const PreClass& pre, int* dIC) {
  if (!pre.drowIds || !pre.hv || !pre.dflops) return fail(SPGEMM_ERR_ARG, "classification pointers are null");
  const int* hv = pre.hv;
  for (int b = 0; b < 8; ++b)
    if (hv[b] > hv[b + 1]) return fail(SPGEMM_ERR_ARG, "hv is not monotone");
  if (hv[8] != m + 1 || hv[1] != 0 || hv[2] < 1) return fail(SPGEMM_ERR_ARG, "hv does not describe %d rows", m);
  clear_stale_hip_error();
  HIPCHK(hipMemsetAsync(h->dsmall, 0, sizeof(HostMirror), h->stream));
  if (m > 0) {
    int* n67 = h->dsmall->scratch2;
    hipLaunchKernelGGL(k_unpack_classify, dim3(cdiv(m, 256)), dim3(256), 0, h->stream, m, pre.drowIds, pre.dflops,
                       h->rowFlops, dIC, hv[7] - 1, n67);
    hipLaunchKernelGGL(k_binptr_from_hv, dim3(1), dim3(64), 0, h->stream, h->dsmall->binPtr, h->dsmall->slotBase, hv[2], hv[3], hv[4], hv[5],
                       hv[6], hv[7], m, n67);
  }
  HIPCHK(hipGetLastError());
  return SPGEMM_OK;
}

extern "C" int hip_sgpuSpMM(spgemm_handle* h, const int* dIA, const int* dJA, const float* dA, int nnzA,
                            const int* dIB, const int* dJB, const float* dB, int nnzB, int m, int k, int n,
                            const int* drowIds, const int hv[SPGEMM_HV_LEN], const int* dflops, int** dIC, int** dJC,
                            float** dC, int* nnzC) {
  PreClass pre{drowIds, hv, dflops};
  return spgemm_device(h, dIA, dJA, dA, nnzA, dIB, dJB, dB, nnzB, m, k, n, &pre, dIC, dJC, dC, nnzC);
}

// ------------------------------------------------------------------------------------------------
// host arrays in / host arrays out
// ------------------------------------------------------------------------------------------------
static int validate_host_csr(const int* I, const int* J, int rows, int cols, int nnz, const char* who) {
  if (I[0] != 0) return fail(SPGEMM_ERR_INPUT, "%s: rowPtr[0]=%d, expected 0", who, I[0]);
  for (int i = 0; i < rows; ++i)
    if (I[i + 1] < I[i]) return fail(SPGEMM_ERR_INPUT, "%s: rowPtr decreases at row %d", who, i);
  if (I[rows] != nnz) return fail(SPGEMM_ERR_INPUT, "%s: rowPtr[rows]=%d but nnz=%d", who, I[rows], nnz);
  for (int p = 0; p < nnz; ++p)
    if ((unsigned)J[p] >= (unsigned)cols) return fail(SPGEMM_ERR_INPUT, "%s: colInd[%d]=%d outside [0,%d)", who, p, J[p], cols);
  return SPGEMM_OK;
}

// Pageable host memory <-> device at PCIe speed.  The caller's arrays are plain malloc memory (the reference's
// CSR::dispose() is free(), nlibs/CSR.h:323-327), and for the outputs they are fresh: a blocking hipMemcpy moves them
// through the runtime's single staging thread (page faults of 1.8 GB of untouched memory included) at a fraction of the
// link rate.  Here T threads each own a contiguous slice, a private stream and two pinned 4 MB slots: while one slot is
// on the link the thread moves the other between slot and destination, so the faults and the memcpy of the slices run
// in parallel and the link stays busy.  Slots are allocated once per process.
namespace {
struct CopyLanes {
  static constexpr int T = 8;
  static constexpr size_t SLOT = size_t(4) << 20;
  char* slot[T][2] = {};
  hipEvent_t done[T][2] = {};                         // "the copy out of this slot has finished" (host -> device direction)
  hipStream_t st[T] = {};
  int device = -1;
  std::mutex mu;
  int ensure(int dev) {
    if (device == dev && slot[0][0]) return SPGEMM_OK;
    release();
    for (int t = 0; t < T; ++t) {
      HIPCHK(hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking));
      for (int b = 0; b < 2; ++b) {
        HIPCHK(hipHostMalloc((void**)&slot[t][b], SLOT, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&done[t][b], hipEventDisableTiming));
      }
    }
    device = dev;
    return SPGEMM_OK;
  }
  void release() {
    for (int t = 0; t < T; ++t) {
      for (int b = 0; b < 2; ++b) {
        if (slot[t][b]) hipHostFree(slot[t][b]);
        if (done[t][b]) hipEventDestroy(done[t][b]);
        slot[t][b] = nullptr; done[t][b] = nullptr;
      }
      if (st[t]) hipStreamDestroy(st[t]);
      st[t] = nullptr;
    }
    device = -1;
  }
};
CopyLanes& lanes() { static CopyLanes* l = new CopyLanes(); return *l; }

// one thread's slice; toDevice: host -> device, else device -> host
hipError_t lane_copy(int dev, hipStream_t st, char* const slots[2], hipEvent_t const done[2], char* host, char* devp, size_t bytes,
                     bool toDevice) {
  hipError_t e = hipSetDevice(dev);
  if (e != hipSuccess) return e;
  const size_t S = CopyLanes::SLOT;
  const size_t n = (bytes + S - 1) / S;
  if (toDevice) {
    for (size_t c = 0; c < n; ++c) {
      const size_t off = c * S, len = std::min(S, bytes - off);
      char* sl = slots[c & 1];
      // wait only for THIS slot's previous transfer (chunk c-2): chunk c-1 stays on the link while this one is staged
      // (round 3 synchronised the stream here, which also waited for chunk c-1: the two slots gave no double buffering)
      if (c >= 2 && (e = hipEventSynchronize(done[c & 1])) != hipSuccess) return e;
      memcpy(sl, host + off, len);
      if ((e = hipMemcpyAsync(devp + off, sl, len, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
      if ((e = hipEventRecord(done[c & 1], st)) != hipSuccess) return e;
    }
    return hipStreamSynchronize(st);
  }
  // The destination is fresh malloc memory: without help every 4 KB page of it costs a fault inside the memcpy below (and
  // the faults of the threads contend for the process's address-space lock: measured 11 GB/s for 1.8 GB over 8 threads).
  // Ask for huge pages once and populate every chunk in one call right before it is filled -- while the next chunk is on
  // the link.  Both are hints (errors ignored: older kernels).
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
  auto page_in = [](char* p, size_t len, int advice) {
    const uintptr_t a = ((uintptr_t)p + 4095) & ~uintptr_t(4095), b = ((uintptr_t)p + len) & ~uintptr_t(4095);
    if (b > a) (void)madvise((void*)a, b - a, advice);
  };
#ifdef MADV_HUGEPAGE
  page_in(host, bytes, MADV_HUGEPAGE);
#endif
  size_t pendOff = 0, pendLen = 0;
  int pendSlot = -1;
  for (size_t c = 0; c < n; ++c) {
    const size_t off = c * S, len = std::min(S, bytes - off);
    if ((e = hipMemcpyAsync(slots[c & 1], devp + off, len, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if (pendSlot >= 0) {                                                      // the previous chunk, already landed
      page_in(host + pendOff, pendLen, MADV_POPULATE_WRITE);
      memcpy(host + pendOff, slots[pendSlot], pendLen);
    } else {
      page_in(host, std::min(bytes, S), MADV_POPULATE_WRITE);         // nothing to move yet: prepare the first chunk
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    pendOff = off; pendLen = len; pendSlot = (int)(c & 1);
  }
  if (pendSlot >= 0) { page_in(host + pendOff, pendLen, MADV_POPULATE_WRITE); memcpy(host + pendOff, slots[pendSlot], pendLen); }
  return hipSuccess;
}
}  // namespace

// several (host, device, bytes) pairs in one go: the bytes of all of them are dealt to the lanes in 4 MB chunks
static int copy_pageable(int dev, const CopyJob* jobs, int njobs, bool toDevice) {
  size_t total = 0;
  for (int i = 0; i < njobs; ++i) total += jobs[i].bytes;
  if (!total) return SPGEMM_OK;
  if (total < (size_t(1) << 20)) {                       // small: not worth the threads
    for (int i = 0; i < njobs; ++i)
      if (jobs[i].bytes) HIPCHK(toDevice ? hipMemcpy(jobs[i].dev, jobs[i].host, jobs[i].bytes, hipMemcpyHostToDevice)
                                        : hipMemcpy(jobs[i].host, jobs[i].dev, jobs[i].bytes, hipMemcpyDeviceToHost));
    return SPGEMM_OK;
  }
  CopyLanes& L = lanes();
  std::lock_guard<std::mutex> lk(L.mu);
  {
    int cur = dev;                                       // the lanes' streams belong to `dev`: make them there
    (void)hipGetDevice(&cur);
    HIPCHK(hipSetDevice(dev));
    const int rc = L.ensure(dev);
    if (cur != dev) (void)hipSetDevice(cur);
    CHK(rc);
  }
  // cut every job into T slices (all lanes work on the same job at the same time: neighbouring pages, one job after the other)
  hipError_t errs[CopyLanes::T];
  std::vector<std::thread> th;
  for (int t = 0; t < CopyLanes::T; ++t) {
    errs[t] = hipSuccess;
    th.emplace_back([&, t] {
      for (int i = 0; i < njobs && errs[t] == hipSuccess; ++i) {
        const size_t b = jobs[i].bytes;
        const size_t per = ((b + CopyLanes::T - 1) / CopyLanes::T + 4095) & ~size_t(4095);
        const size_t lo = std::min(b, per * (size_t)t), hi = std::min(b, lo + per);
        if (hi > lo) errs[t] = lane_copy(dev, L.st[t], L.slot[t], L.done[t], (char*)jobs[i].host + lo, (char*)jobs[i].dev + lo, hi - lo, toDevice);
      }
    });
  }
  for (auto& x : th) x.join();
  for (int t = 0; t < CopyLanes::T; ++t)
    if (errs[t] != hipSuccess) return fail(SPGEMM_ERR_HIP, "pipelined %s copy failed: %s", toDevice ? "h2d" : "d2h", hipGetErrorString(errs[t]));
  return SPGEMM_OK;
}

extern "C" int hip_CSR_SpMM(const int* IA, const int* JA, const float* A, int nnzA, const int* IB, const int* JB,
                            const float* B, int nnzB, int** IC, int** JC, float** C, int* nnzC, int m, int k, int n) {
  if (!IC || !JC || !C || !nnzC) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  *IC = nullptr; *JC = nullptr; *C = nullptr; *nnzC = 0;
  if (m < 0 || k < 0 || n < 0) return fail(SPGEMM_ERR_ARG, "negative dimension");
  CHK(check_common(IA, JA, A, nnzA, "A"));
  CHK(check_common(IB, JB, B, nnzB, "B"));
  CHK(validate_host_csr(IA, JA, m, k, nnzA, "A"));
  CHK(validate_host_csr(IB, JB, k, n, nnzB, "B"));
  spgemm_handle* h = nullptr;
  CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  int *dIA = nullptr, *dJA = nullptr, *dIB = nullptr, *dJB = nullptr, *dIC = nullptr, *dJC = nullptr;
  float *dA = nullptr, *dB = nullptr, *dC = nullptr;
  int *hIC = nullptr, *hJC = nullptr;
  float* hC = nullptr;
  auto cleanup = [&](int rc) {
    for (void* p : {(void*)dIA, (void*)dJA, (void*)dA, (void*)dIB, (void*)dJB, (void*)dB, (void*)dIC, (void*)dJC, (void*)dC})
      pool().release(p);
    if (rc != SPGEMM_OK) { free(hIC); free(hJC); free(hC); }
    return rc;
  };
  const bool same = (IA == IB && JA == JB && A == B && nnzA == nnzB && m == k);  // C = A*A: upload once
  const auto t0 = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  int rc;
  std::vector<CopyJob> up;
#define UP(dst, src, bytes)                                                                        \
  if ((rc = spgemm_hip_malloc((void**)&dst, (bytes)))) return cleanup(rc);                          \
  up.push_back(CopyJob{(void*)(src), (void*)dst, (size_t)(bytes)});
  UP(dIA, IA, sizeof(int) * ((size_t)m + 1));
  UP(dJA, JA, sizeof(int) * (size_t)nnzA);
  UP(dA, A, sizeof(float) * (size_t)nnzA);
  if (!same) {
    UP(dIB, IB, sizeof(int) * ((size_t)k + 1));
    UP(dJB, JB, sizeof(int) * (size_t)nnzB);
    UP(dB, B, sizeof(float) * (size_t)nnzB);
  }
#undef UP
  if ((rc = copy_pageable(h->device, up.data(), (int)up.size(), true))) return cleanup(rc);
  h->host_api.ms_h2d = ms_since(t0);
  const auto t1 = std::chrono::steady_clock::now();
  int nz = 0;
  rc = hip_gpuSpMM(h, dIA, dJA, dA, nnzA, same ? dIA : dIB, same ? dJA : dJB, same ? dA : dB, nnzB, m, k, n, &dIC,
                   &dJC, &dC, &nz);
  if (rc) return cleanup(rc);
  h->host_api.ms_device = ms_since(t1);
  const auto t2 = std::chrono::steady_clock::now();
  // outputs must be malloc()ed: the caller's CSR::dispose() is free() (nlibs/CSR.h:323-327)
  hIC = (int*)malloc(sizeof(int) * ((size_t)m + 1));
  hJC = (int*)malloc(sizeof(int) * (size_t)std::max(nz, 1));
  hC = (float*)malloc(sizeof(float) * (size_t)std::max(nz, 1));
  if (!hIC || !hJC || !hC) return cleanup(fail(SPGEMM_ERR_NOMEM, "host malloc of C failed"));
  const CopyJob down[3] = {{hIC, dIC, sizeof(int) * ((size_t)m + 1)}, {hJC, dJC, sizeof(int) * (size_t)nz},
                           {hC, dC, sizeof(float) * (size_t)nz}};
  if ((rc = copy_pageable(h->device, down, 3, false))) return cleanup(rc);
  h->host_api.ms_d2h = ms_since(t2);
  h->host_api.ms_total = ms_since(t0);
  h->host_api.bytes_h2d = 0;
  for (auto& j : up) h->host_api.bytes_h2d += (long long)j.bytes;
  h->host_api.bytes_d2h = (long long)(down[0].bytes + down[1].bytes + down[2].bytes);
  *IC = hIC; *JC = hJC; *C = hC; *nnzC = nz;
  return cleanup(SPGEMM_OK);
}

// phases of the latest hip_CSR_SpMM (host arrays in, host arrays out) on the default handle
extern "C" int spgemm_hip_host_api_stats(spgemm_host_api_stats* out) {
  if (!out) return fail(SPGEMM_ERR_ARG, "null argument");
  spgemm_handle* h = nullptr;
  CHK(default_handle(&h));
  *out = h->host_api;
  return SPGEMM_OK;
}

// ------------------------------------------------------------------------------------------------
// R-MCL
// ------------------------------------------------------------------------------------------------
// nnzIn >= 0: the caller knows nnz(C) (it is on the host after every SpGEMM of this library) and the choice of lanes per
// row needs no device read; nnzIn < 0: one 4-byte copy fetches it.
static int rmcl_prune_impl(spgemm_handle* h, int m, long long nnzIn, const int* dIC, const int* dJC, const float* dC,
                           int** dIN, int** dJN, float** dCN, int* nnzN) {
  if (!dIN || !dJN || !dCN || !nnzN) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  *dIN = nullptr; *dJN = nullptr; *dCN = nullptr; *nnzN = 0;
  if (m < 0 || !dIC) return fail(SPGEMM_ERR_ARG, "bad argument");
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  h->sym_m = -1;                                   // scan scratch and the small device block are shared with a pending phase
  clear_stale_hip_error();
  CHK(ws_ensure(h, m));
  int* newPtr = nullptr; int* JN = nullptr; float* CN = nullptr; float* th = nullptr; float* ks = nullptr;
  auto cleanup = [&](int rc) { pool().release(newPtr); pool().release(JN); pool().release(CN); pool().release(th); pool().release(ks); return rc; };
  auto hipfail = [&](const char* what) { return cleanup(fail(SPGEMM_ERR_HIP, "rmcl prune: %s: %s", what, hipGetErrorString(hipGetLastError()))); };
  if (hipSuccess != pool().alloc((void**)&newPtr, sizeof(int) * ((size_t)m + 1)) ||
      hipSuccess != pool().alloc((void**)&th, sizeof(float) * (size_t)std::max(m, 1)) ||
      hipSuccess != pool().alloc((void**)&ks, sizeof(float) * (size_t)std::max(m, 1)))
    return hipfail("device allocation failed");
  hipStream_t s = h->stream;
  if (hipMemsetAsync(&h->dsmall->nnzC64, 0, sizeof(unsigned long long), s) != hipSuccess) return hipfail("memset");
  // lanes per row: 16 for short rows; a whole wave once rows average ~100 entries
  if (m > 0 && nnzIn < 0) {
    int tmp = 0;
    if (hipMemcpyAsync(&tmp, dIC + m, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess)
      return hipfail("reading nnz");
    nnzIn = tmp;
  }
  const bool wide = m > 0 && nnzIn >= 96ll * m;
  const int grid = wide ? clampi(cdiv(m, 4), 1, h->numCU * 32) : clampi(cdiv(m, 16), 1, h->numCU * 16);
  if (m > 0) {
    if (wide) hipLaunchKernelGGL(k_rmcl_stats<64>, dim3(grid), dim3(256), 0, s, m, dIC, dC, newPtr, th, ks);
    else hipLaunchKernelGGL(k_rmcl_stats<16>, dim3(grid), dim3(256), 0, s, m, dIC, dC, newPtr, th, ks);
    int rc = launch_scan(h, newPtr, m, &h->dsmall->nnzC64);
    if (rc) return cleanup(rc);
  } else {
    hipMemsetAsync(newPtr, 0, sizeof(int), s);
  }
  if (hipMemcpyAsync(&h->hsmall->nnzC64, &h->dsmall->nnzC64, sizeof(unsigned long long), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return hipfail("kept-entry count");
  const int nz = (int)h->hsmall->nnzC64;
  if (hipSuccess != pool().alloc((void**)&JN, sizeof(int) * (size_t)std::max(nz, 1)) ||
      hipSuccess != pool().alloc((void**)&CN, sizeof(float) * (size_t)std::max(nz, 1)))
    return hipfail("device allocation failed");
  if (m > 0 && nz > 0) {
    if (wide) hipLaunchKernelGGL(k_rmcl_compact<64>, dim3(grid), dim3(256), 0, s, m, dIC, dJC, dC, newPtr, th, ks, JN, CN);
    else hipLaunchKernelGGL(k_rmcl_compact<16>, dim3(grid), dim3(256), 0, s, m, dIC, dJC, dC, newPtr, th, ks, JN, CN);
    if (hipGetLastError() != hipSuccess) return hipfail("compaction launch");
  }
  if (hipStreamSynchronize(s) != hipSuccess) return hipfail("compaction");
  pool().release(th); pool().release(ks);
  *dIN = newPtr; *dJN = JN; *dCN = CN; *nnzN = nz;
  return SPGEMM_OK;
}

extern "C" int hip_rmcl_prune(spgemm_handle* h, int m, const int* dIC, const int* dJC, float* dC, int** dIN, int** dJN,
                              float** dCN, int* nnzN) {
  return rmcl_prune_impl(h, m, -1, dIC, dJC, dC, dIN, dJN, dCN, nnzN);
}

extern "C" int hip_rmcl_prune_n(spgemm_handle* h, int m, int nnz, const int* dIC, const int* dJC, const float* dC,
                                int** dIN, int** dJN, float** dCN, int* nnzN) {
  if (nnz < 0) return fail(SPGEMM_ERR_ARG, "negative nnz");
  return rmcl_prune_impl(h, m, nnz, dIC, dJC, dC, dIN, dJN, dCN, nnzN);
}

// Expansion and prune of one R-MCL iteration as ONE operator: C = A*B is never materialised.  Every numeric kernel
// applies the row rule to the finished row while it still sits in LDS and writes only the kept, normalised entries
// (about a quarter of the product) to the front of the row's range of a scratch C; k_rmcl_move packs them.
// What the reference's loop does in three steps (gpu SpGEMM, inflate+threshold kernels of dutil.cuh, thrust::remove;
// gpu_csr_kernel.cu:218-270) and hip_gpuSpMM + hip_rmcl_prune do in two.
// Rows that sit at [starts[r], starts[r] + len[r]) of (J, V) -- where the epilogues of the fused R-MCL step leave them --
// packed into a CSR of their own: the scan of the lengths is the row pointer, k_rmcl_move copies the rows.
static int rmcl_pack_rows(spgemm_handle* h, int m, const int* starts, const int* len, const int* J, const float* V,
                          int** pI, int** pJ, float** pV, int* pn) {
  *pI = nullptr; *pJ = nullptr; *pV = nullptr; *pn = 0;
  hipStream_t s = h->stream;
  int* I = nullptr; int* JN = nullptr; float* CN = nullptr;
  auto bad = [&](int rc) { pool().release(I); pool().release(JN); pool().release(CN); return rc; };
  if (hipSuccess != pool().alloc((void**)&I, sizeof(int) * ((size_t)m + 1))) return fail(SPGEMM_ERR_NOMEM, "device allocation failed");
  unsigned long long total = 0;
  int rc;
  if (hipMemsetAsync(I, 0, sizeof(int) * ((size_t)m + 1), s) != hipSuccess ||
      hipMemcpyAsync(I, len, sizeof(int) * (size_t)m, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return bad(fail(SPGEMM_ERR_HIP, "pack: copy of the row lengths"));
  if ((rc = launch_scan(h, I, m, &h->dsmall->kept64))) return bad(rc);
  if (hipMemcpyAsync(&h->hsmall->kept64, &h->dsmall->kept64, sizeof(total), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return bad(fail(SPGEMM_ERR_HIP, "pack: scan of the row lengths"));
  total = h->hsmall->kept64;
  if (total > 0x7fffffffULL) return bad(fail(SPGEMM_ERR_OVERFLOW, "pack: %llu entries", total));
  const int nz = (int)total;
  if (hipSuccess != pool().alloc((void**)&JN, sizeof(int) * (size_t)std::max(nz, 1)) ||
      hipSuccess != pool().alloc((void**)&CN, sizeof(float) * (size_t)std::max(nz, 1)))
    return bad(fail(SPGEMM_ERR_NOMEM, "device allocation failed"));
  if (nz > 0 && m > 0) {
    hipLaunchKernelGGL(k_rmcl_move<16>, dim3(clampi(cdiv(m, 16), 1, h->numCU * 16)), dim3(256), 0, s, m, starts, I, J, V, JN, CN);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return bad(fail(SPGEMM_ERR_HIP, "pack: move"));
  }
  *pI = I; *pJ = JN; *pV = CN; *pn = nz;
  return SPGEMM_OK;
}

// One R-MCL iteration (expansion with the row rule fused into the numeric epilogues) on device arrays.
//   dIBlen != nullptr   B is NOT packed: row j is [dIB[j], dIB[j] + dIBlen[j]) of (dJB, dB), dIBse[j] holds the same
//                       extent as one {start, end} pair (what the classification gathers); nnzB = the arrays' extent
//   mode RMCL_EXTENTS   the result is left where the epilogues wrote it: *dIN = the scratch row starts (m + 1 entries),
//                       *dLenN = the kept entries per row, *dSEN = the {start, end} pairs, *dJN / *dCN = the scratch
//                       arrays, *nnzN = their extent (P).
//                       The next iteration reads it as its unpacked B: no scan, no copy, no allocation of packed arrays.
//   mode RMCL_BLOCK     (sharded loop) the rows stay in the scratch arrays too, but their lengths are scanned: *dIN = the
//                       scratch row starts, *dLenN = the row pointer of the PACKED block (m + 1 entries, [m] = *nnzN =
//                       kept entries), *dJN / *dCN = the scratch arrays.  The caller packs the rows wherever the block
//                       belongs (its slice of the next replicated Mt: k_rmcl_move), with no packed copy in between.
//                       (The paths that give up on the fused step return a packed result and *dLenN = nullptr.)
enum { RMCL_EXTENTS = 0, RMCL_PACK = 1, RMCL_BLOCK = 2 };
static int rmcl_expand_prune_core(spgemm_handle* h, const int* dIA, const int* dJA, const float* dA, int nnzA,
                                  const int* dIB, const int* dIBlen, const int2* dIBse, const int* dJB, const float* dB,
                                  int nnzB, int m, int k, int n, int mode, int** dIN, int** dLenN, int2** dSEN,
                                  int** dJN, float** dCN, int* nnzN, bool* errPending = nullptr) {
  const bool pack = mode == RMCL_PACK;
  *dIN = nullptr; *dLenN = nullptr; *dSEN = nullptr; *dJN = nullptr; *dCN = nullptr; *nnzN = 0;
  if (h->failNext > 0) { --h->failNext; return fail(SPGEMM_ERR_INTERNAL, "forced failure (spgemm_hip_debug_fail_next)"); }
  HIPCHK(hipSetDevice(h->device));
  auto two_steps = [&]() {                           // no rows, no products, or a product too large for the scratch C
    int *bI = nullptr, *bJ = nullptr, bn = nnzB;
    float* bV = nullptr;
    int rc = SPGEMM_OK;
    if (dIBlen && (rc = rmcl_pack_rows(h, k, dIB, dIBlen, dJB, dB, &bI, &bJ, &bV, &bn))) return rc;
    int *cI = nullptr, *cJ = nullptr, cn = 0;
    float* cA = nullptr;
    rc = spgemm_device(h, dIA, dJA, dA, nnzA, dIBlen ? bI : dIB, dIBlen ? bJ : dJB, dIBlen ? bV : dB, bn, m, k, n, nullptr,
                       &cI, &cJ, &cA, &cn);
    if (!rc) rc = rmcl_prune_impl(h, m, cn, cI, cJ, cA, dIN, dJN, dCN, nnzN);
    pool().release(cI); pool().release(cJ); pool().release(cA);
    pool().release(bI); pool().release(bJ); pool().release(bV);
    return rc;
  };
  if (m == 0) return two_steps();
  CHK(ws_ensure(h, m));
  int* dIC = nullptr; int* cnt = nullptr; int* dJC = nullptr; float* dC = nullptr; int* JN = nullptr; float* CN = nullptr;
  auto cleanup = [&](int rc) {
    // a failure may leave kernels queued that still write these blocks (and, in the loop form, the previous iteration's):
    // the pool hands a released block to other handles and streams, so nothing goes back before the stream has drained
    if (rc) (void)hipStreamSynchronize(h->stream);
    for (void* q : {(void*)dIC, (void*)cnt, (void*)dJC, (void*)dC, (void*)JN, (void*)CN}) pool().release(q);
    return rc;
  };
  auto hipfail = [&](const char* what) { return cleanup(fail(SPGEMM_ERR_HIP, "rmcl expand+prune: %s: %s", what, hipGetErrorString(hipGetLastError()))); };
  if (hipSuccess != pool().alloc((void**)&dIC, sizeof(int) * ((size_t)m + 1)) ||
      hipSuccess != pool().alloc((void**)&cnt, sizeof(int) * ((size_t)m + 1)))
    return hipfail("device allocation failed");
  hipStream_t s = h->stream;
  h->sym_m = -1;
  hipEventRecord(h->ev[0], s);
  h->cur_rowIds = h->rowIds;
  int rc = launch_classify(h, dIA, dJA, dIB, m, nnzA, dIC, dIBse);
  if (rc) return cleanup(rc);
  hipEventRecord(h->ev[1], s);
  if (hipMemcpyAsync(h->hmid, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipEventRecord(h->evMid, s) != hipSuccess)
    return hipfail("classification copy");
  if (hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)m + 1), s) != hipSuccess) return hipfail("memset");
  if (hipEventSynchronize(h->evMid) != hipSuccess) return hipfail("classification");
  if (errPending && *errPending) {                   // the previous iteration of the loop returned without waiting for its
    *errPending = false;                             // numeric kernels: their flags reached the host before this event
    if (h->hsmall->err) return cleanup(fail(SPGEMM_ERR_INTERNAL, "device invariant broken in the previous iteration (flags=%d)", h->hsmall->err));
  }
  const HostMirror mid = *h->hmid;
  const unsigned long long P = mid.totalP;
  // products beyond which the scratch C (8 bytes per product) is not made and the step runs as SpGEMM + prune.
  // SPGEMM_RMCL_MAXP lowers the bound (test hook: the give-up path behind an unpacked Mt on small inputs).
  unsigned long long maxP = 1ull << 30;
  if (const char* e = getenv("SPGEMM_RMCL_MAXP")) maxP = std::min<unsigned long long>(maxP, strtoull(e, nullptr, 10));
  if (P == 0 || P > maxP) {
    if (hipStreamSynchronize(s) != hipSuccess) return hipfail("classification");
    cleanup(0);
    dIC = cnt = dJC = JN = nullptr; dC = CN = nullptr;
    return two_steps();
  }
  // The symbolic pass is SKIPPED for every row of at most 4096 products (all rows of an R-MCL iteration on a sparse graph
  // without hubs).  Its only product is the exact size of every row of C, and C is not kept: the scratch rows are laid
  // out by the rows' product counts (known from the classification), every such row is hashed in a table sized by its
  // products and the epilogue counts the distinct columns itself.  Rows of bin 8 pass through LDS in pieces and need
  // their exact counts: only they get a symbolic kernel (k_sym_big overwrites their entries of dIC).
  const bool nosym = !getenv("SPGEMM_RMCL_SYMBOLIC");
  if (nosym) {
    if (hipMemcpyAsync(dIC, h->rowFlops, sizeof(int) * (size_t)m, hipMemcpyDeviceToDevice, s) != hipSuccess) return hipfail("copy");
    const int nbig = mid.binPtr[NBINS] - mid.binPtr[NBINS - 1];
    if (nbig > 0) {
      KTimer t(h, SPGEMM_K_SYM_BIG, s);
      hipLaunchKernelGGL(k_sym_big, dim3(clampi(nbig, 1, h->numCU)), dim3(BIG_THREADS), sizeof(BigSymShared), s,
                         h->dsmall->binPtr, 8, h->cur_rowIds, dIA, h->sbl, dJB, n, dIC, h->bigBitmaps, h->bm_cap,
                         h->dsmall->qctr + 0 * 32);
      if (hipGetLastError() != hipSuccess) return hipfail("symbolic launch");
    }
  } else if ((rc = launch_symbolic(h, dIA, dJB, m, n, h->cur_rowIds, dIC))) return cleanup(rc);
  hipEventRecord(h->ev[2], s);
  if ((rc = launch_scan(h, dIC, m, &h->dsmall->nnzC64))) return cleanup(rc);
  hipEventRecord(h->ev[3], s);
  if (hipSuccess != pool().alloc((void**)&dJC, sizeof(int) * (size_t)P) ||
      hipSuccess != pool().alloc((void**)&dC, sizeof(float) * (size_t)P))
    return hipfail("device allocation of the scratch product failed");
  hipEventRecord(h->ev[4], s);
  h->mirror = mid;
  if ((rc = launch_numeric(h, dIA, dA, dJB, dB, n, h->cur_rowIds, mid.binPtr, dIC, dJC, dC, cnt, nosym ? 2 : 1))) return cleanup(rc);
  hipEventRecord(h->ev[5], s);
  if ((pack || mode == RMCL_BLOCK) && (rc = launch_scan(h, cnt, m, &h->dsmall->kept64))) return cleanup(rc);
  if (hipMemcpyAsync(h->hsmall, h->dsmall, sizeof(HostMirror), hipMemcpyDeviceToHost, s) != hipSuccess)
    return hipfail("numeric phase");
  if (mode == RMCL_EXTENTS && errPending && h->ktiming == 0) {
    // Loop form: nothing of this iteration is needed on the host before the next one starts -- its classification is
    // queued behind the numeric kernels, and the error flags are looked at after that classification's event (above).
    // The host runs ahead and the GPU does not idle between iterations.  (With per-kernel timing on, the events have to
    // be read now: the call waits as before.)
    int2* se = nullptr;
    if (hipSuccess != pool().alloc((void**)&se, sizeof(int2) * (size_t)m)) return hipfail("device allocation failed");
    hipLaunchKernelGGL(k_zip_extents, dim3(cdiv(m, 256)), dim3(256), 0, s, m, dIC, cnt, se);
    if (hipGetLastError() != hipSuccess) { pool().release(se); return hipfail("extent launch"); }
    spgemm_stats& st0 = h->stats;
    st0.total_flops = (long long)mid.totalP;
    st0.nnzC = -1;
    for (int b = 0; b < NBINS; ++b) st0.bin_rows[b] = mid.binPtr[b + 1] - mid.binPtr[b];
    *errPending = true;
    *dIN = dIC; *dLenN = cnt; *dSEN = se; *dJN = dJC; *dCN = dC; *nnzN = (int)P;
    return SPGEMM_OK;
  }
  if (hipStreamSynchronize(s) != hipSuccess) return hipfail("numeric phase");
  h->mirror = *h->hsmall;
  const HostMirror& hm = h->mirror;
  if (hm.err) return cleanup(fail(SPGEMM_ERR_INTERNAL, "device invariant broken (flags=%d)", hm.err));
  spgemm_stats& st = h->stats;
  st.total_flops = (long long)hm.totalP;
  st.nnzC = nosym ? -1 : (int)std::min<unsigned long long>(hm.nnzC64, 0x7fffffffULL);   // not computed without the symbolic pass
  for (int b = 0; b < NBINS; ++b) st.bin_rows[b] = hm.binPtr[b + 1] - hm.binPtr[b];
  hipEventElapsedTime(&st.ms_classify, h->ev[0], h->ev[1]);
  hipEventElapsedTime(&st.ms_symbolic, h->ev[1], h->ev[2]);
  hipEventElapsedTime(&st.ms_scan_alloc, h->ev[2], h->ev[4]);
  hipEventElapsedTime(&st.ms_numeric, h->ev[4], h->ev[5]);
  hipEventElapsedTime(&st.ms_total, h->ev[0], h->ev[5]);
  collect_kernel_times(h, true);
  grow_bitmaps(h, n);
  if (mode == RMCL_BLOCK) {                          // scratch rows + the row pointer of the packed block
    *dIN = dIC; *dLenN = cnt; *dJN = dJC; *dCN = dC; *nnzN = (int)hm.kept64;
    return SPGEMM_OK;
  }
  if (!pack) {                                       // the rows stay where the epilogues wrote them
    int2* se = nullptr;
    if (hipSuccess != pool().alloc((void**)&se, sizeof(int2) * (size_t)m)) return hipfail("device allocation failed");
    hipLaunchKernelGGL(k_zip_extents, dim3(cdiv(m, 256)), dim3(256), 0, s, m, dIC, cnt, se);
    if (hipGetLastError() != hipSuccess) { pool().release(se); return hipfail("extent launch"); }
    *dIN = dIC; *dLenN = cnt; *dSEN = se; *dJN = dJC; *dCN = dC; *nnzN = (int)P;
    return SPGEMM_OK;                                // stream order: the next classification runs behind the kernel
  }
  const int nz = (int)hm.kept64;                     // <= nnz(C) <= P <= 2^30
  if (hipSuccess != pool().alloc((void**)&JN, sizeof(int) * (size_t)std::max(nz, 1)) ||
      hipSuccess != pool().alloc((void**)&CN, sizeof(float) * (size_t)std::max(nz, 1)))
    return hipfail("device allocation failed");
  if (nz > 0) {
    const bool wide = nz >= 96ll * m;
    if (wide) hipLaunchKernelGGL(k_rmcl_move<64>, dim3(clampi(cdiv(m, 4), 1, h->numCU * 32)), dim3(256), 0, s, m, dIC, cnt, dJC, dC, JN, CN);
    else hipLaunchKernelGGL(k_rmcl_move<16>, dim3(clampi(cdiv(m, 16), 1, h->numCU * 16)), dim3(256), 0, s, m, dIC, cnt, dJC, dC, JN, CN);
    if (hipGetLastError() != hipSuccess) return hipfail("move launch");
  }
  if (hipStreamSynchronize(s) != hipSuccess) return hipfail("move");
  pool().release(dIC); pool().release(dJC); pool().release(dC);
  *dIN = cnt; *dJN = JN; *dCN = CN; *nnzN = nz;
  return SPGEMM_OK;
}

extern "C" int hip_rmcl_expand_prune(spgemm_handle* h, const int* dIA, const int* dJA, const float* dA, int nnzA,
                                     const int* dIB, const int* dJB, const float* dB, int nnzB, int m, int k, int n,
                                     int** dIN, int** dJN, float** dCN, int* nnzN) {
  if (!dIN || !dJN || !dCN || !nnzN) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  *dIN = nullptr; *dJN = nullptr; *dCN = nullptr; *nnzN = 0;
  if (m < 0 || k < 0 || n < 0) return fail(SPGEMM_ERR_ARG, "negative dimension m=%d k=%d n=%d", m, k, n);
  CHK(check_common(dIA, dJA, dA, nnzA, "A"));
  CHK(check_common(dIB, dJB, dB, nnzB, "B"));
  if (!h) CHK(default_handle(&h));
  int* len = nullptr;
  int2* se = nullptr;
  return rmcl_expand_prune_core(h, dIA, dJA, dA, nnzA, dIB, nullptr, nullptr, dJB, dB, nnzB, m, k, n, RMCL_PACK, dIN, &len, &se,
                                dJN, dCN, nnzN);
}

// The R-MCL loop on device arrays (gpuRmclIter, gpus/gpu_csr_kernel.cu:15-40, without its two copies): maxIter iterations
// Mt <- prune(Mgt * Mt) from (tI, tJ, tA), which are left alone; the result is a packed device CSR of the library's
// pool.  Between iterations Mt is NOT packed: every epilogue leaves its kept entries at the front of the row's scratch
// range and the next classification reads {start, kept} per row (k_row_flops' IBlen), so that only the last iteration
// pays for the scan of the counts, the packed arrays and the copy into them.
extern "C" int hip_gpuRmclIter_device(spgemm_handle* h, int maxIter, int rows, int cols, const int* dgI, const int* dgJ,
                                      const float* dgA, int gnnz, const int* dtI, const int* dtJ, const float* dtA,
                                      int tnnz, int** oI, int** oJ, float** oA, int* onnz) {
  if (!oI || !oJ || !oA || !onnz) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  *oI = nullptr; *oJ = nullptr; *oA = nullptr; *onnz = 0;
  if (rows < 0 || cols != rows || maxIter < 0) return fail(SPGEMM_ERR_ARG, "R-MCL needs a square matrix and maxIter >= 0");
  CHK(check_common(dgI, dgJ, dgA, gnnz, "Mgt"));
  CHK(check_common(dtI, dtJ, dtA, tnnz, "Mt"));
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  const bool keep_packed = getenv("SPGEMM_RMCL_PACK") != nullptr;       // A/B switch: pack after every iteration
  const int *bI = dtI, *bLen = nullptr, *bJ = dtJ;
  const int2* bSE = nullptr;
  const float* bV = dtA;
  int bn = tnnz;
  int *cI = nullptr, *cLen = nullptr, *cJ = nullptr;                    // the current Mt when the loop owns it
  int2* cSE = nullptr;
  float* cV = nullptr;
  // An iteration that leaves Mt unpacked returns WITHOUT waiting for its numeric kernels (they read the previous Mt):
  // that Mt goes to `old` and is released one iteration later, after a call that has waited for an event recorded behind
  // those kernels -- the pool hands blocks to other handles and streams, so stream order alone does not protect them.
  std::vector<void*> old;
  auto flush_old = [&]() { for (void* q : old) pool().release(q); old.clear(); };
  auto drop = [&]() {
    for (void* q : {(void*)cI, (void*)cLen, (void*)cSE, (void*)cJ, (void*)cV}) if (q) old.push_back(q);
    cI = cLen = cJ = nullptr; cSE = nullptr; cV = nullptr;
  };
  bool errPending = false;
  auto bail = [&](int rc) { (void)hipStreamSynchronize(h->stream); drop(); flush_old(); return rc; };
  for (int it = 0; it < maxIter; ++it) {
    int *nI = nullptr, *nLen = nullptr, *nJ = nullptr, nn = 0;
    int2* nSE = nullptr;
    float* nV = nullptr;
    const bool pack = keep_packed || it == maxIter - 1;
    const int rc = rmcl_expand_prune_core(h, dgI, dgJ, dgA, gnnz, bI, bLen, bSE, bJ, bV, bn, rows, cols, cols,
                                          pack ? RMCL_PACK : RMCL_EXTENTS, &nI, &nLen, &nSE, &nJ, &nV, &nn, &errPending);
    if (rc) return bail(rc);
    flush_old();                                     // the call above waited behind the kernels that read these
    drop();                                          // the Mt it read itself: released after the next call
    cI = nI; cLen = nLen; cSE = nSE; cJ = nJ; cV = nV;
    bI = cI; bLen = cLen; bSE = cSE; bJ = cJ; bV = cV; bn = nn;
  }
  flush_old();                                       // the last iteration packs and waits for everything
  if (maxIter == 0) {                                // a copy of Mt
    const size_t bi = sizeof(int) * ((size_t)rows + 1), bj = sizeof(int) * (size_t)std::max(tnnz, 1);
    if (hipSuccess != pool().alloc((void**)&cI, bi) || hipSuccess != pool().alloc((void**)&cJ, bj) ||
        hipSuccess != pool().alloc((void**)&cV, bj)) { return bail(fail(SPGEMM_ERR_NOMEM, "device allocation failed")); }
    if (hipMemcpyAsync(cI, dtI, bi, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
        (tnnz > 0 && (hipMemcpyAsync(cJ, dtJ, sizeof(int) * (size_t)tnnz, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                      hipMemcpyAsync(cV, dtA, sizeof(float) * (size_t)tnnz, hipMemcpyDeviceToDevice, h->stream) != hipSuccess)) ||
        hipStreamSynchronize(h->stream) != hipSuccess) { return bail(fail(SPGEMM_ERR_HIP, "copy of Mt")); }
    bn = tnnz;
  } else if (bLen) {                                 // the last iteration gave up on the fused step?  it returns packed: not reached
    return bail(fail(SPGEMM_ERR_INTERNAL, "the last iteration returned an unpacked matrix"));
  }
  *oI = cI; *oJ = cJ; *oA = cV; *onnz = bn;
  return SPGEMM_OK;
}

struct spgemm_group;
static std::mutex g_rmcl_group_mu;                     // the group hip_gpuRmclIter keeps between calls (same device layout)
static spgemm_group* g_rmcl_group = nullptr;
static std::vector<int> g_rmcl_group_devs;
static int g_rmcl_devices_used = 1;
// devices the latest hip_gpuRmclIter of this process computed on (1 unless SPGEMM_RMCL_DEVICES asked for more)
extern "C" int spgemm_hip_rmcl_devices_used(void) { return g_rmcl_devices_used; }
extern "C" int spgemm_hip_group_create(spgemm_group** out, int nshards, const int* devices, int transport);
extern "C" int spgemm_hip_group_destroy(spgemm_group* g);
extern "C" int hip_gpuRmclIter_sharded(spgemm_group* g, int maxIter, int rows, int cols, const int* gIA, const int* gJA,
                                       const float* gA, int gnnz, const int* tIA, const int* tJA, const float* tA, int tnnz,
                                       int** oIA, int** oJA, float** oA, int* onnz);

extern "C" int hip_gpuRmclIter(int maxIter, int rows, int cols, const int* gIA, const int* gJA, const float* gA, int gnnz,
                               const int* tIA, const int* tJA, const float* tA, int tnnz, int** oIA, int** oJA,
                               float** oA, int* onnz) {
  if (!oIA || !oJA || !oA || !onnz) return fail(SPGEMM_ERR_ARG, "output pointer is null");
  if (rows < 0 || cols != rows || maxIter < 0) return fail(SPGEMM_ERR_ARG, "R-MCL needs a square matrix and maxIter >= 0");
  CHK(check_common(gIA, gJA, gA, gnnz, "Mgt"));
  CHK(check_common(tIA, tJA, tA, tnnz, "Mt"));
  CHK(validate_host_csr(gIA, gJA, rows, cols, gnnz, "Mgt"));
  CHK(validate_host_csr(tIA, tJA, rows, cols, tnnz, "Mt"));
  {
    // Several devices, when the caller asks for them: Mgt's rows are cut into flops-balanced blocks, one per GPU, Mt is
    // replicated and the pruned blocks are gathered every iteration (sharded.hpp) -- behind the same call the reference's
    // driver makes (nlibs/qrmcl.cc:149-152).  OPT-IN (round 3 sharded over every visible device by default: a 100-row graph
    // was cut over 8 GPUs, and a process-per-GPU job would have opened contexts and communicators on all of them from every
    // rank): SPGEMM_RMCL_DEVICES=N|all uses N devices (0..N-1); SPGEMM_RMCL_SHARDS=K asks for K LOGICAL shards spread
    // round-robin over those devices (the one-GPU rehearsal of the sharded loop).  Unset: one device, the default handle's.
    // The group is kept for the next call with the same layout (the C++ mirror's --stats mode calls this once per iteration:
    // handles, RCCL communicators and pinned buffers are made once).
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    int use = 1;
    if (const char* e = getenv("SPGEMM_RMCL_DEVICES")) use = !strcasecmp(e, "all") ? ndev : std::max(1, std::min(ndev, atoi(e)));
    int shards = use;
    if (const char* e = getenv("SPGEMM_RMCL_SHARDS")) shards = std::max(1, std::min(64, atoi(e)));
    g_rmcl_devices_used = 1;
    if (shards > 1 && rows >= shards) {
      std::vector<int> devs((size_t)shards);
      for (int i = 0; i < shards; ++i) devs[(size_t)i] = i % std::max(use, 1);
      std::lock_guard<std::mutex> lk(g_rmcl_group_mu);
      int rc = SPGEMM_OK;
      if (!g_rmcl_group || g_rmcl_group_devs != devs) {
        if (g_rmcl_group) { spgemm_hip_group_destroy(g_rmcl_group); g_rmcl_group = nullptr; }
        rc = spgemm_hip_group_create(&g_rmcl_group, shards, devs.data(), SPGEMM_XCHG_AUTO);
        if (rc == SPGEMM_OK) g_rmcl_group_devs = devs;
        else g_rmcl_group = nullptr;
      }
      if (rc == SPGEMM_OK)
        rc = hip_gpuRmclIter_sharded(g_rmcl_group, maxIter, rows, cols, gIA, gJA, gA, gnnz, tIA, tJA, tA, tnnz, oIA, oJA, oA, onnz);
      if (rc == SPGEMM_OK) { g_rmcl_devices_used = std::min(use, shards); return rc; }
      // the caller asked for an R-MCL result, not for a particular number of GPUs: say what went wrong and carry on with one
      fprintf(stderr, "hip_gpuRmclIter: the %d-shard loop failed (%s); running on one device\n", shards, spgemm_hip_last_error());
    }
  }
  spgemm_handle* h = nullptr;
  CHK(default_handle(&h));
  int *dgI = nullptr, *dgJ = nullptr, *dtI = nullptr, *dtJ = nullptr;
  float *dgA = nullptr, *dtA = nullptr;
  auto cleanup = [&](int rc) { for (void* p : {(void*)dgI, (void*)dgJ, (void*)dgA, (void*)dtI, (void*)dtJ, (void*)dtA}) pool().release(p); return rc; };
  int rc;
#define UP(dst, src, bytes) \
  if ((rc = spgemm_hip_malloc((void**)&dst, (bytes))) || (rc = spgemm_hip_memcpy_h2d(dst, src, (bytes)))) return cleanup(rc);
  UP(dgI, gIA, sizeof(int) * ((size_t)rows + 1)); UP(dgJ, gJA, sizeof(int) * (size_t)gnnz); UP(dgA, gA, sizeof(float) * (size_t)gnnz);
  UP(dtI, tIA, sizeof(int) * ((size_t)rows + 1)); UP(dtJ, tJA, sizeof(int) * (size_t)tnnz); UP(dtA, tA, sizeof(float) * (size_t)tnnz);
#undef UP
  int curnnz = tnnz;
  {
    int *nI = nullptr, *nJ = nullptr, nn = 0;
    float* nA = nullptr;
    if ((rc = hip_gpuRmclIter_device(h, maxIter, rows, cols, dgI, dgJ, dgA, gnnz, dtI, dtJ, dtA, tnnz, &nI, &nJ, &nA, &nn))) return cleanup(rc);
    pool().release(dtI); pool().release(dtJ); pool().release(dtA);
    dtI = nI; dtJ = nJ; dtA = nA; curnnz = nn;
  }
  int* hI = (int*)malloc(sizeof(int) * ((size_t)rows + 1));
  int* hJ = (int*)malloc(sizeof(int) * (size_t)std::max(curnnz, 1));
  float* hA = (float*)malloc(sizeof(float) * (size_t)std::max(curnnz, 1));
  if (!hI || !hJ || !hA) { free(hI); free(hJ); free(hA); return cleanup(fail(SPGEMM_ERR_NOMEM, "host malloc failed")); }
  if ((rc = spgemm_hip_memcpy_d2h(hI, dtI, sizeof(int) * ((size_t)rows + 1))) ||
      (rc = spgemm_hip_memcpy_d2h(hJ, dtJ, sizeof(int) * (size_t)curnnz)) ||
      (rc = spgemm_hip_memcpy_d2h(hA, dtA, sizeof(float) * (size_t)curnnz))) { free(hI); free(hJ); free(hA); return cleanup(rc); }
  *oIA = hI; *oJA = hJ; *oA = hA; *onnz = curnnz;
  return cleanup(SPGEMM_OK);
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
extern "C" int hip_csr_sort_rows(spgemm_handle* h, int m, const int* dIC, int* dJC, float* dC) {
  if (m < 0 || !dIC) return fail(SPGEMM_ERR_ARG, "bad argument");
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  if (m == 0) return SPGEMM_OK;
  clear_stale_hip_error();
  hipStream_t s = h->stream;
  int* dcnt = nullptr;
  int* JS = nullptr;
  float* CS = nullptr;
  auto cleanup = [&](int rc) { pool().release(dcnt); pool().release(JS); pool().release(CS); return rc; };
  auto hipfail = [&](const char* what) { return cleanup(fail(SPGEMM_ERR_HIP, "sort rows: %s: %s", what, hipGetErrorString(hipGetLastError()))); };
  if (pool().alloc((void**)&dcnt, sizeof(int)) != hipSuccess) return hipfail("allocation");
  if (hipMemsetAsync(dcnt, 0, sizeof(int), s) != hipSuccess) return hipfail("memset");
  hipLaunchKernelGGL(k_sort_rows, dim3(clampi(m, 1, h->numCU * 8)), dim3(256), 0, s, m, dIC, dJC, dC, dcnt);
  int info[2] = {0, 0};                                // unsorted rows longer than SORT_MAX, nnz
  if (hipMemcpyAsync(&info[0], dcnt, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipMemcpyAsync(&info[1], dIC + m, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return hipfail("short rows");
  if (info[0] > 0 && info[1] > 0) {
    // the long rows sort between their segment of (JC, C) and the same segment of a scratch copy
    if (pool().alloc((void**)&JS, sizeof(int) * (size_t)info[1]) != hipSuccess ||
        pool().alloc((void**)&CS, sizeof(float) * (size_t)info[1]) != hipSuccess)
      return hipfail("scratch allocation");
    // column bits: the columns of a valid CSR are < 2^31; take the width from the largest possible key
    const int keyBits = 31;
    hipLaunchKernelGGL(k_sort_long_rows, dim3(clampi(info[0], 1, h->numCU * 4)), dim3(256), 0, s, m, dIC, dJC, dC, JS, CS, keyBits);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return hipfail("long rows");
  }
  return cleanup(SPGEMM_OK);
}

extern "C" int spgemm_hip_selftest(spgemm_handle* h) {
  if (!h) CHK(default_handle(&h));
  HIPCHK(hipSetDevice(h->device));
  const int nb = 64;
  std::vector<int> host(nb * WAVE);
  unsigned x = 12345u;
  for (auto& v : host) { x = x * 1664525u + 1013904223u; v = (int)((x >> 8) & 0xfffff); }
  int* din = nullptr;
  int* dbad = nullptr;
  HIPCHK(hipMalloc((void**)&din, sizeof(int) * host.size()));
  HIPCHK(hipMalloc((void**)&dbad, sizeof(int)));
  HIPCHK(hipMemcpy(din, host.data(), sizeof(int) * host.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dbad, 0, sizeof(int)));
  clear_stale_hip_error();
  hipLaunchKernelGGL(k_selftest, dim3(nb), dim3(WAVE), 0, h->stream, din, dbad);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  int bad = -1;
  HIPCHK(hipMemcpy(&bad, dbad, sizeof(int), hipMemcpyDeviceToHost));
  hipFree(din);
  hipFree(dbad);
  if (bad != 0) return fail(SPGEMM_ERR_INTERNAL, "wave primitive self-test: %d lanes disagree", bad);
  return SPGEMM_OK;
}


// ------------------------------------------------------------------------------------------------
// COO -> CSR on the device (the step in front of the path)
// ------------------------------------------------------------------------------------------------
#include "coo_device.hpp"

// ------------------------------------------------------------------------------------------------
// multi-GPU: groups of shards, sharded SpGEMM, sharded R-MCL
// ------------------------------------------------------------------------------------------------
#include "sharded.hpp"

#ifdef SMF_STAMPS
// diagnostic build only: the phase cycles k_num_bighash has accumulated since the last call (and reset)
extern "C" int spgemm_hip_debug_stamps(unsigned long long* out16) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(z)) != hipSuccess) return fail(SPGEMM_ERR_HIP, "stamps");
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return fail(SPGEMM_ERR_HIP, "stamps reset");
  return SPGEMM_OK;
}
extern "C" int spgemm_hip_debug_chain_stamps(unsigned long long* out16) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_cstamps), sizeof(z)) != hipSuccess) return fail(SPGEMM_ERR_HIP, "stamps");
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_cstamps), z, sizeof(z)) != hipSuccess) return fail(SPGEMM_ERR_HIP, "stamps reset");
  return SPGEMM_OK;
}
#endif
