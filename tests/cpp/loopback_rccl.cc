// TEST INFRASTRUCTURE, not part of the product: an in-process stand-in for the nine RCCL entry points libspgemm_hip.so
// loads at run time (csrc/sharded.hpp: RcclApi), selected with SPGEMM_RCCL_LIB.  The test box has ONE GPU and real RCCL
// refuses two ranks on one device, so the code path that runs on an 8-GPU node with peers != self -- communicator per
// rank, ncclAllGather of the segment sizes, grouped ncclSend/ncclRecv of the row segments -- cannot run there.  Here the
// RANKS ARE THREADS of one process sharing the GPU: point-to-point messages are matched per (source, destination) pair
// in posting order, exactly as NCCL matches them, and delivered by a device-to-device copy on the receiver's stream that
// waits for the sender's stream; a mismatch of message sizes between a send and its receive is an error, an unmatched
// receive blocks (the test times out).  What this checks is the library's call pattern: who sends what to whom, the
// counts and offsets, and that every rank posts the matching operations.  It says nothing about RCCL itself.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace {

struct Msg {
  const void* buf = nullptr;
  size_t bytes = 0;
  hipEvent_t ready = nullptr, done = nullptr;
  bool consumed = false;
};

struct World {
  int nranks = 0, joined = 0, left = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> box;   // (source, destination) -> posted sends
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;

struct Op { bool send; void* buf; size_t bytes; int peer; ncclComm_t comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t elem_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

}  // namespace

struct ncclComm {
  std::shared_ptr<World> w;
  int rank = 0;
};

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::random_device rd;
  for (size_t i = 0; i < sizeof(id->internal); ++i) id->internal[i] = (char)(rd() & 0xff);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto& slot = g_worlds[std::string(id.internal, sizeof(id.internal))];
    if (!slot) { slot = std::make_shared<World>(); slot->nranks = nranks; }
    w = slot;
  }
  if (w->nranks != nranks) return ncclInvalidArgument;
  std::unique_lock<std::mutex> lk(w->mu);
  ++w->joined;
  w->cv.notify_all();
  w->cv.wait(lk, [&] { return w->joined >= w->nranks; });              // as ncclCommInitRank: returns when all are in
  auto* c = new ncclComm;
  c->w = w;
  c->rank = rank;
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  delete comm;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (loopback stand-in)";
    case ncclInvalidUsage: return "invalid usage: a receive met a send of another size (loopback stand-in)";
    case ncclUnhandledCudaError: return "HIP call failed (loopback stand-in)";
    default: return "error (loopback stand-in)";
  }
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}

static ncclResult_t run_ops(std::vector<Op>& ops) {
  std::vector<std::pair<std::shared_ptr<Msg>, hipStream_t>> mine;
  for (Op& o : ops) {                                  // 1. post every send
    if (!o.send) continue;
    auto m = std::make_shared<Msg>();
    m->buf = o.buf;
    m->bytes = o.bytes;
    if (hipEventCreateWithFlags(&m->ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->done, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(m->ready, o.stream) != hipSuccess)
      return ncclUnhandledCudaError;
    World& w = *o.comm->w;
    { std::lock_guard<std::mutex> lk(w.mu); w.box[{o.comm->rank, o.peer}].push_back(m); }
    w.cv.notify_all();
    mine.push_back({m, o.stream});
  }
  ncclResult_t res = ncclSuccess;
  for (Op& o : ops) {                                  // 2. serve every receive, in posting order per source
    if (o.send) continue;
    World& w = *o.comm->w;
    std::shared_ptr<Msg> m;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      auto& q = w.box[{o.peer, o.comm->rank}];
      w.cv.wait(lk, [&] { return !q.empty(); });
      m = q.front();
      q.pop_front();
    }
    if (m->bytes != o.bytes) res = ncclInvalidUsage;
    else if (hipStreamWaitEvent(o.stream, m->ready, 0) != hipSuccess ||
             (o.bytes && hipMemcpyAsync(o.buf, m->buf, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) ||
             hipEventRecord(m->done, o.stream) != hipSuccess)
      res = ncclUnhandledCudaError;
    { std::lock_guard<std::mutex> lk(w.mu); m->consumed = true; }
    w.cv.notify_all();
  }
  for (auto& ms : mine) {                              // 3. my buffers are free again once the receivers have copied
    World& w = *ops.front().comm->w;
    { std::unique_lock<std::mutex> lk(w.mu); w.cv.wait(lk, [&] { return ms.first->consumed; }); }
    if (res == ncclSuccess && hipStreamWaitEvent(ms.second, ms.first->done, 0) != hipSuccess) res = ncclUnhandledCudaError;
  }
  return res;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return ops.empty() ? ncclSuccess : run_ops(ops);
}

static ncclResult_t post(bool send, void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  const size_t es = elem_size(t);
  if (!comm || !es || peer < 0 || peer >= comm->w->nranks) return ncclInvalidArgument;
  t_ops.push_back({send, buf, count * es, peer, comm, s});
  if (t_depth == 0) {                                  // outside a group: the operation runs at once
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
  }
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return post(true, const_cast<void*>(buf), count, t, peer, comm, s);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return post(false, buf, count, t, peer, comm, s);
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t t, ncclComm_t comm,
                           hipStream_t s) {
  const size_t es = elem_size(t);
  if (!comm || !es) return ncclInvalidArgument;
  ncclGroupStart();
  for (int p = 0; p < comm->w->nranks; ++p) {
    post(true, const_cast<void*>(sendbuff), sendcount, t, p, comm, s);
    post(false, static_cast<char*>(recvbuff) + (size_t)p * sendcount * es, sendcount, t, p, comm, s);
  }
  return ncclGroupEnd();
}

}  // extern "C"
