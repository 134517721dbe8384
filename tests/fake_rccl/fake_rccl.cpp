// TEST INFRASTRUCTURE, never part of the product path: a stand-in for librccl's C API for processes that SHARE ONE GPU.
//
// RCCL refuses two ranks on one device, and a builder / driver box has one GPU: licos_allreduce_weighted_direct's grouped
// send / recv loop, its chunk offsets, scratch sizing and stream ordering (licos_amd/csrc/collective.hip, the replacement of
// /root/reference/licos/federation_utils.py:27-85) could otherwise only ever run with nranks == 1, i.e. never through its
// loop.  collective.hip loads this library INSTEAD of librccl only when LICOS_RCCL_TEST_LIB names it
// (tests/test_gpu_stale.py); nothing in licos_amd/ refers to it.
//
// Transport: a POSIX shared-memory segment named after the unique id.  Per ordered rank pair one mailbox
// {sent, received, payload}; ncclSend = wait until the mailbox is free, hipMemcpy device -> mailbox, publish;
// ncclRecv = wait for the publication, hipMemcpy mailbox -> device, acknowledge.  Inside ncclGroupStart / ncclGroupEnd
// the calls are queued and run at the outermost GroupEnd, all sends before all receives (what makes the grouped
// exchange deadlock-free).  Stream order: the stream is synchronised before the first copy out, the copies are
// synchronous - work queued on the stream afterwards sees the received data, as with the real library.
// ncclAllReduce(sum, f32): every rank publishes its buffer in its slot, a barrier, every rank adds the slots in rank order
// (identical bits on every rank), a barrier.  Every wait is bounded (LICOS_FAKE_RCCL_TIMEOUT_S, default 60): a peer that
// died shows up as an error code, not as a hang.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

struct UniqueId { char internal[128]; };
enum { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4 };

struct Mailbox {
  std::atomic<uint64_t> sent, received;
  char pad[112];
};
struct Header {
  std::atomic<uint64_t> arrived[4];  // barrier: arrivals per phase
  std::atomic<uint64_t> attached;
  char pad[88];
};

struct Comm {
  int nranks, rank;
  size_t box_bytes, total;
  char name[64];
  unsigned char *base;
  uint64_t barrier_no;
  Header *hdr() const { return reinterpret_cast<Header *>(base); }
  Mailbox *box(int src, int dst) const { return reinterpret_cast<Mailbox *>(base + sizeof(Header)) + (src * nranks + dst); }
  unsigned char *payload(int src, int dst) const {
    return base + sizeof(Header) + sizeof(Mailbox) * (size_t)nranks * nranks + (size_t)(src * nranks + dst) * box_bytes;
  }
};

struct Op {
  bool send;
  void *buf;
  size_t bytes;
  int peer;
  Comm *comm;
  hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double timeout_s() {
  const char *e = getenv("LICOS_FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 60.0;
}

template <class Pred>
bool wait_until(Pred ok) {
  const auto t0 = std::chrono::steady_clock::now();
  const double limit = timeout_s();
  for (long spin = 0; !ok(); ++spin) {
    if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spin & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return false;
  }
  return true;
}

int do_send(const Op &o) {
  Comm *c = o.comm;
  if (o.bytes > c->box_bytes) return kInvalidArgument;
  Mailbox *m = c->box(c->rank, o.peer);
  if (!wait_until([&] { return m->sent.load(std::memory_order_acquire) == m->received.load(std::memory_order_acquire); })) return kSystemError;
  if (hipMemcpy(c->payload(c->rank, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledCudaError;
  m->sent.fetch_add(1, std::memory_order_release);
  return kSuccess;
}

int do_recv(const Op &o) {
  Comm *c = o.comm;
  if (o.bytes > c->box_bytes) return kInvalidArgument;
  Mailbox *m = c->box(o.peer, c->rank);
  if (!wait_until([&] { return m->sent.load(std::memory_order_acquire) > m->received.load(std::memory_order_acquire); })) return kSystemError;
  if (hipMemcpy(o.buf, c->payload(o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return kUnhandledCudaError;
  m->received.fetch_add(1, std::memory_order_release);
  return kSuccess;
}

int run_ops(std::vector<Op> &ops) {
  // everything queued on the streams so far must have produced the send buffers / finished with the receive buffers
  for (const Op &o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return kUnhandledCudaError;
  for (const Op &o : ops)
    if (o.send) {
      const int rc = do_send(o);
      if (rc != kSuccess) return rc;
    }
  for (const Op &o : ops)
    if (!o.send) {
      const int rc = do_recv(o);
      if (rc != kSuccess) return rc;
    }
  return kSuccess;
}

int barrier(Comm *c) {
  const uint64_t no = c->barrier_no++;
  std::atomic<uint64_t> &a = c->hdr()->arrived[no & 3];
  // phase counters are monotonic: arrival k of generation g is count g * nranks + k (four phases in rotation, a phase is
  // reused only after three later barriers have completed)
  a.fetch_add(1, std::memory_order_acq_rel);
  const uint64_t want = (no / 4 + 1) * (uint64_t)c->nranks;
  return wait_until([&] { return a.load(std::memory_order_acquire) >= want; }) ? kSuccess : kSystemError;
}

size_t dtype_bytes(int dtype) { return dtype == 7 ? 4 : 0; }  // ncclFloat32 only: all the product uses

}  // namespace

extern "C" {

int ncclGetUniqueId(UniqueId *id) {
  if (!id) return kInvalidArgument;
  std::memset(id, 0, sizeof(*id));
  const uint64_t t = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof(id->internal), "/licos_fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)t);
  return kSuccess;
}

int ncclCommInitRank(void **comm, int nranks, UniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > 8 || rank < 0 || rank >= nranks) return kInvalidArgument;
  const char *e = getenv("LICOS_FAKE_RCCL_BOX_BYTES");
  Comm *c = new Comm();
  c->nranks = nranks;
  c->rank = rank;
  c->box_bytes = e ? (size_t)atoll(e) : ((size_t)16 << 20);
  c->barrier_no = 0;
  std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
  c->total = sizeof(Header) + sizeof(Mailbox) * (size_t)nranks * nranks + (size_t)nranks * nranks * c->box_bytes;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->total) != 0) {  // (a fresh segment is zero-filled: counters start at 0)
    if (fd >= 0) close(fd);
    delete c;
    return kSystemError;
  }
  void *p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return kSystemError;
  }
  c->base = static_cast<unsigned char *>(p);
  c->hdr()->attached.fetch_add(1, std::memory_order_acq_rel);
  if (!wait_until([&] { return c->hdr()->attached.load(std::memory_order_acquire) >= (uint64_t)nranks; })) return kSystemError;
  if (rank == 0) shm_unlink(c->name);  // everybody holds a mapping: the name can go (nothing is left behind when a rank dies)
  *comm = c;
  return kSuccess;
}

int ncclCommDestroy(void *comm) {
  Comm *c = static_cast<Comm *>(comm);
  if (!c) return kSuccess;
  munmap(c->base, c->total);
  delete c;
  return kSuccess;
}

int ncclGroupStart() {
  ++g_depth;
  return kSuccess;
}

int ncclGroupEnd() {
  if (g_depth <= 0) return kInvalidArgument;
  if (--g_depth > 0) return kSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
  Comm *c = static_cast<Comm *>(comm);
  if (!c || !buf || peer < 0 || peer >= c->nranks || peer == c->rank || !dtype_bytes(dtype)) return kInvalidArgument;
  g_ops.push_back(Op{true, const_cast<void *>(buf), count * dtype_bytes(dtype), peer, c, stream});
  if (g_depth > 0) return kSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream) {
  Comm *c = static_cast<Comm *>(comm);
  if (!c || !buf || peer < 0 || peer >= c->nranks || peer == c->rank || !dtype_bytes(dtype)) return kInvalidArgument;
  g_ops.push_back(Op{false, buf, count * dtype_bytes(dtype), peer, c, stream});
  if (g_depth > 0) return kSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}

int ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, int dtype, int op, void *comm, hipStream_t stream) {
  Comm *c = static_cast<Comm *>(comm);
  if (!c || !sendbuf || !recvbuf || dtype != 7 || op != 0) return kInvalidArgument;
  const size_t bytes = count * 4;
  if (bytes > c->box_bytes) return kInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return kUnhandledCudaError;
  // slot of rank r = the diagonal mailbox (r, r), which no send ever uses
  if (hipMemcpy(c->payload(c->rank, c->rank), sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledCudaError;
  int rc = barrier(c);
  if (rc != kSuccess) return rc;
  std::vector<float> acc(count);
  std::memcpy(acc.data(), c->payload(0, 0), bytes);
  for (int r = 1; r < c->nranks; ++r) {
    const float *src = reinterpret_cast<const float *>(c->payload(r, r));
    for (size_t i = 0; i < count; ++i) acc[i] += src[i];
  }
  rc = barrier(c);  // nobody overwrites a slot before everybody has read it
  if (rc != kSuccess) return rc;
  if (hipMemcpy(recvbuf, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return kUnhandledCudaError;
  return kSuccess;
}

const char *ncclGetErrorString(int rc) {
  switch (rc) {
    case kSuccess: return "no error";
    case kUnhandledCudaError: return "fake rccl: a HIP call failed";
    case kSystemError: return "fake rccl: timed out waiting for a peer, or shared memory could not be set up";
    case kInvalidArgument: return "fake rccl: invalid argument (only float32 sum, buffers up to LICOS_FAKE_RCCL_BOX_BYTES)";
    default: return "fake rccl: internal error";
  }
}

}  // extern "C"
