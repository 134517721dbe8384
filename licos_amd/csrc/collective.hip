// licos_allreduce_weighted: the federated weight blend of /root/reference/licos/federation_utils.py:27-85
// as ONE call on one stream - scale the flat fp32 bucket by this rank's coefficient, RCCL all-reduce(SUM) over xGMI,
// normalise by the reduced coefficient sum (the bucket's last element carries it).  RCCL is bound at run time
// (dlopen; the copy PyTorch already loaded is reused when there is one), so the library has no link-time dependency
// on it and a process never holds two copies.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.hpp"

namespace licos {
namespace {

struct NcclUniqueId { char internal[128]; };  // rccl.h: NCCL_UNIQUE_ID_BYTES = 128
typedef void *ncclComm_t;
enum { kNcclFloat32 = 7, kNcclSum = 0 };  // ncclDataType_t / ncclRedOp_t values (nccl.h, stable ABI)

struct Rccl {
  int (*GetUniqueId)(NcclUniqueId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = nullptr;
    // LICOS_RCCL_TEST_LIB (tests only): a stand-in library with librccl's C API for processes that share ONE GPU
    // (tests/fake_rccl/fake_rccl.cpp) - RCCL refuses two ranks on one device, so on a one-GPU box the direct schedule's
    // send / recv loop below can only be exercised through it.  Unset in production: then this is librccl or nothing.
    const char *test_lib = getenv("LICOS_RCCL_TEST_LIB");
    if (test_lib && *test_lib) {
      h = dlopen(test_lib, RTLD_NOW | RTLD_LOCAL);
      if (!h) return;  // (never fall through to the real library when a test asked for the stand-in)
    }
    if (!h)
      for (const char *name : {"librccl.so", "librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // PyTorch's copy, if the process has one
        if (h) break;
      }
    if (!h)
      for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
      }
    if (!h) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString && r.Send && r.Recv &&
           r.GroupStart && r.GroupEnd;
  });
  return r;
}

// x[0 .. n-2] *= coef; x[n-1] = coef  (the coefficient rides in the bucket's last element)
__global__ void scale_and_tag_kernel(float *__restrict__ x, long n, float coef) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    x[e] = (e == n - 1) ? coef : x[e] * coef;
}
// x[0 .. n-2] /= x[n-1]
__global__ void normalise_kernel(float *__restrict__ x, long n) {
  const float inv = 1.0f / x[n - 1];
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n - 1; e += (long)gridDim.x * blockDim.x) x[e] *= inv;
}

// own[i] += sum over the peers p != rank, in ascending p, of scratch[p * chunk + i]  (fixed order: every rank computes
// ITS chunk once and ships the result, so all ranks end with identical bits)
__global__ void reduce_peers_kernel(float *__restrict__ own, const float *__restrict__ scratch, long chunk, int nranks, int rank) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < chunk; e += (long)gridDim.x * blockDim.x) {
    float acc = own[e];
    for (int p = 0; p < nranks; ++p)
      if (p != rank) acc += scratch[(size_t)p * chunk + e];
    own[e] = acc;
  }
}

}  // namespace
}  // namespace licos

using namespace licos;

#define LICOS_NCCL_CHECK(expr, what)                                                                  \
  do {                                                                                                \
    const int rc_ = (expr);                                                                           \
    if (rc_ != 0) return fail(LICOS_EHIP, "%s: RCCL error %d (%s)", what, rc_, rccl().GetErrorString(rc_)); \
  } while (0)

extern "C" {

int licos_comm_unique_id(void *out128) {
  LICOS_REQUIRE(out128 != nullptr, "comm_unique_id: NULL buffer");
  LICOS_REQUIRE(rccl().ok, "comm_unique_id: librccl.so could not be loaded");
  LICOS_NCCL_CHECK(rccl().GetUniqueId(static_cast<NcclUniqueId *>(out128)), "comm_unique_id");
  return LICOS_OK;
}

int licos_comm_init(void **comm, int nranks, int rank, const void *id128) {
  LICOS_REQUIRE(comm && id128 && nranks > 0 && rank >= 0 && rank < nranks, "comm_init: bad arguments");
  LICOS_REQUIRE(rccl().ok, "comm_init: librccl.so could not be loaded");
  NcclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  LICOS_NCCL_CHECK(rccl().CommInitRank(&c, nranks, id, rank), "comm_init");
  *comm = c;
  return LICOS_OK;
}

int licos_comm_destroy(void *comm) {
  if (!comm) return LICOS_OK;
  LICOS_REQUIRE(rccl().ok, "comm_destroy: librccl.so could not be loaded");
  LICOS_NCCL_CHECK(rccl().CommDestroy(static_cast<ncclComm_t>(comm)), "comm_destroy");
  return LICOS_OK;
}

int licos_allreduce_weighted(float *bucket, long n, float coef, void *comm, void *stream) {
  LICOS_REQUIRE(bucket && n >= 2 && comm, "allreduce_weighted: bad arguments (the bucket's last element carries the coefficient)");
  LICOS_REQUIRE(rccl().ok, "allreduce_weighted: librccl.so could not be loaded");
  hipStream_t s = as_stream(stream);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_and_tag_kernel, dim3(blocks), dim3(256), 0, s, bucket, n, coef);
  LICOS_LAUNCH_CHECK();
  LICOS_NCCL_CHECK(rccl().AllReduce(bucket, bucket, (size_t)n, kNcclFloat32, kNcclSum, static_cast<ncclComm_t>(comm), s), "allreduce_weighted");
  hipLaunchKernelGGL(normalise_kernel, dim3(blocks), dim3(256), 0, s, bucket, n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

// The same blend on the DIRECT schedule: xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per
// GPU), so a 12 MB bucket does not need a ring's 2 (N - 1) dependent steps.  Two steps, every link busy in both:
//   1. rank r sends chunk p of its (scaled) bucket to rank p, for every peer, and receives the peers' chunk r  (grouped
//      ncclSend / ncclRecv); it adds them to its own chunk r in a fixed order
//   2. rank r sends the reduced chunk r to every peer and receives theirs in place
// `n_alloc` >= nranks * ceil(n / nranks) elements must be addressable behind `bucket` (the tail past n is padding the
// caller owns); `scratch`: nranks * ceil(n / nranks) floats.
int licos_allreduce_weighted_direct(float *bucket, long n, long n_alloc, float coef, void *comm, int nranks, int rank,
                                    float *scratch, void *stream) {
  LICOS_REQUIRE(bucket && scratch && n >= 2 && comm && nranks >= 1 && rank >= 0 && rank < nranks, "allreduce_weighted_direct: bad arguments");
  LICOS_REQUIRE(rccl().ok, "allreduce_weighted_direct: librccl.so could not be loaded");
  const long chunk = (n + nranks - 1) / nranks;
  LICOS_REQUIRE(n_alloc >= chunk * nranks, "allreduce_weighted_direct: the bucket must be padded to %ld elements (%d ranks)", chunk * nranks, nranks);
  hipStream_t s = as_stream(stream);
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_and_tag_kernel, dim3(blocks), dim3(256), 0, s, bucket, n, coef);
  LICOS_LAUNCH_CHECK();
  if (nranks > 1) {
    LICOS_NCCL_CHECK(rccl().GroupStart(), "allreduce_weighted_direct");
    for (int p = 0; p < nranks; ++p) {
      if (p == rank) continue;
      LICOS_NCCL_CHECK(rccl().Send(bucket + (size_t)p * chunk, (size_t)chunk, kNcclFloat32, p, c, s), "allreduce_weighted_direct");
      LICOS_NCCL_CHECK(rccl().Recv(scratch + (size_t)p * chunk, (size_t)chunk, kNcclFloat32, p, c, s), "allreduce_weighted_direct");
    }
    LICOS_NCCL_CHECK(rccl().GroupEnd(), "allreduce_weighted_direct");
    const int rb = (int)((chunk + 255) / 256 < 2048 ? (chunk + 255) / 256 : 2048);
    hipLaunchKernelGGL(reduce_peers_kernel, dim3(rb), dim3(256), 0, s, bucket + (size_t)rank * chunk, scratch, chunk, nranks, rank);
    LICOS_LAUNCH_CHECK();
    LICOS_NCCL_CHECK(rccl().GroupStart(), "allreduce_weighted_direct");
    for (int p = 0; p < nranks; ++p) {
      if (p == rank) continue;
      LICOS_NCCL_CHECK(rccl().Send(bucket + (size_t)rank * chunk, (size_t)chunk, kNcclFloat32, p, c, s), "allreduce_weighted_direct");
      LICOS_NCCL_CHECK(rccl().Recv(bucket + (size_t)p * chunk, (size_t)chunk, kNcclFloat32, p, c, s), "allreduce_weighted_direct");
    }
    LICOS_NCCL_CHECK(rccl().GroupEnd(), "allreduce_weighted_direct");
  }
  hipLaunchKernelGGL(normalise_kernel, dim3(blocks), dim3(256), 0, s, bucket, n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // extern "C"
