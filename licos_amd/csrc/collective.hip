// licos_allreduce_weighted: the federated weight blend of /root/reference/licos/federation_utils.py:27-85
// as ONE call on one stream - scale the flat fp32 bucket by this rank's coefficient, RCCL all-reduce(SUM) over xGMI,
// normalise by the reduced coefficient sum (the bucket's last element carries it).  RCCL is bound at run time
// (dlopen; the copy PyTorch already loaded is reused when there is one), so the library has no link-time dependency
// on it and a process never holds two copies.
#include <dlfcn.h>

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
  const char *(*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void *h = nullptr;
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
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
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

}  // extern "C"
