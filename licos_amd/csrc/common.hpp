// Shared helpers for the LICOS gfx950 library (internal; the public ABI is include/licos_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <type_traits>
#include <utility>

#include "licos_hip.h"

namespace licos {

std::string &last_error_ref();
int fail(int code, const char *fmt, ...);

#define LICOS_REQUIRE(cond, ...)                              \
  do {                                                        \
    if (!(cond)) return ::licos::fail(LICOS_EINVAL, __VA_ARGS__); \
  } while (0)

#define LICOS_HIP_CHECK(expr)                                                                    \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess)                                                                        \
      return ::licos::fail(LICOS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                           __FILE__, __LINE__);                                                  \
  } while (0)

// every launch goes through this: catches bad launch configurations immediately
#define LICOS_LAUNCH_CHECK() LICOS_HIP_CHECK(hipGetLastError())

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

constexpr int WAVE = 64;

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>) - for bodies that need the index as a
// constant expression (sched_group_barrier sizes)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}


// Raises a kernel's dynamic-LDS limit to `bytes` on the CURRENT device, once per (device, kernel): the library is
// re-entrant per (device, stream) - a process-wide "already set" flag would skip the other devices of a
// multi-device process and race between threads.  host.cpp; returns a LICOS_* code.
int ensure_dynamic_lds(const void *kernel, int bytes);
#define LICOS_ENSURE_LDS(kernel, bytes)                                                  \
  do {                                                                                   \
    const int rc_ = ::licos::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), (int)(bytes)); \
    if (rc_ != LICOS_OK) return rc_;                                                     \
  } while (0)

}  // namespace licos
