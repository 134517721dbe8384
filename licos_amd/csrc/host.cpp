// Host-side entry points: error plumbing, device query, integer CDF construction and
// the encoder reciprocal table.  No device code here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "common.hpp"

namespace licos {
std::string &last_error_ref() {
  static thread_local std::string s;
  return s;
}
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}
int ensure_dynamic_lds(const void *kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, int> done;  // (device, kernel) -> limit already granted
  int dev = 0;
  LICOS_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int &have = done[std::make_pair(dev, kernel)];
  if (have >= bytes) return LICOS_OK;
  LICOS_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
  return LICOS_OK;
}
}  // namespace licos

using namespace licos;

extern "C" {

const char *licos_last_error(void) { return last_error_ref().c_str(); }
int licos_abi_version(void) { return LICOS_ABI_VERSION; }

int licos_query(int device, licos_device_props *out) {
  LICOS_REQUIRE(out != nullptr, "licos_query: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(LICOS_EHIP, "licos_query: no HIP device (%s)", hipGetErrorString(e));
  LICOS_REQUIRE(device >= 0 && device < n, "licos_query: device %d out of range (%d devices)", device, n);
  hipDeviceProp_t p;
  LICOS_HIP_CHECK(hipGetDeviceProperties(&p, device));
  out->compute_units = p.multiProcessorCount;
  out->wavefront_size = p.warpSize;
  out->lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
  out->clock_khz = p.clockRate;
  out->hbm_bytes = p.totalGlobalMem;
  std::memset(out->arch, 0, sizeof(out->arch));
  std::strncpy(out->arch, p.gcnArchName, sizeof(out->arch) - 1);
  return LICOS_OK;
}

// Integer CDF from a float pmf.  Behaviour follows CompressAI's pmf_to_quantized_cdf
// (SURVEY.md section 8(c)): round-half-away scaling, renormalise with integer division, prefix
// sum, then give every zero-width bin one count taken from the narrowest bin wider than 1.
int licos_pmf_to_quantized_cdf(const float *pmf, int n, int precision, int32_t *cdf_out) {
  LICOS_REQUIRE(pmf && cdf_out && n > 0, "pmf_to_quantized_cdf: bad arguments");
  LICOS_REQUIRE(precision > 0 && precision <= 16, "pmf_to_quantized_cdf: precision %d unsupported", precision);
  for (int i = 0; i < n; ++i)
    if (!(pmf[i] >= 0.0f) || !std::isfinite(pmf[i]))
      return fail(LICOS_EDOMAIN, "Invalid `pmf`, non-finite or negative element found: %g", (double)pmf[i]);
  std::vector<uint32_t> c(n + 1);
  const float scale = (float)(1u << precision);
  uint32_t total = 0;
  c[0] = 0;
  for (int i = 0; i < n; ++i) {
    c[i + 1] = (uint32_t)std::round(pmf[i] * scale);
    total += c[i + 1];
  }
  if (total == 0) return fail(LICOS_EDOMAIN, "Invalid `pmf`: at least one element must have a non-zero probability.");
  const uint64_t one = 1ull << precision;
  uint32_t run = 0;
  for (int i = 0; i <= n; ++i) {
    run += (uint32_t)((one * c[i]) / total);
    c[i] = run;
  }
  c[n] = (uint32_t)one;
  for (int i = 0; i < n; ++i) {
    if (c[i] != c[i + 1]) continue;
    int donor = -1;
    uint32_t donor_width = ~0u;
    for (int j = 0; j < n; ++j) {
      const uint32_t width = c[j + 1] - c[j];
      if (width > 1 && width < donor_width) { donor_width = width; donor = j; }
    }
    if (donor < 0) return fail(LICOS_EDOMAIN, "pmf_to_quantized_cdf: cannot make every bin non-empty");
    if (donor < i) for (int j = donor + 1; j <= i; ++j) c[j]--;
    else for (int j = i + 1; j <= donor; ++j) c[j]++;
  }
  for (int i = 0; i <= n; ++i) cdf_out[i] = (int32_t)c[i];
  return LICOS_OK;
}

// 16-byte encoder record per (row, symbol): q = mulhi64(x, rcp) >> shift equals x / freq for
// every 64-bit x (Alverson's round-up reciprocal), so x' = x + bias + q * (2^16 - freq)
// reproduces ((x / freq) << 16) + (x % freq) + start bit for bit.  freq == 1 uses rcp = 2^64-1,
// shift 0 (q = x - 1) with the bias corrected by 2^16 - 1.
struct EncRec { uint64_t rcp; uint32_t bias; uint16_t freq; uint16_t shift; };
static_assert(sizeof(EncRec) == 16, "EncRec must be 16 bytes");

int licos_rans_build_enc_table(const int32_t *cdf, const int32_t *cdf_len, int rows, int stride, void *table_out) {
  LICOS_REQUIRE(cdf && cdf_len && table_out && rows > 0 && stride > 1, "rans_build_enc_table: bad arguments");
  EncRec *t = static_cast<EncRec *>(table_out);
  std::memset(t, 0, sizeof(EncRec) * (size_t)rows * stride);
  for (int r = 0; r < rows; ++r) {
    const int len = cdf_len[r];
    LICOS_REQUIRE(len >= 2 && len <= stride, "rans_build_enc_table: row %d has cdf length %d (stride %d)", r, len, stride);
    for (int s = 0; s + 1 < len; ++s) {
      const int64_t start = cdf[(size_t)r * stride + s];
      const int64_t freq = (int64_t)cdf[(size_t)r * stride + s + 1] - start;
      LICOS_REQUIRE(start >= 0 && freq > 0 && start + freq <= 65536, "rans_build_enc_table: row %d symbol %d has start %ld freq %ld", r, s, (long)start, (long)freq);
      EncRec &e = t[(size_t)r * stride + s];
      e.freq = (uint16_t)(freq & 0xFFFF);  // 65536 (single-symbol row) wraps to 0; handled below
      if (freq == 65536) {
        // x / 65536 == x >> 16 : rcp = 2^63, shift = 15 -> mulhi(x, 2^63) >> 15 == x >> 16
        e.rcp = 1ull << 63; e.shift = 15; e.bias = (uint32_t)start;
      } else if (freq == 1) {
        e.rcp = ~0ull; e.shift = 0; e.bias = (uint32_t)(start + 65535);
      } else {
        int sh = 0;
        while (freq > (1ll << sh)) ++sh;
        const unsigned __int128 num = ((unsigned __int128)1 << (sh + 63)) + (unsigned __int128)(freq - 1);
        e.rcp = (uint64_t)(num / (unsigned __int128)freq);
        e.shift = (uint16_t)(sh - 1);
        e.bias = (uint32_t)start;
      }
    }
  }
  return LICOS_OK;
}

}  // extern "C"
