// Host-side rANS coder (one std::thread per group of streams), bit-identical to the device coder (rans.hip) and to
// CompressAI's RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes
// (compressai/cpp_exts/rans/rans_interface.cpp over third_party/ryg_rans/rans64.h; reached from
// /root/reference/eval_utils.py:201 net.compress and eval_script.py:138-165, which code ONE whole granule = one stream).
//
// Why it exists next to the device kernels: the rANS recurrence is sequential inside a stream, a GPU lane spends
// ~160 ns (encode) / ~330 ns (decode) per symbol whatever the batch, a host core 3-6 ns.  One lane per stream wins
// from a few hundred streams up; for a handful of tiles, or one multi-megasymbol granule, the host cores do
// ("C++ host (thread-per-tile) and/or HIP").  The transforms, quantisation and
// dequantisation stay on the device; only int32 symbols cross PCIe (196 KB per 256x256 tile).
//
// Format facts restated from SURVEY.md section 8(a) row A7: 64-bit state, L = 2^31, 32-bit words, 16-bit
// probabilities, 4-bit bypass digits; symbols coded in reverse so the decoder reads forward; an escape symbol (index
// max = cdf_len - 2) is followed by the nibble count in base-15 "unary" chunks and then the nibbles of
// raw = (v < 0 ? -2v - 1 : 2 (v - max)), least significant first.
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"

namespace licos {
namespace {

constexpr uint64_t RANS_L = 1ull << 31;
constexpr int PREC = 16, BYPASS_BITS = 4, BYPASS_MAX = 15;

struct EncRec { uint64_t rcp; uint32_t bias; uint16_t freq; uint16_t shift; };  // licos_rans_build_enc_table (host.cpp)

struct Tables {
  const int32_t *cdf, *cdf_len, *offset;
  int cdf_stride, rows;
};

inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

// writes words backwards from `top`; returns the new write pointer or nullptr on overflow of `floor`
struct WordSink {
  uint32_t *ptr, *floor;
  bool ok = true;
  inline void put(uint32_t w) {
    if (ptr == floor) { ok = false; return; }
    *--ptr = w;
  }
};

inline void enc_bypass(uint64_t &x, WordSink &out, uint32_t nibble) {
  // freq = 2^(16-4): x_max = ((L >> 16) << 32) << 12 = 2^59
  if (x >= (1ull << 59)) { out.put((uint32_t)x); x >>= 32; }
  x = (x << BYPASS_BITS) | nibble;
}

// one stream, symbols sym[i * stride] for i = n-1 .. 0; row of position i = idx ? idx[i * stride] : i / plane
long encode_stream(const int32_t *sym, const int32_t *idx, long stride, int n, int plane, const Tables &t,
                   const EncRec *enc, uint32_t *buf, long cap_words) {
  WordSink out{buf + cap_words, buf};
  uint64_t x = RANS_L;
  int row = -1, max_value = 0, off = 0;
  const int32_t *cdf = nullptr;
  const EncRec *er = nullptr;
  int next_row_change = n;  // channel-id rows change every `plane` positions: re-derive only then
  for (int i = n - 1; i >= 0; --i) {
    if (idx) {
      const int r = idx[(long)i * stride];
      if (r != row) {
        if (r < 0 || r >= t.rows) return -2;
        row = r;
        cdf = t.cdf + (size_t)row * t.cdf_stride;
        er = enc + (size_t)row * t.cdf_stride;
        max_value = t.cdf_len[row] - 2;
        off = t.offset[row];
      }
    } else if (i < next_row_change) {
      row = i / plane;
      if (row >= t.rows) return -2;
      next_row_change = row * plane;
      cdf = t.cdf + (size_t)row * t.cdf_stride;
      er = enc + (size_t)row * t.cdf_stride;
      max_value = t.cdf_len[row] - 2;
      off = t.offset[row];
    }
    if (max_value < 0) return -2;
    int value = sym[(long)i * stride] - off;
    if (value < 0 || value >= max_value) {
      // escape: the list order is [symbol max][count chunks][nibbles lsb first]; coding runs through it backwards
      const uint32_t raw = value < 0 ? (uint32_t)(-2 * (int64_t)value - 1) : (uint32_t)(2 * ((int64_t)value - max_value));
      int nb = 0;
      while (nb < 8 && (raw >> (nb * BYPASS_BITS)) != 0) ++nb;
      for (int j = nb - 1; j >= 0; --j) enc_bypass(x, out, (raw >> (j * BYPASS_BITS)) & BYPASS_MAX);
      int val = nb, chunks = 0;
      while (val >= BYPASS_MAX) { val -= BYPASS_MAX; ++chunks; }
      enc_bypass(x, out, (uint32_t)val);
      for (int c = 0; c < chunks; ++c) enc_bypass(x, out, BYPASS_MAX);
      value = max_value;
    }
    const EncRec &e = er[value];
    const uint32_t freq = e.freq ? e.freq : 65536u;
    if (x >= ((uint64_t)freq << 47)) { out.put((uint32_t)x); x >>= 32; }  // x_max = ((L >> 16) << 32) * freq
    // ((x / freq) << 16) + (x % freq) + start  ==  x + bias + (x / freq) * (2^16 - freq), q by Alverson reciprocal
    const uint64_t q = mulhi64(x, e.rcp) >> e.shift;
    x = x + e.bias + q * (uint64_t)(65536u - freq);
    (void)cdf;
  }
  out.put((uint32_t)(x >> 32));
  out.put((uint32_t)x);
  if (!out.ok) return -1;
  return (long)((buf + cap_words) - out.ptr);
}

struct WordSource {
  const uint8_t *p, *end;
  bool overrun = false;
  inline uint32_t get() {
    if (p + 4 > end) { overrun = true; return 0; }
    uint32_t w;
    std::memcpy(&w, p, 4);
    p += 4;
    return w;
  }
};

inline uint32_t dec_bypass(uint64_t &x, WordSource &in) {
  const uint32_t v = (uint32_t)x & BYPASS_MAX;
  x >>= BYPASS_BITS;
  if (x < RANS_L) x = (x << 32) | in.get();
  return v;
}

// per-row search accelerator: lut[row][cf >> 8] = largest s with cdf[s] <= (cf & ~255)  (then scan upwards)
struct DecLut {
  std::vector<uint16_t> first;  // [rows][256]
  void build(const Tables &t) {
    first.assign((size_t)t.rows * 256, 0);
    for (int r = 0; r < t.rows; ++r) {
      const int32_t *cdf = t.cdf + (size_t)r * t.cdf_stride;
      const int len = t.cdf_len[r];
      int s = 0;
      for (int b = 0; b < 256; ++b) {
        const int cf = b << 8;
        while (s + 2 < len && cdf[s + 1] <= cf) ++s;
        first[(size_t)r * 256 + b] = (uint16_t)s;
      }
    }
  }
};

int decode_stream(const uint8_t *data, long nbytes, int32_t *sym, const int32_t *idx, long stride, int n, int plane,
                  const Tables &t, const DecLut &lut) {
  WordSource in{data, data + nbytes};
  uint64_t x = in.get();
  x |= (uint64_t)in.get() << 32;
  if (in.overrun) {  // not even the initial state: nothing to decode from
    for (int k = 0; k < n; ++k) sym[(long)k * stride] = 0;
    return 1;
  }
  int row = -1, max_value = 0, off = 0, len = 0;
  const int32_t *cdf = nullptr;
  const uint16_t *first = nullptr;
  int next_row_change = 0;
  for (int i = 0; i < n; ++i) {
    bool change = false;
    int r = row;
    if (idx) {
      r = idx[(long)i * stride];
      change = r != row;
    } else if (i >= next_row_change) {
      r = i / plane;
      next_row_change = (r + 1) * plane;
      change = true;
    }
    if (change) {
      if (r < 0 || r >= t.rows) return -2;
      row = r;
      cdf = t.cdf + (size_t)row * t.cdf_stride;
      first = lut.first.data() + (size_t)row * 256;
      len = t.cdf_len[row];
      max_value = len - 2;
      off = t.offset[row];
      if (max_value < 0) return -2;
    }
    const uint32_t cf = (uint32_t)x & 0xFFFF;
    int s = first[cf >> 8];
    while (s + 2 < len && (uint32_t)cdf[s + 1] <= cf) ++s;  // first s with cdf[s + 1] > cf, limited to the row
    const uint32_t start = (uint32_t)cdf[s], range = (uint32_t)cdf[s + 1] - start;
    x = (uint64_t)range * (x >> PREC) + cf - start;
    if (x < RANS_L) x = (x << 32) | in.get();
    int value = s;
    if (s == max_value) {
      int val = (int)dec_bypass(x, in), nb = val;
      while (val == BYPASS_MAX) {
        val = (int)dec_bypass(x, in);
        nb += val;
        if (in.overrun) break;
      }
      uint32_t raw = 0;
      for (int j = 0; j < nb; ++j) {
        const uint32_t d = dec_bypass(x, in);
        if (j < 8) raw |= d << (j * BYPASS_BITS);
        if (in.overrun) break;
      }
      const int v = (int)(raw >> 1);
      value = (raw & 1) ? -v - 1 : v + max_value;
    }
    sym[(long)i * stride] = value + off;
    if (in.overrun) {  // a truncated / corrupt stream: finish with zeros, never read outside it
      for (int k = i + 1; k < n; ++k) sym[(long)k * stride] = 0;
      return 1;
    }
  }
  return 0;
}

template <class F>
void parallel_streams(int batch, int nthreads, F &&f) {
  nthreads = nthreads < 1 ? 1 : (nthreads > batch ? batch : nthreads);
  if (nthreads == 1) {
    for (int b = 0; b < batch; ++b) f(b);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  pool.reserve(nthreads);
  for (int tix = 0; tix < nthreads; ++tix)
    pool.emplace_back([&]() {
      for (int b = next.fetch_add(1); b < batch; b = next.fetch_add(1)) f(b);
    });
  for (auto &th : pool) th.join();
}

}  // namespace
}  // namespace licos

using namespace licos;

extern "C" {

int licos_rans_encode_host(const int32_t *symbols, const int32_t *indexes, long sym_stride_b, long sym_stride_i, int n,
                           int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset,
                           int rows, const void *enc_table, uint8_t *out, long cap_bytes_per_stream, int64_t *nbytes,
                           int batch, int nthreads) {
  LICOS_REQUIRE(symbols && cdf && cdf_len && offset && enc_table && out && nbytes, "rans_encode_host: NULL buffer");
  LICOS_REQUIRE(n >= 0 && batch >= 0 && rows > 0 && cdf_stride > 1, "rans_encode_host: bad shape");
  LICOS_REQUIRE(indexes || plane > 0, "rans_encode_host: plane must be positive without explicit indexes");
  LICOS_REQUIRE(cap_bytes_per_stream >= 8 && cap_bytes_per_stream % 4 == 0, "rans_encode_host: capacity must be a multiple of 4, >= 8");
  const Tables t{cdf, cdf_len, offset, cdf_stride, rows};
  const EncRec *enc = static_cast<const EncRec *>(enc_table);
  std::atomic<int> worst{0};
  parallel_streams(batch, nthreads, [&](int b) {
    uint32_t *buf = reinterpret_cast<uint32_t *>(out + (size_t)b * cap_bytes_per_stream);
    const long cap_words = cap_bytes_per_stream / 4;
    const long nw = encode_stream(symbols + (size_t)b * sym_stride_b, indexes ? indexes + (size_t)b * sym_stride_b : nullptr,
                                  sym_stride_i, n, plane, t, enc, buf, cap_words);
    if (nw < 0) {
      nbytes[b] = 0;
      int code = (int)-nw, prev = worst.load();
      while (code > prev && !worst.compare_exchange_weak(prev, code)) {}
      return;
    }
    // the stream was written from the top of its slot: move it to the front
    if (nw != cap_words) std::memmove(buf, buf + (cap_words - nw), (size_t)nw * 4);
    nbytes[b] = nw * 4;
  });
  if (worst.load() == 1) return fail(LICOS_EOVERFLOW, "rans_encode_host: a stream does not fit %ld bytes", cap_bytes_per_stream);
  if (worst.load() == 2) return fail(LICOS_EINVAL, "rans_encode_host: CDF row out of range or empty");
  return LICOS_OK;
}

int licos_rans_decode_host(const uint8_t *in, const int64_t *byte_off, const int32_t *indexes, long sym_stride_b,
                           long sym_stride_i, int n, int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len,
                           const int32_t *offset, int rows, int32_t *symbols, int32_t *status, int batch, int nthreads) {
  LICOS_REQUIRE(in && byte_off && cdf && cdf_len && offset && symbols && status, "rans_decode_host: NULL buffer");
  LICOS_REQUIRE(n >= 0 && batch >= 0 && rows > 0 && cdf_stride > 1, "rans_decode_host: bad shape");
  LICOS_REQUIRE(indexes || plane > 0, "rans_decode_host: plane must be positive without explicit indexes");
  for (int r = 0; r < rows; ++r)
    LICOS_REQUIRE(cdf_len[r] >= 2 && cdf_len[r] <= cdf_stride, "rans_decode_host: row %d has cdf length %d", r, cdf_len[r]);
  const Tables t{cdf, cdf_len, offset, cdf_stride, rows};
  DecLut lut;
  lut.build(t);
  std::atomic<int> worst{0};
  parallel_streams(batch, nthreads, [&](int b) {
    const long nb = (long)(byte_off[b + 1] - byte_off[b]);
    int rc = nb < 0 ? 2 : decode_stream(in + byte_off[b], nb, symbols + (size_t)b * sym_stride_b,
                                        indexes ? indexes + (size_t)b * sym_stride_b : nullptr, sym_stride_i, n, plane, t, lut);
    if (rc < 0) rc = 2;
    int prev = worst.load();
    while (rc > prev && !worst.compare_exchange_weak(prev, rc)) {}
  });
  if (worst.load() == 2) return fail(LICOS_EINVAL, "rans_decode_host: CDF row out of range or bad stream offsets");
  status[0] = worst.load();  // 1: some stream ended before all its symbols were decoded
  return LICOS_OK;
}

}  // extern "C"
