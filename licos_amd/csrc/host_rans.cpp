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
#include <pthread.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <type_traits>
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


// K streams in lockstep (same n, same addressing).  The chain of one symbol - table entry, renormalise, reciprocal
// multiply, add - is ~11 dependent cycles with nothing beside it; K independent chains in one loop body fill the core's
// issue width (measured: 3.5 -> ~1.2 ns per symbol and thread at K = 4).  Same arithmetic, same bytes as encode_stream.
template <int K, typename SymT = int32_t>
void encode_streams(const SymT *const (&sym)[K], const int32_t *const (&idx)[K], long stride, int n, int plane, const Tables &t,
                    const EncRec *enc, uint32_t *const (&buf)[K], long cap_words, long (&nwords)[K], bool packed = false) {
  WordSink out[K];
  uint64_t x[K];
  int row[K], max_value[K], off[K], next_row_change[K];
  const EncRec *er[K];
  bool bad[K];
  for (int k = 0; k < K; ++k) {
    out[k] = WordSink{buf[k] + cap_words, buf[k]};
    x[k] = RANS_L;
    row[k] = -1;
    max_value[k] = off[k] = 0;
    next_row_change[k] = n;
    er[k] = nullptr;
    bad[k] = false;
  }
  bool planes = plane > 0 && (long)((n + plane - 1) / plane) <= (long)t.rows;
  for (int k = 0; k < K; ++k) planes = planes && idx[k] == nullptr;
  if (planes) {
    // channel-plane rows (the entropy bottleneck): the row is fixed over `plane` positions - its table pointer, symbol
    // range and offset leave the per-symbol work, which is then ~14 instructions per stream
    for (int r = (n - 1) / plane; r >= 0; --r) {
      const EncRec *e_row = enc + (size_t)r * t.cdf_stride;
      const int maxv = t.cdf_len[r] - 2, o = t.offset[r];
      if (maxv < 0) {
        for (int k = 0; k < K; ++k) bad[k] = true;
        break;
      }
      const int hi = (r + 1) * plane < n ? (r + 1) * plane : n;
      for (int i = hi - 1; i >= r * plane; --i) {
#pragma GCC unroll 8
        for (int k = 0; k < K; ++k) {
          int value = sym[k][(long)i * stride] - o;
          if (__builtin_expect((unsigned)value >= (unsigned)maxv, 0)) {
            const uint32_t raw = value < 0 ? (uint32_t)(-2 * (int64_t)value - 1) : (uint32_t)(2 * ((int64_t)value - maxv));
            int nb = 0;
            while (nb < 8 && (raw >> (nb * BYPASS_BITS)) != 0) ++nb;
            for (int j = nb - 1; j >= 0; --j) enc_bypass(x[k], out[k], (raw >> (j * BYPASS_BITS)) & BYPASS_MAX);
            int val = nb, chunks = 0;
            while (val >= BYPASS_MAX) { val -= BYPASS_MAX; ++chunks; }
            enc_bypass(x[k], out[k], (uint32_t)val);
            for (int c = 0; c < chunks; ++c) enc_bypass(x[k], out[k], BYPASS_MAX);
            value = maxv;
          }
          const EncRec &e = e_row[value];
          const uint32_t freq = e.freq ? e.freq : 65536u;
          if (__builtin_expect(x[k] >= ((uint64_t)freq << 47), 0)) { out[k].put((uint32_t)x[k]); x[k] >>= 32; }
          const uint64_t q = mulhi64(x[k], e.rcp) >> e.shift;
          x[k] = x[k] + e.bias + q * (uint64_t)(65536u - freq);
        }
      }
    }
  }
  bool indexed = !planes;
  for (int k = 0; k < K; ++k) indexed = indexed && idx[k] != nullptr;
  if (indexed) {
    // explicit per-symbol rows (the scale hyperprior's y stream: the row follows the predicted scale and changes from one
    // symbol to the next): the row's table pointer, range and offset are re-derived for EVERY symbol without a branch - a
    // "did the row change" test mispredicts on a quarter of the symbols of a trained model
    for (int i = n - 1; i >= 0; --i) {
#pragma GCC unroll 8
      for (int k = 0; k < K; ++k) {
        if (__builtin_expect(bad[k], 0)) continue;
        // packed (licos_rans_encode_host_packed): one word per symbol, row << 16 | (symbol & 0xFFFF) - half the bytes
        // over PCIe when the words come from the device (idx[k] == sym[k] then)
        const int32_t w = (int32_t)sym[k][(long)i * stride];
        const int r = packed ? (int)((uint32_t)w >> 16) : idx[k][(long)i * stride];
        if (__builtin_expect((unsigned)r >= (unsigned)t.rows, 0)) { bad[k] = true; continue; }
        const EncRec *e_row = enc + (size_t)r * t.cdf_stride;
        const int maxv = t.cdf_len[r] - 2, o = t.offset[r];
        if (__builtin_expect(maxv < 0, 0)) { bad[k] = true; continue; }
        int value = (packed ? (int)(int16_t)(uint16_t)(w & 0xFFFF) : w) - o;
        if (__builtin_expect((unsigned)value >= (unsigned)maxv, 0)) {
          const uint32_t raw = value < 0 ? (uint32_t)(-2 * (int64_t)value - 1) : (uint32_t)(2 * ((int64_t)value - maxv));
          int nb = 0;
          while (nb < 8 && (raw >> (nb * BYPASS_BITS)) != 0) ++nb;
          for (int j = nb - 1; j >= 0; --j) enc_bypass(x[k], out[k], (raw >> (j * BYPASS_BITS)) & BYPASS_MAX);
          int val = nb, chunks = 0;
          while (val >= BYPASS_MAX) { val -= BYPASS_MAX; ++chunks; }
          enc_bypass(x[k], out[k], (uint32_t)val);
          for (int c = 0; c < chunks; ++c) enc_bypass(x[k], out[k], BYPASS_MAX);
          value = maxv;
        }
        const EncRec &e = e_row[value];
        const uint32_t freq = e.freq ? e.freq : 65536u;
        if (__builtin_expect(x[k] >= ((uint64_t)freq << 47), 0)) { out[k].put((uint32_t)x[k]); x[k] >>= 32; }
        const uint64_t q = mulhi64(x[k], e.rcp) >> e.shift;
        x[k] = x[k] + e.bias + q * (uint64_t)(65536u - freq);
      }
    }
  }
  for (int i = (planes || indexed) ? -1 : n - 1; i >= 0; --i) {
#pragma GCC unroll 8
    for (int k = 0; k < K; ++k) {
      if (idx[k]) {
        const int r = idx[k][(long)i * stride];
        if (__builtin_expect(r != row[k], 0)) {
          if (r < 0 || r >= t.rows) { bad[k] = true; continue; }
          row[k] = r;
          er[k] = enc + (size_t)r * t.cdf_stride;
          max_value[k] = t.cdf_len[r] - 2;
          off[k] = t.offset[r];
        }
      } else if (__builtin_expect(i < next_row_change[k], 0)) {
        const int r = i / plane;
        if (r >= t.rows) { bad[k] = true; continue; }
        row[k] = r;
        next_row_change[k] = r * plane;
        er[k] = enc + (size_t)r * t.cdf_stride;
        max_value[k] = t.cdf_len[r] - 2;
        off[k] = t.offset[r];
      }
      if (__builtin_expect(bad[k] || max_value[k] < 0, 0)) { bad[k] = true; continue; }
      int value = sym[k][(long)i * stride] - off[k];
      if (__builtin_expect(value < 0 || value >= max_value[k], 0)) {
        const uint32_t raw = value < 0 ? (uint32_t)(-2 * (int64_t)value - 1) : (uint32_t)(2 * ((int64_t)value - max_value[k]));
        int nb = 0;
        while (nb < 8 && (raw >> (nb * BYPASS_BITS)) != 0) ++nb;
        for (int j = nb - 1; j >= 0; --j) enc_bypass(x[k], out[k], (raw >> (j * BYPASS_BITS)) & BYPASS_MAX);
        int val = nb, chunks = 0;
        while (val >= BYPASS_MAX) { val -= BYPASS_MAX; ++chunks; }
        enc_bypass(x[k], out[k], (uint32_t)val);
        for (int c = 0; c < chunks; ++c) enc_bypass(x[k], out[k], BYPASS_MAX);
        value = max_value[k];
      }
      const EncRec &e = er[k][value];
      const uint32_t freq = e.freq ? e.freq : 65536u;
      if (__builtin_expect(x[k] >= ((uint64_t)freq << 47), 0)) { out[k].put((uint32_t)x[k]); x[k] >>= 32; }
      const uint64_t q = mulhi64(x[k], e.rcp) >> e.shift;
      x[k] = x[k] + e.bias + q * (uint64_t)(65536u - freq);
    }
  }
  for (int k = 0; k < K; ++k) {
    out[k].put((uint32_t)(x[k] >> 32));
    out[k].put((uint32_t)x[k]);
    nwords[k] = bad[k] ? -2 : (!out[k].ok ? -1 : (long)((buf[k] + cap_words) - out[k].ptr));
  }
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


// K streams in lockstep: see encode_streams.  A decode step is state -> cumulative value -> table search -> multiply ->
// (rarely) one word in: ~25 dependent cycles alone.
// SymT = int16_t (licos_rans_decode_host_sym16): a decoded value outside 16 bits ends the stream with rc 3 (the caller
// decodes the batch again into 32-bit symbols).
template <typename SymT>
static inline bool put_symbol(SymT *p, int v) {
  *p = (SymT)v;
  return sizeof(SymT) >= 4 || v == (int)(SymT)v;
}

template <int K, typename IdxT = int32_t, typename SymT = int32_t>
void decode_streams(const uint8_t *const (&data)[K], const long (&nbytes)[K], SymT *const (&sym)[K], const IdxT *const (&idx)[K],
                    long stride, int n, int plane, const Tables &t, const DecLut &lut, int (&rc)[K]) {
  WordSource in[K];
  uint64_t x[K];
  int row[K], max_value[K], off[K], len[K], next_row_change[K];
  const int32_t *cdf[K];
  const uint16_t *first[K];
  bool done[K];
  for (int k = 0; k < K; ++k) {
    in[k] = WordSource{data[k], data[k] + nbytes[k]};
    x[k] = in[k].get();
    x[k] |= (uint64_t)in[k].get() << 32;
    rc[k] = 0;
    done[k] = false;
    row[k] = -1;
    max_value[k] = off[k] = len[k] = next_row_change[k] = 0;
    cdf[k] = nullptr;
    first[k] = nullptr;
    if (in[k].overrun) {
      for (int j = 0; j < n; ++j) sym[k][(long)j * stride] = 0;
      rc[k] = 1;
      done[k] = true;
    }
  }
  bool planes = plane > 0 && (long)((n + plane - 1) / plane) <= (long)t.rows;
  for (int k = 0; k < K; ++k) planes = planes && idx[k] == nullptr && !done[k];
  int i0 = 0;
  if (planes) {
    // channel-plane rows: the row's table pointers, length and offset are fixed over `plane` positions.  The segment
    // ends early (and the general loop below takes over at i0) as soon as a stream runs out of bytes.
    bool stop = false;
    for (int r = 0; r * plane < n && !stop; ++r) {
      if (t.cdf_len[r] - 2 < 0) break;  // (the general loop reports it)
      const int32_t *c_row = t.cdf + (size_t)r * t.cdf_stride;
      const uint16_t *f_row = lut.first.data() + (size_t)r * 256;
      const int ln = t.cdf_len[r], maxv = ln - 2, o = t.offset[r];
      const int hi = (r + 1) * plane < n ? (r + 1) * plane : n;
      for (int i = r * plane; i < hi && !stop; ++i) {
#pragma GCC unroll 8
        for (int k = 0; k < K; ++k) {
          const uint32_t cf = (uint32_t)x[k] & 0xFFFF;
          int s2 = f_row[cf >> 8];
          while (s2 + 2 < ln && (uint32_t)c_row[s2 + 1] <= cf) ++s2;
          const uint32_t start = (uint32_t)c_row[s2], range = (uint32_t)c_row[s2 + 1] - start;
          x[k] = (uint64_t)range * (x[k] >> PREC) + cf - start;
          if (__builtin_expect(x[k] < RANS_L, 0)) x[k] = (x[k] << 32) | in[k].get();
          int value = s2;
          if (__builtin_expect(s2 == maxv, 0)) {
            int val = (int)dec_bypass(x[k], in[k]), nb = val;
            while (val == BYPASS_MAX) {
              val = (int)dec_bypass(x[k], in[k]);
              nb += val;
              if (in[k].overrun) break;
            }
            uint32_t raw = 0;
            for (int j = 0; j < nb; ++j) {
              const uint32_t d = dec_bypass(x[k], in[k]);
              if (j < 8) raw |= d << (j * BYPASS_BITS);
              if (in[k].overrun) break;
            }
            const int v = (int)(raw >> 1);
            value = (raw & 1) ? -v - 1 : v + maxv;
          }
          if (__builtin_expect(!put_symbol(&sym[k][(long)i * stride], value + o), 0)) {
            rc[k] = 3;
            done[k] = true;
            stop = true;
          }
          if (__builtin_expect(in[k].overrun, 0)) {  // finish this stream here; the others continue in the general loop
            for (int j = i + 1; j < n; ++j) sym[k][(long)j * stride] = 0;
            rc[k] = 1;
            done[k] = true;
            stop = true;
          }
        }
        i0 = i + 1;
      }
    }
  }
  bool indexed = !planes;
  for (int k = 0; k < K; ++k) indexed = indexed && idx[k] != nullptr;
  if (indexed) {
    // explicit per-symbol rows: every row-dependent quantity re-derived per symbol, no "row changed" branch (see encode)
    for (int i = 0; i < n; ++i) {
#pragma GCC unroll 8
      for (int k = 0; k < K; ++k) {
        if (__builtin_expect(done[k], 0)) continue;
        const int r = idx[k][(long)i * stride];
        if (__builtin_expect((unsigned)r >= (unsigned)t.rows || t.cdf_len[r] - 2 < 0, 0)) { rc[k] = -2; done[k] = true; continue; }
        const int32_t *c_row = t.cdf + (size_t)r * t.cdf_stride;
        const uint16_t *f_row = lut.first.data() + (size_t)r * 256;
        const int ln = t.cdf_len[r], maxv = ln - 2, o = t.offset[r];
        const uint32_t cf = (uint32_t)x[k] & 0xFFFF;
        int s2 = f_row[cf >> 8];
        while (s2 + 2 < ln && (uint32_t)c_row[s2 + 1] <= cf) ++s2;
        const uint32_t start = (uint32_t)c_row[s2], range = (uint32_t)c_row[s2 + 1] - start;
        x[k] = (uint64_t)range * (x[k] >> PREC) + cf - start;
        if (__builtin_expect(x[k] < RANS_L, 0)) x[k] = (x[k] << 32) | in[k].get();
        int value = s2;
        if (__builtin_expect(s2 == maxv, 0)) {
          int val = (int)dec_bypass(x[k], in[k]), nb = val;
          while (val == BYPASS_MAX) {
            val = (int)dec_bypass(x[k], in[k]);
            nb += val;
            if (in[k].overrun) break;
          }
          uint32_t raw = 0;
          for (int j = 0; j < nb; ++j) {
            const uint32_t d = dec_bypass(x[k], in[k]);
            if (j < 8) raw |= d << (j * BYPASS_BITS);
            if (in[k].overrun) break;
          }
          const int v = (int)(raw >> 1);
          value = (raw & 1) ? -v - 1 : v + maxv;
        }
        if (__builtin_expect(!put_symbol(&sym[k][(long)i * stride], value + o), 0)) { rc[k] = 3; done[k] = true; }
        if (__builtin_expect(in[k].overrun, 0)) {
          for (int j = i + 1; j < n; ++j) sym[k][(long)j * stride] = 0;
          rc[k] = 1;
          done[k] = true;
        }
      }
    }
    i0 = n;
  }
  for (int i = i0; i < n; ++i) {
#pragma GCC unroll 8
    for (int k = 0; k < K; ++k) {
      if (__builtin_expect(done[k], 0)) continue;
      bool change = false;
      int r = row[k];
      if (idx[k]) {
        r = idx[k][(long)i * stride];
        change = r != row[k];
      } else if (i >= next_row_change[k]) {
        r = i / plane;
        next_row_change[k] = (r + 1) * plane;
        change = true;
      }
      if (__builtin_expect(change, 0)) {
        if (r < 0 || r >= t.rows || t.cdf_len[r] - 2 < 0) { rc[k] = -2; done[k] = true; continue; }
        row[k] = r;
        cdf[k] = t.cdf + (size_t)r * t.cdf_stride;
        first[k] = lut.first.data() + (size_t)r * 256;
        len[k] = t.cdf_len[r];
        max_value[k] = len[k] - 2;
        off[k] = t.offset[r];
      }
      const uint32_t cf = (uint32_t)x[k] & 0xFFFF;
      int s2 = first[k][cf >> 8];
      while (s2 + 2 < len[k] && (uint32_t)cdf[k][s2 + 1] <= cf) ++s2;
      const uint32_t start = (uint32_t)cdf[k][s2], range = (uint32_t)cdf[k][s2 + 1] - start;
      x[k] = (uint64_t)range * (x[k] >> PREC) + cf - start;
      if (__builtin_expect(x[k] < RANS_L, 0)) x[k] = (x[k] << 32) | in[k].get();
      int value = s2;
      if (__builtin_expect(s2 == max_value[k], 0)) {
        int val = (int)dec_bypass(x[k], in[k]), nb = val;
        while (val == BYPASS_MAX) {
          val = (int)dec_bypass(x[k], in[k]);
          nb += val;
          if (in[k].overrun) break;
        }
        uint32_t raw = 0;
        for (int j = 0; j < nb; ++j) {
          const uint32_t d = dec_bypass(x[k], in[k]);
          if (j < 8) raw |= d << (j * BYPASS_BITS);
          if (in[k].overrun) break;
        }
        const int v = (int)(raw >> 1);
        value = (raw & 1) ? -v - 1 : v + max_value[k];
      }
      if (__builtin_expect(!put_symbol(&sym[k][(long)i * stride], value + off[k]), 0)) { rc[k] = 3; done[k] = true; }
      if (__builtin_expect(in[k].overrun, 0)) {
        for (int j = i + 1; j < n; ++j) sym[k][(long)j * stride] = 0;
        rc[k] = 1;
        done[k] = true;
      }
    }
  }
}

// A persistent pool: the coder is called per batch of a few tiles (1 ms of work at B = 16), where starting and joining
// 16 std::threads per call cost as much as the coding.  Workers sleep on a condition variable between calls; the
// calling thread works too.  One call at a time (calls from several threads serialise on `run_mutex`).  After fork()
// the child owns no workers: they are started again on first use.
class Pool {
 public:
  static Pool &instance() {
    static Pool *p = new Pool();  // never destroyed: workers may still be parked at process exit
    return *p;
  }
  void run(int njobs, int nthreads, const std::function<void(int)> &f) {
    nthreads = nthreads < 1 ? 1 : (nthreads > njobs ? njobs : nthreads);
    if (nthreads <= 1) {
      for (int j = 0; j < njobs; ++j) f(j);
      return;
    }
    std::lock_guard<std::mutex> serial(run_mutex_);
    {
      std::unique_lock<std::mutex> lk(m_);
      while ((int)workers_ < nthreads - 1) {
        std::thread(&Pool::worker, this, workers_).detach();
        ++workers_;
      }
      job_ = &f;
      njobs_ = njobs;
      next_.store(0);
      active_ = nthreads - 1;   // workers 0 .. nthreads-2 take part in this call
      pending_ = nthreads - 1;
      ++generation_;
    }
    cv_.notify_all();
    for (int j = next_.fetch_add(1); j < njobs; j = next_.fetch_add(1)) f(j);
    std::unique_lock<std::mutex> lk(m_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  Pool() { pthread_atfork(nullptr, nullptr, &Pool::after_fork_child); }
  static void after_fork_child() {
    Pool &p = instance();
    new (&p.m_) std::mutex();
    new (&p.run_mutex_) std::mutex();
    new (&p.cv_) std::condition_variable();
    new (&p.done_cv_) std::condition_variable();
    p.workers_ = 0;
    p.pending_ = 0;
    p.job_ = nullptr;
  }
  void worker(unsigned id) {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)> *job;
      int njobs;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if ((int)id >= active_) continue;  // not needed for this call
        job = job_;
        njobs = njobs_;
      }
      for (int j = next_.fetch_add(1); j < njobs; j = next_.fetch_add(1)) (*job)(j);
      std::unique_lock<std::mutex> lk(m_);
      if (--pending_ == 0) done_cv_.notify_one();
    }
  }
  std::mutex m_, run_mutex_;
  std::condition_variable cv_, done_cv_;
  unsigned workers_ = 0;
  unsigned long generation_ = 0;
  const std::function<void(int)> *job_ = nullptr;
  int njobs_ = 0, active_ = 0, pending_ = 0;
  std::atomic<int> next_{0};
};

// streams per job: up to 4 in lockstep (encode_streams / decode_streams), fewer when that would leave threads idle
inline int lockstep_width(int batch, int nthreads) {
  nthreads = nthreads < 1 ? 1 : nthreads;
  const int per_thread = batch / nthreads;
  return per_thread >= 4 ? 4 : per_thread >= 2 ? 2 : 1;
}

}  // namespace
}  // namespace licos

using namespace licos;

template <typename SymT>
static int encode_host(const SymT *symbols, const int32_t *indexes, long sym_stride_b, long sym_stride_i, int n,
                       int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset,
                       int rows, const void *enc_table, uint8_t *out, long cap_bytes_per_stream, int64_t *nbytes,
                       int batch, int nthreads, bool packed) {
  LICOS_REQUIRE(symbols && cdf && cdf_len && offset && enc_table && out && nbytes, "rans_encode_host: NULL buffer");
  LICOS_REQUIRE(n >= 0 && batch >= 0 && rows > 0 && cdf_stride > 1, "rans_encode_host: bad shape");
  LICOS_REQUIRE(indexes || plane > 0, "rans_encode_host: plane must be positive without explicit indexes");
  LICOS_REQUIRE(cap_bytes_per_stream >= 8 && cap_bytes_per_stream % 4 == 0, "rans_encode_host: capacity must be a multiple of 4, >= 8");
  const Tables t{cdf, cdf_len, offset, cdf_stride, rows};
  const EncRec *enc = static_cast<const EncRec *>(enc_table);
  std::atomic<int> worst{0};
  const long cap_words = cap_bytes_per_stream / 4;
  auto finish = [&](int b, long nw) {
    uint32_t *buf = reinterpret_cast<uint32_t *>(out + (size_t)b * cap_bytes_per_stream);
    if (nw < 0) {
      nbytes[b] = 0;
      int code = (int)-nw, prev = worst.load();
      while (code > prev && !worst.compare_exchange_weak(prev, code)) {}
      return;
    }
    // the stream was written from the top of its slot: move it to the front
    if (nw != cap_words) std::memmove(buf, buf + (cap_words - nw), (size_t)nw * 4);
    nbytes[b] = nw * 4;
  };
  const int K = lockstep_width(batch, nthreads);
  const int njobs = (batch + K - 1) / K;
  auto run_group = [&](auto kc, int b0) {
    constexpr int KK = decltype(kc)::value;
    const SymT *sp[KK];
    const int32_t *ip[KK];
    uint32_t *bp[KK];
    long nw[KK];
    for (int k = 0; k < KK; ++k) {
      sp[k] = symbols + (size_t)(b0 + k) * sym_stride_b;
      ip[k] = indexes ? indexes + (size_t)(b0 + k) * sym_stride_b : nullptr;
      bp[k] = reinterpret_cast<uint32_t *>(out + (size_t)(b0 + k) * cap_bytes_per_stream);
    }
    encode_streams<KK, SymT>(sp, ip, sym_stride_i, n, plane, t, enc, bp, cap_words, nw, packed);
    for (int k = 0; k < KK; ++k) finish(b0 + k, nw[k]);
  };
  Pool::instance().run(njobs, nthreads, [&](int j) {
    const int b0 = j * K, cnt = batch - b0 < K ? batch - b0 : K;
    if (cnt == 4) run_group(std::integral_constant<int, 4>{}, b0);
    else if (cnt == 2) run_group(std::integral_constant<int, 2>{}, b0);
    else if (packed || !std::is_same<SymT, int32_t>::value)
      for (int k = 0; k < cnt; ++k) run_group(std::integral_constant<int, 1>{}, b0 + k);
    else if constexpr (std::is_same<SymT, int32_t>::value)
      for (int k = 0; k < cnt; ++k) {
        uint32_t *buf = reinterpret_cast<uint32_t *>(out + (size_t)(b0 + k) * cap_bytes_per_stream);
        finish(b0 + k, encode_stream(symbols + (size_t)(b0 + k) * sym_stride_b, indexes ? indexes + (size_t)(b0 + k) * sym_stride_b : nullptr,
                                     sym_stride_i, n, plane, t, enc, buf, cap_words));
      }
  });
  if (worst.load() == 1) return fail(LICOS_EOVERFLOW, "rans_encode_host: a stream does not fit %ld bytes", cap_bytes_per_stream);
  if (worst.load() == 2) return fail(LICOS_EINVAL, "rans_encode_host: CDF row out of range or empty");
  return LICOS_OK;
}

template <typename IdxT, typename SymT = int32_t>
static int decode_host(const uint8_t *in, const int64_t *byte_off, const IdxT *indexes, long sym_stride_b,
                       long sym_stride_i, int n, int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len,
                       const int32_t *offset, int rows, SymT *symbols, int32_t *status, int batch, int nthreads) {
  LICOS_REQUIRE(in && byte_off && cdf && cdf_len && offset && symbols && status, "rans_decode_host: NULL buffer");
  LICOS_REQUIRE(n >= 0 && batch >= 0 && rows > 0 && cdf_stride > 1, "rans_decode_host: bad shape");
  LICOS_REQUIRE(indexes || plane > 0, "rans_decode_host: plane must be positive without explicit indexes");
  for (int r = 0; r < rows; ++r)
    LICOS_REQUIRE(cdf_len[r] >= 2 && cdf_len[r] <= cdf_stride, "rans_decode_host: row %d has cdf length %d", r, cdf_len[r]);
  const Tables t{cdf, cdf_len, offset, cdf_stride, rows};
  DecLut lut;
  lut.build(t);
  std::atomic<int> worst{0};
  auto note = [&](int rc) {
    if (rc < 0) rc = 4;  // (above 3 = "a symbol outside 16 bits": an error wins over it)
    int prev = worst.load();
    while (rc > prev && !worst.compare_exchange_weak(prev, rc)) {}
  };
  const int K = lockstep_width(batch, nthreads);
  const int njobs = (batch + K - 1) / K;
  auto one = [&](int b) {
    const long nb = (long)(byte_off[b + 1] - byte_off[b]);
    if (nb < 0) return note(4);
    if constexpr (std::is_same<IdxT, int32_t>::value && std::is_same<SymT, int32_t>::value) {
      note(decode_stream(in + byte_off[b], nb, symbols + (size_t)b * sym_stride_b,
                         indexes ? indexes + (size_t)b * sym_stride_b : nullptr, sym_stride_i, n, plane, t, lut));
    } else {
      const uint8_t *dp[1] = {in + byte_off[b]};
      const long nbs[1] = {nb};
      SymT *sp[1] = {symbols + (size_t)b * sym_stride_b};
      const IdxT *ip[1] = {indexes ? indexes + (size_t)b * sym_stride_b : nullptr};
      int rc[1];
      decode_streams<1, IdxT, SymT>(dp, nbs, sp, ip, sym_stride_i, n, plane, t, lut, rc);
      note(rc[0]);
    }
  };
  auto run_group = [&](auto kc, int b0) {
    constexpr int KK = decltype(kc)::value;
    const uint8_t *dp[KK];
    long nb[KK];
    SymT *sp[KK];
    const IdxT *ip[KK];
    int rc[KK];
    for (int k = 0; k < KK; ++k) {
      nb[k] = (long)(byte_off[b0 + k + 1] - byte_off[b0 + k]);
      if (nb[k] < 0) {  // bad offsets: the single-stream path reports it
        for (int q = 0; q < KK; ++q) one(b0 + q);
        return;
      }
      dp[k] = in + byte_off[b0 + k];
      sp[k] = symbols + (size_t)(b0 + k) * sym_stride_b;
      ip[k] = indexes ? indexes + (size_t)(b0 + k) * sym_stride_b : nullptr;
    }
    decode_streams<KK, IdxT, SymT>(dp, nb, sp, ip, sym_stride_i, n, plane, t, lut, rc);
    for (int k = 0; k < KK; ++k) note(rc[k]);
  };
  Pool::instance().run(njobs, nthreads, [&](int j) {
    const int b0 = j * K, cnt = batch - b0 < K ? batch - b0 : K;
    if (cnt == 4) run_group(std::integral_constant<int, 4>{}, b0);
    else if (cnt == 2) run_group(std::integral_constant<int, 2>{}, b0);
    else
      for (int k = 0; k < cnt; ++k) one(b0 + k);
  });
  if (worst.load() == 4) return fail(LICOS_EINVAL, "rans_decode_host: CDF row out of range or bad stream offsets");
  status[0] = worst.load();  // 1: some stream ended before all its symbols were decoded; 3 (16-bit symbols): a value did not fit
  return LICOS_OK;
}

extern "C" {

int licos_rans_encode_host(const int32_t *symbols, const int32_t *indexes, long sym_stride_b, long sym_stride_i, int n,
                           int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset,
                           int rows, const void *enc_table, uint8_t *out, long cap_bytes_per_stream, int64_t *nbytes,
                           int batch, int nthreads) {
  return encode_host(symbols, indexes, sym_stride_b, sym_stride_i, n, plane, cdf, cdf_stride, cdf_len, offset, rows, enc_table, out,
                     cap_bytes_per_stream, nbytes, batch, nthreads, false);
}

int licos_rans_encode_host_packed(const int32_t *packed, long stride_b, int n, const int32_t *cdf, int cdf_stride,
                                  const int32_t *cdf_len, const int32_t *offset, int rows, const void *enc_table, uint8_t *out,
                                  long cap_bytes_per_stream, int64_t *nbytes, int batch, int nthreads) {
  return encode_host(packed, packed, stride_b, 1, n, 0, cdf, cdf_stride, cdf_len, offset, rows, enc_table, out, cap_bytes_per_stream,
                     nbytes, batch, nthreads, true);
}

int licos_rans_decode_host(const uint8_t *in, const int64_t *byte_off, const int32_t *indexes, long sym_stride_b,
                           long sym_stride_i, int n, int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len,
                           const int32_t *offset, int rows, int32_t *symbols, int32_t *status, int batch, int nthreads) {
  return decode_host<int32_t>(in, byte_off, indexes, sym_stride_b, sym_stride_i, n, plane, cdf, cdf_stride, cdf_len, offset, rows,
                              symbols, status, batch, nthreads);
}

int licos_rans_encode_host_sym16(const int16_t *symbols, long stride_b, int n, int plane, const int32_t *cdf, int cdf_stride,
                                 const int32_t *cdf_len, const int32_t *offset, int rows, const void *enc_table, uint8_t *out,
                                 long cap_bytes_per_stream, int64_t *nbytes, int batch, int nthreads) {
  LICOS_REQUIRE(plane > 0, "rans_encode_host_sym16: channel-plane rows only");
  return encode_host<int16_t>(symbols, nullptr, stride_b, 1, n, plane, cdf, cdf_stride, cdf_len, offset, rows, enc_table, out,
                              cap_bytes_per_stream, nbytes, batch, nthreads, false);
}

int licos_rans_decode_host_sym16(const uint8_t *in, const int64_t *byte_off, long stride_b, int n, int plane, const int32_t *cdf,
                                 int cdf_stride, const int32_t *cdf_len, const int32_t *offset, int rows, int16_t *symbols,
                                 int32_t *status, int batch, int nthreads) {
  LICOS_REQUIRE(plane > 0, "rans_decode_host_sym16: channel-plane rows only");
  return decode_host<int32_t, int16_t>(in, byte_off, static_cast<const int32_t *>(nullptr), stride_b, 1, n, plane, cdf, cdf_stride,
                                       cdf_len, offset, rows, symbols, status, batch, nthreads);
}

int licos_rans_decode_host_rows8(const uint8_t *in, const int64_t *byte_off, const uint8_t *rows8, long stride_b, int n,
                                 const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset, int rows,
                                 int32_t *symbols, int32_t *status, int batch, int nthreads) {
  LICOS_REQUIRE(rows8, "rans_decode_host_rows8: NULL rows");
  return decode_host<uint8_t>(in, byte_off, rows8, stride_b, 1, n, 0, cdf, cdf_stride, cdf_len, offset, rows, symbols, status, batch,
                              nthreads);
}

}  // extern "C"
