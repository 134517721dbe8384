// Decoder image of a family of quantised CDFs (the GaussianConditional's scale table: 64 rows of 3 .. ~3100 symbols).
//
// What a decode step needs from the tables is "which symbol holds the 16-bit value cf, and its cdf bounds" - on the
// serial state -> state chain of a lane that owns a stream, where every dependent memory round trip is paid by the
// whole wave.  The image turns that into ONE 8-byte read in the common case:
//
//   meta[row]   (16 B)  { byte offset of the row's first record << 5 | shift, offset, cdf16 base | max_value << 16,
//                         first record }
//   rec[...]    ( 8 B)  per (row, bucket = cf >> shift): the two ADJACENT symbols s, s+1 that cover the most of the
//                       bucket, as { cdf[s] | freq(s) << 16, freq(s+1) | s << 16 }.  With d = cf - cdf[s] (mod 2^32):
//                       d < freq(s) -> s; else d < freq(s) + freq(s+1) -> s+1; else the pair missed.  The escape
//                       symbol (the row's last) is never part of a pair: an out-of-range value always takes the miss
//                       path, which is where its bypass nibbles are read - one rare branch per symbol, not two.
//   cdf16[...]  ( 2 B)  every row's symbol starts cdf[0 .. len-2], for the rare value outside a record's pair
//                       (a symbol of a few counts at the edge of a row: hit about as often as such a symbol occurs)
//
// Buckets per row are powers of two chosen by the builder (rans_gc.hip, host side) under a byte budget so that the
// probability mass NOT covered by the records is as small as possible.  `image_lookup` below is the one definition
// of the search, shared by the decode kernel (tables in LDS) and by the host-side self check (licos_rans_image_lookup).
#pragma once
#include <cstdint>

#ifndef __HIPCC__
#define __host__
#define __device__
#endif

namespace licos {

struct ImageMeta { uint32_t pack; int32_t offset; uint32_t cdf_base_max, rec_base; };  // pack = record bytes << 5 | shift
constexpr int IMAGE_PACK_SHIFT = 5;

struct ImageHeader {  // 32 bytes at the start of the blob; all offsets in bytes from the start, 16-byte aligned
  uint32_t magic, rows, n_rec, n_cdf, off_meta, off_rec, off_cdf, total_bytes;
};
constexpr uint32_t IMAGE_MAGIC = 0x4C494D47u;  // "LIMG"

// The pair part of the search: straight-line selects on the record of cf's bucket.  `miss` = cf lies outside the pair.
__host__ __device__ inline void image_pair(const uint2 r, uint32_t cf, int &s, uint32_t &off, uint32_t &freq, bool &miss) {
  const uint32_t c0 = r.x & 0xFFFFu, f0 = r.x >> 16, f1 = r.y & 0xFFFFu;
  const uint32_t d = cf - c0;  // wraps to a huge value when cf < c0: caught by `miss`
  const bool adv = d >= f0;
  miss = d >= f0 + f1;
  freq = adv ? f1 : f0;
  off = adv ? d - f0 : d;
  s = (int)(r.y >> 16) + (adv ? 1 : 0);
}

// The rare part: the symbol lies between the neighbouring buckets' pairs (a record's s is never below the symbol
// holding its bucket's first value, nor above the one holding its last) - a search over a few entries of the row.
__host__ __device__ inline void image_search(const ImageMeta m, const uint2 *rec, const uint16_t *cdf16, const uint2 r, uint32_t cf,
                                             int &s, uint32_t &off, uint32_t &freq) {
  const uint16_t *row = cdf16 + (m.cdf_base_max & 0xFFFFu);
  const int max_value = (int)(m.cdf_base_max >> 16);
  const uint2 *rr = rec + m.rec_base;
  const uint32_t shift = m.pack & ((1u << IMAGE_PACK_SHIFT) - 1u);
  const uint32_t j = cf >> shift;
  int a, b;  // largest s in [a, b] with row[s] <= cf
  if (cf < (r.x & 0xFFFFu)) {
    a = j > 0 ? (int)(rr[j - 1].y >> 16) : 0;
    b = (int)(r.y >> 16) - 1;
  } else {
    a = (int)(r.y >> 16) + 1;
    b = (j + 1 < (65536u >> shift)) ? (int)(rr[j + 1].y >> 16) : max_value;
  }
  a = a > max_value ? max_value : a;
  b = b < a ? a : b;
  while (a < b) {
    const int mid = (a + b + 1) >> 1;
    if ((uint32_t)row[mid] <= cf) a = mid; else b = mid - 1;
  }
  s = a;
  const uint32_t lo = row[a];
  off = cf - lo;
  freq = (a == max_value ? 65536u : (uint32_t)row[a + 1]) - lo;
}

// Symbol s, cf - cdf[s] and freq(s) for value cf in row `m` (host-side self check; the kernel calls the two parts).
__host__ __device__ inline void image_lookup(const ImageMeta m, const uint2 *rec, const uint16_t *cdf16, uint32_t cf, int &s,
                                             uint32_t &off, uint32_t &freq, bool &miss) {
  const uint32_t shift = m.pack & ((1u << IMAGE_PACK_SHIFT) - 1u);
  const uint2 r = rec[m.rec_base + (cf >> shift)];
  image_pair(r, cf, s, off, freq, miss);
  if (miss) image_search(m, rec, cdf16, r, cf, s, off, freq);
}

}  // namespace licos
