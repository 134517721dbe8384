// Decoder image of a family of quantised CDFs (the GaussianConditional's scale table: 64 rows of 3 .. ~3100 symbols).
//
// What a decode step needs from the tables is "which symbol holds the 16-bit value cf, and its cdf bounds" - on the
// serial state -> state chain of a lane that owns a stream, where every dependent memory round trip is paid by the
// whole wave.  The image turns that into ONE 8-byte read in the common case:
//
//   meta[row]   (16 B)  { first record, shift, cdf16 base | max_value << 16, offset }
//   rec[...]    ( 8 B)  per (row, bucket = cf >> shift): the two ADJACENT symbols s, s+1 that cover the most of the
//                       bucket: { cdf[s] | (cdf[s+1]-1) << 16, (cdf[s+2]-1) | s << 16 }  ("-1" keeps 65536 in 16 bits)
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

struct ImageMeta { uint32_t rec_base, shift, cdf_base_max, offset; };  // cdf_base_max = cdf16 base | max_value << 16

struct ImageHeader {  // 32 bytes at the start of the blob; all offsets in bytes from the start, 16-byte aligned
  uint32_t magic, rows, n_rec, n_cdf, off_meta, off_rec, off_cdf, total_bytes;
};
constexpr uint32_t IMAGE_MAGIC = 0x4C494D47u;  // "LIMG"

// Symbol s and its [lo, hi) cdf interval for value cf in row `m`.  `fallback` reports whether the record's pair missed.
__host__ __device__ inline void image_lookup(const ImageMeta m, const uint2 *rec, const uint16_t *cdf16, uint32_t cf, int &s,
                                             uint32_t &lo, uint32_t &hi_m1, bool &fallback) {
  const uint2 r = rec[m.rec_base + (cf >> m.shift)];
  const uint32_t c0 = r.x & 0xFFFFu, c1m = r.x >> 16, c2m = r.y & 0xFFFFu;
  const bool adv = cf > c1m;
  fallback = cf < c0 || cf > c2m;
  lo = adv ? c1m + 1u : c0;
  hi_m1 = adv ? c2m : c1m;
  s = (int)(r.y >> 16) + (adv ? 1 : 0);
  if (fallback) {
    // the pair missed: the symbol lies between the neighbouring buckets' pairs (a record's s is never below the symbol
    // holding its bucket's first value, nor above the one holding its last) - a search over a few entries of the row
    const uint16_t *row = cdf16 + (m.cdf_base_max & 0xFFFFu);
    const int max_value = (int)(m.cdf_base_max >> 16);
    const uint2 *rr = rec + m.rec_base;
    const uint32_t j = cf >> m.shift;
    int a, b;  // largest s in [a, b] with row[s] <= cf
    if (cf < c0) {
      a = j > 0 ? (int)(rr[j - 1].y >> 16) : 0;
      b = (int)(r.y >> 16) - 1;
    } else {
      a = (int)(r.y >> 16) + 2;
      b = (j + 1 < (65536u >> m.shift)) ? (int)(rr[j + 1].y >> 16) : max_value;
    }
    while (a < b) {
      const int mid = (a + b + 1) >> 1;
      if ((uint32_t)row[mid] <= cf) a = mid; else b = mid - 1;
    }
    s = a;
    lo = row[a];
    hi_m1 = (a == max_value ? 65536u : (uint32_t)row[a + 1]) - 1u;
  }
}

}  // namespace licos
