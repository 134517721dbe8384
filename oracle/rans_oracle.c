/*
 * CPU oracle: rANS coder + pmf->quantised-CDF, plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
 * algorithm restated here is the one CompressAI ships in
 *   compressai/cpp_exts/rans/rans_interface.cpp   (encode_with_indexes /
 *                                                  decode_with_indexes)
 *   compressai/cpp_exts/ops/ops.cpp               (pmf_to_quantized_cdf)
 *   third_party/ryg_rans/rans64.h                 (64-bit rANS primitives)
 * none of which is present under /root/reference (CompressAI is an un-pinned
 * pip dependency, /root/reference/environment.yml:27-29).  The reference
 * reaches this code at /root/reference/eval_utils.py:201 (net.compress) and
 * /root/reference/eval_script.py:72 (net.update()).  SURVEY.md section 8(a) row A7
 * and section 8(c) hold the restated definition this file follows.
 *
 * Conventions: 64-bit state, lower bound L = 2^31, 32-bit word renormalisation,
 * 16-bit probability precision, 4-bit bypass ("escape") coding.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RANS_L (1ull << 31)
#define PRECISION 16
#define BYPASS_BITS 4
#define BYPASS_MAX ((1 << BYPASS_BITS) - 1)

/* ---- pmf -> quantised cdf (ops.cpp: pmf_to_quantized_cdf) ---------------- */
/* returns 0 ok, -1 invalid pmf entry, -2 all-zero pmf, -3 nothing to steal */
int oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision,
                                uint32_t *cdf /* n+1 */) {
  for (int i = 0; i < n; ++i)
    if (!(pmf[i] >= 0.0f) || !isfinite(pmf[i])) return -1;
  cdf[0] = 0;
  for (int i = 0; i < n; ++i)
    /* float product, then C round(): half away from zero */
    cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision));
  uint32_t total = 0;
  for (int i = 0; i <= n; ++i) total += cdf[i];
  if (total == 0) return -2;
  for (int i = 0; i <= n; ++i)
    cdf[i] = (uint32_t)((((uint64_t)1 << precision) * cdf[i]) / total);
  for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
  cdf[n] = 1u << precision;
  for (int i = 0; i < n; ++i) {
    if (cdf[i] != cdf[i + 1]) continue;
    uint32_t best_freq = ~0u;
    int best = -1;
    for (int j = 0; j < n; ++j) {
      uint32_t f = cdf[j + 1] - cdf[j];
      if (f > 1 && f < best_freq) { best_freq = f; best = j; }
    }
    if (best < 0) return -3;
    if (best < i) {
      for (int j = best + 1; j <= i; ++j) cdf[j]--;
    } else {
      for (int j = i + 1; j <= best; ++j) cdf[j]++;
    }
  }
  return 0;
}

/* ---- encoder -------------------------------------------------------------- */
typedef struct { uint16_t start; uint16_t range; uint8_t bypass; } sym_t;

/* returns number of bytes written to `out` (>=0), -1 on overflow of out_cap,
 * -2 on bad symbol/index, -3 alloc failure */
long oracle_rans_encode(const int32_t *symbols, const int32_t *indexes, int n,
                        const int32_t *cdfs, int cdf_stride,
                        const int32_t *cdf_sizes, const int32_t *offsets,
                        uint8_t *out, long out_cap) {
  /* pass 1: the symbol list, in coding order (rans_interface.cpp pushes to a
   * vector, then pops from the back) */
  size_t cap = (size_t)n * 2 + 16, cnt = 0;
  sym_t *list = (sym_t *)malloc(cap * sizeof(sym_t));
  if (!list) return -3;
  for (int i = 0; i < n; ++i) {
    int c = indexes[i];
    const int32_t *cdf = cdfs + (size_t)c * cdf_stride;
    int32_t max_value = cdf_sizes[c] - 2;
    int32_t value = symbols[i] - offsets[c];
    uint32_t raw = 0;
    if (value < 0) { raw = (uint32_t)(-2 * value - 1); value = max_value; }
    else if (value >= max_value) { raw = (uint32_t)(2 * (value - max_value)); value = max_value; }
    if (max_value < 0) { free(list); return -2; }
    if (cnt + 24 > cap) {
      cap *= 2;
      sym_t *nl = (sym_t *)realloc(list, cap * sizeof(sym_t));
      if (!nl) { free(list); return -3; }
      list = nl;
    }
    list[cnt].start = (uint16_t)cdf[value];
    list[cnt].range = (uint16_t)(cdf[value + 1] - cdf[value]);
    list[cnt].bypass = 0;
    cnt++;
    if (value == max_value) {
      int32_t nb = 0;
      while (nb < 8 && (raw >> (nb * BYPASS_BITS)) != 0) ++nb; /* 32-bit raw: at most 8 nibbles */
      int32_t val = nb;
      while (val >= BYPASS_MAX) {
        list[cnt].start = BYPASS_MAX; list[cnt].range = 0; list[cnt].bypass = 1; cnt++;
        val -= BYPASS_MAX;
      }
      list[cnt].start = (uint16_t)val; list[cnt].range = 0; list[cnt].bypass = 1; cnt++;
      for (int32_t j = 0; j < nb; ++j) {
        list[cnt].start = (uint16_t)((raw >> (j * BYPASS_BITS)) & BYPASS_MAX);
        list[cnt].range = 0; list[cnt].bypass = 1; cnt++;
      }
    }
  }
  /* pass 2: code in reverse, words written backwards */
  size_t nwords = cnt + 2; /* every item emits at most one word; flush = 2 */
  uint32_t *buf = (uint32_t *)malloc(nwords * sizeof(uint32_t));
  if (!buf) { free(list); return -3; }
  uint32_t *ptr = buf + nwords;
  uint64_t x = RANS_L;
  while (cnt > 0) {
    sym_t s = list[--cnt];
    if (!s.bypass) {
      uint64_t x_max = ((RANS_L >> PRECISION) << 32) * (uint64_t)s.range;
      if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
      x = ((x / s.range) << PRECISION) + (x % s.range) + s.start;
    } else {
      uint32_t freq = 1u << (PRECISION - BYPASS_BITS);
      uint64_t x_max = ((RANS_L >> PRECISION) << 32) * (uint64_t)freq;
      if (x >= x_max) { *--ptr = (uint32_t)x; x >>= 32; }
      x = (x << BYPASS_BITS) | s.start;
    }
  }
  ptr -= 2;
  ptr[0] = (uint32_t)x;
  ptr[1] = (uint32_t)(x >> 32);
  long nbytes = (long)((buf + nwords) - ptr) * 4;
  long ret = nbytes;
  if (nbytes > out_cap) ret = -1;
  else memcpy(out, ptr, (size_t)nbytes); /* little-endian host */
  free(buf);
  free(list);
  return ret;
}

/* ---- decoder -------------------------------------------------------------- */
static inline uint32_t rd32(const uint8_t *p) {
  uint32_t v; memcpy(&v, p, 4); return v;
}

/* returns 0 ok, -1 if the stream is read past its end */
int oracle_rans_decode(const uint8_t *in, long nbytes, const int32_t *indexes,
                       int n, const int32_t *cdfs, int cdf_stride,
                       const int32_t *cdf_sizes, const int32_t *offsets,
                       int32_t *out) {
  long pos = 0;
  int over = 0;
#define NEXT_WORD() ((pos + 4 <= nbytes) ? (pos += 4, rd32(in + pos - 4)) : (over = 1, 0u))
  uint64_t x = (uint64_t)NEXT_WORD();
  x |= (uint64_t)NEXT_WORD() << 32;
  for (int i = 0; i < n; ++i) {
    int c = indexes[i];
    const int32_t *cdf = cdfs + (size_t)c * cdf_stride;
    int32_t max_value = cdf_sizes[c] - 2;
    uint32_t cf = (uint32_t)(x & ((1u << PRECISION) - 1));
    /* std::find_if over cdf[0 .. cdf_size): first entry > cf, minus one */
    int32_t s = 0;
    while (s + 1 < cdf_sizes[c] && (uint32_t)cdf[s + 1] <= cf) ++s;
    uint32_t start = (uint32_t)cdf[s], range = (uint32_t)(cdf[s + 1] - cdf[s]);
    x = (uint64_t)range * (x >> PRECISION) + cf - start;
    if (x < RANS_L) x = (x << 32) | NEXT_WORD();
    int32_t value = s;
    if (value == max_value) {
      int32_t val, nb;
      val = (int32_t)(x & BYPASS_MAX); x >>= BYPASS_BITS;
      if (x < RANS_L) x = (x << 32) | NEXT_WORD();
      nb = val;
      while (val == BYPASS_MAX) {
        val = (int32_t)(x & BYPASS_MAX); x >>= BYPASS_BITS;
        if (x < RANS_L) x = (x << 32) | NEXT_WORD();
        nb += val;
      }
      uint32_t raw = 0;
      for (int32_t j = 0; j < nb; ++j) {
        val = (int32_t)(x & BYPASS_MAX); x >>= BYPASS_BITS;
        if (x < RANS_L) x = (x << 32) | NEXT_WORD();
        if (j < 8) raw |= (uint32_t)val << (j * BYPASS_BITS);
      }
      value = (int32_t)(raw >> 1);
      if (raw & 1) value = -value - 1;
      else value += max_value;
    }
    out[i] = value + offsets[c];
  }
#undef NEXT_WORD
  return over ? -1 : 0;
}
