// Batched rANS coder: one stream per image, one lane per stream.
//
// Stream format (bit-exact with CompressAI's RansEncoder/RansDecoder):
// 64-bit state, L = 2^31, 32-bit little-endian words, 16-bit probabilities, 4-bit bypass nibbles
// for values outside the tabulated range.  The recurrence is strictly sequential inside a
// stream, so parallelism is across streams only: lane b of the grid owns stream b and walks its
// symbols in reverse (the encoder emits words back to front).  Words are staged in an
// interleaved scratch [word][stream] so that lanes advancing in lock-step write coalesced.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>

namespace licos {

constexpr uint64_t RANS_L = 1ull << 31;

struct EncRec { uint64_t rcp; uint32_t bias; uint16_t freq; uint16_t shift; };

struct WordSink {
  uint32_t *words;
  int B, b, wp;
  bool overflow;
  __device__ inline void put(uint32_t w) {
    if (wp > 0) { --wp; words[(size_t)wp * B + b] = w; }
    else overflow = true;
  }
  // Branch-free form for the per-symbol renormalisation: the word is ALWAYS stored to the next free slot (a slot
  // below wp is scratch until a real emission claims it, which overwrites whatever sits there) and the slot is
  // claimed only when `emit` is set.
  __device__ inline void put_if(bool emit, uint32_t w) {
    const int slot = wp > 0 ? wp - 1 : 0;
    words[(size_t)slot * B + b] = w;
    overflow = overflow || (emit && wp <= 0);
    wp = emit ? slot : wp;
  }
};

__device__ inline void put_bits4(uint64_t &x, WordSink &sink, uint32_t val) {
  // Rans64EncPutBits with nbits = 4: freq = 2^12, x_max = 2^59
  if (x >= (1ull << 59)) { sink.put((uint32_t)x); x >>= 32; }
  x = (x << 4) | val;
}

__global__ __launch_bounds__(64) void rans_encode_kernel(const int32_t *__restrict__ symbols,
                                                         const int32_t *__restrict__ indexes, long ssb, long ssi, int n,
                                                         int plane, int cdf_stride, const int32_t *__restrict__ cdf_len,
                                                         const int32_t *__restrict__ offset,
                                                         const EncRec *__restrict__ table, uint32_t *__restrict__ words,
                                                         int cap_words, int32_t *__restrict__ nwords,
                                                         int32_t *__restrict__ status, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  WordSink sink{words, B, b, cap_words, false};
  uint64_t x = RANS_L;
  const int32_t *sp = symbols + (size_t)b * ssb;
  const int32_t *ip = indexes ? indexes + (size_t)b * ssb : nullptr;
  for (int i = n - 1; i >= 0; --i) {
    const int32_t s = sp[(size_t)i * ssi];
    const int c = ip ? ip[(size_t)i * ssi] : i / plane;
    const int32_t max_value = cdf_len[c] - 2;
    int32_t value = s - offset[c];
    if (value < 0 || value >= max_value) {
      const uint32_t raw = (value < 0) ? (uint32_t)(-2 * value - 1) : (uint32_t)(2 * (value - max_value));
      value = max_value;
      int nb = 0;
      while (nb < 8 && (raw >> (nb * 4)) != 0) ++nb;
      // coding order is [symbol, count nibbles (15,15,..,rem), raw nibbles low->high]; emit reversed
      for (int j = nb - 1; j >= 0; --j) put_bits4(x, sink, (raw >> (j * 4)) & 15u);
      const int k15 = nb / 15, rem = nb - 15 * k15;
      put_bits4(x, sink, (uint32_t)rem);
      for (int t = 0; t < k15; ++t) put_bits4(x, sink, 15u);
    }
    const EncRec rec = table[(size_t)c * cdf_stride + value];
    const uint32_t freq = rec.freq ? rec.freq : 65536u;
    if (x >= ((uint64_t)freq << 47)) { sink.put((uint32_t)x); x >>= 32; }
    const uint64_t q = __umul64hi(x, rec.rcp) >> rec.shift;
    x = x + rec.bias + q * (uint64_t)(65536u - freq);
  }
  sink.put((uint32_t)(x >> 32));
  sink.put((uint32_t)x);
  nwords[b] = cap_words - sink.wp;
  if (sink.overflow) atomicOr(status, 1);
}

__global__ __launch_bounds__(256) void rans_compact_kernel(const uint32_t *__restrict__ words, int cap_words,
                                                           const int32_t *__restrict__ nwords,
                                                           const int64_t *__restrict__ byte_off,
                                                           uint32_t *__restrict__ out, int B) {
  const int b = blockIdx.y;
  const int nw = nwords[b];
  uint32_t *dst = out + byte_off[b] / 4;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nw; k += gridDim.x * blockDim.x)
    dst[k] = words[(size_t)(cap_words - nw + k) * B + b];
}

struct WordSource {
  const uint32_t *p;
  int nw, pos;
  bool over;
  __device__ inline uint32_t next() {
    if (pos < nw) return p[pos++];
    over = true;
    return 0u;
  }
};

__device__ inline uint32_t get_bits4(uint64_t &x, WordSource &src) {
  const uint32_t val = (uint32_t)(x & 15u);
  x >>= 4;
  if (x < RANS_L) x = (x << 32) | src.next();
  return val;
}

__global__ __launch_bounds__(64) void rans_decode_kernel(const uint8_t *__restrict__ in,
                                                         const int64_t *__restrict__ byte_off,
                                                         const int32_t *__restrict__ indexes, long ssb, long ssi, int n,
                                                         int plane, const int32_t *__restrict__ cdf, int cdf_stride,
                                                         const int32_t *__restrict__ cdf_len,
                                                         const int32_t *__restrict__ offset, int32_t *__restrict__ symbols,
                                                         int32_t *__restrict__ status, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  WordSource src{reinterpret_cast<const uint32_t *>(in + byte_off[b]), (int)((byte_off[b + 1] - byte_off[b]) / 4), 0, false};
  uint64_t x = (uint64_t)src.next();
  x |= (uint64_t)src.next() << 32;
  int32_t *sp = symbols + (size_t)b * ssb;
  const int32_t *ip = indexes ? indexes + (size_t)b * ssb : nullptr;
  for (int i = 0; i < n; ++i) {
    const int c = ip ? ip[(size_t)i * ssi] : i / plane;
    const int32_t *row = cdf + (size_t)c * cdf_stride;
    const int len = cdf_len[c];
    const int32_t max_value = len - 2;
    const uint32_t cf = (uint32_t)(x & 0xFFFFu);
    int lo = 0, hi = len - 1;  // row[lo] <= cf < row[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((uint32_t)row[mid] <= cf) lo = mid; else hi = mid;
    }
    const uint32_t start = (uint32_t)row[lo], range = (uint32_t)row[lo + 1] - start;
    x = (uint64_t)range * (x >> 16) + cf - start;
    if (x < RANS_L) x = (x << 32) | src.next();
    int32_t value = lo;
    if (value == max_value) {
      uint32_t val = get_bits4(x, src);
      int nb = (int)val;
      while (val == 15u && nb < 64) { val = get_bits4(x, src); nb += (int)val; }
      uint32_t raw = 0;
      for (int j = 0; j < nb; ++j) {
        const uint32_t nib = get_bits4(x, src);
        if (j < 8) raw |= nib << (j * 4);
      }
      value = (int32_t)(raw >> 1);
      value = (raw & 1u) ? -value - 1 : value + max_value;
    }
    sp[(size_t)i * ssi] = value + offset[c];
  }
  if (src.over) atomicOr(status, 1);
}


// ---------------------------------------------------------------------------------------------------
// Plane-indexed fast path (indexes == NULL: the CDF row of position i is i / plane, i.e. the
// EntropyBottleneck case).  One wave = 64 streams walking the same (channel, position) in lock-step,
// so per channel the wave stages that row's table in LDS once and every per-symbol lookup is an LDS
// read instead of a dependent global load; symbol loads are batched ahead of the serial chain, and
// the decoder keeps the next two stream words in registers so renormalisation never waits on HBM.
constexpr int SYM_BATCH = 8;

__global__ __launch_bounds__(256) void rans_encode_plane_kernel(const int32_t *__restrict__ symbols, long ssb, long ssi,
                                                               int C, int plane, int cdf_stride,
                                                               const int32_t *__restrict__ cdf_len,
                                                               const int32_t *__restrict__ offset,
                                                               const EncRec *__restrict__ table,
                                                               uint32_t *__restrict__ words, int cap_words,
                                                               int32_t *__restrict__ nwords, int32_t *__restrict__ status,
                                                               int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  EncRec *s_tab = reinterpret_cast<EncRec *>(smem_raw);
  // a workgroup is blockDim.x / 64 independent waves that share the staged table (see coder_waves() below)
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  const int b_raw = (blockIdx.x * (nthr >> 6) + (tid >> 6)) * 64 + lane;
  const bool live = b_raw < B;
  const int b = live ? b_raw : B - 1;  // idle lanes shadow the last stream (no stores)
  WordSink sink{words, B, b, cap_words, false};
  uint64_t x = RANS_L;
  const int32_t *sp = symbols + (size_t)b * ssb;
  // Stream-major symbols (ssi == 1: licos_conv5x5s2_f16_symbols writes them so): a lane's symbols are one contiguous run
  // walked backwards, channel boundaries included, so the requests run ENC_AHEAD batches (two 16-byte loads each) ahead of
  // the coding - a wave's load touches 64 different cache lines, a round trip of microseconds that one batch of coding
  // (1.2 us) does not cover: with one batch of look-ahead the launch took 15.9 ms instead of 7.1 (round 5, first form).
  constexpr int ENC_AHEAD = 4;
  const bool vec_ok = ssi == 1 && (plane % (SYM_BATCH * ENC_AHEAD)) == 0 && (ssb & 3) == 0 && (reinterpret_cast<uintptr_t>(symbols) & 15) == 0;  // (uniform)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  long g_next = 0;  // next batch to request: batch g is positions n - 8 (g + 1) .. n - 8 g - 1 of the stream
  const long n_total = (long)C * plane;
  // The requests are inline assembly with a counted wait in front of their use: left to the compiler, the loop-carried
  // ring makes it wait for vmcnt(0) once per four batches - the 32 unconditional word stores of those batches included, a
  // full store round trip per 32 symbols.  Behind slot j's request come three more batches of 2 loads + 8 stores before
  // the slot is used again (the slot is re-requested right behind its own batch): s_waitcnt vmcnt(30) covers exactly the
  // request (escapes only add stores).  Past the front of the stream the address is clamped: every request is issued,
  // whatever its position.
  //
  // The ring lives in NAMED registers, v[200:231], that no C++ value ever occupies: a request only clobbers them, and the
  // statement that waits is the one that DEFINES the slot's values (physical-register output constraints).  The first form
  // of this ring (round 5, commit "plane encoder's requests as counted asm loads") passed the slot through "+v" operands of
  // the wait; the register allocator is free to give such an operand another register and COPY the ring's register into it
  // in front of the statement - in front of the wait, that is: a copy of data still in flight, right whenever the load
  // happened to be back already and wrong (56 streams of 4096 in one run of tools/eb_coder_bench.py) whenever it was not.
  // Nothing the compiler can see holds a value that is still on its way now; tests/test_host.py checks in the shipped
  // library's disassembly that these registers appear in the ring's own statements only.
#define LICOS_ENC_RING_REQUEST(LO, HI)                                                                                            \
  asm volatile("global_load_dwordx4 v[" #LO ":" #LO "+3], %0, off\n\tglobal_load_dwordx4 v[" #HI ":" #HI "+3], %0, off offset:16" \
               :: "v"(q) : "memory", LICOS_ENC_RING_CLOBBER)
#define LICOS_ENC_RING_CLOBBER                                                                                                      \
  "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", \
      "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231"
  auto request = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    long pos = n_total - (long)SYM_BATCH * (g_next + 1);
    pos = pos < 0 ? 0 : pos;
    const i32x4 *q = reinterpret_cast<const i32x4 *>(sp + pos);
    // (each statement names its own slot and declares the whole ring clobbered: nothing of the compiler's lives there)
    if constexpr (j == 0) LICOS_ENC_RING_REQUEST(200, 204);
    if constexpr (j == 1) LICOS_ENC_RING_REQUEST(208, 212);
    if constexpr (j == 2) LICOS_ENC_RING_REQUEST(216, 220);
    if constexpr (j == 3) LICOS_ENC_RING_REQUEST(224, 228);
    ++g_next;
  };
  auto landed = [&](auto jc, i32x4 &lo, i32x4 &hi) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j == 0) asm volatile("s_waitcnt vmcnt(30)" : "={v[200:203]}"(lo), "={v[204:207]}"(hi)::"memory");
    if constexpr (j == 1) asm volatile("s_waitcnt vmcnt(30)" : "={v[208:211]}"(lo), "={v[212:215]}"(hi)::"memory");
    if constexpr (j == 2) asm volatile("s_waitcnt vmcnt(30)" : "={v[216:219]}"(lo), "={v[220:223]}"(hi)::"memory");
    if constexpr (j == 3) asm volatile("s_waitcnt vmcnt(30)" : "={v[224:227]}"(lo), "={v[228:231]}"(hi)::"memory");
  };
  if (vec_ok && live) {
    static_assert(ENC_AHEAD == 4, "four named slots");
    static_for<ENC_AHEAD>([&](auto jc) { request(jc); });
    // the first pass over the ring has fewer operations behind its requests than the loop's counted wait assumes: they land here
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  for (int c = C - 1; c >= 0; --c) {
    const int len = cdf_len[c];
    const int32_t max_value = len - 2;
    const int32_t off = offset[c];
    __syncthreads();
    for (int e = tid; e < len - 1; e += nthr) s_tab[e] = table[(size_t)c * cdf_stride + e];
    __syncthreads();
    if (!live) continue;  // idle lanes (last block only) sit out the coding loop: no per-symbol predication for them
    // symbols are fetched one batch ahead, so their load latency hides under the coding of the current batch
    int32_t sv_next[SYM_BATCH];
    auto fetch = [&](int p1, int32_t (&dst)[SYM_BATCH]) {
#pragma unroll
      for (int k = 0; k < SYM_BATCH; ++k)
        dst[k] = (p1 - 1 - k >= 0) ? sp[(size_t)((size_t)c * plane + (p1 - 1 - k)) * ssi] : 0;
    };
    // one batch of nb <= SYM_BATCH symbols (FULL: nb == SYM_BATCH, no per-symbol bound checks)
    auto code_batch = [&](auto full_c, const int32_t (&sv)[SYM_BATCH], int nb) {
      constexpr bool FULL = decltype(full_c)::value;
      // table records for the whole batch first: they do not depend on the coder state, so the LDS
      // latency stays off the serial x -> x chain below
      EncRec rec[SYM_BATCH];
#pragma unroll
      for (int k = 0; k < SYM_BATCH; ++k) {
        if (FULL || k < nb) {
          int32_t value = sv[k] - off;
          value = (value < 0 || value >= max_value) ? max_value : value;
          rec[k] = s_tab[value];
        }
      }
      auto code_symbol = [&](int k) {
        // straight-line renormalise + encode: x >= freq << 47 compares the high words (the low 47 bits of the bound are 0)
        const uint32_t freq = rec[k].freq ? rec[k].freq : 65536u;
        const bool emit = (uint32_t)(x >> 32) >= (freq << 15);
        sink.put_if(emit, (uint32_t)x);
        x = emit ? (x >> 32) : x;
        const uint64_t q = __umul64hi(x, rec[k].rcp) >> rec[k].shift;
        x = x + rec[k].bias + q * (uint64_t)(65536u - freq);
      };
      // whether any symbol of the batch is out of its channel's range is known before the chain starts: a batch without
      // escapes - nearly all of them - is ONE basic block of SYM_BATCH chained symbols, so the scheduler can fill the
      // 8-cycle bubbles between dependent instructions with the neighbours' bookkeeping (a branch per symbol ends the block)
      bool any_escape = false;
#pragma unroll
      for (int k = 0; k < SYM_BATCH; ++k) {
        const int32_t value = sv[k] - off;
        any_escape = any_escape || ((FULL || k < nb) && (value < 0 || value >= max_value));
      }
      if (FULL && __builtin_expect(!__any(any_escape), 1)) {
#pragma unroll
        for (int k = 0; k < SYM_BATCH; ++k) code_symbol(k);
        return;
      }
#pragma unroll
      for (int k = 0; k < SYM_BATCH; ++k) {
        if (!FULL && k >= nb) break;
        const int32_t value = sv[k] - off;
        if (value < 0 || value >= max_value) {  // rare: this stream codes an out-of-range value
          const uint32_t raw = (value < 0) ? (uint32_t)(-2 * value - 1) : (uint32_t)(2 * (value - max_value));
          int nbyp = 0;
          while (nbyp < 8 && (raw >> (nbyp * 4)) != 0) ++nbyp;
          for (int j = nbyp - 1; j >= 0; --j) put_bits4(x, sink, (raw >> (j * 4)) & 15u);
          put_bits4(x, sink, (uint32_t)nbyp);  // nbyp <= 8 < 15: a single count nibble
        }
        code_symbol(k);
      }
    };
    if (vec_ok) {
      static_assert(SYM_BATCH == 8, "two int4 per batch");
      for (int jj = 0; jj < plane / SYM_BATCH; jj += ENC_AHEAD) {
        static_for<ENC_AHEAD>([&](auto jc) {
          i32x4 lo, hi;
          landed(jc, lo, hi);
          const int32_t sv[SYM_BATCH] = {hi.w, hi.z, hi.y, hi.x, lo.w, lo.z, lo.y, lo.x};
          code_batch(std::true_type{}, sv, SYM_BATCH);
          request(jc);  // (behind the batch's coding, three batches ahead of the slot's next use)
        });
      }
      continue;
    }
    fetch(plane, sv_next);
    int p1 = plane;
    for (; p1 >= SYM_BATCH; p1 -= SYM_BATCH) {
      int32_t sv[SYM_BATCH];
#pragma unroll
      for (int k = 0; k < SYM_BATCH; ++k) sv[k] = sv_next[k];
      if (p1 - SYM_BATCH > 0) fetch(p1 - SYM_BATCH, sv_next);
      code_batch(std::true_type{}, sv, SYM_BATCH);
    }
    if (p1 > 0) code_batch(std::false_type{}, sv_next, p1);
  }
  if (vec_ok && live) {
    // the requests that ran past the front of the stream are still in flight: they land before the wave goes on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", LICOS_ENC_RING_CLOBBER);
  }
  if (live) {
    sink.put((uint32_t)(x >> 32));
    sink.put((uint32_t)x);
    nwords[b] = cap_words - sink.wp;
    if (sink.overflow) atomicOr(status, 1);
  }
}

// Per-lane ring of the next stream words in LDS (RING slots, [slot][lane]).  The serial loop itself issues
// no global memory operation: when ANY lane runs low the whole wave tops its rings up together (all loads
// issued before the first is consumed: one memory round trip per refill, and refills are rare - a stream
// averages well under one word per symbol), and decoded symbols leave through a 16-deep LDS buffer.
constexpr int RING = 64, RING_LOW = 24, SYM_BUF = 16, LUT_BITS = 10;

struct RingSource {
  const uint32_t *p;
  uint32_t *ring;  // LDS: word (i & (RING-1)) of this lane at ring[(i & (RING-1)) * 64]
  int nw, rd, filled, last;
  bool over;
  // `fallback`: any readable device word, addressed instead of the stream when the stream is empty
  __device__ __forceinline__ void init(const uint32_t *ptr, int n, uint32_t *lane_ring, const uint32_t *fallback) {
    p = n > 0 ? ptr : fallback; nw = n; last = n > 0 ? n - 1 : 0; ring = lane_ring; rd = 0; over = false; filled = 0;
    top_up();
    top_up();
  }
  // up to 16 more words for every lane that has room (uniform control flow, per-lane predication).  The loads are
  // issued as asm with their own wait: loads the compiler knows about make it put a conservative s_waitcnt vmcnt(0)
  // into every iteration of the symbol loop (the refill is a rarely taken branch of it), and that wait also waits
  // for the symbol stores of the last flush - a memory round trip per SYM_BUF symbols on the serial chain.
  __device__ __forceinline__ void top_up() {
    uint32_t w[16];
    const bool room = filled - rd <= RING - 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = filled + j;
      const uint32_t *a = p + (i < last ? i : last);  // past the end: re-read the last word, zeroed below
      asm volatile("global_load_dword %0, %1, off" : "=v"(w[j]) : "v"(a));
    }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
                   "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
    if (room) {
#pragma unroll
      for (int j = 0; j < 16; ++j) ring[((filled + j) & (RING - 1)) * 64] = (filled + j < nw) ? w[j] : 0u;
      filled += 16;
    }
  }
  __device__ __forceinline__ void refill_if_low() {
    if (__builtin_expect(__any(filled - rd <= RING_LOW), 0)) top_up();
  }
  // branch-free pair for the common renormalisation: peek() the next word early, advance(true) if it was used.
  // The ring read carries no guard (a guarded read becomes a branch with its own s_waitcnt, which serialises this LDS
  // round trip with the table lookup's): words past the end of the stream are stored as zeros by top_up(), and the ring
  // checks keep rd < filled (a block of SYM_BUF symbols consumes at most SYM_BUF <= RING_LOW words between checks).
  __device__ __forceinline__ uint32_t peek() const { return ring[(rd & (RING - 1)) * 64]; }
  __device__ __forceinline__ void advance(bool used) {
    over = over || (used && rd >= nw);
    rd += used ? 1 : 0;
  }
  __device__ __forceinline__ uint32_t next() {
    if (rd >= nw) over = true;
    const uint32_t w = ring[(rd & (RING - 1)) * 64];
    ++rd;
    return w;
  }
};

__device__ inline uint32_t get_bits4p(uint64_t &x, RingSource &src) {
  const uint32_t val = (uint32_t)(x & 15u);
  x >>= 4;
  if (x < RANS_L) x = (x << 32) | src.next();
  return val;
}

__global__ __launch_bounds__(256) void rans_decode_plane_kernel(const uint8_t *__restrict__ in,
                                                               const int64_t *__restrict__ byte_off, long ssb, long ssi,
                                                               int C, int plane, const int32_t *__restrict__ cdf,
                                                               int cdf_stride, const int32_t *__restrict__ cdf_len,
                                                               const int32_t *__restrict__ offset,
                                                               int32_t *__restrict__ symbols, int32_t *__restrict__ status,
                                                               int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // a workgroup is blockDim.x / 64 independent waves (own ring and symbol buffer) that share the channel's tables
  const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  uint32_t *s_ring = reinterpret_cast<uint32_t *>(smem_raw) + wave * (RING * 64);            // [waves][RING][64] stream-word rings
  int32_t *s_out = reinterpret_cast<int32_t *>(smem_raw) + nwave * (RING * 64) + wave * (SYM_BUF * 64);  // [waves][SYM_BUF][64]
  uint2 *s_lut = reinterpret_cast<uint2 *>(reinterpret_cast<uint32_t *>(smem_raw) + nwave * (RING + SYM_BUF) * 64);  // [1 << LUT_BITS]
  uint32_t *s_cdf = reinterpret_cast<uint32_t *>(s_lut + (1 << LUT_BITS));    // [cdf_stride]
  const int b_raw = (blockIdx.x * nwave + wave) * 64 + lane;
  const bool live = b_raw < B;
  const int b = live ? b_raw : B - 1;
  RingSource src;
  src.init(reinterpret_cast<const uint32_t *>(in + byte_off[b]), (int)((byte_off[b + 1] - byte_off[b]) / 4), s_ring + lane,
           reinterpret_cast<const uint32_t *>(byte_off));
  uint64_t x = (uint64_t)src.next();
  x |= (uint64_t)src.next() << 32;
  int32_t *sp = symbols + (size_t)b * ssb;
  for (int c = 0; c < C; ++c) {
    const int len = cdf_len[c] < 2 ? 2 : cdf_len[c];
    const int32_t max_value = len - 2;
    const int32_t off = offset[c];
    __syncthreads();
    for (int e = tid; e < len; e += nthr) s_cdf[e] = (uint32_t)cdf[(size_t)c * cdf_stride + e];
    __syncthreads();
    // One record per value of the top LUT_BITS bits of cf, holding everything the common case needs so that the
    // x -> x chain of a symbol carries ONE LDS round trip: with s = the largest symbol whose cdf[s] <= the bucket's
    // first value, the record is { cdf[s], cdf[s+1] - 1, cdf[s+2] - 1, s } (16 bits each; "- 1" keeps 65536 in range).
    // A bucket of 64 values that starts in symbol s reaches at most into s+1 unless a whole symbol of frequency < 64
    // lies inside it; that case (the far tails) is detected by cf > cdf[s+2] - 1 and walks the table.
    // Each thread fills a run of consecutive keys: one binary search, then a forward walk.
    const int KPL = (1 << LUT_BITS) / nthr;
    {
      const uint32_t key0 = (uint32_t)(tid * KPL) << (16 - LUT_BITS);
      int lo = 0, hi = len - 1;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_cdf[mid] <= key0) lo = mid; else hi = mid;
      }
      for (int k = 0; k < KPL; ++k) {
        const uint32_t key = (uint32_t)(tid * KPL + k) << (16 - LUT_BITS);
        while (lo + 1 < len - 1 && s_cdf[lo + 1] <= key) ++lo;
        const uint32_t c0 = s_cdf[lo], c1 = s_cdf[lo + 1], c2 = s_cdf[lo + 2 < len ? lo + 2 : len - 1];
        s_lut[tid * KPL + k] = make_uint2((c0 & 0xFFFFu) | ((c1 - 1u) << 16), ((c2 - 1u) & 0xFFFFu) | ((uint32_t)lo << 16));
      }
    }
    __syncthreads();
    if (!live) continue;  // idle lanes (last block only) sit out the decoding loop
    // blocks of SYM_BUF symbols: ring check | SYM_BUF symbols with no global-memory operation | flush.  A symbol takes at
    // most one word off the ring outside the (rare) bypass path, and RING_LOW >= SYM_BUF.
    static_assert(RING_LOW >= SYM_BUF, "a block may consume SYM_BUF words before the next ring check");
    for (int p0 = 0; p0 < plane; p0 += SYM_BUF) {
      src.refill_if_low();
      const int nb = plane - p0 < SYM_BUF ? plane - p0 : SYM_BUF;
      for (int k = 0; k < nb; ++k) {
        const uint32_t cf = (uint32_t)(x & 0xFFFFu);
        const uint2 rec = s_lut[cf >> (16 - LUT_BITS)];
        // the next stream word is read before it is known to be needed: its LDS latency runs beside the lookup's
        const uint32_t w_next = src.peek();
        const uint32_t c1m = rec.x >> 16, c2m = rec.y & 0xFFFFu;
        const bool adv = cf > c1m;
        uint32_t lo = adv ? c1m + 1u : (rec.x & 0xFFFFu);
        uint32_t him = adv ? c2m : c1m;  // cdf[s + 1] - 1 of the symbol taken
        int s = (int)(rec.y >> 16) + (adv ? 1 : 0);
        if (__builtin_expect(__any(cf > c2m), 0)) {  // uniform, per-lane walk: some stream sits in a bucket that holds three or more symbols
          if (cf > c2m) {
            s = (int)(rec.y >> 16) + 2;
            if (s > len - 2) s = len - 2;  // (malformed tables only)
            lo = s_cdf[s];
            uint32_t hi = s_cdf[s + 1];
            while (s < len - 2 && hi <= cf) {
              ++s;
              lo = hi;
              hi = s_cdf[s + 1];
            }
            him = hi - 1u;
          }
        }
        x = (uint64_t)(him + 1u - lo) * (x >> 16) + cf - lo;
        const bool need = x < RANS_L;
        x = need ? ((x << 32) | w_next) : x;
        src.advance(need);
        int32_t value = s;
        if (__builtin_expect(__any(value == max_value), 0)) {  // uniform and rare: some stream hit the escape symbol
          if (value == max_value) {
            uint32_t val = get_bits4p(x, src);
            int nbyp = (int)val;
            while (val == 15u && nbyp < 64) { val = get_bits4p(x, src); nbyp += (int)val; }
            uint32_t raw = 0;
            for (int j = 0; j < nbyp; ++j) {
              const uint32_t nib = get_bits4p(x, src);
              if (j < 8) raw |= nib << (j * 4);
            }
            value = (int32_t)(raw >> 1);
            value = (raw & 1u) ? -value - 1 : value + max_value;
          }
          src.refill_if_low();  // the escape path may have drained several words
        }
        s_out[k * 64 + lane] = value + off;
      }
      for (int k = 0; k < nb; ++k) sp[(size_t)((size_t)c * plane + p0 + k) * ssi] = s_out[k * 64 + lane];
    }
  }
  if (live && src.over) atomicOr(status, 1);
}


// ---- round 5: the same decoder, counted by INSTRUCTIONS ---------------------------------------------------------------
// A coder wave has its SIMD to itself and the hardware issues it ONE instruction per four cycles, scalar ones, waits and
// s_nop included (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost'): the kernel above spends ~60 instructions on a
// symbol - 240 of its ~275 cycles are issue slots, not the 132-cycle dependent chain round 4 priced.  This form spends
// 26:
//  * the table record is 16 bytes { cdf[s], cdf[s+1], cdf[s+2], s + offset } read by one ds_read_b128: nothing to
//    unpack, no SDWA forms (each of which cost an s_nop against the VCC hazard), the symbol's offset already added;
//  * the table sits at LDS address 0 (an immediate, no base to add) and every wave's word ring on a 16-KB boundary, so a
//    ring address is one v_and_or_b32 of the shifted read counter;
//  * ONE rare-case test per symbol: buckets that can yield the escape symbol carry cdf[s+2] = 0, which sends them down the
//    slow path that already exists for buckets of three or more symbols; that path (out of line) finishes the symbol
//    itself - walk, state update, bypass nibbles;
//  * freq * (x >> 16): one v_mad_u64_u32 for the low 32 bits of x >> 16 (carrying cf - lo as its addend) and a full-rate
//    v_mad_u32_u24 for the 15 bits above, instead of two 64-bit multiplies and the moves between them;
//  * reads past the end of a stream are detected ONCE, after the last symbol (the read counter only grows);
//  * four symbols per pass of the loop; with stream-major symbols (ssi == 1, what codec.factorized asks for) they leave
//    as one 16-byte store from registers - no LDS buffer, no flush loop.
constexpr int DEC_LUT_BYTES = 16 << LUT_BITS;
#define LICOS_OPAQUE(v) asm("" : "+v"(v))
#ifdef LICOS_STAMPS  // (diagnostic builds via tools/ab_build.sh, never the product: s_memtime cycles of wave 0 of every workgroup)
__device__ unsigned long long g_dec_stamps[4];
#endif
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4_t lds_cu32x4;
typedef int32_t i32x4_t __attribute__((ext_vector_type(4)));

template <int CW, bool SM>
__global__ __launch_bounds__(64 * CW) void rans_decode_plane4_kernel(const uint8_t *__restrict__ in, const int64_t *__restrict__ byte_off,
                                                                     long ssb, long ssi, int C, int plane,
                                                                     const int32_t *__restrict__ cdf, int cdf_stride,
                                                                     const int32_t *__restrict__ cdf_len,
                                                                     const int32_t *__restrict__ offset, int32_t *__restrict__ symbols,
                                                                     int32_t *__restrict__ status, int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int nthr = 64 * CW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // [table 16 KB][CW word rings, 16 KB each][CW symbol buffers, 4 KB each (strided output only)][the channel's cdf]
  u32x4_t *s_lut = reinterpret_cast<u32x4_t *>(smem_raw);
  uint32_t *s_ring = reinterpret_cast<uint32_t *>(smem_raw + DEC_LUT_BYTES) + wave * (RING * 64);
  int32_t *s_out = reinterpret_cast<int32_t *>(smem_raw + DEC_LUT_BYTES + CW * RING * 256) + wave * (SYM_BUF * 64);
  uint32_t *s_cdf = reinterpret_cast<uint32_t *>(smem_raw + DEC_LUT_BYTES + CW * (RING + SYM_BUF) * 256);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem_raw;
  if (lds0 != 0) {  // (the addresses below are absolute: never taken - the kernel has no static LDS in front of this block)
    if (tid == 0) atomicOr(status, 4);
    return;
  }
  const uint32_t ring_addr = DEC_LUT_BYTES + (uint32_t)wave * (RING * 256) + (uint32_t)lane * 4u;
  const int b_raw = (blockIdx.x * CW + wave) * 64 + lane;
  const bool live = b_raw < B;
  const int b = live ? b_raw : B - 1;
  RingSource src;
  src.init(reinterpret_cast<const uint32_t *>(in + byte_off[b]), (int)((byte_off[b + 1] - byte_off[b]) / 4), s_ring + lane,
           reinterpret_cast<const uint32_t *>(byte_off));
  uint32_t xlo = src.ring[0];
  uint32_t xhi = src.ring[64];
  src.rd = 2;
  constexpr uint32_t LUT_MASK = ((1u << LUT_BITS) - 1u) << 4;
  uint64_t rans_l = RANS_L;
  asm volatile("" : "+s"(rans_l));
#ifdef LICOS_STAMPS
  unsigned long long st_tab = 0, st_loop = 0, st_prev = __builtin_amdgcn_s_memtime();
#endif
  int32_t *sp = symbols + (size_t)b * ssb;
  for (int c = 0; c < C; ++c) {
    const int len = cdf_len[c] < 2 ? 2 : cdf_len[c];
    const int32_t max_value = len - 2;
    const int32_t off = offset[c];
    __syncthreads();
    for (int e = tid; e < len; e += nthr) s_cdf[e] = (uint32_t)cdf[(size_t)c * cdf_stride + e];
    __syncthreads();
    // one record per value of the top LUT_BITS bits of cf; s = the largest symbol whose cdf[s] <= the bucket's first value
    constexpr int KPL = (1 << LUT_BITS) / nthr;
    {
      const uint32_t key0 = (uint32_t)(tid * KPL) << (16 - LUT_BITS);
      int lo = 0, hi = len - 1;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_cdf[mid] <= key0) lo = mid; else hi = mid;
      }
      for (int k = 0; k < KPL; ++k) {
        const uint32_t key = (uint32_t)(tid * KPL + k) << (16 - LUT_BITS);
        while (lo + 1 < len - 1 && s_cdf[lo + 1] <= key) ++lo;
        // (a bucket that reaches the escape symbol within its first two symbols, or has no cdf[s + 2], is always "rare")
        const uint32_t c2 = (lo + 1 >= max_value || lo + 2 > len - 1) ? 0u : s_cdf[lo + 2];
        s_lut[tid * KPL + k] = u32x4_t{s_cdf[lo], s_cdf[lo + 1], c2, (uint32_t)(lo + off)};
      }
    }
    __syncthreads();
#ifdef LICOS_STAMPS
    { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_tab += now_ - st_prev; st_prev = now_; }
#endif
    if (!live) continue;  // idle lanes (last block only) sit out the decoding loop

    // one symbol; returns its value (offset included)
    auto decode_one = [&]() -> int32_t {
      const u32x4_t rec = *(lds_cu32x4 *)(uintptr_t)((xlo >> (16 - LUT_BITS - 4)) & LUT_MASK);
      // the next stream word is read before it is known to be needed: its LDS latency runs beside the lookup's
      const uint32_t w_next = *(lds_cu32 *)(uintptr_t)((((uint32_t)src.rd << 8) & (uint32_t)((RING - 1) << 8)) | ring_addr);
      uint32_t Xlo = __builtin_amdgcn_alignbit(xhi, xlo, 16), Xhi = xhi >> 16;  // x >> 16
      const bool adv = (xlo & 0xFFFFu) >= rec.y;
      const uint32_t lo = adv ? rec.y : rec.x, hi = adv ? rec.z : rec.y;
      int32_t value = (int32_t)rec.w + (adv ? 1 : 0);
      uint32_t freq = hi - lo, d = (xlo & 0xFFFFu) - lo;
      if (__builtin_expect(__any((xlo & 0xFFFFu) >= rec.z), 0)) {  // uniform and rare
        if ((xlo & 0xFFFFu) >= rec.z) {
          // this lane's symbol is finished HERE (walk, state update, bypass nibbles), and the common update below is handed
          // an identity - frequency 2^16 at cdf 0 of the state as it then is - so that the fast path carries one test only
          const uint32_t cf = xlo & 0xFFFFu;
          int sidx = (int)rec.w - off;
          if (sidx > len - 2) sidx = len - 2;  // (malformed tables only)
          uint32_t slo = s_cdf[sidx], shi = s_cdf[sidx + 1];
          while (sidx < len - 2 && shi <= cf) {
            ++sidx;
            slo = shi;
            shi = s_cdf[sidx + 1];
          }
          value = sidx + off;
          uint64_t x = ((uint64_t)xhi << 32) | xlo;
          auto pull = [&]() {  // (reads past the end show in the read counter: checked after the last symbol)
            if (x < RANS_L) {
              x = (x << 32) | src.ring[(src.rd & (RING - 1)) * 64];
              ++src.rd;
            }
          };
          x = (uint64_t)(shi - slo) * (x >> 16) + cf - slo;
          pull();
          if (sidx == max_value) {  // the escape symbol: the value follows in bypass nibbles
            auto bits4 = [&]() -> uint32_t {
              const uint32_t v4 = (uint32_t)(x & 15u);
              x >>= 4;
              pull();
              return v4;
            };
            uint32_t val = bits4();
            int nbyp = (int)val;
            while (val == 15u && nbyp < 64) { val = bits4(); nbyp += (int)val; }
            uint32_t raw = 0;
            for (int j = 0; j < nbyp; ++j) {
              const uint32_t nib = bits4();
              if (j < 8) raw |= nib << (j * 4);
            }
            int32_t v = (int32_t)(raw >> 1);
            v = (raw & 1u) ? -v - 1 : v + max_value;
            value = v + off;
          }
          freq = 65536u;
          d = (uint32_t)x & 0xFFFFu;
          Xlo = (uint32_t)(x >> 16);
          Xhi = (uint32_t)(x >> 48);
        }
        src.refill_if_low();  // the escape path may have drained several words
      }
      // x' = freq * (x >> 16) + d: a 64-bit multiply-add for the low 32 bits of x >> 16, a full-rate 24-bit one for the 15 above
      const uint64_t p = (uint64_t)Xlo * freq + (uint64_t)d;
      const uint32_t plo = (uint32_t)p, nhi = __umul24(Xhi, freq) + (uint32_t)(p >> 32);
      const bool need = (((uint64_t)nhi << 32) | plo) < rans_l;  // (an opaque 2^31: as a literal the compare becomes mask + compare-with-zero)
      xhi = need ? plo : nhi;
      xlo = need ? w_next : plo;
      src.rd += need ? 1 : 0;
      return value;
    };

    const size_t cbase = (size_t)c * plane;
    if (SM) {  // plane % 4 == 0, 16-byte aligned rows (the launcher checks)
      i32x4_t *dst = reinterpret_cast<i32x4_t *>(sp + cbase);
      for (int p0 = 0; p0 < plane; p0 += SYM_BUF) {
        src.refill_if_low();
        const int nq = (plane - p0 < SYM_BUF ? plane - p0 : SYM_BUF) >> 2;
        for (int q = 0; q < nq; ++q) {
          i32x4_t v;
          v.x = decode_one();
          v.y = decode_one();
          v.z = decode_one();
          v.w = decode_one();
          *dst++ = v;
        }
      }
    } else {
      static_assert(RING_LOW >= SYM_BUF, "a block may consume SYM_BUF words before the next ring check");
      for (int p0 = 0; p0 < plane; p0 += SYM_BUF) {
        src.refill_if_low();
        const int nb = plane - p0 < SYM_BUF ? plane - p0 : SYM_BUF;
        for (int k = 0; k < nb; ++k) s_out[k * 64 + lane] = decode_one();
        for (int k = 0; k < nb; ++k) sp[(size_t)(cbase + p0 + k) * ssi] = s_out[k * 64 + lane];
      }
    }
#ifdef LICOS_STAMPS
    { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_loop += now_ - st_prev; st_prev = now_; }
#endif
  }
#ifdef LICOS_STAMPS
  if (tid == 0) {
    atomicAdd(&g_dec_stamps[0], st_tab);
    atomicAdd(&g_dec_stamps[1], st_loop);
    atomicAdd(&g_dec_stamps[2], 1ull);
  }
#endif
  if (live && src.rd > src.nw) atomicOr(status, 1);
}

// ---------------------------------------------------------------------------------------------------
// Per-element-indexed fast path (GaussianConditional: the CDF row of every symbol is data).  The 64 rows have
// very different lengths (3 .. ~3100 entries); the short, frequent ones are staged in LDS (as many leading rows
// as fit IDX_CAP entries, decided by the kernel itself from cdf_len), the long ones stay in global memory.
// Symbols and row indexes are fetched a batch ahead; the decoder shares the plane kernel's LDS word ring and
// buffered symbol stores.
constexpr int IDX_CAP = 2560, IDX_MAX_ROWS = 256, IDX_LUT_ROWS = 64, IDX_LUT_BITS = 6, IDX_LUT_N = 1 << IDX_LUT_BITS;

struct IdxTables {
  int *s_len, *s_off, *s_base;  // [rows] cdf length, symbol offset, first staged entry (or -1)
};

__device__ inline void idx_stage_rows(int rows, const int32_t *cdf_len, const int32_t *offset, int *s_len, int *s_off,
                                      int *s_base, int lane) {
  for (int r = lane; r < rows; r += 64) { s_len[r] = cdf_len[r]; s_off[r] = offset[r]; }
  __syncthreads();
  if (lane == 0) {
    int used = 0;
    for (int r = 0; r < rows; ++r) {
      if (used >= 0 && used + s_len[r] <= IDX_CAP) { s_base[r] = used; used += s_len[r]; }
      else { s_base[r] = -1; used = -1; }  // only a leading run of rows is staged
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(64) void rans_encode_indexed_kernel(const int32_t *__restrict__ symbols,
                                                                 const int32_t *__restrict__ indexes, long ssb, long ssi, int n,
                                                                 int rows, int cdf_stride, const int32_t *__restrict__ cdf_len,
                                                                 const int32_t *__restrict__ offset,
                                                                 const EncRec *__restrict__ table, uint32_t *__restrict__ words,
                                                                 int cap_words, int32_t *__restrict__ nwords,
                                                                 int32_t *__restrict__ status, int B) {
  __shared__ int s_len[IDX_MAX_ROWS], s_off[IDX_MAX_ROWS], s_base[IDX_MAX_ROWS];
  __shared__ __attribute__((aligned(16))) EncRec s_tab[IDX_CAP];
  const int lane = threadIdx.x;
  const int b_raw = blockIdx.x * 64 + lane;
  const bool live = b_raw < B;
  const int b = live ? b_raw : B - 1;
  idx_stage_rows(rows, cdf_len, offset, s_len, s_off, s_base, lane);
  for (int r = 0; r < rows; ++r) {
    const int base = s_base[r];
    if (base < 0) break;
    for (int e = lane; e < s_len[r] - 1; e += 64) s_tab[base + e] = table[(size_t)r * cdf_stride + e];
  }
  __syncthreads();
  if (!live) return;  // the tables are staged: idle lanes (last block only) have nothing left to do
  WordSink sink{words, B, b, cap_words, false};
  uint64_t x = RANS_L;
  const int32_t *sp = symbols + (size_t)b * ssb;
  const int32_t *ip = indexes + (size_t)b * ssb;
  int32_t sv_next[SYM_BATCH], iv_next[SYM_BATCH];
  auto fetch = [&](int i1) {
#pragma unroll
    for (int k = 0; k < SYM_BATCH; ++k) {
      const int i = i1 - 1 - k;
      sv_next[k] = (i >= 0) ? sp[(size_t)i * ssi] : 0;
      iv_next[k] = (i >= 0) ? ip[(size_t)i * ssi] : 0;
    }
  };
  int fetch_from = 0;
  auto fetch_next = [&]() { fetch(fetch_from); };
  // one batch of nb <= SYM_BATCH symbols (FULL: nb == SYM_BATCH, no per-symbol bound checks); same straight-line
  // step as the plane kernel
  auto code_batch = [&](auto full_c, int nb, bool more) {
    constexpr bool FULL = decltype(full_c)::value;
    int32_t sv[SYM_BATCH], mx[SYM_BATCH];
    EncRec rec[SYM_BATCH];
#pragma unroll
    for (int k = 0; k < SYM_BATCH; ++k) {
      const int c = iv_next[k];
      mx[k] = s_len[c] - 2;
      sv[k] = sv_next[k] - s_off[c];
      const int32_t v = (sv[k] < 0 || sv[k] >= mx[k]) ? mx[k] : sv[k];
      const int base = s_base[c];
      rec[k] = (base >= 0) ? s_tab[base + v] : table[(size_t)c * cdf_stride + v];
    }
    if (more) fetch_next();
#pragma unroll
    for (int k = 0; k < SYM_BATCH; ++k) {
      if (!FULL && k >= nb) break;
      const int32_t value = sv[k];
      const bool escape = value < 0 || value >= mx[k];
      if (__builtin_expect(__any(escape), 0)) {  // uniform and rare
        if (escape) {
          const uint32_t raw = (value < 0) ? (uint32_t)(-2 * value - 1) : (uint32_t)(2 * (value - mx[k]));
          int nbyp = 0;
          while (nbyp < 8 && (raw >> (nbyp * 4)) != 0) ++nbyp;
          for (int j = nbyp - 1; j >= 0; --j) put_bits4(x, sink, (raw >> (j * 4)) & 15u);
          put_bits4(x, sink, (uint32_t)nbyp);
        }
      }
      const uint32_t freq = rec[k].freq ? rec[k].freq : 65536u;
      const bool emit = (uint32_t)(x >> 32) >= (freq << 15);
      sink.put_if(emit, (uint32_t)x);
      x = emit ? (x >> 32) : x;
      const uint64_t q = __umul64hi(x, rec[k].rcp) >> rec[k].shift;
      x = x + rec[k].bias + q * (uint64_t)(65536u - freq);
    }
  };
  fetch(n);
  int i1 = n;
  for (; i1 >= SYM_BATCH; i1 -= SYM_BATCH) {
    fetch_from = i1 - SYM_BATCH;
    code_batch(std::true_type{}, SYM_BATCH, i1 - SYM_BATCH > 0);
  }
  if (i1 > 0) code_batch(std::false_type{}, i1, false);
  sink.put((uint32_t)(x >> 32));
  sink.put((uint32_t)x);
  nwords[b] = cap_words - sink.wp;
  if (sink.overflow) atomicOr(status, 1);
}

__global__ __launch_bounds__(64) void rans_decode_indexed_kernel(const uint8_t *__restrict__ in,
                                                                 const int64_t *__restrict__ byte_off,
                                                                 const int32_t *__restrict__ indexes, long ssb, long ssi, int n,
                                                                 int rows, const int32_t *__restrict__ cdf, int cdf_stride,
                                                                 const int32_t *__restrict__ cdf_len,
                                                                 const int32_t *__restrict__ offset, int32_t *__restrict__ symbols,
                                                                 int32_t *__restrict__ status, int B) {
  __shared__ int s_len[IDX_MAX_ROWS], s_off[IDX_MAX_ROWS], s_base[IDX_MAX_ROWS];
  __shared__ uint32_t s_cdf[IDX_CAP];
  __shared__ uint32_t s_ring[RING * 64];
  __shared__ int32_t s_out[SYM_BUF * 64];
  // per-row search LUT over the top IDX_LUT_BITS bits of cf: s_lut[r][k] = largest s with row[s] <= k << (16 - bits),
  // plus one closing entry len - 2; the symbol of cf then lies in [s_lut[k], s_lut[k+1] + 1) and the binary search
  // below needs a step or two instead of log2(len) - with 64 streams in lock-step every saved step is saved 64 times
  __shared__ uint16_t s_lut[IDX_LUT_ROWS * (IDX_LUT_N + 1)];
  const bool use_lut = rows <= IDX_LUT_ROWS;
  const int lane = threadIdx.x;
  const int b_raw = blockIdx.x * 64 + lane;
  const bool live = b_raw < B;
  const int b = live ? b_raw : B - 1;
  idx_stage_rows(rows, cdf_len, offset, s_len, s_off, s_base, lane);
  for (int r = 0; r < rows; ++r) {
    const int base = s_base[r];
    if (base < 0) break;
    for (int e = lane; e < s_len[r]; e += 64) s_cdf[base + e] = (uint32_t)cdf[(size_t)r * cdf_stride + e];
  }
  __syncthreads();
  if (use_lut) {
    for (int e = lane; e < rows * (IDX_LUT_N + 1); e += 64) {
      const int r = e / (IDX_LUT_N + 1), k = e - r * (IDX_LUT_N + 1);
      const int len = s_len[r];
      int lo = len - 2;
      if (k < IDX_LUT_N) {
        const uint32_t key = (uint32_t)k << (16 - IDX_LUT_BITS);
        const int base = s_base[r];
        lo = 0;
        int hi = len - 1;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          const uint32_t v = base >= 0 ? s_cdf[base + mid] : (uint32_t)cdf[(size_t)r * cdf_stride + mid];
          if (v <= key) lo = mid; else hi = mid;
        }
      }
      s_lut[e] = (uint16_t)lo;
    }
    __syncthreads();
  }
  RingSource src;
  src.init(reinterpret_cast<const uint32_t *>(in + byte_off[b]), (int)((byte_off[b + 1] - byte_off[b]) / 4), s_ring + lane,
           reinterpret_cast<const uint32_t *>(byte_off));
  uint64_t x = (uint64_t)src.next();
  x |= (uint64_t)src.next() << 32;
  int32_t *sp = symbols + (size_t)b * ssb;
  const int32_t *ip = indexes + (size_t)b * ssb;
  int32_t iv_next[SYM_BUF];
  auto fetch = [&](int i0) {
#pragma unroll
    for (int k = 0; k < SYM_BUF; ++k) iv_next[k] = (i0 + k < n) ? ip[(size_t)(i0 + k) * ssi] : 0;
  };
  fetch(0);
  for (int i0 = 0; i0 < n; i0 += SYM_BUF) {
    int32_t iv[SYM_BUF];
#pragma unroll
    for (int k = 0; k < SYM_BUF; ++k) iv[k] = iv_next[k];
    if (i0 + SYM_BUF < n) fetch(i0 + SYM_BUF);
    const int nb = (n - i0) < SYM_BUF ? (n - i0) : SYM_BUF;
#pragma unroll
    for (int k = 0; k < SYM_BUF; ++k) {
      if (k >= nb) break;
      src.refill_if_low();
      const int c = iv[k];
      const int len = s_len[c];
      const int32_t max_value = len - 2;
      const int base = s_base[c];
      const uint32_t cf = (uint32_t)(x & 0xFFFFu);
      int lo = 0, hi = len - 1;  // row[lo] <= cf < row[hi]
      if (use_lut) {
        const uint16_t *lr = s_lut + c * (IDX_LUT_N + 1) + (cf >> (16 - IDX_LUT_BITS));
        lo = lr[0];
        hi = lr[1] + 1;
      }
      uint32_t vlo, vhi;
      if (base >= 0) {
        const uint32_t *row = s_cdf + base;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (row[mid] <= cf) lo = mid; else hi = mid;
        }
        vlo = row[lo];
        vhi = row[lo + 1];
      } else {
        const int32_t *row = cdf + (size_t)c * cdf_stride;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if ((uint32_t)row[mid] <= cf) lo = mid; else hi = mid;
        }
        vlo = (uint32_t)row[lo];
        vhi = (uint32_t)row[lo + 1];
      }
      x = (uint64_t)(vhi - vlo) * (x >> 16) + cf - vlo;
      if (x < RANS_L) x = (x << 32) | src.next();
      int32_t value = lo;
      if (value == max_value) {
        uint32_t val = get_bits4p(x, src);
        int nbp = (int)val;
        while (val == 15u && nbp < 64) { val = get_bits4p(x, src); nbp += (int)val; }
        uint32_t raw = 0;
        for (int j = 0; j < nbp; ++j) {
          const uint32_t nib = get_bits4p(x, src);
          if (j < 8) raw |= nib << (j * 4);
        }
        value = (int32_t)(raw >> 1);
        value = (raw & 1u) ? -value - 1 : value + max_value;
      }
      s_out[k * 64 + lane] = value + s_off[c];
    }
    if (live)
      for (int k = 0; k < nb; ++k) sp[(size_t)(i0 + k) * ssi] = s_out[k * 64 + lane];
  }
  if (live && src.over) atomicOr(status, 1);
}

}  // namespace licos

using namespace licos;

// Waves per workgroup of the plane coders.  A coder wave is a latency chain that leaves its CU almost idle - but a CU
// that holds one cannot take a workgroup of the 8-wave transform kernels, which need every register of all four SIMDs
// (2 waves x 256 VGPRs each), and the coders run BESIDE the transforms of the neighbouring chunk.  One wave per
// workgroup spreads 4096 streams over 64 CUs (a quarter of the chip closed to the transforms for the 10-20 ms a coder
// launch lasts); two waves per workgroup close 32, four - one per SIMD, so no wave shares an issue port - 16.  Measured
// on the bench step: 1 wave 230.7 ms, 2 waves 227.6 ms, 4 waves 228.4 ms (the waves of a workgroup wait for each other
// at every channel's table: a 4-wave encode launch takes 8.7 instead of 7.9 ms, and the last one of a step is exposed).
// LICOS_CODER_WAVES (1, 2 or 4) overrides for A/B runs.
// Round 5: the instruction-counted plane decoder prefers FOUR (95 against 102 ns per symbol alone; a 16 384-tile decompress
// 100.6 -> 99.2 ms, 1024 tiles 10.4 -> 9.6), the stream-major encoder two (98 against 144 ns: its waves meet at every
// channel's table) - profiles/r05_coder_waves_ab.log.
static int coder_waves(int B, bool decoder = false) {
  static const int forced = [] { const char *e = getenv("LICOS_CODER_WAVES"); return e ? atoi(e) : 0; }();
  int w = (forced == 1 || forced == 2 || forced == 4) ? forced : decoder ? 4 : 2;
  while (w > 1 && B <= 64 * (w / 2)) w /= 2;  // small batches: no more waves than there are streams for
  return w;
}

extern "C" {

#ifdef LICOS_STAMPS
int licos_debug_dec_stamps(unsigned long long *out, int reset) {
  if (out) LICOS_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_stamps), sizeof(unsigned long long) * 4));
  if (reset) {
    unsigned long long z[4] = {};
    LICOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_dec_stamps), z, sizeof(z)));
  }
  return LICOS_OK;
}
#endif

int licos_rans_encode_batch(const int32_t *symbols, const int32_t *indexes, long ssb, long ssi, int n, int plane,
                            const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset,
                            const void *enc_table, uint32_t *words, int cap_words, int32_t *nwords, int32_t *status,
                            int B, void *stream) {
  (void)cdf;
  LICOS_REQUIRE(symbols && cdf_len && offset && enc_table && words && nwords && status, "rans_encode_batch: NULL buffer");
  LICOS_REQUIRE(B > 0 && n > 0 && cap_words >= 2 && cdf_stride > 1, "rans_encode_batch: bad sizes B=%d n=%d cap=%d", B, n, cap_words);
  LICOS_REQUIRE(indexes || plane > 0, "rans_encode_batch: need indexes or a plane size");
  if (!indexes && n % plane == 0 && (size_t)cdf_stride * sizeof(EncRec) <= 64 * 1024) {
    const int cw = coder_waves(B);
    hipLaunchKernelGGL(rans_encode_plane_kernel, dim3(cdiv(B, 64 * cw)), dim3(64 * cw), (size_t)cdf_stride * sizeof(EncRec),
                       as_stream(stream), symbols, ssb, ssi, n / plane, plane, cdf_stride, cdf_len, offset,
                       static_cast<const EncRec *>(enc_table), words, cap_words, nwords, status, B);
  } else if (indexes && plane > 0 && plane <= IDX_MAX_ROWS) {
    // with explicit indexes `plane` carries the number of CDF rows (see licos_hip.h)
    hipLaunchKernelGGL(rans_encode_indexed_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), symbols, indexes, ssb,
                       ssi, n, plane, cdf_stride, cdf_len, offset, static_cast<const EncRec *>(enc_table), words, cap_words,
                       nwords, status, B);
  } else {
    hipLaunchKernelGGL(rans_encode_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), symbols, indexes, ssb,
                       ssi, n, plane, cdf_stride, cdf_len, offset, static_cast<const EncRec *>(enc_table), words,
                       cap_words, nwords, status, B);
  }
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_rans_compact(const uint32_t *words, int cap_words, const int32_t *nwords, const int64_t *byte_off,
                       uint8_t *out, int B, void *stream) {
  LICOS_REQUIRE(words && nwords && byte_off && out && B > 0 && B <= 65535 && cap_words > 0, "rans_compact: bad arguments");
  LICOS_REQUIRE(((uintptr_t)out & 3) == 0, "rans_compact: out must be 4-byte aligned");
  hipLaunchKernelGGL(rans_compact_kernel, dim3(cdiv(cap_words, 256) < 8 ? cdiv(cap_words, 256) : 8, B), dim3(256), 0,
                     as_stream(stream), words, cap_words, nwords, byte_off, reinterpret_cast<uint32_t *>(out), B);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_rans_decode_batch(const uint8_t *in, const int64_t *byte_off, const int32_t *indexes, long ssb, long ssi,
                            int n, int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len,
                            const int32_t *offset, int32_t *symbols, int32_t *status, int B, void *stream) {
  LICOS_REQUIRE(in && byte_off && cdf && cdf_len && offset && symbols && status, "rans_decode_batch: NULL buffer");
  LICOS_REQUIRE(B > 0 && n > 0 && cdf_stride > 1, "rans_decode_batch: bad sizes");
  LICOS_REQUIRE(indexes || plane > 0, "rans_decode_batch: need indexes or a plane size");
  LICOS_REQUIRE(((uintptr_t)in & 3) == 0, "rans_decode_batch: input must be 4-byte aligned");
  int cw = coder_waves(B);
  size_t dec_lds = (size_t)cw * (RING + SYM_BUF) * 64 * 4 + ((size_t)8 << LUT_BITS) + (size_t)cdf_stride * 4;
  if (dec_lds > 156 * 1024) {  // very long tables: one wave per workgroup
    cw = 1;
    dec_lds = (size_t)(RING + SYM_BUF) * 64 * 4 + ((size_t)8 << LUT_BITS) + (size_t)cdf_stride * 4;
  }
  static const bool dec4 = [] { const char *e = getenv("LICOS_RANS_DEC4"); return !e || atoi(e) != 0; }();  // (A/B: 0 = the round-4 kernel)
  if (dec4 && !indexes && n % plane == 0 && cdf_stride <= 65535) {
    int w4 = coder_waves(B, true);
    auto lds4 = [&](int w) { return (size_t)DEC_LUT_BYTES + (size_t)w * (RING + SYM_BUF) * 256 + (size_t)cdf_stride * 4; };
    while (w4 > 1 && lds4(w4) > 156 * 1024) w4 /= 2;
    if (lds4(w4) <= 156 * 1024) {
      const bool sm = ssi == 1 && plane % 4 == 0 && ssb % 4 == 0 && ((uintptr_t)symbols & 15) == 0;
      auto launch4 = [&](auto kern) -> int {
        LICOS_ENSURE_LDS(kern, lds4(w4));
        hipLaunchKernelGGL(kern, dim3(cdiv(B, 64 * w4)), dim3(64 * w4), lds4(w4), as_stream(stream), in, byte_off, ssb, ssi, n / plane, plane,
                           cdf, cdf_stride, cdf_len, offset, symbols, status, B);
        LICOS_LAUNCH_CHECK();
        return LICOS_OK;
      };
      switch (w4) {
        case 1: return sm ? launch4(rans_decode_plane4_kernel<1, true>) : launch4(rans_decode_plane4_kernel<1, false>);
        case 2: return sm ? launch4(rans_decode_plane4_kernel<2, true>) : launch4(rans_decode_plane4_kernel<2, false>);
        default: return sm ? launch4(rans_decode_plane4_kernel<4, true>) : launch4(rans_decode_plane4_kernel<4, false>);
      }
    }
  }
  if (!indexes && n % plane == 0 && dec_lds <= 156 * 1024 && cdf_stride <= 65535) {
    LICOS_ENSURE_LDS(rans_decode_plane_kernel, dec_lds);
    hipLaunchKernelGGL(rans_decode_plane_kernel, dim3(cdiv(B, 64 * cw)), dim3(64 * cw), dec_lds,
                       as_stream(stream), in, byte_off, ssb, ssi, n / plane, plane, cdf, cdf_stride, cdf_len, offset,
                       symbols, status, B);
  } else if (indexes && plane > 0 && plane <= IDX_MAX_ROWS) {
    hipLaunchKernelGGL(rans_decode_indexed_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), in, byte_off, indexes,
                       ssb, ssi, n, plane, cdf, cdf_stride, cdf_len, offset, symbols, status, B);
  } else {
    hipLaunchKernelGGL(rans_decode_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), in, byte_off, indexes,
                       ssb, ssi, n, plane, cdf, cdf_stride, cdf_len, offset, symbols, status, B);
  }
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // extern "C"
