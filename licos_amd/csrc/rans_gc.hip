// Scale-conditioned rANS path (CompressAI GaussianConditional.compress / decompress, BASELINE config 5): every symbol
// of a tile's y stream (192 x 32 x 32 = 196 608 of them for a 512 x 512 tile) is coded with the CDF row its predicted
// scale selects.  The stream format is rans.hip's (bit-exact with rans_interface.cpp); what differs is where the time
// goes: a lane owns a stream and the recurrence is strictly serial, so a launch lasts n_symbols x (latency of one
// state -> state step) whatever the batch, and everything that can be computed WITHOUT the coder state is moved out of
// that chain into throughput kernels that fill the chip:
//
//   encode   gc_encode_prepare_kernel   (y, scales) -> one 16-byte encoder record per symbol (table row by scale,
//                                       symbol by rounding, escape folded in), transposed to [position][stream]
//            rans_encode_records_kernel the serial part: renormalise + reciprocal multiply per record, no table at all
//   decode   gc_decode_prepare_kernel   scales -> one table-row byte per symbol, [block of 16][stream][16]
//            rans_decode_image_kernel   the serial part: per symbol ONE 8-byte LDS read of the decoder image
//                                       (rans_image.hpp) resolves symbol and interval; row bytes arrive by LDS-DMA two
//                                       blocks ahead and decoded symbols leave by counted stores, so the loop never
//                                       waits for memory
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include <cstdlib>

#include "common.hpp"
#include "rans_image.hpp"

namespace licos {
namespace gc {

constexpr uint64_t RANS_L = 1ull << 31;
constexpr int SYM_BLK = 16;          // symbols per decode block (= one 16-byte row-index granule per stream)
// LDS ring of stream words per lane.  A refill tops up every lane that has room for 16 more words (level <= RING - 16), and
// is triggered when ANY lane is down to RING_LOW: the gap between the two is what makes a refill lift the whole wave well
// above the trigger - with RING = 32 the two coincided, only the triggering lanes were topped up, and some lane of the 64
// triggered a full memory round trip nearly every block (93 of 600 cycles per symbol, in-kernel stamps).  A block takes at
// most SYM_BLK words off the ring between checks.
constexpr int RING = 64, RING_LOW = 24;
constexpr int IDEPTH = 4;            // LDS-DMA slots of row-index granules per wave
constexpr uint32_t REC_ESCAPE = 0x80000000u;  // flag in EncRec's (freq | shift << 16) dword: raw value in the aux plane

struct EncRec { uint64_t rcp; uint32_t bias; uint16_t freq; uint16_t shift; };

// ---------------------------------------------------------------------------------------------- encode: prepare
__device__ inline int scale_row(float s, const float *__restrict__ table, int levels, float bound) {
  s = fmaxf(s, bound);
  int idx = levels - 1;
  for (int t = 0; t < levels - 1; ++t) idx -= (s <= table[t]) ? 1 : 0;
  return idx;
}

// block = 64 streams x 32 positions; latents and scales are read along positions (NCHW rows), records leave along
// streams (the coder's lanes), through LDS.
// BY_SCALE: the row of an element is chosen by its predicted scale (GaussianConditional); otherwise it is the element's
// channel, position / plane, and the symbol is round(y - median[channel]) (EntropyBottleneck: `scales` = the medians).
// A thread owns 4 consecutive positions of 2 streams.  Its 8 elements go through the kernel's three dependent memory
// steps TOGETHER - all latents and scales (16-byte loads when n % 4 == 0), then all row lengths / offsets, then all 8
// table records - so a workgroup pays three round trips, not the 24 of an element-at-a-time loop (round 4: 6.9 -> see
// DESIGN 6.1 per 2048 tiles of 196 608 symbols; the launch sits on the main stream in front of every chunk's coder).
template <bool BY_SCALE>
__global__ __launch_bounds__(256) void gc_encode_prepare_kernel(const float *__restrict__ y, const float *__restrict__ scales,
                                                               const float *__restrict__ table, int levels, float bound,
                                                               const uint4 *__restrict__ enc_table, int cdf_stride,
                                                               const int32_t *__restrict__ cdf_len,
                                                               const int32_t *__restrict__ offset, uint4 *__restrict__ rec,
                                                               int32_t *__restrict__ aux, int B, long n) {
  __shared__ uint4 t_rec[32][65];
  __shared__ int32_t t_aux[32][65];
  const long i0 = (long)blockIdx.x * 32;
  const int b0 = blockIdx.y * 64;
  {
    const int q = threadIdx.x & 7, r0 = threadIdx.x >> 3;  // positions 4q .. 4q + 3 of streams r0 and r0 + 32
    const long i = i0 + 4 * q;
    // (n % 4 == 0 and 16-byte aligned planes: every stream starts on a 16-byte boundary, and i + 3 < n whenever i < n)
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | (BY_SCALE ? reinterpret_cast<uintptr_t>(scales) : 0)) & 15) == 0;
    float yv[2][4], sv[2][4];
    bool live[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int b = b0 + r0 + 32 * h;
      const size_t at = (size_t)b * n + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        live[h][j] = b < B && i + j < n;
        yv[h][j] = 0.f;
        sv[h][j] = 0.f;
      }
      if (vec) {
        if (live[h][0]) {
          const float4 a4 = *reinterpret_cast<const float4 *>(y + at);
          yv[h][0] = a4.x; yv[h][1] = a4.y; yv[h][2] = a4.z; yv[h][3] = a4.w;
          if (BY_SCALE) {
            const float4 s4 = *reinterpret_cast<const float4 *>(scales + at);
            sv[h][0] = s4.x; sv[h][1] = s4.y; sv[h][2] = s4.z; sv[h][3] = s4.w;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (live[h][j]) {
            yv[h][j] = y[at + j];
            if (BY_SCALE) sv[h][j] = scales[at + j];
          }
      }
    }
    int c[2][4];
    if (BY_SCALE) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          sv[h][j] = fmaxf(sv[h][j], bound);
          c[h][j] = levels - 1;
        }
      for (int t = 0; t < levels - 1; ++t) {  // (scale_row for the 8 elements at once: one table value per step)
        const float tv = table[t];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) c[h][j] -= (sv[h][j] <= tv) ? 1 : 0;
      }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[h][j] = live[h][j] ? (int)((i + j) / levels) : 0;  // (`levels` = the plane size)
    }
    int32_t maxv[2][4], offs[2][4];
    float med[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        maxv[h][j] = cdf_len[c[h][j]] - 2;
        offs[h][j] = offset[c[h][j]];
        med[h][j] = BY_SCALE ? 0.f : scales[c[h][j]];
      }
    int32_t v[2][4], raw[2][4];
    bool esc[2][4];
    uint4 e[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float centred = BY_SCALE ? yv[h][j] : yv[h][j] - med[h][j];
        int32_t vv = (int32_t)rintf(centred) - offs[h][j];  // round-half-to-even, as torch.round
        int32_t rr = 0;
        bool ee = false;
        if (vv < 0) { rr = -2 * vv - 1; vv = maxv[h][j]; ee = true; }
        else if (vv >= maxv[h][j]) { rr = 2 * (vv - maxv[h][j]); vv = maxv[h][j]; ee = true; }
        v[h][j] = vv; raw[h][j] = rr; esc[h][j] = ee;
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) e[h][j] = enc_table[(size_t)c[h][j] * cdf_stride + v[h][j]];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 ev = e[h][j];
        // the serial kernel's form of the last dword: 2^16 - freq (what the update multiplies by, and what the
        // renormalisation bound is computed from) | shift << 16 | escape flag
        ev.w = ((0x10000u - (ev.w & 0xFFFFu)) & 0xFFFFu) | (ev.w & 0x7FFF0000u) | (esc[h][j] ? REC_ESCAPE : 0u);
        if (!live[h][j]) { ev = make_uint4(0, 0, 0, 0); raw[h][j] = 0; }
        t_rec[4 * q + j][r0 + 32 * h] = ev;
        t_aux[4 * q + j][r0 + 32 * h] = raw[h][j];
      }
  }
  __syncthreads();
  {
    const int r = threadIdx.x & 63;
    const int b = b0 + r;
    for (int p = threadIdx.x >> 6; p < 32; p += 4) {
      const long i = i0 + p;
      if (i < n && b < B) {
        rec[(size_t)i * B + b] = t_rec[p][r];
        aux[(size_t)i * B + b] = t_aux[p][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- encode: serial part
// 16 B per lane, global -> LDS, no register round trip (LDS address = wave base + lane * 16); counted in vmcnt only.
__device__ __forceinline__ void dma16(const void *gsrc, void *lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(m0v)
               : "memory");
}

// Output words of a stream, back to front, in the interleaved scratch [row][stream]: rows 1 .. cap hold the words (the
// first word written goes to row cap), row 0 is a dump row - once a stream has used all its rows, further words land
// there and the stream is flagged - so the bookkeeping of a word is one select, one subtract and one max on a 32-bit
// byte offset.  Every store is issued from here as inline assembly (scalar base + vector offset): the kernel's waits
// count vector-memory operations (see below) and must know each one.
struct WordSink {
  const uint32_t *base;  // row 0 of this stream (wave-uniform part folded in by the compiler or not - a 64-bit pair either way)
  uint32_t off;          // byte offset of the next free row; 0 = the dump row
  uint32_t stride;       // bytes between two rows
  __device__ inline void store(uint32_t w) const {
    // (the base through an s_mov: an SGPR the allocator brings back from a VGPR lane with v_readlane must not be read by a
    // vector-memory instruction within five wait states, and nothing inside an asm statement is padded)
    uint64_t base_copy;
    asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dword %1, %2, %0" : "=&s"(base_copy) : "v"(off), "v"(w), "s"(base) : "memory");
  }
  __device__ inline void claim(bool emit) {
    const uint32_t dec = emit ? stride : 0u;
    off = off >= dec ? off - dec : 0u;
  }
  // branch-free: the word is always stored to the next free row, the row is claimed only when `emit` is set
  __device__ inline void put_if(bool emit, uint32_t w) { store(w); claim(emit); }
  __device__ inline void put(uint32_t w) { put_if(true, w); }
};

__device__ inline void put_bits4(uint64_t &x, WordSink &sink, uint32_t val) {
  if (x >= (1ull << 59)) { sink.put((uint32_t)x); x >>= 32; }  // Rans64EncPutBits, nbits = 4
  x = (x << 4) | val;
}

constexpr int ENC_BATCH = 8, ENC_DEPTH = 8, ENC_AHEAD = 4;  // records are requested ENC_AHEAD batches (32 symbols, ~6 us) ahead

// Records arrive by LDS-DMA ENC_AHEAD batches ahead; words leave by stores nobody waits for.  vmcnt discipline of batch t:
//     wait for batch t's records | read them | request batch t + ENC_AHEAD | ENC_BATCH symbols, one store each (+ escapes)
// Everything issued after batch t's request may stay in flight: `s_waitcnt vmcnt(N)` with N = the number of those
// operations (each batch issues ENC_BATCH request pieces and at least ENC_BATCH stores) waits for that request alone.
// (The first version fetched records into registers and let the compiler place the waits: `vmcnt(0)` at every batch,
// i.e. a full round trip of the stores just issued per 8 symbols - 235 ns per symbol, most of it that wait.)
// Round 5: the records in REGISTERS instead of an LDS ring.  An LDS-DMA request costs its wave 80 - 100 cycles of issue (M0
// juggling, the instruction itself: in-kernel stamps of mfma_first16.hip), one per symbol here - a quarter of the ~330 cycles
// a symbol took; a plain 16-byte load per lane and symbol costs a tenth of that and needs no ds_read behind it.  Four
// batches of eight records (128 VGPRs - the kernel has the file to itself) are in flight; the loads are inline assembly
// with a counted wait in front of their use: behind a batch's request (issued right behind its own coding, so that the
// slot's old values need no copy) come three batches of 8 loads + at least 8 stores - s_waitcnt vmcnt(48).  Requests past
// the front of the stream re-read record 0 (never coded): every batch issues its eight loads whatever its position.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rans_encode_records_regs_kernel(const uint4 *__restrict__ rec, const int32_t *__restrict__ aux,
                                                                              long n, uint32_t *__restrict__ words, int cap_words,
                                                                              int32_t *__restrict__ nwords, int32_t *__restrict__ status,
                                                                              int B) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_b0 = (blockIdx.x * WAVES + wave) * 64;
  if (wave_b0 >= B) return;
  const bool live = wave_b0 + lane < B;
  const int b = live ? wave_b0 + lane : B - 1;  // idle lanes shadow the last stream; their stores go to the same rows with the same words
  const int32_t *ap = aux + b;
  const long nbat = (n + ENC_BATCH - 1) / ENC_BATCH;
  const unsigned voff = (unsigned)b * 16u;
  const uint64_t row_bytes = (uint64_t)B * 16u;
  static_assert(ENC_BATCH == 8 && ENC_AHEAD == 4, "eight loads per request, vmcnt(48) = three batches of 8 loads + 8 stores");
  // The ring lives in NAMED registers, v[128:255] (slot j, record k: v[128 + 32 j + 4 k ...]), that no C++ value ever
  // occupies: a request only clobbers them, and the statement that waits is the one that DEFINES the slot's values
  // (physical-register output constraints).  Passed through "+v" operands of the wait - the first form of this kernel -
  // the allocator may give the operand another register and copy the ring's register into it IN FRONT of the wait: a copy
  // of data still in flight (csrc/rans.hip, rans_encode_plane_kernel, where it was caught).  tests/test_host.py checks in
  // the shipped library's disassembly that the compiler's own values stay below v128 in this kernel.
#define LICOS_REC_RING_CLOBBER_10(B0)                                                                                                      \
  "v" #B0 "0", "v" #B0 "1", "v" #B0 "2", "v" #B0 "3", "v" #B0 "4", "v" #B0 "5", "v" #B0 "6", "v" #B0 "7", "v" #B0 "8", "v" #B0 "9"
  // (v128 .. v255 by decades: 12x = v120..v129 would name v120-v127 too, so the list starts at 128 explicitly)
#define LICOS_REC_RING_CLOBBER                                                                                                             \
  "v128", "v129", LICOS_REC_RING_CLOBBER_10(13), LICOS_REC_RING_CLOBBER_10(14), LICOS_REC_RING_CLOBBER_10(15), LICOS_REC_RING_CLOBBER_10(16), \
      LICOS_REC_RING_CLOBBER_10(17), LICOS_REC_RING_CLOBBER_10(18), LICOS_REC_RING_CLOBBER_10(19), LICOS_REC_RING_CLOBBER_10(20),            \
      LICOS_REC_RING_CLOBBER_10(21), LICOS_REC_RING_CLOBBER_10(22), LICOS_REC_RING_CLOBBER_10(23), LICOS_REC_RING_CLOBBER_10(24), "v250",    \
      "v251", "v252", "v253", "v254", "v255"
#define LICOS_REC_LD(S, K, OP) "global_load_dwordx4 v[" #S "+4*" #K ":" #S "+4*" #K "+3], %0, %" #OP "\n\t"
#define LICOS_REC_RING_REQUEST(S)                                                                                                          \
  asm volatile(LICOS_REC_LD(S, 0, 1) LICOS_REC_LD(S, 1, 2) LICOS_REC_LD(S, 2, 3) LICOS_REC_LD(S, 3, 4) LICOS_REC_LD(S, 4, 5)                 \
                   LICOS_REC_LD(S, 5, 6) LICOS_REC_LD(S, 6, 7) LICOS_REC_LD(S, 7, 8)                                                         \
               :: "v"(voff), "s"(base[0]), "s"(base[1]), "s"(base[2]), "s"(base[3]), "s"(base[4]), "s"(base[5]), "s"(base[6]), "s"(base[7])  \
               : "memory", LICOS_REC_RING_CLOBBER)
#define LICOS_REC_RING_LANDED(WAIT, S0, S1, S2, S3, S4, S5, S6, S7)                                                                        \
  asm volatile(WAIT : "={v[" S0 "]}"(cur[0]), "={v[" S1 "]}"(cur[1]), "={v[" S2 "]}"(cur[2]), "={v[" S3 "]}"(cur[3]), "={v[" S4 "]}"(cur[4]), \
               "={v[" S5 "]}"(cur[5]), "={v[" S6 "]}"(cur[6]), "={v[" S7 "]}"(cur[7])::"memory")
  auto request = [&](auto jc, long t) {
    constexpr int j = decltype(jc)::value;
    // records n - 1 - 8 t - k, k = 0 .. 7: scalar row bases (wave-uniform), one 32-bit lane offset
    long i0 = n - 1 - t * ENC_BATCH;
    uint64_t base[ENC_BATCH];
#pragma unroll
    for (int k = 0; k < ENC_BATCH; ++k) {
      const long i = i0 - k < 0 ? 0 : i0 - k;
      base[k] = reinterpret_cast<uint64_t>(rec) + (uint64_t)i * row_bytes;
    }
    if constexpr (j == 0) LICOS_REC_RING_REQUEST(128);
    if constexpr (j == 1) LICOS_REC_RING_REQUEST(160);
    if constexpr (j == 2) LICOS_REC_RING_REQUEST(192);
    if constexpr (j == 3) LICOS_REC_RING_REQUEST(224);
  };
  auto landed = [&](auto jc, auto counted, u32x4 (&cur)[ENC_BATCH]) {
    constexpr int j = decltype(jc)::value;
    constexpr bool COUNTED = decltype(counted)::value;
#define LICOS_REC_LANDED_SLOT(WAIT)                                                                                                      \
    if constexpr (j == 0) LICOS_REC_RING_LANDED(WAIT, "128:131", "132:135", "136:139", "140:143", "144:147", "148:151", "152:155", "156:159"); \
    if constexpr (j == 1) LICOS_REC_RING_LANDED(WAIT, "160:163", "164:167", "168:171", "172:175", "176:179", "180:183", "184:187", "188:191"); \
    if constexpr (j == 2) LICOS_REC_RING_LANDED(WAIT, "192:195", "196:199", "200:203", "204:207", "208:211", "212:215", "216:219", "220:223"); \
    if constexpr (j == 3) LICOS_REC_RING_LANDED(WAIT, "224:227", "228:231", "232:235", "236:239", "240:243", "244:247", "248:251", "252:255");
    if constexpr (COUNTED) { LICOS_REC_LANDED_SLOT("s_waitcnt vmcnt(48)") } else { LICOS_REC_LANDED_SLOT("s_waitcnt vmcnt(0)") }
  };
  WordSink sink{words, ((uint32_t)cap_words * (uint32_t)B + (uint32_t)b) * 4u, (uint32_t)B * 4u};
  uint64_t x = RANS_L;
  auto code_symbol = [&](const u32x4 r) {
    const uint32_t cfreq = r.w & 0xFFFFu;  // 2^16 - freq
    const uint32_t shift = (r.w >> 16) & 0x7FFFu;
    const uint64_t rcp = ((uint64_t)r.y << 32) | r.x;
    const bool emit = (uint32_t)(x >> 32) + (cfreq << 15) >= 0x80000000u;
    sink.put_if(emit, (uint32_t)x);
    x = emit ? (x >> 32) : x;
    const uint64_t q = __umul64hi(x, rcp) >> shift;
    x = x + r.z + q * (uint64_t)cfreq;
  };
  auto code_escape = [&](long i) {
    const uint32_t raw = (uint32_t)ap[(size_t)i * B];
    int nbyp = 0;
    while (nbyp < 8 && (raw >> (nbyp * 4)) != 0) ++nbyp;
    for (int j = nbyp - 1; j >= 0; --j) put_bits4(x, sink, (raw >> (j * 4)) & 15u);
    put_bits4(x, sink, (uint32_t)nbyp);
  };
  // batch t from `cur`: ONE basic block of eight chained symbols when no lane of the wave carries an escape
  auto code_batch = [&](const u32x4 (&cur)[ENC_BATCH], long t) {
    const long i1 = n - t * ENC_BATCH;  // symbols i1 - 1 ... i1 - ENC_BATCH
    uint32_t flags = 0;
#pragma unroll
    for (int k = 0; k < ENC_BATCH; ++k) flags |= cur[k].w;
    if (__builtin_expect(i1 >= ENC_BATCH && !__any((flags & REC_ESCAPE) != 0), 1)) {
#pragma unroll
      for (int k = 0; k < ENC_BATCH; ++k) code_symbol(cur[k]);
      return;
    }
#pragma unroll
    for (int k = 0; k < ENC_BATCH; ++k) {
      if (i1 - 1 - k < 0) break;
      if ((cur[k].w & REC_ESCAPE) != 0) code_escape(i1 - 1 - k);
      code_symbol(cur[k]);
    }
  };
  // prologue: four batches requested, landed before the loop (the first pass has fewer operations behind its requests
  // than the loop's counted wait assumes)
  static_for<ENC_AHEAD>([&](auto jc) { request(jc, (long)decltype(jc)::value); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // whole groups of four FULL batches: counted waits; the ragged end (the last group, whose last batch may be short and
  // issue fewer stores than the count assumes) waits for everything
  const long full_groups = (n / ENC_BATCH) / ENC_AHEAD;
  long t = 0;
  for (long g = 0; g < full_groups; ++g) {
    static_for<ENC_AHEAD>([&](auto jc) {
      u32x4 cur[ENC_BATCH];
      landed(jc, std::true_type{}, cur);
      code_batch(cur, t);
      request(jc, t + ENC_AHEAD);
      ++t;
    });
  }
  // the requests that ran past the front of the stream are still in flight: every slot lands before its tail batch is coded
  static_for<ENC_AHEAD>([&](auto jc) {
    u32x4 cur[ENC_BATCH];
    landed(jc, std::false_type{}, cur);
    if (t < nbat) code_batch(cur, t);
    ++t;
  });
  if (live) {
    sink.put((uint32_t)(x >> 32));
    sink.put((uint32_t)x);
    const uint32_t row = sink.off / sink.stride;
    nwords[b] = cap_words - (int)row;
    if (row == 0) atomicOr(status, 1);
  }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rans_encode_records_kernel(const uint4 *__restrict__ rec, const int32_t *__restrict__ aux,
                                                                         long n, uint32_t *__restrict__ words, int cap_words,
                                                                         int32_t *__restrict__ nwords, int32_t *__restrict__ status,
                                                                         int B) {
  extern __shared__ __attribute__((aligned(16))) uint4 s_recs[];  // [WAVES][ENC_DEPTH][ENC_BATCH][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_b0 = (blockIdx.x * WAVES + wave) * 64;
  if (wave_b0 >= B) return;  // no barrier below
  const bool live = wave_b0 + lane < B;
  const int b = live ? wave_b0 + lane : B - 1;  // idle lanes shadow the last stream; their stores are masked off below
  uint4 *ring = s_recs + wave * (ENC_DEPTH * ENC_BATCH * 64);
  const uint4 *rp = rec + b;
  const int32_t *ap = aux + b;
  const long nbat = (n + ENC_BATCH - 1) / ENC_BATCH;
  auto request = [&](long t) {
#pragma unroll
    for (int k = 0; k < ENC_BATCH; ++k) {
      long i = n - 1 - t * ENC_BATCH - k;
      i = i < 0 ? 0 : i;  // past the front of the stream (last batch only): a harmless re-read, never coded
      dma16(rp + (size_t)i * B, ring + ((t & (ENC_DEPTH - 1)) * ENC_BATCH + k) * 64);
    }
  };
  // (the scratch of a launch stays below 4 GB: checked by the launcher)
  WordSink sink{words, ((uint32_t)cap_words * (uint32_t)B + (uint32_t)b) * 4u, (uint32_t)B * 4u};
  uint64_t x = RANS_L;
  for (int t = 0; t < ENC_AHEAD; ++t)
    if (t < nbat) request(t);
  for (long t = 0; t < nbat; ++t) {
    // operations issued after batch t's request: the stores of min(t, ENC_AHEAD) batches and the requests of the batches
    // up to t + ENC_AHEAD - 1 that exist - ENC_BATCH apiece (escapes only add stores): all of them may stay in flight
    {
      const long later = (t < ENC_AHEAD ? t : ENC_AHEAD) + (nbat - 1 - t < ENC_AHEAD - 1 ? nbat - 1 - t : ENC_AHEAD - 1);
      static_assert(ENC_BATCH == 8 && ENC_AHEAD == 4, "the counted waits below are written for 8 operations per batch, up to 7 batches");
      switch (later) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
      }
    }
    uint4 cur[ENC_BATCH];
#pragma unroll
    for (int k = 0; k < ENC_BATCH; ++k) cur[k] = ring[((t & (ENC_DEPTH - 1)) * ENC_BATCH + k) * 64 + lane];
    if (t + ENC_AHEAD < nbat) request(t + ENC_AHEAD);
    const long i1 = n - t * ENC_BATCH;  // symbols i1 - 1 ... i1 - ENC_BATCH
    // one symbol of the chain: renormalise (one word out when the state would overflow), divide by the reciprocal, add
    auto code_symbol = [&](const uint4 r) {
      const uint32_t cfreq = r.w & 0xFFFFu;  // 2^16 - freq
      const uint32_t shift = (r.w >> 16) & 0x7FFFu;
      const uint64_t rcp = ((uint64_t)r.y << 32) | r.x;
      // x >= freq << 47  <=>  hi(x) >= (2^16 - cfreq) << 15  <=>  hi(x) + (cfreq << 15) >= 2^31   (hi(x) < 2^31)
      const bool emit = (uint32_t)(x >> 32) + (cfreq << 15) >= 0x80000000u;
      sink.put_if(emit, (uint32_t)x);
      x = emit ? (x >> 32) : x;
      const uint64_t q = __umul64hi(x, rcp) >> shift;
      x = x + r.z + q * (uint64_t)cfreq;
    };
    auto code_escape = [&](const uint4 r, long i) {  // a value outside its row's range: bypass nibbles in front of the symbol
      const uint32_t raw = (uint32_t)ap[(size_t)i * B];
      int nbyp = 0;
      while (nbyp < 8 && (raw >> (nbyp * 4)) != 0) ++nbyp;
      // coding order is [symbol, count nibble, raw nibbles low -> high]; emitted reversed
      for (int j = nbyp - 1; j >= 0; --j) put_bits4(x, sink, (raw >> (j * 4)) & 15u);
      put_bits4(x, sink, (uint32_t)nbyp);  // nbyp <= 8 < 15: a single count nibble
    };
    auto code_batch = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
      // the escape flags of the whole batch are known up front (they ride in the records): a batch without any - nearly
      // all of them - is ONE basic block of ENC_BATCH chained symbols, so the scheduler can fill the 8-cycle bubbles
      // between dependent instructions with the neighbouring symbols' bookkeeping instead of stopping at a branch per symbol
      uint32_t flags = 0;
#pragma unroll
      for (int k = 0; k < ENC_BATCH; ++k) flags |= cur[k].w;
      if (FULL && __builtin_expect(!__any((flags & REC_ESCAPE) != 0), 1)) {
#pragma unroll
        for (int k = 0; k < ENC_BATCH; ++k) code_symbol(cur[k]);
        return;
      }
#pragma unroll
      for (int k = 0; k < ENC_BATCH; ++k) {
        if (!FULL && i1 - 1 - k < 0) break;
        if ((cur[k].w & REC_ESCAPE) != 0) code_escape(cur[k], i1 - 1 - k);
        code_symbol(cur[k]);
      }
    };
    if (live) {
      if (__builtin_expect(i1 >= ENC_BATCH, 1)) code_batch(std::true_type{});
      else code_batch(std::false_type{});
    }
  }
  if (live) {
    sink.put((uint32_t)(x >> 32));
    sink.put((uint32_t)x);
    // rows used = cap - (next free row); a stream that reached the dump row is reported as overflowed (it may have
    // fitted exactly: the caller's retry with the worst-case capacity settles that)
    const uint32_t row = sink.off / sink.stride;
    nwords[b] = cap_words - (int)row;
    if (row == 0) atomicOr(status, 1);
  }
}

// ---------------------------------------------------------------------------------------------- decode: prepare
// scales -> row byte per symbol in granules of 16 positions: idx16[block][stream][16].  block = 64 streams x 64 positions.
__global__ __launch_bounds__(256) void gc_decode_prepare_kernel(const float *__restrict__ scales, const float *__restrict__ table,
                                                               int levels, float bound, uint4 *__restrict__ idx16,
                                                               unsigned int *__restrict__ row_hist, int B, long n) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[64][80];
  __shared__ unsigned int s_hist[256];
  if (row_hist) s_hist[threadIdx.x] = 0;
  if (row_hist) __syncthreads();
  const long i0 = (long)blockIdx.x * 64;
  const int b0 = blockIdx.y * 64;
  {
    // a thread owns 4 consecutive positions of 4 streams; the 16 scales are loaded before the first is looked at
    const int q = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    const long i = i0 + 4 * q;
    const bool vec = (n & 3) == 0 && (reinterpret_cast<uintptr_t>(scales) & 15) == 0;
    float sv[4][4];
    bool live[4][4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int b = b0 + r0 + 16 * h;
      const size_t at = (size_t)b * n + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        live[h][j] = b < B && i + j < n;
        sv[h][j] = 0.f;
      }
      if (vec) {
        if (live[h][0]) {
          const float4 s4 = *reinterpret_cast<const float4 *>(scales + at);
          sv[h][0] = s4.x; sv[h][1] = s4.y; sv[h][2] = s4.z; sv[h][3] = s4.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (live[h][j]) sv[h][j] = scales[at + j];
      }
    }
    int c[4][4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sv[h][j] = fmaxf(sv[h][j], bound);
        c[h][j] = levels - 1;
      }
    for (int t = 0; t < levels - 1; ++t) {
      const float tv = table[t];
#pragma unroll
      for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[h][j] -= (sv[h][j] <= tv) ? 1 : 0;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = r0 + 16 * h;
      uint32_t packed = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = live[h][j] ? c[h][j] : 0;
        if (row_hist && live[h][j] && (((4 * q + j) ^ r) & 7) == 0) atomicAdd(&s_hist[cc], 1u);  // a 1-in-8 sample of the rows in use
        packed |= (uint32_t)cc << (8 * j);
      }
      *reinterpret_cast<uint32_t *>(&tile[r][4 * q]) = packed;
    }
  }
  __syncthreads();
  if (row_hist && s_hist[threadIdx.x]) atomicAdd(&row_hist[threadIdx.x], s_hist[threadIdx.x]);
  {
    const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int b = b0 + r;
    const long blk = i0 / 16 + q;
    if (b < B && blk * 16 < n) idx16[(size_t)blk * B + b] = *reinterpret_cast<const uint4 *>(&tile[r][q * 16]);
  }
}

// ---------------------------------------------------------------------------------------------- decode: serial part
// Per-lane ring of the next stream words in LDS ([slot][lane]); see rans.hip RingSource - same discipline, half the depth.
struct RingSource {
  const uint32_t *p;
  uint32_t *ring;
  int nw, rd, filled, last;
  bool over;
  __device__ __forceinline__ void init(const uint32_t *ptr, int n, uint32_t *lane_ring, const uint32_t *fallback) {
    p = n > 0 ? ptr : fallback; nw = n; last = n > 0 ? n - 1 : 0; ring = lane_ring; rd = 0; over = false; filled = 0;
    top_up();
    top_up();
  }
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

__device__ __forceinline__ uint32_t get_bits4(uint64_t &x, RingSource &src) {
  const uint32_t val = (uint32_t)(x & 15u);
  x >>= 4;
  if (x < RANS_L) x = (x << 32) | src.next();
  return val;
}

typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2_t lds_cu32x2;
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rans_decode_image_kernel(const uint8_t *__restrict__ in,
                                                                       const int64_t *__restrict__ byte_off,
                                                                       const uint4 *__restrict__ idx16, int rows_shared, long n,
                                                                       const uint4 *__restrict__ image, int image_bytes,
                                                                       int off_meta, int off_rec, int off_cdf,
                                                                       int32_t *__restrict__ symbols, long ssb, long ssi,
                                                                       int32_t *__restrict__ status, int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *s_ring = reinterpret_cast<uint32_t *>(smem) + wave * (RING * 64);
  uint4 *s_idx = reinterpret_cast<uint4 *>(smem + WAVES * RING * 64 * 4) + wave * (IDEPTH * 64);
  unsigned char *s_img = smem + WAVES * (RING * 64 * 4 + IDEPTH * 64 * 16);
  for (int e = tid; e < image_bytes / 16; e += 64 * WAVES) reinterpret_cast<uint4 *>(s_img)[e] = image[e];
  __syncthreads();
  const int wave_b0 = (blockIdx.x * WAVES + wave) * 64;
  if (wave_b0 >= B) return;  // (no barrier below) a wave without streams would also break the store count the waits rely on
  const ImageMeta *s_meta = reinterpret_cast<const ImageMeta *>(s_img + off_meta);
  // absolute LDS addresses for the speculative fast path (the kernel has no static LDS: the dynamic block starts at 0; the
  // slow path and everything rare keep their pointers)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;
  if (lds0 != 0) {  // (never taken)
    if (tid == 0) atomicOr(status, 4);
    return;
  }
  const uint32_t ring_addr = (uint32_t)wave * (RING * 256) + (uint32_t)lane * 4u;
  const uint32_t img_addr = (uint32_t)WAVES * (RING * 64 * 4 + IDEPTH * 64 * 16);
  uint64_t rans_l = RANS_L;
  asm volatile("" : "+s"(rans_l));  // (opaque: as a literal the 64-bit compare becomes mask + compare-with-zero)
  const uint2 *s_rec = reinterpret_cast<const uint2 *>(s_img + off_rec);
  const uint16_t *s_cdf = reinterpret_cast<const uint16_t *>(s_img + off_cdf);
  const bool live = wave_b0 + lane < B;
  const int b = live ? wave_b0 + lane : B - 1;  // idle lanes shadow the last stream; they never store
  RingSource src;
  src.init(reinterpret_cast<const uint32_t *>(in + byte_off[b]), (int)((byte_off[b + 1] - byte_off[b]) / 4), s_ring + lane,
           reinterpret_cast<const uint32_t *>(byte_off));
  uint64_t x = (uint64_t)src.next();
  x |= (uint64_t)src.next() << 32;
  int32_t *sp = symbols + (size_t)b * ssb;
  const long nblk = (n + SYM_BLK - 1) / SYM_BLK;
  // granule of block j at ip[j * gstride]; rows_shared: every stream walks the same rows (one granule per block: the
  // entropy bottleneck's channel pattern), all lanes then fetch the same 16 bytes
  const uint4 *ip = idx16 + (rows_shared ? 0 : b);
  const size_t gstride = rows_shared ? 1 : (size_t)B;
  // Row-index granules travel global -> LDS two blocks ahead.  vmcnt discipline of a block j (all vector-memory
  // operations of the loop are issued by this code, in this order):
  //     wait for granule j | read it | request granule j + 2 | SYM_BLK symbols | SYM_BLK stores
  // Granule j was requested in block j - 2; issued after it: that block's stores, granule j + 1's request, block
  // j - 1's stores.  `s_waitcnt vmcnt(SYM_BLK)` leaves exactly the youngest SYM_BLK operations - block j - 1's stores,
  // every block but the last stores all SYM_BLK - in flight and so guarantees granules j and j + 1 have landed, without
  // waiting for stores issued a moment ago.  (A ring refill drains everything; that only makes the counts smaller.)
  // One symbol: bucket -> record (ONE 8-byte LDS read) -> three selects -> multiply-add -> renormalise, written on
  // 32-bit halves in the order the dependence chain runs.  Everything a rare event needs (bounded search in the row,
  // escape nibbles, ring refill) sits behind ONE wave-uniform branch.  `pack` / `offset`: the row's ImageMeta head.
#ifdef LICOS_GC_STAMPS
  long dbg_slow_n = 0, dbg_slow_cyc = 0, dbg_refill_cyc = 0, dbg_wait_cyc = 0, dbg_rb_n = 0, dbg_rb_cyc = 0, dbg_meta_cyc = 0, dbg_fast_cyc = 0;
  const long dbg_t0 = clock64();
#endif
  auto step = [&](uint32_t row, uint32_t pack, int32_t offset) -> int32_t {
    const uint32_t x_lo = (uint32_t)x, x_hi = (uint32_t)(x >> 32);
    const uint32_t cf = x_lo & 0xFFFFu;
    const uint2 r = *reinterpret_cast<const uint2 *>(s_img + (pack >> IMAGE_PACK_SHIFT) + ((cf >> (pack & 31u)) << 3));
    // the next stream word is read before it is known to be needed: its LDS latency runs beside the lookup's
    const uint32_t w_next = src.peek();
    int s;
    uint32_t off, freq;
    bool miss;
    image_pair(r, cf, s, off, freq, miss);
    int32_t value;
    // x = freq * (x >> 16) + (cf - start), then renormalise from the word read above
    auto update = [&]() {
      const uint32_t xs_lo = __builtin_amdgcn_alignbit(x_hi, x_lo, 16), xs_hi = x_hi >> 16;
      uint64_t nx = (uint64_t)freq * xs_lo + off;
      nx += (uint64_t)(freq * xs_hi) << 32;  // freq <= 2^16, xs_hi < 2^15
      const bool need = nx < RANS_L;
      x = need ? ((nx << 32) | w_next) : nx;
      src.rd += need ? 1 : 0;
    };
    if (__builtin_expect(__any(miss), 0)) {  // uniform and rare: some stream's value lies outside its bucket's pair of symbols
#ifdef LICOS_GC_STAMPS
      const long st0 = clock64();
      ++dbg_slow_n;
#endif
      bool escaped = false;
      if (miss) {
        const ImageMeta m = s_meta[row];
        image_search(m, s_rec, s_cdf, r, cf, s, off, freq);
        escaped = s == (int)(m.cdf_base_max >> 16);
      }
      update();
      value = s;
      if (escaped) {  // the row's escape symbol: the value follows as bypass nibbles
        const int max_value = s;
        uint32_t val = get_bits4(x, src);
        int nbyp = (int)val;
        while (val == 15u && nbyp < 64) { val = get_bits4(x, src); nbyp += (int)val; }
        uint32_t raw = 0;
        for (int t = 0; t < nbyp; ++t) {
          const uint32_t nib = get_bits4(x, src);
          if (t < 8) raw |= nib << (t * 4);
        }
        value = (int32_t)(raw >> 1);
        value = (raw & 1u) ? -value - 1 : value + max_value;
      }
      src.refill_if_low();  // the escape path may have drained several words
#ifdef LICOS_GC_STAMPS
      dbg_slow_cyc += clock64() - st0;
#endif
    } else {
      update();
      value = s;
    }
    return value + offset;
  };

  const long nfull = n / SYM_BLK;  // whole blocks; a ragged tail is decoded symbol by symbol at the end
  int32_t *op = sp;                // where the next symbol of this stream goes (a running pointer: one 64-bit add per store)
  dma16(ip, s_idx);
  if (nblk > 1) dma16(ip + gstride, s_idx + 64);
  for (long j = 0; j < nfull; ++j) {
#ifdef LICOS_GC_STAMPS
    const long sr0 = clock64();
#endif
    src.refill_if_low();
#ifdef LICOS_GC_STAMPS
    const long sr1 = clock64();
    dbg_refill_cyc += sr1 - sr0;
#endif
    if (j == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#ifdef LICOS_GC_STAMPS
    dbg_wait_cyc += clock64() - sr1;
#endif
    static_assert(SYM_BLK == 16, "the counted wait above is written for 16 stores per block");
    const uint4 g = s_idx[(j & (IDEPTH - 1)) * 64 + lane];
    if (j + 2 < nblk) dma16(ip + (size_t)(j + 2) * gstride, s_idx + ((j + 2) & (IDEPTH - 1)) * 64);
    const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
    // The rows' metadata depends on the row bytes only: a quad's four reads are issued one quad AHEAD, so their LDS round
    // trip (52 cycles per symbol when taken at the head of each quad: in-kernel stamps) runs under the previous quad's chain.
    uint2 mm[2][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mm[0][k] = *reinterpret_cast<const uint2 *>(s_meta + ((gw[0] >> (k * 8)) & 0xFFu));
#pragma unroll
    for (int q = 0; q < SYM_BLK / 4; ++q) {
      if (q + 1 < SYM_BLK / 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) mm[(q + 1) & 1][k] = *reinterpret_cast<const uint2 *>(s_meta + ((gw[q + 1] >> (k * 8)) & 0xFFu));
      }
      int32_t v[4];
      // LICOS_GC_SPEC symbols at a time run SPECULATIVELY as one basic block - lookup, selects, multiply-add,
      // renormalise, no branch: a branch per symbol costs ~120 of a step's ~290 cycles (in-kernel stamps: 170 cycles per
      // symbol for the branch-free block).  A lane whose value fell outside its bucket's pair carries a wrong state from
      // there on - harmlessly: every address it forms stays inside its tables - and is flagged; if any lane of the wave
      // was, the state is restored and the group is redone symbol by symbol through the full step.  Pays once misses are
      // rare (usage-weighted image: GaussianConditional.note_row_usage).
#ifndef LICOS_GC_SPEC
#define LICOS_GC_SPEC 2
#endif
      static_assert(LICOS_GC_SPEC == 0 || LICOS_GC_SPEC == 2 || LICOS_GC_SPEC == 4, "group of 0 (off), 2 or 4 symbols");
#pragma unroll
      for (int k0 = 0; k0 < 4; k0 += (LICOS_GC_SPEC ? LICOS_GC_SPEC : 4)) {
        if (LICOS_GC_SPEC == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = step((gw[q] >> (k * 8)) & 0xFFu, mm[q & 1][k].x, (int32_t)mm[q & 1][k].y);
          continue;
        }
        const uint64_t x0 = x;
        const int rd0 = src.rd;
        bool missed = false;
#pragma unroll
        for (int k = k0; k < k0 + LICOS_GC_SPEC; ++k) {
          // (round 5, by instruction count - a lone wave pays ~7 cycles for each, DESIGN 6.3: table and ring through absolute
          // LDS addresses - no base to add, the ring's wrap and base in one v_and_or -, the 15 bits above x >> 16's low word
          // through a full-rate 24-bit multiply-add instead of a second 64-bit one, the compare against an opaque 2^31)
          const uint32_t pack = mm[q & 1][k].x;
          const uint32_t x_lo = (uint32_t)x, x_hi = (uint32_t)(x >> 32);
          const uint32_t cf = x_lo & 0xFFFFu;
          const u32x2_t rr = *(lds_cu32x2 *)(uintptr_t)(img_addr + (pack >> IMAGE_PACK_SHIFT) + ((cf >> (pack & 31u)) << 3));
          const uint2 r = make_uint2(rr.x, rr.y);
          const uint32_t w_next = *(lds_cu32 *)(uintptr_t)((((uint32_t)src.rd << 8) & (uint32_t)((RING - 1) << 8)) | ring_addr);
          int s;
          uint32_t off, freq;
          bool miss;
          image_pair(r, cf, s, off, freq, miss);
          missed = missed || miss;
          const uint32_t xs_lo = __builtin_amdgcn_alignbit(x_hi, x_lo, 16), xs_hi = x_hi >> 16;
          const uint64_t p = (uint64_t)freq * xs_lo + off;
          const uint32_t plo = (uint32_t)p, nhi = __umul24(xs_hi, freq & 0xFFFFFFu) + (uint32_t)(p >> 32);  // freq <= 2^16, xs_hi < 2^15
          const bool need = (((uint64_t)nhi << 32) | plo) < rans_l;
          x = need ? (((uint64_t)plo << 32) | w_next) : (((uint64_t)nhi << 32) | plo);
          src.rd += need ? 1 : 0;
          v[k] = s + (int32_t)mm[q & 1][k].y;
        }
        if (__builtin_expect(__any(missed), 0)) {
          x = x0;
          src.rd = rd0;
#pragma unroll
          for (int k = k0; k < k0 + LICOS_GC_SPEC; ++k) v[k] = step((gw[q] >> (k * 8)) & 0xFFu, mm[q & 1][k].x, (int32_t)mm[q & 1][k].y);
        }
      }
      if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          asm volatile("global_store_dword %0, %1, off" ::"v"(op), "v"(v[k]) : "memory");
          op += ssi;
        }
      }
    }
  }
  if (nfull < nblk) {  // the last, ragged block: nothing counts on its stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    src.refill_if_low();
    const uint8_t *gb = reinterpret_cast<const uint8_t *>(s_idx + (nfull & (IDEPTH - 1)) * 64 + lane);
    for (long i = nfull * SYM_BLK; i < n; ++i) {
      const uint32_t row = gb[i - nfull * SYM_BLK];
      const uint2 mm = *reinterpret_cast<const uint2 *>(s_meta + row);
      const int32_t v = step(row, mm.x, (int32_t)mm.y);
      if (live) sp[(size_t)i * ssi] = v;
    }
  }
#ifdef LICOS_GC_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    long *d = reinterpret_cast<long *>(status + 2);
    d[0] = clock64() - dbg_t0; d[1] = dbg_slow_n; d[2] = dbg_slow_cyc; d[3] = dbg_refill_cyc; d[4] = dbg_wait_cyc; d[5] = dbg_rb_n; d[6] = dbg_rb_cyc; d[7] = dbg_meta_cyc; d[8] = dbg_fast_cyc;
  }
#endif
  if (live && (src.over || src.rd > src.nw)) atomicOr(status, 1);  // words past a stream's end read as zeros; flagged here
}

}  // namespace gc
}  // namespace licos

using namespace licos;
using namespace licos::gc;

// ---------------------------------------------------------------------------------------------- host: decoder image
namespace {

// Mass of a row's 2^k buckets that the best adjacent-symbol pair per bucket does not cover (of 65536), and optionally
// the records themselves.  c[0 .. len-1], c[len-1] == 65536, symbols 0 .. len-2; the last symbol E (the escape) is
// never part of a pair: its values always take the miss path.
double row_records(const int32_t *c, int len, int k, uint2 *out) {
  const int shift = 16 - k, nb = 1 << k, width = 1 << shift;
  const int E = len - 2;
  long bad = 0;
  int f = 0;
  for (int j = 0; j < nb; ++j) {
    const int key0 = j << shift, key1 = key0 + width;  // [key0, key1)
    while (c[f + 1] <= key0) ++f;                       // f: symbol holding key0
    int l = f;
    while (c[l + 1] < key1) ++l;                        // l: symbol holding key1 - 1
    int best = f;
    long best_cov = 0;
    uint32_t f0 = 0, f1 = 0;
    if (f < E) {
      best_cov = -1;
      for (int s = f; s <= std::min(std::max(f, l - 1), E - 1); ++s) {
        const int hi = c[std::min(s + 2, E)];  // the pair ends where the escape symbol starts
        const long cov = (long)std::min(hi, key1) - std::max(c[s], key0);
        if (cov > best_cov) { best_cov = cov; best = s; }
      }
      f0 = (uint32_t)(c[best + 1] - c[best]);
      f1 = best + 1 < E ? (uint32_t)(c[best + 2] - c[best + 1]) : 0u;
    }
    bad += width - best_cov;
    if (out) out[j] = make_uint2(((uint32_t)c[best] & 0xFFFFu) | (f0 << 16), (f1 & 0xFFFFu) | ((uint32_t)best << 16));
  }
  return (double)bad / 65536.0;
}

constexpr int K_MIN = 2, K_MAX = 12;

}  // namespace

extern "C" {

int licos_rans_image_build(const int32_t *cdf, const int32_t *cdf_len, const int32_t *offset, int rows, int stride,
                           const float *row_weight, long budget_bytes, void *out, long *out_bytes) {
  LICOS_REQUIRE(cdf && cdf_len && offset && out && out_bytes && rows > 0 && rows <= 256 && stride > 1, "rans_image_build: bad arguments");
  long n_cdf = 0;
  for (int r = 0; r < rows; ++r) {
    const int len = cdf_len[r];
    LICOS_REQUIRE(len >= 3 && len <= stride && len - 2 <= 65535, "rans_image_build: row %d has cdf length %d", r, len);
    const int32_t *c = cdf + (size_t)r * stride;
    LICOS_REQUIRE(c[0] == 0 && c[len - 1] == 65536, "rans_image_build: row %d does not span [0, 65536]", r);
    for (int s = 0; s + 1 < len; ++s) LICOS_REQUIRE(c[s + 1] > c[s], "rans_image_build: row %d symbol %d has no mass", r, s);
    n_cdf += len - 1;
  }
  LICOS_REQUIRE(n_cdf <= 65535, "rans_image_build: %ld cdf entries exceed the 16-bit row bases", n_cdf);
  auto align16 = [](long v) { return (v + 15) & ~15L; };
  const long off_meta = align16(sizeof(ImageHeader));
  const long off_rec = off_meta + align16((long)rows * sizeof(ImageMeta));
  const long cdf_bytes = align16(n_cdf * 2 + 2);
  const long rec_budget = (budget_bytes - off_rec - cdf_bytes) / 8;
  if (rec_budget < (long)rows << K_MIN)
    return fail(LICOS_EINVAL, "rans_image_build: %ld bytes cannot hold %d rows (%ld cdf entries)", budget_bytes, rows, n_cdf);
  // greedy over the uncovered mass bad[r][k] of every row at every bucket count: all rows start at 2^K_MIN buckets;
  // the jump k -> k' (any k' > k: the mass is not monotonic in k) that removes the most weighted mass per added record
  // goes first, until the budget is spent or nothing is left to gain
  std::vector<int> k(rows, K_MIN);
  std::vector<double> bad((size_t)rows * (K_MAX + 1), 0.0);
  for (int r = 0; r < rows; ++r)
    for (int kk = K_MIN; kk <= K_MAX; ++kk) bad[(size_t)r * (K_MAX + 1) + kk] = row_records(cdf + (size_t)r * stride, cdf_len[r], kk, nullptr);
  long used = (long)rows << K_MIN;
  for (;;) {
    int pick = -1, pick_k = 0;
    double best = 0.0;
    for (int r = 0; r < rows; ++r) {
      const double w = row_weight ? (double)row_weight[r] : 1.0;
      const double *br = &bad[(size_t)r * (K_MAX + 1)];
      for (int kk = k[r] + 1; kk <= K_MAX; ++kk) {
        const long add = (1L << kk) - (1L << k[r]);
        if (used + add > rec_budget) break;
        const double gain = w * (br[k[r]] - br[kk]) / (double)add;
        if (gain > best) { best = gain; pick = r; pick_k = kk; }
      }
    }
    if (pick < 0) break;
    used += (1L << pick_k) - (1L << k[pick]);
    k[pick] = pick_k;
  }
  const long off_cdf = off_rec + align16(used * 8);
  const long total = off_cdf + cdf_bytes;
  LICOS_REQUIRE(total <= budget_bytes, "rans_image_build: internal size error");
  unsigned char *blob = static_cast<unsigned char *>(out);
  std::memset(blob, 0, (size_t)total);
  ImageHeader h{IMAGE_MAGIC, (uint32_t)rows, (uint32_t)used, (uint32_t)n_cdf, (uint32_t)off_meta, (uint32_t)off_rec, (uint32_t)off_cdf, (uint32_t)total};
  std::memcpy(blob, &h, sizeof(h));
  ImageMeta *meta = reinterpret_cast<ImageMeta *>(blob + off_meta);
  uint2 *rec = reinterpret_cast<uint2 *>(blob + off_rec);
  uint16_t *c16 = reinterpret_cast<uint16_t *>(blob + off_cdf);
  long rec_at = 0, cdf_at = 0;
  for (int r = 0; r < rows; ++r) {
    const int len = cdf_len[r];
    const int32_t *c = cdf + (size_t)r * stride;
    meta[r] = ImageMeta{(uint32_t)((off_rec + rec_at * 8) << IMAGE_PACK_SHIFT) | (uint32_t)(16 - k[r]), offset[r],
                        (uint32_t)cdf_at | ((uint32_t)(len - 2) << 16), (uint32_t)rec_at};
    row_records(c, len, k[r], rec + rec_at);
    for (int s = 0; s + 1 < len; ++s) c16[cdf_at + s] = (uint16_t)c[s];
    rec_at += 1L << k[r];
    cdf_at += len - 1;
  }
  *out_bytes = total;
  return LICOS_OK;
}

int licos_rans_image_lookup(const void *image, int row, int cf, int32_t *out3) {
  LICOS_REQUIRE(image && out3, "rans_image_lookup: NULL argument");
  const unsigned char *blob = static_cast<const unsigned char *>(image);
  ImageHeader h;
  std::memcpy(&h, blob, sizeof(h));
  LICOS_REQUIRE(h.magic == IMAGE_MAGIC && row >= 0 && (uint32_t)row < h.rows && cf >= 0 && cf < 65536, "rans_image_lookup: bad image, row or value");
  const ImageMeta m = reinterpret_cast<const ImageMeta *>(blob + h.off_meta)[row];
  int s;
  uint32_t off, freq;
  bool fb;
  image_lookup(m, reinterpret_cast<const uint2 *>(blob + h.off_rec), reinterpret_cast<const uint16_t *>(blob + h.off_cdf), (uint32_t)cf, s, off, freq, fb);
  out3[0] = s;
  out3[1] = (int32_t)((uint32_t)cf - off);
  out3[2] = (int32_t)((uint32_t)cf - off + freq);
  return fb ? 1 : 0;
}

int licos_gc_encode_prepare(const float *y, const float *scales, const float *scale_table, int levels, float scale_bound,
                            const void *enc_table, int cdf_stride, const int32_t *cdf_len, const int32_t *offset, void *rec,
                            int32_t *aux, int B, long n, void *stream) {
  LICOS_REQUIRE(y && scales && scale_table && enc_table && cdf_len && offset && rec && aux, "gc_encode_prepare: NULL buffer");
  LICOS_REQUIRE(levels > 0 && levels <= 256 && B > 0 && n > 0 && cdf_stride > 1, "gc_encode_prepare: bad sizes");
  LICOS_REQUIRE((n + 31) / 32 < (1L << 31) && (B + 63) / 64 < 65536, "gc_encode_prepare: too many symbols or streams");
  hipLaunchKernelGGL(gc_encode_prepare_kernel<true>, dim3((unsigned)((n + 31) / 32), (unsigned)((B + 63) / 64)), dim3(256), 0,
                     as_stream(stream), y, scales, scale_table, levels, scale_bound, static_cast<const uint4 *>(enc_table),
                     cdf_stride, cdf_len, offset, static_cast<uint4 *>(rec), aux, B, n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_eb_encode_prepare(const float *y, const float *medians, int C, int plane, const void *enc_table, int cdf_stride,
                            const int32_t *cdf_len, const int32_t *offset, void *rec, int32_t *aux, int B, void *stream) {
  LICOS_REQUIRE(y && medians && enc_table && cdf_len && offset && rec && aux, "eb_encode_prepare: NULL buffer");
  LICOS_REQUIRE(C > 0 && plane > 0 && B > 0 && cdf_stride > 1, "eb_encode_prepare: bad sizes");
  const long n = (long)C * plane;
  LICOS_REQUIRE((n + 31) / 32 < (1L << 31) && (B + 63) / 64 < 65536, "eb_encode_prepare: too many symbols or streams");
  hipLaunchKernelGGL(gc_encode_prepare_kernel<false>, dim3((unsigned)((n + 31) / 32), (unsigned)((B + 63) / 64)), dim3(256), 0,
                     as_stream(stream), y, medians, nullptr, plane, 0.f, static_cast<const uint4 *>(enc_table), cdf_stride, cdf_len,
                     offset, static_cast<uint4 *>(rec), aux, B, n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_rans_encode_records(const void *rec, const int32_t *aux, long n, uint32_t *words, int cap_words, int32_t *nwords,
                              int32_t *status, int B, void *stream) {
  LICOS_REQUIRE(rec && aux && words && nwords && status && B > 0 && n > 0 && cap_words >= 2, "rans_encode_records: bad arguments");
  LICOS_REQUIRE(((long)cap_words + 1) * B * 4 < (1L << 32), "rans_encode_records: scratch of %d x %d words exceeds 4 GB - code fewer streams per launch", cap_words + 1, B);
  // two waves per workgroup: 2048 streams occupy 16 CUs (a CU that holds a coder wave cannot take a workgroup of the
  // 8-wave transform kernels, and the coders run beside the neighbouring chunk's transforms)
  auto launch = [&](auto kern, int waves) -> int {
    const size_t lds = (size_t)waves * ENC_DEPTH * ENC_BATCH * 64 * sizeof(uint4);
    LICOS_ENSURE_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3(cdiv(B, 64 * waves)), dim3(64 * waves), lds, as_stream(stream), static_cast<const uint4 *>(rec), aux, n,
                       words, cap_words, nwords, status, B);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  };
  static const bool regs = [] { const char *e = getenv("LICOS_GC_ENC_REGS"); return !e || atoi(e) != 0; }();  // (A/B: 0 = the LDS-DMA ring)
  if (regs) {
    auto launch_r = [&](auto kern, int waves) -> int {
      hipLaunchKernelGGL(kern, dim3(cdiv(B, 64 * waves)), dim3(64 * waves), 0, as_stream(stream), static_cast<const uint4 *>(rec), aux, n, words,
                         cap_words, nwords, status, B);
      LICOS_LAUNCH_CHECK();
      return LICOS_OK;
    };
    return B > 64 ? launch_r(rans_encode_records_regs_kernel<2>, 2) : launch_r(rans_encode_records_regs_kernel<1>, 1);
  }
  return B > 64 ? launch(rans_encode_records_kernel<2>, 2) : launch(rans_encode_records_kernel<1>, 1);
}

int licos_gc_decode_prepare(const float *scales, const float *scale_table, int levels, float scale_bound, void *idx16,
                            unsigned int *row_hist, int B, long n, void *stream) {
  LICOS_REQUIRE(scales && scale_table && idx16 && levels > 0 && levels <= 256 && B > 0 && n > 0, "gc_decode_prepare: bad arguments");
  LICOS_REQUIRE((n + 63) / 64 < (1L << 31) && (B + 63) / 64 < 65536, "gc_decode_prepare: too many symbols or streams");
  hipLaunchKernelGGL(gc_decode_prepare_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((B + 63) / 64)), dim3(256), 0,
                     as_stream(stream), scales, scale_table, levels, scale_bound, static_cast<uint4 *>(idx16), row_hist, B, n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

long licos_rans_image_budget(int waves) {
  if (waves != 1 && waves != 2) return 0;
  return 160L * 1024 - 256 - (long)waves * (RING * 64 * 4 + IDEPTH * 64 * 16);
}

int licos_rans_decode_image(const uint8_t *in, const int64_t *byte_off, const void *idx16, int rows_shared, long n,
                            const void *image, const void *image_host_header, int32_t *symbols, long sym_stride_b,
                            long sym_stride_i, int32_t *status, int B, void *stream) {
  LICOS_REQUIRE(in && byte_off && idx16 && image && image_host_header && symbols && status && B > 0 && n > 0, "rans_decode_image: bad arguments");
  LICOS_REQUIRE(((uintptr_t)in & 3) == 0 && ((uintptr_t)image & 15) == 0, "rans_decode_image: misaligned input");
  ImageHeader h;
  std::memcpy(&h, image_host_header, sizeof(h));
  LICOS_REQUIRE(h.magic == IMAGE_MAGIC && h.total_bytes % 16 == 0, "rans_decode_image: not a decoder image");
  const int waves = B > 64 ? 2 : 1;
  const size_t fixed = (size_t)waves * (RING * 64 * 4 + IDEPTH * 64 * 16);
  const size_t lds = fixed + h.total_bytes;
  LICOS_REQUIRE(lds <= 160 * 1024, "rans_decode_image: image of %u bytes does not fit beside %d waves", h.total_bytes, waves);
  auto launch = [&](auto kern) -> int {
    LICOS_ENSURE_LDS(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(cdiv(B, 64 * waves)), dim3(64 * waves), lds, as_stream(stream), in, byte_off,
                       static_cast<const uint4 *>(idx16), rows_shared, n, static_cast<const uint4 *>(image), (int)h.total_bytes,
                       (int)h.off_meta, (int)h.off_rec, (int)h.off_cdf, symbols, sym_stride_b, sym_stride_i, status, B);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  };
  return waves == 2 ? launch(rans_decode_image_kernel<2>) : launch(rans_decode_image_kernel<1>);
}

}  // extern "C"
