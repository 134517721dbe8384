// First analysis stage g_a[0] for 1..3 input bands: Conv2d(C -> 128, 5x5, stride 2, padding 2) + GDN, two independent
// 4-wave workgroups per CU.
//
// The stage is a short K loop in front of a long epilogue (GDN as a second MFMA GEMM + ~1000 vector instructions per wave
// and tile) and 4 MiB of stores per 256^2 tile.  The 8-wave kernel it replaces for these shapes (mfma_conv3x3t.hip: 3x3
// stride-1 over the space-to-depth image) keeps ONE workgroup per CU, so the two waves of a SIMD run their K loops together
// and their epilogues together - matrix pipe and vector issue take turns.  Here:
//   * K = (ky; kx, c): a K step is one kernel row, its 16 slots hold the 5 C <= 15 values (kx, c) of that row - 5 steps
//     instead of the 9 of the 3x3 form (whose 16-channel chunk carries 12 live values and whose taps 44 % zeros), 20 KB of
//     weight fragments instead of 36;
//   * the input is an interleaved, zero-bordered fp16 image [H + 4][(W + 4) C (+ pad)] (licos_nchw_f32_to_hwc_pad_f16): the
//     16 values a lane needs for (output pixel, ky) are CONTIGUOUS (8 halfs at offset 2 x C + 8 h of the patch row), patch
//     rows are contiguous 16-byte granules for LDS-DMA, and there is no bounds logic at all;
//   * with 20 + 32 (gamma) + 1 + 2 x 8 KB of LDS a workgroup is 69 KB: TWO 4-wave workgroups share a CU, each with its own
//     schedule - no cross-wave choreography, and one workgroup's store drain and barrier waits are the other's time to issue.
// A workgroup walks a run of 8 x 32 output tiles with the weights, gamma, beta and bias resident; per tile only the 8 KB
// patch arrives (double buffered, requested two tiles ahead); vmcnt discipline as in mfma_conv3x3t.hip / mfma_deconv8.hip.
// Measured per 4096 tiles of 3 x 256^2: 4.13 ms against 5.02 for the 3x3 form; without stores 3.03, without the
// GDN arithmetic 3.63 - the 4 MiB/tile store stream and the epilogue each fill most of the time.
// The default entry point is the IN-PLACE form further down (conv5x5s2_first_raw_kernel: the same K loop and epilogue fed
// from the NCHW fp32 image itself, no layout pass); this form serves widths that are not multiples of 4.
#include <cstdlib>

#include "mfma_deconv8.hpp"

namespace licos {

struct FirstArgs {
  const _Float16 *x;   // [B][H + 4][RSG] interleaved fp16 with a 2-pixel zero border
  const half8 *wp;     // [5 ky][MT][64] A fragments
  const float *bias, *beta;
  const bf16x8 *gamma;
  _Float16 *y_blk;
  int B, H, W, Ho, Wo, Cout, tiles_x, tiles_y, rsg;  // rsg: halfs per padded input row (a multiple of 8)
};

__host__ __device__ constexpr int first_row_halfs(int W, int C) { return ((W + 4) * C + 8 + 7) / 8 * 8; }

template <int C>
struct FirstGeom {
  static constexpr int MT = 4, NT = 2, TH = 8, TW = 32;
  static constexpr int PR = 2 * TH + 3;                               // input rows of a tile
  static constexpr int GR = (2 * (TW - 1) * C + 16 + 7) / 8;          // 16-byte granules per patch row (26 at C = 3)
  static constexpr int PG = PR * GR, PQ = (PG + 63) / 64, PATCH_PAD = PQ * 64, NPP = (PQ + 3) / 4;
  static constexpr int W_GRAN = 5 * MT * 64, GAMMA_GRAN = MT * MT * 2 * 64, NWP = (5 * MT + 3) / 4, NGP = (GAMMA_GRAN / 64 + 3) / 4;
  static_assert(5 * C <= 16, "one kernel row (kx, c) per 16-slot K step");
  static_assert(GR * 8 >= (2 * TW + 3) * C, "a patch row holds every input column of the tile");
};

template <int C, int EPI>
__global__ __launch_bounds__(256, 2) void conv5x5s2_first_kernel(FirstArgs a, int run) {
  using G = FirstGeom<C>;
  constexpr int MT = G::MT, NT = G::NT;
  constexpr bool NORM = (EPI == EPI_GDN);
  constexpr int NSTORE = NT * MT * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_pbuf = reinterpret_cast<half8 *>(smem);  // [2][PATCH_PAD]
  half8 *s_w = s_pbuf + 2 * G::PATCH_PAD;           // [5 MT 64] resident
  float *s_bias = reinterpret_cast<float *>(s_w + G::W_GRAN);
  float *s_beta = s_bias + 32 * MT;
  bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(s_beta + 32 * MT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y, runs = (tiles + run - 1) / run;
  int b, item;
  xcd_work_item(blockIdx.x, a.B, runs, b, item);
  const int t_first = item * run, t_count = (t_first + run <= tiles) ? run : tiles - t_first;

  const unsigned char *xb = reinterpret_cast<const unsigned char *>(a.x) + (size_t)b * (a.H + 4) * a.rsg * 2;
  // my granules of a patch: idx = 64 q + lane -> (patch row, granule) -> byte offset from the tile's first granule
  int p_off[G::NPP];
#pragma unroll
  for (int i = 0; i < G::NPP; ++i) {
    int idx = (wave + 4 * i) * 64 + lane;
    idx = idx < G::PG ? idx : G::PG - 1;  // (the tail of the last piece re-reads the last granule)
    const int prow = idx / G::GR, g = idx - prow * G::GR;
    p_off[i] = prow * a.rsg * 2 + g * 16;
  }
  auto dma_patch = [&](int t, int buf) {
    const int tile = t_first + t;
    const int ty0 = (tile / a.tiles_x) * G::TH, tx0 = (tile % a.tiles_x) * G::TW;
    // padded row of input row 2 ty0 - 2 is 2 ty0, padded column of input column 2 tx0 - 2 is 2 tx0
    const unsigned char *src = xb + ((size_t)(2 * ty0) * a.rsg + (size_t)(2 * tx0) * C) * 2;
#pragma unroll
    for (int i = 0; i < G::NPP; ++i) {
      const int q = wave + 4 * i;
      if (q < G::PQ) glds16(src + p_off[i], s_pbuf + buf * G::PATCH_PAD + q * 64);
    }
  };

  // resident operands + the first patch
#pragma unroll
  for (int i = 0; i < G::NWP; ++i) {
    const int q = wave + 4 * i;
    if (q < 5 * MT) glds16(a.wp + q * 64 + lane, s_w + q * 64);
  }
  if (wave == 0) glds16((lane < 32 || !NORM) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias);
  if (NORM) {
#pragma unroll
    for (int i = 0; i < G::NGP; ++i) {
      const int q = wave + 4 * i;
      if (q < G::GAMMA_GRAN / 64) glds16(a.gamma + q * 64 + lane, s_gamma + q * 64);
    }
  }
  dma_patch(0, 0);

  f32x16 acc[MT][NT];
  auto acc_init = [&]() {  // accumulators start at the bias: register q of tile mt is channel 32mt + (q&3) + 8(q>>2) + 4h
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(s_bias + 32 * mt + 8 * g + 4 * h);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt][4 * g + 0] = bv.x;
          acc[mt][nt][4 * g + 1] = bv.y;
          acc[mt][nt][4 * g + 2] = bv.z;
          acc[mt][nt][4 * g + 3] = bv.w;
        }
      }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  acc_init();
  if (t_count > 1) dma_patch(1, 1);  // tile 1's patch lands under tile 0's MFMAs

  // byte offset of this lane's 8 halfs inside a patch row: pixel column 2 r, slots k = 8 h .. 8 h + 7 of (kx, c)
  const int lane_boff = 4 * r * C + 16 * h;
  const int Cout16 = (a.Cout + 15) >> 4;
  _Float16 *y_img = a.y_blk + (size_t)b * Cout16 * a.Ho * a.Wo * 16;
  bool counted = false;  // the youngest NSTORE operations of this wave are the previous tile's stores
  for (int t = 0; t < t_count; ++t) {
    const int cur = t & 1;
    const unsigned char *pb = reinterpret_cast<const unsigned char *>(s_pbuf + cur * G::PATCH_PAD) + lane_boff;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      half8 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        // (4-byte aligned at odd C: four dword reads)
        const unsigned *p = reinterpret_cast<const unsigned *>(pb + (2 * (wave * NT + nt) + ky) * G::GR * 16);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {p[0], p[1], p[2], p[3]};
        bf[nt] = __builtin_bit_cast(half8, v);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const half8 af = s_w[(ky * MT + mt) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    // the patch of tile t+1 (requested before the previous epilogue, or above) has landed: everything older than this
    // wave's last NSTORE operations is complete
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < t_count) dma_patch(t + 2, cur);  // buffer `cur` is free now
    asm volatile("" ::: "memory");               // the stores below stay behind that request
    const int tile = t_first + t;
    const int ty0 = (tile / a.tiles_x) * G::TH, tx0 = (tile % a.tiles_x) * G::TW;
    long pix[NT];
    bool all_live = a.Cout >= 32 * MT - 15;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = ty0 + wave * NT + nt, ox = tx0 + r;
      pix[nt] = (oy < a.Ho && ox < a.Wo) ? (long)oy * a.Wo + ox : -1;
      all_live = all_live && oy < a.Ho;  // wave-uniform: a live row issues its stores whatever its columns
    }
    tile8_epilogue<MT, NT, EPI>(acc, s_gamma, s_beta, y_img, (size_t)a.Ho * a.Wo, Cout16, pix, lane);
    counted = all_live;
    if (t + 1 < t_count) acc_init();
  }
}

// ---- the same stage reading the NCHW fp32 image in place (no layout pass) ----------------------------------------------
// The fp32 rows of a tile arrive by LDS-DMA as they lie in memory - per band and input row 18 granules of 4 pixels,
// columns 2 tx0 - 4 .. 2 tx0 + 67; with W a multiple of 4 and tile origins multiples of 64 a granule is entirely inside
// the image or entirely outside it, so zero padding is a per-granule choice of source address (16 zero bytes ride at the
// end of the packed weights) - and the workgroup turns them into the interleaved fp16 patch itself, LDS to LDS: a thread
// takes the C bands' granules of a 4-pixel group (16-byte reads) and writes their 4 C interleaved halfs as 2 C dwords,
// one or two groups per tile, between the K loop and the epilogue.  One raw
// buffer and one patch buffer (78 KB per workgroup: still two per CU):
//   K loop (patch) | wait: raw rows of tile t+1 landed | barrier | repack raw -> patch | barrier | request raw rows of
//   tile t+2 | epilogue of tile t, stores
// LICOS_STAMPS (diagnostic builds, never the product): per-phase s_memtime cycles of wave 0, see mfma_first16.hip
#ifdef LICOS_STAMPS
__device__ unsigned long long g_first_stamps[32];
#define F_STAMP(i)                                                     \
  do {                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
    st_acc[i] += now_ - st_prev;                                       \
    st_prev = now_;                                                    \
  } while (0)
#else
#define F_STAMP(i) do {} while (0)
#endif
// LICOS_ABL_FIRST (diagnostic builds): 5 = no LDS-to-LDS repack (timing only: the patch keeps the first tile's values)
#ifndef LICOS_ABL_FIRST
#define LICOS_ABL_FIRST 0
#endif
#ifndef LICOS_FIRST_PRIO_E
#define LICOS_FIRST_PRIO_E 0
#endif

struct FirstRawArgs {
  const float *x;      // NCHW fp32 [B][C][H][W]
  const half8 *wp;     // [5 ky][MT][64] A fragments + one granule of zeros
  const float *bias, *beta;
  const bf16x8 *gamma;
  _Float16 *y_blk;
  int B, H, W, Ho, Wo, Cout, tiles_x, tiles_y;
};

template <int C>
struct FirstRawGeom {
  using G = FirstGeom<C>;
  static constexpr int SG = 18;                       // fp32 granules per (band, input row)
  static constexpr int SGT = C * G::PR * SG;          // granules of a tile's raw rows (1026 at C = 3)
  static constexpr int SQ = (SGT + 63) / 64, S_PAD = SQ * 64, NSP = (SQ + 3) / 4;  // wave-wide DMA pieces; per wave
  static_assert(4 * SG - 2 >= 2 * G::TW + 3, "18 granules cover the 67 input columns of a tile");
};

template <int C, int EPI>
__global__ __launch_bounds__(256, 2) void conv5x5s2_first_raw_kernel(FirstRawArgs a, int run) {
  using G = FirstGeom<C>;
  using R = FirstRawGeom<C>;
  constexpr int MT = G::MT, NT = G::NT;
  constexpr bool NORM = (EPI == EPI_GDN);
  constexpr int NSTORE = NT * MT * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_raw = reinterpret_cast<half8 *>(smem);  // [S_PAD] granules of 4 fp32 pixels
  half8 *s_p = s_raw + R::S_PAD;                    // [PATCH_PAD] the interleaved fp16 patch
  half8 *s_w = s_p + G::PATCH_PAD;                  // [5 MT 64] resident
  float *s_bias = reinterpret_cast<float *>(s_w + G::W_GRAN);
  float *s_beta = s_bias + 32 * MT;
  bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(s_beta + 32 * MT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y, runs = (tiles + run - 1) / run;
  int b, item;
  xcd_work_item(blockIdx.x, a.B, runs, b, item);
  const int t_first = item * run, t_count = (t_first + run <= tiles) ? run : tiles - t_first;
  const float *xb = a.x + (size_t)b * C * a.H * a.W;
  const half8 *zero = a.wp + G::W_GRAN;

  // my granules to request: idx = tid + 256 i -> (band, patch row, granule of 4 columns)
  int g_rel[R::NSP], g_rc[R::NSP];
#pragma unroll
  for (int i = 0; i < R::NSP; ++i) {
    int idx = tid + 256 * i;
    idx = idx < R::SGT ? idx : R::SGT - 1;  // (the tail of the last piece repeats the last granule)
    const int c = idx / (G::PR * R::SG), rem = idx - c * (G::PR * R::SG);
    const int prow = rem / R::SG, g = rem - prow * R::SG;
    g_rel[i] = (c * a.H + prow) * a.W + 4 * g;  // element offset from (row 2 ty0 - 2, column 2 tx0 - 4) of band 0
    g_rc[i] = prow | (4 * g) << 8;
  }
  auto dma_raw = [&](int t) {
    const int tile = t_first + t;
    const int iy0 = 2 * (tile / a.tiles_x) * G::TH - 2, ixb = 2 * (tile % a.tiles_x) * G::TW - 4;
    const float *org = xb + (long)iy0 * a.W + ixb;
#pragma unroll
    for (int i = 0; i < R::NSP; ++i) {
      const int q = wave + 4 * i;
      if (q >= R::SQ) continue;
      const bool ok = (unsigned)(iy0 + (g_rc[i] & 255)) < (unsigned)a.H && (unsigned)(ixb + (g_rc[i] >> 8)) < (unsigned)a.W;
      glds16(ok ? static_cast<const void *>(org + g_rel[i]) : static_cast<const void *>(zero), s_raw + q * 64);
    }
  };
  // my pixel groups to interleave: m = tid + 256 i -> (patch row, granule): the C bands' granules of 4 pixels become 4 C
  // consecutive halfs of the patch row = 2 C dwords at patch column 4 g - 2 (patch column 0 = input column 2 tx0 - 2).  The
  // first granule of a row holds two columns left of the patch (its first C dwords are skipped), the last one a single
  // live column (ceil(C / 2) dwords: at odd C the extra half is a pad slot of the row tail).
  constexpr int NRG = (G::PR * R::SG + 255) / 256;
  int m_src[NRG], m_dst[NRG], m_lo[NRG], m_hi[NRG];
#pragma unroll
  for (int i = 0; i < NRG; ++i) {
    const int m = tid + 256 * i;
    const int prow = m / R::SG, g = m - prow * R::SG;
    m_src[i] = m;                                                  // granule index inside band 0's rows
    m_dst[i] = 2 * (prow * 8 * G::GR + (4 * g - 2) * C);           // byte offset inside the patch
    m_lo[i] = m >= G::PR * R::SG ? 2 * C : (g == 0 ? C : 0);       // first and one-past-last dword written
    m_hi[i] = g == R::SG - 1 ? (C + 1) / 2 : 2 * C;
  }
  auto repack = [&]() {
    unsigned char *pb = reinterpret_cast<unsigned char *>(s_p);
#pragma unroll
    for (int i = 0; i < NRG; ++i) {
      _Float16 hv[4 * C];
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(s_raw + c * (G::PR * R::SG) + (m_src[i] < G::PR * R::SG ? m_src[i] : 0));
        hv[0 * C + c] = (_Float16)v.x;
        hv[1 * C + c] = (_Float16)v.y;
        hv[2 * C + c] = (_Float16)v.z;
        hv[3 * C + c] = (_Float16)v.w;
      }
#pragma unroll
      for (int d = 0; d < 2 * C; ++d) {
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        const half2v pr = {hv[2 * d], hv[2 * d + 1]};
        if (d >= m_lo[i] && d < m_hi[i]) *reinterpret_cast<unsigned *>(pb + m_dst[i] + 4 * d) = __builtin_bit_cast(unsigned, pr);
      }
    }
  };

  // resident operands, the first tile's raw rows; the patch's row tails (never rewritten) start as zeros
#pragma unroll
  for (int i = 0; i < G::NWP; ++i) {
    const int q = wave + 4 * i;
    if (q < 5 * MT) glds16(a.wp + q * 64 + lane, s_w + q * 64);
  }
  if (wave == 0) glds16((lane < 32 || !NORM) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias);
  if (NORM) {
#pragma unroll
    for (int i = 0; i < G::NGP; ++i) {
      const int q = wave + 4 * i;
      if (q < G::GAMMA_GRAN / 64) glds16(a.gamma + q * 64 + lane, s_gamma + q * 64);
    }
  }
  dma_raw(0);
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int e = tid; e < G::PATCH_PAD; e += 256) s_p[e] = zero8;

  f32x16 acc[MT][NT];
  auto acc_init = [&]() {  // accumulators start at the bias: register q of tile mt is channel 32mt + (q&3) + 8(q>>2) + 4h
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(s_bias + 32 * mt + 8 * g + 4 * h);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt][4 * g + 0] = bv.x;
          acc[mt][nt][4 * g + 1] = bv.y;
          acc[mt][nt][4 * g + 2] = bv.z;
          acc[mt][nt][4 * g + 3] = bv.w;
        }
      }
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  repack();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (t_count > 1) dma_raw(1);  // tile 1's rows land under tile 0's MFMAs
  acc_init();

  const int lane_boff = 4 * r * C + 16 * h;
  const int Cout16 = (a.Cout + 15) >> 4;
  _Float16 *y_img = a.y_blk + (size_t)b * Cout16 * a.Ho * a.Wo * 16;
  bool counted = false;  // the youngest NSTORE operations of this wave are the previous tile's stores
#ifdef LICOS_STAMPS
  unsigned long long st_acc[8] = {}, st_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int t = 0; t < t_count; ++t) {
    F_STAMP(7);
    const unsigned char *pb = reinterpret_cast<const unsigned char *>(s_p) + lane_boff;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
      half8 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned *p = reinterpret_cast<const unsigned *>(pb + (2 * (wave * NT + nt) + ky) * G::GR * 16);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = {p[0], p[1], p[2], p[3]};
        bf[nt] = __builtin_bit_cast(half8, v);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const half8 af = s_w[(ky * MT + mt) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    // the raw rows of tile t+1 (requested before the previous epilogue, or above) have landed: everything older than this
    // wave's last NSTORE operations is complete; after the barrier nobody reads the patch of tile t any more
    F_STAMP(0);
    if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    F_STAMP(1);
    __builtin_amdgcn_s_barrier();
    F_STAMP(2);
    if (t + 1 < t_count && LICOS_ABL_FIRST != 5) repack();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    F_STAMP(3);
    __builtin_amdgcn_s_barrier();
    F_STAMP(4);
    if (t + 2 < t_count) dma_raw(t + 2);  // the raw buffer is free now
    asm volatile("" ::: "memory");        // the stores below stay behind that request
    const int tile = t_first + t;
    const int ty0 = (tile / a.tiles_x) * G::TH, tx0 = (tile % a.tiles_x) * G::TW;
    long pix[NT];
    bool all_live = a.Cout >= 32 * MT - 15;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = ty0 + wave * NT + nt, ox = tx0 + r;
      pix[nt] = (oy < a.Ho && ox < a.Wo) ? (long)oy * a.Wo + ox : -1;
      all_live = all_live && oy < a.Ho;
    }
    F_STAMP(5);
#if LICOS_FIRST_PRIO_E
    __builtin_amdgcn_s_setprio(LICOS_FIRST_PRIO_E);  // (A/B) the epilogue's dependent chains ahead of the other workgroup's K loop
#endif
    tile8_epilogue<MT, NT, EPI>(acc, s_gamma, s_beta, y_img, (size_t)a.Ho * a.Wo, Cout16, pix, lane);
#if LICOS_FIRST_PRIO_E
    __builtin_amdgcn_s_setprio(0);
#endif
    F_STAMP(6);
    counted = all_live;
    if (t + 1 < t_count) acc_init();
  }
#ifdef LICOS_STAMPS
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(&g_first_stamps[i], st_acc[i]);
    atomicAdd(&g_first_stamps[8], (unsigned long long)t_count);
  }
#endif
}

// NCHW fp32 -> [B][H + 4][rsg] fp16, value (c, iy, ix) at half (iy + 2) * rsg + (ix + 2) * C + c, zeros elsewhere.  A
// workgroup turns LP_ROWS padded rows at a time: coalesced 4-byte loads per band, the interleave through an LDS image of
// the rows (border and row tail stay zero from the start), coalesced 16-byte stores.
constexpr int LP_ROWS = 4;
template <int C>
__global__ __launch_bounds__(256) void nchw_to_hwc_pad_kernel(const float *__restrict__ x, half8 *__restrict__ out, int H, int W, int rsg,
                                                              long rows_total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16 *s_row = reinterpret_cast<_Float16 *>(smem);  // [LP_ROWS][rsg]
  const int tid = threadIdx.x, gpr = rsg / 8, Hp = H + 4;
  for (int i = tid; i < LP_ROWS * rsg; i += 256) s_row[i] = (_Float16)0.f;
  __syncthreads();
  const long groups = (rows_total + LP_ROWS - 1) / LP_ROWS;
  for (long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long row0 = grp * LP_ROWS;
#pragma unroll
    for (int k = 0; k < LP_ROWS; ++k) {
      const long rowi = row0 + k;
      const int prow = (int)(rowi % Hp), iy = prow - 2;
      const long b = rowi / Hp;
      if (rowi < rows_total && iy >= 0 && iy < H) {  // (workgroup-uniform)
        const float *src = x + ((size_t)b * C * H + iy) * W;
        for (int ix = tid; ix < W; ix += 256) {
#pragma unroll
          for (int c = 0; c < C; ++c) s_row[k * rsg + (ix + 2) * C + c] = (_Float16)src[(size_t)c * H * W + ix];
        }
      }
    }
    __syncthreads();
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < LP_ROWS; ++k) {
      const long rowi = row0 + k;
      const int iy = (int)(rowi % Hp) - 2;
      if (rowi >= rows_total) break;
      const bool data = iy >= 0 && iy < H;
      for (int g = tid; g < gpr; g += 256) out[rowi * gpr + g] = data ? *reinterpret_cast<const half8 *>(s_row + k * rsg + 8 * g) : zero8;
    }
    __syncthreads();
  }
}

// w: Conv2d weight [Cout][C][5][5] fp32 -> A fragments [ky][mt][lane][8]: row = 32 mt + (lane & 31) = output channel,
// k = 8 * (lane >> 5) + e = kx * C + c (k >= 5 C: zero)
__global__ void pack_conv_w_first_kernel(const float *__restrict__ w, int C, int Cout, int MT, _Float16 *__restrict__ out, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long f = i >> 9;
    const int mt = (int)(f % MT), ky = (int)(f / MT);
    const int co = 32 * mt + (lane & 31), k = 8 * (lane >> 5) + e;
    const int kx = k / C, c = k - kx * C;
    float v = 0.f;
    if (co < Cout && k < 5 * C) v = w[(((size_t)co * C + c) * 5 + ky) * 5 + kx];
    out[i] = (_Float16)v;
  }
}

template <int C, int EPI>
static int launch_first(const FirstArgs &a, hipStream_t s) {
  using G = FirstGeom<C>;
  const int tiles = a.tiles_x * a.tiles_y;
  static const int run_max = [] { const char *e = getenv("LICOS_FIRST_RUN"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
  const int run = tiles >= run_max ? run_max : tiles;  // tiles per workgroup (the resident operands are 53 KB per run)
  const size_t lds = (size_t)16 * (2 * G::PATCH_PAD + G::W_GRAN + 16 * G::MT + (EPI == EPI_GDN ? G::GAMMA_GRAN : 0));
  auto kern = conv5x5s2_first_kernel<C, EPI>;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)cdiv(tiles, run) * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "conv5x5s2_first_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a, run);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int C>
static int launch_first_epi(const FirstArgs &a, int epi, hipStream_t s) {
  if (epi == EPI_GDN) return launch_first<C, EPI_GDN>(a, s);
  if (epi == EPI_NONE) return launch_first<C, EPI_NONE>(a, s);
  if (epi == EPI_RELU) return launch_first<C, EPI_RELU>(a, s);
  return fail(LICOS_EINVAL, "conv5x5s2_first_f16: epilogue %d not supported (none, GDN, ReLU)", epi);
}

template <int C, int EPI>
static int launch_first_raw(const FirstRawArgs &a, hipStream_t s) {
  using G = FirstGeom<C>;
  using R = FirstRawGeom<C>;
  const int tiles = a.tiles_x * a.tiles_y;
  // tiles per workgroup: the resident operands (53 KB) and the first tile's exposed request are paid once per run - 16 for
  // large calls (4.74 against 4.80 ms per 4096 tiles at 8), fewer while the call has fewer than ~1024 workgroups to fill
  // the chip's 512 slots with (a single tile: 64 workgroups of one tile each instead of 8 of eight)
  static const int run_env = [] { const char *e = getenv("LICOS_FIRST_RUN"); return e ? atoi(e) : 0; }();
  long want = (long)a.B * tiles / 1024;
  want = want < 1 ? 1 : (want > 16 ? 16 : want);
  const int run_max = run_env > 0 ? run_env : (int)want;
  const int run = tiles >= run_max ? run_max : tiles;
  const size_t lds = (size_t)16 * (R::S_PAD + G::PATCH_PAD + G::W_GRAN + 16 * G::MT + (EPI == EPI_GDN ? G::GAMMA_GRAN : 0));
  auto kern = conv5x5s2_first_raw_kernel<C, EPI>;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)cdiv(tiles, run) * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "conv5x5s2_first_nchw_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a, run);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int C>
static int launch_first_raw_epi(const FirstRawArgs &a, int epi, hipStream_t s) {
  if (epi == EPI_GDN) return launch_first_raw<C, EPI_GDN>(a, s);
  if (epi == EPI_NONE) return launch_first_raw<C, EPI_NONE>(a, s);
  if (epi == EPI_RELU) return launch_first_raw<C, EPI_RELU>(a, s);
  return fail(LICOS_EINVAL, "conv5x5s2_first_nchw_f16: epilogue %d not supported (none, GDN, ReLU)", epi);
}

}  // namespace licos

using namespace licos;

extern "C" {

#ifdef LICOS_STAMPS
int licos_debug_first_stamps(unsigned long long *out, int reset) {
  if (out) LICOS_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_first_stamps), sizeof(unsigned long long) * 32));
  if (reset) {
    unsigned long long z[32] = {};
    LICOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_first_stamps), z, sizeof(z)));
  }
  return LICOS_OK;
}
#endif

size_t licos_hwc_pad_f16_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || C > 3 || H <= 0 || W <= 0) return 0;
  // + slack: tiles that hang over the bottom / right edge request (masked) rows past the last image
  return ((size_t)B * (H + 4) + 2 * FirstGeom<1>::TH + 4) * first_row_halfs(W, C) * 2 + 1024;
}

int licos_nchw_f32_to_hwc_pad_f16(const float *x_nchw, void *out, int B, int C, int H, int W, void *stream) {
  LICOS_REQUIRE(x_nchw && out, "nchw_f32_to_hwc_pad_f16: null buffer");
  LICOS_REQUIRE(B > 0 && C > 0 && C <= 3 && H > 0 && W > 0, "nchw_f32_to_hwc_pad_f16: needs 1..3 channels");
  LICOS_REQUIRE(((uintptr_t)out & 15) == 0, "nchw_f32_to_hwc_pad_f16: output must be 16-byte aligned");
  const int rsg = first_row_halfs(W, C);
  const long rows_total = (long)B * (H + 4), groups = (rows_total + LP_ROWS - 1) / LP_ROWS;
  const unsigned blocks = (unsigned)(groups < 256 * 16 ? groups : 256 * 16);
  const size_t lds = (size_t)LP_ROWS * rsg * 2;
  LICOS_REQUIRE(lds <= 64 * 1024, "nchw_f32_to_hwc_pad_f16: image too wide");
  half8 *o = reinterpret_cast<half8 *>(out);
  hipStream_t s = as_stream(stream);
  if (C == 1) hipLaunchKernelGGL(nchw_to_hwc_pad_kernel<1>, dim3(blocks), dim3(256), lds, s, x_nchw, o, H, W, rsg, rows_total);
  else if (C == 2) hipLaunchKernelGGL(nchw_to_hwc_pad_kernel<2>, dim3(blocks), dim3(256), lds, s, x_nchw, o, H, W, rsg, rows_total);
  else hipLaunchKernelGGL(nchw_to_hwc_pad_kernel<3>, dim3(blocks), dim3(256), lds, s, x_nchw, o, H, W, rsg, rows_total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

size_t licos_packed_conv_w_first_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cin > 3 || Cout <= 0 || Cout > 128) return 0;
  return (size_t)5 * 4 * 64 * 16 + 64;  // + a granule of zeros (and padding): the in-place form's source for padding
}

int licos_pack_conv_w_first_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  LICOS_REQUIRE(w && packed && Cin > 0 && Cin <= 3 && Cout > 0 && Cout <= 128, "pack_conv_w_first_f16: needs 1..3 input and <= 128 output channels");
  const long total = (long)5 * 4 * 64 * 8;
  LICOS_HIP_CHECK(hipMemsetAsync(static_cast<unsigned char *>(packed) + total * 2, 0, 64, as_stream(stream)));
  hipLaunchKernelGGL(pack_conv_w_first_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w, Cin, Cout, 4,
                     reinterpret_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_conv5x5s2_first_f16(const void *x_hwc_pad, const void *w_packed_first, const float *bias, const void *gdn_packed, int epilogue,
                              void *y_blk16, int B, int Cin, int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(x_hwc_pad && w_packed_first && bias && y_blk16, "conv5x5s2_first_f16: null buffer");
  LICOS_REQUIRE(B > 0 && Cin > 0 && Cin <= 3 && H > 0 && W > 0 && Cout > 0 && Cout <= 128,
                "conv5x5s2_first_f16: needs 1..3 input and <= 128 output channels");
  LICOS_REQUIRE(epilogue != EPI_GDN || gdn_packed, "conv5x5s2_first_f16: the GDN epilogue needs packed gamma/beta");
  LICOS_REQUIRE(((uintptr_t)x_hwc_pad & 15) == 0 && ((uintptr_t)w_packed_first & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)y_blk16 & 15) == 0,
                "conv5x5s2_first_f16: buffers must be 16-byte aligned");
  FirstArgs a{};
  a.x = static_cast<const _Float16 *>(x_hwc_pad);
  a.wp = static_cast<const half8 *>(w_packed_first);
  a.bias = bias;
  a.gamma = static_cast<const bf16x8 *>(gdn_packed);
  a.beta = gdn_packed ? reinterpret_cast<const float *>(static_cast<const unsigned char *>(gdn_packed) + (size_t)4 * 4 * 2 * 1024) : bias;
  a.y_blk = static_cast<_Float16 *>(y_blk16);
  a.B = B;
  a.H = H;
  a.W = W;
  a.Ho = (H - 1) / 2 + 1;
  a.Wo = (W - 1) / 2 + 1;
  a.Cout = Cout;
  a.tiles_x = cdiv(a.Wo, FirstGeom<1>::TW);
  a.tiles_y = cdiv(a.Ho, FirstGeom<1>::TH);
  a.rsg = first_row_halfs(W, Cin);
  LICOS_REQUIRE((long)a.Ho * a.Wo * ((Cout + 15) / 16) * 32 < (1L << 32), "conv5x5s2_first_f16: an image's output must stay below 4 GB (32-bit store offsets)");
  LICOS_REQUIRE((long)(H + 4 + 2 * FirstGeom<1>::TH + 4) * a.rsg * 2 < (1L << 31), "conv5x5s2_first_f16: image too large");
  hipStream_t s = as_stream(stream);
  switch (Cin) {
    case 1: return launch_first_epi<1>(a, epilogue, s);
    case 2: return launch_first_epi<2>(a, epilogue, s);
    default: return launch_first_epi<3>(a, epilogue, s);
  }
}

int licos_conv5x5s2_first_nchw_f16(const float *x_nchw, const void *w_packed_first, const float *bias, const void *gdn_packed, int epilogue,
                                   void *y_blk16, int B, int Cin, int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(x_nchw && w_packed_first && bias && y_blk16, "conv5x5s2_first_nchw_f16: null buffer");
  LICOS_REQUIRE(B > 0 && Cin > 0 && Cin <= 3 && H > 0 && W > 0 && Cout > 0 && Cout <= 128,
                "conv5x5s2_first_nchw_f16: needs 1..3 input and <= 128 output channels");
  LICOS_REQUIRE(W % 4 == 0, "conv5x5s2_first_nchw_f16: the width must be a multiple of 4 (16-byte granules of the fp32 rows); use "
                            "licos_nchw_f32_to_hwc_pad_f16 + licos_conv5x5s2_first_f16 otherwise");
  LICOS_REQUIRE(epilogue != EPI_GDN || gdn_packed, "conv5x5s2_first_nchw_f16: the GDN epilogue needs packed gamma/beta");
  LICOS_REQUIRE(((uintptr_t)x_nchw & 15) == 0 && ((uintptr_t)w_packed_first & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)y_blk16 & 15) == 0,
                "conv5x5s2_first_nchw_f16: buffers must be 16-byte aligned");
  FirstRawArgs a{};
  a.x = x_nchw;
  a.wp = static_cast<const half8 *>(w_packed_first);
  a.bias = bias;
  a.gamma = static_cast<const bf16x8 *>(gdn_packed);
  a.beta = gdn_packed ? reinterpret_cast<const float *>(static_cast<const unsigned char *>(gdn_packed) + (size_t)4 * 4 * 2 * 1024) : bias;
  a.y_blk = static_cast<_Float16 *>(y_blk16);
  a.B = B;
  a.H = H;
  a.W = W;
  a.Ho = (H - 1) / 2 + 1;
  a.Wo = (W - 1) / 2 + 1;
  a.Cout = Cout;
  a.tiles_x = cdiv(a.Wo, FirstGeom<1>::TW);
  a.tiles_y = cdiv(a.Ho, FirstGeom<1>::TH);
  LICOS_REQUIRE((long)a.Ho * a.Wo * ((Cout + 15) / 16) * 32 < (1L << 32), "conv5x5s2_first_nchw_f16: an image's output must stay below 4 GB (32-bit store offsets)");
  LICOS_REQUIRE((long)Cin * H * W < (1L << 30), "conv5x5s2_first_nchw_f16: image too large");
  hipStream_t s = as_stream(stream);
  switch (Cin) {
    case 1: return launch_first_raw_epi<1>(a, epilogue, s);
    case 2: return launch_first_raw_epi<2>(a, epilogue, s);
    default: return launch_first_raw_epi<3>(a, epilogue, s);
  }
}

}  // extern "C"
