// First analysis stage g_a[0] for 4 < C <= 16 input bands (the 13 merged Sentinel-2 bands of raw_image_folder.py:168-174,
// configs 3 - 5) on the NCHW fp32 image IN PLACE: Conv2d(C -> 128, 5x5, stride 2, padding 2) + GDN / ReLU, blk16 fp16 out.
//
// Until round 4 these models paid a layout pass first (licos_nchw_f32_to_blk16: 13.6 MB in, 8.4 MB out per 13 x 512^2
// tile - 11 ms per 2048 tiles, HBM-bound) and then ran the generic 8-wave kernel (mfma_conv8.hip) on a one-chunk blk16
// image, one tile per workgroup with every operand's first request exposed.  Here the same K loop (kernel rows over the
// even / odd input-row planes, weights streamed per kernel row) and the same epilogue run on planes the workgroup fills
// ITSELF from the fp32 image: a thread owns one (plane row, group of 4 pixels) position, loads the 16 bytes of each band
// there (a group is entirely inside the image or entirely outside it: W a multiple of 4, tile origins multiples of 64),
// and writes each pixel's bands as 8-byte pieces (4 channels) of the planes' [half][row][x-parity][x/2] granules - four
// bands at a time, so that at most 8 loads (32 registers) are in flight beside the accumulators; channel slots C .. 15
// are written as zeros.  A workgroup walks a run of tiles with gamma / beta / bias resident; the planes are single-
// buffered and refilled in the steps in which the K loop does not read them (requested at a step's top, written behind
// its MFMAs):
//     steps ky=0, ky=2 (even plane): the ODD rows of this tile, bands 0-7 then 8-15
//     steps ky=1, ky=3 (odd plane) : the EVEN rows of the NEXT tile, bands 0-7 then 8-15
// Same MFMA order as the blk16 route (cin chunk, ky in 0 2 4 1 3, kx, channel tile), same fp16 rounding of the input:
// the output is bit-identical to licos_nchw_f32_to_blk16 + licos_conv5x5s2_f16 (tests/test_gpu_fp16.py).
#include <cstdlib>

// This stage's output is never re-read before it has left every cache: non-temporal stores (A/B on one box, duo form,
// 1024 tiles of 13 x 512^2: 9.61 ms against 9.72 with the write-through `sc1` the other tile kernels use; plain stores 9.86).
#ifndef LICOS_STORE_BITS
#define LICOS_STORE_BITS "nt"
#endif
// (A/B builds) the duo kernel's raw-row staging: 0 = by the K role into the other group's planes; 1 = by the epilogue role
// into its own (the epilogue's blocks then store through a raw buffer so that the waits can count past them): correct and
// SLOWER - 9.83 against 9.45 ms per 1024 tiles (profiles/r05_first16_stage_in_epilogue_ab.log): the epilogue role, not the K
// role, is what a period waits for.
#ifndef LICOS_F16D_STAGE_E
#define LICOS_F16D_STAGE_E 0
#endif
#include "mfma_deconv8.hpp"

namespace licos {

#ifndef LICOS_F16D_PRIO_E
#define LICOS_F16D_PRIO_E 1
#endif
#ifndef LICOS_F16D_PRIO_YOUNG  // (A/B) static extra priority for waves 4 - 7, the arbitration losers of every SIMD pair
#define LICOS_F16D_PRIO_YOUNG 0
#endif
// LICOS_STAMPS (diagnostic builds via tools/ab_build.sh, never the product): wave 0 of every workgroup adds the s_memtime
// cycles it spent in each phase of a tile into g_first16_stamps; licos_debug_first16_stamps copies them out.
#ifdef LICOS_STAMPS
__device__ unsigned long long g_first16_stamps[64];
#define F16_STAMP(i)                                                   \
  do {                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();      \
    st_acc[i] += now_ - st_prev;                                       \
    st_prev = now_;                                                    \
  } while (0)
#else
#define F16_STAMP(i) do {} while (0)
#endif

typedef float lds_f32x4_t __attribute__((ext_vector_type(4)));  // (a native vector: HIP's float4 has no address-space-qualified copy)

struct First16Args {
  const float *x;      // NCHW fp32 [B][C][H][W]
  const half8 *wp;     // licos_pack_conv_w_f16 of the [128][C][5][5] weight: [ky][kx][mt][64] A fragments (one cin chunk)
  const float *bias, *beta;
  const bf16x8 *gamma;
  _Float16 *y_blk;
  int B, C, H, W, Ho, Wo, Cout, tiles_x, tiles_y;
  int ywalk;  // a run walks DOWN a tile column (x-neighbouring runs are neighbouring work items) instead of along a tile row
};

struct First16Geom {
  static constexpr int MT = 4, NT = 2, TH = 16, TW = 32;
  static constexpr int PWH = 36, ROWG = 2 * PWH;            // granules per (half, row): both x parities
  static constexpr int NR_E = TH + 2, NR_O = TH + 1;        // even / odd input rows of a tile
  static constexpr int EVEN_GRAN = 2 * NR_E * ROWG, ODD_GRAN = 2 * NR_O * ROWG;
  static constexpr int W_GRAN = 5 * MT * 64, GAMMA_GRAN = MT * MT * 2 * 64;
  static constexpr int SG = 18;                              // 4-pixel groups per band and row: columns 2 tx0 - 4 .. 2 tx0 + 67
  static constexpr int LDS_BYTES = 16 * (EVEN_GRAN + ODD_GRAN + 2 * W_GRAN + GAMMA_GRAN) + 2 * 32 * MT * 4;
  static_assert(NR_E * SG <= 512, "one (plane row, pixel group) position per thread");
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void conv5x5s2_first16_kernel(First16Args a, int run) {
  using G = First16Geom;
  constexpr int MT = G::MT, NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_even = reinterpret_cast<half8 *>(smem);
  half8 *s_odd = s_even + G::EVEN_GRAN;
  half8 *s_wbuf = s_odd + G::ODD_GRAN;  // [2][W_GRAN]
  bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(s_wbuf + 2 * G::W_GRAN);
  float *s_bias = reinterpret_cast<float *>(s_gamma + G::GAMMA_GRAN);  // [32 MT] bias, [32 MT] beta
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y;
  const int runs = a.ywalk ? a.tiles_x * ((a.tiles_y + run - 1) / run) : (tiles + run - 1) / run;
  int b, item;
  xcd_work_item(blockIdx.x, a.B, runs, b, item);
  int t_first = item * run, t_count = (t_first + run <= tiles) ? run : tiles - t_first, t_stride = 1;
  if (a.ywalk) {  // item -> (tile column, run of tile rows); `runs` was computed by the launcher's formula for this order
    const int col = item % a.tiles_x, yr = item / a.tiles_x;
    t_first = yr * run * a.tiles_x + col;
    t_stride = a.tiles_x;
    t_count = (yr * run + run <= a.tiles_y) ? run : a.tiles_y - yr * run;
  }
  const float *xb = a.x + (size_t)b * a.C * a.H * a.W;
#ifdef LICOS_STAMPS
  unsigned long long st_acc[24] = {}, st_prev = __builtin_amdgcn_s_memtime();
#endif

  // ---- raw rows -> planes --------------------------------------------------------------------------------------------
  // thread tid < nrows * SG owns (plane row j, group g) of a plane; `half` = bands 8 half .. 8 half + 7 (two batches of 4)
  const int rj = tid / G::SG, rg = tid - rj * G::SG;
  const size_t band = (size_t)a.H * a.W;
  struct Raw {
    float4 v[8];
  };
  auto raw_load = [&](Raw &rw, int tile, int nrows, int parity, int half) {
    const int iy = 2 * (tile / a.tiles_x) * G::TH - 2 + 2 * rj + parity, ix = 2 * (tile % a.tiles_x) * G::TW - 4 + 4 * rg;
    const bool ok = rj < nrows && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const float *p = xb + (ok ? (size_t)iy * a.W + ix : 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = 8 * half + k;
      rw.v[k] = (ok && c < a.C) ? *reinterpret_cast<const float4 *>(p + (size_t)c * band) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto raw_store = [&](const Raw &rw, half8 *plane, int nrows, int half) {
    if (rj >= nrows) return;
    // pixel p of the group is patch column 4 g + p - 2 = 2 xh + par: p = 0, 1 -> xh = 2 g - 1 (par 0, 1), p = 2, 3 -> xh = 2 g
    unsigned char *gr = reinterpret_cast<unsigned char *>(plane + (half * nrows + rj) * G::ROWG + 2 * rg);
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const float4 &b0 = rw.v[4 * qq], &b1 = rw.v[4 * qq + 1], &b2 = rw.v[4 * qq + 2], &b3 = rw.v[4 * qq + 3];
      const half4v p0 = {(_Float16)b0.x, (_Float16)b1.x, (_Float16)b2.x, (_Float16)b3.x};
      const half4v p1 = {(_Float16)b0.y, (_Float16)b1.y, (_Float16)b2.y, (_Float16)b3.y};
      const half4v p2 = {(_Float16)b0.z, (_Float16)b1.z, (_Float16)b2.z, (_Float16)b3.z};
      const half4v p3 = {(_Float16)b0.w, (_Float16)b1.w, (_Float16)b2.w, (_Float16)b3.w};
      unsigned char *q = gr + 8 * qq;
      if (rg > 0) {
        *reinterpret_cast<half4v *>(q - 16) = p0;
        *reinterpret_cast<half4v *>(q + (G::PWH - 1) * 16) = p1;
      }
      *reinterpret_cast<half4v *>(q) = p2;
      *reinterpret_cast<half4v *>(q + G::PWH * 16) = p3;
    }
  };

  auto dma_w = [&](int ky, int buf) {  // one kernel row of A fragments: [kx][mt][64]
    const half8 *wsrc = a.wp + (size_t)ky * (5 * MT * 64) + lane;
#pragma unroll
    for (int i = 0; i < (5 * MT + 7) / 8; ++i) {
      const int q = wave + 8 * i;
      if (q < 5 * MT) glds16(wsrc + q * 64, s_wbuf + buf * G::W_GRAN + q * 64);
    }
  };

  // resident operands, the first weights, zeroed planes (channel slots C .. 15 and the row tails are never written again)
  dma_w(0, 0);
  static_assert(MT == 4, "bias + beta = one 64-lane piece");
  if (wave == 0) glds16((lane < 32 || EPI != EPI_GDN) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias);
  if (EPI == EPI_GDN) {
#pragma unroll
    for (int i = 0; i < (G::GAMMA_GRAN / 64 + 7) / 8; ++i) {
      const int q = wave + 8 * i;
      if (q < G::GAMMA_GRAN / 64) glds16(a.gamma + q * 64 + lane, s_gamma + q * 64);
    }
  }
  Raw rw;
  raw_load(rw, t_first, G::NR_E, 0, 0);
  raw_store(rw, s_even, G::NR_E, 0);
  raw_load(rw, t_first, G::NR_E, 0, 1);
  raw_store(rw, s_even, G::NR_E, 1);

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
  acc_init();

  int base_e[NT], base_o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ty = wave * NT + nt;
    base_e[nt] = h * (G::NR_E * G::ROWG) + ty * G::ROWG + r;
    base_o[nt] = h * (G::NR_O * G::ROWG) + ty * G::ROWG + r;
  }
  const int Cout16 = (a.Cout + 15) >> 4;
  _Float16 *y_img = a.y_blk + (size_t)b * Cout16 * a.Ho * a.Wo * 16;
  // (Measured and not kept: the counted wait of mfma_first.hip around the epilogue - the next tile's second weight row
  // requested before the stores, no raw rows in the first step, s_waitcnt vmcnt(16) behind it: 22.3 against 22.0 ms per
  // 2048 tiles of 13 x 512^2.  Every step ends with vmcnt(0).  Second half of round 4, also not kept: one-dword "touches" of
  // the 128-byte lines of the rows two steps ahead (compiler-visible loads summed into a dead register, the step's wait
  // leaving them in flight) so that the rows come from L2: 24.6 - 25.1 against 21.7 ms - the kernel is at its register
  // limit (255 VGPRs and 4 spills with the touch's address arithmetic) and the extra requests queue in front of the rows.)
  int wcur = 0;
  for (int t = 0; t < t_count; ++t) {
    const int tile = t_first + t * t_stride;
    F16_STAMP(22);
#pragma unroll
    for (int si = 0; si < 5; ++si) {
      const int ky = (si < 3) ? 2 * si : 2 * si - 5;  // 0, 2, 4, 1, 3
      // the next step's weights (the next tile's first kernel row behind the last step)
      if (si < 4) dma_w((si + 1 < 3) ? 2 * (si + 1) : 2 * (si + 1) - 5, wcur ^ 1);
      else if (t + 1 < t_count) dma_w(0, wcur ^ 1);
      // raw rows travel through registers across the step's MFMAs: requested here, converted and written behind them
      if (si < 2) raw_load(rw, tile, G::NR_O, 1, si);
      if (si >= 3 && t + 1 < t_count) raw_load(rw, tile + t_stride, G::NR_E, 0, si - 3);
      const half8 *s_patch = (si < 3) ? s_even : s_odd;
      const half8 *s_w = s_wbuf + wcur * G::W_GRAN;
      const int rowoff = (ky >> 1) * G::ROWG;
      {
        constexpr int NI = 5 * MT;
        const int pb0 = ((si < 3) ? base_e[0] : base_o[0]) + rowoff, pb1 = ((si < 3) ? base_e[1] : base_o[1]) + rowoff;
        half8 a_cur = s_w[lane], a_nxt = s_w[64 + lane], b_cur[NT], b_nxt[NT];
        b_nxt[0] = b_cur[0] = s_patch[pb0];
        b_nxt[1] = b_cur[1] = s_patch[pb1];
        static_for<NI>([&](auto itc) {
          constexpr int it = decltype(itc)::value, mt = it % MT, kx = it / MT;
          constexpr bool more_a = it + 2 < NI, more_b = (mt == MT - 2) && (kx + 1 < 5);
          constexpr int boff = ((kx + 1) & 1) * G::PWH + ((kx + 1) >> 1);
          half8 a_nn = a_nxt;
          if (more_a) a_nn = s_w[(it + 2) * 64 + lane];
          if (more_b) {
            b_nxt[0] = s_patch[pb0 + boff];
            b_nxt[1] = s_patch[pb1 + boff];
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur, b_cur[nt], acc[mt][nt], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, (more_a ? 1 : 0) + (more_b ? NT : 0), 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
          a_cur = a_nxt;
          a_nxt = a_nn;
          if (mt == MT - 1) {
            b_cur[0] = b_nxt[0];
            b_cur[1] = b_nxt[1];
          }
        });
      }
      // the plane this step does not read takes the rows requested at its top (the odd plane is free in steps 0 - 2, the
      // even one - for the next tile - in steps 3 and 4)
      F16_STAMP(4 * si + 0);
      if (si < 2) raw_store(rw, s_odd, G::NR_O, si);
      if (si >= 3 && t + 1 < t_count) raw_store(rw, s_even, G::NR_E, si - 3);
      F16_STAMP(4 * si + 1);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      F16_STAMP(4 * si + 2);
      __builtin_amdgcn_s_barrier();
      F16_STAMP(4 * si + 3);
      wcur ^= 1;
    }
    const int oy0 = (tile / a.tiles_x) * G::TH, ox0 = (tile % a.tiles_x) * G::TW;
    long pix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = oy0 + wave * NT + nt, ox = ox0 + r;
      pix[nt] = (oy < a.Ho && ox < a.Wo) ? (long)oy * a.Wo + ox : -1;
    }
    F16_STAMP(20);
    tile8_epilogue<MT, NT, EPI>(acc, s_gamma, s_bias + 32 * MT, y_img, (size_t)a.Ho * a.Wo, Cout16, pix, lane);
    F16_STAMP(21);
    if (t + 1 < t_count) acc_init();
  }
#ifdef LICOS_STAMPS
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 24; ++i) atomicAdd(&g_first16_stamps[i], st_acc[i]);
    atomicAdd(&g_first16_stamps[24], (unsigned long long)t_count);
  }
#endif
}

// ---- the duo form: two groups of four waves take turns between K loop and epilogue ---------------------------------------
// Stamps of the kernel above (profiles/r05_first16_stamps.log): of a tile's 44.5 k cycles 25 k are work.  The rest is one
// pattern: the epilogue issues the tile's 128 KB of stores in one burst, the first K step's raw-row loads queue behind them
// in the CU's memory path (step 0: 9.9 k cycles against 1.7 k for a step that loads nothing), every wave then waits for
// them - memory and arithmetic take turns instead of overlapping.  Two independent workgroups per CU would fix that and do
// not fit (planes + streamed weight rows + gamma: 93 KB each).  Here ONE workgroup holds two groups of four waves, half a
// tile apart: while group g runs the K loop of its 8 x 32 tile (the matrix pipe; loads), the other group runs the epilogue
// of its previous tile (vector issue, GDN on the matrix pipe between; stores).  The waves w and w + 4 that share a SIMD are
// in different groups, the weight rows and gamma are used by one group at a time and exist once, each group has its own
// planes (2 x 41 KB instead of 81 KB for one 16 x 32 tile): 157 KB of LDS.  Every workgroup barrier is joined by all
// eight waves: the K group at the end of each of its five steps, the epilogue group at five points of its epilogue.
// The K group also stages the OTHER group's next tile (nobody reads those planes meanwhile): 3 rounds of (row, pixel group,
// band half) jobs over its 256 threads, a round's loads issued at a step's top and converted at the end of the NEXT step -
// two steps of flight, behind which the step's weight request is waited for with a counted vmcnt.
// A work item is a pair of neighbouring tile columns (group 0: the left one, group 1: the right one) walked DOWN `run` tile
// rows: the line both groups' patches share is requested half a period apart by the same CU, neighbouring work items
// (neighbouring column pairs) run on the same XCD.  Same MFMA order per output as the kernel above: bit-identical output.
struct First16DuoGeom {
  static constexpr int MT = 4, NT = 2, TH = 8, TW = 32;
  static constexpr int PWH = 34, ROWG = 2 * PWH;           // patch columns 0 .. 66 -> xh 0 .. 33 per x parity
  static constexpr int NR_E = TH + 2, NR_O = TH + 1;
  static constexpr int EVEN_GRAN = 2 * NR_E * ROWG, ODD_GRAN = 2 * NR_O * ROWG, SET_GRAN = EVEN_GRAN + ODD_GRAN;
  static constexpr int W_GRAN = 5 * MT * 64, GAMMA_GRAN = MT * MT * 2 * 64;
  static constexpr int SG = 18;
  static constexpr int POS_E = NR_E * SG, POS = (NR_E + NR_O) * SG, POS_PAD = 384;  // (row, group) positions; padded to 6 waves per band half
  static constexpr int LDS_BYTES = 16 * (2 * SET_GRAN + 2 * W_GRAN + GAMMA_GRAN) + 2 * (2 * 32 * MT * 4);  // bias | beta, twice
  static_assert(POS <= POS_PAD && 2 * POS_PAD == 3 * 256, "three rounds of 256 jobs; a wave's jobs of a round share the band half");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void conv5x5s2_first16_duo_kernel(First16Args a, int run) {
  using G = First16DuoGeom;
  constexpr int MT = G::MT, NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_set = reinterpret_cast<half8 *>(smem);  // [2 groups][even plane | odd plane]
  half8 *s_wbuf = s_set + 2 * G::SET_GRAN;          // [2][W_GRAN]
  bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(s_wbuf + 2 * G::W_GRAN);
  float *s_bias = reinterpret_cast<float *>(s_gamma + G::GAMMA_GRAN);  // [32 MT] bias, [32 MT] beta
  const int tid = threadIdx.x, lane = tid & 63, tk = tid & 255;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wl = wave & 3;
  const int h = lane >> 5, r = lane & 31;
  const int pairs = (a.tiles_x + 1) >> 1, yruns = (a.tiles_y + run - 1) / run;
  int b, item;
  xcd_work_item(blockIdx.x, a.B, pairs * yruns, b, item);
  const int cp = item % pairs, yr = item / pairs;
  const int rows = (yr * run + run <= a.tiles_y) ? run : a.tiles_y - yr * run;
  const int n_tiles = 2 * rows;  // tile k: column 2 cp + (k & 1), tile row yr * run + (k >> 1); group k & 1 owns it
  const float *xb = a.x + (size_t)b * a.C * a.H * a.W;
#ifdef LICOS_STAMPS
  unsigned long long st_acc[20] = {}, st_prev = __builtin_amdgcn_s_memtime();
#endif

  // ---- staging jobs of this thread: round q -> (band half, plane row, pixel group) --------------------------------------
  // job j = tk + 256 q of 768: band half = j / 384, position = j % 384 (>= 342: none) -> even plane rows first, then odd
  int j_src[3], j_dst[3];  // j_src: iy_rel | ix_rel << 8 | half << 16 | valid << 17 | rg << 18; j_dst: byte offset inside a plane set
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int j = tk + 256 * q, half = j >= G::POS_PAD, pos = j - G::POS_PAD * half;
    const bool valid = pos < G::POS, even = pos < G::POS_E;
    const int pe = even ? pos : pos - G::POS_E;
    const int prow = pe / G::SG, rg = pe - prow * G::SG;
    j_src[q] = (2 * prow + (even ? 0 : 1)) | (4 * rg) << 8 | half << 16 | (valid ? 1 : 0) << 17 | rg << 18;
    j_dst[q] = 16 * ((even ? 0 : G::EVEN_GRAN) + (half * (even ? G::NR_E : G::NR_O) + prow) * G::ROWG + 2 * rg);
  }
  struct Raw {
    float4 v[8];
  };
  // The image as a raw buffer (band c at soffset c * band bytes): a lane whose position lies outside the image - or belongs
  // to no job, or to no tile (k >= n_tiles) - asks for an offset past num_records and gets zeros: no branch, no select.
  // Every wave issues the same number of loads per round whatever its lanes' positions (8 for the first band half, C - 8
  // for the second: wave-uniform): the counted waits below rely on it.
  const unsigned band_bytes = (unsigned)(a.H * a.W) * 4u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xb), 0, (int)((unsigned)a.C * band_bytes), 0x00020000);
  const unsigned OOB = 0x80000000u;  // (the launcher keeps an image below 2^31 bytes)
  auto raw_load = [&](Raw &rw, int q, int k) {
    const int ty = yr * run + (k >> 1), tx = 2 * cp + (k & 1);
    const int iy = 2 * ty * G::TH - 2 + (j_src[q] & 255), ix = 2 * tx * G::TW - 4 + ((j_src[q] >> 8) & 255);
    const bool ok = k < n_tiles && ((j_src[q] >> 17) & 1) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const unsigned off = ((unsigned)(iy * a.W + ix) * 4u & 0x7FFFFFFFu) | (ok ? 0u : OOB);
    const int half = __builtin_amdgcn_readfirstlane((j_src[q] >> 16) & 1);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int c = 8 * half + kk;
      if (c < a.C) rw.v[kk] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, c * band_bytes, 0));
      else rw.v[kk] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto raw_store = [&](const Raw &rw, int q, int set) {
    // A use of every loaded register in UNIFORM control flow: the conversions below sit in lane-dependent branches, and a
    // wait the compiler places inside a branch leaves its scoreboard "pending" on the path around it - it then waits for
    // vmcnt(0) at the next reuse of these registers, which is right behind the next step's weight request (and, in the
    // period's first step, behind this wave's own epilogue stores): 1 - 8 k cycles per tile in the first version's stamps.
    asm volatile("" ::"v"(rw.v[0].w), "v"(rw.v[1].w), "v"(rw.v[2].w), "v"(rw.v[3].w), "v"(rw.v[4].w), "v"(rw.v[5].w), "v"(rw.v[6].w),
                 "v"(rw.v[7].w));
    if (!((j_src[q] >> 17) & 1)) return;
    const int rg = j_src[q] >> 18;
    // pixel p of the group is patch column 4 g + p - 2 = 2 xh + par: p = 0, 1 -> xh = 2 g - 1 (par 0, 1), p = 2, 3 -> xh = 2 g;
    // xh = -1 (g = 0) and xh = 34 (g = 17) lie outside the patch
    unsigned char *gr = reinterpret_cast<unsigned char *>(s_set + set * G::SET_GRAN) + j_dst[q];
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const float4 &b0 = rw.v[4 * qq], &b1 = rw.v[4 * qq + 1], &b2 = rw.v[4 * qq + 2], &b3 = rw.v[4 * qq + 3];
      const half4v p0 = {(_Float16)b0.x, (_Float16)b1.x, (_Float16)b2.x, (_Float16)b3.x};
      const half4v p1 = {(_Float16)b0.y, (_Float16)b1.y, (_Float16)b2.y, (_Float16)b3.y};
      const half4v p2 = {(_Float16)b0.z, (_Float16)b1.z, (_Float16)b2.z, (_Float16)b3.z};
      const half4v p3 = {(_Float16)b0.w, (_Float16)b1.w, (_Float16)b2.w, (_Float16)b3.w};
      unsigned char *qd = gr + 8 * qq;
      if (rg > 0) {
        *reinterpret_cast<half4v *>(qd - 16) = p0;
        *reinterpret_cast<half4v *>(qd + (G::PWH - 1) * 16) = p1;
      }
      if (rg < G::SG - 1) {
        *reinterpret_cast<half4v *>(qd) = p2;
        *reinterpret_cast<half4v *>(qd + G::PWH * 16) = p3;
      }
    }
  };
  // one kernel row of A fragments [kx][mt][64], requested by the four waves of the K group (5 pieces each)
  // (ONE scalar base, the piece's place in the lane offset: a scalar base per piece is 25 hoisted SGPR pairs, and SGPRs
  // spilled to VGPR lanes come back through v_readlane right in front of an asm statement that reads them as an address -
  // a hazard the compiler does not pad for inside an asm statement)
  const unsigned w_lane_off = (unsigned)(wl * 64 + lane) * 16u;
  auto dma_w = [&](int ky, int buf) {
#pragma unroll
    for (int i = 0; i < 5 * MT / 4; ++i)
      glds16_s(a.wp, w_lane_off + (unsigned)(ky * (5 * MT * 64) + 4 * i * 64) * 16u, s_wbuf + buf * G::W_GRAN + (wl + 4 * i) * 64);
  };
  static_assert((5 * MT) % 4 == 0, "whole pieces per wave");

  // ---- prologue: resident operands, the first weights (group 0), tile 0 into set 0 (group 1) -------------------------------
  if (wave == 0) glds16((lane < 32 || EPI != EPI_GDN) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias);
  if (wave == 1) glds16((lane < 32 || EPI != EPI_GDN) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias + 2 * 32 * MT);  // (acc_init's second copy)
  if (EPI == EPI_GDN) {
#pragma unroll
    for (int i = 0; i < G::GAMMA_GRAN / 64 / 8; ++i) glds16(a.gamma + (wave + 8 * i) * 64 + lane, s_gamma + (wave + 8 * i) * 64);
  }
  if (grp == 0) {
    dma_w(0, 0);
  } else {
    Raw rw0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      raw_load(rw0, q, 0);
      raw_store(rw0, q, 0);
    }
  }
  f32x16 acc[MT][NT];
  // (the bias sits 153 KB into LDS, beyond a ds_read's 16-bit offset field: one opaque lane base, immediate offsets from it -
  // left to itself the compiler kept sixteen address registers for these reads alive across the loop and spilled them)
  const float *s_bias_lane = s_bias + 4 * h;
  auto acc_init = [&]() {  // accumulators start at the bias: register q of tile mt is channel 32mt + (q&3) + 8(q>>2) + 4h
    // (opaque as an LDS ADDRESS, not as a generic pointer: through a generic pointer the reads become flat loads, whose wait
    // is vmcnt(0) + lgkmcnt(0) - a full round trip of the epilogue's stores issued a moment ago, once per tile)
    unsigned sb_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float *)s_bias_lane;
    asm volatile("" : "+v"(sb_addr));
    const __attribute__((address_space(3))) float *sb = (const __attribute__((address_space(3))) float *)(uintptr_t)sb_addr;
#ifdef LICOS_F16D_ACC_DIRECT
    // every accumulator quad straight from LDS (the second pixel tile from a second copy of the bias, so that the compiler
    // does not share the read and copy 128 registers): 32 ds_read_b128 and no v_mov
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const lds_f32x4_t bv = *(const __attribute__((address_space(3))) lds_f32x4_t *)(sb + nt * (2 * 32 * MT) + 32 * mt + 8 * g);
          acc[mt][nt][4 * g + 0] = bv.x;
          acc[mt][nt][4 * g + 1] = bv.y;
          acc[mt][nt][4 * g + 2] = bv.z;
          acc[mt][nt][4 * g + 3] = bv.w;
        }
#else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const lds_f32x4_t bv = *(const __attribute__((address_space(3))) lds_f32x4_t *)(sb + 32 * mt + 8 * g);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt][4 * g + 0] = bv.x;
          acc[mt][nt][4 * g + 1] = bv.y;
          acc[mt][nt][4 * g + 2] = bv.z;
          acc[mt][nt][4 * g + 3] = bv.w;
        }
      }
#endif
  };
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  acc_init();

  int base_e[NT], base_o[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ty = wl * NT + nt;
    base_e[nt] = grp * G::SET_GRAN + h * (G::NR_E * G::ROWG) + ty * G::ROWG + r;
    base_o[nt] = grp * G::SET_GRAN + G::EVEN_GRAN + h * (G::NR_O * G::ROWG) + ty * G::ROWG + r;
  }
  const int Cout16 = (a.Cout + 15) >> 4;
  _Float16 *y_img = a.y_blk + (size_t)b * Cout16 * a.Ho * a.Wo * 16;
  // loads of the second band half a wave issues per round (the first: 8)
  const int n_hi = a.C > 8 ? a.C - 8 : 0;
  int wcur = 0;
  auto idle_period = [&]() {  // (a period in which this group has no tile: it only keeps the barriers' count)
#pragma unroll
    for (int si = 0; si < 5; ++si) {
      __builtin_amdgcn_s_barrier();
      wcur ^= 1;
    }
  };
  // Group 0: [K E] x rows, idle.  Group 1: idle, [K E] x rows.  Tile p's K role is period p, its epilogue period p + 1; one
  // straight-line loop body per wave (a role switch inside the loop made the allocator spill the accumulators).
#if LICOS_F16D_STAGE_E
  if (grp == 1) {  // period 0: this group's first tile into its own planes
    Raw rw1;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      raw_load(rw1, q, 1);
      raw_store(rw1, q, 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    idle_period();
  }
#else
  if (grp == 1) idle_period();
#endif
  for (int p = grp; p < n_tiles; p += 2) {
    {
      // ---- K role: tile p from this group's planes; tile p + 1 into the other group's ---------------------------------------
#ifdef LICOS_F16D_PRIO
    __builtin_amdgcn_s_setprio(LICOS_F16D_PRIO);  // (A/B) the K role's MFMAs ahead of the SIMD partner's epilogue stream
#endif
#if LICOS_F16D_PRIO_E
    if (LICOS_F16D_PRIO_YOUNG && grp == 1) __builtin_amdgcn_s_setprio(LICOS_F16D_PRIO_YOUNG);
    else __builtin_amdgcn_s_setprio(0);
#endif
    Raw rw;
#pragma unroll
    for (int si = 0; si < 5; ++si) {
      const int ky = (si < 3) ? 2 * si : 2 * si - 5;  // 0, 2, 4, 1, 3
#ifdef LICOS_STAMPS
      if (si == 0) {  // (diagnostic: the five pieces of the first step's weight request one by one)
        F16_STAMP(13);
#pragma unroll
        for (int i = 0; i < 5 * MT / 4; ++i) {
          glds16_s(a.wp, w_lane_off + (unsigned)(2 * (5 * MT * 64) + 4 * i * 64) * 16u, s_wbuf + (wcur ^ 1) * G::W_GRAN + (wl + 4 * i) * 64);
          F16_STAMP(14 + i);
        }
      } else
#endif
      if (si < 4) dma_w((si + 1 < 3) ? 2 * (si + 1) : 2 * (si + 1) - 5, wcur ^ 1);
      else if (p + 1 < n_tiles) dma_w(0, wcur ^ 1);
      asm volatile("" ::: "memory");  // the round's loads stay behind the weight request (the counted wait below)
      if (si == 0) F16_STAMP(11);
#if !LICOS_F16D_STAGE_E
      if (si == 0 || si == 2 || si == 4) raw_load(rw, si >> 1, p + 1);
#endif
      if (si == 0) F16_STAMP(12);
      {
        constexpr int NI = 5 * MT;
        const half8 *s_w = s_wbuf + wcur * G::W_GRAN;
        const int rowoff = (ky >> 1) * G::ROWG;
        const int pb0 = ((si < 3) ? base_e[0] : base_o[0]) + rowoff, pb1 = ((si < 3) ? base_e[1] : base_o[1]) + rowoff;
        half8 a_cur = s_w[lane], a_nxt = s_w[64 + lane], b_cur[NT], b_nxt[NT];
        b_nxt[0] = b_cur[0] = s_set[pb0];
        b_nxt[1] = b_cur[1] = s_set[pb1];
        static_for<NI>([&](auto itc) {
          constexpr int it = decltype(itc)::value, mt = it % MT, kx = it / MT;
          constexpr bool more_a = it + 2 < NI, more_b = (mt == MT - 2) && (kx + 1 < 5);
          constexpr int boff = ((kx + 1) & 1) * G::PWH + ((kx + 1) >> 1);
          half8 a_nn = a_nxt;
          if (more_a) a_nn = s_w[(it + 2) * 64 + lane];
          if (more_b) {
            b_nxt[0] = s_set[pb0 + boff];
            b_nxt[1] = s_set[pb1 + boff];
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur, b_cur[nt], acc[mt][nt], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, (more_a ? 1 : 0) + (more_b ? NT : 0), 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
          a_cur = a_nxt;
          a_nxt = a_nn;
          if (mt == MT - 1) {
            b_cur[0] = b_nxt[0];
            b_cur[1] = b_nxt[1];
          }
        });
      }
      F16_STAMP(si);
#if !LICOS_F16D_STAGE_E
      if (si == 1 || si == 3 || si == 4) raw_store(rw, si == 4 ? 2 : si >> 1, grp ^ 1);
#endif
      F16_STAMP(5);
      // the weight request of this step has landed; a round requested at this step's top (steps 0 and 2) stays in flight:
      // its loads are the wave's youngest operations - 8, or C - 8 for a wave of the second band half
      if (!LICOS_F16D_STAGE_E && (si == 0 || si == 2)) {
        const int half = __builtin_amdgcn_readfirstlane((j_src[si >> 1] >> 16) & 1);
        const int nl = half ? n_hi : 8;
        if (nl == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (nl == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      F16_STAMP(6);
      __builtin_amdgcn_s_barrier();
      F16_STAMP(7);
      wcur ^= 1;
    }
    }
    {
      // ---- epilogue role: five barriers on the way ------------------------------------------------------------------------------
#ifdef LICOS_F16D_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
#if LICOS_F16D_PRIO_E
      // the epilogue's DEPENDENT chains (norm MFMAs behind their gamma reads, rsq behind the norm) ahead of the SIMD partner's
      // independent K-loop MFMAs: 9.61 against 9.71 ms; the other way round (K role first) changed nothing
      if (LICOS_F16D_PRIO_YOUNG && grp == 1) __builtin_amdgcn_s_setprio(LICOS_F16D_PRIO_E + LICOS_F16D_PRIO_YOUNG);
      else __builtin_amdgcn_s_setprio(LICOS_F16D_PRIO_E);
#endif
#if LICOS_F16D_STAGE_E
      // The epilogue role stages the group's OWN next tile (p + 2) into its own planes, which nobody reads this period: a
      // round's eight (C - 8) loads are requested behind one barrier and converted two epilogue blocks later.  The blocks
      // store through a raw buffer - every block issues its two stores whatever the exec mask - so the compiler can count
      // past them to the round's loads instead of waiting for the stores it has just issued.
      Raw rws;
      raw_load(rws, 0, p + 2);
      struct JoinStage : EpilogueBufferStores {
        decltype(raw_load) &ld;
        decltype(raw_store) &st;
        Raw &rw;
        int &wcur;
        int p, grp;
        __device__ __forceinline__ void operator()(int blk) const {
          if (blk == 1 || blk == 3 || blk == 5) {
            st(rw, blk >> 1, grp);
            if (blk < 5) ld(rw, (blk >> 1) + 1, p + 2);
          }
          if (blk == 1 || blk == 3 || blk == 5 || blk == 6) {
            if (blk == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staged planes are complete behind the period's last barrier
            __builtin_amdgcn_s_barrier();
            wcur ^= 1;
          }
        }
      };
      const unsigned out_bytes = (unsigned)Cout16 * a.Ho * a.Wo * 32u;
      JoinStage join{{__builtin_amdgcn_make_buffer_rsrc(y_img, 0, (int)out_bytes, 0x00020000)}, raw_load, raw_store, rws, wcur, p, grp};
#else
      auto join = [&](int blk) {
        if (blk == 1 || blk == 3 || blk == 5 || blk == 6) {
          F16_STAMP(8);
          __builtin_amdgcn_s_barrier();
          F16_STAMP(9);
          wcur ^= 1;
        }
      };
#endif
      const int oy0 = (yr * run + (p >> 1)) * G::TH, ox0 = (2 * cp + (p & 1)) * G::TW;
      long pix[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + wl * NT + nt, ox = ox0 + r;
        pix[nt] = (oy < a.Ho && ox < a.Wo) ? (long)oy * a.Wo + ox : -1;
      }
      tile8_epilogue<MT, NT, EPI>(acc, s_gamma, s_bias + 32 * MT, y_img, (size_t)a.Ho * a.Wo, Cout16, pix, lane, join);
      F16_STAMP(8);
      if (p + 2 < n_tiles) acc_init();
      F16_STAMP(10);
      __builtin_amdgcn_s_barrier();
      F16_STAMP(9);
      wcur ^= 1;
    }
  }
  if (grp == 0) idle_period();
#ifdef LICOS_STAMPS
  if (tk == 0) {  // wave 0 of each group: group g's counters at 32 g
#pragma unroll
    for (int i = 0; i < 19; ++i) atomicAdd(&g_first16_stamps[32 * grp + i], st_acc[i]);
    atomicAdd(&g_first16_stamps[32 * grp + 24], (unsigned long long)rows);
  }
#endif
}

template <int EPI>
static int launch_first16(const First16Args &a_in, hipStream_t s) {
  using G = First16Geom;
  const int tiles = a_in.tiles_x * a_in.tiles_y;
  // tiles per workgroup: the resident operands (gamma, bias, the first weights) and the first tile's exposed rows are paid
  // once per run; short runs while the call has too few workgroups to fill the chip
  static const int run_env = [] { const char *e = getenv("LICOS_FIRST16_RUN"); return e ? atoi(e) : 0; }();
  long want = (long)a_in.B * tiles / 512;
  want = want < 1 ? 1 : (want > 8 ? 8 : want);
  const int run_max = run_env > 0 ? run_env : (int)want;
  static const int duo_env = [] { const char *e = getenv("LICOS_FIRST16_DUO"); return e ? atoi(e) : 1; }();
  if (duo_env && (long)a_in.C * a_in.H * a_in.W * 4 < (1L << 31)) {  // (the duo form reads the image through 32-bit buffer offsets)
    using D = First16DuoGeom;
    First16Args d = a_in;
    d.tiles_x = cdiv(d.Wo, D::TW);
    d.tiles_y = cdiv(d.Ho, D::TH);
    const int pairs = (d.tiles_x + 1) / 2;
    // tile rows per workgroup (two tiles each): the resident operands and the first tile's exposed rows are paid once per
    // run; short runs while the call has too few workgroups to fill the chip
    long want_d = (long)d.B * pairs * d.tiles_y / 1024;
    want_d = want_d < 1 ? 1 : (want_d > 8 ? 8 : want_d);
    const int run_d = run_env > 0 ? run_env : (int)want_d;
    auto kern_d = conv5x5s2_first16_duo_kernel<EPI>;
    LICOS_ENSURE_LDS(kern_d, D::LDS_BYTES);
    const long blocks_d = (long)pairs * cdiv(d.tiles_y, run_d) * d.B;
    LICOS_REQUIRE(blocks_d < (1L << 31), "conv5x5s2_first16_nchw_f16: grid too large");
    hipLaunchKernelGGL(kern_d, dim3((unsigned)blocks_d), dim3(512), D::LDS_BYTES, s, d, run_d);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  }
  static const int ywalk_env = [] { const char *e = getenv("LICOS_FIRST16_YWALK"); return e ? atoi(e) : 0; }();
  First16Args a = a_in;
  a.ywalk = ywalk_env;
  const int run = a.ywalk ? (a.tiles_y >= run_max ? run_max : a.tiles_y) : (tiles >= run_max ? run_max : tiles);
  auto kern = conv5x5s2_first16_kernel<EPI>;
  LICOS_ENSURE_LDS(kern, G::LDS_BYTES);
  const long blocks = (a.ywalk ? (long)a.tiles_x * cdiv(a.tiles_y, run) : (long)cdiv(tiles, run)) * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "conv5x5s2_first16_nchw_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), G::LDS_BYTES, s, a, run);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // namespace licos

using namespace licos;

extern "C" {

#ifdef LICOS_STAMPS
int licos_debug_first16_stamps(unsigned long long *out, int reset) {
  if (out) LICOS_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_first16_stamps), sizeof(unsigned long long) * 64));
  if (reset) {
    unsigned long long z[64] = {};
    LICOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_first16_stamps), z, sizeof(z)));
  }
  return LICOS_OK;
}
#endif

int licos_conv5x5s2_first16_nchw_f16(const float *x_nchw, const void *w_packed, const float *bias, const void *gdn_packed, int epilogue,
                                     void *y_blk16, int B, int Cin, int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(x_nchw && w_packed && bias && y_blk16, "conv5x5s2_first16_nchw_f16: null buffer");
  LICOS_REQUIRE(B > 0 && Cin > 0 && Cin <= 16 && H > 0 && W > 0 && Cout > 32 && Cout <= 128,
                "conv5x5s2_first16_nchw_f16: needs 1..16 input bands and 33..128 output channels (the four-tile weight packing)");
  LICOS_REQUIRE(W % 4 == 0, "conv5x5s2_first16_nchw_f16: the width must be a multiple of 4 (16-byte granules of the fp32 rows); use "
                            "licos_nchw_f32_to_blk16 + licos_conv5x5s2_f16 otherwise");
  LICOS_REQUIRE(epilogue != EPI_GDN || gdn_packed, "conv5x5s2_first16_nchw_f16: the GDN epilogue needs packed gamma/beta");
  LICOS_REQUIRE(((uintptr_t)x_nchw & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)y_blk16 & 15) == 0,
                "conv5x5s2_first16_nchw_f16: buffers must be 16-byte aligned");
  First16Args a{};
  a.x = x_nchw;
  a.wp = static_cast<const half8 *>(w_packed);
  a.bias = bias;
  a.gamma = static_cast<const bf16x8 *>(gdn_packed);
  a.beta = gdn_packed ? reinterpret_cast<const float *>(static_cast<const unsigned char *>(gdn_packed) + (size_t)4 * 4 * 2 * 1024) : bias;
  a.y_blk = static_cast<_Float16 *>(y_blk16);
  a.B = B;
  a.C = Cin;
  a.H = H;
  a.W = W;
  a.Ho = (H - 1) / 2 + 1;
  a.Wo = (W - 1) / 2 + 1;
  a.Cout = Cout;
  a.tiles_x = cdiv(a.Wo, First16Geom::TW);
  a.tiles_y = cdiv(a.Ho, First16Geom::TH);
  LICOS_REQUIRE((long)a.Ho * a.Wo * ((Cout + 15) / 16) * 32 < (1L << 32), "conv5x5s2_first16_nchw_f16: an image's output must stay below 4 GB (32-bit store offsets)");
  LICOS_REQUIRE((long)Cin * H * W < (1L << 30), "conv5x5s2_first16_nchw_f16: image too large");
  hipStream_t s = as_stream(stream);
  if (epilogue == EPI_GDN) return launch_first16<EPI_GDN>(a, s);
  if (epilogue == EPI_NONE) return launch_first16<EPI_NONE>(a, s);
  if (epilogue == EPI_RELU) return launch_first16<EPI_RELU>(a, s);
  return fail(LICOS_EINVAL, "conv5x5s2_first16_nchw_f16: epilogue %d not supported (none, GDN, ReLU)", epilogue);
}

}  // extern "C"
