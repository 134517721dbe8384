// MFMA stride-2 5x5 convolution, 8-wave variant for wide maps (output >= 16 x 32): a workgroup owns a
// 16 x 32 output tile (512 pixels) x all output channels; wave w owns output rows 2w, 2w+1 (NT = 2).
//
// Compared with mfma_conv.hip (4 waves, 2 workgroups per CU, the input rows of every kernel row re-staged):
//   * the input patch of a cin chunk is staged ONCE, as two "parity planes" - the 18 even and the 17 odd
//     input rows of the tile - because kernel rows 0, 2, 4 read only even rows (row 2*ty + ky, as slot
//     ty + ky/2 of the even plane) and kernel rows 1, 3 only odd ones.  The K loop visits kernel rows in the
//     order 0, 2, 4, 1, 3: while the three even steps run, the odd plane of the same chunk is DMA'd in; while
//     the two odd steps run, the even plane of the NEXT chunk is.  Each plane is single-buffered.
//   * the weight fragments of a step (double-buffered) are shared by 8 waves instead of 4.
// Together the LDS-DMA pieces per MFMA drop by ~55 %, which is what the 4-wave kernel's ablation says it is
// paying for.  One workgroup per CU, two waves per SIMD, one s_barrier per step.
#include "mfma_deconv8.hpp"

namespace licos {

template <int MT>
struct Conv8Geom {
  static constexpr int TH = 16, TW = 32, NT = 2;
  static constexpr int PWH = 36, ROWG = 2 * PWH;           // granules per (half, row): both x parities
  static constexpr int NR_E = TH + 2, NR_O = TH + 1;       // even / odd input rows of the tile
  static constexpr int EVEN_GRAN = 2 * NR_E * ROWG;        // 2592
  static constexpr int ODD_GRAN = 2 * NR_O * ROWG;         // 2448
  static constexpr int W_GRAN = 5 * MT * 64;
  static constexpr int EVEN_Q = (EVEN_GRAN + 63) / 64, ODD_Q = (ODD_GRAN + 63) / 64, W_Q = W_GRAN / 64;
  static constexpr int EVEN_PAD = EVEN_Q * 64, ODD_PAD = ODD_Q * 64;
  static constexpr int TOTAL_GRAN = EVEN_PAD + ODD_PAD + 2 * W_GRAN;
  static constexpr int GAMMA_GRAN = MT * MT * 2 * 64;
  static constexpr int LDS_BYTES = 16 * (TOTAL_GRAN > GAMMA_GRAN ? TOTAL_GRAN : GAMMA_GRAN);
  static constexpr int NPE = (EVEN_Q + 7) / 8, NPO = (ODD_Q + 7) / 8, NPW = (W_Q + 7) / 8;  // pieces per wave
  static constexpr int ODD_PER_STEP = (ODD_Q + 2) / 3, EVEN_PER_STEP = (EVEN_Q + 1) / 2;
};

// PAIR: maps whose output is 16 pixels wide (the last analysis stage, 32^2 -> 16^2 at 256^2 tiles).  Two images share a
// pixel tile: lanes r < 16 of a row are image 2b, lanes r >= 16 image 2b + 1.  A parity plane row holds 36 granules per x
// parity; one image needs 18 of them (input columns -2 .. 33), so the second image's patch sits at granule 18 and its
// lanes read at r - 16 + 18 = r + 2.  Wide layers (Cout = 192: six 32-channel tiles, too many accumulators for two waves
// per SIMD) run as a.halves channel groups of MT tiles each, one workgroup per group; the input patch of the second
// group comes out of the XCD's L2.
//
// TILE_EPI (fused GDN, blk16 output - every GDN stage of the analysis transform): the epilogue of the tile kernels
// (mfma_deconv8.hpp: squares converted ONCE per pixel tile instead of once per 32-channel output tile, gamma / beta /
// bias in their own LDS region, requested by LDS-DMA in the prologue instead of copied through registers after the K
// loop; the accumulators start at the bias).  The epilogue is vector-issue bound: 576 fewer vector
// instructions per wave and tile.
template <int MT, int EPI, bool PAIR = false, bool TILE_EPI = false>
__global__ __launch_bounds__(512, 2) void conv5x5s2_mfma8_kernel(MfmaArgs a) {
  using G = Conv8Geom<MT>;
  constexpr int NT = G::NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_even = reinterpret_cast<half8 *>(smem);
  half8 *s_odd = s_even + G::EVEN_PAD;
  half8 *s_wbuf = s_odd + G::ODD_PAD;  // [2][W_GRAN]
  bf16x8 *s_gamma_t = reinterpret_cast<bf16x8 *>(s_wbuf + 2 * G::W_GRAN);             // TILE_EPI: [GAMMA_GRAN]
  float *s_bias_t = reinterpret_cast<float *>(s_gamma_t + G::GAMMA_GRAN);              // TILE_EPI: [32 MT] bias, [32 MT] beta
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int b, tile, mt0 = 0;
  if (PAIR) {
    xcd_work_item(blockIdx.x, (a.B + 1) >> 1, a.tiles_y * a.halves, b, tile);
    mt0 = (tile % a.halves) * MT;
    tile /= a.halves;
  } else {
    xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  }
  const int oy0 = PAIR ? tile * G::TH : (tile / a.tiles_x) * G::TH, ox0 = PAIR ? 0 : (tile % a.tiles_x) * G::TW;
  const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;
  const int img = PAIR ? (r >> 4) : 0;  // which image of the pair this lane's pixel belongs to

  int base_e[NT], base_o[NT], oy[NT], ox[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ty = wave * NT + nt;
    oy[nt] = (PAIR && 2 * b + img >= a.B) ? -1 : oy0 + ty;  // (an odd batch: the last pair's second image does not exist)
    ox[nt] = PAIR ? (r & 15) : ox0 + r;
    base_e[nt] = h * (G::NR_E * G::ROWG) + ty * G::ROWG + r + 2 * img;
    base_o[nt] = h * (G::NR_O * G::ROWG) + ty * G::ROWG + r + 2 * img;
  }
  f32x16 acc[MT][NT];
  if (!TILE_EPI) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;
  }

  const size_t plane = (size_t)a.H * a.W;
  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)(PAIR ? 2 * b : b) * a.Cin16 * plane * 2;
  const half8 *zero = reinterpret_cast<const half8 *>(a.zero16);

  // per-lane source offset (half8 units inside a chunk plane) of this wave's patch pieces; -1 = outside the image
  int e_off[G::NPE], o_off[G::NPO];
  auto piece_offset = [&](int q, int nrows, int row_parity, int limit) {
    const int d = q * 64 + lane;
    if (d >= limit) return -1;
    const int hh = d / (nrows * G::ROWG), rem = d - hh * (nrows * G::ROWG);
    const int j = rem / G::ROWG, r2 = rem - j * G::ROWG;
    const int par = r2 / G::PWH;
    int xh = r2 - par * G::PWH, second = 0;
    if (PAIR && xh >= G::PWH / 2) {
      xh -= G::PWH / 2;
      second = 1;
    }
    const int iy = iy0 + 2 * j + row_parity, ix = ix0 + 2 * xh + par;
    if (iy < 0 || iy >= a.H || ix < 0 || ix >= a.W || (second && 2 * b + 1 >= a.B)) return -1;
    return second * (int)(a.Cin16 * plane * 2) + (iy * a.W + ix) * 2 + hh;
  };
#pragma unroll
  for (int i = 0; i < G::NPE; ++i) e_off[i] = piece_offset(wave + 8 * i, G::NR_E, 0, G::EVEN_GRAN);
#pragma unroll
  for (int i = 0; i < G::NPO; ++i) o_off[i] = piece_offset(wave + 8 * i, G::NR_O, 1, G::ODD_GRAN);

  // DMA helpers: pieces [q_lo, q_hi) of a plane / the weight slab of one kernel row
  auto dma_even = [&](int cc, int q_lo, int q_hi) {
    const half8 *xin = xb + (size_t)cc * plane * 2;
#pragma unroll
    for (int i = 0; i < G::NPE; ++i) {
      const int q = wave + 8 * i;
      if (q >= q_lo && q < q_hi && q < G::EVEN_Q) glds16(e_off[i] >= 0 ? xin + e_off[i] : zero, s_even + q * 64);
    }
  };
  auto dma_odd = [&](int cc, int q_lo, int q_hi) {
    const half8 *xin = xb + (size_t)cc * plane * 2;
#pragma unroll
    for (int i = 0; i < G::NPO; ++i) {
      const int q = wave + 8 * i;
      if (q >= q_lo && q < q_hi && q < G::ODD_Q) glds16(o_off[i] >= 0 ? xin + o_off[i] : zero, s_odd + q * 64);
    }
  };
  auto dma_w = [&](int cc, int ky, int buf) {
    // one kernel row of a cin chunk = [kx][32-channel tile][64 lanes]; a channel group takes MT of the w_mt_total tiles
    const int mtt = PAIR ? a.w_mt_total : MT;
    const half8 *wsrc = a.wp + (size_t)(cc * 5 + ky) * (5 * mtt * 64) + lane;
#pragma unroll
    for (int i = 0; i < G::NPW; ++i) {
      const int q = wave + 8 * i;
      if (q < G::W_Q) glds16(wsrc + (PAIR ? (q / MT) * mtt + mt0 + q % MT : q) * 64, s_wbuf + buf * G::W_GRAN + q * 64);
    }
  };

  dma_even(0, 0, G::EVEN_Q);
  dma_w(0, 0, 0);
  if (TILE_EPI) {
    static_assert(!TILE_EPI || MT == 4, "bias + beta = one 64-lane piece");
    if (wave == 0) glds16(lane < 32 ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias_t);
#pragma unroll
    for (int i = 0; i < (G::GAMMA_GRAN / 64 + 7) / 8; ++i) {
      const int q = wave + 8 * i;
      if (q < G::GAMMA_GRAN / 64) glds16(a.gamma + q * 64 + lane, s_gamma_t + q * 64);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (TILE_EPI) {  // accumulators start at the bias: register q of tile mt is channel 32mt + (q&3) + 8(q>>2) + 4h
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(s_bias_t + 32 * mt + 8 * g + 4 * h);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt][4 * g + 0] = bv.x;
          acc[mt][nt][4 * g + 1] = bv.y;
          acc[mt][nt][4 * g + 2] = bv.z;
          acc[mt][nt][4 * g + 3] = bv.w;
        }
      }
  }
  int wcur = 0;
  for (int cc = 0; cc < a.Cin16; ++cc) {
#pragma unroll
    for (int si = 0; si < 5; ++si) {
      const int ky = (si < 3) ? 2 * si : 2 * si - 5;  // 0, 2, 4, 1, 3
      // prefetch: next step's weights, and a share of the plane that is not being read
      if (si < 4) dma_w(cc, (si + 1 < 3) ? 2 * (si + 1) : 2 * (si + 1) - 5, wcur ^ 1);
      else if (cc + 1 < a.Cin16) dma_w(cc + 1, 0, wcur ^ 1);
      if (si < 3) dma_odd(cc, si * G::ODD_PER_STEP, (si + 1) * G::ODD_PER_STEP);
      else if (cc + 1 < a.Cin16) dma_even(cc + 1, (si - 3) * G::EVEN_PER_STEP, (si - 2) * G::EVEN_PER_STEP);
      const half8 *s_patch = (si < 3) ? s_even : s_odd;
      const half8 *s_w = s_wbuf + wcur * G::W_GRAN;
      const int rowoff = (ky >> 1) * G::ROWG;
      // 5 kx x MT A fragments, each against the NT pixel tiles; the LDS reads run two items ahead of their MFMAs
      // (pinned by sched_group_barrier; counted lgkmcnt waits since the LDS-DMA is issued as inline assembly)
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
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      wcur ^= 1;
    }
  }
  if (TILE_EPI) {
    const int Cout16 = (a.Cout + 15) >> 4;
    long pix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      pix[nt] = (oy[nt] >= 0 && oy[nt] < a.Ho && ox[nt] < a.Wo) ? (long)oy[nt] * a.Wo + ox[nt] : -1;
    tile8_epilogue<MT, NT, EPI>(acc, s_gamma_t, s_bias_t + 32 * MT, a.y_blk + (size_t)(PAIR ? 2 * b + img : b) * Cout16 * a.Ho * a.Wo * 16,
                                (size_t)a.Ho * a.Wo, Cout16, pix, lane);
    return;
  }
  const bf16x8 *gam = a.gamma;
  if (epi_norm(EPI)) {
    bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(smem);
    for (int g = tid; g < epi_gamma_gran(EPI, MT); g += 512) s_gamma[g] = a.gamma[g];
    __syncthreads();
    gam = s_gamma;
  }
  epilogue_store<MT, NT, EPI>(acc, a, gam, PAIR ? 2 * b + img : b, oy, ox, lane, 32 * mt0);
}

template <int MT, int EPI, bool TILE_EPI = false>
static int launch_conv8(const MfmaArgs &a0, hipStream_t s) {
  using G = Conv8Geom<MT>;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.Wo, G::TW);
  a.tiles_y = cdiv(a.Ho, G::TH);
  const size_t kloop = (size_t)G::TOTAL_GRAN * 16, gam = (size_t)epi_gamma_gran(EPI, MT) * 16;
  const size_t lds = TILE_EPI ? (size_t)(G::TOTAL_GRAN + G::GAMMA_GRAN) * 16 + 2 * 32 * MT * sizeof(float) : (kloop > gam ? kloop : gam);
  auto kern = conv5x5s2_mfma8_kernel<MT, EPI, false, TILE_EPI>;
  LICOS_ENSURE_LDS(kern, lds);
  LICOS_REQUIRE((long)a.tiles_x * a.tiles_y * a.B < (1L << 31), "conv5x5s2_f16: grid too large");
  LICOS_REQUIRE(!TILE_EPI || (long)a.Ho * a.Wo * ((a.Cout + 15) / 16) * 32 < (1L << 32), "conv5x5s2_f16: an image's output must stay below 4 GB (32-bit store offsets)");
  hipLaunchKernelGGL(kern, dim3(a.tiles_x * a.tiles_y * a.B), dim3(512), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int MT, int EPI>
static int launch_conv8_pair(const MfmaArgs &a0, int mt_total, hipStream_t s) {
  using G = Conv8Geom<MT>;
  MfmaArgs a = a0;
  a.tiles_x = 1;
  a.tiles_y = a.Ho / G::TH;
  a.w_mt_total = mt_total;
  a.halves = mt_total / MT;
  auto kern = conv5x5s2_mfma8_kernel<MT, EPI, true>;
  const size_t lds = (size_t)G::TOTAL_GRAN * 16;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)((a.B + 1) / 2) * a.tiles_y * a.halves;
  LICOS_REQUIRE(blocks < (1L << 31), "conv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

// returns LICOS_OK after launching, or 1 when this variant does not apply (caller falls back to the 4-wave kernel)
int mfma_try_conv8(const MfmaArgs &a, int MT, int epi, hipStream_t s) {
  // 16-pixel-wide outputs, two images per pixel tile; 192 output channels as two groups of 96 (no norm across groups).
  // Also for a batch of one (half the lanes idle): a tile's bytes must not depend on the batch it was coded in, and the
  // 4-wave kernel adds its partial sums in a different order.
  if (a.Wo == 16 && a.W == 32 && a.Ho >= 16 && (a.Ho % 16) == 0 && !a.accum && (long)a.Cin16 * a.H * a.W * 2 < (1L << 30)) {
    if (MT == 6 && epi == EPI_NONE) return launch_conv8_pair<3, EPI_NONE>(a, 6, s);
    if (MT == 6 && epi == EPI_RELU) return launch_conv8_pair<3, EPI_RELU>(a, 6, s);
    if (MT == 4 && epi == EPI_NONE) return launch_conv8_pair<4, EPI_NONE>(a, 4, s);
    if (MT == 4 && epi == EPI_RELU) return launch_conv8_pair<4, EPI_RELU>(a, 4, s);
  }
  if (MT != 4 || a.Ho < 16 || a.Wo < 32 || (a.Ho % 16) != 0) return 1;
  if (epi == EPI_GDN) return (a.y_blk && !a.accum && !a.out_split3) ? launch_conv8<4, EPI_GDN, true>(a, s) : launch_conv8<4, EPI_GDN>(a, s);
  if (epi == EPI_GDN32) return launch_conv8<4, EPI_GDN32>(a, s);
  if (epi == EPI_NONE) return launch_conv8<4, EPI_NONE>(a, s);
  if (epi == EPI_RELU) return launch_conv8<4, EPI_RELU>(a, s);
  return 1;
}

}  // namespace licos
