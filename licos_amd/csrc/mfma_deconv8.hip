// MFMA stride-2 5x5 transposed convolution, 8-wave variant for wide maps (input >= 16 x 32, 128 output channels).
//
// Same phase decomposition as mfma_deconv.hip (each of the 4 output phases is a stride-1 conv with 3x3 / 3x2 / 2x3 /
// 2x2 taps; one workgroup = one phase of one input tile; the workgroup exits right after its stores), but:
//   * the tile is 16 x 32 input pixels and the workgroup has 8 waves (wave w owns input rows 2w, 2w+1; NT = 2), so
//     the weight fragments of a step are shared by 8 waves and the patch halo shrinks: 5.6 instead of 9.3 LDS-DMA
//     pieces per wave per 50 MFMAs;
//   * a K step is a whole cin chunk (ALL taps of the phase: 32-72 MFMAs per wave between two barriers instead of
//     16-24), written as one static instruction sequence per phase so the LDS reads run one fragment ahead of their
//     MFMAs across tap and kernel-row boundaries;
//   * gamma has its own LDS region and arrives by LDS-DMA during the first step (no register round trip and no extra
//     barrier in front of the epilogue); the bias is the accumulators' initial value;
//   * the (I)GDN epilogue squares and converts each accumulator ONCE per pixel tile (the 4-wave kernel's shared
//     epilogue redoes it for every 32-channel output tile).
// 146 KB of LDS, one workgroup (2 waves per SIMD) per CU.
//
// PAIR: maps that are 16 pixels wide (the first synthesis stage of a 256^2 tile, 16^2 -> 32^2).  Two images share a pixel
// tile: lanes r < 16 of a row are image 2b, lanes r >= 16 image 2b + 1; a patch row (36 granules) holds the 18 columns
// -1 .. 16 of each image, so the second image's lanes read at r - 16 + 18 = r + 2.
#include <cstdlib>

#include "mfma_deconv8.hpp"

namespace licos {

// F32: NCHW fp32 output (the fp32 parity path's one-launch split-operand layers; epilogue NONE / RELU): 128 dword stores
// per wave and phase - more than a counted vmcnt can leave in flight, so every step waits for vmcnt(0).
template <int MT, int EPI, bool PAIR = false, bool F32 = false>
__global__ __launch_bounds__(512, 2) void deconv5x5s2_mfma8_kernel(MfmaArgs a) {
  using G = Deconv8Geom<MT>;
  constexpr int NT = G::NT;
  constexpr bool NORM = (EPI == EPI_GDN || EPI == EPI_IGDN) && LICOS_ABL != 2;
  constexpr int NSTORE = NT * MT * 2;  // 16-byte store instructions of one epilogue when every row of the wave is live
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_pbuf = reinterpret_cast<half8 *>(smem);   // [2][PATCH_PAD]
  half8 *s_wbuf = s_pbuf + 2 * G::PATCH_PAD;         // [2][W_GRAN_MAX]
  float *s_bias = reinterpret_cast<float *>(s_wbuf + 2 * G::W_GRAN_MAX);  // [32 MT] bias, then [32 MT] beta
  float *s_beta = s_bias + 32 * MT;
  bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(s_beta + 32 * MT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int b, tile;
  xcd_work_item(blockIdx.x, PAIR ? (a.B + 1) >> 1 : a.B, a.tiles_x * a.tiles_y, b, tile);
  const int nphase = a.s1conv ? 1 : 4;
  const int ty0 = (tile / a.tiles_x) * G::TH, tx0 = (tile % a.tiles_x) * G::TW;
  const int img = PAIR ? (r >> 4) : 0;  // which image of the pair this lane's pixel belongs to

  const size_t plane = (size_t)a.H * a.W;
  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)(PAIR ? 2 * b : b) * a.Cin16 * plane * 2;
  const half8 *zero = reinterpret_cast<const half8 *>(a.zero16);

  // per-lane source offset of this wave's patch pieces inside a chunk plane (-1: outside the image / padding); the
  // same for every phase and chunk
  int p_off[G::NPP];
#pragma unroll
  for (int i = 0; i < G::NPP; ++i) {
    const int d = (wave + 8 * i) * 64 + lane;
    const int hh = d / G::HALF, rem = d - hh * G::HALF;
    const int j = rem / G::RS;
    int q = rem - j * G::RS, second = 0;
    if (PAIR && q >= G::RS / 2) {
      q -= G::RS / 2;
      second = 1;
    }
    const int iy = ty0 - 1 + j, ix = tx0 - 1 + q;
    const bool ok = d < G::PATCH_GRAN && (PAIR || q < G::TW + 2) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && 2 * b + second < (PAIR ? a.B : 2 * a.B);
    const int pix = a.in_xsplit ? (iy * 2 + (ix & 1)) * (a.W >> 1) + (ix >> 1) : iy * a.W + ix;
    p_off[i] = ok ? second * (int)(a.Cin16 * plane * 2) + pix * 2 + hh : -1;
  }
  auto dma_patch = [&](int cc, int buf) {
    const half8 *xin = xb + (size_t)cc * plane * 2;
#pragma unroll
    for (int i = 0; i < G::NPP; ++i) {
      const int q = wave + 8 * i;
      if (q < G::PQ) glds16(p_off[i] >= 0 ? xin + p_off[i] : zero, s_pbuf + buf * G::PATCH_PAD + q * 64);
    }
  };
  auto dma_w = [&](int phase, int cc, int buf) {  // all taps of (phase, cin chunk): ntap * MT pieces
    const int ntap = ((phase >> 1) ? 2 : 3) * ((phase & 1) ? 2 : 3);
    const int tap0 = (phase == 0) ? 0 : (phase == 1) ? 9 : (phase == 2) ? 15 : 21;
    const half8 *wsrc = a.wp + ((size_t)tap0 * a.Cin16 + (size_t)cc * ntap) * MT * 64 + lane;
#pragma unroll
    for (int i = 0; i < G::NWP; ++i) {
      const int q = wave + 8 * i;
      if (q < ntap * MT) glds16(wsrc + q * 64, s_wbuf + buf * G::W_GRAN_MAX + q * 64);
    }
  };
  // operands of K step `cc` of `phase`, or of the first step of the next phase when cc runs past the last chunk
  auto dma_step = [&](int phase, int cc, int buf) {
    if (cc >= a.Cin16) {
      cc -= a.Cin16;
      ++phase;
    }
    if (phase < nphase) {
      dma_w(phase, cc, buf);
      dma_patch(cc, buf);
    }
  };

  dma_step(0, 0, 0);
  // bias and beta live in LDS: a global load after the prologue would be one more vmcnt event whose wait drains the
  // LDS-DMA requests and the stores in front of it (see the counted wait below)
  static_assert(MT == 4, "bias + beta = one 64-lane piece");
  if (wave == 0) glds16((lane < 32 || !NORM) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias);
  if (NORM) {
#pragma unroll
    for (int i = 0; i < G::NGP; ++i) {
      const int q = wave + 8 * i;
      if (q < G::GAMMA_GRAN / 64) glds16(a.gamma + q * 64 + lane, s_gamma + q * 64);
    }
  }

  int base[NT], oyh[NT], oxh[NT];  // output position of the input pixel at phase (0, 0); row -1 = outside the map
  bool all_live = !F32 && a.Cout >= 32 * MT - 15;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ty = wave * NT + nt, tx = PAIR ? (r & 15) : r;
    const bool in = (ty0 + ty) < a.H && (tx0 + tx) < a.W && (!PAIR || 2 * b + img < a.B);
    oyh[nt] = in ? ty0 + ty : -1;
    oxh[nt] = tx0 + tx;
    all_live = all_live && (ty0 + ty) < a.H;  // wave-uniform: a live row issues its stores whatever its columns
    base[nt] = h * G::HALF + (ty + 1) * G::RS + (r + 1) + 2 * img;  // patch row 0 / column 0 = input row ty0-1 / column tx0-1
  }
  // accumulators start at the bias: channel of register q in tile mt is 32mt + (q&3) + 8(q>>2) + 4h
  f32x16 acc[MT][NT];
  auto acc_init = [&]() {
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

  // The workgroup walks the phases of its tile one after the other: ONE stream of K steps (phase, cin chunk), each
  // step's operands requested one step ahead into the other buffer, with a phase's epilogue between its last step and
  // the next phase's first.  Stores and LDS-DMA share vmcnt (in issue order), so around an epilogue the order is:
  //   barrier of the last step | request the operands of the next phase's SECOND step | epilogue arithmetic, stores |
  //   MFMAs of the next phase's first step | s_waitcnt vmcnt(NSTORE) | barrier
  // - the counted wait lets the NSTORE youngest operations (the stores) stay in flight and still guarantees the
  // requests issued before them have landed; the stores are waited for one whole step later, by the vmcnt(0) of the
  // second step.  Waves that issue fewer stores (rows outside the map, fewer channels) wait for vmcnt(0).
  int cur = 0;
  bool requested = false;  // the operands of the coming step's successor are already on their way
  auto kloop = [&](auto nky_c, auto nkx_c, int phase) {
    constexpr int NKY = decltype(nky_c)::value, NKX = decltype(nkx_c)::value;
    for (int cc = 0; cc < a.Cin16; ++cc) {
      const bool counted = requested && all_live;
      if (!requested && LICOS_ABL != 1) dma_step(phase, cc + 1, cur ^ 1);  // the next step's operands land under this step's MFMAs
      requested = false;
      if (LICOS_ABL != 4)
        deconv8_chunk<MT, NT, NKY, NKX, G::RS>(acc, s_pbuf + cur * G::PATCH_PAD, s_wbuf + cur * G::W_GRAN_MAX, base, lane);
      // my DMA pieces have landed; after the barrier so have everyone's, and every wave is done reading the buffers
      // the next step overwrites
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
    }
  };
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  const int Cout16 = (a.Cout + 15) >> 4;

  // straight-line code per phase (a loop over phases with a switch makes the register allocator spill)
  auto run_phase = [&](auto nky_c, auto nkx_c, auto phase_c) {
    constexpr int phase = decltype(phase_c)::value;
    kloop(nky_c, nkx_c, phase);
    if (phase + 1 < nphase && a.Cin16 > 1 && LICOS_ABL != 1) {
      dma_step(phase + 1, 1, cur ^ 1);
      requested = true;
    }
    asm volatile("" ::: "memory");  // the stores below stay behind the requests above (see the counted wait)

    // ---- epilogue: (I)GDN per pixel tile, fp16 pack, 16-byte stores ------------------------------------------
    constexpr int py = phase >> 1, px = phase & 1;
    long pix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int oy = a.s1conv ? oyh[nt] : 2 * oyh[nt] + py, ox = a.s1conv ? oxh[nt] : 2 * oxh[nt] + px;
      const bool live = oyh[nt] >= 0 && oy < a.Ho && ox < a.Wo;
      // x-split output: row oy as [even-x pixels][odd-x pixels] - this phase's pixels of the row are one run
      pix[nt] = !live ? -1 : (a.out_xsplit ? ((long)oy * 2 + px) * a.W + oxh[nt] : (long)oy * a.Wo + ox);
      if (PAIR && live) pix[nt] += (long)img * Cout16 * a.Ho * a.Wo;  // the second image's planes
    }
    if constexpr (F32) {
      static_assert(!F32 || EPI == EPI_NONE || EPI == EPI_RELU, "the fp32 output has no (I)GDN epilogue");
      const unsigned plane_px = (unsigned)(a.Ho * a.Wo);
      float *yb = a.y_nchw + (size_t)(PAIR ? 2 * b : b) * a.Cout * plane_px;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (pix[nt] < 0) continue;
        // (PAIR: pix carries the second image's blk16 plane offset, img * Cout16 planes; here the image is Cout planes)
        const unsigned p0 = (unsigned)(pix[nt] - (PAIR ? (long)img * Cout16 * plane_px : 0)) + (PAIR ? (unsigned)img * a.Cout * plane_px : 0u);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int c = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * h;
            float v = acc[mt][nt][q];
            if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
            if (c < a.Cout) yb[(unsigned)c * plane_px + p0] = v;
          }
      }
    } else {
      tile8_epilogue<MT, NT, EPI>(acc, s_gamma, s_beta, a.y_blk + (size_t)(PAIR ? 2 * b : b) * Cout16 * a.Ho * a.Wo * 16,
                                  (size_t)a.Ho * a.Wo, Cout16, pix, lane);
    }
    if (phase + 1 < nphase) acc_init();
  };
  run_phase(I3{}, I3{}, std::integral_constant<int, 0>{});
  if (nphase > 1) {
    run_phase(I3{}, I2{}, std::integral_constant<int, 1>{});
    run_phase(I2{}, I3{}, std::integral_constant<int, 2>{});
    run_phase(I2{}, I2{}, std::integral_constant<int, 3>{});
  }
}

template <int MT, int EPI, bool PAIR = false, bool F32 = false>
static int launch_deconv8(const MfmaArgs &a0, hipStream_t s) {
  using G = Deconv8Geom<MT>;
  MfmaArgs a = a0;
  a.tiles_x = PAIR ? 1 : cdiv(a.W, G::TW);
  a.tiles_y = cdiv(a.H, G::TH);
  const size_t lds = (size_t)16 * (G::KLOOP_GRAN + 16 * MT + ((EPI == EPI_GDN || EPI == EPI_IGDN) ? G::GAMMA_GRAN : 0));
  auto kern = deconv5x5s2_mfma8_kernel<MT, EPI, PAIR, F32>;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)a.tiles_x * a.tiles_y * (PAIR ? (a.B + 1) / 2 : a.B);  // a workgroup walks all four phases of its tile
  LICOS_REQUIRE(blocks < (1L << 31), "deconv5x5s2_f16: grid too large");
  LICOS_REQUIRE((long)a.Ho * a.Wo * ((a.Cout + 15) / 16) * (F32 ? 64 : 32) * (PAIR ? 2 : 1) < (1L << 32), "deconv5x5s2_f16: an image's output must stay below 4 GB (32-bit store offsets)");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

// returns LICOS_OK after launching, or 1 when this variant does not apply (caller falls back to the 4-wave kernel)
bool mfma_deconv8_applies(int MT, int Cin16, int H, int W, bool blk_out, bool accum, bool s1conv) {
  static const bool enabled = [] { const char *e = getenv("LICOS_DECONV8"); return !(e && e[0] == '0'); }();
  // (W == 16: two images per pixel tile - also for a batch of one, so that a tile's result does not depend on the batch)
  return enabled && MT == 4 && H >= 16 && (W >= 32 || (W == 16 && !s1conv)) && blk_out && !accum && (Cin16 >= 2 || s1conv);
}

int mfma_try_deconv8(const MfmaArgs &a, int MT, int epi, hipStream_t s) {
  // NCHW fp32 output, no accumulation, no (I)GDN, no clamp: the fp32 parity path's layers
  static const bool f32_enabled = [] { const char *e = getenv("LICOS_DECONV8_F32"); return !(e && e[0] == '0'); }();
  if (f32_enabled && a.y_nchw && !a.y_blk && !a.accum && !a.clamp01 && !a.in_xsplit && !a.out_xsplit && (epi == EPI_NONE || epi == EPI_RELU) &&
      mfma_deconv8_applies(MT, a.Cin16, a.H, a.W, true, false, a.s1conv != 0) && a.Cout > 32) {
    if (a.W == 16) {
      LICOS_REQUIRE((long)a.Cin16 * a.H * a.W * 2 < (1L << 30), "deconv5x5s2_f16: image too large");
      return epi == EPI_NONE ? launch_deconv8<4, EPI_NONE, true, true>(a, s) : launch_deconv8<4, EPI_RELU, true, true>(a, s);
    }
    return epi == EPI_NONE ? launch_deconv8<4, EPI_NONE, false, true>(a, s) : launch_deconv8<4, EPI_RELU, false, true>(a, s);
  }
  if (!mfma_deconv8_applies(MT, a.Cin16, a.H, a.W, a.y_blk != nullptr && !a.out_split3, a.accum != 0, a.s1conv != 0)) return 1;
  if (a.W == 16) {
    LICOS_REQUIRE((long)a.Cin16 * a.H * a.W * 2 < (1L << 30), "deconv5x5s2_f16: image too large");
    if (epi == EPI_IGDN) return launch_deconv8<4, EPI_IGDN, true>(a, s);
    if (epi == EPI_NONE) return launch_deconv8<4, EPI_NONE, true>(a, s);
    if (epi == EPI_RELU) return launch_deconv8<4, EPI_RELU, true>(a, s);
    if (epi == EPI_GDN) return launch_deconv8<4, EPI_GDN, true>(a, s);
    return 1;
  }
  if (epi == EPI_IGDN) return launch_deconv8<4, EPI_IGDN>(a, s);
  if (epi == EPI_GDN) return launch_deconv8<4, EPI_GDN>(a, s);
  if (epi == EPI_NONE) return launch_deconv8<4, EPI_NONE>(a, s);
  if (epi == EPI_RELU) return launch_deconv8<4, EPI_RELU>(a, s);
  return 1;
}

}  // namespace licos
