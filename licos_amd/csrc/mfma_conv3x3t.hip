// 3x3 stride-1 convolution over ONE 16-channel chunk with resident weights: the first analysis stage g_a[0] for <= 4
// input channels (a 5x5 stride-2 conv over C channels == a 3x3 stride-1 conv over the 4C channels of the 2x2
// space-to-depth image; licos/model_utils.py:31-37 creates that layer, eval_utils.py:200 / train.py:190 run it).
//
// The stage is 72 K-loop MFMAs per wave per 16 x 32 output tile against a 64-MFMA GDN epilogue and 128 KB of stores:
// what costs is everything AROUND the K loop.  So a workgroup (8 waves, one per CU) keeps the 36 KB of weight
// fragments, gamma, beta and the bias resident in LDS and walks a run of TILES output tiles: per tile only the 21 KB
// input patch arrives (double buffered, requested two tiles ahead), and the previous tile's stores drain under the
// next tile's MFMAs - same vmcnt discipline as mfma_deconv8.hip: a tile's order is
//   barrier | request the patch of tile t+2 | epilogue of tile t, stores | MFMAs of tile t+1 | s_waitcnt vmcnt(NSTORE) | barrier
// so the wait leaves the NSTORE youngest operations (the stores) in flight and still covers the request before them.
#include "mfma_deconv8.hpp"

namespace licos {

template <int MT, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3s1_tiles_kernel(MfmaArgs a, int run) {
  using G = Deconv8Geom<MT>;
  constexpr int NT = G::NT;
  constexpr bool NORM = (EPI == EPI_GDN || EPI == EPI_IGDN);
  constexpr int NSTORE = NT * MT * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_pbuf = reinterpret_cast<half8 *>(smem);   // [2][PATCH_PAD]
  half8 *s_w = s_pbuf + 2 * G::PATCH_PAD;            // [9 MT 64] resident
  float *s_bias = reinterpret_cast<float *>(s_w + G::W_GRAN_MAX);
  float *s_beta = s_bias + 32 * MT;
  bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(s_beta + 32 * MT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int tiles = a.tiles_x * a.tiles_y, runs = (tiles + run - 1) / run;
  int b, item;
  xcd_work_item(blockIdx.x, a.B, runs, b, item);
  const int t_first = item * run, t_count = (t_first + run <= tiles) ? run : tiles - t_first;

  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)b * a.H * a.W * 2;  // one chunk per image
  const half8 *zero = reinterpret_cast<const half8 *>(a.zero16);
  auto dma_patch = [&](int t, int buf) {  // patch of the run's tile t
    const int tile = t_first + t;
    const int ty0 = (tile / a.tiles_x) * G::TH, tx0 = (tile % a.tiles_x) * G::TW;
#pragma unroll
    for (int i = 0; i < G::NPP; ++i) {
      const int q = wave + 8 * i;
      if (q >= G::PQ) continue;
      const int d = q * 64 + lane;
      const int hh = d / G::HALF, rem = d - hh * G::HALF;
      const int j = rem / G::RS, c = rem - j * G::RS;
      const int iy = ty0 - 1 + j, ix = tx0 - 1 + c;
      const bool ok = d < G::PATCH_GRAN && c < G::TW + 2 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      glds16(ok ? xb + (iy * a.W + ix) * 2 + hh : zero, s_pbuf + buf * G::PATCH_PAD + q * 64);
    }
  };

  // resident operands + the first patch
#pragma unroll
  for (int i = 0; i < G::NWP; ++i) {
    const int q = wave + 8 * i;
    if (q < 9 * MT) glds16(a.wp + q * 64 + lane, s_w + q * 64);
  }
  static_assert(MT == 4, "bias + beta = one 64-lane piece");
  if (wave == 0) glds16((lane < 32 || !NORM) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias);
  if (NORM) {
#pragma unroll
    for (int i = 0; i < G::NGP; ++i) {
      const int q = wave + 8 * i;
      if (q < G::GAMMA_GRAN / 64) glds16(a.gamma + q * 64 + lane, s_gamma + q * 64);
    }
  }
  dma_patch(0, 0);

  int base[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) base[nt] = h * G::HALF + (wave * NT + nt + 1) * G::RS + (r + 1);
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

  const int Cout16 = (a.Cout + 15) >> 4;
  _Float16 *y_img = a.y_blk + (size_t)b * Cout16 * a.Ho * a.Wo * 16;
  bool counted = false;  // the youngest NSTORE operations of this wave are the previous tile's stores
  for (int t = 0; t < t_count; ++t) {
    const int cur = t & 1;
    deconv8_chunk<MT, NT, 3, 3, G::RS>(acc, s_pbuf + cur * G::PATCH_PAD, s_w, base, lane);
    // the patch of tile t+1 (requested before the previous epilogue, or above) has landed: everything older than
    // this wave's last NSTORE operations is complete
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

template <int MT, int EPI>
static int launch_conv3x3t(const MfmaArgs &a0, hipStream_t s) {
  using G = Deconv8Geom<MT>;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.W, G::TW);
  a.tiles_y = cdiv(a.H, G::TH);
  const int tiles = a.tiles_x * a.tiles_y;
  const int run = tiles >= 8 ? 8 : tiles;  // tiles per workgroup
  const size_t lds = (size_t)16 * (2 * G::PATCH_PAD + G::W_GRAN_MAX + 16 * MT + ((EPI == EPI_GDN || EPI == EPI_IGDN) ? G::GAMMA_GRAN : 0));
  auto kern = conv3x3s1_tiles_kernel<MT, EPI>;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)cdiv(tiles, run) * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "conv3x3s1_f16: grid too large");
  LICOS_REQUIRE((long)a.Ho * a.Wo * ((a.Cout + 15) / 16) * 32 < (1L << 32), "conv3x3s1_f16: an image's output must stay below 4 GB (32-bit store offsets)");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, a, run);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

// returns LICOS_OK after launching, or 1 when this variant does not apply (caller falls back to the general kernels)
int mfma_try_conv3x3_tiles(const MfmaArgs &a, int MT, int epi, hipStream_t s) {
  if (!a.s1conv || a.Cin16 != 1 || MT != 4 || a.H < 16 || a.W < 32 || !a.y_blk || a.accum || a.out_split3) return 1;
  if (epi == EPI_GDN) return launch_conv3x3t<4, EPI_GDN>(a, s);
  if (epi == EPI_NONE) return launch_conv3x3t<4, EPI_NONE>(a, s);
  if (epi == EPI_RELU) return launch_conv3x3t<4, EPI_RELU>(a, s);
  return 1;
}

}  // namespace licos
