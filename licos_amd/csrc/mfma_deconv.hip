// MFMA stride-2 5x5 transposed convolution (see mfma_common.hpp for the design notes).
//
// out[2ty+py][2tx+px] = sum over ky = py (mod 2), kx = px (mod 2) of
//                       in[ty + (py+2-ky)/2][tx + (px+2-kx)/2] * w[ky][kx]
// so each of the 4 output phases is a stride-1 convolution with 3x3 / 3x2 / 2x3 / 2x2 taps.  A
// workgroup computes one phase of a TH x TW input tile (all output channels).  The chunk's (TH+2) x (TW+2)
// input patch is LDS-DMA'd once (every kernel row reads it at its own row offset) into the buffer the NEXT chunk's
// steps will read; the K loop walks (cin chunk) x (kernel row of the phase), each step's weight row arriving in the
// buffer the next step reads while the MFMAs of the current one run.  49 KB of LDS; the workgroup exits right
// after its stores (stores and LDS-DMA share vmcnt on gfx950: a loop over phases would wait for them).
#include "mfma_common.hpp"

namespace licos {

template <int MT, int NT, int TH, int TW, int EPI>
__global__ __launch_bounds__(256, (MT <= 6 ? 2 : 1)) void deconv5x5s2_mfma_kernel(MfmaArgs a) {
  using G = DeconvStepGeom<MT, TH, TW>;
  static_assert(TH * TW == 128 * NT, "tile must hold 4 waves x NT x 32 pixels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_pbuf = reinterpret_cast<half8 *>(smem);   // [2][PATCH_PAD]
  half8 *s_wbuf = s_pbuf + 2 * G::PATCH_PAD;         // [2][W_GRAN_MAX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int b, item;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y * (a.s1conv ? 1 : 4), b, item);
  const int phase = a.s1conv ? 0 : (item & 3), tile = a.s1conv ? item : (item >> 2);
  const int py = phase >> 1, px = phase & 1;
  const int nky = py ? 2 : 3, nkx = px ? 2 : 3, ntap = nky * nkx;
  const int phase_tap0 = (phase == 0) ? 0 : (phase == 1) ? 9 : (phase == 2) ? 15 : 21;
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;

  int base[NT], oy[NT], ox[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = (wave * NT + nt) * 32 + r;
    const int ty = p / TW, tx = p % TW;
    const bool in = (ty0 + ty) < a.H && (tx0 + tx) < a.W;
    oy[nt] = in ? (a.s1conv ? ty0 + ty : 2 * (ty0 + ty) + py) : -1;
    ox[nt] = a.s1conv ? tx0 + tx : 2 * (tx0 + tx) + px;
    base[nt] = h * G::HALF + (ty + 1) * G::RS + (tx + 1);  // patch row 0 / column 0 = input row ty0-1 / column tx0-1
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;

  const size_t plane = (size_t)a.H * a.W;
  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)b * a.Cin16 * plane * 2;
  const half8 *zero = reinterpret_cast<const half8 *>(a.zero16);

  // per-lane source offset of this wave's patch pieces inside a chunk plane (-1: outside the image or padding);
  // fixed for the whole K loop, because every kernel row reads the same patch
  constexpr int NPP = (G::PQ + 3) / 4, NWP = (3 * MT + 3) / 4;
  int p_off[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int d = (wave + 4 * i) * 64 + lane;
    const int hh = d / G::HALF, rem = d - hh * G::HALF;
    const int j = rem / G::RS, q = rem - j * G::RS;
    const int iy = ty0 - 1 + j, ix = tx0 - 1 + q;
    const bool ok = d < G::PATCH_GRAN && q < TW + 2 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    p_off[i] = ok ? (iy * a.W + ix) * 2 + hh : -1;
  }
  auto dma_patch = [&](int cc, int buf, int q_lo, int q_hi) {
    const half8 *xin = xb + (size_t)cc * plane * 2;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int q = wave + 4 * i;
      if (q >= q_lo && q < q_hi && q < G::PQ) glds16(p_off[i] >= 0 ? xin + p_off[i] : zero, s_pbuf + buf * G::PATCH_PAD + q * 64);
    }
  };
  const int wq = nkx * MT;  // weight pieces per step
  auto dma_w = [&](int cc, int iky, int buf) {
    const half8 *wsrc = a.wp + ((size_t)phase_tap0 * a.Cin16 + (size_t)cc * ntap + (size_t)iky * nkx) * MT * 64 + lane;
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int q = wave + 4 * i;
      if (q < wq) glds16(wsrc + q * 64, s_wbuf + buf * G::W_GRAN_MAX + q * 64);
    }
  };
  const int per_step = (G::PQ + nky - 1) / nky;  // pieces of the next chunk's patch issued per step

  dma_patch(0, 0, 0, G::PQ);
  dma_w(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int wcur = 0;
  for (int cc = 0; cc < a.Cin16; ++cc) {
    const int pcur = cc & 1;
    const bool last_cc = cc + 1 == a.Cin16;
    for (int iky = 0; iky < nky; ++iky) {
      // prefetch: the next step's kernel row, and this step's share of the next chunk's patch
      if (iky + 1 < nky) dma_w(cc, iky + 1, wcur ^ 1);
      else if (!last_cc) dma_w(cc + 1, 0, wcur ^ 1);
      if (!last_cc) dma_patch(cc + 1, pcur ^ 1, iky * per_step, (iky + 1) * per_step);
      const half8 *s_patch = s_pbuf + pcur * G::PATCH_PAD + (1 - iky) * G::RS;  // ky = py + 2*iky -> dy = (py + 2 - ky) / 2 = 1 - iky
      const half8 *s_w = s_wbuf + wcur * G::W_GRAN_MAX;
    // LDS reads run one item (one A fragment = NT MFMAs) ahead of their use, pinned by sched_group_barrier, so
    // the LDS latency sits under the previous item's MFMAs instead of in front of its own
    auto taps = [&](auto first_c, auto count_c) {  // taps [FIRST, FIRST + COUNT) of this kernel row
      constexpr int FIRST = decltype(first_c)::value, NI = decltype(count_c)::value * MT;
      half8 bf[NT], af = s_w[FIRST * MT * 64 + lane];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = s_patch[base[nt] + (1 - FIRST)];  // dx = (px + 2 - kx) / 2 = 1 - ikx
      static_for<NI>([&](auto itc) {
        constexpr int it = decltype(itc)::value, ikx = FIRST + it / MT, mt = it % MT;
        constexpr bool next_tap = (mt == MT - 1 && it + 1 < NI);
        half8 af_n = af, bf_n[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf_n[nt] = bf[nt];
        if (it + 1 < NI) af_n = s_w[(FIRST * MT + it + 1) * 64 + lane];
        if (next_tap) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bf_n[nt] = s_patch[base[nt] + (1 - (ikx + 1))];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, next_tap ? 1 + NT : (it + 1 < NI ? 1 : 0), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
        af = af_n;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = bf_n[nt];
      });
    };
    taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    if (nkx == 3) taps(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});  // wave-uniform: even-x phases
      // my DMA pieces have landed; after the barrier so have everyone's, and every wave is done reading the
      // buffers the next step overwrites
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      wcur ^= 1;
    }
  }
  const bf16x8 *gam = a.gamma;
  if (epi_norm(EPI)) {
    bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(smem);
    for (int g = tid; g < epi_gamma_gran(EPI, MT); g += 256) s_gamma[g] = a.gamma[g];
    __syncthreads();
    gam = s_gamma;
  }
  epilogue_store<MT, NT, EPI>(acc, a, gam, b, oy, ox, lane);
}

template <int MT, int NT, int TH, int TW, int EPI>
static int launch_deconv(const MfmaArgs &a0, hipStream_t s) {
  using G = DeconvStepGeom<MT, TH, TW>;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, TH);
  // gamma fragments share the K-loop buffers' space; only (I)GDN epilogues need room for them
  const size_t kloop = (size_t)G::KLOOP_GRAN * 16, gam = (size_t)epi_gamma_gran(EPI, MT) * 16;
  const size_t lds = kloop > gam ? kloop : gam;
  auto kern = deconv5x5s2_mfma_kernel<MT, NT, TH, TW, EPI>;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)a.tiles_x * a.tiles_y * (a.s1conv ? 1 : 4) * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "deconv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int MT, int EPI>
static int dispatch_tile(const MfmaArgs &a, int width, hipStream_t s) {
  if constexpr (MT <= 4) {
    if (width >= 32) return launch_deconv<MT, 2, 8, 32, EPI>(a, s);
    return launch_deconv<MT, 2, 16, 16, EPI>(a, s);
  } else {
    if (width >= 32) return launch_deconv<MT, 1, 4, 32, EPI>(a, s);
    return launch_deconv<MT, 1, 8, 16, EPI>(a, s);
  }
}

int mfma_dispatch_deconv(const MfmaArgs &a, int MT, int epi, int width, hipStream_t s) {
  const int rc3 = mfma_try_conv3x3_tiles(a, MT, epi, s);  // first analysis stage: resident weights, a run of tiles per workgroup
  if (rc3 != 1) return rc3;
  const int rc8 = mfma_try_deconv8(a, MT, epi, s);  // 8-wave form for wide maps and 128 output channels
  if (rc8 != 1) return rc8;
  if (MT == 1 && epi == EPI_NONE) return dispatch_tile<1, EPI_NONE>(a, width, s);
  if (MT == 4 && epi == EPI_NONE) return dispatch_tile<4, EPI_NONE>(a, width, s);
  if (MT == 4 && epi == EPI_GDN) return dispatch_tile<4, EPI_GDN>(a, width, s);
  if (MT == 4 && epi == EPI_IGDN) return dispatch_tile<4, EPI_IGDN>(a, width, s);
  if (MT == 4 && epi == EPI_GDN32) return dispatch_tile<4, EPI_GDN32>(a, width, s);
  if (MT == 4 && epi == EPI_IGDN32) return dispatch_tile<4, EPI_IGDN32>(a, width, s);
  if (MT == 6 && epi == EPI_NONE) return dispatch_tile<6, EPI_NONE>(a, width, s);
  if (MT == 6 && epi == EPI_GDN) return dispatch_tile<6, EPI_GDN>(a, width, s);
  if (MT == 6 && epi == EPI_IGDN) return dispatch_tile<6, EPI_IGDN>(a, width, s);
  if (MT == 4 && epi == EPI_RELU) return dispatch_tile<4, EPI_RELU>(a, width, s);
  if (MT == 6 && epi == EPI_RELU) return dispatch_tile<6, EPI_RELU>(a, width, s);
  if (MT == 10 && epi == EPI_NONE) return dispatch_tile<10, EPI_NONE>(a, width, s);
  if (MT == 10 && epi == EPI_RELU) return dispatch_tile<10, EPI_RELU>(a, width, s);  // h_s[4] of the q6-8 hyperprior (N -> M = 320, ReLU)
  return fail(LICOS_EINVAL, "mfma deconv: %d output channels with epilogue %d not instantiated", 32 * MT, epi);
}

}  // namespace licos

// -----------------------------------------------------------------------------------------------------
// Last synthesis stage (128 -> 1/3/13 channels, NCHW fp32 out, optional clamp): the output is a few
// channels wide, so one 32-row accumulator tile per phase holds everything and a workgroup can keep
// all FOUR phases of its 8x32 input tile in registers (4 x NT x 16).  The input patch is then staged
// once per cin chunk instead of once per phase and kernel row (the stage is input-read bound: 4 MB
// of activations per tile against 0.8 MB of output), and a lane owns the 2x2 output pixels of its
// input pixel, so it stores px = 0/1 as one 8-byte pair: fully coalesced rows.
namespace licos {

// Weight fragments are stored COMPACT: only the first RP (4/8/16/32 >= Cout) rows of each 32-row A fragment are
// real, so TPP = 32 / RP taps share one 64-granule LDS-DMA piece (granule = piece*64 + half*32 + tap_in_piece*RP
// + row); lanes holding rows >= RP use a zero operand.  For RGB that is 4 KB of weights per cin chunk, not 25.
template <int NT, int TH, int TW, int RP>
__global__ __launch_bounds__(256, 2) void deconv5x5s2_fewch_kernel(MfmaArgs a) {
  using G = DeconvGeom<TH, TW>;
  static_assert(TH * TW == 128 * NT, "tile must hold 4 waves x NT x 32 pixels");
  constexpr int TPP = 32 / RP, WP = (25 + TPP - 1) / TPP;  // taps per piece, weight pieces per chunk
  constexpr int HALF = round_up(G::PH * G::RS, 32), PATCH_GRAN = 2 * HALF, BUF_GRAN = PATCH_GRAN + WP * 64;
  static_assert(PATCH_GRAN % 64 == 0, "patch must be whole LDS-DMA pieces");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_buf = reinterpret_cast<half8 *>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int b, tile;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  const int ty0 = (tile / a.tiles_x) * TH, tx0 = (tile % a.tiles_x) * TW;

  int base[NT], iy[NT], ix[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = (wave * NT + nt) * 32 + r;
    const int ty = p / TW, tx = p % TW;
    iy[nt] = ty0 + ty;
    ix[nt] = tx0 + tx;
    base[nt] = h * HALF + (ty + 1) * G::RS + (tx + 1);
  }
  f32x16 acc[4][NT];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[ph][nt][q] = 0.f;

  const size_t plane = (size_t)a.H * a.W;
  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)b * a.Cin16 * plane * 2;
  const half8 *zero = reinterpret_cast<const half8 *>(a.zero16);

  constexpr int PQ = PATCH_GRAN / 64, NPP = (PQ + 3) / 4, NWP = (WP + 3) / 4;
  const half8 *p_src[NPP];
  bool p_ok[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int d = (wave + 4 * i) * 64 + lane;
    const int hh = d / HALF, rem = d - hh * HALF;
    const int j = rem / G::RS, q = rem - j * G::RS;
    const int yy = ty0 - 1 + j, xx = tx0 - 1 + q;
    p_ok[i] = j < G::PH && q < G::PW && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    p_src[i] = xb + ((ptrdiff_t)yy * a.W + xx) * 2 + hh;
  }
  auto stage = [&](int cc, int buf) {
    half8 *dst = s_buf + buf * BUF_GRAN;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int q = wave + 4 * i;
      if (q < PQ) glds16(p_ok[i] ? p_src[i] + (size_t)cc * plane * 2 : zero, dst + q * 64);
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int q = wave + 4 * i;
      if (q < WP) glds16(a.wp + ((size_t)cc * WP + q) * 64 + lane, dst + PATCH_GRAN + q * 64);
    }
  };

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int cc = 0; cc < a.Cin16; ++cc) {
    const int cur = cc & 1;
    if (cc + 1 < a.Cin16) stage(cc + 1, cur ^ 1);
    const half8 *s_patch = s_buf + cur * BUF_GRAN;
    const half8 *s_w = s_patch + PATCH_GRAN;
    // the 9 distinct shifted views of the patch, shared by the phases that use them
    half8 bf[3][3][NT];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[dy + 1][dx + 1][nt] = s_patch[base[nt] + dy * G::RS + dx];
    int T = 0;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int py = ph >> 1, px = ph & 1;
#pragma unroll
      for (int iky = 0; iky < (py ? 2 : 3); ++iky)
#pragma unroll
        for (int ikx = 0; ikx < (px ? 2 : 3); ++ikx) {
          half8 af = s_w[(T / TPP) * 64 + h * 32 + (T % TPP) * RP + (r & (RP - 1))];
          if (RP < 32 && r >= RP) af = half8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[2 - iky][2 - ikx][nt], acc[ph][nt], 0, 0, 0);
          ++T;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // store: register q of a lane is channel (q&3) + 8(q>>2) + 4h; px = 0/1 of one output row go out as a pair
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    if (iy[nt] >= a.H || ix[nt] >= a.W) continue;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = (q & 3) + 8 * (q >> 2) + 4 * h;
      if (c >= a.Cout) continue;
      const float bias = a.bias[c];
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        float2 v;
        v.x = acc[2 * py][nt][q] + bias;
        v.y = acc[2 * py + 1][nt][q] + bias;
        if (a.clamp01) {
          v.x = fminf(fmaxf(v.x, 0.f), 1.f);
          v.y = fminf(fmaxf(v.y, 0.f), 1.f);
        }
        float *dst = a.y_nchw + (((size_t)b * a.Cout + c) * a.Ho + 2 * iy[nt] + py) * a.Wo + 2 * ix[nt];
        *reinterpret_cast<float2 *>(dst) = v;
      }
    }
  }
}

// ---- the same stage for 5 .. 16 output channels (the 13-band Sentinel-2 models) on v_mfma_f32_16x16x32_f16 -----------
// With 13 output channels a 32-row MFMA tile is 60 % zeros and the kernel above is MFMA-bound on padding (1.0 PFLOP/s
// of issued work, 0.4 useful).  The 16 x 16 x 32 shape has 16 rows: K = 32 = TWO cin chunks of 16 channels per step
// (lane k-group g = lane / 16: chunk g / 2 of the pair, half g % 2), so the nine shifted patch views of a pixel tile are
// still loaded once per step and shared by all 25 taps, and the matrix pipes do a quarter of the cycles per useful MAC.
// 8 waves own a 16 x 32 input tile (wave w: rows 2w, 2w + 1 as four 16-pixel tiles), all four output phases in registers
// (4 x 4 x 4 accumulators), patch and weights of a chunk PAIR staged by LDS-DMA; two workgroups per CU.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Few16Geom {
  static constexpr int TH = 16, TW = 32;
  using G = DeconvGeom<TH, TW>;
  static constexpr int HALF = round_up(G::PH * G::RS, 32);  // granules of one 8-channel half of one chunk's patch
  static constexpr int CHUNK_GRAN = 2 * HALF;
  static constexpr int PATCH_GRAN = 2 * CHUNK_GRAN;         // two chunks per K step
  static constexpr int W_GRAN = 25 * 64;                    // one A fragment (64 lanes x 16 B) per tap
  static constexpr int BUF_GRAN = PATCH_GRAN + W_GRAN;
  static_assert(PATCH_GRAN % 64 == 0, "patch must be whole LDS-DMA pieces");
};

__global__ __launch_bounds__(512, 4) void deconv5x5s2_few16_kernel(MfmaArgs a) {
  using F = Few16Geom;
  using G = F::G;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_buf = reinterpret_cast<half8 *>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px_l = lane & 15, g = lane >> 4;
  int b, tile;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  const int ty0 = (tile / a.tiles_x) * F::TH, tx0 = (tile % a.tiles_x) * F::TW;

  // pixel tile t of this wave: row 2 wave + t / 2, columns 16 (t % 2) .. + 15; this lane's granule at tap offset (0, 0)
  int base[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    base[t] = (g >> 1) * F::CHUNK_GRAN + (g & 1) * F::HALF + (2 * wave + (t >> 1) + 1) * G::RS + (16 * (t & 1) + px_l + 1);
  f32x4 acc[4][4];
#pragma unroll
  for (int ph = 0; ph < 4; ++ph)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[ph][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const size_t plane = (size_t)a.H * a.W;
  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)b * a.Cin16 * plane * 2;
  const half8 *zero = reinterpret_cast<const half8 *>(a.zero16);
  const int npairs = a.Cin16 / 2;

  constexpr int PQ = F::PATCH_GRAN / 64, NPP = (PQ + 7) / 8, NWP = (25 + 7) / 8;
  const half8 *p_src[NPP];
  bool p_ok[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int d = (wave + 8 * i) * 64 + lane;
    const int cc = d / F::CHUNK_GRAN, r1 = d - cc * F::CHUNK_GRAN;
    const int hh = r1 / F::HALF, rem = r1 - hh * F::HALF;
    const int j = rem / G::RS, q = rem - j * G::RS;
    const int yy = ty0 - 1 + j, xx = tx0 - 1 + q;
    p_ok[i] = d < F::PATCH_GRAN && j < G::PH && q < G::PW && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    p_src[i] = xb + ((size_t)cc * plane + (ptrdiff_t)yy * a.W + xx) * 2 + hh;
  }
  auto stage = [&](int pair, int buf) {
    half8 *dst = s_buf + buf * F::BUF_GRAN;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int q = wave + 8 * i;
      if (q < PQ) glds16(p_ok[i] ? p_src[i] + (size_t)pair * 2 * plane * 2 : zero, dst + q * 64);
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int q = wave + 8 * i;
      if (q < 25) glds16(a.wp + ((size_t)pair * 25 + q) * 64 + lane, dst + F::PATCH_GRAN + q * 64);
    }
  };

  // ONE buffer per workgroup (69 KB): two workgroups share a CU and fill each other's gaps - one stages or stores while the
  // other multiplies - which a double buffer inside a single resident workgroup (137 KB, nothing beside it) did not do:
  // its output stores (106 KB of fp32 per tile) and its first stage overlapped with nothing (5.6 ms per 512 tiles; this form
  // is measured in DESIGN.md section 5).
  for (int pair = 0; pair < npairs; ++pair) {
    if (pair) __builtin_amdgcn_s_barrier();  // every wave has read the previous pair's fragments
    stage(pair, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const half8 *s_patch = s_buf;
    const half8 *s_w = s_patch + F::PATCH_GRAN + lane;
    // kernel row iky reads the patch at dy = 1 - iky, kernel column ikx at dx = 1 - ikx, whatever the phase: a shifted
    // view of the four pixel tiles is loaded once and serves every phase's tap at that (iky, ikx)
#pragma unroll
    for (int iky = 0; iky < 3; ++iky) {
#pragma unroll
      for (int ikx = 0; ikx < 3; ++ikx) {
        half8 bf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) bf[t] = s_patch[base[t] + (1 - iky) * G::RS + (1 - ikx)];
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          if (iky >= (py ? 2 : 3)) continue;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int ph = 2 * py + px, nkx = px ? 2 : 3;
            if (ikx >= nkx) continue;
            const int tap0 = ph == 0 ? 0 : ph == 1 ? 9 : ph == 2 ? 15 : 21;  // phase-major tap order of the packed weights
            const half8 af = s_w[(tap0 + iky * nkx + ikx) * 64];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[ph][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[t], acc[ph][t], 0, 0, 0);
          }
        }
      }
    }
  }
  // store: register i of a lane is channel 4 g + i of pixel px_l; the two column phases of one output row go out as a pair
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int iy = ty0 + 2 * wave + (t >> 1), ix = tx0 + 16 * (t & 1) + px_l;
    if (iy >= a.H || ix >= a.W) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 4 * g + i;
      if (c >= a.Cout) continue;
      const float bias = a.bias[c];
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        float2 v;
        v.x = acc[2 * py][t][i] + bias;
        v.y = acc[2 * py + 1][t][i] + bias;
        if (a.clamp01) {
          v.x = fminf(fmaxf(v.x, 0.f), 1.f);
          v.y = fminf(fmaxf(v.y, 0.f), 1.f);
        }
        float *dst = a.y_nchw + (((size_t)b * a.Cout + c) * a.Ho + 2 * iy + py) * a.Wo + 2 * ix;
        *reinterpret_cast<float2 *>(dst) = v;
      }
    }
  }
}

static int launch_few16(const MfmaArgs &a0, hipStream_t s) {
  using F = Few16Geom;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.W, F::TW);
  a.tiles_y = cdiv(a.H, F::TH);
  const size_t lds = (size_t)F::BUF_GRAN * 16;
  LICOS_ENSURE_LDS(deconv5x5s2_few16_kernel, lds);
  LICOS_REQUIRE((long)a.tiles_x * a.tiles_y * a.B < (1L << 31), "deconv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(deconv5x5s2_few16_kernel, dim3(a.tiles_x * a.tiles_y * a.B), dim3(512), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int RP>
static int launch_fewch(const MfmaArgs &a0, hipStream_t s) {
  using G = DeconvGeom<8, 32>;
  constexpr int TPP = 32 / RP, WP = (25 + TPP - 1) / TPP;
  constexpr int HALF = round_up(G::PH * G::RS, 32), BUF_GRAN = 2 * HALF + WP * 64;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.W, 32);
  a.tiles_y = cdiv(a.H, 8);
  const size_t lds = (size_t)2 * BUF_GRAN * 16;
  auto kern = deconv5x5s2_fewch_kernel<2, 8, 32, RP>;
  LICOS_ENSURE_LDS(kern, lds);
  LICOS_REQUIRE((long)a.tiles_x * a.tiles_y * a.B < (1L << 31), "deconv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3(a.tiles_x * a.tiles_y * a.B), dim3(256), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

// rows kept per weight fragment for `Cout` output channels (must match licos_pack_deconv_w_fewch_f16)
int mfma_launch_deconv_fewch(const MfmaArgs &a, hipStream_t s) {
  if (a.Cout <= 4) return launch_fewch<4>(a, s);
  if (a.Cout <= 8) return launch_fewch<8>(a, s);
  if (fewch_uses_16x16x32(a.Cin16 * 16, a.Cout)) return launch_few16(a, s);  // (same rule as the weight packer)
  if (a.Cout <= 16) return launch_fewch<16>(a, s);
  return launch_fewch<32>(a, s);
}

}  // namespace licos
