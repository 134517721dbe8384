// MFMA stride-2 5x5 convolution (see mfma_common.hpp for the design notes).
//
// K loop = (cin chunk of 16) x (kernel row ky): 5*Cin/16 steps.  Each step needs the TH input rows
// 2*ty + ky of the chunk (both x parities) and one kernel row of weight fragments (5 taps x MT KB).
// Both are brought in by LDS-DMA (global_load_lds_dwordx4, no VGPRs) into the buffer the NEXT step
// will read while the MFMAs of the current step run on the other buffer: one s_barrier per step,
// 2 workgroups per CU (77 KB of LDS each at MT=4).
#include "mfma_common.hpp"

namespace licos {

template <int MT, int NT, int TH, int TW, int EPI>
__global__ __launch_bounds__(256, (MT <= 6 ? 2 : 1)) void conv5x5s2_mfma_kernel(MfmaArgs a) {
  using G = ConvStepGeom<MT, TH, TW>;
  static_assert(TH * TW == 128 * NT, "tile must hold 4 waves x NT x 32 pixels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_buf = reinterpret_cast<half8 *>(smem);  // [2][BUF_GRAN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int b, tile;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  const int oy0 = (tile / a.tiles_x) * TH, ox0 = (tile % a.tiles_x) * TW;
  const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;

  int base[NT], oy[NT], ox[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = (wave * NT + nt) * 32 + r;
    const int ty = p / TW, tx = p % TW;
    oy[nt] = oy0 + ty;
    ox[nt] = ox0 + tx;
    base[nt] = h * G::HALF + ty * G::ROWG + tx;
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

  // LDS-DMA pieces (64 granules each) are dealt round-robin to the 4 waves.  Everything about a patch
  // piece except the kernel row is fixed for the whole K loop, so it is computed once here: the
  // granule's offset from the chunk's (iy0, ix0) pixel, its first input row and whether its column
  // lies inside the image.
  constexpr int PQ = G::PATCH_GRAN / 64, WQ = G::W_GRAN / 64;
  constexpr int NPP = (PQ + 3) / 4, NWP = (WQ + 3) / 4;
  int p_off[NPP], p_row[NPP];
#pragma unroll
  for (int i = 0; i < NPP; ++i) {
    const int d = (wave + 4 * i) * 64 + lane;
    const int hh = d / G::HALF, rem = d - hh * G::HALF;
    const int j = rem / G::ROWG, r2 = rem - j * G::ROWG;
    const int par = r2 / G::PWH, xh = r2 - par * G::PWH;
    const int ix = ix0 + 2 * xh + par;
    p_off[i] = ((2 * j) * a.W + ix) * 2 + hh;                 // half8 units from row iy0 of the chunk plane
    p_row[i] = (ix >= 0 && ix < a.W) ? iy0 + 2 * j : -(1 << 20);  // out-of-image columns never validate
  }
  auto stage = [&](int step, int buf) {
    const int cc = step / 5, ky = step - 5 * cc;
    const half8 *xrow = xb + (size_t)cc * plane * 2 + (ptrdiff_t)(iy0 + ky) * a.W * 2;
    const half8 *wsrc = a.wp + (size_t)step * G::W_GRAN + lane;
    half8 *dst = s_buf + buf * G::BUF_GRAN;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
      const int q = wave + 4 * i;
      if (q < PQ) {
        const bool ok = (unsigned)(p_row[i] + ky) < (unsigned)a.H;
        glds16(ok ? xrow + p_off[i] : zero, dst + q * 64);
      }
    }
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int q = wave + 4 * i;
      if (q < WQ) glds16(wsrc + q * 64, dst + G::PATCH_GRAN + q * 64);
    }
  };

  const int S = a.Cin16 * 5;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < S; ++s) {
    const int cur = s & 1;
    if (s + 1 < S) stage(s + 1, cur ^ 1);
    const half8 *s_patch = s_buf + cur * G::BUF_GRAN;
    const half8 *s_w = s_patch + G::PATCH_GRAN;
#pragma unroll
    for (int kx = 0; kx < 5; ++kx) {
      half8 bf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[nt] = s_patch[base[nt] + (kx & 1) * G::PWH + (kx >> 1)];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const half8 af = s_w[(kx * MT + mt) * 64 + lane];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }
    // my DMA pieces for the next step have landed; after the barrier so have everyone's, and every
    // wave is done reading `cur`, which the step after next overwrites
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const bf16x8 *gam = a.gamma;
  if (epi_norm(EPI)) {
    // gamma fragments are read by all 4 waves: stage them once in the (now free) LDS
    bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(smem);
    for (int g = tid; g < epi_gamma_gran(EPI, MT); g += 256) s_gamma[g] = a.gamma[g];
    __syncthreads();
    gam = s_gamma;
  }
  epilogue_store<MT, NT, EPI>(acc, a, gam, b, oy, ox, lane);
}

template <int MT, int NT, int TH, int TW, int EPI>
static int launch_conv(const MfmaArgs &a0, hipStream_t s) {
  using G = ConvStepGeom<MT, TH, TW>;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.Wo, TW);
  a.tiles_y = cdiv(a.Ho, TH);
  // gamma fragments share the K-loop buffers' space; only (I)GDN epilogues need room for them
  const size_t kloop = (size_t)2 * G::BUF_GRAN * 16, gam = (size_t)epi_gamma_gran(EPI, MT) * 16;
  const size_t lds = kloop > gam ? kloop : gam;
  auto kern = conv5x5s2_mfma_kernel<MT, NT, TH, TW, EPI>;
  LICOS_ENSURE_LDS(kern, lds);
  LICOS_REQUIRE((long)a.tiles_x * a.tiles_y * a.B < (1L << 31), "conv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3(a.tiles_x * a.tiles_y * a.B), dim3(256), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int MT, int EPI>
static int dispatch_tile(const MfmaArgs &a, int width, hipStream_t s) {
  if constexpr (MT <= 4) {
    if (width >= 32) return launch_conv<MT, 2, 8, 32, EPI>(a, s);
    return launch_conv<MT, 2, 16, 16, EPI>(a, s);
  } else {
    // wide channel counts: one pixel tile per wave keeps the accumulators within the register file
    if (width >= 32) return launch_conv<MT, 1, 4, 32, EPI>(a, s);
    return launch_conv<MT, 1, 8, 16, EPI>(a, s);
  }
}

int mfma_dispatch_conv(const MfmaArgs &a, int MT, int epi, int width, hipStream_t s) {
  // wide maps: the 8-wave parity-plane variant (mfma_conv8.hip); LICOS_CONV8=0 keeps the 4-wave kernel for A/B runs
  static const bool use_conv8 = [] { const char *e = getenv("LICOS_CONV8"); return !(e && e[0] == '0'); }();
  if (use_conv8) {
    const int rc = mfma_try_conv8(a, MT, epi, s);
    if (rc <= 0) return rc;
  }
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
  return fail(LICOS_EINVAL, "mfma conv: %d output channels with epilogue %d not instantiated", 32 * MT, epi);
}

}  // namespace licos
