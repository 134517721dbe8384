// MFMA conv kernels (see mfma_common.hpp for the design notes).
#include "mfma_common.hpp"

namespace licos {

// ---- stride-2 5x5 convolution --------------------------------------------------------------------
template <int MT, int NT, int TH, int TW, int EPI>
__global__ __launch_bounds__(256, 2) void conv5x5s2_mfma_kernel(MfmaArgs a) {
  using G = ConvGeom<TH, TW>;
  static_assert(TH * TW == 128 * NT, "tile must hold 4 waves x NT x 32 pixels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_patch = reinterpret_cast<half8 *>(smem);
  half8 *s_w = reinterpret_cast<half8 *>(smem + G::PATCH_BYTES);  // [5 taps][MT][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int b = blockIdx.y;
  const int oy0 = (blockIdx.x / a.tiles_x) * TH, ox0 = (blockIdx.x % a.tiles_x) * TW;
  const int iy0 = 2 * oy0 - 2, ix0 = 2 * ox0 - 2;

  int base[NT], oy[NT], ox[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = (wave * NT + nt) * 32 + r;
    const int ty = p / TW, tx = p % TW;
    oy[nt] = oy0 + ty;
    ox[nt] = ox0 + tx;
    base[nt] = h * G::HALF + (2 * ty) * (2 * G::PWH) + tx;
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;

  const size_t plane = (size_t)a.H * a.W;
  for (int cc = 0; cc < a.Cin16; ++cc) {
    const half8 *xin = reinterpret_cast<const half8 *>(a.x) + ((size_t)b * a.Cin16 + cc) * plane * 2;
    for (int ky = 0; ky < 5; ++ky) {
      __syncthreads();  // everyone is done reading the previous slab (and patch, when cc changes)
      if (ky == 0) {
        // patch: global order (row, x, half) is contiguous per row; scatter to [half][row][parity][x/2]
        for (int g = tid; g < G::PH * G::PW * 2; g += 256) {
          const int hh = g & 1, q = (g >> 1) % G::PW, row = (g >> 1) / G::PW;
          const int iy = iy0 + row, ix = ix0 + q;
          half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = xin[((size_t)iy * a.W + ix) * 2 + hh];
          s_patch[hh * G::HALF + (row * 2 + (q & 1)) * G::PWH + (q >> 1)] = v;
        }
      }
      const half8 *wsrc = a.wp + ((size_t)(cc * 5 + ky) * 5 * MT) * 64;
      for (int g = tid; g < 5 * MT * 64; g += 256) s_w[g] = wsrc[g];
      __syncthreads();
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        half8 bf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = s_patch[base[nt] + ky * (2 * G::PWH) + (kx & 1) * G::PWH + (kx >> 1)];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const half8 af = s_w[(kx * MT + mt) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
  epilogue_store<MT, NT, EPI>(acc, a, b, oy, ox, lane);
}

template <int MT, int NT, int TH, int TW, int EPI>
static int launch_conv(const MfmaArgs &a0, hipStream_t s) {
  using G = ConvGeom<TH, TW>;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.Wo, TW);
  a.tiles_y = cdiv(a.Ho, TH);
  const size_t lds = G::PATCH_BYTES + (size_t)5 * MT * 1024;
  auto kern = conv5x5s2_mfma_kernel<MT, NT, TH, TW, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    LICOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  LICOS_REQUIRE((long)a.tiles_x * a.tiles_y < (1L << 31) && a.B <= 65535, "conv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3(a.tiles_x * a.tiles_y, a.B), dim3(256), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}


template <int MT, int EPI>
static int dispatch_tile(const MfmaArgs &a, int width, hipStream_t s) {
  if (MT <= 4) {
    if (width >= 32) return launch_conv<MT, 2, 8, 32, EPI>(a, s);
    return launch_conv<MT, 2, 16, 16, EPI>(a, s);
  }
  // wide channel counts: one pixel tile per wave keeps the accumulators within the register file
  if (width >= 32) return launch_conv<MT, 1, 4, 32, EPI>(a, s);
  return launch_conv<MT, 1, 8, 16, EPI>(a, s);
}

int mfma_dispatch_conv(const MfmaArgs &a, int MT, int epi, int width, hipStream_t s) {
  if (MT == 1 && epi == EPI_NONE) return dispatch_tile<1, EPI_NONE>(a, width, s);
  if (MT == 4 && epi == EPI_NONE) return dispatch_tile<4, EPI_NONE>(a, width, s);
  if (MT == 4 && epi == EPI_GDN) return dispatch_tile<4, EPI_GDN>(a, width, s);
  if (MT == 4 && epi == EPI_IGDN) return dispatch_tile<4, EPI_IGDN>(a, width, s);
  if (MT == 6 && epi == EPI_NONE) return dispatch_tile<6, EPI_NONE>(a, width, s);
  if (MT == 6 && epi == EPI_GDN) return dispatch_tile<6, EPI_GDN>(a, width, s);
  if (MT == 6 && epi == EPI_IGDN) return dispatch_tile<6, EPI_IGDN>(a, width, s);
  return fail(LICOS_EINVAL, "mfma conv: %d output channels with epilogue %d not instantiated", 32 * MT, epi);
}

}  // namespace licos
