// MFMA deconv kernels (see mfma_common.hpp for the design notes).
#include "mfma_common.hpp"

namespace licos {

// ---- stride-2 5x5 transposed convolution, one output phase per workgroup ---------------------------
// out[2ty+py][2tx+px] = sum over ky = py (mod 2), kx = px (mod 2) of in[ty + (py+2-ky)/2][tx + (px+2-kx)/2] w[ky][kx]
template <int MT, int NT, int TH, int TW, int EPI>
__global__ __launch_bounds__(256, 2) void deconv5x5s2_mfma_kernel(MfmaArgs a) {
  using G = DeconvGeom<TH, TW>;
  static_assert(TH * TW == 128 * NT, "tile must hold 4 waves x NT x 32 pixels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_patch = reinterpret_cast<half8 *>(smem);
  half8 *s_w = reinterpret_cast<half8 *>(smem + G::PATCH_BYTES);  // [<=9 taps][MT][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int b = blockIdx.y;
  const int phase = blockIdx.x & 3, tile = blockIdx.x >> 2;
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
    oy[nt] = in ? 2 * (ty0 + ty) + py : -1;
    ox[nt] = 2 * (tx0 + tx) + px;
    base[nt] = h * G::HALF + (ty + 1) * G::RS + (tx + 1);
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
    __syncthreads();
    for (int g = tid; g < G::PH * G::PW * 2; g += 256) {
      const int hh = g & 1, q = (g >> 1) % G::PW, row = (g >> 1) / G::PW;
      const int iy = ty0 - 1 + row, ix = tx0 - 1 + q;
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = xin[((size_t)iy * a.W + ix) * 2 + hh];
      s_patch[hh * G::HALF + row * G::RS + q] = v;
    }
    const half8 *wsrc = a.wp + ((size_t)phase_tap0 * a.Cin16 + (size_t)cc * ntap) * MT * 64;
    for (int g = tid; g < ntap * MT * 64; g += 256) s_w[g] = wsrc[g];
    __syncthreads();
    for (int iky = 0; iky < nky; ++iky) {
      const int dy = 1 - iky;  // ky = py + 2*iky  ->  dy = (py + 2 - ky) / 2
      for (int ikx = 0; ikx < nkx; ++ikx) {
        const int dx = 1 - ikx;
        const int t = iky * nkx + ikx;
        half8 bf[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = s_patch[base[nt] + dy * G::RS + dx];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const half8 af = s_w[(t * MT + mt) * 64 + lane];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt], acc[mt][nt], 0, 0, 0);
        }
      }
    }
  }
  epilogue_store<MT, NT, EPI>(acc, a, a.gamma, b, oy, ox, lane);
}

template <int MT, int NT, int TH, int TW, int EPI>
static int launch_deconv(const MfmaArgs &a0, hipStream_t s) {
  using G = DeconvGeom<TH, TW>;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.W, TW);
  a.tiles_y = cdiv(a.H, TH);
  const size_t lds = G::PATCH_BYTES + (size_t)9 * MT * 1024;
  auto kern = deconv5x5s2_mfma_kernel<MT, NT, TH, TW, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    LICOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  LICOS_REQUIRE((long)a.tiles_x * a.tiles_y * 4 < (1L << 31) && a.B <= 65535, "deconv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3(a.tiles_x * a.tiles_y * 4, a.B), dim3(256), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}


template <int MT, int EPI>
static int dispatch_tile(const MfmaArgs &a, int width, hipStream_t s) {
  if (MT <= 4) {
    if (width >= 32) return launch_deconv<MT, 2, 8, 32, EPI>(a, s);
    return launch_deconv<MT, 2, 16, 16, EPI>(a, s);
  }
  // wide channel counts: one pixel tile per wave keeps the accumulators within the register file
  if (width >= 32) return launch_deconv<MT, 1, 4, 32, EPI>(a, s);
  return launch_deconv<MT, 1, 8, 16, EPI>(a, s);
}

int mfma_dispatch_deconv(const MfmaArgs &a, int MT, int epi, int width, hipStream_t s) {
  if (MT == 1 && epi == EPI_NONE) return dispatch_tile<1, EPI_NONE>(a, width, s);
  if (MT == 4 && epi == EPI_NONE) return dispatch_tile<4, EPI_NONE>(a, width, s);
  if (MT == 4 && epi == EPI_GDN) return dispatch_tile<4, EPI_GDN>(a, width, s);
  if (MT == 4 && epi == EPI_IGDN) return dispatch_tile<4, EPI_IGDN>(a, width, s);
  if (MT == 6 && epi == EPI_NONE) return dispatch_tile<6, EPI_NONE>(a, width, s);
  if (MT == 6 && epi == EPI_GDN) return dispatch_tile<6, EPI_GDN>(a, width, s);
  if (MT == 6 && epi == EPI_IGDN) return dispatch_tile<6, EPI_IGDN>(a, width, s);
  return fail(LICOS_EINVAL, "mfma deconv: %d output channels with epilogue %d not instantiated", 32 * MT, epi);
}

}  // namespace licos
