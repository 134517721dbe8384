// Last synthesis stage for very few output channels (Cout <= 4: RGB / single-band tiles), scatter form.
//
// The gather forms (mfma_deconv.hip) spend one 32-row MFMA tile per output PHASE with 3 live rows: as many MFMA
// cycles as a 128 -> 32 channel layer.  Here the contraction is done once per INPUT pixel for all 25 taps:
//     Y[c][tap][p] = sum_cin W[cin][c][tap] * X[cin][p]          (M = 25 taps per channel, K = Cin, N = pixels)
// - one 32-row tile per output channel, 25 rows live, 8x fewer MFMAs - and every Y value is then added to the one
// output pixel it belongs to, out[2*iy + ky - 2][2*ix + kx - 2], in an LDS image of the workgroup's output tile.
// Contributions are accumulated as 2^-20 fixed point with integer ds_add, so the sum does not depend on the order
// in which waves arrive (bit-reproducible output); |values| < 2048 by a wide margin for image data in [0, 1].
// The input (TH+2) x (TW+2) pixel patch goes straight from global memory into B fragments (no reuse between waves,
// so no LDS staging); the packed weights (Cout x Cin/16 KiB) are staged in LDS once.  The stage is then bound by
// reading its input once (4 MiB per 256x256 tile), not by MFMA issue.
#include "mfma_common.hpp"

namespace licos {

struct ScatterArgs {
  const half8 *x;     // blk16 input [B][Cin16][H][W][16]
  const half8 *wp;    // [Cout][Cin16][64 lanes] A fragments: row = tap (25 live), k = channel within the chunk
  const float *bias;  // [Cout]
  float *y;           // NCHW fp32 [B][Cout][2H][2W]
  int B, Cin16, H, W, Cout, tiles_x, tiles_y, clamp01, in_xsplit;
};

constexpr int SC_TH = 8, SC_TW = 32, SC_PW = SC_TW + 2, SC_NPX = (SC_TH + 2) * SC_PW;  // 340 patch pixels
constexpr int SC_NT = 3;                                                               // pixel tiles of 32 per wave (4 x 3 x 32 >= 340)
constexpr int SC_OH = 2 * SC_TH, SC_OW = 2 * SC_TW;
constexpr float SC_FIX = 1048576.f;  // 2^20

template <int MT, int CC>
__global__ __launch_bounds__(256, 2) void deconv5x5s2_scatter_kernel(ScatterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int *s_out = reinterpret_cast<int *>(smem);                               // [MT][SC_OH][SC_OW] fixed point
  half8 *s_w = reinterpret_cast<half8 *>(smem + MT * SC_OH * SC_OW * 4);     // [MT][CC][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int b, tile;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  const int ty0 = (tile / a.tiles_x) * SC_TH, tx0 = (tile % a.tiles_x) * SC_TW;

  for (int g = tid; g < MT * CC * 64; g += 256) s_w[g] = a.wp[g];
  for (int e = tid; e < MT * SC_OH * SC_OW; e += 256) s_out[e] = 0;

  // my pixels: patch pixel p = 32 * (wave * NT + nt) + r, row-major over the (TH+2) x (TW+2) patch
  int src_off[SC_NT], oyb[SC_NT], oxb[SC_NT];
#pragma unroll
  for (int nt = 0; nt < SC_NT; ++nt) {
    const int p = 32 * (wave * SC_NT + nt) + r;
    const int prow = p / SC_PW, pcol = p - prow * SC_PW;
    const int iy = ty0 - 1 + prow, ix = tx0 - 1 + pcol;
    const bool ok = p < SC_NPX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    const int pix = a.in_xsplit ? (iy * 2 + (ix & 1)) * (a.W >> 1) + (ix >> 1) : iy * a.W + ix;
    src_off[nt] = ok ? pix * 2 + h : -1;
    oyb[nt] = 2 * (prow - 1) - 2;  // output row (tile-local) of tap ky = 0
    oxb[nt] = 2 * (pcol - 1) - 2;
  }
  const size_t plane2 = (size_t)a.H * a.W * 2;
  const half8 *xb = a.x + (size_t)b * a.Cin16 * plane2;
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  // every B fragment this wave will need is requested up front (NT x CC 16-byte loads per lane, 24 KiB per wave
  // at Cin = 128): the stage is a streaming read, so bytes in flight are what matter, not overlap inside a wave
  half8 bf[SC_NT][CC];
#pragma unroll
  for (int nt = 0; nt < SC_NT; ++nt)
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) bf[nt][cc] = src_off[nt] >= 0 ? xb[(size_t)cc * plane2 + src_off[nt]] : zero8;
  __syncthreads();  // weights staged, output image zeroed

#pragma unroll
  for (int nt = 0; nt < SC_NT; ++nt) {
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][q] = 0.f;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(mt * CC + cc) * 64 + lane], bf[nt][cc], acc[mt], 0, 0, 0);
    // scatter: register q of a 32x32 tile holds row (q&3) + 8(q>>2) + 4h = the tap, column r = my pixel
    const bool px_ok = src_off[nt] >= 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int t0 = (q & 3) + 8 * (q >> 2), t1 = t0 + 4;  // tap for the lower / upper half-wave
      if (t0 >= 25) continue;                               // rows 25..31 carry no tap in either half
      const int ky0 = t0 / 5, kx0 = t0 - 5 * ky0, ky1 = t1 / 5, kx1 = t1 - 5 * ky1;
      const int ky = h ? ky1 : ky0, kx = h ? kx1 : kx0;
      const int oy = oyb[nt] + ky, ox = oxb[nt] + kx;
      const bool ok = px_ok && (h ? t1 < 25 : true) && (unsigned)oy < (unsigned)SC_OH && (unsigned)ox < (unsigned)SC_OW;
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          atomicAdd(&s_out[(mt * SC_OH + oy) * SC_OW + ox], __float2int_rn(acc[mt][q] * SC_FIX));
      }
    }
  }
  __syncthreads();
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  for (int e = tid; e < MT * SC_OH * SC_OW; e += 256) {
    const int c = e / (SC_OH * SC_OW), rem = e - c * (SC_OH * SC_OW);
    const int oy = 2 * ty0 + rem / SC_OW, ox = 2 * tx0 + rem % SC_OW;
    if (c < a.Cout && oy < Ho && ox < Wo) {
      float v = (float)s_out[e] * (1.f / SC_FIX) + a.bias[c];
      if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
      a.y[(((size_t)b * a.Cout + c) * Ho + oy) * Wo + ox] = v;
    }
  }
}

// w: ConvTranspose2d weight [Cin][Cout][5][5] fp32 -> A fragments [c][cc][lane][8]: row = lane & 31 = tap (ky*5+kx),
// k = 8 * (lane >> 5) + e = channel within chunk cc
__global__ void pack_deconv_w_scatter_kernel(const float *__restrict__ w, int Cin, int Cout, _Float16 *__restrict__ out, long total) {
  const int Cin16 = (Cin + 15) / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long f = i >> 9;
    const int cc = (int)(f % Cin16), c = (int)(f / Cin16);
    const int tap = lane & 31, cin = 16 * cc + 8 * (lane >> 5) + e;
    float v = 0.f;
    if (tap < 25 && cin < Cin && c < Cout) v = w[((size_t)cin * Cout + c) * 25 + tap];
    out[i] = (_Float16)v;
  }
}

template <int MT, int CC>
static int launch_scatter_cc(const ScatterArgs &a, hipStream_t s) {
  const size_t lds = (size_t)MT * SC_OH * SC_OW * 4 + (size_t)MT * CC * 64 * 16;
  auto kern = deconv5x5s2_scatter_kernel<MT, CC>;
  LICOS_ENSURE_LDS(kern, lds);
  const long blocks = (long)a.tiles_x * a.tiles_y * a.B;
  LICOS_REQUIRE(blocks < (1L << 31), "deconv5x5s2_scatter_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

template <int MT>
static int launch_scatter(const ScatterArgs &a, hipStream_t s) {
  if (a.Cin16 == 8) return launch_scatter_cc<MT, 8>(a, s);    // N = 128 (qualities 1-5)
  if (a.Cin16 == 12) return launch_scatter_cc<MT, 12>(a, s);  // N = 192 (qualities 6-8)
  return fail(LICOS_EINVAL, "deconv5x5s2_scatter_f16: instantiated for 128 and 192 input channels, got %d chunks of 16", a.Cin16);
}

}  // namespace licos

using namespace licos;

extern "C" {

size_t licos_packed_deconv_w_scatter_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cout > 4) return 0;
  return (size_t)Cout * ((Cin + 15) / 16) * 64 * 16;
}

int licos_pack_deconv_w_scatter_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  LICOS_REQUIRE(w && packed && Cin > 0 && Cout > 0 && Cout <= 4, "pack_deconv_w_scatter_f16: needs 1..4 output channels");
  const long total = (long)Cout * ((Cin + 15) / 16) * 64 * 8;
  hipLaunchKernelGGL(pack_deconv_w_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), w, Cin,
                     Cout, reinterpret_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_deconv5x5s2_scatter_f16(const void *x_blk16, const void *w_packed_scatter, const float *bias, float *y_nchw, int clamp01,
                                  int B, int Cin, int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(x_blk16 && w_packed_scatter && bias && y_nchw, "deconv5x5s2_scatter_f16: null buffer");
  LICOS_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && Cout <= 4, "deconv5x5s2_scatter_f16: needs 1..4 output channels");
  LICOS_REQUIRE((long)H * W * 2 < (1L << 30), "deconv5x5s2_scatter_f16: image too large");
  ScatterArgs a{};
  a.x = reinterpret_cast<const half8 *>(x_blk16);
  a.wp = reinterpret_cast<const half8 *>(w_packed_scatter);
  a.bias = bias;
  a.y = y_nchw;
  a.B = B;
  a.Cin16 = (Cin + 15) / 16;
  a.H = H;
  a.W = W;
  a.Cout = Cout;
  a.tiles_x = cdiv(W, SC_TW);
  a.tiles_y = cdiv(H, SC_TH);
  a.clamp01 = clamp01 & 1;
  a.in_xsplit = (clamp01 >> 1) & 1;
  LICOS_REQUIRE(!a.in_xsplit || W % 2 == 0, "deconv5x5s2_scatter_f16: x-split input needs an even width");
  hipStream_t s = as_stream(stream);
  switch (Cout) {
    case 1: return launch_scatter<1>(a, s);
    case 2: return launch_scatter<2>(a, s);
    case 3: return launch_scatter<3>(a, s);
    default: return launch_scatter<4>(a, s);
  }
}

}  // extern "C"
