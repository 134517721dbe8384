// Weight gradient of the 5x5 stride-2 (transposed) convolutions on the matrix cores.
//
//   dW[s][c][ky][kx] += scale * sum over images b and small-map pixels (y, x) of
//                       S[b][s][y][x] * L[b][c][2y + ky - 2][2x + kx - 2]
// with S the small map (the conv's output gradient / the transposed conv's input) and L the large one (the conv's
// input / the transposed conv's output gradient): both Conv2d ([cout][cin][5][5], s = cout, c = cin) and
// ConvTranspose2d ([cin][cout][5][5], s = cin, c = cout) weight layouts come out directly.
//
// The contraction runs over (pixel, image).  Both maps are given "batch-minor": [b/16][H][W][2][channel][8 images] fp16
// (licos_nchw_f32_split_bm8), so an MFMA 32x32x16 step is exactly one pixel - A = 32 small-map channels x 16 images,
// B = 16 images x 32 large-map channels, whatever the tap offset: no im2col, no unaligned access - and the 32 lanes
// of a fragment load (32 consecutive channels x 8 images) read 512 contiguous bytes.
// A wave owns 128 small channels x 32 large channels x TWO taps (128 accumulator registers) over a strip of small-map
// rows and reads its fragments straight from L2 (no reuse inside a wave that LDS staging would serve), four pixels'
// worth in flight at a time.  Every strip writes its partial sums to its own slice of a scratch buffer and a second
// kernel adds the slices in strip order: no atomics (64 strips adding into the same 400 K weights were the
// bottleneck of a first version) and a bit-reproducible gradient.
// Precision: called three times on split operands (hi*hi, hi*lo, lo*hi with the residuals scaled by 2^k and
// `scale` = 2^-k), as the forward path does.
#include "mfma_common.hpp"

namespace licos {

struct WgradArgs {
  const half8 *small;  // [nbc][Hs][Ws][2][Cs] granules of 8 images
  const half8 *large;  // [nbc][Hl][Wl][2][Cl]
  float *part;         // [n_strips][25][Cs][Cl] per-strip partial sums (tap-major: a wave stores 128-byte rows)
  int Cs, Cl, nbc, Hs, Ws, Hl, Wl, rows_per_strip;
};

constexpr int WG_MT = 4;  // 128 small-map channels per wave
constexpr int WG_PX = 4;  // pixels whose fragments are in flight together

__global__ __launch_bounds__(64, 2) void wgrad5x5s2_mfma_kernel(WgradArgs a) {
  const int lane = threadIdx.x, h = lane >> 5, r = lane & 31;
  const int tap0 = 2 * blockIdx.x;                      // taps tap0, tap0 + 1 (the last pair has one)
  const int cl0 = 32 * blockIdx.y;                      // large-map channel tile
  const int n_strips = (a.Hs + a.rows_per_strip - 1) / a.rows_per_strip;
  const int cs0 = 128 * (blockIdx.z / n_strips);        // small-map channel tile
  const int y0 = (blockIdx.z % n_strips) * a.rows_per_strip;
  const int y1 = min(y0 + a.rows_per_strip, a.Hs);
  const int ky[2] = {tap0 / 5, (tap0 + 1) / 5}, kx[2] = {tap0 % 5, (tap0 + 1) % 5};
  const bool tap_ok[2] = {true, tap0 + 1 < 25};

  f32x16 acc[WG_MT][2];
#pragma unroll
  for (int mt = 0; mt < WG_MT; ++mt)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][t][q] = 0.f;

  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const int cl = cl0 + r;
  const bool cl_ok = cl < a.Cl;
  for (int bc = 0; bc < a.nbc; ++bc) {
    // granule of (pixel p, image half h, channel c) = base[(p * 2 + h) * C + c]
    const half8 *sp[WG_MT];
    bool cs_ok[WG_MT];
#pragma unroll
    for (int mt = 0; mt < WG_MT; ++mt) {
      const int cs = cs0 + 32 * mt + r;
      cs_ok[mt] = cs < a.Cs;
      sp[mt] = a.small + ((size_t)bc * a.Hs * a.Ws * 2 + h) * a.Cs + (cs_ok[mt] ? cs : 0);
    }
    const half8 *lp = a.large + ((size_t)bc * a.Hl * a.Wl * 2 + h) * a.Cl + (cl_ok ? cl : 0);
    const size_t ss = (size_t)2 * a.Cs, ls = (size_t)2 * a.Cl;  // granules per pixel
    for (int y = y0; y < y1; ++y) {
      for (int x0 = 0; x0 < a.Ws; x0 += WG_PX) {
        // all fragments of WG_PX pixels are requested before the first MFMA: the loop is a stream of 16-byte loads
        // from L2, so what matters is bytes in flight (6 KiB per pixel and wave)
        half8 af[WG_PX][WG_MT], bf[WG_PX][2];
#pragma unroll
        for (int j = 0; j < WG_PX; ++j) {
          const int x = x0 + j;
          const bool px_ok = x < a.Ws;
          const size_t ps = ((size_t)y * a.Ws + (px_ok ? x : 0)) * ss;
#pragma unroll
          for (int mt = 0; mt < WG_MT; ++mt) af[j][mt] = (px_ok && cs_ok[mt]) ? sp[mt][ps] : zero8;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int ly = 2 * y + ky[t] - 2, lx = 2 * x + kx[t] - 2;
            const bool ok = px_ok && tap_ok[t] && cl_ok && ly >= 0 && ly < a.Hl && lx >= 0 && lx < a.Wl;
            bf[j][t] = ok ? lp[((size_t)ly * a.Wl + lx) * ls] : zero8;
          }
        }
#pragma unroll
        for (int j = 0; j < WG_PX; ++j)
#pragma unroll
          for (int mt = 0; mt < WG_MT; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
              acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[j][mt], bf[j][t], acc[mt][t], 0, 0, 0);
      }
    }
  }
  // D[row = small channel][col = large channel]: register q holds row (q&3) + 8(q>>2) + 4h of the tile, column r
  float *mine = a.part + (size_t)(blockIdx.z % n_strips) * a.Cs * a.Cl * 25;
#pragma unroll
  for (int mt = 0; mt < WG_MT; ++mt)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!tap_ok[t] || !cl_ok) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cs = cs0 + 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (cs < a.Cs) mine[((size_t)(tap0 + t) * a.Cs + cs) * a.Cl + cl] = acc[mt][t][q];
      }
    }
}

// dw[cs][cl][tap] += scale * sum over strips (in order) of part[strip][tap][cs][cl]
__global__ void wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dw, long n, int n_strips, float scale,
                                    long cscl) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long tap = i / cscl, pair = i - tap * cscl;  // i walks the partials' layout: coalesced reads
    float sum = 0.f;
    for (int s = 0; s < n_strips; ++s) sum += part[(size_t)s * n + i];
    float *d = dw + pair * 25 + tap;
    *d = fmaf(sum, scale, *d);
  }
}

// NCHW fp32 [B][C][HW] -> batch-minor fp16 pair [ceil(B/16)][HW][2][C][8 images]: y_hi = fp16(x),
// y_lo = fp16((x - y_hi) * 2^shift).  A block moves a 32-channel x 32-pixel tile of one group of 8 images through
// LDS, so both the reads (along pixels) and the writes (along channels) are coalesced.
__global__ __launch_bounds__(256) void nchw_split_bm8_kernel(const float *__restrict__ x, _Float16 *__restrict__ y_hi,
                                                            _Float16 *__restrict__ y_lo, int B, int C, long HW, float res_scale) {
  __shared__ float s[32][33][8];
  const int tid = threadIdx.x;
  const long p0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32, g = blockIdx.z;  // g = bc * 2 + h: images 8g .. 8g + 7
  for (int e = tid; e < 32 * 32 * 8; e += 256) {
    const int pl = e & 31, cl = (e >> 5) & 31, j = e >> 10;
    const int b = 8 * g + j, c = c0 + cl;
    const long p = p0 + pl;
    s[cl][pl][j] = (b < B && c < C && p < HW) ? x[((size_t)b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += 256) {
    const int cl = e & 31, pl = e >> 5;
    const int c = c0 + cl;
    const long p = p0 + pl;
    if (c >= C || p >= HW) continue;
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = s[cl][pl][j];
      hi[j] = (_Float16)v;
      lo[j] = (_Float16)((v - (float)hi[j]) * res_scale);
    }
    const size_t o = ((((size_t)(g >> 1) * HW + p) * 2 + (g & 1)) * C + c) * 8;
    *reinterpret_cast<half8 *>(y_hi + o) = hi;
    *reinterpret_cast<half8 *>(y_lo + o) = lo;
  }
}

}  // namespace licos

using namespace licos;

extern "C" int licos_nchw_f32_split_bm8(const float *x, void *y_hi, void *y_lo, int B, int C, int H, int W, int lo_shift,
                                        void *stream) {
  LICOS_REQUIRE(x && y_hi && y_lo && B > 0 && C > 0 && H > 0 && W > 0 && lo_shift >= 0 && lo_shift <= 24, "nchw_f32_split_bm8: bad arguments");
  const long HW = (long)H * W;
  const int groups = 2 * cdiv(B, 16);
  LICOS_REQUIRE(cdiv(C, 32) <= 65535 && groups <= 65535, "nchw_f32_split_bm8: too many channels / images");
  hipLaunchKernelGGL(nchw_split_bm8_kernel, dim3((unsigned)cdiv(HW, 32), cdiv(C, 32), groups), dim3(256), 0, as_stream(stream), x,
                     static_cast<_Float16 *>(y_hi), static_cast<_Float16 *>(y_lo), B, C, HW, ldexpf(1.f, lo_shift));
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

extern "C" int licos_wgrad5x5s2_strips(int Cs, int Cl, int Hs) {
  // strips of small-map rows so that the launch has a few thousand waves (13 tap pairs x channel tiles x strips)
  const long jobs = 13L * cdiv(Cl, 32) * cdiv(Cs, 128);
  long strips = 4096 / jobs;
  if (strips < 1) strips = 1;
  if (strips > Hs) strips = Hs;
  const int rows = cdiv(Hs, (int)strips);
  return cdiv(Hs, rows);
}

extern "C" int licos_wgrad5x5s2_f16(const void *small_bm8, const void *large_bm8, float *scratch, float *dw, int Cs, int Cl, int nbc,
                                    int Hs, int Ws, int Hl, int Wl, int scale_down, void *stream) {
  LICOS_REQUIRE(small_bm8 && large_bm8 && scratch && dw, "wgrad5x5s2_f16: NULL buffer");
  LICOS_REQUIRE(Cs > 0 && Cl > 0 && nbc > 0 && Hs > 0 && Ws > 0 && Hl > 0 && Wl > 0, "wgrad5x5s2_f16: bad shape");
  LICOS_REQUIRE(scale_down >= 0 && scale_down <= 63, "wgrad5x5s2_f16: bad scale");
  LICOS_REQUIRE((long)Hl * Wl * 2 * Cl < (1L << 31) && (long)Hs * Ws * 2 * Cs < (1L << 31), "wgrad5x5s2_f16: map too large");
  WgradArgs a{};
  a.small = static_cast<const half8 *>(small_bm8);
  a.large = static_cast<const half8 *>(large_bm8);
  a.part = scratch;
  a.Cs = Cs;
  a.Cl = Cl;
  a.nbc = nbc;
  a.Hs = Hs;
  a.Ws = Ws;
  a.Hl = Hl;
  a.Wl = Wl;
  const int n_strips = licos_wgrad5x5s2_strips(Cs, Cl, Hs);
  a.rows_per_strip = cdiv(Hs, n_strips);
  LICOS_REQUIRE(cdiv(Hs, a.rows_per_strip) == n_strips, "wgrad5x5s2_f16: internal strip arithmetic");
  hipLaunchKernelGGL(wgrad5x5s2_mfma_kernel, dim3(13, cdiv(Cl, 32), cdiv(Cs, 128) * n_strips), dim3(64), 0, as_stream(stream), a);
  LICOS_LAUNCH_CHECK();
  const long n = (long)Cs * Cl * 25;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), scratch, dw, n, n_strips,
                     ldexpf(1.f, -scale_down), (long)Cs * Cl);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}
