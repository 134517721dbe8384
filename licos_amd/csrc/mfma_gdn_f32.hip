// fp32 GDN / IGDN forward ([CAI] layers/gdn.py: norm = conv2d(x^2, gamma, beta); out = x * rsqrt(norm) or x * sqrt(norm))
// for 128 channels in ONE pass over the data: the 128 x 128 channel product of every pixel runs on the matrix cores at
// fp32 accuracy through the fp16 split - hi = fp16(v), lo = fp16((v - hi) * 2^11),
// D = hi.hi + 2^-11 (hi.lo + lo.hi), the cross terms in their own accumulators - with x^2 staged as (x / 16)^2 and gamma
// as 256 gamma so that activations in the hundreds stay inside fp16, as in the backward pass's 1x1 products.
//
// The vector-ALU kernel this replaces (conv_f32.hip gdn_f32_kernel: 128 fmas per output element) was 36 % of the fp32
// parity path's kernel time (25.6 ms per 1024 tiles at 128 x 128) and 1.2 ms of the 16-ms training step.  Here a PAIR of
// waves owns a 32-pixel tile: x (128 x 32 fp32) is staged in LDS once - it is the source of the B fragments (8 channels
// of one pixel per lane, squared and split in registers, by both waves) and the operand of the epilogue -, each wave
// takes 64 of the 128 output channels (two 32-row accumulator tiles and their cross-term twins), gamma's hi / lo A
// fragments are built once per workgroup and stay in LDS (64 KB), the next tile's x travels to registers while the
// current one multiplies.  8 waves per CU: one wave's conversions and epilogue run under the other's MFMAs.
//
// Epilogue: norm >= beta_min > 0, so rsqrt is v_rsq_f32 (1 ulp) + one Newton step (the correctly rounded sqrt and
// division of the first version were 25 instructions per element, more than the matrix work), sqrt = norm * rsqrt.
// SPLIT3: the result leaves as the split operand of the next layer's one-launch convolution (3 C fp16 channels,
// licos_nchw_f32_split3_blk16's layout) instead of NCHW fp32 - 6 bytes per element written instead of 4 written, 4 read
// again and 6 written by a separate split pass.  Traffic: 4 B in, 4 or 6 B out per element; 96 MFMAs per 32 pixels.
#include "mfma_common.hpp"

namespace licos {

constexpr int GF_C = 128, GF_PX = 32, GF_RS = 36;  // channels, pixels per tile, LDS row stride of the x tile (floats)
constexpr int GF_TILES = 4;                        // tiles in flight per workgroup (a pair of waves each)
constexpr int GF_GAMMA_BYTES = 4 * 8 * 2 * 64 * 16, GF_X_FLOATS = GF_C * GF_RS;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

template <bool INVERSE, bool SPLIT3>
__global__ __launch_bounds__(512) void gdn_f32_mfma_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float *__restrict__ y,
                                                           _Float16 *__restrict__ y3, float *__restrict__ n_out, int HW,
                                                           long tiles_total, int tiles_per_image) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_gam = reinterpret_cast<half8 *>(smem);                                   // [(it * 8 + ks) * 2 + part][lane]
  float *s_beta = reinterpret_cast<float *>(smem + GF_GAMMA_BYTES);                 // [128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave >> 1, half = wave & 1;                                      // tile slot; which 64 output channels
  float *s_x = s_beta + GF_C + slot * GF_X_FLOATS;                                  // this pair's [128][GF_RS] x tile
  const int p = lane & 31, h = lane >> 5;
  // gamma as A fragments of v_mfma_f32_32x32x16_f16: element e of lane (r, h) of fragment (it, ks) is
  // 256 gamma[32 it + r][16 ks + 8 h + e], split hi / lo
  for (int f = tid; f < 4 * 8 * 64; f += 512) {
    const int it = f >> 9, ks = (f >> 6) & 7, ln = f & 63, r = ln & 31, hh = ln >> 5;
    const float *g = gamma + (size_t)(32 * it + r) * GF_C + 16 * ks + 8 * hh;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = g[e] * 256.f;  // (a power of two: the fused multiply-convert and the rounded product agree)
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)((v - (float)hi[e]) * 2048.f);
    }
    s_gam[((it * 8 + ks) * 2 + 0) * 64 + ln] = hi;
    s_gam[((it * 8 + ks) * 2 + 1) * 64 + ln] = lo;
  }
  if (tid < GF_C) s_beta[tid] = beta[tid];

  const long stride = (long)gridDim.x * GF_TILES;
  // a tile's 128 x 32 floats = 1024 float4 (channel idx / 8, quad idx % 8): each wave of the pair fetches 8 per lane
  f32x4 nxt[8];
  auto fetch = [&](long tile) {
    const long b = tile / tiles_per_image;
    const float *xb = x + (size_t)b * GF_C * HW + (tile - b * tiles_per_image) * GF_PX;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = half * 512 + lane + 64 * k;
      nxt[k] = *reinterpret_cast<const f32x4 *>(xb + (unsigned)((idx >> 3) * HW + 4 * (idx & 7)));
    }
  };
  long t = (long)blockIdx.x * GF_TILES + slot;
  if (t < tiles_total) fetch(t);
  for (long t0 = (long)blockIdx.x * GF_TILES; t0 < tiles_total; t0 += stride, t += stride) {  // (workgroup-uniform trip count: barriers inside)
    const bool live = t < tiles_total;
    if (live) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = half * 512 + lane + 64 * k;
        *reinterpret_cast<f32x4 *>(s_x + (idx >> 3) * GF_RS + 4 * (idx & 7)) = nxt[k];
      }
    }
    __syncthreads();  // the tile is staged (first trip: gamma and beta too)
    if (t + stride < tiles_total) fetch(t + stride);  // the next tile's loads run under this tile's products
    if (live) {
      f32x16 acc[2], accx[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          acc[j][q] = s_beta[32 * (2 * half + j) + (q & 3) + 8 * (q >> 2) + 4 * h];
          accx[j][q] = 0.f;
        }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        half8 bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = s_x[(16 * ks + 8 * h + e) * GF_RS + p] * 0.0625f;
          const float sq = pin_f32(v * v);  // one value for the high part and its residual (mfma_common.hpp)
          bh[e] = (_Float16)sq;
          bl[e] = (_Float16)((sq - (float)bh[e]) * 2048.f);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int it = 2 * half + j;
          const half8 ah = s_gam[((it * 8 + ks) * 2 + 0) * 64 + lane], al = s_gam[((it * 8 + ks) * 2 + 1) * 64 + lane];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
          accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[j], 0, 0, 0);
          accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[j], 0, 0, 0);
        }
      }
      const long b = t / tiles_per_image;
      const int p0 = (int)(t - b * tiles_per_image) * GF_PX;
      // out[j][q]: the accumulator register's output channel is 32 (2 half + j) + (q & 3) + 8 (q >> 2) + 4 h
      float out[2][16];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int i = 32 * (2 * half + j) + (q & 3) + 8 * (q >> 2) + 4 * h;
          const float norm = acc[j][q] + accx[j][q] * (1.f / 2048.f);
          if (!SPLIT3 && n_out) n_out[(size_t)b * GF_C * HW + p0 + p + (unsigned)(i * HW)] = norm;  // kept for the backward pass
          const float r0 = __builtin_amdgcn_rsqf(norm);
          float sc;
          if (INVERSE) {
            const float s0 = norm * r0;
            sc = fmaf(fmaf(-s0, s0, norm), 0.5f * r0, s0);
          } else {
            sc = fmaf(0.5f * r0, fmaf(-norm * r0, r0, 1.f), r0);
          }
          out[j][q] = s_x[i * GF_RS + p] * sc;
        }
      if (!SPLIT3) {
        float *yb = y + (size_t)b * GF_C * HW + p0 + p;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int i = 32 * (2 * half + j) + (q & 3) + 8 * (q >> 2) + 4 * h;
            yb[(unsigned)(i * HW)] = out[j][q];
          }
      } else {
        // blk16 chunks of 3 C channels [hi 2^-5 | (v - hi) 2^6 | hi]; a lane holds channels {0-3, 8-11} (+ 4 h) of a chunk:
        // one v_permlane32_swap per dword hands the lower lane channels 0-7 and the upper lane 8-15 (16-byte stores)
        _Float16 *yb = y3 + ((size_t)b * 24 * HW + p0 + p) * 16 + 8 * h;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            unsigned lo3[3][2], hi3[3][2];
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              half2v ph[2], pm[2], pl[2];  // [low / high four channels]
#pragma unroll
              for (int side = 0; side < 2; ++side)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  const float v = pin_f32(out[j][8 * gp + 4 * side + 2 * d + e]);
                  const _Float16 hv = (_Float16)v;
                  ph[side][e] = hv;
                  pl[side][e] = (_Float16)((v - (float)hv) * 64.f);
                }
              const half2v k5 = {(_Float16)0.03125f, (_Float16)0.03125f};
              pm[0] = ph[0] * k5;  // (a power of two: the fp16 product rounds as the conversion of the fp32 product would)
              pm[1] = ph[1] * k5;
#pragma unroll
              for (int part = 0; part < 3; ++part) {
                const half2v *src = part == 0 ? pm : part == 1 ? pl : ph;
                const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, src[0]),
                                                                 __builtin_bit_cast(unsigned, src[1]), false, false);
                lo3[part][d] = sw[0];
                hi3[part][d] = sw[1];
              }
            }
            const int chunk = 2 * (2 * half + j) + gp;
#pragma unroll
            for (int part = 0; part < 3; ++part)
              *reinterpret_cast<uint4 *>(yb + (size_t)(part * 8 + chunk) * HW * 16) =
                  make_uint4(lo3[part][0], lo3[part][1], hi3[part][0], hi3[part][1]);
          }
      }
    }
    __syncthreads();  // every wave is done with the x tiles the next trip overwrites
  }
}

template <bool INVERSE, bool SPLIT3>
static int launch_gdn_f32(const float *x, const float *gamma_eff, const float *beta_eff, float *y, _Float16 *y3, float *n_out, int B,
                          long HW, hipStream_t s) {
  const int tiles_per_image = (int)(HW / GF_PX);
  const long tiles_total = (long)B * tiles_per_image;
  const size_t lds = (size_t)GF_GAMMA_BYTES + (size_t)GF_C * 4 + (size_t)GF_TILES * GF_X_FLOATS * 4;
  auto kern = gdn_f32_mfma_kernel<INVERSE, SPLIT3>;
  LICOS_ENSURE_LDS(kern, lds);
  const long want = (tiles_total + GF_TILES - 1) / GF_TILES;
  const int grid = (int)(want < 256 ? want : 256);  // one 140-KB workgroup per CU, each wave pair walking its share of the tiles
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, x, gamma_eff, beta_eff, y, y3, n_out, (int)HW, tiles_total, tiles_per_image);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int mfma_launch_gdn_f32(const float *x, const float *gamma_eff, const float *beta_eff, float *y, void *y_split3, float *norm_out,
                        int B, long HW, int inverse, hipStream_t s) {
  LICOS_REQUIRE((long)GF_C * HW * 16 < (1L << 31), "gdn_f32: an image's plane set must stay below 2^31 bytes (32-bit offsets)");
  _Float16 *y3 = static_cast<_Float16 *>(y_split3);
  if (y3) return inverse ? launch_gdn_f32<true, true>(x, gamma_eff, beta_eff, nullptr, y3, nullptr, B, HW, s)
                         : launch_gdn_f32<false, true>(x, gamma_eff, beta_eff, nullptr, y3, nullptr, B, HW, s);
  return inverse ? launch_gdn_f32<true, false>(x, gamma_eff, beta_eff, y, nullptr, norm_out, B, HW, s)
                 : launch_gdn_f32<false, false>(x, gamma_eff, beta_eff, y, nullptr, norm_out, B, HW, s);
}

}  // namespace licos
