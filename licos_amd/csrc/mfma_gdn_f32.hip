// fp32 GDN / IGDN forward ([CAI] layers/gdn.py: norm = conv2d(x^2, gamma, beta); out = x * rsqrt(norm) or x * sqrt(norm))
// for 128 channels in ONE pass over the data: the 128 x 128 channel product of every pixel runs on the matrix cores at
// fp32 accuracy through the three-pass fp16 split - hi = fp16(v), lo = fp16((v - hi) * 2^11),
// D = hi.hi + 2^-11 (hi.lo + lo.hi), the cross terms in their own accumulators - with x^2 staged as (x / 16)^2 and gamma
// as 256 gamma so that activations in the hundreds stay inside fp16, as in the backward pass's 1x1 products.
//
// The vector-ALU kernel this replaces (conv_f32.hip gdn_f32_kernel: 128 fmas per output element) was 36 % of the fp32
// parity path's kernel time (25.6 ms per 1024 tiles at 128 x 128) and 1.2 ms of the 16-ms training step.  Here a wave
// owns a 32-pixel tile: x (128 x 32 fp32) is staged in LDS once - it is both the source of the B fragments (8 channels
// of one pixel per lane, squared and split in registers) and the operand of the epilogue -, gamma's hi / lo A fragments
// are built once per workgroup and stay in LDS (64 KB), the next tile's x travels to registers while the current one
// multiplies.  Traffic: x in, y out (2 x 8.4 MB per 128 x 128 tile); 96 MFMAs per 32 pixels.
#include "mfma_common.hpp"

namespace licos {

constexpr int GF_C = 128, GF_PX = 32, GF_RS = 36;  // channels, pixels per wave tile, LDS row stride of the x tile (floats)
constexpr int GF_GAMMA_BYTES = 4 * 8 * 2 * 64 * 16, GF_X_FLOATS = GF_C * GF_RS;

__global__ __launch_bounds__(256) void gdn_f32_mfma_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, float *__restrict__ y, long HW,
                                                           long tiles_total, int tiles_per_image, int inverse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_gam = reinterpret_cast<half8 *>(smem);                                   // [(it * 8 + ks) * 2 + part][lane]
  float *s_beta = reinterpret_cast<float *>(smem + GF_GAMMA_BYTES);                 // [128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *s_x = s_beta + GF_C + wave * GF_X_FLOATS;                                  // this wave's [128][GF_RS] x tile
  const int p = lane & 31, h = lane >> 5;
  // gamma as A fragments of v_mfma_f32_32x32x16_f16: element e of lane (r, h) of fragment (it, ks) is
  // 256 gamma[32 it + r][16 ks + 8 h + e], split hi / lo
  for (int f = tid; f < 4 * 8 * 64; f += 256) {
    const int it = f >> 9, ks = (f >> 6) & 7, ln = f & 63, r = ln & 31, hh = ln >> 5;
    const float *g = gamma + (size_t)(32 * it + r) * GF_C + 16 * ks + 8 * hh;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = g[e] * 256.f;
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)((v - (float)hi[e]) * 2048.f);
    }
    s_gam[((it * 8 + ks) * 2 + 0) * 64 + ln] = hi;
    s_gam[((it * 8 + ks) * 2 + 1) * 64 + ln] = lo;
  }
  if (tid < GF_C) s_beta[tid] = beta[tid];
  __syncthreads();

  const long stride = (long)gridDim.x * 4;
  long t = (long)blockIdx.x * 4 + wave;
  // a tile's 128 x 32 floats = 1024 float4: lane fetches 16 of them (channel idx / 8, quad idx % 8)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 nxt[16];
  auto fetch = [&](long tile) {
    const long b = tile / tiles_per_image, p0 = (tile - b * tiles_per_image) * GF_PX;
    const float *xb = x + (size_t)b * GF_C * HW + p0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = lane + 64 * k;
      nxt[k] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(idx >> 3) * HW + 4 * (idx & 7));
    }
  };
  if (t < tiles_total) fetch(t);
  for (; t < tiles_total; t += stride) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = lane + 64 * k;
      *reinterpret_cast<f32x4 *>(s_x + (idx >> 3) * GF_RS + 4 * (idx & 7)) = nxt[k];
    }
    if (t + stride < tiles_total) fetch(t + stride);  // the next tile's loads run under this tile's products
    f32x16 acc[4], accx[4];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        acc[it][q] = s_beta[32 * it + (q & 3) + 8 * (q >> 2) + 4 * h];
        accx[it][q] = 0.f;
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
      for (int it = 0; it < 4; ++it) {
        const half8 ah = s_gam[((it * 8 + ks) * 2 + 0) * 64 + lane], al = s_gam[((it * 8 + ks) * 2 + 1) * 64 + lane];
        acc[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[it], 0, 0, 0);
        accx[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accx[it], 0, 0, 0);
        accx[it] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accx[it], 0, 0, 0);
      }
    }
    const long b = t / tiles_per_image, p0 = (t - b * tiles_per_image) * GF_PX;
    float *yb = y + (size_t)b * GF_C * HW + p0 + p;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 32 * it + (q & 3) + 8 * (q >> 2) + 4 * h;  // the accumulator register's output channel
        const float norm = acc[it][q] + accx[it][q] * (1.f / 2048.f);
        const float xv = s_x[i * GF_RS + p];
        const float sc = inverse ? sqrtf(norm) : 1.0f / sqrtf(norm);
        yb[(size_t)i * HW] = xv * sc;
      }
  }
}

int mfma_launch_gdn_f32(const float *x, const float *gamma_eff, const float *beta_eff, float *y, int B, long HW, int inverse,
                        hipStream_t s) {
  const int tiles_per_image = (int)(HW / GF_PX);
  const long tiles_total = (long)B * tiles_per_image;
  const size_t lds = (size_t)GF_GAMMA_BYTES + (size_t)GF_C * 4 + (size_t)4 * GF_X_FLOATS * 4;
  LICOS_ENSURE_LDS(gdn_f32_mfma_kernel, lds);
  const long want = (tiles_total + 3) / 4;
  const int grid = (int)(want < 256 ? want : 256);  // one 140-KB workgroup per CU, each wave walking its share of the tiles
  hipLaunchKernelGGL(gdn_f32_mfma_kernel, dim3(grid), dim3(256), lds, s, x, gamma_eff, beta_eff, y, HW, tiles_total,
                     tiles_per_image, inverse);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // namespace licos
