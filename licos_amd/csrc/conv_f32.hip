// 32-bit path: direct Conv2d / ConvTranspose2d / GDN on NCHW fp32.
//
// This is the precision-reference path of the product (latents within 1e-5 of the CPU
// oracle, symbols and therefore rANS bytes identical); it is a VALU kernel family with
// LDS-staged input patches and weights.  The throughput path is conv_mfma.hip.
#include "common.hpp"
#include "mfma_common.hpp"

namespace licos {

// ---------------------------------------------------------------------------------------------
// Conv2d: one workgroup = 8x32 output pixels x 16 output channels of one image.
// Thread t owns pixel (t / 32, t % 32) and 16 accumulators.  Input channels are consumed in
// chunks of CIB with the (8*s+K-s) x (32*s+K-s) patch and the [CIB][K*K][16] weight slab in LDS.
constexpr int TOH = 8, TOW = 32, COB = 16, CIB = 4;

template <int K, int S>
__global__ __launch_bounds__(256) void conv2d_f32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, float *__restrict__ y,
                                                         int Cin, int H, int W, int Cout, int Ho, int Wo, int pad,
                                                         int tiles_x, int flags) {
  const int relu = flags & LICOS_CONV_RELU;
  const bool abs_in = flags & LICOS_CONV_ABS_INPUT;
  constexpr int PH = (TOH - 1) * S + K, PW = (TOW - 1) * S + K;
  __shared__ float s_x[CIB][PH][PW + 1];
  __shared__ __attribute__((aligned(16))) float s_w[CIB][K * K][COB];
  const int tid = threadIdx.x;
  const int ty = tid / TOW, tx = tid % TOW;
  const int tile = blockIdx.x, b = blockIdx.z, co0 = blockIdx.y * COB;
  const int oy0 = (tile / tiles_x) * TOH, ox0 = (tile % tiles_x) * TOW;
  const int iy0 = oy0 * S - pad, ix0 = ox0 * S - pad;
  float acc[COB];
#pragma unroll
  for (int j = 0; j < COB; ++j) acc[j] = 0.f;
  const float *xb = x + (size_t)b * Cin * H * W;
  for (int c0 = 0; c0 < Cin; c0 += CIB) {
    __syncthreads();
    for (int e = tid; e < CIB * PH * PW; e += 256) {
      const int c = e / (PH * PW), r = (e / PW) % PH, q = e % PW;
      const int iy = iy0 + r, ix = ix0 + q, ci = c0 + c;
      float v = 0.f;
      if (ci < Cin && iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[((size_t)ci * H + iy) * W + ix];
      s_x[c][r][q] = abs_in ? fabsf(v) : v;
    }
    for (int e = tid; e < CIB * K * K * COB; e += 256) {
      const int j = e % COB, tap = (e / COB) % (K * K), c = e / (COB * K * K);
      const int ci = c0 + c, co = co0 + j;
      float v = 0.f;
      if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * (K * K) + tap];
      s_w[c][tap][j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float xv = s_x[c][ty * S + ky][tx * S + kx];
          const float4 *wp = reinterpret_cast<const float4 *>(&s_w[c][ky * K + kx][0]);
#pragma unroll
          for (int j4 = 0; j4 < COB / 4; ++j4) {
            const float4 wv = wp[j4];
            acc[j4 * 4 + 0] = fmaf(xv, wv.x, acc[j4 * 4 + 0]);
            acc[j4 * 4 + 1] = fmaf(xv, wv.y, acc[j4 * 4 + 1]);
            acc[j4 * 4 + 2] = fmaf(xv, wv.z, acc[j4 * 4 + 2]);
            acc[j4 * 4 + 3] = fmaf(xv, wv.w, acc[j4 * 4 + 3]);
          }
        }
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy < Ho && ox < Wo) {
#pragma unroll
    for (int j = 0; j < COB; ++j) {
      const int co = co0 + j;
      if (co < Cout) {
        float v = acc[j] + (bias ? bias[co] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        y[(((size_t)b * Cout + co) * Ho + oy) * Wo + ox] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ConvTranspose2d in gather form: out[oy][ox] = sum over taps with (oy + pad - ky) % S == 0 of
// in[(oy+pad-ky)/S][(ox+pad-kx)/S] * w[ci][co][ky][kx].  Same 8x32x16 tiling; the input patch of a
// tile spans rows floor((oy0+pad-K+1)/S) .. floor((oy0+TOH-1+pad)/S).
template <int K, int S>
__global__ __launch_bounds__(256) void deconv2d_f32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ bias, float *__restrict__ y,
                                                           int Cin, int H, int W, int Cout, int Ho, int Wo, int pad,
                                                           int tiles_x, int relu) {
  constexpr int PH = (TOH + K - 2) / S + 2, PW = (TOW + K - 2) / S + 2;
  __shared__ float s_x[CIB][PH][PW + 1];
  __shared__ __attribute__((aligned(16))) float s_w[CIB][K * K][COB];
  const int tid = threadIdx.x;
  const int ty = tid / TOW, tx = tid % TOW;
  const int tile = blockIdx.x, b = blockIdx.z, co0 = blockIdx.y * COB;
  const int oy0 = (tile / tiles_x) * TOH, ox0 = (tile % tiles_x) * TOW;
  // floor division of possibly negative numerators
  auto fdiv = [](int a, int d) { return (a >= 0) ? a / d : -((-a + d - 1) / d); };
  const int iy0 = fdiv(oy0 + pad - (K - 1), S), ix0 = fdiv(ox0 + pad - (K - 1), S);
  float acc[COB];
#pragma unroll
  for (int j = 0; j < COB; ++j) acc[j] = 0.f;
  const float *xb = x + (size_t)b * Cin * H * W;
  const int oy = oy0 + ty, ox = ox0 + tx;
  for (int c0 = 0; c0 < Cin; c0 += CIB) {
    __syncthreads();
    for (int e = tid; e < CIB * PH * PW; e += 256) {
      const int c = e / (PH * PW), r = (e / PW) % PH, q = e % PW;
      const int iy = iy0 + r, ix = ix0 + q, ci = c0 + c;
      float v = 0.f;
      if (ci < Cin && iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[((size_t)ci * H + iy) * W + ix];
      s_x[c][r][q] = v;
    }
    for (int e = tid; e < CIB * K * K * COB; e += 256) {
      const int j = e % COB, tap = (e / COB) % (K * K), c = e / (COB * K * K);
      const int ci = c0 + c, co = co0 + j;
      float v = 0.f;
      if (ci < Cin && co < Cout) v = w[((size_t)ci * Cout + co) * (K * K) + tap];
      s_w[c][tap][j] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int ny = oy + pad - ky;
        if (ny % S != 0 && S > 1) continue;  // (negative ny: % keeps the sign, still != 0 when odd)
        const int ry = fdiv(ny, S) - iy0;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int nx = ox + pad - kx;
          if (nx % S != 0 && S > 1) continue;
          const int rx = fdiv(nx, S) - ix0;
          const float xv = s_x[c][ry][rx];
          const float4 *wp = reinterpret_cast<const float4 *>(&s_w[c][ky * K + kx][0]);
#pragma unroll
          for (int j4 = 0; j4 < COB / 4; ++j4) {
            const float4 wv = wp[j4];
            acc[j4 * 4 + 0] = fmaf(xv, wv.x, acc[j4 * 4 + 0]);
            acc[j4 * 4 + 1] = fmaf(xv, wv.y, acc[j4 * 4 + 1]);
            acc[j4 * 4 + 2] = fmaf(xv, wv.z, acc[j4 * 4 + 2]);
            acc[j4 * 4 + 3] = fmaf(xv, wv.w, acc[j4 * 4 + 3]);
          }
        }
      }
    }
  }
  if (oy < Ho && ox < Wo) {
#pragma unroll
    for (int j = 0; j < COB; ++j) {
      const int co = co0 + j;
      if (co < Cout) {
        float v = acc[j] + (bias ? bias[co] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        y[(((size_t)b * Cout + co) * Ho + oy) * Wo + ox] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GDN reparametrisation: out = max(raw, bound)^2 - pedestal for beta (C) and gamma (C*C).
__global__ void gdn_reparam_kernel(const float *__restrict__ beta_raw, const float *__restrict__ gamma_raw,
                                   float beta_bound, float gamma_bound, float pedestal, float *__restrict__ beta_eff,
                                   float *__restrict__ gamma_eff, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    const float v = fmaxf(beta_raw[i], beta_bound);
    beta_eff[i] = v * v - pedestal;
  }
  if (i < C * C) {
    const float v = fmaxf(gamma_raw[i], gamma_bound);
    gamma_eff[i] = v * v - pedestal;
  }
}

// GDN: workgroup = 64 pixels of one image x all channels.  x^2 of the 64 pixels is staged in
// LDS [C][64]; wave g handles output channels g, g+4, ...; gamma rows are wave-uniform so they
// come through the scalar path.
__global__ __launch_bounds__(256) void gdn_f32_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float *__restrict__ y, int C,
                                                      int HW, int inverse) {
  extern __shared__ __attribute__((aligned(16))) float s_sq[];  // [C][64]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int p = blockIdx.x * 64 + lane;
  const bool live = p < HW;
  const float *xb = x + (size_t)b * C * HW;
  for (int c = wave; c < C; c += 4) {
    const float v = live ? xb[(size_t)c * HW + p] : 0.f;
    s_sq[c * 64 + lane] = v * v;
  }
  __syncthreads();
  for (int i = wave; i < C; i += 4) {
    const float *g = gamma + (size_t)i * C;
    float norm = beta[i];
    for (int j = 0; j < C; ++j) norm = fmaf(g[j], s_sq[j * 64 + lane], norm);
    if (live) {
      const float xv = xb[(size_t)i * HW + p];
      const float sc = inverse ? sqrtf(norm) : 1.0f / sqrtf(norm);
      y[((size_t)b * C + i) * HW + p] = xv * sc;
    }
  }
}

}  // namespace licos

using namespace licos;

extern "C" {

int licos_conv2d_f32(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int H, int W,
                     int Cout, int K, int stride, int pad, int relu /* LICOS_CONV_* flags */, void *stream) {
  LICOS_REQUIRE(x && w && y, "conv2d_f32: NULL buffer");
  LICOS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv2d_f32: bad shape B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
  LICOS_REQUIRE(pad >= 0 && pad < K, "conv2d_f32: pad %d unsupported for K=%d", pad, K);
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  LICOS_REQUIRE(Ho > 0 && Wo > 0, "conv2d_f32: empty output");
  const int tiles_x = cdiv(Wo, TOW), tiles_y = cdiv(Ho, TOH);
  LICOS_REQUIRE(B <= 65535 && cdiv(Cout, COB) <= 65535, "conv2d_f32: grid too large");
  dim3 grid(tiles_x * tiles_y, cdiv(Cout, COB), B), block(256);
  hipStream_t s = as_stream(stream);
#define LICOS_CONV_CASE(KK, SS)                                                                                     \
  if (K == KK && stride == SS) {                                                                                    \
    hipLaunchKernelGGL((conv2d_f32_kernel<KK, SS>), grid, block, 0, s, x, w, bias, y, Cin, H, W, Cout, Ho, Wo, pad, \
                       tiles_x, relu);                                                                              \
    LICOS_LAUNCH_CHECK();                                                                                           \
    return LICOS_OK;                                                                                                \
  }
  LICOS_CONV_CASE(5, 2)
  LICOS_CONV_CASE(3, 1)
  LICOS_CONV_CASE(1, 1)
  LICOS_CONV_CASE(5, 1)
  LICOS_CONV_CASE(3, 2)
#undef LICOS_CONV_CASE
  return fail(LICOS_EINVAL, "conv2d_f32: kernel %d stride %d not instantiated", K, stride);
}

int licos_deconv2d_f32(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int H, int W,
                       int Cout, int K, int stride, int pad, int out_pad, int relu, void *stream) {
  LICOS_REQUIRE(x && w && y, "deconv2d_f32: NULL buffer");
  LICOS_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "deconv2d_f32: bad shape");
  LICOS_REQUIRE(pad >= 0 && pad < K && out_pad >= 0 && out_pad < stride + (stride == 1), "deconv2d_f32: pad/out_pad unsupported");
  const int Ho = (H - 1) * stride - 2 * pad + K + out_pad, Wo = (W - 1) * stride - 2 * pad + K + out_pad;
  LICOS_REQUIRE(Ho > 0 && Wo > 0, "deconv2d_f32: empty output");
  const int tiles_x = cdiv(Wo, TOW), tiles_y = cdiv(Ho, TOH);
  LICOS_REQUIRE(B <= 65535 && cdiv(Cout, COB) <= 65535, "deconv2d_f32: grid too large");
  dim3 grid(tiles_x * tiles_y, cdiv(Cout, COB), B), block(256);
  hipStream_t s = as_stream(stream);
#define LICOS_DECONV_CASE(KK, SS)                                                                                     \
  if (K == KK && stride == SS) {                                                                                      \
    hipLaunchKernelGGL((deconv2d_f32_kernel<KK, SS>), grid, block, 0, s, x, w, bias, y, Cin, H, W, Cout, Ho, Wo, pad, \
                       tiles_x, relu);                                                                                \
    LICOS_LAUNCH_CHECK();                                                                                             \
    return LICOS_OK;                                                                                                  \
  }
  LICOS_DECONV_CASE(5, 2)
  LICOS_DECONV_CASE(3, 1)
  LICOS_DECONV_CASE(3, 2)
#undef LICOS_DECONV_CASE
  return fail(LICOS_EINVAL, "deconv2d_f32: kernel %d stride %d not instantiated", K, stride);
}

int licos_gdn_reparam_f32(const float *beta_raw, const float *gamma_raw, float beta_bound, float gamma_bound,
                          float pedestal, float *beta_eff, float *gamma_eff, int C, void *stream) {
  LICOS_REQUIRE(beta_raw && gamma_raw && beta_eff && gamma_eff && C > 0, "gdn_reparam_f32: bad arguments");
  hipLaunchKernelGGL(gdn_reparam_kernel, dim3(cdiv((long)C * C, 256)), dim3(256), 0, as_stream(stream), beta_raw,
                     gamma_raw, beta_bound, gamma_bound, pedestal, beta_eff, gamma_eff, C);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_gdn_f32_split3_applies(int C, int HW) { return C == 128 && HW > 0 && HW % 32 == 0 && HW < (1 << 20); }

int licos_gdn_f32_split3(const float *x, const float *gamma_eff, const float *beta_eff, void *y_blk16, int B, int C, int HW,
                         int inverse, void *stream) {
  LICOS_REQUIRE(x && gamma_eff && beta_eff && y_blk16 && B > 0, "gdn_f32_split3: bad arguments");
  LICOS_REQUIRE(licos_gdn_f32_split3_applies(C, HW), "gdn_f32_split3: C=%d HW=%d is not served by the one-pass kernel (ask licos_gdn_f32_split3_applies)", C, HW);
  LICOS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y_blk16 & 15) == 0, "gdn_f32_split3: buffers must be 16-byte aligned");
  return mfma_launch_gdn_f32(x, gamma_eff, beta_eff, nullptr, y_blk16, nullptr, B, HW, inverse, as_stream(stream));
}

int licos_gdn_f32_fwd_norm(const float *x, const float *gamma_eff, const float *beta_eff, float *y, float *norm_out, int B, int C,
                           int HW, int inverse, void *stream) {
  LICOS_REQUIRE(x && gamma_eff && beta_eff && y && norm_out && B > 0, "gdn_f32_fwd_norm: bad arguments");
  LICOS_REQUIRE(licos_gdn_f32_split3_applies(C, HW), "gdn_f32_fwd_norm: C=%d HW=%d is not served by the one-pass kernel (ask licos_gdn_f32_split3_applies)", C, HW);
  LICOS_REQUIRE(((uintptr_t)x & 15) == 0, "gdn_f32_fwd_norm: x must be 16-byte aligned");
  return mfma_launch_gdn_f32(x, gamma_eff, beta_eff, y, nullptr, norm_out, B, HW, inverse, as_stream(stream));
}

int licos_gdn_bwd_fused_f32(const float *x, const float *dy, const float *norm, const float *gamma_eff, float *dx, float *t_out,
                            void *t_absmax, int B, int C, int HW, int inverse, void *stream) {
  LICOS_REQUIRE(x && dy && norm && gamma_eff && dx && t_out && B > 0, "gdn_bwd_fused_f32: bad arguments");
  LICOS_REQUIRE(licos_gdn_f32_split3_applies(C, HW), "gdn_bwd_fused_f32: C=%d HW=%d is not served by the one-pass kernel (ask licos_gdn_f32_split3_applies)", C, HW);
  LICOS_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)norm | (uintptr_t)t_out) & 15) == 0, "gdn_bwd_fused_f32: buffers must be 16-byte aligned");
  return mfma_launch_gdn_bwd_f32(x, dy, norm, gamma_eff, dx, t_out, static_cast<unsigned int *>(t_absmax), B, HW, inverse, as_stream(stream));
}

int licos_gdn_f32(const float *x, const float *gamma_eff, const float *beta_eff, float *y, int B, int C, int HW,
                  int inverse, void *stream) {
  LICOS_REQUIRE(x && gamma_eff && beta_eff && y, "gdn_f32: NULL buffer");
  LICOS_REQUIRE(B > 0 && B <= 65535 && C > 0 && HW > 0, "gdn_f32: bad shape");
  // 128 channels, whole 32-pixel tiles, 16-byte aligned rows: the one-pass matrix-core kernel (mfma_gdn_f32.hip)
  static const bool use_mfma = [] { const char *e = getenv("LICOS_GDN_F32_MFMA"); return !(e && e[0] == '0'); }();
  if (use_mfma && licos_gdn_f32_split3_applies(C, HW) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0)
    return mfma_launch_gdn_f32(x, gamma_eff, beta_eff, y, nullptr, nullptr, B, HW, inverse, as_stream(stream));
  const size_t lds = (size_t)C * 64 * sizeof(float);
  LICOS_REQUIRE(lds <= 64 * 1024, "gdn_f32: C=%d needs %zu B of LDS (max 65536)", C, lds);
  hipLaunchKernelGGL(gdn_f32_kernel, dim3(cdiv(HW, 64), B), dim3(256), lds, as_stream(stream), x, gamma_eff, beta_eff,
                     y, C, HW, inverse);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // extern "C"
