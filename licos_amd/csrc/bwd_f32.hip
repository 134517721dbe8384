// Backward kernels of the 32-bit path: weight / bias gradients of Conv2d and
// ConvTranspose2d and the GDN backward.  Input gradients of the convolutions need no kernel of their own:
// dgrad(Conv2d) is licos_deconv2d_f32 and dgrad(ConvTranspose2d) is licos_conv2d_f32 on the same weights.
//
// licos/train.py:193 (`out_criterion["loss"].backward()`) is where the reference runs this arithmetic
// through torch autograd; the maths restated here is the standard adjoint of the forward definitions in
// conv_f32.hip (cross-correlation) and of CompressAI layers/gdn.py.
#include <cmath>

#include "common.hpp"

namespace licos {

// ---------------------------------------------------------------------------------------------------
// Weight gradient as a GEMM over pixels:  dW[co][n] += sum_px g[co][px] * col[n][px],  n = ci*K*K + tap,
// col[n][px] = inp[ci][oy*S - pad + ky][ox*S - pad + kx]  (optionally squared: the GDN gamma gradient).
// Workgroup = 64 (co) x 64 (n) outputs, thread = 4 x 4; the pixel dimension is cut into 16-pixel steps
// dealt round-robin to gridDim.z slices whose partial sums meet in fp32 atomics.
constexpr int WG_M = 64, WG_N = 64, WG_K = 16;

__global__ __launch_bounds__(256) void conv2d_wgrad_f32_kernel(const float *__restrict__ inp, const float *__restrict__ g,
                                                               float *__restrict__ dw, int B, int Ci, int H, int W, int Co,
                                                               int Ho, int Wo, int K, int S, int pad, int square) {
  __shared__ __attribute__((aligned(16))) float s_a[WG_K][WG_M];  // g    [px][co]
  __shared__ __attribute__((aligned(16))) float s_b[WG_K][WG_N];  // col  [px][n]
  const int tid = threadIdx.x;
  const int tm = tid >> 4, tn = tid & 15;
  const int co0 = blockIdx.y * WG_M, n0 = blockIdx.x * WG_N;
  const int KK = K * K, N = Ci * KK;
  const int chunks_x = (Wo + WG_K - 1) / WG_K;
  const long chunks = (long)B * Ho * chunks_x;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (long ch = blockIdx.z; ch < chunks; ch += gridDim.z) {
    const int cx = (int)(ch % chunks_x);
    const int oy = (int)((ch / chunks_x) % Ho);
    const int b = (int)(ch / ((long)chunks_x * Ho));
    const int ox0 = cx * WG_K;
    __syncthreads();
    for (int e = tid; e < WG_K * WG_M; e += 256) {
      const int px = e % WG_K, m = e / WG_K;  // px fastest: coalesced reads of g rows
      const int co = co0 + m, ox = ox0 + px;
      float v = 0.f;
      if (co < Co && ox < Wo) v = g[(((size_t)b * Co + co) * Ho + oy) * Wo + ox];
      s_a[px][m] = v;
    }
    for (int e = tid; e < WG_K * WG_N; e += 256) {
      const int px = e % WG_K, nn = e / WG_K;
      const int n = n0 + nn, ox = ox0 + px;
      float v = 0.f;
      if (n < N && ox < Wo) {
        const int ci = n / KK, tap = n - ci * KK;
        const int iy = oy * S - pad + tap / K, ix = ox * S - pad + tap % K;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = inp[(((size_t)b * Ci + ci) * H + iy) * W + ix];
      }
      s_b[px][nn] = square ? v * v : v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < WG_K; ++k) {
      const float4 av = *reinterpret_cast<const float4 *>(&s_a[k][tm * 4]);
      const float4 bv = *reinterpret_cast<const float4 *>(&s_b[k][tn * 4]);
      const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + tm * 4 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tn * 4 + j;
      if (co < Co && n < N) atomicAdd(&dw[(size_t)co * N + n], acc[i][j]);
    }
  }
}

// db[c] = sum over (b, p) of dy[b][c][p]; one 1024-thread workgroup per channel, 16-byte loads, double accumulation
// in a fixed order (bit-reproducible)
__global__ __launch_bounds__(1024) void bias_grad_f32_kernel(const float *__restrict__ dy, float *__restrict__ db, int B,
                                                             int C, long HW) {
  __shared__ double s_red[16];
  const int c = blockIdx.x;
  double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
  if ((HW & 3) == 0) {
    const long n4 = HW >> 2;
    for (int b = 0; b < B; ++b) {
      const float4 *p = reinterpret_cast<const float4 *>(dy + ((size_t)b * C + c) * HW);
      for (long e = threadIdx.x; e < n4; e += 1024) {
        const float4 v = p[e];
        l0 += (double)v.x;
        l1 += (double)v.y;
        l2 += (double)v.z;
        l3 += (double)v.w;
      }
    }
  } else {
    for (int b = 0; b < B; ++b) {
      const float *p = dy + ((size_t)b * C + c) * HW;
      for (long e = threadIdx.x; e < HW; e += 1024) l0 += (double)p[e];
    }
  }
  double local = (l0 + l1) + (l2 + l3);
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += s_red[w];
    db[c] = (float)t;
  }
}

// ---------------------------------------------------------------------------------------------------
// GDN backward (y = x * n^p, n = beta + gamma . x^2, p = -1/2, or +1/2 for the inverse):
//   t_i  = dL/dn_i = dy_i * x_i * p * n_i^(p-1)
//   dx_i = dy_i * n_i^p + 2 x_i * sum_k gamma[k][i] t_k
// Workgroup = 64 pixels x all channels (as the forward kernel); t is also written out, because
// dgamma = sum_px t_i x_j^2 and dbeta = sum_px t_i are the weight / bias gradient kernels above applied
// to (x, t) as a 1x1 convolution.
__global__ __launch_bounds__(256) void gdn_bwd_f32_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                          const float *__restrict__ gamma, const float *__restrict__ gamma_t,
                                                          const float *__restrict__ beta, float *__restrict__ dx,
                                                          float *__restrict__ t_out, int C, int HW, int inverse, int keep_n) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];  // [C][64] x^2, [C][64] t, and (keep_n) [C][64] n
  float *s_sq = s_mem, *s_t = s_mem + (size_t)C * 64, *s_n = s_mem + (size_t)2 * C * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y;
  const int p = blockIdx.x * 64 + lane;
  const bool live = p < HW;
  const size_t base = (size_t)b * C * HW;
  for (int c = wave; c < C; c += 4) {
    const float v = live ? x[base + (size_t)c * HW + p] : 0.f;
    s_sq[c * 64 + lane] = v * v;
  }
  __syncthreads();
  // Each wave takes channels wave, wave + 4, ...; four of them share every LDS read of x^2 (or t): the dot products
  // are LDS-read bound otherwise.  Summation order per channel is unchanged (j ascending).
  for (int i0 = wave; i0 < C; i0 += 16) {
    float n[4];
    const float *gr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u;
      gr[u] = gamma + (size_t)(i < C ? i : 0) * C;
      n[u] = i < C ? beta[i] : 1.f;
    }
    for (int j = 0; j < C; ++j) {
      const float sq = s_sq[j * 64 + lane];
#pragma unroll
      for (int u = 0; u < 4; ++u) n[u] = fmaf(gr[u][j], sq, n[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u;
      if (i >= C) continue;
      const float xv = live ? x[base + (size_t)i * HW + p] : 0.f;
      const float g = live ? dy[base + (size_t)i * HW + p] : 0.f;
      // n^(p-1) * p: forward -1/2 * n^(-3/2); inverse +1/2 * n^(-1/2)
      const float rs = 1.0f / sqrtf(n[u]);
      const float t = inverse ? 0.5f * g * xv * rs : -0.5f * g * xv * rs / n[u];
      s_t[i * 64 + lane] = t;
      if (keep_n) s_n[i * 64 + lane] = n[u];
      if (live) t_out[base + (size_t)i * HW + p] = t;
    }
  }
  __syncthreads();
  for (int i0 = wave; i0 < C; i0 += 16) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, n[4];
    const float *gc[4], *gr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u;
      gc[u] = gamma_t + (size_t)(i < C ? i : 0) * C;  // row i of gamma^T = column i of gamma
      gr[u] = gamma + (size_t)(i < C ? i : 0) * C;
      n[u] = i < C ? beta[i] : 1.f;
    }
    for (int k = 0; k < C; ++k) {
      const float tv = s_t[k * 64 + lane];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = fmaf(gc[u][k], tv, acc[u]);
    }
    if (!keep_n) {  // C too wide to keep n in LDS: one more dot product per channel
      for (int j = 0; j < C; ++j) {
        const float sq = s_sq[j * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) n[u] = fmaf(gr[u][j], sq, n[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 4 * u;
      if (i >= C || !live) continue;
      const float nn = keep_n ? s_n[i * 64 + lane] : n[u];
      const float xv = x[base + (size_t)i * HW + p];
      const float g = dy[base + (size_t)i * HW + p];
      const float f = inverse ? sqrtf(nn) : 1.0f / sqrtf(nn);
      dx[base + (size_t)i * HW + p] = g * f + 2.0f * xv * acc[u];
    }
  }
}

// ---- GDN around an MFMA norm product (licos_conv1x1_f16): the element-wise halves ------------------------------
// mode 0: y = x * n^p                        (p = -1/2, +1/2 for the inverse)
// mode 1: t = dy * x * p * n^(p-1)           (dL/dn; also what dgamma / dbeta are reduced from)
// mode 2: dx = dy * n^p + 2 x u              (u = gamma^T . t)
__global__ void gdn_pointwise_f32_kernel(const float *__restrict__ x, const float *__restrict__ n, const float *__restrict__ dy,
                                         const float *__restrict__ u, float *__restrict__ out, long count, int inverse, int mode) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (long)gridDim.x * blockDim.x) {
    const float nn = n[e], xv = x[e];
    const float rs = 1.0f / sqrtf(nn);
    if (mode == 0) {
      out[e] = inverse ? xv * sqrtf(nn) : xv * rs;
    } else if (mode == 1) {
      const float g = dy[e];
      out[e] = inverse ? 0.5f * g * xv * rs : -0.5f * g * xv * rs / nn;
    } else {
      const float f = inverse ? sqrtf(nn) : rs;
      out[e] = dy[e] * f + 2.0f * xv * u[e];
    }
  }
}

// NonNegativeParametrizer backward: eff = max(raw, bound)^2 - pedestal with CompressAI's LowerBound gradient
// (passes where raw >= bound or the incoming gradient is negative):  d_raw = pass ? d_eff * 2 * max(raw, bound) : 0
__global__ void reparam_bwd_f32_kernel(const float *__restrict__ raw, const float *__restrict__ d_eff, float bound,
                                       float *__restrict__ d_raw, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float r = raw[e];
    const float gm = d_eff[e] * 2.0f * fmaxf(r, bound);
    d_raw[e] = (r >= bound || gm < 0.f) ? gm : 0.f;
  }
}

__global__ void transpose_sq_f32_kernel(const float *__restrict__ a, float *__restrict__ at, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * C) at[(size_t)(i % C) * C + i / C] = a[i];
}

__global__ void adam_f32_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                float *__restrict__ v, long n, float lr_t, float beta1, float beta2, float eps,
                                float inv_sqrt_bc2, float grad_scale) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float gr = g[e] * grad_scale;
    const float mm = beta1 * m[e] + (1.f - beta1) * gr;
    const float vv = beta2 * v[e] + (1.f - beta2) * gr * gr;
    m[e] = mm;
    v[e] = vv;
    p[e] -= lr_t * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
  }
}

__global__ __launch_bounds__(256) void sumsq_f32_kernel(const float *__restrict__ x, long n, double *__restrict__ out) {
  __shared__ double s_red[4];
  double local = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) local += (double)x[e] * x[e];
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

}  // namespace licos

using namespace licos;

extern "C" {

int licos_conv2d_wgrad_f32(const float *inp, const float *g, float *dw, int B, int Ci, int H, int W, int Co, int K,
                           int stride, int pad, int square_input, void *stream) {
  LICOS_REQUIRE(inp && g && dw, "conv2d_wgrad_f32: NULL buffer");
  LICOS_REQUIRE(B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0 && K > 0 && stride > 0 && pad >= 0, "conv2d_wgrad_f32: bad shape");
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  LICOS_REQUIRE(Ho > 0 && Wo > 0, "conv2d_wgrad_f32: empty output");
  const int N = Ci * K * K;
  const long chunks = (long)B * Ho * cdiv(Wo, WG_K);
  const int gx = cdiv(N, WG_N), gy = cdiv(Co, WG_M);
  // enough K-slices to fill the chip (>= ~2048 workgroups) without more atomics than useful
  long gz = 2048 / ((long)gx * gy) + 1;
  if (gz > chunks) gz = chunks;
  if (gz > 1024) gz = 1024;
  if (gz < 1) gz = 1;
  LICOS_HIP_CHECK(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Co * N, as_stream(stream)));
  hipLaunchKernelGGL(conv2d_wgrad_f32_kernel, dim3(gx, gy, (unsigned)gz), dim3(256), 0, as_stream(stream), inp, g, dw, B, Ci,
                     H, W, Co, Ho, Wo, K, stride, pad, square_input);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_bias_grad_f32(const float *dy, float *db, int B, int C, long HW, void *stream) {
  LICOS_REQUIRE(dy && db && B > 0 && C > 0 && HW > 0, "bias_grad_f32: bad arguments");
  hipLaunchKernelGGL(bias_grad_f32_kernel, dim3(C), dim3(1024), 0, as_stream(stream), dy, db, B, C, HW);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_gdn_bwd_f32(const float *x, const float *dy, const float *gamma_eff, const float *beta_eff, float *gamma_t_scratch,
                      float *dx, float *t_out, int B, int C, int HW, int inverse, void *stream) {
  LICOS_REQUIRE(x && dy && gamma_eff && beta_eff && gamma_t_scratch && dx && t_out, "gdn_bwd_f32: NULL buffer");
  LICOS_REQUIRE(B > 0 && B <= 65535 && C > 0 && HW > 0, "gdn_bwd_f32: bad shape");
  const int keep_n = (size_t)3 * C * 64 * sizeof(float) <= 160 * 1024;  // n kept in LDS between the two passes when it fits
  const size_t lds = (size_t)(keep_n ? 3 : 2) * C * 64 * sizeof(float);
  LICOS_REQUIRE(lds <= 160 * 1024, "gdn_bwd_f32: C=%d needs %zu B of LDS", C, lds);
  hipLaunchKernelGGL(transpose_sq_f32_kernel, dim3(cdiv((long)C * C, 256)), dim3(256), 0, as_stream(stream), gamma_eff,
                     gamma_t_scratch, C);
  LICOS_LAUNCH_CHECK();
  LICOS_ENSURE_LDS(gdn_bwd_f32_kernel, 160 * 1024);
  hipLaunchKernelGGL(gdn_bwd_f32_kernel, dim3(cdiv(HW, 64), B), dim3(256), lds, as_stream(stream), x, dy, gamma_eff,
                     gamma_t_scratch, beta_eff, dx, t_out, C, HW, inverse, keep_n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_gdn_pointwise_f32(const float *x, const float *n, const float *dy, const float *u, float *out, long count, int inverse,
                            int mode, void *stream) {
  LICOS_REQUIRE(x && n && out && count > 0 && mode >= 0 && mode <= 2, "gdn_pointwise_f32: bad arguments");
  LICOS_REQUIRE(mode == 0 || dy, "gdn_pointwise_f32: modes 1 and 2 need dy");
  LICOS_REQUIRE(mode != 2 || u, "gdn_pointwise_f32: mode 2 needs u");
  const int blocks = (int)((count + 255) / 256 < 16384 ? (count + 255) / 256 : 16384);
  hipLaunchKernelGGL(gdn_pointwise_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, dy, u, out, count, inverse, mode);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_reparam_bwd_f32(const float *raw, const float *d_eff, float bound, float *d_raw, long n, void *stream) {
  LICOS_REQUIRE(raw && d_eff && d_raw && n > 0, "reparam_bwd_f32: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(reparam_bwd_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), raw, d_eff, bound, d_raw, n);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_adam_f32(float *p, const float *g, float *m, float *v, long n, float lr, float beta1, float beta2, float eps,
                   int step, float grad_scale, void *stream) {
  LICOS_REQUIRE(p && g && m && v && n > 0 && step > 0, "adam_f32: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(adam_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n, (float)(lr / bc1), beta1,
                     beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_sumsq_f32(const float *x, long n, double *out, void *stream) {
  LICOS_REQUIRE(x && out && n > 0, "sumsq_f32: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(sumsq_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, out);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // extern "C"
