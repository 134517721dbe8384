// Entropy bottleneck kernels (fp32): parameter packing, quantise, factorised likelihood,
// dequantise, and the reductions the loss / metrics need.
//
// Packed per-channel parameter record (licos_eb_pack), for filters f = (1, f1..fn, 1):
//   for layer i = 0..n:  softplus(matrix_i) [f(i+1)][f(i)] row-major, bias_i [f(i+1)],
//                        tanh(factor_i) [f(i+1)]   (factor block absent on the last layer)
#include "common.hpp"

namespace licos {

constexpr int EB_MAX_LAYERS = 8;
constexpr int EB_MAX_WIDTH = 16;

struct EbShape {
  int n_layers;              // len(filters) + 1
  int f[EB_MAX_LAYERS + 1];  // (1, filters..., 1)
  int per_channel;
};

static int make_shape(const int *filters, int nfilt, EbShape *s) {
  if (!filters || nfilt < 1 || nfilt + 1 > EB_MAX_LAYERS) return -1;
  s->n_layers = nfilt + 1;
  s->f[0] = 1;
  for (int i = 0; i < nfilt; ++i) {
    if (filters[i] < 1 || filters[i] > EB_MAX_WIDTH) return -1;
    s->f[i + 1] = filters[i];
  }
  s->f[nfilt + 1] = 1;
  int n = 0;
  for (int i = 0; i < s->n_layers; ++i) {
    n += s->f[i + 1] * s->f[i] + s->f[i + 1];
    if (i < s->n_layers - 1) n += s->f[i + 1];
  }
  s->per_channel = n;
  return 0;
}

struct EbPackArgs {
  const float *matrix[EB_MAX_LAYERS];
  const float *bias[EB_MAX_LAYERS];
  const float *factor[EB_MAX_LAYERS];
};

__device__ inline float softplus_f(float x) {
  // torch.nn.functional.softplus (beta=1, threshold=20)
  return x > 20.f ? x : log1pf(expf(x));
}

__global__ void eb_pack_kernel(EbPackArgs a, EbShape s, int C, float *__restrict__ packed) {
  const int c = blockIdx.x;
  if (c >= C) return;
  float *out = packed + (size_t)c * s.per_channel;
  int base = 0;
  for (int i = 0; i < s.n_layers; ++i) {
    const int rows = s.f[i + 1], cols = s.f[i];
    for (int e = threadIdx.x; e < rows * cols; e += blockDim.x)
      out[base + e] = softplus_f(a.matrix[i][(size_t)c * rows * cols + e]);
    base += rows * cols;
    for (int e = threadIdx.x; e < rows; e += blockDim.x) out[base + e] = a.bias[i][(size_t)c * rows + e];
    base += rows;
    if (i < s.n_layers - 1) {
      for (int e = threadIdx.x; e < rows; e += blockDim.x) out[base + e] = tanhf(a.factor[i][(size_t)c * rows + e]);
      base += rows;
    }
  }
}

// A workgroup's partial sum of log2-likelihoods, snapped to a multiple of 2^-20 before it is added to the per-image total
// with a double atomicAdd: sums of such multiples below 2^32 are exact in a double, hence independent of the order the
// workgroups arrive in - the total is bit-reproducible from run to run, like every other reduction of the training path
// (each partial itself is summed in a fixed order).  Costs < 5e-7 bit per workgroup.
__device__ inline double fixed_point_partial(double t) { return rint(t * 1048576.0) * (1.0 / 1048576.0); }

// ---- quantise ---------------------------------------------------------------------------------
__global__ void eb_quantize_kernel(const float *__restrict__ y, const float *__restrict__ medians,
                                   const float *__restrict__ noise, float *__restrict__ y_hat,
                                   int32_t *__restrict__ symbols, long ssb, long ssi, int mode, int C, int HW,
                                   long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int p = (int)(e % HW);
    const int c = (int)((e / HW) % C);
    const long b = e / ((long)HW * C);
    const float v = y[e];
    const float m = medians[c];
    if (mode == 1) {
      y_hat[e] = v + noise[e];
    } else {
      const float r = rintf(v - m);  // round-half-to-even, as torch.round
      if (y_hat) y_hat[e] = r + m;
      if (symbols) symbols[b * ssb + ((long)c * HW + p) * ssi] = (int32_t)r;
    }
  }
}

// Symbols in the coder's interleaved layout [position][stream]: a thread block owns 64 streams x 64
// positions and transposes through LDS so that both the latent reads (contiguous along positions)
// and the symbol writes (contiguous along streams) are coalesced.
__global__ __launch_bounds__(256) void eb_symbols_T_kernel(const float *__restrict__ y, const float *__restrict__ medians,
                                                           int32_t *__restrict__ symbols, long ssi, int B, int C, int HW) {
  __shared__ int32_t tile[64][65];
  const long n = (long)C * HW;
  const long i0 = (long)blockIdx.x * 64;
  const int b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {  // r: stream within the tile, tx: position
    const long i = i0 + tx;
    const int b = b0 + r;
    int32_t v = 0;
    if (i < n && b < B) {
      const int c = (int)(i / HW);
      v = (int32_t)rintf(y[(size_t)b * n + i] - medians[c]);
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {  // r: position, tx: stream
    const long i = i0 + r;
    const int b = b0 + tx;
    if (i < n && b < B) symbols[i * ssi + b] = tile[tx][r];
  }
}

// inverse: symbols [position][stream] -> y_hat as fp16 blk16 and/or NCHW fp32
__global__ __launch_bounds__(256) void eb_dequantize_T_kernel(const int32_t *__restrict__ symbols, long ssi,
                                                              const float *__restrict__ medians, float *__restrict__ y_nchw,
                                                              _Float16 *__restrict__ y_blk, int B, int C, int HW) {
  __shared__ int32_t tile[64][65];
  const long n = (long)C * HW;
  const long i0 = (long)blockIdx.x * 64;
  const int b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {  // r: position, tx: stream
    const long i = i0 + r;
    const int b = b0 + tx;
    tile[r][tx] = (i < n && b < B) ? symbols[i * ssi + b] : 0;
  }
  __syncthreads();
  const int C16 = (C + 15) / 16;
  for (int r = ty; r < 64; r += 4) {  // r: stream, tx: position
    const long i = i0 + tx;
    const int b = b0 + r;
    if (i < n && b < B) {
      const int c = (int)(i / HW), p = (int)(i - (long)c * HW);
      const float val = (float)tile[tx][r] + medians[c];
      if (y_nchw) y_nchw[(size_t)b * n + i] = val;
      if (y_blk) y_blk[(((size_t)b * C16 + (c >> 4)) * HW + p) * 16 + (c & 15)] = (_Float16)val;
    }
  }
}

// The decoder's case: blk16 fp16 output only.  A block owns one 16-channel chunk x 16 positions x 64 streams: the symbol
// reads are 256-byte rows along the streams, and every stream's 16 pixels x 16 channels leave as ONE 512-byte run (16 bytes
// per lane, consecutive lanes consecutive granules) - the kernel above writes blk16 two bytes at a time at a 32-byte
// stride.  Channels past C (C % 16 != 0) are written as zeros.
constexpr int DQ_ROW = 16 * 16 + 8;  // halfs per stream in LDS: 528 bytes (16-byte aligned rows)
__global__ __launch_bounds__(256) void eb_dequantize_blk_T_kernel(const int32_t *__restrict__ symbols, long ssi,
                                                                  const float *__restrict__ medians, _Float16 *__restrict__ y_blk,
                                                                  int B, int C, int HW) {
  __shared__ __attribute__((aligned(16))) _Float16 s_t[64 * DQ_ROW];
  const int groups = HW / 16;
  const int c16 = blockIdx.x / groups, p0 = (blockIdx.x - c16 * groups) * 16;
  const int b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 256; r += 4) {  // r = 16 * channel + position; tx: stream
    const int c = 16 * c16 + (r >> 4), p = r & 15, b = b0 + tx;
    float v = 0.f;
    if (c < C && b < B) v = (float)symbols[((long)c * HW + p0 + p) * ssi + b] + medians[c];
    s_t[tx * DQ_ROW + p * 16 + (r >> 4)] = (_Float16)v;
  }
  __syncthreads();
  const int C16 = (C + 15) / 16;
  for (int g = threadIdx.x; g < 64 * 32; g += 256) {  // 32 granules of 16 bytes per stream
    const int sl = g >> 5, gi = g & 31, b = b0 + sl;
    if (b < B) {
      const uint4 val = *reinterpret_cast<const uint4 *>(s_t + sl * DQ_ROW + gi * 8);
      *reinterpret_cast<uint4 *>(y_blk + (((size_t)b * C16 + c16) * HW + p0) * 16 + gi * 8) = val;
    }
  }
}

// ---- likelihood -------------------------------------------------------------------------------
__device__ inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// Evaluates the cumulative logits F(v) of one channel; parameters in LDS (s_p).
template <int F1, int F2, int F3, int F4>
__device__ inline float eb_logits_fixed(const float *s_p, float v) {
  // filters (F1,F2,F3,F4): five layers, all extents compile-time -> registers only
  constexpr int f[6] = {1, F1, F2, F3, F4, 1};
  float cur[EB_MAX_WIDTH], nxt[EB_MAX_WIDTH];
  cur[0] = v;
  int base = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int rows = f[i + 1], cols = f[i];
#pragma unroll
    for (int r = 0; r < rows; ++r) {
      // torch.matmul on (rows x cols) @ (cols x n): a dot product per output
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < cols; ++q) acc = fmaf(s_p[base + r * cols + q], cur[q], acc);
      nxt[r] = acc + s_p[base + rows * cols + r];
    }
    base += rows * cols + rows;
    if (i < 4) {
#pragma unroll
      for (int r = 0; r < rows; ++r) nxt[r] = nxt[r] + s_p[base + r] * tanhf(nxt[r]);
      base += rows;
    }
#pragma unroll
    for (int r = 0; r < rows; ++r) cur[r] = nxt[r];
  }
  return cur[0];
}

__device__ inline float eb_logits_generic(const float *s_p, const EbShape &s, float v) {
  float cur[EB_MAX_WIDTH], nxt[EB_MAX_WIDTH];
  cur[0] = v;
  int base = 0;
  for (int i = 0; i < s.n_layers; ++i) {
    const int rows = s.f[i + 1], cols = s.f[i];
    for (int r = 0; r < rows; ++r) {
      float acc = 0.f;
      for (int q = 0; q < cols; ++q) acc = fmaf(s_p[base + r * cols + q], cur[q], acc);
      nxt[r] = acc + s_p[base + rows * cols + r];
    }
    base += rows * cols + rows;
    if (i < s.n_layers - 1) {
      for (int r = 0; r < rows; ++r) nxt[r] = nxt[r] + s_p[base + r] * tanhf(nxt[r]);
      base += rows;
    }
    for (int r = 0; r < rows; ++r) cur[r] = nxt[r];
  }
  return cur[0];
}

// VARIANT: 0 generic, 1 = (3,3,3,3), 2 = (1,1,3,3), 3 = (13,13,3,3)
template <int VARIANT>
__global__ __launch_bounds__(256) void eb_likelihood_kernel(const float *__restrict__ v, const float *__restrict__ packed,
                                                            EbShape s, float *__restrict__ lik, float bound, int form,
                                                            double *__restrict__ sum_log2, int C, int HW) {
  extern __shared__ float s_p[];
  __shared__ double s_red[4];
  const int c = blockIdx.x, b = blockIdx.y;
  for (int e = threadIdx.x; e < s.per_channel; e += blockDim.x) s_p[e] = packed[(size_t)c * s.per_channel + e];
  __syncthreads();
  const size_t plane = ((size_t)b * C + c) * HW;
  double local = 0.0;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const float x = v[plane + p];
    float lo, up;
    if (VARIANT == 1) { lo = eb_logits_fixed<3, 3, 3, 3>(s_p, x - 0.5f); up = eb_logits_fixed<3, 3, 3, 3>(s_p, x + 0.5f); }
    else if (VARIANT == 2) { lo = eb_logits_fixed<1, 1, 3, 3>(s_p, x - 0.5f); up = eb_logits_fixed<1, 1, 3, 3>(s_p, x + 0.5f); }
    else if (VARIANT == 3) { lo = eb_logits_fixed<13, 13, 3, 3>(s_p, x - 0.5f); up = eb_logits_fixed<13, 13, 3, 3>(s_p, x + 0.5f); }
    else { lo = eb_logits_generic(s_p, s, x - 0.5f); up = eb_logits_generic(s_p, s, x + 0.5f); }
    float l;
    if (form == 0) {
      l = sigmoid_f(up) - sigmoid_f(lo);
    } else {
      const float t = lo + up;
      const float sg = (t > 0.f) ? -1.f : ((t < 0.f) ? 1.f : 0.f);
      l = fabsf(sigmoid_f(sg * up) - sigmoid_f(sg * lo));
    }
    l = fmaxf(l, bound);
    lik[plane + p] = l;
    local += (double)log2f(l);
  }
  if (sum_log2) {
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
      atomicAdd(&sum_log2[b], fixed_point_partial(t));
    }
  }
}

// ---- dequantise -------------------------------------------------------------------------------
__global__ void eb_dequantize_kernel(const int32_t *__restrict__ symbols, long ssb, long ssi,
                                     const float *__restrict__ medians, float *__restrict__ y_nchw,
                                     _Float16 *__restrict__ y_blk, int C, int H, int W, long total) {
  const int HW = H * W;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int p = (int)(e % HW);
    const int c = (int)((e / HW) % C);
    const long b = e / ((long)HW * C);
    const float val = (float)symbols[b * ssb + ((long)c * HW + p) * ssi] + medians[c];
    if (y_nchw) y_nchw[e] = val;
    if (y_blk) {
      const int C16 = (C + 15) / 16;
      y_blk[(((size_t)b * C16 + (c >> 4)) * HW + p) * 16 + (c & 15)] = (_Float16)val;
    }
  }
}

// ---- squared-difference reduction ---------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_sqdiff_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                            long n, int clamp01, double *__restrict__ out) {
  __shared__ double s_red[4];
  double local = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float av = a[e];
    if (clamp01) av = fminf(fmaxf(av, 0.f), 1.f);
    const float d = av - b[e];
    local += (double)(d * d);
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_red[wave] = local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
}

// ---- SSIM statistics of one scale (MS-SSIM metric) ------------------------------------------------------
// 11-tap Gaussian window, "valid" region, per plane sums of the ssim and cs maps.  A 256-thread block owns a 16x16
// patch of the (H-10)x(W-10) output map: the 26x26 inputs of both images go to LDS, a horizontal pass leaves the
// five filtered moments (x, y, xx, yy, xy) for 26 rows x 16 columns, the vertical pass finishes them per thread.
struct SsimWin { float w[11]; };

__global__ __launch_bounds__(256) void ssim_stats_kernel(const float *__restrict__ x, const float *__restrict__ y, int H, int W,
                                                         SsimWin win, float C1, float C2, double *__restrict__ sums) {
  __shared__ float s_in[2][26][27];
  __shared__ float s_h[5][26][17];
  __shared__ double s_red[2][4];
  const int plane = blockIdx.z, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16, Ho = H - 10, Wo = W - 10;
  const float *xp = x + (size_t)plane * H * W, *yp = y + (size_t)plane * H * W;
  for (int e = threadIdx.x; e < 26 * 26; e += 256) {
    const int r = e / 26, c = e - r * 26;
    const int iy = oy0 + r, ix = ox0 + c;
    const bool in = iy < H && ix < W;
    s_in[0][r][c] = in ? xp[(size_t)iy * W + ix] : 0.f;
    s_in[1][r][c] = in ? yp[(size_t)iy * W + ix] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 26 * 16; e += 256) {
    const int r = e >> 4, c = e & 15;
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float a = s_in[0][r][c + k], b = s_in[1][r][c + k], w = win.w[k];
      m[0] += w * a;
      m[1] += w * b;
      m[2] += w * (a * a);
      m[3] += w * (b * b);
      m[4] += w * (a * b);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) s_h[q][r][c] = m[q];
  }
  __syncthreads();
  double ssim_v = 0.0, cs_v = 0.0;
  if (oy0 + ty < Ho && ox0 + tx < Wo) {
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
      for (int q = 0; q < 5; ++q) m[q] += win.w[k] * s_h[q][ty + k][tx];
    const float mu1_sq = m[0] * m[0], mu2_sq = m[1] * m[1], mu1_mu2 = m[0] * m[1];
    const float sigma1_sq = m[2] - mu1_sq, sigma2_sq = m[3] - mu2_sq, sigma12 = m[4] - mu1_mu2;
    const float cs = (2.f * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2);
    cs_v = (double)cs;
    ssim_v = (double)(((2.f * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs);
  }
  for (int off = 32; off > 0; off >>= 1) {
    ssim_v += __shfl_down(ssim_v, off, 64);
    cs_v += __shfl_down(cs_v, off, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s_red[0][wave] = ssim_v;
    s_red[1][wave] = cs_v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(sums + 2 * plane, s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]);
    atomicAdd(sums + 2 * plane + 1, s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]);
  }
}

// ---- Gaussian conditional ---------------------------------------------------------------------------
__device__ inline float std_cumulative(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }

__global__ __launch_bounds__(256) void gc_likelihood_kernel(const float *__restrict__ v, const float *__restrict__ scales,
                                                            float *__restrict__ lik, float scale_bound, float lik_bound,
                                                            double *__restrict__ sum_log2, long per_image) {
  __shared__ double s_red[4];
  const int b = blockIdx.y;
  double local = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < per_image; e += (long)gridDim.x * blockDim.x) {
    const size_t i = (size_t)b * per_image + e;
    const float s = fmaxf(scales[i], scale_bound);
    const float a = fabsf(v[i]);
    float l = std_cumulative((0.5f - a) / s) - std_cumulative((-0.5f - a) / s);
    l = fmaxf(l, lik_bound);
    lik[i] = l;
    local += (double)log2f(l);
  }
  if (sum_log2) {
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sum_log2[b], fixed_point_partial(s_red[0] + s_red[1] + s_red[2] + s_red[3]));
  }
}

__global__ void gc_build_indexes_kernel(const float *__restrict__ scales, const float *__restrict__ table, int levels,
                                        float scale_bound, int32_t *__restrict__ indexes, long sb, long si, long n,
                                        long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / n, i = e - b * n;
    const float s = fmaxf(scales[e], scale_bound);
    int idx = levels - 1;
    for (int t = 0; t < levels - 1; ++t) idx -= (s <= table[t]) ? 1 : 0;
    indexes[b * sb + i * si] = idx;
  }
}

// 16-bit symbols in the plain [stream][position] layout, for the tiles the HOST codes (codec.compress_chunked /
// decompress_chunked: half the bytes of the int32 form over PCIe, whose copies are blit kernels that share the CUs with
// the transforms).  quantise: 4 positions per thread; `flag` is raised when a symbol does not fit (the caller then uses
// the 32-bit form).
__global__ __launch_bounds__(256) void eb_symbols16_kernel(const float *__restrict__ y, const float *__restrict__ medians,
                                                           int16_t *__restrict__ sym, int32_t *__restrict__ flag, int C, int HW,
                                                           long total) {
  const bool vec = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(sym) & 7) == 0;
  bool over = false;
  for (long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < total; e += (long)gridDim.x * blockDim.x * 4) {
    const int m = total - e < 4 ? (int)(total - e) : 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int16_t q[4];
    if (vec) {
      const float4 y4 = *reinterpret_cast<const float4 *>(y + e);
      v[0] = y4.x; v[1] = y4.y; v[2] = y4.z; v[3] = y4.w;
    } else {
      for (int j = 0; j < m; ++j) v[j] = y[e + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = (int)(((e + (vec ? 0 : j)) / HW) % C);  // (HW % 4 == 0: the four share a channel)
      const float r = rintf(v[j] - medians[c]);             // round-half-to-even, as torch.round
      over = over || (j < m && !(r >= -32768.f && r <= 32767.f));
      q[j] = (int16_t)(int32_t)fminf(fmaxf(r, -32768.f), 32767.f);
    }
    if (vec) *reinterpret_cast<short4 *>(sym + e) = make_short4(q[0], q[1], q[2], q[3]);
    else
      for (int j = 0; j < m; ++j) sym[e + j] = q[j];
  }
  if (over) atomicOr(flag, 1);
}

// dequantise to fp16 blk16 [B][C16][HW][16]: a block owns 16 channels x 64 positions of one stream - 8-byte reads along
// the positions, transposed through LDS, 16-byte writes of 8 channels per position
// (S = int16_t: the host coder's symbols; S = int32_t: the device decoder's stream-major ones, round 5 - `ssb`: symbols per stream row)
template <typename S>
__global__ __launch_bounds__(256) void eb_dequantize16_blk_kernel(const S *__restrict__ sym, long ssb, const float *__restrict__ medians,
                                                                  _Float16 *__restrict__ y_blk, int C, int HW) {
  __shared__ _Float16 tile[64][16 + 8];
  const int C16 = (C + 15) / 16;
  const int p0 = blockIdx.x * 64, c16 = blockIdx.y, b = blockIdx.z;
  {
    const int ch = threadIdx.x >> 4, qd = threadIdx.x & 15;
    const int c = c16 * 16 + ch;
    typedef S S4 __attribute__((ext_vector_type(4)));
    S4 q = {0, 0, 0, 0};
    float md = 0.f;
    if (c < C) {
      q = *reinterpret_cast<const S4 *>(sym + (size_t)b * ssb + (size_t)c * HW + p0 + 4 * qd);
      md = medians[c];
    }
    tile[4 * qd + 0][ch] = c < C ? (_Float16)((float)q.x + md) : (_Float16)0.f;
    tile[4 * qd + 1][ch] = c < C ? (_Float16)((float)q.y + md) : (_Float16)0.f;
    tile[4 * qd + 2][ch] = c < C ? (_Float16)((float)q.z + md) : (_Float16)0.f;
    tile[4 * qd + 3][ch] = c < C ? (_Float16)((float)q.w + md) : (_Float16)0.f;
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int pos = threadIdx.x >> 1, half = threadIdx.x & 1;
    const uint4 v = *reinterpret_cast<const uint4 *>(&tile[pos][8 * half]);
    *reinterpret_cast<uint4 *>(y_blk + (((size_t)b * C16 + c16) * HW + p0 + pos) * 16 + 8 * half) = v;
  }
}

// ... and to NCHW fp32 (the parity path)
__global__ __launch_bounds__(256) void eb_dequantize16_nchw_kernel(const int16_t *__restrict__ sym, const float *__restrict__ medians,
                                                                   float *__restrict__ y, int C, int HW, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    y[e] = (float)sym[e] + medians[(int)((e / HW) % C)];
}

// What the HOST coder needs of a tile's y stream, in as few bytes as PCIe allows (the host share of a scale-hyperprior
// call, codec.compress_hyper / decompress_hyper): one word per symbol, table row << 16 | (round(y) & 0xFFFF), for the
// encoder (`flag` is raised when a symbol does not fit 16 bits: the caller then codes the call on the device); one row
// byte per symbol for the decoder.  Plain [stream][position] layout; 4 elements per thread, 16-byte loads when the
// planes allow.
template <bool PACK>
__global__ __launch_bounds__(256) void gc_host_words_kernel(const float *__restrict__ y, const float *__restrict__ scales,
                                                           const float *__restrict__ table, int levels, float scale_bound,
                                                           int32_t *__restrict__ packed, uint8_t *__restrict__ rows8,
                                                           int32_t *__restrict__ flag, long total) {
  // (the outputs' alignment counts too: these are public C-ABI entry points, a caller may pass an offset view of a staging
  // buffer - that takes the element-wise path instead of a misaligned vector store)
  const bool vec = (total & 3) == 0 && ((reinterpret_cast<uintptr_t>(scales) | (PACK ? reinterpret_cast<uintptr_t>(y) : 0)) & 15) == 0 &&
                   (PACK ? (reinterpret_cast<uintptr_t>(packed) & 15) == 0 : (reinterpret_cast<uintptr_t>(rows8) & 3) == 0);
  bool over = false;
  for (long e = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; e < total; e += (long)gridDim.x * blockDim.x * 4) {
    float sv[4], yv[4] = {0.f, 0.f, 0.f, 0.f};
    const int m = total - e < 4 ? (int)(total - e) : 4;
    if (vec) {
      const float4 s4 = *reinterpret_cast<const float4 *>(scales + e);
      sv[0] = s4.x; sv[1] = s4.y; sv[2] = s4.z; sv[3] = s4.w;
      if (PACK) {
        const float4 y4 = *reinterpret_cast<const float4 *>(y + e);
        yv[0] = y4.x; yv[1] = y4.y; yv[2] = y4.z; yv[3] = y4.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sv[j] = j < m ? scales[e + j] : 0.f;
        if (PACK) yv[j] = j < m ? y[e + j] : 0.f;
      }
    }
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sv[j] = fmaxf(sv[j], scale_bound);
      c[j] = levels - 1;
    }
    for (int t = 0; t < levels - 1; ++t) {
      const float tv = table[t];
#pragma unroll
      for (int j = 0; j < 4; ++j) c[j] -= (sv[j] <= tv) ? 1 : 0;
    }
    if (PACK) {
      int32_t w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r = rintf(yv[j]);  // round-half-to-even, as torch.round
        over = over || !(r >= -32768.f && r <= 32767.f);
        w[j] = (int32_t)(((uint32_t)c[j] << 16) | ((uint32_t)(int32_t)fminf(fmaxf(r, -32768.f), 32767.f) & 0xFFFFu));
      }
      if (vec) *reinterpret_cast<int4 *>(packed + e) = make_int4(w[0], w[1], w[2], w[3]);
      else
        for (int j = 0; j < m; ++j) packed[e + j] = w[j];
    } else {
      if (vec) *reinterpret_cast<uchar4 *>(rows8 + e) = make_uchar4((uint8_t)c[0], (uint8_t)c[1], (uint8_t)c[2], (uint8_t)c[3]);
      else
        for (int j = 0; j < m; ++j) rows8[e + j] = (uint8_t)c[j];
    }
  }
  if (PACK && over) atomicOr(flag, 1);
}

// The same for the coder's interleaved layout [position][stream] (stride_b == 1, stride_i == B): a block owns 64
// streams x 64 positions and transposes through LDS, so that the scale reads (contiguous along positions) and the
// index writes (contiguous along streams) are both coalesced - written straight, every 4-byte index lands in a
// cache line of its own (15.6 ms per 2048 tiles of 13 x 512 x 512 instead of ~1).
__global__ __launch_bounds__(256) void gc_build_indexes_T_kernel(const float *__restrict__ scales, const float *__restrict__ table,
                                                                 int levels, float scale_bound, int32_t *__restrict__ indexes,
                                                                 long si, int B, long n) {
  __shared__ int32_t tile[64][65];
  const long i0 = (long)blockIdx.x * 64;
  const int b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {  // r: stream within the tile, tx: position
    const long i = i0 + tx;
    const int b = b0 + r;
    int idx = 0;
    if (i < n && b < B) {
      const float s = fmaxf(scales[(size_t)b * n + i], scale_bound);
      idx = levels - 1;
      for (int t = 0; t < levels - 1; ++t) idx -= (s <= table[t]) ? 1 : 0;
    }
    tile[r][tx] = idx;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {  // r: position, tx: stream
    const long i = i0 + r;
    const int b = b0 + tx;
    if (i < n && b < B) indexes[i * si + b] = tile[tx][r];
  }
}

// ---- granule pre/post-processing ---------------------------------------------------------
__global__ void dn12_to_grid8_kernel(const uint16_t *__restrict__ dn, float *__restrict__ out, long n, int full_range) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const double v = (double)dn[e] / 4095.0;
    out[e] = full_range ? (float)v : (float)(rint(v * 255.0) / 255.0);
  }
}

// Bilinear resampling of a batch of 2-D planes with torch.nn.functional.interpolate's arithmetic (ATen
// UpSampleKernel: source index = scale * dst (align_corners) or max(scale * (dst + .5) - .5, 0); i0 = min(int(src),
// in - 1); l1 = clamp(src - i0, 0, 1); a dimension whose size does not change is copied).
__device__ inline void bilinear_tap(int dst, int in, int out, float scale, int align, int &i0, int &i1, float &l0, float &l1) {
  if (in == out) {
    i0 = i1 = dst;
    l0 = 1.f;
    l1 = 0.f;
    return;
  }
  float src = align ? scale * (float)dst : scale * ((float)dst + 0.5f) - 0.5f;
  if (!align && src < 0.f) src = 0.f;
  i0 = min((int)src, in - 1);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l0 = 1.f - l1;
}

__global__ void resample_bilinear_kernel(const float *__restrict__ src, float *__restrict__ dst, long planes, int Hin, int Win,
                                         int Hout, int Wout, float scale_h, float scale_w, int align) {
  const long total = planes * Hout * Wout;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(e % Wout), oy = (int)((e / Wout) % Hout);
    const long p = e / ((long)Wout * Hout);
    int y0, y1, x0, x1;
    float hy0, hy1, wx0, wx1;
    bilinear_tap(oy, Hin, Hout, scale_h, align, y0, y1, hy0, hy1);
    bilinear_tap(ox, Win, Wout, scale_w, align, x0, x1, wx0, wx1);
    const float *s = src + p * (long)Hin * Win;
    const float top = __fadd_rn(__fmul_rn(wx0, s[(long)y0 * Win + x0]), __fmul_rn(wx1, s[(long)y0 * Win + x1]));
    const float bot = __fadd_rn(__fmul_rn(wx0, s[(long)y1 * Win + x0]), __fmul_rn(wx1, s[(long)y1 * Win + x1]));
    dst[e] = __fadd_rn(__fmul_rn(hy0, top), __fmul_rn(hy1, bot));
  }
}

// dir 0: image -> tiles, 1: tiles -> image.  One thread per tile element; x fastest (coalesced both sides).
// With a margin m > 0 tiles overlap: tile (iy, ix) covers image rows iy*S - m .. iy*S - m + T - 1, S = T - 2m, and only
// its central S x S pixels are written back - every reconstructed pixel then sits >= m pixels inside the tile that
// coded it (the transforms' zero padding at a tile border otherwise shows as a seam).
__global__ void tile_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int H, int W, int T, int m, int ny,
                            int nx, long total, int dir) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int tx = (int)(e % T);
    const int ty = (int)((e / T) % T);
    const int c = (int)((e / ((long)T * T)) % C);
    const long t = e / ((long)T * T * C);
    const int ix = (int)(t % nx), iy = (int)((t / nx) % ny);
    const long b = t / ((long)nx * ny);
    const int S = T - 2 * m;
    const int y = iy * S - m + ty, x = ix * S - m + tx;
    const bool in = y >= 0 && x >= 0 && y < H && x < W;
    const size_t img = (((size_t)b * C + c) * H + y) * W + x;
    if (dir == 0) dst[e] = in ? src[img] : 0.f;
    else if (in && ty >= m && ty < T - m && tx >= m && tx < T - m) dst[img] = src[e];
  }
}

__global__ void scale_f32_kernel(float *__restrict__ x, long n, float alpha, const float *__restrict__ inv_alpha) {
  const float a = inv_alpha ? alpha / inv_alpha[0] : alpha;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) x[e] *= a;
}

}  // namespace licos

using namespace licos;

extern "C" {

int licos_eb_packed_size(const int *filters, int nfilt) {
  EbShape s;
  if (make_shape(filters, nfilt, &s) != 0) return fail(LICOS_EINVAL, "eb: unsupported filters (need 1..%d layers of width 1..%d)", EB_MAX_LAYERS - 1, EB_MAX_WIDTH);
  return s.per_channel;
}

int licos_eb_pack(const float *const *matrices, const float *const *biases, const float *const *factors,
                  const int *filters, int nfilt, int C, float *packed, void *stream) {
  EbShape s;
  LICOS_REQUIRE(make_shape(filters, nfilt, &s) == 0, "eb_pack: unsupported filters");
  LICOS_REQUIRE(matrices && biases && factors && packed && C > 0, "eb_pack: bad arguments");
  EbPackArgs a{};
  for (int i = 0; i < s.n_layers; ++i) {
    a.matrix[i] = matrices[i];
    a.bias[i] = biases[i];
    a.factor[i] = (i < s.n_layers - 1) ? factors[i] : nullptr;
    LICOS_REQUIRE(a.matrix[i] && a.bias[i] && (i == s.n_layers - 1 || a.factor[i]), "eb_pack: NULL parameter for layer %d", i);
  }
  hipLaunchKernelGGL(eb_pack_kernel, dim3(C), dim3(64), 0, as_stream(stream), a, s, C, packed);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_eb_quantize(const float *y, const float *medians, const float *noise, float *y_hat, int32_t *symbols,
                      long ssb, long ssi, int mode, int B, int C, int HW, void *stream) {
  LICOS_REQUIRE(y && medians && B > 0 && C > 0 && HW > 0, "eb_quantize: bad arguments");
  LICOS_REQUIRE(mode >= 0 && mode <= 2, "eb_quantize: mode %d", mode);
  LICOS_REQUIRE(mode != 1 || (noise && y_hat), "eb_quantize: noise mode needs noise and y_hat");
  LICOS_REQUIRE(mode != 0 || y_hat, "eb_quantize: dequantize mode needs y_hat");
  LICOS_REQUIRE(mode != 2 || symbols, "eb_quantize: symbols mode needs symbols");
  if (mode == 2) y_hat = nullptr;
  if (mode == 1) symbols = nullptr;
  if (mode == 2 && ssb == 1 && B <= 65535 * 64) {
    const long n = (long)C * HW;
    hipLaunchKernelGGL(eb_symbols_T_kernel, dim3((unsigned)((n + 63) / 64), (B + 63) / 64), dim3(256), 0, as_stream(stream),
                       y, medians, symbols, ssi, B, C, HW);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  }
  const long total = (long)B * C * HW;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(eb_quantize_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), y, medians, noise, y_hat,
                     symbols, ssb, ssi, mode, C, HW, total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_eb_likelihood(const float *v, const float *packed, const int *filters, int nfilt, float *lik, float bound,
                        int form, double *sum_log2, int B, int C, int HW, void *stream) {
  EbShape s;
  LICOS_REQUIRE(make_shape(filters, nfilt, &s) == 0, "eb_likelihood: unsupported filters");
  LICOS_REQUIRE(v && packed && lik && B > 0 && B <= 65535 && C > 0 && HW > 0, "eb_likelihood: bad arguments");
  LICOS_REQUIRE(form == 0 || form == 1, "eb_likelihood: form %d", form);
  int variant = 0;
  if (nfilt == 4 && s.f[3] == 3 && s.f[4] == 3) {
    if (s.f[1] == 3 && s.f[2] == 3) variant = 1;
    else if (s.f[1] == 1 && s.f[2] == 1) variant = 2;
    else if (s.f[1] == 13 && s.f[2] == 13) variant = 3;
  }
  const size_t lds = (size_t)s.per_channel * sizeof(float);
  dim3 grid(C, B), block(HW >= 256 ? 256 : 64 * cdiv(HW, 64));
  hipStream_t st = as_stream(stream);
#define LICOS_EB_CASE(V)                                                                                          \
  case V:                                                                                                         \
    hipLaunchKernelGGL((eb_likelihood_kernel<V>), grid, block, lds, st, v, packed, s, lik, bound, form, sum_log2, \
                       C, HW);                                                                                    \
    break;
  switch (variant) {
    LICOS_EB_CASE(0)
    LICOS_EB_CASE(1)
    LICOS_EB_CASE(2)
    LICOS_EB_CASE(3)
  }
#undef LICOS_EB_CASE
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_eb_dequantize(const int32_t *symbols, long ssb, long ssi, const float *medians, float *y_nchw,
                        void *y_blk16, int B, int C, int H, int W, void *stream) {
  LICOS_REQUIRE(symbols && medians && (y_nchw || y_blk16) && B > 0 && C > 0 && H > 0 && W > 0, "eb_dequantize: bad arguments");
  if (ssb == 1 && B <= 65535 * 64 && !y_nchw && (H * W) % 16 == 0 && (long)(H * W / 16) * ((C + 15) / 16) < (1L << 31) &&
      ((uintptr_t)y_blk16 & 15) == 0) {
    hipLaunchKernelGGL(eb_dequantize_blk_T_kernel, dim3((unsigned)((H * W / 16) * ((C + 15) / 16)), (B + 63) / 64), dim3(256), 0,
                       as_stream(stream), symbols, ssi, medians, static_cast<_Float16 *>(y_blk16), B, C, H * W);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  }
  // stream-major symbols (what the round-5 plane decoder writes, 16 bytes per lane and four symbols): the 16-bit form's kernel
  if (ssi == 1 && !y_nchw && (H * W) % 64 == 0 && B <= 65535 && (C + 15) / 16 <= 65535 && (ssb & 3) == 0 && ((uintptr_t)symbols & 15) == 0 &&
      ((uintptr_t)y_blk16 & 15) == 0) {
    hipLaunchKernelGGL(eb_dequantize16_blk_kernel<int32_t>, dim3(H * W / 64, (C + 15) / 16, B), dim3(256), 0, as_stream(stream), symbols, ssb,
                       medians, static_cast<_Float16 *>(y_blk16), C, H * W);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  }
  if (ssb == 1 && B <= 65535 * 64) {
    const long n = (long)C * H * W;
    hipLaunchKernelGGL(eb_dequantize_T_kernel, dim3((unsigned)((n + 63) / 64), (B + 63) / 64), dim3(256), 0, as_stream(stream),
                       symbols, ssi, medians, y_nchw, static_cast<_Float16 *>(y_blk16), B, C, H * W);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  }
  const long total = (long)B * C * H * W;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(eb_dequantize_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), symbols, ssb, ssi, medians,
                     y_nchw, static_cast<_Float16 *>(y_blk16), C, H, W, total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_gc_likelihood(const float *v, const float *scales, float *lik, float scale_bound, float lik_bound,
                        double *sum_log2, int B, int C, int HW, void *stream) {
  LICOS_REQUIRE(v && scales && lik && B > 0 && B <= 65535 && C > 0 && HW > 0, "gc_likelihood: bad arguments");
  const long per = (long)C * HW;
  const int bx = (int)((per + 255) / 256 < 256 ? (per + 255) / 256 : 256);
  hipLaunchKernelGGL(gc_likelihood_kernel, dim3(bx, B), dim3(256), 0, as_stream(stream), v, scales, lik, scale_bound,
                     lik_bound, sum_log2, per);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_gc_build_indexes(const float *scales, const float *table, int levels, float scale_bound, int32_t *indexes,
                           long stride_b, long stride_i, int B, long n, void *stream) {
  LICOS_REQUIRE(scales && table && indexes && levels > 0 && B > 0 && n > 0, "gc_build_indexes: bad arguments");
  const long total = (long)B * n;
  if (stride_b == 1 && stride_i >= B && B >= 16 && (n + 63) / 64 < (1L << 31) && (B + 63) / 64 < 65536) {
    hipLaunchKernelGGL(gc_build_indexes_T_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((B + 63) / 64)), dim3(256), 0,
                       as_stream(stream), scales, table, levels, scale_bound, indexes, stride_i, B, n);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  }
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(gc_build_indexes_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), scales, table, levels,
                     scale_bound, indexes, stride_b, stride_i, n, total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_eb_symbols16(const float *y, const float *medians, int16_t *symbols, int32_t *flag, int B, int C, int HW, void *stream) {
  LICOS_REQUIRE(y && medians && symbols && flag && B > 0 && C > 0 && HW > 0, "eb_symbols16: bad arguments");
  const long total = (long)B * C * HW, quads = (total + 3) / 4;
  const int blocks = (int)((quads + 255) / 256 < 65536 ? (quads + 255) / 256 : 65536);
  hipLaunchKernelGGL(eb_symbols16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), y, medians, symbols, flag, C, HW, total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_eb_dequantize16(const int16_t *symbols, const float *medians, float *y_nchw, void *y_blk16, int B, int C, int H, int W,
                          void *stream) {
  LICOS_REQUIRE(symbols && medians && (y_nchw || y_blk16) && B > 0 && C > 0 && H > 0 && W > 0, "eb_dequantize16: bad arguments");
  const int HW = H * W;
  if (y_blk16) {
    LICOS_REQUIRE(HW % 64 == 0 && B <= 65535 && (C + 15) / 16 <= 65535 && ((uintptr_t)y_blk16 & 15) == 0 && ((uintptr_t)symbols & 7) == 0,
                  "eb_dequantize16: the blk16 form needs H * W to be a multiple of 64 and aligned buffers");
    hipLaunchKernelGGL(eb_dequantize16_blk_kernel<int16_t>, dim3(HW / 64, (C + 15) / 16, B), dim3(256), 0, as_stream(stream), symbols,
                       (long)C * HW, medians, static_cast<_Float16 *>(y_blk16), C, HW);
    LICOS_LAUNCH_CHECK();
  }
  if (y_nchw) {
    const long total = (long)B * C * HW;
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(eb_dequantize16_nchw_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), symbols, medians, y_nchw, C, HW, total);
    LICOS_LAUNCH_CHECK();
  }
  return LICOS_OK;
}

int licos_gc_pack_symbols(const float *y, const float *scales, const float *table, int levels, float scale_bound, int32_t *packed,
                          int32_t *flag, int B, long n, void *stream) {
  LICOS_REQUIRE(y && scales && table && packed && flag && levels > 0 && levels <= 65536 && B > 0 && n > 0, "gc_pack_symbols: bad arguments");
  const long total = (long)B * n, quads = (total + 3) / 4;
  const int blocks = (int)((quads + 255) / 256 < 65536 ? (quads + 255) / 256 : 65536);
  hipLaunchKernelGGL(gc_host_words_kernel<true>, dim3(blocks), dim3(256), 0, as_stream(stream), y, scales, table, levels, scale_bound,
                     packed, static_cast<uint8_t *>(nullptr), flag, total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_gc_build_rows8(const float *scales, const float *table, int levels, float scale_bound, uint8_t *rows8, int B, long n,
                         void *stream) {
  LICOS_REQUIRE(scales && table && rows8 && levels > 0 && levels <= 256 && B > 0 && n > 0, "gc_build_rows8: bad arguments (at most 256 table rows)");
  const long total = (long)B * n, quads = (total + 3) / 4;
  const int blocks = (int)((quads + 255) / 256 < 65536 ? (quads + 255) / 256 : 65536);
  hipLaunchKernelGGL(gc_host_words_kernel<false>, dim3(blocks), dim3(256), 0, as_stream(stream), static_cast<const float *>(nullptr),
                     scales, table, levels, scale_bound, static_cast<int32_t *>(nullptr), rows8, static_cast<int32_t *>(nullptr), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_dn12_to_grid8_f32(const uint16_t *dn, float *out, long n, int full_range, void *stream) {
  LICOS_REQUIRE(dn && out && n > 0, "dn12_to_grid8_f32: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(dn12_to_grid8_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), dn, out, n, full_range);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_resample_bilinear_f32(const float *src, float *dst, long planes, int Hin, int Win, int Hout, int Wout,
                                float scale_h, float scale_w, int align_corners, void *stream) {
  LICOS_REQUIRE(src && dst && planes > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "resample_bilinear_f32: bad arguments");
  const long total = planes * Hout * Wout;
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(resample_bilinear_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, planes, Hin, Win, Hout,
                     Wout, scale_h, scale_w, align_corners);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

static int tile_launch(const float *src, float *dst, int B, int C, int H, int W, int T, int m, int dir, void *stream) {
  LICOS_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && T > 0, "tile: bad arguments");
  LICOS_REQUIRE(m >= 0 && 2 * m < T, "tile: the margin must leave a positive tile core");
  const int ny = cdiv(H, T - 2 * m), nx = cdiv(W, T - 2 * m);
  const long total = (long)B * ny * nx * C * T * T;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(tile_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src, dst, C, H, W, T, m, ny, nx, total, dir);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_tile_f32(const float *img, float *tiles, int B, int C, int H, int W, int T, void *stream) {
  return tile_launch(img, tiles, B, C, H, W, T, 0, 0, stream);
}

int licos_untile_f32(const float *tiles, float *img, int B, int C, int H, int W, int T, void *stream) {
  return tile_launch(tiles, img, B, C, H, W, T, 0, 1, stream);
}

int licos_tile_overlap_f32(const float *img, float *tiles, int B, int C, int H, int W, int T, int margin, void *stream) {
  return tile_launch(img, tiles, B, C, H, W, T, margin, 0, stream);
}

int licos_untile_overlap_f32(const float *tiles, float *img, int B, int C, int H, int W, int T, int margin, void *stream) {
  return tile_launch(tiles, img, B, C, H, W, T, margin, 1, stream);
}

int licos_scale_f32(float *x, long n, float alpha, const float *inv_alpha_dev, void *stream) {
  LICOS_REQUIRE(x && n > 0, "scale_f32: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(scale_f32_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, alpha, inv_alpha_dev);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_ssim_stats_f32(const float *x, const float *y, int planes, int H, int W, const float *window11, float C1, float C2,
                         double *sums, void *stream) {
  LICOS_REQUIRE(x && y && window11 && sums && planes > 0, "ssim_stats_f32: bad arguments");
  LICOS_REQUIRE(H >= 11 && W >= 11, "ssim_stats_f32: planes must be at least 11 x 11 (the window)");
  LICOS_REQUIRE(planes <= 65535, "ssim_stats_f32: too many planes in one call");
  SsimWin win;
  for (int k = 0; k < 11; ++k) win.w[k] = window11[k];
  hipLaunchKernelGGL(ssim_stats_kernel, dim3(cdiv(W - 10, 16), cdiv(H - 10, 16), planes), dim3(256), 0, as_stream(stream), x, y, H, W,
                     win, C1, C2, sums);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_reduce_sqdiff(const float *a, const float *b, long n, int clamp01, double *out, void *stream) {
  LICOS_REQUIRE(a && b && out && n > 0, "reduce_sqdiff: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(reduce_sqdiff_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a, b, n, clamp01, out);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // extern "C"
