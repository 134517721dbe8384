// The GDN gamma gradient on the matrix cores: dG[i][j] = sum over images b and pixels p of t[b][i][p] * x[b][j][p]^2
// (t = dL/dnorm; [CAI] layers/gdn.py norm = conv2d(x^2, gamma, beta), differentiated by train.py:193 loss.backward()) -
// a 128 x 128 Gram-type product whose K dimension is ALL pixels of the batch (up to 262 144 for cfg/raw_merged.toml's
// step), so the pixel axis is what an MFMA K step walks: both operands are NCHW fp32, 8 consecutive pixels of one channel
// are one lane's K fragment.
//
// fp32 accuracy through the three-pass split: every value v is staged in LDS as hi = fp16(v) and
// lo = fp16((v - hi) * 2^11); D = hi.hi + 2^-11 (hi.lo + lo.hi), the two parts in separate accumulators.  x^2 is staged as
// (x / 16)^2 (and the result multiplied by 256) so that activations in the hundreds stay inside fp16, as in the forward norm;
// t - a loss gradient, 1e-6 .. 1e-9 per element for a mean-reduced rate-distortion loss, i.e. BELOW fp16's normal range -
// is staged as t * 2^k with k chosen by a pre-pass from max|t| so that the largest magnitude lands near 2^14: values down
// to 2^-28 of the maximum keep a normal high part (exact: a power of two), the result is multiplied by 2^-k.
// A workgroup (4 waves; wave w owns output rows 32w .. 32w+31, all 128 columns) walks a contiguous range of 32-pixel
// slabs and writes ONE partial matrix; a second kernel adds the partials in workgroup order (no atomics: the gradient is
// bit-reproducible).
#include "mfma_common.hpp"

namespace licos {

constexpr int GR_C = 128, GR_PT = 32, GR_ROW = GR_PT / 8 + 1;  // 5 granules per channel row (odd stride); 40 KB of LDS

// stage one slab: [2 tensors][hi, lo][128 channels][GR_ROW granules of 8 pixels]
__device__ inline void gram_stage(const float *__restrict__ t, const float *__restrict__ x, size_t img_off, long HW, long p0,
                                  half8 *s, int tid, float t_scale) {
  // 2 tensors x 128 channels x 4 granules = 1024 granule jobs over 256 threads; a job = 8 consecutive pixels (two float4)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int job = tid + 256 * k;
    const int which = job >> 9, rem = job & 511, c = rem >> 2, gq = rem & 3;
    const float *src = (which ? x : t) + img_off + (size_t)c * HW + p0 + 8 * gq;
    float v[8];
    const long left = HW - (p0 + 8 * gq);
    if (left >= 8) {
      const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (e < left) ? src[e] : 0.f;
    }
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // (pin_f32: one fp32 value for both the high part and its residual - mfma_common.hpp)
      const float val = pin_f32(which ? (v[e] * 0.0625f) * (v[e] * 0.0625f) : v[e] * t_scale);
      hi[e] = (_Float16)val;
      lo[e] = (_Float16)((val - (float)hi[e]) * 2048.f);
    }
    s[((which * 2 + 0) * GR_C + c) * GR_ROW + gq] = hi;
    s[((which * 2 + 1) * GR_C + c) * GR_ROW + gq] = lo;
  }
}

// max |t| as the bit pattern of a non-negative float (ordered like an unsigned integer)
__global__ void gram_absmax_kernel(const float *__restrict__ t, long n, unsigned int *__restrict__ out) {
  unsigned int m = 0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const unsigned int b = __float_as_uint(t[e]) & 0x7FFFFFFFu;
    m = b > m && b < 0x7F800000u ? b : m;  // (infinities / NaNs do not choose the scale)
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned int o = __shfl_xor(m, off);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// 2^k with max|t| * 2^k in [2^13, 2^14)  (1 when t is all zero)
__device__ inline float gram_t_scale(unsigned int absmax_bits) {
  if (absmax_bits == 0) return 1.f;
  const int e = (int)(absmax_bits >> 23) - 127;  // floor(log2(max)) (denormal maxima: e = -127, clamped below)
  int k = 13 - e;
  k = k > 120 ? 120 : (k < -120 ? -120 : k);
  return __uint_as_float((unsigned int)(127 + k) << 23);
}

__global__ __launch_bounds__(256) void gram_partial_kernel(const float *__restrict__ t, const float *__restrict__ x, float *__restrict__ part,
                                                          const unsigned int *__restrict__ absmax, int B, long HW, int slabs_per_image,
                                                          int slabs_per_wg) {
  __shared__ __attribute__((aligned(16))) half8 s[2 * 2 * GR_C * GR_ROW];  // 40 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  f32x16 acc_hh[4], acc_x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc_hh[j][q] = 0.f;
      acc_x[j][q] = 0.f;
    }
  const float t_scale = gram_t_scale(*absmax);
  const long total = (long)B * slabs_per_image;
  const long s_begin = (long)blockIdx.x * slabs_per_wg, s_end = (s_begin + slabs_per_wg < total) ? s_begin + slabs_per_wg : total;
  for (long sl = s_begin; sl < s_end; ++sl) {
    const long b = sl / slabs_per_image, p0 = (sl - b * slabs_per_image) * GR_PT;
    __syncthreads();  // the previous slab's fragments have been read
    gram_stage(t, x, (size_t)b * GR_C * HW, HW, p0, s, tid, t_scale);
    __syncthreads();
    // A = t rows of this wave's tile (channel 32 wave + r), B = x^2 rows of tile j; K step = 16 pixels = granules 2ks + h
#pragma unroll
    for (int ks = 0; ks < GR_PT / 16; ++ks) {
      const half8 a_hi = s[((0 * 2 + 0) * GR_C + 32 * wave + r) * GR_ROW + 2 * ks + h];
      const half8 a_lo = s[((0 * 2 + 1) * GR_C + 32 * wave + r) * GR_ROW + 2 * ks + h];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const half8 b_hi = s[((1 * 2 + 0) * GR_C + 32 * j + r) * GR_ROW + 2 * ks + h];
        const half8 b_lo = s[((1 * 2 + 1) * GR_C + 32 * j + r) * GR_ROW + 2 * ks + h];
        acc_hh[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc_hh[j], 0, 0, 0);
        acc_x[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc_x[j], 0, 0, 0);
        acc_x[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc_x[j], 0, 0, 0);
      }
    }
  }
  // D[row][col]: register q of a lane is row (q&3) + 8(q>>2) + 4h of the tile, column r
  float *out = part + (size_t)blockIdx.x * GR_C * GR_C;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = 32 * wave + (q & 3) + 8 * (q >> 2) + 4 * h, col = 32 * j + r;
      out[row * GR_C + col] = (acc_hh[j][q] + acc_x[j][q] * (1.f / 2048.f)) * (256.f / t_scale);
    }
}

__global__ void gram_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, int n_part) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GR_C * GR_C) return;
  float sum = 0.f;
  for (int p = 0; p < n_part; ++p) sum += part[(size_t)p * GR_C * GR_C + i];
  out[i] = sum;
}

}  // namespace licos

using namespace licos;

extern "C" {

int licos_gdn_gamma_grad_parts(int B, long HW) {
  const long slabs = (long)B * ((HW + GR_PT - 1) / GR_PT);
  return (int)(slabs < 128 ? slabs : 128);  // enough workgroups to stream the operands, few enough partials to add cheaply
}

static int gamma_grad(const float *t, const float *x, float *scratch, float *dgamma, int B, int C, long HW, bool absmax_ready, void *stream);

int licos_gdn_gamma_grad_f32(const float *t, const float *x, float *scratch, float *dgamma, int B, int C, long HW, void *stream) {
  return gamma_grad(t, x, scratch, dgamma, B, C, HW, false, stream);
}

int licos_gdn_gamma_grad_scaled_f32(const float *t, const float *x, float *scratch, float *dgamma, int B, int C, long HW, void *stream) {
  return gamma_grad(t, x, scratch, dgamma, B, C, HW, true, stream);
}

static int gamma_grad(const float *t, const float *x, float *scratch, float *dgamma, int B, int C, long HW, bool absmax_ready, void *stream) {
  LICOS_REQUIRE(t && x && scratch && dgamma && B > 0 && HW > 0, "gdn_gamma_grad_f32: bad arguments");
  LICOS_REQUIRE(C == GR_C, "gdn_gamma_grad_f32: built for 128 channels (got %d)", C);
  LICOS_REQUIRE(HW % 4 == 0 && ((uintptr_t)t & 15) == 0 && ((uintptr_t)x & 15) == 0, "gdn_gamma_grad_f32: rows must be 16-byte aligned (H*W %% 4 == 0)");
  const int spi = (int)((HW + GR_PT - 1) / GR_PT);
  const long slabs = (long)B * spi;
  const int parts = licos_gdn_gamma_grad_parts(B, HW);
  const int per = (int)((slabs + parts - 1) / parts);
  const int grid = (int)((slabs + per - 1) / per);
  hipStream_t s = as_stream(stream);
  unsigned int *absmax = reinterpret_cast<unsigned int *>(scratch + (size_t)parts * GR_C * GR_C);  // the scratch's last word
  if (!absmax_ready) {  // (licos_gdn_bwd_fused_f32 leaves max|t| there as a by-product: no pass over t)
    LICOS_HIP_CHECK(hipMemsetAsync(absmax, 0, sizeof(unsigned int), s));
    const long nt = (long)B * GR_C * HW;
    hipLaunchKernelGGL(gram_absmax_kernel, dim3((int)((nt + 255) / 256 < 1024 ? (nt + 255) / 256 : 1024)), dim3(256), 0, s, t, nt, absmax);
    LICOS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(gram_partial_kernel, dim3(grid), dim3(256), 0, s, t, x, scratch, absmax, B, HW, spi, per);
  LICOS_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_reduce_kernel, dim3(GR_C * GR_C / 256), dim3(256), 0, s, scratch, dgamma, grid);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // extern "C"
