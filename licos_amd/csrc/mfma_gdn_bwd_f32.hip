// fp32 GDN / IGDN backward for 128 channels in ONE pass (what licos/train.py:193 `loss.backward()` computes through
// [CAI] layers/gdn.py):  with n = beta + gamma . x^2 kept by the forward kernel (mfma_gdn_f32.hip, norm_out) and p = -1/2
// (GDN) or +1/2 (IGDN),
//     t  = dy * x * p * n^(p-1)                 (dL/dn: what dgamma and dbeta are reduced from)
//     u  = gamma^T . t                          (128 x 128 channel product per pixel)
//     dx = dy * n^p + 2 x u
// Before: recompute n (three 1x1 MFMA passes + an operand split), a pointwise kernel for t, another split, three more
// passes for u, a pointwise kernel for dx - ~70 bytes of traffic per element and 11 launches per layer.  Here a pair of
// waves owns a 32-pixel tile as in the forward kernel: t is computed while the tile is staged (and written out, 16-byte
// stores), the product runs on the matrix cores through the fp16 split with gamma^T's fragments resident in LDS, the
// epilogue re-reads x, dy, n (L2: the workgroup loaded them a moment ago) and writes dx: 12 B in, 8 B out per element.
//
// Gradients have no natural scale (1e-3 .. 1e-12): every pixel's column of t is multiplied by a power of two that brings
// its largest element to [1, 2) before the split - the sum over channels is per pixel, so u's column is scaled back
// exactly - and nothing underflows fp16 whatever the loss scale.
#include "mfma_common.hpp"

namespace licos {

constexpr int GB_C = 128, GB_PX = 32, GB_RS = 36, GB_TILES = 4;
constexpr int GB_GAMMA_BYTES = 4 * 8 * 2 * 64 * 16, GB_T_FLOATS = GB_C * GB_RS;

typedef float f32x4g __attribute__((ext_vector_type(4)));

// n^(-1/2) to ~1 ulp: v_rsq_f32 + one Newton step (n >= beta_min > 0)
__device__ __forceinline__ float rsqrt_newton(float n) {
  const float r0 = __builtin_amdgcn_rsqf(n);
  return fmaf(0.5f * r0, fmaf(-n * r0, r0, 1.f), r0);
}

template <bool INVERSE>
__global__ __launch_bounds__(512) void gdn_bwd_f32_mfma_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                               const float *__restrict__ nrm, const float *__restrict__ gamma,
                                                               float *__restrict__ dx, float *__restrict__ t_out,
                                                               unsigned int *__restrict__ t_absmax, int HW, long tiles_total,
                                                               int tiles_per_image) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_gam = reinterpret_cast<half8 *>(smem);  // [(it * 8 + ks) * 2 + part][lane]: 256 gamma^T, hi / lo
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave >> 1, half = wave & 1;
  float *s_t = reinterpret_cast<float *>(smem + GB_GAMMA_BYTES) + slot * GB_T_FLOATS;  // this pair's [128][GB_RS] tile of t
  const int p = lane & 31, h = lane >> 5;
  // A fragments of v_mfma_f32_32x32x16_f16 for u[i] = sum_j gamma[j][i] t[j]: element e of lane (r, hh) of fragment
  // (it, ks) is 256 gamma[16 ks + 8 hh + e][32 it + r]
  for (int f = tid; f < 4 * 8 * 64; f += 512) {
    const int it = f >> 9, ks = (f >> 6) & 7, ln = f & 63, r = ln & 31, hh = ln >> 5;
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = gamma[(size_t)(16 * ks + 8 * hh + e) * GB_C + 32 * it + r] * 256.f;
      hi[e] = (_Float16)v;
      lo[e] = (_Float16)((v - (float)hi[e]) * 2048.f);
    }
    s_gam[((it * 8 + ks) * 2 + 0) * 64 + ln] = hi;
    s_gam[((it * 8 + ks) * 2 + 1) * 64 + ln] = lo;
  }

  const long stride = (long)gridDim.x * GB_TILES;
  // a tile's 128 x 32 floats = 1024 float4 (channel idx / 8, quad idx % 8): each wave of the pair handles 8 per lane.
  // tn: t of the NEXT tile, computed from loads that were issued before the current tile's products
  f32x4g tn[8];
  auto load_t = [&](long tile) {
    const long b = tile / tiles_per_image;
    const size_t base = (size_t)b * GB_C * HW + (tile - b * tiles_per_image) * GB_PX;
    int hw = HW;
    asm volatile("" : "+s"(hw));  // (as in the epilogue: keeps the piece offsets out of loop-invariant registers)
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 4) {  // (four pieces at a time: twelve 16-byte loads in flight, 48 registers)
      f32x4g xv[4], gv[4], nv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = half * 512 + lane + 64 * (k0 + k);
        const unsigned off = (unsigned)((idx >> 3) * hw + 4 * (idx & 7));
        xv[k] = *reinterpret_cast<const f32x4g *>(x + base + off);
        gv[k] = *reinterpret_cast<const f32x4g *>(dy + base + off);
        nv[k] = *reinterpret_cast<const f32x4g *>(nrm + base + off);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = half * 512 + lane + 64 * (k0 + k);
        const unsigned off = (unsigned)((idx >> 3) * hw + 4 * (idx & 7));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float rs = rsqrt_newton(nv[k][c]);
          // (gdn_pointwise_f32_kernel mode 1) IGDN: dy x / (2 sqrt n); GDN: - dy x / (2 n sqrt n)
          tn[k0 + k][c] = INVERSE ? 0.5f * gv[k][c] * xv[k][c] * rs : -0.5f * gv[k][c] * xv[k][c] * rs * (rs * rs);
        }
        *reinterpret_cast<f32x4g *>(t_out + base + off) = tn[k0 + k];
      }
      asm volatile("" ::: "memory");  // keeps the second batch's loads behind the first batch's arithmetic
    }
  };
  long t = (long)blockIdx.x * GB_TILES + slot;
  if (t < tiles_total) load_t(t);
  for (long t0 = (long)blockIdx.x * GB_TILES; t0 < tiles_total; t0 += stride, t += stride) {  // (workgroup-uniform trip count)
    const bool live = t < tiles_total;
    if (live) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = half * 512 + lane + 64 * k;
        *reinterpret_cast<f32x4g *>(s_t + (idx >> 3) * GB_RS + 4 * (idx & 7)) = tn[k];
      }
    }
    __syncthreads();  // the tile of t is staged (first trip: gamma^T too)
    if (live) {
      // this pixel's scale: largest |t| over its 128 channels -> a power of two that brings it to [1, 2)
      float m = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(s_t[(16 * ks + 8 * h + e) * GB_RS + p]));
      m = fmaxf(m, __shfl_xor(m, 32));
      if (t_absmax && half == 0) {  // max|t| of the whole tensor, a by-product the gamma-gradient kernel scales its operand by
        float mt = m;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) mt = fmaxf(mt, __shfl_xor(mt, d));
        if (lane == 0) atomicMax(t_absmax, __float_as_uint(mt));  // (bit patterns of non-negative floats order like the floats)
      }
      int ex = (int)((__float_as_uint(m) >> 23) & 255u) - 127;  // floor(log2 m); zeros / subnormals: the smallest scale
      ex = ex < -100 ? -100 : ex;
      const float up = __uint_as_float((unsigned)(127 - ex) << 23), down = __uint_as_float((unsigned)(127 + ex) << 23) * (1.f / 256.f);
      f32x16 acc[2], accx[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = accx[j][q] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        half8 bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = pin_f32(s_t[(16 * ks + 8 * h + e) * GB_RS + p] * up);  // one value for the high part and its residual
          bh[e] = (_Float16)v;
          bl[e] = (_Float16)((v - (float)bh[e]) * 2048.f);
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
      const size_t base = (size_t)b * GB_C * HW + (t - b * tiles_per_image) * GB_PX;  // (wave-uniform: scalar base + 32-bit lane offsets)
      const float *xb = x + base, *gb = dy + base, *nb = nrm + base;
      float *db = dx + base;
      int hw = HW;
      asm volatile("" : "+s"(hw));  // (opaque per trip: hoisted out of the loop, the 128 element addresses were spilled to scratch)
      // dx = dy n^p + 2 x u; the accumulator register's channel is 32 (2 half + j) + (q & 3) + 8 (q >> 2) + 4 h
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const unsigned off = (unsigned)((32 * (2 * half + j) + (q & 3) + 8 * (q >> 2) + 4 * h) * hw + p);
          const float u = (acc[j][q] + accx[j][q] * (1.f / 2048.f)) * down;
          const float nn = nb[off], rs = rsqrt_newton(nn);
          const float f = INVERSE ? nn * rs : rs;
          db[off] = fmaf(gb[off], f, 2.f * xb[off] * u);
        }
        asm volatile("" ::: "memory");  // one accumulator tile's 48 loads at a time
      }
    }
    if (t + stride < tiles_total) load_t(t + stride);  // (after the products: the loads' registers are free again)
    __syncthreads();  // every wave is done with the tiles the next trip overwrites
  }
}

template <bool INVERSE>
static int launch_gdn_bwd(const float *x, const float *dy, const float *norm, const float *gamma_eff, float *dx, float *t_out,
                          unsigned int *t_absmax, int B, long HW, hipStream_t s) {
  const int tiles_per_image = (int)(HW / GB_PX);
  const long tiles_total = (long)B * tiles_per_image;
  const size_t lds = (size_t)GB_GAMMA_BYTES + (size_t)GB_TILES * GB_T_FLOATS * 4;
  auto kern = gdn_bwd_f32_mfma_kernel<INVERSE>;
  LICOS_ENSURE_LDS(kern, lds);
  const long want = (tiles_total + GB_TILES - 1) / GB_TILES;
  const int grid = (int)(want < 256 ? want : 256);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, x, dy, norm, gamma_eff, dx, t_out, t_absmax, (int)HW, tiles_total, tiles_per_image);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int mfma_launch_gdn_bwd_f32(const float *x, const float *dy, const float *norm, const float *gamma_eff, float *dx, float *t_out,
                            unsigned int *t_absmax, int B, long HW, int inverse, hipStream_t s) {
  LICOS_REQUIRE((long)GB_C * HW * 16 < (1L << 31), "gdn_bwd_fused_f32: an image's plane set must stay below 2^31 bytes (32-bit offsets)");
  if (t_absmax) LICOS_HIP_CHECK(hipMemsetAsync(t_absmax, 0, sizeof(unsigned int), s));
  return inverse ? launch_gdn_bwd<true>(x, dy, norm, gamma_eff, dx, t_out, t_absmax, B, HW, s)
                 : launch_gdn_bwd<false>(x, dy, norm, gamma_eff, dx, t_out, t_absmax, B, HW, s);
}

}  // namespace licos
