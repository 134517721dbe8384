// 16-bit MFMA path (gfx950): 5x5 stride-2 Conv2d (+GDN) and ConvTranspose2d (+IGDN) as LDS-tiled
// implicit GEMMs on v_mfma_f32_32x32x16_f16, fp32 accumulate.  Shared by mfma_conv.hip, mfma_deconv.hip and
// mfma_pack.hip.
//
// Orientation: D[cout][pixel] = sum_k Wp[cout][k] * X[k][pixel], k = (cin chunk of 16, tap, cin in
// chunk).  Weights are the MFMA A operand, activations the B operand, so an accumulator tile holds
// 32 output channels (registers) x 32 pixels (lanes).  That is the orientation in which
//   * the GDN norm  beta_i + sum_j gamma[i][j] x_j^2  is a second MFMA GEMM that takes the squared
//     accumulator tile as its B operand with no lane movement (MI355X guide, "accumulator tile as
//     the next MFMA's operand"): gamma is pre-packed k-permuted to match;
//   * a lane stores 8 consecutive channels of one pixel (16 B after one v_permlane32_swap per dword).
//
// Activations between stages: blk16 = [B][C/16][H][W][16] fp16 (32 B per pixel per 16-channel chunk).
// A workgroup (4 waves) owns 256 (NT=2) or 128 (NT=1) output pixels x all output channels; wave w owns NT
// pixel-tiles of 32.  The K loop walks (cin chunk) x (kernel row); each step's input rows and weight
// fragments arrive by LDS-DMA into the buffer the next step reads (see ConvStepGeom / DeconvStepGeom).
// LDS granules (16 B = 8 channels of a pixel) are arranged so the 32 lanes of a B-fragment read
// (consecutive output x) hit consecutive granules: conflict-free ds_read_b128.
#pragma once
#include "common.hpp"
#include <utility>

namespace licos {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int EPI_NONE = LICOS_EPI_NONE, EPI_GDN = LICOS_EPI_GDN, EPI_IGDN = LICOS_EPI_IGDN, EPI_RELU = LICOS_EPI_RELU;
// internal: (I)GDN with LICOS_EPI_NORM32 - the norm at fp32 accuracy (fp16-split gamma fragments, licos_pack_gdn_f32split)
constexpr int EPI_GDN32 = 4, EPI_IGDN32 = 5;
__host__ __device__ constexpr bool epi_norm(int epi) { return epi == EPI_GDN || epi == EPI_IGDN || epi == EPI_GDN32 || epi == EPI_IGDN32; }
__host__ __device__ constexpr bool epi_norm32(int epi) { return epi == EPI_GDN32 || epi == EPI_IGDN32; }
// 16-byte granules of the packed gamma fragments an epilogue keeps in LDS
__host__ __device__ constexpr int epi_gamma_gran(int epi, int MT) { return epi_norm(epi) ? MT * MT * 2 * 64 * (epi_norm32(epi) ? 2 : 1) : 0; }

// Pins `v` to ONE fp32 register value.  A split operand needs it: with contraction on (HIP's default), `(_Float16)(a * a)`
// may compile to a fused multiply-convert (v_fma_mixlo_f16: the EXACT product rounded to fp16 once) at one use and to
// v_cvt_pk_f16_f32 of the fp32-rounded product at another.  In the rare double-rounding cases (~5 per 100 000 values)
// the two high parts differ by one fp16 ulp, the residual is then taken against the wrong one, and the element is off
// by 2^-11 relative (found by the fp32 GDN parity test).  __fmul_rn alone does not stop it.
__device__ __forceinline__ float pin_f32(float v) {
  asm("" : "+v"(v));
  return v;
}
__host__ __device__ constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- geometry shared by host and device ---------------------------------------------------------
// Double-buffered stride-2 conv (mfma_conv.hip): per K-step (cin chunk, kernel row ky) only the TH input
// rows 2*ty + ky that step reads are staged, as granules [half][ty][x-parity][x/2].
template <int MT, int TH, int TW>
struct ConvStepGeom {
  static constexpr int PWH = (TW == 16) ? 24 : round_up(TW + 2, 4);  // 2*PWH % 16 == 0 keeps 2-row tiles conflict-free
  static constexpr int ROWG = 2 * PWH;            // granules per (half, ty): both parities
  static constexpr int HALF = TH * ROWG;
  static constexpr int PATCH_GRAN = 2 * HALF;     // multiple of 64 for every instantiated tile
  static constexpr int W_GRAN = 5 * MT * 64;      // one kernel row of A fragments
  static constexpr int BUF_GRAN = PATCH_GRAN + W_GRAN;
  static constexpr int GAMMA_GRAN = MT * MT * 2 * 64;
  static constexpr int LDS_BYTES = 16 * ((2 * BUF_GRAN > GAMMA_GRAN) ? 2 * BUF_GRAN : GAMMA_GRAN);
  static_assert(PATCH_GRAN % 64 == 0, "patch must be a whole number of wave-wide LDS-DMA pieces");
};

// XCD-aware work mapping (speed only, never correctness).  Workgroups are dealt round-robin over the 8
// XCDs, each with a private L2, so linear ids l and l+8 share an L2.  The grid is 1-D over B images x
// `per_image` work items; this remap hands each XCD whole images (8 images in flight, one per XCD, their
// items advancing together) so that tiles sharing halo rows - and the 4 output phases of a deconv tile,
// which read the same patch and write the two halves of the same cache lines - meet in one L2.
__device__ inline void xcd_work_item(int l, int B, int per_image, int &b, int &item) {
  const int group = 8 * per_image;
  const int g = l / group, r = l - g * group;
  if ((g + 1) * 8 <= B) {
    b = g * 8 + (r & 7);
    item = r >> 3;
  } else {  // tail group with fewer than 8 images: plain order
    b = g * 8 + r / per_image;
    item = r % per_image;
  }
}

// 16 B global -> LDS without a register round trip: LDS address = wave-uniform base + lane * 16.
// Issued as inline assembly on purpose.  With the builtin (__builtin_amdgcn_global_load_lds) in flight the compiler
// treats lgkmcnt as unordered and turns every wait in front of an MFMA into s_waitcnt lgkmcnt(0) - also for
// ds_reads issued two instructions earlier - which defeats reading LDS fragments ahead of their use.  The hardware
// counts an LDS-DMA in vmcnt only, and every kernel here already orders its LDS reads behind the data with an explicit
// `s_waitcnt vmcnt(0)` + s_barrier, so nothing depends on the compiler knowing about the transfer.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
#ifdef LICOS_GLDS_BUILTIN  // A/B build only (tools/ab_build.sh): the compiler-visible form
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
  return;
#endif
  const unsigned m0v = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
  unsigned keep;  // M0 is compiler-reserved: written and restored inside the one statement that uses it
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(m0v));
}

// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit byte offset per lane: no address VGPR pair per request
// (a kernel that requests a dozen different pieces per step otherwise keeps a dozen hoisted 64-bit addresses alive).
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, void *lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(
      (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
  const uint64_t bits = reinterpret_cast<uint64_t>(sbase);
  const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bits >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bits);
  unsigned keep;
  uint64_t base_copy;
  // (the base goes through an s_mov inside the statement: a base the compiler has just produced with v_readfirstlane /
  // v_readlane needs five wait states before a vector-memory instruction may read it as its scalar address, and the
  // compiler pads nothing inside an asm statement; an SALU read of it is interlocked)
  asm volatile("s_mov_b64 %1, %4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(base_copy)
               : "v"(voff), "s"(m0v), "s"(sb));
}

// Transposed conv (mfma_deconv.hip): a workgroup owns ONE output phase (py, px) of a TH x TW input tile.  The
// (TH+2) x (TW+2) input patch of a cin chunk is staged once, as granules [half][row][x], double-buffered by chunk;
// a K-step is (cin chunk, kernel row of the phase): it reads the patch at row offset dy = 1 - iky and needs that
// kernel row's 2 or 3 taps of weight fragments, double-buffered by step.
template <int MT, int TH, int TW>
struct DeconvStepGeom {
  static constexpr int RS = (TW == 16) ? 32 : round_up(TW + 2, 4);  // row stride; % 16 == 0 for 2-row pixel tiles
  static constexpr int PH = TH + 2;                // input rows ty0-1 .. ty0+TH: every kernel row reads TH of them
  static constexpr int HALF = PH * RS;
  static constexpr int PATCH_GRAN = 2 * HALF;
  static constexpr int PQ = (PATCH_GRAN + 63) / 64;  // wave-wide LDS-DMA pieces per patch
  static constexpr int PATCH_PAD = PQ * 64;
  static constexpr int W_GRAN_MAX = 3 * MT * 64;   // one kernel row of a phase: up to 3 taps
  static constexpr int KLOOP_GRAN = 2 * PATCH_PAD + 2 * W_GRAN_MAX;
  static constexpr int GAMMA_GRAN = MT * MT * 2 * 64;
  static constexpr int LDS_BYTES = 16 * ((KLOOP_GRAN > GAMMA_GRAN) ? KLOOP_GRAN : GAMMA_GRAN);
};

template <int TH, int TW>
struct DeconvGeom {  // full (TH+2) x (TW+2) input patch of a transposed-conv tile
  static constexpr int PH = TH + 2;
  static constexpr int PW = TW + 2;
  static constexpr int RS = (TW == 16) ? 32 : round_up(TW + 2, 4);
};

struct MfmaArgs {
  const _Float16 *x;      // blk16 input
  const half8 *wp;        // packed weights (A fragments)
  const float *bias;      // [32*MT] fp32 (zero padded)
  const bf16x8 *gamma;    // packed gamma fragments [it][jt][s][lane]
  const float *beta;      // [32*MT]
  _Float16 *y_blk;        // blk16 output (or null)
  float *y_nchw;          // NCHW fp32 output (or null)
  int B, Cin16, H, W;     // input geometry
  int Ho, Wo, Cout;       // output geometry; Cout = real channel count (<= 32*MT)
  int tiles_x, tiles_y;
  int clamp01;
  int accum;              // NCHW fp32 output: y += out_scale * result (then ReLU / clamp), for the split-operand fp32 passes
  float out_scale;        // 2^-k of LICOS_EPI_SCALE_DOWN(k); 1 otherwise
  int s1conv;             // deconv kernel used as a 3x3 stride-1 conv (space-to-depth first stage): one 'phase', no upsampling
  const void *zero16;     // 16 bytes of zeros in global memory (source of out-of-image granules)
  int in_xsplit, out_xsplit;  // LICOS_EPI_IN_XSPLIT / LICOS_EPI_OUT_XSPLIT (mfma_deconv8.hip only)
  int w_mt_total, halves;     // mfma_conv8.hip pair mode: 32-channel tiles in the packed weights, channel groups per tile
  int out_split3;             // LICOS_EPI_OUT_SPLIT3: y_blk holds 3 Cout channels, the split operand of the next fp32 convolution
  const float *sym_medians;   // licos_conv5x5s2_f16_symbols: y_nchw receives int32 rint(result - median[channel]) instead of the fp32 result
};

// ---- epilogue: bias, (I)GDN, store ----------------------------------------------------------------
// One accumulator tile (32 channels x 32 pixels) leaves as: blk16 fp16 | the 3 Cout-channel split operand of the next fp32
// convolution | NCHW fp32 (optionally accumulated into).  `scale`: the (I)GDN factor per register, or null.
template <int EPI>
__device__ __forceinline__ void epilogue_store_tile(const f32x16 &acc, const f32x16 *scale, const MfmaArgs &a, int b, int oy, int ox,
                                                    int lane, int it, int c_base) {
  const int h = lane >> 5;
  const int Cout16 = (a.Cout + 15) >> 4;
  const bool live = oy < a.Ho && ox < a.Wo && oy >= 0;
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
  if (a.y_blk && a.out_split3) {
    // chunks [hi 2^-5 | (v - hi) 2^6 | hi] of 3 Cout channels (licos_nchw_f32_split3_blk16); lane / permlane pattern as the
    // blk16 store below
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      unsigned lo3[3][2], hi3[3][2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        half2v ph[2], pm[2], pl[2];
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float v = acc[8 * gp + 4 * side + 2 * d + e];
            if (epi_norm(EPI)) v *= (*scale)[8 * gp + 4 * side + 2 * d + e];
            if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
            v = pin_f32(v);
            const _Float16 hv = (_Float16)v;
            ph[side][e] = hv;
            pl[side][e] = (_Float16)((v - (float)hv) * 64.f);
          }
        const half2v k5 = {(_Float16)0.03125f, (_Float16)0.03125f};
        pm[0] = ph[0] * k5;
        pm[1] = ph[1] * k5;
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          const half2v *src = part == 0 ? pm : part == 1 ? pl : ph;
          const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, src[0]), __builtin_bit_cast(unsigned, src[1]), false, false);
          lo3[part][d] = sw[0];
          hi3[part][d] = sw[1];
        }
      }
      const int chunk = (c_base >> 4) + 2 * it + gp;
      if (live && chunk < Cout16) {
#pragma unroll
        for (int part = 0; part < 3; ++part) {
          _Float16 *dst = a.y_blk + ((((size_t)b * 3 * Cout16 + part * Cout16 + chunk) * a.Ho + oy) * a.Wo + ox) * 16 + 8 * h;
          *reinterpret_cast<uint4 *>(dst) = make_uint4(lo3[part][0], lo3[part][1], hi3[part][0], hi3[part][1]);
        }
      }
    }
  } else if (a.y_blk) {
    // blk16: a 16-channel chunk of a pixel is 32 B.  A lane holds channels {0-3, 8-11} (+4 for the
    // upper half-wave) of the chunk; one v_permlane32_swap per dword hands the lower lane channels
    // 0-7 and the upper lane 8-15, so each lane stores ONE 16-byte piece (half as many stores).
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      unsigned lo[2], hi[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        float v0[2], v1[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          v0[e] = acc[8 * gp + 2 * d + e];
          v1[e] = acc[8 * gp + 4 + 2 * d + e];
          if (epi_norm(EPI)) {
            v0[e] *= (*scale)[8 * gp + 2 * d + e];
            v1[e] *= (*scale)[8 * gp + 4 + 2 * d + e];
          }
          if (EPI == EPI_RELU) {
            v0[e] = fmaxf(v0[e], 0.f);
            v1[e] = fmaxf(v1[e], 0.f);
          }
        }
        half2v p0 = {(_Float16)v0[0], (_Float16)v0[1]}, p1 = {(_Float16)v1[0], (_Float16)v1[1]};
        lo[d] = __builtin_bit_cast(unsigned, p0);
        hi[d] = __builtin_bit_cast(unsigned, p1);
        const auto sw = __builtin_amdgcn_permlane32_swap(lo[d], hi[d], false, false);
        lo[d] = sw[0];
        hi[d] = sw[1];
      }
      const int chunk = (c_base >> 4) + 2 * it + gp;
      if (live && chunk < Cout16) {
        _Float16 *dst = a.y_blk + ((((size_t)b * Cout16 + chunk) * a.Ho + oy) * a.Wo + ox) * 16 + 8 * h;
        *reinterpret_cast<uint4 *>(dst) = make_uint4(lo[0], lo[1], hi[0], hi[1]);
      }
    }
  } else {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = c_base + 32 * it + 8 * g + 4 * h;  // 4 consecutive channels c0..c0+3
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[4 * g + e];
        if (epi_norm(EPI)) v *= (*scale)[4 * g + e];
        if (live && c0 + e < a.Cout) {
          float *dst = a.y_nchw + (((size_t)b * a.Cout + c0 + e) * a.Ho + oy) * a.Wo + ox;
          if (a.sym_medians) {  // the entropy bottleneck's symbol, [stream][position]: round-half-to-even as torch.round
            *reinterpret_cast<int32_t *>(dst) = (int32_t)rintf(v - a.sym_medians[c0 + e]);
            continue;
          }
          if (a.accum) v = fmaf(v, a.out_scale, *dst);
          if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
          if (a.clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
          *dst = v;
        }
      }
    }
  }
}

template <int MT, int NT, int EPI>
__device__ inline void epilogue_store(f32x16 (&acc)[MT][NT], const MfmaArgs &a, const bf16x8 *gamma, int b,
                                      const int (&oy)[NT], const int (&ox)[NT], int lane, int c_base = 0) {
  // b may differ per lane (two images side by side in one pixel tile); c_base: first output channel of this
  // workgroup's channel group (a multiple of 32; epilogues without a norm only)
  const int h = lane >> 5;
  // bias: channel of register q in tile mt is 32mt + (q&3) + 8(q>>2) + 4h
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *reinterpret_cast<const float4 *>(a.bias + c_base + 32 * mt + 8 * g + 4 * h);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[mt][nt][4 * g + 0] += bv.x;
        acc[mt][nt][4 * g + 1] += bv.y;
        acc[mt][nt][4 * g + 2] += bv.z;
        acc[mt][nt][4 * g + 3] += bv.w;
      }
    }
  }
  // (static_for, not `#pragma unroll`: with the fp32-norm body the compiler kept this loop rolled and indexed the
  // accumulators through scratch)
  static_for<MT>([&](auto itc) {
    constexpr int it = decltype(itc)::value;
    if constexpr (EPI == EPI_GDN32 || EPI == EPI_IGDN32) {
      // the norm at fp32 accuracy: (acc / 16)^2 split hi + 2^-11 lo in registers (the accumulator tile is the B operand as
      // it stands), 256 gamma split the same way as A fragments [it][jt][s][hi | lo][lane]; hi.hi in `norm`, the two
      // cross terms in `normx` (mfma_gdn_f32.hip is the stand-alone form of this).  One pixel tile at a time - norm,
      // factor, store - : both at once do not fit the register file next to the accumulators.
      const half8 *g32 = reinterpret_cast<const half8 *>(gamma);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x16 norm, normx;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4 *>(a.beta + 32 * it + 8 * g + 4 * h);
          norm[4 * g + 0] = bv.x;
          norm[4 * g + 1] = bv.y;
          norm[4 * g + 2] = bv.z;
          norm[4 * g + 3] = bv.w;
          normx[4 * g + 0] = normx[4 * g + 1] = normx[4 * g + 2] = normx[4 * g + 3] = 0.f;
        }
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const half8 gh = g32[(((it * MT + jt) * 2 + s) * 2 + 0) * 64 + lane], gl = g32[(((it * MT + jt) * 2 + s) * 2 + 1) * 64 + lane];
            half8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              // (volatile: keeps the compiler from sharing the 128 registers of scaled / squared / split accumulators
              // between the four output tiles - recomputing them is cheaper than the spills)
              float t = acc[jt][nt][8 * s + e];
              asm volatile("" : "+v"(t));
              const float v = t * 0.0625f;
              const float sq = pin_f32(v * v);
              bh[e] = (_Float16)sq;
              bl[e] = (_Float16)((sq - (float)bh[e]) * 2048.f);
            }
            norm = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, bh, norm, 0, 0, 0);
            normx = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, bl, normx, 0, 0, 0);
            normx = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, bh, normx, 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          // norm >= beta_min > 0: v_rsq_f32 (1 ulp) + one Newton step; sqrt = norm * rsqrt
          const float n = norm[q] + normx[q] * (1.f / 2048.f);
          const float r0 = __builtin_amdgcn_rsqf(n);
          if (EPI == EPI_IGDN32) {
            const float s0 = n * r0;
            norm[q] = fmaf(fmaf(-s0, s0, n), 0.5f * r0, s0);
          } else {
            norm[q] = fmaf(0.5f * r0, fmaf(-n * r0, r0, 1.f), r0);
          }
        }
        epilogue_store_tile<EPI>(acc[it][nt], &norm, a, b, oy[nt], ox[nt], lane, it, c_base);
      }
    } else {
      f32x16 scale[NT];
      if (EPI == EPI_GDN || EPI == EPI_IGDN) {
        // norm tile `it` = beta + sum_jt sum_s gamma(it, jt, s) x sq(acc[jt], regs 8s..8s+7)
        f32x16 norm[NT];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4 *>(a.beta + 32 * it + 8 * g + 4 * h);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            norm[nt][4 * g + 0] = bv.x;
            norm[nt][4 * g + 1] = bv.y;
            norm[nt][4 * g + 2] = bv.z;
            norm[nt][4 * g + 3] = bv.w;
          }
        }
#pragma unroll
        for (int jt = 0; jt < MT; ++jt) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const bf16x8 gfrag = gamma[((it * MT + jt) * 2 + s) * 64 + lane];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              bf16x8 sq;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float v = acc[jt][nt][8 * s + e];
                sq[e] = (__bf16)(v * v);
              }
              norm[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfrag, sq, norm[nt], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            scale[nt][q] = (EPI == EPI_GDN) ? __builtin_amdgcn_rsqf(norm[nt][q]) : __builtin_amdgcn_sqrtf(norm[nt][q]);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) epilogue_store_tile<EPI>(acc[it][nt], &scale[nt], a, b, oy[nt], ox[nt], lane, it, c_base);
    }
  });
}


static inline int mt_for(int Cout) {
  if (Cout <= 32) return 1;
  if (Cout <= 128) return 4;
  if (Cout <= 192) return 6;
  if (Cout <= 320) return 10;  // q6-8 latents (M = 320)
  return 0;
}

// defined in mfma_conv.hip / mfma_deconv.hip
int mfma_dispatch_conv(const MfmaArgs &a, int MT, int epi, int width, hipStream_t s);
int mfma_try_conv8(const MfmaArgs &a, int MT, int epi, hipStream_t s);  // mfma_conv8.hip; 1 = not applicable
int mfma_dispatch_deconv(const MfmaArgs &a, int MT, int epi, int width, hipStream_t s);
int mfma_try_deconv8(const MfmaArgs &a, int MT, int epi, hipStream_t s);  // mfma_deconv8.hip; 1 = not applicable
int mfma_try_conv3x3_tiles(const MfmaArgs &a, int MT, int epi, hipStream_t s);  // mfma_conv3x3t.hip; 1 = not applicable
bool mfma_deconv8_applies(int MT, int Cin16, int H, int W, bool blk_out, bool accum, bool s1conv);
// 9 .. 16 output channels over a whole number of channel PAIRS of chunks: the 16 x 16 x 32 kernel and its weight layout
// ([pair][tap][lane][8]); everything else: the 32-row kernel and its compact layout.  One rule for packer and launcher.
inline bool fewch_uses_16x16x32(int Cin, int Cout) { return Cout > 8 && Cout <= 16 && Cin % 32 == 0; }
int mfma_launch_deconv_fewch(const MfmaArgs &a, hipStream_t s);
// fp32 GDN / IGDN over 128 channels on the matrix cores (mfma_gdn_f32.hip); HW must be a multiple of 32 and rows 16-byte
// aligned.  Output: NCHW fp32 `y`, or (y_split3 != null) the 3 C-channel split operand of licos_nchw_f32_split3_blk16.
// norm_out (NCHW fp32 output only, nullable): beta + gamma . x^2, what the backward pass starts from.
int mfma_launch_gdn_f32(const float *x, const float *gamma_eff, const float *beta_eff, float *y, void *y_split3, float *norm_out,
                        int B, long HW, int inverse, hipStream_t s);
// GDN / IGDN backward for 128 channels in one pass (mfma_gdn_bwd_f32.hip): dx and t = dL/dnorm from x, dy and the forward's norm
// t_absmax (nullable): receives the bit pattern of max|t|
int mfma_launch_gdn_bwd_f32(const float *x, const float *dy, const float *norm, const float *gamma_eff, float *dx, float *t_out,
                            unsigned int *t_absmax, int B, long HW, int inverse, hipStream_t s);

}  // namespace licos
