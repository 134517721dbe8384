// Shared by the 8-wave tile kernels (mfma_deconv8.hip: transposed conv, four phases per workgroup; mfma_conv3x3t.hip:
// 3x3 stride-1 conv with resident weights, several tiles per workgroup): tile geometry, the K step of one cin chunk and
// the (I)GDN / pack / store epilogue.
#pragma once
#include <type_traits>
#include "mfma_common.hpp"

// LICOS_ABL (dev builds via tools/ab_build.sh, never the product): timing ablations.
//   1 no LDS-DMA inside the K loop   2 no (I)GDN arithmetic in the epilogue   3 no stores   4 no MFMAs in the K loop
#ifndef LICOS_ABL
#define LICOS_ABL 0
#endif
// Output stores of the tile kernels carry sc1: written through and NOT kept in the XCD's L2 (MI355X_MICROARCH.md, "stores
// of each flavour").  A stage's output is gigabytes that nothing re-reads before it has left a 4 MB L2 anyway, but kept
// there it evicts the input patches the transposed conv re-reads once per phase: with plain stores that kernel fetched
// 4.1x its input from HBM (profiles/r02_pmc_traffic_deconv_s4_plain_stores.json).  LICOS_STORE_SC1=0: A/B builds only.
#ifndef LICOS_STORE_SC1
#define LICOS_STORE_SC1 1
#endif
#ifndef LICOS_STORE_BITS  // (A/B builds: the cache-policy bits of that store, as assembler text)
#define LICOS_STORE_BITS "sc1"
#endif

namespace licos {

template <int MT>
struct Deconv8Geom {
  static constexpr int TH = 16, TW = 32, NT = 2;
  static constexpr int RS = 36;                       // patch row stride in granules (34 used)
  static constexpr int PH = TH + 2;
  static constexpr int HALF = PH * RS;                // 648
  static constexpr int PATCH_GRAN = 2 * HALF;         // 1296
  static constexpr int PQ = (PATCH_GRAN + 63) / 64;   // 21 wave-wide pieces
  static constexpr int PATCH_PAD = PQ * 64;
  static constexpr int W_GRAN_MAX = 9 * MT * 64;      // all taps of the largest phase, one cin chunk
  static constexpr int GAMMA_GRAN = MT * MT * 2 * 64;
  static constexpr int KLOOP_GRAN = 2 * PATCH_PAD + 2 * W_GRAN_MAX;
  static constexpr int NPP = (PQ + 7) / 8, NWP = (9 * MT + 7) / 8, NGP = (GAMMA_GRAN / 64 + 7) / 8;  // pieces per wave
};

// one cin chunk of one phase: NKY x NKX taps x MT A fragments, each against the NT pixel tiles of the wave.  The LDS
// reads run TWO items (one item = one A fragment = NT MFMAs) ahead of their use, pinned by sched_group_barrier: a
// ds_read_b128 takes longer to come back than the NT MFMAs of one item take to issue.
template <int MT, int NT, int NKY, int NKX, int RS>
__device__ __forceinline__ void deconv8_chunk(f32x16 (&acc)[MT][NT], const half8 *s_patch, const half8 *s_w,
                                              const int (&base)[NT], int lane) {
  constexpr int NTAP = NKY * NKX, NI = NTAP * MT;
  static_assert(MT >= 2, "the B fragments of the next tap are requested two items before its first use");
  // tap t = iky * NKX + ikx reads the patch at (dy, dx) = (1 - iky, 1 - ikx) (see mfma_deconv.hip)
  half8 a_cur = s_w[lane], a_nxt = s_w[64 + lane], b_cur[NT], b_nxt[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = b_cur[nt] = s_patch[base[nt] + RS + 1];
  static_for<NI>([&](auto itc) {
    constexpr int it = decltype(itc)::value, mt = it % MT, tap = it / MT;
    constexpr bool more_a = it + 2 < NI, more_b = (mt == MT - 2) && (tap + 1 < NTAP);
    constexpr int iky_n = (tap + 1) / NKX, ikx_n = (tap + 1) % NKX;
    half8 a_nn = a_nxt;
    if (more_a) a_nn = s_w[(it + 2) * 64 + lane];
    if (more_b) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = s_patch[base[nt] + (1 - iky_n) * RS + (1 - ikx_n)];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur, b_cur[nt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, (more_a ? 1 : 0) + (more_b ? NT : 0), 0);
    __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
    a_cur = a_nxt;
    a_nxt = a_nn;
    if (mt == MT - 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_cur[nt] = b_nxt[nt];
    }
  });
}

// ---- epilogue of one accumulator set: (I)GDN per pixel tile (squares converted once), fp16 pack, 16-byte stores ----
// pix[nt]: index of the lane's output pixel inside the image's [Ho * Wo] plane, or -1 (outside the map: nothing stored).
// y_img: the image's first output element; a 16-channel chunk of the image is plane_px * 16 halfs.
// `hook(block)` runs after each of the NT * MT (pixel tile, 32-channel tile) blocks, block = nt * MT + it: kernels whose
// waves take turns between K loop and epilogue (mfma_first16.hip, duo form) join the workgroup's barriers there.
struct EpilogueNoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
// A hook type that derives from this stores through a raw buffer (`out`: the image's output, offsets below 2^31): a lane
// without a pixel asks for an offset past num_records instead of being masked off, so EVERY block issues its two stores
// whatever the exec mask - a wave that keeps loads in flight across the epilogue can then count them (mfma_first16.hip).
struct EpilogueBufferStores {
  __amdgpu_buffer_rsrc_t out;
};
template <int MT, int NT, int EPI, class Hook = EpilogueNoHook>
__device__ __forceinline__ void tile8_epilogue(f32x16 (&acc)[MT][NT], const bf16x8 *s_gamma, const float *s_beta, _Float16 *y_img,
                                               size_t plane_px, int Cout16, const long (&pix)[NT], int lane, Hook hook = Hook()) {
  constexpr bool NORM = (EPI == EPI_GDN || EPI == EPI_IGDN) && LICOS_ABL != 2;
  const int h = lane >> 5;
  const unsigned chunk_bytes = (unsigned)plane_px * 32u;  // one 16-channel chunk of the image
  // the image's base as a scalar pair (it is workgroup-uniform; the readfirstlanes tell the compiler so)
  const uint64_t y_bits = reinterpret_cast<uint64_t>(y_img);
  const _Float16 *y_base = reinterpret_cast<const _Float16 *>(
      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(y_bits >> 32)) << 32) |
      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)y_bits));  // (the builtin returns int: no sign extension)
  unsigned pix_off[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) pix_off[nt] = (unsigned)pix[nt] * 32u + 16u * h;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bf16x8 sq[MT][2];
    if (NORM) {
#pragma unroll
      for (int jt = 0; jt < MT; ++jt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = acc[jt][nt][8 * s + e];
            sq[jt][s][e] = (__bf16)(v * v);
          }
    }
    const bool live = pix[nt] >= 0;
#pragma unroll
    for (int it = 0; it < MT; ++it) {
      f32x16 scale;
      if (NORM) {
        f32x16 norm;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *reinterpret_cast<const float4 *>(s_beta + 32 * it + 8 * g + 4 * h);
          norm[4 * g + 0] = bv.x;
          norm[4 * g + 1] = bv.y;
          norm[4 * g + 2] = bv.z;
          norm[4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int jt = 0; jt < MT; ++jt)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            norm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(s_gamma[((it * MT + jt) * 2 + s) * 64 + lane], sq[jt][s], norm, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q)
          scale[q] = (EPI == EPI_GDN) ? __builtin_amdgcn_rsqf(norm[q]) : __builtin_amdgcn_sqrtf(norm[q]);
      }
      // a lane holds channels {0-3, 8-11} (+4 for the upper half-wave) of each 16-channel chunk; one
      // v_permlane32_swap per dword hands the lower lane channels 0-7 and the upper lane 8-15: one 16-byte store each
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        unsigned lo[2], hi[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          float v0[2], v1[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            v0[e] = acc[it][nt][8 * gp + 2 * d + e];
            v1[e] = acc[it][nt][8 * gp + 4 + 2 * d + e];
            if (NORM) {
              v0[e] *= scale[8 * gp + 2 * d + e];
              v1[e] *= scale[8 * gp + 4 + 2 * d + e];
            }
            if (EPI == EPI_RELU) {
              v0[e] = fmaxf(v0[e], 0.f);
              v1[e] = fmaxf(v1[e], 0.f);
            }
          }
          typedef _Float16 half2v __attribute__((ext_vector_type(2)));
          half2v p0 = {(_Float16)v0[0], (_Float16)v0[1]}, p1 = {(_Float16)v1[0], (_Float16)v1[1]};
          lo[d] = __builtin_bit_cast(unsigned, p0);
          hi[d] = __builtin_bit_cast(unsigned, p1);
          const auto sw = __builtin_amdgcn_permlane32_swap(lo[d], hi[d], false, false);
          lo[d] = sw[0];
          hi[d] = sw[1];
        }
        const int chunk = 2 * it + gp;
        if constexpr (std::is_base_of<EpilogueBufferStores, Hook>::value) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 val = {lo[0], lo[1], hi[0], hi[1]};
          const unsigned off = (live && chunk < Cout16) ? pix_off[nt] + (unsigned)chunk * chunk_bytes : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b128(val, hook.out, off, 0, 2 /* nt */);
        } else if (live && chunk < Cout16 && (LICOS_ABL != 3 || lo[0] == 0x12345678u)) {
          if (LICOS_STORE_SC1) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 val = {lo[0], lo[1], hi[0], hi[1]};
            // scalar base + 32-bit byte offset: one vector add per store instead of a 64-bit address (the epilogue is
            // vector-issue bound; the launchers check that an image's output stays below 4 GB).
            // (the s_nop covers the ISA's manual wait state between a store of more than 64 bits and the next write of
            // its data registers, which the compiler cannot see through the asm)
            // (the base goes through an s_mov inside the statement: when the register allocator has spilled it to a VGPR
            // lane it comes back through v_readlane, and a vector-memory instruction that reads an SGPR within five wait
            // states of a VALU write of it reads garbage - a hazard the compiler pads for only where it can see the
            // instruction; an SALU read is interlocked)
            const unsigned off = pix_off[nt] + (unsigned)chunk * chunk_bytes;
            uint64_t base_copy;
            asm volatile("s_mov_b64 %0, %3\n\tglobal_store_dwordx4 %1, %2, %0 " LICOS_STORE_BITS "\n\ts_nop 1"
                         : "=&s"(base_copy)
                         : "v"(off), "v"(val), "s"(y_base)
                         : "memory");
          } else {
            _Float16 *dst = y_img + ((size_t)chunk * plane_px + (size_t)pix[nt]) * 16 + 8 * h;
            *reinterpret_cast<uint4 *>(dst) = make_uint4(lo[0], lo[1], hi[0], hi[1]);
          }
        }
      }
      // one 32-channel tile at a time: letting the scheduler interleave the four norm chains costs 48 more live
      // registers than the kernel has, and a spill reload is a vmcnt event (see the counted waits of the callers)
#ifndef LICOS_EPI_NO_FENCE  // (A/B builds: let the scheduler overlap two blocks' chains where registers allow)
      __builtin_amdgcn_sched_barrier(0);
#endif
      hook(nt * MT + it);
    }
  }
}

}  // namespace licos
