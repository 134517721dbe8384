// MFMA stride-2 5x5 transposed convolution, 8-wave variant for wide maps (input >= 16 x 32, 128 output channels).
//
// Same phase decomposition as mfma_deconv.hip (each of the 4 output phases is a stride-1 conv with 3x3 / 3x2 / 2x3 /
// 2x2 taps; one workgroup = one phase of one input tile; the workgroup exits right after its stores), but:
//   * the tile is 16 x 32 input pixels and the workgroup has 8 waves (wave w owns input rows 2w, 2w+1; NT = 2), so
//     the weight fragments of a step are shared by 8 waves and the patch halo shrinks: 5.6 instead of 9.3 LDS-DMA
//     pieces per wave per 50 MFMAs;
//   * a K step is a whole cin chunk (ALL taps of the phase: 32-72 MFMAs per wave between two barriers instead of
//     16-24), written as one static instruction sequence per phase so the LDS reads run one fragment ahead of their
//     MFMAs across tap and kernel-row boundaries;
//   * gamma has its own LDS region and arrives by LDS-DMA during the first step (no register round trip and no extra
//     barrier in front of the epilogue); the bias is the accumulators' initial value;
//   * the (I)GDN epilogue squares and converts each accumulator ONCE per pixel tile (the 4-wave kernel's shared
//     epilogue redoes it for every 32-channel output tile).
// 146 KB of LDS, one workgroup (2 waves per SIMD) per CU.
#include <cstdlib>

#include "mfma_common.hpp"

// LICOS_ABL (dev builds via tools/ab_build.sh, never the product): timing ablations of this kernel.
//   1 no LDS-DMA inside the K loop   2 no (I)GDN arithmetic in the epilogue   3 no stores   4 no MFMAs in the K loop
#ifndef LICOS_ABL
#define LICOS_ABL 0
#endif
#ifndef LICOS_STAGGER
#define LICOS_STAGGER 0
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
template <int MT, int NT, int NKY, int NKX, int RS, class Mid>
__device__ __forceinline__ void deconv8_chunk(f32x16 (&acc)[MT][NT], const half8 *s_patch, const half8 *s_w,
                                              const int (&base)[NT], int lane, Mid &&mid) {
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
    if (it == NI / 2) mid();  // half-way hook (the younger half of the waves requests its operands here)
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

template <int MT, int EPI>
__global__ __launch_bounds__(512, 2) void deconv5x5s2_mfma8_kernel(MfmaArgs a) {
  using G = Deconv8Geom<MT>;
  constexpr int NT = G::NT;
  constexpr bool NORM = (EPI == EPI_GDN || EPI == EPI_IGDN) && LICOS_ABL != 2;
  constexpr int NSTORE = NT * MT * 2;  // 16-byte store instructions of one epilogue when every row of the wave is live
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_pbuf = reinterpret_cast<half8 *>(smem);   // [2][PATCH_PAD]
  half8 *s_wbuf = s_pbuf + 2 * G::PATCH_PAD;         // [2][W_GRAN_MAX]
  float *s_bias = reinterpret_cast<float *>(s_wbuf + 2 * G::W_GRAN_MAX);  // [32 MT] bias, then [32 MT] beta
  float *s_beta = s_bias + 32 * MT;
  bf16x8 *s_gamma = reinterpret_cast<bf16x8 *>(s_beta + 32 * MT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  int b, tile;
  xcd_work_item(blockIdx.x, a.B, a.tiles_x * a.tiles_y, b, tile);
  const int nphase = a.s1conv ? 1 : 4;
  const int ty0 = (tile / a.tiles_x) * G::TH, tx0 = (tile % a.tiles_x) * G::TW;

  const size_t plane = (size_t)a.H * a.W;
  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)b * a.Cin16 * plane * 2;
  const half8 *zero = reinterpret_cast<const half8 *>(a.zero16);

  // per-lane source offset of this wave's patch pieces inside a chunk plane (-1: outside the image / padding); the
  // same for every phase and chunk
  int p_off[G::NPP];
#pragma unroll
  for (int i = 0; i < G::NPP; ++i) {
    const int d = (wave + 8 * i) * 64 + lane;
    const int hh = d / G::HALF, rem = d - hh * G::HALF;
    const int j = rem / G::RS, q = rem - j * G::RS;
    const int iy = ty0 - 1 + j, ix = tx0 - 1 + q;
    const bool ok = d < G::PATCH_GRAN && q < G::TW + 2 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    const int pix = a.in_xsplit ? (iy * 2 + (ix & 1)) * (a.W >> 1) + (ix >> 1) : iy * a.W + ix;
    p_off[i] = ok ? pix * 2 + hh : -1;
  }
  auto dma_patch = [&](int cc, int buf) {
    const half8 *xin = xb + (size_t)cc * plane * 2;
#pragma unroll
    for (int i = 0; i < G::NPP; ++i) {
      const int q = wave + 8 * i;
      if (q < G::PQ) glds16(p_off[i] >= 0 ? xin + p_off[i] : zero, s_pbuf + buf * G::PATCH_PAD + q * 64);
    }
  };
  auto dma_w = [&](int phase, int cc, int buf) {  // all taps of (phase, cin chunk): ntap * MT pieces
    const int ntap = ((phase >> 1) ? 2 : 3) * ((phase & 1) ? 2 : 3);
    const int tap0 = (phase == 0) ? 0 : (phase == 1) ? 9 : (phase == 2) ? 15 : 21;
    const half8 *wsrc = a.wp + ((size_t)tap0 * a.Cin16 + (size_t)cc * ntap) * MT * 64 + lane;
#pragma unroll
    for (int i = 0; i < G::NWP; ++i) {
      const int q = wave + 8 * i;
      if (q < ntap * MT) glds16(wsrc + q * 64, s_wbuf + buf * G::W_GRAN_MAX + q * 64);
    }
  };
  // operands of K step `cc` of `phase`, or of the first step of the next phase when cc runs past the last chunk
  auto dma_step = [&](int phase, int cc, int buf) {
    if (cc >= a.Cin16) {
      cc -= a.Cin16;
      ++phase;
    }
    if (phase < nphase) {
      dma_w(phase, cc, buf);
      dma_patch(cc, buf);
    }
  };

  dma_step(0, 0, 0);
  // bias and beta live in LDS: a global load after the prologue would be one more vmcnt event whose wait drains the
  // LDS-DMA requests and the stores in front of it (see the counted wait below)
  static_assert(MT == 4, "bias + beta = one 64-lane piece");
  if (wave == 0) glds16((lane < 32 || !NORM) ? a.bias + 4 * (lane & 31) : a.beta + 4 * (lane & 31), s_bias);
  if (NORM) {
#pragma unroll
    for (int i = 0; i < G::NGP; ++i) {
      const int q = wave + 8 * i;
      if (q < G::GAMMA_GRAN / 64) glds16(a.gamma + q * 64 + lane, s_gamma + q * 64);
    }
  }

  int base[NT], oyh[NT], oxh[NT];  // output position of the input pixel at phase (0, 0); row -1 = outside the map
  bool all_live = a.Cout >= 32 * MT - 15;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int ty = wave * NT + nt, tx = r;
    const bool in = (ty0 + ty) < a.H && (tx0 + tx) < a.W;
    oyh[nt] = in ? ty0 + ty : -1;
    oxh[nt] = tx0 + tx;
    all_live = all_live && (ty0 + ty) < a.H;  // wave-uniform: a live row issues its stores whatever its columns
    base[nt] = h * G::HALF + (ty + 1) * G::RS + (tx + 1);  // patch row 0 / column 0 = input row ty0-1 / column tx0-1
  }
  // accumulators start at the bias: channel of register q in tile mt is 32mt + (q&3) + 8(q>>2) + 4h
  f32x16 acc[MT][NT];
  auto acc_init = [&]() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(s_bias + 32 * mt + 8 * g + 4 * h);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[mt][nt][4 * g + 0] = bv.x;
          acc[mt][nt][4 * g + 1] = bv.y;
          acc[mt][nt][4 * g + 2] = bv.z;
          acc[mt][nt][4 * g + 3] = bv.w;
        }
      }
  };

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  acc_init();

  // The workgroup walks the phases of its tile one after the other: ONE stream of K steps (phase, cin chunk), each
  // step's operands requested one step ahead into the other buffer, with a phase's epilogue between its last step and
  // the next phase's first.  Stores and LDS-DMA share vmcnt (in issue order), so around an epilogue the order is:
  //   barrier of the last step | request the operands of the next phase's SECOND step | epilogue arithmetic, stores |
  //   MFMAs of the next phase's first step | s_waitcnt vmcnt(NSTORE) | barrier
  // - the counted wait lets the NSTORE youngest operations (the stores) stay in flight and still guarantees the
  // requests issued before them have landed; the stores are waited for one whole step later, by the vmcnt(0) of the
  // second step.  Waves that issue fewer stores (rows outside the map, fewer channels) wait for vmcnt(0).
  int cur = 0;
  bool requested = false;  // the operands of the coming step's successor are already on their way
  auto kloop = [&](auto nky_c, auto nkx_c, int phase) {
    constexpr int NKY = decltype(nky_c)::value, NKX = decltype(nkx_c)::value;
    for (int cc = 0; cc < a.Cin16; ++cc) {
      const bool counted = requested && all_live;
      // waves 0-3 request the next step's operands at the start of the step, waves 4-7 (their SIMD partners) half-way
      // through it: a request costs its wave ~60 issue cycles per piece, which the partner's MFMAs cover only if the
      // two are not doing it at the same moment
      const bool want = !requested && LICOS_ABL != 1;
      const bool late = LICOS_STAGGER && wave >= 4;
      if (want && !late) dma_step(phase, cc + 1, cur ^ 1);
      requested = false;
      if (LICOS_ABL != 4)
        deconv8_chunk<MT, NT, NKY, NKX, G::RS>(acc, s_pbuf + cur * G::PATCH_PAD, s_wbuf + cur * G::W_GRAN_MAX, base, lane,
                                               [&]() { if (want && late) dma_step(phase, cc + 1, cur ^ 1); });
      else if (want && late) dma_step(phase, cc + 1, cur ^ 1);
      // my DMA pieces have landed; after the barrier so have everyone's, and every wave is done reading the buffers
      // the next step overwrites
      if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
    }
  };
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  const int Cout16 = (a.Cout + 15) >> 4;

  // straight-line code per phase (a loop over phases with a switch makes the register allocator spill)
  auto run_phase = [&](auto nky_c, auto nkx_c, auto phase_c) {
    constexpr int phase = decltype(phase_c)::value;
    kloop(nky_c, nkx_c, phase);
    if (phase + 1 < nphase && a.Cin16 > 1 && LICOS_ABL != 1) {
      dma_step(phase + 1, 1, cur ^ 1);
      requested = true;
    }
    asm volatile("" ::: "memory");  // the stores below stay behind the requests above (see the counted wait)

    // ---- epilogue: (I)GDN per pixel tile, fp16 pack, 16-byte stores ------------------------------------------
    constexpr int py = phase >> 1, px = phase & 1;
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
      const int oy = a.s1conv ? oyh[nt] : 2 * oyh[nt] + py, ox = a.s1conv ? oxh[nt] : 2 * oxh[nt] + px;
      const bool live = oyh[nt] >= 0 && oy < a.Ho && ox < a.Wo;
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
          if (live && chunk < Cout16 && (LICOS_ABL != 3 || lo[0] == 0x12345678u)) {
            // x-split output: row oy as [even-x pixels][odd-x pixels] - this phase's pixels of the row are one run
            const size_t pix = a.out_xsplit ? ((size_t)oy * 2 + px) * a.W + oxh[nt] : (size_t)oy * a.Wo + ox;
            _Float16 *dst = a.y_blk + (((size_t)b * Cout16 + chunk) * a.Ho * a.Wo + pix) * 16 + 8 * h;
            *reinterpret_cast<uint4 *>(dst) = make_uint4(lo[0], lo[1], hi[0], hi[1]);
          }
        }
        // one 32-channel tile at a time: letting the scheduler interleave the four norm chains costs 48 more live
        // registers than the kernel has, and a spill reload is a vmcnt event (see above)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (phase + 1 < nphase) acc_init();
  };
  run_phase(I3{}, I3{}, std::integral_constant<int, 0>{});
  if (nphase > 1) {
    run_phase(I3{}, I2{}, std::integral_constant<int, 1>{});
    run_phase(I2{}, I3{}, std::integral_constant<int, 2>{});
    run_phase(I2{}, I2{}, std::integral_constant<int, 3>{});
  }
}

template <int MT, int EPI>
static int launch_deconv8(const MfmaArgs &a0, hipStream_t s) {
  using G = Deconv8Geom<MT>;
  MfmaArgs a = a0;
  a.tiles_x = cdiv(a.W, G::TW);
  a.tiles_y = cdiv(a.H, G::TH);
  const size_t lds = (size_t)16 * (G::KLOOP_GRAN + 16 * MT + ((EPI == EPI_GDN || EPI == EPI_IGDN) ? G::GAMMA_GRAN : 0));
  auto kern = deconv5x5s2_mfma8_kernel<MT, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    LICOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const long blocks = (long)a.tiles_x * a.tiles_y * a.B;  // a workgroup walks all four phases of its tile
  LICOS_REQUIRE(blocks < (1L << 31), "deconv5x5s2_f16: grid too large");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, s, a);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

// returns LICOS_OK after launching, or 1 when this variant does not apply (caller falls back to the 4-wave kernel)
bool mfma_deconv8_applies(int MT, int Cin16, int H, int W, bool blk_out, bool accum, bool s1conv) {
  static const bool enabled = [] { const char *e = getenv("LICOS_DECONV8"); return !(e && e[0] == '0'); }();
  return enabled && MT == 4 && H >= 16 && W >= 32 && blk_out && !accum && (Cin16 >= 2 || s1conv);
}

int mfma_try_deconv8(const MfmaArgs &a, int MT, int epi, hipStream_t s) {
  if (!mfma_deconv8_applies(MT, a.Cin16, a.H, a.W, a.y_blk != nullptr, a.accum != 0, a.s1conv != 0)) return 1;
  if (epi == EPI_IGDN) return launch_deconv8<4, EPI_IGDN>(a, s);
  if (epi == EPI_GDN) return launch_deconv8<4, EPI_GDN>(a, s);
  if (epi == EPI_NONE) return launch_deconv8<4, EPI_NONE>(a, s);
  if (epi == EPI_RELU) return launch_deconv8<4, EPI_RELU>(a, s);
  return 1;
}

}  // namespace licos
