// Dev probe (not product): what K-loop STRUCTURE sustains the most matrix-core FLOP/s on random fp16 operands on a
// power-limited MI355X?  Every variant does the same work per workgroup and step - a 128-channel x 512-pixel output
// tile against 6 taps x 16 channels (or 3 taps x 32 channels) of operands read from LDS with the access pattern of
// mfma_deconv8.hip - and differs only in MFMA shape, waves per SIMD and register tile:
//   A  32x32x16, 8 waves (2 per SIMD), 4 x 2 tiles per wave: 0.75 ds_read_b128 per MFMA      (the product's K loop)
//   B  32x32x16, 4 waves (1 per SIMD), 4 x 4 tiles per wave: 0.50 ds_read_b128 per MFMA
//   C  16x16x32, 4 waves (1 per SIMD), 8 x 8 tiles per wave: 0.25 ds_read_b128 per (16-cycle) MFMA
//   D  16x16x32, 8 waves (2 per SIMD), 8 x 4 tiles per wave: 0.375 per MFMA                    (tools/experiments/mfma_deconv8k)
// Optional LDS-DMA of the next step's operands (the product's traffic: 21 patch pieces + 4 weight pieces per tap and 16
// channels).  Interleaved rounds in one process, random operands, in-kernel clock from s_memtime / s_memrealtime.
//   hipcc -O3 --offload-arch=gfx950 -o build/kloop_probe tools/experiments/kloop_probe.hip && build/kloop_probe [steps] [rounds]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(m0v));
}

constexpr int RS = 36, PH = 18;
constexpr int HALF = 656;  // 18 * 36 = 648 granules, padded to a multiple of 16 (16x16x32 B reads: halves 0 banks apart)

// K16: steps of 6 taps x 16 channels; image per buffer: patch [2 halves][HALF] + weights [6 taps][4 frags][64]
// K32: steps of 3 taps x 32 channels; patch [2 chunks][2 halves][HALF] + weights [3 taps][8 frags][64]
template <bool K32>
struct Img {
  static constexpr int NTAP = K32 ? 3 : 6;
  static constexpr int PATCH = (K32 ? 4 : 2) * HALF;
  static constexpr int PATCH_PIECES = (PATCH + 63) / 64;
  static constexpr int W = NTAP * (K32 ? 8 : 4) * 64;
  static constexpr int BUF = PATCH_PIECES * 64 + W;
  static constexpr int LDS = 2 * BUF * 16;
};

template <bool K32, int WAVES, int DMA>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void kloop(const half8 *__restrict__ src, long src_gran, float *out, unsigned long long *stamps,
                                                              int nsteps) {
  using I = Img<K32>;
  constexpr int NTAP = I::NTAP;
  constexpr int NROW = 16 / WAVES;  // input rows per wave: 2 (8 waves) or 4 (4 waves)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s = reinterpret_cast<half8 *>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // fill both buffers with random operands
  for (int i = tid; i < 2 * I::BUF; i += WAVES * 64) s[i] = src[(i + (long)blockIdx.x * 977) % src_gran];
  __syncthreads();
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && tid == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  // DMA sources: weights from a small (L2-resident) region, patch pieces streamed from the big buffer
  const half8 *wsrc = src + lane;
  long pstream = ((long)blockIdx.x * 7919 * 64) % (src_gran - 64 * 64);

  if constexpr (!K32) {
    constexpr int MT = 4, NT = NROW;  // 32-pixel tiles: one per row
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m][n][q] = 0.f;
    const int h = lane >> 5, r = lane & 31;
    int base[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) base[n] = h * HALF + (wave * NROW + n + 1) * RS + r + 1;
    int cur = 0;
    for (int st = 0; st < nsteps; ++st) {
      half8 *buf = s + cur * I::BUF;
      half8 *nb = s + (cur ^ 1) * I::BUF;
      constexpr int PPW = (I::PATCH_PIECES + WAVES - 1) / WAVES, WPW = NTAP * 4 / WAVES;  // pieces per wave: patch, weights
      auto piece = [&](int k) {  // the wave's k-th piece of the next step's operands
        if (k < PPW) {
          const int q = wave + WAVES * k;
          if (q < I::PATCH_PIECES) glds16(src + pstream + q * 64 + lane, nb + q * 64);
        } else if (k < PPW + WPW) {
          const int q = wave + WAVES * (k - PPW);
          glds16(wsrc + ((st * 24 + q) & 1023) * 64, nb + I::PATCH_PIECES * 64 + q * 64);
        }
      };
      if (DMA == 1) {
#pragma unroll
        for (int k = 0; k < PPW + WPW; ++k) piece(k);
      }
      // DMA == 9 (round 5): the UPPER BOUND of phase-shared patch staging - the patch of a cin chunk requested once for all
      // four output phases (every 4th step), the weights every step.  Not realisable in the product as it stands: the four
      // phases' accumulators of a 16 x 32 tile are 512 registers per lane (or the 8 chunks' patches 166 KB of LDS).
      if (DMA == 9) {
#pragma unroll
        for (int k = 0; k < PPW + WPW; ++k)
          if (k >= PPW || (st & 3) == 0) piece(k);
      }
      // DMA >= 2: one piece every DMA / 2 items (an item = one A fragment's MFMAs), waves of the upper half (the SIMD
      // partners) one slot later (DMA odd: same slot)
      const int off = (DMA >= 2 && DMA != 9 && !(DMA & 1) && wave >= WAVES / 2) ? 1 : 0;
      const half8 *sp = buf, *sw = buf + I::PATCH_PIECES * 64;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int toff = (1 - t / 3) * RS + (1 - t % 3);
        half8 b[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = sp[base[n] + toff];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (DMA >= 2 && DMA != 9) {
            constexpr int SP = (DMA / 2) < 1 ? 1 : DMA / 2;
            const int item = t * MT + m - off;
            if (item >= 0 && item % SP == 0 && item / SP < PPW + WPW) piece(item / SP);
          }
          const half8 a = sw[(t * MT + m) * 64 + lane];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            // one wave per SIMD: accumulators pinned to the AGPR half of the file (left to itself the compiler shuffles them
            // between the halves: 2.4 v_accvgpr moves per MFMA)
            if constexpr (WAVES == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m][n]) : "v"(a), "v"(b[n]));
            else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[n], acc[m][n], 0, 0, 0);
          }
        }
      }
      if (DMA) {
        pstream += I::PATCH_PIECES * 64;
        if (pstream > src_gran - 64 * 64) pstream -= src_gran - 64 * 64;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
    }
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += acc[m][n][q];
    out[(long)blockIdx.x * WAVES * 64 + tid] = sum;
  } else {
    constexpr int MT = 8, NT = NROW * 2;  // 16-pixel tiles: two per row
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[m][n][q] = 0.f;
    const int g = lane >> 4, n16 = lane & 15;
    int base[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) base[n] = (g >> 1) * 2 * HALF + (g & 1) * HALF + (wave * NROW + (n >> 1) + 1) * RS + n16 + 16 * (n & 1) + 1;
    int cur = 0;
    for (int st = 0; st < nsteps; ++st) {
      half8 *buf = s + cur * I::BUF;
      if (DMA) {
        half8 *nb = s + (cur ^ 1) * I::BUF;
        for (int q = wave; q < I::PATCH_PIECES; q += WAVES) glds16(src + pstream + q * 64 + lane, nb + q * 64);
        for (int q = wave; q < NTAP * 8; q += WAVES) glds16(wsrc + ((st * 24 + q) & 1023) * 64, nb + I::PATCH_PIECES * 64 + q * 64);
        pstream += I::PATCH_PIECES * 64;
        if (pstream > src_gran - 64 * 64) pstream -= src_gran - 64 * 64;
      }
      const half8 *sp = buf, *sw = buf + I::PATCH_PIECES * 64;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const int toff = (1 - t / 3) * RS + (1 - t % 3);
        half8 b[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) b[n] = sp[base[n] + toff];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const half8 a = sw[(t * MT + m) * 64 + lane];
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (WAVES == 4) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[m][n]) : "v"(a), "v"(b[n]));
            else acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[n], acc[m][n], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur ^= 1;
    }
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += acc[m][n][q];
    out[(long)blockIdx.x * WAVES * 64 + tid] = sum;
  }
  if (blockIdx.x == 0 && tid == 0) {
    stamps[0] = __builtin_amdgcn_s_memtime() - t0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

struct Variant {
  const char *name;
  void (*launch)(const half8 *, long, float *, unsigned long long *, int, int, hipStream_t);
};

template <bool K32, int WAVES, int DMA>
static void launch(const half8 *src, long gran, float *out, unsigned long long *stamps, int nsteps, int grid, hipStream_t st) {
  auto k = kloop<K32, WAVES, DMA>;
  static bool once = false;
  if (!once) {
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, Img<K32>::LDS));
    once = true;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), Img<K32>::LDS, st, src, gran, out, stamps, nsteps);
}

int main(int argc, char **argv) {
  const int nsteps = argc > 1 ? atoi(argv[1]) : 1000;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const int zero = argc > 3 ? atoi(argv[3]) : 0;
  const int grid = 1024;
  const long gran = 4L << 20;  // 64 MB of operands
  std::vector<_Float16> h(gran * 8);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (size_t i = 0; i < h.size(); ++i) h[i] = zero ? (_Float16)0.f : (_Float16)(nd(rng) * ((i >> 3) & 1 ? 0.05f : 1.f));
  half8 *src;
  float *out;
  unsigned long long *stamps;
  CK(hipMalloc(&src, gran * 16));
  CK(hipMalloc(&out, (size_t)grid * 512 * 4));
  CK(hipMalloc(&stamps, 64));
  CK(hipMemcpy(src, h.data(), gran * 16, hipMemcpyHostToDevice));
  const Variant vs[] = {
      {"A 32x32x16 2w/SIMD 4x2                ", launch<false, 8, 0>}, {"D 16x16x32 2w/SIMD 8x4                ", launch<true, 8, 0>},
      {"A +DMA all at step start              ", launch<false, 8, 1>}, {"D +DMA all at step start              ", launch<true, 8, 1>},
      {"A +DMA 1 piece / item, partners +1    ", launch<false, 8, 2>}, {"A +DMA 1 piece / item, same slot      ", launch<false, 8, 3>},
      {"A +DMA 1 piece / 2 items, partners +1 ", launch<false, 8, 4>}, {"A +DMA 1 piece / 2 items, same slot   ", launch<false, 8, 5>},
      {"A +DMA 1 piece / 3 items, partners +1 ", launch<false, 8, 6>}, {"A +DMA 1 piece / 4 items, partners +1 ", launch<false, 8, 8>},
      {"A +DMA weights / step, patch / 4 steps", launch<false, 8, 9>},
  };
  const int NV = sizeof(vs) / sizeof(vs[0]);
  std::vector<std::vector<float>> ms(NV);
  std::vector<double> clk(NV, 0.0);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // warm the chip: ~1 s of launches before anything is timed
  for (int i = 0; i < 40; ++i) vs[0].launch(src, gran, out, stamps, nsteps, grid, 0);
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < NV; ++v) {
      vs[v].launch(src, gran, out, stamps, nsteps, grid, 0);  // untimed: the variant's own steady state
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 3; ++i) vs[v].launch(src, gran, out, stamps, nsteps, grid, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      ms[v].push_back(t / 3);
      unsigned long long st[2];
      CK(hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost));
      clk[v] = (double)st[0] / (double)st[1] * 0.1;  // GHz
    }
  const double flop = 2.0 * 128 * 512 * 96 * (double)nsteps * grid;  // per launch: 128 ch x 512 px x (6 taps x 16 ch) per step
  printf("steps %d grid %d rounds %d %s operands; FLOP per launch %.3e\n", nsteps, grid, rounds, zero ? "ZERO" : "random", flop);
  for (int v = 0; v < NV; ++v) {
    std::sort(ms[v].begin(), ms[v].end());
    const float med = ms[v][ms[v].size() / 2];
    printf("%s median %.3f ms (min %.3f max %.3f)  %.0f TFLOP/s = %.3f of 2.5 PF   clock of block 0 %.2f GHz\n", vs[v].name, med, ms[v].front(),
           ms[v].back(), flop / med / 1e9, flop / med / 1e9 / 2500, clk[v]);
  }
  return 0;
}
