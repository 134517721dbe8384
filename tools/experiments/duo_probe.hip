// Dev probe (not product): would the four-phase transposed-conv kernel gain from running as TWO independent 4-wave
// workgroups per CU instead of one 8-wave workgroup?  In the 8-wave kernel all waves reach a phase's (I)GDN epilogue
// together (one barrier per K step), so the matrix pipes idle for the ~1000 vector instructions per wave of each of the
// four epilogues of a tile; two independent workgroups drift apart and one's epilogue runs beside the other's K loop.
// The price: the weight fragments of a K step serve 256 instead of 512 pixels (1.6x the LDS-DMA pieces per MFMA) and
// gamma no longer fits LDS next to the operands (here it is read from the weight buffers: timing only).
//
// Both variants run the product's K-loop access pattern (6 taps x 16 channels per step, LDS-DMA of the next step's
// operands, one barrier per step) and the product's real epilogue (tile8_epilogue of mfma_deconv8.hpp: squares, norm
// MFMAs, sqrt, scale, pack, 16-byte sc1 stores) after every 8 steps (a tile = 4 epilogues per 200 tap-chunks).
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -Iinclude -Ilicos_amd/csrc -o build/duo_probe tools/experiments/duo_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mfma_deconv8.hpp"

using namespace licos;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

namespace licos {  // (common.hpp declares these; the probe does not link the library)
std::string &last_error_ref() {
  static std::string s;
  return s;
}
int fail(int code, const char *, ...) { return code; }
}  // namespace licos

constexpr int RS = 36, NTAP = 6, MT = 4, NT = 2;

template <int WAVES>
struct Img {
  static constexpr int ROWS = 2 * WAVES;             // input rows of the tile
  static constexpr int HALF = (ROWS + 2) * RS;
  static constexpr int PATCH_PIECES = (2 * HALF + 63) / 64;
  static constexpr int W = NTAP * MT * 64;
  static constexpr int BUF = PATCH_PIECES * 64 + W;  // granules
  // 8 waves: gamma has its own region (as in the product); 4 waves: gamma is read from the operand buffers (timing only)
  static constexpr int GAMMA = MT * MT * 2 * 64;
  static constexpr int LDS = (2 * BUF + 64 + (WAVES == 8 ? GAMMA : 0)) * 16;
};

// DMA: 0 none, 1 all pieces at the step's start, 2 one piece per item (an item = one A fragment's MFMAs)
template <int WAVES, int DMA, bool EPI, bool STAG = false, int PRIO = 0>
__global__ __launch_bounds__(WAVES * 64, 2) void duo(const half8 *__restrict__ src, long src_gran, _Float16 *y, int nimg, unsigned long long *stamps,
                                                     int nsteps) {
  using I = Img<WAVES>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s = reinterpret_cast<half8 *>(smem);
  float *s_beta = reinterpret_cast<float *>(s + 2 * I::BUF);  // 64 granules: bias + beta
  const bf16x8 *s_gamma = reinterpret_cast<const bf16x8 *>(WAVES == 8 ? s + 2 * I::BUF + 64 : s);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < I::LDS / 16; i += WAVES * 64) s[i] = src[(i + (long)blockIdx.x * 977) % src_gran];
  __syncthreads();
  if (tid < 256) s_beta[tid] = 1.0f + 0.001f * tid;
  __syncthreads();
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && tid == 0) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  const half8 *wsrc = src + lane;
  long pstream = ((long)blockIdx.x * 7919 * 64) % (src_gran - 64 * 64);

  f32x16 acc[MT][NT];
  auto acc_init = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bv = *reinterpret_cast<const float4 *>(s_beta + 32 * m + 8 * g + 4 * (lane >> 5));
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[m][n][4 * g + 0] = bv.x - 1.f;
          acc[m][n][4 * g + 1] = bv.y - 1.f;
          acc[m][n][4 * g + 2] = bv.z - 1.f;
          acc[m][n][4 * g + 3] = bv.w - 1.f;
        }
      }
  };
  acc_init();
  const int h = lane >> 5, r = lane & 31;
  int base[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) base[n] = h * I::HALF + (wave * 2 + n + 1) * RS + r + 1;
  int cur = 0, nepi = 0;
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);  // K loop above the epilogue: the matrix wave wins the SIMD's issue arbitration
  for (int st = 0; st < nsteps; ++st) {
    half8 *buf = s + cur * I::BUF;
    half8 *nb = s + (cur ^ 1) * I::BUF;
    constexpr int PPW = (I::PATCH_PIECES + WAVES - 1) / WAVES, WPW = NTAP * MT / WAVES;
    auto piece = [&](int k) {
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
    const half8 *sp = buf, *sw = buf + I::PATCH_PIECES * 64;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const int toff = (1 - t / 3) * RS + (1 - t % 3);
      half8 b[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = sp[base[n] + toff];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (DMA == 2 && t * MT + m < PPW + WPW) piece(t * MT + m);
        const half8 a = sw[(t * MT + m) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[n], acc[m][n], 0, 0, 0);
      }
    }
    if (DMA) {
      pstream += I::PATCH_PIECES * 64;
      if (pstream > src_gran - 64 * 64) pstream -= src_gran - 64 * 64;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur ^= 1;
    // STAG: workgroups reach their epilogues at different steps (identical programs started together otherwise stay in
    // lockstep - and two workgroups of a CU would both be in their epilogues at once)
    const int stag = STAG ? (int)((blockIdx.x * 5u + (blockIdx.x >> 8) * 3u) & 7u) : 0;
    if (EPI && ((st + stag) & 7) == 7) {
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      // one phase's epilogue: the lane's two output pixels of a 128 x 128 x-split map, image chosen per (workgroup, epilogue)
      const int img = (int)((blockIdx.x * 131u + (unsigned)nepi * 17u) % (unsigned)nimg);
      long pix[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) pix[n] = ((long)(((wave * 2 + n) * 2 + (nepi & 1)) * 2 + ((nepi >> 1) & 1)) * 64 + r) % (128 * 128);
      tile8_epilogue<MT, NT, EPI_IGDN>(acc, s_gamma, s_beta + 128, y + (size_t)img * 128 * 128 * 128, (size_t)128 * 128, 8, pix, lane);
      acc_init();
      ++nepi;
      if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += acc[m][n][q];
  if (sum == 12345.678f) y[tid] = (_Float16)sum;  // keeps the accumulators live
  if (blockIdx.x == 0 && tid == 0) {
    stamps[0] = __builtin_amdgcn_s_memtime() - t0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

struct Variant {
  const char *name;
  void (*launch)(const half8 *, long, _Float16 *, int, unsigned long long *, int, int, hipStream_t);
  int waves;
};

template <int WAVES, int DMA, bool EPI, bool STAG = false, int PRIO = 0>
static void launch(const half8 *src, long gran, _Float16 *y, int nimg, unsigned long long *stamps, int nsteps, int grid, hipStream_t st) {
  auto k = duo<WAVES, DMA, EPI, STAG, PRIO>;
  static bool once = false;
  if (!once) {
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, Img<WAVES>::LDS));
    int nb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, WAVES * 64, Img<WAVES>::LDS));
    fprintf(stderr, "waves %d dma %d epi %d: LDS %d B, %d workgroup(s) per CU\n", WAVES, DMA, (int)EPI, Img<WAVES>::LDS, nb);
    once = true;
  }
  // the same work per launch: a 4-wave workgroup covers half the pixels of an 8-wave one
  hipLaunchKernelGGL(k, dim3(grid * (8 / WAVES)), dim3(WAVES * 64), Img<WAVES>::LDS, st, src, gran, y, nimg, stamps, nsteps);
}

int main(int argc, char **argv) {
  const int nsteps = argc > 1 ? atoi(argv[1]) : 1000;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const int zero = argc > 3 ? atoi(argv[3]) : 0;
  const int grid = 1024;
  const long gran = 4L << 20;  // 64 MB of operands
  const int nimg = 256;        // 1 GB of output images (128 ch x 128 x 128 fp16)
  std::vector<_Float16> h(gran * 8);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (size_t i = 0; i < h.size(); ++i) h[i] = zero ? (_Float16)0.f : (_Float16)(nd(rng) * ((i >> 3) & 1 ? 0.05f : 1.f));
  half8 *src;
  _Float16 *y;
  unsigned long long *stamps;
  CK(hipMalloc(&src, gran * 16));
  CK(hipMalloc(&y, (size_t)nimg * 128 * 128 * 128 * 2));
  CK(hipMalloc(&stamps, 64));
  CK(hipMemcpy(src, h.data(), gran * 16, hipMemcpyHostToDevice));
  const Variant vs[] = {
      {"8 waves x 1 WG/CU, K loop only, no DMA   ", launch<8, 0, false>, 8}, {"4 waves x 2 WG/CU, K loop only, no DMA   ", launch<4, 0, false>, 4},
      {"8 waves x 1 WG/CU, DMA at step start     ", launch<8, 1, false>, 8}, {"4 waves x 2 WG/CU, DMA at step start     ", launch<4, 1, false>, 4},
      {"8 waves x 1 WG/CU, DMA 1 per item        ", launch<8, 2, false>, 8}, {"4 waves x 2 WG/CU, DMA 1 per item        ", launch<4, 2, false>, 4},
      {"8 waves x 1 WG/CU, DMA start + EPILOGUE  ", launch<8, 1, true>, 8},  {"4 waves x 2 WG/CU, DMA start + EPILOGUE  ", launch<4, 1, true>, 4},
      {"8 waves x 1 WG/CU, DMA/item + EPILOGUE   ", launch<8, 2, true>, 8},  {"4 waves x 2 WG/CU, DMA/item + EPILOGUE   ", launch<4, 2, true>, 4},
      {"8 waves x 1 WG/CU, no DMA + EPILOGUE     ", launch<8, 0, true>, 8},  {"4 waves x 2 WG/CU, no DMA + EPILOGUE     ", launch<4, 0, true>, 4},
      {"4 waves x 2 WG/CU, no DMA + EPI STAGGERED", launch<4, 0, true, true>, 4}, {"4 waves x 2, DMA start + EPI STAGGERED   ", launch<4, 1, true, true>, 4},
      {"4 waves x 2, DMA/item + EPI STAGGERED    ", launch<4, 2, true, true>, 4}, {"8 waves x 1, DMA/item + EPI STAGGERED    ", launch<8, 2, true, true>, 8},
      {"4 x 2, no DMA + EPI STAG, K loop prio 1  ", launch<4, 0, true, true, 1>, 4}, {"4 x 2, DMA/item + EPI STAG, K loop prio 1", launch<4, 2, true, true, 1>, 4},
      {"4 x 2, no DMA + EPI STAG, K loop prio 3  ", launch<4, 0, true, true, 3>, 4}, {"4 x 2, DMA/item + EPI STAG, K loop prio 3", launch<4, 2, true, true, 3>, 4},
      {"4 x 2, DMA/item + EPI no stag, prio 1    ", launch<4, 2, true, false, 1>, 4}, {"8 x 1, DMA/item + EPI, K loop prio 1     ", launch<8, 2, true, false, 1>, 8},
  };
  const int NV = sizeof(vs) / sizeof(vs[0]);
  std::vector<std::vector<float>> ms(NV);
  std::vector<double> clk(NV, 0.0);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; ++i) vs[0].launch(src, gran, y, nimg, stamps, nsteps, grid, 0);
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (int v = 0; v < NV; ++v) {
      vs[v].launch(src, gran, y, nimg, stamps, nsteps, grid, 0);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 3; ++i) vs[v].launch(src, gran, y, nimg, stamps, nsteps, grid, 0);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t;
      CK(hipEventElapsedTime(&t, e0, e1));
      ms[v].push_back(t / 3);
      unsigned long long st[2];
      CK(hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost));
      clk[v] = (double)st[0] / (double)st[1] * 0.1;
    }
  const double flop = 2.0 * 128 * 512 * 96 * (double)nsteps * grid;  // conv FLOPs per launch (the norm MFMAs are not counted)
  printf("steps %d grid %d rounds %d %s operands; conv FLOP per launch %.3e\n", nsteps, grid, rounds, zero ? "ZERO" : "random", flop);
  for (int v = 0; v < NV; ++v) {
    std::sort(ms[v].begin(), ms[v].end());
    const float med = ms[v][ms[v].size() / 2];
    printf("%s median %.3f ms (min %.3f max %.3f)  %.0f TFLOP/s = %.3f of 2.5 PF conv-only   clock %.2f GHz\n", vs[v].name, med, ms[v].front(),
           ms[v].back(), flop / med / 1e9, flop / med / 1e9 / 2500, clk[v]);
  }
  return 0;
}
