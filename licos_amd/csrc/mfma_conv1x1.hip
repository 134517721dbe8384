// 1x1 convolution on the matrix cores: y[b][co][p] (+)= sum_ci W[co][ci] * x[b][ci][p] - the channel-mixing product of
// GDN / IGDN (norm = beta + gamma . x^2) and of their backward pass (gamma^T . t) on the fp32 path, run as three
// fp16 passes over split operands like the 5x5 layers.
//
// No spatial structure, so no patch staging: a wave owns 64 pixels x all output channels, requests every B fragment
// (16 channels x 32 pixels, 1 KiB contiguous in the blk16 layout) up front and reads the A fragments (weights,
// C x C / 16 KiB in all) from LDS, where the workgroup staged them once.
#include "mfma_common.hpp"

namespace licos {

template <int MT, int CC>
__global__ __launch_bounds__(256, 2) void conv1x1_mfma_kernel(MfmaArgs a, int HW) {
  constexpr int NT = (MT <= 4) ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  half8 *s_w = reinterpret_cast<half8 *>(smem);  // [MT][CC][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, r = lane & 31;
  const int b = blockIdx.y;
  const int p0 = (blockIdx.x * 4 + wave) * (32 * NT);
  for (int g = tid; g < MT * CC * 64; g += 256) s_w[g] = a.wp[g];

  const half8 *xb = reinterpret_cast<const half8 *>(a.x) + (size_t)b * CC * HW * 2;
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  half8 bf[NT][CC];
  int oy[NT], ox[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = p0 + 32 * nt + r;
    const bool ok = p < HW;
    oy[nt] = ok ? p / a.Wo : -1;
    ox[nt] = ok ? p % a.Wo : 0;
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) bf[nt][cc] = ok ? xb[((size_t)cc * HW + p) * 2 + h] : zero8;
  }
  __syncthreads();
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;
#pragma unroll
  for (int cc = 0; cc < CC; ++cc)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const half8 af = s_w[(mt * CC + cc) * 64 + lane];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt][cc], acc[mt][nt], 0, 0, 0);
    }
  epilogue_store<MT, NT, EPI_NONE>(acc, a, nullptr, b, oy, ox, lane);
}

// W [Cout][Cin] fp32 (row-major) -> A fragments [mt][cc][lane][8]: row = 32 mt + lane % 32, k = 16 cc + 8 (lane / 32) + e
__global__ void pack_conv1x1_w_kernel(const float *__restrict__ w, int Cout, int Cin, int CC, _Float16 *__restrict__ out, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long f = i >> 9;
    const int cc = (int)(f % CC), mt = (int)(f / CC);
    const int co = 32 * mt + (lane & 31), ci = 16 * cc + 8 * (lane >> 5) + e;
    out[i] = (_Float16)((co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] : 0.f);
  }
}

template <int MT, int CC>
static int launch_conv1x1(const MfmaArgs &a, int HW, hipStream_t s) {
  constexpr int NT = (MT <= 4) ? 2 : 1;
  const size_t lds = (size_t)MT * CC * 64 * 16;
  auto kern = conv1x1_mfma_kernel<MT, CC>;
  LICOS_ENSURE_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(cdiv(HW, 4 * 32 * NT), a.B), dim3(256), lds, s, a, HW);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

}  // namespace licos

using namespace licos;

extern "C" {

size_t licos_packed_conv1x1_w_bytes(int Cin, int Cout) {
  const int MT = mt_for(Cout);
  if (Cin <= 0 || MT <= 0) return 0;
  return (size_t)MT * ((Cin + 15) / 16) * 64 * 16;
}

int licos_pack_conv1x1_w_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  const int MT = mt_for(Cout);
  LICOS_REQUIRE(w && packed && Cin > 0 && MT > 0, "pack_conv1x1_w_f16: unsupported Cin=%d Cout=%d", Cin, Cout);
  const int CC = (Cin + 15) / 16;
  const long total = (long)MT * CC * 512;
  hipLaunchKernelGGL(pack_conv1x1_w_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, Cout, Cin, CC,
                     static_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_conv1x1_f16(const void *x_blk16, const void *w_packed, const float *bias, int epilogue, float *y_nchw, int B, int Cin,
                      int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(x_blk16 && w_packed && bias && y_nchw, "conv1x1_f16: NULL buffer");
  LICOS_REQUIRE(B > 0 && B <= 65535 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv1x1_f16: bad shape");
  const int flags = epilogue & ~0xff;
  LICOS_REQUIRE((epilogue & 0xff) == LICOS_EPI_NONE, "conv1x1_f16: epilogue must be LICOS_EPI_NONE (+ accumulate flags)");
  const int down = (flags >> 12) & 63;
  MfmaArgs a{};
  a.x = static_cast<const _Float16 *>(x_blk16);
  a.wp = static_cast<const half8 *>(w_packed);
  a.bias = bias;
  a.y_nchw = y_nchw;
  a.B = B;
  a.Cin16 = (Cin + 15) / 16;
  a.H = H;
  a.W = W;
  a.Ho = H;
  a.Wo = W;
  a.Cout = Cout;
  a.accum = (flags & LICOS_EPI_ACCUMULATE) ? 1 : 0;
  a.out_scale = ldexpf(1.f, -down);
  const int MT = mt_for(Cout), HW = H * W;
  hipStream_t s = as_stream(stream);
  if (MT == 4 && a.Cin16 == 8) return launch_conv1x1<4, 8>(a, HW, s);
  if (MT == 6 && a.Cin16 == 12) return launch_conv1x1<6, 12>(a, HW, s);
  return fail(LICOS_EINVAL, "conv1x1_f16: instantiated for 128 -> 128 and 192 -> 192 channels, got %d -> %d", Cin, Cout);
}

}  // extern "C"
