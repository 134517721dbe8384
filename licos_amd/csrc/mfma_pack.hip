// MFMA path: operand packing, layout conversion and the C entry points.
#include <cstdlib>

#include "mfma_common.hpp"

namespace licos {

// ---- packing kernels -----------------------------------------------------------------------------------
// conv weights [Cout][Cin][5][5] fp32 -> [cc][ky][kx][mt][lane][8] fp16 A-fragments
__global__ void pack_conv_w_kernel(const float *__restrict__ w, int Cin, int Cout, int Cin16, int MT,
                                   _Float16 *__restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long rest = e >> 9;
    const int mt = (int)(rest % MT); rest /= MT;
    const int tap = (int)(rest % 25); rest /= 25;
    const int cc = (int)rest;
    const int co = 32 * mt + (lane & 31), ci = 16 * cc + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 25 + tap];
    out[e] = (_Float16)v;
  }
}

// deconv weights [Cin][Cout][5][5] fp32 -> [phase][cc][tap in phase][mt][lane][8]
__global__ void pack_deconv_w_kernel(const float *__restrict__ w, int Cin, int Cout, int Cin16, int MT,
                                     _Float16 *__restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long rest = e >> 9;
    const int mt = (int)(rest % MT); rest /= MT;
    // rest = phase_tap0 * Cin16 + cc * ntap + t
    int phase = 0, tap0 = 0;
    const int ntaps[4] = {9, 6, 6, 4}, starts[4] = {0, 9, 15, 21};
    for (int p = 3; p >= 0; --p)
      if (rest >= (long)starts[p] * Cin16) { phase = p; tap0 = starts[p]; break; }
    rest -= (long)tap0 * Cin16;
    const int ntap = ntaps[phase];
    const int cc = (int)(rest / ntap), t = (int)(rest % ntap);
    const int py = phase >> 1, px = phase & 1, nkx = px ? 2 : 3;
    const int ky = py + 2 * (t / nkx), kx = px + 2 * (t % nkx);
    const int co = 32 * mt + (lane & 31), ci = 16 * cc + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)ci * Cout + co) * 25 + ky * 5 + kx];
    out[e] = (_Float16)v;
  }
}

// compact deconv weights for the few-channel last stage: [cc][piece][half][tap in piece][row] x 8 halfs, taps in
// phase-major order (the order deconv5x5s2_fewch_kernel walks them)
__global__ void pack_deconv_w_fewch_kernel(const float *__restrict__ w, int Cin, int Cout, int Cin16, int RP,
                                           _Float16 *__restrict__ out, long total) {
  const int TPP = 32 / RP, WP = (25 + TPP - 1) / TPP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7);
    long g = e >> 3;                       // granule
    const int in_piece = (int)(g & 63);
    g >>= 6;
    const int piece = (int)(g % WP), cc = (int)(g / WP);
    const int hh = in_piece >> 5, rem = in_piece & 31;
    const int T = piece * TPP + rem / RP, row = rem % RP;
    float v = 0.f;
    if (T < 25) {
      const int phase = T < 9 ? 0 : T < 15 ? 1 : T < 21 ? 2 : 3;
      const int tap0 = phase == 0 ? 0 : phase == 1 ? 9 : phase == 2 ? 15 : 21;
      const int py = phase >> 1, px = phase & 1, nkx = px ? 2 : 3;
      const int t = T - tap0;
      const int ky = py + 2 * (t / nkx), kx = px + 2 * (t % nkx);
      const int ci = 16 * cc + 8 * hh + j;
      if (row < Cout && ci < Cin) v = w[((size_t)ci * Cout + row) * 25 + ky * 5 + kx];
    }
    out[e] = (_Float16)v;
  }
}

// the same weights as v_mfma_f32_16x16x32_f16 A fragments for deconv5x5s2_few16_kernel: [chunk pair][tap][lane][8], lane
// (row = lane % 16, k-group g = lane / 16) holding channels 32 pair + 16 (g / 2) + 8 (g % 2) + 0..7 of output channel `row`
__global__ void pack_deconv_w_few16_kernel(const float *__restrict__ w, int Cin, int Cout, _Float16 *__restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    const long tp = e >> 9;
    const int T = (int)(tp % 25), pair = (int)(tp / 25);
    const int row = lane & 15, g = lane >> 4;
    const int phase = T < 9 ? 0 : T < 15 ? 1 : T < 21 ? 2 : 3;
    const int tap0 = phase == 0 ? 0 : phase == 1 ? 9 : phase == 2 ? 15 : 21;
    const int py = phase >> 1, px = phase & 1, nkx = px ? 2 : 3;
    const int t = T - tap0;
    const int ky = py + 2 * (t / nkx), kx = px + 2 * (t % nkx);
    const int ci = 32 * pair + 16 * (g >> 1) + 8 * (g & 1) + j;
    float v = 0.f;
    if (row < Cout && ci < Cin) v = w[((size_t)ci * Cout + row) * 25 + ky * 5 + kx];
    out[e] = (_Float16)v;
  }
}

// GDN: gamma_eff = max(gamma, bound)^2 - pedestal as bf16 A-fragments, k-permuted for the
// accumulator-as-B-operand product: element e of lane (r, h) of fragment (it, jt, s) is
// gamma[32it + r][32jt + 16s + 8(e>>2) + 4h + (e&3)].  beta_eff (fp32, padded) follows.
__global__ void pack_gdn_kernel(const float *__restrict__ beta_raw, const float *__restrict__ gamma_raw,
                                float beta_bound, float gamma_bound, float pedestal, int C, int MT,
                                __bf16 *__restrict__ gout, float *__restrict__ bout) {
  const long total = (long)MT * MT * 2 * 64 * 8;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int el = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long rest = e >> 9;
    const int s = (int)(rest & 1); rest >>= 1;
    const int jt = (int)(rest % MT), it = (int)(rest / MT);
    const int i = 32 * it + (lane & 31), j = 32 * jt + 16 * s + 8 * (el >> 2) + 4 * (lane >> 5) + (el & 3);
    float v = 0.f;
    if (i < C && j < C) {
      const float g = fmaxf(gamma_raw[(size_t)i * C + j], gamma_bound);
      v = g * g - pedestal;
    }
    gout[e] = (__bf16)v;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 32 * MT; i += gridDim.x * blockDim.x) {
    float v = 1.f;  // padded channels: norm = 1 keeps rsqrt/sqrt finite
    if (i < C) {
      const float t = fmaxf(beta_raw[i], beta_bound);
      v = t * t - pedestal;
    }
    bout[i] = v;
  }
}

// The same reparametrised gamma for LICOS_EPI_NORM32: 256 gamma split hi + 2^-11 lo in fp16, fragments
// [it][jt][s][hi | lo][lane][8] (k-permuted as above), then beta [32 MT] fp32
__global__ void pack_gdn_f32split_kernel(const float *__restrict__ beta_raw, const float *__restrict__ gamma_raw,
                                         float beta_bound, float gamma_bound, float pedestal, int C, int MT,
                                         _Float16 *__restrict__ gout, float *__restrict__ bout) {
  const long total = (long)MT * MT * 2 * 64 * 8;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int el = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long rest = e >> 9;
    const int s = (int)(rest & 1); rest >>= 1;
    const int jt = (int)(rest % MT), it = (int)(rest / MT);
    const int i = 32 * it + (lane & 31), j = 32 * jt + 16 * s + 8 * (el >> 2) + 4 * (lane >> 5) + (el & 3);
    float v = 0.f;
    if (i < C && j < C) {
      const float g = fmaxf(gamma_raw[(size_t)i * C + j], gamma_bound);
      v = (g * g - pedestal) * 256.f;
    }
    v = pin_f32(v);
    const _Float16 hi = (_Float16)v;
    const size_t frag = (size_t)((it * MT + jt) * 2 + s) * 2;
    gout[((frag + 0) * 64 + lane) * 8 + el] = hi;
    gout[((frag + 1) * 64 + lane) * 8 + el] = (_Float16)((v - (float)hi) * 2048.f);
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 32 * MT; i += gridDim.x * blockDim.x) {
    float v = 1.f;  // padded channels: norm = 1 keeps rsqrt/sqrt finite
    if (i < C) {
      const float t = fmaxf(beta_raw[i], beta_bound);
      v = t * t - pedestal;
    }
    bout[i] = v;
  }
}

// space-to-depth conv weights: [Cout][Cin][5][5] -> phase-0-style fragments [cc][t = iky*3+ikx][mt][lane][8] of the
// equivalent 3x3 stride-1 conv over channels c*4 + py*2 + px (see licos_conv5x5s2_s2d_f16)
__global__ void pack_conv_w_s2d_kernel(const float *__restrict__ w, int Cin, int Cout, int C16, int MT,
                                       _Float16 *__restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long rest = e >> 9;
    const int mt = (int)(rest % MT); rest /= MT;
    const int t = (int)(rest % 9), cc = (int)(rest / 9);
    const int co = 32 * mt + (lane & 31), kidx = 16 * cc + 8 * (lane >> 5) + j;
    const int c = kidx >> 2, py = (kidx >> 1) & 1, px = kidx & 1;
    const int dy = 1 - t / 3, dx = 1 - t % 3;
    const int ky = 2 * dy + 2 + py, kx = 2 * dx + 2 + px;
    float v = 0.f;
    if (co < Cout && c < Cin && ky <= 4 && kx <= 4) v = w[((size_t)co * Cin + c) * 25 + ky * 5 + kx];
    out[e] = (_Float16)v;
  }
}

// NCHW fp32 -> 2x2 space-to-depth blk16: out[b][chunk][y/2][x/2][k], k = c*4 + (y&1)*2 + (x&1)
__global__ void nchw_to_s2d_blk16_kernel(const float *__restrict__ x, _Float16 *__restrict__ y, int C, int C16, int H,
                                         int W, long total) {
  const int H2 = H / 2, W2 = W / 2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int x2 = (int)(e % W2);
    const int y2 = (int)((e / W2) % H2);
    const int cc = (int)((e / ((long)W2 * H2)) % C16);
    const long b = e / ((long)W2 * H2 * C16);
    half8 lo, hi;
    // s2d channel kidx = 4c + 2py + px: the two px of a (c, py) pair are adjacent in memory - one 8-byte load
#pragma unroll
    for (int jp = 0; jp < 8; ++jp) {
      const int kidx = cc * 16 + 2 * jp, c = kidx >> 2, py = (kidx >> 1) & 1;
      float2 v = make_float2(0.f, 0.f);
      if (c < C) v = *reinterpret_cast<const float2 *>(x + (((size_t)b * C + c) * H + 2 * y2 + py) * W + 2 * x2);
      if (jp < 4) {
        lo[2 * jp] = (_Float16)v.x;
        lo[2 * jp + 1] = (_Float16)v.y;
      } else {
        hi[2 * jp - 8] = (_Float16)v.x;
        hi[2 * jp - 7] = (_Float16)v.y;
      }
    }
    half8 *dst = reinterpret_cast<half8 *>(y + (size_t)e * 16);
    dst[0] = lo;
    dst[1] = hi;
  }
}

// 3x3 stride-1 conv weights [Cout][Cin][3][3] -> [cc][t = iky*3+ikx][mt][lane][8]; the kernel reads the input at
// (dy, dx) = (1 - iky, 1 - ikx), i.e. tap (ky, kx) = (2 - iky, 2 - ikx)
__global__ void pack_conv3x3_w_kernel(const float *__restrict__ w, int Cin, int Cout, int C16, int MT,
                                      _Float16 *__restrict__ out, long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
    long rest = e >> 9;
    const int mt = (int)(rest % MT); rest /= MT;
    const int t = (int)(rest % 9), cc = (int)(rest / 9);
    const int co = 32 * mt + (lane & 31), ci = 16 * cc + 8 * (lane >> 5) + j;
    const int ky = 2 - t / 3, kx = 2 - t % 3;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
    out[e] = (_Float16)v;
  }
}

// y = fp16(x); with y_res also the residual fp16((x - float(y)) * 2^shift): x = y + y_res * 2^-shift to ~22 bits, the operand split of the
// "fp32 through three fp16 MFMA passes" path (hi*hi + hi*lo + lo*hi, fp32 accumulation)
__global__ void nchw_to_blk16_kernel(const float *__restrict__ x, _Float16 *__restrict__ y, _Float16 *__restrict__ y_res,
                                     int C, int C16, long HW, long total, int abs_input, float res_scale) {
  // one thread per (b, chunk, pixel): writes 16 halfs (32 B)
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long p = e % HW;
    const int cc = (int)((e / HW) % C16);
    const long b = e / (HW * C16);
    half8 lo, hi, rlo, rhi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c0 = cc * 16 + j, c1 = c0 + 8;
      float v0 = (c0 < C) ? x[((size_t)b * C + c0) * HW + p] : 0.f;
      float v1 = (c1 < C) ? x[((size_t)b * C + c1) * HW + p] : 0.f;
      if (abs_input & 1) { v0 = fabsf(v0); v1 = fabsf(v1); }
      if (abs_input & 2) {  // (x / 16)^2: the GDN norm operand, pre-scaled so that |x| up to ~4000 stays inside fp16
        v0 *= 0.0625f; v1 *= 0.0625f;
        v0 *= v0; v1 *= v1;
      }
      v0 = pin_f32(v0); v1 = pin_f32(v1);  // one fp32 value for the high part and its residual (mfma_common.hpp)
      lo[j] = (_Float16)v0;
      hi[j] = (_Float16)v1;
      rlo[j] = (_Float16)((v0 - (float)lo[j]) * res_scale);  // scaled up so the residual keeps all 11 bits (no fp16 subnormals)
      rhi[j] = (_Float16)((v1 - (float)hi[j]) * res_scale);
    }
    half8 *dst = reinterpret_cast<half8 *>(y + (size_t)e * 16);
    dst[0] = lo;
    dst[1] = hi;
    if (y_res) {
      half8 *rd = reinterpret_cast<half8 *>(y_res + (size_t)e * 16);
      rd[0] = rlo;
      rd[1] = rhi;
    }
  }
}

// The split operand of the ONE-launch fp32 convolution (licos_hip.h, licos_nchw_f32_split3_blk16): x = hi + (x - hi) as
// 3 C channels [hi 2^-5 | (x - hi) 2^6 | hi], to be met by weights [(w - w_hi) 2^5 | w_hi 2^-6 | w_hi] along cin - the
// three products hi.lo + lo.hi + hi.hi become one K loop over 3 C channels, one fp32 accumulator, one store.  The small
// cross terms come FIRST: every step of the K loop rounds the running sum, and while that sum is 2^-11 of its final
// size those roundings cost nothing - the other order measured 2.1e-6 of max|y| on a 128 -> 192 layer, this one 1e-6.
// FAN (C a multiple of 16): a thread reads 16 channels of a pixel once and writes its three 32-byte pieces (chunks cc,
// C16 + cc, 2 C16 + cc); otherwise a thread builds one output chunk from whichever channels fall into it.
template <bool FAN>
__global__ void nchw_to_blk16_split3_kernel(const float *__restrict__ x, _Float16 *__restrict__ y, int C, int C16out, long HW,
                                            long total, int flags) {
  const int C16in = FAN ? C / 16 : C16out;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long p = e % HW;
    const int cc = (int)((e / HW) % C16in);
    const long b = e / (HW * C16in);
    half8 out[3][2];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ch = cc * 16 + j;                      // FAN: input channel; else: output channel
      const int part = FAN ? 0 : ch / C, c = FAN ? ch : ch - part * C;
      float v = (FAN || ch < 3 * C) ? x[((size_t)b * C + c) * HW + p] : 0.f;
      if (flags & 1) v = fabsf(v);
      v = pin_f32(v);  // one fp32 value for the high part and its residual (mfma_common.hpp)
      const _Float16 hi = (_Float16)v;
      const float hf = (float)hi;
      const _Float16 mid = (_Float16)(hf * 0.03125f), lo = (_Float16)((v - hf) * 64.f);
      if (FAN) {
        out[0][j >> 3][j & 7] = mid;
        out[1][j >> 3][j & 7] = lo;
        out[2][j >> 3][j & 7] = hi;
      } else {
        out[0][j >> 3][j & 7] = part == 0 ? mid : part == 1 ? lo : hi;
      }
    }
#pragma unroll
    for (int part = 0; part < (FAN ? 3 : 1); ++part) {
      half8 *dst = reinterpret_cast<half8 *>(y + (((size_t)b * C16out + (size_t)part * C16in + cc) * HW + p) * 16);
      dst[0] = out[part][0];
      dst[1] = out[part][1];
    }
  }
}

__global__ void blk16_to_nchw_kernel(const _Float16 *__restrict__ x, float *__restrict__ y, int C, int C16, long HW,
                                     long total) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long p = e % HW;
    const int c = (int)((e / HW) % C);
    const long b = e / (HW * C);
    y[e] = (float)x[(((size_t)b * C16 + (c >> 4)) * HW + p) * 16 + (c & 15)];
  }
}

}  // namespace licos

using namespace licos;

extern "C" {

size_t licos_packed_conv_w_bytes(int Cin, int Cout) {
  const int MT = mt_for(Cout);
  if (MT == 0 || Cin <= 0) return 0;
  return (size_t)((Cin + 15) / 16) * 25 * MT * 1024;
}

int licos_pack_conv_w_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  const int MT = mt_for(Cout);
  LICOS_REQUIRE(w && packed && Cin > 0 && MT > 0, "pack_conv_w_f16: unsupported Cin=%d Cout=%d", Cin, Cout);
  const int Cin16 = (Cin + 15) / 16;
  const long total = (long)Cin16 * 25 * MT * 512;
  hipLaunchKernelGGL(pack_conv_w_kernel, dim3(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096), dim3(256), 0,
                     as_stream(stream), w, Cin, Cout, Cin16, MT, static_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_pack_deconv_w_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  const int MT = mt_for(Cout);
  LICOS_REQUIRE(w && packed && Cin > 0 && MT > 0, "pack_deconv_w_f16: unsupported Cin=%d Cout=%d", Cin, Cout);
  const int Cin16 = (Cin + 15) / 16;
  const long total = (long)Cin16 * 25 * MT * 512;
  hipLaunchKernelGGL(pack_deconv_w_kernel, dim3(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096), dim3(256), 0,
                     as_stream(stream), w, Cin, Cout, Cin16, MT, static_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

static int fewch_rows(int Cout) { return Cout <= 4 ? 4 : Cout <= 8 ? 8 : Cout <= 16 ? 16 : 32; }

size_t licos_packed_deconv_w_fewch_bytes(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cout > 32) return 0;
  if (fewch_uses_16x16x32(Cin, Cout)) return (size_t)(Cin / 32) * 25 * 1024;
  const int RP = fewch_rows(Cout), TPP = 32 / RP, WP = (25 + TPP - 1) / TPP;
  return (size_t)((Cin + 15) / 16) * WP * 1024;
}

int licos_pack_deconv_w_fewch_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  LICOS_REQUIRE(w && packed && Cin > 0 && Cout > 0 && Cout <= 32, "pack_deconv_w_fewch_f16: unsupported Cin=%d Cout=%d", Cin, Cout);
  if (fewch_uses_16x16x32(Cin, Cout)) {
    const long total = (long)(Cin / 32) * 25 * 512;
    hipLaunchKernelGGL(pack_deconv_w_few16_kernel, dim3(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096), dim3(256), 0,
                       as_stream(stream), w, Cin, Cout, static_cast<_Float16 *>(packed), total);
    LICOS_LAUNCH_CHECK();
    return LICOS_OK;
  }
  const int RP = fewch_rows(Cout), TPP = 32 / RP, WP = (25 + TPP - 1) / TPP, Cin16 = (Cin + 15) / 16;
  const long total = (long)Cin16 * WP * 512;
  hipLaunchKernelGGL(pack_deconv_w_fewch_kernel, dim3(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096), dim3(256), 0,
                     as_stream(stream), w, Cin, Cout, Cin16, RP, static_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

size_t licos_packed_gdn_bytes(int C) {
  const int MT = mt_for(C);
  if (MT == 0) return 0;
  return (size_t)MT * MT * 2 * 1024 + (size_t)32 * MT * sizeof(float);
}

int licos_pack_gdn_bf16(const float *beta_raw, const float *gamma_raw, float beta_bound, float gamma_bound,
                        float pedestal, int C, void *packed, void *stream) {
  const int MT = mt_for(C);
  LICOS_REQUIRE(beta_raw && gamma_raw && packed && MT > 0, "pack_gdn_bf16: unsupported C=%d", C);
  __bf16 *g = static_cast<__bf16 *>(packed);
  float *bta = reinterpret_cast<float *>(static_cast<unsigned char *>(packed) + (size_t)MT * MT * 2 * 1024);
  hipLaunchKernelGGL(pack_gdn_kernel, dim3(64), dim3(256), 0, as_stream(stream), beta_raw, gamma_raw, beta_bound,
                     gamma_bound, pedestal, C, MT, g, bta);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

size_t licos_packed_gdn_f32split_bytes(int C) {
  if (C <= 0 || C > 128) return 0;  // LICOS_EPI_NORM32 is instantiated for the 128-channel (four-tile) kernels
  return (size_t)4 * 4 * 2 * 2 * 1024 + (size_t)32 * 4 * sizeof(float);
}

int licos_pack_gdn_f32split(const float *beta_raw, const float *gamma_raw, float beta_bound, float gamma_bound,
                            float pedestal, int C, void *packed, void *stream) {
  LICOS_REQUIRE(beta_raw && gamma_raw && packed && licos_packed_gdn_f32split_bytes(C) > 0 && mt_for(C) == 4, "pack_gdn_f32split: unsupported C=%d (65..128)", C);
  const int MT = 4;
  _Float16 *g = static_cast<_Float16 *>(packed);
  float *bta = reinterpret_cast<float *>(static_cast<unsigned char *>(packed) + (size_t)MT * MT * 2 * 2 * 1024);
  hipLaunchKernelGGL(pack_gdn_f32split_kernel, dim3(64), dim3(256), 0, as_stream(stream), beta_raw, gamma_raw, beta_bound,
                     gamma_bound, pedestal, C, MT, g, bta);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_nchw_f32_to_s2d_blk16(const float *x, void *y_blk16, int B, int C, int H, int W, void *stream) {
  LICOS_REQUIRE(x && y_blk16 && B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "nchw_f32_to_s2d_blk16: bad arguments (H, W must be even)");
  LICOS_REQUIRE(((uintptr_t)x & 7) == 0, "nchw_f32_to_s2d_blk16: input must be 8-byte aligned");
  const int C16 = (4 * C + 15) / 16;
  const long total = (long)B * C16 * (H / 2) * (W / 2);
  hipLaunchKernelGGL(nchw_to_s2d_blk16_kernel, dim3(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192), dim3(256), 0,
                     as_stream(stream), x, static_cast<_Float16 *>(y_blk16), C, C16, H, W, total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_pack_conv_w_s2d_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  const int MT = mt_for(Cout);
  LICOS_REQUIRE(w && packed && Cin > 0 && MT > 0, "pack_conv_w_s2d_f16: unsupported Cin=%d Cout=%d", Cin, Cout);
  const int C16 = (4 * Cin + 15) / 16;
  const long total = (long)C16 * 9 * MT * 512;
  hipLaunchKernelGGL(pack_conv_w_s2d_kernel, dim3(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096), dim3(256), 0,
                     as_stream(stream), w, Cin, Cout, C16, MT, static_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_pack_conv3x3_w_f16(const float *w, int Cin, int Cout, void *packed, void *stream) {
  const int MT = mt_for(Cout);
  LICOS_REQUIRE(w && packed && Cin > 0 && MT > 0, "pack_conv3x3_w_f16: unsupported Cin=%d Cout=%d", Cin, Cout);
  const int C16 = (Cin + 15) / 16;
  const long total = (long)C16 * 9 * MT * 512;
  hipLaunchKernelGGL(pack_conv3x3_w_kernel, dim3(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096), dim3(256), 0,
                     as_stream(stream), w, Cin, Cout, C16, MT, static_cast<_Float16 *>(packed), total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

static int nchw_to_blk16_launch(const float *x, void *y_blk16, void *y_res, int B, int C, int H, int W, int abs_input,
                                int res_shift, void *stream, const char *who) {
  LICOS_REQUIRE(x && y_blk16 && B > 0 && C > 0 && H > 0 && W > 0, "%s: bad arguments", who);
  const int C16 = (C + 15) / 16;
  const long HW = (long)H * W, total = (long)B * C16 * HW;
  hipLaunchKernelGGL(nchw_to_blk16_kernel, dim3(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192), dim3(256), 0,
                     as_stream(stream), x, static_cast<_Float16 *>(y_blk16), static_cast<_Float16 *>(y_res), C, C16, HW, total,
                     abs_input, ldexpf(1.f, res_shift));
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_nchw_f32_to_blk16(const float *x, void *y_blk16, int B, int C, int H, int W, int abs_input, void *stream) {
  return nchw_to_blk16_launch(x, y_blk16, nullptr, B, C, H, W, abs_input, 0, stream, "nchw_f32_to_blk16");
}

int licos_nchw_f32_split_blk16(const float *x, void *y_hi_blk16, void *y_lo_blk16, int B, int C, int H, int W, int abs_input,
                               int lo_shift, void *stream) {
  LICOS_REQUIRE(y_lo_blk16 && lo_shift >= 0 && lo_shift <= 24, "nchw_f32_split_blk16: bad residual buffer / shift");
  return nchw_to_blk16_launch(x, y_hi_blk16, y_lo_blk16, B, C, H, W, abs_input, lo_shift, stream, "nchw_f32_split_blk16");
}

int licos_nchw_f32_split3_blk16(const float *x, void *y_blk16, int B, int C, int H, int W, int abs_input, void *stream) {
  LICOS_REQUIRE(x && y_blk16 && B > 0 && C > 0 && H > 0 && W > 0, "nchw_f32_split3_blk16: bad arguments");
  const int C16out = (3 * C + 15) / 16;
  const long HW = (long)H * W;
  if (C % 16 == 0) {
    const long total = (long)B * (C / 16) * HW;
    hipLaunchKernelGGL(nchw_to_blk16_split3_kernel<true>, dim3(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192), dim3(256), 0,
                       as_stream(stream), x, static_cast<_Float16 *>(y_blk16), C, C16out, HW, total, abs_input);
  } else {
    const long total = (long)B * C16out * HW;
    hipLaunchKernelGGL(nchw_to_blk16_split3_kernel<false>, dim3(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192), dim3(256), 0,
                       as_stream(stream), x, static_cast<_Float16 *>(y_blk16), C, C16out, HW, total, abs_input);
  }
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

int licos_blk16_to_nchw_f32(const void *x_blk16, float *y, int B, int C, int H, int W, void *stream) {
  LICOS_REQUIRE(x_blk16 && y && B > 0 && C > 0 && H > 0 && W > 0, "blk16_to_nchw_f32: bad arguments");
  const int C16 = (C + 15) / 16;
  const long HW = (long)H * W, total = (long)B * C * HW;
  hipLaunchKernelGGL(blk16_to_nchw_kernel, dim3(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192), dim3(256), 0,
                     as_stream(stream), static_cast<const _Float16 *>(x_blk16), y, C, C16, HW, total);
  LICOS_LAUNCH_CHECK();
  return LICOS_OK;
}

// 16 zero bytes in device memory: the LDS-DMA source for halo granules that fall outside the image.
static const void *zero_page() {
  static void *p = [] {
    void *q = nullptr;
    if (hipMalloc(&q, 256) != hipSuccess) return (void *)nullptr;
    if (hipMemset(q, 0, 256) != hipSuccess) return (void *)nullptr;
    return q;
  }();
  return p;
}

// the kernels' epilogue code of a public `epilogue` word: (I)GDN with LICOS_EPI_NORM32 -> EPI_GDN32 / EPI_IGDN32
static int kernel_epi(int epilogue) {
  const int epi = epilogue & 0xff;
  if ((epilogue & LICOS_EPI_NORM32) && (epi == EPI_GDN || epi == EPI_IGDN)) return epi == EPI_GDN ? EPI_GDN32 : EPI_IGDN32;
  return epi;
}

static int fill_args(MfmaArgs &a, const void *x, const void *wp, const float *bias, const void *gdn, int epi,
                     void *y_blk, float *y_nchw, int B, int Cin, int H, int W, int Cout, int *MT_out, const char *who) {
  LICOS_REQUIRE(x && wp && bias, "%s: NULL buffer", who);
  LICOS_REQUIRE((y_blk != nullptr) != (y_nchw != nullptr), "%s: exactly one of y_blk16 / y_nchw must be given", who);
  LICOS_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0, "%s: bad shape", who);
  const int MT = mt_for(Cout);
  LICOS_REQUIRE(MT > 0 && Cout > 0, "%s: Cout=%d unsupported (max 320)", who, Cout);
  LICOS_REQUIRE((epi != EPI_GDN && epi != EPI_IGDN) || gdn, "%s: (I)GDN epilogue needs packed gamma/beta", who);
  const int accum = (epi & LICOS_EPI_ACCUMULATE) ? 1 : 0;
  const int down = (epi >> 12) & 63;  // LICOS_EPI_SCALE_DOWN(k)
  const int norm32 = (epi & LICOS_EPI_NORM32) ? 1 : 0, split3 = (epi & LICOS_EPI_OUT_SPLIT3) ? 1 : 0;
  epi &= 0xff;
  LICOS_REQUIRE(!norm32 || ((epi == EPI_GDN || epi == EPI_IGDN) && MT == 4), "%s: LICOS_EPI_NORM32 goes with an (I)GDN epilogue over 65..128 channels", who);
  LICOS_REQUIRE(!split3 || (y_blk && Cout % 16 == 0 && !accum), "%s: LICOS_EPI_OUT_SPLIT3 needs a blk16 output buffer (3 Cout channels) and Cout a multiple of 16", who);
  LICOS_REQUIRE(down == 0 || accum, "%s: LICOS_EPI_SCALE_DOWN goes with LICOS_EPI_ACCUMULATE", who);
  LICOS_REQUIRE(epi >= 0 && epi <= 3, "%s: bad epilogue %d", who, epi);
  LICOS_REQUIRE(!accum || (y_nchw && epi != EPI_GDN && epi != EPI_IGDN), "%s: LICOS_EPI_ACCUMULATE needs an NCHW fp32 output and no (I)GDN", who);
  LICOS_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wp & 15) == 0 && ((uintptr_t)bias & 15) == 0, "%s: buffers must be 16-byte aligned", who);
  a.x = static_cast<const _Float16 *>(x);
  a.wp = static_cast<const half8 *>(wp);
  a.bias = bias;
  a.gamma = static_cast<const bf16x8 *>(gdn);
  a.beta = gdn ? reinterpret_cast<const float *>(static_cast<const unsigned char *>(gdn) + (size_t)MT * MT * 2 * 1024 * (norm32 ? 2 : 1)) : nullptr;
  a.out_split3 = split3;
  a.y_blk = static_cast<_Float16 *>(y_blk);
  a.y_nchw = y_nchw;
  a.B = B;
  a.Cin16 = (Cin + 15) / 16;
  a.H = H;
  a.W = W;
  a.Cout = Cout;
  a.clamp01 = 0;
  a.accum = accum;
  a.out_scale = ldexpf(1.f, -down);
  a.s1conv = 0;
  a.in_xsplit = a.out_xsplit = 0;
  a.zero16 = zero_page();
  LICOS_REQUIRE(a.zero16 != nullptr, "%s: could not allocate the zero page", who);
  *MT_out = MT;
  return LICOS_OK;
}

int licos_conv5x5s2_f16(const void *x_blk16, const void *w_packed, const float *bias, const void *gdn_packed,
                        int epilogue, void *y_blk16, float *y_nchw, int B, int Cin, int H, int W, int Cout,
                        void *stream) {
  MfmaArgs a{};
  int MT = 0;
  int rc = fill_args(a, x_blk16, w_packed, bias, gdn_packed, epilogue, y_blk16, y_nchw, B, Cin, H, W, Cout, &MT, "conv5x5s2_f16");
  if (rc != LICOS_OK) return rc;
  a.Ho = (H - 1) / 2 + 1;
  a.Wo = (W - 1) / 2 + 1;
  return mfma_dispatch_conv(a, MT, kernel_epi(epilogue), a.Wo, as_stream(stream));
}

int licos_conv5x5s2_f16_symbols(const void *x_blk16, const void *w_packed, const float *bias, const float *medians,
                                int32_t *symbols, int B, int Cin, int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(medians && symbols, "conv5x5s2_f16_symbols: NULL buffer");
  MfmaArgs a{};
  int MT = 0;
  int rc = fill_args(a, x_blk16, w_packed, bias, nullptr, EPI_NONE, nullptr, reinterpret_cast<float *>(symbols), B, Cin, H, W, Cout, &MT,
                     "conv5x5s2_f16_symbols");
  if (rc != LICOS_OK) return rc;
  a.Ho = (H - 1) / 2 + 1;
  a.Wo = (W - 1) / 2 + 1;
  a.sym_medians = medians;
  return mfma_dispatch_conv(a, MT, EPI_NONE, a.Wo, as_stream(stream));
}

int licos_deconv5x5s2_f16(const void *x_blk16, const void *w_packed, const float *bias, const void *gdn_packed,
                          int epilogue, void *y_blk16, float *y_nchw, int clamp01, int B, int Cin, int H, int W,
                          int Cout, void *stream) {
  MfmaArgs a{};
  int MT = 0;
  int rc = fill_args(a, x_blk16, w_packed, bias, gdn_packed, epilogue, y_blk16, y_nchw, B, Cin, H, W, Cout, &MT, "deconv5x5s2_f16");
  if (rc != LICOS_OK) return rc;
  a.Ho = 2 * H;
  a.Wo = 2 * W;
  a.clamp01 = clamp01;
  a.in_xsplit = (epilogue & LICOS_EPI_IN_XSPLIT) ? 1 : 0;
  a.out_xsplit = (epilogue & LICOS_EPI_OUT_XSPLIT) ? 1 : 0;
  if (a.in_xsplit || a.out_xsplit) {
    LICOS_REQUIRE(mfma_deconv8_applies(MT, a.Cin16, H, W, y_blk16 != nullptr, a.accum != 0, false),
                  "deconv5x5s2_f16: the x-split layout is not available for this stage (see licos_deconv5x5s2_f16_layouts)");
    LICOS_REQUIRE(!a.in_xsplit || W % 2 == 0, "deconv5x5s2_f16: x-split input needs an even width");
  }
  return mfma_dispatch_deconv(a, MT, kernel_epi(epilogue), W, as_stream(stream));
}

int licos_deconv5x5s2_f16_layouts(int Cin, int H, int W, int Cout) {
  if (Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return 0;
  if (!mfma_deconv8_applies(mt_for(Cout), (Cin + 15) / 16, H, W, true, false, false)) return 0;
  return LICOS_EPI_OUT_XSPLIT | (W % 2 == 0 ? LICOS_EPI_IN_XSPLIT : 0);
}

}  // extern "C"

extern "C" int licos_conv5x5s2_s2d_f16(const void *x_s2d_blk16, const void *w_packed_s2d, const float *bias,
                                       const void *gdn_packed, int epilogue, void *y_blk16, float *y_nchw, int B, int Cin,
                                       int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv5x5s2_s2d_f16: H and W must be even");
  MfmaArgs a{};
  int MT = 0;
  int rc = fill_args(a, x_s2d_blk16, w_packed_s2d, bias, gdn_packed, epilogue, y_blk16, y_nchw, B, 4 * Cin, H / 2, W / 2, Cout, &MT, "conv5x5s2_s2d_f16");
  if (rc != LICOS_OK) return rc;
  a.Ho = H / 2;
  a.Wo = W / 2;
  a.s1conv = 1;  // 3x3 stride-1 taps = output phase (0,0) of the transposed-conv kernel without the upsampling
  return mfma_dispatch_deconv(a, MT, kernel_epi(epilogue), W / 2, as_stream(stream));
}

extern "C" int licos_deconv5x5s2_fewch_f16(const void *x_blk16, const void *w_packed_fewch, const float *bias, float *y_nchw,
                                           int clamp01, int B, int Cin, int H, int W, int Cout, void *stream) {
  LICOS_REQUIRE(Cout > 0 && Cout <= 32, "deconv5x5s2_fewch_f16: Cout=%d (this kernel is for <= 32 output channels)", Cout);
  LICOS_REQUIRE(((uintptr_t)y_nchw & 7) == 0, "deconv5x5s2_fewch_f16: output must be 8-byte aligned");
  MfmaArgs a{};
  int MT = 0;
  int rc = fill_args(a, x_blk16, w_packed_fewch, bias, nullptr, EPI_NONE, nullptr, y_nchw, B, Cin, H, W, Cout, &MT, "deconv5x5s2_fewch_f16");
  if (rc != LICOS_OK) return rc;
  a.Ho = 2 * H;
  a.Wo = 2 * W;
  a.clamp01 = clamp01;
  return mfma_launch_deconv_fewch(a, as_stream(stream));
}

extern "C" int licos_conv3x3s1_f16(const void *x_blk16, const void *w_packed, const float *bias, const void *gdn_packed,
                                   int epilogue, void *y_blk16, float *y_nchw, int B, int Cin, int H, int W, int Cout,
                                   void *stream) {
  MfmaArgs a{};
  int MT = 0;
  int rc = fill_args(a, x_blk16, w_packed, bias, gdn_packed, epilogue, y_blk16, y_nchw, B, Cin, H, W, Cout, &MT, "conv3x3s1_f16");
  if (rc != LICOS_OK) return rc;
  a.Ho = H;
  a.Wo = W;
  a.s1conv = 1;
  return mfma_dispatch_deconv(a, MT, kernel_epi(epilogue), W, as_stream(stream));
}
