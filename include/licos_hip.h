/*
 * licos_hip.h - C ABI of the MI355X (gfx950) implementation of the LICOS
 * learned-image-compression hot path.
 *
 * This is the drop-in boundary.  The reference reaches the same
 * work through CompressAI's torch modules and its two pybind11 extensions;
 * each entry point below names the reference interface it replaces
 * (paths under /root/reference, or the upstream CompressAI file when the code
 * lives in that un-vendored dependency).
 *
 * Conventions
 *   - every function returns 0 on success, a negative LICOS_E* code otherwise;
 *     licos_last_error() returns a thread-local message for the last failure;
 *   - no exceptions cross the boundary, no ownership is transferred: every
 *     buffer is caller-owned (device pointers unless the name says host);
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and the
 *     library never synchronises the device;
 *   - activations of the 32-bit path are NCHW fp32 exactly like torch; the
 *     16-bit MFMA path keeps activations in the blocked layout
 *     [B][C/16][H][W][16] fp16 ("blk16") between stages.
 */
#ifndef LICOS_HIP_H
#define LICOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LICOS_OK 0
#define LICOS_EINVAL (-1)   /* bad argument / unsupported shape            */
#define LICOS_EHIP (-2)     /* HIP runtime error (message has the detail)  */
#define LICOS_EDOMAIN (-3)  /* invalid pmf, see licos_pmf_to_quantized_cdf */
#define LICOS_EOVERFLOW (-4)

#define LICOS_ABI_VERSION 1

typedef struct licos_device_props {
  int compute_units;
  int wavefront_size;
  int lds_bytes_per_cu;
  int clock_khz;
  size_t hbm_bytes;
  char arch[32]; /* e.g. "gfx950" */
} licos_device_props;

const char *licos_last_error(void);
int licos_abi_version(void);
/* device discovery (replaces torch.cuda.is_available()/device choice, licos/train.py:74) */
int licos_query(int device, licos_device_props *out);

/* ---------------------------------------------------------------- host side
 * CompressAI cpp_exts/ops/ops.cpp pmf_to_quantized_cdf, called from
 * EntropyBottleneck.update() <- /root/reference/eval_script.py:72,88.
 * cdf_out has n+1 entries.  LICOS_EDOMAIN on negative/non-finite/all-zero pmf. */
int licos_pmf_to_quantized_cdf(const float *pmf_host, int n, int precision, int32_t *cdf_out_host);

/* Encoder-side table derived from the integer CDFs: per (row, symbol) a 16-byte
 * record {rcp_freq:u64, bias:u32, freq:u16|rcp_shift:u16} implementing the
 * exact x/freq of rans64.h Rans64EncPut by reciprocal multiplication.
 * rows x stride int32 CDFs in, rows x stride records out (host memory). */
int licos_rans_build_enc_table(const int32_t *cdf_host, const int32_t *cdf_len_host, int rows, int stride,
                               void *table_out_host /* rows*stride*16 bytes */);

/* ------------------------------------------------- 32-bit path (NCHW fp32)
 * torch.nn.Conv2d.forward as instantiated by CompressAI models/google.py
 * `conv()` and licos/model_utils.py:31-37.  Cross-correlation, square kernel. */
#define LICOS_CONV_RELU 1      /* ReLU on the output (bmshj2018-factorized-relu, hyperprior h_a / h_s) */
#define LICOS_CONV_ABS_INPUT 2 /* |x| on the input   (ScaleHyperprior: h_a(torch.abs(y)))              */
int licos_conv2d_f32(const float *x, const float *w /*[Cout][Cin][K][K]*/, const float *bias /*nullable*/,
                     float *y, int B, int Cin, int H, int W, int Cout, int K, int stride, int pad,
                     int flags, void *stream);
/* torch.nn.ConvTranspose2d.forward (CompressAI `deconv()`, licos/model_utils.py:38-45);
 * w is [Cin][Cout][K][K]; Hout = (H-1)*stride - 2*pad + K + out_pad. */
int licos_deconv2d_f32(const float *x, const float *w, const float *bias, float *y, int B, int Cin, int H,
                       int W, int Cout, int K, int stride, int pad, int out_pad, int relu, void *stream);
/* CompressAI ops/parametrizers.py NonNegativeParametrizer applied to GDN's
 * beta (C) and gamma (C*C): out = max(raw, bound)^2 - pedestal. */
int licos_gdn_reparam_f32(const float *beta_raw, const float *gamma_raw, float beta_bound, float gamma_bound,
                          float pedestal, float *beta_eff, float *gamma_eff, int C, void *stream);
/* CompressAI layers/gdn.py GDN.forward: y = x * rsqrt(beta + gamma . x^2)
 * (sqrt when inverse != 0).  x, y: [B][C][HW] fp32; gamma_eff [C][C]. */
int licos_gdn_f32(const float *x, const float *gamma_eff, const float *beta_eff, float *y, int B, int C,
                  int HW, int inverse, void *stream);
/* The same layer with its result written as the split operand of the NEXT layer's one-launch fp32 convolution
 * (blk16 fp16, 3 C channels: licos_nchw_f32_split3_blk16's layout) instead of NCHW fp32 - the inference chain of the
 * fp32 parity path ((I)GDN followed by a 5x5 stride-2 (transposed) convolution).  Served by the one-pass matrix-core
 * kernel only: licos_gdn_f32_split3_applies(C, HW) != 0 (128 channels, HW a multiple of 32). */
int licos_gdn_f32_split3_applies(int C, int HW);
int licos_gdn_f32_split3(const float *x, const float *gamma_eff, const float *beta_eff, void *y_blk16, int B, int C,
                         int HW, int inverse, void *stream);

/* ------------------------------------------------ backward, 32-bit path
 * What licos/train.py:193 (`loss.backward()`) computes through torch autograd for the transforms.
 * Input gradients need no entry point of their own: dgrad(Conv2d) = licos_deconv2d_f32 and
 * dgrad(ConvTranspose2d) = licos_conv2d_f32 on the same weight tensor.
 *
 * Weight gradient of a K x K, stride-S cross-correlation: dw[co][ci][ky][kx] = sum_{b,oy,ox} g[b][co][oy][ox] *
 * inp[b][ci][oy*S-pad+ky][ox*S-pad+kx] (inp squared first when square_input != 0: the GDN gamma gradient as a 1x1 case).
 * Conv2d: inp = x, g = dy.  ConvTranspose2d: inp = dy, g = x (dw then has the [Cin][Cout][K][K] layout). dw is overwritten. */
int licos_conv2d_wgrad_f32(const float *inp /*[B][Ci][H][W]*/, const float *g /*[B][Co][Ho][Wo]*/, float *dw, int B,
                           int Ci, int H, int W, int Co, int K, int stride, int pad, int square_input, void *stream);
/* The GDN gamma gradient (the 1x1, square_input case above) for 128 channels on the matrix cores:
 * dgamma_eff[i][j] = sum_{b,p} t[b][i][p] * x[b][j][p]^2, fp32-grade through the three-pass fp16 split, partial matrices per
 * workgroup added in a fixed order (bit-reproducible); t is staged as t * 2^k, k from a max|t| pre-pass, so gradients far
 * below fp16's range (1e-6 .. 1e-12) keep their bits.  scratch: licos_gdn_gamma_grad_parts(B, HW) * 128 * 128 + 4 floats. */
int licos_gdn_gamma_grad_parts(int B, long HW);
int licos_gdn_gamma_grad_f32(const float *t, const float *x, float *scratch, float *dgamma, int B, int C, long HW, void *stream);
/* the same with max|t| already in the scratch's last word (written by licos_gdn_bwd_fused_f32) */
int licos_gdn_gamma_grad_scaled_f32(const float *t, const float *x, float *scratch, float *dgamma, int B, int C, long HW, void *stream);
int licos_bias_grad_f32(const float *dy /*[B][C][HW]*/, float *db /*[C]*/, int B, int C, long HW, void *stream);
/* GDN backward: dx, and t = dL/dnorm (dgamma_eff = licos_conv2d_wgrad_f32(x, t, K=1, square_input=1), dbeta_eff =
 * licos_bias_grad_f32(t)); gamma_t_scratch: C*C floats. */
int licos_gdn_bwd_f32(const float *x, const float *dy, const float *gamma_eff, const float *beta_eff,
                      float *gamma_t_scratch, float *dx, float *t_out, int B, int C, int HW, int inverse, void *stream);
/* The same backward pass for the shapes of licos_gdn_f32_split3_applies (128 channels, HW a multiple of 32) as ONE kernel on
 * the matrix cores: licos_gdn_f32_fwd_norm is licos_gdn_f32 that also writes norm = beta + gamma . x^2 (NCHW fp32),
 * licos_gdn_bwd_fused_f32 turns x, dy and that norm into dx and t (12 B in, 8 B out per element; each pixel's column of t
 * is scaled by a power of two before its fp16 split, so gradients of any magnitude keep their bits).  t_absmax
 * (nullable, 4 bytes): receives the bit pattern of max|t| - pass the last word of licos_gdn_gamma_grad_f32's scratch and call
 * licos_gdn_gamma_grad_scaled_f32, which then skips its own pass over t. */
int licos_gdn_f32_fwd_norm(const float *x, const float *gamma_eff, const float *beta_eff, float *y, float *norm_out, int B, int C,
                           int HW, int inverse, void *stream);
int licos_gdn_bwd_fused_f32(const float *x, const float *dy, const float *norm, const float *gamma_eff, float *dx, float *t_out,
                            void *t_absmax, int B, int C, int HW, int inverse, void *stream);
/* NonNegativeParametrizer backward incl. CompressAI's LowerBound gradient rule. */
int licos_reparam_bwd_f32(const float *raw, const float *d_eff, float bound, float *d_raw, long n, void *stream);

/* torch.optim.Adam.step (licos/train.py:196,200; defaults amsgrad=False, weight_decay=0) on one flat fp32 tensor:
 * m = b1 m + (1-b1) g', v = b2 v + (1-b2) g'^2, p -= lr / (1-b1^t) * m / (sqrt(v / (1-b2^t)) + eps), g' = g * grad_scale
 * (grad_scale carries clip_grad_norm_'s coefficient, train.py:194-195). */
int licos_adam_f32(float *p, const float *g, float *m, float *v, long n, float lr, float beta1, float beta2, float eps,
                   int step, float grad_scale, void *stream);
/* sum of squares into *out (double, zeroed by the caller): the total-norm reduction of clip_grad_norm_. */
int licos_sumsq_f32(const float *x, long n, double *out, void *stream);

/* ------------------------------------------- entropy bottleneck (fp32 math)
 * CompressAI entropy_models/entropy_models.py EntropyBottleneck, constructed at
 * licos/model_utils.py:25-29.
 *
 * licos_eb_pack: softplus(matrices) / biases / tanh(factors) of all layers
 * flattened per channel into `packed` [C][per_channel] (per_channel returned by
 * licos_eb_packed_size).  nfilt = len(filters); filters e.g. {3,3,3,3}. */
int licos_eb_packed_size(const int *filters_host, int nfilt);
int licos_eb_pack(const float *const *matrices_dev_host_array, const float *const *biases_dev_host_array,
                  const float *const *factors_dev_host_array, const int *filters_host, int nfilt, int C,
                  float *packed, void *stream);
/* EntropyModel.quantize.  y, noise, y_hat: [B][C][HW] fp32; medians [C].
 * mode 0 ("dequantize"): y_hat = rint(y - m) + m (half-to-even, as torch.round)
 * mode 1 ("noise"):      y_hat = y + noise
 * mode 2 ("symbols"):    only symbols are written.
 * symbols (nullable unless mode 2): int32, element (b, c, p) at
 *   b*sym_stride_b + (c*HW + p)*sym_stride_i   (lets the coder read coalesced).
 * y_hat_blk16 (nullable): also writes rint(y-m)+m as fp16 blk16 for the 16-bit g_s. */
int licos_eb_quantize(const float *y, const float *medians, const float *noise, float *y_hat, int32_t *symbols,
                      long sym_stride_b, long sym_stride_i, int mode, int B, int C, int HW, void *stream);
/* EntropyBottleneck._likelihood + LowerBound: lik = max(sigmoid(F(v+.5)) -
 * sigmoid(F(v-.5)), bound); form 0 = current releases, 1 = sign-flip variant
 * of the 1.1/1.2-era releases.  sum_log2 (nullable, double[B], zeroed by the
 * caller) receives sum over (c,p) of log2(lik) per image - the bpp numerator of
 * /root/reference/eval_utils.py:172-186 and RateDistortionLoss. */
int licos_eb_likelihood(const float *v, const float *packed, const int *filters_host, int nfilt, float *lik,
                        float bound, int form, double *sum_log2, int B, int C, int HW, void *stream);
/* Backward of licos_eb_likelihood ([CAI] EntropyBottleneck._likelihood + LowerBound's gradient rule: the gradient
 * passes where lik >= bound or g < 0; /root/reference/licos/train.py:186-200 loss.backward()).  g_lik = dL/dlik.
 * dv = dL/dv; dparams_slices [licos_eb_likelihood_bwd_slices(B, HW)][C][per_channel] in licos_eb_pack's record order
 * (matrix, bias, factor per layer) but w.r.t. the RAW parameters; the caller adds the slices. */
int licos_eb_likelihood_bwd_slices(int B, int HW);
int licos_eb_likelihood_bwd(const float *v, const float *g_lik, const float *packed, const int *filters_host, int nfilt,
                            float bound, int form, float *dv, float *dparams_slices, int B, int C, int HW, void *stream);
/* Backward of licos_gc_likelihood: dL/dv and dL/dscales (both LowerBounds with CompressAI's gradient rule). */
int licos_gc_likelihood_bwd(const float *v, const float *scales, const float *g_lik, float scale_bound, float lik_bound,
                            float *dv, float *dscale, long n, void *stream);
/* out = g * (ref > 0) (mode 0: ReLU backward) or g * sign(ref) (mode 1: |x| backward) - the point-wise masks of the
 * hyper transforms' backward pass ([CAI] models/google.py ScaleHyperprior h_a / h_s). */
int licos_mask_mul_f32(const float *g, const float *ref, float *out, long n, int mode, void *stream);
/* symbols -> y_hat = float(symbol) + median (EntropyModel.dequantize);
 * writes NCHW fp32 (nullable) and/or blk16 fp16 (nullable). */
int licos_eb_dequantize(const int32_t *symbols, long sym_stride_b, long sym_stride_i, const float *medians,
                        float *y_hat_nchw, void *y_hat_blk16, int B, int C, int H, int W, void *stream);

/* --------------------------------------------- Gaussian conditional (ScaleHyperprior, config 5)
 * CompressAI entropy_models.py GaussianConditional._likelihood + LowerBound:
 * s = max(scales, scale_bound); lik = max(Phi((.5-|v|)/s) - Phi((-.5-|v|)/s), lik_bound), Phi(x) = erfc(-x/sqrt 2)/2.
 * v, scales, lik: [B][C][HW] fp32; sum_log2 as in licos_eb_likelihood. */
int licos_gc_likelihood(const float *v, const float *scales, float *lik, float scale_bound, float lik_bound,
                        double *sum_log2, int B, int C, int HW, void *stream);
/* GaussianConditional.build_indexes: idx = (levels-1) - #{t in table[:-1] : max(s, bound) <= t}; written in the
 * coder's addressing (element (b, i) at b*stride_b + i*stride_i, i = c*HW + p). */
int licos_gc_build_indexes(const float *scales, const float *table, int levels, float scale_bound, int32_t *indexes,
                           long stride_b, long stride_i, int B, long n, void *stream);
/* 16-bit symbols in the plain [stream][position] layout for the tiles the HOST codes (EntropyBottleneck.compress /
 * decompress, [CAI] entropy_models.py: symbols = round(y - median), y_hat = symbols + median) - half the PCIe bytes of the
 * int32 form.  symbols16: flag[0] |= 1 when a symbol does not fit int16 (use licos_eb_quantize then).
 * dequantize16: y_nchw fp32 [B][C][H][W] and / or y_blk16 fp16 [B][ceil(C/16)][H*W][16] (needs H*W % 64 == 0). */
int licos_eb_symbols16(const float *y, const float *medians, int16_t *symbols, int32_t *flag, int B, int C, int HW, void *stream);
int licos_eb_dequantize16(const int16_t *symbols, const float *medians, float *y_nchw, void *y_blk16, int B, int C, int H, int W,
                          void *stream);
/* What the host coder (licos_rans_*_host_packed / _rows8 below) needs of a y stream, in as few bytes as PCIe allows, when
 * the host cores code part of a scale-hyperprior call beside the device (GaussianConditional.compress / decompress,
 * [CAI] entropy_models.py: indexes = build_indexes(scales), symbols = round(y)).  Plain [stream][position] layout.
 * pack: packed[b*n + i] = row << 16 | (round(y) & 0xFFFF); flag[0] |= 1 if a symbol does not fit int16 (the caller must
 *       then use the int32 form).  rows8: one table-row byte per symbol (levels <= 256). */
int licos_gc_pack_symbols(const float *y, const float *scales, const float *table, int levels, float scale_bound, int32_t *packed,
                          int32_t *flag, int B, long n, void *stream);
int licos_gc_build_rows8(const float *scales, const float *table, int levels, float scale_bound, uint8_t *rows8, int B, long n,
                         void *stream);

/* Sentinel-2 raw DN -> model input grid (the step before the path):
 * /root/reference/licos/raw_image_folder.py:192-196 with use_full_range=False: x = DN / 4095 (raw_utils.py:128),
 * then skimage.img_as_ubyte(x) / 255 = rint(x * 255) / 255.  dn: uint16 [n]; out: fp32 [n]. */
int licos_dn12_to_grid8_f32(const uint16_t *dn, float *out, long n, int full_range, void *stream);
/* Band resampling of Sentinel-2 granules (/root/reference/licos/raw_utils.py:134-244: image_band_upsample /
 * image_band_reshape call torch.nn.functional.interpolate(mode="bilinear"), align_corners=True when up-sampling,
 * the default (False) when down-sampling).  src: [planes][Hin][Win] fp32 -> dst: [planes][Hout][Wout].  scale_h /
 * scale_w are the source-coordinate steps the caller derives as ATen does: (in-1)/(out-1) with align_corners, else
 * 1/scale_factor; a dimension with Hin == Hout (Win == Wout) is copied. */
int licos_resample_bilinear_f32(const float *src, float *dst, long planes, int Hin, int Win, int Hout, int Wout,
                                float scale_h, float scale_w, int align_corners, void *stream);
/* Cuts (B, C, H, W) images into T x T tiles (zero padded at the right/bottom edge) laid out as a tile batch
 * (B*ny*nx, C, T, T), and the inverse (crop back).  Whole granules (raw_utils.py:131: up to 2304 x 2592) then
 * ride the batched tile codec instead of one 4.5-M-symbol stream. */
int licos_tile_f32(const float *img, float *tiles, int B, int C, int H, int W, int T, void *stream);
int licos_untile_f32(const float *tiles, float *img, int B, int C, int H, int W, int T, void *stream);
/* The same with overlapping tiles: tile (iy, ix) starts at (iy*S - margin, ix*S - margin), S = T - 2*margin, zero outside
 * the image; the inverse writes back only each tile's central S x S pixels.  There are ceil(H/S) x ceil(W/S) tiles per
 * image.  The reference codes whole images (eval_script.py:138-165) and so has no tile seams; a margin of a few latent
 * cells (e.g. 32) keeps the transforms' zero padding at tile borders out of the reconstruction. */
int licos_tile_overlap_f32(const float *img, float *tiles, int B, int C, int H, int W, int T, int margin, void *stream);
int licos_untile_overlap_f32(const float *tiles, float *img, int B, int C, int H, int W, int T, int margin, void *stream);

/* mean-squared-error numerator: sum over all elements of (a-b)^2 into *out (double, zeroed by caller);
 * /root/reference/eval_utils.py:145-156, RateDistortionLoss mse term.  clamp01 != 0 clamps `a` first. */
int licos_reduce_sqdiff(const float *a, const float *b, long n, int clamp01, double *out, void *stream);

/* One scale of MS-SSIM (/root/reference/eval_utils.py:159-169 -> pytorch_msssim.ms_ssim, a dependency absent from
 * the reference tree): 11-tap window (HOST pointer, 11 floats), "valid" filtering of x, y, x^2, y^2, xy, then
 * cs = (2 s12 + C2)/(s1 + s2 + C2), ssim = (2 m1 m2 + C1)/(m1^2 + m2^2 + C1) * cs.  x, y: [planes][H][W] fp32;
 * sums: double [planes][2] (zeroed by the caller) receives the sums of the ssim and cs maps over the
 * (H-10) x (W-10) valid region. */
int licos_ssim_stats_f32(const float *x, const float *y, int planes, int H, int W, const float *window11, float C1, float C2,
                         double *sums, void *stream);

/* ------------------------------------------------------------------- rANS
 * CompressAI cpp_exts/rans/rans_interface.cpp RansEncoder.encode_with_indexes /
 * RansDecoder.decode_with_indexes (reached from /root/reference/eval_utils.py:201),
 * batched: one stream per image, one GPU lane per stream.
 *
 * symbols: int32, stream b, position i at b*sym_stride_b + i*sym_stride_i.
 * indexes: nullable; same addressing; when NULL the CDF row of position i is
 *          i / plane  (EntropyBottleneck._build_indexes: the channel id).  With explicit indexes
 *          (GaussianConditional) `plane` may carry the number of CDF rows (<= 256) to select the kernel
 *          that stages the short rows in LDS; 0 selects the generic kernel.
 * cdf [rows][cdf_stride] int32, cdf_len[rows], offset[rows]: device copies of
 *          _quantized_cdf / _cdf_length / _offset.
 * enc_table: device copy of licos_rans_build_enc_table's output.
 * words: scratch uint32 [cap_words][B] (word w of stream b at w*B + b, filled
 *          from the top); nwords[B] receives each stream's word count;
 *          status[0] is set non-zero if any stream overflowed cap_words. */
int licos_rans_encode_batch(const int32_t *symbols, const int32_t *indexes, long sym_stride_b, long sym_stride_i,
                            int n, int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len,
                            const int32_t *offset, const void *enc_table, uint32_t *words, int cap_words,
                            int32_t *nwords, int32_t *status, int B, void *stream);
/* The same coder on the HOST cores, one std::thread per group of streams, bit-identical streams (CompressAI's
 * RansEncoder.encode_with_indexes / RansDecoder.decode_with_indexes are host code too; /root/reference/eval_script.py:138-165
 * codes one whole granule = ONE stream, where a single GPU lane loses to a single core).  All pointers are HOST memory.
 * Same symbol / index addressing as above; `rows` = number of CDF rows; enc_table = licos_rans_build_enc_table's output.
 * encode: stream b is written to out[b * cap_bytes_per_stream ...) (front-aligned), nbytes[b] = its length;
 *         LICOS_EOVERFLOW if a stream needs more than cap_bytes_per_stream (8 * n + 16 always suffices).
 * decode: stream b = in[byte_off[b] .. byte_off[b+1]); status[0] = 1 if a stream ended early (its remaining symbols are 0;
 *         never reads outside a stream).  nthreads <= 1 runs on the calling thread. */
int licos_rans_encode_host(const int32_t *symbols, const int32_t *indexes, long sym_stride_b, long sym_stride_i, int n,
                           int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset,
                           int rows, const void *enc_table, uint8_t *out, long cap_bytes_per_stream, int64_t *nbytes,
                           int B, int nthreads);
int licos_rans_decode_host(const uint8_t *in, const int64_t *byte_off /*[B+1]*/, const int32_t *indexes, long sym_stride_b,
                           long sym_stride_i, int n, int plane, const int32_t *cdf, int cdf_stride, const int32_t *cdf_len,
                           const int32_t *offset, int rows, int32_t *symbols, int32_t *status, int B, int nthreads);
/* The host coder on the compact forms licos_gc_pack_symbols / licos_gc_build_rows8 produce (explicit per-symbol rows):
 * encode from packed words [B][n] (stream b at packed + b*stride_b); decode with row bytes [B][n] into int32 symbols
 * [B][n] (both at b*stride_b).  Same streams, same errors as the two functions above. */
int licos_rans_encode_host_packed(const int32_t *packed, long stride_b, int n, const int32_t *cdf, int cdf_stride,
                                  const int32_t *cdf_len, const int32_t *offset, int rows, const void *enc_table, uint8_t *out,
                                  long cap_bytes_per_stream, int64_t *nbytes, int B, int nthreads);
int licos_rans_decode_host_rows8(const uint8_t *in, const int64_t *byte_off /*[B+1]*/, const uint8_t *rows8, long stride_b, int n,
                                 const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset, int rows,
                                 int32_t *symbols, int32_t *status, int B, int nthreads);
/* ... and on 16-bit symbols [B][n] with channel-plane rows (licos_eb_symbols16 / licos_eb_dequantize16).  decode:
 * status[0] = 3 when a decoded value does not fit int16 (decode the batch again with licos_rans_decode_host). */
int licos_rans_encode_host_sym16(const int16_t *symbols, long stride_b, int n, int plane, const int32_t *cdf, int cdf_stride,
                                 const int32_t *cdf_len, const int32_t *offset, int rows, const void *enc_table, uint8_t *out,
                                 long cap_bytes_per_stream, int64_t *nbytes, int B, int nthreads);
int licos_rans_decode_host_sym16(const uint8_t *in, const int64_t *byte_off /*[B+1]*/, long stride_b, int n, int plane,
                                 const int32_t *cdf, int cdf_stride, const int32_t *cdf_len, const int32_t *offset, int rows,
                                 int16_t *symbols, int32_t *status, int B, int nthreads);
/* gathers each stream's words (in stream order) into one packed little-endian
 * byte buffer: stream b occupies out[byte_off[b] .. byte_off[b] + 4*nwords[b]).
 * byte_off[B] = exclusive prefix sum of 4*nwords (computed by the caller). */
int licos_rans_compact(const uint32_t *words, int cap_words, const int32_t *nwords, const int64_t *byte_off,
                       uint8_t *out, int B, void *stream);
/* decoder: stream b = in[byte_off[b] .. byte_off[b+1]) ; writes symbols with the
 * same addressing as above; status[0] non-zero if a stream ran past its end.
 * Round 5: with per-channel tables (no indexes) and stream-major symbols (ssi == 1, ssb and plane multiples of 4,
 * 16-byte-aligned rows) four symbols leave as one 16-byte store from registers - the fastest form (csrc/rans.hip,
 * rans_decode_plane4_kernel); licos_eb_dequantize reads that layout with 16-byte loads. */
int licos_rans_decode_batch(const uint8_t *in, const int64_t *byte_off /*[B+1]*/, const int32_t *indexes,
                            long sym_stride_b, long sym_stride_i, int n, int plane, const int32_t *cdf,
                            int cdf_stride, const int32_t *cdf_len, const int32_t *offset, int32_t *symbols,
                            int32_t *status, int B, void *stream);

/* --- the scale-conditioned fast path ([CAI] entropy_models.py GaussianConditional.compress / decompress: one CDF row
 * per ELEMENT, chosen by the predicted scale; rans_interface.cpp encode_with_indexes / decode_with_indexes) -------------
 * Same streams as the generic entry points above, byte for byte; the work that does not depend on the coder state is
 * split off into throughput kernels so that the serial, one-lane-per-stream part touches no table (encode) or one
 * 8-byte LDS record (decode) per symbol.  No means (CompressAI's `means=None`): symbols are round(y).
 *
 * licos_gc_encode_prepare: y, scales [B][n] fp32 (NCHW order) -> rec [n][B] 16-byte records, aux [n][B] int32 (raw
 * escape values).  scale_table: the `levels` sorted scales of build_indexes(); enc_table: licos_rans_build_enc_table's
 * output on the device.  licos_rans_encode_records: the serial part; `words` holds (cap_words + 1) rows of B words - row 0
 * is a dump row, stream b's nwords[b] words end up in rows cap_words + 1 - nwords[b] .. cap_words (pass words + B to
 * licos_rans_compact); nwords / status as licos_rans_encode_batch (status is also set when a stream fills its rows exactly).
 *
 * licos_rans_image_build (host): turns the integer CDF table into the decoder image (meta, bucket records, 16-bit
 * symbol starts; licos_amd/csrc/rans_image.hpp) of at most budget_bytes (licos_rans_image_budget(waves) = what fits
 * in LDS beside the decode kernel's rings).  row_weight: optional expected use of each row (NULL = uniform) - it only
 * steers how the record budget is split between rows (a row used more gets finer buckets, so fewer values take the slow
 * search); any weighting decodes every stream exactly.  licos_gc_decode_prepare can collect the statistic (row_hist).
 * licos_rans_image_lookup (host, test hook): symbol and [lo, hi) for a 16-bit value; returns 1 if the slow search ran.
 *
 * licos_gc_decode_prepare: scales [B][n] -> row bytes idx16 [ceil(n/16)][B][16].  licos_rans_decode_image: the serial
 * part; `image` on the device, `image_host_header` = the first 32 bytes of the same image on the host. */
int licos_gc_encode_prepare(const float *y, const float *scales, const float *scale_table, int levels, float scale_bound,
                            const void *enc_table, int cdf_stride, const int32_t *cdf_len, const int32_t *offset, void *rec,
                            int32_t *aux, int B, long n, void *stream);
int licos_rans_encode_records(const void *rec, const int32_t *aux, long n, uint32_t *words, int cap_words, int32_t *nwords,
                              int32_t *status, int B, void *stream);
long licos_rans_image_budget(int waves);
int licos_rans_image_build(const int32_t *cdf_host, const int32_t *cdf_len_host, const int32_t *offset_host, int rows, int stride,
                           const float *row_weight, long budget_bytes, void *image_out_host, long *image_bytes);
int licos_rans_image_lookup(const void *image_host, int row, int cf, int32_t *symbol_lo_hi /*[3]*/);
int licos_gc_decode_prepare(const float *scales, const float *scale_table, int levels, float scale_bound, void *idx16,
                            unsigned int *row_hist /* optional uint32[256]: a sampled count of the rows in use, accumulated */,
                            int B, long n, void *stream);
int licos_rans_decode_image(const uint8_t *in, const int64_t *byte_off /*[B+1]*/, const void *idx16,
                            int rows_shared /* 1: idx16 is [ceil(n/16)][16], the same rows for every stream */, long n,
                            const void *image, const void *image_host_header, int32_t *symbols, long sym_stride_b,
                            long sym_stride_i, int32_t *status, int B, void *stream);
/* The same record encoder for the ENTROPY BOTTLENECK ([CAI] EntropyBottleneck.compress: row = channel of the element,
 * symbol = round(y - median[channel])): y [B][C][plane] fp32 -> rec / aux as above; decode with licos_rans_decode_image,
 * rows_shared = 1 and the channel pattern as idx16 (C <= 256). */
int licos_eb_encode_prepare(const float *y, const float *medians, int C, int plane, const void *enc_table, int cdf_stride,
                            const int32_t *cdf_len, const int32_t *offset, void *rec, int32_t *aux, int B, void *stream);

/* ------------------------------------------------------- federated averaging
 * Replaces the file-based pair-wise blend of /root/reference/licos/federation_utils.py:47-53 by one
 * collective over a flat fp32 bucket of the whole floating state: every rank scales its bucket by its
 * coefficient (this kernel), RCCL all-reduces it (torch.distributed "nccl"), and the result is
 * divided by the all-reduced coefficient sum carried in the bucket's last element.
 * x[i] *= alpha for i < n;  alpha_dev (nullable): device pointer whose value is used as 1/(*alpha_dev). */
int licos_scale_f32(float *x, long n, float alpha, const float *inv_alpha_dev, void *stream);

/* ----------------------------------------------- 16-bit MFMA path (gfx950)
 * The fused hot path: 5x5 stride-2 Conv2d (+GDN) and ConvTranspose2d (+IGDN)
 * stages of CompressAI models/google.py FactorizedPrior.g_a / g_s
 * (instantiated at licos/model_utils.py:19, run at licos/train.py:190 and
 * eval_utils.py:200-201) as LDS-tiled implicit GEMMs on v_mfma_f32_32x32x16_f16
 * with the GDN normalisation as a second (bf16) MFMA GEMM in the epilogue.
 *
 * Packed operands (device, produced once per weight version):
 *   conv weights  : licos_pack_conv_w_f16   [Cin16][25][Cout/32] MFMA A-fragments
 *   deconv weights: licos_pack_deconv_w_f16 same, taps regrouped per output phase
 *   GDN           : licos_pack_gdn_bf16     reparametrised gamma as bf16 A-fragments (k-permuted
 *                                           for accumulator-as-operand use) + fp32 beta            */
size_t licos_packed_conv_w_bytes(int Cin, int Cout);
int licos_pack_conv_w_f16(const float *w /*[Cout][Cin][5][5]*/, int Cin, int Cout, void *packed, void *stream);
int licos_pack_deconv_w_f16(const float *w /*[Cin][Cout][5][5]*/, int Cin, int Cout, void *packed, void *stream);
size_t licos_packed_gdn_bytes(int C);
/* for LICOS_EPI_NORM32 (0 bytes: C not served) */
size_t licos_packed_gdn_f32split_bytes(int C);
int licos_pack_gdn_f32split(const float *beta_raw, const float *gamma_raw, float beta_bound, float gamma_bound,
                            float pedestal, int C, void *packed, void *stream);
int licos_pack_gdn_bf16(const float *beta_raw, const float *gamma_raw, float beta_bound, float gamma_bound,
                        float pedestal, int C, void *packed, void *stream);
/* NCHW fp32 -> blk16 fp16 (channels zero-padded to a multiple of 16); abs_input != 0 stores |x| (ScaleHyperprior h_a) */
int licos_nchw_f32_to_blk16(const float *x, void *y_blk16, int B, int C, int H, int W, int abs_input, void *stream);
/* 3x3 stride-1 padding-1 Conv2d on the MFMA path (hyperprior h_a[0], h_s[4]): same kernel as one output phase of the
 * transposed conv, weights packed by licos_pack_conv3x3_w_f16 ([Cout][Cin][3][3] in). */
int licos_pack_conv3x3_w_f16(const float *w, int Cin, int Cout, void *packed, void *stream);
int licos_conv3x3s1_f16(const void *x_blk16, const void *w_packed, const float *bias, const void *gdn_packed, int epilogue,
                        void *y_blk16, float *y_nchw, int B, int Cin, int H, int W, int Cout, void *stream);
int licos_blk16_to_nchw_f32(const void *x_blk16, float *y, int B, int C, int H, int W, void *stream);

/* As licos_nchw_f32_to_blk16, plus the residual y_lo = fp16((x - float(y_hi)) * 2^lo_shift).  abs_input: bit 0 |x|,
 * bit 1 (x / 16)^2 (the GDN norm operand: gamma is then passed multiplied by 256). */
int licos_nchw_f32_split_blk16(const float *x, void *y_hi_blk16, void *y_lo_blk16, int B, int C, int H, int W, int abs_input,
                               int lo_shift, void *stream);

/* The split operand of the ONE-launch fp32 convolution: NCHW fp32 -> blk16 fp16 with 3 C channels (zero-padded to a
 * multiple of 16), channel part * C + c =  hi * 2^-5  |  (x - hi) * 2^6  |  hi = fp16(x)   for part 0 | 1 | 2.
 * Met along cin by the weights  [ (w - w_hi) * 2^5 | w_hi * 2^-6 | w_hi ]  (split on the caller's side, packed by
 * licos_pack_conv_w_f16 / licos_pack_conv3x3_w_f16 with Cin = 3 C), the convolution's K loop sums
 * hi.(w - w_hi) + (x - hi).w_hi + hi.w_hi in ONE fp32 accumulator (the small cross terms first: the running sum is
 * rounded at every step, and early roundings of a small sum cost nothing): an fp32 convolution to ~1e-6 (the lo.lo
 * term, 2^-22, is dropped) in one launch and one store - the form of torch's conv2d / conv_transpose2d on the fp32
 * parity path (licos/train.py:190, eval_utils.py:200).  The power-of-two factors keep every part inside fp16's normal range for
 * |x| in 2e-3 .. 6e4 and |w| above 4e-3; below, a part loses at most 3e-8 absolute.  abs_input: bit 0 |x|. */
int licos_nchw_f32_split3_blk16(const float *x, void *y_blk16, int B, int C, int H, int W, int abs_input, void *stream);

/* First analysis stage for few input channels (Cin <= 4: RGB, single Sentinel-2 band): a 5x5 stride-2 conv over
 * Cin channels equals a 3x3 stride-1 conv over the 4*Cin channels of the 2x2 space-to-depth image (channel
 * c*4 + (y&1)*2 + (x&1) at half resolution).  That turns K = 25 taps x 16 padded channels into 9 x 16 and the
 * strided halo patch into a one-pixel halo.  H, W (even) are the ORIGINAL image size. */
int licos_nchw_f32_to_s2d_blk16(const float *x, void *y_blk16, int B, int C, int H, int W, void *stream);
int licos_pack_conv_w_s2d_f16(const float *w /*[Cout][Cin][5][5]*/, int Cin, int Cout, void *packed, void *stream);
int licos_conv5x5s2_s2d_f16(const void *x_s2d_blk16, const void *w_packed_s2d, const float *bias,
                            const void *gdn_packed, int epilogue, void *y_blk16, float *y_nchw, int B, int Cin,
                            int H, int W, int Cout, void *stream);

/* First analysis stage for 1..3 input bands and <= 128 output channels (the fp16 path's default for RGB / single-band
 * tiles): K steps = kernel rows (ky; kx, c) over an interleaved, zero-bordered fp16 image
 * [B][H + 4][round_up((W + 4) C + 8, 8)] that licos_nchw_f32_to_hwc_pad_f16 writes (value (c, iy, ix) at half
 * (iy + 2) * row + (ix + 2) * C + c); two independent 4-wave workgroups per CU; output blk16 fp16, epilogue
 * LICOS_EPI_NONE / _GDN / _RELU.  CompressAI FactorizedPrior.g_a[0] (+ GDN g_a[1]) as licos/model_utils.py:31-37 re-sizes it. */
size_t licos_hwc_pad_f16_bytes(int B, int C, int H, int W);
int licos_nchw_f32_to_hwc_pad_f16(const float *x_nchw, void *x_hwc_pad, int B, int C, int H, int W, void *stream);
size_t licos_packed_conv_w_first_bytes(int Cin, int Cout);
int licos_pack_conv_w_first_f16(const float *w /*[Cout][Cin][5][5]*/, int Cin, int Cout, void *packed, void *stream);
int licos_conv5x5s2_first_f16(const void *x_hwc_pad, const void *w_packed_first, const float *bias, const void *gdn_packed,
                              int epilogue, void *y_blk16, int B, int Cin, int H, int W, int Cout, void *stream);
/* The same stage on the NCHW fp32 image IN PLACE (W a multiple of 4): the fp32 rows arrive by LDS-DMA, zero padding is a
 * per-granule source choice, the workgroup interleaves and converts them LDS to LDS - no layout pass, bit-identical output. */
int licos_conv5x5s2_first_nchw_f16(const float *x_nchw, const void *w_packed_first, const float *bias, const void *gdn_packed,
                                   int epilogue, void *y_blk16, int B, int Cin, int H, int W, int Cout, void *stream);
/* The first analysis stage for 4 < Cin <= 16 bands (the 13 merged Sentinel-2 bands: /root/reference/licos/raw_image_folder.py:168-174,
 * model_utils.py:31-37 re-sizes g_a[0] to them) on the NCHW fp32 image IN PLACE (W a multiple of 4; 33 .. 128 output
 * channels): replaces licos_nchw_f32_to_blk16 + licos_conv5x5s2_f16 for these models, bit-identical output.  The
 * workgroup fills the K loop's even / odd input-row planes itself - 16-byte loads of 4 pixels per band, converted and
 * written as 4-channel pieces of the planes' granules in the steps in which the K loop does not read the plane - and
 * walks a run of tiles with gamma / beta / bias resident.  w_packed: licos_pack_conv_w_f16 of the [Cout][Cin][5][5] weight. */
int licos_conv5x5s2_first16_nchw_f16(const float *x_nchw, const void *w_packed, const float *bias, const void *gdn_packed,
                                     int epilogue, void *y_blk16, int B, int Cin, int H, int W, int Cout, void *stream);

/* Last synthesis stage (Cout <= 32, NCHW fp32 out): all four output phases per workgroup, weights stored compact
 * (only ceil-pow2(Cout) rows per fragment).  CompressAI FactorizedPrior.g_s[6], replaced per licos/model_utils.py:38-45. */
size_t licos_packed_deconv_w_fewch_bytes(int Cin, int Cout);
int licos_pack_deconv_w_fewch_f16(const float *w /*[Cin][Cout][5][5]*/, int Cin, int Cout, void *packed, void *stream);
int licos_deconv5x5s2_fewch_f16(const void *x_blk16, const void *w_packed_fewch, const float *bias, float *y_nchw,
                                int clamp01, int B, int Cin, int H, int W, int Cout, void *stream);

/* The same stage for 1..4 output channels (RGB / single-band tiles) in scatter form: one contraction per INPUT pixel
 * over all 25 taps (one 32-row MFMA tile per output channel), each product added to its output pixel
 * out[2*iy + ky - 2][2*ix + kx - 2] in an LDS image of the output tile (2^-20 fixed point, order independent). */
size_t licos_packed_deconv_w_scatter_bytes(int Cin, int Cout);
int licos_pack_deconv_w_scatter_f16(const float *w /*[Cin][Cout][5][5]*/, int Cin, int Cout, void *packed, void *stream);
int licos_deconv5x5s2_scatter_f16(const void *x_blk16, const void *w_packed_scatter, const float *bias, float *y_nchw,
                                  int clamp01 /* bit 0: clamp to [0,1]; bit 1: x is x-split (LICOS_EPI_IN_XSPLIT) */, int B,
                                  int Cin, int H, int W, int Cout, void *stream);

/* The same stage for 1..3 output channels in row-walking form (the fp16 path's default for RGB / single-band tiles):
 * Z[(py, kx, c)][y][x] = sum over (dy, cin) as ONE 32-row MFMA tile per 32 input pixels, B operands = the fragments of input
 * rows y-1, y, y+1 held in registers while a wave walks down its 32-column strip (every input row is read once per
 * row block), the x shift out[.., 2x + px] = sum_kx Z[..kx..][x + (px + 2 - kx) / 2] through LDS between neighbouring lanes.
 * fp32 sums in a fixed order.  CompressAI FactorizedPrior.g_s[6] as licos/model_utils.py:38-45 re-sizes it (3 / 1 bands).
 * Round 5: also 5..16 output channels out of 113..128 input channels (the 13 merged Sentinel-2 bands of
 * licos/raw_image_folder.py:168-174; csrc/mfma_rows16.hip): ten 16-row tiles (py, kx) on v_mfma_f32_16x16x32_f16, a wave
 * walks 32 columns, the x shift by DPP between neighbouring lanes and through LDS only at a wave's edges; same three
 * entry points, same argument meaning (`licos_packed_deconv_w_rows_bytes` returns 0 for a pair it does not serve).  A workgroup
 * is 8 waves x 32 columns above 128 columns, 4 x 32 in two row groups up to 128, 2 x 32 in four up to 64; where most of a
 * strip's waves would have no pixels (below ~48 columns, 65 - 95, 129 - 191) licos_deconv5x5s2_fewch_f16 is faster. */
size_t licos_packed_deconv_w_rows_bytes(int Cin, int Cout);
int licos_pack_deconv_w_rows_f16(const float *w /*[Cin][Cout][5][5]*/, int Cin, int Cout, void *packed, void *stream);
int licos_deconv5x5s2_rows_f16(const void *x_blk16, const void *w_packed_rows, const float *bias, float *y_nchw,
                               int clamp01 /* bit 0: clamp to [0,1]; bit 1: x is x-split (LICOS_EPI_IN_XSPLIT) */, int B,
                               int Cin, int H, int W, int Cout, void *stream);

/* 1x1 convolution on the matrix cores, NCHW fp32 output: the channel product of GDN / IGDN ([CAI] layers/gdn.py:
 * norm = conv2d(x^2, gamma, beta)) and of its backward pass (gamma^T . t), as three split-operand passes (`epilogue` =
 * LICOS_EPI_NONE, then LICOS_EPI_ACCUMULATE | LICOS_EPI_SCALE_DOWN(k)).  w: [Cout][Cin] fp32 row-major.  Instantiated
 * for 128 and 192 channels.  licos_gdn_pointwise_f32 supplies the element-wise halves: mode 0 y = x n^p, mode 1
 * t = dy x p n^(p-1), mode 2 dx = dy n^p + 2 x u (p = -1/2, or +1/2 with `inverse`). */
size_t licos_packed_conv1x1_w_bytes(int Cin, int Cout);
int licos_pack_conv1x1_w_f16(const float *w /*[Cout][Cin]*/, int Cin, int Cout, void *packed, void *stream);
int licos_conv1x1_f16(const void *x_blk16, const void *w_packed, const float *bias, int epilogue, float *y_nchw, int B, int Cin,
                      int H, int W, int Cout, void *stream);
int licos_gdn_pointwise_f32(const float *x, const float *n, const float *dy, const float *u, float *out, long count, int inverse,
                            int mode, void *stream);

/* Weight gradient of a 5x5 stride-2 Conv2d / ConvTranspose2d (licos/train.py:195 loss.backward()) on the matrix cores:
 *   dw[s][c][ky][kx] += 2^-scale_down * sum_{b,y,x} small[b][s][y][x] * large[b][c][2y+ky-2][2x+kx-2]
 * small = the conv's output gradient (s = cout) or the transposed conv's input (s = cin); large = the conv's input
 * or the transposed conv's output gradient.  Both maps are "batch-minor" fp16: [ceil(B/16)][H][W][2][channel][8 images]
 * (licos_nchw_f32_split_bm8: hi = fp16(x), lo = fp16((x - hi) * 2^lo_shift)).  dw: fp32, accumulated (caller zeroes).
 * The small map's rows are cut into licos_wgrad5x5s2_strips(Cs, Cl, Hs) strips whose partial sums go through
 * `scratch` (strips * Cs * Cl * 25 floats) and are added in strip order: no atomics, bit-reproducible. */
int licos_nchw_f32_split_bm8(const float *x, void *y_hi, void *y_lo, int B, int C, int H, int W, int lo_shift, void *stream);
int licos_wgrad5x5s2_strips(int Cs, int Cl, int Hs);
int licos_wgrad5x5s2_f16(const void *small_bm8, const void *large_bm8, float *scratch, float *dw, int Cs, int Cl, int nbc, int Hs,
                         int Ws, int Hl, int Wl, int scale_down, void *stream);

/* --------------------------------------------- federated weight average over RCCL / xGMI
 * Replaces the file-and-lock blend of /root/reference/licos/federation_utils.py:27-85 (and the mpi4py scalars of
 * licos/main.py:96-107 stay with the host framework).  One communicator per process: rank 0 makes the 128-byte id,
 * the host framework hands it to every rank (any channel: torch.distributed's store, MPI, a file), every rank calls
 * licos_comm_init with its device current.  licos_allreduce_weighted, in place on `bucket` (n floats, the LAST one
 * reserved): bucket[:n-1] <- sum_r coef_r * bucket_r[:n-1] / sum_r coef_r, all on `stream`, nothing synchronised. */
int licos_comm_unique_id(void *out128);
int licos_comm_init(void **comm, int nranks, int rank, const void *id128);
int licos_comm_destroy(void *comm);
int licos_allreduce_weighted(float *bucket, long n, float coef, void *comm, void *stream);
/* The same blend on the direct schedule (xGMI is a point-to-point mesh): grouped ncclSend / ncclRecv of
 * per-rank chunks, a fixed-order local reduction, grouped exchange of the reduced chunks - two steps instead of a ring's
 * 2 (N - 1).  n_alloc >= nranks * ceil(n / nranks) elements addressable behind `bucket`; scratch: that many floats. */
int licos_allreduce_weighted_direct(float *bucket, long n, long n_alloc, float coef, void *comm, int nranks, int rank,
                                    float *scratch, void *stream);

#define LICOS_EPI_NONE 0
#define LICOS_EPI_GDN 1
#define LICOS_EPI_IGDN 2
#define LICOS_EPI_RELU 3 /* bmshj2018-factorized-relu: ReLU in place of (I)GDN */
/* OR-ed into `epilogue` (NCHW fp32 output, epilogue NONE or RELU): y_nchw += result, then ReLU / clamp on the sum.
 * Three passes over fp16-split operands (licos_nchw_f32_split_blk16: x = hi + 2^-k lo; weights split the same way on
 * the host side): hi*hi (+bias), hi*lo, lo*hi with fp32 accumulation reproduce an fp32 convolution to ~1e-6 - the fp32
 * parity path of torch's conv2d / conv_transpose2d (licos/train.py:190, eval_utils.py:200) at MFMA speed. */
#define LICOS_EPI_ACCUMULATE 0x100
/* with LICOS_EPI_ACCUMULATE: y_nchw += 2^-k * result, k = 0..63 - undoes the 2^k by which a residual operand was scaled
 * up before its conversion to fp16 (licos_nchw_f32_split_blk16's lo_shift, the weight residual likewise) */
#define LICOS_EPI_SCALE_DOWN(k) (((k) & 63) << 12)
/* OR-ed into `epilogue` with LICOS_EPI_GDN / LICOS_EPI_IGDN (65..128 output channels): the norm at fp32 accuracy -
 * `gdn_packed` is then licos_pack_gdn_f32split's buffer (256 gamma split hi + 2^-11 lo in fp16; squares of the
 * accumulators split the same way in registers; hi.hi + 2^-11 (hi.lo + lo.hi) on the matrix cores, rsqrt / sqrt to
 * 1 ulp).  On a split-operand input (licos_nchw_f32_split3_blk16) this is the fp32 parity path's conv + GDN as ONE kernel:
 * the same values as licos_gdn_f32 applied to the convolution's fp32 result. */
#define LICOS_EPI_NORM32 0x40000
/* OR-ed into `epilogue` (blk16 output, Cout a multiple of 16): `y_blk16` receives 3 Cout channels, the split operand of
 * the NEXT fp32 convolution (licos_nchw_f32_split3_blk16's layout) - built from the fp32 result in registers. */
#define LICOS_EPI_OUT_SPLIT3 0x80000
/* OR-ed into licos_deconv5x5s2_f16's `epilogue`: the blk16 input / output is in the x-split form
 *   [B][C/16][H][2][W/2][16]   (each row as two half rows: its even-x pixels, then its odd-x pixels; W even).
 * One output phase of a transposed convolution writes every other pixel of a row; x-split, that is one contiguous run
 * (whole 1-KiB stores, whole cache lines) instead of 32-byte pieces at a 64-byte stride.  Only some kernels take or
 * produce it: ask licos_deconv5x5s2_f16_layouts() for the stage; the last-stage scatter kernel takes it as input
 * (bit 1 of its `clamp01` argument). */
#define LICOS_EPI_IN_XSPLIT 0x200
#define LICOS_EPI_OUT_XSPLIT 0x400
/* bit mask (LICOS_EPI_IN_XSPLIT | LICOS_EPI_OUT_XSPLIT or 0) of the layout flags licos_deconv5x5s2_f16 accepts for a
 * blk16-output stage of this shape (H, W = input size) */
int licos_deconv5x5s2_f16_layouts(int Cin, int H, int W, int Cout);
/* x: blk16 [B][Cin16/16][H][W][16]; out: blk16 fp16 (y_blk16) or NCHW fp32 (y_nchw), exactly one non-NULL.
 * Cout_real <= Cout_packed: channels beyond Cout_real are not stored.  H, W are the INPUT size. */
int licos_conv5x5s2_f16(const void *x_blk16, const void *w_packed, const float *bias, const void *gdn_packed,
                        int epilogue, void *y_blk16, float *y_nchw, int B, int Cin, int H, int W, int Cout,
                        void *stream);
/* The LAST analysis stage with the entropy bottleneck's quantiser in its epilogue: symbols[b][c][h][w] =
 * (int32) rint(conv(x)[b][c][h][w] + bias[c] - medians[c]), round-half-to-even - CompressAI EntropyBottleneck.compress's
 * `symbols = round(x - medians).int()` on g_a's output, the one call /root/reference/eval_utils.py:199-204 makes
 * (`net.compress`).  The layout is the coder's [stream][position] (NCHW order inside a stream): licos_rans_encode_batch
 * reads it with sym_stride_b = Cout * Ho * Wo, sym_stride_i = 1.  Same accumulators as licos_conv5x5s2_f16 with an NCHW
 * fp32 output followed by licos_eb_quantize: identical symbols, no fp32 latent in memory.  `medians`: [Cout] fp32. */
int licos_conv5x5s2_f16_symbols(const void *x_blk16, const void *w_packed, const float *bias, const float *medians,
                                int32_t *symbols, int B, int Cin, int H, int W, int Cout, void *stream);
int licos_deconv5x5s2_f16(const void *x_blk16, const void *w_packed, const float *bias, const void *gdn_packed,
                          int epilogue, void *y_blk16, float *y_nchw, int clamp01, int B, int Cin, int H, int W,
                          int Cout, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LICOS_HIP_H */
