"""Tensor-level wrappers over the C ABI (include/licos_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute
step is one of the hand-written HIP kernels.  All wrappers refuse CPU tensors -
there is deliberately no CPU fallback in the product.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.LicosError(
                "licos_amd: the HIP path needs tensors on a ROCm device (MI355X); got a CPU tensor. "
                "There is no CPU fallback in this package."
            )
        if not t.is_contiguous():
            raise ValueError("licos_amd: tensor must be contiguous")


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


# Optional device timing of the transform stages (engine._timed, bench.py): while `stage_event_sink` is a list, every
# stage launch is bracketed by HIP events on the stream it is launched on - around the C call itself, so that the output
# allocation in front of it (which may have to free cached blocks, and hipFree waits for the whole device) is outside.
stage_event_sink = None


def _launch(cfn, *args):
    if stage_event_sink is None:
        return cfn(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = cfn(*args)
    e1.record()
    stage_event_sink.append((e0, e1))
    return rc


def _is_pinned_host(t):
    return t is not None and (not t.is_cuda) and t.is_pinned() and t.is_contiguous()


def _f32(t):
    if t.dtype != torch.float32:
        raise ValueError(f"licos_amd: expected float32, got {t.dtype}")
    return t


# ----------------------------------------------------------------------------- weights epoch
# The kernels behind adam_f32 / scale_f32 (and any collective on a FlatState bucket) write parameter memory through raw
# pointers, which torch's per-tensor version counter never sees.  Every cache of packed operands (engine.py, layers.py,
# entropy_models.py) therefore keys on this package-level epoch as well; every raw-pointer writer bumps it.
_weights_epoch = 0


def weights_epoch():
    return _weights_epoch


def touch_weights():
    """Declare that parameter memory may have changed behind torch's back (invalidates all packed-operand caches)."""
    global _weights_epoch
    _weights_epoch += 1


# ----------------------------------------------------------------------------- host-side helpers
def pmf_to_quantized_cdf(pmf, precision=16):
    """CompressAI ``_CXX.pmf_to_quantized_cdf`` equivalent (host, int32 out)."""
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(pmf.size + 1, dtype=np.int32)
    rc = _lib.load().licos_pmf_to_quantized_cdf(pmf.ctypes.data, pmf.size, precision, out.ctypes.data)
    _lib.check(rc, "pmf_to_quantized_cdf")
    return out


def rans_build_enc_table(cdf, cdf_len):
    cdf = np.ascontiguousarray(np.asarray(cdf, dtype=np.int32))
    cdf_len = np.ascontiguousarray(np.asarray(cdf_len, dtype=np.int32))
    rows, stride = cdf.shape
    table = np.zeros((rows, stride, 16), dtype=np.uint8)
    rc = _lib.load().licos_rans_build_enc_table(cdf.ctypes.data, cdf_len.ctypes.data, rows, stride, table.ctypes.data)
    _lib.check(rc, "rans_build_enc_table")
    return table


def query(device=0):
    class Props(ctypes.Structure):
        _fields_ = [("compute_units", ctypes.c_int), ("wavefront_size", ctypes.c_int),
                    ("lds_bytes_per_cu", ctypes.c_int), ("clock_khz", ctypes.c_int),
                    ("hbm_bytes", ctypes.c_size_t), ("arch", ctypes.c_char * 32)]
    p = Props()
    _lib.check(_lib.load().licos_query(device, ctypes.addressof(p)), "query")
    return {"compute_units": p.compute_units, "wavefront_size": p.wavefront_size,
            "lds_bytes_per_cu": p.lds_bytes_per_cu, "clock_khz": p.clock_khz, "hbm_bytes": p.hbm_bytes,
            "arch": p.arch.decode()}


# ----------------------------------------------------------------------------- 32-bit path
# ---------------------------------------------------------------------------------------------------------------
# fp32 on the fp16 matrix cores.  A 5x5 stride-2 (transposed) convolution in fp32 is the throughput kernels run on
# split operands: x = x_hi + x_lo, w = w_hi + w_lo (hi = the fp16 rounding, lo = the residual, ~11 more bits), and
#     y = bias + x_hi*w_hi + x_hi*w_lo + x_lo*w_hi        (fp32 accumulation; x_lo*w_lo ~ 2^-22 is dropped)
# as ONE convolution over 3 Cin channels: activations [x_hi 2^-5 | x_lo 2^6 | x_hi] (licos_nchw_f32_split3_blk16), weights
# [w_lo 2^5 | w_hi 2^-6 | w_hi] along cin - one K loop, one accumulator, one store of the NCHW fp32 result (the earlier
# form, three launches accumulating into the output, re-read and re-wrote it twice: the first analysis stage spent
# 25 ms per 1024 tiles on 43 GB of output traffic).  Against float64 this is as accurate as torch's fp32 convolution
# on the CPU (max error 1e-7..1e-6 of max|y|; the direct fp32 VALU kernels: up to 2e-6) and an order of magnitude faster
# than those kernels.  It serves the fp32 parity path and the training step (forward and dgrad); LICOS_FP32_MFMA=0
# keeps the VALU kernels.
FP32_MFMA = os.environ.get("LICOS_FP32_MFMA", "1") != "0"
X3_SHIFT = 11  # (1x1 products, wgrad) residual parts are stored as fp16((v - hi) * 2^11): full 11 bits instead of fp16 subnormals
_x3_zero_bias = {}


def _x3_ok(cin, cout, relu, kind):
    """Channel counts / epilogues the MFMA kernels are instantiated for."""
    if cin > 320 or cout > 320:
        return False
    if not relu:
        return True
    return 32 < cout <= (192 if kind == "conv" else 320)  # ReLU: 4- and 6-tile kernels; 10 tiles only stride 1 / transposed


def _x3_weights(w, kind):
    """Packed fragments of [(w - w_hi) 2^5 | w_hi 2^-6 | w_hi] along cin.  Not cached: a cache keyed on the weight's
    address and version would serve stale fragments once a freed tensor's address is reused; split + pack are a few
    tiny kernels."""
    wf = w.detach().float()
    hi = wf.half().float()
    cat = torch.cat(((wf - hi) * 32.0, hi * 0.015625, hi), dim=0 if kind == "deconv" else 1)
    if kind == "conv3":
        return pack_conv3x3_w_f16(cat)
    return pack_conv_w_f16(cat, kind == "deconv")


def nchw_f32_split_blk16(x, abs_input=False, square16=False):
    """(hi, lo) blk16 fp16 parts of x (|x| with abs_input; (x/16)^2 with square16 - the GDN norm operand)."""
    _dev(x)
    b, c, h, w = x.shape
    hi = torch.empty((b, (c + 15) // 16, h, w, 16), device=x.device, dtype=torch.float16)
    lo = torch.empty_like(hi)
    _lib.check(_lib.load().licos_nchw_f32_split_blk16(_p(_f32(x)), _p(hi), _p(lo), b, c, h, w, int(bool(abs_input)) | (2 if square16 else 0), X3_SHIFT,
                                                      _stream()), "nchw_f32_split_blk16")
    return hi, lo


def nchw_f32_split3_blk16(x, abs_input=False):
    """blk16 fp16 tensor of 3 C channels [hi 2^-5 | (x - hi) 2^6 | hi] (licos_hip.h)."""
    _dev(x)
    b, c, h, w = x.shape
    y = torch.empty((b, (3 * c + 15) // 16, h, w, 16), device=x.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_nchw_f32_split3_blk16(_p(_f32(x)), _p(y), b, c, h, w, int(bool(abs_input)), _stream()),
               "nchw_f32_split3_blk16")
    return y


class Split3:
    """An activation tensor already in the split-operand form (blk16 fp16, 3 C channels) - what gdn_f32_split3 hands
    the next layer's convolution in place of NCHW fp32."""

    def __init__(self, blk, channels):
        self.blk, self.channels = blk, channels

    @property
    def shape(self):
        b, _, h, w, _ = self.blk.shape
        return (b, self.channels, h, w)


def _conv_x3(x, w, bias, relu, abs_input, kind, gdn=None, split3_out=False):
    """kind: "conv" (5x5 s2), "deconv" (5x5 s2 transposed, output padding 1), "conv3" (3x3 s1).
    gdn = (packed f32split operand, inverse): the layer's (I)GDN runs in the convolution's epilogue (EPI_NORM32);
    split3_out: the result is the next convolution's split operand (Split3) instead of NCHW fp32."""
    cin = x.shape[1]
    cout = w.shape[1] if kind == "deconv" else w.shape[0]
    fn = {"conv": conv5x5s2_f16, "deconv": deconv5x5s2_f16, "conv3": conv3x3s1_f16}[kind]
    if isinstance(x, Split3):
        if abs_input:
            raise ValueError("licos_amd: |x| is taken when the operand is split, not afterwards")
        blk = x.blk
    else:
        blk = nchw_f32_split3_blk16(x.contiguous(), abs_input)
    if kind == "deconv" and cout <= 32 and not relu and gdn is None and not split3_out:
        # the last synthesis stage: all four output phases of a tile from one staged patch (mfma_deconv.hip, few-channel form)
        wf = w.detach().float()
        hi = wf.half().float()
        wp = pack_deconv_w_fewch_f16(torch.cat(((wf - hi) * 32.0, hi * 0.015625, hi), dim=0))
        return deconv5x5s2_fewch_f16(blk, wp, pad_bias(bias, cout, blk.device), 3 * cin, cout)
    epi = EPI_RELU if relu else EPI_NONE
    if gdn is not None:
        epi = (EPI_IGDN if gdn[1] else EPI_GDN) | EPI_NORM32
    if split3_out:
        epi |= EPI_OUT_SPLIT3
    y = fn(blk, _x3_weights(w, kind), pad_bias(bias, cout, blk.device), None if gdn is None else gdn[0], epi, 3 * cin, cout,
           out_nchw=not split3_out)
    return Split3(y, cout) if split3_out else y


def x3_route(cin, cout, k, stride, pad, out_pad=None, relu=False):
    """Whether conv2d_f32 (out_pad None) / deconv2d_f32 sends this layer through the one-launch split-operand form."""
    if not FP32_MFMA:
        return False
    if out_pad is None:
        return ((k, stride, pad) == (5, 2, 2) and _x3_ok(cin, cout, relu, "conv")) or ((k, stride, pad) == (3, 1, 1) and _x3_ok(cin, cout, relu, "conv3"))
    return (k, stride, pad, out_pad) == (5, 2, 2, 1) and _x3_ok(cin, cout, relu, "deconv")


def conv2d_f32(x, w, bias, stride, pad, relu=False, abs_input=False, gdn=None, split3_out=False):
    """gdn / split3_out (see _conv_x3): only for layers on the split-operand route (x3_route) with a norm32_ok() GDN."""
    if isinstance(x, Split3) or gdn is not None or split3_out:
        if not x3_route(x.shape[1], w.shape[0], w.shape[2], stride, pad, None, relu):
            raise ValueError("licos_amd: split operands / fused fp32 GDN need a layer on the split-operand route")
        return _conv_x3(x, w, bias, relu, abs_input, "conv" if w.shape[2] == 5 else "conv3", gdn, split3_out)
    _dev(x, w, bias)
    b, cin, h, wd = x.shape
    cout, cin_w, k, k2 = w.shape
    if cin_w != cin or k != k2:
        raise ValueError(f"conv2d_f32: weight {tuple(w.shape)} does not match input {tuple(x.shape)}")
    if FP32_MFMA and (k, stride, pad) == (5, 2, 2) and _x3_ok(cin, cout, relu, "conv"):
        return _conv_x3(x, w, bias, relu, abs_input, "conv")
    if FP32_MFMA and (k, stride, pad) == (3, 1, 1) and _x3_ok(cin, cout, relu, "conv3"):
        return _conv_x3(x, w, bias, relu, abs_input, "conv3")
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    y = torch.empty((b, cout, ho, wo), device=x.device, dtype=torch.float32)
    rc = _lib.load().licos_conv2d_f32(_p(_f32(x)), _p(_f32(w)), _p(bias), _p(y), b, cin, h, wd, cout, k, stride, pad,
                                      int(bool(relu)) | (2 if abs_input else 0), _stream())
    _lib.check(rc, "conv2d_f32")
    return y


def deconv2d_f32(x, w, bias, stride, pad, out_pad, relu=False, gdn=None, split3_out=False):
    if isinstance(x, Split3) or gdn is not None or split3_out:
        if not x3_route(x.shape[1], w.shape[1], w.shape[2], stride, pad, out_pad, relu):
            raise ValueError("licos_amd: split operands / fused fp32 GDN need a layer on the split-operand route")
        return _conv_x3(x, w, bias, relu, False, "deconv", gdn, split3_out)
    _dev(x, w, bias)
    b, cin, h, wd = x.shape
    cin_w, cout, k, k2 = w.shape
    if cin_w != cin or k != k2:
        raise ValueError(f"deconv2d_f32: weight {tuple(w.shape)} does not match input {tuple(x.shape)}")
    if FP32_MFMA and (k, stride, pad, out_pad) == (5, 2, 2, 1) and _x3_ok(cin, cout, relu, "deconv"):
        return _conv_x3(x, w, bias, relu, False, "deconv")
    ho, wo = (h - 1) * stride - 2 * pad + k + out_pad, (wd - 1) * stride - 2 * pad + k + out_pad
    y = torch.empty((b, cout, ho, wo), device=x.device, dtype=torch.float32)
    rc = _lib.load().licos_deconv2d_f32(_p(_f32(x)), _p(_f32(w)), _p(bias), _p(y), b, cin, h, wd, cout, k, stride,
                                        pad, out_pad, int(relu), _stream())
    _lib.check(rc, "deconv2d_f32")
    return y


def gdn_reparam_f32(beta_raw, gamma_raw, beta_bound, gamma_bound, pedestal):
    _dev(beta_raw, gamma_raw)
    c = beta_raw.numel()
    beta = torch.empty_like(beta_raw)
    gamma = torch.empty_like(gamma_raw)
    rc = _lib.load().licos_gdn_reparam_f32(_p(_f32(beta_raw)), _p(_f32(gamma_raw)), beta_bound, gamma_bound, pedestal,
                                           _p(beta), _p(gamma), c, _stream())
    _lib.check(rc, "gdn_reparam_f32")
    return beta, gamma


GDN_MFMA = os.environ.get("LICOS_GDN_MFMA", "1") != "0"


def _conv1x1_x3(x, w, bias, square16=False):
    """y = bias + w . x over channels (NCHW fp32 in and out) as three split-operand MFMA passes; with square16 the
    operand is (x/16)^2 and w must already carry the factor 256."""
    b, c, h, wd = x.shape
    cout = w.shape[0]
    xh, xl = nchw_f32_split_blk16(x.contiguous(), square16=square16)
    wf = w.detach().float().contiguous()
    hi = wf.half().float()
    lib = _lib.load()
    packs = []
    for part in (hi, (wf - hi) * float(2 ** X3_SHIFT)):
        pk = torch.empty(lib.licos_packed_conv1x1_w_bytes(c, cout) // 2, device=x.device, dtype=torch.float16)
        _lib.check(lib.licos_pack_conv1x1_w_f16(_p(part), c, cout, _p(pk), _stream()), "pack_conv1x1_w_f16")
        packs.append(pk)
    bp = pad_bias(bias, cout, x.device)
    zk = (32 * mfma_tiles(cout), str(x.device))
    zero = _x3_zero_bias.get(zk)
    if zero is None:
        zero = _x3_zero_bias[zk] = torch.zeros(zk[0], device=x.device, dtype=torch.float32)
    y = torch.empty((b, cout, h, wd), device=x.device, dtype=torch.float32)
    down = EPI_ACCUMULATE | (X3_SHIFT << 12)
    for xin, pk, bb, epi in ((xh, packs[0], bp, EPI_NONE), (xh, packs[1], zero, down), (xl, packs[0], zero, down)):
        _lib.check(lib.licos_conv1x1_f16(_p(xin), _p(pk), _p(bb), epi, _p(y), b, c, h, wd, cout, _stream()), "conv1x1_f16")
    return y


def _gdn_pointwise(x, n, dy, u, inverse, mode):
    out = torch.empty_like(x)
    _lib.check(_lib.load().licos_gdn_pointwise_f32(_p(x), _p(n), _p(dy), _p(u), _p(out), x.numel(), int(inverse), mode, _stream()),
               "gdn_pointwise_f32")
    return out


def _gdn_mfma_ok(x):
    return GDN_MFMA and FP32_MFMA and x.dim() == 4 and x.shape[1] in (128, 192)


def gdn_f32_split3_applies(c, hw):
    return FP32_MFMA and bool(_lib.load().licos_gdn_f32_split3_applies(c, hw))


def gdn_f32_split3(x, gamma_eff, beta_eff, inverse=False):
    """GDN / IGDN whose result is the next convolution's split operand (Split3) instead of NCHW fp32."""
    _dev(x, gamma_eff, beta_eff)
    b, c, h, w = x.shape
    y = torch.empty((b, 3 * c // 16, h, w, 16), device=x.device, dtype=torch.float16)
    rc = _lib.load().licos_gdn_f32_split3(_p(_f32(x)), _p(gamma_eff), _p(beta_eff), _p(y), b, c, h * w, int(inverse), _stream())
    _lib.check(rc, "gdn_f32_split3")
    return Split3(y, c)


def gdn_f32_fwd_norm(x, gamma_eff, beta_eff, inverse=False):
    """(y, norm): the forward of a training step - norm = beta + gamma . x^2 is what gdn_bwd_fused_f32 starts from.
    Shapes of gdn_f32_split3_applies only."""
    _dev(x, gamma_eff, beta_eff)
    b, c, h, w = x.shape
    y, n = torch.empty_like(x), torch.empty_like(x)
    rc = _lib.load().licos_gdn_f32_fwd_norm(_p(_f32(x)), _p(gamma_eff), _p(beta_eff), _p(y), _p(n), b, c, h * w, int(inverse), _stream())
    _lib.check(rc, "gdn_f32_fwd_norm")
    return y, n


def gdn_bwd_fused_f32(x, dy, norm, gamma_eff, inverse, want_dgamma=False):
    """(dx, t, dgamma_eff or None) - dx and t in one kernel (licos_hip.h licos_gdn_bwd_fused_f32), which also leaves
    max|t| where the gamma-gradient kernel looks for its operand scale (no extra pass over t)."""
    _dev(x, dy, norm, gamma_eff)
    b, c, h, w = x.shape
    lib = _lib.load()
    dx, t = torch.empty_like(x), torch.empty_like(x)
    scratch = absmax = None
    if want_dgamma:
        parts = lib.licos_gdn_gamma_grad_parts(b, h * w)
        scratch = torch.empty(parts * c * c + 4, device=x.device, dtype=torch.float32)
        absmax = scratch[parts * c * c:]
    xc, dyc = _f32(x), _f32(dy)
    rc = lib.licos_gdn_bwd_fused_f32(_p(xc), _p(dyc), _p(norm), _p(gamma_eff), _p(dx), _p(t), _p(absmax), b, c, h * w,
                                     int(inverse), _stream())
    _lib.check(rc, "gdn_bwd_fused_f32")
    dg = None
    if want_dgamma:
        dg = torch.empty((c, c), device=x.device, dtype=torch.float32)
        rc = lib.licos_gdn_gamma_grad_scaled_f32(_p(t), _p(xc), _p(scratch), _p(dg), b, c, h * w, _stream())
        _lib.check(rc, "gdn_gamma_grad_scaled_f32")
    return dx, t, dg


def gdn_f32(x, gamma_eff, beta_eff, inverse=False):
    _dev(x, gamma_eff, beta_eff)
    b, c = x.shape[:2]
    hw = x[0, 0].numel()
    # (128 channels over whole 32-pixel tiles run as ONE pass on the matrix cores, mfma_gdn_f32.hip; other shapes on the
    # vector-ALU kernel)
    y = torch.empty_like(x)
    rc = _lib.load().licos_gdn_f32(_p(_f32(x)), _p(gamma_eff), _p(beta_eff), _p(y), b, c, hw, int(inverse), _stream())
    _lib.check(rc, "gdn_f32")
    return y


# ----------------------------------------------------------------------------- backward (32-bit path)
def conv2d_wgrad_f32(inp, g, ci, co, k, stride, pad, square_input=False):
    """dw [co][ci][k][k] = sum g[b][co][oy][ox] * inp[b][ci][oy*s-p+ky][ox*s-p+kx]."""
    _dev(inp, g)
    b, _, h, w = inp.shape
    if WGRAD_MFMA and FP32_MFMA and (k, stride, pad) == (5, 2, 2) and not square_input:
        return _wgrad5x5s2_x3(inp, g, ci, co)
    if GDN_MFMA and square_input and (k, stride, pad) == (1, 1, 0) and ci == co == 128 and (h * w) % 4 == 0:
        # the GDN gamma gradient: a 128 x 128 product over all pixels of the batch, on the matrix cores
        lib = _lib.load()
        # partial matrices + one word for the kernel's max|t| pre-pass (licos_hip.h)
        scratch = torch.empty(lib.licos_gdn_gamma_grad_parts(b, h * w) * 128 * 128 + 4, device=inp.device, dtype=torch.float32)
        dw = torch.empty((co, ci, 1, 1), device=inp.device, dtype=torch.float32)
        rc = lib.licos_gdn_gamma_grad_f32(_p(_f32(g.contiguous())), _p(_f32(inp.contiguous())), _p(scratch), _p(dw), b, 128, h * w, _stream())
        _lib.check(rc, "gdn_gamma_grad_f32")
        return dw
    dw = torch.empty((co, ci, k, k), device=inp.device, dtype=torch.float32)
    rc = _lib.load().licos_conv2d_wgrad_f32(_p(_f32(inp)), _p(_f32(g)), _p(dw), b, ci, h, w, co, k, stride, pad,
                                            int(square_input), _stream())
    _lib.check(rc, "conv2d_wgrad_f32")
    return dw


WGRAD_MFMA = os.environ.get("LICOS_WGRAD_MFMA", "1") != "0"


def _split_bm8(x):
    """NCHW fp32 -> batch-minor fp16 pair [ceil(B/16)][H][W][2][C][8] (hi, 2^11-scaled residual)."""
    b, c, h, w = x.shape
    shape = ((b + 15) // 16, h, w, 2, c, 8)
    hi = torch.empty(shape, device=x.device, dtype=torch.float16)
    lo = torch.empty(shape, device=x.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_nchw_f32_split_bm8(_p(_f32(x)), _p(hi), _p(lo), b, c, h, w, X3_SHIFT, _stream()),
               "nchw_f32_split_bm8")
    return hi, lo


def _wgrad5x5s2_x3(inp, g, ci, co):
    """The same weight gradient on the matrix cores: both maps in batch-minor fp16 (the 16 images of a pixel = one
    MFMA K step), split hi / 2^-11 lo, three passes, per-strip partial sums added in a fixed order."""
    b, _, hl, wl = inp.shape
    hs, ws = g.shape[2:]
    nbc = (b + 15) // 16
    l_hi, l_lo = _split_bm8(inp.detach().contiguous())
    s_hi, s_lo = _split_bm8(g.detach().contiguous())
    dw = torch.zeros((co, ci, 5, 5), device=inp.device, dtype=torch.float32)
    lib = _lib.load()
    scratch = torch.empty(lib.licos_wgrad5x5s2_strips(co, ci, hs) * co * ci * 25, device=inp.device, dtype=torch.float32)
    for small, large, down in ((s_hi, l_hi, 0), (s_hi, l_lo, X3_SHIFT), (s_lo, l_hi, X3_SHIFT)):
        rc = lib.licos_wgrad5x5s2_f16(_p(small), _p(large), _p(scratch), _p(dw), co, ci, nbc, hs, ws, hl, wl, down, _stream())
        _lib.check(rc, "wgrad5x5s2_f16")
    return dw


def bias_grad_f32(dy):
    _dev(dy)
    b, c = dy.shape[:2]
    db = torch.empty(c, device=dy.device, dtype=torch.float32)
    _lib.check(_lib.load().licos_bias_grad_f32(_p(_f32(dy)), _p(db), b, c, dy[0, 0].numel(), _stream()), "bias_grad_f32")
    return db


def gdn_bwd_f32(x, dy, gamma_eff, beta_eff, inverse):
    _dev(x, dy, gamma_eff, beta_eff)
    b, c = x.shape[:2]
    hw = x[0, 0].numel()
    if _gdn_mfma_ok(x):
        x, dy = _f32(x).contiguous(), _f32(dy).contiguous()
        n = _conv1x1_x3(x, gamma_eff * 256.0, beta_eff, square16=True)
        t = _gdn_pointwise(x, n, dy, None, inverse, 1)                       # dL/dn
        u = _conv1x1_x3(t, gamma_eff.t().contiguous(), None)                 # gamma^T . t
        return _gdn_pointwise(x, n, dy, u, inverse, 2), t
    dx, t = torch.empty_like(x), torch.empty_like(x)
    scratch = torch.empty_like(gamma_eff)
    rc = _lib.load().licos_gdn_bwd_f32(_p(_f32(x)), _p(_f32(dy)), _p(gamma_eff), _p(beta_eff), _p(scratch), _p(dx), _p(t),
                                       b, c, hw, int(inverse), _stream())
    _lib.check(rc, "gdn_bwd_f32")
    return dx, t


def reparam_bwd_f32(raw, d_eff, bound):
    _dev(raw, d_eff)
    out = torch.empty_like(raw)
    rc = _lib.load().licos_reparam_bwd_f32(_p(_f32(raw)), _p(_f32(d_eff.contiguous())), bound, _p(out), raw.numel(), _stream())
    _lib.check(rc, "reparam_bwd_f32")
    return out


def adam_f32(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
    _dev(p, g, m, v)
    rc = _lib.load().licos_adam_f32(_p(_f32(p)), _p(_f32(g)), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step,
                                    grad_scale, _stream())
    _lib.check(rc, "adam_f32")
    touch_weights()


def sumsq_f32(x, out):
    _dev(x, out)
    _lib.check(_lib.load().licos_sumsq_f32(_p(_f32(x)), x.numel(), _p(out), _stream()), "sumsq_f32")


# ----------------------------------------------------------------------------- entropy bottleneck
def _filters_arr(filters):
    return (ctypes.c_int * len(filters))(*[int(f) for f in filters])


def eb_packed_size(filters):
    arr = _filters_arr(filters)
    return _lib.check(_lib.load().licos_eb_packed_size(ctypes.cast(arr, ctypes.c_void_p), len(filters)), "eb_packed_size")


def eb_pack(matrices, biases, factors, filters, channels):
    _dev(*matrices, *biases, *factors)
    n = len(filters) + 1
    per = eb_packed_size(filters)
    packed = torch.empty((channels, per), device=matrices[0].device, dtype=torch.float32)
    PA = ctypes.c_void_p * n
    ma = PA(*[m.data_ptr() for m in matrices])
    ba = PA(*[b.data_ptr() for b in biases])
    fa = PA(*([f.data_ptr() for f in factors] + [0]))
    arr = _filters_arr(filters)
    rc = _lib.load().licos_eb_pack(ctypes.cast(ma, ctypes.c_void_p), ctypes.cast(ba, ctypes.c_void_p),
                                   ctypes.cast(fa, ctypes.c_void_p), ctypes.cast(arr, ctypes.c_void_p), len(filters),
                                   channels, _p(packed), _stream())
    _lib.check(rc, "eb_pack")
    return packed


def _p_off(t, elem_off):
    return ctypes.c_void_p(t.data_ptr() + elem_off * t.element_size())


def eb_quantize(y, medians, mode, noise=None, symbols=None, sym_stride_b=0, sym_stride_i=1, want_y_hat=True,
                sym_offset=0):
    """mode: 'dequantize' | 'noise' | 'symbols'.  y: (B, C, *spatial) fp32.  `sym_offset`: element offset of
    this batch's first stream inside a larger interleaved symbol buffer."""
    # (`symbols` may be a page-locked host tensor: the kernel then stores across PCIe itself - hipHostMalloc memory is
    # mapped into the device's address space - instead of a device buffer plus a copy engine transfer, which runs at
    # 15 GB/s for the 10 - 50 MB of a small batch: tools/split_probe.py)
    _dev(y, medians, noise, None if (symbols is not None and _is_pinned_host(symbols)) else symbols)
    b, c = y.shape[:2]
    hw = y[0, 0].numel()
    m = {"dequantize": 0, "noise": 1, "symbols": 2}[mode]
    y_hat = torch.empty_like(y) if (want_y_hat and m != 2) else None
    sp = _p(symbols) if symbols is None else _p_off(symbols, sym_offset)
    rc = _lib.load().licos_eb_quantize(_p(_f32(y)), _p(_f32(medians)), _p(noise), _p(y_hat), sp,
                                       sym_stride_b, sym_stride_i, m, b, c, hw, _stream())
    _lib.check(rc, "eb_quantize")
    return y_hat


def eb_likelihood(v, packed, filters, bound, form=0, sum_log2=None):
    _dev(v, packed, sum_log2)
    b, c = v.shape[:2]
    hw = v[0, 0].numel()
    lik = torch.empty_like(v)
    arr = _filters_arr(filters)
    rc = _lib.load().licos_eb_likelihood(_p(_f32(v)), _p(packed), ctypes.cast(arr, ctypes.c_void_p), len(filters),
                                         _p(lik), bound, form, _p(sum_log2), b, c, hw, _stream())
    _lib.check(rc, "eb_likelihood")
    return lik


def eb_likelihood_bwd(v, g_lik, packed, filters, bound, form=0):
    """(dL/dv, dL/dpacked-record [C][per_channel] w.r.t. the raw parameters) of eb_likelihood."""
    g_lik = g_lik.contiguous()  # autograd may hand over an expanded (stride-0) gradient, e.g. for loss = lik.sum()
    _dev(v, g_lik, packed)
    b, c = v.shape[:2]
    hw = v[0, 0].numel()
    dv = torch.empty_like(v)
    ns = int(_lib.load().licos_eb_likelihood_bwd_slices(b, hw))
    dps = torch.empty((ns, c, packed.shape[1]), device=v.device, dtype=torch.float32)
    arr = _filters_arr(filters)
    rc = _lib.load().licos_eb_likelihood_bwd(_p(_f32(v)), _p(_f32(g_lik.contiguous())), _p(packed), ctypes.cast(arr, ctypes.c_void_p),
                                             len(filters), bound, form, _p(dv), _p(dps), b, c, hw, _stream())
    _lib.check(rc, "eb_likelihood_bwd")
    return dv, (dps[0] if ns == 1 else dps.sum(0))


def gc_likelihood_bwd(v, scales, g_lik, scale_bound, lik_bound):
    g_lik = g_lik.contiguous()
    _dev(v, scales, g_lik)
    dv, ds = torch.empty_like(v), torch.empty_like(v)
    rc = _lib.load().licos_gc_likelihood_bwd(_p(_f32(v)), _p(_f32(scales)), _p(_f32(g_lik.contiguous())), scale_bound, lik_bound,
                                             _p(dv), _p(ds), v.numel(), _stream())
    _lib.check(rc, "gc_likelihood_bwd")
    return dv, ds


def mask_mul_f32(g, ref, mode):
    """g * (ref > 0) for mode "relu", g * sign(ref) for mode "abs"."""
    g = g.contiguous()
    _dev(g, ref)
    out = torch.empty_like(g)
    rc = _lib.load().licos_mask_mul_f32(_p(_f32(g)), _p(_f32(ref)), _p(out), g.numel(), {"relu": 0, "abs": 1}[mode], _stream())
    _lib.check(rc, "mask_mul_f32")
    return out


def eb_dequantize(symbols, sym_stride_b, sym_stride_i, medians, b, c, h, w, want_nchw=True, blk16=None, sym_offset=0):
    # (`symbols` may be a page-locked host tensor, read across PCIe by the kernel itself: see eb_quantize)
    _dev(None if _is_pinned_host(symbols) else symbols, medians, blk16)
    y = torch.empty((b, c, h, w), device=medians.device, dtype=torch.float32) if want_nchw else None
    rc = _lib.load().licos_eb_dequantize(_p_off(symbols, sym_offset), sym_stride_b, sym_stride_i, _p(medians), _p(y), _p(blk16),
                                         b, c, h, w, _stream())
    _lib.check(rc, "eb_dequantize")
    return y


def eb_symbols16(y, medians, symbols16, flag):
    """symbols16 int16 [B][n] (device) = round(y - median[channel]); flag int32 [1] (device) |= 1 on a symbol outside
    int16 - what the host coder's 16-bit form reads (rans_encode_host_sym16)."""
    _dev(y, medians, symbols16, flag)
    b, c = y.shape[:2]
    hw = y[0, 0].numel()
    rc = _lib.load().licos_eb_symbols16(_p(_f32(y)), _p(_f32(medians)), _p(symbols16), _p(flag), b, c, hw, _stream())
    _lib.check(rc, "eb_symbols16")
    return symbols16


def eb_dequantize16(symbols16, medians, b, c, h, w, want_nchw=True, blk16=None):
    """y_hat of int16 symbols [B][n] (device): NCHW fp32 (returned) and / or fp16 blk16 (written into `blk16`)."""
    _dev(symbols16, medians, blk16)
    y = torch.empty((b, c, h, w), device=medians.device, dtype=torch.float32) if want_nchw else None
    rc = _lib.load().licos_eb_dequantize16(_p(symbols16), _p(medians), _p(y), _p(blk16), b, c, h, w, _stream())
    _lib.check(rc, "eb_dequantize16")
    return y


def reduce_sqdiff(a, b, clamp01=False):
    _dev(a, b)
    out = torch.zeros(1, device=a.device, dtype=torch.float64)
    rc = _lib.load().licos_reduce_sqdiff(_p(_f32(a)), _p(_f32(b)), a.numel(), int(clamp01), _p(out), _stream())
    _lib.check(rc, "reduce_sqdiff")
    return out


def gc_likelihood(v, scales, scale_bound, lik_bound, sum_log2=None):
    _dev(v, scales, sum_log2)
    b, c = v.shape[:2]
    hw = v[0, 0].numel()
    lik = torch.empty_like(v)
    rc = _lib.load().licos_gc_likelihood(_p(_f32(v)), _p(_f32(scales)), _p(lik), scale_bound, lik_bound, _p(sum_log2),
                                         b, c, hw, _stream())
    _lib.check(rc, "gc_likelihood")
    return lik


def gc_build_indexes(scales, table, scale_bound, indexes, stride_b, stride_i):
    _dev(scales, table, indexes)
    b = scales.shape[0]
    n = scales[0].numel()
    rc = _lib.load().licos_gc_build_indexes(_p(_f32(scales)), _p(_f32(table)), table.numel(), scale_bound, _p(indexes),
                                            stride_b, stride_i, b, n, _stream())
    _lib.check(rc, "gc_build_indexes")
    return indexes


def gc_pack_symbols(y, scales, table, scale_bound, packed, flag):
    """packed int32 [B][n] (device) = table row << 16 | (round(y) & 0xFFFF); flag int32 [1] (device) |= 1 on a symbol
    outside int16 - the host coder's compact input (rans_encode_host_packed)."""
    _dev(y, scales, table, packed, flag)
    b = scales.shape[0]
    n = scales[0].numel()
    rc = _lib.load().licos_gc_pack_symbols(_p(_f32(y)), _p(_f32(scales)), _p(_f32(table)), table.numel(), scale_bound, _p(packed),
                                           _p(flag), b, n, _stream())
    _lib.check(rc, "gc_pack_symbols")
    return packed


def gc_build_rows8(scales, table, scale_bound, rows8):
    """rows8 uint8 [B][n] (device): the table row of every symbol, for rans_decode_host_rows8."""
    _dev(scales, table, rows8)
    b = scales.shape[0]
    n = scales[0].numel()
    rc = _lib.load().licos_gc_build_rows8(_p(_f32(scales)), _p(_f32(table)), table.numel(), scale_bound, _p(rows8), b, n, _stream())
    _lib.check(rc, "gc_build_rows8")
    return rows8


def scale_f32(x, alpha, inv_alpha_dev=None):
    """In place x *= alpha (or alpha / inv_alpha_dev[0])."""
    _dev(x, inv_alpha_dev)
    _lib.check(_lib.load().licos_scale_f32(_p(_f32(x)), x.numel(), float(alpha), _p(inv_alpha_dev), _stream()), "scale_f32")
    touch_weights()
    return x


# ----------------------------------------------------------------------------- rANS
def rans_encode_batch(symbols, sym_stride_b, sym_stride_i, n, plane, cdf, cdf_len, offset, enc_table, cap_words,
                      batch, indexes=None, sym_offset=0):
    """Returns (words scratch [cap_words, B] u32, nwords [B] i32, status [1] i32) on the device."""
    _dev(symbols, cdf, cdf_len, offset, enc_table, indexes)
    dev = symbols.device
    words = torch.empty((cap_words, batch), device=dev, dtype=torch.int32)
    nwords = torch.empty(batch, device=dev, dtype=torch.int32)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    ip = _p(indexes) if indexes is None else _p_off(indexes, sym_offset)
    rc = _lib.load().licos_rans_encode_batch(_p_off(symbols, sym_offset), ip, sym_stride_b, sym_stride_i, n, plane, _p(cdf),
                                             cdf.shape[1], _p(cdf_len), _p(offset), _p(enc_table), _p(words),
                                             cap_words, _p(nwords), _p(status), batch, _stream())
    _lib.check(rc, "rans_encode_batch")
    return words, nwords, status


def rans_compact(words, nwords, byte_off, total_bytes, out=None, off_offset=0):
    """`byte_off` holds absolute byte offsets into `out`; `off_offset` selects this batch's first entry."""
    _dev(words, nwords, byte_off, out)
    cap, batch = words.shape
    if out is None:
        out = torch.empty(max(int(total_bytes), 4), device=words.device, dtype=torch.uint8)
    rc = _lib.load().licos_rans_compact(_p(words), cap, _p(nwords), _p_off(byte_off, off_offset), _p(out), batch, _stream())
    _lib.check(rc, "rans_compact")
    return out


def rans_decode_batch(data, byte_off, sym_stride_b, sym_stride_i, n, plane, cdf, cdf_len, offset, symbols, batch,
                      indexes=None, sym_offset=0, status=None, off_offset=None):
    _dev(data, byte_off, cdf, cdf_len, offset, symbols, indexes)
    if status is None:
        status = torch.zeros(1, device=data.device, dtype=torch.int32)
    ip = _p(indexes) if indexes is None else _p_off(indexes, sym_offset)
    if off_offset is None:
        off_offset = sym_offset
    rc = _lib.load().licos_rans_decode_batch(_p(data), _p_off(byte_off, off_offset), ip, sym_stride_b, sym_stride_i, n, plane,
                                             _p(cdf), cdf.shape[1], _p(cdf_len), _p(offset), _p_off(symbols, sym_offset), _p(status),
                                             batch, _stream())
    _lib.check(rc, "rans_decode_batch")
    return status


def rans_image_budget(waves=2):
    """Bytes of LDS a decoder image may take beside the decode kernel's rings (licos_rans_decode_image)."""
    return int(_lib.load().licos_rans_image_budget(int(waves)))


def rans_image_build(cdf, cdf_len, offset, budget_bytes=None, row_weight=None):
    """Decoder image (numpy uint8 blob) of an integer CDF table: per-row bucket records + 16-bit symbol starts
    (licos_amd/csrc/rans_image.hpp).  Host-side, once per table."""
    cdf_a, cp = _np_i32(cdf)
    len_a, lp = _np_i32(cdf_len)
    off_a, op = _np_i32(offset)
    budget = rans_image_budget(2) if budget_bytes is None else int(budget_bytes)
    out = np.zeros(budget, dtype=np.uint8)
    used = ctypes.c_long(0)
    wp = ctypes.c_void_p(0)
    if row_weight is not None:
        w = np.ascontiguousarray(row_weight, dtype=np.float32)
        wp = ctypes.c_void_p(w.ctypes.data)
    rc = _lib.load().licos_rans_image_build(cp, lp, op, cdf_a.shape[0], cdf_a.shape[1], wp, budget,
                                            ctypes.c_void_p(out.ctypes.data), ctypes.addressof(used))
    _lib.check(rc, "rans_image_build")
    return out[: used.value].copy()


def rans_image_lookup(image, row, cf):
    """(symbol, lo, hi, slow_path) of value cf in `row` through the image - the decode kernel's own search, on the host."""
    out = (ctypes.c_int32 * 3)()
    rc = _lib.load().licos_rans_image_lookup(ctypes.c_void_p(image.ctypes.data), int(row), int(cf), ctypes.addressof(out))
    _lib.check(rc, "rans_image_lookup")
    return out[0], out[1], out[2], bool(rc)


def gc_encode_prepare(y, scales, scale_table, scale_bound, enc_table, cdf_len, offset, cdf_stride):
    """(rec [n][B] 16-byte records as int32 [n][B][4], aux int32 [n][B]) for licos_rans_encode_records."""
    _dev(y, scales, scale_table, enc_table, cdf_len, offset)
    b = y.shape[0]
    n = y[0].numel()
    rec = torch.empty((n, b, 4), device=y.device, dtype=torch.int32)
    aux = torch.empty((n, b), device=y.device, dtype=torch.int32)
    rc = _lib.load().licos_gc_encode_prepare(_p(_f32(y)), _p(_f32(scales)), _p(_f32(scale_table)), scale_table.numel(), scale_bound,
                                             _p(enc_table), cdf_stride, _p(cdf_len), _p(offset), _p(rec), _p(aux), b, n, _stream())
    _lib.check(rc, "gc_encode_prepare")
    return rec, aux


def rans_encode_records(rec, aux, cap_words):
    """The serial encoder over prepared records: (words [cap_words][B], nwords [B], status [1])."""
    _dev(rec, aux)
    n, b = aux.shape
    words = torch.empty((cap_words + 1, b), device=aux.device, dtype=torch.int32)  # row 0: the kernel's dump row
    nwords = torch.empty(b, device=aux.device, dtype=torch.int32)
    status = torch.zeros(1, device=aux.device, dtype=torch.int32)
    rc = _lib.load().licos_rans_encode_records(_p(rec), _p(aux), n, _p(words), cap_words, _p(nwords), _p(status), b, _stream())
    _lib.check(rc, "rans_encode_records")
    return words[1:], nwords, status


def gc_decode_prepare(scales, scale_table, scale_bound, row_hist=None):
    """Row byte per symbol in the decoder's granule layout: uint8 [ceil(n/16)][B][16].  `row_hist` (int32 [256] on the
    device, optional) accumulates a sampled histogram of the rows in use."""
    _dev(scales, scale_table, row_hist)
    b = scales.shape[0]
    n = scales[0].numel()
    idx16 = torch.empty(((n + 15) // 16, b, 16), device=scales.device, dtype=torch.uint8)
    rc = _lib.load().licos_gc_decode_prepare(_p(_f32(scales)), _p(_f32(scale_table)), scale_table.numel(), scale_bound, _p(idx16),
                                             _p(row_hist), b, n, _stream())
    _lib.check(rc, "gc_decode_prepare")
    return idx16


def eb_encode_prepare(y, medians, enc_table, cdf_len, offset, cdf_stride):
    """Entropy-bottleneck latents y (B, C, *spatial) -> (rec [n][B][4] int32, aux [n][B]) for rans_encode_records."""
    _dev(y, medians, enc_table, cdf_len, offset)
    b, c = y.shape[:2]
    plane = y[0, 0].numel()
    n = c * plane
    rec = torch.empty((n, b, 4), device=y.device, dtype=torch.int32)
    aux = torch.empty((n, b), device=y.device, dtype=torch.int32)
    rc = _lib.load().licos_eb_encode_prepare(_p(_f32(y)), _p(_f32(medians)), c, plane, _p(enc_table), cdf_stride, _p(cdf_len), _p(offset),
                                             _p(rec), _p(aux), b, _stream())
    _lib.check(rc, "eb_encode_prepare")
    return rec, aux


def rans_decode_image(data, byte_off, idx16, n, image_dev, image_host, symbols, sym_stride_b, sym_stride_i, batch, status=None,
                      sym_offset=0, rows_shared=False):
    _dev(data, byte_off, idx16, image_dev, symbols)
    if status is None:
        status = torch.zeros(1, device=data.device, dtype=torch.int32)
    rc = _lib.load().licos_rans_decode_image(_p(data), _p(byte_off), _p(idx16), int(bool(rows_shared)), n, _p(image_dev),
                                             ctypes.c_void_p(image_host.ctypes.data), _p_off(symbols, sym_offset), sym_stride_b,
                                             sym_stride_i, _p(status), batch, _stream())
    _lib.check(rc, "rans_decode_image")
    return status


def host_threads():
    """Worker threads for the host coder: this process's share of the cores it may run on.  With one process per GPU
    the ranks of a node usually share one affinity mask (torchrun does not pin): the mask is divided by the node-local
    world size (LOCAL_WORLD_SIZE from the launcher, else WORLD_SIZE), so 8 ranks on a 128-core host start 16 threads
    each, not 8 x 16 on top of each other's cores.  Capped at 16: beyond that the per-call thread start-up outweighs
    the work of the batches the host coder is chosen for.  LICOS_HOST_THREADS overrides."""
    forced = os.environ.get("LICOS_HOST_THREADS")
    if forced:
        return max(1, int(forced))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1)
    except ValueError:
        local_world = 1
    return max(1, min(16, n // max(1, local_world)))


# Coder placement (licos_amd/entropy_models.py): a GPU lane spends ~160 ns (encode) / ~220 ns (decode) per symbol of
# its stream whatever the batch, a host core ~5 / ~10 ns; with T host threads the host wins below ~32 T streams.
# LICOS_HOST_CODER = "0" never, "1" always, otherwise automatic with this many streams per host thread as the limit.
HOST_CODER = os.environ.get("LICOS_HOST_CODER", "auto")
HOST_CODER_STREAMS_PER_THREAD = 24


def host_coder_preferred(batch):
    if HOST_CODER == "0":
        return False
    if HOST_CODER == "1":
        return True
    return batch <= HOST_CODER_STREAMS_PER_THREAD * host_threads()


def _np_i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, ctypes.c_void_p(a.ctypes.data)


def rans_encode_host(symbols, n, plane, cdf, cdf_len, offset, enc_table, indexes=None, nthreads=None):
    """Host coder.  symbols: int32 numpy [B][n] (one contiguous stream per row); tables: numpy (host copies).
    Returns (out uint8 [B][cap], nbytes int64 [B])."""
    sym, sp = _np_i32(symbols)
    b = sym.shape[0]
    cdf_a, cp = _np_i32(cdf)
    len_a, lp = _np_i32(cdf_len)
    off_a, op = _np_i32(offset)
    ip = ctypes.c_void_p(0)
    if indexes is not None:
        idx, ip = _np_i32(indexes)
    nthreads = host_threads() if nthreads is None else int(nthreads)
    cap = 4 * (n // 2 + 64)
    for attempt in range(2):
        out = np.empty((b, cap), dtype=np.uint8)
        nbytes = np.zeros(b, dtype=np.int64)
        rc = _lib.load().licos_rans_encode_host(sp, ip, n, 1, n, plane, cp, cdf_a.shape[1], lp, op, cdf_a.shape[0],
                                                ctypes.c_void_p(enc_table.ctypes.data), ctypes.c_void_p(out.ctypes.data), cap,
                                                ctypes.c_void_p(nbytes.ctypes.data), b, nthreads)
        if rc != -4 or attempt == 1:
            break
        cap = 8 * n + 16  # worst case: under two words per symbol
    _lib.check(rc, "rans_encode_host")
    return out, nbytes


def rans_encode_host_packed(packed, n, cdf, cdf_len, offset, enc_table, nthreads=None):
    """Host coder on packed words (ops.gc_pack_symbols): int32 numpy [B][n].  Returns (out uint8 [B][cap], nbytes int64 [B])."""
    pk, pp = _np_i32(packed)
    b = pk.shape[0]
    cdf_a, cp = _np_i32(cdf)
    len_a, lp = _np_i32(cdf_len)
    off_a, op = _np_i32(offset)
    nthreads = host_threads() if nthreads is None else int(nthreads)
    cap = 4 * (n // 2 + 64)
    for attempt in range(2):
        out = np.empty((b, cap), dtype=np.uint8)
        nbytes = np.zeros(b, dtype=np.int64)
        rc = _lib.load().licos_rans_encode_host_packed(pp, n, n, cp, cdf_a.shape[1], lp, op, cdf_a.shape[0],
                                                       ctypes.c_void_p(enc_table.ctypes.data), ctypes.c_void_p(out.ctypes.data), cap,
                                                       ctypes.c_void_p(nbytes.ctypes.data), b, nthreads)
        if rc != -4 or attempt == 1:
            break
        cap = 8 * n + 16  # worst case: under two words per symbol
    _lib.check(rc, "rans_encode_host_packed")
    return out, nbytes


def rans_encode_host_sym16(symbols16, n, plane, cdf, cdf_len, offset, enc_table, nthreads=None):
    """Host coder on 16-bit symbols, channel-plane rows: int16 numpy [B][n].  Returns (out uint8 [B][cap], nbytes int64 [B])."""
    sym = np.ascontiguousarray(symbols16, dtype=np.int16)
    b = sym.shape[0]
    cdf_a, cp = _np_i32(cdf)
    len_a, lp = _np_i32(cdf_len)
    off_a, op = _np_i32(offset)
    nthreads = host_threads() if nthreads is None else int(nthreads)
    cap = 4 * (n // 2 + 64)
    for attempt in range(2):
        out = np.empty((b, cap), dtype=np.uint8)
        nbytes = np.zeros(b, dtype=np.int64)
        rc = _lib.load().licos_rans_encode_host_sym16(ctypes.c_void_p(sym.ctypes.data), n, n, plane, cp, cdf_a.shape[1], lp, op,
                                                      cdf_a.shape[0], ctypes.c_void_p(enc_table.ctypes.data),
                                                      ctypes.c_void_p(out.ctypes.data), cap, ctypes.c_void_p(nbytes.ctypes.data), b, nthreads)
        if rc != -4 or attempt == 1:
            break
        cap = 8 * n + 16  # worst case: under two words per symbol
    _lib.check(rc, "rans_encode_host_sym16")
    return out, nbytes


def rans_decode_host_sym16(data, byte_off, n, plane, cdf, cdf_len, offset, batch, out, nthreads=None):
    """Host decoder into 16-bit symbols (int16 numpy [B][n], channel-plane rows).  Returns the status: 1 a stream ended
    early, 3 a value does not fit int16 (decode again with rans_decode_host)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off64 = np.ascontiguousarray(byte_off, dtype=np.int64)
    cdf_a, cp = _np_i32(cdf)
    len_a, lp = _np_i32(cdf_len)
    off_a, op = _np_i32(offset)
    status = np.zeros(1, dtype=np.int32)
    nthreads = host_threads() if nthreads is None else int(nthreads)
    assert out.dtype == np.int16 and out.flags["C_CONTIGUOUS"] and out.shape == (batch, n)
    rc = _lib.load().licos_rans_decode_host_sym16(ctypes.c_void_p(data.ctypes.data), ctypes.c_void_p(off64.ctypes.data), n, n, plane, cp,
                                                  cdf_a.shape[1], lp, op, cdf_a.shape[0], ctypes.c_void_p(out.ctypes.data),
                                                  ctypes.c_void_p(status.ctypes.data), batch, nthreads)
    _lib.check(rc, "rans_decode_host_sym16")
    return int(status[0])


def rans_decode_host_rows8(data, byte_off, rows8, n, cdf, cdf_len, offset, batch, out, nthreads=None):
    """Host decoder with one table-row byte per symbol (ops.gc_build_rows8): rows8 uint8 numpy [B][n], out int32 [B][n].
    Returns the status (1: a stream ended early)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off64 = np.ascontiguousarray(byte_off, dtype=np.int64)
    rows8 = np.ascontiguousarray(rows8, dtype=np.uint8)
    cdf_a, cp = _np_i32(cdf)
    len_a, lp = _np_i32(cdf_len)
    off_a, op = _np_i32(offset)
    status = np.zeros(1, dtype=np.int32)
    nthreads = host_threads() if nthreads is None else int(nthreads)
    assert out.dtype == np.int32 and out.flags["C_CONTIGUOUS"] and out.shape == (batch, n) and rows8.shape == (batch, n)
    rc = _lib.load().licos_rans_decode_host_rows8(ctypes.c_void_p(data.ctypes.data), ctypes.c_void_p(off64.ctypes.data),
                                                  ctypes.c_void_p(rows8.ctypes.data), n, n, cp, cdf_a.shape[1], lp, op, cdf_a.shape[0],
                                                  ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(status.ctypes.data), batch, nthreads)
    _lib.check(rc, "rans_decode_host_rows8")
    return int(status[0])


def rans_decode_host(data, byte_off, n, plane, cdf, cdf_len, offset, batch, indexes=None, nthreads=None, out=None):
    """Host decoder.  data: uint8 numpy (all streams), byte_off int64 [B+1].  Returns (symbols int32 [B][n], status)."""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off64 = np.ascontiguousarray(byte_off, dtype=np.int64)
    cdf_a, cp = _np_i32(cdf)
    len_a, lp = _np_i32(cdf_len)
    off_a, op = _np_i32(offset)
    ip = ctypes.c_void_p(0)
    if indexes is not None:
        idx, ip = _np_i32(indexes)
    sym = np.empty((batch, n), dtype=np.int32) if out is None else out
    status = np.zeros(1, dtype=np.int32)
    nthreads = host_threads() if nthreads is None else int(nthreads)
    rc = _lib.load().licos_rans_decode_host(ctypes.c_void_p(data.ctypes.data), ctypes.c_void_p(off64.ctypes.data), ip, n, 1, n,
                                            plane, cp, cdf_a.shape[1], lp, op, cdf_a.shape[0], ctypes.c_void_p(sym.ctypes.data),
                                            ctypes.c_void_p(status.ctypes.data), batch, nthreads)
    _lib.check(rc, "rans_decode_host")
    return sym, int(status[0])


# ----------------------------------------------------------------------------- 16-bit MFMA path
EPI_NONE, EPI_GDN, EPI_IGDN, EPI_RELU = 0, 1, 2, 3
EPI_ACCUMULATE = 0x100  # licos_hip.h LICOS_EPI_ACCUMULATE
EPI_IN_XSPLIT, EPI_OUT_XSPLIT = 0x200, 0x400  # licos_hip.h: blk16 rows stored as [even-x pixels][odd-x pixels]
EPI_NORM32, EPI_OUT_SPLIT3 = 0x40000, 0x80000  # licos_hip.h: (I)GDN norm at fp32 accuracy; output = the next fp32 convolution's split operand


def deconv_layouts(cin, h, w, cout):
    """Layout flags (EPI_IN_XSPLIT | EPI_OUT_XSPLIT or 0) the transposed-conv stage of this shape accepts."""
    return int(_lib.load().licos_deconv5x5s2_f16_layouts(int(cin), int(h), int(w), int(cout)))


def blk16_xsplit(x_blk, inverse=False):
    """blk16 [B][C16][H][W][16] <-> its x-split form (same shape, rows re-ordered); host-side helper for tests / tools."""
    b, c16, h, w, k = x_blk.shape
    if inverse:
        return x_blk.reshape(b, c16, h, 2, w // 2, k).permute(0, 1, 2, 4, 3, 5).reshape(b, c16, h, w, k).contiguous()
    return x_blk.reshape(b, c16, h, w // 2, 2, k).permute(0, 1, 2, 4, 3, 5).reshape(b, c16, h, w, k).contiguous()


def mfma_tiles(cout):
    """Number of 32-channel accumulator tiles the MFMA kernels use for `cout` output channels."""
    if cout <= 32:
        return 1
    if cout <= 128:
        return 4
    if cout <= 192:
        return 6
    if cout <= 320:
        return 10
    raise ValueError(f"licos_amd: the fp16 MFMA path supports at most 320 output channels, got {cout}")


def pack_conv_w_f16(w, transposed=False):
    """fp32 conv weight -> fp16 MFMA A-fragments (+ returns padded fp32 bias holder size)."""
    _dev(w)
    if transposed:
        cin, cout = w.shape[:2]
    else:
        cout, cin = w.shape[:2]
    if tuple(w.shape[2:]) != (5, 5):
        raise ValueError("licos_amd: the fp16 MFMA path implements 5x5 stride-2 stages only")
    nbytes = _lib.load().licos_packed_conv_w_bytes(cin, cout)
    if nbytes == 0:
        raise ValueError(f"licos_amd: unsupported channel counts Cin={cin} Cout={cout} for the fp16 path")
    packed = torch.empty(nbytes // 2, device=w.device, dtype=torch.float16)
    fn = _lib.load().licos_pack_deconv_w_f16 if transposed else _lib.load().licos_pack_conv_w_f16
    _lib.check(fn(_p(_f32(w.contiguous())), cin, cout, _p(packed), _stream()), "pack_conv_w_f16")
    return packed


def pad_bias(bias, cout, device):
    n = 32 * mfma_tiles(cout)
    out = torch.zeros(n, device=device, dtype=torch.float32)
    if bias is not None:
        out[:cout] = bias.detach()
    return out


def pack_gdn_bf16(beta_raw, gamma_raw, beta_bound, gamma_bound, pedestal):
    _dev(beta_raw, gamma_raw)
    c = beta_raw.numel()
    nbytes = _lib.load().licos_packed_gdn_bytes(c)
    if nbytes == 0:
        raise ValueError(f"licos_amd: GDN over {c} channels unsupported on the fp16 path")
    packed = torch.empty(nbytes, device=beta_raw.device, dtype=torch.uint8)
    rc = _lib.load().licos_pack_gdn_bf16(_p(_f32(beta_raw)), _p(_f32(gamma_raw)), beta_bound, gamma_bound, pedestal, c,
                                         _p(packed), _stream())
    _lib.check(rc, "pack_gdn_bf16")
    return packed


def pack_gdn_f32split(beta_raw, gamma_raw, beta_bound, gamma_bound, pedestal):
    """Packed operand of EPI_NORM32 (None: this channel count is not served)."""
    _dev(beta_raw, gamma_raw)
    c = beta_raw.numel()
    nbytes = _lib.load().licos_packed_gdn_f32split_bytes(c)
    if nbytes == 0 or mfma_tiles(c) != 4:
        return None
    packed = torch.empty(nbytes, device=beta_raw.device, dtype=torch.uint8)
    rc = _lib.load().licos_pack_gdn_f32split(_p(_f32(beta_raw)), _p(_f32(gamma_raw)), beta_bound, gamma_bound, pedestal, c,
                                             _p(packed), _stream())
    _lib.check(rc, "pack_gdn_f32split")
    return packed


def nchw_f32_to_blk16(x, abs_input=False):
    _dev(x)
    b, c, h, w = x.shape
    y = torch.empty((b, (c + 15) // 16, h, w, 16), device=x.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_nchw_f32_to_blk16(_p(_f32(x)), _p(y), b, c, h, w, int(abs_input), _stream()),
               "nchw_f32_to_blk16")
    return y


def blk16_to_nchw_f32(x, c):
    _dev(x)
    b, c16, h, w, _ = x.shape
    y = torch.empty((b, c, h, w), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().licos_blk16_to_nchw_f32(_p(x), _p(y), b, c, h, w, _stream()), "blk16_to_nchw_f32")
    return y


def _out_nchw(out, shape, device):
    if out is None:
        return torch.empty(shape, device=device, dtype=torch.float32)
    if tuple(out.shape) != tuple(shape) or out.dtype != torch.float32:
        raise ValueError(f"licos_amd: output tensor must be float32 of shape {shape}")
    return out


def conv5x5s2_f16(x_blk, w_packed, bias_padded, gdn_packed, epilogue, cin, cout, out_nchw=False, out=None):
    _dev(x_blk, w_packed, bias_padded, gdn_packed, out)
    b, c16, h, w, _ = x_blk.shape
    if c16 != (cin + 15) // 16 or x_blk.dtype != torch.float16:
        raise ValueError("conv5x5s2_f16: input is not the blk16 fp16 layout of `cin` channels")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if out_nchw:
        y = _out_nchw(out, (b, cout, ho, wo), x_blk.device)
        yb, yn = None, y
    else:
        y = torch.empty((b, (3 if epilogue & EPI_OUT_SPLIT3 else 1) * ((cout + 15) // 16), ho, wo, 16), device=x_blk.device, dtype=torch.float16)
        yb, yn = y, None
    rc = _launch(_lib.load().licos_conv5x5s2_f16, _p(x_blk), _p(w_packed), _p(bias_padded), _p(gdn_packed), epilogue, _p(yb),
                                         _p(yn), b, cin, h, w, cout, _stream())
    _lib.check(rc, "conv5x5s2_f16")
    return y


def conv5x5s2_f16_symbols(x_blk, w_packed, bias_padded, medians, cin, cout, out=None):
    """The last analysis stage with the entropy bottleneck's quantiser in its epilogue (licos_conv5x5s2_f16_symbols):
    int32 symbols (B, cout, Ho, Wo) - the coder's [stream][position] layout - written into `out` when given."""
    _dev(x_blk, w_packed, bias_padded, medians, out)
    b, c16, h, w, _ = x_blk.shape
    if c16 != (cin + 15) // 16 or x_blk.dtype != torch.float16:
        raise ValueError("conv5x5s2_f16_symbols: input is not the blk16 fp16 layout of `cin` channels")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if out is None:
        out = torch.empty((b, cout, ho, wo), device=x_blk.device, dtype=torch.int32)
    elif out.dtype != torch.int32 or tuple(out.shape) != (b, cout, ho, wo) or not out.is_contiguous():
        raise ValueError("conv5x5s2_f16_symbols: `out` must be a contiguous int32 (B, cout, Ho, Wo) tensor")
    med = _f32(medians)
    if med.numel() != cout:
        raise ValueError("conv5x5s2_f16_symbols: one median per output channel")
    rc = _launch(_lib.load().licos_conv5x5s2_f16_symbols, _p(x_blk), _p(w_packed), _p(bias_padded), _p(med), _p(out), b, cin, h, w, cout,
                 _stream())
    _lib.check(rc, "conv5x5s2_f16_symbols")
    return out


def deconv5x5s2_f16(x_blk, w_packed, bias_padded, gdn_packed, epilogue, cin, cout, out_nchw=False, clamp01=False,
                    out=None):
    _dev(x_blk, w_packed, bias_padded, gdn_packed, out)
    b, c16, h, w, _ = x_blk.shape
    if c16 != (cin + 15) // 16 or x_blk.dtype != torch.float16:
        raise ValueError("deconv5x5s2_f16: input is not the blk16 fp16 layout of `cin` channels")
    ho, wo = 2 * h, 2 * w
    if out_nchw:
        y = _out_nchw(out, (b, cout, ho, wo), x_blk.device)
        yb, yn = None, y
    else:
        y = torch.empty((b, (3 if epilogue & EPI_OUT_SPLIT3 else 1) * ((cout + 15) // 16), ho, wo, 16), device=x_blk.device, dtype=torch.float16)
        yb, yn = y, None
    rc = _launch(_lib.load().licos_deconv5x5s2_f16, _p(x_blk), _p(w_packed), _p(bias_padded), _p(gdn_packed), epilogue, _p(yb),
                                           _p(yn), int(clamp01), b, cin, h, w, cout, _stream())
    _lib.check(rc, "deconv5x5s2_f16")
    return y


def nchw_f32_to_s2d_blk16(x):
    _dev(x)
    b, c, h, w = x.shape
    y = torch.empty((b, (4 * c + 15) // 16, h // 2, w // 2, 16), device=x.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_nchw_f32_to_s2d_blk16(_p(_f32(x)), _p(y), b, c, h, w, _stream()), "nchw_f32_to_s2d_blk16")
    return y


def pack_conv_w_s2d_f16(w):
    _dev(w)
    cout, cin = w.shape[:2]
    mt = mfma_tiles(cout)
    packed = torch.empty(((4 * cin + 15) // 16) * 9 * mt * 512, device=w.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_pack_conv_w_s2d_f16(_p(_f32(w.contiguous())), cin, cout, _p(packed), _stream()),
               "pack_conv_w_s2d_f16")
    return packed


def conv5x5s2_s2d_f16(x_s2d, w_packed, bias_padded, gdn_packed, epilogue, cin, cout, h, w, out_nchw=False, out=None):
    """h, w: ORIGINAL (even) image size; x_s2d from nchw_f32_to_s2d_blk16."""
    _dev(x_s2d, w_packed, bias_padded, gdn_packed, out)
    b = x_s2d.shape[0]
    ho, wo = h // 2, w // 2
    if out_nchw:
        y = _out_nchw(out, (b, cout, ho, wo), x_s2d.device)
        yb, yn = None, y
    else:
        y = torch.empty((b, (cout + 15) // 16, ho, wo, 16), device=x_s2d.device, dtype=torch.float16)
        yb, yn = y, None
    rc = _launch(_lib.load().licos_conv5x5s2_s2d_f16, _p(x_s2d), _p(w_packed), _p(bias_padded), _p(gdn_packed), epilogue, _p(yb),
                                             _p(yn), b, cin, h, w, cout, _stream())
    _lib.check(rc, "conv5x5s2_s2d_f16")
    return y


def nchw_f32_to_hwc_pad_f16(x):
    """NCHW fp32 (1..3 bands) -> the interleaved, zero-bordered fp16 image of the first analysis stage (csrc/mfma_first.hip);
    returns a flat fp16 buffer (with the slack the kernel's edge tiles may touch)."""
    _dev(x)
    b, c, h, w = x.shape
    nbytes = _lib.load().licos_hwc_pad_f16_bytes(b, c, h, w)
    if nbytes == 0:
        raise ValueError(f"licos_amd: the interleaved first-stage image supports 1..3 bands, got {c}")
    y = torch.empty(nbytes // 2, device=x.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_nchw_f32_to_hwc_pad_f16(_p(_f32(x)), _p(y), b, c, h, w, _stream()), "nchw_f32_to_hwc_pad_f16")
    return y


def pack_conv_w_first_f16(w):
    _dev(w)
    cout, cin = w.shape[:2]
    nbytes = _lib.load().licos_packed_conv_w_first_bytes(cin, cout)
    if nbytes == 0 or tuple(w.shape[2:]) != (5, 5):
        raise ValueError(f"licos_amd: the first-stage kernel takes 1..3 input and <= 128 output channels, 5x5; got {tuple(w.shape)}")
    packed = torch.empty(nbytes // 2, device=w.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_pack_conv_w_first_f16(_p(_f32(w.contiguous())), cin, cout, _p(packed), _stream()),
               "pack_conv_w_first_f16")
    return packed


def conv5x5s2_first_f16(x_hwc, w_packed, bias_padded, gdn_packed, epilogue, b, cin, cout, h, w):
    """b, h, w: batch and ORIGINAL image size of the NCHW tensor `x_hwc` was made from; returns blk16 fp16."""
    _dev(x_hwc, w_packed, bias_padded, gdn_packed)
    if x_hwc.dtype != torch.float16 or x_hwc.numel() * 2 < _lib.load().licos_hwc_pad_f16_bytes(b, cin, h, w):
        raise ValueError("conv5x5s2_first_f16: input is not the buffer nchw_f32_to_hwc_pad_f16 makes for this shape")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((b, (cout + 15) // 16, ho, wo, 16), device=x_hwc.device, dtype=torch.float16)
    rc = _launch(_lib.load().licos_conv5x5s2_first_f16, _p(x_hwc), _p(w_packed), _p(bias_padded), _p(gdn_packed), epilogue, _p(y),
                                               b, cin, h, w, cout, _stream())
    _lib.check(rc, "conv5x5s2_first_f16")
    return y


def conv5x5s2_first_nchw_f16(x, w_packed, bias_padded, gdn_packed, epilogue, cout):
    """First analysis stage on the NCHW fp32 image in place (1..3 bands, width a multiple of 4); returns blk16 fp16."""
    _dev(x, w_packed, bias_padded, gdn_packed)
    b, cin, h, w = x.shape
    if x.dtype != torch.float32 or w % 4:
        raise ValueError("conv5x5s2_first_nchw_f16: needs a float32 image whose width is a multiple of 4")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((b, (cout + 15) // 16, ho, wo, 16), device=x.device, dtype=torch.float16)
    rc = _launch(_lib.load().licos_conv5x5s2_first_nchw_f16, _p(x), _p(w_packed), _p(bias_padded), _p(gdn_packed), epilogue, _p(y),
                                                    b, cin, h, w, cout, _stream())
    _lib.check(rc, "conv5x5s2_first_nchw_f16")
    return y


def conv5x5s2_first16_nchw_f16(x, w_packed, bias_padded, gdn_packed, epilogue, cout):
    """First analysis stage for 5..16 bands on the NCHW fp32 image in place (width a multiple of 4; csrc/mfma_first16.hip);
    w_packed: pack_conv_w_f16 of the layer's weight.  Returns blk16 fp16."""
    _dev(x, w_packed, bias_padded, gdn_packed)
    b, cin, h, w = x.shape
    if x.dtype != torch.float32 or w % 4:
        raise ValueError("conv5x5s2_first16_nchw_f16: needs a float32 image whose width is a multiple of 4")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((b, (cout + 15) // 16, ho, wo, 16), device=x.device, dtype=torch.float16)
    rc = _launch(_lib.load().licos_conv5x5s2_first16_nchw_f16, _p(x), _p(w_packed), _p(bias_padded), _p(gdn_packed), epilogue, _p(y),
                 b, cin, h, w, cout, _stream())
    _lib.check(rc, "conv5x5s2_first16_nchw_f16")
    return y


def pack_deconv_w_fewch_f16(w):
    _dev(w)
    cin, cout = w.shape[:2]
    nbytes = _lib.load().licos_packed_deconv_w_fewch_bytes(cin, cout)
    if nbytes == 0:
        raise ValueError(f"licos_amd: few-channel deconv supports at most 32 output channels, got {cout}")
    packed = torch.empty(nbytes // 2, device=w.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_pack_deconv_w_fewch_f16(_p(_f32(w.contiguous())), cin, cout, _p(packed), _stream()),
               "pack_deconv_w_fewch_f16")
    return packed


def deconv5x5s2_fewch_f16(x_blk, w_packed, bias_padded, cin, cout, clamp01=False, out=None):
    _dev(x_blk, w_packed, bias_padded, out)
    b, c16, h, w, _ = x_blk.shape
    if c16 != (cin + 15) // 16 or x_blk.dtype != torch.float16:
        raise ValueError("deconv5x5s2_fewch_f16: input is not the blk16 fp16 layout of `cin` channels")
    y = _out_nchw(out, (b, cout, 2 * h, 2 * w), x_blk.device)
    rc = _launch(_lib.load().licos_deconv5x5s2_fewch_f16, _p(x_blk), _p(w_packed), _p(bias_padded), _p(y), int(clamp01), b, cin, h,
                                                 w, cout, _stream())
    _lib.check(rc, "deconv5x5s2_fewch_f16")
    return y


def pack_deconv_w_scatter_f16(w):
    _dev(w)
    cin, cout = w.shape[:2]
    nbytes = _lib.load().licos_packed_deconv_w_scatter_bytes(cin, cout)
    if nbytes == 0:
        raise ValueError(f"licos_amd: scatter-form deconv supports 1..4 output channels, got {cout}")
    packed = torch.empty(nbytes // 2, device=w.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_pack_deconv_w_scatter_f16(_p(_f32(w.contiguous())), cin, cout, _p(packed), _stream()),
               "pack_deconv_w_scatter_f16")
    return packed


def deconv5x5s2_scatter_f16(x_blk, w_packed, bias, cin, cout, clamp01=False, out=None, in_xsplit=False):
    _dev(x_blk, w_packed, bias, out)
    b, c16, h, w, _ = x_blk.shape
    if c16 != (cin + 15) // 16 or x_blk.dtype != torch.float16:
        raise ValueError("deconv5x5s2_scatter_f16: input is not the blk16 fp16 layout of `cin` channels")
    y = _out_nchw(out, (b, cout, 2 * h, 2 * w), x_blk.device)
    rc = _launch(_lib.load().licos_deconv5x5s2_scatter_f16, _p(x_blk), _p(w_packed), _p(bias), _p(y), int(bool(clamp01)) | (2 if in_xsplit else 0),
                                                   b, cin, h, w, cout, _stream())
    _lib.check(rc, "deconv5x5s2_scatter_f16")
    return y

def pack_deconv_w_rows_f16(w):
    _dev(w)
    cin, cout = w.shape[:2]
    nbytes = _lib.load().licos_packed_deconv_w_rows_bytes(cin, cout)
    if nbytes == 0:
        raise ValueError(f"licos_amd: row-walking deconv supports 1..3 output channels (5..16 from 113..128 input channels), got {cin} -> {cout}")
    packed = torch.empty(nbytes // 2, device=w.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_pack_deconv_w_rows_f16(_p(_f32(w.contiguous())), cin, cout, _p(packed), _stream()),
               "pack_deconv_w_rows_f16")
    return packed


def deconv5x5s2_rows_f16(x_blk, w_packed, bias, cin, cout, clamp01=False, out=None, in_xsplit=False):
    """Last synthesis stage, row-walking form: 1..3 output channels (csrc/mfma_rows.hip) or 5..16 out of 113..128
    (csrc/mfma_rows16.hip): NCHW fp32 out."""
    _dev(x_blk, w_packed, bias, out)
    b, c16, h, w, _ = x_blk.shape
    if c16 != (cin + 15) // 16 or x_blk.dtype != torch.float16:
        raise ValueError("deconv5x5s2_rows_f16: input is not the blk16 fp16 layout of `cin` channels")
    y = _out_nchw(out, (b, cout, 2 * h, 2 * w), x_blk.device)
    rc = _launch(_lib.load().licos_deconv5x5s2_rows_f16, _p(x_blk), _p(w_packed), _p(bias), _p(y), int(bool(clamp01)) | (2 if in_xsplit else 0),
                                                b, cin, h, w, cout, _stream())
    _lib.check(rc, "deconv5x5s2_rows_f16")
    return y


def pack_conv3x3_w_f16(w):
    _dev(w)
    cout, cin = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3):
        raise ValueError("pack_conv3x3_w_f16: expected a 3x3 kernel")
    packed = torch.empty(((cin + 15) // 16) * 9 * mfma_tiles(cout) * 512, device=w.device, dtype=torch.float16)
    _lib.check(_lib.load().licos_pack_conv3x3_w_f16(_p(_f32(w.contiguous())), cin, cout, _p(packed), _stream()),
               "pack_conv3x3_w_f16")
    return packed


def conv3x3s1_f16(x_blk, w_packed, bias_padded, gdn_packed, epilogue, cin, cout, out_nchw=False, out=None):
    _dev(x_blk, w_packed, bias_padded, gdn_packed, out)
    b, c16, h, w, _ = x_blk.shape
    if c16 != (cin + 15) // 16 or x_blk.dtype != torch.float16:
        raise ValueError("conv3x3s1_f16: input is not the blk16 fp16 layout of `cin` channels")
    if out_nchw:
        y = _out_nchw(out, (b, cout, h, w), x_blk.device)
        yb, yn = None, y
    else:
        y = torch.empty((b, (3 if epilogue & EPI_OUT_SPLIT3 else 1) * ((cout + 15) // 16), h, w, 16), device=x_blk.device, dtype=torch.float16)
        yb, yn = y, None
    rc = _launch(_lib.load().licos_conv3x3s1_f16, _p(x_blk), _p(w_packed), _p(bias_padded), _p(gdn_packed), epilogue, _p(yb),
                                         _p(yn), b, cin, h, w, cout, _stream())
    _lib.check(rc, "conv3x3s1_f16")
    return y
