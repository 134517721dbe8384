"""Host-side driver of the fused 16-bit MFMA pipeline.

A transform (``g_a`` / ``g_s``) is a chain of (conv | deconv)[+ GDN] stages.  Each stage is one
HIP kernel (licos_amd/csrc/mfma_{conv,deconv}.hip) reading and writing the blk16 fp16 layout; only
the two ends of the chain touch NCHW fp32.  Packed operands (MFMA weight fragments, padded bias,
reparametrised bf16 gamma fragments) are cached per module and rebuilt when a parameter's
version or storage changes or when the package's weights epoch moves (ops.touch_weights: the fused
Adam, the federated average and every other raw-pointer writer bump it, since torch's version counter
does not see their writes) - so modules swapped in after construction (model_utils.py:31-45),
optimiser steps and averaging steps are all picked up lazily.
"""
import os
import weakref

import torch
import torch.nn as nn

from . import ops
from .layers import GDN, conv_geometry

_cache = weakref.WeakKeyDictionary()

# Optional per-stage device timing (bench.py): when set to a dict, every stage launch is bracketed by
# HIP events recorded on the stream the kernel is launched on; key = (kind, Cin, Cout, H, W, tiles in the launch,
# stage carries a fused GDN / IGDN).
stage_events = None


# (stages launched on fewer tiles than this are not timed: the sub-chunks of the host's share are a hundred small launches
# per step, and two events around each are ~2 ms of a 196-ms step; bench.py sets it for its timed region)
stage_events_min_batch = int(os.environ.get("LICOS_STAGE_EVENTS_MIN", "0"))


def _timed(key, fn):
    if stage_events is None or key[5] < stage_events_min_batch:
        return fn()
    sink = ops.stage_event_sink = []  # filled by ops._launch around the stage's C call (not around its output allocation)
    try:
        out = fn()
    finally:
        ops.stage_event_sink = None
    if sink:
        stage_events.setdefault(key, []).append((sink[0][0], sink[-1][1]))
    return out


def _pver(p):
    return None if p is None else (p.data_ptr(), p._version, str(p.device), ops.weights_epoch())


SCATTER_LAST = os.environ.get("LICOS_SCATTER", "1") != "0"  # A/B switch for the scatter-form last stage
FIRST_ROWS = os.environ.get("LICOS_FIRST", "1") != "0"  # kernel-row first stage (1..3 bands) instead of the space-to-depth 3x3 form
FIRST_RAW = os.environ.get("LICOS_FIRST_RAW", "1") != "0"  # ... reading the NCHW fp32 image in place (no layout pass) when W % 4 == 0
ROWS_LAST = os.environ.get("LICOS_ROWS", "1") != "0"  # row-walking last stage (1..3 bands) instead of the scatter form
ROWS16_LAST = os.environ.get("LICOS_ROWS16_LAST", "1") != "0"  # 5..16 bands: the last stage row-walking (csrc/mfma_rows16.hip), 0 = few16
FIRST16 = os.environ.get("LICOS_FIRST16", "1") != "0"  # 5..16 bands: the first stage on the NCHW fp32 image in place (csrc/mfma_first16.hip)


def _packed_conv(m, s2d=False, fewch=False, first=False):
    key = (_pver(m.weight), _pver(m.bias), s2d, fewch, first)
    ent = _cache.get(m)
    if ent is None or ent[0] != key:
        transposed = isinstance(m, nn.ConvTranspose2d)
        geo = conv_geometry(m)
        k3 = (not transposed) and geo[:3] == (3, 1, 1)
        if not k3 and (geo[:3] != (5, 2, 2) or (transposed and geo[3] != 1)):
            raise ValueError("licos_amd: the fp16 MFMA path implements 5x5 stride-2 (de)convolutions and 3x3 "
                             "stride-1 convolutions; use precision='fp32' for other shapes")
        cout = m.out_channels
        if first:
            wp = ops.pack_conv_w_first_f16(m.weight.detach())
        elif fewch == "rows":
            wp = ops.pack_deconv_w_rows_f16(m.weight.detach())
        elif fewch == "scatter":
            wp = ops.pack_deconv_w_scatter_f16(m.weight.detach())
        elif fewch:
            wp = ops.pack_deconv_w_fewch_f16(m.weight.detach())
        elif k3:
            wp = ops.pack_conv3x3_w_f16(m.weight.detach())
        elif s2d:
            wp = ops.pack_conv_w_s2d_f16(m.weight.detach())
        else:
            wp = ops.pack_conv_w_f16(m.weight.detach(), transposed=transposed)
        bp = ops.pad_bias(m.bias, cout, m.weight.device)
        ent = (key, wp, bp)
        _cache[m] = ent
    return ent[1], ent[2]


def _packed_gdn(m):
    key = (_pver(m.beta), _pver(m.gamma))
    ent = _cache.get(m)
    if ent is None or ent[0] != key:
        bb, gb, ped = m.reparam_args()
        ent = (key, ops.pack_gdn_bf16(m.beta.detach(), m.gamma.detach(), bb, gb, ped))
        _cache[m] = ent
    return ent[1]


def stages(seq):
    """[(conv module, gdn module or None)] for a transform chain; validates the pattern."""
    mods = list(seq)
    out = []
    i = 0
    while i < len(mods):
        m = mods[i]
        if not isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            raise TypeError(f"licos_amd: fp16 path expects conv/deconv stages, found {type(m).__name__}")
        g = None
        if i + 1 < len(mods) and isinstance(mods[i + 1], GDN):
            g = mods[i + 1]
            if g.in_channels != m.out_channels:
                raise ValueError("licos_amd: GDN width does not match the preceding conv")
            i += 1
        elif i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
            g = "relu"
            i += 1
        elif i + 1 < len(mods) and not isinstance(mods[i + 1], (nn.Conv2d, nn.ConvTranspose2d)):
            raise TypeError(f"licos_amd: fp16 path does not fuse {type(mods[i + 1]).__name__}; use precision='fp32'")
        out.append((m, g))
        i += 1
    return out


def symbols_fusable(seq):
    """Can the chain's last stage carry the entropy bottleneck's quantiser in its epilogue (run_chain_fp16's `symbols`)?
    A plain 5x5 stride-2 convolution with nothing behind it - every analysis transform of the zoo."""
    st = stages(seq)
    m, g = st[-1]
    return (len(st) > 1 and g is None and isinstance(m, nn.Conv2d) and not isinstance(m, nn.ConvTranspose2d)
            and conv_geometry(m)[:3] == (5, 2, 2))


def run_chain_fp16(seq, x=None, x_blk=None, clamp01=False, out=None, symbols=None):
    """Runs the chain on NCHW fp32 `x` (or an already blocked fp16 `x_blk`); returns NCHW fp32
    (written into `out` when given).  symbols = (medians [C], int32 out (B, C, h, w) or None): the last stage writes
    rint(y - median) as int32 instead of y (symbols_fusable(seq) must hold) and that tensor is returned."""
    st = stages(seq)
    if symbols is not None and not symbols_fusable(seq):
        raise ValueError("licos_amd: this chain's last stage cannot carry the quantiser")
    s2d_first = False
    if x_blk is None:
        if x.dtype != torch.float32:
            raise ValueError("licos_amd: inputs must be float32")
        if x.shape[1] != st[0][0].in_channels:
            raise ValueError(f"expected {st[0][0].in_channels} input channels, got {x.shape[1]}")
        h0, w0 = x.shape[2], x.shape[3]
        # few input channels: 5x5 s2 over C == 3x3 s1 over the 4C channels of the 2x2 space-to-depth image
        abs_in = bool(getattr(seq, "abs_input", False))
        s2d_first = (isinstance(st[0][0], nn.Conv2d) and not isinstance(st[0][0], nn.ConvTranspose2d)
                     and conv_geometry(st[0][0])[:3] == (5, 2, 2) and not abs_in
                     and x.shape[1] <= 4 and h0 % 2 == 0 and w0 % 2 == 0)
        # 1..3 bands into <= 128 channels: K steps = kernel rows over the interleaved image (csrc/mfma_first.hip)
        first_rows = (FIRST_ROWS and isinstance(st[0][0], nn.Conv2d) and not isinstance(st[0][0], nn.ConvTranspose2d)
                      and conv_geometry(st[0][0])[:3] == (5, 2, 2) and not abs_in and x.shape[1] <= 3
                      and st[0][0].out_channels <= 128 and len(st) > 1 and (st[0][1] is None or st[0][1] == "relu" or not st[0][1].inverse)
                      and min(h0, w0) >= 16)
        # (the in-place forms read 16-byte granules of the fp32 rows: a view whose storage offset breaks that alignment takes
        # the layout-pass routes, which accept any pointer)
        x = x.contiguous()
        aligned16 = x.data_ptr() % 16 == 0
        first_raw = first_rows and FIRST_RAW and w0 % 4 == 0 and aligned16
        # 5..16 bands (the 13 merged Sentinel-2 bands) into 33..128 channels: in place as well, no blk16 layout pass
        first16 = (FIRST16 and isinstance(st[0][0], nn.Conv2d) and not isinstance(st[0][0], nn.ConvTranspose2d)
                   and conv_geometry(st[0][0])[:3] == (5, 2, 2) and not abs_in and 4 < x.shape[1] <= 16
                   and 32 < st[0][0].out_channels <= 128 and len(st) > 1
                   and (st[0][1] is None or st[0][1] == "relu" or not st[0][1].inverse) and w0 % 4 == 0 and min(h0, w0) >= 16
                   and aligned16)
        if first_rows or first16:
            s2d_first = False
            cur = x.contiguous() if (first_raw or first16) else ops.nchw_f32_to_hwc_pad_f16(x.contiguous())
        else:
            cur = ops.nchw_f32_to_s2d_blk16(x.contiguous()) if s2d_first else ops.nchw_f32_to_blk16(x.contiguous(), abs_in)
    else:
        first_rows = first16 = False
        cur = x_blk
    xsplit = False  # layout of `cur`: blk16, or its x-split form (ops.EPI_OUT_XSPLIT) between two kernels that agree on it

    def scatter_last(i):
        m, g = st[i]
        return (i == len(st) - 1 and isinstance(m, nn.ConvTranspose2d) and g is None and SCATTER_LAST
                and m.out_channels <= 4 and m.in_channels in (128, 192))

    def rows16_last(i, w):
        """5..16 bands out of 128 channels: the row-walking form on 16 x 16 x 32 tiles (csrc/mfma_rows16.hip) for maps that
        give its waves columns: a workgroup is 8 waves x 32 columns for maps wider than 128, 4 x 32 in two row groups up to
        128, 2 x 32 in four up to 64; where most of a strip's waves would idle the LDS-patch form wins (tools/last16_probe.py:
        2048 maps of 128^2 3.6 against 4.1 ms, 8192 of 64^2 3.7 against 4.0, 8192 of 32^2 2.5 against 1.1)."""
        m, g = st[i]
        return (i == len(st) - 1 and isinstance(m, nn.ConvTranspose2d) and g is None and ROWS_LAST and ROWS16_LAST
                and 5 <= m.out_channels <= 16 and 112 < m.in_channels <= 128 and (w >= 192 or 96 <= w <= 128 or 48 <= w <= 64))

    def takes_xsplit(i, h, w):
        """Does stage i, fed an h x w map, read the x-split layout?"""
        if i >= len(st) or w % 2:
            return False
        m, g = st[i]
        if not isinstance(m, nn.ConvTranspose2d):
            return False
        if scatter_last(i) or rows16_last(i, w):
            return True
        return i < len(st) - 1 and bool(ops.deconv_layouts(m.in_channels, h, w, m.out_channels) & ops.EPI_IN_XSPLIT)

    for idx, (m, g) in enumerate(st):
        last = idx == len(st) - 1
        fewch = last and isinstance(m, nn.ConvTranspose2d) and g is None and m.out_channels <= 32
        if scatter_last(idx):
            fewch = "rows" if (ROWS_LAST and m.out_channels <= 3) else "scatter"
        elif last and cur.dim() == 5 and rows16_last(idx, cur.shape[3]):
            fewch = "rows"
        wp, bp = _packed_conv(m, s2d=(s2d_first and idx == 0), fewch=fewch, first=(first_rows and idx == 0))
        if fewch in ("rows", "scatter"):
            key = ("deconv", m.in_channels, m.out_channels, cur.shape[2], cur.shape[3], cur.shape[0], False)
            last_op = ops.deconv5x5s2_rows_f16 if fewch == "rows" else ops.deconv5x5s2_scatter_f16
            cur = _timed(key, lambda: last_op(cur, wp, bp, m.in_channels, m.out_channels,
                                              clamp01=clamp01, out=out, in_xsplit=xsplit))
            xsplit = False  # NCHW fp32 from here
            continue
        if fewch:
            key = ("deconv", m.in_channels, m.out_channels, cur.shape[2], cur.shape[3], cur.shape[0], False)
            cur = _timed(key, lambda: ops.deconv5x5s2_fewch_f16(cur, wp, bp, m.in_channels, m.out_channels,
                                                                clamp01=clamp01, out=out))
            continue
        gp = _packed_gdn(g) if isinstance(g, GDN) else None
        epi = ops.EPI_NONE if g is None else ops.EPI_RELU if g == "relu" else (ops.EPI_IGDN if g.inverse else ops.EPI_GDN)
        norm = gp is not None
        if first16 and idx == 0:
            key = ("conv", m.in_channels, m.out_channels, h0, w0, x.shape[0], norm)
            cur = _timed(key, lambda: ops.conv5x5s2_first16_nchw_f16(cur, wp, bp, gp, epi, m.out_channels))
            continue
        if first_rows and idx == 0:
            key = ("conv", m.in_channels, m.out_channels, h0, w0, x.shape[0], norm)
            if first_raw:
                cur = _timed(key, lambda: ops.conv5x5s2_first_nchw_f16(cur, wp, bp, gp, epi, m.out_channels))
            else:
                cur = _timed(key, lambda: ops.conv5x5s2_first_f16(cur, wp, bp, gp, epi, x.shape[0], m.in_channels, m.out_channels, h0, w0))
            continue
        if s2d_first and idx == 0:
            key = ("conv", m.in_channels, m.out_channels, h0, w0, cur.shape[0], norm)
            cur = _timed(key, lambda: ops.conv5x5s2_s2d_f16(cur, wp, bp, gp, epi, m.in_channels, m.out_channels, h0, w0,
                                                            out_nchw=last, out=out if last else None))
            continue
        if isinstance(m, nn.Conv2d) and not isinstance(m, nn.ConvTranspose2d) and conv_geometry(m)[:3] == (3, 1, 1):
            key = ("conv3", m.in_channels, m.out_channels, cur.shape[2], cur.shape[3], cur.shape[0], norm)
            cur = _timed(key, lambda: ops.conv3x3s1_f16(cur, wp, bp, gp, epi, m.in_channels, m.out_channels,
                                                        out_nchw=last, out=out if last else None))
        elif isinstance(m, nn.ConvTranspose2d):
            hh, ww = cur.shape[2], cur.shape[3]
            key = ("deconv", m.in_channels, m.out_channels, hh, ww, cur.shape[0], norm)
            # one output phase = every other pixel of a row: hand the next kernel the x-split layout when both sides
            # speak it (whole-line stores here, nothing lost there: LDS-DMA addresses are per lane anyway)
            flags = ops.EPI_IN_XSPLIT if xsplit else 0
            out_split = (not last and bool(ops.deconv_layouts(m.in_channels, hh, ww, m.out_channels) & ops.EPI_OUT_XSPLIT)
                         and takes_xsplit(idx + 1, 2 * hh, 2 * ww))
            if out_split:
                flags |= ops.EPI_OUT_XSPLIT
            cur = _timed(key, lambda: ops.deconv5x5s2_f16(cur, wp, bp, gp, epi | flags, m.in_channels, m.out_channels,
                                                          out_nchw=last, clamp01=clamp01 and last,
                                                          out=out if last else None))
            xsplit = out_split
        elif last and symbols is not None:
            key = ("conv", m.in_channels, m.out_channels, cur.shape[2], cur.shape[3], cur.shape[0], norm)
            cur = _timed(key, lambda: ops.conv5x5s2_f16_symbols(cur, wp, bp, symbols[0], m.in_channels, m.out_channels, out=symbols[1]))
        else:
            key = ("conv", m.in_channels, m.out_channels, cur.shape[2], cur.shape[3], cur.shape[0], norm)
            cur = _timed(key, lambda: ops.conv5x5s2_f16(cur, wp, bp, gp, epi, m.in_channels, m.out_channels,
                                                        out_nchw=last, out=out if last else None))
    assert not xsplit, "a transform chain must not end in the x-split layout"
    return cur
