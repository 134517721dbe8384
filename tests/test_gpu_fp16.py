"""GPU tests of the 16-bit MFMA path.  Each stage is checked against torch-CPU convolution of the
SAME fp16-rounded operands (so only accumulation order and the output rounding differ), the fused
(I)GDN epilogue against the fp32 definition, and the whole model against the oracle on the
quantities BASELINE.json names for the fp16 configuration: bpp, PSNR, symbol mismatch rate."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import licos_amd
from licos_amd import engine, ops
from oracle import model as om

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elementwise_close(a, b, rtol, atol_of_max):
    """|a - b| <= rtol |b| + atol_of_max * max|b| for EVERY element (the max-norm rel_err above hides small entries)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return bool(((a - b).abs() <= rtol * b.abs() + atol_of_max * float(b.abs().max())).all())


def h16(t):
    return t.half().float()


@pytest.mark.parametrize("c,h,w", [(3, 8, 8), (16, 5, 7), (13, 16, 16), (192, 4, 4), (130, 3, 5)])
def test_blk16_layout_roundtrip(c, h, w):
    g = torch.Generator().manual_seed(c)
    x = h16(torch.randn(2, c, h, w, generator=g))
    blk = ops.nchw_f32_to_blk16(x.to(DEV))
    c16 = (c + 15) // 16
    assert tuple(blk.shape) == (2, c16, h, w, 16)
    ref = torch.zeros(2, c16 * 16, h, w)
    ref[:, :c] = x
    ref = ref.reshape(2, c16, 16, h, w).permute(0, 1, 3, 4, 2)
    assert torch.equal(blk.cpu().float(), ref)
    assert torch.equal(ops.blk16_to_nchw_f32(blk, c).cpu(), x)


@pytest.mark.parametrize("cin,cout,h,w", [(3, 128, 64, 64), (128, 128, 64, 64), (128, 128, 32, 32), (128, 192, 32, 32),
                                           (13, 128, 40, 72), (128, 128, 22, 38), (1, 128, 16, 16), (128, 192, 64, 80),
                                           # the 8-wave kernel (output >= 16 x 32, rows % 16 == 0): the bench geometry
                                           # 128^2 -> 64^2, ragged-x maps (Wo % 32 != 0), the 576 x 648 -> 288 x 324 granule map
                                           (128, 128, 128, 128), (128, 128, 128, 80), (128, 128, 64, 144), (128, 128, 576, 648)])
def test_conv_stage_exact_operands(cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + cout + h)
    x = h16(torch.randn(2, cin, h, w, generator=g))
    wt = h16(torch.randn(cout, cin, 5, 5, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, stride=2, padding=2)
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    wp = ops.pack_conv_w_f16(wt.to(DEV))
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    out = ops.conv5x5s2_f16(xb, wp, bp, None, ops.EPI_NONE, cin, cout, out_nchw=True)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-5
    assert elementwise_close(out, ref, 1e-5, 2e-5)
    outb = ops.conv5x5s2_f16(xb, wp, bp, None, ops.EPI_NONE, cin, cout, out_nchw=False)
    assert rel_err(ops.blk16_to_nchw_f32(outb, cout), ref) < 1e-3  # fp16 output rounding
    assert elementwise_close(ops.blk16_to_nchw_f32(outb, cout), ref, 4.9e-4, 2e-5)  # one fp16 rounding of the fp32 sum


@pytest.mark.parametrize("batch,cin,cout,h,w,relu", [(3, 128, 192, 32, 32, False), (5, 128, 192, 32, 32, True), (2, 128, 128, 64, 32, False),
                                                     (7, 192, 128, 32, 32, True), (4, 128, 192, 96, 32, False), (2, 16, 192, 32, 32, False)])
def test_conv_two_images_per_pixel_tile(batch, cin, cout, h, w, relu):
    """16-pixel-wide outputs (the last analysis stage of a 256^2 tile): the 8-wave kernel puts two images side by side
    in one 16 x 32 pixel tile and runs 192 channels as two groups of 96.  Odd batches (a pair with one image), several
    tile rows, both channel-group counts, the ReLU epilogue; against torch conv2d on the same fp16-exact operands."""
    g = torch.Generator().manual_seed(batch * 1000 + cin + cout + h)
    x = h16(torch.randn(batch, cin, h, w, generator=g))
    wt = h16(torch.randn(cout, cin, 5, 5, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x, wt, b, stride=2, padding=2)
    if relu:
        ref = ref.relu()
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    wp = ops.pack_conv_w_f16(wt.to(DEV))
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    epi = ops.EPI_RELU if relu else ops.EPI_NONE
    out = ops.conv5x5s2_f16(xb, wp, bp, None, epi, cin, cout, out_nchw=True)
    assert out.shape == ref.shape
    assert elementwise_close(out, ref, 1e-5, 2e-5)
    outb = ops.conv5x5s2_f16(xb, wp, bp, None, epi, cin, cout, out_nchw=False)
    assert elementwise_close(ops.blk16_to_nchw_f32(outb, cout), ref, 4.9e-4, 2e-5)
    # the result must not depend on which images share a tile: the last image alone in a batch of one, bit for bit
    one = ops.conv5x5s2_f16(ops.nchw_f32_to_blk16(x[batch - 1:].to(DEV)), wp, bp, None, epi, cin, cout, out_nchw=True)
    assert torch.equal(one, out[batch - 1:])


@pytest.mark.parametrize("cin,cout,h,w", [(192, 128, 16, 16), (192, 128, 4, 4), (128, 128, 32, 32), (128, 128, 64, 64),
                                           (128, 3, 32, 32), (128, 13, 20, 36), (128, 1, 16, 16), (128, 128, 11, 19),
                                           (128, 7, 9, 40), (192, 24, 8, 33),
                                           # the 13-band last stage (16 x 16 x 32 kernel): odd width (8-byte stores), 9 and 16 channels
                                           (128, 13, 9, 35), (128, 9, 33, 64), (128, 16, 16, 34),
                                           # the 8-wave kernel (input >= 16 x 32, <= 128 output channels): whole and ragged
                                           # 16 x 32 tiles, the bench geometry, a granule-shaped map, 96 channels
                                           (128, 128, 40, 80), (128, 128, 18, 35), (128, 96, 16, 32), (192, 128, 17, 64),
                                           (128, 128, 128, 128), (128, 128, 72, 81)])
def test_deconv_stage_exact_operands(cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + cout + h)
    x = h16(torch.randn(2, cin, h, w, generator=g))
    wt = h16(torch.randn(cin, cout, 5, 5, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x, wt, b, stride=2, padding=2, output_padding=1)
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    wp = ops.pack_conv_w_f16(wt.to(DEV), transposed=True)
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    out = ops.deconv5x5s2_f16(xb, wp, bp, None, ops.EPI_NONE, cin, cout, out_nchw=True)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-5
    if cout > 32:
        outb = ops.deconv5x5s2_f16(xb, wp, bp, None, ops.EPI_NONE, cin, cout, out_nchw=False)
        gotb = ops.blk16_to_nchw_f32(outb, cout).cpu()
        assert rel_err(gotb, ref) < 1e-3
        # element-wise: one fp16 rounding of the fp32 sum (2^-11 relative) plus the summation-order noise of the sum
        assert bool(((gotb - ref).abs() <= 4.9e-4 * ref.abs() + 2e-5 * float(ref.abs().max())).all())
    outc = ops.deconv5x5s2_f16(xb, wp, bp, None, ops.EPI_NONE, cin, cout, out_nchw=True, clamp01=True)
    assert rel_err(outc, ref.clamp(0, 1)) < 2e-5
    if cout <= 32:  # the all-phase few-channel kernel with compact weights
        wf = ops.pack_deconv_w_fewch_f16(wt.to(DEV))
        outf = ops.deconv5x5s2_fewch_f16(xb, wf, bp, cin, cout)
        assert rel_err(outf, ref) < 2e-5
        outf = ops.deconv5x5s2_fewch_f16(xb, wf, bp, cin, cout, clamp01=True)
        assert rel_err(outf, ref.clamp(0, 1)) < 2e-5


@pytest.mark.parametrize("batch,cin,cout,h,w,igdn", [(3, 192, 128, 16, 16, True), (5, 192, 128, 16, 16, False), (2, 128, 128, 33, 16, True),
                                                     (1, 192, 128, 16, 16, True), (4, 128, 96, 16, 16, False)])
def test_deconv_two_images_per_pixel_tile(batch, cin, cout, h, w, igdn):
    """16-pixel-wide inputs (the first synthesis stage of a 256^2 tile): the 8-wave transposed conv puts two images side
    by side in one 16 x 32 pixel tile.  Odd batches, ragged tile rows, with and without the fused IGDN; against torch
    conv_transpose2d (+ the oracle's IGDN arithmetic) on fp16-exact operands, and bit for bit against the same image
    coded alone."""
    from licos_amd import engine
    from licos_amd.layers import GDN
    g = torch.Generator().manual_seed(batch * 100 + cin + h)
    x = h16(torch.randn(batch, cin, h, w, generator=g))
    wt = h16(torch.randn(cin, cout, 5, 5, generator=g) * 0.04)
    b = torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x, wt, b, stride=2, padding=2, output_padding=1)
    gp, e = None, ops.EPI_NONE
    if igdn:
        m = GDN(cout, inverse=True).to(DEV)
        with torch.no_grad():
            m.gamma.add_((0.02 * torch.rand(cout, cout, generator=g)).to(DEV))
            beta, gamma = m.effective()
        gp, e = engine._packed_gdn(m), ops.EPI_IGDN
        norm = F.conv2d(ref * ref, gamma.cpu().reshape(cout, cout, 1, 1), beta.cpu())
        ref = ref * torch.sqrt(norm)
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    wp = ops.pack_conv_w_f16(wt.to(DEV), transposed=True)
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    assert ops.deconv_layouts(cin, h, w, cout) == (ops.EPI_IN_XSPLIT | ops.EPI_OUT_XSPLIT)
    out = ops.deconv5x5s2_f16(xb, wp, bp, gp, e, cin, cout)
    got = ops.blk16_to_nchw_f32(out, cout).cpu()
    assert got.shape == ref.shape
    tol = 4e-3 if igdn else 4.9e-4  # the fused norm runs on bf16 squares (test_fused_gdn_stage's bound)
    assert bool(((got - ref).abs() <= tol * ref.abs() + (2e-3 if igdn else 2e-5) * float(ref.abs().max())).all())
    one = ops.deconv5x5s2_f16(ops.nchw_f32_to_blk16(x[batch - 1:].to(DEV)), wp, bp, gp, e, cin, cout)
    assert torch.equal(one, out[batch - 1:])


@pytest.mark.parametrize("cin,cout,h,w,epi", [(128, 128, 32, 32, "igdn"), (128, 128, 40, 80, "none"), (192, 96, 18, 34, "relu"),
                                               (128, 128, 64, 64, "igdn"),
                                               # 16-pixel-wide maps: two images per pixel tile
                                               (192, 128, 16, 16, "igdn"), (128, 128, 40, 16, "relu")])
def test_deconv_xsplit_layouts_are_bit_exact(cin, cout, h, w, epi):
    """The x-split activation layout (rows as [even-x pixels][odd-x pixels], LICOS_EPI_IN/OUT_XSPLIT) is a pure
    re-ordering: every combination of input / output layout gives bit for bit the values of the plain blk16 call, in
    the 8-wave transposed-conv kernel and in the scatter-form last stage."""
    from licos_amd import engine
    from licos_amd.layers import GDN
    g = torch.Generator().manual_seed(cin + h)
    x = h16(torch.randn(2, cin, h, w, generator=g))
    wt = h16(torch.randn(cin, cout, 5, 5, generator=g) * 0.04)
    b = torch.randn(cout, generator=g)
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    wp = ops.pack_conv_w_f16(wt.to(DEV), transposed=True)
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    gp, e = None, ops.EPI_NONE
    if epi == "igdn":
        m = GDN(cout, inverse=True).to(DEV)
        with torch.no_grad():
            m.gamma.add_((0.02 * torch.rand(cout, cout, generator=g)).to(DEV))
        gp, e = engine._packed_gdn(m), ops.EPI_IGDN
    elif epi == "relu":
        e = ops.EPI_RELU
    lay = ops.deconv_layouts(cin, h, w, cout)
    assert lay == (ops.EPI_IN_XSPLIT | ops.EPI_OUT_XSPLIT)
    ref = ops.deconv5x5s2_f16(xb, wp, bp, gp, e, cin, cout)
    xs = ops.blk16_xsplit(xb)
    assert not torch.equal(xs, xb)
    o1 = ops.deconv5x5s2_f16(xb, wp, bp, gp, e | ops.EPI_OUT_XSPLIT, cin, cout)
    o2 = ops.deconv5x5s2_f16(xs, wp, bp, gp, e | ops.EPI_IN_XSPLIT, cin, cout)
    o3 = ops.deconv5x5s2_f16(xs, wp, bp, gp, e | ops.EPI_IN_XSPLIT | ops.EPI_OUT_XSPLIT, cin, cout)
    assert torch.equal(ops.blk16_xsplit(o1, inverse=True), ref)
    assert torch.equal(o2, ref)
    assert torch.equal(o3, o1)
    if cin in (128, 192):  # scatter-form last stage reading the x-split layout
        w3 = h16(torch.randn(cin, 3, 5, 5, generator=g) * 0.04)
        ws = ops.pack_deconv_w_scatter_f16(w3.to(DEV))
        b3 = torch.randn(3, generator=g).to(DEV)
        assert torch.equal(ops.deconv5x5s2_scatter_f16(xs, ws, b3, cin, 3, in_xsplit=True),
                           ops.deconv5x5s2_scatter_f16(xb, ws, b3, cin, 3))
    # stages that have no x-split form say so, and the flags are refused there
    assert ops.deconv_layouts(192, 8, 8, 128) == 0 and ops.deconv_layouts(128, 64, 64, 192) == 0 and ops.deconv_layouts(192, 16, 24, 128) == 0
    small = torch.zeros(1, 12, 8, 8, 16, device=DEV, dtype=torch.float16)
    with pytest.raises(ValueError):
        ops.deconv5x5s2_f16(small, ops.pack_conv_w_f16(torch.zeros(192, 128, 5, 5, device=DEV), transposed=True),
                            ops.pad_bias(torch.zeros(128, device=DEV), 128, DEV), None, ops.EPI_NONE | ops.EPI_OUT_XSPLIT, 192, 128)


@pytest.mark.parametrize("cin,cout,h,w", [(128, 3, 32, 32), (128, 1, 16, 16), (128, 3, 128, 128), (128, 2, 9, 40), (128, 4, 20, 36),
                                           (192, 3, 8, 33), (192, 1, 5, 3), (128, 3, 1, 1)])
def test_scatter_deconv_exact_operands(cin, cout, h, w):
    """Scatter-form last stage: same operands as torch-CPU conv_transpose2d; fp32 MFMA sums, then 2^-20 fixed-point
    accumulation of the <= 9 contributions per output pixel (|error| <= 9 * 2^-21), ragged tiles, every Cout."""
    g = torch.Generator().manual_seed(cin + cout + h)
    x = h16(torch.randn(2, cin, h, w, generator=g))
    wt = h16(torch.randn(cin, cout, 5, 5, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x, wt, b, stride=2, padding=2, output_padding=1)
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    ws = ops.pack_deconv_w_scatter_f16(wt.to(DEV))
    bd = b.to(DEV)
    out = ops.deconv5x5s2_scatter_f16(xb, ws, bd, cin, cout)
    assert out.shape == ref.shape
    assert float((out.cpu() - ref).abs().max()) < 9 * 2.0 ** -21 + 2e-5 * float(ref.abs().max())
    outc = ops.deconv5x5s2_scatter_f16(xb, ws, bd, cin, cout, clamp01=True)
    assert float((outc.cpu() - ref.clamp(0, 1)).abs().max()) < 9 * 2.0 ** -21 + 2e-5
    # integer accumulation: the result does not depend on the order in which waves arrive
    for _ in range(3):
        assert torch.equal(ops.deconv5x5s2_scatter_f16(xb, ws, bd, cin, cout), out)


def test_scatter_deconv_rejects_unsupported_shapes():
    with pytest.raises(ValueError):
        ops.pack_deconv_w_scatter_f16(torch.zeros(128, 5, 5, 5, device=DEV))
    xb = torch.zeros(1, 4, 8, 8, 16, device=DEV, dtype=torch.float16)  # 64 input channels: not instantiated
    ws = ops.pack_deconv_w_scatter_f16(torch.zeros(64, 3, 5, 5, device=DEV))
    with pytest.raises(ValueError):
        ops.deconv5x5s2_scatter_f16(xb, ws, torch.zeros(3, device=DEV), 64, 3)


@pytest.mark.parametrize("cin,cout,h,w", [(128, 3, 32, 32), (128, 1, 16, 16), (128, 3, 128, 128), (128, 2, 9, 40), (128, 3, 20, 36),
                                           (192, 3, 8, 33), (192, 1, 5, 3), (128, 3, 1, 1), (128, 3, 7, 130), (128, 1, 4, 300),
                                           (128, 3, 37, 128),
                                           # 5..16 bands: csrc/mfma_rows16.hip (256-column strips, two 16-pixel tiles per wave)
                                           (128, 13, 20, 36), (128, 13, 9, 35), (128, 16, 16, 34), (128, 5, 3, 300), (128, 13, 37, 256),
                                           (120, 13, 8, 16), (128, 9, 1, 1), (128, 13, 5, 510)])
def test_rows_deconv_exact_operands(cin, cout, h, w):
    """Row-walking last stage (csrc/mfma_rows.hip): same operands as torch-CPU conv_transpose2d, fp32 sums: ragged widths
    (dead lanes read zeros through the buffer bounds), maps wider than one 128-column strip (126 live columns per strip),
    row blocks that end inside a turn of the register ring, 1..3 bands, both channel counts."""
    g = torch.Generator().manual_seed(cin + cout + h)
    x = h16(torch.randn(2, cin, h, w, generator=g))
    wt = h16(torch.randn(cin, cout, 5, 5, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=2, output_padding=1).float()
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    ws = ops.pack_deconv_w_rows_f16(wt.to(DEV))
    bd = b.to(DEV)
    out = ops.deconv5x5s2_rows_f16(xb, ws, bd, cin, cout)
    assert out.shape == ref.shape
    assert float((out.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    outc = ops.deconv5x5s2_rows_f16(xb, ws, bd, cin, cout, clamp01=True)
    assert float((outc.cpu() - ref.clamp(0, 1)).abs().max()) < 4e-5
    assert torch.equal(ops.deconv5x5s2_rows_f16(xb, ws, bd, cin, cout), out)
    if w % 2 == 0:  # the x-split input layout is a pure re-ordering
        assert torch.equal(ops.deconv5x5s2_rows_f16(ops.blk16_xsplit(xb), ws, bd, cin, cout, in_xsplit=True), out)
    if cout > 4:  # against the LDS-patch form of the same stage (csrc/mfma_deconv.hip)
        if cin == 128:
            fw = ops.deconv5x5s2_fewch_f16(xb, ops.pack_deconv_w_fewch_f16(wt.to(DEV)), ops.pad_bias(b, cout, DEV), cin, cout)
            assert float((fw - out).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
        return
    # against the scatter form (2^-20 fixed-point sums of the same products)
    sc = ops.deconv5x5s2_scatter_f16(xb, ops.pack_deconv_w_scatter_f16(wt.to(DEV)), bd, cin, cout)
    assert float((sc - out).abs().max()) < 9 * 2.0 ** -21 + 2e-5 * float(ref.abs().max())


def test_rows_deconv_is_batch_and_block_invariant():
    """A tile's output bits do not depend on the batch it runs in (large batches walk 32-row blocks, small ones 8-row
    blocks): the per-pixel sum order is the same in both."""
    g = torch.Generator().manual_seed(5)
    x = h16(torch.randn(40, 128, 64, 64, generator=g))  # 40 x 2 row blocks of 32 < 2048 -> 8-row blocks ...
    wt = h16(torch.randn(128, 3, 5, 5, generator=g) * 0.05)
    b = torch.randn(3, generator=g).to(DEV)
    ws = ops.pack_deconv_w_rows_f16(wt.to(DEV))
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    one = ops.deconv5x5s2_rows_f16(xb[:1].contiguous(), ws, b, 128, 3)
    big = ops.deconv5x5s2_rows_f16(xb.repeat(64, 1, 1, 1, 1), ws, b, 128, 3)  # ... 2560 x 2 >= 2048 -> 32-row blocks
    assert torch.equal(big[:1], one) and torch.equal(big[40:41], one)


def test_rows16_deconv_is_batch_and_block_invariant():
    """The same for the 13-band form: 8-row blocks for small calls, 32-row blocks for large ones, identical bits."""
    g = torch.Generator().manual_seed(6)
    x = h16(torch.randn(10, 128, 64, 64, generator=g))
    wt = h16(torch.randn(128, 13, 5, 5, generator=g) * 0.05)
    b = torch.randn(13, generator=g).to(DEV)
    ws = ops.pack_deconv_w_rows_f16(wt.to(DEV))
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    one = ops.deconv5x5s2_rows_f16(xb[:1].contiguous(), ws, b, 128, 13)
    big = ops.deconv5x5s2_rows_f16(xb.repeat(60, 1, 1, 1, 1), ws, b, 128, 13)  # 600 x 2 row blocks of 32 >= 1024 -> 32-row blocks
    assert torch.equal(big[:1], one) and torch.equal(big[10:11], one)
    del big
    huge = ops.deconv5x5s2_rows_f16(xb.repeat(110, 1, 1, 1, 1), ws, b, 128, 13)  # 1100 x 1 row block of 64 >= 1024 -> 64-row blocks
    assert torch.equal(huge[:1], one) and torch.equal(huge[1099:1100], ops.deconv5x5s2_rows_f16(xb[9:10].contiguous(), ws, b, 128, 13))


def test_rows_deconv_rejects_unsupported_shapes():
    with pytest.raises(ValueError):
        ops.pack_deconv_w_rows_f16(torch.zeros(128, 4, 5, 5, device=DEV))
    xb = torch.zeros(1, 4, 8, 8, 16, device=DEV, dtype=torch.float16)  # 64 input channels: not instantiated
    ws = ops.pack_deconv_w_rows_f16(torch.zeros(64, 3, 5, 5, device=DEV))
    with pytest.raises(ValueError):
        ops.deconv5x5s2_rows_f16(xb, ws, torch.zeros(3, device=DEV), 64, 3)


@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("h,w", [(64, 64), (32, 32)])
def test_fused_gdn_epilogue(inverse, h, w):
    c = 128
    g = torch.Generator().manual_seed(h + inverse)
    sd = {}
    om._gdn_init(sd, "g.", c)
    sd["g.gamma"] = sd["g.gamma"] + 0.03 * torch.rand(c, c, generator=g)
    sd["g.beta"] = sd["g.beta"] * (0.5 + torch.rand(c, generator=g))
    m = licos_amd.GDN(c, inverse=inverse)
    m.load_state_dict({k[2:]: v for k, v in sd.items()})
    m = m.to(DEV)
    gp = engine._packed_gdn(m)
    x = h16(torch.randn(2, c, h, w, generator=g))
    wt = h16(torch.randn(c, c, 5, 5, generator=g) * 0.03)
    b = torch.randn(c, generator=g)
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    bp = ops.pad_bias(b.to(DEV), c, DEV)
    epi = ops.EPI_IGDN if inverse else ops.EPI_GDN
    if inverse:
        pre = F.conv_transpose2d(x, wt.transpose(0, 1).contiguous(), b, stride=2, padding=2, output_padding=1)
        wp = ops.pack_conv_w_f16(wt.transpose(0, 1).contiguous().to(DEV), transposed=True)
        out = ops.deconv5x5s2_f16(xb, wp, bp, gp, epi, c, c)
    else:
        pre = F.conv2d(x, wt, b, stride=2, padding=2)
        wp = ops.pack_conv_w_f16(wt.to(DEV))
        out = ops.conv5x5s2_f16(xb, wp, bp, gp, epi, c, c)
    ref = om.gdn(pre, sd, "g.", inverse=inverse)
    err = rel_err(ops.blk16_to_nchw_f32(out, c), ref)
    print(f"fused {'IGDN' if inverse else 'GDN'} rel err {err:.2e}")
    assert err < 4e-3  # bf16 gamma / x^2 operands + fp16 output


@pytest.mark.parametrize("cin,kind", [(3, "aid"), (1, "s2"), (13, "s2-merged")])
def test_model_fp16_matches_oracle_rates(cin, kind):
    sd = om.perturb_state(om.make_factorized_state(cin, quality=1, seed=42), seed=11, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, cin, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision("fp16")
    net.update(force=True)
    om.eb_update(sd)
    x = om.synthetic_tiles(2, cin, 256, seed=4, kind=kind)
    with torch.no_grad():
        out = net(x.to(DEV))
        comp = net.compress(x.to(DEV))
        dec = net.decompress(comp["strings"], comp["shape"])
        y16 = net.g_a(x.to(DEV))
    ref = om.forward(x, sd)
    y_err = rel_err(y16, ref["y"])
    sym16 = torch.round(y16.cpu() - sd["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1))
    mism = float((sym16 != om.eb_symbols(ref["y"], sd)).float().mean())
    bpp16, bpp = licos_amd.metrics.compute_bpp(out), om.compute_bpp(ref)
    psnr16 = licos_amd.metrics.compute_psnr(out["x_hat"].clamp(0, 1), x.to(DEV))
    psnr = om.compute_psnr(ref["x_hat"].clamp(0, 1), x)
    nbytes16 = sum(len(s) for s in comp["strings"][0])
    nbytes = sum(len(s) for s in om.compress(x, sd)["strings"][0])
    print(f"cin={cin}: y rel err {y_err:.2e}, symbol mismatch {mism:.4f}, bpp {bpp16:.4f} vs {bpp:.4f}, "
          f"PSNR {psnr16:.3f} vs {psnr:.3f} dB, bytes {nbytes16} vs {nbytes}")
    # ~3x what is measured (y 8.5e-4, mismatch 0.05-0.08 %, bpp < 1e-4 relative, PSNR < 0.001 dB, length within 4 bytes)
    assert y_err < 3e-3
    assert mism < 3e-3
    assert abs(bpp16 - bpp) < 1e-3 * bpp
    assert abs(psnr16 - psnr) < 0.01
    assert abs(nbytes16 - nbytes) < 1e-3 * nbytes + 16
    # self-consistency: the decoder reproduces forward()'s reconstruction from the bytes alone
    assert rel_err(dec["x_hat"], out["x_hat"].clamp(0, 1)) < 1e-6


@pytest.mark.parametrize("cin", [3, 1])
def test_default_fp16_path_runs_the_band_sized_end_stage_kernels(cin, monkeypatch):
    """The fp16 path of an RGB / single-band model goes through the in-place first stage (csrc/mfma_first.hip) and the
    row-walking last stage (csrc/mfma_rows.hip) - a silent fall-back to the older forms would keep every parity test green
    and lose 8 % of the headline - and agrees with those older forms (same operands, other fp32 summation orders)."""
    from licos_amd import synthetic
    net = licos_amd.get_model("bmshj2018-factorized", False, cin, 3).to(DEV).eval().set_precision("fp16")
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=7)
    x = om.synthetic_tiles(5, cin, 256, seed=11).to(DEV)
    calls = {"first": 0, "rows": 0}
    real_first, real_rows = ops.conv5x5s2_first_nchw_f16, ops.deconv5x5s2_rows_f16

    def first(*a, **k):
        calls["first"] += 1
        return real_first(*a, **k)

    def rows(*a, **k):
        calls["rows"] += 1
        return real_rows(*a, **k)

    monkeypatch.setattr(ops, "conv5x5s2_first_nchw_f16", first)
    monkeypatch.setattr(ops, "deconv5x5s2_rows_f16", rows)
    with torch.no_grad():
        y = net.g_a(x)
        xh = net.g_s(y)
    assert calls == {"first": 1, "rows": 1}
    monkeypatch.setattr(engine, "FIRST_ROWS", False)
    monkeypatch.setattr(engine, "ROWS_LAST", False)
    with torch.no_grad():
        y_old = net.g_a(x)
        xh_old = net.g_s(y)
    assert calls == {"first": 1, "rows": 1}  # the switches really select the older kernels
    assert rel_err(y, y_old) < 5e-3 and rel_err(xh, xh_old) < 1e-4
    # width not a multiple of 4: the layout-pass form of the same first stage
    monkeypatch.setattr(engine, "FIRST_ROWS", True)
    x2 = x[:2, :, :, :254].contiguous()
    with torch.no_grad():
        y2 = net.g_a(x2)
    assert calls["first"] == 1 and tuple(y2.shape[-2:]) == (16, 16)
    monkeypatch.setattr(engine, "FIRST_ROWS", False)
    with torch.no_grad():
        assert rel_err(y2, net.g_a(x2)) < 5e-3


def test_bench_scale_batch_invariants():
    """The bench workload (BASELINE configs[1]: q=3, 3-channel 256x256 tiles, fp16) at a batch far beyond what the
    oracle can follow, checked through size-independent properties: decode(encode(x)) equals forward()'s
    reconstruction bit for bit, the bytes do not depend on the pipeline chunking or on the run, every tile's string
    equals the string the same tile gets when coded alone, and the byte count agrees with the likelihood bpp."""
    from licos_amd import synthetic
    torch.manual_seed(3)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(DEV).eval().set_precision("fp16")
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=0)
    b = 1536
    x = synthetic.tiles(b, 3, 256, seed=77, device=DEV)
    with torch.no_grad():
        net.chunk = 512
        c1 = net.compress(x)
        d1 = net.decompress(c1["strings"], c1["shape"])["x_hat"]
        net.chunk = 1000  # ragged chunks
        c2 = net.compress(x)
        d2 = net.decompress([[bytes(s) for s in c2["strings"][0]]], c2["shape"])["x_hat"]
        fwd = net(x)
        pick = [0, 511, 512, 999, 1000, b - 1]  # chunk seams of both chunkings
        alone = [net.compress(x[i:i + 1])["strings"][0][0] for i in pick]
    s1, s2 = [bytes(s) for s in c1["strings"][0]], [bytes(s) for s in c2["strings"][0]]
    assert len(s1) == b and s1 == s2
    assert [s1[i] for i in pick] == [bytes(s) for s in alone]
    assert torch.equal(d1, d2)
    assert torch.equal(d1, fwd["x_hat"].clamp(0, 1))
    bpp_bytes = 8.0 * sum(len(s) for s in s1) / (b * 256 * 256)
    bpp_lik = licos_amd.metrics.compute_bpp(fwd)
    assert abs(bpp_bytes - bpp_lik) < 0.02 * bpp_lik + 0.002  # coder overhead: a few bytes per stream


@pytest.mark.parametrize("cin,h,w", [(3, 304, 464), (1, 16, 16), (13, 272, 336), (3, 1152, 1296)])
def test_fp16_whole_images_of_any_multiple_of_16(cin, h, w):
    """Images are not always 256x256 tiles: eval_script.py feeds whole granules (raw_utils.py:131: 1152x1296,
    2304x2592 - multiples of 16, not of 64).  Every stage kernel must handle ragged tiles; the fp16 path is checked
    against the fp32 path of the same module and against its own forward()."""
    from licos_amd import synthetic
    net = licos_amd.get_model("bmshj2018-factorized", False, cin, 2).to(DEV).eval()
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=5)
    size = -(-max(h, w) // 8) * 8
    x = om.synthetic_tiles(1, cin, size, seed=h + w)[..., :h, :w].contiguous().to(DEV)
    with torch.no_grad():
        ref = net(x)  # fp32 path (parity-tested against the oracle)
        net.set_precision("fp16")
        out = net(x)
        comp = net.compress(x)
        dec = net.decompress(comp["strings"], comp["shape"])["x_hat"]
    assert tuple(out["x_hat"].shape) == (1, cin, h, w) and tuple(comp["shape"]) == (h // 16, w // 16)
    assert torch.equal(dec, out["x_hat"].clamp(0, 1))
    psnr16 = licos_amd.metrics.compute_psnr(out["x_hat"].clamp(0, 1), x)
    psnr32 = licos_amd.metrics.compute_psnr(ref["x_hat"].clamp(0, 1), x)
    bpp16, bpp32 = licos_amd.metrics.compute_bpp(out), licos_amd.metrics.compute_bpp(ref)
    assert abs(psnr16 - psnr32) < 0.1 and abs(bpp16 - bpp32) < 0.01 * bpp32 + 1e-3, (psnr16, psnr32, bpp16, bpp32)


def test_fp16_codec_accepts_plain_and_edited_string_lists(monkeypatch):
    """compress() returns a list subclass that remembers its packed host buffer; decompress must give the
    same result for that object, for a plain list of the same bytes, and for a reordered plain list."""
    monkeypatch.setattr(ops, "HOST_CODER", "0")  # the chunk-pipelined device coder (batches this small go to the host otherwise)
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=11, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision("fp16")
    net.update(force=True)
    net.chunk = 2  # 3 chunks for 5 tiles: exercises the pipeline seams and the ragged last chunk
    x = om.synthetic_tiles(5, 3, 64, seed=8).to(DEV)
    with torch.no_grad():
        comp = net.compress(x)
        a = net.decompress(comp["strings"], comp["shape"])["x_hat"]
        plain = [bytes(s) for s in comp["strings"][0]]
        b = net.decompress([plain], comp["shape"])["x_hat"]
        c = net.decompress([plain[::-1]], comp["shape"])["x_hat"]
        net.chunk = 1024
        d = net.decompress([plain], comp["shape"])["x_hat"]
    assert torch.equal(a, b) and torch.equal(a, d)
    assert torch.equal(a, c.flip(0))
    assert all(isinstance(s, bytes) for s in comp["strings"][0])
    assert np.frombuffer(np.array(comp["strings"]), dtype=np.uint8).size > 0  # eval_utils.py:202-204 idiom


def test_fp16_codec_host_and_device_coder_agree(monkeypatch):
    """Same tiles through the chunk-pipelined device coder and through the host coder: identical strings, identical
    reconstruction; and each side decodes the other's strings."""
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=11, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision("fp16")
    net.update(force=True)
    net.chunk = 8
    x = om.synthetic_tiles(21, 3, 64, seed=5).to(DEV)
    with torch.no_grad():
        monkeypatch.setattr(ops, "HOST_CODER", "0")
        cd = net.compress(x)
        dd = net.decompress(cd["strings"], cd["shape"])["x_hat"]
        monkeypatch.setattr(ops, "HOST_CODER", "1")
        ch = net.compress(x)
        dh = net.decompress(ch["strings"], ch["shape"])["x_hat"]
        dcross = net.decompress([list(cd["strings"][0])], cd["shape"])["x_hat"]
        monkeypatch.setattr(ops, "HOST_CODER", "0")
        dcross2 = net.decompress([list(ch["strings"][0])], ch["shape"])["x_hat"]
    assert [bytes(s) for s in cd["strings"][0]] == [bytes(s) for s in ch["strings"][0]]
    assert tuple(cd["shape"]) == tuple(ch["shape"])
    assert torch.equal(dd, dh) and torch.equal(dd, dcross) and torch.equal(dd, dcross2)


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
def test_strings_and_tiles_do_not_depend_on_the_coder_placement(monkeypatch, precision):
    """Split placement of the serial coder (codec.host_share): whichever share of a call the host codes - nothing, the
    call's last tiles (compress) / first tiles (decompress), a share that ends inside a packed segment, everything - the
    byte strings are the same and so are the decoded tiles; also when decompress is handed a plain list, and when the
    tiles the host encoded ride behind the last packed segment in ONE device launch."""
    from licos_amd import codec
    from licos_amd.codec import PackedStrings
    monkeypatch.setattr(ops, "HOST_CODER", "auto")
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=5, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision(precision)
    net.update(force=True)
    net.chunk = 16
    x = om.synthetic_tiles(40, 3, 64, seed=33).to(DEV)
    share = {"enc": 0, "dec": 0}
    monkeypatch.setattr(codec.placement, "host_share", lambda batch, direction: min(batch, share[direction]))
    monkeypatch.setattr(codec.config, "host_sub", 1)  # sub-chunks of a few tiles: several of them per call
    monkeypatch.setattr(ops, "host_threads", lambda: 4)
    with torch.no_grad():
        c0 = net.compress(x)
        ref_strings = [bytes(s_) for s_ in c0["strings"][0]]
        ref = net.decompress(c0["strings"], c0["shape"])["x_hat"]
        assert isinstance(c0["strings"][0], PackedStrings) and len(c0["strings"][0].segments) == 3
        for enc, dec in ((12, 0), (12, 10), (12, 20), (3, 37), (40, 0), (40, 40), (0, 40), (0, 7)):
            share["enc"], share["dec"] = enc, dec
            c = net.compress(x)
            assert [bytes(s_) for s_ in c["strings"][0]] == ref_strings, (enc, dec)
            assert sum(n for _, n, _, _ in c["strings"][0].segments) == 40 - enc
            for strings in (c["strings"], [[bytes(s_) for s_ in c["strings"][0]]]):
                got = net.decompress(strings, c["shape"])["x_hat"]
                assert torch.equal(got, ref), (enc, dec)
        # chunks of 32: the 18 tiles left of the packed segment and the 12 the host encoded share one device launch
        net.chunk = 32
        share["enc"], share["dec"] = 12, 10
        c = net.compress(x)
        assert [bytes(s_) for s_ in c["strings"][0]] == ref_strings and len(c["strings"][0].segments) == 1
        assert torch.equal(net.decompress(c["strings"], c["shape"])["x_hat"], ref)
        net.chunk = 16
    # a corrupt stream among the host's tiles raises like one among the device's
    share["enc"], share["dec"] = 0, 10
    bad = list(ref_strings)
    bad[3] = bad[3][:8]
    with torch.no_grad(), pytest.raises(ValueError):
        net.decompress([bad], c0["shape"])


def test_mutated_packed_strings_decode_what_the_list_holds(monkeypatch):
    """The list compress() returns may be edited by the caller (a corruption experiment, a swap between tiles): the
    packed host buffers it remembers are then stale and decompress must decode the list's bytes, including when the
    replacement has the SAME length as the original (the round-1 check sampled three lengths per segment)."""
    from licos_amd.codec import PackedStrings
    monkeypatch.setattr(ops, "HOST_CODER", "0")  # PackedStrings is what the chunk-pipelined device coder returns
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=11, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision("fp16")
    net.update(force=True)
    net.chunk = 16
    x = om.synthetic_tiles(48, 3, 32, seed=21).to(DEV)
    with torch.no_grad():
        comp = net.compress(x)
        lst = comp["strings"][0]
        assert isinstance(lst, PackedStrings) and lst.still_packed()
        ref = net.decompress(comp["strings"], comp["shape"])["x_hat"]
        lens = {}
        pair = None
        for i, s in enumerate(lst):
            if len(s) in lens and lst[lens[len(s)]] != s:
                pair = (lens[len(s)], i)
                break
            lens[len(s)] = i
        assert pair is not None, "no two tiles with equal stream length; enlarge the batch"
        i, j = pair
        lst[j] = lst[i]                       # same length, different content, not at a sampled position necessarily
        assert not lst.still_packed()
        got = net.decompress(comp["strings"], comp["shape"])["x_hat"]
        plain = net.decompress([[bytes(s) for s in lst]], comp["shape"])["x_hat"]
        assert torch.equal(got, plain)
        assert torch.equal(got[j], ref[i]) and not torch.equal(got[j], ref[j])
        keep = [k for k in range(48) if k != j]
        assert torch.equal(got[keep], ref[keep])
        # deletions / insertions / reversal are caught too
        c2 = net.compress(x)
        l2 = c2["strings"][0]
        l2.reverse()
        assert not l2.still_packed()
        assert torch.equal(net.decompress(c2["strings"], c2["shape"])["x_hat"], ref.flip(0))
        c3 = net.compress(x)
        del c3["strings"][0][5]
        assert torch.equal(net.decompress(c3["strings"], c3["shape"])["x_hat"], torch.cat((ref[:5], ref[6:])))


@pytest.mark.parametrize("cin,h,w", [(3, 64, 64), (1, 32, 48), (3, 256, 256), (4, 18, 70), (2, 16, 16)])
def test_first_stage_space_to_depth_equals_conv(cin, h, w):
    """5x5 stride-2 conv over <=4 channels computed as a 3x3 stride-1 conv over the 2x2 space-to-depth image."""
    g = torch.Generator().manual_seed(cin * 7 + h)
    x = h16(torch.rand(2, cin, h, w, generator=g))
    wt = h16(torch.randn(128, cin, 5, 5, generator=g) * 0.2)
    b = torch.randn(128, generator=g)
    ref = F.conv2d(x, wt, b, stride=2, padding=2)
    xs = ops.nchw_f32_to_s2d_blk16(x.to(DEV))
    wp = ops.pack_conv_w_s2d_f16(wt.to(DEV))
    bp = ops.pad_bias(b.to(DEV), 128, DEV)
    out = ops.conv5x5s2_s2d_f16(xs, wp, bp, None, ops.EPI_NONE, cin, 128, h, w, out_nchw=True)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("cin,cout,h,w,epi", [(3, 128, 64, 64, "none"), (1, 128, 32, 48, "none"), (3, 128, 256, 256, "gdn"),
                                               (2, 128, 18, 70, "relu"), (3, 96, 16, 16, "none"), (3, 128, 50, 130, "gdn"),
                                               (1, 128, 256, 256, "gdn"), (3, 128, 37, 65, "none"), (3, 128, 37, 64, "gdn"),
                                               (1, 128, 19, 100, "relu"), (2, 128, 16, 260, "none")])
def test_first_stage_kernel_rows_equals_conv(cin, cout, h, w, epi):
    """First analysis stage with K steps = kernel rows over the interleaved zero-bordered fp16 image (csrc/mfma_first.hip,
    two 4-wave workgroups per CU): same operands as torch-CPU conv2d; ragged tiles, odd sizes, 1..3 bands, every epilogue;
    identical to the space-to-depth form up to the fp32 summation order."""
    from licos_amd.layers import GDN
    g = torch.Generator().manual_seed(cin * 7 + h)
    x = h16(torch.rand(3, cin, h, w, generator=g))
    wt = h16(torch.randn(cout, cin, 5, 5, generator=g) * 0.2)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=2).float()
    xi = ops.nchw_f32_to_hwc_pad_f16(x.to(DEV))
    rs = ((w + 4) * cin + 8 + 7) // 8 * 8
    img = xi[: 3 * (h + 4) * rs].view(3, h + 4, rs).float().cpu()
    want = torch.zeros(3, h + 4, rs)
    want[:, 2:h + 2, 2 * cin:(w + 2) * cin] = x.permute(0, 2, 3, 1).reshape(3, h, w * cin)
    assert torch.equal(img, want)  # interleaved, 2-pixel zero border, zero row tail
    wp = ops.pack_conv_w_first_f16(wt.to(DEV))
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    gp, e, tol = None, ops.EPI_NONE, 2e-3  # fp16 output
    if epi == "gdn":
        sd = {}
        om._gdn_init(sd, "g.", cout)
        sd["g.gamma"] = sd["g.gamma"] + 0.03 * torch.rand(cout, cout, generator=g)
        m = GDN(cout)
        m.load_state_dict({k[2:]: v for k, v in sd.items()})
        gp, e, tol = engine._packed_gdn(m.to(DEV)), ops.EPI_GDN, 4e-3
        ref = om.gdn(ref, sd, "g.")
    elif epi == "relu":
        e, ref = ops.EPI_RELU, ref.clamp_min(0)
    out = ops.conv5x5s2_first_f16(xi, wp, bp, gp, e, 3, cin, cout, h, w)
    got = ops.blk16_to_nchw_f32(out, cout)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < tol
    assert torch.equal(ops.conv5x5s2_first_f16(xi, wp, bp, gp, e, 3, cin, cout, h, w), out)
    if h % 2 == 0 and w % 2 == 0 and min(h, w) >= 32 and cout == 128:  # against the 3x3 form on the same operands
        xs = ops.nchw_f32_to_s2d_blk16(x.to(DEV))
        old = ops.conv5x5s2_s2d_f16(xs, ops.pack_conv_w_s2d_f16(wt.to(DEV)), bp, gp, e, cin, cout, h, w)
        assert rel_err(ops.blk16_to_nchw_f32(old, cout), got) < tol
    # a tile's bits do not depend on the batch it is in
    one = ops.conv5x5s2_first_f16(ops.nchw_f32_to_hwc_pad_f16(x[2:3].to(DEV)), wp, bp, gp, e, 1, cin, cout, h, w)
    assert torch.equal(one, out[2:3])
    if w % 4 == 0:  # the in-place form (fp32 rows by LDS-DMA, interleaved LDS to LDS): the same bits, no layout pass
        xd = x.to(DEV)
        raw = ops.conv5x5s2_first_nchw_f16(xd, wp, bp, gp, e, cout)
        assert torch.equal(raw, out)
        assert torch.equal(ops.conv5x5s2_first_nchw_f16(xd[1:2].contiguous(), wp, bp, gp, e, cout), out[1:2])
    else:
        with pytest.raises(ValueError):
            ops.conv5x5s2_first_nchw_f16(x.to(DEV), wp, bp, gp, e, cout)


@pytest.mark.parametrize("cin,cout,h,w,epi", [(13, 128, 512, 512, "gdn"), (13, 128, 256, 256, "gdn"), (13, 128, 64, 64, "none"),
                                               (13, 128, 70, 100, "gdn"), (5, 128, 37, 68, "relu"), (16, 96, 48, 96, "none"),
                                               (13, 128, 16, 260, "gdn"), (8, 128, 33, 64, "none")])
def test_first_stage_in_place_for_many_bands(cin, cout, h, w, epi):
    """g_a[0] for 5..16 bands on the NCHW fp32 image in place (csrc/mfma_first16.hip: the 13 merged Sentinel-2 bands of
    configs 3 - 5): same operands as torch-CPU conv2d (+ GDN / ReLU); bit-identical to the route it replaces - blk16 layout
    pass + the 8-wave kernel - wherever that route runs the 8-wave kernel (same MFMA order); ragged tiles, odd heights,
    runs of tiles that end inside an image; a tile's bits do not depend on the batch it is in."""
    from licos_amd.layers import GDN
    g = torch.Generator().manual_seed(cin * 11 + h)
    nb = 2 if h * w > 100000 else 3
    x = h16(torch.rand(nb, cin, h, w, generator=g))
    wt = h16(torch.randn(cout, cin, 5, 5, generator=g) * 0.1)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=2, padding=2).float()
    wp = ops.pack_conv_w_f16(wt.to(DEV))
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    gp, e, tol = None, ops.EPI_NONE, 2e-3  # fp16 output
    if epi == "gdn":
        sd = {}
        om._gdn_init(sd, "g.", cout)
        sd["g.gamma"] = sd["g.gamma"] + 0.03 * torch.rand(cout, cout, generator=g)
        m = GDN(cout)
        m.load_state_dict({k[2:]: v for k, v in sd.items()})
        gp, e, tol = engine._packed_gdn(m.to(DEV)), ops.EPI_GDN, 4e-3
        ref = om.gdn(ref, sd, "g.")
    elif epi == "relu":
        e, ref = ops.EPI_RELU, ref.clamp_min(0)
    xd = x.to(DEV)
    out = ops.conv5x5s2_first16_nchw_f16(xd, wp, bp, gp, e, cout)
    got = ops.blk16_to_nchw_f32(out, cout)
    assert got.shape == ref.shape
    assert rel_err(got, ref) < tol
    assert torch.equal(ops.conv5x5s2_first16_nchw_f16(xd, wp, bp, gp, e, cout), out)  # deterministic
    old = ops.conv5x5s2_f16(ops.nchw_f32_to_blk16(xd), wp, bp, gp, e, cin, cout)
    assert rel_err(ops.blk16_to_nchw_f32(old, cout), got) < tol
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    if ho % 16 == 0 and wo >= 32 and cout == 128 and epi == "gdn":
        # the replaced route is the 8-wave kernel with the tile epilogue there (accumulators start at the bias, as here): the
        # same bits.  (Without a norm that kernel adds the bias last: equal up to the fp32 summation order, checked above.)
        assert torch.equal(old, out)
    assert torch.equal(ops.conv5x5s2_first16_nchw_f16(xd[1:2].contiguous(), wp, bp, gp, e, cout), out[1:2])
    with pytest.raises(ValueError):
        ops.conv5x5s2_first16_nchw_f16(xd[:, :, :, : w - 2].contiguous(), wp, bp, gp, e, cout)  # W % 4


def test_models_with_13_bands_take_the_in_place_first_stage(monkeypatch):
    """The default fp16 path of a 13-band model runs g_a[0] through licos_conv5x5s2_first16_nchw_f16, and its output equals
    the blk16-route's (LICOS_FIRST16=0) bit for bit."""
    torch.manual_seed(3)
    net = licos_amd.get_model("bmshj2018-factorized", False, 13, 1).to(DEV).eval().set_precision("fp16")
    x = torch.rand(2, 13, 256, 256, device=DEV)
    calls = []
    real = ops.conv5x5s2_first16_nchw_f16
    monkeypatch.setattr(ops, "conv5x5s2_first16_nchw_f16", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        y = net.g_a(x)
        assert calls
        monkeypatch.setattr(engine, "FIRST16", False)
        y_old = net.g_a(x)
    assert torch.equal(y, y_old)


def test_first_stage_kernel_rows_rejects_unsupported_shapes():
    with pytest.raises(ValueError):
        ops.pack_conv_w_first_f16(torch.zeros(128, 4, 5, 5, device=DEV))
    with pytest.raises(ValueError):
        ops.pack_conv_w_first_f16(torch.zeros(192, 3, 5, 5, device=DEV))
    with pytest.raises(ValueError):
        ops.nchw_f32_to_hwc_pad_f16(torch.zeros(1, 4, 16, 16, device=DEV))
    xi = ops.nchw_f32_to_hwc_pad_f16(torch.zeros(1, 3, 16, 16, device=DEV))
    wp = ops.pack_conv_w_first_f16(torch.zeros(128, 3, 5, 5, device=DEV))
    with pytest.raises(ValueError):  # a buffer made for a smaller shape
        ops.conv5x5s2_first_f16(xi, wp, torch.zeros(128, device=DEV), None, ops.EPI_NONE, 1, 3, 128, 64, 64)


def _relu_state(cin, seed):
    """A factorized-relu state: the factorized state without the GDN entries."""
    sd = om.perturb_state(om.make_factorized_state(cin, quality=1, seed=42), seed=seed, y_gain=20.0)
    return {k: v for k, v in sd.items() if ".beta" not in k and ".gamma" not in k}


def _relu_forward(x, sd):
    y = x
    for li in range(4):
        y = F.conv2d(y, sd[f"g_a.{2 * li}.weight"], sd[f"g_a.{2 * li}.bias"], stride=2, padding=2)
        if li < 3:
            y = F.relu(y)
    return y


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-5), ("fp16", 2e-3)])
def test_factorized_relu_variant(precision, tol):
    """bmshj2018-factorized-relu (allowed by licos/model_utils.py:20-24): ReLU fused as a conv flag (fp32) or
    as the MFMA epilogue (fp16)."""
    sd = _relu_state(3, 3)
    net = licos_amd.get_model("bmshj2018-factorized-relu", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision(precision)
    net.update(force=True)
    x = om.synthetic_tiles(2, 3, 128, seed=5)
    with torch.no_grad():
        y = net.g_a(x.to(DEV))
        comp = net.compress(x.to(DEV))
        dec = net.decompress(comp["strings"], comp["shape"])
        out = net(x.to(DEV))
    assert rel_err(y, _relu_forward(x, sd)) < tol
    assert rel_err(dec["x_hat"], out["x_hat"].clamp(0, 1)) < 1e-6


def test_quality6_topology_fp16():
    """q6-8: N = 192, M = 320 (CompressAI zoo cfgs): 6- and 10-tile accumulators, 20 cin chunks in g_s[0]."""
    sd = om.perturb_state(om.make_factorized_state(3, quality=6, seed=42), seed=9, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 6)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision("fp16")
    net.update(force=True)
    om.eb_update(sd)
    x = om.synthetic_tiles(2, 3, 128, seed=1)
    with torch.no_grad():
        out = net(x.to(DEV))
        y16 = net.g_a(x.to(DEV))
        comp = net.compress(x.to(DEV))
        dec = net.decompress(comp["strings"], comp["shape"])
    ref = om.forward(x, sd)
    assert tuple(y16.shape) == (2, 320, 8, 8)
    assert rel_err(y16, ref["y"]) < 1e-2
    assert abs(licos_amd.metrics.compute_bpp(out) - om.compute_bpp(ref)) < 0.01 * om.compute_bpp(ref)
    assert rel_err(dec["x_hat"], out["x_hat"].clamp(0, 1)) < 1e-6
    net.set_precision("fp32")
    with torch.no_grad():
        y32 = net.g_a(x.to(DEV))
    assert rel_err(y32, ref["y"]) < 1e-5


@pytest.mark.parametrize("cin,cout,h,w,relu", [(192, 128, 32, 32, True), (128, 192, 32, 32, True), (128, 192, 8, 8, True),
                                                (20, 40, 9, 21, False)])
def test_conv3x3_stage_exact_operands(cin, cout, h, w, relu):
    """3x3 stride-1 conv (hyperprior h_a[0] / h_s[4]) on the MFMA path, with the ReLU epilogue."""
    g = torch.Generator().manual_seed(cin + cout + h)
    x = h16(torch.randn(2, cin, h, w, generator=g))
    wt = h16(torch.randn(cout, cin, 3, 3, generator=g) * 0.05)
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(torch.abs(x), wt, b, stride=1, padding=1)
    if relu:
        ref = F.relu(ref)
    xb = ops.nchw_f32_to_blk16(x.to(DEV), abs_input=True)
    wp = ops.pack_conv3x3_w_f16(wt.to(DEV))
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    out = ops.conv3x3s1_f16(xb, wp, bp, None, ops.EPI_RELU if relu else ops.EPI_NONE, cin, cout, out_nchw=True)
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("b,c,h,w", [(70, 192, 16, 16), (3, 20, 4, 12), (65, 32, 6, 6), (130, 16, 16, 2)])
def test_dequantise_to_blk16_layouts_agree(b, c, h, w):
    """symbols [position][stream] -> y_hat: the coalesced blk16 kernel (H*W % 16 == 0), the element-wise blk16 kernel
    and the NCHW fp32 output hold the same values (sym + median, one fp16 rounding in blk16)."""
    g = torch.Generator().manual_seed(b + c)
    sym = torch.randint(-300, 300, (c * h * w, b), generator=g, dtype=torch.int32).to(DEV)
    med = torch.randn(c, generator=g).to(DEV)
    ref = ops.eb_dequantize(sym, 1, b, med, b, c, h, w)                       # NCHW fp32
    blk = torch.full((b, (c + 15) // 16, h, w, 16), 7.0, device=DEV, dtype=torch.float16)
    ops.eb_dequantize(sym, 1, b, med, b, c, h, w, want_nchw=False, blk16=blk)
    assert torch.equal(ops.blk16_to_nchw_f32(blk, c), ref.half().float())
    if c % 16:  # the channels that pad the last chunk are zeros, whatever the buffer held
        full = blk.permute(0, 1, 4, 2, 3).reshape(b, -1, h, w)
        assert float(full[:, c:].abs().max()) == 0.0 or (h * w) % 16 != 0
    want = sym.t().reshape(b, c, h, w).float() + med.view(1, c, 1, 1)
    assert torch.equal(ref, want)


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
def test_host_tiles_as_16_bit_symbols(monkeypatch, precision):
    """The host's tiles cross PCIe as 16-bit symbols (codec.config.sym16: licos_eb_symbols16 / licos_eb_dequantize16 and the
    host coder's 16-bit entries): the strings and tiles of the 32-bit form; latents beyond 16 bits make compress fall back
    to it for the call and decompress for the sub-chunk concerned."""
    from licos_amd import codec
    monkeypatch.setattr(ops, "HOST_CODER", "1")
    monkeypatch.setattr(ops, "host_threads", lambda: 2)
    monkeypatch.setattr(codec.config, "host_sub", 2)  # sub-chunks of 4 tiles
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=6, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision(precision)
    net.update(force=True)
    x = om.synthetic_tiles(22, 3, 128, seed=34).to(DEV)  # 8 x 8 latents: the blk16 form of the 16-bit dequantise applies
    calls = {"enc16": 0, "dec16": 0, "dec32": 0}
    for name, key in (("rans_encode_host_sym16", "enc16"), ("rans_decode_host_sym16", "dec16"), ("rans_decode_host", "dec32")):
        real = getattr(ops, name)
        monkeypatch.setattr(ops, name, lambda *a, _r=real, _k=key, **k: (calls.__setitem__(_k, calls[_k] + 1), _r(*a, **k))[1])
    with torch.no_grad():
        monkeypatch.setattr(codec.config, "sym16", False)
        c0 = net.compress(x)
        ref = net.decompress(c0["strings"], c0["shape"])["x_hat"]
        assert calls["enc16"] == 0 and calls["dec16"] == 0 and calls["dec32"] > 0
        monkeypatch.setattr(codec.config, "sym16", True)
        calls["dec32"] = 0
        c = net.compress(x)
        got = net.decompress(c["strings"], c["shape"])["x_hat"]
        assert calls["enc16"] > 0 and calls["dec16"] > 0 and calls["dec32"] == 0
        assert [bytes(s_) for s_ in c["strings"][0]] == [bytes(s_) for s_ in c0["strings"][0]]
        assert torch.equal(got, ref)
        # latents up to +-10^5: compress finds the flag of a sub-chunk raised and codes the call with 32-bit symbols; decompress
        # decodes the sub-chunks concerned twice (16-bit attempt, then 32-bit)
        ymax = float(net.g_a(x).abs().max())
        net.g_a[6].weight.mul_(1.0e5 / ymax)
        net.g_a[6].bias.mul_(1.0e5 / ymax)
        assert float(net.g_a(x).abs().max()) > 5.0e4
        monkeypatch.setattr(codec.config, "sym16", False)
        b0 = net.compress(x)
        bref = net.decompress(b0["strings"], b0["shape"])["x_hat"]
        monkeypatch.setattr(codec.config, "sym16", True)
        calls.update(enc16=0, dec16=0, dec32=0)
        b1 = net.compress(x)
        assert [bytes(s_) for s_ in b1["strings"][0]] == [bytes(s_) for s_ in b0["strings"][0]]
        assert max(len(s_) for s_ in b1["strings"][0]) > 3 * max(len(s_) for s_ in c["strings"][0])
        bgot = net.decompress(b1["strings"], b1["shape"])["x_hat"]
        assert calls["dec16"] > 0 and 1 <= calls["dec32"] <= calls["dec16"]
        assert torch.allclose(bgot, bref, rtol=0, atol=0, equal_nan=True)  # (latents this large overflow the synthesis)


@pytest.mark.parametrize("batch,cin,cout,h,w", [(3, 128, 192, 32, 32), (1, 128, 192, 32, 32), (4, 128, 320, 32, 32), (2, 128, 192, 40, 72),
                                                (2, 128, 128, 64, 64)])
def test_quantiser_in_the_last_analysis_stage(batch, cin, cout, h, w):
    """licos_conv5x5s2_f16_symbols (CompressAI EntropyBottleneck.compress's `round(x - medians).int()` on g_a's
    output, /root/reference/eval_utils.py:199-204): the symbols of the convolution with the quantiser in its epilogue are
    EXACTLY those of the same convolution's fp32 output followed by licos_eb_quantize - ties included, the accumulators are
    the same - in the coder's [stream][position] layout; two images per pixel tile, an odd batch, M = 320, a ragged map."""
    g = torch.Generator().manual_seed(cout + h + batch)
    x = h16(torch.randn(batch, cin, h, w, generator=g))
    wt = h16(torch.randn(cout, cin, 5, 5, generator=g) * 0.08)
    b = torch.randn(cout, generator=g)
    med = (torch.randn(cout, generator=g) * 0.7).to(DEV)
    med[::7] = 0.5  # medians on the grid's half steps put latents on rounding ties
    xb = ops.nchw_f32_to_blk16(x.to(DEV))
    wp = ops.pack_conv_w_f16(wt.to(DEV))
    bp = ops.pad_bias(b.to(DEV), cout, DEV)
    y = ops.conv5x5s2_f16(xb, wp, bp, None, ops.EPI_NONE, cin, cout, out_nchw=True)
    ho, wo = y.shape[2], y.shape[3]
    want = torch.empty((batch, cout * ho * wo), device=DEV, dtype=torch.int32)
    ops.eb_quantize(y, med, "symbols", symbols=want, sym_stride_b=cout * ho * wo, sym_stride_i=1)
    got = ops.conv5x5s2_f16_symbols(xb, wp, bp, med, cin, cout)
    assert got.dtype == torch.int32 and tuple(got.shape) == (batch, cout, ho, wo)
    assert torch.equal(got.view(batch, -1), want)
    assert int(got.abs().max()) > 3  # (the operands do spread the symbols)
    # ... and the oracle's quantiser on the same latents
    ref = torch.round(y.cpu() - med.cpu().view(1, -1, 1, 1)).to(torch.int32)
    assert torch.equal(got.cpu(), ref)


def test_fused_quantiser_and_stream_major_symbols_change_no_byte(monkeypatch):
    """The chunk pipeline with the quantiser in g_a[6]'s epilogue and the plane encoder reading [stream][position] symbols
    (codec.config.eb_stream_major) writes the strings of the unfused pipeline (transposing quantise kernel, [position]
    [stream] symbols), tile for tile - several chunks, a ragged last chunk, an odd chunk size."""
    from licos_amd import codec
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=7)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision("fp16")
    net.update(force=True)
    monkeypatch.setattr(ops, "HOST_CODER", "0")
    x = om.synthetic_tiles(23, 3, 64, seed=5).to(DEV)
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(codec.config, "eb_stream_major", fused)
        for chunk in (7, 16):
            net.chunk = chunk
            with torch.no_grad():
                c = net.compress(x)
            out[(fused, chunk)] = [bytes(s) for s in c["strings"][0]]
    assert out[(True, 7)] == out[(False, 7)] == out[(True, 16)] == out[(False, 16)]
    assert len(set(out[(True, 7)])) > 1 and min(len(s) for s in out[(True, 7)]) > 8


def test_stream_major_coders_agree_with_position_major_ones_at_scale():
    """The prefetching plane encoder and the instruction-counted plane decoder (round 5) on 2048 streams of real latents, three
    times over: words and symbols equal to the [position][stream] forms' every time.  (The first form of the encoder's
    prefetch ring coded a wave's streams from stale registers once in a dozen launches of this size - and never in the small
    batches of the other tests: loads are slower when the whole chip is asking.)"""
    from licos_amd import checkpoint, synthetic
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(DEV).eval().set_precision("fp16")
    wf = os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar")
    if os.path.exists(wf):
        checkpoint.load_checkpoint(wf, net)
    else:
        with torch.no_grad():
            synthetic.make_trained_like(net, seed=3)
    net.update(force=True)
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, table = eb.coder_tables()
    B, plane = 2048, 256
    nsym = cdf.shape[0] * plane
    x = synthetic.tiles(B, 3, 256, seed=9, device=DEV)
    with torch.no_grad():
        sy = engine.run_chain_fp16(net.g_a, x=x, symbols=(eb.medians_vec(), None))
    sym_sm = sy.reshape(B, nsym).contiguous()
    sym_pm = sym_sm.t().contiguous()
    del x, sy
    cap = nsym // 2 + 64
    w0, n0, s0 = ops.rans_encode_batch(sym_pm, 1, B, nsym, plane, cdf, cdf_len, offset, table, cap, B)
    assert int(s0) == 0
    byte_off = torch.zeros(B + 1, device=DEV, dtype=torch.int64)
    byte_off[1:] = torch.cumsum(n0.to(torch.int64) * 4, 0)
    data = ops.rans_compact(w0, n0, byte_off, int(byte_off[-1]))
    mx = int(n0.max())
    for _ in range(3):
        w1, n1, s1 = ops.rans_encode_batch(sym_sm, nsym, 1, nsym, plane, cdf, cdf_len, offset, table, cap, B)
        assert int(s1) == 0 and torch.equal(n0, n1)
        assert torch.equal(torch.where(torch.arange(cap, device=DEV)[:, None] >= cap - n0[None, :], w0, 0)[cap - mx:],
                           torch.where(torch.arange(cap, device=DEV)[:, None] >= cap - n1[None, :], w1, 0)[cap - mx:])
        out = torch.empty((B, nsym), device=DEV, dtype=torch.int32)
        st = ops.rans_decode_batch(data, byte_off, nsym, 1, nsym, plane, cdf, cdf_len, offset, out, B, off_offset=0)
        assert int(st) == 0 and torch.equal(out, sym_sm)
