"""GPU parity of the scale hyperprior (BASELINE config 5) on the fp32 path, stage by stage on
identical inputs (see test_gpu_parity.py for why end-to-end symbols may differ on rounding ties)."""
import numpy as np
import pytest
import torch

import licos_amd
from oracle import model as om
from oracle import rans

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["device-coder", "host-coder"])
def _coder_placement(request, monkeypatch):
    """Every test runs twice: rANS on the GPU (one lane per stream) and on the host cores (licos_rans_*_host) - the
    product picks by batch size (ops.host_coder_preferred), the bytes must not depend on it."""
    from licos_amd import ops as _ops
    monkeypatch.setattr(_ops, "HOST_CODER", "0" if request.param == "device-coder" else "1")
DEV = "cuda:0"


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _state(cin, seed):
    sd = om.perturb_state(om.make_hyperprior_state(cin, quality=1, seed=42), seed=seed, y_gain=40.0)
    g = torch.Generator().manual_seed(seed)
    # random-init h_s gives scales < 0.11 everywhere (index 0 only): spread them over the table
    sd["h_s.4.weight"] = sd["h_s.4.weight"] * 40
    sd["h_s.4.bias"] = 6 * torch.rand(sd["h_s.4.bias"].shape, generator=g) - 1.0
    sd["h_a.4.weight"] = sd["h_a.4.weight"] * 20
    return sd


@pytest.mark.parametrize("cin,size", [(3, 128), (13, 64), (13, 512)])  # (13, 512): BASELINE config 5's tile, one of them
def test_hyperprior_fp32_stagewise(cin, size):
    sd = _state(cin, 5)
    net = licos_amd.get_model("bmshj2018-hyperprior", False, cin, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    net.update(force=True)
    om.hyper_update(sd)
    assert torch.equal(net.gaussian_conditional._quantized_cdf.cpu(), sd["gaussian_conditional._quantized_cdf"])
    x = om.synthetic_tiles(1 if size >= 512 else 2, cin, size, seed=6, kind="aid" if cin == 3 else "s2-merged")
    ref = om.hyper_forward(x, sd)
    gc, eb = net.gaussian_conditional, net.entropy_bottleneck
    with torch.no_grad():
        y = net.g_a(x.to(DEV))
        assert rel_err(y, ref["y"]) < 1e-5
        z = net.h_a(ref["y"].to(DEV))                       # |y| fused into the first conv
        assert rel_err(z, ref["z"]) < 1e-5
        z_hat, z_lik = eb(ref["z"].to(DEV))
        assert torch.equal(z_hat.cpu(), ref["z_hat"])
        rl = ref["likelihoods"]["z"]
        assert bool(((z_lik.cpu() - rl).abs() <= 1e-5 * rl + 3e-7).all())
        scales = net.h_s(ref["z_hat"].to(DEV))
        assert rel_err(scales, ref["scales"]) < 1e-5
        s_ref = ref["scales"].to(DEV)
        y_hat, y_lik = gc(ref["y"].to(DEV), s_ref)
        assert torch.equal(y_hat.cpu(), ref["y_hat"])
        rl = ref["likelihoods"]["y"]
        assert bool(((y_lik.cpu() - rl).abs() <= 1e-5 * rl + 3e-7).all())
        idx = gc.build_indexes(s_ref)
        ref_idx = om.gc_build_indexes(ref["scales"], sd)
        assert torch.equal(idx.cpu(), ref_idx)
        assert int(ref_idx.max()) > 20  # the fixture really exercises many table rows
        # coder with per-element table rows: bytes identical to the oracle's on identical inputs
        y_strings = gc.compress(ref["y"].to(DEV), gc.build_indexes_interleaved(s_ref))
        p = "gaussian_conditional."
        for i in range(x.shape[0]):
            want = rans.encode_with_indexes(ref["y_hat"][i].int().reshape(-1).numpy(), ref_idx[i].reshape(-1).numpy(),
                                            sd[p + "_quantized_cdf"].numpy(), sd[p + "_cdf_length"].numpy(),
                                            sd[p + "_offset"].numpy())
            assert y_strings[i] == want
        y_dec = gc.decompress(y_strings, gc.build_indexes_interleaved(s_ref), tuple(ref["y"].shape[1:]))
        assert torch.equal(y_dec.cpu(), ref["y_hat"])
        # module API end to end: self-consistent, two string lists, shape = z's spatial size
        out = net(x.to(DEV))
        comp = net.compress(x.to(DEV))
        dec = net.decompress(comp["strings"], comp["shape"])
    assert len(comp["strings"]) == 2 and tuple(comp["shape"]) == (size // 64, size // 64)
    assert rel_err(dec["x_hat"], out["x_hat"].clamp(0, 1)) < 1e-6
    ref_c = om.hyper_compress(x, sd)
    n_gpu = sum(len(s) for lst in comp["strings"] for s in lst)
    n_ref = sum(len(s) for lst in ref_c["strings"] for s in lst)
    assert abs(n_gpu - n_ref) <= 0.01 * n_ref + 16
    bpp = licos_amd.metrics.compute_bpp(out)
    assert abs(bpp - om.compute_bpp(ref)) < 2e-3 * om.compute_bpp(ref)


def test_hyperprior_fp16_config5_size():
    """BASELINE configs[4] shape: 13-band 512x512 tiles, all four transforms on the fp16 MFMA path (h_a / h_s: 3x3
    stride-1 stages as one phase of the transposed-conv kernel, ReLU epilogues, |y| folded into the layout conversion).
    Rates and quality track the oracle; decode(encode(x)) is self-consistent."""
    sd = _state(13, 7)
    net = licos_amd.get_model("bmshj2018-hyperprior", False, 13, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).eval().set_precision("fp16")
    net.update(force=True)
    om.hyper_update(sd)
    x = om.synthetic_tiles(1, 13, 512, seed=2, kind="s2-merged")
    with torch.no_grad():
        out = net(x.to(DEV))
        comp = net.compress(x.to(DEV))
        dec = net.decompress(comp["strings"], comp["shape"])
    ref = om.hyper_forward(x, sd)
    assert tuple(comp["shape"]) == (8, 8) and tuple(dec["x_hat"].shape) == (1, 13, 512, 512)
    assert rel_err(dec["x_hat"], out["x_hat"].clamp(0, 1)) < 1e-6
    bpp, bpp_ref = licos_amd.metrics.compute_bpp(out), om.compute_bpp(ref)
    psnr = licos_amd.metrics.compute_psnr(out["x_hat"].clamp(0, 1), x.to(DEV))
    psnr_ref = om.compute_psnr(ref["x_hat"].clamp(0, 1), x)
    print(f"hyperprior fp16 13x512x512: bpp {bpp:.4f} vs {bpp_ref:.4f}, PSNR {psnr:.3f} vs {psnr_ref:.3f}")
    assert abs(bpp - bpp_ref) < 0.01 * bpp_ref and abs(psnr - psnr_ref) < 0.1


@pytest.mark.parametrize("model", ["bmshj2018-factorized", "bmshj2018-factorized-relu", "bmshj2018-hyperprior"])
@pytest.mark.parametrize("quality", [2, 7])
def test_every_zoo_model_and_width_round_trips(model, quality):
    """Both channel tiers (q1-5: N=128, M=192; q6-8: N=192, M=320) of every model LICOS's get_model admits
    (model_utils.py:20-24), both precisions: compress -> decompress reproduces forward()'s reconstruction."""
    from licos_amd import synthetic
    net = licos_amd.get_model(model, False, 3, quality).to(DEV).eval()
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=quality)
    x = torch.round(torch.rand(2, 3, 128, 192, device=DEV) * 255) / 255
    for prec in ("fp32", "fp16"):
        net.set_precision(prec)
        with torch.no_grad():
            out = net(x)
            comp = net.compress(x)
            dec = net.decompress(comp["strings"], comp["shape"])["x_hat"]
        assert bool(torch.isfinite(out["x_hat"]).all())
        assert float((dec - out["x_hat"].clamp(0, 1)).abs().max()) < (1e-5 if prec == "fp32" else 1e-6), (model, quality, prec)


def test_hyper_codec_host_share_changes_nothing(monkeypatch):
    """compress_hyper / decompress_hyper with the call's last tiles coded by the host cores (codec.hyper_host_share: y
    symbols and table rows, z symbols over PCIe, the host coder with explicit per-symbol rows): the same y and z strings
    as the all-device call, and decompress - whatever it decodes where, also when its host share is not the encoder's -
    returns the same tiles."""
    from licos_amd import codec, ops
    monkeypatch.setattr(ops, "HOST_CODER", "0")  # the large-batch pipeline whatever the batch
    torch.manual_seed(11)
    net = licos_amd.get_model("bmshj2018-hyperprior", False, 3, 1).to(DEV).eval().set_precision("fp16")
    with torch.no_grad():
        licos_amd.synthetic.make_trained_like(net, seed=3)
    net.update(force=True)
    net.chunk = 8 * 16  # _chunk_for scales by the tile area: 8 tiles of 128^2 per pipeline chunk
    x = licos_amd.synthetic.tiles(21, 3, 128, seed=9, device=DEV)
    share = {"enc": 0, "dec": 0}
    monkeypatch.setattr(codec.placement, "hyper_host_share", lambda batch, direction="enc": share[direction])
    with torch.no_grad():
        c0 = net.compress(x)
        ref = net.decompress(c0["strings"], c0["shape"])["x_hat"]
        ys0, zs0 = [bytes(s_) for s_ in c0["strings"][0]], [bytes(s_) for s_ in c0["strings"][1]]
        monkeypatch.setattr(ops, "host_threads", lambda: 2)
        for enc, dec in ((5, 5), (9, 3), (3, 9), (0, 6), (6, 0), (7, 7), (21, 21), (21, 0), (0, 21)):
            share["enc"], share["dec"] = enc, dec
            c = net.compress(x)
            assert [bytes(s_) for s_ in c["strings"][0]] == ys0 and [bytes(s_) for s_ in c["strings"][1]] == zs0, (enc, dec)
            assert sum(n for _, n, _, _ in c["strings"][0].segments) == 21 - enc
            assert torch.equal(net.decompress(c["strings"], c["shape"])["x_hat"], ref), (enc, dec)
            plain = [[bytes(s_) for s_ in lst] for lst in c["strings"]]
            assert torch.equal(net.decompress(plain, c["shape"])["x_hat"], ref), (enc, dec)
        # a truncated y string among the host's tiles is reported like one among the device's
        share["enc"], share["dec"] = 0, 6
        bad = [[bytes(s_) for s_ in lst] for lst in c0["strings"]]
        bad[0][19] = bad[0][19][:8]
        with pytest.raises(ValueError):
            net.decompress(bad, c0["shape"])
        torch.cuda.synchronize()
        # latents beyond 16 bits: the host's packed words (row << 16 | symbol) cannot carry them - the call is coded on the
        # device as a whole, same strings; the decoder's host share (rows down, 32-bit symbols up) is not concerned
        net.g_a[6].weight.mul_(4000.0)
        share["enc"], share["dec"] = 0, 0
        big0 = net.compress(x)
        ymax = max(abs(int(v)) for v in net.decompress(big0["strings"], big0["shape"])["x_hat"].shape)  # (shape only: the call must work)
        assert ymax > 0
        share["enc"], share["dec"] = 6, 6
        big = net.compress(x)
        assert sum(n for _, n, _, _ in big["strings"][0].segments) == 21  # no tile went to the host
        assert [bytes(s_) for s_ in big["strings"][0]] == [bytes(s_) for s_ in big0["strings"][0]]
        assert max(len(s_) for s_ in big["strings"][0]) > 2 * max(len(s_) for s_ in ys0)  # (escapes everywhere: the latents are large)
        assert torch.equal(net.decompress(big["strings"], big["shape"])["x_hat"],
                           net.decompress([[bytes(s_) for s_ in lst] for lst in big0["strings"]], big0["shape"])["x_hat"])


def test_hyper_codec_mid_size_calls_run_the_host_pipeline(monkeypatch, request):
    """A host-coded call of at least 4 tiles per host thread takes the sub-chunk pipeline with every tile in the host's
    share (codec.hyper_fast_path / hyper_host_share through the real policy): the plain module path's strings and tiles."""
    from licos_amd import codec, ops
    if ops.HOST_CODER == "0":
        pytest.skip("device-coder placement: the policy under test is the host's")
    monkeypatch.setattr(ops, "HOST_CODER", "auto")
    monkeypatch.setattr(ops, "host_threads", lambda: 2)
    torch.manual_seed(5)
    net = licos_amd.get_model("bmshj2018-hyperprior", False, 13, 1).to(DEV).eval().set_precision("fp16")
    with torch.no_grad():
        licos_amd.synthetic.make_trained_like(net, seed=4)
    net.update(force=True)
    x = licos_amd.synthetic.tiles(19, 13, 128, seed=2, kind="s2-merged", device=DEV)
    assert ops.host_coder_preferred(19) and codec.hyper_fast_path(net, 19) and not codec.hyper_fast_path(net, 7)
    assert codec.hyper_host_share(19, "enc") == 19 == codec.hyper_host_share(19, "dec")
    calls = []
    real = codec.compress_hyper
    monkeypatch.setattr(codec, "compress_hyper", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        c = net.compress(x)
        d = net.decompress(c["strings"], c["shape"])["x_hat"]
        assert calls == [1]
        monkeypatch.setattr(codec.config, "host_split", False)  # the plain module path: transform, copy, one host coder call
        assert not codec.hyper_fast_path(net, 19)
        c0 = net.compress(x)
        d0 = net.decompress(c0["strings"], c0["shape"])["x_hat"]
        assert calls == [1]
    assert [[bytes(s_) for s_ in lst] for lst in c["strings"]] == [[bytes(s_) for s_ in lst] for lst in c0["strings"]]
    assert torch.equal(d, d0)
