"""CPU: the C-ABI library loads and exports every declared symbol, its host-side entry points
agree with the oracle, and the host mirror of the reference interface behaves like it."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import licos_amd
from licos_amd import _lib, ops
from oracle import model as om
from oracle import rans

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "licos_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(licos_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.SO_PATH)
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/licos_hip.h but not exported"
    # and the Python binding table covers them all
    assert set(names) == set(_lib.SIGNATURES.keys())
    assert _lib.load().licos_abi_version() == 1


def test_cpu_tensors_fail_loudly():
    x = torch.zeros(1, 3, 16, 16)
    w = torch.zeros(4, 3, 5, 5)
    with pytest.raises(_lib.LicosError):
        ops.conv2d_f32(x, w, None, 2, 2)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    with pytest.raises(_lib.LicosError):
        net(torch.zeros(1, 3, 64, 64))


def test_pmf_to_quantized_cdf_matches_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "pmf_kat.npz"))
    for i in range(int(g["n"])):
        assert np.array_equal(ops.pmf_to_quantized_cdf(g[f"pmf{i}"], 16), g[f"cdf{i}"])
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(2, 120))
        p = rng.random(n).astype(np.float32) ** float(rng.integers(1, 9))
        p /= p.sum()
        assert np.array_equal(ops.pmf_to_quantized_cdf(p, 16), rans.pmf_to_quantized_cdf(p, 16))
    for bad in ([0.5, -0.1], [0.5, float("nan")], [0.0, 0.0]):
        with pytest.raises(ValueError):
            ops.pmf_to_quantized_cdf(np.array(bad, dtype=np.float32))


def test_encoder_reciprocal_table_is_exact():
    """q = mulhi64(x, rcp) >> shift must equal x // freq over the whole state range the coder
    uses (x < freq << 47) and x + bias + q*(65536-freq) must equal the textbook update."""
    sd = om.perturb_state(om.make_factorized_state(3, 1), seed=1)
    om.eb_update(sd)
    cdf = sd["entropy_bottleneck._quantized_cdf"].numpy()
    ln = sd["entropy_bottleneck._cdf_length"].numpy()
    # add rows with freq 1, a power of two, 65535 and the single-symbol 65536 case
    extra = np.zeros((2, cdf.shape[1]), dtype=np.int32)
    extra[0, :5] = [0, 1, 2, 32770, 65536]
    extra[1, :2] = [0, 65536]
    cdf = np.concatenate([cdf, extra])
    ln = np.concatenate([ln, np.array([5, 2], dtype=np.int32)])
    table = ops.rans_build_enc_table(cdf, ln)
    rng = np.random.default_rng(0)
    rows = list(range(0, 192, 37)) + [192, 193]
    for r in rows:
        for s in range(ln[r] - 1):
            rec = table[r, s].tobytes()
            rcp = int.from_bytes(rec[0:8], "little")
            bias = int.from_bytes(rec[8:12], "little")
            freq16 = int.from_bytes(rec[12:14], "little")
            shift = int.from_bytes(rec[14:16], "little")
            start, freq = int(cdf[r, s]), int(cdf[r, s + 1] - cdf[r, s])
            assert (freq16 or 65536) == freq
            hi = freq << 47
            xs = [1 << 31, (1 << 31) + 1, hi - 1, hi // 2, freq * 12345 + 1] + [int(v) for v in rng.integers(1 << 31, hi, size=64, dtype=np.uint64)]
            for x in xs:
                q = ((x * rcp) >> 64) >> shift
                new = x + bias + q * (65536 - freq)
                assert new == ((x // freq) << 16) + (x % freq) + start, (r, s, x)


def test_state_dict_surface_and_channel_surgery():
    net = licos_amd.get_model("bmshj2018-factorized", False, 13, 1)
    assert isinstance(net.g_a[0], torch.nn.Conv2d) and net.g_a[0].in_channels == 13
    assert isinstance(net.g_s[6], torch.nn.ConvTranspose2d) and net.g_s[6].out_channels == 13
    assert net.entropy_bottleneck.filters == (13, 13, 3, 3) and net.entropy_bottleneck.channels == 192
    assert sum(p.numel() for p in net.parameters()) == 3108237
    keys = set(net.state_dict().keys())
    ref = set(om.make_factorized_state(13, 1).keys())
    assert keys == ref
    with pytest.raises(ValueError):
        # a zoo model outside the three raw-data ones
        from licos_amd import zoo
        zoo.image_models["dummy"] = zoo.bmshj2018_factorized
        try:
            licos_amd.get_model("dummy", False, 1, 1)
        finally:
            del zoo.image_models["dummy"]
    q6 = licos_amd.image_models["bmshj2018-factorized"](quality=6, pretrained=False)
    assert q6.g_a[0].out_channels == 192 and q6.g_a[6].out_channels == 320


def test_load_state_dict_accepts_both_eb_namings_and_updated_tables():
    sd = om.perturb_state(om.make_factorized_state(3, 1), seed=5)
    om.eb_update(sd)
    old = {}
    for k, v in sd.items():
        m = re.match(r"entropy_bottleneck\.(matrices|biases|factors)\.(\d+)", k)
        if m:
            k = "entropy_bottleneck._" + {"matrices": "matrix", "biases": "bias", "factors": "factor"}[m.group(1)] + m.group(2)
        old[k] = v
    for variant in (sd, old):
        net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
        net.load_state_dict(variant)
        assert tuple(net.entropy_bottleneck._quantized_cdf.shape) == tuple(sd["entropy_bottleneck._quantized_cdf"].shape)
        for k, v in net.state_dict().items():
            assert torch.equal(v, sd[k]), k


def test_update_builds_the_oracles_tables():
    for cin, seed in ((3, 7), (1, 8), (13, 9)):
        sd = om.perturb_state(om.make_factorized_state(cin, 1), seed=seed)
        net = licos_amd.get_model("bmshj2018-factorized", False, cin, 1)
        net.load_state_dict(sd)
        assert net.update() is True and net.update() is False and net.update(force=True) is True
        om.eb_update(sd)
        for k in ("_quantized_cdf", "_cdf_length", "_offset"):
            assert torch.equal(getattr(net.entropy_bottleneck, k).cpu(), sd["entropy_bottleneck." + k]), k
        assert abs(float(net.aux_loss()) - float(om.eb_aux_loss(sd))) < 1e-3


def test_compress_before_update_raises_like_compressai():
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    with pytest.raises(ValueError, match="Uninitialized CDFs"):
        net.entropy_bottleneck.compress(torch.zeros(1, 192, 4, 4))


def test_aux_optimizer_split_and_loss_surface():
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    conf = {"net": {"type": "Adam", "lr": 1e-4}, "aux": {"type": "Adam", "lr": 1e-3}}
    opt = licos_amd.net_aux_optimizer(net, conf)
    aux = [p for g in opt["aux"].param_groups for p in g["params"]]
    assert len(aux) == 1 and aux[0] is net.entropy_bottleneck.quantiles
    crit = licos_amd.RateDistortionLoss(lmbda=1e-2)
    x = torch.rand(2, 3, 32, 32)
    out = {"x_hat": x + 0.01, "likelihoods": {"y": torch.full((2, 192, 2, 2), 0.5)}}
    res = crit(out, x)
    ref = om.rate_distortion_loss(out, x, 1e-2)
    for k in ("loss", "mse_loss", "bpp_loss"):
        assert abs(float(res[k]) - float(ref[k])) < 1e-6


def test_hyperprior_surface_and_tables():
    """bmshj2018-hyperprior (BASELINE config 5): CompressAI's parameter count, state_dict keys, the LICOS
    channel surgery (EB over N channels with (in,in,3,3) filters) and host-built tables == oracle."""
    stock = licos_amd.image_models["bmshj2018-hyperprior"](quality=1, pretrained=False)
    assert sum(p.numel() for p in stock.parameters()) == 5075843
    net = licos_amd.get_model("bmshj2018-hyperprior", False, 13, 5)
    sd = om.make_hyperprior_state(13, 5)
    assert set(net.state_dict().keys()) == set(sd.keys())
    assert net.entropy_bottleneck.channels == 128 and net.entropy_bottleneck.filters == (13, 13, 3, 3)
    assert om.count_parameters(sd) == sum(p.numel() for p in net.parameters())
    net.load_state_dict(sd)
    assert net.update() is True and net.update() is False
    om.hyper_update(sd)
    for mod in ("gaussian_conditional", "entropy_bottleneck"):
        for k in ("_quantized_cdf", "_cdf_length", "_offset"):
            assert torch.equal(getattr(getattr(net, mod), k), sd[f"{mod}.{k}"]), (mod, k)
    assert torch.allclose(net.gaussian_conditional.scale_table, om.get_scale_table())
    # a checkpoint saved after update() reloads with its tables
    net2 = licos_amd.get_model("bmshj2018-hyperprior", False, 13, 5)
    net2.load_state_dict(net.state_dict())
    assert torch.equal(net2.gaussian_conditional._quantized_cdf, net.gaussian_conditional._quantized_cdf)


def test_stream_container_round_trip(tmp_path):
    """checkpoint.write_strings / read_strings: length-prefixed container, ragged and empty strings included."""
    import io
    from licos_amd import checkpoint
    out = {"strings": [[b"", b"\x00\x01\x02", bytes(range(256)) * 3], [b"a", b"", b"xyz"]], "shape": (16, 16)}
    buf = io.BytesIO()
    checkpoint.write_strings(buf, out)
    buf.seek(0)
    back = checkpoint.read_strings(buf)
    assert back == out
    with pytest.raises(ValueError):
        checkpoint.read_strings(io.BytesIO(b"nope" + b"\0" * 16))
    trunc = io.BytesIO(buf.getvalue()[:-5])
    with pytest.raises(ValueError):
        checkpoint.read_strings(trunc)
    with pytest.raises(ValueError):
        checkpoint.write_strings(io.BytesIO(), {"strings": [[b"a"], [b"a", b"b"]], "shape": (1, 1)})


def test_checkpoint_dict_layout(tmp_path):
    """The checkpoint file is the reference's dict (federation_utils.py:69-78) and the over-time path follows
    licos/utils.py:82-111."""
    import torch
    from licos_amd import checkpoint

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.arange(4.0))

    net = Net()
    state = checkpoint.make_state(net, batch_idx=7, loss=0.25, local_time=12.5)
    assert set(state) == {"batch_idx", "state_dict", "loss", "local_time"}
    cfg = {"save_path": str(tmp_path / "results" / "modelX")}
    os.makedirs(tmp_path / "results", exist_ok=True)
    checkpoint.save_model_checkpoint_over_time(cfg, 12.5, 3, state)
    expect = tmp_path / "results" / "modelX" / "modelX_time_checkpoints" / "modelX_rank_3_sim_time=12.5.pth.tar"
    assert expect.exists()
    back = torch.load(expect, weights_only=False)
    assert back["batch_idx"] == 7 and back["local_time"] == 12.5
    assert torch.equal(back["state_dict"]["w"], net.w.detach())


def test_codec_py_record_layout():
    """checkpoint.write_image / read_image: CompressAI examples/codec.py's per-image record, byte for byte as restated
    from its source (header bytes, big-endian uint32 fields), for a factorized (1 string list) and a hyperprior (2) result."""
    import io
    import struct
    from licos_amd import checkpoint
    out = {"strings": [[b"abcd" * 3, b"xy" * 4], [b"zz" * 2, b""]], "shape": (4, 6)}
    buf = io.BytesIO()
    n = checkpoint.write_image(buf, out, 1, "bmshj2018-hyperprior", 5, (64, 96))
    raw = buf.getvalue()
    assert n == len(raw)
    assert raw[:2] == bytes([2, (0 << 4) | 4])                         # model id (current table), metric mse, quality 5
    assert struct.unpack(">2I", raw[2:10]) == (64, 96)                 # original size
    assert struct.unpack(">3I", raw[10:22]) == (4, 6, 2)               # latent shape, number of string lists
    assert struct.unpack(">I", raw[22:26]) == (8,) and raw[26:34] == b"xy" * 4
    assert struct.unpack(">I", raw[34:38]) == (0,) and len(raw) == 38
    buf.seek(0)
    model, metric, quality, size, back = checkpoint.read_image(buf)
    assert (model, metric, quality, size) == ("bmshj2018-hyperprior", "mse", 5, (64, 96))
    assert back == {"strings": [[b"xy" * 4], [b""]], "shape": (4, 6)}
    # the older zoo order (no -relu variant): hyperprior is id 1
    b2 = io.BytesIO()
    checkpoint.write_image(b2, out, 0, "bmshj2018-hyperprior", 1, (64, 96), ids="legacy")
    assert b2.getvalue()[0] == 1
    b2.seek(0)
    assert checkpoint.read_image(b2, ids="legacy")[0] == "bmshj2018-hyperprior"
    with pytest.raises(ValueError):
        checkpoint.write_image(io.BytesIO(), out, 0, "bmshj2018-factorized-relu", 1, (64, 96), ids="legacy")
    with pytest.raises(ValueError):
        checkpoint.read_image(io.BytesIO(raw[:20]))


def test_bench_flop_accounting_matches_the_survey():
    """bench.py prices a stage by its convolution MACs plus, when a GDN / IGDN is fused into it, that layer's
    C x C MACs per output pixel ("conv + GDN MACs x 2").  The eight stages of one 3-channel 256^2
    tile must add up to the survey's 11.056 GFLOP, and g_a[2]'s convolution alone to its 3.3554 GFLOP."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("licos_bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g_a = [("conv", 3, 128, 256, 256, True), ("conv", 128, 128, 128, 128, True), ("conv", 128, 128, 64, 64, True),
           ("conv", 128, 192, 32, 32, False)]
    g_s = [("deconv", 192, 128, 16, 16, True), ("deconv", 128, 128, 32, 32, True), ("deconv", 128, 128, 64, 64, True),
           ("deconv", 128, 3, 128, 128, False)]
    total = sum(bench.stage_flops(*s[:5], norm=s[5]) for s in g_a + g_s)
    assert abs(total - 11.056e9) < 2e6
    assert abs(bench.stage_flops("conv", 128, 128, 128, 128) - 3.3554432e9) < 1
    assert abs(bench.stage_flops("deconv", 128, 128, 64, 64) - 3.3554432e9) < 1
    # the fused norms: 128 x 128 MACs per output pixel
    assert bench.stage_flops("deconv", 128, 128, 64, 64, norm=True) - bench.stage_flops("deconv", 128, 128, 64, 64) == 2.0 * 128 * 128 * 128 * 128
    # the first stage as the kernel runs it (3 x 3 stride 1 over the space-to-depth image) is the same work
    assert bench.stage_flops("conv", 3, 128, 256, 256) == 2.0 * 128 * 128 * 25 * 3 * 128


def test_decoder_image_is_exact_on_every_value():
    """Host-side check of the image the kernel searches (the same inline function, licos_rans_image_lookup): every row,
    a dense sweep of 16-bit values, under the default budget and a starved one."""
    sd = {}
    om.gc_update(sd)
    cdf, cdf_len, offset = (sd["gaussian_conditional." + k].numpy() for k in ("_quantized_cdf", "_cdf_length", "_offset"))
    for budget in (None, 64 * 1024):
        img = ops.rans_image_build(cdf, cdf_len, offset, budget_bytes=budget)
        slow = 0
        for r in range(cdf.shape[0]):
            row = cdf[r, : cdf_len[r]]
            cfs = np.unique(np.concatenate((np.arange(0, 65536, 97), row[:-1], np.maximum(row[1:] - 1, 0))))
            want = np.searchsorted(row, cfs, side="right") - 1
            for cf, s in zip(cfs.tolist(), want.tolist()):
                got = ops.rans_image_lookup(img, r, cf)
                assert got[:3] == (s, int(row[s]), int(row[s + 1])), (budget, r, cf, got)
                slow += got[3]
        assert slow > 0  # the bounded search is exercised too


def test_bench_spawner_stops_all_ranks_when_one_fails():
    """`bench.py --gpus 2` from a plain shell starts its own ranks.  Here (no GPU) every rank fails at start-up: the parent
    must notice, end the group and exit non-zero promptly, with the failing rank's stderr relayed - not wait on rank 0."""
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["CUDA_VISIBLE_DEVICES"] = ""
    env["HIP_VISIBLE_DEVICES"] = ""
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--batch", "8",
                        "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True,
                       timeout=300)
    assert p.returncode != 0
    assert time.monotonic() - t0 < 240
    assert b"exited with status" in p.stderr and b"stderr (tail)" in p.stderr


def test_fp32_path_shape_queries_and_argument_checks():
    """Host-side answers of the fp32 path's C ABI (no GPU): which shapes the one-pass GDN kernels / the fp32-norm epilogue
    serve, operand sizes, and that bad arguments come back as LICOS_EINVAL with a message instead of a launch."""
    from licos_amd import _lib
    lib = _lib.load()
    assert lib.licos_gdn_f32_split3_applies(128, 64 * 64) == 1
    assert lib.licos_gdn_f32_split3_applies(128, 33) == 0 and lib.licos_gdn_f32_split3_applies(192, 64 * 64) == 0
    assert lib.licos_gdn_f32_split3_applies(128, 0) == 0 and lib.licos_gdn_f32_split3_applies(128, 1 << 20) == 0
    assert lib.licos_packed_gdn_f32split_bytes(128) == 4 * 4 * 2 * 2 * 1024 + 512
    assert lib.licos_packed_gdn_f32split_bytes(192) == 0 and lib.licos_packed_gdn_f32split_bytes(0) == 0
    for call in (lambda: lib.licos_gdn_f32_split3(None, None, None, None, 1, 128, 4096, 0, None),
                 lambda: lib.licos_gdn_bwd_fused_f32(None, None, None, None, None, None, None, 1, 128, 4096, 0, None),
                 lambda: lib.licos_gdn_f32_fwd_norm(None, None, None, None, None, 1, 128, 4096, 0, None),
                 lambda: lib.licos_nchw_f32_split3_blk16(None, None, 1, 3, 8, 8, 0, None),
                 lambda: lib.licos_pack_gdn_f32split(None, None, 0.0, 0.0, 0.0, 128, None, None)):
        assert call() == -1  # LICOS_EINVAL
        assert lib.licos_last_error()


def test_band_sized_end_stage_queries_and_argument_checks():
    """Host-side answers of the first / last stage entry points for 1..3 bands (csrc/mfma_first.hip, mfma_rows.hip): operand
    sizes, and bad arguments as LICOS_EINVAL with a message instead of a launch (no GPU needed)."""
    from licos_amd import _lib
    lib = _lib.load()
    # last stage: 3 row shifts x Cin / 16 A fragments of 1 KB; 1..3 bands only
    assert lib.licos_packed_deconv_w_rows_bytes(128, 3) == 3 * 8 * 1024 and lib.licos_packed_deconv_w_rows_bytes(192, 1) == 3 * 12 * 1024
    assert lib.licos_packed_deconv_w_rows_bytes(128, 4) == 0 and lib.licos_packed_deconv_w_rows_bytes(0, 3) == 0
    # first stage: 5 kernel rows x 4 channel tiles of 1 KB + the zero granule; 1..3 bands into <= 128 channels
    assert lib.licos_packed_conv_w_first_bytes(3, 128) == 5 * 4 * 1024 + 64
    assert lib.licos_packed_conv_w_first_bytes(4, 128) == 0 and lib.licos_packed_conv_w_first_bytes(3, 192) == 0
    # interleaved zero-bordered image: (H + 4) rows of round_up((W + 4) C + 8, 8) halfs per image, plus edge-tile slack
    rs = ((256 + 4) * 3 + 8 + 7) // 8 * 8
    n = lib.licos_hwc_pad_f16_bytes(2, 3, 256, 256)
    assert n >= 2 * 260 * rs * 2 and n <= (2 * 260 + 32) * rs * 2 + 4096 and lib.licos_hwc_pad_f16_bytes(1, 4, 256, 256) == 0
    one = ctypes.c_void_p(16)  # a non-null, 16-byte aligned stand-in: the checks below fail before any dereference
    for call in (lambda: lib.licos_deconv5x5s2_rows_f16(None, None, None, None, 0, 1, 128, 8, 8, 3, None),
                 lambda: lib.licos_deconv5x5s2_rows_f16(one, one, one, one, 0, 1, 128, 8, 8, 4, None),      # 4 bands
                 lambda: lib.licos_deconv5x5s2_rows_f16(one, one, one, one, 2, 1, 128, 8, 7, 3, None),      # x-split needs an even width
                 lambda: lib.licos_deconv5x5s2_rows_f16(one, one, one, one, 0, 1, 64, 8, 8, 3, None),       # 64 input channels
                 lambda: lib.licos_conv5x5s2_first_nchw_f16(None, None, None, None, 0, None, 1, 3, 64, 64, 128, None),
                 lambda: lib.licos_conv5x5s2_first_nchw_f16(one, one, one, None, 0, one, 1, 3, 64, 66, 128, None),  # W % 4
                 lambda: lib.licos_conv5x5s2_first_nchw_f16(one, one, one, None, 1, one, 1, 3, 64, 64, 128, None),  # GDN without gamma
                 lambda: lib.licos_conv5x5s2_first_f16(one, one, one, None, 0, one, 1, 4, 64, 64, 128, None),       # 4 bands
                 lambda: lib.licos_pack_deconv_w_rows_f16(one, 128, 4, one, None),
                 lambda: lib.licos_pack_conv_w_first_f16(one, 3, 192, one, None),
                 lambda: lib.licos_nchw_f32_to_hwc_pad_f16(one, one, 1, 4, 16, 16, None)):
        assert call() == -1  # LICOS_EINVAL
        assert lib.licos_last_error()


def test_spawner_returns_a_failing_ranks_status(capsys):
    """bench.py --gpus N (bare): a rank that gives up with a non-zero status - the native-RCCL watchdog exits 3 after
    printing what it has - must become the spawner's status, the other ranks are stopped, and rank 0's line is relayed."""
    import argparse
    import sys
    import bench
    code = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
            "if r == 0:\n"
            "    print('{\"partial\": true}', flush=True)\n"
            "    sys.exit(3)\n"
            "time.sleep(60)\n")
    import time
    t0 = time.monotonic()
    status = bench.spawn_workers(argparse.Namespace(gpus=3), limit_s=30.0, cmd=[sys.executable, "-c", code])
    assert status == 3
    assert time.monotonic() - t0 < 25  # the sleeping ranks were stopped, not waited for
    assert '{"partial": true}' in capsys.readouterr().out
    ok = bench.spawn_workers(argparse.Namespace(gpus=2), limit_s=30.0, cmd=[sys.executable, "-c", "pass"])
    assert ok == 0


def test_hyper_retry_chunk_keeps_the_word_sink_below_4_gb():
    """compress_hyper's worst-case retry (cap_words = 2 ny + 8) at M = 320, 512^2 tiles: 2048 streams x 655 369 words
    x 4 B = 5.4 GB would be refused by licos_rans_encode_records (32-bit sink offsets); the retry shrinks the chunk."""
    from licos_amd import codec
    ny = 320 * 32 * 32
    c = codec.hyper_retry_chunk(2048, ny)
    assert 1 <= c < 2048 and (2 * ny + 8 + 1) * c * 4 < 2 ** 32 and (2 * ny + 8 + 1) * (c + 1) * 4 >= 2 ** 32
    assert codec.hyper_retry_chunk(2048, 192 * 32 * 32) == 2048  # M = 192: 3.2 GB, fits as it is
    assert codec.hyper_retry_chunk(4, 10 ** 9) == 1


def test_host_share_of_a_call(monkeypatch):
    """Split placement of the serial coder (codec.host_share): all of a small batch, the exposed end of a large one -
    as many tiles as the host threads code during ONE device coder launch - and nothing when the host coder is off."""
    from licos_amd import codec, ops
    monkeypatch.setattr(ops, "HOST_CODER", "auto")
    monkeypatch.setattr(ops, "host_threads", lambda: 16)
    monkeypatch.setattr(codec.config, "host_split", True)
    monkeypatch.setattr(codec.placement.rate, "cap_state", {})
    monkeypatch.setattr(codec.placement.rate, "factor", {"enc": 1.0, "dec": 1.0})
    cap_e, cap_d = codec.host_capacity("enc"), codec.host_capacity("dec")
    assert cap_e == int(0.85 * 16 * codec.config.dev_ns["enc"] / codec.config.host_ns["enc"]) // 32 * 32 and 0 < cap_d < cap_e and cap_d % 32 == 0
    for b in (1, 16, 64, 384):
        assert codec.host_share(b, "enc") == b and codec.host_share(b, "dec") == b  # the host-only batches of rounds 2 - 3
    # split: the call's first tiles (above the host-only batches, whichever limit is the larger)
    assert codec.host_share(cap_d, "dec") == cap_d and codec.host_share(max(cap_d, 384) + 1, "dec") == cap_d
    edge = int(codec.config.enc_all_host * cap_e)
    assert codec.host_share(cap_e, "enc") == cap_e and codec.host_share(edge, "enc") == edge  # a little over: still all
    assert codec.host_share(edge + 1, "enc") == int(codec.config.enc_tail * cap_e)  # ... then the call's last tiles
    assert codec.host_share(16384, "enc") == int(codec.config.enc_tail * cap_e) and codec.host_share(16384, "dec") == cap_d  # the exposed ends
    monkeypatch.setattr(codec.config, "host_split", False)
    assert codec.host_share(1000, "enc") == 0 and codec.host_share(64, "enc") == 64
    monkeypatch.setattr(ops, "HOST_CODER", "0")
    assert codec.host_share(1, "enc") == 0 and codec.host_share(16384, "dec") == 0
    monkeypatch.setattr(ops, "HOST_CODER", "1")
    assert codec.host_share(5000, "enc") == 5000
    monkeypatch.setattr(ops, "host_threads", lambda: 1)
    monkeypatch.setattr(ops, "HOST_CODER", "auto")
    monkeypatch.setattr(codec.config, "host_split", True)
    assert codec.host_share(3 * codec.host_capacity("dec") + 1, "dec") == codec.host_capacity("dec") < 64


def test_host_capacity_follows_the_measured_host_rate(monkeypatch):
    """codec.note_host_rate: a host that codes slower than nominal (a shared host under other tenants' load) shrinks the
    share the next calls give it; a quiet one restores it; small sub-chunks and a faster-than-nominal host change nothing."""
    from licos_amd import codec, ops
    monkeypatch.setattr(ops, "host_threads", lambda: 16)
    monkeypatch.setattr(codec.placement.rate, "factor", {"enc": 1.0, "dec": 1.0})
    monkeypatch.setattr(codec.placement.rate, "cap_state", {})
    cap0 = codec.host_capacity("enc")
    nsym = 49152
    codec.note_host_rate("enc", 8, nsym, 1.0)               # too few tiles to say anything
    assert codec.host_capacity("enc") == cap0
    codec.note_host_rate("enc", 256, nsym, 256 * nsym * 0.5e-9 / 16)   # faster than nominal: the factor stays 1
    assert codec.host_capacity("enc") == cap0
    for _ in range(6):
        codec.note_host_rate("enc", 256, nsym, 256 * nsym * 9.0e-9 / 16)  # 5 x slower than the expected 1.8 ns
    assert cap0 / 6.0 < codec.host_capacity("enc") < cap0 / 4.0  # (on the grid of 2 tiles per thread)
    assert codec.host_capacity("dec") == int(0.85 * 16 * codec.config.dev_ns["dec"] / codec.config.host_ns["dec"]) // 32 * 32  # per direction
    for _ in range(10):
        codec.note_host_rate("enc", 256, nsym, 256 * nsym * 1.8e-9 / 16)
    assert codec.host_capacity("enc") >= 0.99 * cap0


def test_host_rate_recovers_through_the_small_subchunks_a_small_share_hands_out(monkeypatch):
    """after a busy-host phase the decode share of a large call drops to a few tiles per thread, `ramp` then
    hands out only small sub-chunks - and those must be able to bring the factor back (a small sample that reads FAST is
    evidence; one that reads slow is not, its per-call overheads dominate).  Driven through host_share and host_subchunks."""
    from licos_amd import codec, ops
    pl = codec.placement
    monkeypatch.setattr(ops, "HOST_CODER", "auto")
    monkeypatch.setattr(ops, "host_threads", lambda: 16)
    monkeypatch.setattr(codec.config, "host_split", True)
    monkeypatch.setattr(pl.rate, "factor", {"enc": 1.0, "dec": 1.0})
    monkeypatch.setattr(pl.rate, "cap_state", {})
    nsym = 49152
    share0 = pl.host_share(16384, "dec")
    assert share0 >= 256
    for _ in range(8):  # a busy phase: full sub-chunks decode 8 x slower than nominal
        pl.note_host_rate("dec", 256, nsym, 256 * nsym * 8 * codec.config.expect_ns["dec"] * 1e-9 / 16)
    busy = pl.host_share(16384, "dec")
    assert 0 < busy <= share0 // 6
    subs = pl.host_subchunks(busy)
    assert subs and max(m for _, m in subs) < 4 * 16  # nothing the old rule (>= 4 tiles per thread) would have accepted
    for _, m in subs:  # slow small samples change nothing ...
        pl.note_host_rate("dec", m, nsym, m * nsym * 20 * codec.config.expect_ns["dec"] * 1e-9 / 16)
    assert pl.host_share(16384, "dec") == busy
    for _ in range(40):  # ... a quiet host seen through the same small sub-chunks restores the share, call by call
        for _, m in pl.host_subchunks(pl.host_share(16384, "dec")):
            pl.note_host_rate("dec", m, nsym, m * nsym * codec.config.expect_ns["dec"] * 1e-9 / 16)
    assert pl.host_share(16384, "dec") >= 0.9 * share0
    pl.note_host_rate("dec", 8, nsym, 1.0)  # fewer tiles than threads: no sample at all
    assert pl.host_share(16384, "dec") >= 0.9 * share0


def test_hyper_host_share_policy(monkeypatch):
    """codec.hyper_host_share (scale hyperprior, large calls): what the host threads code during one exposed device coder
    launch, in steps of 4 x threads, at most a third of the call, nothing when the host coder or the split is off; it
    moves only when the measured host rate has moved it by a whole step."""
    from licos_amd import codec, ops
    monkeypatch.setattr(ops, "HOST_CODER", "auto")
    monkeypatch.setattr(ops, "host_threads", lambda: 16)
    monkeypatch.setattr(codec.config, "host_split", True)
    monkeypatch.setattr(codec.placement.rate, "factor", {"enc": 1.0, "dec": 1.0})
    monkeypatch.setattr(codec.placement.rate, "hyper_share", {})
    monkeypatch.setattr(codec.config, "hyper_share", -1)
    for d in ("enc", "dec"):
        cap = 0.85 * 16 * codec.config.hyper_dev_ns[d] / codec.config.hyper_host_ns[d]
        want = int(cap) // 64 * 64
        assert want >= 128
        assert codec.hyper_host_share(4096, d) == want == codec.hyper_host_share(2048, d)
        assert codec.hyper_host_share(600, d) == 192 and codec.hyper_host_share(400, d) == 128  # a third of the call at most
        codec.placement.rate.factor[d] = cap / (want - 40.0)   # the host a little slower: under a step, the share stays
        assert codec.hyper_host_share(4096, d) == want
        codec.placement.rate.factor[d] = 2.0                    # half the rate: half the share (on the grid)
        assert codec.hyper_host_share(4096, d) == int(cap / 2) // 64 * 64
    monkeypatch.setattr(codec.config, "host_split", False)
    assert codec.hyper_host_share(4096) == 0
    monkeypatch.setattr(codec.config, "host_split", True)
    monkeypatch.setattr(ops, "HOST_CODER", "0")
    assert codec.hyper_host_share(4096, "dec") == 0


def test_prefetch_ring_registers_belong_to_the_ring_alone(tmp_path):
    """The stream-major plane encoder (csrc/rans.hip) and the record encoder (csrc/rans_gc.hip) keep symbols on their way from
    memory in NAMED registers (v200-v231 / v128-v255, four slots) and define the values only in the statement that waits for
    them: the first form passed them through ordinary asm operands, and the register allocator copied them in front of the
    wait (intermittently stale symbols).  Checked in the shipped library's disassembly, in address order: from a slot's
    request to that slot's own counted wait nothing touches its registers, nothing touches a slot between the prologue and
    its first wait in the loop (cyclically: behind its request at the loop's end), and nothing else is ever loaded there."""
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    so = os.path.join(ROOT, "licos_amd", "liblicos_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(so)):
        pytest.skip("needs llvm-objdump and the built library")
    shutil.copy(so, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    # kernel -> (first ring register, registers per slot, the loop's counted wait)
    wanted = {"rans_encode_plane_kernel": (200, 8, 30), "rans_encode_records_regs_kernel": (128, 32, 48)}
    checked = {}

    def regs_of(text):
        out = {int(r) for r in re.findall(r"\bv(\d+)\b", text)}
        for lo_, hi_ in re.findall(r"v\[(\d+):(\d+)\]", text):
            out.update(range(int(lo_), int(hi_) + 1))
        return out

    def check(kernel, body):
        first, per_slot, counted = wanted[next(k for k in wanted if k in kernel)]
        slot_of = lambda r: (r - first) // per_slot
        in_flight = [False] * 4
        before_first_wait = None  # set at the prologue's vmcnt(0)
        n_counted = n_loads = 0
        for ins in body:
            parts = ins.split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins)
            if m:
                if int(m.group(1)) == 0:
                    in_flight = [False] * 4
                    if before_first_wait is None:
                        before_first_wait = [True] * 4
                elif int(m.group(1)) == counted:
                    j = n_counted % 4
                    n_counted += 1
                    in_flight[j] = False
                    if before_first_wait is not None:
                        before_first_wait[j] = False
                continue
            ring = {r for r in regs_of(args) if r >= first}
            if not ring:
                continue
            slots = {slot_of(r) for r in ring}
            if op == "global_load_dwordx4" and regs_of(args.split(",")[0]) == ring:  # a request
                assert len(slots) == 1
                in_flight[slots.pop()] = True
                n_loads += 1
                continue
            assert not op.startswith(("global_load", "buffer_load", "flat_load", "ds_read")) or not (regs_of(args.split(",")[0]) & ring), \
                "%s: a load of the compiler's own into the ring: %s" % (kernel, ins)
            for j in slots:
                assert not in_flight[j], "%s: slot %d touched between its request and its wait: %s" % (kernel, j, ins)
                assert before_first_wait is None or not before_first_wait[j], \
                    "%s: slot %d touched in front of its first wait in the loop: %s" % (kernel, j, ins)
        assert n_loads >= 16 and n_counted >= 4, (kernel, n_loads, n_counted)
        checked[kernel] = (n_loads, n_counted)

    for obj in sorted(tmp_path.glob("*gfx950")):
        dis = subprocess.run([objdump, "-d", str(obj)], capture_output=True, text=True, check=True).stdout
        name, body = None, []
        for line in dis.splitlines() + ["0 <end>:"]:
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                if name:
                    check(name, body)
                name = m.group(1) if any(k in m.group(1) for k in wanted) else None
                body = []
            elif name and line.strip():
                body.append(line.split("//")[0].strip())
    assert any("rans_encode_plane_kernel" in k for k in checked) and any("rans_encode_records_regs_kernel" in k for k in checked), checked
