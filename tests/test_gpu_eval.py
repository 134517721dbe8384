"""GPU checks of the evaluation helpers: MS-SSIM against the oracle restatement,
process_img's contract, checkpoint save -> load -> identical bit streams."""
import io
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import metrics as ometrics  # noqa: E402
from oracle import model as om  # noqa: E402


@pytest.mark.parametrize("shape,noise", [((2, 3, 256, 256), 0.05), ((1, 1, 256, 256), 0.2), ((1, 13, 192, 176), 0.02),
                                         ((1, 2, 161, 333), 0.1)])
def test_msssim_matches_oracle(shape, noise):
    import licos_amd
    g = torch.Generator().manual_seed(5)
    x = om.synthetic_tiles(shape[0], shape[1], -(-max(shape[2:]) // 8) * 8, seed=3)[..., : shape[2], : shape[3]].contiguous()
    y = (x + noise * torch.randn(x.shape, generator=g)).clamp(0, 1)
    ref = ometrics.ms_ssim(x, y, data_range=1.0)
    got = licos_amd.metrics.compute_msssim(x.cuda(), y.cuda())
    assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (got, ref)  # fp32 sums in a different order
    assert abs(licos_amd.metrics.compute_msssim(x.cuda(), x.cuda()) - 1.0) < 1e-6


def test_msssim_rejects_small_images():
    import licos_amd
    x = torch.rand(1, 3, 160, 256, device="cuda")
    with pytest.raises(AssertionError):
        licos_amd.metrics.compute_msssim(x, x)


def _drive_like_process_img(img, net):
    """What the reference's evaluation helper asks of the module (/root/reference/eval_utils.py:189-210, which stays the
    caller's own code - INTEGRATION.md): one forward and one compress of a (C, H, W) image, byte count taken over
    np.array(strings), reconstruction clamped and cropped to the input."""
    with torch.no_grad():
        out = net.forward(img.unsqueeze(0))
        coded = net.compress(img.unsqueeze(0))
    nbytes = np.frombuffer(np.array(coded["strings"]), dtype=np.uint8).size
    out["x_hat"].clamp_(0, 1)
    out["x_hat"] = out["x_hat"][..., : img.shape[1], : img.shape[2]]
    return out, out["x_hat"].squeeze().cpu(), torch.mean((out["x_hat"] - img).abs(), axis=1).squeeze().cpu(), nbytes


def test_process_img_contract():
    import licos_amd
    from licos_amd import metrics as eval_utils, synthetic
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1).cuda().eval()
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=1)
    net.update(force=True)
    img = synthetic.tiles(1, 3, 256, seed=4, device="cuda")[0, :, :200, :232].contiguous()  # not a multiple of 16: x_hat is cropped back
    pad = torch.zeros(3, 208, 240, device="cuda")
    pad[:, :200, :232] = img
    out_net, rec, diff, nbytes = _drive_like_process_img(pad, net)
    assert tuple(rec.shape) == (3, 208, 240) and tuple(diff.shape) == (208, 240)
    assert float(out_net["x_hat"].min()) >= 0.0 and float(out_net["x_hat"].max()) <= 1.0
    comp = net.compress(pad.unsqueeze(0))
    assert nbytes == sum(len(s) for lst in comp["strings"] for s in lst)
    bpp = eval_utils.compute_bpp(out_net)
    assert 0.0 < bpp < 24.0 and math.isfinite(eval_utils.compute_psnr(out_net["x_hat"], pad.unsqueeze(0)))


def test_checkpoint_round_trip_gives_identical_streams(tmp_path):
    import licos_amd
    from licos_amd import checkpoint, synthetic
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1).cuda().eval()
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=2)
    net.update(force=True)
    x = synthetic.tiles(3, 3, 256, seed=9, device="cuda")
    a = net.compress(x)
    path = str(tmp_path / "central.pth.tar")
    checkpoint.save_checkpoint(checkpoint.make_state(net, 11, 0.5, 3.0), False, filename=path)
    net2 = licos_amd.get_model("bmshj2018-factorized", False, 3, 1).cuda().eval()
    ck = checkpoint.load_checkpoint(path, net2, device="cuda")
    assert ck["batch_idx"] == 11 and ck["local_time"] == 3.0
    b = net2.compress(x)
    assert [bytes(s) for s in a["strings"][0]] == [bytes(s) for s in b["strings"][0]]
    buf = io.BytesIO()
    checkpoint.write_strings(buf, a)
    buf.seek(0)
    back = checkpoint.read_strings(buf)
    xa = net.decompress(a["strings"], a["shape"])["x_hat"]
    xb = net2.decompress(back["strings"], back["shape"])["x_hat"]
    assert torch.equal(xa, xb)
