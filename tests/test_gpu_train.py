"""GPU: one training step in the shape of licos/train.py:186-200 (forward with noise, RD loss, backward,
clip, Adam; aux loss on the quantiles) - HIP forward, stock-PyTorch backward (licos_amd/autograd.py) -
against the oracle differentiated by torch on the CPU."""
import pytest
import torch

import licos_amd
from oracle import model as om

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_train_step_matches_cpu_autograd():
    sd = om.perturb_state(om.make_factorized_state(3, quality=1, seed=42), seed=13, y_gain=20.0)
    net = licos_amd.get_model("bmshj2018-factorized", False, 3, 1)
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = om.synthetic_tiles(2, 3, 64, seed=3)
    g = torch.Generator().manual_seed(0)
    noise = torch.rand(2, 192, 4, 4, generator=g) - 0.5
    crit = licos_amd.RateDistortionLoss(lmbda=1e-2)
    conf = {"net": {"type": "Adam", "lr": 1e-4}, "aux": {"type": "Adam", "lr": 1e-3}}
    opt = licos_amd.net_aux_optimizer(net, conf)
    opt["net"].zero_grad()
    opt["aux"].zero_grad()
    out = net(x.to(DEV), noise=noise.to(DEV))
    res = crit(out, x.to(DEV))
    res["loss"].backward()
    # CPU reference: the oracle's forward under torch autograd
    ref_sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0 and "bound" not in k
                  and "pedestal" not in k and "target" not in k else v) for k, v in sd.items()}
    ref_out = om.forward(x, ref_sd, training=True, noise=noise)
    ref_res = om.rate_distortion_loss(ref_out, x, 1e-2)
    ref_res["loss"].backward()
    assert abs(float(res["loss"]) - float(ref_res["loss"])) < 1e-4 * abs(float(ref_res["loss"]))
    checked = 0
    for name, p in net.named_parameters():
        rg = ref_sd[name].grad
        if rg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        err = float((p.grad.cpu() - rg).abs().max() / rg.abs().max().clamp_min(1e-20))
        assert err < 2e-3, (name, err)
        checked += 1
    assert checked >= 40
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    before = net.g_a[0].weight.detach().clone()
    opt["net"].step()
    assert not torch.equal(before, net.g_a[0].weight.detach())
    aux = net.aux_loss()
    aux.backward()
    assert net.entropy_bottleneck.quantiles.grad is not None
    opt["aux"].step()
    # the fp16 path refuses autograd loudly
    net.set_precision("fp16")
    with pytest.raises(NotImplementedError):
        net(x.to(DEV))
    with torch.no_grad():
        net.eval()(x.to(DEV))
