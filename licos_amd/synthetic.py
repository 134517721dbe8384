"""Synthetic inputs and "trained-like" weights for benchmarks and smoke runs (no datasets or
checkpoints can be fetched here).

* ``tiles``: the deterministic tile recipe of SURVEY.md section 8(d) - AID-style RGB on the 8-bit grid
  (train.py:33-39 ``ToTensor()`` of 8-bit images) or Sentinel-2-shaped 12-bit DN -> 8-bit levels
  (raw_utils.py:128, raw_image_folder.py:192-196; merged = 13 channels with channel 12 zero,
  raw_image_folder.py:172).
* ``make_trained_like``: random-init networks produce |y| < 0.3 (every symbol 0) and a 4-bpp prior.
  This gives the entropy bottleneck a per-channel logistic prior with scales spread log-uniformly
  (as trained codecs have: many near-deterministic channels, a few wide ones) and rescales the last
  analysis conv so the latents actually follow it.  The arithmetic per tile is identical to a
  trained model of the same topology; only the stream length depends on it.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def tiles(batch, channels=3, size=256, seed=0, kind="aid", device="cpu"):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    out = torch.empty((batch, channels, size, size), dtype=torch.float32, device=device)
    step = 64
    for b0 in range(0, batch, step):
        n = min(step, batch - b0)
        low = torch.randn((n, channels, size // 8, size // 8), generator=gen).to(device)
        f = F.interpolate(low, scale_factor=8, mode="bilinear", align_corners=False)
        f = f + 0.05 * torch.randn((n, channels, size, size), generator=gen).to(device)
        lo = f.amin(dim=(1, 2, 3), keepdim=True)
        hi = f.amax(dim=(1, 2, 3), keepdim=True)
        f = (f - lo) / (hi - lo)
        if kind != "aid":
            f = torch.round(f * 4095.0) / 4095.0
        f = torch.round(f * 255.0) / 255.0
        if kind == "s2-merged" and channels == 13:
            f[:, 12] = 0.0
        out[b0:b0 + n] = f
    return out


def _softplus_inv(v):
    return math.log(math.expm1(v))


@torch.no_grad()
def make_trained_like(net, seed=0, scale_range=(0.02, 2.5), sample=None):
    """In-place: perturb GDN, give the EB per-channel logistic priors, calibrate g_a[-1].
    `net` must already sit on its device with precision set; `sample` is a batch of tiles."""
    gen = torch.Generator().manual_seed(seed)
    dev = next(net.parameters()).device
    for m in net.modules():
        if m.__class__.__name__ == "GDN":
            c = m.in_channels
            m.gamma.add_((0.01 * torch.rand(c, c, generator=gen)).to(dev))
            m.beta.mul_((0.7 + 0.6 * torch.rand(c, generator=gen)).to(dev))
    eb = net.entropy_bottleneck
    c = eb.channels
    lo, hi = scale_range
    hyper = hasattr(net, "gaussian_conditional")
    sigma = torch.exp(torch.empty(c).uniform_(math.log(lo), math.log(hi), generator=gen))
    # with zero factors the MLP is affine: F(v) = v * prod_i(sum of softplus(matrix_i)) + const.  The
    # stock init makes that slope 1/init_scale; rescale layer 0 per channel to slope 1/sigma_c.
    f = (1,) + eb.filters + (1,)
    layer_scale = eb.init_scale ** (1 / (len(eb.filters) + 1))
    base = 1.0 / layer_scale / f[1]
    m0 = torch.empty_like(eb.matrices[0].cpu())
    for ch in range(c):
        m0[ch].fill_(_softplus_inv(base * eb.init_scale / float(sigma[ch])))
    eb.matrices[0].copy_(m0.to(dev))
    for b in eb.biases:
        b.zero_()
    for fac in eb.factors:
        fac.zero_()
    t = math.log(2 / eb.tail_mass - 1)
    med = 0.3 * torch.randn(c, generator=gen) * sigma
    # medians shift the logistic: bias of the last layer = -median / sigma
    eb.biases[-1].copy_((-(med / sigma)).reshape(c, 1, 1).to(dev))
    q = torch.stack((med - t * sigma, med, med + t * sigma), dim=1).reshape(c, 1, 3)
    eb.quantiles.copy_(q.to(dev))
    if sample is None:
        cin = net.g_a[0].in_channels
        sample = tiles(4, cin, 256, seed=1234, kind="aid" if cin == 3 else "s2", device=dev)

    def calibrate(transform, inp, target_std, target_mean):
        """Rescale the last conv of `transform` so its output channels have the wanted spread."""
        last = transform[len(transform) - 1]
        out = transform(inp)
        mean = out.mean(dim=(0, 2, 3))
        std = out.std(dim=(0, 2, 3)).clamp_min(1e-8)
        gain = target_std.to(dev) / std
        last.weight.mul_(gain.reshape(-1, 1, 1, 1))
        last.bias.copy_((last.bias - mean) * gain + target_mean.to(dev))

    if not hyper:
        # channel c of y spread like logistic(med_c, sigma_c); std of a logistic = pi/sqrt(3) * scale
        calibrate(net.g_a, sample, 1.8138 * sigma, med)
    else:
        # scale hyperprior: y_c ~ N(0, s_c) with s_c log-uniform; h_s answers ~s_c (bias-dominated), z follows
        # the factorised logistic prior set up above
        m = net.g_a[len(net.g_a) - 1].out_channels
        s_y = torch.exp(torch.empty(m).uniform_(math.log(0.15), math.log(4.0), generator=gen))
        calibrate(net.g_a, sample, s_y, torch.zeros(m))
        y = net.g_a(sample)
        calibrate(net.h_a, y, 1.8138 * sigma, med)
        hs_last = net.h_s[len(net.h_s) - 2]  # conv before the final ReLU
        hs_last.weight.mul_(0.05)
        hs_last.bias.copy_(s_y.to(dev))
    net.update(force=True)
    return net
