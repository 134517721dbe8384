"""bpp / PSNR / MS-SSIM as /root/reference/eval_utils.py:145-186 define them, with the reductions done by the HIP
kernels (sum of log2-likelihoods, squared differences, windowed SSIM statistics)."""
import ctypes
import math

import torch
import torch.nn.functional as F

from . import _lib, ops

MS_SSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def compute_psnr(a, b):
    mse = float(ops.reduce_sqdiff(a.contiguous(), b.contiguous()).item()) / a.numel()
    return -10 * math.log10(mse)


def compute_bpp(out_net):
    size = out_net["x_hat"].size()
    num_pixels = size[0] * size[2] * size[3]
    return sum(torch.log(lik).sum() / (-math.log(2) * num_pixels) for lik in out_net["likelihoods"].values()).item()


def _gauss_window(size=11, sigma=1.5):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def compute_msssim(a, b, data_range=1.0, size_average=True):
    """eval_utils.py:159-169: pytorch_msssim.ms_ssim(a, b, data_range=1.0) - 5 scales, 11-tap sigma-1.5 window,
    K = (0.01, 0.03), 2x2 average pooling between scales, prod(relu(cs_i)^w_i) * relu(ssim_5)^w_5, mean over
    (batch, channel).  a, b: (B, C, H, W) fp32 on the GPU, smaller side > 160."""
    ops._dev(a, b)
    if a.shape != b.shape or a.dim() != 4:
        raise ValueError("Input images should have the same 4-d dimensions (B, C, H, W)")
    if min(a.shape[-2:]) <= (11 - 1) * 2 ** 4:
        raise AssertionError("Image size should be larger than 160 due to the 4 downsamplings in ms-ssim")
    lib = _lib.load()
    win = _gauss_window()
    win_c = (ctypes.c_float * 11)(*[float(v) for v in win])
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    x, y = ops._f32(a.contiguous()), ops._f32(b.contiguous())
    bsz, ch = x.shape[:2]
    vals = []
    for level in range(5):
        h, w = x.shape[-2:]
        sums = torch.zeros(bsz * ch, 2, device=x.device, dtype=torch.float64)
        rc = lib.licos_ssim_stats_f32(ops._p(x), ops._p(y), bsz * ch, h, w, ctypes.cast(win_c, ctypes.c_void_p), c1, c2, ops._p(sums),
                                      ops._stream())
        _lib.check(rc, "ssim_stats")
        mean = (sums / float((h - 10) * (w - 10))).to(torch.float32).view(bsz, ch, 2)
        if level < 4:
            vals.append(torch.relu(mean[..., 1]))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
        else:
            vals.append(torch.relu(mean[..., 0]))
    stack = torch.stack(vals, dim=0)  # (level, batch, channel)
    weights = torch.tensor(MS_SSIM_WEIGHTS, device=x.device, dtype=torch.float32).view(-1, 1, 1)
    ms = torch.prod(stack ** weights, dim=0)
    return ms.mean().item() if size_average else ms.mean(1)
