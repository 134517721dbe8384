"""bpp / PSNR exactly as /root/reference/eval_utils.py:145-156,172-186 define them, with the
reductions done by the HIP kernels (sum of log2-likelihoods, squared differences)."""
import math

import torch

from . import ops


def compute_psnr(a, b):
    mse = float(ops.reduce_sqdiff(a.contiguous(), b.contiguous()).item()) / a.numel()
    return -10 * math.log10(mse)


def compute_bpp(out_net):
    size = out_net["x_hat"].size()
    num_pixels = size[0] * size[2] * size[3]
    return sum(torch.log(lik).sum() / (-math.log(2) * num_pixels) for lik in out_net["likelihoods"].values()).item()
