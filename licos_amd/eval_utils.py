"""The model-facing half of /root/reference/eval_utils.py (lines 145-210): compute_psnr, compute_msssim,
compute_bpp and process_img with the same signatures and return values.  (The JPEG/PIL comparison helpers of that
file are outside the path.)"""
import numpy as np
import torch

from .metrics import compute_bpp, compute_msssim, compute_psnr  # noqa: F401  (same names as the reference)


def process_img(img, net):
    """eval_utils.py:189-210: forward + compress of one (C, H, W) image; returns (out_net, reconstructed, diff,
    compressed_size_in_bytes) with x_hat clamped to [0, 1] and cropped to the input size."""
    with torch.no_grad():
        out_net = net.forward(img.unsqueeze(0))
        compressed_img = net.compress(img.unsqueeze(0))
    compressed_size_in_bytes = np.frombuffer(np.array(compressed_img["strings"]), dtype=np.uint8).size
    out_net["x_hat"].clamp_(0, 1)
    out_net["x_hat"] = out_net["x_hat"][..., : img.shape[1], : img.shape[2]]
    reconstructed = out_net["x_hat"].squeeze().cpu()
    diff = torch.mean((out_net["x_hat"] - img).abs(), axis=1).squeeze().cpu()
    return out_net, reconstructed, diff, compressed_size_in_bytes
