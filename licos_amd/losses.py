"""RateDistortionLoss with CompressAI's interface (``losses/rate_distortion.py``), built at
/root/reference/licos/train.py:123 and evaluated at train.py:192,288."""
import math

import torch
import torch.nn as nn


class RateDistortionLoss(nn.Module):
    def __init__(self, lmbda=1e-2, metric="mse", return_type="all"):
        super().__init__()
        if metric != "mse":
            raise NotImplementedError(f"{metric} is not implemented!")
        self.lmbda = lmbda
        self.return_type = return_type

    def forward(self, output, target):
        n, _, h, w = target.size()
        num_pixels = n * h * w
        out = {}
        out["bpp_loss"] = sum(
            (torch.log(lik).sum() / (-math.log(2) * num_pixels)) for lik in output["likelihoods"].values()
        )
        out["mse_loss"] = torch.mean((output["x_hat"] - target) ** 2)
        out["loss"] = self.lmbda * 255 ** 2 * out["mse_loss"] + out["bpp_loss"]
        return out if self.return_type == "all" else out[self.return_type]
