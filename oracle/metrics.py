"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU statement of the MS-SSIM the reference reports
(/root/reference/eval_utils.py:159-169 -> `pytorch_msssim.ms_ssim(a, b, data_range=1.0)`).  pytorch_msssim is not in
the reference tree nor installed here; this restates its published algorithm (v0.2/1.0: separable 11-tap Gaussian
"valid" filtering with torch conv2d, K = (0.01, 0.03), 5 scales with weights (0.0448, 0.2856, 0.3001, 0.2363,
0.1333), avg_pool2d(2, padding = size % 2) between scales, relu on the per-channel terms) - PARITY UNPINNED."""
import torch
import torch.nn.functional as F

WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _fspecial_gauss_1d(size, sigma):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).view(1, 1, 1, size)


def _gaussian_filter(x, win):
    c = x.shape[1]
    out = x
    for i, s in enumerate(x.shape[2:]):
        if s >= win.shape[-1]:
            out = F.conv2d(out, win.transpose(2 + i, -1).repeat(c, 1, 1, 1), stride=1, padding=0, groups=c)
    return out


def _ssim(x, y, data_range, win, k=(0.01, 0.03)):
    c1, c2 = (k[0] * data_range) ** 2, (k[1] * data_range) ** 2
    mu1, mu2 = _gaussian_filter(x, win), _gaussian_filter(y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = _gaussian_filter(x * x, win) - mu1_sq
    sigma2_sq = _gaussian_filter(y * y, win) - mu2_sq
    sigma12 = _gaussian_filter(x * y, win) - mu1_mu2
    cs_map = (2 * sigma12 + c2) / (sigma1_sq + sigma2_sq + c2)
    ssim_map = ((2 * mu1_mu2 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1), torch.flatten(cs_map, 2).mean(-1)


def ms_ssim(x, y, data_range=1.0, win_size=11, win_sigma=1.5):
    assert min(x.shape[-2:]) > (win_size - 1) * 2 ** 4
    win = _fspecial_gauss_1d(win_size, win_sigma)
    mcs = []
    for i in range(5):
        ssim_c, cs = _ssim(x, y, data_range, win)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, kernel_size=2, padding=pad)
            y = F.avg_pool2d(y, kernel_size=2, padding=pad)
    stack = torch.stack(mcs + [torch.relu(ssim_c)], dim=0)
    w = torch.tensor(WEIGHTS, dtype=torch.float32).view(-1, 1, 1)
    return torch.prod(stack ** w, dim=0).mean().item()
