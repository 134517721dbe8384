"""CPU probe (no GPU): what would a Winograd-style reduction of g_a[2]'s stride-2 5x5 convolution cost in accuracy on
fp16 operands?  The stride-2 conv is four stride-1 phase convs on the input's parity planes (3x3, 3x2, 2x3, 2x2 taps);
F(2,3) along every 3-tap dimension and F(2,2) along every 2-tap one needs 49 multiplies per 2x2 outputs and channel pair
instead of 100.  Operands are rounded to fp16 where the MFMA would read them (transformed inputs, transformed weights),
products accumulate in fp32, the output transform is fp32 - against the same rounding of the direct form.
  python tools/experiments/winograd_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import licos_amd
from licos_amd import checkpoint, synthetic

torch.manual_seed(0)
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3)
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
w = net.g_a[2].weight.detach().double()              # [128][128][5][5]
b = net.g_a[2].bias.detach().double()
x0 = synthetic.tiles(2, 3, 256, seed=5, kind="aid", device="cpu").double()
sd = {k: v.detach().double() for k, v in net.state_dict().items()}
ped = (2.0 ** -18) ** 2
beta = torch.clamp(sd["g_a.1.beta"], min=(1e-6 + ped) ** 0.5) ** 2 - ped      # GDN reparametrisation (CompressAI layers/gdn.py)
gamma = torch.clamp(sd["g_a.1.gamma"], min=2.0 ** -18) ** 2 - ped
y0 = torch.nn.functional.conv2d(x0, sd["g_a.0.weight"], sd["g_a.0.bias"], stride=2, padding=2)
a = y0 * torch.rsqrt(torch.nn.functional.conv2d(y0 * y0, gamma[:, :, None, None], beta))   # the real input of g_a[2]: [2][128][128][128]
ref = torch.nn.functional.conv2d(a, w, b, stride=2, padding=2)


def h16(t):
    return t.to(torch.float16).to(torch.float64)


bias4 = b[None, :, None, None]
direct = torch.nn.functional.conv2d(h16(a), h16(w), None, stride=2, padding=2).float().double() + bias4

# F(2,3): Y = A^T [(G g) * (B^T d)], d = 4 inputs, g = 3 taps;  F(2,2): 3 multiplies for 2 outputs of a 2-tap filter
BT3 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G3 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT3 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
BT2 = torch.tensor([[1, -1, 0], [0, 1, 0], [0, -1, 1]], dtype=torch.float64)      # d = 3 inputs
G2 = torch.tensor([[1, 0], [1, 1], [0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 0], [0, 1, 1]], dtype=torch.float64)
T = {3: (BT3, G3, AT3), 2: (BT2, G2, AT2)}

ap = torch.nn.functional.pad(a, (2, 2, 2, 2))
Ho, Wo = ref.shape[-2:]


def winograd(rnd):
    out = torch.zeros_like(ref)
    mults = 0
    for py in (0, 1):            # row parity of the padded input: taps ky = py, py + 2, (py + 4)
        for px in (0, 1):
            ty, tx = (3 if py == 0 else 2), (3 if px == 0 else 2)
            plane = ap[:, :, py::2, px::2]                       # [B][C][Ho + 2][Wo + 2]
            g = w[:, :, py::2, px::2]                            # [O][C][ty][tx]
            BTy, Gy, ATy = T[ty]
            BTx, Gx, ATx = T[tx]
            U = rnd(torch.einsum("ai,ocij,bj->ocab", Gy, g, Gx))                     # transformed weights
            ny, nx = BTy.shape[1], BTx.shape[1]                  # input tile edge: 4 or 3
            d = plane.unfold(2, ny, 2).unfold(3, nx, 2)[:, :, :Ho // 2, :Wo // 2]   # tiles of 2 x 2 outputs: [B][C][Ho/2][Wo/2][ny][nx]
            V = rnd(torch.einsum("ai,bcyxij,ej->bcyxae", BTy, d, BTx))               # transformed inputs
            M = torch.einsum("ocae,bcyxae->boyxae", U, V)
            if rnd is h16:
                M = M.float().double()                                               # fp32 accumulation over the channels
            Y = torch.einsum("pa,boyxae,qe->boyxpq", ATy, M, ATx)                    # [B][O][Ho/2][Wo/2][2][2]
            out += Y.permute(0, 1, 2, 4, 3, 5).reshape(ref.shape)
            mults += U.shape[-2] * U.shape[-1]
    return out + bias4, mults


wino, mults = winograd(h16)
exact, _ = winograd(lambda t: t)


def err(t):
    e = (t - ref)
    return float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(e.abs().max() / ref.abs().max())


print("multiplies per 2 x 2 outputs and channel pair: direct 100, Winograd-style %d" % mults)
print("direct form, fp16 operands : rel rms %.2e, max %.2e of the output's range" % err(direct))
print("Winograd-style, fp16 operands: rel rms %.2e, max %.2e" % err(wino))
print("sanity: the same algebra in fp64 against the reference: max %.1e" % float((exact - ref).abs().max()))
