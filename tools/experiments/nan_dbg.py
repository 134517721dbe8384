import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import licos_amd
from licos_amd import ops
torch.manual_seed(0)
conv = torch.nn.Conv2d(3, 128, 5, 2, 2).cuda()
x = torch.rand(1, 3, 64, 64).cuda()
for mode in ("gamma0", "diag", "full"):
    m = licos_amd.GDN(128).cuda()
    with torch.no_grad():
        if mode == "gamma0":
            m.gamma.copy_(m.gamma_reparam.init(torch.zeros(128, 128).cuda()))
        if mode == "full":
            m.gamma.add_(0.01 * torch.rand(128, 128).cuda())
        c = ops.conv2d_f32(x, conv.weight.detach(), conv.bias.detach(), 2, 2)
        beta, gamma = m.effective()
        ref = ops.gdn_f32(c, gamma, beta, False)
        y = ops.conv2d_f32(x, conv.weight.detach(), conv.bias.detach(), 2, 2, gdn=(m.packed_f32split(), False))
    n_ref = (c / ref) ** 2
    n_got = (c / y) ** 2
    print(mode, "rel err", float((y - ref).abs().max() / ref.abs().max()))
    print("  norm ref", n_ref[0, :4, 5, 5].tolist(), "got", n_got[0, :4, 5, 5].tolist())
    print("  c", c[0, :4, 5, 5].tolist())
