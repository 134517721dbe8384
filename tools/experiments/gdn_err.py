"""Error of the fp32 GDN against float64, one-pass matrix-core kernel (default) or the vector-ALU kernel (LICOS_GDN_F32_MFMA=0)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from licos_amd import ops

torch.manual_seed(0)
C = 128
for scale in (0.05, 1.0, 8.0, 100.0):
    x = (torch.randn(4, C, 64, 64) * scale)
    gamma = (torch.rand(C, C) * 0.02 + torch.eye(C) * 0.1)
    beta = torch.rand(C) + 0.5
    for inverse in (False, True):
        y = ops.gdn_f32(x.cuda(), gamma.cuda(), beta.cuda(), inverse=inverse).cpu().double()
        norm = torch.einsum("ij,bjhw->bihw", gamma.double(), x.double() ** 2) + beta.double().view(1, -1, 1, 1)
        ref = x.double() * (norm.sqrt() if inverse else norm.rsqrt())
        rel = ((y - ref).abs() / ref.abs().clamp_min(1e-30))
        print("mfma=%s scale %-6g inverse %d  rms rel %.3e  max rel %.3e" % (os.environ.get("LICOS_GDN_F32_MFMA", "1"), scale, inverse,
                                                                             float((rel ** 2).mean().sqrt()), float(rel.max())))
