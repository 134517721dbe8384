import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import licos_amd
m = licos_amd.GDN(128).cuda()
with torch.no_grad():
    m.gamma.add_(0.01 * torch.rand(128, 128).cuda())
beta, gamma = m.effective()
p = m.packed_f32split()
g = p[:65536].view(torch.float16).view(4, 4, 2, 2, 64, 8).float().cpu()
b = p[65536:].view(torch.float32).cpu()
print("beta ok", bool(torch.equal(b, beta.cpu())), b[:4])
err = 0.0
for it in range(4):
    for jt in range(4):
        for s in range(2):
            for lane in range(64):
                for el in range(8):
                    i = 32 * it + (lane & 31)
                    j = 32 * jt + 16 * s + 8 * (el >> 2) + 4 * (lane >> 5) + (el & 3)
                    v = float(g[it, jt, s, 0, lane, el]) + float(g[it, jt, s, 1, lane, el]) / 2048.0
                    err = max(err, abs(v - 256.0 * float(gamma[i, j])))
print("max abs err of hi + lo / 2048 against 256 gamma:", err, "max 256 gamma", float(gamma.max()) * 256)
