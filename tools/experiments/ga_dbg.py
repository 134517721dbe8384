"""Which latents of the fp32 analysis transform miss the element-wise bound, with the fused GDN on and off."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
from oracle import model as om

cin, kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1, sys.argv[2] if len(sys.argv) > 2 else "s2"
import licos_amd
sd = om.perturb_state(om.make_factorized_state(cin, quality=1, seed=42), seed=11)
net = licos_amd.get_model("bmshj2018-factorized", False, cin, 1)
net.load_state_dict(sd)
net = net.to("cuda").eval().set_precision("fp32")
x = om.synthetic_tiles(2, cin, 256, seed=3, kind=kind)
ref = om.forward(x, sd)["y"].double()
with torch.no_grad():
    y = net.g_a(x.cuda()).cpu().double()
d = (y - ref).abs()
bound = 1e-5 * ref.abs() + 1e-6 * float(ref.abs().max())
bad = d > bound
print("fused", os.environ.get("LICOS_GDN_F32_MFMA", "1"), "violations", int(bad.sum()), "of", bad.numel(), "worst ratio", float((d / bound).max()),
      "rel_err", float((y - ref).norm() / ref.norm()), "max|ref|", float(ref.abs().max()))
idx = bad.nonzero()[:8]
for i in idx:
    i = tuple(int(v) for v in i)
    print(i, float(y[i]), float(ref[i]), float(d[i]), float(bound[i]))
# per-layer: x through g_a stage by stage
