"""Dev tool: time the 128->128 transposed-conv stage (+IGDN) alone, random blk16 input.
  python tools/d8_probe.py [tiles=1024] [hw=64] [reps=5]"""
import sys, os, torch, torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from licos_amd import engine
from licos_amd.layers import GDN, deconv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
plain = len(sys.argv) > 4 and sys.argv[4] == "nogdn"  # no (I)GDN epilogue: K loop + stores only
dev = torch.device("cuda:0")
torch.manual_seed(0)
# the probed stage sits between a producer and a consumer of its own kind, as in g_s (x-split layouts on both sides)
mods = [deconv(128, 128), GDN(128, inverse=True), deconv(128, 128), GDN(128, inverse=True), deconv(128, 3)]
if plain:
    mods = [m for m in mods if not isinstance(m, GDN)]
seq = nn.Sequential(*mods).to(dev).eval()
x = torch.randn(B, 8, hw // 2, hw // 2, 16, device=dev).half()
with torch.no_grad():
    for it in range(2 + reps):
        if it == 2:
            engine.stage_events = {}
        engine.run_chain_fp16(seq, x_blk=x)
    torch.cuda.synchronize()
k = [k for k in engine.stage_events if k[3] == hw][0]
v = engine.stage_events[k]
ms = sorted(a.elapsed_time(b) for a, b in v)
fl = 2.0 * (2 * hw) ** 2 * 25 / 4 * 128 * 128 * B
print("%s deconv8=%s so=%s: median %.3f ms min %.3f  -> %.0f TFLOP/s (%.3f of 2.5 PF)" % (
    k[:5], os.environ.get("LICOS_DECONV8", "1"), os.path.basename(os.environ.get("LICOS_HIP_SO", "product")),
    ms[len(ms) // 2], ms[0], fl / ms[len(ms) // 2] / 1e9, fl / ms[len(ms) // 2] / 1e9 / 2500))
