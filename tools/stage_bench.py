"""Stage timing helper (dev tool): times each MFMA stage of g_a/g_s on B tiles."""
import sys, os, json, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import engine, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3  # input channels (3 RGB, 1 split band, 13 merged bands)
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, C, 3).to(dev).eval().set_precision("fp16")
x = synthetic.tiles(B, C, 256, seed=1, device=dev)
with torch.no_grad():
    for it in range(3):
        engine.stage_events = {} if it == 2 else None
        y = net.g_a(x)
        xh = net.g_s(y)
    torch.cuda.synchronize()
ev = engine.stage_events
tot = 0
for k, v in ev.items():
    ms = sum(a.elapsed_time(b) for a, b in v) / len(v)
    tot += ms
    print("%-28s %8.3f ms" % ("%s_%d_%d_%dx%d" % k[:5], ms))
print("total %.3f ms for %d tiles -> %.1f us/tile" % (tot, B, 1e3 * tot / B))
