"""Dev tool: host-side section breakdown of one compress()+decompress() step (synchronising)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import codec, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
net.chunk = chunk
with torch.no_grad():
    synthetic.make_trained_like(net, seed=0)
    x = synthetic.tiles(B, 3, 256, seed=1, device=dev)
    for it in range(3):
        codec.trace.timings = {} if it == 2 else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = net.compress(x)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        d = net.decompress(c["strings"], c["shape"])
        torch.cuda.synchronize(); t2 = time.perf_counter()
print("B=%d chunk=%d: compress %.1f ms, decompress %.1f ms" % (B, chunk, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
for k, v in codec.trace.timings.items():
    print("  %-36s %8.2f ms" % (k, 1e3 * v))
print("bytes per tile: %.0f" % (sum(len(s) for s in c["strings"][0]) / B))
