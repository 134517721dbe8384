"""Dev probe of the strict-parity (fp32) path of the headline codec: wall time of compress / decompress through the chunk
pipeline.   python tools/fp32_probe.py [tiles] [chunk]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd  # noqa: E402
from licos_amd import synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp32")
with torch.no_grad():
    synthetic.make_trained_like(net, seed=0)
net.chunk = chunk
x = synthetic.tiles(B, 3, 256, seed=7, device=dev)
with torch.no_grad():
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c = net.compress(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        d = net.decompress(c["strings"], c["shape"])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("iter %d: compress %.1f ms, decompress %.1f ms, %.0f tiles/s" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), B / (t2 - t0)), flush=True)
