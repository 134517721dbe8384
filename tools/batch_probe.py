"""Dev probe: median encode / decode wall time of the factorized codec at a few batch sizes (bench.py's `batches` cells),
for A/B runs of a switch (LICOS_SYM16=0, LICOS_HOST_SPLIT=0, ...).   python tools/batch_probe.py [sizes ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import checkpoint, synthetic
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
net.chunk = int(os.environ.get("PROBE_CHUNK", "4096"))
sizes = [int(a) for a in sys.argv[1:]] or [16, 64, 256, 1024]
x = synthetic.tiles(max(sizes), 3, 256, seed=5, kind="aid", device=dev)
with torch.no_grad():
    for b in sizes:
        xb = x[:b].contiguous()
        enc, dec = [], []
        for it in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c = net.compress(xb)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            d = net.decompress(c["strings"], c["shape"])
            torch.cuda.synchronize(); t2 = time.perf_counter()
            if it >= 3:
                enc.append(1e3 * (t1 - t0)); dec.append(1e3 * (t2 - t1))
        enc.sort(); dec.sort()
        e, dd = enc[len(enc) // 2], dec[len(dec) // 2]
        print("B %5d: encode %.3f ms decode %.3f ms -> %.0f tiles/s" % (b, e, dd, 1e3 * b / (e + dd)), flush=True)
