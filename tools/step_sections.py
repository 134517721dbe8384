"""Dev tool: wall-clock sections of one compress() + decompress() of the bench workload (codec.trace.timings; every mark
synchronises the device, so the sections do not overlap the way the un-instrumented step does)."""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import synthetic, codec, checkpoint
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
net.update(force=True)
net.chunk = 4096
x = synthetic.tiles(16384, 3, 256, seed=5, device=dev)
with torch.no_grad():
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = net.compress(x); torch.cuda.synchronize(); t1 = time.perf_counter()
        d = net.decompress(c["strings"], c["shape"]); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("plain: compress %.1f ms, decompress %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    codec.trace.timings = {}
    c = net.compress(x); d = net.decompress(c["strings"], c["shape"]); torch.cuda.synchronize()
    for k, v in codec.trace.timings.items():
        print("  %-40s %7.1f ms" % (k, 1e3 * v))
