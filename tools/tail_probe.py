"""Dev probe: the timeline of the end of a large factorized compress call (codec.trace.host_trace): when the device chunks are
drained, what each host sub-chunk waits for and takes, when the host is done and when the last device chunk is.
  python tools/tail_probe.py [tiles = 16384]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import codec, checkpoint, synthetic
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
net.chunk = 4096
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x = synthetic.tiles(B, 3, 256, seed=5, kind="aid", device=dev)
with torch.no_grad():
    for it in range(5):
        codec.trace.host_trace = [] if it == 4 else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = net.compress(x)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tr, codec.trace.host_trace = codec.trace.host_trace, ([] if it == 4 else None)
        d = net.decompress(c["strings"], c["shape"])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        trd, codec.trace.host_trace = codec.trace.host_trace, None
        print("iter %d: compress %.2f ms decompress %.2f ms" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
for e in tr:
    if e[0] == "enc":
        print("  host sub-chunk %4d tiles at %7.2f ms: waited %.2f, coded %.2f, strings %.2f" % (e[1], 1e3 * (e[5] - t0), e[2], e[3], e[4]))
    else:
        print("  %-16s %5d at %7.2f ms" % (e[0], e[1], 1e3 * (e[2] - t0)))
print("  (end of compress at %.2f ms)" % (1e3 * (t1 - t0)))
print("decompress (from %.2f ms):" % 0.0)
for e in trd:
    if e[0] == "dec":
        print("  host sub-chunk %4d tiles at %7.2f ms: joined %.2f, decoded %.2f, queued its upload + transforms in %.2f" % (e[1], 1e3 * (e[5] - t1), e[2], e[3], e[4]))
    else:
        print("  %-20s %5d at %7.2f ms" % (e[0], e[1], 1e3 * (e[2] - t1)))
print("  (end of decompress at %.2f ms)" % (1e3 * (t2 - t1)))
