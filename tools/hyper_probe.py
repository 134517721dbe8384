"""Dev probe of the scale-hyperprior codec (BASELINE config 5): wall time of compress / decompress, device time of the
serial coder launches, per-stage times.   python tools/hyper_probe.py [tiles] [chunk = 2048] [--synthetic]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd  # noqa: E402
from licos_amd import checkpoint, codec, engine, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
torch.manual_seed(42)
net = licos_amd.get_model("bmshj2018-hyperprior", False, 13, 5).to(dev).eval().set_precision("fp16")
w = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "licos_amd", "weights", "hyperprior_q5_c13.pth.tar")
if os.path.exists(w) and "--synthetic" not in sys.argv:
    checkpoint.load_checkpoint(w, net)
else:
    with torch.no_grad():
        synthetic.make_trained_like(net, seed=0)
net.chunk = chunk
x = synthetic.tiles(B, 13, 512, seed=300, kind="s2-merged", device=dev)
with torch.no_grad():
    iters = int(os.environ.get("PROBE_ITERS", "3"))
    for it in range(iters):
        if it >= 2:
            if it > 2:
                print("   coder ms:", {k: ["%.1f" % e0.elapsed_time(e1) for e0, e1 in v] for k, v in codec.trace.coder_events.items()},
                      "stages %.1f ms" % sum(e0.elapsed_time(e1) for v in engine.stage_events.values() for e0, e1 in v))
            codec.trace.coder_events, engine.stage_events = {}, {}
        torch.cuda.synchronize()
        codec.trace.host_trace = [] if "--trace" in sys.argv else None
        t0 = time.perf_counter()
        c = net.compress(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        d = net.decompress(c["strings"], c["shape"])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if codec.trace.host_trace is not None:
            tr, codec.trace.host_trace = codec.trace.host_trace, None
            print("  enc factor %.2f" % codec.placement.rate.factor["enc"])
            for e in tr:
                print("  ", e[:2], ["%.1f" % (1e3 * (v - t0)) if e[0] not in ("hyper-enc", "hyper-dec") else v for v in e[2:]])
        ms_ = torch.cuda.memory_stats()
        print("   reserved %.1f GB, allocated peak %.1f GB, alloc retries %d, hipMalloc calls %d, hipFree calls %d" % (
            ms_["reserved_bytes.all.current"] / 2**30, ms_["allocated_bytes.all.peak"] / 2**30, ms_["num_alloc_retries"],
            ms_["segment.all.allocated"], ms_["segment.all.freed"]))
        print("iter %d: compress %.1f ms, decompress %.1f ms, %.0f tiles/s" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), B / (t2 - t0)), flush=True)
nsym = {"y": 192 * 32 * 32, "z": 128 * 8 * 8}
for k, evs in codec.trace.coder_events.items():
    ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    print(k, ["%.2f" % m for m in ms], "ns/symbol %.1f" % (1e6 * sorted(ms)[len(ms) // 2] / nsym[k[0]]))
tot = 0.0
for k, evs in engine.stage_events.items():
    ms = sum(e0.elapsed_time(e1) for e0, e1 in evs)
    tot += ms
    print("%s_%d_%d_%dx%d_b%d" % k[:6], "%.2f ms total over %d launches" % (ms, len(evs)))
print("stage total %.1f ms" % tot)
nbytes = sum(len(s) for lst in c["strings"] for s in lst)
print("bpp %.4f psnr %.2f" % (nbytes * 8.0 / (B * 512 * 512), licos_amd.metrics.compute_psnr(d["x_hat"], x)))
