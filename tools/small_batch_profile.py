"""Dev tool: where a small-batch compress / decompress spends its wall time - cProfile of 300 round trips of B tiles plus the
HIP-event time of every transform stage.   python tools/small_batch_profile.py [B]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd  # noqa: E402
from licos_amd import checkpoint, engine, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
x = synthetic.tiles(B, 3, 256, seed=5, kind="aid", device=dev)
with torch.no_grad():
    for _ in range(20):
        c = net.compress(x)
        d = net.decompress(c["strings"], c["shape"])
    torch.cuda.synchronize()
    engine.stage_events = {}
    for _ in range(20):
        c = net.compress(x)
        d = net.decompress(c["strings"], c["shape"])
    torch.cuda.synchronize()
    ev, engine.stage_events = engine.stage_events, None
    tot = 0.0
    for key, evs in ev.items():
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2]
        tot += ms
        print("stage %-40s %.1f us" % ("%s_%d_%d_%dx%d_b%d" % key[:6], 1e3 * ms))
    print("sum of the stages' kernels: %.1f us" % (1e3 * tot))
    for name, fn in (("compress", lambda: net.compress(x)), ("decompress", lambda: net.decompress(c["strings"], c["shape"]))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        torch.cuda.synchronize()
        print("%s: %.1f us per call" % (name, 1e6 * (time.perf_counter() - t0) / 300))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(300):
            fn()
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats("tottime").print_stats(18)
