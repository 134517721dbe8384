"""Dev probe: is the small-batch analysis chain launch-bound, and what does a captured hipGraph (torch.cuda.CUDAGraph over
the library's own launches) save?   python tools/graph_probe.py [B ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import checkpoint, synthetic, ops, engine
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
eb = net.entropy_bottleneck
med = eb.medians_vec()


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


with torch.no_grad():
    for b in [int(a) for a in sys.argv[1:]] or [1, 16, 64]:
        x = synthetic.tiles(b, 3, 256, seed=5, kind="aid", device=dev)
        nsym = 192 * 16 * 16
        stage = torch.empty((b, nsym), dtype=torch.int16, pin_memory=True)
        flag = torch.zeros(1, device=dev, dtype=torch.int32)

        def eager():
            y = net.g_a(x)
            hsym = torch.empty((b, nsym), device=dev, dtype=torch.int16)
            ops.eb_symbols16(y.contiguous(), med, hsym, flag)
            stage.copy_(hsym, non_blocking=True)

        t_eager = timeit(eager)
        xs = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                y = net.g_a(xs)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = net.g_a(xs)
            hsym = torch.empty((b, nsym), device=dev, dtype=torch.int16)
            ops.eb_symbols16(y.contiguous(), med, hsym, flag)
            stage.copy_(hsym, non_blocking=True)

        def graphed():
            xs.copy_(x)
            g.replay()

        ref = stage.clone()
        eager(); torch.cuda.synchronize(); a = stage.clone()
        graphed(); torch.cuda.synchronize()
        same = bool(torch.equal(a, stage))
        t_graph = timeit(graphed)
        # GPU time of the chain alone (events around the eager chain)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); eager(); e1.record(); torch.cuda.synchronize()
        print("B %3d: eager %.3f ms, graph %.3f ms (same symbols: %s), eager span on the GPU %.3f ms" % (b, t_eager, t_graph, same, e0.elapsed_time(e1)), flush=True)
