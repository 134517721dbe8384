"""Dev probe: where a 1024-tile call's encode goes (codec.trace.host_trace) and what the queueing costs."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import codec, checkpoint, synthetic, ops
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
net.chunk = 4096
x = synthetic.tiles(1024, 3, 256, seed=5, kind="aid", device=dev)
with torch.no_grad():
    for it in range(4):
        codec.trace.host_trace = [] if it == 3 else None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c = net.compress(x)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        d = net.decompress(c["strings"], c["shape"])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("iter %d: encode %.2f ms decode %.2f ms" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
        if it == 3:
            tq = time.perf_counter()
            codec.trace.timings = {}
            net.decompress(c["strings"], c["shape"])
            print("decode sections (each mark synchronises):", {k: round(1e3 * v, 2) for k, v in codec.trace.timings.items()})
            codec.trace.timings = None
            for share in (0, 192, 256, 320, 391, 450, 520, 600):
                real = codec.placement.host_share
                codec.placement.host_share = lambda b, d, s_=share: (min(b, s_) if d == "dec" else real(b, d))
                ts = []
                for _ in range(4):
                    torch.cuda.synchronize(); ta = time.perf_counter()
                    net.decompress(c["strings"], c["shape"])
                    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - ta))
                print("decode with host share %d: median %.2f ms (min %.2f)" % (share, sorted(ts)[2], min(ts)))
                codec.placement.host_share = real
    print(codec.trace.host_trace)
    codec.trace.host_trace = None
    # the GPU side alone: transforms + quantise of the same sub-chunks, no host work
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for (s0, m) in codec.ramp(1024, 32, 256):
        y = net.g_a(x[s0:s0 + m])
    torch.cuda.synchronize()
    print("g_a over the ramp: %.2f ms; one launch of 1024: " % (1e3 * (time.perf_counter() - t0)), end="")
    t0 = time.perf_counter(); y = net.g_a(x); torch.cuda.synchronize(); print("%.2f ms" % (1e3 * (time.perf_counter() - t0)))
    import numpy as np
    eb = net.entropy_bottleneck
    hcdf, hlen, hoff, htable = eb.coder_tables_host()
    sym = torch.empty((256, 49152), device=dev, dtype=torch.int32)
    ops.eb_quantize(y[:256].contiguous(), eb.medians_vec(), "symbols", symbols=sym, sym_stride_b=49152, sym_stride_i=1)
    h = torch.empty((256, 49152), dtype=torch.int32, pin_memory=True); h.copy_(sym); torch.cuda.synchronize()
    for k in range(3):
        t0 = time.perf_counter(); out, nb = ops.rans_encode_host(h.numpy(), 49152, 256, hcdf, hlen, hoff, htable); t1 = time.perf_counter()
        ss = [out[i, : int(nb[i])].tobytes() for i in range(256)]; t2 = time.perf_counter()
        print("host encode of 256 tiles from pinned: %.2f ms, strings %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    hp = h.numpy().copy()
    t0 = time.perf_counter(); out, nb = ops.rans_encode_host(hp, 49152, 256, hcdf, hlen, hoff, htable); print("from pageable copy: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
