"""Dev tool: encode / decode wall time against the host's share of the serial coding (codec.host_share), and the PCIe
rates its symbol traffic sees.   python tools/split_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import codec, checkpoint, synthetic, ops
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval().set_precision("fp16")
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
net.chunk = 4096
# PCIe
for mb in (25, 150):
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    hbuf = torch.empty(mb << 20, dtype=torch.uint8, pin_memory=True)
    for name, fn in (("D2H", lambda: hbuf.copy_(d, non_blocking=True)), ("H2D", lambda: d.copy_(hbuf, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        print("%s %d MB pinned: %.1f GB/s" % (name, mb, 5 * mb / 1024 / (time.perf_counter() - t0)))
real_share = codec.placement.host_share


def run(B, share_enc, share_dec, reps=4):
    x = synthetic.tiles(B, 3, 256, seed=5, kind="aid", device=dev)
    codec.placement.host_share = lambda batch, direction: min(batch, share_enc if direction == "enc" else share_dec)
    enc, dec = [], []
    with torch.no_grad():
        for i in range(reps + 1):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            c = net.compress(x)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            d = net.decompress(c["strings"], c["shape"])
            torch.cuda.synchronize(); t2 = time.perf_counter()
            if i:
                enc.append(1e3 * (t1 - t0)); dec.append(1e3 * (t2 - t1))
    enc.sort(); dec.sort()
    print("B %5d host share enc %4d dec %4d: encode %.2f ms, decode %.2f ms" % (B, share_enc, share_dec, enc[len(enc) // 2], dec[len(dec) // 2]), flush=True)
    return c


for zc in (True, False):
    codec.config.zero_copy = zc
    print("ZERO_COPY", zc)
    for B in (16, 64, 256):
        run(B, B, B, reps=8)
    for se, sd in ((1024, 1024), (1024, 512), (1024, 384), (768, 384)):
        run(1024, se, sd, reps=5)
codec.config.zero_copy = True
for B in (4096, 16384):
    for se, sd in ((0, 0), (512, 256), (1024, 512), (1536, 768)):
        run(B, se, sd, reps=4 if B < 16384 else 3)
print("default shares:", real_share(1024, "enc"), real_share(1024, "dec"), real_share(16384, "enc"), real_share(16384, "dec"))
# section timings of one B = 1024 call at the default split
codec.placement.host_share = real_share
x = synthetic.tiles(1024, 3, 256, seed=5, kind="aid", device=dev)
with torch.no_grad():
    net.decompress(*[net.compress(x)[k] for k in ("strings", "shape")])
    codec.trace.timings = {}
    c = net.compress(x); net.decompress(c["strings"], c["shape"])
    print({k: round(1e3 * v, 2) for k, v in codec.trace.timings.items()})
    codec.trace.timings = None
