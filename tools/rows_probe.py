"""Dev tool: the last synthesis stage alone (row-walking / scatter form), N timed launches on B tiles of 128 x 128 x 128."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from licos_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
form = sys.argv[2] if len(sys.argv) > 2 else "rows"
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
xb = (torch.randn(B, 8, 128, 128, 16, device=dev, generator=g) * 0.5).half()
w = torch.randn(128, cout, 5, 5, device=dev, generator=g) * 0.05
b = torch.zeros(cout, device=dev)
pack, run = (ops.pack_deconv_w_rows_f16, ops.deconv5x5s2_rows_f16) if form == "rows" else (ops.pack_deconv_w_scatter_f16, ops.deconv5x5s2_scatter_f16)
wp = pack(w)
out = torch.empty(B, cout, 256, 256, device=dev)
ts = []
for it in range(25):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(xb, wp, b, 128, cout, clamp01=True, out=out, in_xsplit=True)
    e1.record()
    torch.cuda.synchronize()
    if it >= 5:
        ts.append(e0.elapsed_time(e1))
ts.sort()
gb = (xb.numel() * 2 + out.numel() * 4) / 1e9
print("%s B=%d cout=%d: median %.3f ms  min %.3f  max %.3f  -> %.2f TB/s algorithmic (%.1f GB)" % (form, B, cout, ts[len(ts) // 2], ts[0], ts[-1], gb / ts[len(ts) // 2], gb))
