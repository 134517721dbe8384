"""Dev tool: the first analysis stage alone (layout pass + conv + GDN), N timed launches on B tiles of 3 x 256 x 256."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import ops, engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
form = sys.argv[2] if len(sys.argv) > 2 else "rows"
cin = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(B, cin, 256, 256, device=dev, generator=g)
w = torch.randn(128, cin, 5, 5, device=dev, generator=g) * 0.2
bp = ops.pad_bias(torch.zeros(128, device=dev), 128, dev)
gp = engine._packed_gdn(licos_amd.GDN(128).to(dev))
if form == "rows":
    wp = ops.pack_conv_w_first_f16(w)
    prep = lambda: ops.nchw_f32_to_hwc_pad_f16(x)
    conv = lambda xi: ops.conv5x5s2_first_f16(xi, wp, bp, gp, ops.EPI_GDN, B, cin, 128, 256, 256)
elif form == "raw":
    wp = ops.pack_conv_w_first_f16(w)
    prep = lambda: x
    conv = lambda xi: ops.conv5x5s2_first_nchw_f16(xi, wp, bp, gp, ops.EPI_GDN, 128)
else:
    wp = ops.pack_conv_w_s2d_f16(w)
    prep = lambda: ops.nchw_f32_to_s2d_blk16(x)
    conv = lambda xi: ops.conv5x5s2_s2d_f16(xi, wp, bp, gp, ops.EPI_GDN, cin, 128, 256, 256)
tp, tc = [], []
for it in range(25):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    xi = prep()
    e[1].record()
    y = conv(xi)
    e[2].record()
    torch.cuda.synchronize()
    if it >= 5:
        tp.append(e[0].elapsed_time(e[1]))
        tc.append(e[1].elapsed_time(e[2]))
    del xi, y
tp.sort(); tc.sort()
gb = (x.numel() * 4 + B * 128 * 128 * 128 * 2) / 1e9
print("%s B=%d cin=%d: layout pass median %.3f ms, conv+GDN median %.3f ms (min %.3f) -> %.2f TB/s algorithmic over both (%.1f GB)"
      % (form, B, cin, tp[len(tp) // 2], tc[len(tc) // 2], tc[0], gb / (tp[len(tp) // 2] + tc[len(tc) // 2]), gb))
