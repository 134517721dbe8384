"""Dev tool: the 13-band last synthesis stage alone - row-walking form (csrc/mfma_rows16.hip) against the LDS-patch form
(deconv5x5s2_few16_kernel, csrc/mfma_deconv.hip) on B maps of 128 x 256 x 256.   python tools/last16_probe.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from licos_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
xb = (torch.randn((B, 8, H, H, 16), device=dev, generator=g) * 0.5).to(torch.float16)
w = torch.randn(128, 13, 5, 5, device=dev, generator=g) * 0.05
b = torch.randn(13, device=dev, generator=g)
wr, wf, bp = ops.pack_deconv_w_rows_f16(w), ops.pack_deconv_w_fewch_f16(w), ops.pad_bias(b, 13, dev)
out = torch.empty((B, 13, 2 * H, 2 * H), device=dev, dtype=torch.float32)
gb = B * (128 * H * H * 2 + 13 * 4 * H * H * 4) / 1e9


def timed(fn, reps=8):
    ts = []
    for _ in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


ms_r = timed(lambda: ops.deconv5x5s2_rows_f16(xb, wr, b, 128, 13, clamp01=True, out=out))
a = out.clone()
ms_f = timed(lambda: ops.deconv5x5s2_fewch_f16(xb, wf, bp, 128, 13, clamp01=True, out=out))
err = float((a - out).abs().max())
print("last stage 128 -> 13 at %d^2, %d maps (%.1f GB algorithmic): rows16 %.3f ms = %.2f TB/s (%.3f of 8) | few16 %.3f ms = %.2f TB/s (%.3f); "
      "max |difference| %.2e" % (H, B, gb, ms_r, gb / ms_r, gb / ms_r / 8, ms_f, gb / ms_f, gb / ms_f / 8, err))
