"""Dev tool: what plain torch streaming kernels sustain on this box - a pure write (fill_), a copy (read + write) and a
widening conversion with a stage's read : write ratio - as reference rates beside the HBM-bound stages' `frac_hbm`.
  python tools/hbm_roof_probe.py [GB]"""
import sys

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device("cuda:0")
n = int(gb * 1e9) // 4
a = torch.empty(n, device=dev, dtype=torch.float32)
b = torch.empty(n, device=dev, dtype=torch.float32)
h = torch.empty(n, device=dev, dtype=torch.float16)


def timed(fn, reps=7):
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


ms = timed(lambda: a.fill_(1.0))
print("fill_ fp32, %.1f GB written: %.3f ms = %.2f TB/s" % (4e-9 * n, ms, 4e-9 * n / ms))
ms = timed(lambda: b.copy_(a))
print("copy_ fp32, %.1f GB read + %.1f GB written: %.3f ms = %.2f TB/s" % (4e-9 * n, 4e-9 * n, ms, 8e-9 * n / ms))
ms = timed(lambda: h.copy_(a))
print("fp32 -> fp16, %.1f GB read + %.1f GB written: %.3f ms = %.2f TB/s" % (4e-9 * n, 2e-9 * n, ms, 6e-9 * n / ms))
ms = timed(lambda: a.copy_(h))
print("fp16 -> fp32, %.1f GB read + %.1f GB written: %.3f ms = %.2f TB/s" % (2e-9 * n, 4e-9 * n, ms, 6e-9 * n / ms))
ms = timed(lambda: torch.sum(a))
print("sum fp32, %.1f GB read: %.3f ms = %.2f TB/s" % (4e-9 * n, ms, 4e-9 * n / ms))
