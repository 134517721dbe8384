"""Latency of the scale-conditioned serial coders alone (no transforms beside them): ns per symbol of
licos_rans_encode_records / licos_rans_decode_image for `streams` streams of 196 608 symbols whose scales follow the
trained hyperprior's row histogram (half the elements on row 0, the rest spread over rows 15-26; profiles/ r03 training log).
  python tools/gc_coder_bench.py [streams] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from licos_amd import ops  # noqa: E402
from licos_amd.entropy_models import EntropyBottleneck, GaussianConditional  # noqa: E402
from licos_amd.models import get_scale_table  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 192 * 32 * 32
dev = torch.device("cuda:0")
gc = GaussianConditional(None).to(dev)
gc.update_scale_table(get_scale_table())
g = torch.Generator(device="cuda").manual_seed(1)
table = get_scale_table().to(dev)
u = torch.rand((B, n), device=dev, generator=g)
rows = torch.where(u < 0.5, torch.zeros_like(u), 15 + torch.floor((u - 0.5) * 2 * 12)).long()
scales = (table[rows] * 1.02).reshape(B, 192, 32, 32).contiguous()
y = (torch.randn((B, n), device=dev, generator=g).reshape(B, 192, 32, 32) * scales).contiguous()
cdf, cdf_len, offset, tab = gc.coder_tables()
image_dev, image_host = gc.coder_image()
bound = gc.lower_bound_scale.bound_value


def timed(fn):
    ts = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[1:])[len(ts[1:]) // 2], out


rec, aux = ops.gc_encode_prepare(y, scales, gc.scale_table, bound, tab, cdf_len, offset, cdf.shape[1])
ms_e, (words, nwords, status) = timed(lambda: ops.rans_encode_records(rec, aux, n // 2 + 64))
host = nwords.cpu().numpy().astype(np.int64)
off = np.zeros(B + 1, dtype=np.int64)
np.cumsum(host * 4, out=off[1:])
off_dev = torch.from_numpy(off).to(dev)
data = ops.rans_compact(words, nwords, off_dev, int(off[-1]))
idx16 = ops.gc_decode_prepare(scales, gc.scale_table, bound, row_hist=gc.row_histogram())
sym = torch.empty((n, B), device=dev, dtype=torch.int32)
ms_prior, _ = timed(lambda: ops.rans_decode_image(data, off_dev, idx16, n, image_dev, image_host, sym, 1, B, B))
rebuilt = gc.note_row_usage()  # the rows this data uses now steer the image's record budget
image_dev, image_host = gc.coder_image()
print("decode with the prior-weighted image: %.2f ms = %.1f ns/symbol; image rebuilt from the observed rows: %s" % (ms_prior, 1e6 * ms_prior / n, rebuilt))
dbg = torch.zeros(32, device=dev, dtype=torch.int32)
ms_d, st = timed(lambda: ops.rans_decode_image(data, off_dev, idx16, n, image_dev, image_host, sym, 1, B, B, status=dbg))
d = dbg[2:22].cpu().view(torch.int64).tolist()
if d[0]:
    print("decode stamps (wave 0): total %d cycles = %.1f per symbol; slow path entered %d times (%.2f%% of symbols), %.1f cycles each = %.1f per symbol; "
          "refill checks %.1f, vmcnt waits %.1f cycles per symbol; quads rolled back %d (%.1f%%), %.0f cycles each = %.1f per symbol"
          % (d[0], d[0] / n, d[1], 100.0 * d[1] / n, d[2] / max(d[1], 1), d[2] / n, d[3] / n, d[4] / n, d[5], 400.0 * d[5] / n,
             d[6] / max(d[5], 1), d[6] / n))
    print("   per symbol: metadata phase %.1f, speculative quad %.1f cycles" % (d[7] / n, d[8] / n))
ok = bool(torch.equal(sym.t().reshape(B, 192, 32, 32), torch.round(y).int()))
print("streams %d x %d symbols: encode %.2f ms = %.1f ns/symbol, decode %.2f ms = %.1f ns/symbol, %.3f bits/symbol, round trip %s"
      % (B, n, ms_e, 1e6 * ms_e / n, ms_d, 1e6 * ms_d / n, 8.0 * off[-1] / (B * n), "ok" if ok else "MISMATCH"))
