"""Dev tool: the entropy bottleneck's plane encoder alone (licos_rans_encode_batch, csrc/rans.hip) on B streams of 192 x 16 x 16 symbols
drawn from the shipped factorized model's tables, for both symbol layouts - [position][stream] (what licos_eb_quantize writes) and
[stream][position] (what licos_conv5x5s2_f16_symbols writes) - with the words compared.
  python tools/eb_coder_bench.py [streams] [reps]"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import licos_amd  # noqa: E402
from licos_amd import checkpoint, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3).to(dev).eval()
wf = os.path.join(ROOT, "licos_amd", "weights", "factorized_q3_c3.pth.tar")
if os.path.exists(wf):
    checkpoint.load_checkpoint(wf, net)
net.update(force=True)
eb = net.entropy_bottleneck
cdf, cdf_len, offset, table = eb.coder_tables()
C, plane = cdf.shape[0], 256
nsym = C * plane
# the symbols of real (synthetic-tile) latents through the shipped model: what the bench codes
from licos_amd import engine, synthetic  # noqa: E402
net.set_precision("fp16")
x = synthetic.tiles(B, 3, 256, seed=5, device=dev)
with torch.no_grad():
    sy = engine.run_chain_fp16(net.g_a, x=x, symbols=(eb.medians_vec(), None))
sym_sm = sy.reshape(B, nsym).contiguous()   # [stream][position]
sym_pm = sym_sm.t().contiguous()            # [position][stream]
del x, sy
cap = nsym // 2 + 64


def timed(fn):
    ts = []
    out = None
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[1:])
    return ts[len(ts) // 2], out


ms_pm, (w0, n0, s0) = timed(lambda: ops.rans_encode_batch(sym_pm, 1, B, nsym, plane, cdf, cdf_len, offset, table, cap, B))
ms_sm, (w1, n1, s1) = timed(lambda: ops.rans_encode_batch(sym_sm, nsym, 1, nsym, plane, cdf, cdf_len, offset, table, cap, B))
assert int(s0) == 0 and int(s1) == 0
assert torch.equal(n0, n1)
# the words sit at the END of each stream's column: compare the live part
mx = int(n0.max())
assert torch.equal(w0[cap - mx:], w1[cap - mx:]) or all(
    torch.equal(w0[cap - int(n0[b]):, b], w1[cap - int(n0[b]):, b]) for b in range(0, B, max(1, B // 64)))
print("plane encoder, %d streams x %d symbols: [position][stream] %.3f ms = %.1f ns/symbol | [stream][position] %.3f ms = %.1f ns/symbol; "
      "identical words, %.3f bit per symbol" % (B, nsym, ms_pm, 1e6 * ms_pm / nsym, ms_sm, 1e6 * ms_sm / nsym, 32.0 * float(n0.float().mean()) / nsym))
