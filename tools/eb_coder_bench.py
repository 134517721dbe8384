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
# the words sit at the END of each stream's column: compare the live part
same = bool(torch.equal(n0, n1)) and all(
    torch.equal(w0[cap - int(n0[b]):, b], w1[cap - int(n0[b]):, b]) for b in range(0, B, max(1, B // 64)))
print("plane encoder, %d streams x %d symbols: [position][stream] %.3f ms = %.1f ns/symbol | [stream][position] %.3f ms = %.1f ns/symbol; "
      "%s, %.3f bit per symbol" % (B, nsym, ms_pm, 1e6 * ms_pm / nsym, ms_sm, 1e6 * ms_sm / nsym, "identical words" if same else "WORDS DIFFER",
                                   32.0 * float(n0.float().mean()) / nsym))
if not same:  # which of the two is wrong: decode both with the image decoder
    for name, (wd, nw) in (("[position][stream]", (w0, n0)), ("[stream][position]", (w1, n1))):
        bo = torch.zeros(B + 1, device=dev, dtype=torch.int64)
        bo[1:] = torch.cumsum(nw.to(torch.int64) * 4, 0)
        dat = ops.rans_compact(wd, nw, bo, int(bo[-1]))
        out = torch.empty((B, nsym), device=dev, dtype=torch.int32)
        st = torch.zeros(1, device=dev, dtype=torch.int32)
        ops.rans_decode_image(dat, bo, eb.channel_rows(plane), nsym, eb.coder_image()[0], eb.coder_image()[1], out, nsym, 1, B, status=st, rows_shared=True)
        bad = (out != sym_sm).any(dim=1)
        print("  encoder %s: %d of %d streams decode wrongly (status %d); first bad streams %s, first bad position of the first %s"
              % (name, int(bad.sum()), B, int(st), bad.nonzero().flatten()[:8].tolist(),
                 (out[bad][0] != sym_sm[bad][0]).nonzero().flatten()[:4].tolist() if bool(bad.any()) else "-"))

# ---- the decoders on the same streams: image decoder (csrc/rans_gc.hip), plane decoder (csrc/rans.hip), both layouts ----
nb = (n0.to(torch.int64) * 4)
byte_off = torch.zeros(B + 1, device=dev, dtype=torch.int64)
byte_off[1:] = torch.cumsum(nb, 0)
data = ops.rans_compact(w0, n0, byte_off, int(byte_off[-1]))
image = eb.coder_image()
rows = eb.channel_rows(plane)


def dec_plane(sm):
    out = torch.empty((B, nsym) if sm else (nsym, B), device=dev, dtype=torch.int32)
    st = ops.rans_decode_batch(data, byte_off, nsym if sm else 1, 1 if sm else B, nsym, plane, cdf, cdf_len, offset, out, B, off_offset=0)
    return out, st


def dec_image(sm):
    out = torch.empty((B, nsym) if sm else (nsym, B), device=dev, dtype=torch.int32)
    st = torch.zeros(1, device=dev, dtype=torch.int32)
    ops.rans_decode_image(data, byte_off, rows, nsym, image[0], image[1], out, nsym if sm else 1, 1 if sm else B, B, status=st,
                          rows_shared=True)
    return out, st


for name, fn, sm in (("image decoder [position][stream]", dec_image, False), ("image decoder [stream][position]", dec_image, True),
                     ("plane decoder [position][stream]", dec_plane, False), ("plane decoder [stream][position]", dec_plane, True)):
    ms, (out, st) = timed(lambda: fn(sm))
    ok = int(st) == 0 and torch.equal(out if sm else out.t(), sym_sm)
    print("%s: %.3f ms = %.1f ns/symbol, symbols %s" % (name, ms, 1e6 * ms / nsym, "identical" if ok else "DIFFER (status %d)" % int(st)))

import ctypes  # noqa: E402
from licos_amd import _lib  # noqa: E402
lib = _lib.load()
if hasattr(lib, "licos_debug_dec_stamps"):  # (a -DLICOS_STAMPS build of rans.hip, tools/ab_build.sh)
    buf = (ctypes.c_ulonglong * 4)()
    lib.licos_debug_dec_stamps(None, 1)
    dec_plane(True)
    torch.cuda.synchronize()
    lib.licos_debug_dec_stamps(buf, 0)
    wgs = max(1, buf[2])
    print("plane decoder stamps (wave 0 of %d workgroups, s_memtime ticks per workgroup): tables %.0f, symbol loops %.0f; per channel %.0f / %.0f, "
          "per symbol %.1f" % (wgs, buf[0] / wgs, buf[1] / wgs, buf[0] / wgs / C, buf[1] / wgs / C, buf[1] / wgs / nsym))
