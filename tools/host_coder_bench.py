"""Dev tool (CPU only): ns per symbol of the product's host rANS coder on streams drawn from the shipped model's own
tables (licos_amd/weights/factorized_q3_c3.pth.tar): 192 channels x 256 positions per tile, symbols sampled from each
channel's quantised pmf.
  python tools/host_coder_bench.py [tiles=64] [threads=8] [reps=5]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import licos_amd
from licos_amd import ops, checkpoint
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
net = licos_amd.get_model("bmshj2018-factorized", False, 3, 3)
checkpoint.load_checkpoint(os.path.join(os.path.dirname(licos_amd.__file__), "weights", "factorized_q3_c3.pth.tar"), net)
net.update(force=True)
eb = net.entropy_bottleneck
cdf = eb._quantized_cdf.numpy().astype(np.int32)
cl = eb._cdf_length.numpy().astype(np.int32)
off = eb._offset.numpy().astype(np.int32)
table = ops.rans_build_enc_table(cdf, cl)
C, plane = cdf.shape[0], 256
n = C * plane
rng = np.random.default_rng(0)
sym = np.empty((B, C, plane), dtype=np.int32)
for c in range(C):
    L = int(cl[c]) - 1
    pmf = np.diff(cdf[c, : L + 1]).astype(np.float64)
    pmf[-1] = 0  # no escapes in the sample (they are < 1e-3 of the symbols on real tiles)
    pmf /= pmf.sum()
    sym[:, c, :] = rng.choice(L, size=(B, plane), p=pmf) + off[c]
sym = sym.reshape(B, n)
for threads in sorted({1, T}):
    enc, dec = [], []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        out, nb = ops.rans_encode_host(sym, n, plane, cdf, cl, off, table, nthreads=threads)
        t1 = time.perf_counter()
        data = np.concatenate([out[b, : nb[b]] for b in range(B)])
        bo = np.concatenate(([0], np.cumsum(nb))).astype(np.int64)
        t2 = time.perf_counter()
        d, st = ops.rans_decode_host(data, bo, n, plane, cdf, cl, off, B, nthreads=threads)
        t3 = time.perf_counter()
        if r:
            enc.append(t1 - t0)
            dec.append(t3 - t2)
    assert st == 0 and np.array_equal(d, sym)
    e, dd = sorted(enc)[len(enc) // 2], sorted(dec)[len(dec) // 2]
    print("threads %2d  B %d: encode %.3f ms = %.2f ns/symbol/thread, decode %.3f ms = %.2f ns/symbol/thread; %.3f bits/symbol"
          % (threads, B, 1e3 * e, 1e9 * e * threads / (B * n), 1e3 * dd, 1e9 * dd * threads / (B * n), 8.0 * nb.sum() / (B * n)))

# ---- explicit per-symbol rows (the scale hyperprior's y stream: GaussianConditional tables, row = scale index) ----------
hy = licos_amd.get_model("bmshj2018-hyperprior", False, 13, 5)
wf = os.path.join(os.path.dirname(licos_amd.__file__), "weights", "hyperprior_q5_c13.pth.tar")
if os.path.exists(wf):
    checkpoint.load_checkpoint(wf, hy)
hy.update(force=True)
gc_ = hy.gaussian_conditional
gcdf = gc_._quantized_cdf.numpy().astype(np.int32)
gcl = gc_._cdf_length.numpy().astype(np.int32)
goff = gc_._offset.numpy().astype(np.int32)
gtable = ops.rans_build_enc_table(gcdf, gcl)
n2 = 192 * 32 * 32
Bh = max(8, B // 4)
# rows as the trained 13-band model uses them: 73 % row 0, the rest spread over rows 20 - 30
rows = np.where(rng.random((Bh, n2)) < 0.73, 0, rng.integers(20, 31, size=(Bh, n2))).astype(np.int32)
sym2 = np.empty((Bh, n2), dtype=np.int32)
for r_ in np.unique(rows):
    L = int(gcl[r_]) - 1
    pmf = np.diff(gcdf[r_, : L + 1]).astype(np.float64)
    pmf[-1] = 0
    pmf /= pmf.sum()
    m_ = rows == r_
    sym2[m_] = rng.choice(L, size=int(m_.sum()), p=pmf) + goff[r_]
for threads in sorted({1, T}):
    enc, dec = [], []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        out, nb = ops.rans_encode_host(sym2, n2, 0, gcdf, gcl, goff, gtable, indexes=rows, nthreads=threads)
        t1 = time.perf_counter()
        data = np.concatenate([out[b, : nb[b]] for b in range(Bh)])
        bo = np.concatenate(([0], np.cumsum(nb))).astype(np.int64)
        t2 = time.perf_counter()
        d, st = ops.rans_decode_host(data, bo, n2, 0, gcdf, gcl, goff, Bh, indexes=rows, nthreads=threads)
        t3 = time.perf_counter()
        if r:
            enc.append(t1 - t0)
            dec.append(t3 - t2)
    assert st == 0 and np.array_equal(d, sym2)
    e, dd = sorted(enc)[len(enc) // 2], sorted(dec)[len(dec) // 2]
    print("explicit rows: threads %2d  B %d: encode %.3f ms = %.2f ns/symbol/thread, decode %.3f ms = %.2f ns/symbol/thread; %.3f bits/symbol"
          % (threads, Bh, 1e3 * e, 1e9 * e * threads / (Bh * n2), 1e3 * dd, 1e9 * dd * threads / (Bh * n2), 8.0 * nb.sum() / (Bh * n2)))
