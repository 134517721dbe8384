"""Chunk-pipelined encode / decode of a tile batch for the 16-bit path.

The rANS recurrence is sequential inside a stream, so a coder launch is latency-bound: one lane per
tile, a few waves in total, a fixed ~N_symbols x chain-latency no matter how many tiles ride along.
The transforms are throughput-bound and fill the chip.  The two therefore overlap almost for free:
the batch is cut into chunks; while the MFMA kernels of chunk k+1 run on the main stream, the coder
kernel of chunk k runs on a side stream (encode), and symmetrically the decoder of chunk k+1 runs
under the synthesis transform of chunk k.  One device->host hand-over of the stream lengths and one
of the packed bytes per call; the byte strings are CompressAI's, one per tile.
"""
import numpy as np
import torch

from . import engine, ops

_side_streams = {}

# Optional host-side section timing (tools/profile_step.py): when a dict, every section boundary
# synchronises the device and accumulates wall-clock seconds.  None in production.
timings = None


class _Section:
    def __init__(self):
        self.t = None

    def mark(self, name):
        if timings is None:
            return
        import time
        torch.cuda.synchronize()
        now = time.perf_counter()
        if self.t is not None:
            timings[name] = timings.get(name, 0.0) + (now - self.t)
        self.t = now


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


class PackedStrings(list):
    """The list of per-tile byte strings ``compress`` returns, which also remembers the page-locked
    host buffer the strings were cut from.  ``decompress`` uploads that buffer directly instead of
    re-joining thousands of small byte objects (the strings themselves are ordinary ``bytes``)."""

    def __init__(self, strings, packed_host, byte_off):
        super().__init__(strings)
        self.packed_host = packed_host  # pinned uint8 tensor
        self.byte_off = byte_off        # np.int64 [B+1]

    def still_packed(self):
        if len(self) + 1 != self.byte_off.size:
            return False
        # cheap integrity check: lengths must still match the offsets (a caller may have edited the list)
        n = len(self)
        for i in (0, n // 2, n - 1):
            if n and len(self[i]) != int(self.byte_off[i + 1] - self.byte_off[i]):
                return False
        return True


def _chunks(total, size):
    return [(s, min(size, total - s)) for s in range(0, total, size)]


def compress_fp16(net, x, chunk=1024, cap_words=None):
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, table = eb.coder_tables()
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("licos_amd: compress expects a float32 (B, C, H, W) tensor")
    x = x.contiguous()
    B = x.shape[0]
    dev = x.device
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    med = eb.medians_vec()
    sec = _Section()
    sec.mark("c.start")
    sym = None
    keep = []  # every tensor the side stream touches stays referenced until the final sync
    shape = None
    per_chunk = []
    for (s0, n) in _chunks(B, chunk):
        y = net.g_a(x[s0:s0 + n])  # MFMA chain, main stream
        if sym is None:
            shape = tuple(y.shape[-2:])
            nsym, plane = y[0].numel(), y[0, 0].numel()
            sym = torch.empty((nsym, B), device=dev, dtype=torch.int32)
            if cap_words is None:
                cap_words = nsym // 2 + 64
        ops.eb_quantize(y, med, "symbols", symbols=sym, sym_stride_b=1, sym_stride_i=B, sym_offset=s0)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            words, nwords, status = ops.rans_encode_batch(sym, 1, B, nsym, plane, cdf, cdf_len, offset, table, cap_words,
                                                          n, sym_offset=s0)
        keep.append((y, words, nwords, status))
        per_chunk.append((s0, n, words, nwords, status))
    done = torch.cuda.Event()
    done.record(side)
    main.wait_event(done)
    sec.mark("c.transforms+encode (device)")
    meta = torch.cat([t for (_, _, _, nw, st) in per_chunk for t in (nw, st)]).cpu().numpy()  # one D2H + sync
    counts = np.empty(B, dtype=np.int64)
    pos = 0
    overflow = False
    for (s0, n, _, _, _) in per_chunk:
        counts[s0:s0 + n] = meta[pos:pos + n]
        overflow |= bool(meta[pos + n])
        pos += n + 1
    if overflow:
        if cap_words >= 2 * nsym + 8:
            raise RuntimeError("licos_amd: rANS scratch overflow at worst-case capacity")
        del keep, per_chunk
        return compress_fp16(net, x, chunk=chunk, cap_words=2 * nsym + 8)
    sec.mark("c.lengths D2H")
    byte_off = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(counts * 4, out=byte_off[1:])
    off_dev = torch.from_numpy(byte_off).to(dev)
    packed = torch.empty(max(int(byte_off[-1]), 4), device=dev, dtype=torch.uint8)
    for (s0, n, words, nwords, _) in per_chunk:
        ops.rans_compact(words, nwords, off_dev, 0, out=packed, off_offset=s0)
    sec.mark("c.compact")
    host_t = torch.empty(packed.numel(), dtype=torch.uint8, pin_memory=True)
    host_t.copy_(packed, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    host = host_t.numpy()
    sec.mark("c.bytes D2H")
    mv = memoryview(host)
    strings = PackedStrings([bytes(mv[byte_off[i]:byte_off[i + 1]]) for i in range(B)], host_t, byte_off)
    sec.mark("c.python bytes objects")
    return {"strings": [strings], "shape": torch.Size(shape)}


def decompress_fp16(net, strings, shape, chunk=1024):
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, _ = eb.coder_tables()
    assert isinstance(strings, list) and len(strings) == 1
    strs = strings[0]
    B = len(strs)
    dev = cdf.device
    C = cdf.shape[0]
    h, w = int(shape[0]), int(shape[1])
    nsym, plane = C * h * w, h * w
    sec = _Section()
    sec.mark("d.start")
    if isinstance(strs, PackedStrings) and strs.still_packed():
        total = int(strs.byte_off[-1])
        data = strs.packed_host[:total].to(dev, non_blocking=True)
        byte_off = torch.from_numpy(strs.byte_off).to(dev, non_blocking=True)
    else:
        data, byte_off = eb.pack_strings(strs, dev)
    sec.mark("d.join + H2D")
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    med = eb.medians_vec()
    sym = torch.empty((nsym, B), device=dev, dtype=torch.int32)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    st = engine.stages(net.g_s)
    cout = st[-1][0].out_channels
    up = 2 ** len(st)
    x_hat = torch.empty((B, cout, h * up, w * up), device=dev, dtype=torch.float32)
    start = torch.cuda.Event()
    start.record(main)
    side.wait_event(start)
    events = []
    for (s0, n) in _chunks(B, chunk):
        with torch.cuda.stream(side):
            ops.rans_decode_batch(data, byte_off, 1, B, nsym, plane, cdf, cdf_len, offset, sym, n, sym_offset=s0,
                                  status=status)
            ev = torch.cuda.Event()
            ev.record(side)
        events.append(ev)
    for (s0, n), ev in zip(_chunks(B, chunk), events):
        main.wait_event(ev)
        y_blk = torch.zeros((n, (C + 15) // 16, h, w, 16), device=dev, dtype=torch.float16) if C % 16 else \
            torch.empty((n, C // 16, h, w, 16), device=dev, dtype=torch.float16)
        ops.eb_dequantize(sym, 1, B, med, n, C, h, w, want_nchw=False, blk16=y_blk, sym_offset=s0)
        engine.run_chain_fp16(net.g_s, x_blk=y_blk, clamp01=True, out=x_hat[s0:s0 + n])
    sec.mark("d.decode+transforms (device)")
    if int(status.item()) != 0:  # synchronises; also keeps data/sym alive until the side stream is done
        raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
    return {"x_hat": x_hat}
