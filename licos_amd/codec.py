"""Chunk-pipelined encode / decode of a tile batch, either precision (fp16 MFMA transforms, or the fp32 parity path's
split-operand transforms (fp32 accuracy on the fp16 matrix cores): the same pipeline, with fp32 NCHW activations bounded per chunk).

The rANS recurrence is sequential inside a stream, so a coder launch is latency-bound: one lane per
tile, a few waves in total, a fixed ~N_symbols x chain-latency no matter how many tiles ride along.
The transforms are throughput-bound and fill the chip.  The two therefore overlap almost for free:
the batch is cut into chunks; while the MFMA kernels of chunk k+1 run on the main stream, the coder
kernel of chunk k runs on a side stream (encode), and symmetrically the decoder of chunk k+1 runs
under the synthesis transform of chunk k.  The host side of a chunk (stream lengths, compaction, copy
of the packed bytes into page-locked memory, building the per-tile ``bytes``) runs on a third stream
while later chunks are still being transformed.  The byte strings are CompressAI's, one per tile.
"""
import numpy as np
import torch

from . import engine, ops

import os

_EB_RECORDS = os.environ.get("LICOS_EB_RECORDS", "0") == "1"
_EB_IMAGE = os.environ.get("LICOS_EB_IMAGE", "1") != "0"  # A/B switch: "0" = the plane decoder of rans.hip
_streams = {}
CODER_STREAMS = 8  # side streams the hyperprior codec spreads its chunks' coder launches over

# Optional host-side section timing (tools/profile_step.py): when a dict, every section boundary
# synchronises the device and accumulates wall-clock seconds.  None in production.
timings = None


# Optional device timing of the serial coder launches (bench.py): when a dict, each one is bracketed by HIP events on
# the stream it runs on; key = "z_encode" | "y_encode" | "z_decode" | "y_decode" -> [(start, end)].
coder_events = None


def _timed_coder(key, fn):
    if coder_events is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    coder_events.setdefault(key, []).append((e0, e1))
    return out


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


def _stream(device, role):
    key = (device.type, device.index, role)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=device)
    return _streams[key]


class PackedStrings(list):
    """The list of per-tile byte strings ``compress`` returns, which also remembers the page-locked
    host buffers (one per pipeline chunk) the strings were cut from.  ``decompress`` uploads those
    buffers directly instead of re-joining thousands of small byte objects - but only while the list
    still holds exactly the strings it was built with: any edit (an entry replaced by different bytes,
    even of the same length, an insertion, a deletion, a re-ordering) makes ``still_packed`` false and
    ``decompress`` decodes what the list holds.  The strings themselves are ordinary ``bytes``."""

    def __init__(self, strings, segments):
        super().__init__(strings)
        self.segments = segments  # [(first tile, tile count, pinned uint8 tensor, np.int64 offsets [n+1])]
        self._built_with = tuple(self)

    def still_packed(self):
        # element-wise comparison in C: identical objects short-cut on identity (a few tens of microseconds for
        # 16384 tiles), a replaced entry is compared by content - so an equal copy is fine and anything else is not
        return (len(self) == len(self._built_with) and sum(n for _, n, _, _ in self.segments) == len(self)
                and tuple(self) == self._built_with)


def _chunks(total, size):
    return [(s, min(size, total - s)) for s in range(0, total, size)]


def compress_chunked(net, x, chunk=1024, cap_words=None):
    """FactorizedPrior.compress for any batch size and either precision (`net.g_a` dispatches on it)."""
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, table = eb.coder_tables()
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("licos_amd: compress expects a float32 (B, C, H, W) tensor")
    x = x.contiguous()
    B = x.shape[0]
    if ops.host_coder_preferred(B):
        # a handful of tiles, or one whole granule: one GPU lane per stream would take ~8 / ~16 ms per 49152 symbols
        # whatever the batch; the host cores code such a batch in a fraction of that (entropy_models._compress_host)
        y = net.g_a(x)
        return {"strings": [eb.compress(y)], "shape": y.size()[-2:]}
    dev = x.device
    main = torch.cuda.current_stream(dev)
    side = _stream(dev, "coder")
    copy = _stream(dev, "copy")
    med = eb.medians_vec()
    sec = _Section()
    sec.mark("c.start")
    sym = None
    shape = None
    # The plane encoder (rans.hip: the channel's records staged in LDS) stays: the record encoder of csrc/rans_gc.hip
    # (licos_eb_encode_prepare + licos_rans_encode_records) is 8 % faster per launch here (6.7 vs 7.3 ms) but its
    # throughput kernel writes 20 B per symbol - 1.8 ms per 4096-tile chunk on the main stream against 0.4 ms for the
    # symbols - and only the LAST launch of a call is exposed: measured, the step did not move.  LICOS_EB_RECORDS=1 switches.
    records = _EB_RECORDS and eb.coder_image() is not None
    queued = []  # every tensor another stream touches stays referenced here until its chunk is drained
    for (s0, n) in _chunks(B, chunk):
        y = net.g_a(x[s0:s0 + n])  # MFMA chain, main stream
        if shape is None:
            shape = tuple(y.shape[-2:])
            nsym, plane = y[0].numel(), y[0, 0].numel()
            if not records:
                sym = torch.empty((nsym, B), device=dev, dtype=torch.int32)
            if cap_words is None:
                cap_words = nsym // 2 + 64
        if records:
            keep = ops.eb_encode_prepare(y.contiguous(), med, table, cdf_len, offset, cdf.shape[1])
        else:
            ops.eb_quantize(y, med, "symbols", symbols=sym, sym_stride_b=1, sym_stride_i=B, sym_offset=s0)
            keep = y
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            if records:
                words, nwords, status = ops.rans_encode_records(keep[0], keep[1], cap_words)
            else:
                words, nwords, status = ops.rans_encode_batch(sym, 1, B, nsym, plane, cdf, cdf_len, offset, table, cap_words,
                                                              n, sym_offset=s0)
            coded = torch.cuda.Event()
            coded.record(side)
        queued.append((s0, n, keep, words, nwords, status, coded))
    sec.mark("c.queue transforms+encode")
    # drain chunk by chunk on the copy stream while later chunks are still in flight
    strings = [None] * B
    segments = []
    overflow = False
    for qi in range(len(queued)):
        (s0, n, keep, words, nwords, status, coded) = queued[qi]
        with torch.cuda.stream(copy):
            copy.wait_event(coded)
            meta = torch.cat((nwords, status)).cpu().numpy()  # synchronises the copy stream only
            if meta[n]:
                overflow = True
                break
            off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(meta[:n].astype(np.int64) * 4, out=off[1:])
            total = int(off[-1])
            packed = torch.empty(max(total, 4), device=dev, dtype=torch.uint8)
            ops.rans_compact(words, nwords, torch.from_numpy(off).to(dev), 0, out=packed)
            host_t = torch.empty(max(total, 4), dtype=torch.uint8, pin_memory=True)
            host_t.copy_(packed, non_blocking=True)
            copy.synchronize()
        queued[qi] = None  # the chunk's records (20 B per symbol) and word scratch go back to the allocator
        del keep, words, nwords, status
        mv = memoryview(host_t.numpy())
        strings[s0:s0 + n] = [bytes(mv[off[i]:off[i + 1]]) for i in range(n)]
        segments.append((s0, n, host_t, off))
    if overflow:
        torch.cuda.synchronize(dev)
        if cap_words >= 2 * nsym + 8:
            raise RuntimeError("licos_amd: rANS scratch overflow at worst-case capacity")
        del queued
        return compress_chunked(net, x, chunk=chunk, cap_words=2 * nsym + 8)
    main.wait_stream(side)
    main.wait_stream(copy)
    sec.mark("c.drain (lengths, compact, D2H, bytes)")
    return {"strings": [PackedStrings(strings, segments)], "shape": torch.Size(shape)}


def decompress_chunked(net, strings, shape, chunk=1024):
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, _ = eb.coder_tables()
    assert isinstance(strings, list) and len(strings) == 1
    strs = strings[0]
    B = len(strs)
    if ops.host_coder_preferred(B):
        y_hat = eb.decompress(list(strs), shape)
        x_hat = net.g_s(y_hat)
        return {"x_hat": x_hat.clamp_(0, 1)}
    dev = cdf.device
    C = cdf.shape[0]
    h, w = int(shape[0]), int(shape[1])
    nsym, plane = C * h * w, h * w
    main = torch.cuda.current_stream(dev)
    side = _stream(dev, "coder")
    med = eb.medians_vec()
    sec = _Section()
    sec.mark("d.start")
    sym = torch.empty((nsym, B), device=dev, dtype=torch.int32)
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    image = eb.coder_image() if _EB_IMAGE else None  # the image decoder (csrc/rans_gc.hip), channel pattern as shared rows
    rows = eb.channel_rows(plane) if image is not None else None
    st = engine.stages(net.g_s)
    cout = st[-1][0].out_channels
    up = 2 ** len(st)
    x_hat = torch.empty((B, cout, h * up, w * up), device=dev, dtype=torch.float32)
    # chunk list with each chunk's packed bytes: straight from compress()'s page-locked segments, or re-packed
    if isinstance(strs, PackedStrings) and strs.still_packed():
        pieces = [(s0, n, host_t, off) for (s0, n, host_t, off) in strs.segments]
    else:
        pieces = [(s0, n, None, None) for (s0, n) in _chunks(B, chunk)]
    start = torch.cuda.Event()
    start.record(main)
    side.wait_event(start)
    events = []
    keep = []
    for (s0, n, host_t, off) in pieces:
        with torch.cuda.stream(side):
            if host_t is not None:
                data = host_t[: max(int(off[-1]), 4)].to(dev, non_blocking=True)
                byte_off = torch.from_numpy(off).to(dev, non_blocking=True)
            else:  # a staging buffer per piece, no sync here: the call's final status read orders everything
                data, byte_off = eb.pack_strings(strs[s0:s0 + n], dev, slot=len(keep))
            if image is not None:
                ops.rans_decode_image(data, byte_off, rows, nsym, image[0], image[1], sym, 1, B, n, status=status, sym_offset=s0,
                                      rows_shared=True)
            else:
                ops.rans_decode_batch(data, byte_off, 1, B, nsym, plane, cdf, cdf_len, offset, sym, n, sym_offset=s0,
                                      status=status, off_offset=0)
            ev = torch.cuda.Event()
            ev.record(side)
        keep.append((data, byte_off))
        events.append(ev)
    sec.mark("d.queue H2D+decode")
    fp16 = net.precision == "fp16"
    for (s0, n, _, _), ev in zip(pieces, events):
        main.wait_event(ev)
        if fp16:
            y_blk = torch.zeros((n, (C + 15) // 16, h, w, 16), device=dev, dtype=torch.float16) if C % 16 else \
                torch.empty((n, C // 16, h, w, 16), device=dev, dtype=torch.float16)
            ops.eb_dequantize(sym, 1, B, med, n, C, h, w, want_nchw=False, blk16=y_blk, sym_offset=s0)
            engine.run_chain_fp16(net.g_s, x_blk=y_blk, clamp01=True, out=x_hat[s0:s0 + n])
        else:  # the parity path: fp32 NCHW latents, y_hat = symbol + median exactly as the reference's decompress
            y_hat = ops.eb_dequantize(sym, 1, B, med, n, C, h, w, sym_offset=s0)
            x_hat[s0:s0 + n] = net.g_s(y_hat).detach().clamp_(0, 1)
    sec.mark("d.decode+transforms (device)")
    if int(status.item()) != 0:  # synchronises; also keeps data/sym alive until the side stream is done
        raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
    return {"x_hat": x_hat}


compress_fp16, decompress_fp16 = compress_chunked, decompress_chunked  # (names of the first two rounds)


# ------------------------------------------------------------------------------------------------ scale hyperprior
def _drain(dev, copy, coded, parts):
    """One pipeline chunk's strings: `parts` = [(words, nwords, status)] per string list (all of the same n streams).
    One D2H of the lengths, compaction of every list into ONE packed buffer, one D2H of that.  Returns
    (overflow flag, page-locked tensor, [np.int64 offsets [n+1]] per list) with the offsets absolute in the buffer."""
    with torch.cuda.stream(copy):
        copy.wait_event(coded)
        n = parts[0][1].numel()
        meta = torch.cat([t for (_, nw, st) in parts for t in (nw, st)]).cpu().numpy()  # synchronises the copy stream only
        offs, base = [], 0
        for k in range(len(parts)):
            seg = meta[k * (n + 1):(k + 1) * (n + 1)]
            if seg[n]:
                return True, None, None
            off = np.empty(n + 1, dtype=np.int64)
            off[0] = base
            np.cumsum(seg[:n].astype(np.int64) * 4, out=off[1:])
            off[1:] += base
            base = int(off[-1])
            offs.append(off)
        packed = torch.empty(max(base, 4), device=dev, dtype=torch.uint8)
        for (words, nwords, _), off in zip(parts, offs):
            ops.rans_compact(words, nwords, torch.from_numpy(off).to(dev), 0, out=packed)
        host_t = torch.empty(max(base, 4), dtype=torch.uint8, pin_memory=True)
        host_t.copy_(packed, non_blocking=True)
        copy.synchronize()
    return False, host_t, offs


def hyper_fast_path(net, batch):
    """The chunk-pipelined scale-hyperprior codec applies: device coder placement and a decoder image that fits LDS."""
    return (not ops.host_coder_preferred(batch)) and net.gaussian_conditional.coder_image() is not None


def hyper_retry_chunk(chunk, ny):
    """Chunk size of the worst-case-capacity retry (cap_words = 2 ny + 8): licos_rans_encode_records addresses its word
    sink with 32-bit byte offsets, (cap + 1) * streams * 4 < 2^32 - an incompressible batch of M = 320 latents at 512^2
    would not fit at the default 2048 tiles per chunk."""
    cap = 2 * ny + 8
    return max(1, min(chunk, ((1 << 32) - 1) // (4 * (cap + 1))))


def compress_hyper(net, x, chunk=512, cap_words=None):
    """ScaleHyperprior.compress ([CAI] models/google.py) for a large batch, either precision: per chunk the four
    transforms run on the main stream, then ONE throughput kernel turns (y, scales) into per-symbol encoder records
    (licos_gc_encode_prepare) and the two serial coder kernels (z: plane coder, y: record coder) run on the side stream
    under the next chunk's transforms.  z_hat is round(z - median) + median computed directly: the reference obtains it
    by decoding the z string it has just written, which returns exactly those integers."""
    eb, gc = net.entropy_bottleneck, net.gaussian_conditional
    zcdf, zlen, zoff, ztab = eb.coder_tables()
    ycdf, ylen, yoff, ytab = gc.coder_tables()
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("licos_amd: compress expects a float32 (B, C, H, W) tensor")
    x = x.contiguous()
    B = x.shape[0]
    dev = x.device
    main = torch.cuda.current_stream(dev)
    copy = _stream(dev, "copy")
    med = eb.medians_vec()
    bound = gc.lower_bound_scale.bound_value
    queued, shape = [], None
    for ci, (s0, n) in enumerate(_chunks(B, chunk)):
        # a coder launch is a latency chain on a handful of CUs: the chunks' launches run side by side, each on a
        # stream of its own (on ONE stream they would queue up behind each other, ~50 ms apiece)
        side = _stream(dev, "coder%d" % (ci % CODER_STREAMS))
        y = net.g_a(x[s0:s0 + n])
        z = net.h_a(y)
        if shape is None:
            shape = tuple(z.shape[-2:])
            ny, nz, zplane = y[0].numel(), z[0].numel(), z[0, 0].numel()
            ycap = (ny // 2 + 64) if cap_words is None else cap_words
            zcap = nz // 2 + 64 if cap_words is None else 2 * nz + 8
        zsym = torch.empty((nz, n), device=dev, dtype=torch.int32)
        ops.eb_quantize(z, med, "symbols", symbols=zsym, sym_stride_b=1, sym_stride_i=n)
        z_hat = ops.eb_quantize(z, med, "dequantize")
        scales = net.h_s(z_hat)
        rec, aux = ops.gc_encode_prepare(y.contiguous(), scales.contiguous(), gc.scale_table, bound, ytab, ylen, yoff, ycdf.shape[1])
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            zpart = _timed_coder("z_encode", lambda: ops.rans_encode_batch(zsym, 1, n, nz, zplane, zcdf, zlen, zoff, ztab, zcap, n))
            ypart = _timed_coder("y_encode", lambda: ops.rans_encode_records(rec, aux, ycap))
            coded = torch.cuda.Event()
            coded.record(side)
        queued.append((s0, n, (y, z, zsym, rec, aux), ypart, zpart, coded))
        del y, z, zsym, z_hat, scales, rec, aux
    ys, zs = [None] * B, [None] * B
    segments = []
    overflow = False
    for i, (s0, n, keep, ypart, zpart, coded) in enumerate(queued):
        overflow, host_t, offs = _drain(dev, copy, coded, [ypart, zpart])
        if overflow:
            break
        queued[i] = None  # the chunk's records (20 B per symbol) and scratch go back to the allocator
        del keep, ypart, zpart
        mv = memoryview(host_t.numpy())
        yo, zo = offs
        ys[s0:s0 + n] = [bytes(mv[yo[i]:yo[i + 1]]) for i in range(n)]
        zs[s0:s0 + n] = [bytes(mv[zo[i]:zo[i + 1]]) for i in range(n)]
        segments.append((s0, n, host_t, yo, zo))
    if overflow:
        torch.cuda.synchronize(dev)
        if cap_words is not None:
            raise RuntimeError("licos_amd: rANS scratch overflow at worst-case capacity")
        del queued
        return compress_hyper(net, x, chunk=hyper_retry_chunk(chunk, ny), cap_words=2 * ny + 8)
    for ci in range(min(CODER_STREAMS, len(segments))):
        main.wait_stream(_stream(dev, "coder%d" % ci))
    main.wait_stream(copy)
    ysegs = [(s0, n, t, yo) for (s0, n, t, yo, _) in segments]
    zsegs = [(s0, n, t, zo) for (s0, n, t, _, zo) in segments]
    return {"strings": [PackedStrings(ys, ysegs), PackedStrings(zs, zsegs)], "shape": torch.Size(shape)}


def _upload(strs, pieces, dev, id_base=0):
    """Per piece (s0, n): (device bytes, device int64 offsets [n+1]) of strs[s0:s0+n] - straight from compress()'s
    page-locked segments when `strs` still is what compress() returned, else re-joined through a staging buffer."""
    from .entropy_models import EntropyBottleneck
    if isinstance(strs, PackedStrings) and strs.still_packed() and [(s, n) for (s, n, _, _) in strs.segments] == list(pieces):
        out = []
        for (_, n, host_t, off) in strs.segments:
            lo, hi = int(off[0]), int(off[-1])
            lo4 = lo & ~3
            data = host_t[lo4: max(hi, lo4 + 4)].to(dev, non_blocking=True)
            out.append((data, torch.from_numpy(off - lo4).to(dev, non_blocking=True)))
        return out
    return [EntropyBottleneck.pack_strings(strs[s0:s0 + n], dev, slot=(id_base + k)) for k, (s0, n) in enumerate(pieces)]


def decompress_hyper(net, strings, shape, chunk=512):
    """ScaleHyperprior.decompress for a large batch: every tile's z string is decoded in one launch, then per chunk
    h_s + the row-byte kernel run on the main stream and the chunk's y decoder on the side stream - ALL chunks' decoders
    are in flight before the first synthesis transform starts, which then overlaps the later chunks' decoding."""
    from . import engine
    eb, gc = net.entropy_bottleneck, net.gaussian_conditional
    zcdf, zlen, zoff, _ = eb.coder_tables()
    gc.note_row_usage()  # (nothing is in flight here) rows seen by earlier calls steer the image's record budget
    image_dev, image_host = gc.coder_image()
    row_hist = gc.row_histogram()
    assert isinstance(strings, list) and len(strings) == 2
    ystrs, zstrs = strings
    B = len(ystrs)
    if len(zstrs) != B:
        raise ValueError("licos_amd: y and z string lists differ in length")
    dev = zcdf.device
    h, w = int(shape[0]), int(shape[1])
    N, M = net.N, net.M
    nz, zplane = N * h * w, h * w
    ny = M * (4 * h) * (4 * w)
    main = torch.cuda.current_stream(dev)
    med = eb.medians_vec()
    bound = gc.lower_bound_scale.bound_value
    pieces = [(s0, n) for (s0, n) in _chunks(B, chunk)]
    # PackedStrings carry compress()'s own chunking; decode in those pieces when it is intact
    if isinstance(ystrs, PackedStrings) and ystrs.still_packed():
        pieces = [(s0, n) for (s0, n, _, _) in ystrs.segments]
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    zsym = torch.empty((nz, B), device=dev, dtype=torch.int32)
    # z: every tile's string in ONE launch (2 ms whatever the batch; a launch per piece would queue them up on this stream)
    zup = _upload(zstrs, pieces, dev)
    if len(zup) > 1:  # (every string is a whole number of 32-bit words: the pieces concatenate without padding)
        zdata = torch.cat([data for (data, _) in zup])
        base, offs = 0, []
        for (data, off) in zup:
            offs.append(off[:-1] + base)
            base += data.numel()
        zoff_all = torch.cat(offs + [torch.tensor([base], device=dev, dtype=torch.int64)])
    else:
        zdata, zoff_all = zup[0]
    _timed_coder("z_decode", lambda: ops.rans_decode_batch(zdata, zoff_all, 1, B, nz, zplane, zcdf, zlen, zoff, zsym, B,
                                                           status=status, off_offset=0))
    z_hat = ops.eb_dequantize(zsym, 1, B, med, B, N, h, w)
    yup = _upload(ystrs, pieces, dev, id_base=len(pieces))  # staging slots behind the z pieces': no slot is shared in a call
    fp16 = net.precision == "fp16"
    st = engine.stages(net.g_s)
    cout = st[-1][0].out_channels
    x_hat = torch.empty((B, cout, 64 * h, 64 * w), device=dev, dtype=torch.float32)
    zeros = torch.zeros(M, device=dev, dtype=torch.float32)
    events, keep = [], []
    for ci, ((s0, n), (data, off)) in enumerate(zip(pieces, yup)):
        side = _stream(dev, "coder%d" % (ci % CODER_STREAMS))
        scales = net.h_s(z_hat[s0:s0 + n])
        idx16 = ops.gc_decode_prepare(scales.contiguous(), gc.scale_table, bound, row_hist=row_hist)
        sym = torch.empty((ny, n), device=dev, dtype=torch.int32)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            _timed_coder("y_decode", lambda: ops.rans_decode_image(data, off, idx16, ny, image_dev, image_host, sym, 1, n, n, status=status))
            ev = torch.cuda.Event()
            ev.record(side)
        events.append(ev)
        keep.append((data, off, idx16, sym))
        del scales
    for (s0, n), ev, (_, _, _, sym) in zip(pieces, events, keep):
        main.wait_event(ev)
        if fp16:
            y_blk = torch.empty((n, M // 16, 4 * h, 4 * w, 16), device=dev, dtype=torch.float16) if M % 16 == 0 else \
                torch.zeros((n, (M + 15) // 16, 4 * h, 4 * w, 16), device=dev, dtype=torch.float16)
            ops.eb_dequantize(sym, 1, n, zeros, n, M, 4 * h, 4 * w, want_nchw=False, blk16=y_blk)
            engine.run_chain_fp16(net.g_s, x_blk=y_blk, clamp01=True, out=x_hat[s0:s0 + n])
        else:
            y_hat = ops.eb_dequantize(sym, 1, n, zeros, n, M, 4 * h, 4 * w)
            x_hat[s0:s0 + n] = net.g_s(y_hat).detach().clamp_(0, 1)
    if int(status.item()) != 0:  # synchronises; also keeps the side stream's tensors alive until it is done
        raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
    return {"x_hat": x_hat}
