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
import time

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
host_trace = None  # dev probe: when a list, the host loops append ("enc", tiles, wait ms, coder ms, strings ms) / ("dec", tiles, join ms, coder ms, queue ms)


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
        # (the segments cover the device-coded tiles: a prefix of the list when the host coded the call's tail)
        return (len(self) == len(self._built_with) and sum(n for _, n, _, _ in self.segments) <= len(self)
                and tuple(self) == self._built_with)


def _split_bytes(mv, off):
    """The byte strings mv[off[i]:off[i+1]] (offsets as Python ints: indexing a memoryview with numpy scalars costs 2.5 x
    the time - 4096 strings are 1 - 2 ms of this thread, and the last chunk's are at the exposed end of compress)."""
    o = off.tolist()
    return [bytes(mv[a:b]) for a, b in zip(o[:-1], o[1:])]


def _chunks(total, size):
    return [(s, min(size, total - s)) for s in range(0, total, size)]


def _ramp(total, first, size):
    """Sub-chunks that double from `first` up to `size`: the host starts coding (or the synthesis transform starts) after
    a few tiles' worth of work instead of a full sub-chunk's, and the later sub-chunks keep the transfers large."""
    out, s0, m = [], 0, max(1, min(first, size))
    while s0 < total:
        n = min(m, total - s0)
        out.append((s0, n))
        s0 += n
        m = min(size, 2 * m)
    return out


# ---- split placement of the serial coder --------------------------------------------------------------------------------
# A device coder launch lasts n_symbols x (latency of one rANS step) whatever the number of streams: ~145 ns per symbol
# encoding, ~115 ns decoding (7.1 / 5.6 ms for a 256^2 tile's 49 152 symbols).  Inside a long call that is hidden under
# the neighbouring chunks' transforms except ONCE per call - the last chunk's encode, the first chunk's decode - and it is
# most of a call of a thousand tiles.  The host cores run the same coder (csrc/host_rans.cpp, bit-identical streams) at
# ~1.8 / ~4 ns per symbol and thread: in the time of one device launch T threads code  T x (device ns) / (host ns)  tiles
# (16 threads: ~1100 encoding, ~390 decoding: the decode figure is what the sub-chunk pipeline sustains - 256 tiles in 2.35 ms,
# 32 in 0.5 ms - not the coder alone).  So the tiles at the exposed end of a call go to the host, in sub-chunks
# that pipeline with their transforms, and the device launch they run beside covers the rest.  LICOS_HOST_SPLIT=0
# switches the split off (A/B); LICOS_HOST_ENC_NS / LICOS_HOST_DEC_NS override the host figures.
HOST_SPLIT = os.environ.get("LICOS_HOST_SPLIT", "1") != "0"
DEV_NS = {"enc": 145.0, "dec": 115.0}
HOST_NS = {"enc": float(os.environ.get("LICOS_HOST_ENC_NS", "1.8")), "dec": float(os.environ.get("LICOS_HOST_DEC_NS", "4.0"))}
# tiles per host thread and sub-chunk: 16 threads x 16 = 256 tiles = 50 MB of int32 symbols per transfer (a 25 MB
# device-to-host copy runs at 15 GB/s on these boxes, a 150 MB one at 53: tools/split_probe.py)
HOST_SUB = 16
SIMPLE_BATCH = 8  # calls of up to this many tiles skip the sub-chunk pipeline (one transform, one copy, one coder call)
# Symbols of the host's tiles cross PCIe through a device buffer and the copy engines (default), or by the quantise /
# dequantise kernels' own stores and loads on the page-locked staging buffer (LICOS_ZERO_COPY=1; measured equal for small
# batches and slower at 1024 tiles: 13.2 vs 10.9 ms)
ZERO_COPY = os.environ.get("LICOS_ZERO_COPY", "0") == "1"
# The host's tiles cross PCIe as 16-bit symbols (licos_eb_symbols16 / licos_eb_dequantize16, licos_rans_*_host_sym16): the
# copies of a sub-chunk are blit kernels that share the CUs with the next sub-chunk's transforms, half the bytes are half
# of that.  A symbol outside 16 bits is flagged by the quantise kernel / the host decoder and the call (the sub-chunk)
# falls back to the 32-bit form.  LICOS_SYM16=0 switches it off (A/B).
SYM16 = os.environ.get("LICOS_SYM16", "1") != "0"
PREQUEUE = int(os.environ.get("LICOS_PREQUEUE", "3"))  # host sub-chunks of a large compress queued before the drains (0: none; A/B)


def _pinned_i16(role, rows, cols):
    """A reusable page-locked int16 [rows, cols] staging area (contiguous whatever the parity of cols)."""
    return _pinned_i32(role, 1, (rows * cols + 1) // 2).view(torch.int16)[0, :rows * cols].view(rows, cols)


# Feedback: what the host coder actually delivered in this process's recent calls, as a factor on HOST_NS (1 = nominal).
# The host is shared on these boxes: a host whose cores are busy with other tenants' work codes 5 - 8 x slower, and a
# share sized for a quiet host would then be the slowest part of the call.  Every sub-chunk of >= 4 tiles per thread
# updates the factor (exponential average); it never goes below 1.
_host_factor = {"enc": 1.0, "dec": 1.0}
_EXPECT_NS = {"enc": 1.8, "dec": 3.0}  # coder time alone per symbol and thread, as the pipeline's sub-chunks see it on a quiet host


def _note_host_rate(direction, tiles, nsym, seconds, expect_ns=None):
    threads = ops.host_threads()
    if tiles < 4 * threads or seconds <= 0:
        return
    ns = 1e9 * seconds * threads / (tiles * nsym)
    f = max(1.0, ns / (_EXPECT_NS[direction] if expect_ns is None else expect_ns))
    _host_factor[direction] = 0.5 * _host_factor[direction] + 0.5 * f


_cap_state = {}
# compress calls of up to this many capacities stay host-only: the host pipeline (transforms, 16-bit symbols over PCIe, coder)
# moves a tile in 5.5 - 7 us at 16 threads, about what the transforms alone take, while a device chunk adds its launch's
# latency and its drain - A/B on one box, ms per call, host-only against the split: 2048 tiles 14.3 - 15.2 vs 16.5, 3000
# tiles 19.9 - 21.1 vs 22.0, 4096 tiles equal; with 12 / 8 threads (capacity 800 / 544) the same up to 1.75 capacities
# (profiles/r04_enc_all_host.log)
ENC_ALL_HOST = float(os.environ.get("LICOS_ENC_ALL_HOST", "2.5"))  # (3.0 measured equal-or-better at 16 threads; 2.5 keeps a margin for hosts with fewer threads, measured to 1.75)
ENC_TAIL = float(os.environ.get("LICOS_ENC_TAIL_FRACTION", "1.0"))  # share of the capacity at the end of a larger compress call


def host_capacity(direction):
    """Tiles the host cores code in the time of ONE device coder launch (independent of the stream length), on a grid of
    2 tiles per thread and with one grid step of hysteresis: the measured host rate moves a little with every call, and
    a share that moved with it would give every call its own tensor shapes (3 751, 3 775, 3 747 ... tiles in a piece) -
    work for the caching allocator, and now and then a hipMalloc in the middle of a step."""
    threads = ops.host_threads()
    raw = 0.85 * threads * DEV_NS[direction] / (HOST_NS[direction] * _host_factor[direction])
    grid = 2 * threads
    key = (direction, threads)
    last = _cap_state.get(key)
    if last is None or abs(raw - (last + 0.5 * grid)) >= grid:
        last = _cap_state[key] = max(0, int(raw) // grid * grid if raw >= grid else int(raw))
    return last


def host_share(batch, direction):
    """How many tiles of a call of `batch` tiles the host codes.  Everything up to the host's capacity for the direction
    (a thousand tiles encoding, four hundred decoding at 16 threads: the device coder's launch latency alone is longer
    than the host takes).  Of a larger call, the tiles at its exposed end: the first `capacity` tiles of a decode - the
    synthesis transform starts on them while the first device launch runs - and the last `capacity` tiles of an encode
    (what the host pipeline finishes beside the last 7-ms device launch and that launch's drain; ENC_TAIL scales it).  `capacity` follows the rate the host coder delivered in this process's recent calls (_note_host_rate).
    The large-call split was measured and left out twice earlier in round 4 (+10 ms per 16 384-tile step); the loss was
    not the split: a piece of decompress that joins a packed segment with the tiles the host encoded built its end offset
    with `torch.tensor([n], device=...)`, a blocking copy that waited for every decode launch queued so far (fixed:
    decompress_chunked).  With that gone: 16 384 tiles 201.3 -> 195.6 ms on one box, 204.5 -> 200.7 on another; 4 096
    tiles 63.3 -> 58.7 ms (tools/batch_probe.py, profiles/r04_large_call_split.log)."""
    if ops.HOST_CODER == "0":
        return 0
    if ops.HOST_CODER == "1" or ops.host_coder_preferred(batch):
        return batch
    if not HOST_SPLIT:
        return 0
    cap = host_capacity(direction)
    if batch <= cap or (direction == "enc" and batch <= ENC_ALL_HOST * cap):
        return batch  # (a compress call a little over the capacity: the host finishing late costs less than a device chunk's drain)
    return cap if direction == "dec" else int(ENC_TAIL * cap)


_pinned = {}


def _pinned_i32(role, rows, cols):
    """A reusable page-locked int32 [rows, cols] staging area per role (page-locked allocations cost milliseconds)."""
    need = rows * cols
    buf = _pinned.get(role)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1 << 18) * 5 // 4, dtype=torch.int32, pin_memory=True)
        _pinned[role] = buf
    return buf[:need].view(rows, cols)


def compress_chunked(net, x, chunk=1024, cap_words=None, sym16=None):
    """FactorizedPrior.compress for any batch size and either precision (`net.g_a` dispatches on it).  Tiles
    [0, B - H) go through the device coder in pipeline chunks, the last H = host_share(B) tiles through the host coder in
    sub-chunks (see "split placement" above); the strings do not depend on the placement."""
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, table = eb.coder_tables()
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("licos_amd: compress expects a float32 (B, C, H, W) tensor")
    x = x.contiguous()
    B = x.shape[0]
    n_host = host_share(B, "enc")
    if n_host == B and B <= SIMPLE_BATCH and ops.host_coder_preferred(B):
        # a handful of tiles: nothing to pipeline - transform, one copy, one host coder call (0.07 ms less per call than
        # the sub-chunk machinery below)
        y = net.g_a(x)
        return {"strings": [eb.compress(y)], "shape": y.size()[-2:]}
    n_dev = B - n_host
    dev = x.device
    main = torch.cuda.current_stream(dev)
    side = _stream(dev, "coder")
    copy = _stream(dev, "copy")
    hcopy = _stream(dev, "hostsym")
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
    for (s0, n) in _chunks(n_dev, chunk):
        y = net.g_a(x[s0:s0 + n])  # MFMA chain, main stream
        if shape is None:
            shape = tuple(y.shape[-2:])
            nsym, plane = y[0].numel(), y[0, 0].numel()
            if not records:
                sym = torch.empty((nsym, n_dev), device=dev, dtype=torch.int32)
            if cap_words is None:
                cap_words = nsym // 2 + 64
        if records:
            keep = ops.eb_encode_prepare(y.contiguous(), med, table, cdf_len, offset, cdf.shape[1])
        else:
            ops.eb_quantize(y, med, "symbols", symbols=sym, sym_stride_b=1, sym_stride_i=n_dev, sym_offset=s0)
            keep = y
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            if records:
                words, nwords, status = ops.rans_encode_records(keep[0], keep[1], cap_words)
            else:
                words, nwords, status = ops.rans_encode_batch(sym, 1, n_dev, nsym, plane, cdf, cdf_len, offset, table, cap_words,
                                                              n, sym_offset=s0)
            coded = torch.cuda.Event()
            coded.record(side)
        queued.append((s0, n, keep, words, nwords, status, coded))
    sec.mark("c.queue transforms+encode")
    strings = [None] * B
    segments = []
    overflow = False

    def drain(qi):
        """One device chunk's strings: lengths, compaction, D2H, bytes - on the copy stream, while later work is in flight."""
        (s0, n, keep, words, nwords, status, coded) = queued[qi]
        with torch.cuda.stream(copy):
            copy.wait_event(coded)
            meta = torch.cat((nwords, status)).cpu().numpy()  # synchronises the copy stream only
            if meta[n]:
                return True
            off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(meta[:n].astype(np.int64) * 4, out=off[1:])
            total = int(off[-1])
            packed = torch.empty(max(total, 4), device=dev, dtype=torch.uint8)
            ops.rans_compact(words, nwords, torch.from_numpy(off).to(dev), 0, out=packed)
            host_t = torch.empty(max(total, 4), dtype=torch.uint8, pin_memory=True)
            host_t.copy_(packed, non_blocking=True)
            copy.synchronize()
        queued[qi] = None  # the chunk's records (20 B per symbol) and word scratch go back to the allocator
        mv = memoryview(host_t.numpy())
        strings[s0:s0 + n] = _split_bytes(mv, off)
        segments.append((s0, n, host_t, off))
        return False

    # The host's first sub-chunks are queued BEFORE the device chunks are drained: a drain is milliseconds of this thread
    # (lengths, compaction, D2H, 4096 bytes objects), and with nothing queued behind the last device chunk's transforms the
    # GPU idled ~2 ms at the very place the call is exposed (tools/tail_probe.py).
    subs = list(_ramp(n_host, 2 * ops.host_threads(), max(1, HOST_SUB * ops.host_threads()))) if n_host else []
    host_state = {"stage": None}
    use16 = (SYM16 if sym16 is None else sym16) and not ZERO_COPY
    hflag = torch.zeros(max(1, len(subs)), device=dev, dtype=torch.int32) if use16 else None
    st_f = _pinned_i32("ef", 1, max(64, len(subs)))[0] if use16 else None

    def queue_sub(k):
        """Sub-chunk k of the host's tiles: transforms + quantise on the main stream, symbols to the page-locked buffer."""
        nonlocal shape, nsym, plane
        (t0, m) = subs[k]
        y = net.g_a(x[n_dev + t0:n_dev + t0 + m])
        if shape is None:
            shape = tuple(y.shape[-2:])
            nsym, plane = y[0].numel(), y[0, 0].numel()
        if host_state["stage"] is None:
            host_state["stage"] = _pinned_i16("enc16", n_host, nsym) if use16 else _pinned_i32("enc", n_host, nsym)
        stage = host_state["stage"]
        if ZERO_COPY:
            ops.eb_quantize(y.contiguous(), med, "symbols", symbols=stage[t0:t0 + m], sym_stride_b=nsym, sym_stride_i=1)
            landed = torch.cuda.Event()
            landed.record(main)
            return (k, t0, m, y, landed)
        if use16:
            hsym = torch.empty((m, nsym), device=dev, dtype=torch.int16)
            ops.eb_symbols16(y.contiguous(), med, hsym, hflag[k:k + 1])
        else:
            hsym = torch.empty((m, nsym), device=dev, dtype=torch.int32)
            ops.eb_quantize(y.contiguous(), med, "symbols", symbols=hsym, sym_stride_b=nsym, sym_stride_i=1)
        ready = torch.cuda.Event()
        ready.record(main)
        # (a stream of its own: on the drains' copy stream these copies would queue up behind / in front of the device
        # chunks' length and byte transfers)
        with torch.cuda.stream(hcopy):
            hcopy.wait_event(ready)
            stage[t0:t0 + m].copy_(hsym, non_blocking=True)
            if use16:
                st_f[k:k + 1].copy_(hflag[k:k + 1], non_blocking=True)
            landed = torch.cuda.Event()
            landed.record(hcopy)
        return (k, t0, m, hsym, landed)

    prequeued = [queue_sub(k) for k in range(min(PREQUEUE, len(subs)))] if len(queued) > 1 else []
    # every device chunk but the last (the last device launch runs beside the host's share below)
    for qi in range(len(queued) - 1):
        if drain(qi):
            overflow = True
            break
    if host_trace is not None:
        host_trace.append(("enc-drained", len(queued) - 1, time.perf_counter()))
    # The host's tiles, a software pipeline in this thread: queue sub-chunk k's transforms + quantise (main stream) and
    # the copy of its symbols [stream][position] to a page-locked buffer (a stream of its own), THEN code sub-chunk k - 1
    # while the GPU works on k.  (Queueing everything first and coding afterwards cost a 1024-tile call 4.5 ms: a
    # sub-chunk's ten launches are ~0.6 ms of Python, during which the host coder had nothing to do.)
    if n_host and not overflow:
        hcdf, hlen, hoff, htable = eb.coder_tables_host()

        def host_encode(entry):
            (k, t0, m, _keep, landed) = entry
            w0 = time.perf_counter()
            landed.synchronize()
            w1 = time.perf_counter()
            stage = host_state["stage"]
            if use16:
                if int(st_f[k]) != 0:
                    raise _HostRange()
                out, nbytes = ops.rans_encode_host_sym16(stage[t0:t0 + m].numpy(), nsym, plane, hcdf, hlen, hoff, htable)
            else:
                out, nbytes = ops.rans_encode_host(stage[t0:t0 + m].numpy(), nsym, plane, hcdf, hlen, hoff, htable)
            w2 = time.perf_counter()
            _note_host_rate("enc", m, nsym, w2 - w1)
            strings[n_dev + t0:n_dev + t0 + m] = [out[i, : int(nbytes[i])].tobytes() for i in range(m)]
            if host_trace is not None:
                host_trace.append(("enc", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (time.perf_counter() - w2), 3), w0))

        try:
            # a software pipeline in this thread: with sub-chunks 0 .. p - 1 queued, queue sub-chunk k + p, then code k
            ahead = max(1, len(prequeued))
            entries = list(prequeued)
            for k in range(len(subs)):
                while len(entries) < min(len(subs), k + ahead + 1):
                    entries.append(queue_sub(len(entries)))
                host_encode(entries[k])
                entries[k] = None
        except _HostRange:  # a symbol outside 16 bits: the whole call again with 32-bit symbols for the host's tiles
            torch.cuda.synchronize(dev)
            del queued
            return compress_chunked(net, x, chunk=chunk, cap_words=cap_words, sym16=False)
        except BaseException:
            torch.cuda.synchronize(dev)  # later sub-chunks' copies still target the shared page-locked buffer: let them land
            raise
    if host_trace is not None:
        host_trace.append(("enc-host-done", n_host, time.perf_counter()))
    if queued and not overflow:
        overflow = drain(len(queued) - 1)
    if host_trace is not None:
        host_trace.append(("enc-last-drained", 0, time.perf_counter()))
    if overflow:
        torch.cuda.synchronize(dev)
        if cap_words >= 2 * nsym + 8:
            raise RuntimeError("licos_amd: rANS scratch overflow at worst-case capacity")
        del queued
        return compress_chunked(net, x, chunk=chunk, cap_words=2 * nsym + 8, sym16=sym16)
    main.wait_stream(side)
    main.wait_stream(copy)
    main.wait_stream(hcopy)
    sec.mark("c.drain (lengths, compact, D2H, bytes)")
    return {"strings": [PackedStrings(strings, segments)], "shape": torch.Size(shape)}


def decompress_chunked(net, strings, shape, chunk=1024):
    """FactorizedPrior.decompress.  The first H = host_share(B) tiles are decoded by the host cores in sub-chunks - the
    synthesis transform starts on them about a millisecond into the call - while the device decodes the others (every
    device launch is queued before the host starts); see "split placement" above."""
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, _ = eb.coder_tables()
    assert isinstance(strings, list) and len(strings) == 1
    strs = strings[0]
    B = len(strs)
    n_host = host_share(B, "dec")
    if n_host == B and B <= SIMPLE_BATCH and ops.host_coder_preferred(B):
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
    n_dev = B - n_host
    sym = torch.empty((nsym, n_dev), device=dev, dtype=torch.int32) if n_dev else None  # device-decoded tiles, [position][stream]
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    image = eb.coder_image() if _EB_IMAGE else None  # the image decoder (csrc/rans_gc.hip), channel pattern as shared rows
    rows = eb.channel_rows(plane) if image is not None else None
    st = engine.stages(net.g_s)
    cout = st[-1][0].out_channels
    up = 2 ** len(st)
    x_hat = torch.empty((B, cout, h * up, w * up), device=dev, dtype=torch.float32)
    # device pieces (first tile, count, packed bytes, offsets) over tiles [n_host, B): straight from compress()'s
    # page-locked segments where they cover them, re-packed otherwise
    pieces = []
    if n_dev:
        covered = n_host
        if isinstance(strs, PackedStrings) and strs.still_packed():
            for (s0, n, host_t, off) in strs.segments:
                if s0 + n <= covered or s0 > covered:
                    continue
                lo = covered - s0
                pieces.append((covered, n - lo, host_t, off[lo:]))
                covered = s0 + n
        pieces += [(covered + t0, m, None, None) for (t0, m) in _chunks(B - covered, chunk)]
        # a device launch lasts as long for 100 streams as for 4096: neighbours that fit one chunk together (the tiles the
        # host encoded, behind the last packed segment) share a launch - their bytes are uploaded one after the other
        merged = []
        for pc in pieces:
            if merged and merged[-1][1] + pc[1] <= chunk:
                merged[-1] = (merged[-1][0], merged[-1][1] + pc[1], merged[-1][2] + [pc])
            else:
                merged.append((pc[0], pc[1], [pc]))
        pieces = merged
    start = torch.cuda.Event()
    start.record(main)
    side.wait_event(start)
    events = []
    keep = []
    nslot = 0
    for (s0, n, parts) in pieces:
        with torch.cuda.stream(side):
            ups = []
            for (p0, pn, host_t, off) in parts:
                if host_t is not None:
                    lo, hi = int(off[0]), int(off[-1])
                    ups.append((host_t[lo: max(hi, lo + 4)].to(dev, non_blocking=True), torch.from_numpy(off - lo).to(dev, non_blocking=True)))
                else:  # a staging buffer per part, no sync here: the call's final status read orders everything
                    ups.append(eb.pack_strings(strs[p0:p0 + pn], dev, slot=nslot))
                    nslot += 1
            if len(ups) == 1:
                data, byte_off = ups[0]
            else:  # (every string is a whole number of 32-bit words: the parts concatenate without padding)
                # (no scalar tensor for the end offset: building one is a blocking copy that waits for every decode launch
                # queued on this stream so far - 10 ms of a 16 384-tile call whose last piece is such a join)
                base, offs = 0, []
                for j, (d_, o_) in enumerate(ups):
                    offs.append((o_ if j == len(ups) - 1 else o_[:-1]) + base)
                    base += int(d_.numel())
                data = torch.cat([d_ for (d_, _) in ups])
                byte_off = torch.cat(offs)
            if image is not None:
                ops.rans_decode_image(data, byte_off, rows, nsym, image[0], image[1], sym, 1, n_dev, n, status=status,
                                      sym_offset=s0 - n_host, rows_shared=True)
            else:
                ops.rans_decode_batch(data, byte_off, 1, n_dev, nsym, plane, cdf, cdf_len, offset, sym, n, sym_offset=s0 - n_host,
                                      status=status, off_offset=0)
            ev = torch.cuda.Event()
            ev.record(side)
        keep.append((data, byte_off))
        events.append(ev)
    sec.mark("d.queue H2D+decode")
    if host_trace is not None:
        host_trace.append(("dec-launches-queued", len(pieces), time.perf_counter()))
    fp16 = net.precision == "fp16"

    def synthesise(s0, n, symbols, stride_b, stride_i, sym_offset=0):
        if fp16:
            y_blk = torch.zeros((n, (C + 15) // 16, h, w, 16), device=dev, dtype=torch.float16) if C % 16 else \
                torch.empty((n, C // 16, h, w, 16), device=dev, dtype=torch.float16)
            ops.eb_dequantize(symbols, stride_b, stride_i, med, n, C, h, w, want_nchw=False, blk16=y_blk, sym_offset=sym_offset)
            engine.run_chain_fp16(net.g_s, x_blk=y_blk, clamp01=True, out=x_hat[s0:s0 + n])
        else:  # the parity path: fp32 NCHW latents, y_hat = symbol + median exactly as the reference's decompress
            y_hat = ops.eb_dequantize(symbols, stride_b, stride_i, med, n, C, h, w, sym_offset=sym_offset)
            x_hat[s0:s0 + n] = net.g_s(y_hat).detach().clamp_(0, 1)

    def synthesise16(s0, n, symbols16):  # 16-bit symbols [stream][position] (the host's tiles)
        if fp16:
            y_blk = torch.empty((n, (C + 15) // 16, h, w, 16), device=dev, dtype=torch.float16)  # (every channel slot is written)
            ops.eb_dequantize16(symbols16, med, n, C, h, w, want_nchw=False, blk16=y_blk)
            engine.run_chain_fp16(net.g_s, x_blk=y_blk, clamp01=True, out=x_hat[s0:s0 + n])
        else:
            y_hat = ops.eb_dequantize16(symbols16, med, n, C, h, w)
            x_hat[s0:s0 + n] = net.g_s(y_hat).detach().clamp_(0, 1)

    def synthesise_device_pieces():
        for (s0, n, _), ev in zip(pieces, events):
            main.wait_event(ev)
            synthesise(s0, n, sym, 1, n_dev, sym_offset=s0 - n_host)

    # The host's tiles, sub-chunk by sub-chunk: decode (this thread blocks, the device decoders run), upload, synthesise.
    # When the call has device pieces as well, the host's tiles are synthesised on a stream of their own and the device
    # pieces' transforms are queued on the main stream right after the FIRST host sub-chunk (whose launches packed the
    # weights: the main stream waits for that event) - they start the moment their decode launch ends, while this
    # thread is still decoding the host's later sub-chunks (queued behind the host loop they started 3 ms late).
    if n_host:
        import contextlib
        hcdf, hlen, hoff, _ = eb.coder_tables_host()
        use16 = SYM16 and not ZERO_COPY and (not fp16 or (h * w) % 64 == 0)
        stage = _pinned_i32("dec", n_host, nsym) if not use16 else None
        stage16 = _pinned_i16("dec16", n_host, nsym) if use16 else None
        sub = max(1, HOST_SUB * ops.host_threads())
        hsyn = _stream(dev, "hostsyn") if pieces else None
        if hsyn is not None:
            hsyn.wait_event(start)
        queued_device = not pieces
        for (t0, m) in _ramp(n_host, 2 * ops.host_threads(), sub):
            w0 = time.perf_counter()
            part = strs[t0:t0 + m]
            lens = np.fromiter((len(b_) for b_ in part), dtype=np.int64, count=m)
            byte_off = np.zeros(m + 1, dtype=np.int64)
            np.cumsum(lens, out=byte_off[1:])
            data = np.frombuffer(b"".join(part), dtype=np.uint8)
            w1 = time.perf_counter()
            wide = not use16
            try:
                if use16:
                    bad = ops.rans_decode_host_sym16(data, byte_off, nsym, plane, hcdf, hlen, hoff, m, out=stage16[t0:t0 + m].numpy())
                    if bad == 3:  # a value outside 16 bits: this sub-chunk again, 32-bit symbols
                        wide = True
                        if stage is None:
                            stage = _pinned_i32("dec", n_host, nsym)
                if wide:
                    _, bad = ops.rans_decode_host(data, byte_off, nsym, plane, hcdf, hlen, hoff, m, out=stage[t0:t0 + m].numpy())
            except BaseException:
                torch.cuda.synchronize(dev)  # earlier sub-chunks' uploads still read the shared page-locked buffer
                raise
            w2 = time.perf_counter()
            _note_host_rate("dec", m, nsym, w2 - w1)
            if bad != 0:
                torch.cuda.synchronize(dev)  # nothing of this call may still be reading its buffers when the exception unwinds
                raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
            with (torch.cuda.stream(hsyn) if hsyn is not None else contextlib.nullcontext()):
                if wide:
                    hsym = stage[t0:t0 + m] if ZERO_COPY else stage[t0:t0 + m].to(dev, non_blocking=True)
                    synthesise(t0, m, hsym, nsym, 1)
                else:
                    hsym = stage16[t0:t0 + m].to(dev, non_blocking=True)
                    synthesise16(t0, m, hsym)
            keep.append((hsym,))
            if host_trace is not None:
                host_trace.append(("dec", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (time.perf_counter() - w2), 3), w0))
            if not queued_device:
                packed_ev = torch.cuda.Event()
                packed_ev.record(hsyn)
                main.wait_event(packed_ev)
                synthesise_device_pieces()
                queued_device = True
        if hsyn is not None:
            main.wait_stream(hsyn)
    else:
        synthesise_device_pieces()
    if host_trace is not None:
        host_trace.append(("dec-all-queued", n_host, time.perf_counter()))
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
    """The chunk-pipelined scale-hyperprior codec applies: a decoder image that fits LDS, and either device coder
    placement (large calls) or a host-coded call of at least one full sub-chunk (4 tiles per host thread), which runs
    the same pipeline with every tile in the host's share - transforms, PCIe and host coding overlap sub-chunk by
    sub-chunk instead of following one another as in the plain module path."""
    if net.gaussian_conditional.coder_image() is None:
        return False
    if not ops.host_coder_preferred(batch):
        return True
    return HOST_SPLIT and batch >= 4 * ops.host_threads()


def hyper_retry_chunk(chunk, ny):
    """Chunk size of the worst-case-capacity retry (cap_words = 2 ny + 8): licos_rans_encode_records addresses its word
    sink with 32-bit byte offsets, (cap + 1) * streams * 4 < 2^32 - an incompressible batch of M = 320 latents at 512^2
    would not fit at the default 2048 tiles per chunk."""
    cap = 2 * ny + 8
    return max(1, min(chunk, ((1 << 32) - 1) // (4 * (cap + 1))))


HYPER_HOST_NS = {"enc": float(os.environ.get("LICOS_HYPER_HOST_ENC_NS", "3.5")),   # host coder with explicit per-symbol rows, ns per symbol
                 "dec": float(os.environ.get("LICOS_HYPER_HOST_DEC_NS", "4.9"))}   # and thread (what balanced the two sides on the box)
HYPER_HOST_CODER_NS = {"enc": 3.7, "dec": 5.0}                                      # the y coder call alone (what _note_host_rate sees)
HYPER_DEV_NS = {"enc": 159.0, "dec": 117.0}                                         # the device y coders' latency per symbol
_hyper_share = {}


def hyper_host_share(batch, direction="enc"):
    """Tiles at the END of a compress_hyper / decompress_hyper call that the host codes (y and z streams): as many as the
    host threads code during the ONE device launch of the y coder that is exposed per call (31 ms encoding, 23 ms
    decoding a 512^2 tile's 196 608 symbols) - their transforms then run beside that launch instead of in front of /
    behind it.  A scale-hyperprior tile is 37 us of transforms on the encode side and the host codes it in 59 us
    (16 threads), so the host keeps up
    with the device for the length of that launch.  The count moves in steps of 4 x threads and only when the measured host
    rate has moved it by a whole step (a change of the device chunks' sizes costs the caching allocator a round of
    hipMalloc, 30 ms)."""
    if ops.HOST_CODER == "0" or not HOST_SPLIT:
        return 0
    if ops.host_coder_preferred(batch):  # a mid-size call (or LICOS_HOST_CODER=1): every tile, no device coder launch at all
        return batch
    if "LICOS_HYPER_SHARE" in os.environ:  # dev probe
        return min(int(os.environ["LICOS_HYPER_SHARE"]), batch // 3)
    threads = ops.host_threads()
    step = 4 * threads
    cap = 0.85 * threads * HYPER_DEV_NS[direction] / (HYPER_HOST_NS[direction] * _host_factor[direction])
    last = _hyper_share.get(direction)
    if last is None or abs(cap - last) >= step:
        last = _hyper_share[direction] = int(cap) // step * step
    return max(0, min(last, batch // 3 // step * step))


class _HostRange(Exception):
    """A y symbol of a host tile does not fit the 16 bits of the packed word (ops.gc_pack_symbols)."""


def compress_hyper(net, x, chunk=512, cap_words=None, host=True):
    """ScaleHyperprior.compress ([CAI] models/google.py) for a large batch, either precision: per chunk the four
    transforms run on the main stream, then ONE throughput kernel turns (y, scales) into per-symbol encoder records
    (licos_gc_encode_prepare) and the two serial coder kernels (z: plane coder, y: record coder) run on the side stream
    under the next chunk's transforms.  z_hat is round(z - median) + median computed directly: the reference obtains it
    by decoding the z string it has just written, which returns exactly those integers.  The call's last
    hyper_host_share(B) tiles are coded by the host cores (one word per y symbol - table row << 16 | symbol - and the z
    symbols, [stream][position] over PCIe, sub-chunk k queued before k - 1 is coded) while the last device launch runs;
    a y symbol outside 16 bits sends the whole call to the device (`host=False`)."""
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
    hcopy = _stream(dev, "hostsym")
    med = eb.medians_vec()
    bound = gc.lower_bound_scale.bound_value
    n_host = hyper_host_share(B) if (cap_words is None and host) else 0
    n_dev = B - n_host
    queued, shape = [], None
    for ci, (s0, n) in enumerate(_chunks(n_dev, chunk)):
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
        # (the z coder's 1.7 ms on a stream of its own: in front of the y coder on ONE stream it lengthened the exposed end of
        # the call by as much)
        zside = _stream(dev, "coder%d" % ((ci + 1) % CODER_STREAMS))  # (the neighbour chunk's: no further hardware queue)
        with torch.cuda.stream(zside):
            zside.wait_event(ready)
            zpart = _timed_coder("z_encode", lambda: ops.rans_encode_batch(zsym, 1, n, nz, zplane, zcdf, zlen, zoff, ztab, zcap, n))
            zcoded = torch.cuda.Event()
            zcoded.record(zside)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            ypart = _timed_coder("y_encode", lambda: ops.rans_encode_records(rec, aux, ycap))
            side.wait_event(zcoded)
            coded = torch.cuda.Event()
            coded.record(side)
        queued.append((s0, n, (y, z, zsym, rec, aux), ypart, zpart, coded))
        del y, z, zsym, z_hat, scales, rec, aux
    ys, zs = [None] * B, [None] * B
    segments = []
    overflow = False

    def drain(i):
        (s0, n, keep, ypart, zpart, coded) = queued[i]
        over, host_t, offs = _drain(dev, copy, coded, [ypart, zpart])
        if over:
            return True
        queued[i] = None  # the chunk's records (20 B per symbol) and scratch go back to the allocator
        mv = memoryview(host_t.numpy())
        yo, zo = offs
        ys[s0:s0 + n] = _split_bytes(mv, yo)
        zs[s0:s0 + n] = _split_bytes(mv, zo)
        segments.append((s0, n, host_t, yo, zo))
        return False

    if host_trace is not None:
        host_trace.append(("hyper-queued", n_host, time.perf_counter()))
    for i in range(len(queued) - 1):
        if drain(i):
            overflow = True
            break
    if host_trace is not None:
        host_trace.append(("hyper-drained", len(queued) - 1, time.perf_counter()))
    if n_host and not overflow:
        # the host's tiles: y symbols and their table rows, z symbols, [stream][position] int32 through page-locked buffers
        hz = eb.coder_tables_host()
        hy = gc.coder_tables_host()
        sub = max(1, 4 * ops.host_threads())
        subs = list(_ramp(n_host, ops.host_threads(), sub))
        hflag = torch.zeros(len(subs), device=dev, dtype=torch.int32)
        st_y = st_z = None
        st_f = _pinned_i32("hf", 1, max(64, len(subs)))[0]

        def host_encode(entry):
            (k, t0, m, _keep, landed) = entry
            w0 = time.perf_counter()
            landed.synchronize()
            if int(st_f[k]) != 0:
                raise _HostRange()
            w1 = time.perf_counter()
            yout, ynb = ops.rans_encode_host_packed(st_y[t0:t0 + m].numpy(), ny, hy[0], hy[1], hy[2], hy[3])
            w2 = time.perf_counter()
            _note_host_rate("enc", m, ny, w2 - w1, expect_ns=HYPER_HOST_CODER_NS["enc"])
            zout, znb = ops.rans_encode_host(st_z[t0:t0 + m].numpy(), nz, zplane, hz[0], hz[1], hz[2], hz[3])
            w3 = time.perf_counter()
            ys[n_dev + t0:n_dev + t0 + m] = [yout[k, : int(ynb[k])].tobytes() for k in range(m)]
            zs[n_dev + t0:n_dev + t0 + m] = [zout[k, : int(znb[k])].tobytes() for k in range(m)]
            if host_trace is not None:
                host_trace.append(("hyper-enc", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (w3 - w2), 3),
                                   round(1e3 * (time.perf_counter() - w3), 3)))

        pending = None
        try:
            for k, (t0, m) in enumerate(subs):
                y = net.g_a(x[n_dev + t0:n_dev + t0 + m])
                z = net.h_a(y)
                if shape is None:
                    shape = tuple(z.shape[-2:])
                    ny, nz, zplane = y[0].numel(), z[0].numel(), z[0, 0].numel()
                if st_y is None:
                    st_y, st_z = _pinned_i32("hy", n_host, ny), _pinned_i32("hz", n_host, nz)
                zsym = torch.empty((m, nz), device=dev, dtype=torch.int32)
                ops.eb_quantize(z, med, "symbols", symbols=zsym, sym_stride_b=nz, sym_stride_i=1)
                z_hat = ops.eb_quantize(z, med, "dequantize")
                scales = net.h_s(z_hat)
                ypk = torch.empty((m, ny), device=dev, dtype=torch.int32)
                ops.gc_pack_symbols(y.contiguous(), scales.contiguous(), gc.scale_table, bound, ypk, hflag[k:k + 1])
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(hcopy):
                    hcopy.wait_event(ready)
                    st_y[t0:t0 + m].copy_(ypk, non_blocking=True)
                    st_z[t0:t0 + m].copy_(zsym, non_blocking=True)
                    st_f[k:k + 1].copy_(hflag[k:k + 1], non_blocking=True)
                    landed = torch.cuda.Event()
                    landed.record(hcopy)
                entry = (k, t0, m, (ypk, zsym), landed)
                del y, z, z_hat, scales
                if pending is not None:
                    host_encode(pending)
                pending = entry
            if pending is not None:
                host_encode(pending)
        except _HostRange:
            torch.cuda.synchronize(dev)
            del queued
            return compress_hyper(net, x, chunk=chunk, host=False)
        except BaseException:
            torch.cuda.synchronize(dev)  # later sub-chunks' copies still target the shared page-locked buffers
            raise
    if host_trace is not None:
        host_trace.append(("hyper-host-done", n_host, time.perf_counter()))
    if queued and not overflow:
        overflow = drain(len(queued) - 1)
    if host_trace is not None:
        host_trace.append(("hyper-last-drained", 0, time.perf_counter()))
    if overflow:
        torch.cuda.synchronize(dev)
        if cap_words is not None:
            raise RuntimeError("licos_amd: rANS scratch overflow at worst-case capacity")
        del queued
        return compress_hyper(net, x, chunk=hyper_retry_chunk(chunk, ny), cap_words=2 * ny + 8)
    for ci in range(min(CODER_STREAMS, len(segments) + 1)):
        main.wait_stream(_stream(dev, "coder%d" % ci))
    main.wait_stream(copy)
    main.wait_stream(hcopy)
    ysegs = [(s0, n, t, yo) for (s0, n, t, yo, _) in segments]
    zsegs = [(s0, n, t, zo) for (s0, n, t, _, zo) in segments]
    return {"strings": [PackedStrings(ys, ysegs), PackedStrings(zs, zsegs)], "shape": torch.Size(shape)}


def _upload(strs, pieces, dev, id_base=0):
    """Per piece (s0, n): (device bytes, device int64 offsets [n+1]) of strs[s0:s0+n] - straight from compress()'s
    page-locked segment when `strs` still is what compress() returned and a segment starts at s0 and covers the piece,
    else re-joined through a staging buffer (the tiles the host encoded have no segment)."""
    from .entropy_models import EntropyBottleneck
    segs = {}
    if isinstance(strs, PackedStrings) and strs.still_packed():
        segs = {s: (n, host_t, off) for (s, n, host_t, off) in strs.segments}
    out = []
    for k, (s0, n) in enumerate(pieces):
        if s0 in segs and segs[s0][0] >= n:
            _, host_t, off = segs[s0]
            off = off[:n + 1]
            lo, hi = int(off[0]), int(off[-1])
            lo4 = lo & ~3
            data = host_t[lo4: max(hi, lo4 + 4)].to(dev, non_blocking=True)
            out.append((data, torch.from_numpy(off - lo4).to(dev, non_blocking=True)))
        else:
            out.append(EntropyBottleneck.pack_strings(strs[s0:s0 + n], dev, slot=(id_base + k)))
    return out


def decompress_hyper(net, strings, shape, chunk=512):
    """ScaleHyperprior.decompress for a large batch: every tile's z string is decoded in one launch, then per chunk
    h_s + the row-byte kernel run on the main stream and the chunk's y decoder on the side stream - ALL chunks' decoders
    are in flight before the first synthesis transform starts, which then overlaps the later chunks' decoding.  The
    call's last hyper_host_share(B, "dec") tiles are decoded by the host cores meanwhile (table rows [stream][position]
    down, symbols up through page-locked buffers, sub-chunk k + 1's rows queued before k is decoded) and synthesised on
    a stream of their own during the first device decoder launch, when the device has nothing else to do."""
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
    n_host = hyper_host_share(B, "dec") if gc.scale_table.numel() <= 256 else 0  # (the host's rows travel as bytes)
    n_dev = B - n_host
    pieces = [(s0, n) for (s0, n) in _chunks(n_dev, chunk)]
    # PackedStrings carry compress()'s own chunking; decode in those pieces when it is intact
    if isinstance(ystrs, PackedStrings) and ystrs.still_packed() and ystrs.segments:
        pieces, covered = [], 0
        for (s0, n, _, _) in ystrs.segments:
            if s0 != covered or covered >= n_dev:
                break
            pieces.append((s0, min(n, n_dev - s0)))
            covered = s0 + pieces[-1][1]
        pieces += [(covered + t0, m) for (t0, m) in _chunks(n_dev - covered, chunk)]  # tiles the host encoded but the device decodes
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    if n_dev == 0:
        # a host-coded call: the z strings too (a device launch would cost its 2 ms of latency for 0.1 ms of host work)
        hz = eb.coder_tables_host()
        zlens = np.fromiter((len(b_) for b_ in zstrs), dtype=np.int64, count=B)
        zbyte_off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(zlens, out=zbyte_off[1:])
        zstage = _pinned_i32("hzd", B, nz)
        _, zbad = ops.rans_decode_host(np.frombuffer(b"".join(zstrs), dtype=np.uint8), zbyte_off, nz, zplane, hz[0], hz[1], hz[2], B,
                                       out=zstage.numpy())
        if zbad != 0:
            raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
        zsym = zstage.to(dev, non_blocking=True)
        z_hat = ops.eb_dequantize(zsym, nz, 1, med, B, N, h, w)
    else:
        zsym = torch.empty((nz, B), device=dev, dtype=torch.int32)
        # z: every tile's string in ONE launch (2 ms whatever the batch; a launch per piece would queue them up on this stream)
        zup = _upload(zstrs, pieces + ([(n_dev, n_host)] if n_host else []), dev)
        if len(zup) > 1:  # (every string is a whole number of 32-bit words: the pieces concatenate without padding)
            zdata = torch.cat([data for (data, _) in zup])
            base, offs = 0, []
            for j, (data, off) in enumerate(zup):
                offs.append((off if j == len(zup) - 1 else off[:-1]) + base)
                base += data.numel()
            zoff_all = torch.cat(offs)
        else:
            zdata, zoff_all = zup[0]
        _timed_coder("z_decode", lambda: ops.rans_decode_batch(zdata, zoff_all, 1, B, nz, zplane, zcdf, zlen, zoff, zsym, B,
                                                               status=status, off_offset=0))
        z_hat = ops.eb_dequantize(zsym, 1, B, med, B, N, h, w)
    yup = _upload(ystrs, pieces, dev, id_base=len(pieces) + 1)  # staging slots behind the z pieces': no slot is shared in a call
    fp16 = net.precision == "fp16"
    st = engine.stages(net.g_s)
    cout = st[-1][0].out_channels
    x_hat = torch.empty((B, cout, 64 * h, 64 * w), device=dev, dtype=torch.float32)
    zeros = torch.zeros(M, device=dev, dtype=torch.float32)

    def synthesise(s0, n, sym, stride_b, stride_i):
        if fp16:
            y_blk = torch.empty((n, M // 16, 4 * h, 4 * w, 16), device=dev, dtype=torch.float16) if M % 16 == 0 else \
                torch.zeros((n, (M + 15) // 16, 4 * h, 4 * w, 16), device=dev, dtype=torch.float16)
            ops.eb_dequantize(sym, stride_b, stride_i, zeros, n, M, 4 * h, 4 * w, want_nchw=False, blk16=y_blk)
            engine.run_chain_fp16(net.g_s, x_blk=y_blk, clamp01=True, out=x_hat[s0:s0 + n])
        else:
            y_hat = ops.eb_dequantize(sym, stride_b, stride_i, zeros, n, M, 4 * h, 4 * w)
            x_hat[s0:s0 + n] = net.g_s(y_hat).detach().clamp_(0, 1)

    # the host's tiles: their table rows come down sub-chunk by sub-chunk (h_s and the row kernel on the stream `hsyn`)
    hsyn = _stream(dev, "hostsyn")
    hcopy = _stream(dev, "hostsym")
    st_i = st_s = None
    if n_host:
        st_i, st_s = _pinned_i32("hr", n_host, ny // 4).view(torch.uint8), _pinned_i32("hy", n_host, ny)  # (M % 4 == 0: ny too)
        zready = torch.cuda.Event()
        zready.record(main)
        hsyn.wait_event(zready)

    def host_rows(t0, m):
        with torch.cuda.stream(hsyn):
            scales = net.h_s(z_hat[n_dev + t0:n_dev + t0 + m])
            yidx = torch.empty((m, ny), device=dev, dtype=torch.uint8)
            ops.gc_build_rows8(scales.contiguous(), gc.scale_table, bound, yidx)
            ready = torch.cuda.Event()
            ready.record(hsyn)
        with torch.cuda.stream(hcopy):
            hcopy.wait_event(ready)
            st_i[t0:t0 + m].copy_(yidx, non_blocking=True)
            landed = torch.cuda.Event()
            landed.record(hcopy)
        return (t0, m, yidx, landed)

    subs = list(_ramp(n_host, ops.host_threads(), max(1, 4 * ops.host_threads()))) if n_host else []
    pending = None
    if subs:
        pending = host_rows(*subs[0])
        packed_ev = torch.cuda.Event()  # (the first h_s call packs the weights: the main stream's own follows it)
        packed_ev.record(hsyn)
        main.wait_event(packed_ev)
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

    def synthesise_device_pieces():
        for (s0, n), ev, (_, _, _, sym) in zip(pieces, events, keep):
            main.wait_event(ev)
            synthesise(s0, n, sym, 1, n)

    if subs:
        hy = gc.coder_tables_host()
        queued_device = False
        for k in range(len(subs)):
            (t0, m, _yidx, landed) = pending
            pending = host_rows(*subs[k + 1]) if k + 1 < len(subs) else None
            w0 = time.perf_counter()
            part = ystrs[n_dev + t0:n_dev + t0 + m]
            lens = np.fromiter((len(b_) for b_ in part), dtype=np.int64, count=m)
            byte_off = np.zeros(m + 1, dtype=np.int64)
            np.cumsum(lens, out=byte_off[1:])
            joined = np.frombuffer(b"".join(part), dtype=np.uint8)
            landed.synchronize()
            w1 = time.perf_counter()
            try:
                bad = ops.rans_decode_host_rows8(joined, byte_off, st_i[t0:t0 + m].numpy(), ny, hy[0], hy[1], hy[2], m,
                                                 out=st_s[t0:t0 + m].numpy())
            except BaseException:
                torch.cuda.synchronize(dev)  # earlier sub-chunks' uploads still read the shared page-locked buffer
                raise
            w2 = time.perf_counter()
            _note_host_rate("dec", m, ny, w2 - w1, expect_ns=HYPER_HOST_CODER_NS["dec"])
            if bad != 0:
                torch.cuda.synchronize(dev)
                raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
            with torch.cuda.stream(hsyn):
                hsym = st_s[t0:t0 + m].to(dev, non_blocking=True)
                synthesise(n_dev + t0, m, hsym, ny, 1)
            keep.append((hsym, _yidx))
            if host_trace is not None:
                host_trace.append(("hyper-dec", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (time.perf_counter() - w2), 3)))
            if not queued_device:  # (the first g_s call packed the weights)
                packed_ev = torch.cuda.Event()
                packed_ev.record(hsyn)
                main.wait_event(packed_ev)
                synthesise_device_pieces()
                queued_device = True
        main.wait_stream(hsyn)
    else:
        synthesise_device_pieces()
    if int(status.item()) != 0:  # synchronises; also keeps the side streams' tensors alive until they are done
        raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
    return {"x_hat": x_hat}
