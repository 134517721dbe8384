"""The factorized codec's chunk pipeline: FactorizedPrior.compress / decompress for any batch size and either precision
(fp16 MFMA transforms, or the fp32 parity path's split-operand transforms: the same pipeline, with fp32 NCHW activations
bounded per chunk).  Reference calls: /root/reference/eval_utils.py:199-204 (`net.compress`, then the strings' byte count).

The rANS recurrence is sequential inside a stream, so a coder launch is latency-bound: one lane per tile, a few waves in
total, a fixed ~N_symbols x chain-latency no matter how many tiles ride along.  The transforms are throughput-bound and
fill the chip.  The two therefore overlap almost for free: the batch is cut into chunks; while the MFMA kernels of chunk
k+1 run on the main stream, the coder kernel of chunk k runs on a side stream (encode), and symmetrically the decoder of
chunk k+1 runs under the synthesis transform of chunk k.  The host side of a chunk (stream lengths, compaction, copy of
the packed bytes into page-locked memory, building the per-tile ``bytes``) runs on a third stream while later chunks are
still being transformed.  Which tiles the host cores code instead: placement.py.  The byte strings are CompressAI's, one
per tile, and do not depend on the placement."""
import time

import numpy as np
import torch

from .. import engine, ops
from .config import config
from . import placement
from .placement import chunks, host_subchunks, note_host_rate
from .staging import HostRange, PackedStrings, exclusive, pinned_i16, pinned_i32, split_bytes, stream
from .trace import Section, trace


@exclusive(lambda net, x, *a, **k: x.device)
def compress_chunked(net, x, chunk=1024, cap_words=None, sym16=None):
    """FactorizedPrior.compress for any batch size and either precision (`net.g_a` dispatches on it).  Tiles
    [0, B - H) go through the device coder in pipeline chunks, the last H = placement.host_share(B) tiles through the host coder in
    sub-chunks (see "split placement" above); the strings do not depend on the placement."""
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, table = eb.coder_tables()
    if x.dtype != torch.float32 or x.dim() != 4:
        raise ValueError("licos_amd: compress expects a float32 (B, C, H, W) tensor")
    x = x.contiguous()
    B = x.shape[0]
    n_host = placement.host_share(B, "enc")
    if n_host == B and B <= config.simple_batch and ops.host_coder_preferred(B):
        # a handful of tiles: nothing to pipeline - transform, one copy, one host coder call (0.07 ms less per call than
        # the sub-chunk machinery below)
        y = net.g_a(x)
        return {"strings": [eb.compress(y)], "shape": y.size()[-2:]}
    n_dev = B - n_host
    dev = x.device
    main = torch.cuda.current_stream(dev)
    side = stream(dev, "coder")
    copy = stream(dev, "copy")
    hcopy = stream(dev, "hostsym")
    med = eb.medians_vec()
    sec = Section()
    sec.mark("c.start")
    sym = None
    shape = None
    # The plane encoder (rans.hip: the channel's records staged in LDS) stays: the record encoder of csrc/rans_gc.hip
    # (licos_eb_encode_prepare + licos_rans_encode_records) is 8 % faster per launch here (6.7 vs 7.3 ms) but its
    # throughput kernel writes 20 B per symbol - 1.8 ms per 4096-tile chunk on the main stream against 0.4 ms for the
    # symbols - and only the LAST launch of a call is exposed: measured, the step did not move.  LICOS_EB_RECORDS=1 switches.
    records = config.eb_records and eb.coder_image() is not None
    # fp16 transforms: the quantiser rides in the last analysis stage's epilogue (licos_conv5x5s2_f16_symbols) and writes the
    # symbols stream-major, [stream][position], which the plane encoder reads as each lane's own 32-byte runs - no fp32
    # latent in memory and no transposing quantise kernel.  The fp32 parity path keeps the separate kernel.
    stream_major = config.eb_stream_major and not records
    fused = (stream_major and net.precision == "fp16" and not getattr(net.g_a, "fp32_only", False)
             and engine.symbols_fusable(net.g_a))
    queued = []  # every tensor another stream touches stays referenced here until its chunk is drained
    for (s0, n) in chunks(n_dev, chunk):
        if fused:
            with torch.no_grad():
                y = engine.run_chain_fp16(net.g_a, x=x[s0:s0 + n], symbols=(med, None))  # int32 (n, C, h, w)
        else:
            y = net.g_a(x[s0:s0 + n])  # MFMA chain, main stream
        if shape is None:
            shape = tuple(y.shape[-2:])
            nsym, plane = y[0].numel(), y[0, 0].numel()
            if not records and not stream_major:
                sym = torch.empty((nsym, n_dev), device=dev, dtype=torch.int32)
            if cap_words is None:
                cap_words = nsym // 2 + 64
        if records:
            keep = ops.eb_encode_prepare(y.contiguous(), med, table, cdf_len, offset, cdf.shape[1])
        else:
            if fused:
                keep = y
            elif stream_major:  # (the fp32 parity path: its own quantise kernel, the same stream-major symbols for the encoder)
                keep = torch.empty((n, nsym), device=dev, dtype=torch.int32)
                ops.eb_quantize(y, med, "symbols", symbols=keep, sym_stride_b=nsym, sym_stride_i=1)
            else:
                ops.eb_quantize(y, med, "symbols", symbols=sym, sym_stride_b=1, sym_stride_i=n_dev, sym_offset=s0)
                keep = y
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            if records:
                words, nwords, status = ops.rans_encode_records(keep[0], keep[1], cap_words)
            elif stream_major:
                words, nwords, status = ops.rans_encode_batch(keep, nsym, 1, nsym, plane, cdf, cdf_len, offset, table, cap_words, n)
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
        strings[s0:s0 + n] = split_bytes(mv, off)
        segments.append((s0, n, host_t, off))
        return False

    # The host's first sub-chunks are queued BEFORE the device chunks are drained: a drain is milliseconds of this thread
    # (lengths, compaction, D2H, 4096 bytes objects), and with nothing queued behind the last device chunk's transforms the
    # GPU idled ~2 ms at the very place the call is exposed (tools/tail_probe.py).
    subs = host_subchunks(n_host)
    host_state = {"stage": None}
    use16 = (config.sym16 if sym16 is None else sym16) and not config.zero_copy
    hflag = torch.zeros(max(1, len(subs)), device=dev, dtype=torch.int32) if use16 else None
    st_f = pinned_i32(dev, "ef", 1, max(64, len(subs)))[0] if use16 else None

    def queue_sub(k):
        """Sub-chunk k of the host's tiles: transforms + quantise on the main stream, symbols to the page-locked buffer."""
        nonlocal shape, nsym, plane
        (t0, m) = subs[k]
        y = net.g_a(x[n_dev + t0:n_dev + t0 + m])
        if shape is None:
            shape = tuple(y.shape[-2:])
            nsym, plane = y[0].numel(), y[0, 0].numel()
        if host_state["stage"] is None:
            host_state["stage"] = pinned_i16(dev, "enc16", n_host, nsym) if use16 else pinned_i32(dev, "enc", n_host, nsym)
        stage = host_state["stage"]
        if config.zero_copy:
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

    prequeued = [queue_sub(k) for k in range(min(config.prequeue, len(subs)))] if len(queued) > 1 else []
    # every device chunk but the last (the last device launch runs beside the host's share below)
    for qi in range(len(queued) - 1):
        if drain(qi):
            overflow = True
            break
    trace.stamp("enc-drained", len(queued) - 1)
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
                    raise HostRange()
                out, nbytes = ops.rans_encode_host_sym16(stage[t0:t0 + m].numpy(), nsym, plane, hcdf, hlen, hoff, htable)
            else:
                out, nbytes = ops.rans_encode_host(stage[t0:t0 + m].numpy(), nsym, plane, hcdf, hlen, hoff, htable)
            w2 = time.perf_counter()
            note_host_rate("enc", m, nsym, w2 - w1)
            strings[n_dev + t0:n_dev + t0 + m] = [out[i, : int(nbytes[i])].tobytes() for i in range(m)]
            if trace.host_trace is not None:
                trace.host_trace.append(("enc", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (time.perf_counter() - w2), 3), w0))

        try:
            # a software pipeline in this thread: with sub-chunks 0 .. p - 1 queued, queue sub-chunk k + p, then code k
            ahead = max(1, len(prequeued))
            entries = list(prequeued)
            for k in range(len(subs)):
                while len(entries) < min(len(subs), k + ahead + 1):
                    entries.append(queue_sub(len(entries)))
                host_encode(entries[k])
                entries[k] = None
        except HostRange:  # a symbol outside 16 bits: the whole call again with 32-bit symbols for the host's tiles
            torch.cuda.synchronize(dev)
            del queued
            return compress_chunked(net, x, chunk=chunk, cap_words=cap_words, sym16=False)
        except BaseException:
            torch.cuda.synchronize(dev)  # later sub-chunks' copies still target the shared page-locked buffer: let them land
            raise
    trace.stamp("enc-host-done", n_host)
    if queued and not overflow:
        overflow = drain(len(queued) - 1)
    trace.stamp("enc-last-drained", 0)
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


@exclusive(lambda net, *a, **k: net.entropy_bottleneck.quantiles.device)
def decompress_chunked(net, strings, shape, chunk=1024):
    """FactorizedPrior.decompress.  The first H = placement.host_share(B) tiles are decoded by the host cores in sub-chunks - the
    synthesis transform starts on them about a millisecond into the call - while the device decodes the others (every
    device launch is queued before the host starts); see "split placement" above."""
    eb = net.entropy_bottleneck
    cdf, cdf_len, offset, _ = eb.coder_tables()
    assert isinstance(strings, list) and len(strings) == 1
    strs = strings[0]
    B = len(strs)
    n_host = placement.host_share(B, "dec")
    if n_host == B and B <= config.simple_batch and ops.host_coder_preferred(B):
        y_hat = eb.decompress(list(strs), shape)
        x_hat = net.g_s(y_hat)
        return {"x_hat": x_hat.clamp_(0, 1)}
    dev = cdf.device
    C = cdf.shape[0]
    h, w = int(shape[0]), int(shape[1])
    nsym, plane = C * h * w, h * w
    main = torch.cuda.current_stream(dev)
    side = stream(dev, "coder")
    med = eb.medians_vec()
    sec = Section()
    sec.mark("d.start")
    n_dev = B - n_host
    # device-decoded tiles: [stream][position] from the round-5 plane decoder (16-byte stores from registers), [position][stream]
    # from the image decoder
    stream_major = config.eb_stream_major and not config.eb_image
    sym = (torch.empty((n_dev, nsym) if stream_major else (nsym, n_dev), device=dev, dtype=torch.int32)) if n_dev else None
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    image = eb.coder_image() if config.eb_image else None  # the image decoder (csrc/rans_gc.hip), channel pattern as shared rows
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
        pieces += [(covered + t0, m, None, None) for (t0, m) in chunks(B - covered, chunk)]
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
            elif stream_major:
                ops.rans_decode_batch(data, byte_off, nsym, 1, nsym, plane, cdf, cdf_len, offset, sym, n, sym_offset=(s0 - n_host) * nsym,
                                      status=status, off_offset=0)
            else:
                ops.rans_decode_batch(data, byte_off, 1, n_dev, nsym, plane, cdf, cdf_len, offset, sym, n, sym_offset=s0 - n_host,
                                      status=status, off_offset=0)
            ev = torch.cuda.Event()
            ev.record(side)
        keep.append((data, byte_off))
        events.append(ev)
    sec.mark("d.queue H2D+decode")
    trace.stamp("dec-launches-queued", len(pieces))
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
            if stream_major:
                synthesise(s0, n, sym, nsym, 1, sym_offset=(s0 - n_host) * nsym)
            else:
                synthesise(s0, n, sym, 1, n_dev, sym_offset=s0 - n_host)

    # The host's tiles, sub-chunk by sub-chunk: decode (this thread blocks, the device decoders run), upload, synthesise.
    # When the call has device pieces as well, the host's tiles are synthesised on a stream of their own and the device
    # pieces' transforms are queued on the main stream right after the FIRST host sub-chunk (whose launches packed the
    # weights: the main stream waits for that event) - they start the moment their decode launch ends, while this
    # thread is still decoding the host's later sub-chunks (queued behind the host loop they started 3 ms late).
    if n_host:
        import contextlib
        hcdf, hlen, hoff, _ = eb.coder_tables_host()
        use16 = config.sym16 and not config.zero_copy and (not fp16 or (h * w) % 64 == 0)
        stage = pinned_i32(dev, "dec", n_host, nsym) if not use16 else None
        stage16 = pinned_i16(dev, "dec16", n_host, nsym) if use16 else None
        hsyn = stream(dev, "hostsyn") if pieces else None
        if hsyn is not None:
            hsyn.wait_event(start)
        queued_device = not pieces
        for (t0, m) in host_subchunks(n_host):
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
                            stage = pinned_i32(dev, "dec", n_host, nsym)
                if wide:
                    _, bad = ops.rans_decode_host(data, byte_off, nsym, plane, hcdf, hlen, hoff, m, out=stage[t0:t0 + m].numpy())
            except BaseException:
                torch.cuda.synchronize(dev)  # earlier sub-chunks' uploads still read the shared page-locked buffer
                raise
            w2 = time.perf_counter()
            note_host_rate("dec", m, nsym, w2 - w1)
            if bad != 0:
                torch.cuda.synchronize(dev)  # nothing of this call may still be reading its buffers when the exception unwinds
                raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
            with (torch.cuda.stream(hsyn) if hsyn is not None else contextlib.nullcontext()):
                if wide:
                    hsym = stage[t0:t0 + m] if config.zero_copy else stage[t0:t0 + m].to(dev, non_blocking=True)
                    synthesise(t0, m, hsym, nsym, 1)
                else:
                    hsym = stage16[t0:t0 + m].to(dev, non_blocking=True)
                    synthesise16(t0, m, hsym)
            keep.append((hsym,))
            if trace.host_trace is not None:
                trace.host_trace.append(("dec", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (time.perf_counter() - w2), 3), w0))
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
    trace.stamp("dec-all-queued", n_host)
    sec.mark("d.decode+transforms (device)")
    if int(status.item()) != 0:  # synchronises; also keeps data/sym alive until the side stream is done
        raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
    return {"x_hat": x_hat}
