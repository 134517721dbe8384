"""The scale hyperprior's chunk pipeline (BASELINE configs[4]): ScaleHyperprior.compress / decompress for large batches,
either precision.  Reference: [CAI] models/google.py ScaleHyperprior.compress / decompress, admitted by
/root/reference/licos/model_utils.py:20-24.  Placement of the host's share: placement.py."""
import time

import numpy as np
import torch

from .. import engine, ops
from .config import config
from . import placement
from .placement import chunks, hyper_retry_chunk, hyper_subchunks, note_host_rate
from .staging import HostRange, PackedStrings, exclusive, pinned_i32, split_bytes, stream
from .trace import timed_coder, trace


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


@exclusive(lambda net, x, *a, **k: x.device)
def compress_hyper(net, x, chunk=512, cap_words=None, host=True):
    """ScaleHyperprior.compress ([CAI] models/google.py) for a large batch, either precision: per chunk the four
    transforms run on the main stream, then ONE throughput kernel turns (y, scales) into per-symbol encoder records
    (licos_gc_encode_prepare) and the two serial coder kernels (z: plane coder, y: record coder) run on the side stream
    under the next chunk's transforms.  z_hat is round(z - median) + median computed directly: the reference obtains it
    by decoding the z string it has just written, which returns exactly those integers.  The call's last
    placement.hyper_host_share(B) tiles are coded by the host cores (one word per y symbol - table row << 16 | symbol - and the z
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
    copy = stream(dev, "copy")
    hcopy = stream(dev, "hostsym")
    med = eb.medians_vec()
    bound = gc.lower_bound_scale.bound_value
    n_host = placement.hyper_host_share(B) if (cap_words is None and host) else 0
    n_dev = B - n_host
    queued, shape = [], None
    for ci, (s0, n) in enumerate(chunks(n_dev, chunk)):
        # a coder launch is a latency chain on a handful of CUs: the chunks' launches run side by side, each on a
        # stream of its own (on ONE stream they would queue up behind each other, ~50 ms apiece)
        side = stream(dev, "coder%d" % (ci % config.coder_streams))
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
        zside = stream(dev, "coder%d" % ((ci + 1) % config.coder_streams))  # (the neighbour chunk's: no further hardware queue)
        with torch.cuda.stream(zside):
            zside.wait_event(ready)
            zpart = timed_coder("z_encode", lambda: ops.rans_encode_batch(zsym, 1, n, nz, zplane, zcdf, zlen, zoff, ztab, zcap, n))
            zcoded = torch.cuda.Event()
            zcoded.record(zside)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            ypart = timed_coder("y_encode", lambda: ops.rans_encode_records(rec, aux, ycap))
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
        ys[s0:s0 + n] = split_bytes(mv, yo)
        zs[s0:s0 + n] = split_bytes(mv, zo)
        segments.append((s0, n, host_t, yo, zo))
        return False

    trace.stamp("hyper-queued", n_host)
    for i in range(len(queued) - 1):
        if drain(i):
            overflow = True
            break
    trace.stamp("hyper-drained", len(queued) - 1)
    if n_host and not overflow:
        # the host's tiles: y symbols and their table rows, z symbols, [stream][position] int32 through page-locked buffers
        hz = eb.coder_tables_host()
        hy = gc.coder_tables_host()
        subs = hyper_subchunks(n_host)
        hflag = torch.zeros(len(subs), device=dev, dtype=torch.int32)
        st_y = st_z = None
        st_f = pinned_i32(dev, "hf", 1, max(64, len(subs)))[0]

        def host_encode(entry):
            (k, t0, m, _keep, landed) = entry
            w0 = time.perf_counter()
            landed.synchronize()
            if int(st_f[k]) != 0:
                raise HostRange()
            w1 = time.perf_counter()
            yout, ynb = ops.rans_encode_host_packed(st_y[t0:t0 + m].numpy(), ny, hy[0], hy[1], hy[2], hy[3])
            w2 = time.perf_counter()
            note_host_rate("enc", m, ny, w2 - w1, expect_ns=config.hyper_host_coder_ns["enc"])
            zout, znb = ops.rans_encode_host(st_z[t0:t0 + m].numpy(), nz, zplane, hz[0], hz[1], hz[2], hz[3])
            w3 = time.perf_counter()
            ys[n_dev + t0:n_dev + t0 + m] = [yout[k, : int(ynb[k])].tobytes() for k in range(m)]
            zs[n_dev + t0:n_dev + t0 + m] = [zout[k, : int(znb[k])].tobytes() for k in range(m)]
            if trace.host_trace is not None:
                trace.host_trace.append(("hyper-enc", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (w3 - w2), 3),
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
                    st_y, st_z = pinned_i32(dev, "hy", n_host, ny), pinned_i32(dev, "hz", n_host, nz)
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
        except HostRange:
            torch.cuda.synchronize(dev)
            del queued
            return compress_hyper(net, x, chunk=chunk, host=False)
        except BaseException:
            torch.cuda.synchronize(dev)  # later sub-chunks' copies still target the shared page-locked buffers
            raise
    trace.stamp("hyper-host-done", n_host)
    if queued and not overflow:
        overflow = drain(len(queued) - 1)
    trace.stamp("hyper-last-drained", 0)
    if overflow:
        torch.cuda.synchronize(dev)
        if cap_words is not None:
            raise RuntimeError("licos_amd: rANS scratch overflow at worst-case capacity")
        del queued
        return compress_hyper(net, x, chunk=hyper_retry_chunk(chunk, ny), cap_words=2 * ny + 8)
    for ci in range(min(config.coder_streams, len(segments) + 1)):
        main.wait_stream(stream(dev, "coder%d" % ci))
    main.wait_stream(copy)
    main.wait_stream(hcopy)
    ysegs = [(s0, n, t, yo) for (s0, n, t, yo, _) in segments]
    zsegs = [(s0, n, t, zo) for (s0, n, t, _, zo) in segments]
    return {"strings": [PackedStrings(ys, ysegs), PackedStrings(zs, zsegs)], "shape": torch.Size(shape)}


def _upload(strs, pieces, dev, id_base=0):
    """Per piece (s0, n): (device bytes, device int64 offsets [n+1]) of strs[s0:s0+n] - straight from compress()'s
    page-locked segment when `strs` still is what compress() returned and a segment starts at s0 and covers the piece,
    else re-joined through a staging buffer (the tiles the host encoded have no segment)."""
    from ..entropy_models import EntropyBottleneck
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


@exclusive(lambda net, *a, **k: net.entropy_bottleneck.quantiles.device)
def decompress_hyper(net, strings, shape, chunk=512):
    """ScaleHyperprior.decompress for a large batch: every tile's z string is decoded in one launch, then per chunk
    h_s + the row-byte kernel run on the main stream and the chunk's y decoder on the side stream - ALL chunks' decoders
    are in flight before the first synthesis transform starts, which then overlaps the later chunks' decoding.  The
    call's last placement.hyper_host_share(B, "dec") tiles are decoded by the host cores meanwhile (table rows [stream][position]
    down, symbols up through page-locked buffers, sub-chunk k + 1's rows queued before k is decoded) and synthesised on
    a stream of their own during the first device decoder launch, when the device has nothing else to do."""
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
    n_host = placement.hyper_host_share(B, "dec") if gc.scale_table.numel() <= 256 else 0  # (the host's rows travel as bytes)
    n_dev = B - n_host
    pieces = [(s0, n) for (s0, n) in chunks(n_dev, chunk)]
    # PackedStrings carry compress()'s own chunking; decode in those pieces when it is intact
    if isinstance(ystrs, PackedStrings) and ystrs.still_packed() and ystrs.segments:
        pieces, covered = [], 0
        for (s0, n, _, _) in ystrs.segments:
            if s0 != covered or covered >= n_dev:
                break
            pieces.append((s0, min(n, n_dev - s0)))
            covered = s0 + pieces[-1][1]
        pieces += [(covered + t0, m) for (t0, m) in chunks(n_dev - covered, chunk)]  # tiles the host encoded but the device decodes
    status = torch.zeros(1, device=dev, dtype=torch.int32)
    if n_dev == 0:
        # a host-coded call: the z strings too (a device launch would cost its 2 ms of latency for 0.1 ms of host work)
        hz = eb.coder_tables_host()
        zlens = np.fromiter((len(b_) for b_ in zstrs), dtype=np.int64, count=B)
        zbyte_off = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(zlens, out=zbyte_off[1:])
        zstage = pinned_i32(dev, "hzd", B, nz)
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
        timed_coder("z_decode", lambda: ops.rans_decode_batch(zdata, zoff_all, 1, B, nz, zplane, zcdf, zlen, zoff, zsym, B,
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
    hsyn = stream(dev, "hostsyn")
    hcopy = stream(dev, "hostsym")
    st_i = st_s = None
    if n_host:
        st_i, st_s = pinned_i32(dev, "hr", n_host, ny // 4).view(torch.uint8), pinned_i32(dev, "hy", n_host, ny)  # (M % 4 == 0: ny too)
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

    subs = hyper_subchunks(n_host)
    pending = None
    if subs:
        pending = host_rows(*subs[0])
        packed_ev = torch.cuda.Event()  # (the first h_s call packs the weights: the main stream's own follows it)
        packed_ev.record(hsyn)
        main.wait_event(packed_ev)
    events, keep = [], []
    for ci, ((s0, n), (data, off)) in enumerate(zip(pieces, yup)):
        side = stream(dev, "coder%d" % (ci % config.coder_streams))
        scales = net.h_s(z_hat[s0:s0 + n])
        idx16 = ops.gc_decode_prepare(scales.contiguous(), gc.scale_table, bound, row_hist=row_hist)
        sym = torch.empty((ny, n), device=dev, dtype=torch.int32)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            timed_coder("y_decode", lambda: ops.rans_decode_image(data, off, idx16, ny, image_dev, image_host, sym, 1, n, n, status=status))
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
            note_host_rate("dec", m, ny, w2 - w1, expect_ns=config.hyper_host_coder_ns["dec"])
            if bad != 0:
                torch.cuda.synchronize(dev)
                raise ValueError("licos_amd: a rANS string ended before all symbols were decoded")
            with torch.cuda.stream(hsyn):
                hsym = st_s[t0:t0 + m].to(dev, non_blocking=True)
                synthesise(n_dev + t0, m, hsym, ny, 1)
            keep.append((hsym, _yidx))
            if trace.host_trace is not None:
                trace.host_trace.append(("hyper-dec", m, round(1e3 * (w1 - w0), 3), round(1e3 * (w2 - w1), 3), round(1e3 * (time.perf_counter() - w2), 3)))
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
