"""CPU: the host rANS coder of the product library (licos_rans_encode_host / licos_rans_decode_host) against the
oracle's C and Python statements of CompressAI's coder - the golden KAT, escapes on both sides incl. multi-nibble and
chained-count bypass, explicit indexes and channel-plane rows, ragged thread counts, truncated streams."""
import os

import numpy as np
import pytest

from licos_amd import ops
from oracle import model as om
from oracle import rans


def _tables(seed=0):
    sd = om.perturb_state(om.make_factorized_state(3, 1), seed=seed)
    om.eb_update(sd)
    cdf = sd["entropy_bottleneck._quantized_cdf"].numpy()
    cl = sd["entropy_bottleneck._cdf_length"].numpy()
    off = sd["entropy_bottleneck._offset"].numpy()
    return cdf, cl, off, ops.rans_build_enc_table(cdf, cl)


def _strings(out, nbytes):
    return [out[b, : int(nbytes[b])].tobytes() for b in range(out.shape[0])]


def test_host_coder_kat(golden_dir):
    g = np.load(os.path.join(golden_dir, "coder_kat.npz"))
    table = ops.rans_build_enc_table(g["cdfs"], g["cdf_len"])
    sym = g["sym"].astype(np.int32).reshape(1, -1)
    idx = g["idx"].astype(np.int32).reshape(1, -1)
    out, nb = ops.rans_encode_host(sym, sym.shape[1], 0, g["cdfs"], g["cdf_len"], g["offset"], table, indexes=idx, nthreads=1)
    assert _strings(out, nb)[0] == g["data"].tobytes()
    dec, st = ops.rans_decode_host(np.frombuffer(g["data"].tobytes(), dtype=np.uint8), np.array([0, g["data"].size]), sym.shape[1],
                                   0, g["cdfs"], g["cdf_len"], g["offset"], 1, indexes=idx, nthreads=1)
    assert st == 0 and np.array_equal(dec, sym)


@pytest.mark.parametrize("batch,threads", [(1, 1), (5, 3), (37, 8)])
def test_host_coder_equals_oracle_on_channel_planes(batch, threads):
    """EntropyBottleneck addressing (row = position / plane), symbols incl. far outliers on both sides."""
    cdf, cl, off, table = _tables(seed=batch)
    rng = np.random.default_rng(batch)
    c, plane = 192, 24
    n = c * plane
    sym = np.rint(rng.standard_normal((batch, n)) * 9).astype(np.int32)
    sym[0, 5] = 1234
    sym[-1, n - 1] = -70000
    sym[batch // 2, 17] = 3_000_000
    sym[0, 100] = -(2 ** 30) + 5      # 8 nibbles (beyond +-2^30 the int32 arithmetic of the C++ original wraps)
    sym[-1, 200] = 2 ** 30 - 7
    idx = np.repeat(np.arange(c, dtype=np.int32), plane)
    out, nb = ops.rans_encode_host(sym, n, plane, cdf, cl, off, table, nthreads=threads)
    got = _strings(out, nb)
    ref = [rans.encode_with_indexes(sym[b], idx, cdf, cl, off) for b in range(batch)]
    assert got == ref
    data = np.frombuffer(b"".join(got), dtype=np.uint8)
    byte_off = np.concatenate(([0], np.cumsum(nb))).astype(np.int64)
    dec, st = ops.rans_decode_host(data, byte_off, n, plane, cdf, cl, off, batch, nthreads=threads)
    assert st == 0 and np.array_equal(dec, sym)
    # a small case against the independent pure-Python coder as well
    k = 600
    o2, n2 = ops.rans_encode_host(sym[:1, :k], k, plane, cdf, cl, off, table, nthreads=1)
    assert _strings(o2, n2)[0] == rans.py_encode_with_indexes(sym[0, :k], idx[:k], cdf, cl, off)


def test_host_coder_explicit_indexes_and_capacity_retry():
    cdf, cl, off, table = _tables(seed=3)
    rng = np.random.default_rng(7)
    n = 5000
    idx = rng.integers(0, 192, size=(3, n)).astype(np.int32)
    sym = np.rint(rng.standard_normal((3, n)) * 4e6).astype(np.int32)  # all escapes, 6-7 nibbles each: needs the worst-case capacity
    out, nb = ops.rans_encode_host(sym, n, 0, cdf, cl, off, table, indexes=idx, nthreads=2)
    got = _strings(out, nb)
    assert got == [rans.encode_with_indexes(sym[b], idx[b], cdf, cl, off) for b in range(3)]
    assert max(nb) > 4 * (n // 2 + 64)
    data = np.frombuffer(b"".join(got), dtype=np.uint8)
    dec, st = ops.rans_decode_host(data, np.concatenate(([0], np.cumsum(nb))), n, 0, cdf, cl, off, 3, indexes=idx, nthreads=2)
    assert st == 0 and np.array_equal(dec, sym)


def test_host_decoder_chained_bypass_count_and_truncation():
    cdfs = np.array([[0, 40000, 65536]], dtype=np.int32)
    cdf_len, offset = np.array([3], dtype=np.int32), np.array([0], dtype=np.int32)
    items = [(0, 40000, False), (40000, 25536, False), (15, 0, True), (2, 0, True)] + [(0, 0, True)] * 17 + [(0, 40000, False)]
    x, words = 1 << 31, []
    for start, rng_, byp in reversed(items):
        freq = (1 << 12) if byp else rng_
        if x >= ((1 << 31 >> 16) << 32) * freq:
            words.append(x & 0xFFFFFFFF)
            x >>= 32
        x = ((x << 4) | start) if byp else ((x // rng_) << 16) + (x % rng_) + start
    words += [x >> 32, x & 0xFFFFFFFF]
    data = np.frombuffer(b"".join(int(w).to_bytes(4, "little") for w in reversed(words)), dtype=np.uint8)
    dec, st = ops.rans_decode_host(data, np.array([0, data.size]), 3, 3, cdfs, cdf_len, offset, 1, nthreads=1)
    assert st == 0 and dec.ravel().tolist() == [0, 1, 0]
    # truncated and garbage streams: flagged, zero-filled, never read outside the stream
    cdf, cl, off, table = _tables(seed=1)
    n = 192 * 8
    sym = np.rint(np.random.default_rng(0).standard_normal((1, n)) * 6).astype(np.int32)
    out, nb = ops.rans_encode_host(sym, n, 8, cdf, cl, off, table, nthreads=1)
    s = out[0, : int(nb[0])]
    dec, st = ops.rans_decode_host(s[: s.size // 2 // 4 * 4].copy(), np.array([0, s.size // 2 // 4 * 4]), n, 8, cdf, cl, off, 1)
    assert st == 1 and dec.shape == (1, n)
    junk = np.random.default_rng(1).integers(0, 256, size=64, dtype=np.uint8)
    dec, st = ops.rans_decode_host(junk, np.array([0, 64]), n, 8, cdf, cl, off, 1)
    assert st == 1
    dec, st = ops.rans_decode_host(np.zeros(4, dtype=np.uint8), np.array([0, 0]), n, 8, cdf, cl, off, 1)
    assert st == 1 and not dec.any()


@pytest.mark.parametrize("batch,threads", [(1, 1), (7, 3), (22, 8)])
def test_host_coder_compact_forms_equal_the_oracle(batch, threads):
    """licos_rans_encode_host_packed (one word per symbol: row << 16 | symbol & 0xFFFF) and licos_rans_decode_host_rows8
    (one row byte per symbol) - what crosses PCIe when the host cores take a share of a scale-hyperprior call - give the
    oracle's bytes and symbols: explicit rows changing from symbol to symbol, escapes on both sides up to the 16-bit ends,
    groups of 4, 2 and single streams, truncated streams."""
    cdf, cl, off, table = _tables(seed=batch)
    rng = np.random.default_rng(100 + batch)
    n = 3001
    idx = rng.integers(0, 192, size=(batch, n)).astype(np.int32)
    sym = np.rint(rng.standard_normal((batch, n)) * 7).astype(np.int32)
    sym[0, 0], sym[0, 9], sym[-1, n - 1], sym[-1, 11] = 32767, -32768, -1234, 4321
    packed = ((idx.astype(np.int64) << 16) | (sym.astype(np.int64) & 0xFFFF)).astype(np.uint32).view(np.int32)
    out, nb = ops.rans_encode_host_packed(packed, n, cdf, cl, off, table, nthreads=threads)
    got = _strings(out, nb)
    assert got == [rans.encode_with_indexes(sym[b], idx[b], cdf, cl, off) for b in range(batch)]
    o2, n2 = ops.rans_encode_host(sym, n, 0, cdf, cl, off, table, indexes=idx, nthreads=threads)
    assert _strings(o2, n2) == got
    data = np.frombuffer(b"".join(got), dtype=np.uint8)
    byte_off = np.concatenate(([0], np.cumsum(nb))).astype(np.int64)
    dec = np.full((batch, n), -7, dtype=np.int32)
    assert ops.rans_decode_host_rows8(data, byte_off, idx.astype(np.uint8), n, cdf, cl, off, batch, out=dec, nthreads=threads) == 0
    assert np.array_equal(dec, sym)
    # the last stream cut short: status 1, the other streams intact
    cut = byte_off.copy()
    cut[-1] -= 8
    dec2 = np.full((batch, n), -7, dtype=np.int32)
    assert ops.rans_decode_host_rows8(data[: cut[-1]], cut, idx.astype(np.uint8), n, cdf, cl, off, batch, out=dec2, nthreads=threads) == 1
    assert np.array_equal(dec2[:-1], sym[:-1])
    # a row outside the table is an error, not a read outside it
    bad = packed.copy()
    bad[0, 5] = (200 << 16) | 3
    with pytest.raises(Exception):
        ops.rans_encode_host_packed(bad, n, cdf, cl, off, table, nthreads=threads)


@pytest.mark.parametrize("batch,threads", [(1, 1), (6, 3), (21, 8)])
def test_host_coder_16_bit_symbols_equal_the_oracle(batch, threads):
    """licos_rans_encode_host_sym16 / licos_rans_decode_host_sym16 (channel-plane rows, int16 symbols: what the factorized
    codec's host share moves over PCIe): the oracle's bytes and symbols, escapes to the 16-bit ends; a stream whose values
    do not fit int16 is reported (status 3), not truncated."""
    cdf, cl, off, table = _tables(seed=10 + batch)
    rng = np.random.default_rng(batch)
    c, plane = 192, 16
    n = c * plane
    sym = np.rint(rng.standard_normal((batch, n)) * 9).astype(np.int32)
    sym[0, 3], sym[0, 8], sym[-1, n - 1] = 32767, -32768, -3000
    idx = np.repeat(np.arange(c, dtype=np.int32), plane)
    out, nb = ops.rans_encode_host_sym16(sym.astype(np.int16), n, plane, cdf, cl, off, table, nthreads=threads)
    got = _strings(out, nb)
    assert got == [rans.encode_with_indexes(sym[b], idx, cdf, cl, off) for b in range(batch)]
    data = np.frombuffer(b"".join(got), dtype=np.uint8)
    byte_off = np.concatenate(([0], np.cumsum(nb))).astype(np.int64)
    dec = np.full((batch, n), -7, dtype=np.int16)
    assert ops.rans_decode_host_sym16(data, byte_off, n, plane, cdf, cl, off, batch, out=dec, nthreads=threads) == 0
    assert np.array_equal(dec.astype(np.int32), sym)
    # a value beyond 16 bits in the last stream: status 3 from the 16-bit decoder, the 32-bit one returns it
    big = sym.copy()
    big[-1, 40] = 40000
    o2, n2 = ops.rans_encode_host(big, n, plane, cdf, cl, off, table, nthreads=threads)
    d2 = np.frombuffer(b"".join(_strings(o2, n2)), dtype=np.uint8)
    bo2 = np.concatenate(([0], np.cumsum(n2))).astype(np.int64)
    assert ops.rans_decode_host_sym16(d2, bo2, n, plane, cdf, cl, off, batch, out=dec, nthreads=threads) == 3
    full, st = ops.rans_decode_host(d2, bo2, n, plane, cdf, cl, off, batch, nthreads=threads)
    assert st == 0 and np.array_equal(full, big)
    # a truncated stream
    cut = byte_off.copy()
    cut[-1] -= 8
    assert ops.rans_decode_host_sym16(data[: cut[-1]], cut, n, plane, cdf, cl, off, batch, out=dec, nthreads=threads) == 1
