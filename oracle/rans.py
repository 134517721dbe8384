"""ctypes front-end of oracle/rans_oracle.c plus a pure-Python cross-check.

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle/__init__.py).
Restates CompressAI ``cpp_exts/rans/rans_interface.cpp`` +
``third_party/ryg_rans/rans64.h`` (reached from /root/reference/eval_utils.py:201).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build():
    src = os.path.join(_HERE, "rans_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.oracle_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                                  ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
        L.oracle_rans_encode.restype = ctypes.c_long
        L.oracle_rans_encode.argtypes = [i32p, i32p, ctypes.c_int, i32p, ctypes.c_int, i32p, i32p,
                                         u8p, ctypes.c_long]
        L.oracle_rans_decode.restype = ctypes.c_int
        L.oracle_rans_decode.argtypes = [u8p, ctypes.c_long, i32p, ctypes.c_int, i32p, ctypes.c_int,
                                         i32p, i32p, i32p]
        _lib = L
    return _lib


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(pmf.size + 1, dtype=np.uint32)
    rc = lib().oracle_pmf_to_quantized_cdf(_p(pmf, ctypes.c_float), pmf.size, precision,
                                           _p(out, ctypes.c_uint32))
    if rc != 0:
        raise ValueError(f"invalid pmf (code {rc})")
    return out.astype(np.int32)


def encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    symbols, indexes = _i32(symbols).ravel(), _i32(indexes).ravel()
    cdfs, cdf_sizes, offsets = _i32(cdfs), _i32(cdf_sizes).ravel(), _i32(offsets).ravel()
    assert cdfs.ndim == 2 and symbols.size == indexes.size
    cap = symbols.size * 8 + 64
    out = np.empty(cap, dtype=np.uint8)
    n = lib().oracle_rans_encode(_p(symbols, ctypes.c_int32), _p(indexes, ctypes.c_int32), symbols.size,
                                 _p(cdfs, ctypes.c_int32), cdfs.shape[1], _p(cdf_sizes, ctypes.c_int32),
                                 _p(offsets, ctypes.c_int32), _p(out, ctypes.c_uint8), cap)
    if n < 0:
        raise RuntimeError(f"oracle_rans_encode failed ({n})")
    return out[:n].tobytes()


def decode_with_indexes(data, indexes, cdfs, cdf_sizes, offsets):
    indexes = _i32(indexes).ravel()
    cdfs, cdf_sizes, offsets = _i32(cdfs), _i32(cdf_sizes).ravel(), _i32(offsets).ravel()
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    out = np.empty(indexes.size, dtype=np.int32)
    rc = lib().oracle_rans_decode(_p(buf, ctypes.c_uint8), buf.size, _p(indexes, ctypes.c_int32), indexes.size,
                                  _p(cdfs, ctypes.c_int32), cdfs.shape[1], _p(cdf_sizes, ctypes.c_int32),
                                  _p(offsets, ctypes.c_int32), _p(out, ctypes.c_int32))
    if rc != 0:
        raise RuntimeError(f"oracle_rans_decode failed ({rc})")
    return out


# ---- pure-Python second statement (small cases only; cross-checks the C) -------
_L = 1 << 31


def py_pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.asarray(pmf, dtype=np.float32)
    if not (np.all(np.isfinite(pmf)) and np.all(pmf >= 0)):
        raise ValueError("invalid pmf")
    scaled = (pmf * np.float32(1 << precision)).astype(np.float32)
    # C round(): half away from zero (values are >= 0 here)
    freqs = [int(np.floor(float(v) + 0.5)) for v in scaled]
    cdf = [0] + freqs
    total = sum(cdf)
    if total == 0:
        raise ValueError("invalid pmf")
    cdf = [((1 << precision) * c) // total for c in cdf]
    for i in range(1, len(cdf)):
        cdf[i] += cdf[i - 1]
    cdf[-1] = 1 << precision
    n = len(cdf) - 1
    for i in range(n):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(n):
                f = cdf[j + 1] - cdf[j]
                if f > 1 and (best_freq is None or f < best_freq):
                    best_freq, best = f, j
            assert best >= 0
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return np.asarray(cdf, dtype=np.int32)


def py_encode_with_indexes(symbols, indexes, cdfs, cdf_sizes, offsets):
    items = []
    for s, c in zip(symbols, indexes):
        cdf = cdfs[c]
        mx = int(cdf_sizes[c]) - 2
        v = int(s) - int(offsets[c])
        raw = 0
        if v < 0:
            raw, v = -2 * v - 1, mx
        elif v >= mx:
            raw, v = 2 * (v - mx), mx
        items.append((int(cdf[v]), int(cdf[v + 1]) - int(cdf[v]), False))
        if v == mx:
            nb = 0
            while (raw >> (4 * nb)) != 0:
                nb += 1
            val = nb
            while val >= 15:
                items.append((15, 0, True))
                val -= 15
            items.append((val, 0, True))
            for j in range(nb):
                items.append(((raw >> (4 * j)) & 15, 0, True))
    x = _L
    words = []
    for start, rng, byp in reversed(items):
        freq = (1 << 12) if byp else rng
        x_max = ((_L >> 16) << 32) * freq
        if x >= x_max:
            words.append(x & 0xFFFFFFFF)
            x >>= 32
        if byp:
            x = (x << 4) | start
        else:
            x = ((x // rng) << 16) + (x % rng) + start
    words.append(x >> 32)
    words.append(x & 0xFFFFFFFF)
    return b"".join(int(w).to_bytes(4, "little") for w in reversed(words))


def py_decode_with_indexes(data, indexes, cdfs, cdf_sizes, offsets):
    words = [int.from_bytes(data[i:i + 4], "little") for i in range(0, len(data), 4)]
    pos = 2
    x = words[0] | (words[1] << 32)
    out = []

    def renorm(x, pos):
        if x < _L:
            x = (x << 32) | words[pos]
            pos += 1
        return x, pos

    for c in indexes:
        cdf = cdfs[c]
        mx = int(cdf_sizes[c]) - 2
        cf = x & 0xFFFF
        s = 0
        while s + 1 < int(cdf_sizes[c]) and int(cdf[s + 1]) <= cf:
            s += 1
        x = (int(cdf[s + 1]) - int(cdf[s])) * (x >> 16) + cf - int(cdf[s])
        x, pos = renorm(x, pos)
        v = s
        if v == mx:
            val = x & 15
            x >>= 4
            x, pos = renorm(x, pos)
            nb = val
            while val == 15:
                val = x & 15
                x >>= 4
                x, pos = renorm(x, pos)
                nb += val
            raw = 0
            for j in range(nb):
                val = x & 15
                x >>= 4
                x, pos = renorm(x, pos)
                raw |= val << (4 * j)
            v = raw >> 1
            v = -v - 1 if (raw & 1) else v + mx
        out.append(v + int(offsets[c]))
    return np.asarray(out, dtype=np.int32)
