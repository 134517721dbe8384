"""What the pipelines share besides policy: side streams, page-locked staging areas, the PackedStrings list.

Staging areas are reused from call to call (page-locked allocations cost milliseconds) and keyed by (device, role); the
pipelines of one device are serialised by `exclusive` - two threads, or two models, coding on the same device take turns
instead of overwriting each other's symbols.  Different devices never share a buffer."""
import functools
import threading

import torch

_streams = {}
_pinned = {}
_locks = {}
_locks_guard = threading.Lock()


def stream(device, role):
    key = (device.type, device.index, role)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=device)
    return _streams[key]


def _dev_key(device):
    if device is None:
        return -1
    return device.index if device.index is not None else torch.cuda.current_device()


def pinned_i32(device, role, rows, cols):
    """A reusable page-locked int32 [rows, cols] staging area per (device, role)."""
    need = rows * cols
    key = (_dev_key(device), role)
    buf = _pinned.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 1 << 18) * 5 // 4, dtype=torch.int32, pin_memory=True)
        _pinned[key] = buf
    return buf[:need].view(rows, cols)


def pinned_i16(device, role, rows, cols):
    """A reusable page-locked int16 [rows, cols] staging area (contiguous whatever the parity of cols)."""
    return pinned_i32(device, role, 1, (rows * cols + 1) // 2).view(torch.int16)[0, :rows * cols].view(rows, cols)


def exclusive(device_of):
    """Decorator: one pipeline call at a time per device (`device_of(*args)` names it).  The reference drives a model
    from one thread; this makes the other case slow instead of wrong."""
    def wrap(fn):
        @functools.wraps(fn)
        def inner(*args, **kwargs):
            key = _dev_key(device_of(*args, **kwargs))
            with _locks_guard:
                lock = _locks.setdefault(key, threading.RLock())
            with lock:
                return fn(*args, **kwargs)
        return inner
    return wrap


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


def split_bytes(mv, off):
    """The byte strings mv[off[i]:off[i+1]] (offsets as Python ints: indexing a memoryview with numpy scalars costs 2.5 x
    the time - 4096 strings are 1 - 2 ms of this thread, and the last chunk's are at the exposed end of compress)."""
    o = off.tolist()
    return [bytes(mv[a:b]) for a, b in zip(o[:-1], o[1:])]


class HostRange(Exception):
    """A symbol of a host tile does not fit the 16 bits of its compact PCIe form."""
