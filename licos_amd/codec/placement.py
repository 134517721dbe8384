"""Placement policy of the serial coder: which tiles of a compress / decompress call the host cores code, in which
sub-chunks.  Pure functions over `config` and the measured host rate - no device work, tested on the CPU (tests/test_host.py).

A device coder launch lasts n_symbols x (latency of one rANS step) whatever the number of streams: ~145 ns per symbol
encoding, ~115 ns decoding (7.1 / 5.6 ms for a 256^2 tile's 49 152 symbols).  Inside a long call that is hidden under
the neighbouring chunks' transforms except ONCE per call - the last chunk's encode, the first chunk's decode - and it is
most of a call of a thousand tiles.  The host cores run the same coder (csrc/host_rans.cpp, bit-identical streams) at
~1.8 / ~4 ns per symbol and thread: in the time of one device launch T threads code  T x (device ns) / (host ns)  tiles
(16 threads: ~1100 encoding, ~390 decoding: the decode figure is what the sub-chunk pipeline sustains - 256 tiles in 2.35 ms,
32 in 0.5 ms - not the coder alone).  So the tiles at the exposed end of a call go to the host, in sub-chunks that
pipeline with their transforms, and the device launch they run beside covers the rest.
Reference call this serves: /root/reference/eval_utils.py:201 (`net.compress` on whatever runs the model)."""
from .. import ops
from .config import config


def chunks(total, size):
    return [(s, min(size, total - s)) for s in range(0, total, size)]


def ramp(total, first, size):
    """Sub-chunks that double from `first` up to `size`: the host starts coding (or the synthesis transform starts) after
    a few tiles' worth of work instead of a full sub-chunk's, and the later sub-chunks keep the transfers large."""
    out, s0, m = [], 0, max(1, min(first, size))
    while s0 < total:
        n = min(m, total - s0)
        out.append((s0, n))
        s0 += n
        m = min(size, 2 * m)
    return out


class HostRate:
    """What the host coder actually delivered in this process's recent calls, as a factor on the nominal ns per symbol
    (1 = nominal, never below).  The host is shared on these boxes: a host whose cores are busy with other tenants' work
    codes 5 - 8 x slower, and a share sized for a quiet host would then be the slowest part of the call.

    A sub-chunk of >= 4 tiles per thread moves the factor either way (exponential average).  A smaller one - down to one
    tile per thread - can only LOWER it: its per-call overheads make a small sample read slow, never fast, so "fast" is
    evidence and "slow" is not.  Without that a busy phase could lock the share small for good: once the capacity is below
    ~6 tiles per thread every sub-chunk `ramp` hands out is a small one, and no sample would ever qualify again."""

    def __init__(self):
        self.factor = {"enc": 1.0, "dec": 1.0}
        self.cap_state = {}
        self.hyper_share = {}

    def reset(self):
        self.__init__()

    def note(self, direction, tiles, nsym, seconds, expect_ns=None):
        threads = ops.host_threads()
        if tiles < threads or seconds <= 0:
            return
        ns = 1e9 * seconds * threads / (tiles * nsym)
        f = max(1.0, ns / (config.expect_ns[direction] if expect_ns is None else expect_ns))
        if tiles >= 4 * threads:
            self.factor[direction] = 0.5 * self.factor[direction] + 0.5 * f
        elif f < self.factor[direction]:
            w = 0.5 * tiles / (4.0 * threads)
            self.factor[direction] = (1.0 - w) * self.factor[direction] + w * f


rate = HostRate()


def note_host_rate(direction, tiles, nsym, seconds, expect_ns=None):
    rate.note(direction, tiles, nsym, seconds, expect_ns)


def host_capacity(direction):
    """Tiles the host cores code in the time of ONE device coder launch (independent of the stream length), on a grid of
    2 tiles per thread and with one grid step of hysteresis: the measured host rate moves a little with every call, and
    a share that moved with it would give every call its own tensor shapes (3 751, 3 775, 3 747 ... tiles in a piece) -
    work for the caching allocator, and now and then a hipMalloc in the middle of a step."""
    threads = ops.host_threads()
    raw = 0.85 * threads * config.dev_ns[direction] / (config.host_ns[direction] * rate.factor[direction])
    grid = 2 * threads
    key = (direction, threads)
    last = rate.cap_state.get(key)
    if last is None or abs(raw - (last + 0.5 * grid)) >= grid:
        last = rate.cap_state[key] = max(0, int(raw) // grid * grid if raw >= grid else int(raw))
    return last


def host_share(batch, direction):
    """How many tiles of a call of `batch` tiles the host codes.  Everything up to the host's capacity for the direction
    (a thousand tiles encoding, four hundred decoding at 16 threads: the device coder's launch latency alone is longer
    than the host takes).  Of a larger call, the tiles at its exposed end: the first `capacity` tiles of a decode - the
    synthesis transform starts on them while the first device launch runs - and the last `capacity` tiles of an encode
    (what the host pipeline finishes beside the last 7-ms device launch and that launch's drain; `enc_tail` scales it).
    `capacity` follows the rate the host coder delivered in this process's recent calls (HostRate)."""
    if ops.HOST_CODER == "0":
        return 0
    if ops.HOST_CODER == "1" or ops.host_coder_preferred(batch):
        return batch
    if not config.host_split:
        return 0
    cap = host_capacity(direction)
    if batch <= cap or (direction == "enc" and batch <= config.enc_all_host * cap):
        return batch  # (a compress call a little over the capacity: the host finishing late costs less than a device chunk's drain)
    return cap if direction == "dec" else int(config.enc_tail * cap)


def host_subchunks(n_host):
    """The host's tiles of a factorized call as ramped sub-chunks [(first, count)]."""
    threads = ops.host_threads()
    return ramp(n_host, 2 * threads, max(1, config.host_sub * threads)) if n_host else []


def hyper_fast_path(net, batch):
    """The chunk-pipelined scale-hyperprior codec applies: a decoder image that fits LDS, and either device coder
    placement (large calls) or a host-coded call of at least one full sub-chunk (4 tiles per host thread), which runs
    the same pipeline with every tile in the host's share - transforms, PCIe and host coding overlap sub-chunk by
    sub-chunk instead of following one another as in the plain module path."""
    if net.gaussian_conditional.coder_image() is None:
        return False
    if not ops.host_coder_preferred(batch):
        return True
    return config.host_split and batch >= 4 * ops.host_threads()


def hyper_retry_chunk(chunk, ny):
    """Chunk size of the worst-case-capacity retry (cap_words = 2 ny + 8): licos_rans_encode_records addresses its word
    sink with 32-bit byte offsets, (cap + 1) * streams * 4 < 2^32 - an incompressible batch of M = 320 latents at 512^2
    would not fit at the default 2048 tiles per chunk."""
    cap = 2 * ny + 8
    return max(1, min(chunk, ((1 << 32) - 1) // (4 * (cap + 1))))


def hyper_host_share(batch, direction="enc"):
    """Tiles at the END of a compress_hyper / decompress_hyper call that the host codes (y and z streams): as many as the
    host threads code during the ONE device launch of the y coder that is exposed per call (31 ms encoding, 23 ms
    decoding a 512^2 tile's 196 608 symbols) - their transforms then run beside that launch instead of in front of /
    behind it.  A scale-hyperprior tile is 37 us of transforms on the encode side and the host codes it in 59 us
    (16 threads), so the host keeps up with the device for the length of that launch.  The count moves in steps of
    4 x threads and only when the measured host rate has moved it by a whole step (a change of the device chunks' sizes
    costs the caching allocator a round of hipMalloc, 30 ms)."""
    if ops.HOST_CODER == "0" or not config.host_split:
        return 0
    if ops.host_coder_preferred(batch):  # a mid-size call (or LICOS_HOST_CODER=1): every tile, no device coder launch at all
        return batch
    if config.hyper_share >= 0:  # dev probe
        return min(config.hyper_share, batch // 3)
    threads = ops.host_threads()
    step = 4 * threads
    cap = 0.85 * threads * config.hyper_dev_ns[direction] / (config.hyper_host_ns[direction] * rate.factor[direction])
    last = rate.hyper_share.get(direction)
    if last is None or abs(cap - last) >= step:
        last = rate.hyper_share[direction] = int(cap) // step * step
    return max(0, min(last, batch // 3 // step * step))


def hyper_subchunks(n_host):
    threads = ops.host_threads()
    return ramp(n_host, threads, max(1, 4 * threads)) if n_host else []


def describe():
    """What the record needs to separate the MI355X from its host (bench.py `host`)."""
    import os
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    return {"threads": ops.host_threads(), "cores_visible": visible, "host_coder": ops.HOST_CODER, "host_split": config.host_split,
            "enc_share": host_share(1 << 20, "enc"), "dec_share": host_share(1 << 20, "dec"),
            "enc_all_host_up_to": int(config.enc_all_host * host_capacity("enc")) if config.host_split and ops.HOST_CODER != "0" else 0,
            "hyper_enc_share": hyper_host_share(1 << 20, "enc"), "hyper_dec_share": hyper_host_share(1 << 20, "dec"),
            "host_factor": {k: round(v, 3) for k, v in rate.factor.items()}}
