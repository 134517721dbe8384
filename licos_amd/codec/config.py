"""Every switch of the chunk codec, read from the environment ONCE (at import) into one object.

Product code reads `config.<field>`; A/B tools and tests set fields on the object (`codec.config.host_split = False`)
instead of patching module constants.  The fields that only exist for A/B measurements say so."""
import os
from dataclasses import dataclass, field


def _flag(name, default):
    v = os.environ.get(name)
    return default if v is None else v != "0"


def _float(name, default):
    v = os.environ.get(name)
    return default if v is None else float(v)


def _int(name, default):
    v = os.environ.get(name)
    return default if v is None else int(v)


@dataclass
class CodecConfig:
    # -- placement of the serial coder (placement.py) ---------------------------------------------------------------------
    host_split: bool = True        # LICOS_HOST_SPLIT=0: the host never takes a SHARE of a call (host-only small calls remain)
    enc_all_host: float = 2.5      # LICOS_ENC_ALL_HOST: compress calls up to this many host capacities stay host-only
    enc_tail: float = 1.0          # LICOS_ENC_TAIL_FRACTION: share of the capacity at the end of a larger compress call
    host_sub: int = 16             # tiles per host thread and sub-chunk (16 threads x 16 = 256 tiles per transfer)
    simple_batch: int = 8          # calls of up to this many tiles skip the sub-chunk pipeline
    prequeue: int = 3              # LICOS_PREQUEUE: host sub-chunks of a large compress queued before the drains
    hyper_share: int = -1          # LICOS_HYPER_SHARE (dev probe): fixed host share of a scale-hyperprior call; -1 = policy
    # nominal coder rates, ns per symbol (device: per lane = per launch; host: per thread)
    dev_ns: dict = field(default_factory=lambda: {"enc": 105.0, "dec": 100.0})  # (round 4: 145 / 115 - before the stream-major symbols and the instruction-counted decoder)
    host_ns: dict = field(default_factory=lambda: {"enc": 1.8, "dec": 4.0})       # LICOS_HOST_ENC_NS / LICOS_HOST_DEC_NS
    expect_ns: dict = field(default_factory=lambda: {"enc": 1.8, "dec": 3.0})     # the coder call alone on a quiet host
    hyper_dev_ns: dict = field(default_factory=lambda: {"enc": 119.0, "dec": 117.0})  # (enc: 159 before the register-ring record encoder of round 5)
    hyper_host_ns: dict = field(default_factory=lambda: {"enc": 3.5, "dec": 4.9})  # LICOS_HYPER_HOST_ENC_NS / _DEC_NS
    hyper_host_coder_ns: dict = field(default_factory=lambda: {"enc": 3.7, "dec": 5.0})
    coder_streams: int = 8         # side streams the hyperprior codec spreads its chunks' coder launches over
    # -- data formats between device and host (factorized.py / hyper.py) --------------------------------------------------
    sym16: bool = True             # LICOS_SYM16=0: the host's tiles cross PCIe as int32 symbols (A/B)
    zero_copy: bool = False        # LICOS_ZERO_COPY=1: quantise / dequantise kernels store to / load from page-locked memory (A/B)
    # -- which device coder (A/B) -----------------------------------------------------------------------------------------
    eb_records: bool = False       # LICOS_EB_RECORDS=1: record encoder for the entropy bottleneck
    eb_image: bool = False         # LICOS_EB_IMAGE=1: the image decoder of rans_gc.hip (round 4) instead of the plane decoder of rans.hip
    eb_stream_major: bool = True   # LICOS_EB_STREAM_MAJOR=0: device plane coder on [position][stream] symbols through the transposing kernels (A/B)

    @classmethod
    def from_env(cls):
        c = cls()
        c.host_split = _flag("LICOS_HOST_SPLIT", c.host_split)
        c.enc_all_host = _float("LICOS_ENC_ALL_HOST", c.enc_all_host)
        c.enc_tail = _float("LICOS_ENC_TAIL_FRACTION", c.enc_tail)
        c.prequeue = _int("LICOS_PREQUEUE", c.prequeue)
        c.hyper_share = _int("LICOS_HYPER_SHARE", c.hyper_share)
        c.host_ns = {"enc": _float("LICOS_HOST_ENC_NS", c.host_ns["enc"]), "dec": _float("LICOS_HOST_DEC_NS", c.host_ns["dec"])}
        c.hyper_host_ns = {"enc": _float("LICOS_HYPER_HOST_ENC_NS", c.hyper_host_ns["enc"]),
                           "dec": _float("LICOS_HYPER_HOST_DEC_NS", c.hyper_host_ns["dec"])}
        c.sym16 = _flag("LICOS_SYM16", c.sym16)
        c.zero_copy = os.environ.get("LICOS_ZERO_COPY", "0") == "1"
        c.eb_records = os.environ.get("LICOS_EB_RECORDS", "0") == "1"
        c.eb_image = _flag("LICOS_EB_IMAGE", c.eb_image)
        c.eb_stream_major = _flag("LICOS_EB_STREAM_MAJOR", c.eb_stream_major)
        return c


config = CodecConfig.from_env()
