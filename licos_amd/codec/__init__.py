"""Chunk-pipelined encode / decode of a tile batch (the reference's `net.compress` / `net.decompress`,
/root/reference/eval_utils.py:199-204), split by concern:

  config.py      every LICOS_* switch of the codec, read once into `config`
  placement.py   which tiles the host cores code (pure policy, CPU-tested): host_share, hyper_host_share, HostRate
  staging.py     side streams, page-locked staging per (device, role), PackedStrings, the per-device call lock
  trace.py       optional measurement hooks (`trace.timings`, `trace.host_trace`, `trace.coder_events`)
  factorized.py  FactorizedPrior pipelines: compress_chunked / decompress_chunked
  hyper.py       ScaleHyperprior pipelines: compress_hyper / decompress_hyper
"""
from . import placement
from .config import CodecConfig, config
from .factorized import compress_chunked, decompress_chunked
from .hyper import compress_hyper, decompress_hyper
from .placement import (host_capacity, host_share, hyper_fast_path, hyper_host_share, hyper_retry_chunk, note_host_rate,
                        ramp)
from .staging import PackedStrings
from .trace import trace

__all__ = ["CodecConfig", "config", "placement", "trace", "PackedStrings", "compress_chunked", "decompress_chunked",
           "compress_hyper", "decompress_hyper", "host_capacity", "host_share", "hyper_fast_path", "hyper_host_share",
           "hyper_retry_chunk", "note_host_rate", "ramp"]
