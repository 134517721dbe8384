"""Optional measurement hooks of the chunk codec; all None (off) in production.

  trace.timings      dict -> every section boundary synchronises the device and accumulates wall-clock seconds (tools/profile_step.py)
  trace.host_trace   list -> the host loops append per-sub-chunk tuples (tools/tail_probe.py, hyper_probe.py --trace)
  trace.coder_events dict -> each serial coder launch is bracketed by HIP events on ITS stream:
                             "z_encode" | "y_encode" | "z_decode" | "y_decode" -> [(start, end)]  (bench.py)
"""
import time

import torch


class _Trace:
    def __init__(self):
        self.timings = None
        self.host_trace = None
        self.coder_events = None

    def note(self, *entry):
        if self.host_trace is not None:
            self.host_trace.append(entry)

    def stamp(self, name, count=0):
        if self.host_trace is not None:
            self.host_trace.append((name, count, time.perf_counter()))


trace = _Trace()


def timed_coder(key, fn):
    if trace.coder_events is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    trace.coder_events.setdefault(key, []).append((e0, e1))
    return out


class Section:
    def __init__(self):
        self.t = None

    def mark(self, name):
        if trace.timings is None:
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        if self.t is not None:
            trace.timings[name] = trace.timings.get(name, 0.0) + (now - self.t)
        self.t = now
