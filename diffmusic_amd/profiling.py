"""Stage ranges of the hot loop: roctx markers (visible to `rocprofv3 --marker-trace`) and, on request, HIP-event stage times.

Disabled by default (one attribute test per stage).  `enable(markers=True)` or DMX_ROCTX=1 pushes / pops a roctx range per
stage (U-Net, VAE forward / backward, HiFi-GAN forward / backward, operator + transform + loss, scheduler update);
`enable(events=True)` additionally brackets every stage with HIP events on torch's current stream so that
`stage_ms()` returns the mean device time per stage (bench.py's per-stage report)."""
import contextlib
import os
from collections import OrderedDict

import torch

_markers = os.environ.get("DMX_ROCTX", "") not in ("", "0")
_events = False
_recs = OrderedDict()          # name -> list of (start_event, end_event)


def enable(markers=None, events=None):
    global _markers, _events
    if markers is not None:
        _markers = bool(markers)
    if events is not None:
        _events = bool(events)
        if _events:
            _recs.clear()


def active():
    return _markers or _events


@contextlib.contextmanager
def _stage(name):
    a = b = None
    if _markers:
        torch.cuda.nvtx.range_push(name)             # roctxRangePush on ROCm builds of torch
    if _events:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    try:
        yield
    finally:
        if _events:
            b.record()
            _recs.setdefault(name, []).append((a, b))
        if _markers:
            torch.cuda.nvtx.range_pop()


_NULL = contextlib.nullcontext()


def stage(name):
    """`with stage("vae_fwd"): ...` -- a null context unless markers or events are enabled."""
    return _stage(name) if (_markers or _events) else _NULL


def stage_ms():
    """Mean device milliseconds per stage over the recorded calls (synchronises)."""
    torch.cuda.synchronize()
    return OrderedDict((k, sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in _recs.items() if v)


def stage_calls():
    return {k: len(v) for k, v in _recs.items()}
