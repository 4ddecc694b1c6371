"""Side streams of the step: cached for eager launches, NEW for every HIP-graph capture.

Eager steps fork onto streams that are created once per (owner, name, device) and kept.  Inside a capture the same names
resolve to streams that belong to THAT capture alone: created through the C ABI (hipStreamCreateWithFlags, non-blocking;
wrapped as torch.cuda.ExternalStream) ahead of the capture, never handed to another capture while fewer than MAX_LIVE streams
exist (64 captures' worth: the entry point re-captures once per epoch), and never destroyed -- autograd nodes and the caching
allocator's deferred record_stream() events keep raw stream handles long after a capture has ended; destroying a stream
they still name takes the process down in the autograd engine's worker thread (measured: tools/capture_sequence.py with
streams destroyed two generations later died in the warm-up backward of its fifth capture).

Why per capture: every capture-time crash of this build inside the ROCm 7.2 runtime (the third capture of
tools/graph_overlap.py in round 1, the fourth training-graph capture of tools/train_times.py in round 3) had one thing in common
-- side-stream OBJECTS that had taken part in an earlier capture, with another fork / join topology, taking part in a new
one; each capture alone was fine, and so was any number of captures of ONE topology.  A stream that joins a capture through
an event wait carries capture state in the runtime (its capture graph, its last captured nodes, the list of streams forked
from it); a stream that has never been in a capture carries none.  The entry point re-captures its training graph whenever the
memory bank's storage changes (once per epoch, main_retrieval.GraphedStep), so the product walks this path too
(tests/test_entry_gpu.py::test_graphed_step_survives_five_recaptures).
"""
import ctypes

import torch

from . import hip

_EAGER = {}            # (owner id, name, device index) -> torch.cuda.Stream
_CAPTURE = {}          # capture id -> {(owner id, name, device index): ExternalStream}
_RETIRED = []          # raw stream handles of finished captures, oldest first (re-used only past MAX_LIVE)
_SPARE = []            # raw streams created in eager code for the next capture (no stream creation while a capture runs)
STATS = {"created": 0, "reused": 0}
SPARE_TARGET = 64      # the training step forks eight streams; a pipelined capture of U loss-only steps 6 U + U
MAX_LIVE = 768         # streams this module creates at most; beyond it the OLDEST retired capture streams go round again


def _new_raw_stream():
    raw = ctypes.c_void_p(0)
    hip._check("nr_stream_create", hip.lib().nr_stream_create(ctypes.byref(raw)))
    STATS["created"] += 1
    return raw.value


def _retire_old(current_cid):
    """The stream sets of finished captures move to the retired list (kept alive: see the module docstring)."""
    for cid in [c for c in _CAPTURE if c != current_cid]:
        _RETIRED.extend(s.cuda_stream for s in _CAPTURE.pop(cid).values())


def _fresh_raw_stream():
    if STATS["created"] >= MAX_LIVE and _RETIRED:
        if STATS["reused"] == 0:
            import warnings
            warnings.warn(f"neighborretr_amd.streams: {MAX_LIVE} capture streams exist; the oldest retired ones are handed to new "
                          "captures from here on (see the module docstring: the condition earlier runtime crashes had in common)")
        STATS["reused"] += 1
        return _RETIRED.pop(0)
    return _new_raw_stream()


def reserve(n):
    """Tops the spare list up to `n` streams NOW (eager code): a capture that forks many streams -- a pipelined capture of U
    loss-only steps forks ~7 U -- must not create streams while it is open.  Raises inside a capture."""
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("streams.reserve() inside a capture: reserve the capture's side streams before opening it")
    while len(_SPARE) < int(n):
        _SPARE.append(_fresh_raw_stream())


def side(owner, name, device):
    """The side stream `name` of `owner` on `device` for the code that is running now: eager -> the cached one; inside a
    capture -> the one of this capture."""
    device = torch.device(device)
    key = (id(owner), name, device.index if device.index is not None else torch.cuda.current_device())
    cid = hip.stream_capture_id() if torch.cuda.is_current_stream_capturing() else 0
    if _CAPTURE and (cid == 0 or any(c != cid for c in _CAPTURE)):
        _retire_old(cid)
    if cid == 0:
        st = _EAGER.get(key)
        if st is None:
            st = _EAGER[key] = torch.cuda.Stream(device=device)
        while len(_SPARE) < SPARE_TARGET:       # (the warm-up steps in front of every capture pass through here)
            _SPARE.append(_fresh_raw_stream())
        return st
    per = _CAPTURE.setdefault(cid, {})
    st = per.get(key)
    if st is None:
        st = per[key] = torch.cuda.ExternalStream(_SPARE.pop() if _SPARE else _fresh_raw_stream(), device=device)
    return st


def forget(owner):
    """Drops the cached eager streams of `owner` (a model that is being discarded)."""
    for key in [k for k in _EAGER if k[0] == id(owner)]:
        del _EAGER[key]
