"""Side streams of the step: cached for eager launches, NEW for every HIP-graph capture.

Eager steps fork onto streams that are created once per (owner, name, device) and kept.  Inside a capture the same names
resolve to streams that belong to THAT capture alone: created through the C ABI (hipStreamCreateWithFlags, non-blocking;
wrapped as torch.cuda.ExternalStream) at their first use under the capture's id, never handed to another capture, and
destroyed later from eager code, once two younger generations exist.

Why: every capture-time crash of this build inside the ROCm 7.2 runtime (the third capture of
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
_RETIRED = []          # [(generation, [raw stream handles])]
_generation = 0
_SPARE = []            # raw streams created in eager code for the next capture (no stream creation while a capture runs)
STATS = {"created": 0, "destroyed": 0}
KEEP_GENERATIONS = 2   # capture stream sets younger than this many generations are left alone
SPARE_TARGET = 12      # the training step forks eight streams


def _new_raw_stream():
    raw = ctypes.c_void_p(0)
    hip._check("nr_stream_create", hip.lib().nr_stream_create(ctypes.byref(raw)))
    STATS["created"] += 1
    return raw.value


def _retire_old(current_cid):
    """Moves the stream sets of finished captures to the retired list; from EAGER code also destroys the ones that are at
    least KEEP_GENERATIONS captures old (by then the allocator has long processed the record_stream() events it defers
    until no capture is under way -- they are recorded on these streams)."""
    global _generation
    for cid in [c for c in _CAPTURE if c != current_cid]:
        _generation += 1
        _RETIRED.append((_generation, [s.cuda_stream for s in _CAPTURE.pop(cid).values()]))
    if current_cid == 0:
        while _RETIRED and _RETIRED[0][0] <= _generation - KEEP_GENERATIONS:
            for raw in _RETIRED.pop(0)[1]:
                hip._check("nr_stream_destroy", hip.lib().nr_stream_destroy(ctypes.c_void_p(raw)))
                STATS["destroyed"] += 1


def side(owner, name, device):
    """The side stream `name` of `owner` on `device` for the code that is running now: eager -> the cached one; inside a
    capture -> the one of this capture."""
    device = torch.device(device)
    key = (id(owner), name, device.index if device.index is not None else torch.cuda.current_device())
    cid = hip.stream_capture_id() if torch.cuda.is_current_stream_capturing() else 0
    if _CAPTURE and (cid == 0 or any(c != cid for c in _CAPTURE)):
        _retire_old(cid)
    elif cid == 0 and _RETIRED:
        _retire_old(0)
    if cid == 0:
        st = _EAGER.get(key)
        if st is None:
            st = _EAGER[key] = torch.cuda.Stream(device=device)
        while len(_SPARE) < SPARE_TARGET:       # (the warm-up steps in front of every capture pass through here)
            _SPARE.append(_new_raw_stream())
        return st
    per = _CAPTURE.setdefault(cid, {})
    st = per.get(key)
    if st is None:
        st = per[key] = torch.cuda.ExternalStream(_SPARE.pop() if _SPARE else _new_raw_stream(), device=device)
    return st


def forget(owner):
    """Drops the cached eager streams of `owner` (a model that is being discarded)."""
    for key in [k for k in _EAGER if k[0] == id(owner)]:
        del _EAGER[key]
