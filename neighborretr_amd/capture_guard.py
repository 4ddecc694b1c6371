"""Stream-topology rules of HIP-graph capture on the ROCm 7.2 runtime, enforced in code.

Two capture shapes take the process down with SIGSEGV inside the runtime at capture time (plain-torch reproducers:
tools/capture_refork.py, tools/capture_nest.py; records: profiles/r02_capture_refork.txt, profiles/r02_capture_nest.txt):

  1. re-fork of a joined stream:  A.wait(B) (B joined into A, A not the capture's origin), later B.wait(A) + more work on B;
  2. parent joins its own child:  C.wait(P) (C forked from P, P itself a forked stream), later P.wait(C).

Both are the same thing seen from the dependency graph: a NON-ORIGIN stream X waits on a stream Y that already depends on
X.  Joins into the capture's origin stream are always fine, so are forks from it (autograd's backward re-enters a side
stream that was joined into the origin: captured and replayed in every training graph), sibling joins, and a fresh stream
in place of the re-used one.  `StreamTopology` keeps, per capture, what every stream depends on (as of its last wait;
dependencies reaching a stream THROUGH the origin are not carried) and raises `CaptureTopologyError` on such a wait
BEFORE the runtime sees it.  `wait_stream` / `record_event` / `wait_event` below are what neighborretr_amd.head uses in
place of the torch calls; outside a capture they are plain pass-throughs.

The pure part (`StreamTopology`) works on integer stream handles and is unit-tested on the CPU.
"""
import torch


class CaptureTopologyError(RuntimeError):
    pass


class StreamTopology:
    """Dependency bookkeeping of ONE capture.  Streams are any hashable handles; `origin` = the stream the capture began on."""

    def __init__(self, origin):
        self.origin = origin
        self.deps = {}            # stream -> set of streams it depends on (excluding what only reaches it through the origin)
        self.parent = {}          # stream -> the stream it was forked from (= the first stream it waited on)
        self.edges = 0

    def wait(self, waiter, on):
        """`waiter` is about to wait on `on` (wait_stream, or wait_event of an event recorded on `on`)."""
        self.edges += 1
        if waiter == on:
            return
        if waiter != self.origin:
            if waiter in self.deps.get(on, ()):
                p, forked_from = self.parent.get(on), False
                while p is not None and not forked_from:
                    forked_from, p = p == waiter, self.parent.get(p)
                shape = ("a forked stream waits on a stream forked from itself (parent joins child on a non-origin stream)"
                         if forked_from else
                         "a stream that was already joined is forked again from the stream that joined it")
                raise CaptureTopologyError(
                    f"capture topology refused: stream {waiter!r} would wait on {on!r}, which already depends on it -- {shape}. "
                    "The ROCm 7.2 runtime segfaults on this shape at capture time (tools/capture_refork.py, "
                    "tools/capture_nest.py); use a fresh side stream, or join into the capture's origin stream instead.")
            self.parent.setdefault(waiter, on)
            carried = set() if on == self.origin else set(self.deps.get(on, ()))
            carried.add(on)
            carried.discard(waiter)
            self.deps.setdefault(waiter, set()).update(carried)
        # the origin may wait on anything; nothing is carried through it


_ACTIVE = {}          # device index -> (capture id, StreamTopology)
_EVENT_STREAM = {}    # id(event) -> (capture id, stream handle)
STATS = {"captures": 0, "edges": 0}


def _topology(stream):
    """The capture's topology when `stream` is capturing, else None."""
    if not torch.cuda.is_current_stream_capturing():
        if _ACTIVE:
            _ACTIVE.clear()
            _EVENT_STREAM.clear()
        return None
    from . import hip
    cid = hip.stream_capture_id(torch.cuda.current_stream())
    if cid == 0:
        return None
    dev = stream.device.index
    hit = _ACTIVE.get(dev)
    if hit is None or hit[0] != cid:
        # first guarded call of this capture: the current stream is taken to be its origin (the step is entered on the
        # stream torch.cuda.graph() captures on)
        hit = _ACTIVE[dev] = (cid, StreamTopology(torch.cuda.current_stream().cuda_stream))
        _EVENT_STREAM.clear()
        STATS["captures"] += 1
    return hit


TRACE = bool(int(__import__("os").environ.get("NR_GUARD_TRACE", "0")))     # developer hook: print every guarded edge


def wait_stream(waiter, on):
    """waiter.wait_stream(on), checked against the capture rules first."""
    hit = _topology(waiter)
    if hit is not None:
        if TRACE:
            print(f"[capture_guard] {waiter.cuda_stream:#x} waits on {on.cuda_stream:#x} (origin {hit[1].origin:#x})", file=__import__("sys").stderr, flush=True)
        hit[1].wait(waiter.cuda_stream, on.cuda_stream)
        STATS["edges"] += 1
    waiter.wait_stream(on)


def may_fork(stream):
    """False when `stream` is being captured and is NOT the capture's origin: a side stream forked from it would have to be
    joined back into a non-origin stream (shape 2 above).  True outside a capture."""
    hit = _topology(stream)
    return hit is None or hit[1].origin == stream.cuda_stream


def record_event(stream=None):
    """A new event recorded on `stream` (default: current); remembered so that wait_event knows where it came from."""
    st = torch.cuda.current_stream() if stream is None else stream
    ev = torch.cuda.Event()
    ev.record(st)
    hit = _topology(st)
    if hit is not None:
        _EVENT_STREAM[id(ev)] = (hit[0], st.cuda_stream, ev)
    return ev


def wait_event(waiter, ev):
    hit = _topology(waiter)
    if hit is not None:
        src = _EVENT_STREAM.get(id(ev))
        if src is not None and src[0] == hit[0]:
            hit[1].wait(waiter.cuda_stream, src[1])
            STATS["edges"] += 1
    waiter.wait_event(ev)
