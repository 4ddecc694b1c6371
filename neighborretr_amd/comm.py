"""The collectives of the sharded step behind ONE seam.

Every collective of the path -- the packed all-gather of the exchange step (modeling.py:274-280 via until_module.py:367-388),
the clustering's batch-wide maximum (cluster.py:473-475), the gathers of global tokens / centrality slices, the row-term
all-reduce, and the reductions of their backward (the AllGather2 pattern, until_module.py:391-412) -- is issued through the
functions below instead of `torch.distributed` directly.  By default they ARE torch.distributed on the default process group
("nccl" = RCCL over xGMI, one GPU per rank; "gloo" only to rehearse ranks that share one card).  Two other communicators
can be put in their place for the duration of a `with use(...)` block:

* `EmulatedWorld(W).comm(r)`: rank r of a W-rank job on ONE GPU.  What the other ranks would contribute is held in
  pre-filled buffers (obtained by running all W ranks' steps in turn until every collective's inputs have settled); each
  collective writes those into place and moves THIS rank's real message through a 1-rank RCCL communicator, so the launch
  sequence, message sizes and rank-local kernel sizes are those of the W-rank job (the wire time of the peers is not).  This
  is how the per-rank step at W = 2 / 4 / 8 is measured on the one GPU a box has (tools/rank_local_times.py).
* `SegmentedStep`: a step that contains collectives as HIP graphs of the rank-local SEGMENTS between them -- the fallback of
  `bench.py --gpus N` / the training entry when a whole-step capture (collectives inside the graph) is refused or fails
  its validation: n+1 graph replays and n eager collectives instead of ~60 eager launches.
"""
import contextlib

import torch
import torch.distributed as dist

_OPS = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}


class TorchComm:
    """torch.distributed on `group` (None = the default process group)."""

    def __init__(self, group=None):
        self.group = group

    @property
    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def backend(self):
        return dist.get_backend(self.group) if dist.is_initialized() else "none"

    def begin_step(self):
        pass

    def all_gather_into_tensor(self, out, inp):
        dist.all_gather_into_tensor(out, inp, group=self.group)

    def all_reduce(self, t, op="sum"):
        dist.all_reduce(t, op=_OPS[op], group=self.group)

    def reduce_scatter_tensor(self, out, inp):
        dist.reduce_scatter_tensor(out, inp, group=self.group)


_DEFAULT = TorchComm()
_active = None


def current():
    return _active if _active is not None else _DEFAULT


@contextlib.contextmanager
def use(comm):
    """Routes the path's collectives through `comm` inside the block (one Python thread per rank drives the path)."""
    global _active
    old, _active = _active, comm
    try:
        yield comm
    finally:
        _active = old


def get_rank():
    return current().rank


def get_world_size():
    return current().world


def backend():
    return current().backend()


def begin_step():
    """Called at the top of every step (modeling.NeighborRetr.forward): communicators that number their collectives restart."""
    current().begin_step()


def all_gather_into_tensor(out, inp):
    current().all_gather_into_tensor(out, inp)


def all_reduce(t, op="sum"):
    current().all_reduce(t, op)


def reduce_scatter_tensor(out, inp):
    current().reduce_scatter_tensor(out, inp)


# ------------------------------------------------------------------------------------------------------------------------------
class EmulatedWorld:
    """W ranks of the sharded step on one GPU, one at a time.  `group`: a 1-rank process group (RCCL) that carries this rank's
    own message of every collective, or None (no process group: the rank's message is a device copy -- CPU-side tests of the
    bookkeeping).  Collectives are matched by their position in the step (every rank issues the same sequence: the
    property the real job relies on too)."""

    def __init__(self, world, group=None, real_collectives=True):
        self.world = int(world)
        self.group = group
        self.real = bool(real_collectives) and dist.is_initialized()
        self.contrib = {}        # sequence number -> [W, ...] what every rank puts into that collective
        self.kind = {}           # sequence number -> ("gather" | "sum" | "max" | "min" | "scatter")
        self.others = {}         # (sequence number, rank) -> the other ranks' combined part of a reduction (frozen form)
        self.recording = True
        self.changed = False

    def comm(self, rank):
        return EmulatedComm(self, int(rank))

    def settle(self, run_rank, max_sweeps=12):
        """run_rank(r): runs rank r's step under `use(self.comm(r))`.  Sweeps over the ranks until no collective's
        contributions change any more (collective k's inputs depend only on the results of collectives before it, so this
        takes at most one sweep per collective of the step), then freezes the peers' data for timed / captured runs.
        Returns the number of sweeps."""
        self.recording = True
        for sweep in range(1, max_sweeps + 1):
            self.changed = False
            for r in range(self.world):
                run_rank(r)
            if not self.changed:
                self.freeze()
                return sweep
        raise RuntimeError(f"the emulated world did not settle in {max_sweeps} sweeps (a collective whose input depends on "
                           "its own result?)")

    def freeze(self):
        self.recording = False
        self.others = {}
        for seq, kind in self.kind.items():
            if kind == "gather":
                continue
            full = self.contrib[seq]
            for r in range(self.world):
                rest = torch.cat((full[:r], full[r + 1:]))
                if kind in ("sum", "scatter"):
                    o = rest.sum(0)
                else:
                    o = rest.max(0).values if kind == "max" else rest.min(0).values
                if kind == "scatter":
                    o = o.view(self.world, -1)[r].clone()
                self.others[(seq, r)] = o.contiguous()

    def _store(self, seq, kind, rank, value):
        full = self.contrib.get(seq)
        v = value.detach().reshape(-1)
        if full is None or full.shape[1] != v.numel() or full.dtype != v.dtype:
            fill = 0
            if kind == "max":
                fill = float("-inf") if v.dtype.is_floating_point else torch.iinfo(v.dtype).min
            elif kind == "min":
                fill = float("inf") if v.dtype.is_floating_point else torch.iinfo(v.dtype).max
            full = self.contrib[seq] = torch.full((self.world, v.numel()), fill, dtype=v.dtype, device=v.device)
            self.changed = True
        if self.kind.setdefault(seq, kind) != kind:
            raise RuntimeError(f"collective #{seq}: {kind} here, {self.kind[seq]} on another rank / in an earlier sweep -- the ranks' "
                               "collective sequences differ")
        if not torch.equal(full[rank].view(torch.uint8), v.contiguous().view(torch.uint8)):       # bitwise (NaN-safe); a host sync: recording sweeps only
            full[rank].copy_(v)
            self.changed = True
        return full


class EmulatedComm:
    def __init__(self, world, rank):
        self.w, self.rank, self.world = world, rank, world.world
        self.seq = 0
        self.n_collectives = 0

    def backend(self):
        return "emulated"

    def begin_step(self):
        self.seq = 0

    def _next(self):
        s = self.seq
        self.seq += 1
        self.n_collectives = max(self.n_collectives, self.seq)
        return s

    def all_gather_into_tensor(self, out, inp):
        seq, W, r = self._next(), self.world, self.rank
        inp = inp.contiguous()
        if self.w.recording:
            full = self.w._store(seq, "gather", r, inp)
        else:
            full = self.w.contrib[seq]
        flat = out.view(-1)
        flat.copy_(full.view(-1))                  # stands in for the peers' writes into this rank's gathered buffer
        slot = flat.view(W, -1)[r]
        if self.w.real:
            dist.all_gather_into_tensor(slot, inp.view(-1), group=self.w.group)       # this rank's own message, through RCCL
        else:
            slot.copy_(inp.view(-1))

    def all_reduce(self, t, op="sum"):
        seq, r = self._next(), self.rank
        if self.w.recording:
            full = self.w._store(seq, op, r, t)
            rest = torch.cat((full[:r], full[r + 1:]))
            other = rest.sum(0) if op == "sum" else (rest.max(0).values if op == "max" else rest.min(0).values)
        else:
            other = self.w.others[(seq, r)]
        if self.w.real:
            dist.all_reduce(t, op=_OPS[op], group=self.w.group)
        other = other.view(t.shape)
        if op == "sum":
            t.add_(other)
        elif op == "max":
            torch.maximum(t, other, out=t)
        else:
            torch.minimum(t, other, out=t)

    def reduce_scatter_tensor(self, out, inp):
        seq, W, r = self._next(), self.world, self.rank
        inp = inp.contiguous()
        if self.w.recording:
            full = self.w._store(seq, "scatter", r, inp)
            rest = torch.cat((full[:r], full[r + 1:]))
            other = rest.sum(0).view(W, -1)[r]
        else:
            other = self.w.others[(seq, r)]
        mine = inp.view(W, -1)[r]
        if self.w.real:
            dist.reduce_scatter_tensor(out.view(-1), mine, group=self.w.group)
        else:
            out.view(-1).copy_(mine)
        out.view(-1).add_(other)


# ------------------------------------------------------------------------------------------------------------------------------
class _SegmentingComm:
    """Used by SegmentedStep.capture(): every collective ENDS the running capture, is noted for eager launch at replay, and
    the next segment's capture begins behind it."""

    def __init__(self, inner, owner):
        self.inner, self.owner = inner, owner
        self.rank, self.world = inner.rank, inner.world

    def backend(self):
        return self.inner.backend()

    def begin_step(self):
        self.inner.begin_step()

    # (the noted calls name the inner communicator only: a closure over `self` would tie the SegmentedStep into a reference
    # cycle, and its graphs would then be destroyed whenever the garbage collector runs -- possibly inside a later capture)
    def all_gather_into_tensor(self, out, inp):
        inner = self.inner
        self.owner._cut(lambda: inner.all_gather_into_tensor(out, inp))

    def all_reduce(self, t, op="sum"):
        inner = self.inner
        self.owner._cut(lambda: inner.all_reduce(t, op))

    def reduce_scatter_tensor(self, out, inp):
        inner = self.inner
        self.owner._cut(lambda: inner.reduce_scatter_tensor(out, inp))


class SegmentedStep:
    """`fn()` -- a step whose collectives go through this module -- captured as the HIP graphs of its rank-local segments.
    `fn` must issue its collectives through comm.* under whatever communicator is current (it must not install its own).
    replay() = graph 0, collective 0 (eager, on the same stream), graph 1, ..., graph n.  All graphs share one memory pool, so
    a tensor produced in one segment (or filled by a collective) is where the later segments read it.  The step must keep its
    work on the capture's stream across a collective (a side stream still unjoined at a collective cannot be cut: the model
    runs the sharded step on one stream for this form, `model.use_side_streams = False`).  Nothing here synchronises the host:
    with RCCL the eager collectives are ordered with the replays by the stream; gloo collectives block the host themselves."""

    def __init__(self, fn, comm=None, capture_error_mode="thread_local"):
        """capture_error_mode "relaxed": for a step whose collectives are also issued from ANOTHER thread -- the backward of a
        training step runs in the autograd engine's worker thread, and a capture begun in thread-local mode may only be ended by
        the thread that began it."""
        self.fn = fn
        self.comm = comm
        self.mode = capture_error_mode
        self.graphs, self.collectives = [], []
        self.stream = torch.cuda.Stream()
        self.pool = torch.cuda.graph_pool_handle()
        self._open = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        g.capture_begin(pool=self.pool, capture_error_mode=self.mode)
        self._open = g

    def _cut(self, thunk):
        self._open.capture_end()
        self.graphs.append(self._open)
        self._open = None
        self.collectives.append(thunk)
        self._begin()

    def capture(self):
        import gc
        inner = self._inner = self.comm if self.comm is not None else current()
        # No garbage collection while a capture is open: a collected CUDAGraph of an earlier form would be destroyed in the
        # middle of this capture (hipGraphExecDestroy: "operation not permitted when stream is capturing", process aborted --
        # measured with tools/rank_local_times.py).  Collect first, like torch.cuda.graph() does, then hold the collector off.
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        torch.cuda.synchronize()
        self.stream.wait_stream(torch.cuda.current_stream())
        ok = False
        try:
            self._capture_segments(inner)
        finally:
            if gc_was_on:
                gc.enable()
        torch.cuda.current_stream().wait_stream(self.stream)
        return self

    def _capture_segments(self, inner):
        ok = False
        with torch.cuda.stream(self.stream):
            self._begin()
            try:
                with use(_SegmentingComm(inner, self)):
                    self.result = self.fn()
                ok = True
            finally:
                if self._open is not None:
                    try:
                        self._open.capture_end()
                    except Exception:
                        if ok:
                            raise
                    if ok:
                        self.graphs.append(self._open)
                    self._open = None

    @property
    def n_segments(self):
        return len(self.graphs)

    def replay(self):
        """On the CURRENT stream (a captured graph replays on any stream; hopping to the capture stream and back costs two
        cross-stream event waits per step, ~100 us measured)."""
        self._inner.begin_step()                 # (communicators that number their collectives: the step starts over)
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.collectives):
                self.collectives[i]()


# ------------------------------------------------------------------------------------------------------------------------------
def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class CollectiveCapture:
    """Choosing the replayed form of a step that contains collectives, TOGETHER with every other rank.

    A rank that drops out of the common sequence of collectives on its own -- because its capture failed, or its validation
    did -- leaves the others waiting in a collective it will never join.  So every decision here is collective: each rank
    records its own verdict, never raises past a collective, and the ranks agree (MIN over a gloo side group with a short
    timeout: host-side, independent of the state of the RCCL stream) after each capture and after each validation; all keep a
    form or all drop to the next one (bench.py: whole-step graph -> segmented graphs -> eager; main_retrieval.GraphedStep)."""

    def __init__(self, world, rank, timeout_s=120, log=None):
        import datetime
        self.world, self.rank = int(world), int(rank)
        self.log = log or (lambda msg: None)
        self.side = None
        if self.world > 1:
            self.side = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))

    def agree(self, ok):
        """True iff EVERY rank says ok."""
        if self.side is None:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.side)
        return bool(int(t.item()))

    def attempt(self, what, eager_pass, make, result, same, freeze=None):
        """One form, tried by every rank at once.
            eager_pass()      the eager step(s) whose result the form must reproduce (called with the bank frozen)
            make()            -> (replay_pass, keep_alive): captures the form; replay_pass() replays what eager_pass ran
            result()          -> the tensor(s) to compare, after either pass
            same(a, b)        -> bool
            freeze(on)        freezes / releases what a step changes for good (the memory bank) around the validation
        Captured frozen and validated first, then captured again for use.  -> (replay_pass, keep_alive) or None, on all ranks alike."""
        if freeze:
            freeze(True)
        try:
            eager_pass()
            _sync()
            want = result()
            want = [t.clone() for t in want] if isinstance(want, (list, tuple)) else want.clone()
            form, err = None, None
            try:
                form = make()
            except Exception as e:          # noqa: BLE001 -- a failed capture is a verdict, not an error
                err = f"{type(e).__name__}: {e}"
                _sync()
            if not self.agree(form is not None):
                if err:
                    self.log(f"rank {self.rank}: {what} capture unavailable ({err})")
                return None
            form[0]()                        # every rank replays: the collectives inside / between the graphs match up
            _sync()
            ok = bool(same(result(), want))
            if not self.agree(ok):
                if not ok:
                    self.log(f"rank {self.rank}: the replayed {what} step differs from the eager one")
                return None
        finally:
            if freeze:
                freeze(False)
        form = None
        try:
            form = make()
        except Exception as e:              # noqa: BLE001
            self.log(f"rank {self.rank}: {what} capture (form to be used) failed ({type(e).__name__}: {e})")
            _sync()
        return form if self.agree(form is not None) else None
