"""Replayable form of an OVERLAPPED owned step of the step-interleaved job (model.interleave_overlap).

The step-interleaved job evaluates the loss of step k on rank k mod W; every rank gathers and pushes every batch.  Serially a
rank pays one loss evaluation (~340 us at configs[1]) and W - 1 exchange-and-push steps (~40 us each) per W steps.  The loss
of step k reads the memory bank as it stood BEFORE batch k was pushed and nothing else that later steps change -- so the owner
copies that state (`modeling.OwnedSlot`: the prepared bf16 shadow, masks, noise counter), pushes the batch at once like every
other rank, and evaluates the loss from the copy on a second stream while its first stream already takes part in the following
steps' exchanges.  Two graphs per owned step:

    A  (exchange half, the step's stream)   pack -> packed all-gather -> unpack into the slot -> bank copy -> absorb
    B  (loss half, the slot's own stream)   loss_step on the slot: prologue, clustering, products, Sinkhorn, row terms

tied together by two events OUTSIDE the graphs: B waits for A; the next A on the same slot waits for B.  Two slots take turns, each
with a loss stream of its own.  A contains the step's one
collective and is captured like any step with collectives (one graph with the RCCL all-gather inside, or comm.SegmentedStep);
B has none and is a plain graph.  Every step's losses and the bank after any number of steps are bit-identical to the serial
forms (tools/rank_local_times.py raises otherwise; tests/test_sharded_gpu.py, tests/test_rank_local_gpu.py).

Measured, emulated on one MI355X (profiles/r04_rank_local.txt, r04_overlap_probe.txt; us per round of W steps, serial form first): W = 2
358 -> 333; W = 4 402 -> 341-378; W = 8 499 -> 360-366 (450 in a process that has built and dropped many other graphs first).  How well the two kinds of
graph overlap depends on the hardware queues their streams land on, so bench.py builds this form on several draws of fresh streams
AND the serial form, validates each and keeps the fastest.  What else was tried (serial round at the time: 616 us at W = 8):
  * ONE slot (the next A waits for this B): 420-560 us; two slots on ONE loss stream: 491-598 (two loss graphs queued on one stream hold
    the exchange graphs up); three slots, a stream each: 432;
  * the pair replayed from the legacy DEFAULT stream: 723 us, the graphs take turns (bench.py runs N > 1 on a pool stream);
  * the whole round as ONE graph with the loss on forked streams: 719 us -- the runtime runs a graph's branches on two hardware
    queues, both of which the loss already uses; the exchange steps were appended to one of them whatever their capture order
    (kernel trace: every exchange kernel behind the loss's local chain on the same queue);
  * stream priorities for either half: 1.0-2.4 ms per round; 6 / 8 hardware queues (GPU_MAX_HW_QUEUES): up to 2 ms.
"""
import torch


class OverlappedOwnedStep:
    def __init__(self, model, exchange_fn, capture_exchange, warm=2):
        """exchange_fn(slot_index): runs model.owned_exchange(..., slot_index=slot_index) on the rank's shard (collectives through
        neighborretr_amd.comm).  capture_exchange(fn) -> an object with .replay() (a CUDAGraph, or a comm.SegmentedStep).
        One (A, B) pair of graphs per slot of the model (`model.owned_slots`, two by default)."""
        self.model = model
        n = max(1, int(model.owned_slots))
        for k in range(max(warm, 1) * n):          # the slots exist, the shadow is built, every kernel has been launched once
            model.owned_loss(exchange_fn(k % n))
        torch.cuda.synchronize()
        self.pairs = []
        self.slots = list(model._owned_ring)       # (the graphs name these buffers: they live as long as this form does)
        for k in range(n):
            A = capture_exchange(lambda k=k: exchange_fn(k))
            slot = model._owned_ring[k]
            B = torch.cuda.CUDAGraph()
            # (thread-local capture mode: the process group's watchdog thread polls its events while this thread captures)
            with torch.cuda.graph(B, stream=slot.stream, capture_error_mode="thread_local"):
                losses = model.owned_loss(slot)
            base = losses[0]._base                 # the five scalars are views of one [5] tensor
            self.pairs.append((A, B, base if base is not None and base.numel() == 5 else torch.stack(losses),
                               torch.cuda.Event(), torch.cuda.Event(), slot.stream))
        self.turn = 0
        self.pending = [False] * n
        # B > 128 solves Sinkhorn as ONE cooperative launch whose workgroups of a direction must all be resident together
        # (nr_sinkhorn_coop_kernel).  The host gate prices that for a launch that has the XCDs to itself: two loss halves on two
        # streams could each hold part of the CUs and starve each other until the bounded spin gives up (NaN losses).  So the
        # loss halves of such a batch run one BEHIND the other (an event edge between consecutive loss halves, whatever their
        # slot); they still run beside the exchange halves, which have no solve.
        from . import hip
        B_glob = int(self.slots[0].batch[0].shape[0]) if self.slots else 0
        self.serial_losses = bool(B_glob > 128 and hip.lib().nr_sinkhorn_cooperative_ok(B_glob))
        self._last_loss = None
        self.losses = self.pairs[0][2]             # of the LAST replayed owned step (valid once its loss half has finished)

    @property
    def A(self):
        return self.pairs[0][0]

    @property
    def n_segments(self):
        return getattr(self.pairs[0][0], "n_segments", 1)

    def replay(self):
        """On the current stream: A; on the loss stream, behind it: B.  Returns at once (no host synchronisation)."""
        k = self.turn % len(self.pairs)
        self.turn += 1
        A, B, losses, ev_a, ev_b, side = self.pairs[k]
        cur = torch.cuda.current_stream()
        if self.pending[k]:
            cur.wait_event(ev_b)                   # the slot is free again: its previous loss half has read it
        A.replay()
        ev_a.record(cur)
        side.wait_event(ev_a)
        if self.serial_losses and self._last_loss is not None and self._last_loss is not ev_b:
            side.wait_event(self._last_loss)       # the previous loss half (another slot's stream) has finished its solve
        with torch.cuda.stream(side):
            B.replay()
            ev_b.record(side)
        self._last_loss = ev_b
        self.pending[k] = True
        self.losses = losses

    def wait(self):
        """Orders the current stream behind the last loss half."""
        k = (self.turn - 1) % len(self.pairs)
        if self.turn and self.pending[k]:
            torch.cuda.current_stream().wait_event(self.pairs[k][4])
