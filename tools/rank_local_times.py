#!/usr/bin/env python
"""What ONE rank of the sharded loss-only step does at W = 2 / 4 / 8, measured on the one GPU of a box.

    python tools/rank_local_times.py [--worlds 2 4 8] [--rank 0] [--B 128 --M 512 --K 20] [--out profiles/r04_rank_local.txt]

The W-rank job is emulated (neighborretr_amd.comm.EmulatedWorld): rank r runs exactly the code of `bench.py --gpus W` --
`NeighborRetr.forward` with config.world_size = W: packed exchange step, the rank's share of the clustering with the batch-wide
maximum exchanged inside, gather of the global tokens, logits, Sinkhorn, the rank's 2/W of the batch x batch product and 1/W of
both bank products, centrality gather, slab row losses, row-term all-reduce, bank push -- on b = B/W samples; every collective
moves THIS rank's real message through a 1-rank RCCL communicator and finds the peers' parts in pre-filled buffers (one device
copy per gather stands in for their writes).  So launch count, kernel sizes and message sizes are the W-rank job's; the
peers' wire time (xGMI) is NOT in these numbers.  Before anything is timed, every emulated rank's losses are checked against
the replicated single-rank step on the same gathered batch.

Timed per W (HIP-graph replays, un-profiled): the whole step as ONE graph (collectives inside), the same step on ONE stream
(= the sum of its kernels' time), the step as SEGMENTED graphs (comm.SegmentedStep: the fallback when a whole-step capture is
refused), the eager step, and the pieces of the critical chain.  (modeling.py:274-298, until_module.py:367-412.)
"""
import argparse
import contextlib
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import comm, hip, modeling, synth  # noqa: E402

CFG = dict(B=128, Nt=24, Nv=12, M=512, K=20)


def init_one_rank_group(backend="nccl"):
    if dist.is_initialized():
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group(backend, rank=0, world_size=1)


def replay_time(replay, reps=200, repeats=3):
    """us per replay: `reps` back-to-back replays between two events, best and median of `repeats`."""
    for _ in range(10):
        replay()
    ts = []
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return float(np.median(ts))


def capture(fn, warm=3):
    """-> (graph, what fn returned inside the capture)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        res = fn()
    return g, res


class RankLocal:
    """The emulated W-rank job around one model: settle(), check(), and timed forms of rank r's step."""

    def __init__(self, model, full, world, device, real_collectives=True):
        self.m, self.full, self.W, self.dev = model, full, int(world), device
        self.B = full["text_feat"].shape[0]
        if self.B % self.W:
            raise ValueError("the global batch must divide over the emulated ranks")
        self.b = self.B // self.W
        self.world = comm.EmulatedWorld(self.W, real_collectives=real_collectives)
        self.rng0 = model._rng_state_on(device).clone()
        self.shards = [{k: full[k][r * self.b:(r + 1) * self.b].contiguous() for k in ("text_feat", "video_feat", "text_mask", "video_mask", "idx")}
                       for r in range(self.W)]
        self.losses = [None] * self.W

    def configure(self, rank):
        c = self.m.config
        c.world_size, c.local_rank = self.W, rank
        self.m.shard_loss = True
        self.m._rng_state.copy_(self.rng0)          # every rank of a real job draws the same batch-wide DPC-KNN noise (same seed)

    def step(self, rank, communicator=None):
        """communicator None: whatever is current (comm.SegmentedStep installs its own around the step)."""
        s = self.shards[rank]
        with (comm.use(communicator) if communicator is not None else contextlib.nullcontext()), torch.no_grad():
            out = self.m(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], 0)
        base = out[0]._base
        return base if base is not None and base.numel() == 5 else torch.stack(out)

    def settle(self):
        self.m.bank_frozen = True

        def run(r):
            self.configure(r)
            self.losses[r] = self.step(r, self.world.comm(r)).clone()
        try:
            return self.world.settle(run)
        finally:
            self.m.bank_frozen = False

    def replicated_losses(self):
        """The single-rank step on the same gathered batch, same noise, bank untouched."""
        c = self.m.config
        c.world_size, c.local_rank = 1, 0
        self.m.shard_loss = None
        self.m._rng_state.copy_(self.rng0)
        self.m.bank_frozen = True
        try:
            f = self.full
            with torch.no_grad():
                out = self.m(f["text_feat"], f["text_mask"], f["video_feat"], f["video_mask"], f["idx"], 0)
            return torch.stack(list(out)).clone()
        finally:
            self.m.bank_frozen = False

    def check(self, tol=2e-5):
        keep = (self.m.interleave_steps,)
        self.m.interleave_steps = False
        try:
            ref = self.replicated_losses()
        finally:
            self.m.interleave_steps, = keep
        worst = max(float((l - ref).abs().max()) for l in self.losses)
        if not worst <= tol:
            raise AssertionError(f"emulated W={self.W}: rank losses differ from the replicated step by {worst:.3e} (> {tol})")
        return worst, ref


def measure(model, full, W, rank, dev, lines):
    rl = RankLocal(model, full, W, dev)
    sweeps = rl.settle()
    worst, ref = rl.check()
    c = rl.world.comm(rank)
    rl.configure(rank)
    out = {"W": W, "b": rl.b, "sweeps": sweeps, "max_dL_vs_replicated": worst}

    def step():
        return rl.step(rank, c)

    def validated(make):
        """Builds a replayable form of the step with the bank frozen and the noise stream rewound, replays it once and returns
        max |dL| against the settled eager losses of this rank (then the caller rebuilds it unfrozen for timing)."""
        model.bank_frozen = True
        try:
            model._rng_state.copy_(rl.rng0)
            replay, result = make()
            model._rng_state.copy_(rl.rng0)
            replay()
            torch.cuda.synchronize()
            return float((result - rl.losses[rank]).abs().max())
        finally:
            model.bank_frozen = False

    def whole():
        g, res = capture(step)
        return g.replay, res

    def segmented():
        for _ in range(2):
            step()
        seg = comm.SegmentedStep(lambda: rl.step(rank), c).capture()
        out["segments"] = seg.n_segments
        return seg.replay, seg.result

    step()
    torch.cuda.synchronize()
    n0 = hip.N_CALLS
    step()
    out["abi_calls"], out["collectives"] = hip.N_CALLS - n0, c.n_collectives
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    out["eager_us"] = (time.perf_counter() - t0) / 50 * 1e6
    # one graph, the shipped two-stream schedule
    out["graph_dL"] = validated(whole)
    out["graph_us"] = replay_time(whole()[0])
    # segmented graphs and the one-stream graph (a side stream cannot stay open across a cut)
    model.use_side_streams = False
    try:
        out["segmented_dL"] = validated(segmented)
        out["segmented_us"] = replay_time(segmented()[0])
        out["one_stream_graph_us"] = replay_time(whole()[0])
    finally:
        model.use_side_streams = True
    # pieces of the critical chain (each its own graph)
    s = rl.shards[rank]
    cfgm = model.config
    tm_all, vm_all = full["text_mask"].float(), full["video_mask"].float()
    with torch.no_grad():
        nz = model._draw_noise(rl.B, CFG["Nt"], CFG["Nv"], dev)

        def exchange():
            from neighborretr_amd.dist import packed_allgather
            with comm.use(c):
                c.begin_step()
                cfgm.shard_loss = True
                packed_allgather(s["text_feat"], s["video_feat"], s["idx"], s["text_mask"], s["video_mask"], cfgm)

        def cluster_and_gather():
            with comm.use(c):
                c.seq = 1                       # behind the exchange step: the maximum exchange, then the token gather
                model._gather_global(*model._merge_sharded(full["text_feat"], full["video_feat"], tm_all, vm_all, nz, rank, W), W)
        out["exchange_us"] = replay_time(capture(exchange)[0].replay)
        out["cluster_gather_us"] = replay_time(capture(cluster_and_gather)[0].replay)
    lines.append(f"W={W} b={rl.b:3d}  settled in {sweeps} sweeps, every rank's losses == replicated step to {worst:.1e}")
    lines.append(f"    rank {rank}: {out['abi_calls']} C-ABI calls + {out['collectives']} collectives per step")
    lines.append(f"    whole step, ONE graph (collectives inside, two streams) : {out['graph_us']:7.1f} us   (replay vs eager |dL| {out['graph_dL']:.1e})")
    lines.append(f"    whole step, one graph, ONE stream (sum of kernel time)  : {out['one_stream_graph_us']:7.1f} us")
    lines.append(f"    SEGMENTED graphs ({out['segments']} segments + {out['collectives']} eager collectives)      : {out['segmented_us']:7.1f} us   (replay vs eager |dL| {out['segmented_dL']:.1e})")
    lines.append(f"    eager launches                                          : {out['eager_us']:7.1f} us")
    lines.append(f"    pieces: exchange step (pack, all-gather, unpack) {out['exchange_us']:6.1f} us;  rank's clustering incl. max exchange + token gather {out['cluster_gather_us']:6.1f} us")
    return out


def measure_interleaved(model, full, W, rank, dev, lines):
    """The step-interleaved job (model.interleave_steps): per step one packed all-gather on every rank; the owner of the step
    (step index mod W) evaluates the loss with the single-rank kernels, the others push the gathered batch into their bank
    replica.  Times rank `rank`'s two kinds of step and what W consecutive steps cost it."""
    rl = RankLocal(model, full, W, dev)
    model.interleave_steps = True
    out = {"W": W, "b": rl.b}
    try:
        c = rl.world.comm(rank)

        def configure(r):
            cfg = model.config
            cfg.world_size, cfg.local_rank = W, r
            model.shard_loss = False

        def settle_run(r):
            configure(r)
            model._rng_state.copy_(rl.rng0)
            model._step_index = r                       # rank r as the owner of "its" step
            rl.losses[r] = rl.step(r, rl.world.comm(r)).clone()
        model.bank_frozen = True
        try:
            out["sweeps"] = rl.world.settle(settle_run)
        finally:
            model.bank_frozen = False
        worst, _ = rl.check(tol=0.0)                    # the owner runs the single-rank kernels on the gathered batch: bit for bit
        out["max_dL_vs_replicated"] = worst
        configure(rank)

        def step_as(own):
            def f():
                model._step_index = rank if own else rank + 1
                s = rl.shards[rank]
                with torch.no_grad():
                    return model(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], 0)
            return f
        times = {}
        for own in (True, False):
            f = step_as(own)
            with comm.use(c):
                f()
                torch.cuda.synchronize()
                n0 = hip.N_CALLS
                f()
                calls = hip.N_CALLS - n0
                g, _ = capture(f)
                t_graph = replay_time(g.replay)
                for _ in range(2):                # (the step's only collective comes before anything is forked: side streams stay on)
                    f()
                seg = comm.SegmentedStep(f, c).capture()
                t_seg, n_seg = replay_time(seg.replay), seg.n_segments
            times[own] = (t_graph, t_seg, n_seg, calls)
        # one ROUND (this rank's own step and the W - 1 others) as ONE graph: what `bench.py --gpus W` replays (--unroll)
        def one_round():
            for k in range(W):
                model._step_index = k
                s_ = rl.shards[rank]
                with torch.no_grad():
                    model(s_["text_feat"], s_["text_mask"], s_["video_feat"], s_["video_mask"], s_["idx"], 0)
        with comm.use(c):
            g_round, _ = capture(one_round)
            out["round_graph_us"] = replay_time(g_round.replay, reps=60)
            del g_round
        out["round_graph_steps_per_s"] = W / out["round_graph_us"] * 1e6
        # the owner's loss BESIDE the following steps (model.interleave_overlap, neighborretr_amd.interleave): two graphs per owned
        # step -- exchange + bank copy + push on this stream, the loss from the copy on the model's loss stream.  Checked first:
        # frozen bank, rewound noise counter -> the replayed pair's losses == the settled owner's losses, bit for bit; and so
        # are those of the eager form (forward() with interleave_overlap: the loss launched on the model's loss stream)
        from neighborretr_amd.interleave import OverlappedOwnedStep
        model.interleave_overlap = True
        out["overlap_us_per_round"] = None
        try:
            s = rl.shards[rank]
            if model._absorb_ready(dict(W=W, b=rl.b, shapes=[tuple(s["text_feat"].shape[1:]), tuple(s["video_feat"].shape[1:])])) is None:
                raise LookupError("the bank cannot absorb a gathered batch (not a device ring of its shapes): no overlapped form")

            def exchange_half(slot_index):
                model._step_index = rank
                return model.owned_exchange(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], slot_index=slot_index)

            def other_step():
                model._step_index = rank + 1
                with torch.no_grad():
                    model(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], 0)
            with comm.use(c):
                model.bank_frozen = True
                try:
                    own = OverlappedOwnedStep(model, exchange_half, lambda fn: capture(fn)[0])
                    out["overlap_dL"] = 0.0
                    for _ in range(2 * len(own.pairs)):                  # every slot's pair of graphs, twice
                        model._rng_state.copy_(rl.rng0)
                        own.replay()
                        torch.cuda.synchronize()
                        out["overlap_dL"] = max(out["overlap_dL"], float((own.losses - rl.losses[rank]).abs().max()))
                    model._rng_state.copy_(rl.rng0)
                    model._step_index = rank
                    with torch.no_grad():
                        ls = model(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], 0)
                    torch.cuda.synchronize()
                    out["overlap_eager_dL"] = float((torch.stack(list(ls)) - rl.losses[rank]).abs().max())
                finally:
                    model.bank_frozen = False
                if not (out["overlap_dL"] == 0.0 and out["overlap_eager_dL"] == 0.0):
                    raise AssertionError(f"overlapped owned step at W={W}: losses differ from the serial owner's ({out})")
                del own
                g_other, _ = capture(other_step)
                # How well the two kinds of graph overlap depends on the hardware queues their streams land on: three draws of
                # fresh streams (what bench.py does too: it keeps the fastest validated form, overlapped or serial)
                draws = []
                for _draw in range(3):
                    model._owned_ring, model._owned = [], None
                    model.owned_slots = 1 if _draw == 2 else 2          # (the third draw: ONE slot, as bench.py tries it)
                    own = OverlappedOwnedStep(model, exchange_half, lambda fn: capture(fn)[0])
                    # NOT on the default stream: replayed there, the exchange graphs and the loss graph take turns (723 us per
                    # round at W = 8 against 524 on a stream of the pool)
                    xs = torch.cuda.Stream()
                    xs.wait_stream(torch.cuda.current_stream())

                    def one_round_overlapped():
                        with torch.cuda.stream(xs):
                            own.replay()
                            for _ in range(W - 1):
                                g_other.replay()
                    import gc
                    gc.collect()                      # (no destruction of discarded graphs in the middle of a timed loop)
                    torch.cuda.synchronize()
                    gc.disable()
                    for _ in range(5):
                        one_round_overlapped()
                    ts = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(60):
                            one_round_overlapped()
                        torch.cuda.synchronize()
                        ts.append((time.perf_counter() - t0) / 60 * 1e6)
                    gc.enable()
                    draws.append(float(np.median(ts)))
                    # the two halves alone: exchange (pack, all-gather, unpack into the slot, bank copy, absorb) and loss
                    out["owned_exchange_us"] = replay_time(own.A.replay)
                    out["owned_loss_us"] = replay_time(own.pairs[0][1].replay)
                    del own
                out["overlap_draws_us"] = draws
                out["overlap_us_per_round"] = min(draws)
                out["overlap_steps_per_s"] = W / out["overlap_us_per_round"] * 1e6
                del g_other
        except LookupError as e:
            out["overlap_note"] = str(e)
        finally:
            model.interleave_overlap = False
            model.owned_slots = 2
        for key, k in (("graph", 0), ("segmented", 1)):
            per_round = times[True][k] + (W - 1) * times[False][k]
            out[key + "_us_per_round"] = per_round
            out[key + "_steps_per_s"] = W / per_round * 1e6
        out["owner_step_us"], out["other_step_us"] = times[True][0], times[False][0]
        out["owner_step_segmented_us"], out["other_step_segmented_us"] = times[True][1], times[False][1]
        out["abi_calls"] = {"owner": times[True][3], "other": times[False][3]}
    finally:
        model.interleave_steps = False
    lines.append(f"W={W} b={rl.b:3d}  step-interleaved: the owner's losses == the single-rank step bit for bit (max |dL| {out['max_dL_vs_replicated']:.1e})")
    lines.append(f"    rank {rank}, a step it OWNS  (exchange + full loss + push): {times[True][3]:3d} C-ABI calls, one graph {times[True][0]:7.1f} us, "
                 f"{times[True][2]} segments + 1 eager collective {times[True][1]:7.1f} us")
    lines.append(f"    rank {rank}, any OTHER step  (exchange + push)            : {times[False][3]:3d} C-ABI calls, one graph {times[False][0]:7.1f} us, "
                 f"{times[False][2]} segments + 1 eager collective {times[False][1]:7.1f} us")
    lines.append(f"    W={W} consecutive steps cost a rank {out['graph_us_per_round']:7.1f} us  ->  {out['graph_steps_per_s']:8.0f} steps/s for the job "
                 f"(segmented form: {out['segmented_us_per_round']:7.1f} us -> {out['segmented_steps_per_s']:8.0f} steps/s); no wire time in these")
    lines.append(f"    the round as ONE graph ({W} steps per replay): {out['round_graph_us']:7.1f} us  ->  {out['round_graph_steps_per_s']:8.0f} steps/s")
    if out["overlap_us_per_round"] is None:
        lines.append(f"    owner's loss beside the following steps: {out.get('overlap_note')}")
        return out
    lines.append(f"    owner's loss BESIDE the following steps (two graphs per owned step; replayed pair and eager form == the serial owner's losses, "
                 f"max |dL| {max(out['overlap_dL'], out['overlap_eager_dL']):.1e}): exchange half {out['owned_exchange_us']:6.1f} us, loss half "
                 f"{out['owned_loss_us']:6.1f} us alone; three draws of streams (two slots, two slots, one slot): " + " / ".join(f"{d_:.0f}" for d_ in out["overlap_draws_us"]) + " us per round; best: "
                 f"W={W} consecutive steps cost a rank {out['overlap_us_per_round']:7.1f} us  ->  {out['overlap_steps_per_s']:8.0f} steps/s for the job")
    return out


def measure_training(model, full, W, rank, dev, lines, graph=True):
    """The sharded TRAINING step (neighborretr_amd.sharded: forward + backward, loss / bank / clustering work sharded over the
    ranks) of rank `rank` at W emulated ranks: five collectives forward (packed gather, clustering maximum, global tokens,
    centrality slices, loss values), four backward (reduce-scatter of the centrality / token gathers' gradients and of the two
    gathered feature tensors).  Checked before timing: every rank's losses == the replicated training step, and the MEAN over
    the ranks of their parameter gradients (what DDP's all-reduce leaves in .grad) == the replicated step's gradients."""
    rl = RankLocal(model, full, W, dev)
    out = {"W": W, "b": rl.b}
    params = [p for p in model.parameters() if p.requires_grad]
    leaves = [(s["text_feat"].clone().requires_grad_(True), s["video_feat"].clone().requires_grad_(True)) for s in rl.shards]

    def train_step(r, communicator=None, functional=False):
        """functional: torch.autograd.grad instead of .backward() -- the form a CAPTURED step takes (main_retrieval.GraphedStep):
        no AccumulateGrad node, whose stream may be one the capture must not touch, takes part."""
        s, (tf_, vf_) = rl.shards[r], leaves[r]
        for p_ in params:
            p_.grad = None
        tf_.grad = vf_.grad = None
        with (comm.use(communicator) if communicator is not None else contextlib.nullcontext()):
            ls = model(tf_, s["text_mask"], vf_, s["video_mask"], s["idx"], 0)
            if functional:
                keep["grads"] = torch.autograd.grad(ls[0], params + [tf_, vf_], allow_unused=True)
            else:
                ls[0].backward()
        return torch.stack([l.detach() for l in ls])
    keep = {}
    grads = [None] * W

    def settle_run(r):
        rl.configure(r)
        rl.losses[r] = train_step(r, rl.world.comm(r)).clone()
        grads[r] = [None if p_.grad is None else p_.grad.clone() for p_ in params]
    model.bank_frozen = True
    try:
        out["sweeps"] = rl.world.settle(settle_run, max_sweeps=24)
        # the replicated training step on the gathered batch
        cfg = model.config
        cfg.world_size, cfg.local_rank = 1, 0
        model.shard_loss = None
        model._rng_state.copy_(rl.rng0)
        for p_ in params:
            p_.grad = None
        tf = full["text_feat"].clone().requires_grad_(True)
        ref = model(tf, full["text_mask"], full["video_feat"], full["video_mask"], full["idx"], 0)
        ref[0].backward()
        ref_l = torch.stack([l.detach() for l in ref])
        out["max_dL_vs_replicated"] = max(float((l - ref_l).abs().max()) for l in rl.losses)
        worst, names = 0.0, {id(p_): n for n, p_ in model.named_parameters()}
        for k, p_ in enumerate(params):
            if p_.grad is None or float(p_.grad.abs().max()) == 0.0:
                continue
            mean = sum(g[k] for g in grads if g[k] is not None) / W
            # (2e-6 absolute slack, as tests/test_sharded_gpu.py: gradients that are rounding noise around zero, e.g. score biases)
            dev_ = max(0.0, float((mean - p_.grad).abs().max()) - 2e-6) / float(p_.grad.abs().max())
            if dev_ > worst:
                worst, out["worst_param"] = dev_, names.get(id(p_), "?")
        out["max_param_grad_dev"] = worst
        own = leaves[rank][0].grad
        want = tf.grad[rank * rl.b:(rank + 1) * rl.b]
        out["feature_grad_dev"] = float((own - want).abs().max() / want.abs().max())
        # drop the replicated step's autograd graph NOW: while its losses live, the parameters' AccumulateGrad nodes live too, on
        # the (default) stream this step ran on -- and a later capture that has to synchronise with them dies in the runtime
        del ref, ref_l, tf, own, want
        if not (out["max_dL_vs_replicated"] <= 1e-3 and worst <= 2e-2 and out["feature_grad_dev"] <= 2e-2):
            raise AssertionError(f"emulated sharded training step at W={W} deviates from the replicated one: {out}")
    finally:
        model.bank_frozen = False
    c = rl.world.comm(rank)
    rl.configure(rank)

    def step():
        return train_step(rank, c)
    step()
    torch.cuda.synchronize()
    n0 = hip.N_CALLS
    step()
    out["abi_calls"], out["collectives"] = hip.N_CALLS - n0, c.n_collectives
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    out["eager_us"] = (time.perf_counter() - t0) / 20 * 1e6
    out["graph_us"] = None
    if graph:
        with model.graph_capture_mode():
            g, _ = capture(lambda: train_step(rank, c, functional=True))
        out["graph_us"] = replay_time(g.replay, reps=50)
        del g
    lines.append(f"W={W} b={rl.b:3d}  sharded TRAINING step (forward + backward): losses == replicated to {out['max_dL_vs_replicated']:.1e}, mean over ranks of "
                 f"the parameter gradients == replicated to {out['max_param_grad_dev']:.1e} of each tensor's largest entry, this rank's feature "
                 f"gradient to {out['feature_grad_dev']:.1e}  (settled in {out['sweeps']} sweeps)")
    lines.append(f"    rank {rank}: {out['abi_calls']} C-ABI calls + {out['collectives']} collectives per step;  eager {out['eager_us']:7.1f} us"
                 + (f";  ONE graph (forward + backward + the 1-rank RCCL collectives) {out['graph_us']:7.1f} us" if out["graph_us"] else ""))
    return out


def check_eval(model, dev, lines, W=4, N=203):
    """The sharded evaluation (neighborretr_amd.evaluator: a row slab of S and its rank counts per rank, fp32 / int32 collectives)
    with W emulated ranks, every collective on the 1-rank RCCL communicator: the ranks every emulated rank ends up with ==
    the single-rank ranks (the reference's `cols`, metrics.py:58-66), single- and multi-sentence."""
    from types import SimpleNamespace
    from neighborretr_amd import evaluator
    t, v, tm, vm = synth.make_samples(4242, "eval", N, CFG["Nt"], CFG["Nv"])
    t, v, tm, vm = (torch.from_numpy(a).to(dev) for a in (t, v, tm, vm))
    tm, vm = tm.float(), vm.float()
    one = SimpleNamespace(world_size=1, local_rank=0)
    with torch.no_grad():
        ref = evaluator.sharded_retrieval_ranks(model, t, v, tm, vm, one)
        ends = np.cumsum(np.random.RandomState(3).randint(1, 4, size=N))
        ends = ends[ends <= N]
        if ends[-1] != N:
            ends = np.append(ends, N)
        V = len(ends)
        ref_ms = evaluator.sharded_multi_sentence_metrics(model, t, v[:V].contiguous(), tm, vm[:V].contiguous(), ends - 1, one)
        world = comm.EmulatedWorld(W)
        got, got_ms = {}, {}

        def run(r):
            c = world.comm(r)
            args = SimpleNamespace(world_size=W, local_rank=r)
            with comm.use(c):
                c.begin_step()
                got[r] = evaluator.sharded_retrieval_ranks(model, t, v, tm, vm, args)
                got_ms[r] = evaluator.sharded_multi_sentence_metrics(model, t, v[:V].contiguous(), tm, vm[:V].contiguous(), ends - 1, args)
        sweeps = world.settle(run)
        run(0)                                        # once more from the frozen buffers
    for r in range(W):
        assert all(np.array_equal(a, b) for a, b in zip(got[r], ref)), f"emulated rank {r}: retrieval ranks differ from the single-rank ones"
        assert got_ms[r][0] == ref_ms[0] and got_ms[r][1]["cols"] == ref_ms[1]["cols"], f"emulated rank {r}: multi-sentence metrics differ"
    lines.append(f"sharded evaluation, W={W} emulated ranks, N={N}: every rank's text->video / video->text rank counts and the multi-sentence "
                 f"metrics ({V} videos) == the single-rank ones (settled in {sweeps} sweeps; fp32 gather, int32 all-reduce / all-gather, MAX "
                 f"all-reduce on the 1-rank RCCL communicator)")


def build(dev, precision="bf16"):
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=CFG["K"]), precision=precision)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(dev).train()
    with torch.no_grad():
        m.clip.logit_scale.fill_(float(np.log(100.0)))
    c = CFG
    full_np = synth.make_problem(1002, c["B"], c["Nt"], c["Nv"], c["M"])
    full = {k: torch.from_numpy(v).to(dev) for k, v in full_np.items()}
    m.mb_feat_t, m.mb_feat_v = full["mb_feat_t"], full["mb_feat_v"]
    m.mb_mask_t, m.mb_mask_v = full["mb_mask_t"], full["mb_mask_v"]
    m.mb_ind = torch.arange(10 ** 6, 10 ** 6 + c["M"], device=dev)
    return m, full


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worlds", type=int, nargs="*", default=[2, 4, 8])
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--B", type=int, default=CFG["B"], help="global batch (default: configs[1])")
    ap.add_argument("--M", type=int, default=CFG["M"])
    ap.add_argument("--K", type=int, default=CFG["K"])
    ap.add_argument("--out", default=None)
    ap.add_argument("--train", action="store_true", help="also the sharded TRAINING step (forward + backward) per world size")
    ap.add_argument("--only_train", action="store_true")
    ap.add_argument("--only_interleaved", action="store_true", help="skip the synchronous sharded step's section")
    ap.add_argument("--eval", action="store_true", help="also check the sharded evaluation's collectives under the emulated world")
    args = ap.parse_args()
    CFG.update(B=args.B, M=args.M, K=args.K)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    init_one_rank_group()
    model, full = build(dev)
    which = {128: "configs[1]", 1024: "configs[2]'s global batch"}.get(CFG["B"], "configs[1]'s token shapes")
    lines = [f"rank-local times of the sharded loss-only step, {which} (B={CFG['B']}, Nt={CFG['Nt']}, Nv={CFG['Nv']}, M={CFG['M']}), emulated "
             f"on one MI355X: this rank's messages through a 1-rank RCCL communicator, peers' parts pre-filled (no wire time)"]
    # the single-rank step for scale
    model.config.world_size = 1

    def one():
        with torch.no_grad():
            model(full["text_feat"], full["text_mask"], full["video_feat"], full["video_mask"], full["idx"], 0)
    lines.append(f"W=1 b={CFG['B']}  the replicated step as one graph: {replay_time(capture(one)[0].replay):7.1f} us")
    if not args.only_train:
        if not args.only_interleaved:
            lines.append("---- SYNCHRONOUS sharded step (every rank takes part in every step's loss; five collectives per step) ----")
            for W in args.worlds:
                measure(model, full, W, min(args.rank, W - 1), dev, lines)
        lines.append("---- STEP-INTERLEAVED (one collective per step; the loss of step k on rank k mod W) ----")
        for W in args.worlds:
            measure_interleaved(model, full, W, min(args.rank, W - 1), dev, lines)
    if args.eval:
        lines.append("---- sharded EVALUATION (neighborretr_amd.evaluator) ----")
        model.config.world_size, model.config.local_rank = 1, 0
        check_eval(model, dev, lines)
    if args.train or args.only_train:
        lines.append("---- sharded TRAINING step (neighborretr_amd.sharded), forward + backward ----")
        for W in args.worlds:
            measure_training(model, full, W, min(args.rank, W - 1), dev, lines)
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as f:
            f.write(text + "\n")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
