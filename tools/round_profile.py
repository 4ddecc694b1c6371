#!/usr/bin/env python
"""Rank 0's round of the emulated W-rank step-interleaved job, replayed a few times -- `W serial`: the serial round as one graph;
`W two [single|fused] [slots] [AB|AO] [one|own|chain]`: the overlapped owned step as two graphs on two streams with the knobs
tools/ovl_probe.sh sweeps.  Run under `rocprofv3 --kernel-trace` (tools/round_timeline.sh) to see which kernels run beside which."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import comm  # noqa: E402
from tools import rank_local_times as RL  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
serial = len(sys.argv) > 2 and sys.argv[2] == "serial"
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
RL.init_one_rank_group()
model, full = RL.build(dev)
rl = RL.RankLocal(model, full, W, dev)
model.interleave_steps = True
model.interleave_overlap = not serial
cfg = model.config
rank = 0


def settle_run(r):
    cfg.world_size, cfg.local_rank = W, r
    model.shard_loss = False
    model._rng_state.copy_(rl.rng0)
    model._step_index = r
    rl.losses[r] = rl.step(r, rl.world.comm(r)).clone()


model.bank_frozen = True
rl.world.settle(settle_run)
model.bank_frozen = False
cfg.world_size, cfg.local_rank = W, rank
c = rl.world.comm(rank)
s = rl.shards[rank]


def own_exchange():
    model._step_index = rank
    return model.owned_exchange(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], slot_index=0)


def other_step(j):
    model._step_index = rank + j
    with torch.no_grad():
        model(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], 0)


def a_round():
    """The serial round: the rank's own step and the W - 1 behind it, one after the other."""
    for j in range(W):
        other_step(j)


if len(sys.argv) > 2 and sys.argv[2] == "two":
    # the two-graph form: exchange graphs on one stream, the loss graphs on another; argv[3]: "fused" = the W - 1 other steps as
    # one graph; argv[4]: slots (1 / 2); argv[5]: issue order "AB" (A, B, others) or "AO" (A, others, B)
    import time
    fused = len(sys.argv) > 3 and sys.argv[3] == "fused"
    n_slots = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    order = sys.argv[5] if len(sys.argv) > 5 else "AB"
    sides_mode = sys.argv[6] if len(sys.argv) > 6 else "one"      # "one": every loss graph on one stream; "own": a stream per slot; "chain": ... and B(k+1) behind B(k)
    model.owned_slots = n_slots
    with comm.use(c):
        def exch(k):
            model._step_index = rank
            return model.owned_exchange(s["text_feat"], s["text_mask"], s["video_feat"], s["video_mask"], s["idx"], slot_index=k)
        for k in range(n_slots):
            exch(k)

        def others():
            for j in range(1, W):
                other_step(j)
        gO, _ = RL.capture(others if fused else (lambda: other_step(1)))
        side = torch.cuda.Stream()
        sides = [side] + [torch.cuda.Stream() if sides_mode != "one" else side for _ in range(n_slots - 1)]
        main = torch.cuda.Stream()
        pairs = []
        last_b = [None]
        for k in range(n_slots):
            side = sides[k]
            gA, _ = RL.capture(lambda k=k: exch(k))
            slot = model._owned_ring[k]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.owned_loss(slot)
            torch.cuda.synchronize()
            gB = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gB, stream=side):
                model.owned_loss(slot)
            pairs.append((gA, gB, torch.cuda.Event(), torch.cuda.Event()))
        turn = [0]

        def rnd():
            gA, gB, evA, evB = pairs[turn[0] % n_slots]
            side = sides[turn[0] % n_slots]
            first = turn[0] < n_slots
            turn[0] += 1
            with torch.cuda.stream(main):
                if not first:
                    main.wait_event(evB)
                gA.replay()
                evA.record(main)
                if order == "AO":
                    for _ in range(1 if fused else W - 1):
                        gO.replay()
                side.wait_event(evA)
                if sides_mode == "chain" and last_b[0] is not None:
                    side.wait_event(last_b[0])
                with torch.cuda.stream(side):
                    gB.replay()
                    evB.record(side)
                last_b[0] = evB
                if order != "AO":
                    for _ in range(1 if fused else W - 1):
                        gO.replay()
        for _ in range(20):
            rnd()
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(40):
                rnd()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            print(f"round us {(time.perf_counter() - t0) / 40 * 1e6:.1f}   (host issue time per round {(t1 - t0) / 40 * 1e6:.1f} us)", flush=True)
    sys.exit(0)
with comm.use(c):
    g, _ = RL.capture(a_round)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("round us", RL.replay_time(g.replay, reps=40))
