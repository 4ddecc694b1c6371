#!/usr/bin/env python
"""ONE process, many HIP-graph captures of different stream topologies, each replay checked against the eager step.

The sequence a long `main_retrieval.py --hip_graph 1` run and bench.py walk between them: the loss-only step (five forked
streams), then main_retrieval.GraphedStep through FIVE bank-generation changes (= five re-captures of the training graph: the
memory bank is replaced at every epoch start), with the clustering form switched in between (grouped HIP kernels on their own
stream | autograd-traced torch ops on two side streams + a two-stream backward): the mix of topologies on which a process
segfaulted inside the ROCm 7.2 runtime at its fourth capture in round 3 (tools/train_times.py before one-child-per-setting),
when the side streams were objects cached for the life of the process.  Side streams now belong to one capture each
(neighborretr_amd/streams.py).  faulthandler prints the Python stack should the process die.

    python tools/capture_sequence.py [--B 32 --M 64]
"""
import argparse
import faulthandler
import os
import sys

import numpy as np
import torch

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from main_retrieval import GraphedStep  # noqa: E402
from neighborretr_amd import modeling, streams, synth  # noqa: E402


def say(msg):
    print(msg, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--M", type=int, default=64)
    ap.add_argument("--K", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, M, K, Nt, Nv = args.B, args.M, args.K, 24, 12
    x = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1003, B, Nt, Nv, M).items()}

    def model():
        m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
        m = m.to(dev).train()
        with torch.no_grad():
            m.clip.logit_scale.fill_(float(np.log(100.0)))
        return m

    def load_bank(m, shift):
        m.mb_feat_t, m.mb_feat_v = x["mb_feat_t"].clone() + shift, x["mb_feat_v"].clone() + shift
        m.mb_mask_t, m.mb_mask_v = x["mb_mask_t"].clone(), x["mb_mask_v"].clone()
        m.mb_ind = torch.arange(5000 + shift, 5000 + shift + M, device=dev)

    def batch(r):
        return (x["text_feat"] + 0.01 * r, x["text_mask"], x["video_feat"] + 0.01 * r, x["video_mask"], x["idx"] + 100 * r)
    captures = 0

    # ---- 1: the loss-only step, as bench.py captures it
    m0 = model()
    load_bank(m0, 0)
    bt = batch(0)
    out = {}

    def loss_only():
        with torch.no_grad():
            out["l"] = torch.stack(m0(*bt, 0))
    m0.bank_frozen = True
    for _ in range(3):
        loss_only()
    m0._rng_state[1] = 77
    loss_only()
    torch.cuda.synchronize()
    want = out["l"].clone()
    say("capture 1: loss-only step")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss_only()
    captures += 1
    m0._rng_state[1] = 77
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["l"], want), (out["l"], want)
    del g

    # ---- 2..7: the training graph through five bank generations, the clustering form alternating
    eager, graphed = model(), model()
    params_g = [p for p in graphed.parameters() if p.requires_grad]
    step = None
    for epoch in range(6):
        fused = epoch % 3 != 1                      # epochs 1 and 4: autograd-traced clustering on two side streams
        eager.fused_training_clustering = graphed.fused_training_clustering = fused
        load_bank(eager, epoch)
        load_bank(graphed, epoch)                   # a new bank generation: the graph is stale
        for r in range(2):
            bt = batch(2 * epoch + r)
            eager.zero_grad(set_to_none=True)
            if step is None:
                say(f"capture {captures + 1}: training graph, epoch {epoch} ({'grouped HIP' if fused else 'traced'} clustering)")
                step = GraphedStep(graphed, bt, params_g)
                captures += 1
            elif graphed._mb_gen != step.generation:
                say(f"capture {captures + 1}: training graph re-captured, epoch {epoch} ({'grouped HIP' if fused else 'traced'} clustering)")
                step.capture()
                captures += 1
            for m_ in (eager, graphed):
                m_._rng_state_on(dev)[1] = 1000 + 2 * epoch + r
            le = eager(*bt, 0)
            le[0].backward()
            lg = step.run(bt)
            torch.cuda.synchronize()
            rel = abs(float(lg[0]) - float(le[0])) / abs(float(le[0]))
            assert rel < 1e-3, (epoch, r, float(lg[0]), float(le[0]))
            ge = torch.cat([p.grad.reshape(-1) for p in eager.parameters() if p.grad is not None])
            gg = torch.cat([p.grad.reshape(-1) for p in graphed.parameters() if p.grad is not None])
            gerr = float((ge - gg).norm() / ge.norm())
            assert gerr < 5e-3, (epoch, r, gerr)
        say(f"  epoch {epoch}: losses and gradients of the replayed step == eager (last: rel {rel:.1e}, grad {gerr:.1e})")
    say(f"ok {captures} captures in one process; capture streams created {streams.STATS['created']}, re-used {streams.STATS['reused']}")


if __name__ == "__main__":
    main()
