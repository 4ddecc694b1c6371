#!/usr/bin/env python
"""How much throughput is there in keeping several loss-only steps in flight?  The whole step captured N times (separate
graphs, separate private pools) and replayed round-robin on N streams WITHOUT the dependencies a real pipeline needs
(bank push -> next step's bank products, prologue -> next prologue): an upper bound on what such a pipeline could give."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import modeling, synth
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
dev = torch.device("cuda")
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.to(dev).train()
p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
m.mb_ind = torch.arange(M, device=dev)
def step():
    with torch.no_grad():
        return m(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0)
for _ in range(3): step()
torch.cuda.synchronize()
graphs = []
for i in range(3):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    graphs.append((g, out))
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
for n in (1, 2, 3):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        steps = 600
        for i in range(steps):
            with torch.cuda.stream(streams[i % n]):
                graphs[i % n][0].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{n} step(s) in flight: {dt / steps * 1e6:7.1f} us per step  ({steps / dt:7.0f} steps/s)", flush=True)
