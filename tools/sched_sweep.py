#!/usr/bin/env python
"""Scheduling knobs of the captured loss-only step (configs[1]): how the clustering launches are interleaved with the local
branch's in capture order (`capture_order`), how many bank chains start early (`bank_early`) -- us per step of each setting,
every setting captured separately in one process, two rounds."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import modeling, synth
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
dev = torch.device("cuda")
p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
BIG = 1 << 30
SETTINGS = [
    ("default ((7,7),(7,inf)) early=2", ((7, 7), (7, BIG)), 2),
    ("early=1", ((7, 7), (7, BIG)), 1),
    ("early=0", ((7, 7), (7, BIG)), 0),
    ("early=2 bb_late", ((7, 6), (7, BIG)), 2, True),
    ("early=1 bb_late", ((7, 6), (7, BIG)), 1, True),
    ("early=0 bb_late", ((7, 6), (7, BIG)), 0, True),
    ("((7,5),(7,inf)) early=2", ((7, 5), (7, BIG)), 2),
    ("((7,9),(7,inf)) early=2", ((7, 9), (7, BIG)), 2),
    ("((7,9),(3,4),(4,inf))", ((7, 9), (3, 4), (4, BIG)), 2),
    ("((4,5),(3,4),(7,inf))", ((4, 5), (3, 4), (7, BIG)), 2),
    ("((7,7),(7,inf))", ((7, 7), (7, BIG)), 2),
    ("((7,11),(7,inf))", ((7, 11), (7, BIG)), 2),
    ("((7,13),(7,inf))", ((7, 13), (7, BIG)), 2),
    ("((3,9),(4,4),(7,inf))", ((3, 9), (4, 4), (7, BIG)), 2),
    ("((1,1),) alternate", ((1, 1),), 2),
    ("((2,1),)", ((2, 1),), 2),
    ("((1,2),)", ((1, 2),), 2),
    ("((14,0),(0,inf)) clustering first", ((14, 0), (0, BIG)), 2),
    ("((0,inf),(14,0)) local first", ((0, BIG), (14, 0)), 2),
]
def build(order, early, bb_late=False):
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(dev).train()
    m.capture_order, m.bank_early, m.bb_late = order, early, bb_late
    m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = (p[k].clone() for k in ("mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v"))
    m.mb_ind = torch.arange(M, device=dev)
    def step():
        with torch.no_grad():
            return m(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0)
    for _ in range(3): step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    return m, g
def timeit(g, n=600):
    for _ in range(30): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
if "--random" in sys.argv[1:]:
    import random
    rng = random.Random(int(sys.argv[sys.argv.index("--random") + 1]))
    SETTINGS = SETTINGS[:1]
    for _ in range(40):
        turns, c = [], 14
        while c > 0:
            a = rng.randint(1, min(7, c)); c -= a
            turns.append((a, rng.randint(0, 9)))
        turns.append((0, BIG))
        SETTINGS.append((str(tuple(turns)).replace(str(BIG), "inf"), tuple(turns), 2))
built = [(st[0], build(*st[1:])) for st in SETTINGS]
for rnd in range(2):
    for name, (m, g) in built:
        print(f"round {rnd}  {name:40s} {timeit(g):7.1f} us", flush=True)
