import os, sys, torch
sys.path.insert(0, os.getcwd())
from neighborretr_amd import ops
from tools.branch_times import graph_time
g = torch.Generator(device="cuda").manual_seed(0)
for B in (256, 512, 1024):
    G = torch.randn(B, B, device="cuda", generator=g) * 8
    t = graph_time(lambda: ops.sinkhorn_targets(G, 0.7, 50), reps=20)
    print(f"sinkhorn_targets B={B}: {t:.1f} us")
