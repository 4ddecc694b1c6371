#!/usr/bin/env python
"""Does a stream-priority split between the step's two branches change the captured step?  The capture's origin stream (the
critical chain: clustering -> logits -> Sinkhorn) and the side streams (local branch, bank chains, push) at the priorities
given; one child process per combination (several captures with different topologies in one process are not safe on this
runtime).  us per replayed step."""
import os
import subprocess
import sys
import time

CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, sys.argv[3])
from neighborretr_amd import modeling, synth
po, ps = int(sys.argv[1]), int(sys.argv[2])
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
dev = torch.device("cuda")
p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.to(dev).train()
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = (p[k].clone() for k in ("mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v"))
m.mb_ind = torch.arange(M, device=dev)
m._lstream = torch.cuda.Stream(device=dev, priority=ps)
m._bstreams = tuple(torch.cuda.Stream(device=dev, priority=ps) for _ in range(3))
origin = torch.cuda.Stream(device=dev, priority=po)
def step():
    with torch.no_grad():
        return m(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0)
with torch.cuda.stream(origin):
    for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=origin):
    step()
with torch.cuda.stream(origin):
    for _ in range(300): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): g.replay()
    torch.cuda.synchronize()
print(f"origin priority {po:2d} | side streams priority {ps:2d}: {(time.perf_counter() - t0) / 1000 * 1e6:7.1f} us per step")
'''

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for po, ps in ((0, 0), (-1, 0), (0, -1), (-1, -1), (0, 0)):
        r = subprocess.run([sys.executable, "-c", CHILD, str(po), str(ps), root], capture_output=True, text=True, timeout=300)
        out = [l for l in r.stdout.splitlines() if "us per step" in l]
        print(out[-1] if out else f"origin {po} side {ps}: exit code {r.returncode} {r.stderr.strip().splitlines()[-1:]}", flush=True)
