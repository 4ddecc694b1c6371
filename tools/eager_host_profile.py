#!/usr/bin/env python
"""Where the host time of an EAGER loss-only step goes (cProfile over 50 steps; configs[1])."""
import cProfile, os, pstats, sys, time
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
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print(f"eager loss-only step: {(time.perf_counter() - t0) * 10:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18); st.sort_stats("cumulative").print_stats(28)
