#!/usr/bin/env python
"""fwd+bwd steps of the configs[1] workload for rocprofv3 (training path, eager)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import modeling, synth
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.cuda().train()
p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
m.mb_ind = torch.arange(M).cuda()
tf = p["text_feat"].clone().requires_grad_(True); vf = p["video_feat"].clone().requires_grad_(True)
import time
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    if it == 3: torch.cuda.synchronize(); t0 = time.perf_counter()
    m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
    ls = m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)
    ls[0].backward()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / (it - 2) * 1e3)
