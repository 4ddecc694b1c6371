#!/usr/bin/env python
"""Eagerly launched training step (configs[1], fused clustering + hand-derived backward): the two modalities' backward on one
stream vs on two streams (cluster_fused.BACKWARD_ON_TWO_STREAMS), alternating in one session; and the traced clustering."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import cluster_fused, modeling, synth
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.cuda().train()
p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
m.mb_ind = torch.arange(M).cuda()
tf = p["text_feat"].clone().requires_grad_(True); vf = p["video_feat"].clone().requires_grad_(True)
def fb():
    m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
    m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)[0].backward()
def timeit(n=20):
    for _ in range(3): fb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rnd in range(3):
    for name, fused, two in (("fused, backward on one stream ", True, False), ("fused, backward on two streams", True, True), ("traced clustering             ", False, False)):
        m.fused_training_clustering = fused
        cluster_fused.BACKWARD_ON_TWO_STREAMS = two
        print(f"round {rnd}  {name}: {timeit():6.2f} ms", flush=True)
