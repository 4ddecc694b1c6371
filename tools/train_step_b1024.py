#!/usr/bin/env python
"""One replicated training step at configs[2]'s batch (B=1024, 24 x 12 tokens, M=512) on one GPU: finite losses and gradients,
timing of 3 eager steps.  A smoke run of the grouped backward kernels at their largest single-GPU shape."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import modeling, synth
B, Nt, Nv, M, K = 1024, 24, 12, 512, 20
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.cuda().train()
m.config.shard_loss = False
p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_problem(1003, B, Nt, Nv, M).items()}
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
m.mb_ind = torch.arange(M).cuda()
tf = p["text_feat"].clone().requires_grad_(True); vf = p["video_feat"].clone().requires_grad_(True)
for it in range(4):
    if it == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
    losses = m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)
    losses[0].backward()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
bad = [n for n, q in m.named_parameters() if q.grad is not None and not torch.isfinite(q.grad).all()]
print("losses", [round(float(x), 4) for x in losses], " ms/step %.2f" % (dt * 1e3), " non-finite parameter gradients:", bad,
      " |d text| %.3e |d video| %.3e" % (float(tf.grad.norm()), float(vf.grad.norm())), " peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
assert not bad and torch.isfinite(tf.grad).all() and torch.isfinite(vf.grad).all()
