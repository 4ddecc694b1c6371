#!/usr/bin/env python
"""The launches bench.py's `roofline` object times -- the step's three similarity products of configs[1] on the bench's
synthetic tokens, replayed back to back from a graph -- and nothing else, for `rocprofv3 --kernel-trace --stats`: the
per-kernel averages of that run are the durations `roofline.avg_launch_us` must agree with."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import head, hip, ops, synth
B, Nt, Nv, M = 128, 24, 12, 512
dev = torch.device("cuda")
full = synth.make_problem(1002, B, Nt, Nv, M)
t = {k: torch.from_numpy(full[k]).to(dev) for k in ("text_feat", "video_feat", "text_mask", "video_mask", "mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v")}
p_bb, p_mlp, p_bank = head.precision_plan(head.PREC_MIXED)       # the bench default ("bf16" plan)
pt, pv = ops.prepare_tokens(t["text_feat"], t["text_mask"], want_lo=True), ops.prepare_tokens(t["video_feat"], t["video_mask"], want_lo=True)
pbt, pbv = ops.prepare_tokens(t["mb_feat_t"], t["mb_mask_t"]), ops.prepare_tokens(t["mb_feat_v"], t["mb_mask_v"])
w = lambda n, N: torch.full((n, N), 1.0 / N, device=dev)
w_t, w_v, w_bt, w_bv = w(B, Nt), w(B, Nv), w(M, Nt), w(M, Nv)
def three():
    ops.local_level(pt, pv, w_t, w_v, B, Nt, B, Nv, p_bb, hip.OUT_FULL)
    ops.local_level(pt, pbv, w_t, w_bv, B, Nt, M, Nv, p_bank, hip.OUT_ROWSUM)
    ops.local_level(pbt, pv, w_bt, w_v, M, Nt, B, Nv, p_bank, hip.OUT_COLSUM)
for _ in range(5): three()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): three()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): g.replay()
e0.record()
for _ in range(50): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"HIP events: {e0.elapsed_time(e1) * 1e3 / 1500:.2f} us per launch (mean of the three products)")
