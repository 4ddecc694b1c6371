#!/usr/bin/env python
"""Training step on a short batch (B not a multiple of 4): which loss term goes NaN?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_entry_gpu import _trainable_model
from util import problem
B, Nt, Nv, M = 8, 24, 12, 40
x = problem(1003, B, Nt, Nv, M, device="cuda")
for nb in (8, 7, 6, 5, 4):
    for grad in (False, True):
        m = _trainable_model()
        m.mb_feat_t, m.mb_feat_v = x["mb_feat_t"].clone(), x["mb_feat_v"].clone()
        m.mb_mask_t, m.mb_mask_v = x["mb_mask_t"].clone(), x["mb_mask_v"].clone()
        m.mb_ind = torch.arange(5000, 5000 + M, device="cuda")
        bt = tuple(t[:nb] for t in (x["text_feat"], x["text_mask"], x["video_feat"], x["video_mask"], x["idx"]))
        with torch.set_grad_enabled(grad):
            ls = m(*bt, 0)
        torch.cuda.synchronize()
        print(nb, "grad" if grad else "nograd", [round(float(l), 4) for l in ls], flush=True)
