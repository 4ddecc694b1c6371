#!/usr/bin/env python
"""nr_row_losses at shapes the parametrised tests do not list (K = B, odd B): every term against the oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nr_oracle as O  # noqa: E402
from neighborretr_amd import ops  # noqa: E402

DEV = "cuda"
for B, K in ((4, 4), (4, 3), (13, 11), (13, 13), (41, 41), (16, 16), (5, 3), (7, 1), (3, 2), (2, 2), (2, 1)):
    g = torch.Generator().manual_seed(B + K)
    S = torch.rand(B, B, generator=g) * 0.12 + torch.eye(B) * 0.02
    G = torch.randn(B, B, generator=g) * 9
    c0 = torch.rand(B, generator=g) * 0.1
    c1 = torch.rand(B, generator=g) * 0.1
    wt = torch.exp(torch.randn(B, generator=g) * 0.01)
    wv = torch.exp(torch.randn(B, generator=g) * 0.01)
    ls = torch.tensor([100.0])
    T = 3.0
    d = lambda t: t.double()
    tr = O.sinkhorn_targets(d(G), 0.7)
    tc = O.sinkhorn_targets(d(G).t(), 0.7)
    bank_v2t = d(c0)[:, None].expand(B, 4)
    bank_t2v = d(c1)[:, None].expand(B, 4)
    ref = [O.centrality_loss(d(S), d(wt), d(wv), 100.0),
           (-(torch.log_softmax(d(G) * T, -1) * tr).sum(-1).mean() - (torch.log_softmax(d(G).t() * T, -1) * tc).sum(-1).mean()) / 2,
           O.neighbor_loss(d(S), bank_t2v, bank_v2t, K, T),
           O.kl_loss(d(G), d(S))]
    ref32 = O.neighbor_loss(S, c1[:, None].expand(B, 4), c0[:, None].expand(B, 4), K, T)
    rl = ops.row_losses(S.to(DEV), G.to(DEV), tr.float().contiguous().to(DEV), tc.float().contiguous().to(DEV), c0.to(DEV), c1.to(DEV),
                        wt.to(DEV), wv.to(DEV), ls.to(DEV), K, T)
    losses = ops.loss_finalize(rl, 1.0, 1.0, 1.0).cpu().double()
    print(f"B={B:3d} K={K:3d}  got {[round(float(x), 6) for x in losses[1:]]}  ref {[round(float(x), 6) for x in ref]}  neighbour fp32 oracle {float(ref32):.6f}")
