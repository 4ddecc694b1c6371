#!/usr/bin/env python
"""Both forms of the token scorers' backward (backward._mlp_backward_hip: hidden layer recomputed from the normalised bf16 pairs
like the forward kernel; backward._mlp_backward: from the raw features through a split-bf16 GEMM) against fp64 on random
tokens: max |error| / max |truth| of (dW1, db1, dW2, db2, dX).  Errors of ~1e-2 in dW1 / db1 / dX with dW2 / db2 at 1e-6 are
ONE ReLU decided differently for a hidden unit whose pre-activation is within rounding of zero.  Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neighborretr_amd import backward, hip, modeling, ops, synth
dev = torch.device("cuda", 0)
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=8), precision="bf16x3")
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.to(dev)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for (A, N) in ((14, 12), (16, 12), (10, 24), (5, 12), (30, 12)):
        x = torch.randn(A, N, 512, generator=g).to(dev)
        mask = torch.ones(A, N).to(dev)
        pt = ops.prepare_tokens(x, mask)
        n = A * N
        dl = (torch.randn(n, generator=g) * 1e-2).to(dev)
        sw = m.scorer_weights("video_weight_fc")
        mlp = m.video_weight_fc
        X = x.reshape(n, 512).double()
        W1, b1, w2 = mlp[0].weight.double(), mlp[0].bias.double(), mlp[2].weight.double().reshape(1, -1)
        h = X @ W1.T + b1
        dh = torch.where(h > 0, dl.double()[:, None] * w2, torch.zeros_like(h))
        truth = (dh.T @ X, dh.sum(0), (dl.double()[:, None] * h.clamp(min=0)).sum(0).reshape(1, -1), dl.double().sum().reshape(1), dh @ W1)
        res = backward._mlp_backward_hip([dict(sw=sw, sets=[(pt, x, dl, hip.PREC_BF16X3)])])[0]
        ref = backward._mlp_backward([x], [dl], mlp[0].weight, mlp[0].bias, mlp[2].weight, n, True)
        e1 = [float((a.double() - t.reshape(a.shape)).abs().max() / t.abs().max()) for a, t in zip(res, truth)]
        e2 = [float((a.double().reshape(t.shape) - t).abs().max() / t.abs().max()) for a, t in zip(ref, truth)]
        print(n, "fused", ["%.1e" % e for e in e1], " torch-form", ["%.1e" % e for e in e2])
