#!/usr/bin/env python
"""Fixed cost of a fused local_level launch: time the step's three products at d = 64 .. 1024 and fit t = t0 + slope * d/64
(t0 = dispatch + first-slice latency + epilogue + tail; slope = one K slice of the main loop)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops
from tools.sim_times import timed
CASES = [(128, 24, 512, 12, hip.PREC_BF16, hip.OUT_ROWSUM, "batch x bank-video bf16"),
         (512, 24, 128, 12, hip.PREC_BF16, hip.OUT_COLSUM, "bank-text x batch bf16"),
         (128, 24, 128, 12, hip.PREC_BF16X3, hip.OUT_FULL, "batch x batch x3")]
g = torch.Generator().manual_seed(0)
for (A, Nt, Bv, Nv, prec, mode, name) in CASES:
    ds, us = [256, 512, 768, 1024], []
    for d in ds:
        t = torch.randn(A, Nt, d, generator=g).cuda(); v = torch.randn(Bv, Nv, d, generator=g).cuda()
        pt = ops.prepare_tokens(t, torch.ones(A, Nt, device="cuda")); pv = ops.prepare_tokens(v, torch.ones(Bv, Nv, device="cuda"))
        wt = torch.full((A, Nt), 1.0 / Nt, device="cuda"); wv = torch.full((Bv, Nv), 1.0 / Nv, device="cuda")
        us.append(min(timed(lambda: ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec, mode), reps=30) for _ in range(3)))
    sl, t0 = np.polyfit(np.array(ds) / 64, us, 1)
    print(f"{name:28s} " + "  ".join(f"d={d}: {u:5.2f}" for d, u in zip(ds, us)) + f"   fit: t0 = {t0:.2f} us, {sl:.3f} us per K slice")
