#!/usr/bin/env python
"""-DNR_TUNE build: the 8-wave ping-pong split-bf16 block (the default) against the 4-wave block (NR_SIM_X3PP=0): same S up
to the order of the weighted sums in the epilogue?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops
g = torch.Generator().manual_seed(1)
for A, Bv in ((128, 128), (125, 131), (256, 64)):
    t = torch.randn(A, 24, 512, generator=g).cuda(); v = torch.randn(Bv, 12, 512, generator=g).cuda()
    tm = (torch.rand(A, 24, generator=g) > 0.2).float().cuda(); vm = (torch.rand(Bv, 12, generator=g) > 0.2).float().cuda()
    pt, pv = ops.prepare_tokens(t, tm), ops.prepare_tokens(v, vm)
    wt = torch.softmax(torch.randn(A, 24, generator=g), -1).cuda(); wv = torch.softmax(torch.randn(Bv, 12, generator=g), -1).cuda()
    os.environ["NR_SIM_X3PP"] = "0"
    S0 = ops.local_level(pt, pv, wt, wv, A, 24, Bv, 12, hip.PREC_BF16X3, hip.OUT_FULL)
    S0 = S0[0] if isinstance(S0, tuple) else S0
    os.environ.pop("NR_SIM_X3PP", None)
    S1 = ops.local_level(pt, pv, wt, wv, A, 24, Bv, 12, hip.PREC_BF16X3, hip.OUT_FULL)
    S1 = S1[0] if isinstance(S1, tuple) else S1
    torch.cuda.synchronize()
    print(A, Bv, "tiles", hip.local_level_tiles(A, 24, Bv, 12, hip.PREC_BF16X3), "max |dS|", float((S0 - S1).abs().max()), "equal", bool(torch.equal(S0, S1)))
