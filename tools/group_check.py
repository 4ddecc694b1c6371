#!/usr/bin/env python
"""The step's three products through one grouped launch (nr_local_level_group) against three launches: same outputs?  and
the time of both forms (graph replays)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops
from tools.sim_times import timed
g = torch.Generator().manual_seed(1)
def prob(A, Bv, prec, mode, ragged=False):
    t = torch.randn(A, 24, 512, generator=g).cuda(); v = torch.randn(Bv, 12, 512, generator=g).cuda()
    tm = (torch.rand(A, 24, generator=g) > 0.2).float().cuda(); vm = (torch.rand(Bv, 12, generator=g) > 0.2).float().cuda()
    pt, pv = ops.prepare_tokens(t, tm, want_lo=True), ops.prepare_tokens(v, vm, want_lo=True)
    wt = torch.softmax(torch.randn(A, 24, generator=g), -1).cuda(); wv = torch.softmax(torch.randn(Bv, 12, generator=g), -1).cuda()
    return (pt, pv, wt, wv, A, 24, Bv, 12, prec, mode)
P = [prob(128, 512, hip.PREC_BF16, hip.OUT_ROWSUM), prob(512, 128, hip.PREC_BF16, hip.OUT_COLSUM), prob(128, 128, hip.PREC_BF16X3, hip.OUT_FULL)]
print("kinds", [hip.local_level_group_kind(q[4], q[5], q[6], q[7], 512, q[8]) for q in P])
ref = [ops.local_level(*q)[0] for q in P]
got = ops.local_level_group(P)
torch.cuda.synchronize()
for name, a, b in zip(("batch x bank-video", "bank-text x batch", "batch x batch x3"), ref, got):
    print(f"{name:20s} shape {tuple(a.shape)}  max |d| {float((a - b).abs().max()):.3e}  equal {bool(torch.equal(a, b))}")
for rnd in range(3):
    t3 = timed(lambda: [ops.local_level(*q) for q in P], reps=30)
    t1 = timed(lambda: ops.local_level_group(P), reps=30)
    t2 = timed(lambda: (ops.local_level_group(P[:2]), ops.local_level(*P[2])), reps=30)
    print(f"round {rnd}: three launches {t3:6.2f} us   one grouped launch {t1:6.2f} us   bank pair grouped + batch product {t2:6.2f} us")
