#!/usr/bin/env python
"""Times of the similarity backward at configs[1] (B=128, M=512, 24 x 12 tokens, d=512): the four products' token gradients
(nr_local_level_bwd_group), the six weight sums (nr_pool_weight_bwd_group) and the operand transposes, each timed with events
over 50 launches.  Run on the GPU box:  python tools/sim_bwd_times.py [bf16x3]
In a -DNR_TUNE build NR_BWD_SLICES (slices per workgroup) and NR_BWD_DBG (1 no generation, 2 no MFMAs, 4 no stores) apply."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops, synth

B, Nt, Nv, M, d = 128, 24, 12, 512, 512
use_lo = len(sys.argv) > 1 and sys.argv[1] == "bf16x3"
dev = torch.device("cuda", 0)
p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
g = torch.Generator().manual_seed(0)
pt, pv = ops.prepare_tokens(p["text_feat"], p["text_mask"]), ops.prepare_tokens(p["video_feat"], p["video_mask"])
pbt, pbv = ops.prepare_tokens(p["mb_feat_t"], p["mb_mask_t"]), ops.prepare_tokens(p["mb_feat_v"], p["mb_mask_v"])
sm = lambda n, N: torch.softmax(torch.randn(n, N, generator=g), -1).to(dev)
w_t, w_v, w_bt, w_bv = sm(B, Nt), sm(B, Nv), sm(M, Nt), sm(M, Nv)
_, aux0 = ops.local_level(pt, pv, w_t, w_v, B, Nt, B, Nv, hip.PREC_BF16X3, hip.OUT_FULL, want_arg=True)
_, aux1 = ops.local_level(pt, pbv, w_t, w_bv, B, Nt, M, Nv, hip.PREC_BF16X3, hip.OUT_FULL, want_arg=True)
_, aux2 = ops.local_level(pbt, pv, w_bt, w_v, M, Nt, B, Nv, hip.PREC_BF16X3, hip.OUT_FULL, want_arg=True)
dS, d_c1, d_c0 = torch.randn(B, B, generator=g).to(dev), torch.randn(B, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
f32 = dict(dtype=torch.float32, device=dev)
d_tn, d_vn = torch.empty((B * Nt, d), **f32), torch.empty((B * Nv, d), **f32)
d_wt, d_wv, d_wbt, d_wbv = (torch.empty((n,), **f32) for n in (B * Nt, B * Nv, M * Nt, M * Nv))


def transposes():
    return ops.transpose_prepared([pv, pt, pbv, pbt], use_lo=use_lo)


T_pv, T_pt, T_pbv, T_pbt = transposes()


def tokens():
    ops.local_level_bwd_group([
        dict(side=0, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pv, w_self=w_t, w_other=w_v, aux=aux0, A=B, Nt=Nt, Bv=B, Nv=Nv, d_x=d_tn),
        dict(side=0, dS=d_c1, ds_mode=1, ds_scale=1.0 / M, other_T=T_pbv, w_self=w_t, w_other=w_bv, aux=aux1, A=B, Nt=Nt, Bv=M, Nv=Nv,
             d_x=d_tn),
        dict(side=1, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pt, w_self=w_v, w_other=w_t, aux=aux0, A=B, Nt=Nt, Bv=B, Nv=Nv, d_x=d_vn),
        dict(side=1, dS=d_c0, ds_mode=2, ds_scale=1.0 / M, other_T=T_pbt, w_self=w_v, w_other=w_bt, aux=aux2, A=M, Nt=Nt, Bv=B, Nv=Nv,
             d_x=d_vn)], use_lo=use_lo)


def weights():
    ops.pool_weight_bwd_group([
        dict(side=0, N=Nt, d_w=d_wt, srcs=[(dS, 0, 1.0, aux0[2], B, B), (d_c1, 1, 1.0 / M, aux1[2], B, M)]),
        dict(side=1, N=Nv, d_w=d_wv, srcs=[(dS, 0, 1.0, aux0[3], B, B), (d_c0, 2, 1.0 / M, aux2[3], M, B)]),
        dict(side=1, N=Nv, d_w=d_wbv, srcs=[(d_c1, 1, 1.0 / M, aux1[3], B, M)]),
        dict(side=0, N=Nt, d_w=d_wbt, srcs=[(d_c0, 2, 1.0 / M, aux2[2], M, B)])])


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tag = "slices=%s dbg=%s %s" % (os.environ.get("NR_BWD_SLICES", "-"), os.environ.get("NR_BWD_DBG", "-"), "bf16x3" if use_lo else "bf16")
print("%-32s token gradients (2 launches) %7.1f us   weight sums %6.1f us   transposes %6.1f us"
      % (tag, timed(tokens), timed(weights), timed(transposes)), flush=True)
