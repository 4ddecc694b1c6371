#!/usr/bin/env python
"""Per-kernel timings of the loss head at the BASELINE configs[1] shape (HIP events, one stream).

    python tools/microbench.py [name ...]      names: sinkhorn sim mlp prepare rowloss cluster
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import head, hip, ops, synth  # noqa: E402

DEV = "cuda"
INNER = 10        # calls per captured graph: a replay costs ~8 us of its own
B, Nt, Nv, M, d = 128, 24, 12, 512, 512


def timeit(fn, reps=50, warm=5):
    """us per call, replayed from a HIP graph (an eager loop is host-bound below ~15 us per call)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    run = fn
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(INNER):
                fn()
        run = g.replay
    except Exception:
        torch.cuda.synchronize()
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / (INNER if run is not fn else 1)      # us


def main():
    want = set(sys.argv[1:]) or {"sinkhorn", "sim", "mlp", "prepare", "rowloss"}
    prob = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
    P = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_params(7).items()}
    g = torch.Generator(device="cpu").manual_seed(0)
    G = (torch.randn(B, B, generator=g) * 9).to(DEV)
    if "sinkhorn" in want:
        for it in (0, 1, 2, 10, 50):
            print(f"sinkhorn iters={it:2d}: {timeit(lambda: ops.sinkhorn_targets(G, 0.7, it)):8.1f} us")
    pt = ops.prepare_tokens(prob["text_feat"], prob["text_mask"])
    pv = ops.prepare_tokens(prob["video_feat"], prob["video_mask"])
    pbt = ops.prepare_tokens(prob["mb_feat_t"], prob["mb_mask_t"])
    pbv = ops.prepare_tokens(prob["mb_feat_v"], prob["mb_mask_v"])
    if "prepare" in want:
        print(f"prepare text  [{B * Nt} tok]: {timeit(lambda: ops.prepare_tokens(prob['text_feat'], prob['text_mask'].float(), want_colsum=True)):8.1f} us")
        print(f"prepare bank_t[{M * Nt} tok]: {timeit(lambda: ops.prepare_tokens(prob['mb_feat_t'], prob['mb_mask_t'].float(), want_lo=False)):8.1f} us")
    w = lambda n, N: torch.full((n, N), 1.0 / N, device=DEV)
    w_t, w_v, w_bt, w_bv = w(B, Nt), w(B, Nv), w(M, Nt), w(M, Nv)
    if "sim" in want:
        f_bb = 2 * d * (B * Nt) * (B * Nv)
        f_bm = 2 * d * (B * Nt) * (M * Nv)
        for name, fn, fl in (
            ("BxB bf16  ", lambda: ops.local_level(pt, pv, w_t, w_v, B, Nt, B, Nv, hip.PREC_BF16), f_bb),
            ("BxB bf16x3", lambda: ops.local_level(pt, pv, w_t, w_v, B, Nt, B, Nv, hip.PREC_BF16X3), f_bb),
            ("BxM bf16  ", lambda: ops.local_level(pt, pbv, w_t, w_bv, B, Nt, M, Nv, hip.PREC_BF16, hip.OUT_ROWSUM), f_bm),
            ("MxB bf16  ", lambda: ops.local_level(pbt, pv, w_bt, w_v, M, Nt, B, Nv, hip.PREC_BF16, hip.OUT_COLSUM), f_bm),
            ("BxM bf16x3", lambda: ops.local_level(pt, pbv, w_t, w_bv, B, Nt, M, Nv, hip.PREC_BF16X3, hip.OUT_ROWSUM), f_bm),
        ):
            us = timeit(fn)
            print(f"sim {name}: {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s algorithmic")
    if "mlp" in want:
        sw = head.ScorerWeights(P["text_weight_fc.0.weight"], P["text_weight_fc.0.bias"], P["text_weight_fc.2.weight"],
                                P["text_weight_fc.2.bias"])
        for name, prep, prec in (("text  bf16x3", pt, hip.PREC_BF16X3), ("video bf16x3", pv, hip.PREC_BF16X3),
                                 ("bank_t bf16 ", pbt, hip.PREC_BF16), ("bank_v bf16 ", pbv, hip.PREC_BF16),
                                 ("bank_t bf16x3", pbt, hip.PREC_BF16X3)):
            us = timeit(lambda: ops.token_logit_parts(prep, sw.w1_hi, sw.w1_lo, sw.b1, sw.w2, prec))
            fl = 2 * d * 1024 * prep.n_tok
            print(f"mlp {name} [{prep.n_tok} tok]: {us:8.1f} us  {fl / us / 1e6:8.1f} TFLOP/s")
        # the four scorer calls of a step with their softmax, one by one and as ONE grouped launch (nr_token_weights_fwd_group)
        fl_all = 2 * d * 1024 * (pt.n_tok + pv.n_tok + pbt.n_tok + pbv.n_tok)
        sets = [(pt, B, Nt, hip.PREC_BF16X3), (pv, B, Nv, hip.PREC_BF16X3), (pbv, M, Nv, hip.PREC_BF16), (pbt, M, Nt, hip.PREC_BF16)]
        calls = [(p_, sw.w1_hi, sw.w1_lo, sw.b1, sw.w2, sw.b2, None, n_, N_) for p_, n_, N_, _ in sets]

        def singly():
            return [ops.token_weights(*c_, pr_) for c_, (_, _, _, pr_) in zip(calls, sets)]
        us1 = timeit(singly)
        usg = timeit(lambda: ops.token_weights_group(calls, [pr_ for _, _, _, pr_ in sets]))
        a_, b_ = singly(), ops.token_weights_group(calls, [pr_ for _, _, _, pr_ in sets])
        dev_ = max(float((x[0] - y[0]).abs().max()) for x, y in zip(a_, b_))
        print(f"mlp 4 sets + softmax, four launches : {us1:8.1f} us  {fl_all / us1 / 1e6:8.1f} TFLOP/s algorithmic")
        print(f"mlp 4 sets + softmax, ONE launch    : {usg:8.1f} us  {fl_all / usg / 1e6:8.1f} TFLOP/s algorithmic   (max |dw| vs the four launches {dev_:.2e})")
    if "rowloss" in want:
        S = torch.rand(B, B, device=DEV) * 0.1
        tr, tc = ops.sinkhorn_targets(G, 0.7, 50)
        v = torch.rand(B, device=DEV) * 0.1
        one = torch.ones(1, device=DEV) * 100
        print(f"row_losses: {timeit(lambda: ops.row_losses(S, G, tr, tc, v, v, v + 1, v + 1, one, 20, 3.0)):8.1f} us")


if __name__ == "__main__":
    main()
