#!/usr/bin/env python
"""Scorer MLP + masked softmax: ONE launch (nr_token_weights_fwd, ops.FUSE_TOKEN_SOFTMAX) against two launches
(nr_token_logits_fwd + nr_token_softmax), for the four scorer calls of the step at configs[1]; HIP-graph replays of ten
calls back to back, us per call; also checks that both forms give bit-identical weights."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import head, hip, modeling, ops, synth  # noqa: E402
from tools.branch_times import graph_time  # noqa: E402

DEV = "cuda"
B, Nt, Nv, M, K = 128, 24, 12, 512, 20


def main():
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(DEV).train()
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
    sw_t, sw_v = m.scorer_weights("text_weight_fc"), m.scorer_weights("video_weight_fc")
    cases = [("batch text  x3", p["text_feat"], p["text_mask"].float(), sw_t, B, Nt, hip.PREC_BF16X3),
             ("batch video x3", p["video_feat"], p["video_mask"].float(), sw_v, B, Nv, hip.PREC_BF16X3),
             ("bank text  bf16", p["mb_feat_t"], p["mb_mask_t"].float(), sw_t, M, Nt, hip.PREC_BF16),
             ("bank video bf16", p["mb_feat_v"], p["mb_mask_v"].float(), sw_v, M, Nv, hip.PREC_BF16)]
    with torch.no_grad():
        for name, feat, mask, sw, n, N, prec in cases:
            prep = ops.prepare_tokens(feat, mask, want_lo=True)
            res = {}
            for fused in (False, True):
                ops.FUSE_TOKEN_SOFTMAX = fused

                def ten():
                    for _ in range(10):
                        res[fused] = head.token_weights(prep, mask, sw, n, N, prec)[0]
                t = graph_time(ten, reps=50) / 10
                print(f"{name:16s} {'one launch ' if fused else 'two launches'}: {t:7.2f} us")
            print(f"{'':16s} identical weights: {bool(torch.equal(res[False], res[True]))}")
    ops.FUSE_TOKEN_SOFTMAX = True


if __name__ == "__main__":
    main()
