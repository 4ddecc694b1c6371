#!/usr/bin/env python
"""Un-profiled timings (HIP-graph replays, configs[1]) of the step's two chains on their own and together:
critical chain = grouped clustering -> global logits -> Sinkhorn(+uniform rows); local chain = prepare, scorers,
three fused products, reductions, centrality weights, row losses (single stream); and the captured step."""
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
    tm, vm = p["text_mask"].float(), p["video_mask"].float()
    btm, bvm = p["mb_mask_t"].float(), p["mb_mask_v"].float()
    hp = m._hp(0.3, 0.7, K, 3.0)
    ls = torch.tensor(100.0, device=DEV)
    with torch.no_grad():
        nz = m._draw_noise(B, Nt, Nv, torch.device(DEV))
        gt, gv = m._merge_grouped(p["text_feat"], p["video_feat"], tm, vm, nz)
        sw_t, sw_v = m.scorer_weights("text_weight_fc"), m.scorer_weights("video_weight_fc")
        rowloss = torch.empty((2, 4, B), device=DEV)

        def clustering():
            return m._merge_grouped(p["text_feat"], p["video_feat"], tm, vm, nz)

        def stage0():
            from neighborretr_amd.cluster_fused import ctm_stage_group
            ctm_stage_group([("text0", p["text_feat"], tm, m.text_ctm0, m.text_block0, nz["t0"]),
                             ("video0", p["video_feat"], vm, m.video_ctm0, m.video_block0, nz["v0"])], m._ctm_cache)

        def tail():
            G = ops.gemm_nt_f32(gt.reshape(B, -1), gv.reshape(B, -1))
            ops.sinkhorn_uniform_rows(G, 0.7, 3.0, rowloss, 50)
            ops.loss_finalize(rowloss, 1, 1, 1)

        def critical():
            a, b = clustering()
            G = ops.gemm_nt_f32(a.reshape(B, -1), b.reshape(B, -1))
            ops.sinkhorn_uniform_rows(G, 0.7, 3.0, rowloss, 50)
            ops.loss_finalize(rowloss, 1, 1, 1)

        def local_and_tail():
            head.head_forward(p["text_feat"], p["video_feat"], tm, vm, p["mb_feat_t"], p["mb_feat_v"], btm, bvm, gt, gv,
                              sw_t, sw_v, hp, ls, head.PREC_MIXED)

        m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], btm, bvm
        m.mb_ind = torch.arange(M, device=DEV)

        def step():
            m(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0)

        t_cl, t_s0, t_tail, t_crit, t_loc = (graph_time(f) for f in (clustering, stage0, tail, critical, local_and_tail))
        print(f"grouped clustering (2 stages)            : {t_cl:7.1f} us   (stage 0 alone {t_s0:6.1f})")
        print(f"tail: logits + Sinkhorn(+uniform) + final: {t_tail:7.1f} us")
        print(f"critical chain alone                     : {t_crit:7.1f} us")
        print(f"local chain + tail, one stream           : {t_loc:7.1f} us   (local alone ~ {t_loc - t_tail:6.1f})")
        print(f"captured step                            : {graph_time(step):7.1f} us")


if __name__ == "__main__":
    main()
