#!/usr/bin/env python
"""Un-profiled timings of the step's branches, each captured in its own HIP graph (configs[1] shape):
text clustering, video clustering, local branch (prepare/scorer/products), tail (G..finalize), bank push."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import head, hip, modeling, ops, synth  # noqa: E402

DEV = "cuda"
B, Nt, Nv, M, K = 128, 24, 12, 512, 20


def graph_time(fn, reps=200):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(10):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


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
        gt = m._merge_one("text", p["text_feat"], tm, nz["t0"], nz["t1"])
        gv = m._merge_one("video", p["video_feat"], vm, nz["v0"], nz["v1"])
        sw_t, sw_v = m.scorer_weights("text_weight_fc"), m.scorer_weights("video_weight_fc")

        def text():
            m._merge_one("text", p["text_feat"], tm, nz["t0"], nz["t1"])

        def video():
            m._merge_one("video", p["video_feat"], vm, nz["v0"], nz["v1"])

        def whole_head():
            head.head_forward(p["text_feat"], p["video_feat"], tm, vm, p["mb_feat_t"], p["mb_feat_v"], btm, bvm, gt, gv,
                              sw_t, sw_v, hp, ls, head.PREC_MIXED)

        def tail_only():
            G = ops.gemm_nt_f32(gt.reshape(B, -1), gv.reshape(B, -1))
            tr, tc = ops.sinkhorn_targets(G, 0.7, 50)
            v = torch.zeros(B, device=DEV)
            rl = ops.row_losses(G * 0.001, G, tr, tc, v, v, v + 1, v + 1, ls.reshape(1), K, 3.0)
            ops.loss_finalize(rl, 1, 1, 1)

        m.use_side_streams = False
        m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], btm, bvm
        m.mb_ind = torch.arange(M, device=DEV)

        def step_serial():
            m(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0)

        print(f"text clustering     : {graph_time(text):8.1f} us")
        print(f"video clustering    : {graph_time(video):8.1f} us")
        print(f"head (local + tail) : {graph_time(whole_head):8.1f} us")
        print(f"tail only           : {graph_time(tail_only):8.1f} us")
        print(f"whole step, 1 stream: {graph_time(step_serial):8.1f} us")
        m.use_side_streams = True
        m.group_clustering = False
        m.bank_side_streams = True
        print(f"whole step, text | video | local streams, bank beside Sinkhorn: {graph_time(step_serial):8.1f} us")
        m.group_clustering = True
        m.bank_side_streams = False
        print(f"whole step, grouped clustering | local, bank serial : {graph_time(step_serial):8.1f} us")
        m.bank_side_streams = True
        print(f"whole step, grouped clustering | local, bank beside Sinkhorn: {graph_time(step_serial):8.1f} us")
        for e in (1, 2):
            m.bank_early = e
            print(f"   ... {e} bank chain(s) beside the clustering instead      : {graph_time(step_serial):8.1f} us")
        m.bank_early = 2
        for name, order in (("7|9|7|rest", ((7, 9), (7, 1 << 30))), ("1:1", ((1, 1),)), ("1:2", ((1, 2),)), ("2:3", ((2, 3),)),
                            ("3|4 then 1:2", ((3, 4), (1, 2))), ("7|all", ((7, 1 << 30),)), ("all|all", ((1 << 30, 1 << 30),))):
            m.capture_order = order
            print(f"   capture order {name:14s}: {graph_time(step_serial):8.1f} us")
        m.bank_early = 0


if __name__ == "__main__":
    main()
