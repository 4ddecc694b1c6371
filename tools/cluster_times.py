#!/usr/bin/env python
"""Clustering alone (configs[1] shape, or B samples: cluster_times.py B), as HIP graphs: per-modality fused path vs the
grouped stage."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import modeling, synth  # noqa: E402
from neighborretr_amd.cluster_fused import ctm_stage_group  # noqa: E402
from tools.branch_times import graph_time  # noqa: E402

DEV = "cuda"
B, Nt, Nv, M, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 128), 24, 12, 512, 20


def main():
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(DEV).train()
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
    tm, vm = p["text_mask"].float(), p["video_mask"].float()
    with torch.no_grad():
        nz = m._draw_noise(B, Nt, Nv, torch.device(DEV))
        cache = {}

        def text():
            m._merge_one("text", p["text_feat"], tm, nz["t0"], nz["t1"])

        def video():
            m._merge_one("video", p["video_feat"], vm, nz["v0"], nz["v1"])

        def stage0():
            return ctm_stage_group([("t0", p["text_feat"], tm, m.text_ctm0, m.text_block0, nz["t0"]),
                                    ("v0", p["video_feat"], vm, m.video_ctm0, m.video_block0, nz["v0"])], cache)

        t0, v0 = stage0()

        def stage1():
            return ctm_stage_group([("t1", t0, None, m.text_ctm1, m.text_block1, nz["t1"]),
                                    ("v1", v0, None, m.video_ctm1, m.video_block1, nz["v1"])], cache)

        def both():
            a, b = stage0()
            ctm_stage_group([("t1", a, None, m.text_ctm1, m.text_block1, nz["t1"]),
                             ("v1", b, None, m.video_ctm1, m.video_block1, nz["v1"])], cache)

        def text_only_grouped():
            (a,) = ctm_stage_group([("t0", p["text_feat"], tm, m.text_ctm0, m.text_block0, nz["t0"])], cache)
            ctm_stage_group([("t1", a, None, m.text_ctm1, m.text_block1, nz["t1"])], cache)

        print(f"text, per-modality path  : {graph_time(text):8.1f} us")
        print(f"video, per-modality path : {graph_time(video):8.1f} us")
        print(f"grouped stage 0 (t+v)    : {graph_time(stage0):8.1f} us")
        print(f"grouped stage 1 (t+v)    : {graph_time(stage1):8.1f} us")
        print(f"grouped both stages (t+v): {graph_time(both):8.1f} us")
        print(f"grouped, text only       : {graph_time(text_only_grouped):8.1f} us")


if __name__ == "__main__":
    main()
