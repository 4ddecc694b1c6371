#!/usr/bin/env python
"""Training step (forward + backward, configs[1]) as one captured HIP graph, with the clustering forward fused or
autograd-traced and the scorer-MLP backward GEMMs on the build's engine or the library: ms per step of each."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import backward, modeling, synth  # noqa: E402

DEV = "cuda"
B, Nt, Nv, M, K = 128, 24, 12, 512, 20


def main():
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(DEV).train()
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
    m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
    m.mb_ind = torch.arange(M, device=DEV)
    tf = p["text_feat"].clone().requires_grad_(True)
    vf = p["video_feat"].clone().requires_grad_(True)

    def fb():
        m.zero_grad(set_to_none=True)
        tf.grad = vf.grad = None
        m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)[0].backward()

    for fused, own, side_streams in ((True, True, True), (True, False, True), (False, True, True), (False, False, True),
                                     (False, True, False)):
        if True:
            m.fused_training_clustering = fused
            m.use_side_streams = side_streams
            backward.OWN_MLP_GEMMS = own
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    fb()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fb()
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 10 * 1e3
            m._scorer_cache.clear(); m._ctm_cache.clear()
            g = torch.cuda.CUDAGraph()
            m.zero_grad(set_to_none=True)
            tf.grad = vf.grad = None
            with torch.cuda.graph(g):
                m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)[0].backward()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                g.replay()
            torch.cuda.synchronize()
            print(f"clustering forward {'fused HIP ' if fused else 'torch ops '} | {'side streams' if side_streams else 'ONE stream  '} | "
                  f"MLP backward GEMMs {'own split-bf16' if own else 'library      '}: "
                  f"eager {eager:6.2f} ms   graph {(time.perf_counter() - t0) / 30 * 1e3:6.2f} ms", flush=True)


if __name__ == "__main__":
    main()
