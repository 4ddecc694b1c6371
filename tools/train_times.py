#!/usr/bin/env python
"""Training step (forward + backward, configs[1]) launched eagerly and replayed from one captured HIP graph, by how its two
heaviest backward parts run: the token clustering (HIP forward + HIP backward | HIP forward + hand-derived torch-op backward |
autograd-traced torch ops) and the scorer MLP (fused HIP backward | torch-op form with GEMMs on the tile engine): ms per step."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import backward, cluster_fused, modeling, synth  # noqa: E402

DEV = "cuda"
B, Nt, Nv, M, K = 128, 24, 12, 512, 20


def main(which):
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

    settings = (("clustering HIP fwd + HIP bwd      | scorer bwd fused HIP", True, True, True),
                ("  same, head as ONE autograd node, clustering on the step's stream (no overlap of the two backward chains)", True, True, True),
                ("clustering HIP fwd + HIP bwd      | scorer bwd torch-op form", True, True, False),
                ("clustering HIP fwd + torch-op bwd | scorer bwd fused HIP", True, False, True),
                ("clustering autograd-traced torch  | scorer bwd fused HIP", False, True, True),
                ("clustering autograd-traced torch  | scorer bwd torch-op form (round 2)", False, True, False))
    # one setting per process: a process that captures several training graphs with DIFFERENT stream topologies on the same
    # side-stream objects segfaults inside the ROCm 7.2 runtime at the fourth capture (each setting alone captures fine:
    # tools/traced_capture_check.py) -- the parent starts a child per setting and never touches the GPU itself
    for name, fused, hip_bwd, mlp_hip in settings[which:which + 1]:
        if True:
            if which == 1:
                backward.SPLIT_HEAD_NODES = False
                m.cluster_side_stream = False
            m.fused_training_clustering = fused
            cluster_fused.HIP_BACKWARD = hip_bwd
            backward.FUSED_MLP_BACKWARD = mlp_hip
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
            print(f"{name:75s}: eager {eager:6.2f} ms   graph {(time.perf_counter() - t0) / 30 * 1e3:6.2f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(int(sys.argv[1]))
    else:
        import subprocess
        for k in range(6):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], capture_output=True, text=True, timeout=600)
            out = [l for l in r.stdout.splitlines() if " ms " in l]
            print(out[-1] if out else f"setting {k}: exit code {r.returncode} {r.stderr.strip().splitlines()[-1:]}", flush=True)
