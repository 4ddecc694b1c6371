#!/usr/bin/env python
"""Phase cycles of nr_ctm_back (sample 0, thread 0) from a -DNR_STAMP build, grouped stage 0 of configs[1]."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import hip, modeling, synth  # noqa: E402
from neighborretr_amd.cluster_fused import ctm_stage_group  # noqa: E402

DEV = "cuda"
NAMES = ["start", "operands arrived (first barrier)", "distances in LDS", "densities", "centres + assignment",
         "shares; token rows landed", "merged + stored"]


def main():
    B, Nt, Nv = 128, 24, 12
    m = modeling.NeighborRetr(modeling.default_config())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(DEV).train()
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(1002, B, Nt, Nv, 64).items()}
    nz = m._draw_noise(B, Nt, Nv, torch.device(DEV))
    with torch.no_grad():
        for _ in range(3):
            ctm_stage_group([("t0", p["text_feat"], p["text_mask"].float(), m.text_ctm0, m.text_block0, nz["t0"]),
                             ("v0", p["video_feat"], p["video_mask"].float(), m.video_ctm0, m.video_block0, nz["v0"])], {})
    buf = (ctypes.c_ulonglong * 16)()
    if hasattr(hip.lib(), "nr_debug_front_stamps"):
        n = hip.lib().nr_debug_front_stamps(buf)
        prev = 0
        for i, name in zip(range(n), ["start", "own rows normalised (issue side)", "all rows (barrier)", "own distances", "end"]):
            print(f"front {buf[i]:8d}  +{buf[i] - prev:6d}  {name}")
            prev = buf[i]
    n = hip.lib().nr_debug_back_stamps(buf)
    prev = 0
    for i in range(n):
        print(f"{buf[i]:8d}  +{buf[i] - prev:6d}  {NAMES[i]}")
        prev = buf[i]


if __name__ == "__main__":
    main()
