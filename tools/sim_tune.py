#!/usr/bin/env python
"""A/B timings of the fused local_level kernel under the environment hooks of a -DNR_TUNE build
(NR_EXTRA_FLAGS=-DNR_TUNE python -m neighborretr_amd.build --force): usage  sim_tune.py VAR=val[,VAR=val] ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops  # noqa: E402
from tools.sim_times import timed  # noqa: E402

DEV = "cuda"
CASES = [(128, 24, 512, 12, hip.PREC_BF16, hip.OUT_ROWSUM, "batch x bank-video bf16"),
         (512, 24, 128, 12, hip.PREC_BF16, hip.OUT_COLSUM, "bank-text x batch bf16"),
         (128, 24, 128, 12, hip.PREC_BF16X3, hip.OUT_FULL, "batch x batch x3")]


def main():
    g = torch.Generator().manual_seed(0)
    settings = [""] + sys.argv[1:]
    data = []
    for (A, Nt, Bv, Nv, prec, mode, name) in CASES:
        t = torch.randn(A, Nt, 512, generator=g).to(DEV)
        v = torch.randn(Bv, Nv, 512, generator=g).to(DEV)
        pt = ops.prepare_tokens(t, torch.ones(A, Nt, device=DEV))
        pv = ops.prepare_tokens(v, torch.ones(Bv, Nv, device=DEV))
        wt = torch.full((A, Nt), 1.0 / Nt, device=DEV)
        wv = torch.full((Bv, Nv), 1.0 / Nv, device=DEV)
        data.append((pt, pv, wt, wv))
    for rnd in range(2):                       # two rounds: the second shows the run-to-run spread
        for s in settings:
            keys = []
            for kv in filter(None, s.split(",")):
                k, v = kv.split("=")
                os.environ[k] = v
                keys.append(k)
            row = []
            for (A, Nt, Bv, Nv, prec, mode, name), (pt, pv, wt, wv) in zip(CASES, data):
                us = timed(lambda: ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec, mode), reps=30)
                row.append(us)
            for k in keys:
                del os.environ[k]
            print(f"round {rnd}  {s or '(default)':40s} " + "  ".join(f"{n}: {u:6.2f} us" for (_, _, _, _, _, _, n), u in zip(CASES, row)), flush=True)


if __name__ == "__main__":
    main()
