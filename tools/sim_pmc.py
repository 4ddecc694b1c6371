#!/usr/bin/env python
"""Driver for rocprofv3 --pmc passes over the fused local_level kernel: the step's three products (configs[1]),
each launched eagerly N times on random data.  Run as  rocprofv3 --kernel-trace --pmc ... -- python3 tools/sim_pmc.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops  # noqa: E402

DEV = "cuda"
N = int(os.environ.get("NR_PMC_LAUNCHES", "20"))
CASES = [(128, 24, 512, 12, hip.PREC_BF16, hip.OUT_ROWSUM), (512, 24, 128, 12, hip.PREC_BF16, hip.OUT_COLSUM),
         (128, 24, 128, 12, hip.PREC_BF16X3, hip.OUT_FULL)]


def main():
    g = torch.Generator().manual_seed(0)
    pair = []
    for (A, Nt, Bv, Nv, prec, mode) in CASES:
        t = torch.randn(A, Nt, 512, generator=g).to(DEV)
        v = torch.randn(Bv, Nv, 512, generator=g).to(DEV)
        pt = ops.prepare_tokens(t, torch.ones(A, Nt, device=DEV))
        pv = ops.prepare_tokens(v, torch.ones(Bv, Nv, device=DEV))
        wt = torch.full((A, Nt), 1.0 / Nt, device=DEV)
        wv = torch.full((Bv, Nv), 1.0 / Nv, device=DEV)
        for _ in range(N):
            ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec, mode)
        torch.cuda.synchronize()
        if prec == hip.PREC_BF16:
            pair.append((pt, pv, wt, wv, A, Nt, Bv, Nv, prec, mode))
    for _ in range(N):                       # the two bank products as one launch of chained tile pairs (nr_sim_pair_kernel)
        ops.local_level_group(pair)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
