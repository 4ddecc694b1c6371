#!/usr/bin/env python
"""Times of the fused local_level kernel on the step's three products and on an evaluation-size product."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import hip, ops  # noqa: E402

DEV = "cuda"
INNER = 10        # calls per captured graph: a replay costs ~8 us of its own


def timed(fn, reps=50):
    """us per call, replayed from a HIP graph (an eager loop is host-bound below ~15 us per call)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(INNER):
            fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / INNER


def main():
    g = torch.Generator().manual_seed(0)
    for (A, Nt, Bv, Nv, prec, mode, name) in [
            (128, 24, 128, 12, hip.PREC_BF16X3, hip.OUT_FULL, "batch x batch, split-bf16"),
            (128, 24, 512, 12, hip.PREC_BF16, hip.OUT_ROWSUM, "batch x bank-video, bf16"),
            (512, 24, 128, 12, hip.PREC_BF16, hip.OUT_COLSUM, "bank-text x batch, bf16"),
            (1000, 24, 1000, 12, hip.PREC_BF16X3, hip.OUT_FULL, "eval 1k x 1k, split-bf16"),
            (1024, 24, 512, 12, hip.PREC_BF16, hip.OUT_ROWSUM, "B=1024 x bank, bf16"),
            (128, 64, 1024, 64, hip.PREC_BF16, hip.OUT_ROWSUM, "ActivityNet 128 x 1024 bank, bf16"),
            (128, 64, 128, 64, hip.PREC_BF16X3, hip.OUT_FULL, "ActivityNet batch x batch, split-bf16"),
            (128, 24, 512, 12, hip.PREC_BF16, hip.OUT_COLSUM, "128 x 512 colsum"),
            (512, 24, 128, 12, hip.PREC_BF16, hip.OUT_ROWSUM, "512 x 128 rowsum"),
            (128, 24, 512, 12, hip.PREC_BF16, hip.OUT_FULL, "128 x 512 full"),
            (256, 24, 256, 12, hip.PREC_BF16, hip.OUT_FULL, "256 x 256 full"),
            (64, 24, 1024, 12, hip.PREC_BF16, hip.OUT_FULL, "64 x 1024 full"),
            (1024, 24, 64, 12, hip.PREC_BF16, hip.OUT_FULL, "1024 x 64 full")]:
        t = torch.randn(A, Nt, 512, generator=g).to(DEV)
        v = torch.randn(Bv, Nv, 512, generator=g).to(DEV)
        pt = ops.prepare_tokens(t, torch.ones(A, Nt, device=DEV))
        pv = ops.prepare_tokens(v, torch.ones(Bv, Nv, device=DEV))
        wt = torch.full((A, Nt), 1.0 / Nt, device=DEV)
        wv = torch.full((Bv, Nv), 1.0 / Nv, device=DEV)
        us = timed(lambda: ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec, mode))
        flops = 2.0 * A * Nt * Bv * Nv * 512 * (3 if prec == hip.PREC_BF16X3 else 1)
        print(f"{name:32s} {us:8.1f} us  {flops / us / 1e6:8.1f} TFLOP/s (MFMA flops issued)  tiles {hip.local_level_tiles(A, Nt, Bv, Nv, prec)}")


if __name__ == "__main__":
    main()
