#!/usr/bin/env python
"""The k=3 token convolution launch (nr_linear_group, conv_n > 0) at the clustering stages' shapes: result against an fp64
convolution of the same split-bf16 operands, and time alone as a graph of 20 launches (the round-5 A/B of a "taps in the block's
columns" kernel against the shipped three-segment loop used this probe: docs/LAB_NOTEBOOK.md, round-5 addendum)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd.cluster_backward_hip import _linear_group  # noqa: E402
from tools.branch_times import graph_time  # noqa: E402

dev = "cuda"


def split(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi.view(torch.int16), lo.view(torch.int16), (hi.double() + lo.double())


def main():
    C = 512
    g = torch.Generator().manual_seed(5)
    for name, sets in (("stage 0, B=128 (24 + 12 tokens)", [(128, 24), (128, 12)]), ("stage 1, B=128 (4 + 3)", [(128, 4), (128, 3)]),
                       ("stage 0, B=16", [(16, 24), (16, 12)]), ("stage 0, B=1024", [(1024, 24), (1024, 12)]),
                       ("ragged: 5 x 24 + 7 x 12", [(5, 24), (7, 12)]), ("20 + 9 tokens (three-segment loop)", [(32, 20), (32, 9)])):
        probs, refs = [], []
        for B, n in sets:
            x = torch.randn(B * n, C, generator=g).to(dev)
            w = (torch.randn(C, 3 * C, generator=g) * 0.03).to(dev)
            bias = torch.randn(C, generator=g).to(dev)
            xh, xl, xd = split(x)
            wh, wl, wd = split(w)
            out = torch.zeros(B * n, C, device=dev)
            probs.append((xh, xl, wh, wl, bias, x, out, B * n, C, 3 * C, 0, n))
            xs = xd.view(B, n, C)
            z = torch.zeros(B, 1, C, dtype=torch.float64, device=dev)
            cat = torch.cat([torch.cat([z, xs[:, :-1]], 1), xs, torch.cat([xs[:, 1:], z], 1)], 2).view(B * n, 3 * C)
            refs.append(cat @ wd.t() + bias.double() + x.double())
        _linear_group(probs)
        torch.cuda.synchronize()
        err = max(float((p[6].double() - r).abs().max() / r.abs().max()) for p, r in zip(probs, refs))
        t = graph_time(lambda: _linear_group(probs))
        fl = sum(2.0 * p[7] * p[8] * p[9] for p in probs)
        print(f"{name:38s} max rel err {err:.2e}   {t:6.1f} us   {3 * fl / t / 1e6:6.0f} TF/s issued")
        assert err < 3e-6, err


if __name__ == "__main__":
    main()
