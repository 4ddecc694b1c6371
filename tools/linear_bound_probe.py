#!/usr/bin/env python
"""What bounds the clustering's big split-bf16 GEMM?  The stage-0 token convolution's shape as a plain product [4608, 1536] x
[1536, 512]^T through nr_linear_group, split-bf16 (three MFMA passes, hi + lo operands: 4 B per element through LDS) against one
bf16 pass (a third of the MFMAs, half the operand bytes), and the kv projection's shape [4608, 512] x [1024, 512]^T likewise."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd.cluster_backward_hip import _linear_group  # noqa: E402
from tools.branch_times import graph_time  # noqa: E402

dev = "cuda"
for name, M, N, K in (("conv shape", 4608, 512, 1536), ("kv shape", 4608, 1024, 512), ("proj shape", 896, 512, 512), ("stage-1 conv shape", 896, 512, 1536)):
    x_hi = torch.randint(-2000, 2000, (M, K), dtype=torch.int16, device=dev)
    x_lo = torch.randint(-2000, 2000, (M, K), dtype=torch.int16, device=dev)
    w_hi = torch.randint(-2000, 2000, (N, K), dtype=torch.int16, device=dev)
    w_lo = torch.randint(-2000, 2000, (N, K), dtype=torch.int16, device=dev)
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    t3 = graph_time(lambda: _linear_group([(x_hi, x_lo, w_hi, w_lo, None, None, out, M, N, K)]))
    t1 = graph_time(lambda: _linear_group([(x_hi, None, w_hi, None, None, None, out, M, N, K)]))
    fl = 2.0 * M * N * K
    print(f"{name:20s} [{M}, {K}] x [{N}, {K}]^T: split-bf16 {t3:6.1f} us ({3 * fl / t3 / 1e6:6.0f} TF/s issued)   one pass {t1:6.1f} us ({fl / t1 / 1e6:6.0f} TF/s)")
