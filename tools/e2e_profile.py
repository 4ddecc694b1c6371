#!/usr/bin/env python
"""configs[4] on one GPU (bench.e2e_bench's step: encoders -> HIP head -> backward), a few steps for rocprofv3 --stats."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
print(bench.e2e_bench(torch.device("cuda"), steps=3))
