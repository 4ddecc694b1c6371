#!/usr/bin/env python
"""Seeded random shapes through the SYNCHRONOUS sharded step on one card: every emulated rank of a W-rank job
(comm.EmulatedWorld, this rank's messages on a 1-rank RCCL communicator) against the replicated single-rank step on the same
gathered batch -- per-rank batches that are not multiples of the kernels' block heights (b = 1, 2, 3, 5 ...), odd token counts,
banks that are not a multiple of anything, K up to B - 3.  Raises on the first mismatch; prints one line per case.
Used by tests/test_fuzz_gpu.py::test_random_sharded_steps_match_the_replicated_step (child process: the process group must not
leak into the test process)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import rank_local_times as rlt  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rlt.init_one_rank_group()
    for seed in range(n_cases):
        r = np.random.RandomState(11000 + seed)
        W = int(r.choice([2, 3, 4, 5, 8]))
        b = int(r.choice([1, 2, 3, 4, 5, 8]))
        B = W * b
        if B < 5:
            b = 3
            B = W * b
        Nt, Nv = int(r.randint(13, 25)), int(r.randint(9, 13))
        M = int(r.choice([B, B + 3, 2 * B + 1, 40]))
        K = int(r.randint(1, B - 2))
        rlt.CFG.update(B=B, Nt=Nt, Nv=Nv, M=M, K=K)
        model, full = rlt.build(dev, precision="bf16x3" if seed % 2 else "bf16")
        rl = rlt.RankLocal(model, full, W, dev)
        sweeps = rl.settle()
        worst, _ = rl.check(tol=2e-5)
        print(f"case {seed}: W={W} b={b} (B={B}) Nt={Nt} Nv={Nv} M={M} K={K}: every rank's losses == replicated step to {worst:.1e} ({sweeps} sweeps)", flush=True)
    print(f"sharded sweep: {n_cases} cases passed")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
