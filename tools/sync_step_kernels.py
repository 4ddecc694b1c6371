#!/usr/bin/env python
"""The synchronous sharded step of ONE rank (emulated world, tools/rank_local_times.RankLocal) replayed as one graph -- for a
kernel trace: rocprofv3 --kernel-trace --stats -- python3 tools/sync_step_kernels.py [B [W]] (bash tools/sync_step_profile.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import rank_local_times as rlt  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rlt.CFG.update(B=B)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    rlt.init_one_rank_group()
    model, full = rlt.build(dev)
    rl = rlt.RankLocal(model, full, W, dev)
    rl.settle()
    c = rl.world.comm(0)
    rl.configure(0)
    g, _ = rlt.capture(lambda: rl.step(0, c))
    print(f"B={B} W={W}: whole sharded step as one graph {rlt.replay_time(g.replay, reps=100, repeats=2):.1f} us")
    torch.cuda.synchronize()
    import torch.distributed as dist
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
