"""GPU: the per-rank sharded step at W > 1 on the ONE card of a box.  comm.EmulatedWorld runs rank r of a W-rank
job with its own messages on a 1-rank RCCL communicator and the peers' parts pre-filled; tools/rank_local_times.py checks
every emulated rank's losses against the replicated single-rank step, and the replays of the whole-step graph (collectives
inside) and of the segmented graphs (comm.SegmentedStep: the fallback form) against the eager step -- it raises otherwise.
Runs in a child process (the RCCL process group must not leak into the test process)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_ranks_match_the_replicated_step_and_replay_as_graphs():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rank_local_times.py"), "--worlds", "2", "4", "--B", "32", "--M", "64",
                        "--K", "8", "--rank", "1", "--eval", "--train"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    print(r.stdout)
    for W, b in ((2, 16), (4, 8)):
        head = [l for l in r.stdout.splitlines() if l.startswith(f"W={W} b={b:3d}")]
        assert head and "every rank's losses == replicated step" in head[0], r.stdout
    # 5 collectives per step: packed exchange, clustering max, global tokens, centrality slices, row terms -> 6 segments
    assert len(re.findall(r"\+ 5 collectives per step", r.stdout)) == 2, r.stdout
    assert len(re.findall(r"SEGMENTED graphs \(6 segments", r.stdout)) == 2, r.stdout
    for dl in re.findall(r"replay vs eager \|dL\| ([0-9.e+-]+)", r.stdout):
        assert float(dl) <= 2e-5, r.stdout
    # the step-interleaved job: the owner's losses are the single-rank step's, bit for bit; a non-owned step is three launches
    assert len(re.findall(r"step-interleaved: the owner's losses == the single-rank step bit for bit \(max \|dL\| 0.0e\+00\)", r.stdout)) == 2
    assert len(re.findall(r"any OTHER step  \(exchange \+ push\) +: +2 C-ABI calls", r.stdout)) == 2, r.stdout
    # ... and with the owner's loss beside the following steps (modeling.OwnedSlot, interleave.OverlappedOwnedStep): the same bits
    assert len(re.findall(r"owner's loss BESIDE the following steps .*max \|dL\| 0.0e\+00\)", r.stdout)) == 2, r.stdout
    # RCCL-only branches on the 1-rank communicator: the evaluator's fp32 / int32 collectives, and the sharded TRAINING step's
    # reduce-scatters (tools/rank_local_times.py raises if any emulated rank's ranks / losses / gradients are off)
    assert "sharded evaluation, W=4 emulated ranks" in r.stdout and "== the single-rank ones" in r.stdout
    assert len(re.findall(r"sharded TRAINING step \(forward \+ backward\): losses == replicated", r.stdout)) == 2, r.stdout
