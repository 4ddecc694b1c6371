#!/usr/bin/env python
"""Evaluation at the MSR-VTT 1k-A size (1000 captions x 1000 videos, 24 / 12 tokens): the sharded metrics path on one rank
(split-bf16 similarity in row chunks + rank counts on the GPU), ms per call."""
import os, sys, time
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import modeling, synth
from neighborretr_amd.evaluator import sharded_metrics
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = modeling.NeighborRetr(modeling.default_config())
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.cuda().eval()
t, v, tm, vm = (torch.from_numpy(a).cuda() for a in synth.make_samples(4242, "test", N, 24, 12))
args = SimpleNamespace(world_size=1)
for chunk in (256, 1000):
    import neighborretr_amd.evaluator as E
    for _ in range(3):
        t2v, v2t = E.sharded_retrieval_ranks(m, t, v, tm.float(), vm.float(), args, chunk=chunk)[:2], None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        E.sharded_retrieval_ranks(m, t, v, tm.float(), vm.float(), args, chunk=chunk)
    torch.cuda.synchronize()
    print(f"N = {N}, row chunks of {chunk}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per evaluation (similarity + rank counts)")
