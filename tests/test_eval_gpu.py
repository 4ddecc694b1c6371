"""GPU: sharded evaluation (neighborretr_amd/evaluator.py).  One process: the slab path with W = 1 equals
RetrievalMetrics on the full matrix and the reference's sort-based ranks (planted exact ties included).  Two gloo ranks
sharing the card: feature gather + index reorder (evaluator.py:173-189) and the slab / collective path give both ranks
the single-process `cols`, R@K and MedR exactly."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import modeling, ops, synth
from util import params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
N, Nt, Nv = 203, 24, 12             # not a multiple of the rank count or of any tile size


def _model():
    m = modeling.NeighborRetr(modeling.default_config())
    m.load_state_dict(params(), strict=False)
    return m.to(DEV).eval()


def _testset():
    t, v, tm, vm = synth.make_samples(4242, "test", N, Nt, Nv)
    v[17] = v[16]                   # two identical videos: exact ties in both directions
    vm[17] = vm[16]
    return tuple(torch.from_numpy(a) for a in (t, v, tm, vm))


def test_slab_ranks_equal_full_matrix_ranks():
    g = torch.Generator().manual_seed(3)
    S = torch.randn(N, N, generator=g) * 0.1 + torch.eye(N) * 0.2
    S[5, 9] = S[5, 5]
    S[9, 5] = S[5, 5]
    S = S.to(DEV)
    ref_t, ref_v = O.compute_metrics(S.cpu().numpy()), O.compute_metrics(S.t().cpu().numpy())
    for W in (1, 2, 3):
        from neighborretr_amd.evaluator import slab_bounds
        from neighborretr_amd.metrics import RetrievalMetrics
        diag = torch.diagonal(S).contiguous()
        gts, ets, gc, ec = [], [], 0, 0
        for r in range(W):
            r0, r1 = slab_bounds(N, W, r)
            a, b, c, d = ops.slab_ranks(S[r0:r1].contiguous(), r0, diag)
            gts.append(a.cpu().numpy()); ets.append(b.cpu().numpy())
            gc, ec = gc + c.cpu().numpy(), ec + d.cpu().numpy()
        t2v = RetrievalMetrics.metrics_from_ranks(RetrievalMetrics.ranks_from_counts(np.concatenate(gts), np.concatenate(ets)))
        v2t = RetrievalMetrics.metrics_from_ranks(RetrievalMetrics.ranks_from_counts(gc, ec))
        assert t2v["cols"] == ref_t["cols"] and v2t["cols"] == ref_v["cols"], W
        for k in ("R1", "R5", "R10", "R50", "MR", "MeanR"):
            assert t2v[k] == ref_t[k] and v2t[k] == ref_v[k]


def _single_process_reference():
    from neighborretr_amd.metrics import RetrievalMetrics
    m = _model()
    t, v, tm, vm = (x.to(DEV) for x in _testset())
    m.precision = "bf16x3"
    with torch.no_grad():
        S, _ = m.get_similarity_logits(t, v, tm, vm)
    return RetrievalMetrics.compute_metrics(S), RetrievalMetrics.compute_metrics(S.t().contiguous()), S


def test_sharded_metrics_single_rank_equal_full_matrix():
    from neighborretr_amd.evaluator import sharded_metrics
    ref_t, ref_v, S = _single_process_reference()
    # the HIP similarity itself ranks like the oracle's fp32 similarity (north_star: identical R@1 ordering)
    t, v, tm, vm = _testset()
    S_o, _ = O.local_level(t, v, tm, vm, params())
    assert O.compute_metrics(S_o.numpy())["cols"] == ref_t["cols"]
    m = _model()
    t2v, v2t = sharded_metrics(m, t.to(DEV), v.to(DEV), tm.to(DEV).float(), vm.to(DEV).float(), SimpleNamespace(world_size=1))
    assert t2v["cols"] == ref_t["cols"] and v2t["cols"] == ref_v["cols"]
    assert len(ref_t["cols"]) > N and max(ref_t["cols"]) > 0          # the planted tie yields an extra hit (metrics.py:58-66)


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from neighborretr_amd.evaluator import gather_eval_features, sharded_metrics
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    args = SimpleNamespace(world_size=world, local_rank=rank)
    m = _model()
    t, v, tm, vm = _testset()
    # a DistributedSampler-style split, the last batch padded with a repeated sample (its duplicate rows are identical)
    mine = torch.arange(rank, N, world)
    if len(mine) < -(-N // world):
        mine = torch.cat((mine, mine[-1:]))
    T, V, TM, VM = gather_eval_features(t[mine].to(DEV), v[mine].to(DEV), mine.to(DEV), tm[mine].to(DEV), vm[mine].to(DEV), args)
    ok_order = bool(torch.equal(T.cpu(), t) and torch.equal(V.cpu(), v) and torch.equal(TM.cpu(), tm.float()))
    t2v, v2t = sharded_metrics(m, T, V, TM, VM, args)
    torch.save({"ok_order": ok_order, "t2v": t2v, "v2t": v2t}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_eval_two_ranks_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    ref_t, ref_v, _ = _single_process_reference()
    world, port = 2, 29641
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(f"{out}.{r}", weights_only=False)
        assert res["ok_order"]
        assert res["t2v"]["cols"] == ref_t["cols"] and res["v2t"]["cols"] == ref_v["cols"]
        for k in ("R1", "R5", "R10", "R50", "MR", "MeanR"):
            assert res["t2v"][k] == ref_t[k] and res["v2t"][k] == ref_v[k]
