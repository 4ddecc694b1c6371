"""GPU: the callers either side of the loss head with the reference's signatures (neighborretr_amd/training.py):
MemoryBankManager (memory_bank.py:22-260), eval_epoch for single- and multi-sentence test sets (evaluator.py:66-291),
train_epoch (trainer.py:18-221); and the multi-sentence rank kernel against the reference's golden vector and the oracle."""
import logging
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import modeling, ops, synth, training
from neighborretr_amd.metrics import RetrievalMetrics
from util import params

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Nt, Nv = 24, 12


class Loader:
    """What the reference's functions need of a DataLoader: len(), iteration over 6-tuples, .dataset."""

    def __init__(self, batches, dataset=None):
        self.batches, self.dataset = batches, dataset

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def _model(**over):
    m = modeling.NeighborRetr(modeling.default_config(**over))
    m.load_state_dict(params(), strict=False)
    return m.to(DEV)


def _batches(t, v, tm, vm, order, bs):
    out = []
    for lo in range(0, len(order), bs):
        ix = order[lo:lo + bs]
        out.append((t[ix], tm[ix].long(), v[ix], vm[ix].long(), ix.clone(), ix.clone()))
    return out


def _args(**over):
    logger = logging.getLogger("test_training")
    return SimpleNamespace(world_size=1, rank=0, local_rank=0, logger=logger, n_display=2, epochs=1, batch_size=16,
                           mb_batch=3, distributed=False, **over)


def test_group_slab_ranks_against_reference_vector_and_oracle():
    g = np.load(os.path.join(GOLD, "multi_sentence.npz"))
    S, cut = g["S"], g["cut_off_points"].tolist()
    group_end = torch.tensor([c + 1 for c in cut], dtype=torch.int32, device=DEV)
    St = torch.from_numpy(S).to(DEV)
    for W in (1, 2, 3):
        from neighborretr_amd.evaluator import slab_bounds
        ranks, gmax = [], None
        for r in range(W):
            r0, r1 = slab_bounds(S.shape[0], W, r)
            gr, eb, gm = ops.group_slab_ranks(St[r0:r1].contiguous(), r0, group_end)
            ranks.append(torch.where(gr < 0, gr, gr + eb))
            gmax = gm if gmax is None else torch.maximum(gmax, gm)
        ranks = torch.cat(ranks)
        assert int((ranks < 0).sum()) == 1                              # the sentence whose own score is NaN
        t2v = RetrievalMetrics.multi_sentence_metrics_from_ranks(ranks[ranks >= 0])
        keys_t = ("R1", "R5", "R10", "R50", "MedianR", "MeanR", "Std_Rank", "MR")
        assert np.allclose([t2v[k] for k in keys_t], g["t2v"], rtol=1e-6), W
        v2t = RetrievalMetrics.compute_metrics(gmax.T.contiguous())
        assert v2t["cols"] == g["v2t_cols"].tolist(), W
    # exact ties inside a row: the stable order (lower video index first) of the oracle
    T = S.copy()
    T[np.isnan(T)] = 0.0
    ends = np.array(cut) + 1
    grp = np.searchsorted(ends, np.arange(T.shape[0]), side="right")
    for i in range(0, T.shape[0], 5):
        T[i, (grp[i] + 3) % T.shape[1]] = T[i, grp[i]]
    gr, eb, gm = ops.group_slab_ranks(torch.from_numpy(T).to(DEV), 0, group_end)
    ref_t, ref_v = O.multi_sentence_metrics(T, cut)
    t2v = RetrievalMetrics.multi_sentence_metrics_from_ranks(gr + eb)
    for k in ("R1", "R5", "R10", "R50", "MedianR", "MeanR", "Std_Rank"):
        assert abs(t2v[k] - ref_t[k]) < 1e-5 * max(1.0, abs(ref_t[k])), k
    assert RetrievalMetrics.compute_metrics(gm.T.contiguous())["cols"] == ref_v["cols"]


def test_eval_epoch_single_sentence_restores_dataset_order():
    N = 150
    t, v, tm, vm = (torch.from_numpy(a) for a in synth.make_samples(91, "test", N, Nt, Nv))
    m = _model().eval()
    m.precision = "bf16x3"
    with torch.no_grad():
        S, _ = m.get_similarity_logits(t.to(DEV), v.to(DEV), tm.to(DEV), vm.to(DEV))
    ref_t, ref_v = O.compute_metrics(S.cpu().numpy()), O.compute_metrics(S.T.cpu().numpy())
    order = torch.randperm(N, generator=torch.Generator().manual_seed(5))       # a shuffled loader
    t2v, v2t = training.eval_epoch(_args(), _model(), Loader(_batches(t, v, tm, vm, order, 32)), torch.device(DEV))
    assert t2v["cols"] == ref_t["cols"] and v2t["cols"] == ref_v["cols"]
    for k in ("R1", "R5", "R10", "R50", "MR", "MeanR"):
        assert t2v[k] == ref_t[k] and v2t[k] == ref_v[k]


def test_eval_epoch_multi_sentence_equals_the_padded_tensor_path():
    V = 41
    sizes = 1 + (np.arange(V) * 3) % 4
    ends = np.cumsum(sizes)
    Ns = int(ends[-1])
    grp = np.searchsorted(ends, np.arange(Ns), side="right")
    t, _, tm, _ = (torch.from_numpy(a) for a in synth.make_samples(92, "test", Ns, Nt, Nv))
    _, v, _, vm = (torch.from_numpy(a) for a in synth.make_samples(93, "test", V, Nt, Nv))
    t = t + 0.4 * v[grp].mean(1, keepdim=True)                                  # captions lean towards their video
    dataset = SimpleNamespace(multi_sentence_per_video=True, cut_off_points=ends.tolist(), sentence_num=Ns, video_num=V)
    # the loader repeats the video for every sentence (dataloader_msvd_retrieval.py); evaluator.py:137-149 keeps one each
    order = torch.arange(Ns)
    batches = _batches(t, v[grp], tm, vm[grp], order, 16)
    t2v, v2t = training.eval_epoch(_args(), _model(), Loader(batches, dataset), torch.device(DEV))
    m = _model().eval()
    m.precision = "bf16x3"
    with torch.no_grad():
        S, _ = m.get_similarity_logits(t.to(DEV), v.to(DEV), tm.to(DEV), vm.to(DEV))
    ref_t, ref_v = O.multi_sentence_metrics(S.cpu().numpy(), (ends - 1).tolist())
    for k in ("R1", "R5", "R10", "R50", "MedianR", "MeanR", "Std_Rank"):
        assert abs(t2v[k] - ref_t[k]) < 1e-5 * max(1.0, abs(ref_t[k])), k
    assert v2t["cols"] == ref_v["cols"]
    # and the product's own padded-tensor functions (the reference's host path) agree with the sharded one
    padded = torch.from_numpy(O.pad_sentence_groups(S.cpu().numpy(), (ends - 1).tolist()))
    host_t = RetrievalMetrics.tensor_text_to_video_metrics(padded)
    assert all(abs(host_t[k] - t2v[k]) < 1e-6 for k in host_t)
    assert RetrievalMetrics.compute_metrics(RetrievalMetrics.tensor_video_to_text_sim(padded).contiguous())["cols"] == v2t["cols"]


def _multi_problem():
    V = 41
    sizes = 1 + (np.arange(V) * 3) % 4
    ends = np.cumsum(sizes)
    Ns = int(ends[-1])
    grp = np.searchsorted(ends, np.arange(Ns), side="right")
    t, _, tm, _ = (torch.from_numpy(a) for a in synth.make_samples(92, "test", Ns, Nt, Nv))
    _, v, _, vm = (torch.from_numpy(a) for a in synth.make_samples(93, "test", V, Nt, Nv))
    t = t + 0.4 * v[grp].mean(1, keepdim=True)
    dataset = SimpleNamespace(multi_sentence_per_video=True, cut_off_points=ends.tolist(), sentence_num=Ns, video_num=V)
    return Loader(_batches(t, v[grp], tm, vm[grp], torch.arange(Ns), 16), dataset)


def _eval_worker(rank, world, port, out_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    args = _args()
    args.world_size, args.rank, args.local_rank, args.distributed = world, rank, rank, True
    N = 150
    t, v, tm, vm = (torch.from_numpy(a) for a in synth.make_samples(91, "test", N, Nt, Nv))
    mine = torch.arange(rank, N, world)                                        # a DistributedSampler's split
    single = training.eval_epoch(args, _model(), Loader(_batches(t, v, tm, vm, mine, 32)), torch.device(DEV))
    multi = training.eval_epoch(args, _model(), _multi_problem(), torch.device(DEV))
    torch.save({"single": single, "multi": multi}, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_eval_epoch_two_ranks_equal_one_rank(tmp_path):
    import torch.multiprocessing as mp
    N = 150
    t, v, tm, vm = (torch.from_numpy(a) for a in synth.make_samples(91, "test", N, Nt, Nv))
    ref_single = training.eval_epoch(_args(), _model(), Loader(_batches(t, v, tm, vm, torch.arange(N), 32)), torch.device(DEV))
    ref_multi = training.eval_epoch(_args(), _model(), _multi_problem(), torch.device(DEV))
    out = str(tmp_path / "res")
    mp.spawn(_eval_worker, args=(2, 29653, out), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(f"{out}.{r}", weights_only=False)
        for got, ref in ((res["single"], ref_single), (res["multi"], ref_multi)):
            for d in (0, 1):
                assert got[d] == ref[d], (r, d)


def test_memory_bank_manager_and_train_epoch():
    N, B = 96, 16
    t, v, tm, vm = (torch.from_numpy(a) for a in synth.make_samples(94, "train", N, Nt, Nv))
    args = _args()
    model = _model(mb_batch=3, batch_size=B, num_neighbors=6)
    loader = Loader(_batches(t, v, tm, vm, torch.arange(N), B))
    manager = training.MemoryBankManager(args)
    n = manager.load_memory_bank(model, loader, torch.device(DEV), epoch=1)
    assert n == 3 * B and model.mb_batch == 3 * B
    assert torch.equal(model.mb_ind.cpu(), torch.arange(3 * B))
    assert torch.equal(model.mb_feat_t.cpu(), t[:3 * B]) and torch.equal(model.mb_feat_v.cpu(), v[:3 * B])
    optimizer = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4)
    before = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
    val = Loader(_batches(t[:40], v[:40], tm[:40], vm[:40], torch.arange(40), 20))
    total, step, best_t, best_v = training.train_epoch(1, args, model, loader, torch.device(DEV), 1, optimizer, None, 0,
                                                       len(loader), val)
    assert step == len(loader) and np.isfinite(total) and total > 0
    assert best_t is not None and best_v is not None and 0 <= best_t["R1"] <= 100
    assert sum(int(not torch.equal(before[k], p.detach())) for k, p in model.named_parameters() if k in before) > 10
    # the bank is a FIFO of 3 batches: after 6 pushes it holds the last three batches, newest first (modeling.py:237-249)
    assert model.mb_ind.cpu().tolist() == list(range(80, 96)) + list(range(64, 80)) + list(range(48, 64))
    manager.clear_memory_bank(model)
    assert model.mb_batch == 0 and model.mb_feat_t.numel() == 0
