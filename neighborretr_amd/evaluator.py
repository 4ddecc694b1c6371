"""Evaluation of the retrieval head with the N x N similarity SHARDED over the ranks.

Reference: training/evaluator.py -- every rank gathers all test features (5 all_gathers, :173-177), scatters them back
into dataset order (:180-189), then EVERY rank computes the whole N x N matrix in 64 x 64 tiles with a device-to-host
copy per tile (:21-63) and ranks it with a NumPy sort (utils/metrics.py:58-66).

Here: one packed all-gather (neighborretr_amd.dist) + one index scatter; rank r then runs the fused local_level kernel
(split-bf16: rank-exact) on ITS row slab only -- texts [r n/W, (r+1) n/W) against all videos, 1/W of the work --, counts
the text->video ranks of its rows and its partial video->text column counts on the GPU (nr_slab_ranks), and three tiny
collectives (diagonal N floats, row counts 2n ints, column counts 2N ints) give every rank the same R@K as the
reference's sort.  Nothing but 4N integers ever leaves the device.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import comm

from . import ops
from .metrics import RetrievalMetrics


def _world(args):
    return int(getattr(args, "world_size", 1))


def rank_sample_indices(n, world, rank):
    """Dataset indices rank `rank` extracts features for: rank, rank + W, ... PADDED to ceil(n / W) entries by wrapping around
    to the start of the dataset -- torch's DistributedSampler(shuffle=False), which the reference's test loader uses
    (dataloaders/data_dataloaders.py).  Every rank therefore hands the SAME number of rows to the packed all-gather (ranks
    issuing a collective with different byte counts hang RCCL); the duplicates overwrite identical rows in dataset_order."""
    import math
    per = math.ceil(n / world)
    return (torch.arange(rank, rank + per * world, world) % n)[:per]


def gather_eval_features(text_feat, video_feat, idx, text_mask, video_mask, args):
    """evaluator.py:173-189: gather every rank's cached features and put them back into dataset order (`idx` = dataset
    index of every local sample; duplicates from a padded last batch overwrite each other with identical rows), trimmed to
    idx.max() + 1.  -> (text_feat, video_feat, text_mask, video_mask) in dataset order, identical on every rank."""
    from .dist import packed_allgather
    with torch.no_grad():
        tf, vf, ix, tm, vm = packed_allgather(text_feat, video_feat, idx, text_mask, video_mask, args)
    return dataset_order(tf, vf, ix, tm, vm)


def dataset_order(text_feat, video_feat, idx, text_mask, video_mask):
    """evaluator.py:180-189: row idx[k] of every output <- row k of the input (later duplicates win), trimmed to
    idx.max() + 1; positions no sample maps to stay zero."""
    with torch.no_grad():
        idx = idx.reshape(-1).long()
        n = int(idx.max().item()) + 1
        out = []
        for t in (text_feat, video_feat, text_mask, video_mask):
            dst = torch.zeros((max(n, t.shape[0]),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dst.index_copy_(0, idx, t)
            out.append(dst[:n].contiguous())
    return tuple(out)


def slab_bounds(n, world, rank):
    """Rows [r0, r1) of rank `rank`: slabs differ by at most one row."""
    base, extra = divmod(n, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def _slab_similarity(model, text_feat, video_feat, text_mask, video_mask, r0, r1, chunk=256):
    """Rows [r0, r1) of the text x video similarity, on the rank-exact (split-bf16) path of the fused kernel."""
    old = model.precision
    model.precision = "bf16x3"
    try:
        with torch.no_grad():
            rows = []
            for lo in range(r0, r1, chunk):                 # [chunk, N] pieces keep the kernel's outputs small
                hi = min(lo + chunk, r1)
                S, _ = model.get_similarity_logits(text_feat[lo:hi], video_feat, text_mask[lo:hi], video_mask, shaped=True)
                rows.append(S)
            if rows:
                return torch.cat(rows, 0)
            return torch.empty((0, video_feat.shape[0]), dtype=torch.float32, device=text_feat.device)
    finally:
        model.precision = old


def sharded_retrieval_ranks(model, text_feat, video_feat, text_mask, video_mask, args, chunk=256):
    """-> (greater_t2v, equal_t2v, greater_v2t, equal_v2t), int64 numpy arrays of length N, identical on every rank:
    for text i the number of videos scoring above / equal to its own video (metrics.py:58-66 on S), and for video j the
    number of texts scoring above / equal to its own text (the same on S.T)."""
    W = _world(args)
    rank = comm.get_rank() if W > 1 else 0
    N = text_feat.shape[0]
    if video_feat.shape[0] != N:
        raise ValueError("single-sentence retrieval: one text per video expected")
    r0, r1 = slab_bounds(N, W, rank)
    dev = text_feat.device
    S_slab = _slab_similarity(model, text_feat, video_feat, text_mask, video_mask, r0, r1, chunk)
    n = r1 - r0
    # the diagonal of the whole matrix: this slab's part, gathered (slabs differ by at most one row: padded to the largest)
    width = -(-N // W)
    mine = torch.full((width,), float("nan"), dtype=torch.float32, device=dev)
    if n:
        mine[:n] = S_slab[torch.arange(n, device=dev), torch.arange(r0, r1, device=dev)]
    if W > 1:
        allv = torch.empty((W * width,), dtype=torch.float32, device=dev)
        comm.all_gather_into_tensor(allv, mine)
        diag = torch.cat([allv[r * width: r * width + (slab_bounds(N, W, r)[1] - slab_bounds(N, W, r)[0])] for r in range(W)])
    else:
        diag = mine[:n]
    diag = diag.contiguous()
    if n:
        g_rows, e_rows, g_cols, e_cols = ops.slab_ranks(S_slab, r0, diag)
    else:
        z = lambda k: torch.zeros((k,), dtype=torch.int32, device=dev)      # noqa: E731
        g_rows, e_rows, g_cols, e_cols = z(0), z(0), z(N), z(N)
    cols = torch.stack((g_cols, e_cols))
    rows_pad = torch.zeros((2, width), dtype=torch.int32, device=dev)
    rows_pad[0, :n], rows_pad[1, :n] = g_rows, e_rows
    if W > 1:
        comm.all_reduce(cols)                              # partial column counts -> complete
        allr = torch.empty((W, 2, width), dtype=torch.int32, device=dev)
        comm.all_gather_into_tensor(allr.view(-1), rows_pad.view(-1))
    else:
        allr = rows_pad[None]
    allr = allr.cpu().numpy()
    gt = np.concatenate([allr[r, 0, :slab_bounds(N, W, r)[1] - slab_bounds(N, W, r)[0]] for r in range(W)])
    et = np.concatenate([allr[r, 1, :slab_bounds(N, W, r)[1] - slab_bounds(N, W, r)[0]] for r in range(W)])
    cols = cols.cpu().numpy()
    return gt.astype(np.int64), et.astype(np.int64), cols[0].astype(np.int64), cols[1].astype(np.int64)


def sharded_metrics(model, text_feat, video_feat, text_mask, video_mask, args):
    """(text->video metrics, video->text metrics) as RetrievalMetrics.compute_metrics(S) / (S.T) would give them."""
    gt, et, gv, ev = sharded_retrieval_ranks(model, text_feat, video_feat, text_mask, video_mask, args)
    t2v = RetrievalMetrics.metrics_from_ranks(RetrievalMetrics.ranks_from_counts(gt, et))
    v2t = RetrievalMetrics.metrics_from_ranks(RetrievalMetrics.ranks_from_counts(gv, ev))
    return t2v, v2t


def sharded_multi_sentence_metrics(model, text_feat, video_feat, text_mask, video_mask, cut_off_points, args, chunk=256):
    """Several captions per video (evaluator.py:114-149 features, :225-262 metrics): text_feat [Ns,...] holds every
    sentence in dataset order, video_feat [V,...] one entry per video, cut_off_points[g] = index of the LAST sentence of
    video g (the dataset's cut_off_points minus one, evaluator.py:98).  -> (text->video, video->text) metric dictionaries.

    The reference pads the Ns x V matrix to [V, max_sentences, V] with -inf on the host and ranks it with two argsorts
    (metrics.py:82-126) and a max over the padded axis (:128-148).  Here rank r scores its slab of sentence rows, one
    launch (nr_group_slab_ranks) gives every row's rank and the slab's per-video best scores, and a MAX all-reduce of
    the V x V best-score matrix + an all-gather of the Ns ranks complete them; no padded tensor exists."""
    W = _world(args)
    rank = comm.get_rank() if W > 1 else 0
    Ns, V = text_feat.shape[0], video_feat.shape[0]
    ends = np.asarray(cut_off_points, dtype=np.int64) + 1
    if len(ends) != V or (np.diff(ends) <= 0).any() or ends[0] <= 0 or ends[-1] != Ns:
        raise ValueError(f"cut_off_points must give {V} non-empty, increasing sentence groups ending at {Ns - 1}")
    dev = text_feat.device
    group_end = torch.from_numpy(ends.astype(np.int32)).to(dev)
    r0, r1 = slab_bounds(Ns, W, rank)
    n = r1 - r0
    S_slab = _slab_similarity(model, text_feat, video_feat, text_mask, video_mask, r0, r1, chunk)
    width = -(-Ns // W)
    mine = torch.zeros((width,), dtype=torch.int32, device=dev)
    if n:
        greater, equal_before, gmax = ops.group_slab_ranks(S_slab, r0, group_end)
        mine[:n] = torch.where(greater < 0, greater, greater + equal_before)
    else:
        gmax = torch.full((V, V), float("-inf"), dtype=torch.float32, device=dev)
    if W > 1:
        comm.all_reduce(gmax, op="max")
        allr = torch.empty((W, width), dtype=torch.int32, device=dev)
        comm.all_gather_into_tensor(allr.view(-1), mine)
        ranks = torch.cat([allr[r, :slab_bounds(Ns, W, r)[1] - slab_bounds(Ns, W, r)[0]] for r in range(W)])
    else:
        ranks = mine[:n]
    t2v = RetrievalMetrics.multi_sentence_metrics_from_ranks(ranks[ranks >= 0])      # < 0: own score NaN / inf, not ranked
    v2t = RetrievalMetrics.compute_metrics(gmax.T.contiguous())          # [video, caption group], metrics.py:146-148
    return t2v, v2t
