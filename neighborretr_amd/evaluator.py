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

from . import ops
from .metrics import RetrievalMetrics


def _world(args):
    return int(getattr(args, "world_size", 1))


def gather_eval_features(text_feat, video_feat, idx, text_mask, video_mask, args):
    """evaluator.py:173-189: gather every rank's cached features and put them back into dataset order (`idx` = dataset
    index of every local sample; duplicates from a padded last batch overwrite each other with identical rows), trimmed to
    idx.max() + 1.  -> (text_feat, video_feat, text_mask, video_mask) in dataset order, identical on every rank."""
    from .dist import packed_allgather
    with torch.no_grad():
        tf, vf, ix, tm, vm = packed_allgather(text_feat, video_feat, idx, text_mask, video_mask, args)
        n = int(ix.max().item()) + 1
        out = []
        for t in (tf, vf, tm, vm):
            dst = torch.zeros((max(n, t.shape[0]),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dst.index_copy_(0, ix, t)
            out.append(dst[:n].contiguous())
    return tuple(out)


def slab_bounds(n, world, rank):
    """Rows [r0, r1) of rank `rank`: slabs differ by at most one row."""
    base, extra = divmod(n, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def sharded_retrieval_ranks(model, text_feat, video_feat, text_mask, video_mask, args, chunk=256):
    """-> (greater_t2v, equal_t2v, greater_v2t, equal_v2t), int64 numpy arrays of length N, identical on every rank:
    for text i the number of videos scoring above / equal to its own video (metrics.py:58-66 on S), and for video j the
    number of texts scoring above / equal to its own text (the same on S.T)."""
    W = _world(args)
    rank = dist.get_rank() if (W > 1 and dist.is_initialized()) else 0
    N = text_feat.shape[0]
    if video_feat.shape[0] != N:
        raise ValueError("single-sentence retrieval: one text per video expected")
    r0, r1 = slab_bounds(N, W, rank)
    dev = text_feat.device
    old = model.precision
    model.precision = "bf16x3"                          # rank-exact path
    try:
        with torch.no_grad():
            rows = []
            for lo in range(r0, r1, chunk):                 # [chunk, N] pieces keep the kernel's outputs small
                hi = min(lo + chunk, r1)
                S, _ = model.get_similarity_logits(text_feat[lo:hi], video_feat, text_mask[lo:hi], video_mask, shaped=True)
                rows.append(S)
            S_slab = torch.cat(rows, 0) if rows else torch.empty((0, N), dtype=torch.float32, device=dev)
    finally:
        model.precision = old
    n = r1 - r0
    # the diagonal of the whole matrix: this slab's part, gathered (slabs differ by at most one row: padded to the largest)
    width = -(-N // W)
    mine = torch.full((width,), float("nan"), dtype=torch.float32, device=dev)
    if n:
        mine[:n] = S_slab[torch.arange(n, device=dev), torch.arange(r0, r1, device=dev)]
    if W > 1:
        allv = torch.empty((W * width,), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allv, mine)
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
        dist.all_reduce(cols)                              # partial column counts -> complete
        allr = torch.empty((W, 2, width), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allr.view(-1), rows_pad.view(-1))
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
