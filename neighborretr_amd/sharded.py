"""The loss of the TRAINING step sharded over the ranks (reference sketch: AllGather2, until_module.py:391-412).

The reference replicates: after the all-gather every rank evaluates the full B x B problem plus both B x M bank
products, and gradients need no reduction because every rank differentiates the whole loss.  At global B = 1024
(BASELINE configs[2]) that is 677 GF per step repeated W times.  Here rank r owns the samples [r b, (r+1) b):

  heavy, on the HIP kernels (differentiable: functional.local_level_sim -> LocalLevelFn):
      S[r-slab, :]  = local_level(text_r, video_all)        [b, B]     rows of the text->video direction
      S[:, r-slab]  = local_level(text_all, video_r)        [B, b]     rows of the video->text direction
      text_r x bank-video [b, M],  bank-text x video_r [M, b]          1/W of both bank products -> centrality slices
  sharded by samples as well: the token clustering of the rank's b samples (fused HIP forward, hand-derived backward).  Its
      masked stage uses the maximum distance over the WHOLE gathered batch (cluster.py:473-475): the ranks all-reduce that
      one number per modality between the stage's front and back kernels; the [b, d] global tokens are all-gathered
  row-local, on the HIP row-loss kernels (SlabRowLossFn: nr_row_losses_fwd_slab / nr_row_losses_bwd_slab -- the same per-row
      code as the replicated loss): the four loss terms of the rank's 2 b rows (until_module.py:56-359);
  light, replicated, in a few torch ops: global logits G = gt gv^T, Sinkhorn targets (HIP kernel, no gradient), centrality
      weights.  (`_direction_terms` / `_neighbor_rows` below restate the row terms in torch ops: the cross-check of the tests.)

Each rank's L_r is ITS rows' share of every term, so sum_r L_r = L (the reference's loss).  Cross-rank values enter
through differentiable collectives whose backward is the matching reduction (all-gather <-> reduce-scatter(sum),
all-reduce <-> all-reduce), so every rank's autograd graph yields dL_r/dtheta and dL_r/dX for ALL gathered samples.
DDP averages parameter gradients over ranks, so L_r is differentiated as W L_r (`_ScaleGrad`): mean_r(W dL_r) = dL,
the reference's effective gradient; the exchange step's backward then AVERAGES the feature gradients over the ranks
(dist.PackedAllGather with args.shard_loss: reduce-scatter / W), which gives every rank dL/dX of its own samples.
Returned values are the FULL losses on every rank (one all-reduce of five numbers), with the partial gradient attached.
"""
import torch

from . import comm, ops
from .functional import local_level_sim

NEG_BIG = -9e15


class _GatherCat(torch.autograd.Function):
    """all-gather + cat on dim 0; backward: every rank's gradient of the gathered tensor summed, this rank's slice."""

    @staticmethod
    def forward(ctx, x, rank, world):
        ctx.rank, ctx.world, ctx.n = rank, world, x.shape[0]
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        comm.all_gather_into_tensor(out, x)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if comm.backend() == "gloo":                          # no reduce_scatter in gloo
            g = g.clone()
            comm.all_reduce(g)
            return g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n], None, None
        out = torch.empty((ctx.n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        comm.reduce_scatter_tensor(out, g)
        return out, None, None


class _AllReduceSum(torch.autograd.Function):
    """y = sum over ranks of x (every rank gets y); backward: sum over ranks of the upstream gradients."""

    @staticmethod
    def forward(ctx, x):
        y = x.clone()
        comm.all_reduce(y)
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        comm.all_reduce(g)
        return g


class _ScaleGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def _neighbor_rows(S, c, K, T, diag_col):
    """until_module.py:161-211 for the rows of a slab: S [n,B] (row k's own sample sits in column diag_col[k]), c [B]
    bank centralities.  Returns the per-row loss [n]."""
    n, B = S.shape
    cols = torch.arange(B, device=S.device)[None, :]
    is_diag = cols == diag_col[:, None]
    s_off = torch.where(is_diag, torch.full_like(S, NEG_BIG), S.detach())
    idx = torch.sort(s_off, dim=-1, descending=True, stable=True)[1][:, :K]          # :100-129
    nb = torch.zeros_like(S, dtype=torch.bool).scatter_(1, idx, True)
    ext = nb | is_diag
    rest = ~ext

    def minmax(X):                                                                   # :65-86
        lo = torch.where(rest, X, torch.full_like(X, 9e15)).min(-1, keepdim=True)[0]
        hi = torch.where(rest, X, torch.full_like(X, -9e15)).max(-1, keepdim=True)[0]
        return (X - lo) / (hi - lo)
    ns = minmax(S)
    nc = minmax(c[None, :].expand(n, -1))
    adj = torch.where(nb, ns - nc, torch.full_like(S, NEG_BIG))                       # :189-193
    p = torch.softmax(adj * T, dim=-1)                                                # :147
    p = torch.where(nb, p, torch.zeros_like(p))
    p = torch.where(is_diag, torch.ones_like(p), p)                                   # fill_diagonal_(1) :157
    masked = torch.where(ext, S, torch.full_like(S, NEG_BIG))                         # :199-203
    lp = torch.log_softmax(masked, dim=-1) * p
    return -lp.sum(-1) / p.sum(-1)                                                    # :206-207


def _direction_terms(S, G, tgt, c, w, ls, K, T, diag_col):
    """The four row terms (summed over the slab's rows) of one direction: S, G, tgt [n,B]; w [n] centrality weights."""
    rows = torch.arange(S.shape[0], device=S.device)
    lp_c = torch.log_softmax(S * ls, dim=-1)[rows, diag_col]                           # until_module.py:315-327
    cent = -(lp_c * w).sum()
    unif = -(torch.log_softmax(G * T, dim=-1) * tgt).sum()                             # :285-289
    p = torch.softmax(S, dim=-1)
    kl = (p * (torch.log_softmax(S, dim=-1) - torch.log_softmax(G, dim=-1))).sum()     # :351-357
    neigh = _neighbor_rows(S, c, K, T, diag_col).sum()
    return cent, unif, neigh, kl


class SlabRowLossFn(torch.autograd.Function):
    """This rank's share of the four loss terms -- its rows of either direction, from the two slabs of S it owns -- on the HIP
    row-loss kernels (nr_row_losses_fwd_slab / nr_row_losses_bwd_slab: the same per-row code as the replicated loss, top-K
    by wave arg-max rounds, ties to the lowest column).  Returns the PARTIAL [5] losses (total, centrality, uniform,
    neighbour, kl) of the rank's rows, normalised like the full loss (sum over ranks = the reference's values)."""

    @staticmethod
    def forward(ctx, S_rows, S_cols, G, tgt_r, tgt_c, c0, c1, wc_text, wc_video, logit_scale, hp, row0):
        args = [t.detach().float().contiguous() for t in (S_rows, S_cols, G, tgt_r, tgt_c, c0, c1, wc_text, wc_video)]
        ls = logit_scale.detach().float().reshape(1).contiguous()
        K, T = int(hp["num_neighbors"]), float(hp["temperature"])
        rowloss = ops.row_losses_slab(args[0], args[1], row0, *args[2:], ls, K, T)
        ctx.save_for_backward(*args, ls)
        ctx.hp, ctx.row0 = dict(hp), int(row0)
        return ops.loss_finalize(rowloss, hp["uniform_weight"], hp["neighbor_weight"], hp["kl_weight"])

    @staticmethod
    def backward(ctx, g):
        from .backward import _coef_rowloss
        S_rows, S_cols, G, tgt_r, tgt_c, c0, c1, wc_text, wc_video, ls = ctx.saved_tensors
        hp, r0 = ctx.hp, ctx.row0
        n, B = S_rows.shape
        coef = _coef_rowloss(g.float().contiguous(), hp, B)
        dS, dG_dir, dC, dwc, dls = ops.row_losses_bwd_slab(S_rows, S_cols, r0, G, tgt_r, tgt_c, c0, c1, wc_text, wc_video, ls,
                                                           int(hp["num_neighbors"]), float(hp["temperature"]), coef)
        dG = torch.zeros_like(G)
        dG[r0:r0 + n] += dG_dir[0]                       # direction 0 read rows of G, direction 1 columns
        dG[:, r0:r0 + n] += dG_dir[1].t()
        d_wt, d_wv = torch.zeros_like(wc_text), torch.zeros_like(wc_video)
        d_wt[r0:r0 + n], d_wv[r0:r0 + n] = dwc[0], dwc[1]
        return (dS[0], dS[1].t().contiguous(), dG, None, None, dC[0].sum(0), dC[1].sum(0), d_wt, d_wv,
                dls.sum().reshape(1), None, None)


def sharded_training_losses(model, text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                            hp, logit_scale, rank, world, noise=None):
    """-> [5] tensor (total, centrality, uniform, neighbour, kl): full values, this rank's share of the gradient (x W)."""
    B, Nt, d = text_feat.shape
    M = mb_feat_v.shape[0]
    K, T = int(hp["num_neighbors"]), float(hp["temperature"])
    if B % world:
        raise ValueError("the gathered batch must divide over the ranks")
    if K > B:
        raise ValueError(f"num_neighbors={K} > batch={B}")
    b, r0 = B // world, rank * (B // world)
    sl = slice(r0, r0 + b)
    text_mask, video_mask, mb_mask_t, mb_mask_v = (m if m.dtype == torch.float32 else m.float()
                                                   for m in (text_mask, video_mask, mb_mask_t, mb_mask_v))
    # ---- token clustering: every rank clusters ITS samples (fused forward + hand-derived backward; the batch-wide maximum
    # distance of the masked stage is exchanged inside, modeling._merge_sharded) and the global tokens are gathered with a
    # differentiable all-gather: d gt of a rank's samples = the sum of every rank's W dL_r / d gt, so the clustering
    # parameters receive W x (the contribution of this rank's samples) and DDP's mean over the ranks is the full gradient.
    # `model.shard_clustering = False`: the replicated form (every rank differentiates the whole clustering).
    if model.shard_clustering and text_feat.is_cuda and text_feat.shape[1] <= 64 and video_feat.shape[1] <= 64 and d % 128 == 0:
        gt_l, gv_l = model._merge_sharded(text_feat, video_feat, text_mask, video_mask, noise or {}, rank, world)
        if gt_l.shape[1] != 1 or gv_l.shape[1] != 1:
            raise RuntimeError("the sharded training loss covers one global token per sample (until_module.py:321)")
        gt = _GatherCat.apply(gt_l.reshape(b, d).float(), rank, world)
        gv = _GatherCat.apply(gv_l.reshape(b, d).float(), rank, world)
    else:
        gt, gv = model.merge_global_features(text_feat, video_feat, text_mask, video_mask, noise)
        if gt.shape[1] != 1 or gv.shape[1] != 1:
            raise RuntimeError("the sharded training loss covers one global token per sample (until_module.py:321)")
        gt, gv = gt.reshape(B, d).float(), gv.reshape(B, d).float()
    # ---- the rank's slabs of S and its slices of the bank centralities (HIP kernels, differentiable)
    S_rows = local_level_sim(model, text_feat[sl], video_feat, text_mask[sl], video_mask)           # [b, B]
    S_cols = local_level_sim(model, text_feat, video_feat[sl], text_mask, video_mask[sl])           # [B, b]
    bank_t2v = local_level_sim(model, text_feat[sl], mb_feat_v, text_mask[sl], mb_mask_v)           # [b, M]
    bank_v2t = local_level_sim(model, mb_feat_t, video_feat[sl], mb_mask_t, video_mask[sl])         # [M, b]
    c1 = _GatherCat.apply(bank_t2v.sum(-1) / M, rank, world)        # [B] text centralities (used by the v2t neighbour loss)
    c0 = _GatherCat.apply(bank_v2t.sum(0) / M, rank, world)         # [B] video centralities (used by the t2v neighbour loss)
    # ---- global logits (un-normalised, modeling.py:526-537 with one global token) and Sinkhorn targets (no gradient)
    G = gt @ gv.t()
    tgt_r, tgt_c = ops.sinkhorn_targets(G.detach().contiguous(), hp["beta"], 50)
    # ---- centrality weights: exp(c <g_hat_i, mean of ALL normalised tokens>), padding included (modeling.py:403-430)
    tn = torch.nn.functional.normalize(text_feat[sl].reshape(-1, d), dim=-1).sum(0)
    vn = torch.nn.functional.normalize(video_feat[sl].reshape(-1, d), dim=-1).sum(0)
    mean_t = _AllReduceSum.apply(tn) / (B * Nt)
    mean_v = _AllReduceSum.apply(vn) / (B * video_feat.shape[1])
    w_text = torch.exp(hp["centrality_scale"] * (torch.nn.functional.normalize(gt[sl], dim=-1) @ mean_t))      # [b]
    w_video = torch.exp(hp["centrality_scale"] * (torch.nn.functional.normalize(gv[sl], dim=-1) @ mean_v))
    # ---- the four terms on this rank's rows of either direction: HIP row-loss kernels on the slabs (forward and backward)
    wt_full = torch.zeros((B,), dtype=torch.float32, device=text_feat.device).index_add(0, torch.arange(r0, r0 + b, device=text_feat.device), w_text)
    wv_full = torch.zeros((B,), dtype=torch.float32, device=text_feat.device).index_add(0, torch.arange(r0, r0 + b, device=text_feat.device), w_video)
    part = SlabRowLossFn.apply(S_rows, S_cols, G, tgt_r, tgt_c, c0, c1, wt_full, wv_full, logit_scale.reshape(1).float(), hp, r0)
    full = part.detach().clone()
    comm.all_reduce(full)                                                              # the reference's (full) values
    # value: full;  gradient: W x this rank's share (DDP's mean over ranks then yields the full-loss gradient)
    return full + (_ScaleGrad.apply(part, float(world)) - part.detach())
