"""The loss of the TRAINING step sharded over the ranks (reference sketch: AllGather2, until_module.py:391-412).

The reference replicates: after the all-gather every rank evaluates the full B x B problem plus both B x M bank
products, and gradients need no reduction because every rank differentiates the whole loss.  At global B = 1024
(BASELINE configs[2]) that is 677 GF per step repeated W times.  Here rank r owns the samples [r b, (r+1) b):

  heavy, on the HIP kernels, ONE autograd node (ShardedLocalFn: the grouped kernels of the replicated training step --
  one prepare launch per token set, fused scorer + softmax, nr_local_level_bwd_group / nr_pool_weight_bwd_group for the
  four products' gradients, the fused scorer backward -- 30 forward+backward launches instead of four LocalLevelFn nodes):
      S[r-slab, :]  = local_level(text_r, video_all)        [b, B]     rows of the text->video direction
      S[:, r-slab]  = local_level(text_all, video_r)        [B, b]     rows of the video->text direction
      text_r x bank-video [b, M],  bank-text x video_r [M, b]          1/W of both bank products -> centrality slices
  sharded by samples as well: the token clustering of the rank's b samples (fused HIP forward, hand-derived backward).  Its
      masked stage uses the maximum distance over the WHOLE gathered batch (cluster.py:473-475): the ranks all-reduce that
      one number per modality between the stage's front and back kernels; the [b, d] global tokens are all-gathered
  row-local, on the HIP row-loss kernels (SlabRowLossFn: nr_row_losses_fwd_slab / nr_row_losses_bwd_slab -- the same per-row
      code as the replicated loss): the four loss terms of the rank's 2 b rows (until_module.py:56-359);
  light, replicated, ONE more node (ShardedGlobalFn): global logits G = gt gv^T (exact-fp32 MFMA), Sinkhorn targets (no
      gradient), centrality weights, the slab row terms -- nr_gemm_nt_f32, nr_centrality_weights_pair, nr_row_losses_fwd_slab
      forward; nr_row_losses_bwd_slab, nr_centrality_weights_bwd_pair, nr_global_logits_bwd backward.
      (tests/slab_terms_torch.py restates the row terms in torch ops: the cross-check of tests/test_sharded_gpu.py.)

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


def _rows(prep, first, count):
    return ops.Prepared(prep.hi[first:first + count], prep.lo[first:first + count] if prep.lo is not None else None,
                        prep.norm[first:first + count], None, count, prep.d)


class ShardedLocalFn(torch.autograd.Function):
    """The token side of rank r's share of the loss: from the gathered features to its two slabs of S, its slices of the two
    bank centrality vectors and the means of the normalised tokens (modeling.py:483-514 x4, :403-424).  Forward and backward on
    the grouped HIP kernels of the replicated training step (backward._local_backward is the B x B form of the same)."""

    @staticmethod
    def forward(ctx, model, masks, r0, b, text_feat, video_feat, mb_feat_t, mb_feat_v, w1t, b1t, w2t, b2t, w1v, b1v, w2v, b2v):
        from . import head, hip
        text_mask, video_mask, mb_mask_t, mb_mask_v = masks
        B, Nt, d = text_feat.shape
        Nv, M = video_feat.shape[1], mb_feat_v.shape[0]
        prec = model._prec()
        p_bb, p_mlp, p_bank = head.precision_plan(prec)
        sw_t, sw_v = model.scorer_weights("text_weight_fc"), model.scorer_weights("video_weight_fc")
        tf, vf = text_feat.detach(), video_feat.detach()
        pt, pv = ops.prepare_tokens_pair(tf, text_mask, vf, video_mask, want_lo=True, want_colsum=True)
        pbt, pbv = ops.prepare_tokens_pair(mb_feat_t, mb_mask_t, mb_feat_v, mb_mask_v, want_lo=True)
        w_t, _ = head.token_weights(pt, text_mask, sw_t, B, Nt, p_mlp)
        w_v, _ = head.token_weights(pv, video_mask, sw_v, B, Nv, p_mlp)
        w_bt, _ = head.token_weights(pbt, mb_mask_t, sw_t, M, Nt, p_bank)
        w_bv, _ = head.token_weights(pbv, mb_mask_v, sw_v, M, Nv, p_bank)
        pt_r, pv_r = _rows(pt, r0 * Nt, b * Nt), _rows(pv, r0 * Nv, b * Nv)
        w_t_r, w_v_r = w_t[r0:r0 + b].contiguous(), w_v[r0:r0 + b].contiguous()
        S_rows, aux_r = ops.local_level(pt_r, pv, w_t_r, w_v, b, Nt, B, Nv, p_bb, hip.OUT_FULL, True)         # S[r0:r0+b, :]
        S_cols, aux_c = ops.local_level(pt, pv_r, w_t, w_v_r, B, Nt, b, Nv, p_bb, hip.OUT_FULL, True)         # S[:, r0:r0+b]
        p1, aux1 = ops.local_level(pt_r, pbv, w_t_r, w_bv, b, Nt, M, Nv, p_bank, hip.OUT_ROWSUM, True)        # text_r x bank-video
        p0, aux2 = ops.local_level(pbt, pv_r, w_bt, w_v_r, M, Nt, b, Nv, p_bank, hip.OUT_COLSUM, True)        # bank-text x video_r
        c1_mine, c0_mine = ops.reduce_parts(p1, 1.0 / M), ops.reduce_parts(p0, 1.0 / M)
        mean_t, mean_v = ops.colsum_pair(pt.colsum, 1.0 / pt.n_tok, pv.colsum, 1.0 / pv.n_tok)
        ctx.st = dict(pt=pt, pv=pv, pbt=pbt, pbv=pbv, pt_r=pt_r, pv_r=pv_r, w_t=w_t, w_v=w_v, w_bt=w_bt, w_bv=w_bv, w_t_r=w_t_r,
                      w_v_r=w_v_r, aux=(aux_r, aux_c, aux1, aux2), masks=masks, r0=int(r0), b=int(b), exact=prec == hip.PREC_BF16X3,
                      plan=(p_bb, p_mlp, p_bank), model=model)
        ctx.save_for_backward(text_feat, video_feat, mb_feat_t, mb_feat_v)
        ctx.set_materialize_grads(False)
        return S_rows, S_cols, c1_mine, c0_mine, mean_t, mean_v

    @staticmethod
    def backward(ctx, dS_rows, dS_cols, d_c1, d_c0, dmean_t, dmean_v):
        from . import hip
        from .backward import _mlp_backward_hip
        st = ctx.st
        if st is None:
            raise RuntimeError("ShardedLocalFn: backward a second time (its saved state is released by the first)")
        text_feat, video_feat, mb_feat_t, mb_feat_v = ctx.saved_tensors
        text_mask, video_mask, mb_mask_t, mb_mask_v = st["masks"]
        B, Nt, d = text_feat.shape
        Nv, M = video_feat.shape[1], mb_feat_v.shape[0]
        r0, b, lo = st["r0"], st["b"], st["exact"]
        aux_r, aux_c, aux1, aux2 = st["aux"]
        dev = text_feat.device
        f32 = dict(dtype=torch.float32, device=dev)
        z = lambda *shape: torch.zeros(shape, **f32)       # noqa: E731 -- a slab that brought no gradient
        dS_rows = z(b, B) if dS_rows is None else dS_rows.float().contiguous()
        dS_cols = z(B, b) if dS_cols is None else dS_cols.float().contiguous()
        d_c1 = z(b) if d_c1 is None else d_c1.float().contiguous()
        d_c0 = z(b) if d_c0 is None else d_c0.float().contiguous()
        pt, pv, pbt, pbv, pt_r, pv_r = st["pt"], st["pv"], st["pbt"], st["pbv"], st["pt_r"], st["pv_r"]
        if not (ops.USE_MFMA_BACKWARD and bool(hip.lib().nr_local_level_bwd_mfma_supported(Nt, Nv, d))):
            raise hip.NrHipError("the sharded training loss needs the matrix-core similarity backward (token counts outside "
                                 "nr_local_level_bwd_mfma_supported)")
        # ---- token gradients of the four products: two grouped launches.  A gradient that only a row range of the tokens
        # receives (the rank's b texts / videos) is accumulated in a buffer of its own and added to the full one afterwards.
        d_tn, d_vn = torch.empty((B * Nt, d), **f32), torch.empty((B * Nv, d), **f32)
        d_tn_r, d_vn_r = torch.empty((b * Nt, d), **f32), torch.empty((b * Nv, d), **f32)
        T_pv, T_pv_r, T_pbv, T_pt_r = ops.transpose_prepared([pv, pv_r, pbv, pt_r], use_lo=lo)
        T_pt, T_pbt = ops.transpose_prepared([pt, pbt], use_lo=lo)
        ops.local_level_bwd_group([
            dict(side=0, dS=dS_cols, ds_mode=0, ds_scale=1.0, other_T=T_pv_r, w_self=st["w_t"], w_other=st["w_v_r"], aux=aux_c,
                 A=B, Nt=Nt, Bv=b, Nv=Nv, d_x=d_tn),
            dict(side=0, dS=dS_rows, ds_mode=0, ds_scale=1.0, other_T=T_pv, w_self=st["w_t_r"], w_other=st["w_v"], aux=aux_r,
                 A=b, Nt=Nt, Bv=B, Nv=Nv, d_x=d_tn_r),
            dict(side=0, dS=d_c1, ds_mode=1, ds_scale=1.0 / M, other_T=T_pbv, w_self=st["w_t_r"], w_other=st["w_bv"], aux=aux1,
                 A=b, Nt=Nt, Bv=M, Nv=Nv, d_x=d_tn_r),
            dict(side=1, dS=dS_rows, ds_mode=0, ds_scale=1.0, other_T=T_pt_r, w_self=st["w_v"], w_other=st["w_t_r"], aux=aux_r,
                 A=b, Nt=Nt, Bv=B, Nv=Nv, d_x=d_vn)], use_lo=lo)
        ops.local_level_bwd_group([
            dict(side=1, dS=dS_cols, ds_mode=0, ds_scale=1.0, other_T=T_pt, w_self=st["w_v_r"], w_other=st["w_t"], aux=aux_c,
                 A=B, Nt=Nt, Bv=b, Nv=Nv, d_x=d_vn_r),
            dict(side=1, dS=d_c0, ds_mode=2, ds_scale=1.0 / M, other_T=T_pbt, w_self=st["w_v_r"], w_other=st["w_bt"], aux=aux2,
                 A=M, Nt=Nt, Bv=b, Nv=Nv, d_x=d_vn_r)], use_lo=lo)
        d_tn[r0 * Nt:(r0 + b) * Nt] += d_tn_r
        d_vn[r0 * Nv:(r0 + b) * Nv] += d_vn_r
        # ---- token-weight gradients, all six sums in one launch
        d_wt, d_wv = torch.empty((B * Nt,), **f32), torch.empty((B * Nv,), **f32)
        d_wt_r, d_wv_r = torch.empty((b * Nt,), **f32), torch.empty((b * Nv,), **f32)
        d_wbt, d_wbv = torch.empty((M * Nt,), **f32), torch.empty((M * Nv,), **f32)
        ops.pool_weight_bwd_group([
            dict(side=0, N=Nt, d_w=d_wt, srcs=[(dS_cols, 0, 1.0, aux_c[2], B, b)]),
            dict(side=0, N=Nt, d_w=d_wt_r, srcs=[(dS_rows, 0, 1.0, aux_r[2], b, B), (d_c1, 1, 1.0 / M, aux1[2], b, M)]),
            dict(side=1, N=Nv, d_w=d_wv, srcs=[(dS_rows, 0, 1.0, aux_r[3], b, B)]),
            dict(side=1, N=Nv, d_w=d_wv_r, srcs=[(dS_cols, 0, 1.0, aux_c[3], B, b), (d_c0, 2, 1.0 / M, aux2[3], M, b)]),
            dict(side=1, N=Nv, d_w=d_wbv, srcs=[(d_c1, 1, 1.0 / M, aux1[3], b, M)]),
            dict(side=0, N=Nt, d_w=d_wbt, srcs=[(d_c0, 2, 1.0 / M, aux2[2], M, b)])])
        d_wt[r0 * Nt:(r0 + b) * Nt] += d_wt_r
        d_wv[r0 * Nv:(r0 + b) * Nv] += d_wv_r
        # ---- normalise / mask / centrality-mean backward, softmax backward, fused scorer backward
        d_text = ops.normalize_bwd(text_feat, pt.norm, text_mask, d_tn, dmean_t)
        d_video = ops.normalize_bwd(video_feat, pv.norm, video_mask, d_vn, dmean_v)
        dl_t = ops.token_softmax_bwd(st["w_t"], d_wt.view(B, Nt))
        dl_v = ops.token_softmax_bwd(st["w_v"], d_wv.view(B, Nv))
        dl_bt = ops.token_softmax_bwd(st["w_bt"], d_wbt.view(M, Nt))
        dl_bv = ops.token_softmax_bwd(st["w_bv"], d_wbv.view(M, Nv))
        _, p_mlp, p_bank = st["plan"]
        model = st["model"]
        (dW1t, db1t, dW2t, db2t, d_text), (dW1v, db1v, dW2v, db2v, d_video) = _mlp_backward_hip([
            dict(sw=model.scorer_weights("text_weight_fc"), add_to=d_text,
                 sets=[(pt, text_feat, dl_t, p_mlp), (pbt, mb_feat_t, dl_bt, p_bank)]),
            dict(sw=model.scorer_weights("video_weight_fc"), add_to=d_video,
                 sets=[(pv, video_feat, dl_v, p_mlp), (pbv, mb_feat_v, dl_bv, p_bank)])])
        ctx.st = None
        return (None, None, None, None, d_text.view(text_feat.shape), d_video.view(video_feat.shape), None, None,
                dW1t, db1t, dW2t, db2t, dW1v, db1v, dW2v, db2v)


class ShardedGlobalFn(torch.autograd.Function):
    """The rest of rank r's share: global logits, Sinkhorn targets, centrality weights and the four row terms of its 2 b rows
    (modeling.py:403-444, until_module.py:56-359) -> the PARTIAL [5] losses (sum over ranks = the reference's values).  Forward
    and backward on the HIP kernels (the slab forms of backward._global_backward)."""

    @staticmethod
    def forward(ctx, hp, r0, S_rows, S_cols, c0, c1, mean_t, mean_v, gt, gv, logit_scale):
        b, B = S_rows.shape
        args = [t.detach().float().contiguous() for t in (S_rows, S_cols, c0, c1, mean_t, mean_v, gt, gv)]
        S_rows, S_cols, c0, c1, mean_t, mean_v, gt2, gv2 = args
        ls = logit_scale.detach().float().reshape(1).contiguous()
        G = ops.gemm_nt_f32(gt2, gv2)
        tgt_r, tgt_c = ops.sinkhorn_targets(G, hp["beta"], 50)
        wc_t, wc_v, cw_aux = ops.centrality_weights_pair(gt2, gv2, mean_t, mean_v, hp["centrality_scale"], True)
        K, T = int(hp["num_neighbors"]), float(hp["temperature"])
        rowloss = ops.row_losses_slab(S_rows, S_cols, r0, G, tgt_r, tgt_c, c0, c1, wc_t, wc_v, ls, K, T)
        ctx.save_for_backward(S_rows, S_cols, c0, c1, mean_t, mean_v, gt2, gv2, ls, G, tgt_r, tgt_c, wc_t, wc_v, *cw_aux)
        ctx.hp, ctx.r0, ctx.shapes = dict(hp), int(r0), (gt.shape, gv.shape)
        return ops.loss_finalize(rowloss, hp["uniform_weight"], hp["neighbor_weight"], hp["kl_weight"])

    @staticmethod
    def backward(ctx, g):
        from .backward import _coef_rowloss
        S_rows, S_cols, c0, c1, mean_t, mean_v, gt2, gv2, ls, G, tgt_r, tgt_c, wc_t, wc_v, gn_t, gn_v, wtok_t, wtok_v = ctx.saved_tensors
        hp, r0 = ctx.hp, ctx.r0
        b, B = S_rows.shape
        coef = _coef_rowloss(g.float().contiguous(), hp, B)
        dS, dG_dir, dC, dwc, dls = ops.row_losses_bwd_slab(S_rows, S_cols, r0, G, tgt_r, tgt_c, c0, c1, wc_t, wc_v, ls,
                                                           int(hp["num_neighbors"]), float(hp["temperature"]), coef)
        dG = torch.zeros_like(G)
        dG[r0:r0 + b] += dG_dir[0]                       # direction 0 read rows of G, direction 1 columns
        dG[:, r0:r0 + b] += dG_dir[1].t()
        dw_t, dw_v = torch.zeros_like(wc_t), torch.zeros_like(wc_v)
        dw_t[r0:r0 + b], dw_v[r0:r0 + b] = dwc[0], dwc[1]
        dg_t, dmean_t, dg_v, dmean_v = ops.centrality_weights_bwd_pair(gt2, gn_t, mean_t, wtok_t, dw_t, gv2, gn_v, mean_v, wtok_v, dw_v,
                                                                       hp["centrality_scale"])
        d_gt, d_gv = ops.global_logits_bwd(dG, gt2, gv2, dg_t, dg_v)
        return (None, None, dS[0], dS[1].t().contiguous(), dC[0].sum(0), dC[1].sum(0), dmean_t, dmean_v,
                d_gt.reshape(ctx.shapes[0]), d_gv.reshape(ctx.shapes[1]), dls.sum().reshape(()))


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
        gg = _GatherCat.apply(torch.stack((gt_l.reshape(b, d).float(), gv_l.reshape(b, d).float()), 1), rank, world)   # [B, 2, d]
        gt, gv = gg[:, 0], gg[:, 1]
    else:
        gt, gv = model.merge_global_features(text_feat, video_feat, text_mask, video_mask, noise)
        if gt.shape[1] != 1 or gv.shape[1] != 1:
            raise RuntimeError("the sharded training loss covers one global token per sample (until_module.py:321)")
        gt, gv = gt.reshape(B, d).float(), gv.reshape(B, d).float()
    # ---- the rank's slabs of S, its slices of the bank centralities and the token means: one node on the grouped HIP kernels
    from .functional import _mlp_params
    S_rows, S_cols, c1_mine, c0_mine, mean_t, mean_v = ShardedLocalFn.apply(
        model, (text_mask, video_mask, mb_mask_t, mb_mask_v), r0, b, text_feat, video_feat, mb_feat_t, mb_feat_v,
        *_mlp_params(model, "text_weight_fc"), *_mlp_params(model, "video_weight_fc"))
    cc = _GatherCat.apply(torch.stack((c0_mine, c1_mine), 1), rank, world)       # [B, 2]: one collective for both vectors
    c0, c1 = cc[:, 0], cc[:, 1]          # video / text centralities (used by the t2v / v2t neighbour loss, modeling.py:393-398)
    # ---- global logits, Sinkhorn targets (no gradient), centrality weights, the four terms on this rank's rows: one node
    part = ShardedGlobalFn.apply(hp, r0, S_rows, S_cols, c0, c1, mean_t, mean_v, gt, gv, logit_scale.reshape(()).float())
    full = part.detach().clone()
    comm.all_reduce(full)                                                              # the reference's (full) values
    # value: full;  gradient: W x this rank's share (DDP's mean over ranks then yields the full-loss gradient)
    return full + (_ScaleGrad.apply(part, float(world)) - part.detach())
