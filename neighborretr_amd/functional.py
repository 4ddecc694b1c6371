"""Differentiable entry points of the HIP loss head (torch.autograd.Function wrappers).

    head_losses(...)      fused _compute_losses  (modeling.py:314-360 minus the clustering stage)
    local_level_sim(...)  local_level / get_similarity_logits (modeling.py:483-514, :625-632)
    global_level_sim(...) global_level (modeling.py:516-539)
    row_loss_terms(...)   the raw [2,4,B] per-row loss terms (used by the until_module classes)

Forward and backward both run on the HIP kernels: the token-scorer MLP backward on the grouped split-bf16 GEMMs, the
clustering stage's backward on nr_ctm_bwd.hip (cluster_backward_hip; widths above 512 channels: the same arithmetic as torch ops
on the GPU, cluster_backward.py).  The only library GEMMs left on the training path are the tiny dG products of the one-token
global logits.
"""
import torch

from . import head, hip, ops


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)


def _mlp_params(model, name):
    m = getattr(model, name)
    return m[0].weight, m[0].bias, m[2].weight, m[2].bias


def head_losses(model, text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                gt, gv, hp, logit_scale, cluster=None):
    """Returns a [5] tensor (total, centrality, uniform, neighbour, kl)."""
    pt_params = _mlp_params(model, "text_weight_fc")
    pv_params = _mlp_params(model, "video_weight_fc")
    if not torch.is_tensor(logit_scale):
        logit_scale = torch.tensor(float(logit_scale), device=text_feat.device)
    if cluster is not None or (gt is not None and _needs_grad(text_feat, video_feat, gt, gv, logit_scale, *pt_params, *pv_params)):
        from .backward import head_loss_nodes
        return head_loss_nodes(model, hp, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                               text_feat, video_feat, gt, gv, logit_scale, (*pt_params, *pv_params),
                               (*_mlp_params(model, "text_weight_fc1"), *_mlp_params(model, "video_weight_fc1")), cluster=cluster)
    losses, _ = head.head_forward(text_feat, video_feat, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t,
                                  mb_mask_v, gt, gv, model.scorer_weights("text_weight_fc"),
                                  model.scorer_weights("video_weight_fc"), hp, logit_scale, model._prec(),
                                  join=model._take_join(), bank_streams=model._bank_streams(text_feat.device),
                                  local_stream=model._local_stream(text_feat.device) if gt is None else None,
                                  bank_early=model.bank_early, capture_order=model.capture_order, bb_late=getattr(model, "bb_late", False),
                                  bank_prepared=model._bank_shadow(mb_feat_t, mb_feat_v), prepared_out=model._last_prepared,
                                  bank_push=getattr(model, "_push_fn", None), slot=model._slot(), pipeline=model._pipeline_for_head(text_feat.device),
                                  local_masks=getattr(model, "_raw_masks", None),
                                  **model._global_scorers(text_feat, video_feat))
    return losses


def local_level_sim(model, text_feat, video_feat, text_mask, video_mask):
    pt_params = _mlp_params(model, "text_weight_fc")
    pv_params = _mlp_params(model, "video_weight_fc")
    if _needs_grad(text_feat, video_feat, *pt_params, *pv_params):
        from .backward import LocalLevelFn
        return LocalLevelFn.apply(model, text_mask, video_mask, text_feat, video_feat, *pt_params, *pv_params)
    return head.similarity_matrix(text_feat, video_feat, text_mask, video_mask,
                                  model.scorer_weights("text_weight_fc"), model.scorer_weights("video_weight_fc"),
                                  model._prec(for_head=False))


def global_level_sim(model, gt, gv):
    if gt.shape[1] == 1 and gv.shape[1] == 1:
        if _needs_grad(gt, gv):
            from .backward import GlobalLogitsFn
            return GlobalLogitsFn.apply(gt, gv)
    else:
        pt_params = _mlp_params(model, "text_weight_fc1")
        pv_params = _mlp_params(model, "video_weight_fc1")
        if _needs_grad(gt, gv, *pt_params, *pv_params):
            from .backward import GlobalLevelMultiFn
            return GlobalLevelMultiFn.apply(model, gt, gv, *pt_params, *pv_params)
    return head.global_logits(gt, gv, model.scorer_weights("text_weight_fc1"), model.scorer_weights("video_weight_fc1"))


def row_loss_terms(S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T):
    if _needs_grad(S, G, bank_c0, bank_c1, wc_text, wc_video, logit_scale):
        from .backward import RowLossFn
        return RowLossFn.apply(S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T)
    return ops.row_losses(S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T)
