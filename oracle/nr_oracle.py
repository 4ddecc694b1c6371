"""CPU oracle for the NeighborRetr similarity / neighbour-weighting / loss head.

TEST INFRASTRUCTURE ONLY.  Nothing under ``neighborretr_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the timed CPU baseline.

This is a restatement (plain torch CPU ops, fp32 or fp64 depending on the input
dtype) of the reference algorithm, written from the semantics recorded in
SURVEY.md section 8(a).  Every function cites the reference lines it follows
(paths relative to the reference checkout).  It is pinned against the reference
itself: ``oracle/capture_golden.py`` imports the reference in the build
container, runs both on the same seeded inputs and stores the reference's
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks this
file against those vectors wherever the tests run.

Parameters are passed as a flat ``dict`` keyed by the reference's state-dict
names (``text_weight_fc.0.weight`` ...), so the same dictionary drives the
reference module, this oracle and the HIP path.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NEG_BIG = -9e15  # the reference's "minus infinity" (modeling.py:486, until_module.py:111)


# ---------------------------------------------------------------------------
# a-4  local_level                                   (modeling.py:483-514)
# ---------------------------------------------------------------------------
def token_weight_logits(feat, P, prefix):
    """Linear(d,2d) -> ReLU -> Linear(2d,1) token scorer (modeling.py:148-153)."""
    h = F.relu(F.linear(feat, P[prefix + ".0.weight"], P[prefix + ".0.bias"]))
    return F.linear(h, P[prefix + ".2.weight"], P[prefix + ".2.bias"]).squeeze(-1)


def token_weights(feat, mask, P, prefix):
    """Masked softmax over the token axis (modeling.py:485-487 / 490-492)."""
    logit = token_weight_logits(feat, P, prefix)
    logit = logit.masked_fill((1 - mask).to(torch.bool), NEG_BIG)
    return torch.softmax(logit, dim=-1)


def local_level_parts(text_feat, video_feat, text_mask, video_mask, P,
                      prefix_t="text_weight_fc", prefix_v="video_weight_fc"):
    """Returns (S, t2v, v2t, w_t, w_v, argmax_v, argmax_t) -- modeling.py:483-514."""
    dt = text_feat.dtype
    w_t = token_weights(text_feat, text_mask, P, prefix_t)          # :485-487
    w_v = token_weights(video_feat, video_mask, P, prefix_v)        # :490-492
    tn = F.normalize(text_feat, dim=-1)                             # :495
    vn = F.normalize(video_feat, dim=-1)                            # :496
    A, Nt, d = tn.shape
    Bv, Nv, _ = vn.shape
    # :499 -- one [A*Nt, d] x [d, Bv*Nv] product, then view as [A,Bv,Nt,Nv]
    R = (tn.reshape(A * Nt, d) @ vn.reshape(Bv * Nv, d).t()).reshape(A, Nt, Bv, Nv).permute(0, 2, 1, 3)
    R = R * text_mask.to(dt)[:, None, :, None]                      # :500  (masked => 0, not -inf)
    R = R * video_mask.to(dt)[None, :, None, :]                     # :501
    pmax, arg_v = R.max(dim=-1)                                     # :504   [A,Bv,Nt]
    t2v = (pmax * w_t[:, None, :]).sum(-1)                          # :505
    qmax, arg_t = R.max(dim=-2)                                     # :508   [A,Bv,Nv]
    v2t = (qmax * w_v[None, :, :]).sum(-1)                          # :509
    S = (t2v + v2t) / 2.0                                           # :512
    return S, t2v, v2t, w_t, w_v, arg_v, arg_t


def local_level(text_feat, video_feat, text_mask, video_mask, P, **kw):
    S = local_level_parts(text_feat, video_feat, text_mask, video_mask, P, **kw)[0]
    return S, S.t()


# ---------------------------------------------------------------------------
# a-8  global_level                                   (modeling.py:516-539)
# ---------------------------------------------------------------------------
def global_level(gt, gv, P):
    """Un-normalised, un-masked variant of local_level on the merged tokens."""
    w_t = torch.softmax(token_weight_logits(gt, P, "text_weight_fc1"), dim=-1)    # :518-519
    w_v = torch.softmax(token_weight_logits(gv, P, "video_weight_fc1"), dim=-1)   # :522-523
    R = torch.einsum("atd,bvd->abtv", gt, gv)                                     # :526
    t2v = (R.max(dim=-1)[0] * w_t[:, None, :]).sum(-1)                            # :529-530
    v2t = (R.max(dim=-2)[0] * w_v[None, :, :]).sum(-1)                            # :533-534
    G = (t2v + v2t) / 2.0
    return G, G.t()


# ---------------------------------------------------------------------------
# a-5  centrality weights                             (modeling.py:403-430)
# ---------------------------------------------------------------------------
def centrality_weights(text_feat, video_feat, gt, gv, centrality_scale, multi_token="raise"):
    """exp(c * mean_j <g_i, x_j>) over ALL B*N tokens, padding included (:423-428).

    One global token per sample (MSR-VTT token counts): [B] weights, as the reference.  With G > 1 global tokens
    (ActivityNet token counts: 3 text / 6 video tokens) the reference's expression yields [B,G] weights and its
    loss then fails to broadcast (until_module.py:321) -- there is NO reference answer.  `multi_token`:
      "raise"  (default) keep the [B,G] weights the reference computes; centrality_weighting_loss raises like it;
      "mean"   the build's documented reduction: w_i = mean_g exp(c * <g_ig, mean_j x_j>), i.e. what
               `diag_log_probs[:, None] * weights` followed by the reference's `.mean()` would have given."""
    d = text_feat.shape[-1]
    tn = F.normalize(text_feat.reshape(-1, d), dim=-1)
    vn = F.normalize(video_feat.reshape(-1, d), dim=-1)
    gtn = F.normalize(gt.squeeze(1), dim=-1)                        # :409  (squeeze is a no-op when G > 1)
    gvn = F.normalize(gv.squeeze(1), dim=-1)
    ct = (gtn @ tn.t()).mean(-1)                                    # [B] or [B,G]
    cv = (gvn @ vn.t()).mean(-1)
    wt, wv = torch.exp(ct * centrality_scale), torch.exp(cv * centrality_scale)
    if multi_token == "mean":
        if wt.dim() == 2:
            wt = wt.mean(-1)
        if wv.dim() == 2:
            wv = wv.mean(-1)
    elif multi_token != "raise":
        raise ValueError(f"multi_token={multi_token!r}")
    return wt, wv


# ---------------------------------------------------------------------------
# a-6  centrality-weighted InfoNCE                    (until_module.py:303-328)
# ---------------------------------------------------------------------------
def centrality_weighting_loss(S_scaled, w):
    lp = F.log_softmax(S_scaled, dim=-1)
    return -(torch.diag(lp) * w).mean()          # [B] * [B,G] raises RuntimeError, as until_module.py:321 does


def centrality_loss(S, w_text, w_video, logit_scale):
    """modeling.py:362-380."""
    return (centrality_weighting_loss(S * logit_scale, w_text)
            + centrality_weighting_loss(S.t() * logit_scale, w_video)) / 2


# ---------------------------------------------------------------------------
# a-7  neighbour adjusting loss                       (until_module.py:56-211)
# ---------------------------------------------------------------------------
def neighbor_mask(S, K):
    """Top-K of every row with the diagonal excluded (:100-129).

    The reference uses torch.sort(descending=True); exact float ties are broken
    here towards the lower column index (what a stable sort gives)."""
    B = S.shape[0]
    if K > B:
        raise IndexError("num_neighbors > batch (until_module.py:119-123 raises the same)")
    eye = torch.eye(B, dtype=S.dtype)
    s = torch.where(eye == 0, S, torch.full_like(S, NEG_BIG))
    idx = torch.sort(s, dim=-1, descending=True, stable=True)[1][:, :K]
    nb = torch.zeros_like(S)
    nb.scatter_(1, idx, 1.0)
    ext = eye.clone()
    ext.scatter_(1, idx, 1.0)
    return nb, ext


def minmax_over_rest(X, ext):
    """(X - min_rest) / (max_rest - min_rest), rest = {j : ext[i,j] == 0}  (:65-86)."""
    lo = torch.where(ext == 0, X, torch.full_like(X, 9e15)).min(-1, keepdim=True)[0]
    hi = torch.where(ext == 0, X, torch.full_like(X, -9e15)).max(-1, keepdim=True)[0]
    return (X - lo) / (hi - lo)


def neighbor_adjusting_parts(S, S_bank, K, T):
    nb, ext = neighbor_mask(S.detach(), K)
    c = S_bank.sum(-1) / S_bank.shape[-1]                                   # :181
    ns = minmax_over_rest(S, ext)                                           # :185
    nc = minmax_over_rest(c[None, :].expand(S.shape[0], -1), ext)           # :182,186
    adj = torch.where(nb == 1, ns - nc, torch.full_like(S, NEG_BIG))        # :189-193
    p = torch.softmax(adj * T, dim=-1)                                      # :147
    p = torch.where(nb == 1, p, torch.zeros_like(p))                        # :150-154
    eye = torch.eye(S.shape[0], dtype=S.dtype)
    p = p * (1 - eye) + eye                                                 # fill_diagonal_(1) :157
    masked = torch.where(ext == 1, S, torch.full_like(S, NEG_BIG))          # :199-203
    lp = F.log_softmax(masked, dim=-1) * p                                  # :206
    rows = -lp.sum(-1) / p.sum(-1)                                          # :207
    return rows.mean(), dict(nb=nb, ext=ext, c=c, ns=ns, nc=nc, p=p, rows=rows)


def neighbor_adjusting_loss(S, S_bank, K, T):
    return neighbor_adjusting_parts(S, S_bank, K, T)[0]


def neighbor_loss(S, bank_t2v, bank_v2t, K, T):
    """modeling.py:393-401: the t2v loss takes the *v2t* bank matrix and vice versa.

    bank_t2v = local_level(text, bank_video)[0]   [B,M]
    bank_v2t = local_level(bank_text, video)[1]   [B,M]  (= [M,B] transposed)"""
    return (neighbor_adjusting_loss(S, bank_v2t, K, T)
            + neighbor_adjusting_loss(S.t(), bank_t2v, K, T)) / 2


# ---------------------------------------------------------------------------
# a-8  uniform regularisation (log-Sinkhorn)          (until_module.py:214-291)
# ---------------------------------------------------------------------------
def sinkhorn_targets(G, beta, iters=50):
    with torch.no_grad():
        m, n = G.shape
        norm = -math.log(m + n)                                      # :241
        u = torch.zeros(m, dtype=G.dtype)
        v = torch.zeros(n, dtype=G.dtype)
        for _ in range(iters):                                        # :248-250
            u = norm - torch.logsumexp(G + v[None, :], dim=1)
            v = norm - torch.logsumexp(G + u[:, None], dim=0)
        Q = (G + u[:, None] + v[None, :] - norm).exp()                # :253-257
        return beta * Q + (1 - beta) * torch.eye(m, n, dtype=G.dtype) # :260-264


def uniform_regularization_loss(G, scale, beta, iters=50):
    tgt = sinkhorn_targets(G.detach(), beta, iters)
    return -(F.log_softmax(G * scale, dim=-1) * tgt).sum(-1).mean()   # :285-289


def uniform_loss(G, temperature, beta):
    """modeling.py:440-442: `temperature` is passed in the logit_scale slot."""
    return (uniform_regularization_loss(G, temperature, beta)
            + uniform_regularization_loss(G.t(), temperature, beta)) / 2


# ---------------------------------------------------------------------------
# a-9  KL(local || global)                            (until_module.py:331-359)
# ---------------------------------------------------------------------------
def kl_divergence_loss(G, S):
    q = F.log_softmax(G, dim=-1)
    p = F.softmax(S, dim=-1)
    return (p * (p.log() - q)).mean()        # kl_div(reduction='mean') divides by B*B


def kl_loss(G, S):
    return (kl_divergence_loss(G, S) + kl_divergence_loss(G.t(), S.t())) / 2   # modeling.py:329-332


# ---------------------------------------------------------------------------
# a-10  merge_global_features (DPC-KNN token merging) (cluster.py, modeling.py:446-481)
# ---------------------------------------------------------------------------
def dpc_knn(x, cluster_num, k, mask, noise, centre_ties="lowest_index"):
    """cluster.py:453-509.  `noise` replaces the reference's torch.rand draw (:483).

    `centre_ties` states how EXACT ties in the centre score are broken (:498, torch.topk there).  The only
    exact ties that occur are the zero scores of padding tokens (density * valid == 0, :488): a sample with fewer
    valid tokens than `cluster_num` takes its extra centres among them.
      "lowest_index"  the tied token with the lower index wins (a stable descending sort) -- the documented rule
                      of this build; what the HIP kernels implement.
      "torch_topk"    whatever torch.topk does on this host (implementation-defined: an nth_element artefact on
                      CPU, a radix select on GPU) -- what the reference executes; used by capture_golden.py and
                      by the CPU test that records on which fixture rows the two rules pick different centres."""
    B, N, C = x.shape
    dist = torch.cdist(x, x) / (C ** 0.5)
    if mask is not None:
        valid = mask > 0
        dist = dist * valid[:, None, :] + (dist.max() + 1) * (~valid[:, None, :])   # :473-475
    near = torch.topk(dist, k=k, dim=-1, largest=False)[0]
    density = (-(near ** 2).mean(-1)).exp() + noise * 1e-6                         # :479-484
    if mask is not None:
        density = density * valid                                                   # :488
    higher = (density[:, None, :] > density[:, :, None]).to(x.dtype)                # :491-492
    dmax = dist.flatten(1).max(-1)[0][:, None, None]
    parent_dist = (dist * higher + dmax * (1 - higher)).min(-1)[0]                  # :494
    score = parent_dist * density                                                   # :497
    if centre_ties == "torch_topk":
        centres = torch.topk(score, k=cluster_num, dim=-1)[1]                       # :498
    elif centre_ties == "lowest_index":
        centres = torch.sort(score, dim=-1, descending=True, stable=True)[1][:, :cluster_num]
    else:
        raise ValueError(f"centre_ties={centre_ties!r}")
    to_centre = torch.gather(dist, 1, centres[:, :, None].expand(B, cluster_num, N))
    assign = to_centre.argmin(dim=1)                                                # :501-502
    assign.scatter_(1, centres, torch.arange(cluster_num)[None, :].expand(B, cluster_num))  # :505-507
    return assign


def merge_tokens(x, assign, cluster_num, tok_w):
    """Weighted average of each cluster's members (cluster.py:512-561)."""
    B, N, C = x.shape
    onehot = F.one_hot(assign, cluster_num).to(x.dtype)             # [B,N,c]
    tot = torch.einsum("bnc,bn->bc", onehot, tok_w) + 1e-6          # :536-539
    nw = tok_w / torch.gather(tot, 1, assign)                       # :540
    return torch.einsum("bnc,bnd->bcd", onehot, x * nw[..., None])  # :543-547


def ctm_stage(x, mask, P, ctm, blk, ratio, k, noise, centre_ties="lowest_index"):
    """CTM (cluster.py:689-717) followed by TCBlock (cluster.py:938-965)."""
    B, N, C = x.shape
    x = x + F.conv1d(x.transpose(1, 2), P[ctm + ".conv.conv.weight"], padding=1).transpose(1, 2)  # :664
    x = F.layer_norm(x, (C,), P[ctm + ".norm.weight"], P[ctm + ".norm.bias"])
    score = F.linear(x, P[ctm + ".score.weight"], P[ctm + ".score.bias"]).squeeze(-1)
    if mask is not None:
        # in-place masked_fill_ on a view of token_score (:703-705): the -inf is
        # also what TCAttention later adds to its logits as `conf_kv`.
        score = score.masked_fill((1 - mask).to(torch.bool), float("-inf"))
    tok_w = score.exp()
    cnum = max(math.ceil(N * ratio), 1)
    assign = dpc_knn(x.detach(), cnum, k, mask, noise, centre_ties)
    merged = merge_tokens(x, assign, cnum, tok_w)
    # TCBlock: q = merged tokens, kv = un-merged tokens, both through norm1
    qn = F.layer_norm(merged, (C,), P[blk + ".norm1.weight"], P[blk + ".norm1.bias"])
    kvn = F.layer_norm(x, (C,), P[blk + ".norm1.weight"], P[blk + ".norm1.bias"])
    H = 8
    hd = C // H
    q = F.linear(qn, P[blk + ".attn.q.weight"], P[blk + ".attn.q.bias"]).reshape(B, cnum, H, hd).transpose(1, 2)
    kv = F.linear(kvn, P[blk + ".attn.kv.weight"], P[blk + ".attn.kv.bias"]).reshape(B, N, 2, H, hd)
    kk = kv[:, :, 0].transpose(1, 2)
    vv = kv[:, :, 1].transpose(1, 2)
    att = (q * hd ** -0.5) @ kk.transpose(-2, -1) + score[:, None, None, :]     # :877-880
    att = att.softmax(-1)
    o = (att @ vv).transpose(1, 2).reshape(B, cnum, C)
    o = F.linear(o, P[blk + ".attn.proj.weight"], P[blk + ".attn.proj.bias"])
    return merged + o


def merged_token_counts(Nt, Nv):
    t0 = max(math.ceil(Nt * (1 / 6)), 1)
    t1 = max(math.ceil(t0 * (1 / 4)), 1)
    v0 = max(math.ceil(Nv * (1 / 4)), 1)
    v1 = max(math.ceil(v0 * (1 / 3)), 1)
    return (t0, t1), (v0, v1)


def merge_global_features(text_feat, video_feat, text_mask, video_mask, P, noise, centre_ties="lowest_index"):
    """modeling.py:446-481.  noise = dict(t0,t1,v0,v1) of [B,N_stage] uniform draws."""
    t = ctm_stage(text_feat, text_mask, P, "text_ctm0", "text_block0", 1 / 6, 3, noise["t0"], centre_ties)
    v = ctm_stage(video_feat, video_mask, P, "video_ctm0", "video_block0", 1 / 4, 3, noise["v0"], centre_ties)
    t = ctm_stage(t, None, P, "text_ctm1", "text_block1", 1 / 4, 3, noise["t1"], centre_ties)
    v = ctm_stage(v, None, P, "video_ctm1", "video_block1", 1 / 3, 3, noise["v1"], centre_ties)
    return t, v


# ---------------------------------------------------------------------------
# a-3  _compute_losses                                (modeling.py:314-360)
# ---------------------------------------------------------------------------
def compute_losses(text_feat, video_feat, text_mask, video_mask,
                   mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v, P, hp, logit_scale, noise,
                   return_parts=False, centrality_multi_token="raise", centre_ties="lowest_index"):
    """hp: dict(centrality_scale, beta, num_neighbors, temperature,
    uniform_weight, neighbor_weight, kl_weight).
    centrality_multi_token: see centrality_weights (only matters with > 1 global token per sample).
    centre_ties: see dpc_knn."""
    S, _ = local_level(text_feat, video_feat, text_mask, video_mask, P)
    gt, gv = merge_global_features(text_feat, video_feat, text_mask, video_mask, P, noise, centre_ties)
    G, _ = global_level(gt, gv, P)
    L_u = uniform_loss(G, hp["temperature"], hp["beta"])
    L_kl = kl_loss(G, S)
    if (gt.shape[1] != 1 or gv.shape[1] != 1) and centrality_multi_token == "raise":
        raise RuntimeError("reference crashes here when >1 global token survives "
                           "(until_module.py:321); parity unpinned for this term")
    w_text, w_video = centrality_weights(text_feat, video_feat, gt, gv, hp["centrality_scale"], centrality_multi_token)
    L_c = centrality_loss(S, w_text, w_video, logit_scale)
    bank_t2v = local_level(text_feat, mb_feat_v, text_mask, mb_mask_v, P)[0]       # :389
    bank_v2t = local_level(mb_feat_t, video_feat, mb_mask_t, video_mask, P)[1]     # :390
    L_n = neighbor_loss(S, bank_t2v, bank_v2t, hp["num_neighbors"], hp["temperature"])
    total = L_c + L_u * hp["uniform_weight"] + L_n * hp["neighbor_weight"] + L_kl * hp["kl_weight"]
    out = (total, L_c, L_u, L_n, L_kl)
    if return_parts:
        return out, dict(S=S, G=G, gt=gt, gv=gv, w_text=w_text, w_video=w_video,
                         bank_t2v=bank_t2v, bank_v2t=bank_v2t)
    return out


# ---------------------------------------------------------------------------
# a-11  memory-bank FIFO                              (modeling.py:222-249)
# ---------------------------------------------------------------------------
def update_memory_bank(bank, batch):
    """bank / batch: tuples (ind, feat_t, feat_v, mask_t, mask_v).  Newest first,
    capacity = rows the bank had before the push; empty bank adopts the batch."""
    if bank[2].shape[0] == 0:
        return tuple(x.clone() for x in batch)
    cap = bank[2].shape[0]
    return tuple(torch.cat((n, o), 0)[:cap] for n, o in zip(batch, bank))


# ---------------------------------------------------------------------------
# a-12  retrieval metrics                             (utils/metrics.py:39-79)
# ---------------------------------------------------------------------------
def compute_metrics(sim):
    """Rank of the diagonal inside every descending-sorted row; exact float
    equality, so a tie with the diagonal yields one extra hit per tied entry."""
    sim = np.asarray(sim)
    sx = np.sort(-sim, axis=1)
    d = np.diag(-sim)[:, None]
    ind = np.where(sx - d == 0)[1]
    n = len(ind)
    return {
        "R1": float(np.sum(ind == 0)) * 100 / n,
        "R5": float(np.sum(ind < 5)) * 100 / n,
        "R10": float(np.sum(ind < 10)) * 100 / n,
        "R50": float(np.sum(ind < 50)) * 100 / n,
        "MR": float(np.median(ind)) + 1,
        "MedianR": float(np.median(ind)) + 1,
        "MeanR": float(np.mean(ind)) + 1,
        "cols": [int(i) for i in ind],
    }


# ---------------------------------------------------------------------------
# a-12b  multi-sentence retrieval  (training/evaluator.py:225-262, utils/metrics.py:82-148)
# ---------------------------------------------------------------------------
def pad_sentence_groups(S, cut_off_points):
    """evaluator.py:236-250: the [n_sentences, n_videos] matrix cut at the groups' last sentences
    (`cut_off_points`, already minus one as in :98) and padded with -inf to [n_videos, max_len, n_videos]."""
    S = np.asarray(S)
    ends = [c + 1 for c in cut_off_points]
    starts = [0] + ends[:-1]
    max_len = max(e - s for s, e in zip(starts, ends))
    return np.stack([np.concatenate((S[s:e], np.full((max_len - e + s, S.shape[1]), -np.inf)), axis=0)
                     for s, e in zip(starts, ends)], axis=0)


def tensor_text_to_video_metrics(sim_tensor, top_k=(1, 5, 10, 50), stable=True):
    """metrics.py:82-126: rank of video i in the row of every valid sentence (i, s), by a double argsort
    (stable=True pins what the sort does among exact ties: lower video index first)."""
    T = np.asarray(sim_tensor)
    stacked = T.transpose(1, 0, 2)                              # [sentence slot, group, video]
    stacked = np.where(np.isnan(stacked), np.inf, stacked)      # torch.argsort(descending=True) puts NaN first
    first = np.argsort(-stacked, axis=-1, kind="stable" if stable else "quicksort")
    second = np.argsort(first, axis=-1, kind="stable")
    ranks = np.diagonal(second, axis1=1, axis2=2).reshape(-1)
    own = np.diagonal(T, axis1=0, axis2=2).reshape(-1)          # [sentence slot, group] flattened the same way
    valid = ranks[~(np.isinf(own) | np.isnan(own))]
    res = {f"R{k}": float(np.sum(valid < k) * 100 / len(valid)) for k in top_k}
    res["MedianR"] = float(np.sort(valid + 1)[(len(valid) - 1) // 2])   # torch.median: the LOWER middle element
    res["MeanR"] = float(np.mean(valid + 1))
    res["Std_Rank"] = float(np.std(valid + 1))
    res["MR"] = res["MedianR"]
    return res


def tensor_video_to_text_sim(sim_tensor):
    """metrics.py:128-148: NaN -> -inf, best sentence of every group, transposed to [video, group]."""
    T = np.array(sim_tensor, copy=True)
    T[T != T] = -np.inf
    return T.max(axis=1).T


def multi_sentence_metrics(S, cut_off_points):
    """evaluator.py:225-262 -> (text->video, video->text) from the sentence x video matrix."""
    padded = pad_sentence_groups(S, cut_off_points)
    return tensor_text_to_video_metrics(padded), compute_metrics(tensor_video_to_text_sim(padded))
