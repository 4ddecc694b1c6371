"""Hand-derived backward of one token-clustering stage (CTM + TCBlock: reference cluster.py:670-717, 834-965) from the
intermediates its fused HIP forward leaves in its workspace.

Why: the training step is bound by the NUMBER of small launches: the autograd-traced stage costs
~37 forward and ~75 backward launches per stage and modality.  With the forward on the grouped HIP kernels (7 launches for
both modalities) and this backward (~45 launches per stage and modality, plain torch ops on the saved tensors: GEMMs,
native layer-norm / softmax backward, gather / scatter-add) the clustering's share of the step drops by more than half.

Forward being differentiated (cluster.py of this package, same arithmetic as the reference):
    y      = x0 + [x0[n-1], x0[n], x0[n+1]] @ Wcat                    token convolution k=3 + residual
    xn     = LayerNorm_ctm(y)
    score  = xn . w_s + b_s,  -inf on masked tokens;  w = exp(score)
    merged = sum_{n in cluster c} (w_n / (sum_{m in c} w_m + 1e-6)) xn_n       (cluster ids from DPC-KNN: no gradient)
    q = Lq(LN1(merged)),  (k, v) = Lkv(LN1(xn)),  p = softmax(scale q k^T + score),  out = merged + Lproj(p v)
`saved`: x0, y, xn, score, w (= exp(score), 0 on masked tokens), assign [B,N] int64, merged, q [B*c,C], kv [B*N,2C], mask.
"""
import math

import torch
import torch.nn.functional as F


def saved_from_modules(ctm, blk, x0, mask, assign):
    """The tensors the fused forward leaves behind, computed with torch ops (tests; CPU)."""
    from .cluster import merge_by_cluster
    with torch.no_grad():
        y = ctm.conv(x0)
        xn = ctm.norm(y)
        score = ctm.score(xn).squeeze(-1)
        if mask is not None:
            score = score.masked_fill((1 - mask).to(torch.bool), float("-inf"))
        w = score.exp()
        c = max(math.ceil(x0.shape[1] * ctm.sample_ratio), 1)
        merged = merge_by_cluster(xn, assign, c, w)
        B, N, C = xn.shape
        q = blk.attn.q(blk.norm1(merged)).reshape(B * c, C)
        kv = blk.attn.kv(blk.norm1(xn)).reshape(B * N, 2 * C)
    return dict(x0=x0, y=y, xn=xn, score=score, w=w, assign=assign, merged=merged, q=q, kv=kv, mask=mask)


def stage_backward(ctm, blk, saved, g):
    """g = d loss / d stage output [B,c,C]  ->  (d loss / d x0 [B,N,C], {parameter: gradient})."""
    x0, y, xn, score, w, assign, merged = (saved[k] for k in ("x0", "y", "xn", "score", "w", "assign", "merged"))
    mask = saved.get("mask")
    B, N, C = xn.shape
    c = merged.shape[1]
    attn = blk.attn
    H = attn.num_heads
    dh = C // H
    scale = attn.scale
    n1 = blk.norm1
    grads = {}
    g = g.contiguous()
    g2 = g.reshape(B * c, C)

    # LN1 forward again for its statistics (and its outputs, which the workspace keeps only as bf16 pairs)
    qn, mean_q, rstd_q = torch.native_layer_norm(merged, (C,), n1.weight, n1.bias, n1.eps)
    kvn, mean_k, rstd_k = torch.native_layer_norm(xn, (C,), n1.weight, n1.bias, n1.eps)

    # attention probabilities again
    q = saved["q"].view(B, c, H, dh).transpose(1, 2)                       # [B,H,c,dh]
    kv = saved["kv"].view(B, N, 2, H, dh)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)        # [B,H,N,dh]
    p = ((q * scale) @ k.transpose(-2, -1) + score[:, None, None, :]).softmax(-1)     # [B,H,c,N]
    out = (p @ v).transpose(1, 2).reshape(B * c, C)

    # proj
    grads[attn.proj.weight] = g2.t() @ out
    if attn.proj.bias is not None:
        grads[attn.proj.bias] = g2.sum(0)
    d_out = (g2 @ attn.proj.weight).view(B, c, H, dh).transpose(1, 2)      # [B,H,c,dh]
    # attention
    d_p = d_out @ v.transpose(-2, -1)                                      # [B,H,c,N]
    d_v = p.transpose(-2, -1) @ d_out                                      # [B,H,N,dh]
    d_logits = torch._softmax_backward_data(d_p, p, -1, p.dtype)
    d_q = (d_logits @ k) * scale
    d_k = (d_logits.transpose(-2, -1) @ q) * scale
    d_score = d_logits.sum((1, 2))                                         # [B,N]: the score bias of every head and query
    d_q2 = d_q.transpose(1, 2).reshape(B * c, C)
    d_kv2 = torch.stack((d_k.transpose(1, 2), d_v.transpose(1, 2)), dim=2).reshape(B * N, 2 * C)
    # q / kv projections
    qn2, kvn2 = qn.reshape(B * c, C), kvn.reshape(B * N, C)
    grads[attn.q.weight] = d_q2.t() @ qn2
    if attn.q.bias is not None:
        grads[attn.q.bias] = d_q2.sum(0)
    grads[attn.kv.weight] = d_kv2.t() @ kvn2
    if attn.kv.bias is not None:
        grads[attn.kv.bias] = d_kv2.sum(0)
    d_qn = (d_q2 @ attn.q.weight).view(B, c, C)
    d_kvn = (d_kv2 @ attn.kv.weight).view(B, N, C)
    # LN1, used twice
    d_merged, dg_a, db_a = torch.ops.aten.native_layer_norm_backward(d_qn, merged, (C,), mean_q, rstd_q, n1.weight, n1.bias,
                                                            [True, True, True])
    d_xn, dg_b, db_b = torch.ops.aten.native_layer_norm_backward(d_kvn, xn, (C,), mean_k, rstd_k, n1.weight, n1.bias,
                                                        [True, True, True])
    grads[n1.weight] = dg_a + dg_b
    grads[n1.bias] = db_a + db_b
    d_merged = d_merged + g                                                # the residual path of the block

    # weighted cluster means
    idx = assign[..., None].expand(B, N, C)
    Dn = d_merged.gather(1, idx)                                           # every token sees its cluster's gradient
    total = torch.zeros((B, c), dtype=w.dtype, device=w.device).scatter_add_(1, assign, w) + 1e-6
    total_n = total.gather(1, assign)
    share = w / total_n
    d_xn = d_xn + Dn * share[..., None]
    d_share = (xn * Dn).sum(-1)
    d_total = torch.zeros((B, c), dtype=w.dtype, device=w.device).scatter_add_(1, assign, -d_share * w / (total_n * total_n))
    d_w = d_share / total_n + d_total.gather(1, assign)
    # score (masked tokens: w = 0 and p = 0, so both terms vanish; the product form keeps -inf out of the arithmetic)
    d_sc = d_w * w + d_score
    if mask is not None:
        d_sc = d_sc * (mask > 0)
    ws = ctm.score.weight                                                  # [1,C]
    d_xn = d_xn + d_sc[..., None] * ws[0]
    grads[ctm.score.weight] = torch.einsum("bn,bnc->c", d_sc, xn)[None]
    if ctm.score.bias is not None:
        grads[ctm.score.bias] = d_sc.sum().reshape(1)
    # LayerNorm of the CTM
    ln = ctm.norm
    _, mean_y, rstd_y = torch.native_layer_norm(y, (C,), ln.weight, ln.bias, ln.eps)
    d_y, dg, db = torch.ops.aten.native_layer_norm_backward(d_xn.contiguous(), y, (C,), mean_y, rstd_y, ln.weight, ln.bias,
                                                   [True, True, True])
    grads[ln.weight], grads[ln.bias] = dg, db
    # token convolution + residual
    wconv = ctm.conv.conv.weight                                           # [C_out, C_in, 3]
    wcat = wconv.permute(2, 1, 0).reshape(3 * wconv.shape[1], wconv.shape[0])
    d_y2 = d_y.reshape(B * N, C)
    d_cat = (d_y2 @ wcat.t()).view(B, N, 3, C)
    d_x0 = d_y + d_cat[:, :, 1]
    if N > 1:
        d_x0 = d_x0 + F.pad(d_cat[:, 1:, 0], (0, 0, 0, 1)) + F.pad(d_cat[:, :-1, 2], (0, 0, 1, 0))
    xp = F.pad(x0, (0, 0, 1, 1))
    cat = torch.cat((xp[:, :-2], xp[:, 1:-1], xp[:, 2:]), dim=-1).reshape(B * N, 3 * C)
    grads[wconv] = (cat.t() @ d_y2).view(3, wconv.shape[1], wconv.shape[0]).permute(2, 1, 0)
    return d_x0, grads
