"""Token clustering stage that produces the one-token "global" feature per sample.

Mirrors the interface of the reference's NeighborRetr/models/cluster.py (CTM :670-717, TCBlock
:891-965, cluster_dpc_knn :453-509, merge_tokens :512-561) with the same parameter names, so a
reference state_dict loads unchanged.  In this round the stage runs on stock PyTorch-ROCm ops
("stays on torch ops in the first slice"); it is written as dense
batched tensor algebra (one-hot matmuls instead of index_add_, no Python loops) so it captures
cleanly into a HIP graph and ports directly to a fused kernel later (row f-1).

Only the pieces the retrieval head actually executes are implemented: the `x`, `mask` and
`token_score` entries of the reference's token dictionaries.  (`idx_token` / `agg_weight` are
bookkeeping the head never reads.)
"""
import math

import torch
import torch.nn.functional as F
from torch import nn


def dpc_knn_assign(x, cluster_num, k, mask=None, noise=None):
    """Density-peaks clustering with k-NN density; returns the cluster id of every token.

    x [B,N,C]; mask [B,N] (>0 = valid) or None; noise [B,N] in [0,1) replaces the reference's
    torch.rand tie-break draw (cluster.py:483) when given, so tests can be deterministic."""
    if x.is_cuda and x.shape[1] <= 64:
        from . import ops                      # two HIP launches instead of ~25 tiny ATen kernels
        return ops.dpc_knn_assign(x, cluster_num, k, mask, noise)
    with torch.no_grad():
        B, N, C = x.shape
        dist = torch.cdist(x, x) / math.sqrt(C)
        valid = None
        if mask is not None:
            valid = mask > 0
            far = dist.max() + 1
            dist = torch.where(valid[:, None, :], dist, far.expand_as(dist))
        knn = dist.topk(k, dim=-1, largest=False).values
        density = torch.exp(-(knn * knn).mean(-1))
        if noise is None:
            noise = torch.rand(density.shape, device=density.device, dtype=density.dtype)
        density = density + noise.to(density.dtype) * 1e-6
        if valid is not None:
            density = density * valid
        denser = density[:, None, :] > density[:, :, None]
        dmax = dist.flatten(1).max(-1).values[:, None, None]
        parent = torch.where(denser, dist, dmax.expand_as(dist)).min(-1).values
        # centres: top `cluster_num` scores, exact ties (zero-score padding tokens, equal densities) -> LOWER index,
        # the rule of the HIP kernels (nr_cluster.hip) and of oracle.dpc_knn(centre_ties="lowest_index");
        # torch.topk (cluster.py:498 in the reference) leaves the tie order to the backend
        centres = (parent * density).sort(dim=-1, descending=True, stable=True).indices[:, :cluster_num]   # [B,c]
        to_centre = dist.gather(1, centres[:, :, None].expand(B, cluster_num, N))
        assign = to_centre.argmin(1)
        ids = torch.arange(cluster_num, device=x.device)[None, :].expand(B, cluster_num)
        assign.scatter_(1, centres, ids)                                           # a centre joins itself
        return assign


def merge_by_cluster(x, assign, cluster_num, tok_w):
    """Weighted mean of every cluster's tokens (weights tok_w [B,N] >= 0), dense one-hot form."""
    onehot = F.one_hot(assign, cluster_num).to(x.dtype)                            # [B,N,c]
    total = torch.einsum("bnc,bn->bc", onehot, tok_w) + 1e-6
    share = tok_w / total.gather(1, assign)
    return torch.einsum("bnc,bnd->bcd", onehot, x * share[..., None])


class TokenConv(nn.Module):
    """Residual k=3 convolution along the token axis (cluster.py:638-667)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, bias=False, padding=padding)

    def forward(self, x):
        # The k=3 convolution over <=64 tokens as ONE GEMM: [B*N, 3C] x [3C, C] on the shifted
        # copies of x.  (nn.functional.conv1d would go through MIOpen, which for this shape runs an
        # im2col + GEMM pair per sample -- ~140 launches per step.)  Same arithmetic, same weights.
        B, N, C = x.shape
        w = self.conv.weight                                    # [C_out, C_in, 3]
        if w.shape[2] != 3 or self.conv.padding[0] != 1:
            return x + self.conv(x.transpose(1, 2)).transpose(1, 2)
        xp = F.pad(x, (0, 0, 1, 1))                             # zero token before and after
        cat = torch.cat((xp[:, :-2], xp[:, 1:-1], xp[:, 2:]), dim=-1)          # [B,N,3C]: x[n-1], x[n], x[n+1]
        wcat = w.permute(2, 1, 0).reshape(3 * w.shape[1], w.shape[0])          # [3C_in, C_out]
        return x + (cat.reshape(B * N, 3 * C) @ wcat).view(B, N, -1)


class CTM(nn.Module):
    """Score tokens, cluster them with DPC-KNN and merge each cluster (cluster.py:670-717)."""

    def __init__(self, sample_ratio, embed_dim, dim_out, k=5):
        super().__init__()
        self.sample_ratio = sample_ratio
        self.dim_out = dim_out
        self.conv = TokenConv(embed_dim, dim_out)
        self.norm = nn.LayerNorm(dim_out)
        self.score = nn.Linear(dim_out, 1)
        self.k = k

    def forward(self, tokens, noise=None, assign=None):
        """assign: optional precomputed cluster ids [B,N] (the fused HIP forward saves them: cluster_fused.ClusterStagesFn
        recomputes this stage in its backward without re-running DPC-KNN, whose indices carry no gradient)."""
        x = self.norm(self.conv(tokens["x"]))
        score = self.score(x).squeeze(-1)
        mask = tokens.get("mask")
        if mask is not None:
            # the reference fills its score view in place with -inf (cluster.py:703-705), which
            # is also what the following attention adds to its logits for masked tokens
            score = score.masked_fill((1 - mask).to(torch.bool), float("-inf"))
        n_out = max(math.ceil(x.shape[1] * self.sample_ratio), 1)
        if assign is None:
            assign = dpc_knn_assign(x, n_out, self.k, mask, noise)
        merged = merge_by_cluster(x, assign, n_out, score.exp())
        down = {"x": merged, "mask": None}
        full = {"x": x, "mask": mask, "token_score": score}
        return down, full


class TCAttention(nn.Module):
    """Merged tokens attend to the un-merged ones, biased by the token scores (cluster.py:834-888)."""

    def __init__(self, dim, num_heads=8, qkv_bias=True):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, q_x, kv_x, kv_score):
        B, Nq, C = q_x.shape
        Nk = kv_x.shape[1]
        H = self.num_heads
        q = self.q(q_x).view(B, Nq, H, C // H).transpose(1, 2)
        kv = self.kv(kv_x).view(B, Nk, 2, H, C // H)
        k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
        att = (q * self.scale) @ k.transpose(-2, -1) + kv_score[:, None, None, :]
        out = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, Nq, C)
        return self.proj(out)


class TCBlock(nn.Module):
    """Pre-norm cross-attention residual block; the reference's block has no MLP (cluster.py:938-965)."""

    def __init__(self, dim, num_heads, qkv_bias=True):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = TCAttention(dim, num_heads, qkv_bias)
        self.apply(self._init_weights)            # cluster.py:918: the block initialises itself on construction

    @staticmethod
    def _init_weights(m):
        """cluster.py:920-932: q / kv / proj weights trunc_normal(std 0.02, cut at +-2), zero biases, LayerNorm 1 / 0."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, inputs):
        down, full = inputs
        x = down["x"]
        upd = self.attn(self.norm1(x), self.norm1(full["x"]), full["token_score"])
        return {"x": x + upd, "mask": None}
