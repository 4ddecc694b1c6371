"""Test infrastructure: the four row terms of one direction restated in torch ops (until_module.py:56-359), for the rows of
a slab of S -- the cross-check of sharded.SlabRowLossFn in tests/test_sharded_gpu.py.  Not part of the product package."""
import torch

NEG_BIG = -9e15


def neighbor_rows(S, c, K, T, diag_col):
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


def direction_terms(S, G, tgt, c, w, ls, K, T, diag_col):
    """The four row terms (summed over the slab's rows) of one direction: S, G, tgt [n,B]; w [n] centrality weights."""
    rows = torch.arange(S.shape[0], device=S.device)
    lp_c = torch.log_softmax(S * ls, dim=-1)[rows, diag_col]                           # until_module.py:315-327
    cent = -(lp_c * w).sum()
    unif = -(torch.log_softmax(G * T, dim=-1) * tgt).sum()                             # :285-289
    p = torch.softmax(S, dim=-1)
    kl = (p * (torch.log_softmax(S, dim=-1) - torch.log_softmax(G, dim=-1))).sum()     # :351-357
    neigh = neighbor_rows(S, c, K, T, diag_col).sum()
    return cent, unif, neigh, kl
