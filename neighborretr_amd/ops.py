"""Tensor-level wrappers, one per C entry point of libnr_hip.so (no autograd here).

Each function allocates its outputs with torch (device memory plumbing only), enqueues the HIP
kernel on torch's current stream and returns without synchronising.
"""
from collections import namedtuple

import torch

from . import hip

Prepared = namedtuple("Prepared", "hi lo norm colsum n_tok d")


_COUNTERS = {}


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


def prepare_tokens(x, mask=None, normalize=True, want_lo=True, want_norm=True, want_colsum=False):
    """x [..., d] f32 -> bf16 hi/lo of normalize(x)*mask  (nr_prepare_tokens)."""
    x = _f32(x).contiguous()
    d = x.shape[-1]
    n_tok = x.numel() // d
    dev = x.device
    hi = torch.empty((n_tok, d), dtype=torch.int16, device=dev)
    lo = torch.empty((n_tok, d), dtype=torch.int16, device=dev) if want_lo else None
    norm = torch.empty((n_tok,), dtype=torch.float32, device=dev) if want_norm else None
    colsum = None
    if want_colsum:
        colsum = torch.empty((hip.prepare_parts(n_tok), d), dtype=torch.float32, device=dev)
    m = None
    if mask is not None:
        m = _f32(mask).contiguous()
        if m.numel() != n_tok:
            raise ValueError("mask does not match the token count")
    hip.call("nr_prepare_tokens", hip.ptr(x, torch.float32), hip.ptr(m, allow_none=True), n_tok, d,
             1 if normalize else 0, hip.ptr(hi), hip.ptr(lo, allow_none=True), hip.ptr(norm, allow_none=True),
             hip.ptr(colsum, allow_none=True), hip.stream_ptr())
    return Prepared(hi, lo, norm, colsum, n_tok, d)


def prepare_tokens_pair(x0, mask0, x1, mask1, want_lo=True, want_colsum=False):
    """prepare_tokens of two token tensors of the same width in ONE launch (nr_prepare_tokens_pair) -> (Prepared, Prepared)."""
    x0, x1 = _f32(x0).contiguous(), _f32(x1).contiguous()
    d = x0.shape[-1]
    if x1.shape[-1] != d:
        raise ValueError("both tensors must have the same feature width")
    dev = x0.device
    out, argv = [], []
    for x, mask in ((x0, mask0), (x1, mask1)):
        n_tok = x.numel() // d
        hi = torch.empty((n_tok, d), dtype=torch.int16, device=dev)
        lo = torch.empty((n_tok, d), dtype=torch.int16, device=dev) if want_lo else None
        norm = torch.empty((n_tok,), dtype=torch.float32, device=dev)
        colsum = torch.empty((hip.prepare_parts(n_tok), d), dtype=torch.float32, device=dev) if want_colsum else None
        m = None
        if mask is not None:
            m = _f32(mask).contiguous()
            if m.numel() != n_tok:
                raise ValueError("mask does not match the token count")
        argv += [hip.ptr(x, torch.float32), hip.ptr(m, allow_none=True), n_tok, hip.ptr(hi), hip.ptr(lo, allow_none=True), hip.ptr(norm),
                 hip.ptr(colsum, allow_none=True)]
        out.append((Prepared(hi, lo, norm, colsum, n_tok, d), m))
    hip.call("nr_prepare_tokens_pair", *argv, d, 1, hip.stream_ptr())
    return out[0][0], out[1][0]


def split_bf16(w, want_lo=True):
    w = _f32(w).contiguous()
    hi = torch.empty(w.shape, dtype=torch.int16, device=w.device)
    lo = torch.empty(w.shape, dtype=torch.int16, device=w.device) if want_lo else None
    hip.call("nr_split_bf16", hip.ptr(w, torch.float32), w.numel(), hip.ptr(hi), hip.ptr(lo, allow_none=True),
             hip.stream_ptr())
    return hi, lo


def token_logit_parts(prep, w1_hi, w1_lo, b1, w2, prec):
    """[H/128, n_tok] partial logits of the token scorer (nr_token_logits_fwd)."""
    H = w1_hi.shape[0]
    parts = torch.empty((H // 128, prep.n_tok), dtype=torch.float32, device=prep.hi.device)
    hip.call("nr_token_logits_fwd", hip.ptr(prep.hi), hip.ptr(prep.lo, allow_none=True), hip.ptr(prep.norm), prep.n_tok,
             prep.d, hip.ptr(w1_hi), hip.ptr(w1_lo, allow_none=True), hip.ptr(b1, torch.float32),
             hip.ptr(w2, torch.float32), H, prec, hip.ptr(parts), hip.stream_ptr())
    return parts


def token_softmax(parts, b2, mask, n_samples, N, want_logits=False):
    w = torch.empty((n_samples, N), dtype=torch.float32, device=parts.device)
    logits = torch.empty((n_samples, N), dtype=torch.float32, device=parts.device) if want_logits else None
    m = _f32(mask).contiguous() if mask is not None else None
    hip.call("nr_token_softmax", hip.ptr(parts), parts.shape[0], hip.ptr(b2, torch.float32), hip.ptr(m, allow_none=True),
             n_samples, N, hip.ptr(w), hip.ptr(logits, allow_none=True), hip.stream_ptr())
    return w, logits


FUSE_TOKEN_SOFTMAX = True      # scorer MLP + masked softmax in one launch (nr_token_weights_fwd); False: two launches (A/B)
_SOFTMAX_SLOTS, _SOFTMAX_SLOT_LEN = 16, 4096


def _softmax_counters(dev, n_tok):
    """Zeroed per-row-tile counters for one nr_token_weights_fwd launch.  Scorer launches of one step may run side by side on
    different streams, so consecutive calls take different slots of a per-device pool (16 slots: a step issues at most six
    scorer launches; a slot comes round again only behind launches that are ordered after its previous user).  The kernel
    leaves its counters zeroed."""
    need = (n_tok + 63) // 64
    if need > _SOFTMAX_SLOT_LEN:
        return None
    key = ("softmax", dev)
    st = _COUNTERS.get(key)
    if st is None:
        st = _COUNTERS[key] = [torch.zeros((_SOFTMAX_SLOTS, _SOFTMAX_SLOT_LEN), dtype=torch.int32, device=dev), 0]
    st[1] = (st[1] + 1) % _SOFTMAX_SLOTS
    return st[0][st[1]]


def token_weights(prep, w1_hi, w1_lo, b1, w2, b2, mask, n_samples, N, prec, want_logits=False):
    """Token scorer + masked softmax (modeling.py:485-492): (w [n,N], logits or None).  One launch
    (nr_token_weights_fwd: the last column block of every row tile does the tile's softmax) when a block shape holds whole
    samples, else nr_token_logits_fwd + nr_token_softmax."""
    dev = prep.hi.device
    counters = _softmax_counters(dev, prep.n_tok) if FUSE_TOKEN_SOFTMAX else None
    if counters is not None:
        H = w1_hi.shape[0]
        parts = torch.empty((H // 128, prep.n_tok), dtype=torch.float32, device=dev)
        w = torch.empty((n_samples, N), dtype=torch.float32, device=dev)
        logits = torch.empty((n_samples, N), dtype=torch.float32, device=dev) if want_logits else None
        m = _f32(mask).contiguous() if mask is not None else None
        hip.N_CALLS += 1
        rc = hip.lib().nr_token_weights_fwd(
            hip.ptr(prep.hi), hip.ptr(prep.lo, allow_none=True), hip.ptr(prep.norm), n_samples, N, prep.d, hip.ptr(w1_hi),
            hip.ptr(w1_lo, allow_none=True), hip.ptr(b1, torch.float32), hip.ptr(w2, torch.float32), hip.ptr(b2, torch.float32),
            H, prec, hip.ptr(m, allow_none=True), hip.ptr(parts), hip.ptr(counters), counters.numel(), hip.ptr(w),
            hip.ptr(logits, allow_none=True), hip.stream_ptr())
        if rc == 0:
            return w, logits
        if rc != hip.NR_EUNSUPPORTED:
            hip._check("nr_token_weights_fwd", rc)
    parts = token_logit_parts(prep, w1_hi, w1_lo, b1, w2, prec)
    return token_softmax(parts, b2, mask, n_samples, N, want_logits)


def token_weights_pair(calls, prec):
    """Two token_weights calls of one precision in ONE launch (nr_token_weights_fwd_pair; bit-identical to the single calls).
    calls: two tuples (prep, w1_hi, w1_lo, b1, w2, b2, mask, n_samples, N).  -> [(w, None), (w, None)]; falls back to two
    launches when the two do not run the same block shape."""
    import ctypes
    dev = calls[0][0].hi.device
    probs, outs, keep = [], [], []
    for prep, w1_hi, w1_lo, b1, w2, b2, mask, n_samples, N in calls:
        counters = _softmax_counters(dev, prep.n_tok) if FUSE_TOKEN_SOFTMAX else None
        if counters is None:
            probs = None
            break
        H = w1_hi.shape[0]
        parts = torch.empty((H // 128, prep.n_tok), dtype=torch.float32, device=dev)
        w = torch.empty((n_samples, N), dtype=torch.float32, device=dev)
        m = _f32(mask).contiguous() if mask is not None else None
        q = hip.TokenWeightsProblem()
        q.tok_hi, q.tok_lo, q.norm = hip.ptr(prep.hi), hip.ptr(prep.lo, allow_none=True), hip.ptr(prep.norm)
        q.w1_hi, q.w1_lo = hip.ptr(w1_hi), hip.ptr(w1_lo, allow_none=True)
        q.b1, q.w2, q.b2 = hip.ptr(b1, torch.float32), hip.ptr(w2, torch.float32), hip.ptr(b2, torch.float32)
        q.mask, q.logit_part, q.counters = hip.ptr(m, allow_none=True), hip.ptr(parts), hip.ptr(counters)
        q.w, q.logits = hip.ptr(w), None
        q.n_samples, q.N, q.d, q.H, q.n_counters = int(n_samples), int(N), int(prep.d), int(H), counters.numel()
        probs.append(q)
        outs.append((w, None))
        keep += [parts, m]
    if probs is not None:
        hip.N_CALLS += 1
        rc = hip.lib().nr_token_weights_fwd_pair(ctypes.byref(probs[0]), ctypes.byref(probs[1]), prec, hip.stream_ptr())
        if rc == 0:
            return outs
        if rc != hip.NR_EUNSUPPORTED:
            hip._check("nr_token_weights_fwd_pair", rc)
        hip.N_CALLS -= 1
    return [token_weights(*c, prec) for c in calls]


def token_weights_group(calls, precs):
    """Up to four token_weights calls, each in its own precision, in ONE launch (nr_token_weights_fwd_group: 192 x 256 blocks,
    split-bf16 sets as three accumulated passes).  calls: tuples (prep, w1_hi, w1_lo, b1, w2, b2, mask, n_samples, N); precs: one
    hip.PREC_* per call.  -> [(w, None), ...], or None when a set does not fit the grouped form (nothing launched: the caller
    issues the calls one by one)."""
    import ctypes
    if not FUSE_TOKEN_SOFTMAX or not 0 < len(calls) <= 4:
        return None
    dev = calls[0][0].hi.device
    arr = (hip.TokenWeightsProblem * len(calls))()
    outs, keep = [], []
    for q, (prep, w1_hi, w1_lo, b1, w2, b2, mask, n_samples, N) in zip(arr, calls):
        counters = _softmax_counters(dev, prep.n_tok)
        if counters is None:
            return None
        H = w1_hi.shape[0]
        parts = torch.empty((H // 128, prep.n_tok), dtype=torch.float32, device=dev)
        w = torch.empty((n_samples, N), dtype=torch.float32, device=dev)
        m = _f32(mask).contiguous() if mask is not None else None
        q.tok_hi, q.tok_lo, q.norm = hip.ptr(prep.hi), hip.ptr(prep.lo, allow_none=True), hip.ptr(prep.norm)
        q.w1_hi, q.w1_lo = hip.ptr(w1_hi), hip.ptr(w1_lo, allow_none=True)
        q.b1, q.w2, q.b2 = hip.ptr(b1, torch.float32), hip.ptr(w2, torch.float32), hip.ptr(b2, torch.float32)
        q.mask, q.logit_part, q.counters = hip.ptr(m, allow_none=True), hip.ptr(parts), hip.ptr(counters)
        q.w, q.logits = hip.ptr(w), None
        q.n_samples, q.N, q.d, q.H, q.n_counters = int(n_samples), int(N), int(prep.d), int(H), counters.numel()
        outs.append((w, None))
        keep += [parts, m]
    pa = (ctypes.c_int * len(calls))(*[int(p) for p in precs])
    hip.N_CALLS += 1
    rc = hip.lib().nr_token_weights_fwd_group(arr, pa, len(calls), hip.stream_ptr())
    if rc == 0:
        return outs
    hip.N_CALLS -= 1
    if rc != hip.NR_EUNSUPPORTED:
        hip._check("nr_token_weights_fwd_group", rc)
    return None


def local_level(prep_t, prep_v, w_t, w_v, A, Nt, Bv, Nv, prec=hip.PREC_BF16, out_mode=hip.OUT_FULL, want_arg=False):
    """Fused token-token similarity (nr_local_level_fwd).  Returns (out, aux); aux = None or
    (arg_v, arg_t, pmax, qmax) kept for the backward pass."""
    dev = prep_t.hi.device
    if prep_t.n_tok != A * Nt or prep_v.n_tok != Bv * Nv:
        raise ValueError("prepared token counts do not match A*Nt / Bv*Nv")
    if out_mode == hip.OUT_FULL:
        out = torch.empty((A, Bv), dtype=torch.float32, device=dev)
    else:
        nr, nc = hip.local_level_tiles(A, Nt, Bv, Nv, prec)
        out = torch.empty((nc, A) if out_mode == hip.OUT_ROWSUM else (nr, Bv), dtype=torch.float32, device=dev)
    arg_v = arg_t = pmax = qmax = None
    if want_arg:
        arg_v = torch.empty((A, Bv, Nt), dtype=torch.uint8, device=dev)
        arg_t = torch.empty((A, Bv, Nv), dtype=torch.uint8, device=dev)
        pmax = torch.empty((A, Bv, Nt), dtype=torch.float32, device=dev)
        qmax = torch.empty((A, Bv, Nv), dtype=torch.float32, device=dev)
    hip.call("nr_local_level_fwd", hip.ptr(prep_t.hi), hip.ptr(prep_t.lo, allow_none=True), hip.ptr(prep_v.hi),
             hip.ptr(prep_v.lo, allow_none=True), hip.ptr(w_t, torch.float32), hip.ptr(w_v, torch.float32),
             A, Nt, Bv, Nv, prep_t.d, prec, out_mode, hip.ptr(out), hip.ptr(arg_v, allow_none=True),
             hip.ptr(arg_t, allow_none=True), hip.ptr(pmax, allow_none=True), hip.ptr(qmax, allow_none=True),
             hip.stream_ptr())
    return out, ((arg_v, arg_t, pmax, qmax) if want_arg else None)


def local_level_group(problems):
    """Several loss-only products of one step: [(prep_t, prep_v, w_t, w_v, A, Nt, Bv, Nv, prec, out_mode), ...] -> list of
    outputs as local_level(...)[0] would return them.  One grid for all of them (nr_local_level_group) when every product
    can join a group; one launch each otherwise."""
    if not problems:
        return []
    groupable = 1 < len(problems) <= hip.LOCAL_LEVEL_GROUP_MAX and all(
        hip.local_level_group_kind(A, Nt, Bv, Nv, pt.d, prec) >= 0 for (pt, pv, wt, wv, A, Nt, Bv, Nv, prec, mode) in problems)
    if not groupable:
        return [local_level(*q)[0] for q in problems]
    arr = (hip.LocalLevelProblem * len(problems))()
    outs = []
    for i, (pt, pv, wt, wv, A, Nt, Bv, Nv, prec, mode) in enumerate(problems):
        dev = pt.hi.device
        if pt.n_tok != A * Nt or pv.n_tok != Bv * Nv:
            raise ValueError("prepared token counts do not match A*Nt / Bv*Nv")
        if mode == hip.OUT_FULL:
            out = torch.empty((A, Bv), dtype=torch.float32, device=dev)
        else:
            nr, nc = hip.local_level_tiles(A, Nt, Bv, Nv, prec)
            out = torch.empty((nc, A) if mode == hip.OUT_ROWSUM else (nr, Bv), dtype=torch.float32, device=dev)
        outs.append(out)
        q = arr[i]
        q.t_hi, q.v_hi = hip.ptr(pt.hi), hip.ptr(pv.hi)
        q.t_lo, q.v_lo = hip.ptr(pt.lo, allow_none=True), hip.ptr(pv.lo, allow_none=True)
        q.w_t, q.w_v, q.out = hip.ptr(wt, torch.float32), hip.ptr(wv, torch.float32), hip.ptr(out)
        q.A, q.Nt, q.Bv, q.Nv, q.d, q.prec, q.out_mode = A, Nt, Bv, Nv, pt.d, int(prec), int(mode)
    hip.call("nr_local_level_group", len(problems), arr, hip.stream_ptr())
    return outs


def reduce_parts(parts, scale):
    out = torch.empty((parts.shape[1],), dtype=torch.float32, device=parts.device)
    hip.call("nr_reduce_parts", hip.ptr(parts, torch.float32), parts.shape[0], parts.shape[1], float(scale), hip.ptr(out),
             hip.stream_ptr())
    return out


def colsum_pair(a, scale_a, b, scale_b):
    """(scale_a * column sums of a, scale_b * column sums of b) in one launch (nr_colsum_group); fixed summation order."""
    from .cluster_backward_hip import _colsum_group
    oa = torch.empty((a.shape[1],), dtype=torch.float32, device=a.device)
    ob = torch.empty((b.shape[1],), dtype=torch.float32, device=b.device)
    _colsum_group([(a, oa, a.shape[0], a.shape[1], scale_a), (b, ob, b.shape[0], b.shape[1], scale_b)])
    return oa, ob


def gemm_nt_f32(a, b):
    a, b = _f32(a).contiguous(), _f32(b).contiguous()
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    hip.call("nr_gemm_nt_f32", hip.ptr(a), hip.ptr(b), M, N, K, hip.ptr(c), hip.stream_ptr())
    return c


def centrality_weights(g, colsum, n_tok, scale, want_aux=False):
    g = _f32(g).contiguous()
    B, d = g.shape
    w = torch.empty((B,), dtype=torch.float32, device=g.device)
    gnorm = torch.empty((B,), dtype=torch.float32, device=g.device) if want_aux else None
    mean = torch.empty((d,), dtype=torch.float32, device=g.device) if want_aux else None
    hip.call("nr_centrality_weights", hip.ptr(g), B, d, hip.ptr(colsum, torch.float32), colsum.shape[0], n_tok,
             float(scale), hip.ptr(w), hip.ptr(gnorm, allow_none=True), hip.ptr(mean, allow_none=True), hip.stream_ptr())
    return w, gnorm, mean


def centrality_weights_pair(gt, gv, mean_t, mean_v, scale, want_aux=False):
    """(w_text [B], w_video [B], aux) from finished token means (nr_centrality_weights_pair).
    gt [B,d] or [B,Gt,d], gv likewise: with several global tokens per sample the weight is the mean over them
    (config.centrality_multi_token = "mean").  aux = None or (gnorm_t, gnorm_v, wtok_t, wtok_v), per global token."""
    if gt.dim() == 2:
        gt, gv = gt[:, None, :], gv[:, None, :]
    B, n_gt, d = gt.shape
    n_gv = gv.shape[1]
    dev = gt.device
    w_t = torch.empty((B,), dtype=torch.float32, device=dev)
    w_v = torch.empty((B,), dtype=torch.float32, device=dev)
    aux = None
    if want_aux:
        aux = tuple(torch.empty((B * n,), dtype=torch.float32, device=dev) for n in (n_gt, n_gv, n_gt, n_gv))
    a = aux or (None,) * 4
    hip.call("nr_centrality_weights_pair", hip.ptr(gt, torch.float32), hip.ptr(gv, torch.float32), B, n_gt, n_gv, d,
             hip.ptr(mean_t, torch.float32), hip.ptr(mean_v, torch.float32), float(scale), hip.ptr(w_t), hip.ptr(w_v),
             hip.ptr(a[0], allow_none=True), hip.ptr(a[1], allow_none=True), hip.ptr(a[2], allow_none=True),
             hip.ptr(a[3], allow_none=True), hip.stream_ptr())
    return w_t, w_v, aux


def dpc_knn_assign(x, cluster_num, k, mask=None, noise=None):
    """Cluster id [B,N] (int64) of every token by DPC-KNN (nr_dpc_knn_assign)."""
    x = _f32(x.detach()).contiguous()
    B, N, C = x.shape
    if noise is None:
        noise = torch.rand((B, N), device=x.device, dtype=torch.float32)
    noise = _f32(noise).contiguous()
    m = _f32(mask).contiguous() if mask is not None else None
    assign = torch.empty((B, N), dtype=torch.int64, device=x.device)
    ws = torch.empty((int(hip.lib().nr_dpc_workspace_bytes(B, N)),), dtype=torch.uint8, device=x.device)
    hip.call("nr_dpc_knn_assign", hip.ptr(x), hip.ptr(m, allow_none=True), hip.ptr(noise), B, N, C, int(k), int(cluster_num),
             hip.ptr(assign), hip.ptr(ws), hip.stream_ptr())
    return assign


def sinkhorn_targets(G, beta, iters=50):
    G = _f32(G).contiguous()
    B = G.shape[0]
    tr = torch.empty_like(G)
    tc = torch.empty_like(G)
    ws = torch.empty((hip.sinkhorn_workspace_bytes(B),), dtype=torch.uint8, device=G.device)
    hip.call("nr_sinkhorn_targets", hip.ptr(G), B, float(beta), int(iters), hip.ptr(tr), hip.ptr(tc), hip.ptr(ws),
             hip.stream_ptr())
    return tr, tc


def sinkhorn_uniform_rows(G, beta, T, rowloss, iters=50):
    """Sinkhorn solve writing the uniform-CE row terms straight into rowloss[:, 1, :] (nr_sinkhorn_uniform_rows);
    the targets themselves are not materialised.  False if the shape is not covered (B > 128 or B % 4)."""
    G = _f32(G).contiguous()
    B = G.shape[0]
    if B > 128 or B % 4:
        return False
    ws = torch.empty((hip.sinkhorn_workspace_bytes(B),), dtype=torch.uint8, device=G.device)
    base = rowloss.view(-1)[B:]                    # rowloss [2,4,B]: term 1 of direction d starts at (4 d + 1) B
    hip.call("nr_sinkhorn_uniform_rows", hip.ptr(G), B, float(beta), int(iters), float(T), hip.ptr(base, torch.float32), 4 * B,
             None, None, hip.ptr(ws), hip.stream_ptr())
    return True


def split_tail_counter(dev, slot=0):
    """The zero-initialised device word shared by the two self-finalizing launches of the split tail (one per
    device: the launch that finishes last resets it, and steps on one device are ordered; steps whose tails may overlap --
    consecutive steps of a pipelined graph, modeling.StepPipeline -- take different `slot`s)."""
    key = ("split_tail", dev, int(slot))
    c = _COUNTERS.get(key)
    if c is None:
        c = _COUNTERS[key] = torch.zeros((1,), dtype=torch.int32, device=dev)
    return c


def forget_split_tail_counter(dev, slot=0):
    """Drops the device's shared finalize word (the next split_tail_counter call allocates a zeroed one)."""
    _COUNTERS.pop(("split_tail", dev, int(slot)), None)


def sinkhorn_uniform_rows_final(G, beta, T, rowloss, counter, wu, wn, wkl, losses, iters=50):
    """nr_sinkhorn_uniform_rows_final: Sinkhorn + uniform-CE row terms into rowloss[:, 1, :], taking part in the shared
    finalize (see row_losses_no_uniform_final)."""
    G = _f32(G).contiguous()
    B = G.shape[0]
    ws = torch.empty((hip.sinkhorn_workspace_bytes(B),), dtype=torch.uint8, device=G.device)
    hip.call("nr_sinkhorn_uniform_rows_final", hip.ptr(G), B, float(beta), int(iters), float(T), hip.ptr(rowloss, torch.float32),
             hip.ptr(counter, torch.int32), float(wu), float(wn), float(wkl), hip.ptr(losses, torch.float32), hip.ptr(ws),
             hip.stream_ptr())


def row_losses_no_uniform_final(S, G, c0_parts, c1_parts, c_scale, wc_text, wc_video, logit_scale, K, T, rowloss, counter,
                                wu, wn, wkl, losses):
    """nr_row_losses_fwd_no_uniform_final: the other row terms from the bank products' partial sums; the workgroup (of
    either launch) that finishes last writes the five losses."""
    B = S.shape[0]
    hip.call("nr_row_losses_fwd_no_uniform_final", hip.ptr(S, torch.float32), hip.ptr(G, torch.float32),
             hip.ptr(c0_parts, torch.float32), c0_parts.shape[0], hip.ptr(c1_parts, torch.float32), c1_parts.shape[0],
             float(c_scale), hip.ptr(wc_text, torch.float32), hip.ptr(wc_video, torch.float32),
             hip.ptr(logit_scale, torch.float32), B, int(K), float(T), hip.ptr(rowloss, torch.float32),
             hip.ptr(counter, torch.int32), float(wu), float(wn), float(wkl), hip.ptr(losses, torch.float32), hip.stream_ptr())


def row_losses_no_uniform_final_cw(S, G, c0_parts, c1_parts, c_scale, g_text, g_video, mean_text, mean_video, centrality_scale,
                                   logit_scale, K, T, rowloss, counter, wu, wn, wkl, losses):
    """nr_row_losses_fwd_no_uniform_final_cw: row_losses_no_uniform_final with the centrality weights computed by the row's own
    wave (one global token per sample: g_text / g_video [B, d], mean_text / mean_video [d])."""
    B, d = S.shape[0], g_text.shape[-1]
    hip.call("nr_row_losses_fwd_no_uniform_final_cw", hip.ptr(S, torch.float32), hip.ptr(G, torch.float32),
             hip.ptr(c0_parts, torch.float32), c0_parts.shape[0], hip.ptr(c1_parts, torch.float32), c1_parts.shape[0],
             float(c_scale), hip.ptr(g_text, torch.float32), hip.ptr(g_video, torch.float32), hip.ptr(mean_text, torch.float32),
             hip.ptr(mean_video, torch.float32), int(d), float(centrality_scale), hip.ptr(logit_scale, torch.float32), B, int(K),
             float(T), hip.ptr(rowloss, torch.float32), hip.ptr(counter, torch.int32), float(wu), float(wn), float(wkl),
             hip.ptr(losses, torch.float32), hip.stream_ptr())


def row_losses_no_uniform(S, G, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T, rowloss):
    """Centrality / neighbour / KL row terms into rowloss[:, (0, 2, 3), :] (nr_row_losses_fwd_no_uniform)."""
    B = S.shape[0]
    hip.call("nr_row_losses_fwd_no_uniform", hip.ptr(S, torch.float32), hip.ptr(G, torch.float32), hip.ptr(bank_c0, torch.float32),
             hip.ptr(bank_c1, torch.float32), hip.ptr(wc_text, torch.float32), hip.ptr(wc_video, torch.float32),
             hip.ptr(logit_scale, torch.float32), B, int(K), float(T), hip.ptr(rowloss, torch.float32), hip.stream_ptr())


def row_losses(S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T):
    B = S.shape[0]
    rowloss = torch.empty((2, 4, B), dtype=torch.float32, device=S.device)
    hip.call("nr_row_losses_fwd", hip.ptr(S, torch.float32), hip.ptr(G, torch.float32), hip.ptr(tgt_rows), hip.ptr(tgt_cols),
             hip.ptr(bank_c0, torch.float32), hip.ptr(bank_c1, torch.float32), hip.ptr(wc_text, torch.float32),
             hip.ptr(wc_video, torch.float32), hip.ptr(logit_scale, torch.float32), B, int(K), float(T),
             hip.ptr(rowloss), hip.stream_ptr())
    return rowloss


def row_losses_slab(S_rows, S_cols, row0, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T):
    """Row terms [2,4,B] of the rows [row0, row0 + n) only (zero elsewhere) from the two slabs of S a rank owns."""
    n, B = S_rows.shape
    rowloss = torch.zeros((2, 4, B), dtype=torch.float32, device=S_rows.device)
    hip.call("nr_row_losses_fwd_slab", hip.ptr(S_rows, torch.float32), hip.ptr(S_cols, torch.float32), int(row0), int(n),
             hip.ptr(G, torch.float32), hip.ptr(tgt_rows), hip.ptr(tgt_cols), hip.ptr(bank_c0, torch.float32),
             hip.ptr(bank_c1, torch.float32), hip.ptr(wc_text, torch.float32), hip.ptr(wc_video, torch.float32),
             hip.ptr(logit_scale, torch.float32), B, int(K), float(T), hip.ptr(rowloss), hip.stream_ptr())
    return rowloss


def row_losses_bwd_slab(S_rows, S_cols, row0, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T, g_rowloss):
    """Backward of row_losses_slab (nr_row_losses_bwd_slab) -> dS_dir, dG_dir, d_c_rows [2,n,B], d_wc, d_ls_rows [2,n]."""
    n, B = S_rows.shape
    dev = S_rows.device
    dS = torch.empty((2, n, B), dtype=torch.float32, device=dev)
    dG = torch.empty((2, n, B), dtype=torch.float32, device=dev)
    dC = torch.empty((2, n, B), dtype=torch.float32, device=dev)
    dwc = torch.empty((2, n), dtype=torch.float32, device=dev)
    dls = torch.empty((2, n), dtype=torch.float32, device=dev)
    hip.call("nr_row_losses_bwd_slab", hip.ptr(S_rows, torch.float32), hip.ptr(S_cols, torch.float32), int(row0), int(n),
             hip.ptr(G, torch.float32), hip.ptr(tgt_rows), hip.ptr(tgt_cols), hip.ptr(bank_c0, torch.float32),
             hip.ptr(bank_c1, torch.float32), hip.ptr(wc_text, torch.float32), hip.ptr(wc_video, torch.float32),
             hip.ptr(logit_scale, torch.float32), B, int(K), float(T), hip.ptr(g_rowloss, torch.float32), hip.ptr(dS), hip.ptr(dG),
             hip.ptr(dC), hip.ptr(dwc), hip.ptr(dls), hip.stream_ptr())
    return dS, dG, dC, dwc, dls


def row_losses_final(S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T, wu, wn, wkl):
    """Row terms [2,4,B] AND the five losses from one launch (nr_row_losses_fwd_final)."""
    B = S.shape[0]
    dev = S.device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)     # one counter per stream: launches on one stream are ordered
    counter = _COUNTERS.get(key)
    if counter is None:
        counter = _COUNTERS[key] = torch.zeros((1,), dtype=torch.int32, device=dev)
    rowloss = torch.empty((2, 4, B), dtype=torch.float32, device=dev)
    losses = torch.empty((5,), dtype=torch.float32, device=dev)
    hip.call("nr_row_losses_fwd_final", hip.ptr(S, torch.float32), hip.ptr(G, torch.float32), hip.ptr(tgt_rows), hip.ptr(tgt_cols),
             hip.ptr(bank_c0, torch.float32), hip.ptr(bank_c1, torch.float32), hip.ptr(wc_text, torch.float32),
             hip.ptr(wc_video, torch.float32), hip.ptr(logit_scale, torch.float32), B, int(K), float(T),
             hip.ptr(rowloss), hip.ptr(counter), float(wu), float(wn), float(wkl), hip.ptr(losses), hip.stream_ptr())
    return rowloss, losses


def loss_finalize(rowloss, wu, wn, wkl):
    B = rowloss.shape[-1]
    losses = torch.empty((5,), dtype=torch.float32, device=rowloss.device)
    hip.call("nr_loss_finalize", hip.ptr(rowloss), B, float(wu), float(wn), float(wkl), hip.ptr(losses), hip.stream_ptr())
    return losses


def row_losses_bwd(S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T, g_rowloss):
    """-> dS_dir [2,B,B], dG_dir [2,B,B], d_c_rows [2,B,B], d_wc [2,B], d_ls_rows [2,B]  (nr_row_losses_bwd)."""
    B = S.shape[0]
    dev = S.device
    dS = torch.empty((2, B, B), dtype=torch.float32, device=dev)
    dG = torch.empty((2, B, B), dtype=torch.float32, device=dev)
    dC = torch.empty((2, B, B), dtype=torch.float32, device=dev)
    dwc = torch.empty((2, B), dtype=torch.float32, device=dev)
    dls = torch.empty((2, B), dtype=torch.float32, device=dev)
    hip.call("nr_row_losses_bwd", hip.ptr(S, torch.float32), hip.ptr(G, torch.float32), hip.ptr(tgt_rows), hip.ptr(tgt_cols),
             hip.ptr(bank_c0, torch.float32), hip.ptr(bank_c1, torch.float32), hip.ptr(wc_text, torch.float32),
             hip.ptr(wc_video, torch.float32), hip.ptr(logit_scale, torch.float32), B, int(K), float(T),
             hip.ptr(g_rowloss, torch.float32), hip.ptr(dS), hip.ptr(dG), hip.ptr(dC), hip.ptr(dwc), hip.ptr(dls),
             hip.stream_ptr())
    return dS, dG, dC, dwc, dls


def rowloss_coef(gs, hp, B):
    """g_rowloss [2,4,B] from the gradients of the five losses (tensors or None) in one launch (nr_rowloss_coef)."""
    dev = next(g for g in gs if g is not None).device
    gs = [None if g is None else _f32(g).contiguous() for g in gs]
    coef = torch.empty((2, 4, B), dtype=torch.float32, device=dev)
    hip.call("nr_rowloss_coef", *[hip.ptr(g, allow_none=True) for g in gs], float(hp["uniform_weight"]),
             float(hp["neighbor_weight"]), float(hp["kl_weight"]), int(B), hip.ptr(coef), hip.stream_ptr())
    return coef


def rowloss_bwd_finish(dS_dir, dG_dir, dC_rows, dls_rows):
    """-> dS [B,B], dG [B,B], d_c0 [B], d_c1 [B], d_ls [] in one launch (nr_rowloss_bwd_finish)."""
    B = dS_dir.shape[1]
    f32 = dict(dtype=torch.float32, device=dS_dir.device)
    dS, dG = torch.empty((B, B), **f32), torch.empty((B, B), **f32)
    dc = torch.empty((2, B), **f32)
    d_ls = torch.empty((1,), **f32)
    hip.call("nr_rowloss_bwd_finish", hip.ptr(dS_dir, torch.float32), hip.ptr(dG_dir, torch.float32), hip.ptr(dC_rows, torch.float32),
             hip.ptr(dls_rows, torch.float32), B, hip.ptr(dS), hip.ptr(dG), hip.ptr(dc[0]), hip.ptr(dc[1]), hip.ptr(d_ls),
             hip.stream_ptr())
    return dS, dG, dc[0], dc[1], d_ls[0]


def centrality_weights_bwd_pair(g_t, gnorm_t, mean_t, w_t, dw_t, g_v, gnorm_v, mean_v, w_v, dw_v, scale):
    """centrality_weights_bwd for both modalities in one launch -> dg_t, dmean_t, dg_v, dmean_v."""
    B, d = g_t.shape
    if g_v.shape != g_t.shape:
        raise ValueError("text and video global tokens of one batch have the same shape")
    dg_t, dg_v = torch.empty_like(g_t), torch.empty_like(g_v)
    dmean = torch.empty((2, d), dtype=torch.float32, device=g_t.device)
    hip.call("nr_centrality_weights_bwd_pair", hip.ptr(g_t, torch.float32), hip.ptr(gnorm_t), hip.ptr(mean_t), hip.ptr(w_t),
             hip.ptr(dw_t.contiguous(), torch.float32), hip.ptr(g_v, torch.float32), hip.ptr(gnorm_v), hip.ptr(mean_v), hip.ptr(w_v),
             hip.ptr(dw_v.contiguous(), torch.float32), B, d, float(scale), hip.ptr(dg_t), hip.ptr(dmean[0]), hip.ptr(dg_v),
             hip.ptr(dmean[1]), hip.stream_ptr())
    return dg_t, dmean[0], dg_v, dmean[1]


def global_logits_bwd(dG, gt2, gv2, add_t, add_v):
    """d_gt = dG gv + add_t, d_gv = dG^T gt + add_v in one launch (nr_global_logits_bwd)."""
    B, d = gt2.shape
    d_gt, d_gv = torch.empty_like(gt2), torch.empty_like(gv2)
    hip.call("nr_global_logits_bwd", hip.ptr(dG, torch.float32), hip.ptr(gt2, torch.float32), hip.ptr(gv2, torch.float32),
             hip.ptr(add_t, torch.float32), hip.ptr(add_v, torch.float32), B, d, hip.ptr(d_gt), hip.ptr(d_gv), hip.stream_ptr())
    return d_gt, d_gv


def add_transposed(a, b):
    """a + b.T for square fp32 matrices."""
    B = a.shape[0]
    out = torch.empty((B, B), dtype=torch.float32, device=a.device)
    hip.call("nr_add_transposed", hip.ptr(a, torch.float32), hip.ptr(b, torch.float32), B, hip.ptr(out), hip.stream_ptr())
    return out


def colsum(a):
    rows, cols = a.shape
    out = torch.empty((cols,), dtype=torch.float32, device=a.device)
    hip.call("nr_colsum", hip.ptr(a, torch.float32), rows, cols, hip.ptr(out), hip.stream_ptr())
    return out


USE_MFMA_BACKWARD = True        # False: the scalar arg-max walk for d_x too (reference for the MFMA path in the tests)


def transpose_prepared(preps, use_lo=True):
    """Several Prepared token sets in the order nr_local_level_bwd_group reads its "other" operand (fragment-major, whole
    slices of 96 tokens) in ONE launch (nr_sim_bwd_operand_group) -> [(hi_f, lo_f or None, tokens covered), ...]."""
    out = []
    for k in range(0, len(preps), hip.SIM_BWD_GROUP_MAX):
        chunk = preps[k:k + hip.SIM_BWD_GROUP_MAX]
        arr = (hip.SimBwdOperand * len(chunk))()
        for a, p in zip(arr, chunk):
            n_tok, d = p.hi.view(-1, p.d).shape
            ldk = (n_tok + 95) // 96 * 96
            dev = p.hi.device
            hi_f = torch.empty((ldk * d,), dtype=torch.int16, device=dev)
            lo_f = torch.empty((ldk * d,), dtype=torch.int16, device=dev) if (use_lo and p.lo is not None) else None
            a.hi, a.lo = p.hi.data_ptr(), (p.lo.data_ptr() if p.lo is not None else None)
            a.out_hi, a.out_lo = hi_f.data_ptr(), (lo_f.data_ptr() if lo_f is not None else None)
            a.n_tok, a.d = int(n_tok), int(d)
            out.append((hi_f, lo_f, ldk))
        hip.call("nr_sim_bwd_operand_group", len(chunk), arr, hip.stream_ptr())
    return out


def local_level_bwd_group(items, use_lo=True):
    """d_x of up to four fused products in one launch + one summing launch (nr_local_level_bwd_group).  items: dicts with
        side, dS, ds_mode, ds_scale, other_T=(hi_t, lo_t, ldk) from transpose_prepared, w_self, w_other, aux, A, Nt, Bv, Nv,
        d_x [n_self_tokens, d] f32 (items with the same d_x are summed), accumulate (first item of a d_x: keep its content)."""
    if not 0 < len(items) <= hip.SIM_BWD_GROUP_MAX:
        raise ValueError("1..%d products per launch" % hip.SIM_BWD_GROUP_MAX)
    arr = (hip.SimBwdItem * len(items))()
    keep = []
    for a, it in zip(arr, items):
        hi_t, lo_t, ldk = it["other_T"]
        arg_v, arg_t = it["aux"][0], it["aux"][1]
        dS = _f32(it["dS"]).contiguous()
        w_s, w_o = _f32(it["w_self"]).contiguous(), _f32(it["w_other"]).contiguous()
        keep += [dS, w_s, w_o]
        a.dS, a.oT_hi, a.oT_lo = dS.data_ptr(), hi_t.data_ptr(), (lo_t.data_ptr() if (use_lo and lo_t is not None) else None)
        a.w_self, a.w_other, a.arg_v, a.arg_t = w_s.data_ptr(), w_o.data_ptr(), arg_v.data_ptr(), arg_t.data_ptr()
        a.d_x, a.ds_scale = it["d_x"].data_ptr(), float(it["ds_scale"])
        a.side, a.ds_mode, a.ldk = int(it["side"]), int(it["ds_mode"]), int(ldk)
        a.A, a.Nt, a.Bv, a.Nv, a.d = int(it["A"]), int(it["Nt"]), int(it["Bv"]), int(it["Nv"]), int(hi_t.numel() // ldk)
        a.accumulate = 1 if it.get("accumulate") else 0
    nbytes = int(hip.lib().nr_local_level_bwd_group_workspace_bytes(len(items), arr))
    if nbytes == 0:
        raise hip.NrHipError("nr_local_level_bwd_group: shape outside the matrix-core backward (nr_local_level_bwd_mfma_supported)")
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=items[0]["d_x"].device)
    hip.call("nr_local_level_bwd_group", len(items), arr, hip.ptr(ws), nbytes, hip.stream_ptr())


def pool_weight_bwd_group(jobs):
    """d_w of several token-weight vectors in one launch (nr_pool_weight_bwd_group).  jobs: dicts with side, N, d_w [n_self*N]
    f32, accumulate, srcs = [(dS, ds_mode, ds_scale, pooled [A,Bv,N] f32, A, Bv), ...] (one or two products)."""
    if not 0 < len(jobs) <= hip.POOLW_GROUP_MAX:
        raise ValueError("1..%d jobs per launch" % hip.POOLW_GROUP_MAX)
    arr = (hip.PoolWJob * len(jobs))()
    keep = []
    for a, j in zip(arr, jobs):
        a.n_src, a.side, a.N, a.accumulate = len(j["srcs"]), int(j["side"]), int(j["N"]), 1 if j.get("accumulate") else 0
        a.d_w = j["d_w"].data_ptr()
        for k, (dS, mode, scale, pool, A, Bv) in enumerate(j["srcs"]):
            dS = _f32(dS).contiguous()
            keep.append(dS)
            a.src[k].dS, a.src[k].pool, a.src[k].ds_scale = dS.data_ptr(), pool.data_ptr(), float(scale)
            a.src[k].ds_mode, a.src[k].A, a.src[k].Bv = int(mode), int(A), int(Bv)
    hip.call("nr_pool_weight_bwd_group", len(jobs), arr, hip.stream_ptr())


def local_level_bwd(side, dS, ds_mode, ds_scale, other, w_self, w_other, aux, A, Nt, Bv, Nv, d_x=None, d_w=None,
                    want_dx=True, accumulate=False, use_lo=True, other_T=None, scalar_dw=False):
    """Arg-max-routed gradient for one operand.  `other` is the Prepared token set of the opposite operand.
    d_x: nr_local_level_bwd_group (matrix cores) where the token counts allow, else nr_local_level_bwd's entry-by-entry walk;
    d_w: nr_pool_weight_bwd_group (scalar_dw=True: the walk's own sum, kept as the cross-check of the tests).
    Returns (d_x [n_self*N, d] or None, d_w [n_self*N])."""
    arg_v, arg_t, pmax, qmax = aux
    n_self = (A * Nt) if side == 0 else (Bv * Nv)
    d = other.d
    dev = w_self.device
    if accumulate and ((want_dx and d_x is None) or d_w is None):
        raise ValueError("accumulate=True needs existing d_x / d_w buffers")
    if want_dx and d_x is None:
        d_x = torch.empty((n_self, d), dtype=torch.float32, device=dev)
    if d_w is None:
        d_w = torch.empty((n_self,), dtype=torch.float32, device=dev)
    mfma = want_dx and USE_MFMA_BACKWARD and bool(hip.lib().nr_local_level_bwd_mfma_supported(Nt, Nv, d))
    if mfma:
        if other_T is None:                          # else: transposed up front, several operands in one launch
            other_T = transpose_prepared([other], use_lo=use_lo)[0]
        local_level_bwd_group([dict(side=side, dS=dS, ds_mode=ds_mode, ds_scale=ds_scale, other_T=other_T, w_self=w_self,
                                    w_other=w_other, aux=aux, A=A, Nt=Nt, Bv=Bv, Nv=Nv, d_x=d_x, accumulate=accumulate)],
                              use_lo=use_lo)
    scalar_dx = want_dx and not mfma
    if scalar_dx or scalar_dw:
        ws = torch.empty((int(hip.lib().nr_local_level_bwd_workspace_bytes(int(side), A, Nt, Bv, Nv, d)),), dtype=torch.uint8,
                         device=dev)
        hip.call("nr_local_level_bwd", int(side), hip.ptr(dS, torch.float32), int(ds_mode), float(ds_scale),
                 hip.ptr(other.hi), hip.ptr(other.lo if use_lo else None, allow_none=True), hip.ptr(w_self, torch.float32),
                 hip.ptr(w_other, torch.float32), hip.ptr(arg_v), hip.ptr(arg_t), hip.ptr(pmax), hip.ptr(qmax),
                 A, Nt, Bv, Nv, d, hip.ptr(d_x if scalar_dx else None, allow_none=True),
                 hip.ptr(d_w if scalar_dw else None, allow_none=True), 1 if accumulate else 0, hip.ptr(ws), hip.stream_ptr())
    if not scalar_dw:
        pool_weight_bwd_group([dict(side=side, N=Nt if side == 0 else Nv, d_w=d_w, accumulate=accumulate,
                                    srcs=[(dS, ds_mode, ds_scale, pmax if side == 0 else qmax, A, Bv)])])
    return (d_x if want_dx else None), d_w


def normalize_bwd(x, norm, mask, d_xn, dmean):
    x = _f32(x).contiguous()
    d = x.shape[-1]
    n_tok = x.numel() // d
    dx = torch.empty_like(x)
    m = _f32(mask).contiguous() if mask is not None else None
    hip.call("nr_normalize_bwd", hip.ptr(x), hip.ptr(norm, torch.float32), hip.ptr(m, allow_none=True),
             hip.ptr(d_xn, allow_none=True), hip.ptr(dmean, allow_none=True), n_tok, d, hip.ptr(dx), hip.stream_ptr())
    return dx


def token_softmax_bwd(w, dw):
    n, N = w.shape
    out = torch.empty_like(w)
    hip.call("nr_token_softmax_bwd", hip.ptr(w, torch.float32), hip.ptr(dw.contiguous(), torch.float32), n, N, hip.ptr(out),
             hip.stream_ptr())
    return out


def centrality_weights_bwd(g, gnorm, mean, w, dw, scale):
    B, d = g.shape
    dg = torch.empty_like(g)
    dmean = torch.empty((d,), dtype=torch.float32, device=g.device)
    hip.call("nr_centrality_weights_bwd", hip.ptr(g, torch.float32), hip.ptr(gnorm), hip.ptr(mean), hip.ptr(w),
             hip.ptr(dw.contiguous(), torch.float32), B, d, float(scale), hip.ptr(dg), hip.ptr(dmean), hip.stream_ptr())
    return dg, dmean


def bank_push(bank, batch, scratch=None):
    """In-place FIFO push: bank <- cat(batch, bank)[:capacity]  (nr_bank_push)."""
    cap, n_new = bank.shape[0], batch.shape[0]
    row_bytes = bank[0].numel() * bank.element_size()
    if batch.dtype != bank.dtype or batch[0].numel() != bank[0].numel():
        raise ValueError("bank / batch row layout mismatch")
    if scratch is None and n_new < cap:
        scratch = torch.empty_like(bank)
    hip.call("nr_bank_push", hip.ptr(bank), hip.ptr(batch.contiguous()), cap, n_new, row_bytes,
             hip.ptr(scratch, allow_none=True), hip.stream_ptr())
    return bank


def mask_piece(mask):
    """(tensor, kind) of a mask for pack_shard: int64 / fp32 masks are converted to the record's u8 by the pack kernel itself
    (kind 1 / 2), u8 ones travel as they are; anything else is converted here."""
    m = mask.contiguous()
    if m.dtype == torch.int64:
        return m, 1
    if m.dtype == torch.float32:
        return m, 2
    return (m if m.dtype == torch.uint8 else m.to(torch.uint8)), 0


def pack_shard(tensors, packed, offsets, kinds=None):
    """nr_pack_shard(_convert): the (contiguous GPU) tensors' bytes at `offsets` inside the uint8 buffer `packed`; a piece of
    kind 1 / 2 (mask_piece) is converted to u8 on the way and takes one byte per ELEMENT."""
    import ctypes
    n = len(tensors)
    for t in tensors:
        hip.ptr(t)
    kinds = list(kinds) if kinds is not None else [0] * n
    srcs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nbytes = (ctypes.c_size_t * n)(*[t.numel() * (1 if k else t.element_size()) for t, k in zip(tensors, kinds)])
    offs = (ctypes.c_size_t * n)(*offsets)
    if any(kinds):
        hip.call("nr_pack_shard_convert", n, srcs, nbytes, offs, (ctypes.c_int * n)(*kinds), hip.ptr(packed, torch.uint8), hip.stream_ptr())
    else:
        hip.call("nr_pack_shard", n, srcs, nbytes, offs, hip.ptr(packed, torch.uint8), hip.stream_ptr())


def copy_group(dsts, srcs):
    """nr_copy_group: dsts[k] <- srcs[k] (contiguous GPU tensors of equal size and dtype, at most 12) in one launch."""
    import ctypes
    n = len(dsts)
    for d_, s_ in zip(dsts, srcs):
        hip.ptr(d_)
        hip.ptr(s_)
        if d_.dtype != s_.dtype or d_.numel() != s_.numel():
            raise ValueError("copy_group: source and destination differ in size or dtype")
    hip.call("nr_copy_group", n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs]), (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts]),
             (ctypes.c_size_t * n)(*[t.numel() * t.element_size() for t in srcs]), hip.stream_ptr())


def unpack_gathered(gathered, world, record_bytes, nbytes, offsets, outs, u8_to_f32):
    """nr_unpack_gathered: [world, record_bytes] uint8 -> the rank-major output tensors `outs`."""
    import ctypes
    n = len(outs)
    for t in outs:
        hip.ptr(t)
    hip.call("nr_unpack_gathered", n, hip.ptr(gathered, torch.uint8), int(world), int(record_bytes),
             (ctypes.c_size_t * n)(*nbytes), (ctypes.c_size_t * n)(*offsets), (ctypes.c_void_p * n)(*[t.data_ptr() for t in outs]),
             (ctypes.c_int * n)(*[1 if c else 0 for c in u8_to_f32]), hip.stream_ptr())


def bank_absorb_gathered(recv, lay, bank, shadow, ring_head, capacity, rng_state):
    """nr_bank_absorb_gathered: the gathered batch in `recv` (dist.packed_gather_raw) goes straight into the memory bank's ring --
    fp32 rows, masks, ids -- and, prepared, into its bf16 shadow; the ring head moves back by the batch and the noise stream's
    counter advances.  bank: dict of the five mb_* tensors; shadow: (text Prepared, video Prepared) or None."""
    dev = recv.device
    key = ("absorb", dev)
    counter = _COUNTERS.get(key)
    if counter is None:                 # the launch's two-level ticket: nr_bank_absorb_counter_words() zeroed words, left zeroed
        counter = _COUNTERS[key] = torch.zeros((hip.lib().nr_bank_absorb_counter_words(),), dtype=torch.int32, device=dev)
    a = hip.BankAbsorbDesc()
    a.gathered = hip.ptr(recv, torch.uint8)
    a.record_bytes = lay["record"]
    a.off_text, a.off_video, a.off_index, a.off_text_mask, a.off_video_mask = lay["offs"]
    (Nt, d), (Nv, _) = lay["shapes"][0], lay["shapes"][1]
    a.world, a.per_rank, a.Nt, a.Nv, a.d, a.capacity = lay["W"], lay["b"], Nt, Nv, d, int(capacity)
    a.bank_text, a.bank_video = hip.ptr(bank["mb_feat_t"], torch.float32), hip.ptr(bank["mb_feat_v"], torch.float32)
    a.bank_text_mask, a.bank_video_mask = hip.ptr(bank["mb_mask_t"], torch.float32), hip.ptr(bank["mb_mask_v"], torch.float32)
    a.bank_index = hip.ptr(bank["mb_ind"], torch.int64)
    if shadow is not None:
        st, sv = shadow
        a.shadow_text_hi, a.shadow_text_lo, a.shadow_text_norm = hip.ptr(st.hi), hip.ptr(st.lo), hip.ptr(st.norm, torch.float32)
        a.shadow_video_hi, a.shadow_video_lo, a.shadow_video_norm = hip.ptr(sv.hi), hip.ptr(sv.lo), hip.ptr(sv.norm, torch.float32)
    a.ring_head = hip.ptr(ring_head, torch.int32)
    a.rng_state = hip.ptr(rng_state, torch.int64)
    a.counter = hip.ptr(counter, torch.int32)
    hip.call("nr_bank_absorb_gathered", a, hip.stream_ptr())


def step_prologue(mask0, mask1, logit_scale, rng_state, n_noise, ring=None):
    """nr_step_prologue: (mask0 fp32, mask1 fp32, exp(logit_scale) [1] or None, noise [n_noise] or None).
    int64 masks are converted by the kernel; fp32 masks pass through untouched.
    ring = (head int32[1] device tensor, advance, capacity): the bank's ring head is moved back by `advance`."""
    dev = (mask0 if mask0 is not None else rng_state).device
    f32 = dict(dtype=torch.float32, device=dev)

    def conv(m):
        if m is None or m.dtype == torch.float32:
            return None, m
        if m.dtype != torch.int64:
            return None, m.float()
        m = m.contiguous()
        return m, torch.empty(m.shape, **f32)
    i0, o0 = conv(mask0)
    i1, o1 = conv(mask1)
    ls_exp = ls = None
    if logit_scale is not None:
        ls = logit_scale.detach().float().reshape(1).contiguous()
        ls_exp = torch.empty((1,), **f32)
    noise = torch.empty((n_noise,), **f32) if n_noise > 0 else None
    if i0 is not None or i1 is not None or ls is not None or noise is not None or ring is not None:
        hip.call("nr_step_prologue", hip.ptr(i0, allow_none=True), i0.numel() if i0 is not None else 0,
                 hip.ptr(o0 if i0 is not None else None, allow_none=True),
                 hip.ptr(i1, allow_none=True), i1.numel() if i1 is not None else 0,
                 hip.ptr(o1 if i1 is not None else None, allow_none=True),
                 hip.ptr(ls, allow_none=True), hip.ptr(ls_exp, allow_none=True),
                 hip.ptr(rng_state if noise is not None else None, torch.int64, allow_none=True),
                 hip.ptr(noise, allow_none=True), int(n_noise),
                 hip.ptr(ring[0] if ring else None, torch.int32, allow_none=True), int(ring[1]) if ring else 0,
                 int(ring[2]) if ring else 0, hip.stream_ptr())
    return o0, o1, ls_exp, noise


def bank_ring_push(banks, batches, head_new, head_dev=None):
    """Ring-buffer push of several bank tensors in one launch (nr_bank_ring_push).  head_dev: int32 [1] device
    tensor holding the head (overrides head_new)."""
    import ctypes
    n = len(banks)
    cap, n_new = banks[0].shape[0], batches[0].shape[0]
    bs = [b.contiguous() for b in batches]
    for bank, batch in zip(banks, bs):
        if bank.dtype != batch.dtype or bank[0].numel() != batch[0].numel() or bank.shape[0] != cap:
            raise ValueError("bank / batch layout mismatch")
    PA = ctypes.c_void_p * n
    pb = PA(*[b.data_ptr() for b in banks])
    pn = PA(*[b.data_ptr() for b in bs])
    rb = (ctypes.c_size_t * n)(*[b[0].numel() * b.element_size() for b in banks])
    for b in list(banks) + bs:
        hip.ptr(b)                      # device / contiguity check
    hip.call("nr_bank_ring_push", n, pb, pn, rb, cap, int(head_new), hip.ptr(head_dev, torch.int32, allow_none=True), n_new,
             hip.stream_ptr())


def diag_ranks(S):
    S = _f32(S).contiguous()
    N = S.shape[0]
    g = torch.empty((N,), dtype=torch.int32, device=S.device)
    e = torch.empty((N,), dtype=torch.int32, device=S.device)
    hip.call("nr_diag_ranks", hip.ptr(S), N, hip.ptr(g), hip.ptr(e), hip.stream_ptr())
    return g, e


def slab_ranks(S_slab, row0, diag):
    """(greater_rows, equal_rows [n], greater_cols, equal_cols [N]) of a row slab against the full diagonal (nr_slab_ranks)."""
    S_slab = _f32(S_slab).contiguous()
    n, N = S_slab.shape
    dev = S_slab.device
    gr = torch.empty((n,), dtype=torch.int32, device=dev)
    er = torch.empty((n,), dtype=torch.int32, device=dev)
    gc = torch.empty((N,), dtype=torch.int32, device=dev)
    ec = torch.empty((N,), dtype=torch.int32, device=dev)
    hip.call("nr_slab_ranks", hip.ptr(S_slab), n, N, int(row0), hip.ptr(diag, torch.float32), hip.ptr(gr), hip.ptr(er),
             hip.ptr(gc), hip.ptr(ec), hip.stream_ptr())
    return gr, er, gc, ec


def group_slab_ranks(S_slab, row0, group_end):
    """(greater_rows, equal_before_rows [n] int32, group_max [G,V] fp32) of a sentence-row slab (nr_group_slab_ranks);
    group_end [G] int32 on the device, its last entry >= row0 + n (the caller built it: not re-read here)."""
    S_slab = _f32(S_slab).contiguous()
    n, V = S_slab.shape
    G = group_end.shape[0]
    dev = S_slab.device
    gr = torch.empty((n,), dtype=torch.int32, device=dev)
    eb = torch.empty((n,), dtype=torch.int32, device=dev)
    gmax = torch.empty((G, V), dtype=torch.float32, device=dev)
    hip.call("nr_group_slab_ranks", hip.ptr(S_slab), n, V, int(row0), hip.ptr(group_end, torch.int32), G, hip.ptr(gr),
             hip.ptr(eb), hip.ptr(gmax), hip.stream_ptr())
    return gr, eb, gmax


def linear_x3(x, w, bias=None, residual=None):
    """Y = X W^T (+ bias) (+ residual) on the split-bf16 MFMA tile engine (nr_linear_x3): x [M,K], w [N,K] fp32, K padded to
    a multiple of 64 with zeros.  ~fp32-grade products (3 bf16 passes); used for the clustering GEMMs and for the
    backward of the token-scorer MLP."""
    x, w = _f32(x).contiguous(), _f32(w).contiguous()
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError("inner dimensions differ")
    if K % 64:
        pad = 64 - K % 64
        x = torch.nn.functional.pad(x, (0, pad))
        w = torch.nn.functional.pad(w, (0, pad))
        K += pad
    xh, xl = split_bf16(x)
    wh, wl = split_bf16(w)
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    b = _f32(bias).contiguous() if bias is not None else None
    r = _f32(residual).contiguous() if residual is not None else None
    hip.call("nr_linear_x3", hip.ptr(xh), hip.ptr(xl), hip.ptr(wh), hip.ptr(wl), hip.ptr(b, allow_none=True), hip.ptr(r, allow_none=True),
             M, N, K, hip.ptr(out), hip.stream_ptr())
    return out
