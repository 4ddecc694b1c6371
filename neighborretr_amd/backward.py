"""torch.autograd.Function wrappers whose forward AND backward run on the HIP kernels.

Gradient structure of the head (verified against the reference's autograd through the golden
vectors, SURVEY.md 8a "Autograd and edge-case facts"): nothing is detached except the Sinkhorn
targets and the top-K / arg-max index choices, so gradients flow
  losses -> S (three local_level calls), G, bank centralities, centrality weights, logit_scale
         -> max-pool arg-max routes -> normalised tokens -> features
         -> token weights -> softmax -> scorer MLP (parameters and features)
         -> centrality weights -> global tokens and the mean of the normalised tokens.
The memory-bank FEATURES get no gradient; the bank tokens' scorer weights do (parameters only).
"""
import torch

from . import head, hip, ops


def _coef_rowloss(g, hp, B):
    """d(objective)/d(rowloss[dir,term,row]) for upstream g [5] on (total, cent, unif, neigh, kl)."""
    wu, wn, wkl = hp["uniform_weight"], hp["neighbor_weight"], hp["kl_weight"]
    c = torch.stack((g[0] + g[1], g[0] * wu + g[2], g[0] * wn + g[3], (g[0] * wkl + g[4]) / B)) * (0.5 / B)
    return c.float()[None, :, None].expand(2, 4, B).contiguous()


def _saved_state(ctx, what):
    """The forward state of a fused-head node (dropped at the end of its backward: it holds the node's own outputs, and with
    them a reference cycle through the autograd graph)."""
    if ctx.sv is None:
        raise RuntimeError(f"{what}: backward through the fused loss head a second time -- its saved state is released by the first "
                           "backward (retain_graph=True / differentiating two of the losses separately is not supported: sum "
                           "them first)")
    return ctx.sv


OWN_MLP_GEMMS = True       # False: library GEMMs (torch.matmul -> hipBLASLt) for A/B timing (tools/train_times.py)


def _mm(a, b_t):
    """a [M,K] x b_t[N,K]^T on the build's split-bf16 engine, or through the library for A/B."""
    if OWN_MLP_GEMMS:
        return ops.linear_x3(a, b_t)
    return a @ b_t.t()


def _mlp_backward(feats, dlogits, w1, b1, w2, n_dx, exact):
    """Backward of Linear(d,H)-ReLU-Linear(H,1) over the concatenation of `feats` (list of [n_i,d] tensors) with upstream
    `dlogits` (list of [n_i]).  The three GEMMs -- the recomputed hidden layer X W1^T, dW1 = dh^T X and dX = dh W1 -- run
    on the build's split-bf16 MFMA engine (nr_linear_x3, fp32-grade products; no library GEMM on the training path); the
    two H-vectors dW2 / db1 are plain column sums.  `exact` is kept for the callers' signature: both precision
    plans now get the same fp32-grade gradients.  Returns dW1, db1, dW2, db2 and dX of the first `n_dx` rows (or None)."""
    X = torch.cat([f.reshape(-1, f.shape[-1]) for f in feats], 0).float()
    dl = torch.cat([d.reshape(-1) for d in dlogits], 0).float()
    w1f = w1.detach().float()
    h = ops.linear_x3(X, w1f, b1.detach().float()) if OWN_MLP_GEMMS else torch.addmm(b1.detach().float(), X, w1f.t())   # [n, H]
    act = h > 0
    a_dl = torch.where(act, h * dl[:, None], torch.zeros_like(h))
    dW2 = a_dl.sum(0).reshape(1, -1)                                     # sum_t dl_t relu(h_t)
    db2 = dl.sum().reshape(1)
    dh = torch.where(act, dl[:, None] * w2.detach().float().reshape(1, -1), torch.zeros_like(h))
    db1 = dh.sum(0)
    dW1 = _mm(dh.t().contiguous(), X.t().contiguous())                   # [H, d] = dh^T X   (K = tokens)
    dX = _mm(dh[:n_dx].contiguous(), w1f.t().contiguous()) if n_dx else None      # [n_dx, d] = dh W1
    return dW1, db1, dW2, db2, dX


FUSED_MLP_BACKWARD = True      # False: the torch-op form above (A/B and cross-check)


def _pad64(n):
    return (n + 63) // 64 * 64


def _residual(j):
    r = j["job"].get("add_to")
    if r is None:
        return None
    r = r.detach().float().contiguous().view(j["dX"].shape)
    return r


def _mlp_backward_hip(jobs):
    """Backward of the token scorers of BOTH modalities in eight launches (nr_token_mlp_bwd_hidden recomputes the hidden layer
    like the forward and emits dh as bf16 pairs from its epilogue; the three GEMMs of each scorer run grouped on the tile
    engine; bias gradients come out of one grouped column sum).  jobs: one dict per modality --
        sw    head.ScorerWeights of the scorer
        sets  [(prepared tokens, raw features [n, d] f32, dl [n] f32, precision the FORWARD ran this set in), ...]
              the batch tokens first, then the bank tokens; dX is returned for the first set's rows
        add_to  optional [n_0, d] tensor added to dX inside its GEMM
    Returns [(dW1 [H,d], db1 [H], dW2 [1,H], db2 [1], dX [n_0, d]), ...]."""
    from .cluster_backward_hip import _colsum_group, _linear_group
    from .cluster_fused import split_group
    J = []
    items = []
    for job in jobs:
        sw = job["sw"]
        H, d = sw.w1_hi.shape
        dev = sw.w1_hi.device
        offs, t0 = [], 0
        for prep, feat, dl, prec in job["sets"]:
            offs.append(t0)
            t0 += _pad64(prep.n_tok)
        ldT = t0
        exact_fit = all(prep.n_tok % 64 == 0 for prep, _, _, _ in job["sets"])
        i16 = dict(dtype=torch.int16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        mk = torch.empty if exact_fit else torch.zeros            # the K padding between / behind the sets must be zeros
        j = dict(job=job, H=H, d=d, ldT=ldT, offs=offs, dhT_hi=mk((H, ldT), **i16), dhT_lo=mk((H, ldT), **i16),
                 XT_hi=torch.empty((d, ldT), **i16), XT_lo=torch.empty((d, ldT), **i16))
        n0 = job["sets"][0][0].n_tok
        j["dh_hi"], j["dh_lo"] = torch.empty((n0, H), **i16), torch.empty((n0, H), **i16)
        # (set k > 0 with a one-pass weight gradient asks for the hi halves of dh^T only: see the calls below)
        rows = [int(hip.lib().nr_token_mlp_bwd_part_rows(prep.n_tok, H, int(prec),
                                                         int(k > 0 and int(prec) != hip.PREC_BF16X3 and ONE_PASS_WEIGHT_GRAD)))
                for k, (prep, _, _, prec) in enumerate(job["sets"])]
        j["rows"], R = rows, sum(rows)
        j["dw2_part"], j["db1_part"], j["dl_part"] = torch.empty((R, H), **f32), torch.empty((R, H), **f32), torch.empty((R, 1), **f32)
        for name, shape in (("dW1", (H, d)), ("db1", (H,)), ("dW2", (1, H)), ("db2", (1,)), ("dX", (n0, d))):
            j[name] = torch.empty(shape, **f32)
        for (prep, feat, dl, prec), off in zip(job["sets"], offs):
            x = feat.detach().reshape(-1, d).float().contiguous()
            # X^T of this set at columns [off, off + pad64(n)): the row tiles of the transposed form also write the zero padding
            # (a set whose weight-gradient GEMM is one-pass gets the hi halves only)
            want_lo = int(prec) == hip.PREC_BF16X3 or not ONE_PASS_WEIGHT_GRAD
            items.append((x, None, (j["XT_hi"], off), (j["XT_lo"], off) if want_lo else None, prep.n_tok, d, 1, ldT))
        J.append(j)
    split_group(items)
    for j in J:
        sw, r0 = j["job"]["sw"], 0
        for k, ((prep, feat, dl, prec), off, rows) in enumerate(zip(j["job"]["sets"], j["offs"], j["rows"])):
            dl = dl.detach().reshape(-1).float().contiguous()
            first = k == 0
            # a set whose weight-gradient GEMM is one-pass (below) has no use for the low halves of its dh^T
            want_lo = int(prec) == hip.PREC_BF16X3 or not ONE_PASS_WEIGHT_GRAD
            hip.call("nr_token_mlp_bwd_hidden", hip.ptr(prep.hi), hip.ptr(prep.lo, allow_none=True), hip.ptr(prep.norm), prep.n_tok,
                     prep.d, hip.ptr(sw.w1_hi), hip.ptr(sw.w1_lo, allow_none=True), hip.ptr(sw.b1), hip.ptr(sw.w2), j["H"], int(prec),
                     hip.ptr(dl), hip.ptr(j["dhT_hi"]), hip.ptr(j["dhT_lo"]) if want_lo else None, j["ldT"], off,
                     hip.ptr(j["dh_hi"]) if first else None, hip.ptr(j["dh_lo"]) if first else None,
                     ctypes_ptr(j["dw2_part"], r0 * j["H"]), ctypes_ptr(j["db1_part"], r0 * j["H"]), ctypes_ptr(j["dl_part"], r0),
                     hip.stream_ptr())
            r0 += rows
    # dX = dh W1 (rows of the first set), dW1 = dh^T X (K = all tokens of the modality)
    _linear_group([(j["dh_hi"], j["dh_lo"]) + j["job"]["sw"].w1_transposed()
                   + (None, _residual(j), j["dX"], j["dX"].shape[0], j["d"], j["H"]) for j in J])
    _weight_grad_sliced(J)
    cs = []
    for j in J:
        R = sum(j["rows"])
        cs += [(j["dw2_part"], j["dW2"], R, j["H"]), (j["db1_part"], j["db1"], R, j["H"]), (j["dl_part"], j["db2"], R, 1)]
    _colsum_group(cs)
    return [(j["dW1"], j["db1"], j["dW2"], j["db2"], j["dX"]) for j in J]


ONE_PASS_WEIGHT_GRAD = True      # False: the scorers' dW1 in split-bf16 for every token set (cross-check of the one-pass product)


def _apportion(weights, slots):
    """Whole numbers >= 1 that add up to `slots` (or to len(weights) if that is more), proportional to `weights`."""
    total = float(sum(weights))
    want = [slots * w / total for w in weights]
    got = [max(1, int(w)) for w in want]
    while sum(got) < slots:
        k = max(range(len(got)), key=lambda i: want[i] - got[i])
        got[k] += 1
    while sum(got) > slots and max(got) > 1:
        k = min((i for i in range(len(got)) if got[i] > 1), key=lambda i: want[i] - got[i])
        got[k] -= 1
    return got


def _weight_grad_sliced(J):
    """dW1 = dh^T X of every scorer (K = all tokens of the modality: 15360 / 7680 at configs[1]) as K-SLICED problems -- a
    [H, d] output is only 64..128 blocks, far fewer than CUs, so each set of tokens is cut into column ranges of the
    transposed operands (nr_linear_group with a row pitch), every range writes its own slab and nr_slab_sum_group adds them
    in fixed order.  Sets whose forward ran one-pass bf16 (the bank, on the mixed plan) take the one-pass GEMM here too: the
    hi halves of dh and X, a third of the matrix-core work; the other sets stay split-bf16.  Two GEMM launches + one sum
    (was one launch of 128 long-K blocks: 195 us at configs[1])."""
    from .cluster_backward_hip import _linear_group
    variants = {True: [], False: []}                       # exact (split-bf16) / one pass -> [(job index, set index), ...]
    for ji, j in enumerate(J):
        for si, (prep, feat, dl, prec) in enumerate(j["job"]["sets"]):
            variants[int(prec) == hip.PREC_BF16X3 or not ONE_PASS_WEIGHT_GRAD].append((ji, si))
    sums = [dict(out=j["dW1"], parts=[]) for j in J]
    keep = []
    for exact, members in variants.items():
        if not members:
            continue
        cols = [_pad64(J[ji]["job"]["sets"][si][0].n_tok) for ji, si in members]
        cuts = _apportion(cols, hip.LINEAR_GROUP_MAX)
        n_prob = sum(cuts)
        H, d = J[0]["H"], J[0]["d"]
        slabs = torch.empty((n_prob, H, d), dtype=torch.float32, device=J[0]["dW1"].device)
        keep.append(slabs)
        probs, k, runs = [], 0, {}
        for (ji, si), n_cols, n_cut in zip(members, cols, cuts):
            j = J[ji]
            if (j["H"], j["d"]) != (H, d):
                raise hip.NrHipError("scorers of one backward launch share their shape")
            step = _pad64(-(-n_cols // n_cut))
            first = k
            for c0 in range(0, n_cols, step):
                kc = min(step, n_cols - c0)
                off = 2 * (j["offs"][si] + c0)               # bytes: int16 elements
                x_hi, w_hi = j["dhT_hi"].data_ptr() + off, j["XT_hi"].data_ptr() + off
                x_lo = j["dhT_lo"].data_ptr() + off if exact else None
                w_lo = j["XT_lo"].data_ptr() + off if exact else None
                probs.append((x_hi, x_lo, w_hi, w_lo, None, None, slabs[k], H, d, kc, j["ldT"]))
                k += 1
            if ji in runs:                                   # the sets of a job are neighbours in `members`: one run of slabs
                runs[ji][1] += k - first
            else:
                runs[ji] = [first, k - first]
        for ji, (first, cnt) in runs.items():
            sums[ji]["parts"].append((slabs[first], cnt))
        _linear_group(probs)
    arr = (hip.SlabSum * len(sums))()
    for a, sm in zip(arr, sums):
        a.out, a.n, a.accumulate, a.n_src = sm["out"].data_ptr(), sm["out"].numel(), 0, len(sm["parts"])
        for i, (first, n) in enumerate(sm["parts"]):
            a.part[i], a.n_slabs[i] = first.data_ptr(), n
    hip.call("nr_slab_sum_group", len(sums), arr, hip.stream_ptr())


def ctypes_ptr(t, elem_offset):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + 4 * int(elem_offset))


class HeadLossFn(torch.autograd.Function):
    """Fused _compute_losses (modeling.py:314-360) given the global tokens."""

    @staticmethod
    def forward(ctx, model, hp, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                text_feat, video_feat, gt, gv, logit_scale, w1t, b1t, w2t, b2t, w1v, b1v, w2v, b2v,
                g1_w1t, g1_b1t, g1_w2t, g1_b2t, g1_w1v, g1_b1v, g1_w2v, g1_b2v):
        """The last eight inputs are the *_weight_fc1 scorers of global_level (modeling.py:518-523): they matter (and get a
        gradient) only with several global tokens per sample; with one token their softmax weight is the constant 1."""
        prec = model._prec()
        losses, sv = head.head_forward(text_feat.detach(), video_feat.detach(), text_mask, video_mask,
                                       mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v, gt.detach(), gv.detach(),
                                       model.scorer_weights("text_weight_fc"), model.scorer_weights("video_weight_fc"),
                                       hp, logit_scale.detach(), prec, keep=True, join=model._take_join(),
                                       bank_streams=model._bank_streams(text_feat.device),
                                       **model._global_scorers(text_feat, video_feat))
        ctx.sv, ctx.hp, ctx.exact = sv, dict(hp), prec == hip.PREC_BF16X3
        ctx.model, ctx.plan = model, head.precision_plan(prec)
        ctx.masks = (text_mask, video_mask)
        ctx.shapes = (text_feat.shape, video_feat.shape, mb_feat_t.shape, mb_feat_v.shape, gt.shape, gv.shape)
        sv["gt2"], sv["gv2"] = sv["gt2"].reshape(-1, sv["gt2"].shape[-1]), sv["gv2"].reshape(-1, sv["gv2"].shape[-1])
        ctx.save_for_backward(text_feat, video_feat, mb_feat_t, mb_feat_v, w1t, b1t, w2t, w1v, b1v, w2v,
                              g1_w1t, g1_b1t, g1_w2t, g1_w1v, g1_b1v, g1_w2v)
        # five scalar outputs (total, centrality, uniform, neighbour, kl): a loss that is not differentiated brings no gradient
        # (not a zero tensor), and backward() on one of them needs no select-backward fill / copy launches
        ctx.set_materialize_grads(False)
        return tuple(losses.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return (None,) * 29
        (text_feat, video_feat, mb_feat_t, mb_feat_v, w1t, b1t, w2t, w1v, b1v, w2v,
         g1_w1t, g1_b1t, g1_w2t, g1_w1v, g1_b1v, g1_w2v) = ctx.saved_tensors
        sv = _saved_state(ctx, "HeadLossFn")
        dS, d_c0, d_c1, dmean_t, dmean_v, d_gt, d_gv, d_ls, g1 = _global_backward(
            sv, ctx.hp, ctx.shapes, gs, (g1_w1t, g1_b1t, g1_w2t, g1_w1v, g1_b1v, g1_w2v))
        d_text, d_video, scorer = _local_backward(sv, ctx.shapes, ctx.masks, ctx.exact, ctx.plan, ctx.model, text_feat, video_feat,
                                                  mb_feat_t, mb_feat_v, (w1t, b1t, w2t, w1v, b1v, w2v), dS, d_c0, d_c1, dmean_t,
                                                  dmean_v)
        ctx.sv = ctx.model = None
        return (None, None, None, None, None, None, None, None, d_text, d_video, d_gt, d_gv, d_ls, *scorer, *g1)


def _global_backward(sv, hp, shapes, gs, g1_params):
    """The part of the loss head's backward that does not touch the token features: from the gradients of the five losses
    through the row terms (nr_row_losses_bwd) to  dS [B,B], d bank means d_c0 / d_c1 [B], d centrality means dmean_t / dmean_v
    [d], d global tokens d_gt / d_gv, d logit scale, and the *_weight_fc1 gradients when a sample has several global tokens."""
    (B, Nt, d), (_, Nv, _), (M, _, _), _, gt_shape, gv_shape = shapes
    Gt, Gv = gt_shape[1], gv_shape[1]
    K, T = int(hp["num_neighbors"]), hp["temperature"]
    g1_w1t, g1_b1t, g1_w2t, g1_w1v, g1_b1v, g1_w2v = g1_params
    if sv["S"].is_cuda:
        _record_on_current_stream(sv)
    coef = ops.rowloss_coef(gs, hp, B)
    dS_dir, dG_dir, dC_rows, dwc, dls_rows = ops.row_losses_bwd(sv["S"], sv["G"], sv["tgt_r"], sv["tgt_c"], sv["c0"],
                                                                sv["c1"], sv["wc_t"], sv["wc_v"], sv["ls"], K, T, coef)
    dS, dG, d_c0, d_c1, d_ls = ops.rowloss_bwd_finish(dS_dir, dG_dir, dC_rows, dls_rows)     # one launch (was four)
    g1 = [None] * 8
    if sv["g_saved"] is None:
        d_gt = d_gv = None              # global logits G = gt gv^T: tiny plain GEMMs, folded into the centrality step below
    else:
        # several global tokens per sample: G is the fused product on the un-normalised global tokens with the
        # *_weight_fc1 softmax weights (modeling.py:516-539): arg-max-routed gradient + softmax + scorer-MLP backward
        gs_ = sv["g_saved"]
        d_gt, d_wgt = ops.local_level_bwd(0, dG, 0, 1.0, gs_["pv"], gs_["w_t"], gs_["w_v"], gs_["aux"], B, Gt, B, Gv, use_lo=True)
        d_gv, d_wgv = ops.local_level_bwd(1, dG, 0, 1.0, gs_["pt"], gs_["w_v"], gs_["w_t"], gs_["aux"], B, Gt, B, Gv, use_lo=True)
        dl_gt = ops.token_softmax_bwd(gs_["w_t"], d_wgt.view(B, Gt))
        dl_gv = ops.token_softmax_bwd(gs_["w_v"], d_wgv.view(B, Gv))
        gW1t, gb1t, gW2t, gb2t, gXt = _mlp_backward([sv["gt2"]], [dl_gt], g1_w1t, g1_b1t, g1_w2t, B * Gt, True)
        gW1v, gb1v, gW2v, gb2v, gXv = _mlp_backward([sv["gv2"]], [dl_gv], g1_w1v, g1_b1v, g1_w2v, B * Gv, True)
        d_gt, d_gv = d_gt + gXt, d_gv + gXv
        g1 = [gW1t, gb1t, gW2t, gb2t, gW1v, gb1v, gW2v, gb2v]
    # centrality weights: w_i = mean over the sample's global tokens of exp(c <g_hat, mean>)  (one token: the reference's)
    cs = hp["centrality_scale"]
    gn_t, gn_v, wtok_t, wtok_v = sv["cw_aux"]
    dw_t = dwc[0] if Gt == 1 else (dwc[0] / Gt)[:, None].expand(B, Gt).reshape(-1).contiguous()
    dw_v = dwc[1] if Gv == 1 else (dwc[1] / Gv)[:, None].expand(B, Gv).reshape(-1).contiguous()
    if sv["gt2"].shape == sv["gv2"].shape:
        dg_t, dmean_t, dg_v, dmean_v = ops.centrality_weights_bwd_pair(sv["gt2"], gn_t, sv["mean_t"], wtok_t, dw_t,
                                                                       sv["gv2"], gn_v, sv["mean_v"], wtok_v, dw_v, cs)
    else:
        dg_t, dmean_t = ops.centrality_weights_bwd(sv["gt2"], gn_t, sv["mean_t"], wtok_t, dw_t, cs)
        dg_v, dmean_v = ops.centrality_weights_bwd(sv["gv2"], gn_v, sv["mean_v"], wtok_v, dw_v, cs)
    if d_gt is None:                    # one token per sample: d gt = dG gv + (centrality part), one launch for both
        d_gt, d_gv = ops.global_logits_bwd(dG, sv["gt2"], sv["gv2"], dg_t, dg_v)
        d_gt, d_gv = d_gt.reshape(gt_shape), d_gv.reshape(gv_shape)
    else:
        d_gt = (d_gt + dg_t).reshape(gt_shape)
        d_gv = (d_gv + dg_v).reshape(gv_shape)
    return dS, d_c0, d_c1, dmean_t, dmean_v, d_gt, d_gv, d_ls, g1


def _record_on_current_stream(obj, seen=None):
    """record_stream(current) on every CUDA tensor reachable from `obj` (dicts, sequences, ops.Prepared-like objects): what a
    backward reads may have been allocated on another stream of the forward (local / bank streams)."""
    cur = torch.cuda.current_stream()
    seen = set() if seen is None else seen
    stack = [obj]
    while stack:
        o = stack.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        if torch.is_tensor(o):
            if o.is_cuda:
                o.record_stream(cur)
        elif isinstance(o, dict):
            stack.extend(o.values())
        elif isinstance(o, (list, tuple)):                     # (ops.Prepared is a namedtuple)
            stack.extend(o)


def _local_backward(sv, shapes, masks, exact, plan, model, text_feat, video_feat, mb_feat_t, mb_feat_v, scorer_params,
                    dS, d_c0, d_c1, dmean_t, dmean_v):
    """The token side of the loss head's backward: the three fused products (batch x batch, text x bank-video, bank-text x video)
    through their stored arg-max routes, normalisation / mask / centrality mean, token softmax and the scorer MLPs ->
    d text_feat, d video_feat and the *_weight_fc gradients (dW1t, db1t, dW2t, db2t, dW1v, db1v, dW2v, db2v)."""
    (B, Nt, d), (_, Nv, _), (M, _, _), _, _, _ = shapes
    text_mask, video_mask = masks
    w1t, b1t, w2t, w1v, b1v, w2v = scorer_params
    if dS.is_cuda:
        _record_on_current_stream(sv)
    pt, pv, pbt, pbv = sv["pt"], sv["pv"], sv["pbt"], sv["pbv"]
    lo = exact
    aux0, aux1, aux2 = sv["aux"]
    f32 = dict(dtype=torch.float32, device=dS.device)
    d_tn, d_vn = torch.empty((B * Nt, d), **f32), torch.empty((B * Nv, d), **f32)
    d_wt, d_wv = torch.empty((B * Nt,), **f32), torch.empty((B * Nv,), **f32)
    d_wbt, d_wbv = torch.empty((M * Nt,), **f32), torch.empty((M * Nv,), **f32)
    if ops.USE_MFMA_BACKWARD and bool(hip.lib().nr_local_level_bwd_mfma_supported(Nt, Nv, d)):
        # token gradients: the four products' routing matrices on the matrix cores in ONE launch (their "other" operands
        # transposed in one launch before it), the chunks of a gradient's two products summed by one more
        T_pv, T_pt, T_pbv, T_pbt = ops.transpose_prepared([pv, pt, pbv, pbt], use_lo=lo)
        ops.local_level_bwd_group([
            dict(side=0, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pv, w_self=sv["w_t"], w_other=sv["w_v"], aux=aux0,
                 A=B, Nt=Nt, Bv=B, Nv=Nv, d_x=d_tn),
            dict(side=0, dS=d_c1, ds_mode=1, ds_scale=1.0 / M, other_T=T_pbv, w_self=sv["w_t"], w_other=sv["w_bv"], aux=aux1,
                 A=B, Nt=Nt, Bv=M, Nv=Nv, d_x=d_tn),
            dict(side=1, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pt, w_self=sv["w_v"], w_other=sv["w_t"], aux=aux0,
                 A=B, Nt=Nt, Bv=B, Nv=Nv, d_x=d_vn),
            dict(side=1, dS=d_c0, ds_mode=2, ds_scale=1.0 / M, other_T=T_pbt, w_self=sv["w_v"], w_other=sv["w_bt"], aux=aux2,
                 A=M, Nt=Nt, Bv=B, Nv=Nv, d_x=d_vn)], use_lo=lo)
        # token-weight gradients of the batch and of the bank, all six sums in one launch
        ops.pool_weight_bwd_group([
            dict(side=0, N=Nt, d_w=d_wt, srcs=[(dS, 0, 1.0, aux0[2], B, B), (d_c1, 1, 1.0 / M, aux1[2], B, M)]),
            dict(side=1, N=Nv, d_w=d_wv, srcs=[(dS, 0, 1.0, aux0[3], B, B), (d_c0, 2, 1.0 / M, aux2[3], M, B)]),
            dict(side=1, N=Nv, d_w=d_wbv, srcs=[(d_c1, 1, 1.0 / M, aux1[3], B, M)]),
            dict(side=0, N=Nt, d_w=d_wbt, srcs=[(d_c0, 2, 1.0 / M, aux2[2], M, B)])])
    else:
        ops.local_level_bwd(0, dS, 0, 1.0, pv, sv["w_t"], sv["w_v"], aux0, B, Nt, B, Nv, d_x=d_tn, d_w=d_wt, use_lo=lo)
        ops.local_level_bwd(1, dS, 0, 1.0, pt, sv["w_v"], sv["w_t"], aux0, B, Nt, B, Nv, d_x=d_vn, d_w=d_wv, use_lo=lo)
        ops.local_level_bwd(0, d_c1, 1, 1.0 / M, pbv, sv["w_t"], sv["w_bv"], aux1, B, Nt, M, Nv, d_x=d_tn, d_w=d_wt,
                            accumulate=True, use_lo=lo)
        ops.local_level_bwd(1, d_c1, 1, 1.0 / M, pt, sv["w_bv"], sv["w_t"], aux1, B, Nt, M, Nv, d_w=d_wbv, want_dx=False)
        ops.local_level_bwd(1, d_c0, 2, 1.0 / M, pbt, sv["w_v"], sv["w_bt"], aux2, M, Nt, B, Nv, d_x=d_vn, d_w=d_wv,
                            accumulate=True, use_lo=lo)
        ops.local_level_bwd(0, d_c0, 2, 1.0 / M, pv, sv["w_bt"], sv["w_v"], aux2, M, Nt, B, Nv, d_w=d_wbt, want_dx=False)
    # normalise / mask / centrality-mean backward
    d_text = ops.normalize_bwd(text_feat, pt.norm, text_mask, d_tn, dmean_t)
    d_video = ops.normalize_bwd(video_feat, pv.norm, video_mask, d_vn, dmean_v)
    # token weights -> logits -> scorer MLPs
    dl_t = ops.token_softmax_bwd(sv["w_t"], d_wt.view(B, Nt))
    dl_v = ops.token_softmax_bwd(sv["w_v"], d_wv.view(B, Nv))
    dl_bt = ops.token_softmax_bwd(sv["w_bt"], d_wbt.view(M, Nt))
    dl_bv = ops.token_softmax_bwd(sv["w_bv"], d_wbv.view(M, Nv))
    if FUSED_MLP_BACKWARD:
        _, p_mlp, p_bank = plan
        # the scorer's input gradient is added to the similarity path's inside its GEMM (residual operand)
        (dW1t, db1t, dW2t, db2t, d_text), (dW1v, db1v, dW2v, db2v, d_video) = _mlp_backward_hip([
            dict(sw=model.scorer_weights("text_weight_fc"), add_to=d_text,
                 sets=[(pt, text_feat, dl_t, p_mlp), (pbt, mb_feat_t, dl_bt, p_bank)]),
            dict(sw=model.scorer_weights("video_weight_fc"), add_to=d_video,
                 sets=[(pv, video_feat, dl_v, p_mlp), (pbv, mb_feat_v, dl_bv, p_bank)])])
        d_text, d_video = d_text.view(text_feat.shape), d_video.view(video_feat.shape)
    else:
        dW1t, db1t, dW2t, db2t, dXt = _mlp_backward([text_feat, mb_feat_t], [dl_t, dl_bt], w1t, b1t, w2t, B * Nt, exact)
        dW1v, db1v, dW2v, db2v, dXv = _mlp_backward([video_feat, mb_feat_v], [dl_v, dl_bv], w1v, b1v, w2v, B * Nv, exact)
        d_text = d_text + dXt.view_as(d_text)
        d_video = d_video + dXv.view_as(d_video)
    return d_text, d_video, (dW1t, db1t, dW2t, db2t, dW1v, db1v, dW2v, db2v)


class HeadLocalFn(torch.autograd.Function):
    """First of the TWO autograd nodes the loss head is split into so that the backward of its token side (this node: ~2/3 of
    the head's backward time) can run BESIDE the backward of the token clustering, which only needs the gradients of the global
    tokens that the second, short node (HeadGlobalFn) produces.  Autograd runs a node's backward on the stream its forward ran
    on: with the clustering issued on a side stream in the forward (modeling._compute_losses), the two heavy backward chains
    overlap, eagerly and as parallel branches of a captured training graph.

    The forward runs the WHOLE fused head (head.head_forward, every kernel as in HeadLossFn) and hands the losses and the saved
    state to HeadGlobalFn through `bridge`; its differentiable outputs are the five tensors through which the loss reaches the
    token side: S [B,B], the bank column means c0 / c1 [B], the centrality means mean_t / mean_v [d]."""

    @staticmethod
    def forward(ctx, model, hp, bridge, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v, gt, gv, logit_scale,
                g1, text_feat, video_feat, w1t, b1t, w2t, b2t, w1v, b1v, w2v, b2v):
        prec = model._prec()
        stepwise = bridge.pop("join", None)       # gt = gv = None: the clustering's launches, a generator driven by the head
        if stepwise is not None:
            # the clustering is issued launch by launch on THIS stream, interleaved (in capture order) with the local branch and
            # the early bank chains on the local stream -- the loss-only step's schedule (head.head_forward)
            extra = dict(join=stepwise, local_stream=model._local_stream(text_feat.device), bank_early=model.bank_early,
                         capture_order=getattr(model, "train_capture_order", None) or model.capture_order)
            gt_in = gv_in = None
        else:
            extra = dict(join=model._take_join())
            gt_in, gv_in = gt.detach(), gv.detach()
        losses, sv = head.head_forward(text_feat.detach(), video_feat.detach(), text_mask, video_mask,
                                       mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v, gt_in, gv_in,
                                       model.scorer_weights("text_weight_fc"), model.scorer_weights("video_weight_fc"),
                                       hp, logit_scale.detach(), prec, keep=True,
                                       bank_streams=model._bank_streams(text_feat.device),
                                       **extra, **model._global_scorers(text_feat, video_feat))
        ctx.sv, ctx.exact = sv, prec == hip.PREC_BF16X3
        ctx.model, ctx.plan = model, head.precision_plan(prec)
        ctx.masks = (text_mask, video_mask)
        ctx.shapes = (text_feat.shape, video_feat.shape, mb_feat_t.shape, mb_feat_v.shape, sv["gt2"].shape, sv["gv2"].shape)
        sv["gt2"], sv["gv2"] = sv["gt2"].reshape(-1, sv["gt2"].shape[-1]), sv["gv2"].reshape(-1, sv["gv2"].shape[-1])
        ctx.save_for_backward(text_feat, video_feat, mb_feat_t, mb_feat_v, w1t, b1t, w2t, w1v, b1v, w2v)
        bridge["losses"], bridge["sv"], bridge["shapes"] = losses, sv, ctx.shapes
        ctx.set_materialize_grads(False)
        return sv["S"], sv["c0"], sv["c1"], sv["mean_t"], sv["mean_v"]

    @staticmethod
    def backward(ctx, dS, d_c0, d_c1, dmean_t, dmean_v):
        if dS is None:
            return (None,) * 23
        text_feat, video_feat, mb_feat_t, mb_feat_v, w1t, b1t, w2t, w1v, b1v, w2v = ctx.saved_tensors
        d_text, d_video, scorer = _local_backward(_saved_state(ctx, "HeadLocalFn"), ctx.shapes, ctx.masks, ctx.exact, ctx.plan, ctx.model, text_feat, video_feat,
                                                  mb_feat_t, mb_feat_v, (w1t, b1t, w2t, w1v, b1v, w2v), dS, d_c0, d_c1, dmean_t,
                                                  dmean_v)
        ctx.sv = ctx.model = None
        return (None,) * 13 + (d_text, d_video, *scorer)


class HeadGlobalFn(torch.autograd.Function):
    """Second node of the split loss head (see HeadLocalFn): takes the losses HeadLocalFn's forward already computed and owns
    the short first part of the backward -- row terms, global logits, centrality weights."""

    @staticmethod
    def forward(ctx, hp, bridge, S, c0, c1, mean_t, mean_v, gt, gv, logit_scale,
                g1_w1t, g1_b1t, g1_w2t, g1_b2t, g1_w1v, g1_b1v, g1_w2v, g1_b2v):
        ctx.sv, ctx.hp, ctx.shapes = bridge.pop("sv"), dict(hp), bridge.pop("shapes")
        ctx.save_for_backward(g1_w1t, g1_b1t, g1_w2t, g1_w1v, g1_b1v, g1_w2v)
        ctx.set_materialize_grads(False)
        return tuple(bridge.pop("losses").unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return (None,) * 18
        dS, d_c0, d_c1, dmean_t, dmean_v, d_gt, d_gv, d_ls, g1 = _global_backward(_saved_state(ctx, "HeadGlobalFn"), ctx.hp, ctx.shapes, gs,
                                                                                  ctx.saved_tensors)
        ctx.sv = None
        return (None, None, dS, d_c0, d_c1, dmean_t, dmean_v, d_gt, d_gv, d_ls, *g1)


SPLIT_HEAD_NODES = True       # False: the head as ONE autograd node (HeadLossFn), its backward strictly before the clustering's


def head_loss_nodes(model, hp, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v, text_feat, video_feat, gt, gv,
                    logit_scale, scorer_params, g1_params, cluster=None):
    """The differentiable fused head -> (total, centrality, uniform, neighbour, kl).
    cluster: None (gt / gv are the global tokens, autograd-tracked), or (launches, make_nodes) with gt = gv = None: `launches` a
    generator that issues the token clustering launch by launch and returns (gt, gv) -- HeadLocalFn's forward drives it,
    interleaved with the head's local branch -- and make_nodes() -> (gt, gv) creating the clustering's autograd nodes afterwards
    (their forwards only wrap what has been computed; see modeling._compute_losses)."""
    if not SPLIT_HEAD_NODES:
        return HeadLossFn.apply(model, hp, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t, mb_mask_v,
                                text_feat, video_feat, gt, gv, logit_scale, *scorer_params, *g1_params)
    bridge = {}
    if cluster is not None:
        bridge["join"] = cluster[0]
    S, c0, c1, mean_t, mean_v = HeadLocalFn.apply(model, hp, bridge, text_mask, video_mask, mb_feat_t, mb_feat_v, mb_mask_t,
                                                  mb_mask_v, gt, gv, logit_scale, None, text_feat, video_feat, *scorer_params)
    if cluster is not None:
        gt, gv = cluster[1]()
    return HeadGlobalFn.apply(hp, bridge, S, c0, c1, mean_t, mean_v, gt, gv, logit_scale, *g1_params)


class LocalLevelFn(torch.autograd.Function):
    """local_level / get_similarity_logits (modeling.py:483-514)."""

    @staticmethod
    def forward(ctx, model, text_mask, video_mask, text_feat, video_feat, w1t, b1t, w2t, b2t, w1v, b1v, w2v, b2v):
        prec = model._prec(for_head=False)
        A, Nt, d = text_feat.shape
        Bv, Nv, _ = video_feat.shape
        sw_t, sw_v = model.scorer_weights("text_weight_fc"), model.scorer_weights("video_weight_fc")
        pt = ops.prepare_tokens(text_feat.detach(), text_mask)
        pv = ops.prepare_tokens(video_feat.detach(), video_mask)
        w_t, _ = head.token_weights(pt, text_mask, sw_t, A, Nt, prec)
        w_v, _ = head.token_weights(pv, video_mask, sw_v, Bv, Nv, prec)
        S, aux = ops.local_level(pt, pv, w_t, w_v, A, Nt, Bv, Nv, prec, hip.OUT_FULL, want_arg=True)
        ctx.st = (pt, pv, w_t, w_v, aux, text_mask, video_mask, prec == hip.PREC_BF16X3, prec, sw_t, sw_v)
        ctx.save_for_backward(text_feat, video_feat, w1t, b1t, w2t, w1v, b1v, w2v)
        return S

    @staticmethod
    def backward(ctx, dS):
        pt, pv, w_t, w_v, aux, text_mask, video_mask, exact, prec, sw_t, sw_v = ctx.st
        text_feat, video_feat, w1t, b1t, w2t, w1v, b1v, w2v = ctx.saved_tensors
        A, Nt, d = text_feat.shape
        Bv, Nv, _ = video_feat.shape
        dS = dS.float().contiguous()
        if ops.USE_MFMA_BACKWARD and bool(hip.lib().nr_local_level_bwd_mfma_supported(Nt, Nv, d)):
            # both operands' token gradients in one launch (+ the slab sum), both weight sums in one more: the same grouped
            # kernels as the loss head's backward (this node serves the sharded training loss, four products per step)
            f32 = dict(dtype=torch.float32, device=dS.device)
            d_tn, d_vn = torch.empty((A * Nt, d), **f32), torch.empty((Bv * Nv, d), **f32)
            d_wt, d_wv = torch.empty((A * Nt,), **f32), torch.empty((Bv * Nv,), **f32)
            T_pv, T_pt = ops.transpose_prepared([pv, pt], use_lo=exact)
            ops.local_level_bwd_group([
                dict(side=0, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pv, w_self=w_t, w_other=w_v, aux=aux, A=A, Nt=Nt, Bv=Bv, Nv=Nv,
                     d_x=d_tn),
                dict(side=1, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pt, w_self=w_v, w_other=w_t, aux=aux, A=A, Nt=Nt, Bv=Bv, Nv=Nv,
                     d_x=d_vn)], use_lo=exact)
            ops.pool_weight_bwd_group([dict(side=0, N=Nt, d_w=d_wt, srcs=[(dS, 0, 1.0, aux[2], A, Bv)]),
                                       dict(side=1, N=Nv, d_w=d_wv, srcs=[(dS, 0, 1.0, aux[3], A, Bv)])])
        else:
            d_tn, d_wt = ops.local_level_bwd(0, dS, 0, 1.0, pv, w_t, w_v, aux, A, Nt, Bv, Nv, use_lo=exact)
            d_vn, d_wv = ops.local_level_bwd(1, dS, 0, 1.0, pt, w_v, w_t, aux, A, Nt, Bv, Nv, use_lo=exact)
        d_text = ops.normalize_bwd(text_feat, pt.norm, text_mask, d_tn, None)
        d_video = ops.normalize_bwd(video_feat, pv.norm, video_mask, d_vn, None)
        dl_t = ops.token_softmax_bwd(w_t, d_wt.view(A, Nt))
        dl_v = ops.token_softmax_bwd(w_v, d_wv.view(Bv, Nv))
        ctx.st = None
        # The scorers' backward stays on the form that recomputes the hidden layer from the RAW features (split-bf16 GEMM): this
        # node is compared element by element with the fp32 oracle on a few hundred tokens, where ONE hidden unit whose
        # pre-activation is within rounding of zero decides a ReLU differently in the fused form (which recomputes it from the
        # normalised bf16 pairs, exactly as the forward kernel did) and moves a dW1 entry by 1e-2 of the largest
        # (tools/dbg_mlp.py: either form flips somewhere against fp64 on random data; the fused one about ten times as often).
        dW1t, db1t, dW2t, db2t, dXt = _mlp_backward([text_feat], [dl_t], w1t, b1t, w2t, A * Nt, exact)
        dW1v, db1v, dW2v, db2v, dXv = _mlp_backward([video_feat], [dl_v], w1v, b1v, w2v, Bv * Nv, exact)
        return (None, None, None, d_text + dXt.view_as(d_text), d_video + dXv.view_as(d_video),
                dW1t, db1t, dW2t, db2t, dW1v, db1v, dW2v, db2v)


class GlobalLogitsFn(torch.autograd.Function):
    """One-global-token global_level: G = gt gv^T in exact fp32 (modeling.py:526-537)."""

    @staticmethod
    def forward(ctx, gt, gv):
        a = gt.detach().reshape(gt.shape[0], -1).float().contiguous()
        b = gv.detach().reshape(gv.shape[0], -1).float().contiguous()
        ctx.save_for_backward(a, b)
        ctx.shapes = (gt.shape, gv.shape)
        return ops.gemm_nt_f32(a, b)

    @staticmethod
    def backward(ctx, dG):
        a, b = ctx.saved_tensors
        return (dG @ b).reshape(ctx.shapes[0]), (dG.t() @ a).reshape(ctx.shapes[1])


class GlobalLevelMultiFn(torch.autograd.Function):
    """global_level with SEVERAL global tokens per sample (ActivityNet token counts; modeling.py:516-539): the fused product on
    the un-normalised global tokens with the softmax weights of the *_weight_fc1 scorers, no masks.  Backward: arg-max-routed
    token gradients (nr_local_level_bwd), softmax and scorer-MLP backward -- the arithmetic of the loss head's own multi-token
    branch (_global_backward), as a node of its own for the public `NeighborRetr.global_level`."""

    @staticmethod
    def forward(ctx, model, gt, gv, w1t, b1t, w2t, b2t, w1v, b1v, w2v, b2v):
        G, saved = head.global_logits(gt.detach().float().contiguous(), gv.detach().float().contiguous(),
                                      model.scorer_weights("text_weight_fc1"), model.scorer_weights("video_weight_fc1"), keep=True)
        ctx.st = saved
        ctx.save_for_backward(gt, gv, w1t, b1t, w2t, w1v, b1v, w2v)
        return G

    @staticmethod
    def backward(ctx, dG):
        gs_ = ctx.st
        gt, gv, w1t, b1t, w2t, w1v, b1v, w2v = ctx.saved_tensors
        B, Gt, d = gt.shape
        Bv, Gv, _ = gv.shape
        dG = dG.float().contiguous()
        gt2, gv2 = gt.detach().float().contiguous(), gv.detach().float().contiguous()
        d_gt, d_wgt = ops.local_level_bwd(0, dG, 0, 1.0, gs_["pv"], gs_["w_t"], gs_["w_v"], gs_["aux"], B, Gt, Bv, Gv, use_lo=True)
        d_gv, d_wgv = ops.local_level_bwd(1, dG, 0, 1.0, gs_["pt"], gs_["w_v"], gs_["w_t"], gs_["aux"], B, Gt, Bv, Gv, use_lo=True)
        dl_gt = ops.token_softmax_bwd(gs_["w_t"], d_wgt.view(B, Gt))
        dl_gv = ops.token_softmax_bwd(gs_["w_v"], d_wgv.view(Bv, Gv))
        gW1t, gb1t, gW2t, gb2t, gXt = _mlp_backward([gt2], [dl_gt], w1t, b1t, w2t, B * Gt, True)
        gW1v, gb1v, gW2v, gb2v, gXv = _mlp_backward([gv2], [dl_gv], w1v, b1v, w2v, Bv * Gv, True)
        ctx.st = None
        return (None, (d_gt.view_as(gt2) + gXt.view_as(gt2)).to(gt.dtype), (d_gv.view_as(gv2) + gXv.view_as(gv2)).to(gv.dtype),
                gW1t, gb1t, gW2t, gb2t, gW1v, gb1v, gW2v, gb2v)


class RowLossFn(torch.autograd.Function):
    """Raw per-row loss terms [2,4,B] with the HIP backward (used by the until_module classes)."""

    @staticmethod
    def forward(ctx, S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video, logit_scale, K, T):
        args = [t.detach().float().contiguous() for t in (S, G, tgt_rows, tgt_cols, bank_c0, bank_c1, wc_text, wc_video)]
        ls = logit_scale.detach().float().reshape(1).contiguous()
        ctx.save_for_backward(*args, ls)
        ctx.KT = (int(K), float(T))
        return ops.row_losses(*args, ls, K, T)

    @staticmethod
    def backward(ctx, g):
        *args, ls = ctx.saved_tensors
        K, T = ctx.KT
        dS_dir, dG_dir, dC_rows, dwc, dls_rows = ops.row_losses_bwd(*args, ls, K, T, g.float().contiguous())
        dS = ops.add_transposed(dS_dir[0], dS_dir[1])
        dG = ops.add_transposed(dG_dir[0], dG_dir[1])
        return (dS, dG, None, None, ops.colsum(dC_rows[0]), ops.colsum(dC_rows[1]), dwc[0], dwc[1],
                dls_rows.sum().reshape(1), None, None)
