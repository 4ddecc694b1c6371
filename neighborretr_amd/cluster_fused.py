"""No-grad forward of the token-clustering stage on fused HIP kernels + library GEMMs.

Same arithmetic as neighborretr_amd/cluster.py (and the reference's cluster.py:670-965); used when no
gradient is required (loss-only forward, evaluation), where the ~45 ATen launches of one
CTM + TCBlock stage collapse to 8:
    nr_shift_concat -> GEMM(+residual) -> nr_ctm_front -> nr_ctm_back
    -> GEMM q, GEMM kv -> nr_tc_attention -> GEMM proj (+residual+bias)
The training path keeps the autograd-traced torch ops of cluster.py.
"""
import math

import torch

from . import hip
from .capture_guard import may_fork, wait_stream


def _p(t, dtype=torch.float32):
    return hip.ptr(t, dtype)


def split_group(items):
    """ONE launch of nr_split_group.  items: (src, src2 or None, hi, lo or None, rows, cols, mode, ld[, group]) with tensors
    (group: tokens per sample, mode 3 only)."""
    for lo in range(0, len(items), hip.SPLIT_MAX):
        chunk = items[lo:lo + hip.SPLIT_MAX]
        arr = (hip.SplitItem * len(chunk))()
        for a, item in zip(arr, chunk):
            src, src2, hi, lo_, rows, cols, mode, ld = item[:8]
            a.group = int(item[8]) if len(item) > 8 else 0
            off = 0
            if isinstance(hi, tuple):                    # (tensor, element offset): write into a wider buffer of pitch ld
                (hi, off), lo_ = hi, (lo_[0] if lo_ is not None else None)
            for t in (src, src2, hi, lo_):
                if t is not None and (not t.is_cuda or not t.is_contiguous()):
                    raise hip.NrHipError("nr_split_group: contiguous GPU tensors only")
            a.src, a.src2 = _addr(src), _addr(src2)
            a.hi = hi.data_ptr() + 2 * off
            a.lo = None if lo_ is None else lo_.data_ptr() + 2 * off
            a.rows, a.cols, a.mode, a.ld = int(rows), int(cols), int(mode), int(ld)
        hip.call("nr_split_group", len(chunk), arr, hip.stream_ptr())


class _StageWeights:
    """Per-stage derived weights, rebuilt when any parameter of the stage changed: the bf16 hi/lo pairs the split-bf16 GEMMs
    read -- forward: the conv kernel as a [C, 3C] matrix, kv / q / proj weights [N, K]; backward (input-gradient GEMMs): the
    TRANSPOSED linears and the transposed convolution [C_in, 3 C_out] -- all written by ONE nr_split_group launch
    (`items`, issued by build_stage_weights for every stale stage of the step together)."""

    def __init__(self, ctm, blk):
        w = ctm.conv.conv.weight.detach().contiguous()                           # [C_out, C_in, 3]
        Co, Ci = w.shape[0], w.shape[1]
        dev = w.device
        attn = blk.attn
        wkv, wq, wp = attn.kv.weight.detach(), attn.q.weight.detach(), attn.proj.weight.detach()
        self._src = (w, wq, wkv, wp)
        self._f32 = {}
        self.items = []

        def pair(name, src, rows, cols, mode, group=0):
            shape = (cols, rows) if mode == 1 else (rows, cols)
            hi = torch.empty(shape, dtype=torch.int16, device=dev)
            lo = torch.empty(shape, dtype=torch.int16, device=dev)
            setattr(self, name + "_hi", hi)
            setattr(self, name + "_lo", lo)
            self.items.append((src.contiguous(), None, hi, lo, rows, cols, mode, rows if mode == 1 else cols, group))
        # both matrix forms of the convolution kernel are read straight from the parameter by the split launch (no permuted
        # fp32 copies: the six torch copies per stage and modality were ~200 us of small launches at the head of every step)
        pair("wconv", w, Co, 3 * Ci, 4, Ci)                                      # [C_out, 3 C_in]: tap k multiplies x[n+k-1]
        pair("wkv", wkv, wkv.shape[0], wkv.shape[1], 0)
        pair("wq", wq, wq.shape[0], wq.shape[1], 0)
        pair("wp", wp, wp.shape[0], wp.shape[1], 0)
        pair("wconv_bt", w, Ci, 3 * Co, 6, Co)                                   # [C_in, 3 C_out], taps reversed: the transposed
        #                                                                          convolution d x0 = d y + conv^T(d y), in place
        pair("wkv_bt", wkv, wkv.shape[0], wkv.shape[1], 1)                       # [C, 2C]: d kvn = d kv Wkv
        pair("wq_bt", wq, wq.shape[0], wq.shape[1], 1)
        pair("wp_bt", wp, wp.shape[0], wp.shape[1], 1)

    def _lazy(self, name, make):
        if name not in self._f32:
            self._f32[name] = make()
        return self._f32[name]

    # fp32 forms for the library-GEMM fallback of the one-stage forward (ctm_stage_fused, small stages); built on first use
    wcat = property(lambda self: self._lazy("wcat", lambda: self._src[0].permute(2, 1, 0).reshape(-1, self._src[0].shape[0]).contiguous()))
    wq_t = property(lambda self: self._lazy("wq_t", lambda: self._src[1].t().contiguous()))
    wkv_t = property(lambda self: self._lazy("wkv_t", lambda: self._src[2].t().contiguous()))
    wp_t = property(lambda self: self._lazy("wp_t", lambda: self._src[3].t().contiguous()))


def stage_params(ctm, blk):
    """The parameters of one stage (CTM then TCBlock, registration order), listed once per module pair: walking the module tree
    with .parameters() three times per stage and step was 0.7 ms of host time per training step."""
    hit = getattr(blk, "_nr_stage_params", None)
    if hit is None or hit[0] is not ctm:
        hit = (ctm, tuple(list(ctm.parameters()) + list(blk.parameters())))
        blk._nr_stage_params = hit
    return hit[1]


def build_stage_weights(cache, stages):
    """Refreshes the derived weights of several stages -- [(key, ctm, blk), ...] -- with ONE split launch for all that are
    stale (in training every stage is, after every optimizer step)."""
    fresh = []
    for key, ctm, blk in stages:
        params = stage_params(ctm, blk)
        ver = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params)
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            sw = _StageWeights(ctm, blk)
            cache[key] = (ver, sw)
            fresh.append(sw)
    items = [it for sw in fresh for it in sw.items]
    if items:
        split_group(items)
    for sw in fresh:
        sw.items = None


def _stage_weights(cache, key, ctm, blk):
    build_stage_weights(cache, [(key, ctm, blk)])
    return cache[key][1]


def ctm_stage_fused(x, mask, ctm, blk, noise, cache, key):
    """One CTM + TCBlock stage, forward only.  x [B,N,C] f32, mask [B,N] fp32 or None -> [B,cnum,C]."""
    x = x.detach().float().contiguous()
    B, N, C = x.shape
    dev = x.device
    sw = _stage_weights(cache, key, ctm, blk)
    st = hip.stream_ptr()
    f32 = dict(dtype=torch.float32, device=dev)
    # token convolution k=3 as one GEMM with the residual folded in: y = x + cat @ Wcat.
    # Big operand (stage 0): split-bf16 on the MFMA tile engine; small (stage 1): library GEMM.
    big = B * N >= 1024 and C % 64 == 0
    if big:
        cat_hi = torch.empty((B * N, 3 * C), dtype=torch.int16, device=dev)
        cat_lo = torch.empty((B * N, 3 * C), dtype=torch.int16, device=dev)
        hip.call("nr_shift_concat_split", _p(x), B, N, C, hip.ptr(cat_hi), hip.ptr(cat_lo), st)
        y = torch.empty((B * N, C), **f32)
        hip.call("nr_linear_x3", hip.ptr(cat_hi), hip.ptr(cat_lo), hip.ptr(sw.wconv_hi), hip.ptr(sw.wconv_lo), None,
                 _p(x), B * N, C, 3 * C, _p(y), st)
    else:
        cat = torch.empty((B * N, 3 * C), **f32)
        hip.call("nr_shift_concat", _p(x), B, N, C, _p(cat), st)
        y = torch.addmm(x.view(B * N, C), cat, sw.wcat)
    # LayerNorm, score, exp, norm1 + pairwise distances  (one launch, workgroup per sample)
    xn = torch.empty((B, N, C), **f32)
    score = torch.empty((B, N), **f32)
    tokw = torch.empty((B, N), **f32)
    dist = torch.empty((B, N, N), **f32)
    smax = torch.empty((B,), **f32)
    kvn = kvn_hi = kvn_lo = None
    if big:
        kvn_hi = torch.empty((B * N, C), dtype=torch.int16, device=dev)
        kvn_lo = torch.empty((B * N, C), dtype=torch.int16, device=dev)
    else:
        kvn = torch.empty((B * N, C), **f32)
    m = None
    if mask is not None:
        m = mask if mask.dtype == torch.float32 else mask.float()
        m = m.contiguous()
    attn = blk.attn
    hip.call("nr_ctm_front", _p(y), hip.ptr(m, allow_none=True), B, N, C, _p(ctm.norm.weight), _p(ctm.norm.bias),
             _p(ctm.score.weight), _p(ctm.score.bias), _p(blk.norm1.weight), _p(blk.norm1.bias), float(ctm.norm.eps),
             _p(xn), hip.ptr(kvn, allow_none=True), hip.ptr(kvn_hi, allow_none=True), hip.ptr(kvn_lo, allow_none=True),
             _p(score), _p(tokw), _p(dist), _p(smax), st)
    # DPC-KNN assignment + weighted cluster means + norm1  (one launch)
    cnum = max(math.ceil(N * ctm.sample_ratio), 1)
    if noise is None:
        noise = torch.rand((B, N), **f32)
    noise = noise.float().contiguous()
    merged = torch.empty((B * cnum, C), **f32)
    merged_pb = torch.empty((B * cnum, C), **f32)
    qn = torch.empty((B * cnum, C), **f32)
    hip.call("nr_ctm_back", _p(dist), _p(smax), hip.ptr(m, allow_none=True), _p(noise), _p(xn), _p(tokw), B, N, C, int(ctm.k),
             cnum, _p(blk.norm1.weight), _p(blk.norm1.bias), _p(attn.proj.bias), float(blk.norm1.eps), _p(merged),
             _p(merged_pb), _p(qn), None, st)
    # projections (library GEMMs) and the score-biased attention
    q = torch.addmm(attn.q.bias, qn, sw.wq_t) if attn.q.bias is not None else qn @ sw.wq_t
    if big:
        kv = torch.empty((B * N, 2 * C), **f32)
        hip.call("nr_linear_x3", hip.ptr(kvn_hi), hip.ptr(kvn_lo), hip.ptr(sw.wkv_hi), hip.ptr(sw.wkv_lo),
                 hip.ptr(attn.kv.bias, allow_none=True), None, B * N, 2 * C, C, _p(kv), st)
    else:
        kv = torch.addmm(attn.kv.bias, kvn, sw.wkv_t) if attn.kv.bias is not None else kvn @ sw.wkv_t
    att = torch.empty((B * cnum, C), **f32)
    hip.call("nr_tc_attention", _p(q), _p(kv), _p(score), B, N, C, cnum, attn.num_heads, _p(att), st)
    out = torch.addmm(merged_pb, att, sw.wp_t)          # merged + proj(att) + proj.bias
    return out.view(B, cnum, C)


def _addr(t):
    """Device address of a tensor (None stays None; an int is taken as an address already)."""
    return None if t is None else (t if isinstance(t, int) else t.data_ptr())


def _workspace_views(ws, B, N, C, cnum):
    """Views of the intermediates a backward pass needs, inside a stage's workspace (nr_ctm_stage_workspace_layout2): fp32
    tensors, and the bf16 pairs (int16) of norm1(xn), norm1(merged) and the attention output the forward GEMMs read."""
    import ctypes
    off = (ctypes.c_size_t * 14)()
    hip.call("nr_ctm_stage_workspace_layout2", B, N, C, cnum, off)
    shapes = {"y": (B, N, C), "xn": (B, N, C), "score": (B, N), "w": (B, N), "merged_pb": (B, cnum, C), "q": (B * cnum, C),
              "kv": (B * N, 2 * C), "smax": (B,)}
    pairs = {"kvn_hi": (B * N, C), "kvn_lo": (B * N, C), "qn_hi": (B * cnum, C), "qn_lo": (B * cnum, C),
             "att_hi": (B * cnum, C), "att_lo": (B * cnum, C)}
    out = {}
    for o, (name, shape) in zip(off, list(shapes.items()) + list(pairs.items())):
        n = 1
        for k in shape:
            n *= k
        if name in shapes:
            out[name] = ws[int(o): int(o) + 4 * n].view(torch.float32).view(shape)
        else:
            out[name] = ws[int(o): int(o) + 2 * n].view(torch.int16).view(shape)
    return out


def ctm_stage_group(problems, cache, stepwise=False, want_assign=False, want_saved=False, exchange=None):
    """One CTM + TCBlock stage of several independent problems (text and video) in the SAME seven launches
    (nr_ctm_stage_fwd).  problems: list of (key, x [B,N,C], mask or None, ctm, blk, noise or None).
    Returns the list of outputs [B,cnum,C]; with stepwise=True returns (outputs, generator) where every
    next() of the generator issues ONE of the seven launches on the then-current stream.  want_assign=True: returns
    (outputs, cluster ids [B,N] int64 per problem) -- what a backward pass needs to recompute the stage; want_saved=True:
    (outputs, per problem a dict of the stage's intermediates: views into its workspace + "x0", "assign", "mask").
    exchange: optional callable(list of per-problem smax tensors [B]) run between the front and the back launch of a stage
    that carries masks -- a rank clustering only ITS samples of a sharded batch puts the maximum over all ranks into
    smax[0] there (the masked stage uses the batch-wide maximum distance, cluster.py:473-475)."""
    import ctypes
    if not 0 < len(problems) <= hip.CTM_MAX_GROUP:
        raise hip.NrHipError(f"1..{hip.CTM_MAX_GROUP} problems per grouped stage")
    descs = (hip.CtmStageDesc * len(problems))()
    keep, outs, assigns, saved, smax = [], [], [], [], []
    build_stage_weights(cache, [(key, ctm, blk) for key, _, _, ctm, blk, _ in problems])       # one split launch, if stale
    for d, (key, x, mask, ctm, blk, noise) in zip(descs, problems):
        # a previous stage's output carries its bf16 pair (written by that stage's proj GEMM): no split launch for it here
        x_pair = getattr(x, "_nr_pair", None)
        x = x.detach().float().contiguous()
        B, N, C = x.shape
        if x_pair is not None and (x_pair[0].shape != (B * N, C) or x_pair[0].device != x.device):
            x_pair = None
        dev = x.device
        hip.ptr(x)                                           # device / contiguity check
        sw = cache[key][1]
        attn = blk.attn
        cnum = max(math.ceil(N * ctm.sample_ratio), 1)
        m = None
        if mask is not None:
            m = (mask if mask.dtype == torch.float32 else mask.float()).contiguous()
        if noise is None:
            noise = torch.rand((B, N), dtype=torch.float32, device=dev)
        noise = noise.float().contiguous()
        nbytes = int(hip.lib().nr_ctm_stage_workspace_bytes(B, N, C, cnum))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty((B, cnum, C), dtype=torch.float32, device=dev)
        out_hi = torch.empty((B * cnum, C), dtype=torch.int16, device=dev)
        out_lo = torch.empty((B * cnum, C), dtype=torch.int16, device=dev)
        out._nr_pair = (out_hi, out_lo)
        assign = torch.empty((B, N), dtype=torch.int64, device=dev) if (want_assign or want_saved) else None
        assigns.append(assign)
        if want_saved or exchange is not None:
            views = _workspace_views(ws, B, N, C, cnum)
            smax.append(views.pop("smax"))
            if want_saved:
                saved.append(dict(views, x0=x, assign=assign, mask=m))
        conv_bias = getattr(ctm.conv.conv, "bias", None)
        tensors = dict(x=x, mask=m, noise=noise, wconv_hi=sw.wconv_hi, wconv_lo=sw.wconv_lo, conv_bias=conv_bias,
                       ln_w=ctm.norm.weight, ln_b=ctm.norm.bias, sc_w=ctm.score.weight, sc_b=ctm.score.bias,
                       n1_w=blk.norm1.weight, n1_b=blk.norm1.bias, wq_hi=sw.wq_hi, wq_lo=sw.wq_lo, q_bias=attn.q.bias,
                       wkv_hi=sw.wkv_hi, wkv_lo=sw.wkv_lo, kv_bias=attn.kv.bias, wp_hi=sw.wp_hi, wp_lo=sw.wp_lo,
                       proj_bias=attn.proj.bias, workspace=ws, out=out, assign=assign,
                       x_hi=x_pair[0] if x_pair else None, x_lo=x_pair[1] if x_pair else None, out_hi=out_hi, out_lo=out_lo)
        d.n_samples, d.N, d.C, d.k, d.cnum, d.heads = B, N, C, int(ctm.k), cnum, int(attn.num_heads)
        d.eps_ctm, d.eps_n1 = float(ctm.norm.eps), float(blk.norm1.eps)
        for name, t in tensors.items():
            if t is not None and (not t.is_cuda or not t.is_contiguous()):
                raise hip.NrHipError(f"{name}: tensors of the clustering stage must be contiguous GPU tensors")
            setattr(d, name, _addr(t))
        keep.append(tensors)
        outs.append(out)
    if stepwise:
        if exchange is not None:
            raise hip.NrHipError("the exchange between front and back is not available in the stepwise form")

        def launches(alive=keep):
            for i in range(hip.CTM_STAGE_LAUNCHES):
                hip.call("nr_ctm_stage_fwd_range", descs, len(problems), i, i + 1, hip.stream_ptr())
                yield
            del alive                            # the tensors behind the descriptors stay alive until here
        return (outs, launches(), saved) if want_saved else (outs, launches())
    if exchange is not None and any(t["mask"] is not None for t in keep):
        hip.call("nr_ctm_stage_fwd_range", descs, len(problems), 0, 3, hip.stream_ptr())
        exchange(smax)
        hip.call("nr_ctm_stage_fwd_range", descs, len(problems), 3, hip.CTM_STAGE_LAUNCHES, hip.stream_ptr())
    else:
        hip.call("nr_ctm_stage_fwd", descs, len(problems), hip.stream_ptr())
    del keep
    if want_saved:
        return outs, saved
    return (outs, assigns) if want_assign else outs


_SAVED = ("x0", "y", "xn", "score", "w", "assign", "merged_pb", "q", "kv", "kvn_hi", "kvn_lo", "qn_hi", "qn_lo", "att_hi", "att_lo")
# The stage's backward: grouped HIP kernels (cluster_backward_hip.stage_backward_group: both modalities in the same nine
# launches) or, False / shapes they do not cover, the same arithmetic as torch ops (cluster_backward.stage_backward, ~125
# launches per modality) -- kept as the reference the kernels are tested against.
HIP_BACKWARD = True
# None: two streams inside a capture (6.3 -> 5.3 ms for the captured training step with the fused clustering), one when the
# step is launched eagerly (the stream switches cost the host more than the overlap returns: 8.4 vs 9.2 ms)
BACKWARD_ON_TWO_STREAMS = None
_BWD_OWNER = object()


def _backward_stream(device):
    from . import streams
    return streams.side(_BWD_OWNER, "cluster_bwd", device)          # cached for eager launches, new for every capture


def _torch_backward(ctx, saved, g_t, g_v):
    """cluster_backward.stage_backward (torch ops) for both modalities: the video stage's backward on a side stream beside the
    text stage's when the step is being captured (~95 small launches each), forked from and joined back into the stream
    autograd runs this node on."""
    from .cluster_backward import stage_backward
    cur = torch.cuda.current_stream()
    two = BACKWARD_ON_TWO_STREAMS if BACKWARD_ON_TWO_STREAMS is not None else torch.cuda.is_current_stream_capturing()
    # on the clustering's own stream (modeling._cluster_stream) this node is itself a fork of the step's stream: no nested fork
    two = two and g_t.is_cuda and may_fork(cur)
    side = _backward_stream(g_t.device) if (two and g_t.is_cuda) else None
    results = [None, None]
    with torch.no_grad():
        for i, ((ctm, blk), sv, g) in enumerate(zip(ctx.modules, saved, (g_t, g_v))):
            pb = blk.attn.proj.bias
            if i == 1 and side is not None:
                wait_stream(side, cur)
                with torch.cuda.stream(side):
                    sv["merged"] = sv["merged_pb"] - pb if pb is not None else sv["merged_pb"]
                    results[i] = stage_backward(ctm, blk, sv, g)
            else:
                sv["merged"] = sv["merged_pb"] - pb if pb is not None else sv["merged_pb"]
                results[i] = stage_backward(ctm, blk, sv, g)
        if side is not None:
            wait_stream(cur, side)
            for t_ in [results[1][0]] + list(results[1][1].values()):
                t_.record_stream(cur)
    return results


class ClusterStagesFn(torch.autograd.Function):
    """One CTM + TCBlock stage of the text AND the video tokens for the TRAINING step: the forward runs the grouped HIP
    kernels (7 launches for both modalities, like the loss-only step) and keeps what they leave in their workspaces; the
    backward is the hand-derived one of cluster_backward.stage_backward on those tensors (~45 torch launches per modality;
    DPC-KNN itself has no gradient: cluster.py:467 runs it under no_grad).  The reference -- and this package's traced
    path -- spend ~37 forward and ~75 backward launches per stage and modality and keep every intermediate alive."""

    @staticmethod
    def forward(ctx, modules, cache, keys, exchange, pre, x_t, mask_t, noise_t, x_v, mask_v, noise_v, *params):
        """pre: None, or ((out_t, out_v), saved) of a stage whose kernels have ALREADY been issued (ctm_stage_group(...,
        stepwise=True, want_saved=True), driven launch by launch from the head's interleaved schedule): this call then only
        creates the autograd node -- on the stream its backward is to run on."""
        (ctm_t, blk_t), (ctm_v, blk_v) = modules
        if pre is not None:
            (out_t, out_v), saved = pre
        else:
            (out_t, out_v), saved = ctm_stage_group([(keys[0], x_t, mask_t, ctm_t, blk_t, noise_t),
                                                     (keys[1], x_v, mask_v, ctm_v, blk_v, noise_v)], cache, want_saved=True,
                                                    exchange=exchange)
        ctx.modules = modules
        ctx.cache, ctx.keys = cache, keys
        ctx.masks = tuple(sv["mask"] for sv in saved)
        ctx.n_params = len(params)
        ctx.save_for_backward(*[sv[k] for sv in saved for k in _SAVED])
        return out_t, out_v

    @staticmethod
    def backward(ctx, g_t, g_v):
        tensors = ctx.saved_tensors
        if g_t.is_cuda:
            # The incoming gradients were produced on the head's stream, and -- when the forward's launches were driven from the
            # head's schedule -- so was everything this node saved: tell the allocator that THIS stream reads them, or the blocks
            # go back to the producing stream's free list the moment autograd drops them and are handed out again there while
            # this stream's kernels still read them (the captured step replayed garbage gradients for exactly that reason).
            cur = torch.cuda.current_stream()
            g_t.record_stream(cur)
            g_v.record_stream(cur)
            for t_ in tensors:
                if t_ is not None and t_.is_cuda:
                    t_.record_stream(cur)
        grads_x, grads_p = [], {}
        saved = []
        for i, mask in enumerate(ctx.masks):
            sv = dict(zip(_SAVED, tensors[i * len(_SAVED): (i + 1) * len(_SAVED)]))
            sv["mask"] = mask
            saved.append(sv)
        from . import cluster_backward_hip as CBH
        if HIP_BACKWARD and all(CBH.supported(sv) for sv in saved):
            with torch.no_grad():
                results = CBH.stage_backward_group([(ctx.keys[i], ctm, blk, saved[i], g)
                                                    for i, ((ctm, blk), g) in enumerate(zip(ctx.modules, (g_t, g_v)))], ctx.cache)
        else:
            results = _torch_backward(ctx, saved, g_t, g_v)
        for d_x0, gp in results:
            grads_x.append(d_x0)
            for p_, gr in gp.items():
                grads_p[id(p_)] = gr
        ordered = []
        for ctm, blk in ctx.modules:
            for p_ in stage_params(ctm, blk):
                ordered.append(grads_p.get(id(p_)) if p_.requires_grad else None)
        assert len(ordered) == ctx.n_params
        return (None, None, None, None, None, grads_x[0], None, None, grads_x[1], None, None) + tuple(ordered)


class _OnThisStream(torch.autograd.Function):
    """Identity.  Its backward runs on the stream its forward ran on (autograd's rule), so a gradient produced on ANOTHER stream
    -- the clustering's backward runs on the clustering's own stream -- reaches what lies behind this node on THIS stream."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        return gs


def route_on_this_stream(params, feats=()):
    """(dict id(parameter) -> pass-through tensor, [pass-through features]) through ONE identity node created on the CURRENT
    stream (the step's own).  The training step hands these to the clustering, which runs -- forward and backward -- on its own
    stream: the gradients of the clustering's parameters and of the leaf features then arrive at their AccumulateGrad nodes from
    a node of the step's stream, the stream those nodes live on (created at first use in the step, or at DDP construction).
    Without it every such gradient is accumulated across streams: one extra synchronisation per parameter and PyTorch's
    "AccumulateGrad node's stream does not match" warning in every step (GPUTEST_r03)."""
    if not torch.is_grad_enabled():
        return {}, list(feats)
    live = [t for t in list(params) + list(feats) if torch.is_tensor(t) and t.requires_grad]
    if not live:
        return {}, list(feats)
    out = dict(zip((id(t) for t in live), _OnThisStream.apply(*live)))
    return {id(p): out[id(p)] for p in params if id(p) in out}, [out.get(id(f), f) for f in feats]


def cluster_stages_train(modules, cache, keys, x_t, mask_t, noise_t, x_v, mask_v, noise_v, exchange=None, pre=None, routed=None):
    """Differentiable grouped stage (ClusterStagesFn); the stage's parameters ride along as explicit inputs so that
    autograd routes their gradients.  exchange: see ctm_stage_group (sample-sharded clustering); pre: see ClusterStagesFn;
    routed: route_on_this_stream's map (parameters passed through a node of the step's stream)."""
    params = [p for ctm, blk in modules for p in stage_params(ctm, blk)]
    if routed:
        params = [routed.get(id(p), p) for p in params]
    return ClusterStagesFn.apply(modules, cache, keys, exchange, pre, x_t, mask_t, noise_t, x_v, mask_v, noise_v, *params)
