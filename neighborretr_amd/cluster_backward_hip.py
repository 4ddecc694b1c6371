"""Backward of one token-clustering stage (CTM + TCBlock, reference cluster.py:453-561, 689-717, 834-888) on grouped HIP
kernels: the arithmetic of cluster_backward.stage_backward (hand-derived, checked against autograd in fp64), for the text
AND the video problem of a stage in the same launches -- nine per stage instead of ~2 x 125 torch launches:

    split    upstream gradient -> bf16 pairs; transposed operands (K = token rows) of the weight-gradient GEMMs
    GEMM     d_att = g Wp
    kernel   score-biased attention backward per sample                          (nr_ctm_attn_bwd)
    GEMM     d_qn = d_q Wq,  d_kvn = d_kv Wkv
    kernel   norm1 x2, residual, weighted cluster means, score, LayerNorm(ctm)   (nr_ctm_mid_bwd)
    GEMM     d_x0 = d_y + conv^T(d_y)                                  (transposed token convolution, read in place)
    split    transposes of d_q, d_kv, d_y
    GEMM     dWproj, dWq, dWkv, dWconv                                            (eight problems, one launch)
    colsum   bias gradients + sums of the per-sample LayerNorm / score parameter gradients

All GEMMs are split-bf16 (fp32-grade products) on the MFMA tile engine.  There is no torch fallback inside: a shape the
kernels do not cover raises NrHipError (callers choose the torch-op backward up front through `supported`).
"""
import torch

from . import hip
from .cluster_fused import _addr, _stage_weights, split_group


def supported(saved):
    """Shapes the grouped kernels cover: N <= 64 tokens, C = 64 * heads <= 512."""
    B, N, C = saved["xn"].shape
    return saved["xn"].is_cuda and N <= 64 and C % 64 == 0 and C <= 512


def _pad64(n):
    return (n + 63) // 64 * 64


def _linear_group(problems):
    """problems: (x_hi, x_lo, w_hi, w_lo, bias, residual, out, M, N, K[, ld[, conv_n]]); operands may be tensors or raw addresses
    (a K-slice of a wider matrix: address of its first column + ld).  conv_n: see NrLinearProblem."""
    for lo in range(0, len(problems), hip.LINEAR_GROUP_MAX):
        chunk = problems[lo:lo + hip.LINEAR_GROUP_MAX]
        arr = (hip.LinearProblem * len(chunk))()
        for a, prob in zip(arr, chunk):
            x_hi, x_lo, w_hi, w_lo, bias, res, out, M, N, K = prob[:10]
            a.x_hi, a.x_lo, a.w_hi, a.w_lo = _addr(x_hi), _addr(x_lo), _addr(w_hi), _addr(w_lo)
            a.bias, a.residual, a.out = _addr(bias), _addr(res), _addr(out)
            a.M, a.N, a.K, a.ld = int(M), int(N), int(K), int(prob[10]) if len(prob) > 10 else 0
            a.conv_n = int(prob[11]) if len(prob) > 11 else 0
        hip.call("nr_linear_group", len(chunk), arr, hip.stream_ptr())


def _colsum_group(items):
    """items: (src [rows, cols] f32, dst [cols], rows, cols[, scale])."""
    arr = (hip.ColsumItem * len(items))()
    for a, it in zip(arr, items):
        src, dst, rows, cols = it[:4]
        a.src, a.dst, a.rows, a.cols = _addr(src), _addr(dst), int(rows), int(cols)
        a.scale = float(it[4]) if len(it) > 4 else 1.0
    hip.call("nr_colsum_group", len(items), arr, hip.stream_ptr())


def stage_backward_group(problems, cache):
    """problems: list of (key, ctm, blk, saved, g) -- `saved` as left by cluster_fused.ctm_stage_group(want_saved=True)
    (workspace views incl. the bf16 pairs kvn / qn / att, plus x0, assign, mask), g = d loss / d stage output [B,c,C].
    Returns [(d loss / d x0 [B,N,C], {parameter: gradient}), ...] in the order of `problems`."""
    P = []
    for key, ctm, blk, sv, g in problems:
        B, N, C = sv["xn"].shape
        c = sv["merged_pb"].shape[1]
        dev = sv["xn"].device
        f32 = dict(dtype=torch.float32, device=dev)
        i16 = dict(dtype=torch.int16, device=dev)
        M, Mc = B * N, B * c
        Mp, Mcp = _pad64(M), _pad64(Mc)
        d = dict(key=key, ctm=ctm, blk=blk, sv=sv, B=B, N=N, C=C, c=c, M=M, Mc=Mc, Mp=Mp, Mcp=Mcp,
                 sw=_stage_weights(cache, key, ctm, blk), g=g.detach().float().contiguous().view(Mc, C))
        for name, shape in (("g_hi", (Mc, C)), ("g_lo", (Mc, C)), ("gT_hi", (C, Mcp)), ("gT_lo", (C, Mcp)),
                            ("attT_hi", (C, Mcp)), ("attT_lo", (C, Mcp)), ("qnT_hi", (C, Mcp)), ("qnT_lo", (C, Mcp)),
                            ("kvnT_hi", (C, Mp)), ("kvnT_lo", (C, Mp)), ("x0T3_hi", (3 * C, Mp)), ("x0T3_lo", (3 * C, Mp)),
                            ("dq_hi", (Mc, C)), ("dq_lo", (Mc, C)), ("dkv_hi", (M, 2 * C)), ("dkv_lo", (M, 2 * C)),
                            ("dy_hi", (M, C)), ("dy_lo", (M, C)), ("dqT_hi", (C, Mcp)), ("dqT_lo", (C, Mcp)),
                            ("dkvT_hi", (2 * C, Mp)), ("dkvT_lo", (2 * C, Mp)), ("dyT_hi", (C, Mp)), ("dyT_lo", (C, Mp))):
            d[name] = torch.empty(shape, **i16)
        for name, shape in (("d_att", (Mc, C)), ("d_q", (Mc, C)), ("d_kv", (M, 2 * C)), ("d_score", (B, N)), ("d_qn", (Mc, C)),
                            ("d_kvn", (M, C)), ("d_y", (M, C)), ("partial", (B, 6 * C)), ("d_x0", (M, C)), ("dWp", (C, C)),
                            ("dWq", (C, C)), ("dWkv", (2 * C, C)), ("dWc", (C, 3 * C)), ("dbp", (C,)), ("dbq", (C,)),
                            ("dbkv", (2 * C,)), ("psum", (6 * C,))):
            d[name] = torch.empty(shape, **f32)
        d["x0"] = sv["x0"].detach().float().contiguous().view(M, C)
        P.append(d)

    # 1. upstream gradient as a bf16 pair; transposed operands that exist before the backward starts
    items = []
    for d in P:
        sv, C, M, Mc, Mp, Mcp = d["sv"], d["C"], d["M"], d["Mc"], d["Mp"], d["Mcp"]
        items += [(d["g"], None, d["g_hi"], d["g_lo"], Mc, C, 0, C),
                  (d["g"], None, d["gT_hi"], d["gT_lo"], Mc, C, 1, Mcp),
                  (sv["att_hi"], sv["att_lo"], d["attT_hi"], d["attT_lo"], Mc, C, 2, Mcp),
                  (sv["qn_hi"], sv["qn_lo"], d["qnT_hi"], d["qnT_lo"], Mc, C, 2, Mcp),
                  (sv["kvn_hi"], sv["kvn_lo"], d["kvnT_hi"], d["kvnT_lo"], M, C, 2, Mp),
                  # the k=3 token neighbourhood of x0, transposed (row 3 i + s = x0[n + s - 1, i]): with it the convolution's
                  # weight gradient comes out of its GEMM as [C_out, C_in, 3], the parameter's own order
                  (d["x0"], None, d["x0T3_hi"], d["x0T3_lo"], M, C, 3, Mp, d["N"])]
    split_group(items)
    # 2. d_att = g Wp
    _linear_group([(d["g_hi"], d["g_lo"], d["sw"].wp_bt_hi, d["sw"].wp_bt_lo, None, None, d["d_att"], d["Mc"], d["C"], d["C"])
                   for d in P])
    # 3. attention backward
    arr = (hip.CtmAttnBwdDesc * len(P))()
    for a, d in zip(arr, P):
        sv = d["sv"]
        a.n_samples, a.N, a.C, a.cnum, a.heads = d["B"], d["N"], d["C"], d["c"], int(d["blk"].attn.num_heads)
        a.q, a.kv, a.score, a.d_att = _addr(sv["q"]), _addr(sv["kv"]), _addr(sv["score"]), _addr(d["d_att"])
        a.d_q, a.d_kv, a.d_score = _addr(d["d_q"]), _addr(d["d_kv"]), _addr(d["d_score"])
        a.dq_hi, a.dq_lo, a.dkv_hi, a.dkv_lo = _addr(d["dq_hi"]), _addr(d["dq_lo"]), _addr(d["dkv_hi"]), _addr(d["dkv_lo"])
    hip.call("nr_ctm_attn_bwd", len(P), arr, hip.stream_ptr())
    # 4. d_qn = d_q Wq, d_kvn = d_kv Wkv
    _linear_group([(d["dq_hi"], d["dq_lo"], d["sw"].wq_bt_hi, d["sw"].wq_bt_lo, None, None, d["d_qn"], d["Mc"], d["C"], d["C"]) for d in P]
                  + [(d["dkv_hi"], d["dkv_lo"], d["sw"].wkv_bt_hi, d["sw"].wkv_bt_lo, None, None, d["d_kvn"], d["M"], d["C"], 2 * d["C"])
                     for d in P])
    # 5. the middle of the stage
    arr = (hip.CtmMidBwdDesc * len(P))()
    for a, d in zip(arr, P):
        sv, ctm, blk = d["sv"], d["ctm"], d["blk"]
        a.n_samples, a.N, a.C, a.cnum = d["B"], d["N"], d["C"], d["c"]
        a.eps_ctm, a.eps_n1 = float(ctm.norm.eps), float(blk.norm1.eps)
        a.d_qn, a.d_kvn, a.g, a.merged_pb = _addr(d["d_qn"]), _addr(d["d_kvn"]), _addr(d["g"]), _addr(sv["merged_pb"])
        a.proj_b, a.xn, a.y, a.tokw = _addr(blk.attn.proj.bias), _addr(sv["xn"]), _addr(sv["y"]), _addr(sv["w"])
        a.d_score, a.mask = _addr(d["d_score"]), _addr(sv.get("mask"))
        a.n1_w, a.ln_w, a.sc_w = _addr(blk.norm1.weight), _addr(ctm.norm.weight), _addr(ctm.score.weight)
        a.assign, a.d_y = _addr(sv["assign"]), _addr(d["d_y"])
        a.dy_hi, a.dy_lo, a.partial = _addr(d["dy_hi"]), _addr(d["dy_lo"]), _addr(d["partial"])
    hip.call("nr_ctm_mid_bwd", len(P), arr, hip.stream_ptr())
    # 6. d_x0 = d_y + conv^T(d_y): the transposed k=3 token convolution read in place from d_y's bf16 pair (the kernel with its
    #    taps reversed: tap s meets row n + s - 1), the residual path as the GEMM's residual operand
    _linear_group([(d["dy_hi"], d["dy_lo"], d["sw"].wconv_bt_hi, d["sw"].wconv_bt_lo, None, d["d_y"], d["d_x0"], d["M"], d["C"],
                    3 * d["C"], 0, d["N"]) for d in P])
    # 7. transposes of what the backward produced
    items = []
    for d in P:
        C, M, Mc, Mp, Mcp = d["C"], d["M"], d["Mc"], d["Mp"], d["Mcp"]
        items += [(d["d_q"], None, d["dqT_hi"], d["dqT_lo"], Mc, C, 1, Mcp),
                  (d["d_kv"], None, d["dkvT_hi"], d["dkvT_lo"], M, 2 * C, 1, Mp),
                  (d["d_y"], None, d["dyT_hi"], d["dyT_lo"], M, C, 1, Mp)]
    split_group(items)
    # 8. weight gradients: K = token rows (zero-padded to a multiple of 64)
    probs = []
    for d in P:
        C, Mp, Mcp = d["C"], d["Mp"], d["Mcp"]
        probs += [(d["gT_hi"], d["gT_lo"], d["attT_hi"], d["attT_lo"], None, None, d["dWp"], C, C, Mcp),
                  (d["dqT_hi"], d["dqT_lo"], d["qnT_hi"], d["qnT_lo"], None, None, d["dWq"], C, C, Mcp),
                  (d["dkvT_hi"], d["dkvT_lo"], d["kvnT_hi"], d["kvnT_lo"], None, None, d["dWkv"], 2 * C, C, Mp),
                  (d["dyT_hi"], d["dyT_lo"], d["x0T3_hi"], d["x0T3_lo"], None, None, d["dWc"], C, 3 * C, Mp)]
    _linear_group(probs)
    # 9. bias gradients and the per-sample partial sums
    items = []
    for d in P:
        items += [(d["g"], d["dbp"], d["Mc"], d["C"]), (d["d_q"], d["dbq"], d["Mc"], d["C"]),
                  (d["d_kv"], d["dbkv"], d["M"], 2 * d["C"]), (d["partial"], d["psum"], d["B"], 6 * d["C"])]
    _colsum_group(items)

    out = []
    for d in P:
        ctm, blk, C = d["ctm"], d["blk"], d["C"]
        attn, n1, ps = blk.attn, blk.norm1, d["psum"]
        wconv = ctm.conv.conv.weight
        grads = {attn.proj.weight: d["dWp"], attn.q.weight: d["dWq"], attn.kv.weight: d["dWkv"],
                 n1.weight: ps[0:C], n1.bias: ps[C:2 * C], ctm.norm.weight: ps[2 * C:3 * C], ctm.norm.bias: ps[3 * C:4 * C],
                 ctm.score.weight: ps[4 * C:5 * C].reshape(1, C),
                 # dWc[o, 3 i + s] = d W[o, i, s]: already the parameter's layout
                 wconv: d["dWc"].view(wconv.shape)}
        if attn.proj.bias is not None:
            grads[attn.proj.bias] = d["dbp"]
        if attn.q.bias is not None:
            grads[attn.q.bias] = d["dbq"]
        if attn.kv.bias is not None:
            grads[attn.kv.bias] = d["dbkv"]
        if ctm.score.bias is not None:
            grads[ctm.score.bias] = ps[5 * C:5 * C + 1]
        out.append((d["d_x0"].view(d["B"], d["N"], C), grads))
    return out
