"""GPU parity of every HIP entry point against the CPU oracle (same seeded inputs).

Tolerances (floating point path; stated per SURVEY.md 8d / BASELINE.json north_star):
  * split-bf16 (BF16X3) contractions: <= 2e-6 on cosine sims (|S| <= 1)
  * single-pass bf16 contractions:    <= 1e-3 on sims ("logits within 1e-3")
  * fp32 kernels (softmax, Sinkhorn, row losses): <= 1e-4 relative to the oracle's fp32
"""
import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import hip, ops
from util import maxdiff, params, problem

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tw(prep, mask, P, prefix, n, N, prec):
    w1h, w1l = ops.split_bf16(P[prefix + ".0.weight"])
    parts = ops.token_logit_parts(prep, w1h, w1l, P[prefix + ".0.bias"], P[prefix + ".2.weight"].reshape(-1).contiguous(), prec)
    return ops.token_softmax(parts, P[prefix + ".2.bias"], mask, n, N, want_logits=True)


def test_library_loads_and_version():
    assert hip.version() == hip.ABI_VERSION == 5


@pytest.mark.parametrize("n,N,d", [(16, 24, 512), (5, 7, 256), (3, 64, 768)])
def test_prepare_tokens(n, N, d):
    g = torch.Generator().manual_seed(n * 100 + N)
    x = torch.randn(n, N, d, generator=g) * 3
    mask = (torch.rand(n, N, generator=g) > 0.3).long()
    prep = ops.prepare_tokens(x.to(DEV), mask.to(DEV), want_colsum=True)
    hi = prep.hi.view(torch.bfloat16).float().cpu()
    lo = prep.lo.view(torch.bfloat16).float().cpu()
    xn = torch.nn.functional.normalize(x.reshape(-1, d), dim=-1)
    ref = xn * mask.reshape(-1, 1)
    assert maxdiff(hi + lo, ref) < 2e-5                  # two bf16 terms carry ~16 mantissa bits
    assert maxdiff(hi, ref) < 5e-3
    assert maxdiff(prep.norm.cpu(), x.reshape(-1, d).norm(dim=-1)) < 1e-4 * float(x.norm(dim=-1).max())
    assert maxdiff(prep.colsum.sum(0).cpu(), xn.sum(0)) < 1e-4
    # masked rows are exact zero vectors
    assert float(hi[mask.reshape(-1) == 0].abs().max()) == 0.0


@pytest.mark.parametrize("prec,tol", [(hip.PREC_BF16X3, 2e-5), (hip.PREC_BF16, 3e-3)])
def test_token_weights(prec, tol):
    x = problem(1001, 16, 24, 12, 128)
    P = params()
    Pg = {k: v.to(DEV) for k, v in P.items()}
    for feat, mask, prefix in ((x["text_feat"], x["text_mask"], "text_weight_fc"),
                               (x["mb_feat_v"], x["mb_mask_v"], "video_weight_fc")):
        n, N, _ = feat.shape
        prep = ops.prepare_tokens(feat.to(DEV), mask.to(DEV))
        w, logits = _tw(prep, mask.to(DEV), Pg, prefix, n, N, prec)
        w_ref = O.token_weights(feat, mask, P, prefix)
        assert maxdiff(w, w_ref) < tol
        lg_ref = O.token_weight_logits(feat, P, prefix)
        valid = mask.bool()
        assert maxdiff(logits.cpu()[valid], lg_ref[valid]) < tol * 10


@pytest.mark.parametrize("B,Nt,Nv,prec", [(512, 24, 12, hip.PREC_BF16X3), (128, 64, 64, hip.PREC_BF16X3), (128, 24, 12, hip.PREC_BF16X3),
                                          (128, 24, 12, hip.PREC_BF16), (5, 7, 24, hip.PREC_BF16X3)])
def test_token_weights_pair_equals_the_two_launches(B, Nt, Nv, prec):
    """nr_token_weights_fwd_pair (the step's text and video scorers in one grid) against the two nr_token_weights_fwd launches:
    identical bits, masks included; counters left zeroed.  Built for crowded grids only (4-wave blocks on a one-deep ring: the
    first two cases); otherwise the entry point refuses and ops.token_weights_pair falls back to the single launches."""
    x = problem(1001 + B, B, Nt, Nv, 16)
    Pg = {k: v.to(DEV) for k, v in params().items()}
    tm, vm = x["text_mask"].to(DEV), x["video_mask"].to(DEV)
    pt = ops.prepare_tokens(x["text_feat"].to(DEV), tm)
    pv = ops.prepare_tokens(x["video_feat"].to(DEV), vm)

    def call(prep, mask, prefix, n, N):
        w1 = Pg[prefix + ".0.weight"]
        hi, lo = ops.split_bf16(w1)
        return (prep, hi, lo, Pg[prefix + ".0.bias"], Pg[prefix + ".2.weight"].reshape(-1).contiguous(), Pg[prefix + ".2.bias"], mask, n, N)
    calls = [call(pt, tm, "text_weight_fc", B, Nt), call(pv, vm, "video_weight_fc", B, Nv)]
    single = [ops.token_weights(*c, prec)[0] for c in calls]
    n0 = hip.N_CALLS
    (w_t, _), (w_v, _) = ops.token_weights_pair(calls, prec)
    torch.cuda.synchronize()
    if (B, Nt, Nv) in ((512, 24, 12), (128, 64, 64)):
        assert hip.N_CALLS - n0 == 1
    assert torch.equal(w_t, single[0]) and torch.equal(w_v, single[1])
    assert maxdiff(w_t, _tw(pt, tm, Pg, "text_weight_fc", B, Nt, prec)[0]) < 1e-5
    assert int((ops._COUNTERS[("softmax", w_t.device)][0] != 0).sum()) == 0


SHAPES = [
    # A, Nt, Bv, Nv
    (5, 24, 3, 64),        # mixed: 24 text tokens x 64 frames
    (3, 64, 21, 12),       # mixed: 64 x 12
    (16, 24, 16, 12),      # C1 batch x batch
    (16, 24, 128, 12),     # C1 batch x bank
    (37, 24, 19, 12),      # ragged tile edges
    (8, 64, 8, 64),        # ActivityNet token counts
    (9, 20, 11, 7),        # token counts that do not divide the tile
    (5, 3, 7, 6),          # merged-token counts of the C4 global level
    (4, 77, 3, 100),       # one sample per tile
]


@pytest.fixture(params=[0, 1], ids=["auto", "lds-epilogue"])
def sim_variant(request, monkeypatch):
    """0: let the library pick (register epilogue for 24/64-token shapes); 1: force the generic LDS
    epilogue of nr_sim.hip (NR_SIM_GENERIC is read by the host entry point on every call)."""
    monkeypatch.setenv("NR_SIM_GENERIC", str(request.param))
    return request.param


@pytest.mark.parametrize("A,Nt,Bv,Nv", SHAPES)
@pytest.mark.parametrize("prec,tol", [(hip.PREC_BF16X3, 2e-6), (hip.PREC_BF16, 1e-3)])
def test_local_level_matches_oracle(A, Nt, Bv, Nv, prec, tol, sim_variant):
    g = torch.Generator().manual_seed(A * 1000 + Bv)
    base = torch.randn(max(A, Bv), 1, 512, generator=g)
    t = base[:A] + 4 * torch.randn(A, Nt, 512, generator=g)
    v = base[:Bv] + 4 * torch.randn(Bv, Nv, 512, generator=g)
    tm = (torch.arange(Nt)[None] < torch.randint(1, Nt + 1, (A, 1), generator=g)).long()
    vm = (torch.arange(Nv)[None] < torch.randint(1, Nv + 1, (Bv, 1), generator=g)).long()
    if Bv > 2:
        vm[2] = 0                                   # a fully masked video: exact-zero column
    w_t = torch.softmax(torch.randn(A, Nt, generator=g), -1)
    w_v = torch.softmax(torch.randn(Bv, Nv, generator=g), -1)
    # oracle with the same token weights
    tn = torch.nn.functional.normalize(t, dim=-1) * tm[..., None]
    vn = torch.nn.functional.normalize(v, dim=-1) * vm[..., None]
    R = torch.einsum("atd,bvd->abtv", tn.double(), vn.double())
    pm, av = R.max(-1)
    qm, at = R.max(-2)
    S_ref = 0.5 * ((pm * w_t[:, None, :].double()).sum(-1) + (qm * w_v[None].double()).sum(-1))

    pt = ops.prepare_tokens(t.to(DEV), tm.to(DEV))
    pv = ops.prepare_tokens(v.to(DEV), vm.to(DEV))
    S, (arg_v, arg_t, pmax, qmax) = ops.local_level(pt, pv, w_t.to(DEV), w_v.to(DEV), A, Nt, Bv, Nv, prec, hip.OUT_FULL, want_arg=True)
    assert maxdiff(S, S_ref) < tol
    if Bv > 2:
        assert float(S[:, 2].abs().max()) == 0.0
    if prec == hip.PREC_BF16X3:
        # arg-max must point at an entry equal (to rounding) to the true max
        Rf = R.float()
        got_p = torch.gather(Rf, 3, arg_v.cpu().long()[..., None]).squeeze(-1)
        assert maxdiff(got_p, pm) < 1e-5
        got_q = torch.gather(Rf, 2, arg_t.cpu().long()[:, :, None, :]).squeeze(2)
        assert maxdiff(got_q, qm) < 1e-5
        assert maxdiff(pmax, pm) < 2e-6 and maxdiff(qmax, qm) < 2e-6
    # bank modes: row / column sums of the same matrix
    nr, nc = hip.local_level_tiles(A, Nt, Bv, Nv, prec)
    rs, _ = ops.local_level(pt, pv, w_t.to(DEV), w_v.to(DEV), A, Nt, Bv, Nv, prec, hip.OUT_ROWSUM)
    cs, _ = ops.local_level(pt, pv, w_t.to(DEV), w_v.to(DEV), A, Nt, Bv, Nv, prec, hip.OUT_COLSUM)
    assert rs.shape == (nc, A) and cs.shape == (nr, Bv)
    assert maxdiff(ops.reduce_parts(rs, 1.0 / Bv), S_ref.mean(1)) < tol
    assert maxdiff(ops.reduce_parts(cs, 1.0 / A), S_ref.mean(0)) < tol


def test_local_level_end_to_end_golden_c1():
    """prepare -> scorer -> similarity against the REFERENCE's own S (tests/golden/c1_b16.npz)."""
    from util import golden
    gd = golden("c1_b16")
    x = problem(1001, 16, 24, 12, 128, device=DEV)
    P = params(device=DEV)
    pt = ops.prepare_tokens(x["text_feat"], x["text_mask"])
    pv = ops.prepare_tokens(x["video_feat"], x["video_mask"])
    w_t, _ = _tw(pt, x["text_mask"], P, "text_weight_fc", 16, 24, hip.PREC_BF16X3)
    w_v, _ = _tw(pv, x["video_mask"], P, "video_weight_fc", 16, 12, hip.PREC_BF16X3)
    assert maxdiff(w_t, gd["w_t"]) < 2e-5 and maxdiff(w_v, gd["w_v"]) < 2e-5
    S, _ = ops.local_level(pt, pv, w_t, w_v, 16, 24, 16, 12, hip.PREC_BF16X3)
    assert maxdiff(S, gd["S"]) < 2e-6
    w_t1, _ = _tw(pt, x["text_mask"], P, "text_weight_fc", 16, 24, hip.PREC_BF16)
    w_v1, _ = _tw(pv, x["video_mask"], P, "video_weight_fc", 16, 12, hip.PREC_BF16)
    S1, _ = ops.local_level(pt, pv, w_t1, w_v1, 16, 24, 16, 12, hip.PREC_BF16)
    assert maxdiff(S1, gd["S"]) < 1e-3


@pytest.mark.parametrize("M,N,K", [(16, 16, 512), (128, 128, 512), (37, 50, 64), (8, 24, 512)])
def test_gemm_nt_f32(M, N, K):
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    c = ops.gemm_nt_f32(a.to(DEV), b.to(DEV))
    ref = a.double() @ b.double().t()
    assert maxdiff(c, ref) < 2e-5 * float(ref.abs().max())


def test_centrality_weights():
    x = problem(1001, 16, 24, 12, 128)
    g = torch.Generator().manual_seed(3)
    gt = torch.randn(16, 1, 512, generator=g)
    gv = torch.randn(16, 1, 512, generator=g)
    wt_ref, wv_ref = O.centrality_weights(x["text_feat"], x["video_feat"], gt, gv, 0.3)
    pt = ops.prepare_tokens(x["text_feat"].to(DEV), x["text_mask"].to(DEV), want_colsum=True)
    pv = ops.prepare_tokens(x["video_feat"].to(DEV), x["video_mask"].to(DEV), want_colsum=True)
    wt, _, _ = ops.centrality_weights(gt[:, 0].to(DEV), pt.colsum, pt.n_tok, 0.3)
    wv, _, _ = ops.centrality_weights(gv[:, 0].to(DEV), pv.colsum, pv.n_tok, 0.3)
    assert maxdiff(wt, wt_ref) < 1e-6 and maxdiff(wv, wv_ref) < 1e-6


@pytest.mark.parametrize("B", [16, 100, 128, 200])
def test_sinkhorn_targets(B):
    g = torch.Generator().manual_seed(B)
    G = torch.randn(B, B, generator=g) * 9 + torch.eye(B) * 10
    tr, tc = ops.sinkhorn_targets(G.to(DEV), 0.7, 50)
    ref_r = O.sinkhorn_targets(G.double(), 0.7)
    ref_c = O.sinkhorn_targets(G.double().t(), 0.7)
    assert maxdiff(tr, ref_r) < 2e-5
    assert maxdiff(tc, ref_c) < 2e-5
    # column sums of Q are 1/... : property the reference's plan has
    Q = (tr.cpu().double() - 0.3 * torch.eye(B, dtype=torch.float64)) / 0.7
    assert maxdiff(Q.sum(0), torch.ones(B, dtype=torch.float64)) < 1e-4


@pytest.mark.parametrize("B,K", [(16, 8), (32, 8), (128, 20), (200, 20), (16, 15), (16, 16), (13, 13), (4, 4), (2, 2), (13, 5), (7, 1)])
def test_row_losses(B, K):
    # (K = B - 1: nothing is left outside the neighbour set, min / max fall back to +-9e15 and every adjusted similarity is 0;
    #  K = B: the sort's last entry -- the diagonal -- is a "neighbour" too and sits in the softmax's denominator, until_module.py:119-123)
    g = torch.Generator().manual_seed(B + K)
    S = torch.rand(B, B, generator=g) * 0.12 + torch.eye(B) * 0.02
    G = torch.randn(B, B, generator=g) * 9
    c0 = torch.rand(B, generator=g) * 0.1
    c1 = torch.rand(B, generator=g) * 0.1
    wt = torch.exp(torch.randn(B, generator=g) * 0.01)
    wv = torch.exp(torch.randn(B, generator=g) * 0.01)
    ls = torch.tensor([100.0])
    T = 3.0
    d = lambda t: t.double()
    tr = O.sinkhorn_targets(d(G), 0.7)
    tc = O.sinkhorn_targets(d(G).t(), 0.7)
    # oracle, term by term (bank matrices with the given row means)
    bank_v2t = d(c0)[:, None].expand(B, 4)
    bank_t2v = d(c1)[:, None].expand(B, 4)
    ref = [O.centrality_loss(d(S), d(wt), d(wv), 100.0),
           (-(torch.log_softmax(d(G) * T, -1) * tr).sum(-1).mean() - (torch.log_softmax(d(G).t() * T, -1) * tc).sum(-1).mean()) / 2,
           O.neighbor_loss(d(S), bank_t2v, bank_v2t, K, T),
           O.kl_loss(d(G), d(S))]
    rl = ops.row_losses(S.to(DEV), G.to(DEV), tr.float().contiguous().to(DEV), tc.float().contiguous().to(DEV), c0.to(DEV), c1.to(DEV),
                        wt.to(DEV), wv.to(DEV), ls.to(DEV), K, T)
    losses = ops.loss_finalize(rl, 1.0, 1.0, 1.0).cpu().double()
    for k, r in enumerate(ref):
        assert abs(float(losses[k + 1]) - float(r)) < 1e-4 * max(1.0, abs(float(r))), (k, float(losses[k + 1]), float(r))
    assert abs(float(losses[0]) - float(sum(ref))) < 2e-4 * float(sum(ref))


def test_row_losses_one_sample_outside_the_neighbour_set_is_nan_as_in_the_reference():
    """K = B - 2: exactly ONE sample is left outside {diagonal, neighbours}; its min-max normalisation divides 0 by 0
    (until_module.py:85) and the reference's neighbour loss is NaN.  Same here -- not a clamp, not a finite number."""
    B, K = 13, 11
    g = torch.Generator().manual_seed(1)
    S = torch.rand(B, B, generator=g) * 0.12
    G = torch.randn(B, B, generator=g)
    v = lambda: torch.rand(B, generator=g) * 0.1
    c0, c1 = v(), v()
    assert torch.isnan(O.neighbor_loss(S, c1[:, None].expand(B, 4), c0[:, None].expand(B, 4), K, 3.0))
    tr, tc = ops.sinkhorn_targets(G.to(DEV), 0.7, 50)
    rl = ops.row_losses(S.to(DEV), G.to(DEV), tr, tc, c0.to(DEV), c1.to(DEV), (1 + v()).to(DEV), (1 + v()).to(DEV),
                        torch.tensor([100.0], device=DEV), K, 3.0)
    losses = ops.loss_finalize(rl, 1.0, 1.0, 1.0).cpu()
    assert torch.isnan(losses[3]) and torch.isnan(losses[0]) and torch.isfinite(losses[[1, 2, 4]]).all()


@pytest.mark.parametrize("B,K", [(16, 8), (16, 15), (16, 16), (13, 13), (9, 1), (130, 20)])
def test_neighbour_term_gradient_matches_oracle_autograd(B, K):
    """d (neighbour loss) / d S and d / d bank centralities through nr_row_losses_bwd against the oracle's autograd (fp64), at the
    edges of K too: K = B - 1 (nothing outside the neighbour set), K = B (the diagonal inside the softmax, until_module.py:119-123)."""
    from neighborretr_amd.backward import RowLossFn
    g = torch.Generator().manual_seed(7 * B + K)
    S0 = torch.rand(B, B, generator=g) * 0.12 + torch.eye(B) * 0.02
    G = torch.randn(B, B, generator=g)
    c0_0, c1_0 = torch.rand(B, generator=g) * 0.1, torch.rand(B, generator=g) * 0.1
    T = 3.0
    Sd, c0d, c1d = (t.double().requires_grad_() for t in (S0, c0_0, c1_0))
    ref = O.neighbor_loss(Sd, c1d[:, None].expand(B, 4), c0d[:, None].expand(B, 4), K, T)
    ref.backward()
    S, c0, c1 = (t.to(DEV).requires_grad_() for t in (S0, c0_0, c1_0))
    tr, tc = ops.sinkhorn_targets(G.to(DEV), 0.7, 50)
    one = torch.ones(B, device=DEV)
    rl = RowLossFn.apply(S, G.to(DEV), tr, tc, c0, c1, one, one, torch.tensor([100.0], device=DEV), K, T)
    loss = rl[:, 2, :].mean()                       # (t2v rows + v2t rows) / 2B = the reference's neighbour loss
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    loss.backward()
    for got, want in ((S.grad, Sd.grad), (c0.grad, c0d.grad), (c1.grad, c1d.grad)):      # fp32 kernel against the fp64 oracle
        assert maxdiff(got, want) < 1e-3 * float(want.abs().max()) + 1e-9


def test_row_losses_rejects_bad_k():
    B = 16
    z = torch.zeros(B, B, device=DEV)
    v = torch.zeros(B, device=DEV)
    with pytest.raises(hip.NrHipError):
        ops.row_losses(z, z, z, z, v, v, v, v, torch.ones(1, device=DEV), 20, 3.0)   # K > B: reference raises too


@pytest.mark.parametrize("B,N,cnum,masked", [(16, 24, 4, True), (16, 12, 3, True), (16, 4, 1, False), (16, 3, 1, False),
                                             (8, 64, 11, True), (8, 16, 6, False), (128, 12, 3, True)])
def test_dpc_knn_assign_matches_oracle(B, N, cnum, masked):
    g = torch.Generator().manual_seed(B * 100 + N)
    x = torch.randn(B, N, 512, generator=g)
    x = torch.nn.functional.layer_norm(x + 0.5 * x[:, :1], (512,))
    mask = None
    if masked:
        ln = torch.randint(1, N + 1, (B, 1), generator=g)
        ln[0] = 1                                   # a sample with a single valid token
        ln[1] = min(2, N)
        mask = (torch.arange(N)[None] < ln).long()
    noise = torch.rand(B, N, generator=g)
    ref = O.dpc_knn(x, cnum, 3, mask, noise)
    got = ops.dpc_knn_assign(x.to(DEV), cnum, 3, None if mask is None else mask.to(DEV), noise.to(DEV)).cpu()
    # every sample and every token (padding included): the oracle's documented tie rule -- exact ties in the centre
    # score go to the lower index (dpc_knn(centre_ties="lowest_index")) -- is the kernel's rule, so samples with
    # fewer valid tokens than centres (zero-score ties among padding tokens) agree too
    assert torch.equal(got, ref)


@pytest.mark.parametrize("M,N,K,bias,res", [(3072, 512, 1536, False, True), (1536, 1024, 512, True, False),
                                             (200, 130, 64, True, True), (64, 512, 512, False, False)])
def test_linear_x3_matches_fp64(M, N, K, bias, res):
    """The split-bf16 linear kernel of the clustering GEMMs against an fp64 product (~fp32 accuracy)."""
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn(M, N, generator=g) if res else None
    xh, xl = ops.split_bf16(x.to(DEV))
    wh, wl = ops.split_bf16(w.to(DEV))
    out = torch.empty(M, N, device=DEV)
    hip.call("nr_linear_x3", hip.ptr(xh), hip.ptr(xl), hip.ptr(wh), hip.ptr(wl), hip.ptr(b.to(DEV) if bias else None, allow_none=True),
             hip.ptr(r.to(DEV) if res else None, allow_none=True), M, N, K, hip.ptr(out), hip.stream_ptr())
    ref = x.double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    if res:
        ref = ref + r.double()
    assert maxdiff(out, ref) < 3e-5 * float(ref.abs().max())


@pytest.mark.parametrize("sets", [[(128, 24), (128, 12)], [(128, 4), (128, 3)], [(5, 24), (7, 12)], [(32, 20), (32, 9)], [(3, 64)]])
def test_token_convolution_launch_matches_fp64(sets):
    """nr_linear_group with conv_n > 0 (the k=3, padding=1 token convolution read in place, cluster.py:664, residual and bias folded
    in) against an fp64 convolution of the same split-bf16 operands: both modalities in one launch, ragged row counts, token
    counts that do not divide the block height."""
    from neighborretr_amd.cluster_backward_hip import _linear_group
    C = 512
    g = torch.Generator().manual_seed(5)

    def split(t):
        hi = t.to(torch.bfloat16)
        lo = (t - hi.float()).to(torch.bfloat16)
        return hi.view(torch.int16), lo.view(torch.int16), hi.double() + lo.double()

    probs, refs = [], []
    for B, n in sets:
        x = torch.randn(B * n, C, generator=g).to(DEV)
        w = (torch.randn(C, 3 * C, generator=g) * 0.03).to(DEV)
        bias = torch.randn(C, generator=g).to(DEV)
        xh, xl, xd = split(x)
        wh, wl, wd = split(w)
        out = torch.zeros(B * n, C, device=DEV)
        probs.append((xh, xl, wh, wl, bias, x, out, B * n, C, 3 * C, 0, n))
        xs = xd.view(B, n, C)
        z = torch.zeros(B, 1, C, dtype=torch.float64, device=DEV)
        cat = torch.cat([torch.cat([z, xs[:, :-1]], 1), xs, torch.cat([xs[:, 1:], z], 1)], 2).view(B * n, 3 * C)
        refs.append(cat @ wd.t() + bias.double() + x.double())
    _linear_group(probs)
    torch.cuda.synchronize()
    for p_, r in zip(probs, refs):
        assert maxdiff(p_[6], r) < 3e-6 * float(r.abs().max())


def test_ctm_front_back_equal_the_separate_kernels():
    """nr_ctm_front / nr_ctm_back (what the step runs) == nr_ctm_norm_score + nr_dpc_knn_assign + nr_merge_ln."""
    B, N, C, cnum = 16, 24, 512, 4
    g = torch.Generator().manual_seed(3)
    y = torch.randn(B, N, C, generator=g).to(DEV)
    mask = (torch.arange(N)[None] < torch.randint(3, N + 1, (B, 1), generator=g)).float().to(DEV)
    noise = torch.rand(B, N, generator=g).to(DEV)
    vec = lambda n, s=1.0: (torch.randn(n, generator=g) * s).to(DEV)
    ln_w, ln_b, sc_w, sc_b = 1 + vec(C, 0.05), vec(C, 0.01), vec(C, 0.05).reshape(1, C), vec(1, 0.01)
    n1_w, n1_b, pb = 1 + vec(C, 0.05), vec(C, 0.01), vec(C, 0.01)
    f = lambda *s: torch.empty(*s, device=DEV)
    xn, kvn, score, tokw = f(B, N, C), f(B * N, C), f(B, N), f(B, N)
    hip.call("nr_ctm_norm_score", hip.ptr(y), hip.ptr(mask), B * N, C, hip.ptr(ln_w), hip.ptr(ln_b), hip.ptr(sc_w), hip.ptr(sc_b),
             hip.ptr(n1_w), hip.ptr(n1_b), 1e-5, hip.ptr(xn), hip.ptr(kvn), hip.ptr(score), hip.ptr(tokw), hip.stream_ptr())
    assign = ops.dpc_knn_assign(xn, cnum, 3, mask, noise)
    merged, mpb, qn = f(B * cnum, C), f(B * cnum, C), f(B * cnum, C)
    hip.call("nr_merge_ln", hip.ptr(xn), hip.ptr(assign), hip.ptr(tokw), B, N, C, cnum, hip.ptr(n1_w), hip.ptr(n1_b), hip.ptr(pb),
             1e-5, hip.ptr(merged), hip.ptr(mpb), hip.ptr(qn), hip.stream_ptr())
    xn2, kvn2, score2, tokw2, dist, smax = f(B, N, C), f(B * N, C), f(B, N), f(B, N), f(B, N, N), f(B)
    hip.call("nr_ctm_front", hip.ptr(y), hip.ptr(mask), B, N, C, hip.ptr(ln_w), hip.ptr(ln_b), hip.ptr(sc_w), hip.ptr(sc_b),
             hip.ptr(n1_w), hip.ptr(n1_b), 1e-5, hip.ptr(xn2), hip.ptr(kvn2), None, None, hip.ptr(score2), hip.ptr(tokw2),
             hip.ptr(dist), hip.ptr(smax), hip.stream_ptr())
    merged2, mpb2, qn2 = f(B * cnum, C), f(B * cnum, C), f(B * cnum, C)
    assign2 = torch.empty(B, N, dtype=torch.int64, device=DEV)
    hip.call("nr_ctm_back", hip.ptr(dist), hip.ptr(smax), hip.ptr(mask), hip.ptr(noise), hip.ptr(xn2), hip.ptr(tokw2), B, N, C, 3, cnum,
             hip.ptr(n1_w), hip.ptr(n1_b), hip.ptr(pb), 1e-5, hip.ptr(merged2), hip.ptr(mpb2), hip.ptr(qn2), hip.ptr(assign2),
             hip.stream_ptr())
    assert torch.equal(xn, xn2) and torch.equal(kvn, kvn2) and torch.equal(tokw, tokw2) and torch.equal(assign, assign2)
    # cluster weight totals and LayerNorm statistics are tree-summed in the fused kernel: equal to rounding
    assert maxdiff(merged, merged2) < 2e-6 * float(merged.abs().max()) and maxdiff(mpb, mpb2) < 2e-6 * float(mpb.abs().max())
    assert maxdiff(qn, qn2) < 4e-6 * float(qn.abs().max())
    # the split-bf16 form of norm1(xn) carries it to ~2^-16
    kh = torch.empty(B * N, C, dtype=torch.int16, device=DEV)
    kl = torch.empty(B * N, C, dtype=torch.int16, device=DEV)
    hip.call("nr_ctm_front", hip.ptr(y), hip.ptr(mask), B, N, C, hip.ptr(ln_w), hip.ptr(ln_b), hip.ptr(sc_w), hip.ptr(sc_b),
             hip.ptr(n1_w), hip.ptr(n1_b), 1e-5, hip.ptr(xn2), None, hip.ptr(kh), hip.ptr(kl), hip.ptr(score2), hip.ptr(tokw2),
             hip.ptr(dist), hip.ptr(smax), hip.stream_ptr())
    rec = kh.view(torch.bfloat16).float() + kl.view(torch.bfloat16).float()
    assert maxdiff(rec, kvn) < 3e-5 * float(kvn.abs().max())


@pytest.mark.parametrize("B,Nt,Nv", [(32, 24, 12), (8, 64, 64), (6, 20, 9)])
def test_grouped_clustering_stage_equals_the_per_modality_path(B, Nt, Nv):
    """nr_ctm_stage_fwd (text and video problems in the same seven launches) against the one-problem fused
    kernels + library GEMMs: identical cluster assignments, outputs to split-bf16 GEMM accuracy."""
    from neighborretr_amd import modeling, synth
    from neighborretr_amd.cluster_fused import ctm_stage_fused, ctm_stage_group
    m = modeling.NeighborRetr(modeling.default_config())
    m.load_state_dict(params(), strict=False)
    m = m.to(DEV).eval()
    prob = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(77, B, Nt, Nv, 8).items()}
    nz = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_noise(77, B, Nt, Nv).items()}
    tm, vm = prob["text_mask"].float(), prob["video_mask"].float()
    with torch.no_grad():
        rt0 = ctm_stage_fused(prob["text_feat"], tm, m.text_ctm0, m.text_block0, nz["t0"], {}, "t0")
        rv0 = ctm_stage_fused(prob["video_feat"], vm, m.video_ctm0, m.video_block0, nz["v0"], {}, "v0")
        rt1 = ctm_stage_fused(rt0, None, m.text_ctm1, m.text_block1, nz["t1"], {}, "t1")
        rv1 = ctm_stage_fused(rv0, None, m.video_ctm1, m.video_block1, nz["v1"], {}, "v1")
        cache = {}
        gt0, gv0 = ctm_stage_group([("t0", prob["text_feat"], tm, m.text_ctm0, m.text_block0, nz["t0"]),
                                    ("v0", prob["video_feat"], vm, m.video_ctm0, m.video_block0, nz["v0"])], cache)
        gt1, gv1 = ctm_stage_group([("t1", gt0, None, m.text_ctm1, m.text_block1, nz["t1"]),
                                    ("v1", gv0, None, m.video_ctm1, m.video_block1, nz["v1"])], cache)
        # a single problem goes through the same entry point
        (st0,) = ctm_stage_group([("t0", prob["text_feat"], tm, m.text_ctm0, m.text_block0, nz["t0"])], cache)
    assert torch.equal(st0, gt0)
    for a, b in ((gt0, rt0), (gv0, rv0), (gt1, rt1), (gv1, rv1)):
        assert a.shape == b.shape
        assert maxdiff(a, b) < 2e-5 * max(1.0, float(b.abs().max()))


def test_step_prologue_masks_scale_and_noise():
    g = torch.Generator().manual_seed(0)
    m0 = (torch.rand(128, 24, generator=g) > 0.3).long().to(DEV)
    m1 = (torch.rand(128, 12, generator=g) > 0.3).long().to(DEV)
    ls = torch.tensor(2.6593, device=DEV)
    rng = torch.tensor([1234, 0], dtype=torch.int64, device=DEV)
    o0, o1, e, n1 = ops.step_prologue(m0, m1, ls, rng, 5504)
    assert torch.equal(o0, m0.float()) and torch.equal(o1, m1.float())
    assert abs(float(e) - float(torch.exp(ls))) < 2e-6 * float(torch.exp(ls))
    assert int(rng[1]) == 1 and int(rng[0]) == 1234
    _, _, _, n2 = ops.step_prologue(m0, m1, None, rng, 5504)
    assert int(rng[1]) == 2
    for n in (n1, n2):
        assert float(n.min()) >= 0.0 and float(n.max()) < 1.0 and abs(float(n.mean()) - 0.5) < 0.02
    assert not torch.equal(n1, n2) and len(torch.unique(n1)) > 5400
    # the same (seed, counter) reproduces the same stream
    rng2 = torch.tensor([1234, 0], dtype=torch.int64, device=DEV)
    assert torch.equal(ops.step_prologue(None, None, None, rng2, 5504)[3], n1)
    # fp32 masks pass through; a captured graph draws fresh numbers at every replay
    f0 = m0.float()
    assert ops.step_prologue(f0, None, None, rng, 0)[0] is f0
    gph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.step_prologue(m0, m1, ls, rng, 64)
        torch.cuda.synchronize()
        with torch.cuda.graph(gph, stream=s):
            out = ops.step_prologue(m0, m1, ls, rng, 64)[3]
    gph.replay()
    torch.cuda.synchronize()
    a = out.clone()
    gph.replay()
    torch.cuda.synchronize()
    assert not torch.equal(a, out)


@pytest.mark.parametrize("B", [16, 128, 300])
def test_row_losses_final_is_bit_identical_to_two_launches(B):
    g = torch.Generator().manual_seed(B)
    S = (torch.rand(B, B, generator=g) * 0.1).to(DEV)
    G = (torch.randn(B, B, generator=g) * 3).to(DEV)
    tr, tc = ops.sinkhorn_targets(G, 0.7, 50)
    v = lambda: (torch.rand(B, generator=g) * 0.1).to(DEV)
    c0, c1, w0, w1 = v(), v(), 1 + v(), 1 + v()
    ls = torch.tensor([100.0], device=DEV)
    K = min(20, B)
    rl = ops.row_losses(S, G, tr, tc, c0, c1, w0, w1, ls, K, 3.0)
    ref = ops.loss_finalize(rl, 1.0, 0.5, 2.0)
    for _ in range(3):                                   # the counter resets itself
        rl2, fused = ops.row_losses_final(S, G, tr, tc, c0, c1, w0, w1, ls, K, 3.0, 1.0, 0.5, 2.0)
        assert torch.equal(rl, rl2) and torch.equal(ref, fused)


def _bwd_problem(A, Bv, Nt, Nv, d, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(A, Nt, d, generator=g).to(DEV)
    v = torch.randn(Bv, Nv, d, generator=g).to(DEV)
    tm = (torch.arange(Nt)[None] < torch.randint(2, Nt + 1, (A, 1), generator=g)).float().to(DEV)
    vm = (torch.arange(Nv)[None] < torch.randint(2, Nv + 1, (Bv, 1), generator=g)).float().to(DEV)
    pt, pv = ops.prepare_tokens(t, tm), ops.prepare_tokens(v, vm)
    w_t = torch.softmax(torch.randn(A, Nt, generator=g), -1).to(DEV)
    w_v = torch.softmax(torch.randn(Bv, Nv, generator=g), -1).to(DEV)
    _, aux = ops.local_level(pt, pv, w_t, w_v, A, Nt, Bv, Nv, hip.PREC_BF16X3, hip.OUT_FULL, want_arg=True)
    return pt, pv, w_t, w_v, aux, g


@pytest.mark.parametrize("side,mode,A,Bv,use_lo,Nt,Nv,d", [
    (0, 0, 40, 56, True, 24, 12, 512), (1, 0, 40, 56, False, 24, 12, 512), (0, 1, 16, 200, False, 24, 12, 512),
    (1, 2, 200, 16, True, 24, 12, 512), (0, 0, 3, 5, False, 24, 12, 512), (1, 1, 130, 70, False, 24, 12, 512),
    (0, 2, 37, 41, False, 12, 24, 512), (1, 0, 37, 41, True, 24, 24, 256), (0, 0, 21, 50, False, 16, 24, 768),
    (1, 0, 9, 11, False, 12, 12, 512)])
def test_mfma_backward_matches_the_scalar_walk(side, mode, A, Bv, use_lo, Nt, Nv, d):
    """nr_local_level_bwd_group (routing-matrix blocks on the matrix cores, all three kernel variants: 512 dims per workgroup,
    256 dims with / without the low halves) and nr_pool_weight_bwd_group against nr_sim_bwd_kernel's entry-by-entry walk:
    same d_x to bf16-coefficient accuracy, same d_w to summation order.  12 x 12 tokens have too many sample pairs per block
    for the matrix-core kernel and must fall back to the walk."""
    pt, pv, w_t, w_v, aux, g = _bwd_problem(A, Bv, Nt, Nv, d, A * 7 + Bv + side)
    dS = (torch.randn(A, Bv, generator=g) if mode == 0 else torch.randn(A if mode == 1 else Bv, generator=g)).to(DEV)
    other, ws_, wo_ = (pv, w_t, w_v) if side == 0 else (pt, w_v, w_t)
    assert bool(hip.lib().nr_local_level_bwd_mfma_supported(Nt, Nv, d)) == ((Nt, Nv) != (12, 12))
    try:
        ops.USE_MFMA_BACKWARD = False
        dx_ref, dw_ref = ops.local_level_bwd(side, dS, mode, 0.37, other, ws_, wo_, aux, A, Nt, Bv, Nv, use_lo=use_lo, scalar_dw=True)
        ops.USE_MFMA_BACKWARD = True
        dx, dw = ops.local_level_bwd(side, dS, mode, 0.37, other, ws_, wo_, aux, A, Nt, Bv, Nv, use_lo=use_lo)
        base, wbase = dx.clone(), dw.clone()
        dx2, dw2 = ops.local_level_bwd(side, dS, mode, 0.37, other, ws_, wo_, aux, A, Nt, Bv, Nv, d_x=base, d_w=wbase,
                                       accumulate=True, use_lo=use_lo)
    finally:
        ops.USE_MFMA_BACKWARD = True
    assert maxdiff(dw, dw_ref) < 2e-6 * max(float(dw_ref.abs().max()), 1e-6)
    assert maxdiff(dw2, 2 * dw) < 2e-6 * max(float(dw_ref.abs().max()), 1e-6)
    scale = float(dx_ref.abs().max())
    assert maxdiff(dx, dx_ref) < 4e-3 * scale                 # coefficients rounded to bf16
    assert float((dx - dx_ref).abs().mean()) < 4e-4 * scale
    assert maxdiff(dx2, 2 * dx) < 1e-5 * scale


@pytest.mark.parametrize("use_lo", [False, True])
def test_grouped_backward_equals_the_products_one_by_one(use_lo):
    """The loss step's four products in one nr_local_level_bwd_group launch (two gradients, two products each) and its six
    weight sums in one nr_pool_weight_bwd_group launch == the same products launched one by one with accumulate."""
    B, M, Nt, Nv, d = 20, 70, 24, 12, 512
    pt, pv, w_t, w_v, aux0, g = _bwd_problem(B, B, Nt, Nv, d, 5)
    _, pbv, _, w_bv, aux1, _ = _bwd_problem(B, M, Nt, Nv, d, 6)
    pbt, _, w_bt, _, aux2, _ = _bwd_problem(M, B, Nt, Nv, d, 7)
    # the bank products pair THIS batch with the bank: recompute their stored indices with the shared operands
    _, aux1 = ops.local_level(pt, pbv, w_t, w_bv, B, Nt, M, Nv, hip.PREC_BF16X3, hip.OUT_FULL, want_arg=True)
    _, aux2 = ops.local_level(pbt, pv, w_bt, w_v, M, Nt, B, Nv, hip.PREC_BF16X3, hip.OUT_FULL, want_arg=True)
    dS, d_c1, d_c0 = torch.randn(B, B, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV)
    # one by one
    r_tn, r_wt = ops.local_level_bwd(0, dS, 0, 1.0, pv, w_t, w_v, aux0, B, Nt, B, Nv, use_lo=use_lo)
    r_vn, r_wv = ops.local_level_bwd(1, dS, 0, 1.0, pt, w_v, w_t, aux0, B, Nt, B, Nv, use_lo=use_lo)
    ops.local_level_bwd(0, d_c1, 1, 1.0 / M, pbv, w_t, w_bv, aux1, B, Nt, M, Nv, d_x=r_tn, d_w=r_wt, accumulate=True, use_lo=use_lo)
    _, r_wbv = ops.local_level_bwd(1, d_c1, 1, 1.0 / M, pt, w_bv, w_t, aux1, B, Nt, M, Nv, want_dx=False)
    ops.local_level_bwd(1, d_c0, 2, 1.0 / M, pbt, w_v, w_bt, aux2, M, Nt, B, Nv, d_x=r_vn, d_w=r_wv, accumulate=True, use_lo=use_lo)
    _, r_wbt = ops.local_level_bwd(0, d_c0, 2, 1.0 / M, pv, w_bt, w_v, aux2, M, Nt, B, Nv, want_dx=False)
    # grouped
    f32 = dict(dtype=torch.float32, device=DEV)
    d_tn, d_vn = torch.full((B * Nt, d), float("nan"), **f32), torch.full((B * Nv, d), float("nan"), **f32)
    T_pv, T_pt, T_pbv, T_pbt = ops.transpose_prepared([pv, pt, pbv, pbt], use_lo=use_lo)
    ops.local_level_bwd_group([
        dict(side=0, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pv, w_self=w_t, w_other=w_v, aux=aux0, A=B, Nt=Nt, Bv=B, Nv=Nv, d_x=d_tn),
        dict(side=0, dS=d_c1, ds_mode=1, ds_scale=1.0 / M, other_T=T_pbv, w_self=w_t, w_other=w_bv, aux=aux1, A=B, Nt=Nt, Bv=M, Nv=Nv,
             d_x=d_tn),
        dict(side=1, dS=dS, ds_mode=0, ds_scale=1.0, other_T=T_pt, w_self=w_v, w_other=w_t, aux=aux0, A=B, Nt=Nt, Bv=B, Nv=Nv, d_x=d_vn),
        dict(side=1, dS=d_c0, ds_mode=2, ds_scale=1.0 / M, other_T=T_pbt, w_self=w_v, w_other=w_bt, aux=aux2, A=M, Nt=Nt, Bv=B, Nv=Nv,
             d_x=d_vn)], use_lo=use_lo)
    d_wt, d_wv = torch.full((B * Nt,), float("nan"), **f32), torch.full((B * Nv,), float("nan"), **f32)
    d_wbt, d_wbv = torch.full((M * Nt,), float("nan"), **f32), torch.full((M * Nv,), float("nan"), **f32)
    ops.pool_weight_bwd_group([
        dict(side=0, N=Nt, d_w=d_wt, srcs=[(dS, 0, 1.0, aux0[2], B, B), (d_c1, 1, 1.0 / M, aux1[2], B, M)]),
        dict(side=1, N=Nv, d_w=d_wv, srcs=[(dS, 0, 1.0, aux0[3], B, B), (d_c0, 2, 1.0 / M, aux2[3], M, B)]),
        dict(side=1, N=Nv, d_w=d_wbv, srcs=[(d_c1, 1, 1.0 / M, aux1[3], B, M)]),
        dict(side=0, N=Nt, d_w=d_wbt, srcs=[(d_c0, 2, 1.0 / M, aux2[2], M, B)])])
    for got, ref in ((d_tn, r_tn), (d_vn, r_vn), (d_wt, r_wt), (d_wv, r_wv), (d_wbt, r_wbt), (d_wbv, r_wbv)):
        assert maxdiff(got, ref) < 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("B,d", [(128, 512), (37, 96)])
def test_the_three_launches_of_the_backward_head(B, d):
    """nr_rowloss_bwd_finish, nr_centrality_weights_bwd_pair and nr_global_logits_bwd against their plain torch forms (and the
    one-modality centrality kernel they replace)."""
    g = torch.Generator().manual_seed(B + d)
    r = lambda *sh: torch.randn(*sh, generator=g).to(DEV)       # noqa: E731
    dS_dir, dG_dir, dC = r(2, B, B), r(2, B, B), r(2, B, B)
    dls = r(2, B)
    dS, dG, d_c0, d_c1, d_ls = ops.rowloss_bwd_finish(dS_dir, dG_dir, dC, dls)
    assert torch.equal(dS, dS_dir[0] + dS_dir[1].T) and torch.equal(dG, dG_dir[0] + dG_dir[1].T)
    assert maxdiff(d_c0, dC[0].double().sum(0).float()) < 1e-5 * B ** 0.5 and maxdiff(d_c1, dC[1].double().sum(0).float()) < 1e-5 * B ** 0.5
    assert abs(float(d_ls) - float(dls.double().sum())) < 1e-5 * B ** 0.5
    gt, gv = r(B, d), r(B, d)
    gn_t, gn_v = gt.norm(dim=-1), gv.norm(dim=-1)
    mean_t, mean_v = r(d) * 0.1, r(d) * 0.1
    w_t, w_v = torch.rand(B, generator=g).to(DEV) + 0.5, torch.rand(B, generator=g).to(DEV) + 0.5
    dw_t, dw_v = r(B), r(B)
    dg_t, dm_t, dg_v, dm_v = ops.centrality_weights_bwd_pair(gt, gn_t, mean_t, w_t, dw_t, gv, gn_v, mean_v, w_v, dw_v, 0.3)
    for (gg, gn, mean, w, dw, dg, dm) in ((gt, gn_t, mean_t, w_t, dw_t, dg_t, dm_t), (gv, gn_v, mean_v, w_v, dw_v, dg_v, dm_v)):
        rdg, rdm = ops.centrality_weights_bwd(gg, gn, mean, w, dw, 0.3)
        assert torch.equal(dg, rdg) and torch.equal(dm, rdm)
    d_gt, d_gv = ops.global_logits_bwd(dG, gt, gv, dg_t, dg_v)
    ref_t = (dG.double() @ gv.double() + dg_t.double()).float()
    ref_v = (dG.double().T @ gt.double() + dg_v.double()).float()
    assert maxdiff(d_gt, ref_t) < 2e-6 * float(ref_t.abs().max()) * B ** 0.5
    assert maxdiff(d_gv, ref_v) < 2e-6 * float(ref_v.abs().max()) * B ** 0.5


def _bf(t):
    """bf16 round-to-nearest-even of an fp32 tensor, as int16 bit patterns (what the split kernels store)."""
    return t.to(torch.bfloat16).view(torch.int16)


@pytest.mark.parametrize("rows,cols,ld", [(200, 96, 256), (64, 64, 64), (130, 70, 132)])
def test_split_group_modes(rows, cols, ld):
    """nr_split_group, every mode, against torch: 0 row-major split, 1 transposed split (zero K padding), 2 bf16-pair transpose,
    3 transposed k=3 token neighbourhood, 4 / 5 / 6 the matrix forms of a convolution kernel."""
    from neighborretr_amd.cluster_fused import split_group
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g).to(DEV)
    i16 = dict(dtype=torch.int16, device=DEV)
    hi_ref = x.to(torch.bfloat16)
    lo_ref = (x - hi_ref.float()).to(torch.bfloat16)
    # mode 1 and mode 2
    t_hi, t_lo = torch.full((cols, ld), 7, **i16), torch.full((cols, ld), 7, **i16)
    p_hi, p_lo = torch.full((cols, ld), 7, **i16), torch.full((cols, ld), 7, **i16)
    split_group([(x, None, t_hi, t_lo, rows, cols, 1, ld),
                 (hi_ref.view(torch.int16).contiguous(), lo_ref.view(torch.int16).contiguous(), p_hi, p_lo, rows, cols, 2, ld)])
    pad = min(ld, (rows + 63) // 64 * 64)
    for got_hi, got_lo in ((t_hi, t_lo), (p_hi, p_lo)):
        assert torch.equal(got_hi[:, :rows], hi_ref.view(torch.int16).T) and torch.equal(got_lo[:, :rows], lo_ref.view(torch.int16).T)
        assert int(got_hi[:, rows:pad].abs().max() if pad > rows else 0) == 0          # the K padding is written as zeros
    # mode 3: samples of `n` tokens
    n = 10 if rows % 10 == 0 else (13 if rows % 13 == 0 else 8)
    n_rows = rows // n * n
    xs = x[:n_rows].contiguous()
    h3, l3 = torch.full((3 * cols, ld), 7, **i16), torch.full((3 * cols, ld), 7, **i16)
    split_group([(xs, None, h3, l3, n_rows, cols, 3, ld, n)])
    xv = xs.view(-1, n, cols)
    z = torch.zeros_like(xv[:, :1])
    shifted = torch.stack([torch.cat([z, xv[:, :-1]], 1), xv, torch.cat([xv[:, 1:], z], 1)], -1).reshape(n_rows, 3 * cols)   # [r, 3c + s]
    assert torch.equal(h3[:, :n_rows], _bf(shifted).T)
    assert torch.equal(l3[:, :n_rows], _bf(shifted - shifted.to(torch.bfloat16).float()).T)
    # modes 4 / 5: W [C_out, C_in, 3]
    Co, Ci = 48, 40
    w = torch.randn(Co, Ci, 3, generator=g).to(DEV)
    a_hi, a_lo = torch.empty((Co, 3 * Ci), **i16), torch.empty((Co, 3 * Ci), **i16)
    b_hi, b_lo = torch.empty((Ci, 3 * Co), **i16), torch.empty((Ci, 3 * Co), **i16)
    c_hi, c_lo = torch.empty((Ci, 3 * Co), **i16), torch.empty((Ci, 3 * Co), **i16)
    split_group([(w, None, a_hi, a_lo, Co, 3 * Ci, 4, 3 * Ci, Ci), (w, None, b_hi, b_lo, Ci, 3 * Co, 5, 3 * Co, Co),
                 (w, None, c_hi, c_lo, Ci, 3 * Co, 6, 3 * Co, Co)])
    wa = w.permute(0, 2, 1).reshape(Co, 3 * Ci)
    wb = w.permute(1, 2, 0).reshape(Ci, 3 * Co)
    wc = w.flip(-1).permute(1, 2, 0).reshape(Ci, 3 * Co)          # mode 6: taps reversed
    assert torch.equal(a_hi, _bf(wa)) and torch.equal(b_hi, _bf(wb)) and torch.equal(c_hi, _bf(wc))
    assert torch.equal(a_lo, _bf(wa - wa.to(torch.bfloat16).float())) and torch.equal(b_lo, _bf(wb - wb.to(torch.bfloat16).float()))


@pytest.mark.parametrize("n_tok,t0", [(4200, 0), (6144, 64)])
def test_scorer_backward_hidden_kernel_block_shapes_agree(n_tok, t0):
    """nr_token_mlp_bwd_hidden on a large one-pass token set: the 192 x 256 block that writes only the hi halves of dh^T
    (what the memory-bank sets of a training step take) against the 128 x 128 block with every output: identical dh^T (hi),
    zero K padding, the same partial sums to summation order."""
    g = torch.Generator().manual_seed(n_tok)
    d, H = 512, 1024
    x = torch.randn(n_tok // 12, 12, d, generator=g).to(DEV)
    prep = ops.prepare_tokens(x, torch.ones(n_tok // 12, 12, device=DEV), want_lo=False)
    n = prep.n_tok
    w1 = (torch.randn(H, d, generator=g) * 0.03).to(DEV)
    b1, w2 = (torch.randn(H, generator=g) * 0.1).to(DEV), torch.randn(H, generator=g).to(DEV)
    w1_hi = w1.to(torch.bfloat16).view(torch.int16).contiguous()
    dl = (torch.randn(n, generator=g) * 1e-2).to(DEV)
    ldT = t0 + (n + 63) // 64 * 64 + 64
    i16 = dict(dtype=torch.int16, device=DEV)

    def run(hi_only):
        rows = int(hip.lib().nr_token_mlp_bwd_part_rows(n, H, hip.PREC_BF16, int(hi_only)))
        dhT_hi = torch.full((H, ldT), 7, **i16)
        dhT_lo = None if hi_only else torch.full((H, ldT), 7, **i16)
        p2, p1, pl = (torch.full((rows, k), float("nan"), device=DEV) for k in (H, H, 1))
        hip.call("nr_token_mlp_bwd_hidden", hip.ptr(prep.hi), None, hip.ptr(prep.norm), n, d, hip.ptr(w1_hi), None, hip.ptr(b1),
                 hip.ptr(w2), H, hip.PREC_BF16, hip.ptr(dl), hip.ptr(dhT_hi), hip.ptr(dhT_lo, allow_none=True), ldT, t0, None, None,
                 hip.ptr(p2), hip.ptr(p1), hip.ptr(pl), hip.stream_ptr())
        return dhT_hi, rows, p2.sum(0), p1.sum(0), pl.sum()
    big, rows_big, a2, a1, al = run(True)
    small, rows_small, b2_, b1_, bl = run(False)
    assert rows_big == 2 * ((n + 191) // 192) and rows_small == 2 * ((n + 127) // 128)
    pad = (n + 63) // 64 * 64
    assert torch.equal(big[:, t0:t0 + pad], small[:, t0:t0 + pad])
    assert int(big[:, t0 + n:t0 + pad].abs().max()) == 0 if pad > n else True        # K padding written as zeros
    assert int((big[:, :t0] != 7).sum()) == 0 and int((big[:, t0 + pad:] != 7).sum()) == 0      # nothing outside the set's columns
    for x_, y_ in ((a2, b2_), (a1, b1_), (al, bl)):
        assert maxdiff(x_, y_) < 1e-5 * max(float(y_.abs().max()), 1e-6)


def test_pack_and_unpack_of_the_exchange_step():
    """nr_pack_shard / nr_unpack_gathered: two ranks' shards packed, concatenated as the all-gather would, unpacked
    rank-major with the u8 masks turned into fp32 multipliers."""
    g = torch.Generator().manual_seed(5)
    W, b, Nt, Nv, d = 2, 5, 24, 12, 512
    shards = []
    for r in range(W):
        shards.append([torch.randn(b, Nt, d, generator=g).to(DEV), torch.randn(b, Nv, d, generator=g).to(DEV),
                       torch.randint(0, 10 ** 6, (b,), generator=g).to(DEV),
                       (torch.rand(b, Nt, generator=g) > 0.3).to(torch.uint8).to(DEV),
                       (torch.rand(b, Nv, generator=g) > 0.3).to(torch.uint8).to(DEV)])
    sizes = [t.numel() * t.element_size() for t in shards[0]]
    offs = [sum(sizes[:k]) for k in range(5)]
    total = (sum(sizes) + 15) // 16 * 16
    recv = torch.zeros(W * total, dtype=torch.uint8, device=DEV)
    for r in range(W):
        ops.pack_shard(shards[r], recv[r * total:(r + 1) * total], offs)
    outs = [torch.empty(W * b, Nt, d, device=DEV), torch.empty(W * b, Nv, d, device=DEV),
            torch.empty(W * b, dtype=torch.int64, device=DEV), torch.empty(W * b, Nt, device=DEV), torch.empty(W * b, Nv, device=DEV)]
    ops.unpack_gathered(recv, W, total, sizes, offs, outs, [False, False, False, True, True])
    for k in range(5):
        ref = torch.cat([shards[r][k] for r in range(W)], 0)
        assert torch.equal(outs[k], ref.float() if k >= 3 else ref)
    # nr_pack_shard_convert: int64 / fp32 masks (what the loaders / the model hand over) become the record's u8 inside the pack
    # launch -- the same record, byte for byte, as packing masks converted with .to(uint8) beforehand
    for mask_dtype in (torch.int64, torch.float32):
        again = torch.zeros_like(recv)
        for r in range(W):
            pieces = shards[r][:3] + [m.to(mask_dtype) for m in shards[r][3:]]
            kinds = [0, 0, 0] + [ops.mask_piece(m)[1] for m in pieces[3:]]
            assert kinds[3:] == ([1, 1] if mask_dtype == torch.int64 else [2, 2])
            ops.pack_shard(pieces, again[r * total:(r + 1) * total], offs, kinds)
        assert torch.equal(again, recv)


@pytest.mark.parametrize("B,M,W", [(128, 512, 8), (32, 64, 2), (24, 40, 3)])
def test_bank_absorb_gathered_equals_unpack_prepare_and_ring_push(B, M, W):
    """nr_bank_absorb_gathered (a step of the step-interleaved job that this rank does not own): from the receive buffer of the
    packed exchange, in ONE launch, what the eager path does in four -- unpack, prepare the batch's tokens, move the ring head,
    push fp32 rows / masks / ids and the prepared rows -- bit for bit; the noise counter advances by one step; the launch's
    two-level ticket (one word per group of 32 workgroups + its own) is all zero again afterwards.  Three launches in a row
    (the head wraps at M = 40)."""
    from types import SimpleNamespace
    from neighborretr_amd import comm, synth
    from neighborretr_amd.dist import packed_gather_raw, unpack_raw
    Nt, Nv, b = 24, 12, B // W
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(31, B, Nt, Nv, M).items()}
    world = comm.EmulatedWorld(W, real_collectives=False)
    cfg = SimpleNamespace(world_size=W)
    recv = lay = None
    for _sweep in range(2):                                       # settle the emulated peers' parts, then read rank 0's buffer
        for r in range(W):
            sl = slice(r * b, (r + 1) * b)
            c = world.comm(r)
            with comm.use(c):
                c.begin_step()
                rv, ly = packed_gather_raw(p["text_feat"][sl].contiguous(), p["video_feat"][sl].contiguous(), p["idx"][sl].contiguous(),
                                           p["text_mask"][sl].contiguous(), p["video_mask"][sl].contiguous(), cfg)
            if r == 0:
                recv, lay = rv, ly
    tf, vf, ix, tm, vm = unpack_raw(recv, lay)
    assert torch.equal(tf, p["text_feat"]) and torch.equal(ix, p["idx"]) and torch.equal(vm, p["video_mask"].float())

    def bank():
        d = {"mb_feat_t": p["mb_feat_t"].clone(), "mb_feat_v": p["mb_feat_v"].clone(), "mb_mask_t": p["mb_mask_t"].float().clone(),
             "mb_mask_v": p["mb_mask_v"].float().clone(), "mb_ind": torch.arange(7000, 7000 + M, device=DEV)}
        sh = (ops.prepare_tokens(d["mb_feat_t"], d["mb_mask_t"], want_lo=True), ops.prepare_tokens(d["mb_feat_v"], d["mb_mask_v"], want_lo=True))
        return d, sh, torch.tensor([3], dtype=torch.int32, device=DEV), torch.tensor([5, 40], dtype=torch.int64, device=DEV)
    got, sh_g, head_g, rng_g = bank()
    ref, sh_r, head_r, rng_r = bank()
    pt, pv = ops.prepare_tokens_pair(tf, tm, vf, vm, want_lo=True)
    names = list(ref)
    for launch in range(3):
        ops.bank_absorb_gathered(recv, lay, got, sh_g, head_g, M, rng_g)
        head_r.sub_(B).remainder_(M)
        banks = [ref[k] for k in names] + [t_ for q, N in ((sh_r[0], Nt), (sh_r[1], Nv)) for t_ in (q.hi.view(M, -1), q.lo.view(M, -1), q.norm.view(M, N))]
        rows = [dict(mb_feat_t=tf, mb_feat_v=vf, mb_mask_t=tm, mb_mask_v=vm, mb_ind=ix)[k] for k in names]
        rows += [t_ for q, N in ((pt, Nt), (pv, Nv)) for t_ in (q.hi.view(B, -1), q.lo.view(B, -1), q.norm.view(B, N))]
        ops.bank_ring_push(banks, rows, 0, head_dev=head_r)
        torch.cuda.synchronize()
        assert int(head_g) == int(head_r), launch
        for k in names:
            assert torch.equal(got[k], ref[k]), (launch, k)
        for a, c in zip(sh_g, sh_r):
            assert torch.equal(a.hi, c.hi) and torch.equal(a.lo, c.lo) and torch.equal(a.norm, c.norm), launch
        assert rng_g.tolist() == [5, 41 + launch]
        assert int((ops._COUNTERS[("absorb", torch.device(DEV, 0))] != 0).sum()) == 0


@pytest.mark.parametrize("B,M,W", [(64, 64, 2), (128, 48, 4), (1024, 512, 8)])
def test_bank_absorb_gathered_of_a_batch_as_large_as_the_bank(B, M, W):
    """world * per_rank >= capacity: the batch's first `capacity` samples become the bank, oldest-first order irrelevant, head 0
    (the reference's cat(batch, bank)[:capacity], modeling.py:244-249) -- fp32 rows, masks, ids and the prepared shadow rows bit for
    bit what unpack + prepare + a plain copy leave; the noise counter advances; the ticket words are zero again."""
    from types import SimpleNamespace
    from neighborretr_amd import comm, synth
    from neighborretr_amd.dist import packed_gather_raw, unpack_raw
    Nt, Nv, b = 24, 12, B // W
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(37, B, Nt, Nv, M).items()}
    world = comm.EmulatedWorld(W, real_collectives=False)
    cfg = SimpleNamespace(world_size=W)
    recv = lay = None
    for _sweep in range(2):
        for r in range(W):
            sl = slice(r * b, (r + 1) * b)
            c = world.comm(r)
            with comm.use(c):
                c.begin_step()
                rv, ly = packed_gather_raw(p["text_feat"][sl].contiguous(), p["video_feat"][sl].contiguous(), p["idx"][sl].contiguous(),
                                           p["text_mask"][sl].contiguous(), p["video_mask"][sl].contiguous(), cfg)
            if r == 0:
                recv, lay = rv, ly
    tf, vf, ix, tm, vm = unpack_raw(recv, lay)
    got = {"mb_feat_t": p["mb_feat_t"].clone(), "mb_feat_v": p["mb_feat_v"].clone(), "mb_mask_t": p["mb_mask_t"].float().clone(),
           "mb_mask_v": p["mb_mask_v"].float().clone(), "mb_ind": torch.arange(7000, 7000 + M, device=DEV)}
    sh = (ops.prepare_tokens(got["mb_feat_t"], got["mb_mask_t"], want_lo=True), ops.prepare_tokens(got["mb_feat_v"], got["mb_mask_v"], want_lo=True))
    head = torch.tensor([min(5, M - 1)], dtype=torch.int32, device=DEV)
    rng = torch.tensor([5, 40], dtype=torch.int64, device=DEV)
    ops.bank_absorb_gathered(recv, lay, got, sh, head, M, rng)
    torch.cuda.synchronize()
    pt, pv = ops.prepare_tokens_pair(tf, tm, vf, vm, want_lo=True)
    assert int(head) == 0 and rng.tolist() == [5, 41]
    want = dict(mb_feat_t=tf, mb_feat_v=vf, mb_mask_t=tm, mb_mask_v=vm, mb_ind=ix)
    for k, v in got.items():
        assert torch.equal(v, want[k][:M]), k
    for a, c, N in ((sh[0], pt, Nt), (sh[1], pv, Nv)):
        assert torch.equal(a.hi.view(-1), c.hi.view(-1)[:M * N * a.d]) and torch.equal(a.lo.view(-1), c.lo.view(-1)[:M * N * a.d])
        assert torch.equal(a.norm.view(-1), c.norm.view(-1)[:M * N])
    assert int((ops._COUNTERS[("absorb", torch.device(DEV, 0))] != 0).sum()) == 0


def test_copy_group_copies_every_piece():
    """nr_copy_group: several tensors of different dtypes and sizes (16-byte multiples and odd byte counts) in one launch."""
    g = torch.Generator().manual_seed(9)
    srcs = [torch.randint(-30000, 30000, (512 * 24, 512), generator=g, dtype=torch.int16).to(DEV), torch.randn(6144, generator=g).to(DEV),
            torch.randn(7, 3, generator=g).to(DEV), torch.tensor([5, 77], dtype=torch.int64, device=DEV),
            torch.randint(0, 255, (13,), generator=g, dtype=torch.uint8).to(DEV)]
    dsts = [torch.zeros_like(t) for t in srcs]
    ops.copy_group(dsts, srcs)
    assert all(torch.equal(a, b) for a, b in zip(dsts, srcs))
    with pytest.raises(ValueError):
        ops.copy_group([torch.zeros(3, device=DEV)], [torch.zeros(4, device=DEV)])


@pytest.mark.parametrize("B", [16, 128])
def test_split_tail_matches_the_single_launch(B):
    """Uniform-CE row terms from the Sinkhorn kernel + the other terms from nr_row_losses_fwd_no_uniform == the
    one-launch path (targets materialised, nr_row_losses_fwd_final)."""
    g = torch.Generator().manual_seed(B + 3)
    S = (torch.rand(B, B, generator=g) * 0.1).to(DEV)
    G = (torch.randn(B, B, generator=g) * 6 + torch.eye(B) * 5).to(DEV)
    v = lambda: (torch.rand(B, generator=g) * 0.1).to(DEV)
    c0, c1, w0, w1 = v(), v(), 1 + v(), 1 + v()
    ls = torch.tensor([100.0], device=DEV)
    K = min(20, B)
    tr, tc = ops.sinkhorn_targets(G, 0.7, 50)
    rl_ref, ref = ops.row_losses_final(S, G, tr, tc, c0, c1, w0, w1, ls, K, 3.0, 1.0, 1.0, 1.0)
    rl = torch.full((2, 4, B), float("nan"), device=DEV)
    assert ops.sinkhorn_uniform_rows(G, 0.7, 3.0, rl, 50)
    ops.row_losses_no_uniform(S, G, c0, c1, w0, w1, ls, K, 3.0, rl)
    got = ops.loss_finalize(rl, 1.0, 1.0, 1.0)
    assert torch.equal(rl[:, [0, 2, 3]], rl_ref[:, [0, 2, 3]])
    assert maxdiff(rl[:, 1], rl_ref[:, 1]) < 2e-5 * float(rl_ref[:, 1].abs().max())
    assert maxdiff(got, ref) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("B", [16, 128])
def test_split_tail_with_the_centrality_weights_inside_the_row_loss_launch(B):
    """nr_row_losses_fwd_no_uniform_final_cw (the loss-only step's final launch computing compute_centrality_weights,
    modeling.py:403-430, itself) == nr_centrality_weights_pair + nr_row_losses_fwd_no_uniform_final: the same five losses and
    the same row terms, bit for bit; the weights against the oracle."""
    g = torch.Generator().manual_seed(B + 11)
    d, K, T, cs = 512, min(20, B), 3.0, 0.3
    S = (torch.rand(B, B, generator=g) * 0.1).to(DEV)
    G = (torch.randn(B, B, generator=g) * 6 + torch.eye(B) * 5).to(DEV)
    parts = lambda n: (torch.rand(n, B, generator=g) * 0.1).to(DEV)
    c0p, c1p = parts(16), parts(5)
    gt, gv = torch.randn(B, 1, d, generator=g).to(DEV), torch.randn(B, 1, d, generator=g).to(DEV)
    mean_t, mean_v = (torch.randn(d, generator=g) * 0.05).to(DEV), (torch.randn(d, generator=g) * 0.05).to(DEV)
    ls = torch.tensor([100.0], device=DEV)
    out = []
    for fused in (False, True):
        rl = torch.full((2, 4, B), float("nan"), device=DEV)
        losses = torch.empty(5, device=DEV)
        counter = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.sinkhorn_uniform_rows_final(G, 0.7, T, rl, counter, 1.0, 0.7, 1.3, losses, 50)
        if fused:
            ops.row_losses_no_uniform_final_cw(S, G, c0p, c1p, 1.0 / 512, gt.view(B, d), gv.view(B, d), mean_t, mean_v, cs, ls, K, T, rl,
                                               counter, 1.0, 0.7, 1.3, losses)
        else:
            w_t, w_v, _ = ops.centrality_weights_pair(gt, gv, mean_t, mean_v, cs, False)
            ops.row_losses_no_uniform_final(S, G, c0p, c1p, 1.0 / 512, w_t, w_v, ls, K, T, rl, counter, 1.0, 0.7, 1.3, losses)
        torch.cuda.synchronize()
        assert int(counter) == 0
        out.append((rl.clone(), losses.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert torch.isfinite(out[1][1]).all()
    gn = torch.nn.functional.normalize(gt.view(B, d).cpu().double(), dim=-1)
    assert maxdiff(w_t, torch.exp(cs * (gn @ mean_t.cpu().double()))) < 1e-6


def test_bank_push_fifo():
    bank = torch.arange(10 * 6, dtype=torch.float32, device=DEV).reshape(10, 2, 3)
    ref = bank.clone()
    for n_new in (3, 10, 12, 1):
        batch = torch.randn(n_new, 2, 3, device=DEV)
        ref = torch.cat((batch, ref), 0)[:10]
        ops.bank_push(bank, batch)
        assert torch.equal(bank, ref)


def test_diag_ranks_match_reference_metrics():
    from util import golden
    from neighborretr_amd import synth
    n = 256
    S = (synth.normal(42, "metrics/S", (n, n)) * 0.1).astype(np.float32)
    S[np.arange(n), np.arange(n)] += 0.25
    for i in range(0, n, 16):
        S[i, (i + 3) % n] = S[i, i]
    for i in range(5, n, 16):
        S[i, (i + 7) % n] = np.nextafter(S[i, i], np.float32(10))
        S[i, (i + 9) % n] = np.nextafter(S[i, i], np.float32(-10))
    gr, eq = ops.diag_ranks(torch.from_numpy(S).to(DEV))
    gr, eq = gr.cpu().numpy(), eq.cpu().numpy()
    cols = np.concatenate([np.arange(g, g + e) for g, e in zip(gr, eq)])
    assert np.array_equal(cols, golden("metrics256")["cols"])


def test_grouped_scorer_launch_equals_the_four_single_launches():
    """nr_token_weights_fwd_group: the step's four token sets (batch text / video in split-bf16, bank video / text in one bf16
    pass) scored by ONE launch of 192 x 256 blocks.  One-pass sets: bit-identical to their single launch; split-bf16 sets (three
    accumulated passes instead of the split tile): equal to 2e-6; sets that do not fit the block (192 % N != 0): None, nothing
    launched."""
    from neighborretr_amd import head, synth
    B, Nt, Nv, M = 128, 24, 12, 512
    prob = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
    P = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_params(7).items()}
    sw = head.ScorerWeights(P["text_weight_fc.0.weight"], P["text_weight_fc.0.bias"], P["text_weight_fc.2.weight"], P["text_weight_fc.2.bias"])
    pt = ops.prepare_tokens(prob["text_feat"], prob["text_mask"])
    pv = ops.prepare_tokens(prob["video_feat"], prob["video_mask"])
    pbt = ops.prepare_tokens(prob["mb_feat_t"], prob["mb_mask_t"])
    pbv = ops.prepare_tokens(prob["mb_feat_v"], prob["mb_mask_v"])
    sets = [(pt, prob["text_mask"], B, Nt, hip.PREC_BF16X3), (pv, prob["video_mask"], B, Nv, hip.PREC_BF16X3),
            (pbv, prob["mb_mask_v"], M, Nv, hip.PREC_BF16), (pbt, prob["mb_mask_t"], M, Nt, hip.PREC_BF16)]
    calls = [(p_, sw.w1_hi, sw.w1_lo, sw.b1, sw.w2, sw.b2, m_.float(), n_, N_) for p_, m_, n_, N_, _ in sets]
    precs = [s_[-1] for s_ in sets]
    one = [ops.token_weights(*c_, pr_)[0] for c_, pr_ in zip(calls, precs)]
    grp = ops.token_weights_group(calls, precs)
    assert grp is not None
    for k, (a, (b, _)) in enumerate(zip(one, grp)):
        assert maxdiff(a.sum(-1), torch.ones(a.shape[0])) < 1e-5
        if precs[k] == hip.PREC_BF16:
            assert torch.equal(a, b), k
        else:
            assert maxdiff(a, b) < 2e-6, (k, maxdiff(a, b))
    # 20 tokens per sample: no whole samples in a 192-row block -> the grouped form declines
    p20 = ops.prepare_tokens(prob["text_feat"][:, :20].contiguous(), prob["text_mask"][:, :20].contiguous())
    assert ops.token_weights_group([(p20, sw.w1_hi, sw.w1_lo, sw.b1, sw.w2, sw.b2, None, B, 20)], [hip.PREC_BF16]) is None
