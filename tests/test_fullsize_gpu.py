"""GPU: BASELINE.json's full sizes (configs[2] global B=1024, configs[3] ActivityNet token counts with
M=1024), checked through size-independent properties -- the oracle is too slow there -- plus direct
oracle comparison for the pieces that are cheap at any size (row losses, Sinkhorn, ranks).

Properties:
  * permuting the videos permutes the columns of S exactly (bit for bit: per-pair arithmetic is
    independent of the tile it lands in); same for texts / rows;
  * a masked token contributes exactly 0: changing its features does not change S at all;
  * row / column partial sums (memory-bank modes) equal the sums of the full matrix;
  * a fully masked video gives an exactly-zero column;
  * Sinkhorn plans have unit column sums; targets are beta*Q + (1-beta)*I;
  * sum of the neighbour-loss positive weights = 2 is implied by loss equality with the oracle.
"""
import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import hip, ops, synth
from util import maxdiff

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tokens(seed, n, N, d=512):
    t = synth.normal(seed, f"fs/{n}/{N}", (n, N, d)).astype(np.float32)
    ln = synth.randint(seed, f"fs/len/{n}/{N}", 1, N, (n,))
    m = (np.arange(N)[None, :] < ln[:, None]).astype(np.float32)
    w = synth.uniform(seed, f"fs/w/{n}/{N}", (n, N)).astype(np.float32) * m + 1e-3
    w = w / w.sum(1, keepdims=True)
    return torch.from_numpy(t).to(DEV), torch.from_numpy(m).to(DEV), torch.from_numpy(w).to(DEV)


@pytest.mark.parametrize("A,Nt,Bv,Nv,prec", [
    (1024, 24, 1024, 12, hip.PREC_BF16),      # configs[2]: global B = 1024, batch x batch
    (1024, 24, 512, 12, hip.PREC_BF16X3),     # configs[2]: batch x bank
    (128, 64, 1024, 64, hip.PREC_BF16),       # configs[3]: ActivityNet token counts, batch x bank (M=1024)
    (1024, 64, 128, 64, hip.PREC_BF16X3),     # configs[3]: bank x batch
    (130, 24, 520, 12, hip.PREC_BF16),        # 192 x 384 blocks with ragged edges (130 = 16*8+2 texts, 520 = 16*32+8 videos)
    (1000, 24, 1000, 12, hip.PREC_BF16X3),    # MSR-VTT 1k-A evaluation size: 192 x 192 split-bf16 blocks, ragged edges
    (250, 24, 90, 12, hip.PREC_BF16X3),       # 96 x 192 split-bf16 blocks, ragged edges
])
def test_local_level_properties_at_full_size(A, Nt, Bv, Nv, prec):
    t, tm, wt = _tokens(11, A, Nt)
    v, vm, wv = _tokens(12, Bv, Nv)
    vm[3] = 0                                               # fully masked video
    pt, pv = ops.prepare_tokens(t, tm), ops.prepare_tokens(v, vm)
    S, _ = ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec)
    assert torch.isfinite(S).all() and float(S.abs().max()) <= 1.0 + 1e-3
    assert float(S[:, 3].abs().max()) == 0.0
    # column / row permutations
    perm_v = torch.from_numpy(np.random.RandomState(0).permutation(Bv)).to(DEV)
    pv2 = ops.prepare_tokens(v[perm_v].contiguous(), vm[perm_v].contiguous())
    S2, _ = ops.local_level(pt, pv2, wt, wv[perm_v].contiguous(), A, Nt, Bv, Nv, prec)
    assert torch.equal(S2, S[:, perm_v])
    perm_t = torch.from_numpy(np.random.RandomState(1).permutation(A)).to(DEV)
    pt2 = ops.prepare_tokens(t[perm_t].contiguous(), tm[perm_t].contiguous())
    S3, _ = ops.local_level(pt2, pv, wt[perm_t].contiguous(), wv, A, Nt, Bv, Nv, prec)
    assert torch.equal(S3, S[perm_t])
    # masked tokens contribute exactly nothing
    t4 = t.clone()
    t4[tm == 0] = 123.0
    S4, _ = ops.local_level(ops.prepare_tokens(t4, tm), pv, wt, wv, A, Nt, Bv, Nv, prec)
    assert torch.equal(S4, S)
    # memory-bank modes
    rs, _ = ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec, hip.OUT_ROWSUM)
    cs, _ = ops.local_level(pt, pv, wt, wv, A, Nt, Bv, Nv, prec, hip.OUT_COLSUM)
    assert maxdiff(ops.reduce_parts(rs, 1.0 / Bv), S.double().mean(1)) < 1e-6
    assert maxdiff(ops.reduce_parts(cs, 1.0 / A), S.double().mean(0)) < 1e-6
    # spot check of 6 x 5 pairs against the oracle arithmetic in fp64
    ia, ib = torch.arange(0, A, max(A // 6, 1))[:6], torch.arange(1, Bv, max(Bv // 5, 1))[:5]
    ia = torch.cat((ia, torch.tensor([A - 1])))              # the ragged last block too
    ib = torch.cat((ib, torch.tensor([Bv - 1])))
    tn = torch.nn.functional.normalize(t[ia].cpu().double(), dim=-1) * tm[ia].cpu().double()[..., None]
    vn = torch.nn.functional.normalize(v[ib].cpu().double(), dim=-1) * vm[ib].cpu().double()[..., None]
    R = torch.einsum("atd,bvd->abtv", tn, vn)
    ref = 0.5 * ((R.max(-1)[0] * wt[ia].cpu().double()[:, None]).sum(-1) + (R.max(-2)[0] * wv[ib].cpu().double()[None]).sum(-1))
    assert maxdiff(S[ia][:, ib], ref) < (2e-6 if prec == hip.PREC_BF16X3 else 1e-3)


@pytest.mark.parametrize("A,Bv", [(64, 64), (64, 70), (128, 1024)])
def test_256x256_blocks_of_64_token_products_equal_the_small_blocks(A, Bv):
    """64 x 64-token products (ActivityNet shape) in one bf16 pass run 256 x 256 blocks -- 4 texts x 4 videos, two texts per wave
    strip (nr_sim_reg_kernel<8, 4, 16, 16, ..., RT = 2>) -- once those fill the chip.  Per-pair arithmetic does not depend on
    the block a pair lands in: a few texts taken alone (2 texts -> the 64 x 64-per-wave blocks of nr_sim_reg / nr_sim) give the
    same rows bit for bit, pooled maxima and arg-max indices of the training form included; ragged column blocks (70 videos)."""
    N = 64
    t, tm, wt = _tokens(41, A, N)
    v, vm, wv = _tokens(42, Bv, N)
    pt, pv = ops.prepare_tokens(t, tm), ops.prepare_tokens(v, vm)
    assert hip.local_level_tiles(A, N, Bv, N, hip.PREC_BF16) == (A // 4, (Bv + 3) // 4)             # 4 texts x 4 videos per block
    S, _ = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, hip.PREC_BF16)
    Sa, aux = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, hip.PREC_BF16, want_arg=True)
    assert torch.equal(S, Sa) and torch.isfinite(S).all()
    for a0 in (0, 1, A // 2 + 1, A - 2):
        sub = ops.prepare_tokens(t[a0:a0 + 2].contiguous(), tm[a0:a0 + 2].contiguous())
        assert hip.local_level_tiles(2, N, Bv, N, hip.PREC_BF16)[0] == 1                              # another block shape
        S2, aux2 = ops.local_level(sub, pv, wt[a0:a0 + 2].contiguous(), wv, 2, N, Bv, N, hip.PREC_BF16, want_arg=True)
        assert torch.equal(S2, S[a0:a0 + 2]), a0
        for big, small in zip(aux, aux2):
            assert torch.equal(small, big[a0:a0 + 2]), a0
    rs, _ = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, hip.PREC_BF16, hip.OUT_ROWSUM)
    cs, _ = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, hip.PREC_BF16, hip.OUT_COLSUM)
    assert maxdiff(ops.reduce_parts(rs, 1.0 / Bv), S.double().mean(1)) < 1e-6
    assert maxdiff(ops.reduce_parts(cs, 1.0 / A), S.double().mean(0)) < 1e-6


def test_sinkhorn_and_row_losses_at_b1024():
    B, K = 1024, 20
    g = torch.Generator().manual_seed(5)
    G = torch.randn(B, B, generator=g) * 6 + torch.eye(B) * 8
    S = torch.rand(B, B, generator=g) * 0.12
    tr, tc = ops.sinkhorn_targets(G.to(DEV), 0.7, 50)
    Q = (tr.cpu().double() - 0.3 * torch.eye(B, dtype=torch.float64)) / 0.7
    assert maxdiff(Q.sum(0), torch.ones(B, dtype=torch.float64)) < 1e-4          # unit column sums
    ref_r = O.sinkhorn_targets(G.double(), 0.7)
    assert maxdiff(tr, ref_r) < 2e-5 and maxdiff(tc, O.sinkhorn_targets(G.double().t(), 0.7)) < 2e-5
    c0, c1 = torch.rand(B, generator=g) * 0.1, torch.rand(B, generator=g) * 0.1
    w = torch.exp(torch.randn(B, generator=g) * 0.01)
    rl = ops.row_losses(S.to(DEV), G.to(DEV), tr, tc, c0.to(DEV), c1.to(DEV), w.to(DEV), w.to(DEV),
                        torch.tensor([100.0], device=DEV), K, 3.0)
    losses = ops.loss_finalize(rl, 1.0, 1.0, 1.0).cpu().double()
    d = lambda x: x.double()
    ref = [O.centrality_loss(d(S), d(w), d(w), 100.0),
           (-(torch.log_softmax(d(G) * 3, -1) * ref_r).sum(-1).mean()
            - (torch.log_softmax(d(G).t() * 3, -1) * O.sinkhorn_targets(d(G).t(), 0.7)).sum(-1).mean()) / 2,
           O.neighbor_loss(d(S), d(c1)[:, None].expand(B, 2), d(c0)[:, None].expand(B, 2), K, 3.0),
           O.kl_loss(d(G), d(S))]
    for k, r in enumerate(ref):
        assert abs(float(losses[k + 1]) - float(r)) < 2e-4 * max(1.0, abs(float(r))), (k, float(losses[k + 1]), float(r))


def test_full_head_forward_b1024_is_finite_and_consistent():
    """configs[2] shape end to end (B=1024, M=512): finite losses; the replicated loss is invariant to a
    joint permutation of the (text, video) pairs (every term is a mean over samples)."""
    from neighborretr_amd import modeling
    from util import params
    B, Nt, Nv, M, K = 1024, 24, 12, 512, 20
    prob = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(3003, B, Nt, Nv, M).items()}
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict(params(), strict=False)
    m = m.to(DEV).train()
    nz = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_noise(3003, B, Nt, Nv).items()}
    c = m.config

    def run(p, noise):
        with torch.no_grad():
            return torch.stack(m._compute_losses(p["text_feat"], p["video_feat"], p["text_mask"], p["video_mask"],
                                                 p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"],
                                                 c.centrality_scale, c.beta, K, c.temperature, torch.tensor(100.0, device=DEV),
                                                 noise=noise)).cpu()
    L = run(prob, nz)
    assert torch.isfinite(L).all() and float(L[0]) > 0
    perm = torch.from_numpy(np.random.RandomState(2).permutation(B)).to(DEV)
    p2 = dict(prob)
    for k in ("text_feat", "video_feat", "text_mask", "video_mask"):
        p2[k] = prob[k][perm].contiguous()
    nz2 = {k: v[perm].contiguous() for k, v in nz.items()}
    L2 = run(p2, nz2)
    assert maxdiff(L, L2) < 2e-3 * float(L.abs().max())


def test_diag_ranks_at_1000():
    """MSR-VTT 1k-A size: GPU rank counting == the reference's sort-based ranks (metrics.py:58-66)."""
    from neighborretr_amd.metrics import RetrievalMetrics
    n = 1000
    S = (synth.normal(9, "ranks", (n, n)) * 0.05).astype(np.float32)
    S[np.arange(n), np.arange(n)] += 0.08
    S[7, 9] = S[7, 7]                                      # an exact tie
    mine = RetrievalMetrics.compute_metrics(torch.from_numpy(S).to(DEV))
    ref = O.compute_metrics(S)
    assert mine["cols"] == ref["cols"]
    for k in ("R1", "R5", "R10", "R50", "MR", "MeanR"):
        assert mine[k] == ref[k]


@pytest.mark.parametrize("A,Bv", [(64, 64), (128, 128), (128, 70)])
def test_split_bf16_64_token_products_on_the_one_pass_blocks(A, Bv):
    """Split-bf16 products of 64 x 64-token samples (configs[3]'s batch x batch product, ActivityNet evaluation) run the ONE-PASS
    256 x 256 blocks as three accumulated passes over K -- Ah Bh, then Ah Bl, then Al Bh (NrGemmTile::run_pp3) -- once those
    fill the chip.  Same three products as the split tile adds up slice by slice, another summation order: against the small
    blocks (a few texts alone) within 1e-6, against fp64 within 2e-6; row / column means; training form's pooled maxima."""
    N = 64
    t, tm, wt = _tokens(51, A, N)
    v, vm, wv = _tokens(52, Bv, N)
    pt, pv = ops.prepare_tokens(t, tm), ops.prepare_tokens(v, vm)
    X3 = hip.PREC_BF16X3
    assert hip.local_level_tiles(A, N, Bv, N, X3) == (A // 4, (Bv + 3) // 4)
    S, _ = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, X3)
    Sa, aux = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, X3, want_arg=True)
    assert torch.equal(S, Sa) and torch.isfinite(S).all()
    for a0 in (0, A // 2 + 1, A - 2):
        sub = ops.prepare_tokens(t[a0:a0 + 2].contiguous(), tm[a0:a0 + 2].contiguous())
        assert hip.local_level_tiles(2, N, Bv, N, X3)[0] == 1                                         # the split tile's own blocks
        S2, aux2 = ops.local_level(sub, pv, wt[a0:a0 + 2].contiguous(), wv, 2, N, Bv, N, X3, want_arg=True)
        assert maxdiff(S2, S[a0:a0 + 2]) < 1e-6, a0
        assert maxdiff(aux2[2], aux[2][a0:a0 + 2]) < 1e-6 and maxdiff(aux2[3], aux[3][a0:a0 + 2]) < 1e-6       # pooled maxima
    rs, _ = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, X3, hip.OUT_ROWSUM)
    cs, _ = ops.local_level(pt, pv, wt, wv, A, N, Bv, N, X3, hip.OUT_COLSUM)
    assert maxdiff(ops.reduce_parts(rs, 1.0 / Bv), S.double().mean(1)) < 1e-6
    assert maxdiff(ops.reduce_parts(cs, 1.0 / A), S.double().mean(0)) < 1e-6
    ia, ib = torch.tensor([0, 1, A // 2, A - 1]), torch.tensor([0, Bv // 3, Bv - 1])
    tn = torch.nn.functional.normalize(t[ia].cpu().double(), dim=-1) * tm[ia].cpu().double()[..., None]
    vn = torch.nn.functional.normalize(v[ib].cpu().double(), dim=-1) * vm[ib].cpu().double()[..., None]
    R = torch.einsum("atd,bvd->abtv", tn, vn)
    ref = 0.5 * ((R.max(-1)[0] * wt[ia].cpu().double()[:, None]).sum(-1) + (R.max(-2)[0] * wv[ib].cpu().double()[None]).sum(-1))
    assert maxdiff(S[ia][:, ib], ref) < 2e-6


@pytest.mark.parametrize("B,M,all_groupable", [(128, 512, True), (130, 520, True), (256, 1024, False)])
def test_grouped_products_equal_the_single_launches(B, M, all_groupable):
    """nr_local_level_group: the step's three products (two bank products in bf16, batch x batch in split-bf16) in one grid
    give, bit for bit, what one nr_local_level_fwd per product gives -- ragged block edges included; a product a group
    cannot take (other token counts) sends the whole list down the single-launch path."""
    t, tm, wt = _tokens(21, B, 24)
    v, vm, wv = _tokens(22, B, 12)
    bt, btm, wbt = _tokens(23, M, 24)
    bv, bvm, wbv = _tokens(24, M, 12)
    pt, pv = ops.prepare_tokens(t, tm, want_lo=True), ops.prepare_tokens(v, vm, want_lo=True)
    pbt, pbv = ops.prepare_tokens(bt, btm), ops.prepare_tokens(bv, bvm)
    probs = [(pt, pbv, wt, wbv, B, 24, M, 12, hip.PREC_BF16, hip.OUT_ROWSUM),
             (pbt, pv, wbt, wv, M, 24, B, 12, hip.PREC_BF16, hip.OUT_COLSUM),
             (pt, pv, wt, wv, B, 24, B, 12, hip.PREC_BF16X3, hip.OUT_FULL)]
    # B = 256: the batch x batch product runs 192 x 192 blocks on its own and cannot join (the call then launches one by one)
    assert all(hip.local_level_group_kind(q[4], q[5], q[6], q[7], 512, q[8]) >= 0 for q in probs) == all_groupable
    ref = [ops.local_level(*q)[0] for q in probs]
    for sel in ((0, 1, 2), (0, 1), (1, 2), (2, 1, 0)):
        got = ops.local_level_group([probs[i] for i in sel])
        for i, g in zip(sel, got):
            assert torch.equal(g, ref[i]), (sel, i)
    # 64-token products cannot join a group: the call falls back to one launch each, same results
    t64, tm64, wt64 = _tokens(25, 8, 64)
    v64, vm64, wv64 = _tokens(26, 8, 64)
    odd = (ops.prepare_tokens(t64, tm64), ops.prepare_tokens(v64, vm64), wt64, wv64, 8, 64, 8, 64, hip.PREC_BF16, hip.OUT_FULL)
    assert hip.local_level_group_kind(8, 64, 8, 64, 512, hip.PREC_BF16) < 0
    got = ops.local_level_group([probs[0], odd])
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ops.local_level(*odd)[0])


def test_chained_tile_pairs_with_ragged_edges():
    """nr_sim_pair_kernel (two bank-shaped products through one K loop per workgroup) on grids whose last row / column blocks
    hang over the operands (520 = 65 x 8 texts, 16.25 x 32 videos): bit-identical to one launch per product, in all three
    output modes."""
    n = 520
    t, tm, wt = _tokens(31, n, 24)
    v, vm, wv = _tokens(32, n, 12)
    t2, tm2, wt2 = _tokens(33, n, 24)
    v2, vm2, wv2 = _tokens(34, n, 12)
    pt, pv, pt2, pv2 = (ops.prepare_tokens(a, b) for a, b in ((t, tm), (v, vm), (t2, tm2), (v2, vm2)))
    for m0, m1 in ((hip.OUT_ROWSUM, hip.OUT_COLSUM), (hip.OUT_FULL, hip.OUT_ROWSUM), (hip.OUT_COLSUM, hip.OUT_FULL)):
        probs = [(pt, pv2, wt, wv2, n, 24, n, 12, hip.PREC_BF16, m0), (pt2, pv, wt2, wv, n, 24, n, 12, hip.PREC_BF16, m1)]
        assert all(hip.local_level_group_kind(n, 24, n, 12, 512, hip.PREC_BF16) == 0 for _ in probs)
        got = ops.local_level_group(probs)
        for q, g in zip(probs, got):
            assert torch.equal(g, ops.local_level(*q)[0])


@pytest.mark.parametrize("B", [192, 256, 512, 960, 1024])
def test_cooperative_sinkhorn_matches_the_oracle(B):
    """128 < B <= 1024, B % 64 == 0: the one-launch cooperative Sinkhorn solve (B/32 workgroups per direction, scaling vectors
    exchanged through global memory, a counter barrier per half-iteration) against the oracle's log-domain iteration
    (until_module.py:235-266) in fp64, both directions; unit column sums; 1 and 50 iterations."""
    g = torch.Generator().manual_seed(B)
    G = torch.randn(B, B, generator=g) * 6 + torch.eye(B) * 8
    for iters in (1, 50):
        tr, tc = ops.sinkhorn_targets(G.to(DEV), 0.7, iters)
        assert torch.isfinite(tr).all() and torch.isfinite(tc).all()
        ref_r = O.sinkhorn_targets(G.double(), 0.7, iters)
        ref_c = O.sinkhorn_targets(G.double().t(), 0.7, iters)
        assert maxdiff(tr, ref_r) < 2e-5 and maxdiff(tc, ref_c) < 2e-5, (B, iters, maxdiff(tr, ref_r), maxdiff(tc, ref_c))
    Q = (tr.cpu().double() - 0.3 * torch.eye(B, dtype=torch.float64)) / 0.7
    assert maxdiff(Q.sum(0), torch.ones(B, dtype=torch.float64)) < 1e-4


@pytest.mark.parametrize("B", [192, 320, 1024])
def test_multilaunch_sinkhorn_fallback_matches_the_oracle_and_the_cooperative_form(B):
    """the multi-launch Sinkhorn -- what nr_sinkhorn_targets runs when nr_sinkhorn_cooperative_ok says the device
    cannot hold a direction's workgroups together, and for every B the cooperative form does not cover -- called on its own
    (nr_sinkhorn_targets_multilaunch): == the oracle (until_module.py:235-266) and == the cooperative launch."""
    g = torch.Generator().manual_seed(B + 1)
    G = (torch.randn(B, B, generator=g) * 6 + torch.eye(B) * 8).to(DEV)
    tr = torch.empty_like(G)
    tc = torch.empty_like(G)
    ws = torch.empty((hip.sinkhorn_workspace_bytes(B),), dtype=torch.uint8, device=DEV)
    hip.call("nr_sinkhorn_targets_multilaunch", hip.ptr(G), B, 0.7, 50, hip.ptr(tr), hip.ptr(tc), hip.ptr(ws), hip.stream_ptr())
    ref_r = O.sinkhorn_targets(G.cpu().double(), 0.7, 50)
    ref_c = O.sinkhorn_targets(G.cpu().double().t(), 0.7, 50)
    assert maxdiff(tr, ref_r) < 2e-5 and maxdiff(tc, ref_c) < 2e-5
    coop = bool(hip.lib().nr_sinkhorn_cooperative_ok(B))
    assert coop == (B % 64 == 0)               # a whole MI355X holds 32 workgroups per XCD: every covered size passes the gate
    tr2, tc2 = ops.sinkhorn_targets(G, 0.7, 50)
    assert maxdiff(tr2, tr) < 2e-5 and maxdiff(tc2, tc) < 2e-5
