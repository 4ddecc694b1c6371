"""GPU: the drop-in model API end to end -- forward losses and gradients against (a) the vectors
captured from the reference (tests/golden) and (b) the CPU oracle's autograd on the same inputs.

Tolerances (flat, absolute): split-bf16 ("bf16x3") path <= 2e-4 on every loss and 2e-3 relative on gradients;
training plan ("bf16") <= 1e-3 on losses / logits (BASELINE.json north_star); retrieval ranks identical on the
bf16x3 path.  The measured deviations are printed (pytest -s).
"""
import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import ops, modeling, synth
from util import golden, maxdiff, noise, params, problem

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(precision, K=20):
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K), precision=precision)
    missing, unexpected = m.load_state_dict(params(), strict=False)
    assert not unexpected
    m = m.to(DEV)
    with torch.no_grad():
        m.clip.logit_scale.fill_(float(np.log(100.0)))
    return m.train()


def _losses(m, x, nz, K):
    c = m.config
    return m._compute_losses(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], x["mb_feat_t"],
                             x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"], c.centrality_scale, c.beta, K,
                             c.temperature, m.clip.logit_scale.exp(), noise=nz)


@pytest.mark.parametrize("name", ["c1_b16", "c2_b128"])
@pytest.mark.parametrize("precision,tol", [("bf16x3", 2e-4), ("bf16", 1e-3)])
def test_forward_losses_match_reference(name, precision, tol):
    g = golden(name)
    B, Nt, Nv, M, K = (int(g[k]) for k in ("B", "Nt", "Nv", "M", "K"))
    x = problem(int(g["seed"]), B, Nt, Nv, M, device=DEV)
    nz = noise(int(g["seed"]), B, Nt, Nv, device=DEV)
    m = _model(precision, K)
    with torch.no_grad():
        losses = torch.stack(_losses(m, x, nz, K)).cpu().numpy()
        S, St = m.get_similarity_logits(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"])
    dL = np.abs(losses - g["losses"])
    dS = maxdiff(S, g["S"])
    print(f"\n[{name} {precision}] |dL| (total, centrality, uniform, neighbour, kl) = {dL.tolist()}  max|dS| = {dS:.2e}")
    assert dL.max() < tol, (losses, g["losses"])
    assert dS < (2e-6 if precision == "bf16x3" else 1e-3)
    assert torch.equal(St, S.T)
    if precision == "bf16x3":
        # identical retrieval ranks (north_star): same `cols` as the reference's metrics on its own S
        from neighborretr_amd.metrics import RetrievalMetrics
        mine = RetrievalMetrics.compute_metrics(S.cpu().numpy())
        ref = O.compute_metrics(g["S"])
        assert mine["cols"] == ref["cols"] and mine["R1"] == ref["R1"]


# Gradient bars per precision plan (relative to the reference's value; measured deviations are printed, pytest -s):
#   "bf16x3": every product split-bf16 (16 mantissa bits), the MFMA backward with the tokens' low halves: <= 2e-3 on the
#             gradient norms, <= 5e-3 of the largest element on slices / row sums / per-parameter norms;
#   "bf16"  : the SHIPPING training plan (modeling.NeighborRetr default).  Forward: bank products + bank scorer in one bf16
#             pass; backward: nr_sim_bwd_mfma rounds the routing coefficients (token weight x upstream gradient) and the other
#             operand's tokens to bf16 = 2^-9 relative per term, of mixed sign, accumulated in fp32.  Measured on the MI355X
#             against both fixtures (round 3): norms <= 1.8e-5, logit scale <= 7.1e-5, slices / row sums <= 1.6e-3, worst
#             per-parameter norm 5.9e-4 -- the split plan measures the same to within 2x except the text slice at c1_b16
#             (3.3e-4 vs 1.6e-3).  So the bars are the SAME for both plans.
GRAD_BARS = {"bf16x3": dict(norm=2e-3, elem=5e-3, param=5e-3), "bf16": dict(norm=2e-3, elem=5e-3, param=5e-3)}


@pytest.mark.parametrize("name", ["c1_b16", "c2_b128"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("fused", [False, True], ids=["traced-clustering", "fused-clustering"])
def test_backward_matches_reference(name, precision, fused):
    """Gradients of the total loss against the REFERENCE's own autograd (fixtures g_* / param_grad_norms captured from the
    imported reference, oracle/capture_golden.py) for every training form that ships: precision plan {bf16x3, bf16 (default)}
    x clustering {autograd-traced torch ops (captured steps), grouped HIP forward + hand-derived backward (eager / DDP steps)}."""
    g = golden(name)
    bar = GRAD_BARS[precision]
    B, Nt, Nv, M, K = (int(g[k]) for k in ("B", "Nt", "Nv", "M", "K"))
    x = problem(int(g["seed"]), B, Nt, Nv, M, device=DEV)
    nz = noise(int(g["seed"]), B, Nt, Nv, device=DEV)
    m = _model(precision, K)
    m.fused_training_clustering = fused
    x["text_feat"].requires_grad_(True)
    x["video_feat"].requires_grad_(True)
    losses = _losses(m, x, nz, K)
    dL = np.abs(torch.stack([l.detach() for l in losses]).cpu().numpy() - g["losses"])
    assert dL.max() < (2e-4 if precision == "bf16x3" else 1e-3), dL
    losses[0].backward()
    gt, gv = x["text_feat"].grad, x["video_feat"].grad
    rel = lambda mine, ref: abs(float(mine) - float(ref)) / max(abs(float(ref)), 1e-30)      # noqa: E731
    dev = dict(
        g_text_norm=rel(gt.norm(), g["g_text_norm"]), g_video_norm=rel(gv.norm(), g["g_video_norm"]),
        g_text_slice=maxdiff(gt[:2, :4, :64], g["g_text_slice"]) / float(np.abs(g["g_text_slice"]).max()),
        g_video_slice=maxdiff(gv[:2, :4, :64], g["g_video_slice"]) / float(np.abs(g["g_video_slice"]).max()),
        g_text_rowsum=maxdiff(gt.sum(-1), g["g_text_rowsum"]) / float(np.abs(g["g_text_rowsum"]).max()),
        g_video_rowsum=maxdiff(gv.sum(-1), g["g_video_rowsum"]) / float(np.abs(g["g_video_rowsum"]).max()),
        g_logit_scale=rel(float(m.clip.logit_scale.grad) / 100.0, g["g_logit_scale"]))       # d/d(exp(p)) = d/dp / exp(p)
    named = dict(m.named_parameters())
    worst = ("", 0.0)
    for n, ref in zip([str(s) for s in g["param_names"]], g["param_grad_norms"]):
        mine = 0.0 if named[n].grad is None else float(named[n].grad.norm())
        e = abs(mine - ref) / max(ref, 1e-3)
        if e > worst[1]:
            worst = (n, e)
    print(f"\n[{name} {precision} {'fused' if fused else 'traced'}] relative gradient deviations vs the reference: "
          + ", ".join(f"{k} {v:.2e}" for k, v in dev.items()) + f"; worst parameter-gradient norm {worst[0]} {worst[1]:.2e}")
    for k in ("g_text_norm", "g_video_norm"):
        assert dev[k] < bar["norm"], (k, dev[k])
    assert dev["g_logit_scale"] < bar["norm"] + 1e-7 / abs(float(g["g_logit_scale"])), dev["g_logit_scale"]
    for k in ("g_text_slice", "g_video_slice", "g_text_rowsum", "g_video_rowsum"):
        assert dev[k] < bar["elem"], (k, dev[k])
    assert worst[1] < bar["param"], worst


def test_split_head_nodes_and_the_clustering_stream_change_nothing_but_the_schedule():
    """The training step with the head as two autograd nodes and the clustering on its own stream (the shipped form: the two
    backward chains overlap) against the head as ONE node with everything on the step's stream: identical losses and
    gradients (same kernels, same order per chain).  Then the shipped form captured as one HIP graph: the replay reproduces
    the eager gradients bit for bit."""
    from neighborretr_amd import backward
    g = golden("c1_b16")
    B, Nt, Nv, M, K = (int(g[k]) for k in ("B", "Nt", "Nv", "M", "K"))
    x = problem(int(g["seed"]), B, Nt, Nv, M, device=DEV)
    nz = noise(int(g["seed"]), B, Nt, Nv, device=DEV)
    m = _model("bf16", K)
    tf, vf = x["text_feat"].requires_grad_(True), x["video_feat"].requires_grad_(True)

    def step():
        m.zero_grad(set_to_none=True)
        tf.grad = vf.grad = None
        losses = _losses(m, x, nz, K)
        losses[0].backward()
        return torch.stack([l.detach() for l in losses])

    def grads():
        out = {"text": tf.grad.clone(), "video": vf.grad.clone()}
        out.update({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
        return out
    try:
        backward.SPLIT_HEAD_NODES, m.cluster_side_stream = False, False
        l_one = step()
        g_one = grads()
    finally:
        backward.SPLIT_HEAD_NODES, m.cluster_side_stream = True, True
    l_two = step()
    g_two = grads()
    torch.cuda.synchronize()
    assert torch.equal(l_one, l_two)
    assert g_one.keys() == g_two.keys() and len(g_two) > 40
    for k in g_one:
        assert torch.equal(g_one[k], g_two[k]), k
    # captured
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), m.graph_capture_mode():
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    m.zero_grad(set_to_none=True)
    tf.grad = vf.grad = None
    gr = torch.cuda.CUDAGraph()
    with m.graph_capture_mode(), torch.cuda.graph(gr):
        losses = _losses(m, x, nz, K)
        losses[0].backward()
    for _ in range(2):
        gr.replay()
    torch.cuda.synchronize()
    g_rep = grads()
    assert torch.equal(torch.stack([l.detach() for l in losses]), l_two)
    for k in g_two:
        assert torch.equal(g_rep[k], g_two[k]), k


@pytest.mark.parametrize("B,Nt,Nv", [(16, 24, 12), (8, 64, 64), (6, 20, 9)])
def test_fused_clustering_matches_oracle_and_torch_path(B, Nt, Nv):
    """The no-grad fused kernels (cluster_fused.py), the autograd torch-op path (cluster.py) and the
    oracle give the same global tokens."""
    x = problem(77, B, Nt, Nv, 4)
    P = params()
    nz = noise(77, B, Nt, Nv)
    gt_o, gv_o = O.merge_global_features(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], P, nz)
    m = _model("bf16")
    xg = {k: v.to(DEV) for k, v in x.items()}
    nzg = {k: v.to(DEV) for k, v in nz.items()}
    with torch.no_grad():
        assert m._can_fuse_clustering(xg["text_feat"], (m.text_ctm0,))
        gt_f, gv_f = m.merge_global_features(xg["text_feat"], xg["video_feat"], xg["text_mask"], xg["video_mask"], nzg)
    m.fuse_clustering = False
    with torch.no_grad():
        gt_t, gv_t = m.merge_global_features(xg["text_feat"], xg["video_feat"], xg["text_mask"], xg["video_mask"], nzg)
    scale = float(gt_o.abs().max())
    # EVERY sample is compared, including those with fewer valid tokens than cluster centres (a video of 1-2 frames
    # and 3 centres takes its extra centres among zero-score padding tokens) and those with an exact density tie
    # between valid tokens (sample 5 of the B=16 case): the oracle states the tie rule the kernels implement --
    # dpc_knn(centre_ties="lowest_index"); where the reference's torch.topk differs is recorded by
    # tests/test_oracle_golden.py::test_centre_tie_rule_and_where_torch_topk_differs.
    for got, ref in ((gt_f, gt_o), (gv_f, gv_o), (gt_t, gt_o), (gv_t, gv_o)):
        assert got.shape == ref.shape
        assert maxdiff(got, ref) < 2e-5 * scale
    assert maxdiff(gt_f, gt_t) < 2e-5 * scale and maxdiff(gv_f, gv_t) < 2e-5 * scale


def test_local_level_backward_matches_oracle_autograd():
    A, Nt, Bv, Nv = 10, 24, 14, 12
    x = problem(31, max(A, Bv), Nt, Nv, 4)
    P = params()
    tf = x["text_feat"][:A].clone().requires_grad_(True)
    vf = x["video_feat"][:Bv].clone().requires_grad_(True)
    tm, vm = x["text_mask"][:A], x["video_mask"][:Bv]
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    S_ref, _ = O.local_level(tf, vf, tm, vm, Pg)
    W = torch.from_numpy(synth.normal(5, "dS", (A, Bv)).astype(np.float32))
    (S_ref * W).sum().backward()

    m = _model("bf16x3")
    tf2 = x["text_feat"][:A].to(DEV).requires_grad_(True)
    vf2 = x["video_feat"][:Bv].to(DEV).requires_grad_(True)
    S, _ = m.local_level(tf2, vf2, tm.to(DEV), vm.to(DEV))
    assert maxdiff(S, S_ref) < 2e-6
    (S * W.to(DEV)).sum().backward()
    assert maxdiff(tf2.grad, tf.grad) < 2e-3 * float(tf.grad.abs().max())
    assert maxdiff(vf2.grad, vf.grad) < 2e-3 * float(vf.grad.abs().max())
    # masked tokens never receive gradient through the similarity
    for n in ("text_weight_fc", "video_weight_fc"):
        for k in ("0.weight", "0.bias", "2.weight", "2.bias"):
            ref = Pg[f"{n}.{k}"].grad
            mine = dict(m.named_parameters())[f"{n}.{k}"].grad
            # the last bias shifts every logit of a sample equally: its true gradient is 0 (softmax
            # invariance), what is left on both sides is rounding noise
            atol = 1e-6 if k == "2.bias" else 1e-9
            assert maxdiff(mine, ref) < 3e-3 * float(ref.abs().max()) + atol, (n, k)


def test_until_module_classes_forward_and_backward():
    from neighborretr_amd.until_module import (CentralityWeightingLoss, KLDivergenceLoss, NeighborAdjustingLoss,
                                               UniformRegularizationLoss)
    B, K, M = 32, 8, 40
    gen = torch.Generator().manual_seed(9)
    S = (torch.rand(B, B, generator=gen) * 0.12)
    G = torch.randn(B, B, generator=gen) * 9
    bank = torch.rand(B, M, generator=gen) * 0.1
    w = torch.exp(torch.randn(B, generator=gen) * 0.05)
    cases = [
        (lambda S_, G_, b_, w_: CentralityWeightingLoss()(S_ * 100.0, w_), lambda S_, G_, b_, w_: O.centrality_weighting_loss(S_ * 100.0, w_)),
        (lambda S_, G_, b_, w_: NeighborAdjustingLoss()(S_, b_, K, 3.0), lambda S_, G_, b_, w_: O.neighbor_adjusting_loss(S_, b_, K, 3.0)),
        (lambda S_, G_, b_, w_: UniformRegularizationLoss()(G_, 3.0, 0.7), lambda S_, G_, b_, w_: O.uniform_regularization_loss(G_, 3.0, 0.7)),
        (lambda S_, G_, b_, w_: KLDivergenceLoss()(G_, S_), lambda S_, G_, b_, w_: O.kl_divergence_loss(G_, S_)),
    ]
    for mine_fn, ref_fn in cases:
        cpu = [t.clone().double().requires_grad_(True) for t in (S, G, bank, w)]
        gpu = [t.clone().to(DEV).requires_grad_(True) for t in (S, G, bank, w)]
        ref = ref_fn(*cpu)
        mine = mine_fn(*gpu)
        assert abs(float(mine.detach()) - float(ref.detach())) < 1e-4 * max(1.0, abs(float(ref.detach())))
        ref.backward()
        mine.backward()
        for a, b in zip(gpu, cpu):
            if b.grad is None:
                assert a.grad is None or float(a.grad.abs().max()) == 0.0
            else:
                assert maxdiff(a.grad, b.grad) < 2e-3 * float(b.grad.abs().max()) + 1e-8
    with pytest.raises(IndexError):
        NeighborAdjustingLoss()(S.to(DEV), bank.to(DEV), B + 1, 3.0)


def test_graph_replays_keep_pushing_the_bank_forward():
    """The ring head lives on the device (moved by the step prologue): every replay of a captured step pushes its
    batch to a NEW place, and the FIFO read back afterwards is cat(newest ... oldest, old bank)[:M]."""
    B, Nt, Nv, M = 8, 24, 12, 40
    x = problem(1003, B, Nt, Nv, M, device=DEV)
    m = _model("bf16", K=4)
    m.mb_feat_t, m.mb_feat_v = x["mb_feat_t"].clone(), x["mb_feat_v"].clone()
    m.mb_mask_t, m.mb_mask_v = x["mb_mask_t"].clone(), x["mb_mask_v"].clone()
    old_ind = torch.arange(1000, 1000 + M, device=DEV)
    m.mb_ind = old_ind.clone()
    idx = x["idx"].clone()
    vid = x["video_feat"].clone()
    pushed = []

    def step():
        with torch.no_grad():
            m(x["text_feat"], x["text_mask"], vid, x["video_mask"], idx, 0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for r in range(2):                                   # eager steps first (also warms the allocator)
            idx.copy_(torch.arange(B, device=DEV) + 10 * r)
            vid.copy_(x["video_feat"] + r)
            pushed.append((idx.clone(), vid.clone()))
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    pushed.append((idx.clone(), vid.clone()))                # the capture itself does not execute: first replay below
    for r in range(2, 5):
        idx.copy_(torch.arange(B, device=DEV) + 10 * r)
        vid.copy_(x["video_feat"] + r)
        if r > 2:
            pushed.append((idx.clone(), vid.clone()))
        else:
            pushed[-1] = (idx.clone(), vid.clone())
        g.replay()
    torch.cuda.synchronize()
    # the persistent prepared shadow of the bank followed every push: it equals a fresh prepare of the raw ring
    assert m._mb_shadow is not None
    for k, (feat, mask) in enumerate((("mb_feat_t", "mb_mask_t"), ("mb_feat_v", "mb_mask_v"))):
        fresh = ops.prepare_tokens(m._mb[feat], m._mb[mask], want_lo=True)
        assert torch.equal(fresh.hi, m._mb_shadow[k].hi) and torch.equal(fresh.lo, m._mb_shadow[k].lo)
        assert torch.equal(fresh.norm, m._mb_shadow[k].norm)
    ref_ind = torch.cat([p[0] for p in reversed(pushed)] + [old_ind])[:M]
    ref_vid = torch.cat([p[1] for p in reversed(pushed)] + [x["mb_feat_v"]])[:M]
    assert torch.equal(m.mb_ind, ref_ind)
    assert torch.equal(m.mb_feat_v, ref_vid)


def test_forward_updates_memory_bank_fifo_and_eval_returns_none():
    B, Nt, Nv, M = 16, 24, 12, 40
    x = problem(1001, B, Nt, Nv, M, device=DEV)
    m = _model("bf16", K=8)
    m.mb_feat_t, m.mb_feat_v = x["mb_feat_t"].clone(), x["mb_feat_v"].clone()
    m.mb_mask_t, m.mb_mask_v = x["mb_mask_t"].clone(), x["mb_mask_v"].clone()
    m.mb_ind = torch.arange(1000, 1000 + M, device=DEV)
    ref_ind = torch.cat((x["idx"], m.mb_ind))[:M]
    ref_v = torch.cat((x["video_feat"], m.mb_feat_v))[:M]
    out = m(x["text_feat"], x["text_mask"], x["video_feat"], x["video_mask"], x["idx"], 0)
    assert len(out) == 5 and all(o.dim() == 0 for o in out)
    assert torch.equal(m.mb_ind, ref_ind) and torch.equal(m.mb_feat_v, ref_v)
    assert m.mb_feat_v.shape[0] == M
    m.eval()
    assert m(x["text_feat"], x["text_mask"], x["video_feat"], x["video_mask"], x["idx"], 0) is None
    # K > B is rejected like the reference's IndexError (until_module.py:119-123)
    m.train()
    m.config.num_neighbors = 20
    with pytest.raises((ValueError, IndexError)):
        m(x["text_feat"], x["text_mask"], x["video_feat"], x["video_mask"], x["idx"], 0)


@pytest.mark.parametrize("M", [8, 16])
def test_a_batch_as_large_as_the_bank_replaces_it(M):
    """modeling.py:244-249: with B >= capacity the bank becomes cat(batch, bank)[:capacity] = the batch's first rows.  Here one
    nr_copy_group launch into the bank's own storage (no cat, no new tensors), from a ring whose head had moved; the next step's
    losses are those of a model whose bank was SET to those rows."""
    B, Nt, Nv = 16, 24, 12
    x = problem(1001, B, Nt, Nv, 40, device=DEV)
    m, ref = _model("bf16", K=8), _model("bf16", K=8)
    m.mb_feat_t, m.mb_feat_v = x["mb_feat_t"][:M].clone(), x["mb_feat_v"][:M].clone()
    m.mb_mask_t, m.mb_mask_v = x["mb_mask_t"][:M].clone(), x["mb_mask_v"][:M].clone()
    m.mb_ind = torch.arange(1000, 1000 + M, device=DEV)
    ptrs = {k: v.data_ptr() for k, v in m._mb.items()}
    with torch.no_grad():
        m(x["text_feat"], x["text_mask"], x["video_feat"], x["video_mask"], x["idx"], 0)
    assert {k: v.data_ptr() for k, v in m._mb.items()} == ptrs                     # in place
    assert torch.equal(m.mb_ind, x["idx"][:M]) and torch.equal(m.mb_feat_v, x["video_feat"][:M])
    assert torch.equal(m.mb_feat_t, x["text_feat"][:M]) and torch.equal(m.mb_mask_v, x["video_mask"][:M].float())
    ref.mb_feat_t, ref.mb_feat_v = x["text_feat"][:M].clone(), x["video_feat"][:M].clone()
    ref.mb_mask_t, ref.mb_mask_v = x["text_mask"][:M].float(), x["video_mask"][:M].float()
    ref.mb_ind = x["idx"][:M].clone()
    y = problem(1002, B, Nt, Nv, 40, device=DEV)
    for mm in (m, ref):
        mm._rng_state_on(torch.device(DEV, 0))[1] = 77
    with torch.no_grad():
        a = torch.stack(m(y["text_feat"], y["text_mask"], y["video_feat"], y["video_mask"], y["idx"], 0))
        b = torch.stack(ref(y["text_feat"], y["text_mask"], y["video_feat"], y["video_mask"], y["idx"], 0))
    assert torch.equal(a, b)


@pytest.mark.parametrize("bank_early", [0, 2])
def test_step_with_the_bank_products_as_chained_tile_pairs(bank_early):
    """head.PAIR_BANK_PRODUCTS (off by default: it loses inside the step): the loss-only step at the
    bench workload with both bank products in one launch of chained tile pairs gives bit-identical losses, whether the bank
    chains run early on the local stream or beside the Sinkhorn solve."""
    from neighborretr_amd import head
    B, Nt, Nv, M, K = 128, 24, 12, 512, 20
    x = problem(1002, B, Nt, Nv, M, device=DEV)
    nz = {k: v.to(DEV) for k, v in noise(1002, B, Nt, Nv).items()}
    out = []
    for pair in (False, True):
        m = _model("bf16", K=K)
        m.bank_early = bank_early
        m.bank_frozen = True
        old = head.PAIR_BANK_PRODUCTS
        head.PAIR_BANK_PRODUCTS = pair
        try:
            with torch.no_grad():
                out.append(torch.stack(_losses(m, x, nz, K)).cpu())
        finally:
            head.PAIR_BANK_PRODUCTS = old
    assert torch.isfinite(out[0]).all() and torch.equal(out[0], out[1])


def test_a_failing_bank_push_does_not_poison_later_steps():
    """the split tail's two self-finalizing launches share one device counter.  A push that raises between them
    must not leave it non-zero -- the following step has to return the same losses as a clean model."""
    B, Nt, Nv, M, K = 32, 24, 12, 64, 8
    x = problem(1005, B, Nt, Nv, M, device=DEV)
    m = _model("bf16", K=K)
    for k in ("mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v"):
        setattr(m, k, x[k].clone().float() if "mask" in k else x[k].clone())
    m.mb_ind = torch.arange(M, device=DEV)
    m.bank_frozen = True

    def run():
        m._rng_state = None
        torch.manual_seed(5)
        with torch.no_grad():
            return torch.stack(m(x["text_feat"], x["text_mask"], x["video_feat"], x["video_mask"], x["idx"], 0)).cpu()
    clean = run()
    assert torch.isfinite(clean).all()
    m.bank_frozen = False
    real = m.update_memory_bank

    def boom(*a, **k):
        raise RuntimeError("injected push failure")
    m.update_memory_bank = boom
    with pytest.raises(RuntimeError, match="injected"):
        run()
    m.update_memory_bank = real
    m.bank_frozen = True
    m._ring_advanced = False
    torch.cuda.synchronize()
    again = run()
    assert torch.equal(again, clean), (again, clean)


def test_capture_guard_fires_inside_a_real_capture_and_the_shipped_step_passes_it():
    """The capture rules in code (neighborretr_amd/capture_guard.py): inside a REAL stream capture the guard refuses the
    re-fork of a joined stream BEFORE the runtime sees the edge (so the capture goes on and ends cleanly -- the crashing
    capture itself is never executed), and the shipped loss-only step, captured, goes through the guard edge by edge."""
    from neighborretr_amd import capture_guard as CG
    x = torch.zeros(1 << 12, device=DEV)
    # three DIFFERENT streams: torch hands out its streams round-robin from a pool of 32, so after enough models a "new"
    # stream can be the very stream the capture runs on (the guard then rightly sees a stream waiting on itself)
    pool = {}
    while len(pool) < 3:
        st = torch.cuda.Stream()
        pool.setdefault(st.cuda_stream, st)
    cap, s1, s2 = pool.values()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        cur = torch.cuda.current_stream()
        assert CG._topology(cur) is not None
        CG.wait_stream(s1, cur)
        CG.wait_stream(s2, cur)
        with torch.cuda.stream(s2):
            a = x + 1
        with torch.cuda.stream(s1):
            b = x + 2
            CG.wait_stream(s1, s2)
        with pytest.raises(CG.CaptureTopologyError, match="already joined"):
            CG.wait_stream(s2, s1)                      # tools/capture_refork.py's crashing edge: refused, never issued
        with torch.cuda.stream(s1):
            c = a + b
        CG.wait_stream(cur, s1)
    g.replay()
    torch.cuda.synchronize()
    assert float(c[0]) == 3.0
    # the shipped step
    B, Nt, Nv, M, K = 32, 24, 12, 64, 8
    p = problem(1006, B, Nt, Nv, M, device=DEV)
    m = _model("bf16", K=K)
    for k in ("mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v"):
        setattr(m, k, p[k].clone().float() if "mask" in k else p[k].clone())
    m.mb_ind = torch.arange(M, device=DEV)
    out = {}

    def step():
        with torch.no_grad():
            out["l"] = torch.stack(m(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    before = dict(CG.STATS)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        step()
    assert CG.STATS["captures"] == before["captures"] + 1
    assert CG.STATS["edges"] >= before["edges"] + 7, CG.STATS     # local fork, side x2, sibling join, push fork, logits event, 2+ joins
    g2.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(out["l"]).all()
