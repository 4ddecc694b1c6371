"""CPU: the oracle restatement against the vectors captured from the reference itself
(tests/golden/*.npz, made by oracle/capture_golden.py).  fp32 tolerances: 2e-5 absolute."""
import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import synth
from util import golden, maxdiff, noise, params, problem

TOL = 2e-5


def _case(name):
    g = golden(name)
    B, Nt, Nv, M, K = (int(g[k]) for k in ("B", "Nt", "Nv", "M", "K"))
    x = problem(int(g["seed"]), B, Nt, Nv, M, blank_video=int(g["blank_video"]))
    return g, x, params(int(g["param_seed"])), noise(int(g["seed"]), B, Nt, Nv), (B, Nt, Nv, M, K)


@pytest.mark.parametrize("name", ["c1_b16", "r32_blank", "c4_b8"])
def test_local_level_and_bank(name):
    g, x, P, nz, (B, Nt, Nv, M, K) = _case(name)
    S, _, _, w_t, w_v, _, _ = O.local_level_parts(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], P)
    assert maxdiff(S, g["S"]) < TOL
    assert maxdiff(w_t, g["w_t"]) < TOL and maxdiff(w_v, g["w_v"]) < TOL
    bt2v = O.local_level(x["text_feat"], x["mb_feat_v"], x["text_mask"], x["mb_mask_v"], P)[0]
    bv2t = O.local_level(x["mb_feat_t"], x["video_feat"], x["mb_mask_t"], x["video_mask"], P)[1]
    assert maxdiff(bt2v, g["bank_t2v"]) < TOL and maxdiff(bv2t, g["bank_v2t"]) < TOL
    nb, _ = O.neighbor_mask(S, K)
    assert np.array_equal(nb.numpy().astype(np.uint8), g["nb_mask"])
    if name == "r32_blank":
        assert float(S[:, 5].abs().max()) == 0.0          # fully masked video -> exact-zero column
        return
    Ln, parts = O.neighbor_adjusting_parts(S, bv2t, K, 3.0)
    assert maxdiff(parts["p"], g["pos_weights_t2v"]) < TOL
    assert maxdiff(parts["p"].sum(-1), torch.full((B,), 2.0)) < 1e-5      # sum of positive weights = 2


@pytest.mark.parametrize("name", ["c1_b16", "c2_b128"])
def test_full_losses_and_grads(name):
    g, x, P, nz, (B, Nt, Nv, M, K) = _case(name)
    hp = dict(synth.DEFAULT_HP, num_neighbors=K)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    tf = x["text_feat"].clone().requires_grad_(True)
    vf = x["video_feat"].clone().requires_grad_(True)
    ls = torch.tensor(100.0, requires_grad=True)
    losses, parts = O.compute_losses(tf, vf, x["text_mask"], x["video_mask"], x["mb_feat_t"], x["mb_feat_v"],
                                     x["mb_mask_t"], x["mb_mask_v"], P, hp, ls, nz, return_parts=True)
    assert maxdiff(torch.stack(losses), g["losses"]) < 1e-4
    assert maxdiff(parts["G"], g["G"]) < 1e-4 and maxdiff(parts["gt"], g["gt"]) < TOL
    assert maxdiff(parts["w_text"], g["w_text"]) < TOL and maxdiff(parts["w_video"], g["w_video"]) < TOL
    losses[0].backward()
    assert abs(float(tf.grad.norm()) - float(g["g_text_norm"])) < 1e-4 * float(g["g_text_norm"])
    assert abs(float(vf.grad.norm()) - float(g["g_video_norm"])) < 1e-4 * float(g["g_video_norm"])
    assert maxdiff(tf.grad[:2, :4, :64], g["g_text_slice"]) < 1e-6
    assert maxdiff(vf.grad[:2, :4, :64], g["g_video_slice"]) < 1e-6
    assert abs(float(ls.grad) - float(g["g_logit_scale"])) < 1e-5
    names = [str(n) for n in g["param_names"]]
    mine = np.array([0.0 if P[n].grad is None else float(P[n].grad.norm()) for n in names])
    assert np.max(np.abs(mine - g["param_grad_norms"])) < 1e-3 * max(1.0, float(g["param_grad_norms"].max()))
    # dead parameters stay without gradient, *_fc1 gets exactly zero
    for n, v in zip(names, g["param_grad_norms"]):
        if "_fc0." in n or "_intra." in n or "_fc1." in n:
            assert v == 0.0


def test_c4_components_and_reference_crash():
    g, x, P, nz, (B, Nt, Nv, M, K) = _case("c4_b8")
    gt, gv = O.merge_global_features(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], P, nz)
    assert gt.shape[1] == 3 and gv.shape[1] == 6
    # the documented reduction for several global tokens: w_i = mean over the sample's tokens of the reference's [B,G] weights
    w_raw = O.centrality_weights(x["text_feat"], x["video_feat"], gt, gv, 0.3)
    w_mean = O.centrality_weights(x["text_feat"], x["video_feat"], gt, gv, 0.3, multi_token="mean")
    assert w_raw[0].shape == (B, 3) and w_raw[1].shape == (B, 6) and w_mean[0].shape == (B,)
    assert maxdiff(w_mean[0], w_raw[0].mean(-1)) == 0.0 and maxdiff(w_mean[1], w_raw[1].mean(-1)) == 0.0
    assert maxdiff(gt, g["gt"]) < TOL and maxdiff(gv, g["gv"]) < TOL
    G, _ = O.global_level(gt, gv, P)
    assert maxdiff(G, g["G"]) < 1e-4
    assert maxdiff(O.sinkhorn_targets(G, 0.7), g["tgt_t2v"]) < 1e-5
    assert abs(float(O.uniform_loss(G, 3.0, 0.7)) - float(g["L_uniform_direct"])) < 1e-4
    assert int(g["centrality_raises"]) == 1          # the reference itself fails at this shape
    with pytest.raises(RuntimeError):
        O.compute_losses(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], x["mb_feat_t"],
                         x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"], P, dict(synth.DEFAULT_HP, num_neighbors=K),
                         torch.tensor(100.0), nz)
    hp = dict(synth.DEFAULT_HP, num_neighbors=K)
    losses, parts = O.compute_losses(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], x["mb_feat_t"],
                                     x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"], P, hp, torch.tensor(100.0), nz,
                                     return_parts=True, centrality_multi_token="mean")
    # under the flag the terms the reference can still compute are the reference's
    assert abs(float(losses[2]) - float(g["L_uniform_direct"])) < 1e-4
    assert abs(float(losses[3]) - float(g["L_neighbor_direct"])) < 1e-4
    assert abs(float(losses[4]) - float(g["L_kl_direct"])) < 1e-4
    # ... and the centrality term is -mean_i w_i log_softmax(100 S)[i,i] with the averaged weights
    lp = torch.log_softmax(parts["S"] * 100.0, -1).diag()
    lpt = torch.log_softmax(parts["S"].t() * 100.0, -1).diag()
    want = (-(lp * parts["w_text"]).mean() - (lpt * parts["w_video"]).mean()) / 2
    assert abs(float(losses[1]) - float(want)) < 1e-6 and torch.isfinite(losses[0])


def _stage0_scores(x, P, which, noise):
    """Centre scores of the stage-0 DPC-KNN call of one modality, as oracle.dpc_knn computes them."""
    import torch.nn.functional as F
    feat, mask = x[which + "_feat"], x[which + "_mask"]
    ctm = which + "_ctm0"
    C = feat.shape[-1]
    y = feat + F.conv1d(feat.transpose(1, 2), P[ctm + ".conv.conv.weight"], padding=1).transpose(1, 2)
    y = F.layer_norm(y, (C,), P[ctm + ".norm.weight"], P[ctm + ".norm.bias"])
    dist = torch.cdist(y, y) / (C ** 0.5)
    valid = mask > 0
    dist = dist * valid[:, None, :] + (dist.max() + 1) * (~valid[:, None, :])
    near = torch.topk(dist, k=3, dim=-1, largest=False)[0]
    density = ((-(near ** 2).mean(-1)).exp() + noise * 1e-6) * valid
    higher = (density[:, None, :] > density[:, :, None]).to(y.dtype)
    dmax = dist.flatten(1).max(-1)[0][:, None, None]
    return (dist * higher + dmax * (1 - higher)).min(-1)[0] * density


@pytest.mark.parametrize("name", ["c1_b16", "c2_b128", "c4_b8"])
def test_centre_tie_rule_on_the_fixtures(name):
    """The build's tie rule for the cluster centres (exact score ties -> lower index; oracle.dpc_knn) against the
    vectors the reference produced with torch.topk (cluster.py:498): on every fixture the two rules pick different
    PADDING centres in the samples that have fewer valid tokens than centres -- and give the same global tokens,
    because an all-padding cluster merges to the zero vector whichever padding token leads it."""
    g, x, P, nz, (B, Nt, Nv, M, K) = _case(name)
    args = (x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], P, nz)
    gt_l, gv_l = O.merge_global_features(*args, centre_ties="lowest_index")
    gt_k, gv_k = O.merge_global_features(*args, centre_ties="torch_topk")
    for mine in ((gt_l, gv_l), (gt_k, gv_k)):
        assert maxdiff(mine[0], g["gt"]) < TOL and maxdiff(mine[1], g["gv"]) < TOL
    # the rows on which torch.topk (this host) and the documented rule choose different centres: exactly the short ones
    (t0, _), (v0, _) = O.merged_token_counts(Nt, Nv)
    for which, nzk, cnum in (("text", "t0", t0), ("video", "v0", v0)):
        score = _stage0_scores(x, P, which, nz[nzk])
        c_topk = torch.topk(score, k=cnum, dim=-1)[1]
        c_low = torch.sort(score, dim=-1, descending=True, stable=True)[1][:, :cnum]
        differ = (c_topk != c_low).any(1)
        short = x[which + "_mask"].sum(1) < cnum
        assert not (differ & ~short).any(), "the rules may only differ where exact (zero-score) ties exist"
        # ... and in the valid part of those rows they agree: the padding centres come after every valid centre
        for r in differ.nonzero().flatten().tolist():
            nv = int(x[which + "_mask"][r].sum())
            assert c_topk[r, :nv].tolist() == c_low[r, :nv].tolist()


def test_centre_tie_rule_and_where_torch_topk_differs():
    """A case where the two rules give DIFFERENT global tokens: seed 77, B=16, video sample 5 has two valid frames
    whose densities tie exactly (equal k-NN sets; the 1e-6 noise is below one ulp of the density), so their centre
    scores tie and the ORDER of the two clusters depends on the rule -- the next stage's token convolution sees them
    in a different order.  The reference's torch.topk puts token 1 first on this host; the documented rule (and the
    HIP kernels, tests/test_head_gpu.py) token 0.  Every other sample is identical under both rules."""
    x = problem(77, 16, 24, 12, 4)
    P, nz = params(), noise(77, 16, 24, 12)
    args = (x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], P, nz)
    gt_l, gv_l = O.merge_global_features(*args, centre_ties="lowest_index")
    gt_k, gv_k = O.merge_global_features(*args, centre_ties="torch_topk")
    assert maxdiff(gt_l, gt_k) == 0.0
    rows = ((gv_l - gv_k).abs().flatten(1).max(1)[0] > 0).nonzero().flatten().tolist()
    score = _stage0_scores(x, P, "video", nz["v0"])
    assert rows in ([5], [])                      # [] if this host's torch.topk happens to agree
    assert int(x["video_mask"][5].sum()) == 2 and float(score[5, 0]) == float(score[5, 1])      # the exact tie
    c_low = torch.sort(score, dim=-1, descending=True, stable=True)[1][:, :3]
    assert c_low[5].tolist() == [0, 1, 2]


def test_blank_video_poisons_reference_losses():
    """A fully masked video makes the REFERENCE's neighbour / global terms NaN; recorded, not fixed."""
    g = golden("r32_blank")
    assert np.isnan(g["losses"]).all() and np.isnan(g["L_neighbor_direct"])


def test_metrics_with_ties():
    g = golden("metrics256")
    n = 256
    S = (synth.normal(42, "metrics/S", (n, n)) * 0.1).astype(np.float32)
    S[np.arange(n), np.arange(n)] += 0.25
    for i in range(0, n, 16):
        S[i, (i + 3) % n] = S[i, i]
    for i in range(5, n, 16):
        S[i, (i + 7) % n] = np.nextafter(S[i, i], np.float32(10))
        S[i, (i + 9) % n] = np.nextafter(S[i, i], np.float32(-10))
    m = O.compute_metrics(S)
    assert np.array_equal(np.array(m["cols"]), g["cols"])
    assert np.allclose([m[k] for k in ("R1", "R5", "R10", "R50", "MR", "MeanR")], g["scalars"])
    assert np.array_equal(np.array(O.compute_metrics(S.T)["cols"]), g["cols_T"])


def test_memory_bank_fifo():
    mk = lambda n, s: (torch.arange(n) + s, torch.randn(n, 3, 4), torch.randn(n, 2, 4), torch.ones(n, 3), torch.ones(n, 2))
    empty = (torch.tensor([], dtype=torch.long), torch.empty(0, 0, 0), torch.empty(0, 0, 0), torch.empty(0, 0), torch.empty(0, 0))
    b0 = mk(6, 0)
    bank = O.update_memory_bank(empty, b0)
    assert bank[0].tolist() == list(range(6))
    bank = O.update_memory_bank(bank, mk(4, 100))
    assert bank[0].tolist() == [100, 101, 102, 103, 0, 1]          # newest first, capacity stays 6
    bank = O.update_memory_bank(bank, mk(8, 200))
    assert bank[0].tolist() == list(range(200, 206))                 # B > capacity: first rows of the batch


def test_multi_sentence_metrics_against_reference_vector():
    """oracle/capture_multi_sentence.py: the reference's own multi-sentence metrics (metrics.py:82-148 on the padded
    tensor of evaluator.py:236-250), NaN scores on and off the own column included."""
    g = golden("multi_sentence")
    t2v, v2t = O.multi_sentence_metrics(g["S"], g["cut_off_points"].tolist())
    keys_t = ("R1", "R5", "R10", "R50", "MedianR", "MeanR", "Std_Rank", "MR")
    assert np.allclose([t2v[k] for k in keys_t], g["t2v"], rtol=1e-6)
    assert np.array_equal(np.array(v2t["cols"]), g["v2t_cols"])
    assert np.allclose([v2t[k] for k in ("R1", "R5", "R10", "R50", "MR", "MeanR")], g["v2t"])


def test_c3_b1024_losses_at_full_size():
    """BASELINE configs[2] (global B = 1024): the oracle's five losses and the reduced forms of its batch similarity against
    the reference's own (oracle/capture_golden_large.py).  ~10 s of CPU."""
    g = golden("c3_b1024")
    B, Nt, Nv, M, K = (int(g[k]) for k in ("B", "Nt", "Nv", "M", "K"))
    x, P, nz = problem(int(g["seed"]), B, Nt, Nv, M), params(int(g["param_seed"])), noise(int(g["seed"]), B, Nt, Nv)
    hp = dict(synth.DEFAULT_HP, num_neighbors=K)
    with torch.no_grad():
        losses, parts = O.compute_losses(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], x["mb_feat_t"],
                                         x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"], P, hp, torch.tensor(100.0), nz,
                                         return_parts=True)
    assert maxdiff(torch.stack(losses), g["losses"]) < 1e-4
    S, G = parts["S"].double(), parts["G"].double()
    assert maxdiff(S.sum(1), g["S_rowsum"]) < 1e-4 and maxdiff(S.sum(0), g["S_colsum"]) < 1e-4
    assert maxdiff(torch.diagonal(S), g["S_diag"]) < TOL and maxdiff(S[:64, :64], g["S_corner"]) < TOL
    assert maxdiff(G[:64, :64], g["G_corner"]) < 1e-3 * float(np.abs(g["G_corner"]).max())
    assert maxdiff(parts["bank_t2v"].mean(-1), g["bank_c_t2v"]) < TOL and maxdiff(parts["bank_v2t"].mean(-1), g["bank_c_v2t"]) < TOL
    ref = O.compute_metrics(parts["S"].numpy())
    assert np.array_equal(np.asarray(ref["cols"]), g["cols"])
