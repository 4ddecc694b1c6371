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
