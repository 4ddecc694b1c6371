"""GPU: the training step with the clustering forward on the grouped HIP kernels and the hand-derived backward
(model.fused_training_clustering, cluster_fused.ClusterStagesFn + cluster_backward.stage_backward) against the same step on
the autograd-traced torch ops of cluster.py: same losses, same gradients for every parameter and for the input features."""
import numpy as np
import pytest
import torch

from neighborretr_amd import modeling, synth
from util import params

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _step(fused, B, Nt, Nv, M, K, seed):
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict(params(), strict=False)
    m = m.to(DEV).train()
    m.fused_training_clustering = fused
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(seed, B, Nt, Nv, M).items()}
    m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
    m.mb_ind = torch.arange(M, device=DEV)
    m._rng_state_on(torch.device(DEV))[1] = 7                     # the same DPC-KNN tie-break noise in both runs
    tf = p["text_feat"].clone().requires_grad_(True)
    vf = p["video_feat"].clone().requires_grad_(True)
    losses = m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)
    losses[0].backward()
    grads = {k: q.grad.detach().clone() for k, q in m.named_parameters() if q.grad is not None}
    return [float(x.detach()) for x in losses], grads, tf.grad.clone(), vf.grad.clone()


@pytest.mark.parametrize("B,Nt,Nv,M,K", [(16, 24, 12, 64, 8), (128, 24, 12, 512, 20)])
def test_fused_training_clustering_matches_the_traced_path(B, Nt, Nv, M, K):
    l0, g0, dt0, dv0 = _step(False, B, Nt, Nv, M, K, 1002)
    l1, g1, dt1, dv1 = _step(True, B, Nt, Nv, M, K, 1002)
    assert np.allclose(l0, l1, rtol=2e-5, atol=2e-6), (l0, l1)
    assert set(g0) == set(g1)
    worst = ("", 0.0)
    # gradients that are analytically ~0 (a stage with ONE cluster is invariant to a shift of all scores: its score bias
    # only acts through the 1e-6 of the merge denominator) are compared on the scale of the whole gradient
    floor = 1e-4 * max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        scale = max(float(g0[k].abs().max()), floor)
        err = float((g0[k] - g1[k]).abs().max()) / scale
        worst = max(worst, (k, err), key=lambda t: t[1])
    print(f"\\n[B={B}] worst parameter-gradient deviation {worst[1]:.2e} ({worst[0]}); "
          f"d text_feat {float((dt0 - dt1).abs().max()) / float(dt0.abs().max()):.2e}, "
          f"d video_feat {float((dv0 - dv1).abs().max()) / float(dv0.abs().max()):.2e}")
    assert worst[1] < 2e-3, worst
    assert float((dt0 - dt1).abs().max()) < 2e-3 * float(dt0.abs().max())
    assert float((dv0 - dv1).abs().max()) < 2e-3 * float(dv0.abs().max())


@pytest.mark.parametrize("B,Nt,Nv,masked", [(16, 24, 12, True), (9, 4, 3, False), (5, 64, 64, True), (3, 11, 16, False)])
def test_hip_stage_backward_matches_the_torch_op_backward(B, Nt, Nv, masked):
    """One stage, text + video in the same launches (cluster_backward_hip.stage_backward_group: nine grouped HIP launches)
    against cluster_backward.stage_backward (torch ops; itself equal to autograd to 1e-10 in fp64, tests/test_host_cpu.py) on
    the SAME saved forward state: d x0 and every parameter gradient.  Shapes: MSR-VTT stage 0 (masks, 4 / 3 clusters), its
    stage 1 (4 -> 1 and 3 -> 1 tokens), ActivityNet stage 0 (64 tokens: the two-chunk attention kernel) and its stage 1.
    Bar: 2e-4 of the largest element of each gradient (split-bf16 GEMMs, fp32 everything else; measured deviations printed)."""
    from neighborretr_amd import cluster_backward as CB, cluster_backward_hip as CBH, cluster_fused as CF
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=4))
    m.load_state_dict(params(), strict=False)
    m = m.to(DEV).train()
    stage1 = Nt <= 16
    mods = ((m.text_ctm1, m.text_block1), (m.video_ctm1, m.video_block1)) if stage1 else \
           ((m.text_ctm0, m.text_block0), (m.video_ctm0, m.video_block0))
    keys = ("text1", "video1") if stage1 else ("text0", "video0")
    x_t = torch.from_numpy(synth.normal(5, "bx_t", (B, Nt, 512)).astype(np.float32)).to(DEV)
    x_v = torch.from_numpy(synth.normal(5, "bx_v", (B, Nv, 512)).astype(np.float32)).to(DEV)
    mk_t = mk_v = None
    if masked:
        mk_t = torch.ones((B, Nt), device=DEV)
        mk_v = torch.ones((B, Nv), device=DEV)
        mk_t[1, Nt // 2:] = 0
        mk_t[2, 2:] = 0                        # fewer valid tokens than clusters
        mk_v[0, Nv - 1:] = 0
        mk_v[3, 1:] = 0
    nz_t = torch.from_numpy(synth.uniform(6, "bn_t", (B, Nt)).astype(np.float32)).to(DEV) if hasattr(synth, "uniform") else torch.rand((B, Nt), device=DEV)
    nz_v = torch.from_numpy(synth.uniform(6, "bn_v", (B, Nv)).astype(np.float32)).to(DEV) if hasattr(synth, "uniform") else torch.rand((B, Nv), device=DEV)
    with torch.no_grad():
        (out_t, out_v), saved = CF.ctm_stage_group([(keys[0], x_t, mk_t, mods[0][0], mods[0][1], nz_t),
                                                    (keys[1], x_v, mk_v, mods[1][0], mods[1][1], nz_v)], m._ctm_cache, want_saved=True)
        g_t = torch.from_numpy(synth.normal(7, "bg_t", tuple(out_t.shape)).astype(np.float32)).to(DEV)
        g_v = torch.from_numpy(synth.normal(7, "bg_v", tuple(out_v.shape)).astype(np.float32)).to(DEV)
        assert all(CBH.supported(sv) for sv in saved)
        got = CBH.stage_backward_group([(keys[i], mods[i][0], mods[i][1], saved[i], g) for i, g in enumerate((g_t, g_v))], m._ctm_cache)
        torch.cuda.synchronize()
        for i, g in enumerate((g_t, g_v)):
            ctm, blk = mods[i]
            sv = dict(saved[i])
            sv["merged"] = sv["merged_pb"] - blk.attn.proj.bias
            d_x0, grads = CB.stage_backward(ctm, blk, sv, g)
            mine_x0, mine = got[i]
            scale = float(d_x0.abs().max())
            err = float((mine_x0 - d_x0).abs().max()) / scale
            worst = ("d_x0", err)
            assert set(id(k) for k in grads) == set(id(k) for k in mine)
            names = {id(p_): n for n, p_ in list(ctm.named_parameters()) + list(blk.named_parameters())}
            # score.bias: a cancelling sum (a shift of all scores of a sample changes nothing but the 1e-6 of the merge
            # denominator and the masked tokens' share): compared on the scale of the largest parameter gradient
            floor = 1e-4 * max(float(v.abs().max()) for v in grads.values())
            worst_bias = 0.0
            for p_, ref in grads.items():
                sc = max(float(ref.abs().max()), floor)
                e = float((mine[p_].reshape(ref.shape) - ref).abs().max()) / sc
                if names[id(p_)] == "score.bias":          # summation-order noise of the cancelling sum: its own, looser bar
                    worst_bias = e
                elif e > worst[1]:
                    worst = (names[id(p_)], e)
            print(f"\n[HIP stage backward, {'text' if i == 0 else 'video'} B={B} N={(Nt, Nv)[i]}] d_x0 {err:.2e}; worst {worst[0]} {worst[1]:.2e}; score.bias {worst_bias:.2e}")
            assert torch.isfinite(mine_x0).all()
            assert worst[1] < 2e-4, worst
            assert worst_bias < 5e-3, worst_bias
