"""GPU: seeded random SHAPES through the whole loss head against the CPU oracle -- batch sizes that are not multiples of the kernels'
block heights, odd token counts, banks smaller / larger than the batch and not a multiple of it, every K from 1 to B, ragged masks.
Token counts stay within 24 / 12 (one global token per sample survives the two clustering stages: modeling.py:188-196; the
multi-token centrality term has no reference answer, until_module.py:321).  Bar: the split-bf16 plan's 2e-4 on every loss."""
import numpy as np
import pytest
import torch

import nr_oracle as O
from neighborretr_amd import modeling
from util import noise, params, problem

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(seed):
    r = np.random.RandomState(1000 + seed)
    B = int(r.randint(2, 49))
    Nt = int(r.randint(13, 25))           # >= 3 tokens must reach stage 1: its DPC-KNN takes the k = 3 nearest (cluster.py:476; the
    Nv = int(r.randint(9, 13))            # reference's torch.topk raises below that -- test_too_few_tokens_for_the_second_stage_raises)
    M = int(r.choice([max(2, B // 2), B, B + 3, 2 * B, 3 * B + 1]))
    K = int(r.randint(1, B + 1))
    return B, Nt, Nv, M, K


@pytest.mark.parametrize("seed", range(24))
def test_random_shapes_match_the_oracle(seed):
    B, Nt, Nv, M, K = _case(seed)
    x = problem(500 + seed, B, Nt, Nv, M)
    nz = noise(500 + seed, B, Nt, Nv)
    P = params()
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K), precision="bf16x3")
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).train()
    with torch.no_grad():
        m.clip.logit_scale.fill_(float(np.log(100.0)))
    c = m.config
    hp = dict(centrality_scale=c.centrality_scale, beta=c.beta, num_neighbors=K, temperature=c.temperature,
              uniform_weight=c.uniform_weight, neighbor_weight=c.neighbor_weight, kl_weight=c.kl_weight)
    ref = torch.stack(O.compute_losses(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], x["mb_feat_t"], x["mb_feat_v"],
                                       x["mb_mask_t"], x["mb_mask_v"], P, hp, 100.0, nz)).numpy()
    xg = {k: v.to(DEV) for k, v in x.items()}
    nzg = {k: v.to(DEV) for k, v in nz.items()}
    with torch.no_grad():
        got = torch.stack(m._compute_losses(xg["text_feat"], xg["video_feat"], xg["text_mask"], xg["video_mask"], xg["mb_feat_t"],
                                            xg["mb_feat_v"], xg["mb_mask_t"], xg["mb_mask_v"], c.centrality_scale, c.beta, K,
                                            c.temperature, m.clip.logit_scale.exp(), noise=nzg)).cpu().numpy()
    # K = B - 2 leaves one sample outside the neighbour set: the reference's min-max normalisation is 0 / 0 there and its
    # neighbour loss (and the total) NaN -- the same entries must be NaN here, every other one within the bar
    assert (np.isnan(got) == np.isnan(ref)).all(), (got, ref)
    ok = ~np.isnan(ref)
    d = np.abs(got[ok] - ref[ok])
    print(f"\n[seed {seed}: B={B} Nt={Nt} Nv={Nv} M={M} K={K}] |dL| = {d.tolist()}{'  (NaN where the reference is NaN)' if not ok.all() else ''}")
    assert d.max() < 2e-4, (got, ref)


def test_too_few_tokens_for_the_second_stage_raises():
    """12 text tokens leave 2 for stage 1, whose DPC-KNN asks for the 3 nearest: the reference's torch.topk raises (cluster.py:476), the
    oracle raises, and so does the HIP path -- no silent clamp."""
    B, Nt, Nv, M, K = 8, 12, 12, 8, 4
    x = problem(3, B, Nt, Nv, M)
    nz = noise(3, B, Nt, Nv)
    P = params()
    with pytest.raises(RuntimeError):
        O.merge_global_features(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], P, nz)
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K), precision="bf16x3")
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).train()
    xg = {k: v.to(DEV) for k, v in x.items()}
    nzg = {k: v.to(DEV) for k, v in nz.items()}
    with torch.no_grad(), pytest.raises((RuntimeError, ValueError)):
        m.merge_global_features(xg["text_feat"], xg["video_feat"], xg["text_mask"], xg["video_mask"], nzg)


# ---- components, seeded random shapes ------------------------------------------------------------------------------------------
def _m(precision="bf16x3"):
    m = modeling.NeighborRetr(modeling.default_config(), precision=precision)
    m.load_state_dict(params(), strict=False)
    return m.to(DEV).eval()


@pytest.mark.parametrize("seed", range(12))
def test_random_rectangular_similarity_matches_the_oracle(seed):
    """get_similarity_logits on A texts x Bv videos (evaluation's shape: the two counts differ), token counts up to 64 x 64, ragged
    masks, one sample with a SINGLE valid token and one video with none at all."""
    from neighborretr_amd import synth
    r = np.random.RandomState(2000 + seed)
    A, Bv = int(r.randint(1, 40)), int(r.randint(1, 40))
    Nt, Nv = int(r.randint(1, 65)), int(r.randint(1, 65))
    t, _, tm, _ = synth.make_samples(900 + seed, "batch", A, Nt, 1, 512, 6.0, True)
    _, v, _, vm = synth.make_samples(950 + seed, "bank", Bv, 1, Nv, 512, 6.0, True)
    tm[0] = 0
    tm[0, 0] = 1
    if Bv > 1:
        vm[1] = 0                                   # a video without a valid frame: all scorer logits -9e15 -> uniform weights, products 0
    t, v, tm, vm = (torch.from_numpy(x) for x in (t, v, tm, vm))
    ref = O.local_level(t.double(), v.double(), tm, vm, {k: p.double() for k, p in params().items()})[0]
    with torch.no_grad():
        S, St = _m().get_similarity_logits(t.to(DEV), v.to(DEV), tm.to(DEV), vm.to(DEV))
    assert S.shape == (A, Bv) and torch.equal(St, S.T)
    d = float((S.cpu().double() - ref).abs().max())
    print(f"\n[seed {seed}: {A} texts x {Bv} videos, {Nt} x {Nv} tokens] max|dS| = {d:.2e}")
    assert d < 2e-6


@pytest.mark.parametrize("seed", range(12))
def test_random_sinkhorn_sizes_match_the_oracle(seed):
    """Every B, not only multiples of 4 / 64: the one-workgroup form, the cooperative form and the multi-launch fallback each get
    sizes they do not own in the parametrised tests; logits up to |G| ~ 40 (exp(100 x cosine) territory)."""
    from neighborretr_amd import ops
    r = np.random.RandomState(3000 + seed)
    B = int(r.choice([r.randint(2, 129), r.randint(129, 400), r.choice([192, 256, 320])]))
    g = torch.Generator().manual_seed(seed)
    G = torch.randn(B, B, generator=g) * float(r.choice([1.0, 9.0, 20.0]))
    tr, tc = ops.sinkhorn_targets(G.to(DEV), 0.7, 50)
    ref_r = O.sinkhorn_targets(G.double(), 0.7)
    ref_c = O.sinkhorn_targets(G.double().t(), 0.7)
    dr, dc = float((tr.cpu().double() - ref_r).abs().max()), float((tc.cpu().double() - ref_c).abs().max())
    print(f"\n[seed {seed}: B={B}] max|d target| = {dr:.2e} / {dc:.2e}")
    assert dr < 2e-5 and dc < 2e-5


@pytest.mark.parametrize("seed", range(12))
def test_random_dpc_knn_shapes_match_the_oracle(seed):
    from neighborretr_amd import ops
    r = np.random.RandomState(4000 + seed)
    B, N = int(r.randint(1, 20)), int(r.randint(3, 65))
    cnum = int(r.randint(1, min(N, 16) + 1))
    g = torch.Generator().manual_seed(seed)
    x = torch.nn.functional.layer_norm(torch.randn(B, N, 512, generator=g), (512,))
    mask = None
    if r.rand() < 0.7:
        mask = (torch.arange(N)[None] < torch.randint(1, N + 1, (B, 1), generator=g)).long()
    nz = torch.rand(B, N, generator=g)
    ref = O.dpc_knn(x, cnum, 3, mask, nz)
    got = ops.dpc_knn_assign(x.to(DEV), cnum, 3, None if mask is None else mask.to(DEV), nz.to(DEV)).cpu()
    assert torch.equal(got, ref), (B, N, cnum)


@pytest.mark.parametrize("seed", range(8))
def test_random_push_sequences_keep_the_bank_equal_to_the_oracle(seed):
    """A run of pushes with varying batch sizes (smaller than, equal to, larger than the bank, not dividing it): the bank the
    model holds -- read back through its public attributes, i.e. in the reference's newest-first order whatever the ring's head is --
    equals the oracle's FIFO (modeling.py:222-249) after every push."""
    r = np.random.RandomState(5000 + seed)
    M = int(r.choice([8, 12, 20]))
    Nt, Nv = 16, 9
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=2), precision="bf16")
    m.load_state_dict(params(), strict=False)
    m = m.to(DEV).train()
    x0 = problem(6000 + seed, 2, Nt, Nv, M)
    bank = (torch.arange(1000, 1000 + M), x0["mb_feat_t"], x0["mb_feat_v"], x0["mb_mask_t"].float(), x0["mb_mask_v"].float())
    m.mb_feat_t, m.mb_feat_v = bank[1].to(DEV), bank[2].to(DEV)
    m.mb_mask_t, m.mb_mask_v = bank[3].to(DEV), bank[4].to(DEV)
    m.mb_ind = bank[0].to(DEV)
    for step in range(7):
        B = int(r.choice([1, 3, 4, M // 2, M - 1, M, M + 3]))
        x = problem(6100 + 10 * seed + step, B, Nt, Nv, 1)
        idx = torch.arange(B) + 100 * step
        bank = O.update_memory_bank(bank, (idx, x["text_feat"], x["video_feat"], x["text_mask"].float(), x["video_mask"].float()))
        with torch.no_grad():
            m.update_memory_bank(idx.to(DEV), x["text_feat"].to(DEV), x["video_feat"].to(DEV), x["text_mask"].to(DEV), x["video_mask"].to(DEV))
        for k, want in zip(("mb_ind", "mb_feat_t", "mb_feat_v", "mb_mask_t", "mb_mask_v"), bank):
            got = getattr(m, k).cpu()
            assert got.shape == want.shape and torch.equal(got.to(want.dtype), want), (seed, step, B, k)


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("fused", [True, False], ids=["fused-clustering", "traced-clustering"])
def test_random_shapes_gradients_match_oracle_autograd(seed, fused):
    """Backward of the whole head at random shapes (the reference-captured gradient fixtures are B = 16 / 128 at 24 x 12 tokens
    only): d total / d features and per-parameter gradient norms against the CPU oracle's autograd, split-bf16 plan."""
    r = np.random.RandomState(7000 + seed)
    B = int(r.randint(5, 25))
    Nt, Nv = int(r.randint(13, 25)), int(r.randint(9, 13))
    M = int(r.choice([B, B + 3, 2 * B]))
    K = int(r.randint(1, B - 2))                      # (K = B - 2 is NaN in the reference; K >= B - 1 has its own tests)
    x = problem(7100 + seed, B, Nt, Nv, M)
    nz = noise(7100 + seed, B, Nt, Nv)
    P = {k: v.clone().requires_grad_(True) for k, v in params().items()}
    tf, vf = x["text_feat"].clone().requires_grad_(True), x["video_feat"].clone().requires_grad_(True)
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K), precision="bf16x3")
    m.load_state_dict(params(), strict=False)
    m = m.to(DEV).train()
    m.fused_training_clustering = fused
    with torch.no_grad():
        m.clip.logit_scale.fill_(float(np.log(100.0)))
    c = m.config
    hp = dict(centrality_scale=c.centrality_scale, beta=c.beta, num_neighbors=K, temperature=c.temperature,
              uniform_weight=c.uniform_weight, neighbor_weight=c.neighbor_weight, kl_weight=c.kl_weight)
    ref = O.compute_losses(tf, vf, x["text_mask"], x["video_mask"], x["mb_feat_t"], x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"],
                           P, hp, 100.0, nz)[0]
    ref.backward()
    xg = {k: v.to(DEV) for k, v in x.items()}
    xg["text_feat"].requires_grad_(True)
    xg["video_feat"].requires_grad_(True)
    nzg = {k: v.to(DEV) for k, v in nz.items()}
    got = m._compute_losses(xg["text_feat"], xg["video_feat"], xg["text_mask"], xg["video_mask"], xg["mb_feat_t"], xg["mb_feat_v"],
                            xg["mb_mask_t"], xg["mb_mask_v"], c.centrality_scale, c.beta, K, c.temperature, m.clip.logit_scale.exp(),
                            noise=nzg)[0]
    assert abs(float(got.detach()) - float(ref.detach())) < 2e-4
    got.backward()
    dev = {}
    for name, mine, want in (("text", xg["text_feat"].grad, tf.grad), ("video", xg["video_feat"].grad, vf.grad)):
        dev[name] = float((mine.cpu() - want).abs().max()) / float(want.abs().max())
    worst = ("", 0.0)
    named = dict(m.named_parameters())
    for n, p in P.items():
        if p.grad is None or n not in named:
            continue
        mine = 0.0 if named[n].grad is None else float(named[n].grad.norm())
        e = abs(mine - float(p.grad.norm())) / max(float(p.grad.norm()), 1e-3)
        if e > worst[1]:
            worst = (n, e)
    print(f"\n[seed {seed} {'fused' if fused else 'traced'}: B={B} Nt={Nt} Nv={Nv} M={M} K={K}] feature-gradient max deviation / largest entry: "
          f"text {dev['text']:.2e}, video {dev['video']:.2e}; worst parameter-gradient norm {worst[0]} {worst[1]:.2e}")
    assert dev["text"] < 5e-3 and dev["video"] < 5e-3 and worst[1] < 5e-3, (dev, worst)


@pytest.mark.parametrize("seed", range(10))
def test_random_metrics_with_planted_ties_match_the_oracle(seed):
    """compute_metrics on random sizes (1 .. 700 rows: one workgroup, many, not a multiple of anything) with exact ties planted on
    and off the diagonal -- a tie with the diagonal is one extra hit per tied entry in the reference (utils/metrics.py:58-66)."""
    from neighborretr_amd.metrics import RetrievalMetrics
    r = np.random.RandomState(8000 + seed)
    n = int(r.choice([1, 2, 3, r.randint(4, 130), r.randint(130, 700)]))
    S = r.randn(n, n).astype(np.float32)
    for _ in range(min(n, 12)):
        i, j = r.randint(n), r.randint(n)
        S[i, j] = S[i, i]                            # ties with the diagonal
        S[r.randint(n), r.randint(n)] = S[r.randint(n), r.randint(n)]
    if n > 3:
        S[2, :] = 0.25                               # a whole row of equal scores
    mine, ref = RetrievalMetrics.compute_metrics(S), O.compute_metrics(S)
    assert mine["cols"] == ref["cols"]
    for k in ("R1", "R5", "R10", "R50", "MR", "MedianR", "MeanR"):
        assert mine[k] == ref[k], (k, mine[k], ref[k])


@pytest.mark.parametrize("seed", range(8))
def test_random_multi_sentence_groups_match_the_oracle(seed):
    """Multi-sentence retrieval (MSVD / ActivityNet style, evaluator.py:225-262) on random caption-group sizes, with exact ties."""
    from neighborretr_amd.metrics import RetrievalMetrics
    r = np.random.RandomState(9000 + seed)
    n_video = int(r.randint(1, 60))
    sizes = r.randint(1, 7, size=n_video)
    cut = list(np.cumsum(sizes) - 1)
    S = r.randn(int(sizes.sum()), n_video).astype(np.float32)
    S[r.randint(S.shape[0], size=5), r.randint(n_video, size=5)] = 0.5
    S[r.randint(S.shape[0], size=5), r.randint(n_video, size=5)] = 0.5
    padded = O.pad_sentence_groups(S, cut)
    ref_tv, ref_vt = O.multi_sentence_metrics(S, cut)
    mine_tv = RetrievalMetrics.tensor_text_to_video_metrics(torch.from_numpy(padded).to(DEV))
    for k, v in ref_tv.items():
        assert abs(mine_tv[k] - v) < 1e-6 * max(1.0, abs(v)), (k, mine_tv[k], v)      # (the reference divides in fp32: metrics.py:113-126)
    vt = RetrievalMetrics.tensor_video_to_text_sim(torch.from_numpy(padded).to(DEV))
    mine_vt = RetrievalMetrics.compute_metrics(vt)
    assert mine_vt["cols"] == ref_vt["cols"]


def test_strided_inputs_and_mask_dtypes_give_the_same_losses():
    """What a caller may hand over: features as strided views (a slice of a wider buffer, an expanded batch), masks as int64 / int32 /
    bool / fp32 -- the step's losses are those of the contiguous fp32 / int64 call, bit for bit."""
    B, Nt, Nv, M, K = 12, 20, 11, 24, 4
    x = problem(42, B, Nt, Nv, M, device=DEV)
    nz = noise(42, B, Nt, Nv, device=DEV)
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K), precision="bf16")
    m.load_state_dict(params(), strict=False)
    m = m.to(DEV).train()
    c = m.config

    def losses(tf, vf, tm, vm):
        with torch.no_grad():
            return torch.stack(m._compute_losses(tf, vf, tm, vm, x["mb_feat_t"], x["mb_feat_v"], x["mb_mask_t"], x["mb_mask_v"],
                                                 c.centrality_scale, c.beta, K, c.temperature, m.clip.logit_scale.exp(), noise=nz))
    want = losses(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"])
    wide_t = torch.zeros(B, Nt, 640, device=DEV)
    wide_t[..., 64:576] = x["text_feat"]
    wide_v = torch.zeros(2 * B, Nv, 512, device=DEV)
    wide_v[::2] = x["video_feat"]
    for cast in (lambda t: t, lambda t: t.int(), lambda t: t.bool(), lambda t: t.float()):
        got = losses(wide_t[..., 64:576], wide_v[::2], cast(x["text_mask"]), cast(x["video_mask"]))
        assert torch.equal(got, want)


def test_random_sharded_steps_match_the_replicated_step():
    """tools/sharded_sweep.py: the synchronous sharded step at random world sizes, per-rank batches of 1 .. 8 samples, odd
    token counts and banks -- every emulated rank against the replicated step (<= 2e-5; the tool raises otherwise)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "sharded_sweep.py"), "8"], capture_output=True, text=True, timeout=900,
                       cwd=root, env=env)
    print(r.stdout)
    assert r.returncode == 0 and "sharded sweep: 8 cases passed" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("seed", range(8))
def test_random_long_token_counts_match_the_oracle(seed):
    """Token counts above 24 / 12 (ActivityNet-like, BASELINE configs[3] is 64 x 64): several global tokens per sample survive the
    clustering, the global logits take the multi-token form (modeling.py:516-539) and the centrality term the documented "mean"
    reduction (the reference itself raises there, until_module.py:321) -- losses against the oracle run the same way."""
    r = np.random.RandomState(12000 + seed)
    B = int(r.randint(4, 20))
    Nt, Nv = int(r.randint(25, 65)), int(r.randint(13, 65))
    M = int(r.choice([B, B + 5, 2 * B]))
    K = int(r.randint(1, B - 2))
    x = problem(12100 + seed, B, Nt, Nv, M)
    nz = noise(12100 + seed, B, Nt, Nv)
    P = params()
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K, centrality_multi_token="mean"), precision="bf16x3")
    m.load_state_dict(P, strict=False)
    m = m.to(DEV).train()
    with torch.no_grad():
        m.clip.logit_scale.fill_(float(np.log(100.0)))
    c = m.config
    hp = dict(centrality_scale=c.centrality_scale, beta=c.beta, num_neighbors=K, temperature=c.temperature,
              uniform_weight=c.uniform_weight, neighbor_weight=c.neighbor_weight, kl_weight=c.kl_weight)
    ref = torch.stack(O.compute_losses(x["text_feat"], x["video_feat"], x["text_mask"], x["video_mask"], x["mb_feat_t"], x["mb_feat_v"],
                                       x["mb_mask_t"], x["mb_mask_v"], P, hp, 100.0, nz, centrality_multi_token="mean")).numpy()
    xg = {k: v.to(DEV) for k, v in x.items()}
    nzg = {k: v.to(DEV) for k, v in nz.items()}
    with torch.no_grad():
        got = torch.stack(m._compute_losses(xg["text_feat"], xg["video_feat"], xg["text_mask"], xg["video_mask"], xg["mb_feat_t"],
                                            xg["mb_feat_v"], xg["mb_mask_t"], xg["mb_mask_v"], c.centrality_scale, c.beta, K,
                                            c.temperature, m.clip.logit_scale.exp(), noise=nzg)).cpu().numpy()
    d = np.abs(got - ref)
    print(f"\n[seed {seed}: B={B} Nt={Nt} Nv={Nv} M={M} K={K}] |dL| = {d.tolist()}")
    assert np.isfinite(got).all() and d.max() < 2e-4, (got, ref)
