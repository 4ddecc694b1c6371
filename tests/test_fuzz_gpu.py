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
