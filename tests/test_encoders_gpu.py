"""GPU: BASELINE configs[4] -- the encoder glue in front of the HIP head ON the MI355X.

1. `encoders.py` on the GPU against the vectors captured from the reference's own CLIP / temporal-transformer code
   (tests/golden/enc_tiny.npz, oracle/capture_encoders.py): fp32 <= 1e-4 of the output's largest element (library GEMMs
   accumulate in another order than the CPU's; the CPU test holds 2e-5), and under the bf16 autocast the model trains
   in: <= 3e-2 (bf16 = 2^-9 per rounding through 2 + 2 residual blocks; the measured figure is printed).
2. configs[4] at its REAL size on one GPU -- B = 128, 12 frames of 3 x 224 x 224, 24 token ids, bank 512, ViT-B/32 towers +
   4-layer temporal transformer in bf16 autocast, random init (no checkpoint offline) -- through size-independent
   properties: finite losses; the losses of the end-to-end forward EQUAL the HIP head run on the encoders' own features
   (same noise, frozen bank); gradients reach both towers, the temporal transformer and the head; the bank push stored
   the encoders' features.  The head itself is pinned at this B by the c2_b128 fixture (tests/test_head_gpu.py).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import capture_encoders as C  # noqa: E402  (seeded parameters / inputs only; never touches the reference when imported)
from neighborretr_amd import encoders, modeling, synth  # noqa: E402
from util import golden, maxdiff  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tiny_modules():
    clip = encoders.ClipEncoders(**C.DIMS).eval()
    clip.load_state_dict(C.seeded_state(clip, 11))
    holder = torch.nn.Module()
    holder.frame_position_embeddings = torch.nn.Embedding(C.DIMS["context_length"], C.DIMS["transformer_width"])
    holder.transformerClip = encoders.TemporalTransformer(C.DIMS["transformer_width"], C.TEMPORAL_LAYERS, C.DIMS["transformer_heads"])
    holder.load_state_dict(C.seeded_state(holder, 12))
    return clip.to(DEV), holder.eval().to(DEV)


@pytest.mark.parametrize("autocast,tol", [(False, 1e-4), (True, 3e-2)], ids=["fp32", "bf16-autocast"])
def test_tiny_encoders_on_gpu_match_reference_outputs(autocast, tol):
    g = golden("enc_tiny")
    clip, holder = _tiny_modules()
    ids, mask, video, vmask = (t.to(DEV) for t in C.inputs())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        t_cls, t_hidden = clip.encode_text(ids, return_hidden=True, mask=mask)
        v_cls, v_hidden = clip.encode_image(video.view(-1, 3, 64, 64), return_hidden=True)
        agg = encoders.aggregate_video_features(v_cls.float().view(3, -1, v_cls.shape[-1]), vmask, holder.frame_position_embeddings,
                                                holder.transformerClip)
    dev = {}
    for name, mine in (("t_cls", t_cls), ("t_hidden", t_hidden), ("v_cls", v_cls), ("v_hidden", v_hidden), ("agg", agg)):
        scale = max(1.0, float(np.abs(g[name]).max()))
        dev[name] = maxdiff(mine.float(), g[name]) / scale
    print(f"\n[enc_tiny on the GPU, {'bf16 autocast' if autocast else 'fp32'}] max|d| / max|ref|: "
          + ", ".join(f"{k} {v:.2e}" for k, v in dev.items()))
    for name, d in dev.items():
        assert d < tol, (name, d)


def test_tiny_model_with_encoders_feeds_the_head_on_gpu():
    """The model-level glue (get_text_video_feat -> fp32 features of the reference's shapes) on the GPU equals the module-level
    outputs pinned above; masks arrive as int64 like the loaders' (dataloader_retrieval.py:256-257)."""
    m = modeling.NeighborRetr(modeling.default_config(num_hidden_layers=C.TEMPORAL_LAYERS), with_encoders=True, encoder_dims=C.DIMS)
    m.clip.load_state_dict(C.seeded_state(m.clip, 11))
    holder = torch.nn.Module()
    holder.frame_position_embeddings, holder.transformerClip = m.frame_position_embeddings, m.transformerClip
    holder.load_state_dict(C.seeded_state(holder, 12))
    m = m.to(DEV).eval()
    m.encoder_dtype = None                                   # fp32: compare with the fp32 fixture
    g = golden("enc_tiny")
    ids, mask, video, vmask = (t.to(DEV) for t in C.inputs())
    with torch.no_grad():
        t, v = m.get_text_video_feat(ids, mask, video, vmask)
    assert t.dtype == torch.float32 and v.dtype == torch.float32
    assert maxdiff(t, g["t_hidden"]) < 1e-4 * max(1.0, float(np.abs(g["t_hidden"]).max()))
    assert maxdiff(v, g["agg"]) < 1e-4 * max(1.0, float(np.abs(g["agg"]).max()))


def test_configs4_full_size_end_to_end_properties():
    B, Nt, Nv, M, K = 128, 24, 12, 512, 20
    torch.manual_seed(4004)
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K, num_hidden_layers=4), with_encoders=True).to(DEV).train()
    assert sum(p.numel() for p in m.clip.parameters()) > 140e6            # ViT-B/32 + text tower, not a toy
    _, _, tm, vm = synth.make_samples(4004, "e2e", B, Nt, Nv, d=8)
    tm, vm = torch.from_numpy(tm).to(DEV), torch.from_numpy(vm).to(DEV)
    ids = encoders.synthetic_text_ids(tm.cpu(), seed=4).to(DEV)
    video = torch.randn((B, Nv, 3, 224, 224), device=DEV, dtype=torch.bfloat16)
    idx = torch.arange(B, device=DEV)
    with torch.no_grad():
        tf, vf = m.get_text_video_feat(ids, tm, video, vm)
    assert tf.shape == (B, Nt, 512) and vf.shape == (B, Nv, 512) and tf.dtype == torch.float32
    assert torch.isfinite(tf).all() and torch.isfinite(vf).all()
    # the bank: M encoded samples (memory_bank.py:80-229 fills it from encoded batches)
    reps = M // B
    bank = dict(mb_feat_t=(tf.repeat(reps, 1, 1) + 0.01 * torch.randn((M, Nt, 512), device=DEV)).contiguous(),
                mb_feat_v=(vf.repeat(reps, 1, 1) + 0.01 * torch.randn((M, Nv, 512), device=DEV)).contiguous(),
                mb_mask_t=tm.repeat(reps, 1).float(), mb_mask_v=vm.repeat(reps, 1).float())
    for k, v in bank.items():
        setattr(m, k, v.clone())
    m.mb_ind = torch.arange(1000, 1000 + M, device=DEV)

    # (1) end-to-end forward == HIP head on the encoders' own features.  Same DPC-KNN noise stream (re-seeded), frozen bank.
    def seeded(fn):
        m._rng_state = None
        torch.manual_seed(77)
        with torch.no_grad():
            return torch.stack(fn()).cpu()
    m.bank_frozen = True
    e2e = seeded(lambda: m(ids, tm, video, vm, idx, 0))
    m.feature_mode = True
    try:
        head_only = seeded(lambda: m(tf, tm, vf, vm, idx, 0))
    finally:
        m.feature_mode = False
    print(f"\n[configs[4] full size] losses end to end {e2e.tolist()}  head on the encoders' features {head_only.tolist()}")
    assert torch.isfinite(e2e).all() and float(e2e[0]) > 0
    # the encoders are deterministic here (eval-free modules, no dropout): identical features -> identical losses
    assert torch.allclose(e2e, head_only, rtol=1e-5, atol=1e-6), (e2e, head_only)
    assert abs(float(e2e[0] - e2e[1:].sum())) < 1e-4 * abs(float(e2e[0]))
    m.bank_frozen = False

    # (2) training step: gradients reach the towers, the temporal transformer, the scorers and the clustering; the push
    # stored the encoders' features of this batch in front of the old bank (modeling.py:222-249)
    m.zero_grad(set_to_none=True)
    out = m(ids, tm, video, vm, idx, 0)
    assert len(out) == 5 and all(torch.isfinite(o) for o in out)
    out[0].backward()
    named = dict(m.named_parameters())
    # the patchify convolution is frozen in the reference (module_clip.py:325-326): no gradient there, by design
    assert not named["clip.visual.conv1.weight"].requires_grad and named["clip.visual.conv1.weight"].grad is None
    for n in ("clip.visual.positional_embedding", "clip.visual.transformer.resblocks.0.attn.in_proj_weight",
              "clip.visual.transformer.resblocks.11.mlp.c_proj.weight",
              "clip.transformer.resblocks.0.attn.in_proj_weight", "clip.token_embedding.weight",
              "transformerClip.resblocks.3.mlp.c_fc.weight", "frame_position_embeddings.weight",
              "text_weight_fc.0.weight", "video_weight_fc.2.weight", "text_ctm0.conv.conv.weight", "video_block1.attn.q.weight",
              "clip.logit_scale"):
        gr = named[n].grad
        assert gr is not None and torch.isfinite(gr).all() and float(gr.abs().max()) > 0, n
    assert torch.equal(m.mb_ind[:B], idx) and torch.equal(m.mb_ind[B:], torch.arange(1000, 1000 + M - B, device=DEV))
    assert maxdiff(m.mb_feat_t[:B], tf) < 1e-5 * float(tf.abs().max()) and maxdiff(m.mb_feat_v[:B], vf) < 1e-5 * float(vf.abs().max())
    assert torch.equal(m.mb_feat_v[B:], bank["mb_feat_v"][:M - B])
