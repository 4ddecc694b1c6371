"""CPU: the encoder glue of BASELINE configs[4] (neighborretr_amd/encoders.py) against vectors captured from the
reference's own CLIP / temporal-transformer code at a tiny size (oracle/capture_encoders.py -> tests/golden/enc_tiny.npz):
same state-dict names and shapes, same outputs for the same seeded parameters (fp32, <= 2e-5)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import capture_encoders as C  # noqa: E402  (seeded parameters / inputs only; it never touches the reference when imported)
from neighborretr_amd import encoders, modeling  # noqa: E402
from util import golden, maxdiff  # noqa: E402


def test_tiny_encoders_match_reference_outputs():
    g = golden("enc_tiny")
    clip = encoders.ClipEncoders(**C.DIMS).eval()
    assert sorted(clip.state_dict().keys()) == [str(k) for k in g["clip_keys"]]
    assert [str(tuple(clip.state_dict()[k].shape)) for k in sorted(clip.state_dict().keys())] == [str(s) for s in g["clip_shapes"]]
    clip.load_state_dict(C.seeded_state(clip, 11))
    holder = torch.nn.Module()
    holder.frame_position_embeddings = torch.nn.Embedding(C.DIMS["context_length"], C.DIMS["transformer_width"])
    holder.transformerClip = encoders.TemporalTransformer(C.DIMS["transformer_width"], C.TEMPORAL_LAYERS, C.DIMS["transformer_heads"])
    assert sorted(holder.state_dict().keys()) == [str(k) for k in g["temporal_keys"]]
    holder.load_state_dict(C.seeded_state(holder, 12))
    holder.eval()
    ids, mask, video, vmask = C.inputs()
    with torch.no_grad():
        t_cls, t_hidden = clip.encode_text(ids, return_hidden=True, mask=mask)
        v_cls, v_hidden = clip.encode_image(video.view(-1, 3, 64, 64), return_hidden=True)
        agg = encoders.aggregate_video_features(v_cls.view(3, -1, v_cls.shape[-1]), vmask, holder.frame_position_embeddings,
                                                holder.transformerClip)
    for name, mine in (("t_cls", t_cls), ("t_hidden", t_hidden), ("v_cls", v_cls), ("v_hidden", v_hidden), ("agg", agg)):
        assert maxdiff(mine, g[name]) < 2e-5 * max(1.0, float(np.abs(g[name]).max())), name


def test_model_with_encoders_has_the_reference_layout_and_runs():
    m = modeling.NeighborRetr(modeling.default_config(num_hidden_layers=2), with_encoders=True, encoder_dims=C.DIMS).eval()
    keys = set(m.state_dict().keys())
    for k in ("clip.visual.conv1.weight", "clip.visual.transformer.resblocks.1.attn.in_proj_weight", "clip.token_embedding.weight",
              "clip.text_projection", "clip.logit_scale", "frame_position_embeddings.weight",
              "transformerClip.resblocks.1.mlp.c_proj.bias", "text_weight_fc.0.weight", "video_ctm1.score.bias"):
        assert k in keys, k
    # modeling.py:199-219: the temporal transformer starts from the text tower's first blocks, the frame positions from
    # its positional embedding
    sd = m.state_dict()
    assert torch.equal(sd["frame_position_embeddings.weight"], sd["clip.positional_embedding"])
    assert torch.equal(sd["transformerClip.resblocks.0.attn.in_proj_weight"], sd["clip.transformer.resblocks.0.attn.in_proj_weight"])
    ids, mask, video, vmask = C.inputs()
    with torch.no_grad():
        t, v = m.get_text_video_feat(ids, mask, video, vmask)
    assert t.shape == (3, 8, 64) and v.shape == (3, 4, 64) and torch.isfinite(t).all() and torch.isfinite(v).all()
    # the 7-D loader layout [b, pair, bs, ts, c, h, w] of dataloader_retrieval.py reshapes the same way
    with torch.no_grad():
        _, v7 = m.get_text_video_feat(ids, mask, video.view(3, 1, 4, 1, 3, 64, 64), vmask)
    assert torch.equal(v7, v)
