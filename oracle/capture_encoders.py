#!/usr/bin/env python
"""Golden vectors for neighborretr_amd/encoders.py from the REFERENCE's own encoder code, at a tiny size.

Runs only in the build container (the reference checkout is mounted at /root/reference).  ViT-B/32 itself needs a
600 MB checkpoint that cannot be fetched here, and its outputs would not fit a fixture; the architecture code is the
same at any size, so the reference classes (module_clip.CLIP, module_cross.Transformer as the temporal transformer,
NeighborRetr.aggregate_video_features) are instantiated with 2 layers / width 64 / 32-pixel patches, filled with seeded
parameters (neighborretr_amd/synth.py, keyed by the state-dict name), run on seeded inputs, and their outputs stored in
tests/golden/enc_tiny.npz.  tests/test_encoders_cpu.py loads the same parameters into the build's modules.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = dict(embed_dim=64, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=32, context_length=16,
            vocab_size=100, transformer_width=64, transformer_heads=1, transformer_layers=2)
TEMPORAL_LAYERS = 2


def seeded_state(module, seed):
    from neighborretr_amd import synth
    out = {}
    for k, v in module.state_dict().items():
        z = synth.normal(seed, "enc/" + k, tuple(v.shape) or (1,)).reshape(v.shape).astype(np.float32)
        if k.endswith(("ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_post.weight", "ln_final.weight")):
            val = 1.0 + 0.05 * z
        elif k.endswith(".bias"):
            val = 0.02 * z
        elif k == "logit_scale":
            val = np.float32(np.log(1 / 0.07)) + 0 * z
        else:
            val = 0.08 * z
        out[k] = torch.from_numpy(np.asarray(val, dtype=np.float32))
    return out


def inputs(seed=5):
    from neighborretr_amd import synth
    b, L, n_v = 3, 8, 4
    ids = synth.randint(seed, "enc/ids", 1, 97, (b, L))
    mask = np.ones((b, L), dtype=np.int64)
    mask[1, 5:] = 0
    mask[2, 3:] = 0
    ids = ids * mask
    for r in range(b):
        ids[r, mask[r].sum() - 1] = 99                       # EOT = the largest id
    video = synth.normal(seed, "enc/video", (b, n_v, 3, 64, 64)).astype(np.float32)
    vmask = np.ones((b, n_v), dtype=np.int64)
    vmask[2, 2:] = 0
    return torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(video), torch.from_numpy(vmask)


def main():
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to capture")
        return 0
    sys.path.insert(0, ROOT)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    stub("timm"); stub("timm.models"); stub("timm.models.layers", drop_path=None)
    stub("boto3"); stub("botocore"); stub("botocore.exceptions", ClientError=Exception)
    stub("ftfy", fix_text=lambda s: s)
    sys.path.insert(0, REF)
    from NeighborRetr.models.module_clip import CLIP
    from NeighborRetr.models.module_cross import Transformer as TransformerClip
    from NeighborRetr.models.modeling import NeighborRetr
    torch.manual_seed(0)
    clip = CLIP(DIMS["embed_dim"], DIMS["image_resolution"], DIMS["vision_layers"], DIMS["vision_width"], DIMS["vision_patch_size"],
                DIMS["context_length"], DIMS["vocab_size"], DIMS["transformer_width"], DIMS["transformer_heads"],
                DIMS["transformer_layers"]).eval()
    clip.load_state_dict(seeded_state(clip, 11))
    holder = torch.nn.Module()
    holder.frame_position_embeddings = torch.nn.Embedding(DIMS["context_length"], DIMS["transformer_width"])
    holder.transformerClip = TransformerClip(width=DIMS["transformer_width"], layers=TEMPORAL_LAYERS, heads=DIMS["transformer_heads"])
    holder.load_state_dict(seeded_state(holder, 12))
    holder.eval()
    ids, mask, video, vmask = inputs()
    with torch.no_grad():
        t_cls, t_hidden = clip.encode_text(ids, return_hidden=True, mask=mask)
        v_cls, v_hidden = clip.encode_image(video.view(-1, 3, 64, 64), return_hidden=True)
        frames = v_cls.float().view(3, -1, v_cls.shape[-1])
        agg = NeighborRetr.aggregate_video_features(holder, frames, vmask)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "enc_tiny.npz"), t_cls=t_cls.numpy(), t_hidden=t_hidden.numpy(),
                        v_cls=v_cls.numpy(), v_hidden=v_hidden.numpy(), agg=agg.numpy(),
                        clip_keys=np.array(sorted(clip.state_dict().keys())),
                        clip_shapes=np.array([str(tuple(clip.state_dict()[k].shape)) for k in sorted(clip.state_dict().keys())]),
                        temporal_keys=np.array(sorted(holder.state_dict().keys())))
    print("captured enc_tiny.npz:", {k: tuple(v.shape) for k, v in dict(t_hidden=t_hidden, v_hidden=v_hidden, agg=agg).items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
