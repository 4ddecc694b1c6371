"""Feature producers of BASELINE configs[4]: CLIP ViT-B/32 image tower, CLIP text tower and the 4-layer
temporal transformer over the frames, as stock PyTorch-ROCm modules that FEED the HIP loss head.

Out of this build's optimisation scope (~99 % of the end-to-end flops, plain transformer encoders): written
for the MI355X runtime that already exists in PyTorch-ROCm -- batch-first tensors, one fused QKV projection per block,
`F.scaled_dot_product_attention` (the flash / memory-efficient kernels on ROCm) under bf16 autocast -- instead of the
reference's LND layout, `nn.MultiheadAttention` with per-head repeated masks and fp16 weights
(models/module_clip.py:303-555, module_transformer.py:62-156, module_cross.py:54-137).

Parameter names and shapes are the reference's (`visual.conv1.weight`, `visual.transformer.resblocks.N.attn.in_proj_weight`,
`transformer.resblocks.N.mlp.c_fc.weight`, `token_embedding.weight`, `text_projection`, `logit_scale`, ...), so a reference
checkpoint (and OpenAI's ViT-B-32.pt state dict) loads unchanged; without one the towers are randomly initialised the
way module_clip.py:412-441 does it (no network here: throughput runs only).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F
from torch import nn


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class Attention(nn.Module):
    """Self-attention with nn.MultiheadAttention's parameter names (in_proj_weight / in_proj_bias / out_proj)."""

    def __init__(self, d_model, n_head):
        super().__init__()
        self.n_head = n_head
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, x, attn_mask=None):
        """x [B,L,d]; attn_mask: None, a boolean [B,1,L,L] (True = may attend) or an additive float mask."""
        B, L, d = x.shape
        qkv = F.linear(x, self.in_proj_weight, self.in_proj_bias).view(B, L, 3, self.n_head, d // self.n_head)
        q, k, v = qkv.permute(2, 0, 3, 1, 4)
        if attn_mask is not None and attn_mask.dtype != torch.bool:
            attn_mask = attn_mask.to(q.dtype)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
        return self.out_proj(o.transpose(1, 2).reshape(B, L, d))


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, eps=1e-5):
        super().__init__()
        self.attn = Attention(d_model, n_head)
        self.ln_1 = nn.LayerNorm(d_model, eps=eps)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = nn.LayerNorm(d_model, eps=eps)

    def forward(self, x, attn_mask=None):
        x = x + self.attn(self.ln_1(x), attn_mask)
        return x + self.mlp(self.ln_2(x))


class Transformer(nn.Module):
    def __init__(self, width, layers, heads, eps=1e-5):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, eps) for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for blk in self.resblocks:
            x = blk(x, attn_mask)
        return x


class VisualTransformer(nn.Module):
    """module_clip.py:303-343: patchify (conv, frozen), class token, positional embedding, ln_pre, transformer."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))
        for p in self.conv1.parameters():
            p.requires_grad = False

    def forward(self, x):
        # Patchify (module_clip.py:325: conv2d with stride = kernel, frozen) as ONE GEMM on the non-overlapping patches,
        # [B g g, 3 p p] x [3 p p, width]: same arithmetic, no MIOpen solver search at the first call (seconds of naive /
        # im2col trials) and no im2col pass per image.
        p = self.conv1.kernel_size[0]
        B, C, H, W = x.shape
        if self.conv1.stride == self.conv1.kernel_size and H % p == 0 and W % p == 0 and self.conv1.bias is None:
            g_h, g_w = H // p, W // p
            patches = x.view(B, C, g_h, p, g_w, p).permute(0, 2, 4, 1, 3, 5).reshape(B * g_h * g_w, C * p * p)
            x = (patches @ self.conv1.weight.view(self.conv1.out_channels, -1).t().to(patches.dtype)).view(B, g_h * g_w, -1)
        else:
            x = self.conv1(x).flatten(2).transpose(1, 2)            # [B, g*g, width]
        cls = self.class_embedding.to(x.dtype).expand(x.shape[0], 1, -1)
        x = torch.cat((cls, x), dim=1) + self.positional_embedding.to(x.dtype)
        return self.transformer(self.ln_pre(x))


class ClipEncoders(nn.Module):
    """The `clip` sub-module of the reference model (module_clip.py:346-555), ViT towers only."""

    def __init__(self, embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                 context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12):
        super().__init__()
        self.context_length = context_length
        self.visual = VisualTransformer(image_resolution, vision_patch_size, vision_width, vision_layers, vision_width // 64,
                                        embed_dim)
        self.transformer = Transformer(transformer_width, transformer_layers, transformer_heads)
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = nn.LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]))
        self.initialize_parameters()
        self.token_embedding.requires_grad = False                  # as written in the reference (an attribute, no effect)

    def initialize_parameters(self):
        """module_clip.py:412-441."""
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std = (self.transformer.width ** -0.5) * ((2 * self.transformer.layers) ** -0.5)
        attn_std = self.transformer.width ** -0.5
        fc_std = (2 * self.transformer.width) ** -0.5
        for block in self.transformer.resblocks:
            nn.init.normal_(block.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(block.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(block.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(block.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=self.transformer.width ** -0.5)

    def encode_image(self, image, return_hidden=False, mask=None):
        hidden = self.visual(image)
        hidden = self.visual.ln_post(hidden) @ self.visual.proj.to(hidden.dtype)
        x = hidden[:, 0, :]
        return (x, hidden) if return_hidden else x

    def encode_text(self, text, return_hidden=False, mask=None):
        """Causal attention restricted to the valid keys (module_clip.py:532-540)."""
        L = text.shape[1]
        x = self.token_embedding(text) + self.positional_embedding[:L]
        allowed = torch.ones((L, L), dtype=torch.bool, device=text.device).tril_()[None, None]
        if mask is not None:
            allowed = allowed & (mask > 0)[:, None, None, :]
        x = self.transformer(x, allowed)
        hidden = self.ln_final(x) @ self.text_projection.to(x.dtype)
        x = hidden[torch.arange(hidden.shape[0], device=text.device), text.argmax(dim=-1)]
        return (x, hidden) if return_hidden else x


class TemporalTransformer(Transformer):
    """`transformerClip` of the reference model (module_cross.py:54-137): the same residual blocks with the TF-style
    LayerNorm of until_module.py:35-48 (epsilon 1e-12 inside the square root = nn.LayerNorm(eps=1e-12))."""

    def __init__(self, width, layers, heads):
        super().__init__(width, layers, heads, eps=1e-12)


def aggregate_video_features(video_feat, video_mask, frame_position_embeddings, transformer_clip):
    """modeling.py:601-623: frame position embeddings, temporal transformer over the valid frames, residual."""
    original = video_feat
    L = video_feat.shape[1]
    pos = frame_position_embeddings(torch.arange(L, device=video_feat.device))
    x = video_feat + pos[None].to(video_feat.dtype)
    bias = ((1.0 - video_mask.to(torch.float32)) * -1000000.0)[:, None, None, :].expand(-1, 1, L, -1)
    return transformer_clip(x, bias) + original


def synthetic_text_ids(text_mask, vocab_size=49408, seed=0):
    """Random BPE ids with the EOT token (the largest id, module_clip.py:551) at the end of every caption's mask."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, vocab_size - 2, text_mask.shape, generator=g)
    length = text_mask.sum(-1).clamp(min=1)
    ids[torch.arange(ids.shape[0]), length - 1] = vocab_size - 1
    return ids * (text_mask > 0)
