"""Deterministic synthetic inputs for the similarity / loss head.

A counter-based generator (splitmix64 -> Box-Muller) written in numpy so that
the golden-capture script, the oracle tests, the GPU parity tests and bench.py
all regenerate bit-identical features, masks, bank contents and parameters from
a (seed, stream-name) pair -- no dependence on torch's RNG or on the device.
Shapes and distributions follow SURVEY.md section 8(d).
"""
import math
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_key(seed, stream):
    h = zlib.crc32(stream.encode()) & 0xFFFFFFFF
    return np.uint64(((int(seed) & 0xFFFFFFFF) << 32) | h)


def uniform(seed, stream, shape):
    """float64 uniforms in (0,1), one per counter value."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.full(1, _stream_key(seed, stream), dtype=np.uint64))[0]
        bits = _splitmix64(_splitmix64(ctr ^ key) + key)
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normal(seed, stream, shape):
    """float64 standard normals (Box-Muller on two independent uniform streams)."""
    u1 = uniform(seed, stream + "/a", shape)
    u2 = uniform(seed, stream + "/b", shape)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def randint(seed, stream, lo, hi, shape):
    """integers in [lo, hi] inclusive."""
    u = uniform(seed, stream, shape)
    return np.minimum((u * (hi - lo + 1)).astype(np.int64) + lo, hi)


def make_samples(seed, stream, n, Nt, Nv, d=512, sigma=6.0, ragged=True):
    """n (text, video) pairs: feat[i,tok,:] = base[i] + sigma*eps; prefix-ones masks.

    Returns float32 text [n,Nt,d], video [n,Nv,d] and int64 masks [n,Nt], [n,Nv]."""
    base = normal(seed, stream + "/base", (n, 1, d))
    text = (base + sigma * normal(seed, stream + "/text", (n, Nt, d))).astype(np.float32)
    video = (base + sigma * normal(seed, stream + "/video", (n, Nv, d))).astype(np.float32)
    if ragged:
        tl = randint(seed, stream + "/tlen", min(3, Nt), Nt, (n,))
        vl = randint(seed, stream + "/vlen", 1, Nv, (n,))
    else:
        tl = np.full((n,), Nt)
        vl = np.full((n,), Nv)
    tmask = (np.arange(Nt)[None, :] < tl[:, None]).astype(np.int64)
    vmask = (np.arange(Nv)[None, :] < vl[:, None]).astype(np.int64)
    return text, video, tmask, vmask


def make_problem(seed, B, Nt, Nv, M, d=512, sigma=6.0, ragged=True):
    """One step's inputs: the current batch and an independent memory bank."""
    t, v, tm, vm = make_samples(seed, "batch", B, Nt, Nv, d, sigma, ragged)
    bt, bv, btm, bvm = make_samples(seed, "bank", M, Nt, Nv, d, sigma, ragged)
    return dict(text_feat=t, video_feat=v, text_mask=tm, video_mask=vm,
                mb_feat_t=bt, mb_feat_v=bv, mb_mask_t=btm, mb_mask_v=bvm,
                idx=np.arange(B, dtype=np.int64))


def merged_token_counts(Nt, Nv):
    """Token counts after each CTM stage (cluster.py:712 with the ratios of modeling.py:188-196)."""
    t0 = max(math.ceil(Nt * (1 / 6)), 1)
    t1 = max(math.ceil(t0 * (1 / 4)), 1)
    v0 = max(math.ceil(Nv * (1 / 4)), 1)
    v1 = max(math.ceil(v0 * (1 / 3)), 1)
    return (t0, t1), (v0, v1)


def make_noise(seed, B, Nt, Nv):
    """The DPC-KNN tie-break draws (cluster.py:483) as explicit inputs, float32 in [0,1)."""
    (t0, _), (v0, _) = merged_token_counts(Nt, Nv)
    return dict(t0=uniform(seed, "noise/t0", (B, Nt)).astype(np.float32),
                t1=uniform(seed, "noise/t1", (B, t0)).astype(np.float32),
                v0=uniform(seed, "noise/v0", (B, Nv)).astype(np.float32),
                v1=uniform(seed, "noise/v1", (B, v0)).astype(np.float32))


HEAD_MLPS = ["text_weight_fc", "video_weight_fc", "text_weight_fc0", "video_weight_fc0",
             "text_weight_fc1", "video_weight_fc1", "text_weight_intra", "video_weight_intra"]


def head_param_shapes(d=512):
    """Every parameter of the loss head, keyed by the reference's state-dict names."""
    shapes = {}
    for m in HEAD_MLPS:                       # modeling.py:137-153
        shapes[m + ".0.weight"] = (2 * d, d)
        shapes[m + ".0.bias"] = (2 * d,)
        shapes[m + ".2.weight"] = (1, 2 * d)
        shapes[m + ".2.bias"] = (1,)
    for mod in ("text", "video"):             # modeling.py:186-197
        for s in (0, 1):
            c, b = f"{mod}_ctm{s}", f"{mod}_block{s}"
            shapes[c + ".conv.conv.weight"] = (d, d, 3)
            shapes[c + ".norm.weight"] = (d,)
            shapes[c + ".norm.bias"] = (d,)
            shapes[c + ".score.weight"] = (1, d)
            shapes[c + ".score.bias"] = (1,)
            shapes[b + ".norm1.weight"] = (d,)
            shapes[b + ".norm1.bias"] = (d,)
            shapes[b + ".attn.q.weight"] = (d, d)
            shapes[b + ".attn.q.bias"] = (d,)
            shapes[b + ".attn.kv.weight"] = (2 * d, d)
            shapes[b + ".attn.kv.bias"] = (2 * d,)
            shapes[b + ".attn.proj.weight"] = (d, d)
            shapes[b + ".attn.proj.bias"] = (d,)
    return shapes


def make_params(seed, d=512):
    """Seeded head parameters: weights N(0,0.02), biases N(0,0.01), norm scales 1+N(0,0.05).

    Biases are deliberately non-zero (the reference zero-initialises them) so the
    parity tests exercise every bias path."""
    out = {}
    for name, shp in head_param_shapes(d).items():
        z = normal(seed, "param/" + name, shp)
        if name.endswith("norm.weight") or name.endswith("norm1.weight"):
            val = 1.0 + 0.05 * z
        elif name.endswith(".bias"):
            val = 0.01 * z
        else:
            val = 0.02 * z
        out[name] = val.astype(np.float32)
    return out


DEFAULT_HP = dict(centrality_scale=0.3, beta=0.7, num_neighbors=20, temperature=3.0,
                  uniform_weight=1.0, neighbor_weight=1.0, kl_weight=1.0)   # args_parser.py:26-41
