#!/usr/bin/env python
"""Time of the token scorers' backward (backward._mlp_backward_hip: both modalities, batch + bank tokens) at configs[1], mixed
and exact plan, with events over 30 calls.  Run on the GPU box:  python tools/scorer_bwd_times.py
In a -DNR_TUNE build NR_LINEAR_TILE="MI,NI,STAGES,WC" forces the tile of every grouped GEMM."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import backward, hip, modeling, ops, synth

B, Nt, Nv, M, d = 128, 24, 12, 512, 512
dev = torch.device("cuda", 0)
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=20))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.to(dev)
p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
pt, pv = ops.prepare_tokens(p["text_feat"], p["text_mask"]), ops.prepare_tokens(p["video_feat"], p["video_mask"])
pbt, pbv = ops.prepare_tokens(p["mb_feat_t"], p["mb_mask_t"]), ops.prepare_tokens(p["mb_feat_v"], p["mb_mask_v"])
g = torch.Generator().manual_seed(0)
dl = lambda n: (torch.randn(n, generator=g) * 1e-3).to(dev)
dl_t, dl_v, dl_bt, dl_bv = dl(B * Nt), dl(B * Nv), dl(M * Nt), dl(M * Nv)


def run(p_bank):
    return backward._mlp_backward_hip([
        dict(sw=m.scorer_weights("text_weight_fc"), sets=[(pt, p["text_feat"], dl_t, hip.PREC_BF16X3), (pbt, p["mb_feat_t"], dl_bt, p_bank)]),
        dict(sw=m.scorer_weights("video_weight_fc"), sets=[(pv, p["video_feat"], dl_v, hip.PREC_BF16X3), (pbv, p["mb_feat_v"], dl_bv, p_bank)])])


def timed(fn, n=30):
    """us per call, replayed from a captured graph (the eager loop is host-bound: ~25 us of Python per launch)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


mix = run(hip.PREC_BF16)
backward.ONE_PASS_WEIGHT_GRAD = False
ref = run(hip.PREC_BF16)                 # same hidden-layer recompute, dW1 of the bank tokens in split-bf16
backward.ONE_PASS_WEIGHT_GRAD = True
err = max(float((a[0] - b[0]).abs().max() / b[0].abs().max()) for a, b in zip(mix, ref))
print("tile=%s/%s  scorer backward (one graph): mixed plan %.1f us, exact plan %.1f us; dW1 one-pass vs split-bf16 bank columns: max |diff| / max |dW1| = %.2e"
      % (os.environ.get("NR_LINEAR_TILE", "-"), os.environ.get("NR_LINEAR_TILE1", "-"), timed(lambda: run(hip.PREC_BF16)), timed(lambda: run(hip.PREC_BF16X3)), err), flush=True)
