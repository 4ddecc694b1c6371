#!/usr/bin/env python
"""Where the HOST time of an eagerly launched training step (configs[1]) goes: cProfile over 20 steps, top functions by own and
by cumulative time.  Run on the GPU box:  python tools/train_host_profile.py"""
import cProfile, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from neighborretr_amd import hip, modeling, synth
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.cuda().train()
p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
m.mb_ind = torch.arange(M).cuda()
tf = p["text_feat"].clone().requires_grad_(True); vf = p["video_feat"].clone().requires_grad_(True)


def fb():
    m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
    m._ctm_cache.clear()                       # as after an optimizer step: the stage weights are stale
    m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)[0].backward()


for _ in range(5):
    fb()
torch.cuda.synchronize()
n0 = hip.N_CALLS
t0 = time.perf_counter()
for _ in range(20):
    fb()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"eager training step {dt * 1e3:.2f} ms, {(hip.N_CALLS - n0) / 20:.0f} C-ABI calls per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fb()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(30)

# the backward runs on autograd's device thread, which cProfile does not see: time its Python pieces by hand
from neighborretr_amd import backward as BW, cluster_backward_hip as CBH, cluster_fused as CF
acc = {}


def timed(mod, name):
    fn = getattr(mod, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    setattr(mod, name, wrap)


for mod, name in ((BW, "_global_backward"), (BW, "_local_backward"), (BW, "_mlp_backward_hip"), (CBH, "stage_backward_group"),
                  (CF, "ctm_stage_group"), (CF, "build_stage_weights")):
    timed(mod, name)
CF.CBH = CBH
t_f = t_b = 0.0
for _ in range(20):
    m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
    m._ctm_cache.clear()
    t0 = time.perf_counter()
    out = m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)[0]
    t1 = time.perf_counter()
    out.backward()
    t2 = time.perf_counter()
    t_f += t1 - t0; t_b += t2 - t1
torch.cuda.synchronize()
print(f"host time per step: forward {t_f / 20 * 1e3:.2f} ms, backward {t_b / 20 * 1e3:.2f} ms")
for k, v in acc.items():
    print(f"  {k:24s} {v / 20 * 1e3:6.2f} ms per step")
