import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neighborretr_amd import modeling, synth
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.cuda().train()
if os.environ.get("NR_TRAIN_ORDER"):          # "a,b;c,d": (clustering launches, local launches) per turn, the last pair repeating
    m.train_capture_order = tuple(tuple((1 << 30) if x == "inf" else int(x) for x in t.split(",")) for t in os.environ["NR_TRAIN_ORDER"].split(";"))
    print("train_capture_order", m.train_capture_order)
p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
m.mb_ind = torch.arange(M).cuda()
tf = p["text_feat"].clone().requires_grad_(True); vf = p["video_feat"].clone().requires_grad_(True)
def f():
    return m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): f()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
m._scorer_cache.clear(); m._ctm_cache.clear()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = f()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): g.replay()
torch.cuda.synchronize()
fwd = (time.perf_counter() - t0) / 100 * 1e6
m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
m._scorer_cache.clear(); m._ctm_cache.clear()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    f()[0].backward()
for _ in range(5): g2.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): g2.replay()
torch.cuda.synchronize()
print("training-mode forward only, one graph: %.1f us;  forward + backward: %.1f us" % (fwd, (time.perf_counter() - t0) / 100 * 1e6))
