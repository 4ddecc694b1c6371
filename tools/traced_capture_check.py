import os, sys, torch
sys.path.insert(0, os.getcwd())
from neighborretr_amd import modeling, synth
B, Nt, Nv, M, K = 128, 24, 12, 512, 20
m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
m = m.cuda().train()
m.fused_training_clustering = False
p = {k: torch.from_numpy(v).cuda() for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
m.mb_ind = torch.arange(M).cuda()
tf = p["text_feat"].clone().requires_grad_(True); vf = p["video_feat"].clone().requires_grad_(True)
def fb():
    m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
    m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)[0].backward()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): fb()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
m.zero_grad(set_to_none=True); tf.grad = vf.grad = None
with torch.cuda.graph(g):
    m(tf, p["text_mask"], vf, p["video_mask"], p["idx"], 0)[0].backward()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize(); print("replayed ok")
