#!/usr/bin/env python
"""nr_bank_absorb_gathered alone: ring head, noise counter and ticket words after each of three launches; time per launch."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import comm, ops, synth
from neighborretr_amd.dist import packed_gather_raw
from types import SimpleNamespace
dev = torch.device("cuda", 0)
B, Nt, Nv, M = 128, 24, 12, 512
p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
with comm.use(comm.EmulatedWorld(1, real_collectives=False).comm(0)):
    recv, lay = packed_gather_raw(p["text_feat"], p["video_feat"], p["idx"], p["text_mask"], p["video_mask"], SimpleNamespace(world_size=1))
bank = {"mb_feat_t": p["mb_feat_t"].clone(), "mb_feat_v": p["mb_feat_v"].clone(), "mb_mask_t": p["mb_mask_t"].float(), "mb_mask_v": p["mb_mask_v"].float(),
        "mb_ind": torch.arange(M, device=dev)}
shadow = (ops.prepare_tokens(bank["mb_feat_t"], bank["mb_mask_t"]), ops.prepare_tokens(bank["mb_feat_v"], bank["mb_mask_v"]))
head = torch.zeros(1, dtype=torch.int32, device=dev)
rng = torch.tensor([5, 0], dtype=torch.int64, device=dev)
for k in range(3):
    ops.bank_absorb_gathered(recv, lay, bank, shadow, head, M, rng)
    torch.cuda.synchronize()
    c = ops._COUNTERS[("absorb", dev)]
    print("after launch", k, "head", int(head), "rng", rng.tolist(), "nonzero ticket words", int((c != 0).sum()), "words", c.numel(), flush=True)
want = ops.prepare_tokens(p["text_feat"], p["text_mask"].float())
h = int(head)
print("newest rows == batch:", torch.equal(bank["mb_feat_t"][h:h + B], p["text_feat"]), torch.equal(shadow[0].hi.view(M, Nt, -1)[h:h + B], want.hi.view(B, Nt, -1)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        ops.bank_absorb_gathered(recv, lay, bank, shadow, head, M, rng)
for _ in range(3):
    g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f"absorb: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us per launch")
