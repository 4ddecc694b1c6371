#!/usr/bin/env python
"""U consecutive loss-only steps captured as ONE HIP graph (same streams, every dependency of the eager sequence kept: step
k+1's prologue follows step k's joins, its bank products follow step k's push): what the launch gap between two replays costs.
One child process per U (several captures with different topologies in one process crash the ROCm 7.2 runtime)."""
import os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def main(U):
    from neighborretr_amd import modeling, synth
    B, Nt, Nv, M, K = 128, 24, 12, 512, 20
    dev = torch.device("cuda")
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(dev).train()
    p = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
    m.mb_feat_t, m.mb_feat_v, m.mb_mask_t, m.mb_mask_v = p["mb_feat_t"], p["mb_feat_v"], p["mb_mask_t"], p["mb_mask_v"]
    m.mb_ind = torch.arange(M, device=dev)

    def step():
        with torch.no_grad():
            return m(p["text_feat"], p["text_mask"], p["video_feat"], p["video_mask"], p["idx"], 0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(U):
            out = step()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        n = 600 // U
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (n * U))
    print(f"{U} step(s) per graph: {best * 1e6:7.1f} us per step  ({1 / best:7.0f} steps/s)  losses {[round(float(x), 4) for x in out]}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        main(int(sys.argv[1]))
    else:
        for U in (1, 2, 4):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(U)], capture_output=True, text=True, timeout=600)
            out = [l for l in r.stdout.splitlines() if "per graph" in l]
            print(out[-1] if out else f"U={U}: exit code {r.returncode} {r.stderr.strip().splitlines()[-2:]}", flush=True)
