#!/usr/bin/env python
"""Cost model of a captured HIP graph on this runtime: replay time per node for chains of tiny kernels -- one stream; two
independent chains on two streams captured one after the other; the same two chains captured interleaved."""
import sys, time
import torch

def replay_us(build, n_nodes, reps=50):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6 / n_nodes

def main():
    N = 400
    for numel in (64, 1 << 20):
        x = torch.zeros(numel, device="cuda"); y = torch.zeros(numel, device="cuda")
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        def one():
            for _ in range(N):
                x.add_(1.0)
        def two_serial():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                for _ in range(N // 2): x.add_(1.0)
            with torch.cuda.stream(s2):
                for _ in range(N // 2): y.add_(1.0)
            cur.wait_stream(s1); cur.wait_stream(s2)
        def two_interleaved():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            for _ in range(N // 2):
                with torch.cuda.stream(s1): x.add_(1.0)
                with torch.cuda.stream(s2): y.add_(1.0)
            cur.wait_stream(s1); cur.wait_stream(s2)
        for name, fn in (("one stream", one), ("two chains, captured one after the other", two_serial), ("two chains, captured interleaved", two_interleaved)):
            print(f"{numel:8d} floats per kernel | {name:42s}: {replay_us(fn, N):6.2f} us per node", flush=True)

main()
