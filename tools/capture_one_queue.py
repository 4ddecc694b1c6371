#!/usr/bin/env python
"""Does a multi-stream capture survive GPU_MAX_HW_QUEUES=1 on this runtime?  Plain torch ops: two side streams forked from
the capture stream and joined back, replayed 3 times.  Children under GPU_MAX_HW_QUEUES = 1, 2, 4; the parent never touches
the GPU."""
import os, subprocess, sys
CHILD = r'''
import torch
x = torch.zeros(1 << 16, device="cuda")
a, b = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    a.wait_stream(cur); b.wait_stream(cur)
    with torch.cuda.stream(a): y = x + 1
    with torch.cuda.stream(b): z = x + 2
    cur.wait_stream(a); cur.wait_stream(b)
    w = y + z
for _ in range(3): g.replay()
torch.cuda.synchronize()
print("ok", float(w[0]))
'''
for q in ("1", "2", "4"):
    r = subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, "GPU_MAX_HW_QUEUES": q}, capture_output=True, text=True)
    print(f"GPU_MAX_HW_QUEUES={q}: exit {r.returncode}  {r.stdout.strip()[-40:]}")
