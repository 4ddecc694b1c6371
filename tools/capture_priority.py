#!/usr/bin/env python
"""Does capturing a graph on a HIGH-PRIORITY stream crash the runtime (round-1 note)?  Child processes, plain torch ops:
default-priority capture stream with a high-priority side stream; high-priority capture stream."""
import subprocess
import sys

CHILD = r'''
import sys, torch
variant = sys.argv[1]
x = torch.zeros(1 << 16, device="cuda")
hi = torch.cuda.Stream(priority=-1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if variant == "side":
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        hi.wait_stream(cur)
        with torch.cuda.stream(hi):
            a = x + 1
        b = x + 2
        cur.wait_stream(hi)
        c = a + b
else:
    with torch.cuda.graph(g, stream=hi):
        a = x + 1
        c = a * 2
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("ok", float(c[0]))
'''

for variant in ("side", "capture"):
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, variant], capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        print(f"{variant:8s}: TIMEOUT (stopping)")
        break
    print(f"{variant:8s}: exit code {r.returncode}  {r.stdout.strip()}  {r.stderr.strip().splitlines()[-1][:200] if r.returncode and r.stderr.strip() else ''}")
