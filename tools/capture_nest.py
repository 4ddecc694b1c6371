#!/usr/bin/env python
"""Which nested fork / join shapes inside ONE stream capture crash the ROCm 7.2 runtime?  Plain torch ops, every variant
in a child process.  Variants: number of roots forked from the capture stream x kids forked from each root; kids joined
back into their ROOT ("r") or straight into the capture stream ("c")."""
import subprocess
import sys

CHILD = r'''
import sys, torch
roots_n, kids_n, join = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
x = torch.zeros(1 << 16, device="cuda")
roots = [torch.cuda.Stream() for _ in range(roots_n)]
kids = [[torch.cuda.Stream() for _ in range(kids_n)] for _ in range(roots_n)]
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
outs = []
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    outs.append(x + 0.5)
    for r, ks in zip(roots, kids):
        r.wait_stream(cur)
        with torch.cuda.stream(r):
            outs.append(x * 3)
            for k in ks:
                k.wait_stream(r)
                with torch.cuda.stream(k):
                    outs.append(x + 1)
            outs.append(x + 2)
            if join == "r":
                for k in ks:
                    r.wait_stream(k)
    for r, ks in zip(roots, kids):
        cur.wait_stream(r)
        if join == "c":
            for k in ks:
                cur.wait_stream(k)
    outs.append(x - 1)
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
print("ok", len(outs))
'''

for spec in (sys.argv[1:] or ["1,1,r", "1,2,r", "2,1,r", "1,2,c", "2,2,c", "2,3,c"]):
    a, b, j = spec.split(",")
    try:
        r = subprocess.run([sys.executable, "-c", CHILD, a, b, j], capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        print(f"roots {a} kids {b} join {j}: TIMEOUT (stopping)")
        break
    print(f"roots {a} x kids {b}, kids joined into {'their root' if j == 'r' else 'the capture stream'}: exit code {r.returncode} {r.stdout.strip()}")
