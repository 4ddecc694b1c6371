#!/usr/bin/env python
"""Reproducer for the capture-time crash of the ROCm 7.2 HIP runtime met in round 2 (and the likely relative of the
round-1 segfault in tools/graph_overlap.py): inside ONE stream capture, a side stream that has already been joined
(`other.wait_stream(side)`) is forked a second time (`side.wait_event(event recorded on other)`) and given more work.
Plain torch ops only.  Every variant runs in a child process; this parent reports the exit codes and never touches the GPU.

    python tools/capture_refork.py            # variants: refork (crashes), fresh (a third stream instead: fine)
"""
import subprocess
import sys

CHILD = r'''
import sys, torch
variant = sys.argv[1]
x = torch.zeros(1 << 16, device="cuda")
if variant.startswith("many") or variant.startswith("nest"):
    # many: N side streams forked from the capture stream and joined back; nest: a forked stream forks N/2 of its own
    n = int(variant.lstrip("manyestw"))
    ss = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    outs = []
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        if variant.startswith("many"):
            for s_ in ss:
                s_.wait_stream(cur)
                with torch.cuda.stream(s_):
                    outs.append(x + 1)
            for s_ in ss:
                cur.wait_stream(s_)
        else:
            half = n // 2
            roots = ss[:2]
            for ri, r in enumerate(roots):
                r.wait_stream(cur)
                with torch.cuda.stream(r):
                    if variant.startswith("nestw"):
                        outs.append(x * 3)           # the forked stream has a node of its own before it forks
                    kids = ss[2 + ri * (half - 1): 2 + (ri + 1) * (half - 1)]
                    for k_ in kids:
                        k_.wait_stream(r)
                        with torch.cuda.stream(k_):
                            outs.append(x + 1)
                    outs.append(x + 2)
                    for k_ in kids:
                        r.wait_stream(k_)
            for r in roots:
                cur.wait_stream(r)
    g.replay(); torch.cuda.synchronize()
    print("ok", len(outs))
    sys.exit(0)
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        a = x + 1
    with torch.cuda.stream(s1):
        b = x + 2
        s1.wait_stream(s2)                       # s2 is joined into s1 here
        ev = torch.cuda.Event(); ev.record(s1)
    late = s2 if variant == "refork" else s3     # refork: s2 gets more work after having been joined
    with torch.cuda.stream(late):
        late.wait_event(ev)
        c = a + b
    with torch.cuda.stream(s1):
        d = b * 2
    cur.wait_stream(s1); cur.wait_stream(late)
g.replay(); torch.cuda.synchronize()
print("ok", float(c[0]), float(d[0]))
'''


def main():
    for variant in (sys.argv[1:] or ("fresh", "refork")):
        try:
            r = subprocess.run([sys.executable, "-c", CHILD, variant], capture_output=True, text=True, timeout=120)
        except subprocess.TimeoutExpired:
            print(f"{variant:7s}: TIMEOUT (stopping)")
            return 0
        print(f"{variant:7s}: exit code {r.returncode}  {r.stdout.strip()}  {r.stderr.strip().splitlines()[-1] if r.returncode and r.stderr.strip() else ''}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
