#!/bin/bash
# Kernel timeline of ONE replay of the captured training step (configs[1]) under rocprofv3 --kernel-trace: start offset,
# duration and queue of every kernel, the busy time and the gaps.  Run on the GPU box from the repo root:
#   bash tools/train_graph_timeline.sh > gpurun_out/train_graph_timeline.txt
repo="$(pwd)"; out="$repo/gpurun_out/train_graph_prof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o tg -- python3 "$repo/tools/train_graph_profile.py" --fused > "$out.log" 2>&1
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the replays are the tail of the trace; one replay = from a step prologue to the next
pro = [i for i, r in enumerate(rows) if "nr_step_prologue" in r["Kernel_Name"]]
g = rows[pro[-2]:pro[-1]]
t0 = int(g[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in g)
busy = 0
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in g] + [(int(r["End_Timestamp"]), -1) for r in g])
depth, last = 0, t0
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print(f"replay: {len(g)} kernels, span {(end - t0) / 1e3:.1f} us, >=1 kernel running {busy / 1e3:.1f} us, summed kernel time {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in g) / 1e3:.1f} us")
for r in g:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{r['Queue_Id']:>3}  {r['Kernel_Name'][:96]}")
PY
