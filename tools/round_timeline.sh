#!/bin/bash
# Kernel timeline of ONE replay of rank 0's overlapped round graph (W = 8 emulated): bash tools/round_timeline.sh [W] [serial]
repo="$(pwd)"; out="$repo/gpurun_out/round_prof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o rt -- python3 "$repo/tools/round_profile.py" "${1:-8}" $2 $3 $4 $5 $6 > "$out.log" 2>&1
f=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = [i for i, r in enumerate(rows) if "nr_unpack_gathered" in r["Kernel_Name"]]
if len(mark) < 3:
    mark = [i for i, r in enumerate(rows) if "nr_step_prologue" in r["Kernel_Name"]]
a, b = mark[-3] - 3, mark[-2] - 3
g = rows[a:b]
t0 = int(g[0]["Start_Timestamp"])
end = max(int(r["End_Timestamp"]) for r in g)
print(f"replay: {len(g)} kernels, span {(end - t0) / 1e3:.1f} us, summed kernel time {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in g) / 1e3:.1f} us")
for r in g:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{r['Queue_Id']:>3}  {r['Kernel_Name'][:80]}")
PY
tail -2 "$out.log"
