#!/bin/bash
# Which hardware queues does one replayed step use, and for how long each?  bash tools/queue_probe.sh "VAR=val ..." [bench args]
# (rocprofv3 --kernel-trace of bench.py --unroll 1; Queue_Id of every kernel of one steady-state step)
repo="$(pwd)"; setting="$1"; shift; out="$repo/gpurun_out/queue_probe"
rm -rf "$out"; cd /tmp && export TMPDIR=/tmp
env $setting rocprofv3 --kernel-trace --output-format csv -d "$out" -o q -- python3 "$repo/bench.py" --no-cpu-baseline --steps 60 --unroll 1 "$@" > "$out.json" 2> "$out.log"
f=$(find "$out" -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$setting" <<'PY'
import csv, sys, collections
rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[1])
st = [i for i, r in enumerate(rows) if r[0].startswith("nr_step_prologue")]
a, b = st[-12], st[-11]
t0 = rows[a][1]
busy = collections.defaultdict(float); names = collections.defaultdict(list)
for n, s, e, q in rows[a:b]:
    busy[q] += (e - s) / 1e3; names[q].append(n.split("(")[0].replace("void ", "")[:28])
print(f"[{sys.argv[2] or 'default'}] step span {(max(r[2] for r in rows[a:b]) - t0) / 1e3:.1f} us, {b - a} kernels, queues: " + "; ".join(f"q{q}: {len(names[q])} kernels, {busy[q]:.0f} us busy" for q in sorted(busy)))
for q in sorted(busy):
    print(f"   q{q}: " + ", ".join(names[q]))
PY
python3 -c "
import json; d=json.loads([l for l in open('$out.json') if l.startswith('{')][-1]); print('   under the profiler:', d['ms_per_step'], 'ms per step')"
