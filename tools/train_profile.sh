#!/bin/bash
# rocprofv3 kernel statistics of eagerly launched training steps (forward + backward) at configs[1]; run on the GPU box from
# the repo root:  bash tools/train_profile.sh [steps]   -> gpurun_out/train_prof/..., summary on stdout
repo="$(pwd)"; steps="${1:-13}"; out="$repo/gpurun_out/train_prof"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o tr -- python3 "$repo/tools/profile_train.py" "$steps" > "$out.log" 2>&1
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$steps" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(f"kernels per step ~ {calls / steps:.0f}; summed kernel time per step {tot / steps / 1e3:.0f} us")
for r in rows[:50]:
    print(r["Name"][:100].ljust(100), f'{int(r["Calls"]) / steps:6.1f} {float(r["TotalDurationNs"]) / steps / 1e3:8.1f} {float(r["AverageNs"]) / 1e3:7.1f}')
PY
