#!/bin/bash
# rocprofv3 kernel trace + statistics of bench.py (step + roofline launches) and the timeline of one steady-state step; run on
# the GPU box from the repo root:  bash tools/bench_profile.sh [tag [bench.py arguments, e.g. --config 2]]
#   -> gpurun_out/<tag>_kernel_stats.csv, <tag>_step_timeline.txt
repo="$(pwd)"; tag="${1:-prof}"; shift; out="$repo/gpurun_out/${tag}_trace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o b -- python3 "$repo/bench.py" --no-cpu-baseline --steps "${NR_PROF_STEPS:-200}" "$@" > "$repo/gpurun_out/${tag}_bench_under_rocprof.json" 2> "$out.log"
cp "$(find "$out" -name '*kernel_stats.csv' | head -1)" "$repo/gpurun_out/${tag}_kernel_stats.csv"
python3 "$repo/tools/step_timeline.py" "$(find "$out" -name '*kernel_trace.csv' | head -1)" "${NR_PROF_TIMELINE_STEPS:-1}" > "$repo/gpurun_out/${tag}_step_timeline.txt"
rm -rf "$out"
cat "$repo/gpurun_out/${tag}_step_timeline.txt"
