#!/bin/bash
# rocprofv3 kernel statistics of tools/rank_local_times.py: bash tools/rank_local_profile.sh <tag> [args of the tool ...]
repo="$(pwd)"; tag="${1:-rl}"; shift; out="$repo/gpurun_out/${tag}_rltrace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o b -- python3 "$repo/tools/rank_local_times.py" "$@" > "$repo/gpurun_out/${tag}_rank_local.txt" 2> "$out.log"
cp "$(find "$out" -name '*kernel_stats.csv' | head -1)" "$repo/gpurun_out/${tag}_rank_local_kernel_stats.csv"
rm -rf "$out"
cat "$repo/gpurun_out/${tag}_rank_local.txt"
