#!/bin/bash
# kernel statistics of the synchronous sharded step of one rank: bash tools/sync_step_profile.sh <tag> [B [W]]
repo="$(pwd)"; tag="${1:-ss}"; shift; out="$repo/gpurun_out/${tag}_sstrace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o b -- python3 "$repo/tools/sync_step_kernels.py" "$@" > "$repo/gpurun_out/${tag}_sync_step.txt" 2> "$out.log"
cp "$(find "$out" -name '*kernel_stats.csv' | head -1)" "$repo/gpurun_out/${tag}_sync_step_kernel_stats.csv"
rm -rf "$out"
cat "$repo/gpurun_out/${tag}_sync_step.txt"
