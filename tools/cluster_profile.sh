#!/bin/bash
# rocprofv3 kernel statistics of the clustering alone (tools/cluster_times.py B): bash tools/cluster_profile.sh [tag [B]]
#   -> gpurun_out/<tag>_cluster_times.txt, <tag>_cluster_kernel_stats.csv
repo="$(pwd)"; tag="${1:-cl}"; B="${2:-128}"; out="$repo/gpurun_out/${tag}_cltrace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o b -- python3 "$repo/tools/cluster_times.py" "$B" > "$repo/gpurun_out/${tag}_cluster_times.txt" 2> "$out.log"
cp "$(find "$out" -name '*kernel_stats.csv' | head -1)" "$repo/gpurun_out/${tag}_cluster_kernel_stats.csv"
rm -rf "$out"
cat "$repo/gpurun_out/${tag}_cluster_times.txt"
