repo="$(pwd)"; out="$repo/gpurun_out/r04/cl1024_trace"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o b -- python3 "$repo/tools/cluster_times.py" 1024 > "$repo/gpurun_out/r04/cl1024.txt" 2> "$out.log"
cp "$(find "$out" -name '*kernel_stats.csv' | head -1)" "$repo/gpurun_out/r04/cl1024_kernel_stats.csv"
rm -rf "$out"
