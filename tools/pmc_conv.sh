#!/bin/bash
# L2 hit rate and fabric traffic of the clustering GEMMs alone (tools/conv_probe.py): bash tools/pmc_conv.sh -> gpurun_out/pmc_conv.txt
repo="$(pwd)"; out="$repo/gpurun_out/pmc_conv"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$out/a" -o p -- python3 "$repo/tools/conv_probe.py" > "$out.a.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/b" -o p -- python3 "$repo/tools/conv_probe.py" > "$out.b.log" 2>&1
cd "$repo"
python tools/pmc_summary.py "$out" nr_linear_group > gpurun_out/pmc_conv.txt
rm -rf "$out"
cat gpurun_out/pmc_conv.txt
