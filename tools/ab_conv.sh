#!/bin/bash
# block shapes of the stage-0 token convolution GEMM (tuning build): grouped clustering alone + the bench line
export NR_HIP_LIB=$(pwd)/neighborretr_amd/libnr_tune.so
for t in "2,2,1" "3,4,1" "3,4,2" "3,2,1" "3,2,2" "4,4,2" "4,2,1" "4,2,2" "2,4,1" "2,4,2" "6,2,1" "6,2,2"; do
  echo "conv tile $t: $(NR_LINEAR_TILE_CONV=$t python tools/cluster_times.py 128 2>&1 | grep 'grouped stage 0')"
done
