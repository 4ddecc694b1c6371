#!/bin/bash
# A/B in one session: centrality weights inside the row-loss launch (NR_FUSE_CW) and the token means off the local chain (NR_COLSUM_OFF)
for rep in 1 2; do for m in "1 1" "0 0" "1 0" "0 1"; do set -- $m
  NR_FUSE_CW=$1 NR_COLSUM_OFF=$2 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no_kernel_profile 2>gpurun_out/tf.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('FUSE_CW=$1 COLSUM_OFF=$2', d['value'], d['ms_per_step'], d['config']['unrolled_graph']['equals_single_step_replays'], d['parity']['pass'], d['parity']['dL'])"
done; done
