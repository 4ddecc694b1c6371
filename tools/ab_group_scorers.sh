#!/bin/bash
# A/B in one session: the step's four scorer calls as ONE launch (NR_GROUP_SCORERS=1) or four
for rep in 1 2; do for m in 1 0; do
  NR_GROUP_SCORERS=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no_kernel_profile "$@" 2>gpurun_out/gs.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('GROUP_SCORERS=$m', d['value'], d['ms_per_step'], d['config']['unrolled_graph']['equals_single_step_replays'], d['parity']['pass'], d['parity']['dL'])"
done; done
