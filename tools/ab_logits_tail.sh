#!/bin/bash
# A/B in one session: the global logits on the tail stream (1) or on the origin (0) of the pipelined graph
for rep in 1 2 3; do for m in 1 0; do
  NR_LOGITS_TAIL=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no_kernel_profile "$@" 2>gpurun_out/lt.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('LOGITS_TAIL=$m', d['value'], d['ms_per_step'], d['config']['unrolled_graph']['equals_single_step_replays'], d['parity']['pass'])"
done; done
