#!/bin/bash
# A/B of where the previous step's row losses are tied into the next step of the pipelined graph: 1 = in front of its bank chain
# (default), 2 = in front of its bank push, 0 = nowhere.  Alternating runs in one session.
for rep in 1 2; do for m in 1 2 0; do
  NR_TAIL_EDGE=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no_kernel_profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NR_TAIL_EDGE=$m', d['value'], d['ms_per_step'], d['config']['unrolled_graph']['equals_single_step_replays'])"
done; done
