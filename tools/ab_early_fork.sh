#!/bin/bash
# A/B of the pipelined graph: the batch half of a step's local branch forked in front of the step's prologue (1, default) or
# behind it (0).  Alternating runs in one session.
for rep in 1 2; do for m in 1 0; do
  NR_EARLY_FORK=$m python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no_kernel_profile "$@" 2>gpurun_out/ef.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('NR_EARLY_FORK=$m', d['value'], d['ms_per_step'], d['config']['unrolled_graph'], d['parity']['pass'])"
done; done
