#!/bin/bash
# A/B in one GPU session of the pipelined bench graph with and without the edge "next step's bank chains wait for this step's row
# losses" (head.TAIL_BEFORE_NEXT_BANK_READS): bash tools/ab_tail.sh [pairs]
run() { "$@" python bench.py --no-cpu-baseline --steps 400 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms_per_step', d['ms_per_step'], 'steps/s', d['value'], 'equals single-step replays:', d['config']['unrolled_graph']['equals_single_step_replays'])" "$*"; }
for i in $(seq 1 "${1:-3}"); do
run env NR_TAIL_EDGE=0
run env NR_TAIL_EDGE=1
done
