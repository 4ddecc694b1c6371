#!/bin/bash
# bench.py three times (ms_per_step, steps/s, sim launch us); optional env assignments as arguments
for r in 1 2 3; do
  env "$@" python bench.py --no-cpu-baseline --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['parity']['pass'])"
done
