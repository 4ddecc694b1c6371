#!/bin/bash
# bench.py alternately with and without an environment assignment (A/B in one GPU session): tools/ab_env.sh VAR=val
for r in 1 2 3; do
  for which in base new; do
    if [ $which = new ]; then out=$(env "$@" python bench.py --no-cpu-baseline --steps 300 2>/dev/null); else out=$(python bench.py --no-cpu-baseline --steps 300 2>/dev/null); fi
    echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['parity']['pass'])"
  done
done
