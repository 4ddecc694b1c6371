#!/bin/bash
# like tools/ab_env.sh with 6 alternations of 1000 steps: tools/ab_env_long.sh VAR=val
for r in 1 2 3 4 5 6; do
  for which in base new; do
    if [ $which = new ]; then out=$(env "$@" python bench.py --no-cpu-baseline --steps 1000 2>/dev/null); else out=$(python bench.py --no-cpu-baseline --steps 1000 2>/dev/null); fi
    echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['ms_per_step'], d['value'])"
  done
done
