#!/bin/bash
# A/B of two builds of libnr_hip.so in one GPU session: tools/ab_bench.sh <base.so> [bench args]
# runs bench.py alternately with the base library (NR_HIP_LIB) and the in-tree one; prints ms_per_step of each run
base=$1; shift
for r in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export NR_HIP_LIB=$base; else unset NR_HIP_LIB; fi
    python bench.py --no-cpu-baseline --steps 300 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'])"
  done
done
