#!/bin/bash
b() { python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  NR_EXTRA_FLAGS=-DNR_PREP_MAX_PARTS=64 python -m neighborretr_amd.build --force > /dev/null 2>&1; b parts64; b parts64
  NR_EXTRA_FLAGS=-DNR_PREP_MAX_PARTS=256 python -m neighborretr_amd.build --force > /dev/null 2>&1; b parts256; b parts256
  NR_EXTRA_FLAGS=-DNR_PREP_MAX_PARTS=128 python -m neighborretr_amd.build --force > /dev/null 2>&1; b parts128; b parts128
done
