#!/bin/bash
# bench.py under runtime environment settings, one per line on stdin ("VAR=val [VAR=val]"; an empty line = default)
while IFS= read -r setting; do
  out=$(env $setting timeout -k 10 120 python bench.py --no-cpu-baseline --steps 500 2>/dev/null | tail -1)
  echo "$out" | python -c "import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %s ms  %s steps/s  parity %s' % ('$setting' or '(default)', d['ms_per_step'], d['value'], d['parity']['pass']))
except Exception as e:
    print('%-50s FAILED' % '$setting')"
done
