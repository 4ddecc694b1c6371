#!/bin/bash
# the batch's two scorer launches as one (head.PAIR_BATCH_SCORERS): bench lines with and without
run() { env "$@" python bench.py --no-cpu-baseline --steps 400 ${CFG} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['abi_calls_per_rank_step'], d['parity']['pass'] if d.get('parity') else None)" "${CFG} $*"; }
for CFG in "--config 1" "--config 2" "--config 3"; do
for i in 1 2; do
run NR_PAIR_SCORERS=0
run NR_PAIR_SCORERS=1
done
done
