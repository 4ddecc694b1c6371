#!/bin/bash
# configs[3] bench under variants of the scorer kernel's block shape (tuning build): bash tools/ab_c3.sh
run() { env "$@" NR_HIP_LIB=$(pwd)/neighborretr_amd/libnr_tune.so python bench.py --config 3 --no-cpu-baseline --steps 80 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'])" "$*"; }
for i in 1 2; do
run NR_X=0
run NR_MLP_SHAPE=3
run NR_MLP_SHAPE=3 NR_MLP_ONE_STAGE=1
run NR_MLP_SHAPE=2
done
