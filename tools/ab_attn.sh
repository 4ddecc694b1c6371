#!/bin/bash
# score-biased attention of the grouped clustering stage: a workgroup per (sample, pair of heads) against the whole-sample form
# (NR_ATTN_WHOLE=1, tuning build), bench lines of the three configs + the grouped clustering alone: bash tools/ab_attn.sh
export NR_HIP_LIB=$(pwd)/neighborretr_amd/libnr_tune.so
run() { env "$@" python bench.py --no-cpu-baseline --steps 200 ${CFG} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'])" "${CFG} $*"; }
for CFG in "--config 1" "--config 2" "--config 3"; do
for i in 1 2; do
run NR_ATTN_WHOLE=1
run NR_ATTN_X=0
done
done
for b in 128 1024; do
NR_ATTN_WHOLE=1 python tools/cluster_times.py $b 2>/dev/null | grep "grouped both" | sed "s/^/whole B=$b /"
python tools/cluster_times.py $b 2>/dev/null | grep "grouped both" | sed "s/^/heads B=$b /"
done
