export NR_HIP_LIB=$PWD/neighborretr_amd/libnr_tune.so
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no_kernel_profile 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run "base"
NR_MLP_ONE_STAGE=1 run "NR_MLP_ONE_STAGE=1"
NR_LINEAR_TILE=1,2,2,2 run "NR_LINEAR_TILE=1,2,2,2"
NR_LINEAR_TILE=1,2,2,2 NR_MLP_ONE_STAGE=1 run "both"
done
