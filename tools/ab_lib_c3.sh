for i in 1 2; do
for lib in libnr_old.so libnr_hip.so; do
NR_HIP_LIB=$(pwd)/neighborretr_amd/$lib python bench.py --config 3 --no-cpu-baseline --steps 80 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'], d['losses'])"
done
done
NR_HIP_LIB=$(pwd)/neighborretr_amd/libnr_old.so python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-160
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-160
