run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms_per_step', d['ms_per_step'], 'steps/s', d['value'], d['config']['unrolled_graph'])" "$*"; }
for i in 1 2; do
run --steps 400 --unroll 10
run --steps 400 --unroll 20
run --steps 400 --unroll 40
run --steps 20 --warmup 5 --unroll 10
run --steps 20 --warmup 5 --unroll 20
done
