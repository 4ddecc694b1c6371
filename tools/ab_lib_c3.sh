#!/bin/bash
# a bench config under two builds of the library: bash tools/ab_lib_c3.sh <old.so> [config]   (new = the in-tree library)
for i in 1 2; do
for lib in "$1" neighborretr_amd/libnr_hip.so; do
NR_HIP_LIB=$(pwd)/$lib python bench.py --config ${2:-3} --no-cpu-baseline --steps 80 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['ms_per_step'], d['value'], d['roofline']['frac'])"
done
done
