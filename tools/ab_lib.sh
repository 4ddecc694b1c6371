#!/bin/bash
# A/B of two builds of the library in ONE GPU session: bash tools/ab_lib.sh <lib A> <lib B> <pairs> -- <command that prints a bench JSON line>
# alternates the command under NR_HIP_LIB=A / B and prints ms_per_step of every run.
a="$1"; b="$2"; n="$3"; shift 4
for i in $(seq 1 "$n"); do
    for lib in "$a" "$b"; do
        NR_HIP_LIB="$(pwd)/$lib" "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', d['ms_per_step'], d['roofline']['avg_launch_us'])"
    done
done
