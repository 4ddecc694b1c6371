#!/bin/bash
# Tile A/B of the scorer backward's GEMMs in a -DNR_TUNE build (built on the GPU box):  bash tools/scorer_bwd_sweep.sh
# NR_LINEAR_TILE forces every grouped GEMM's block, NR_LINEAR_TILE1 only the one-pass launches' ("MI,NI,STAGES,WC").
set -e
NR_EXTRA_FLAGS=-DNR_TUNE python -m neighborretr_amd.build --force > /dev/null 2>&1
python tools/scorer_bwd_times.py
for t in ${TILES1:-2,2,1,4 2,2,2,4 4,2,1,4 4,2,2,4 2,2,2,2 2,4,2,2}; do echo "one-pass tile $t"; NR_LINEAR_TILE1=$t python tools/scorer_bwd_times.py; done
