#!/bin/bash
# Tile A/B of the scorer backward's GEMMs in a -DNR_TUNE build (built on the GPU box):  bash tools/scorer_bwd_sweep.sh
set -e
NR_EXTRA_FLAGS=-DNR_TUNE python -m neighborretr_amd.build --force > /dev/null 2>&1
python tools/scorer_bwd_times.py
for t in ${TILES:-2,2,1,4 2,2,2,4 4,2,1,4 4,2,2,4 2,2,2,2 2,4,2,2}; do NR_LINEAR_TILE=$t python tools/scorer_bwd_times.py; done
