#!/bin/bash
# A/B of the similarity-backward kernel in a -DNR_TUNE build (built on the GPU box, the shipped library is left alone):
#   bash tools/sim_bwd_sweep.sh > gpurun_out/sim_bwd_sweep.txt
set -e
NR_EXTRA_FLAGS=-DNR_TUNE python -m neighborretr_amd.build --force > /dev/null 2>&1
python tools/sim_bwd_times.py
python tools/sim_bwd_times.py bf16x3
for s in ${SLICES:-8 16 20 22 24}; do NR_BWD_SLICES=$s python tools/sim_bwd_times.py; done
for dbg in ${DBG:-1 2 4 8 16}; do NR_BWD_DBG=$dbg python tools/sim_bwd_times.py; done
