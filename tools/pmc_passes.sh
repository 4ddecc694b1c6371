#!/bin/bash
# The rocprofv3 counter passes behind profiles/rNN_pmc_sim.json (run on the GPU box from the repo root):
#   tools/pmc_passes.sh gpurun_out/pmc  &&  python tools/pmc_to_json.py gpurun_out/pmc > profiles/rNN_pmc_sim.json
# One pass per counter group (the SQ block multiplexes 4-8 counters; FETCH_SIZE and WRITE_SIZE each in a pass of their own,
# MI355X_MICROARCH.md), the program directly after `--`, no other trace domain next to --pmc.
out=$(realpath -m "$1"); repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA --output-format csv -d "$out/a" -o p -- python3 "$repo/tools/sim_pmc.py" > "$out.a.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$out/b" -o p -- python3 "$repo/tools/sim_pmc.py" > "$out.b.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$out/c" -o p -- python3 "$repo/tools/sim_pmc.py" > "$out.c.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/d" -o p -- python3 "$repo/tools/sim_pmc.py" > "$out.d.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/e" -o p -- python3 "$repo/tools/sim_pmc.py" > "$out.e.log" 2>&1
