#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU session (run on the GPU box from the repo root):
#   bash tools/round_evidence.sh r04 [A|B]   -> gpurun_out/r04/*     (A: bench lines, rank-local emulation, capture sequence, scorer
#   backward;  B: chain / kernel timings, profiles, counters; both when omitted -- a gpurun call is limited to 20 minutes)
tag="${1:-rNN}"; part="${2:-AB}"; repo="$(pwd)"; out="$repo/gpurun_out/$tag"; mkdir -p "$out"
q() { grep -v "amdgpu.ids" ; }
if [[ "$part" == *A* ]]; then
python bench.py --steps 200 --backward 2> "$out/bench.err" > "$out/bench_final.json"
python bench.py --steps 20 --warmup 5 2>/dev/null > "$out/bench_20steps.json"
python bench.py --steps 200 --no-cpu-baseline --unroll 1 2>/dev/null > "$out/bench_unroll1.json"
python bench.py --steps 200 --no-cpu-baseline --no-pipeline 2>/dev/null > "$out/bench_sequential_unroll.json"
python bench.py --config 2 --steps 50 --warmup 10 2>/dev/null > "$out/bench_c2.json"
python bench.py --config 3 --steps 50 --warmup 10 2>/dev/null > "$out/bench_c3.json"
python tools/rank_local_times.py --worlds 2 4 8 --train --eval --out "gpurun_out/$tag/rank_local.txt" > "$out/rank_local.log" 2>&1
python tools/rank_local_times.py --worlds 8 --only_train --B 1024 --out "gpurun_out/$tag/rank_local_b1024.txt" > "$out/rank_local_b1024.log" 2>&1
python tools/capture_sequence.py 2>&1 | q | grep -v "Warning\|warn\|^  " > "$out/capture_sequence.txt"
python -m pytest tests/test_scorer_backward_gpu.py -q -s 2>&1 | q | grep "^\[\|^  [tv]\|passed\|failed" > "$out/scorer_backward.txt"
fi
if [[ "$part" != *B* ]]; then ls -la "$out"; exit 0; fi
python tools/chain_times.py 2>&1 | q > "$out/chain_times.txt"
python tools/train_times.py 2>&1 | q > "$out/train_times.txt"
python tools/scorer_ab.py 2>&1 | q > "$out/scorer_ab.txt"
python tools/sched_sweep.py 2>&1 | q > "$out/sched_sweep.txt"
python tools/rccl_capture_probe.py 2>&1 | q > "$out/rccl_capture.txt"
python tools/microbench.py 2>&1 | q > "$out/microbench.txt"
bash tools/bench_profile.sh "$tag/step" --unroll 1 > /dev/null 2>&1
bash tools/train_profile.sh 13 > "$out/train_profile.txt" 2>&1
cp "$repo/gpurun_out/train_prof/tr_kernel_stats.csv" "$out/train_kernel_stats.csv" 2>/dev/null
bash tools/train_graph_timeline.sh > "$out/train_graph_timeline.txt" 2>&1
python tools/sim_bwd_times.py 2>&1 | q > "$out/sim_bwd_times.txt"
python tools/sim_bwd_times.py bf16x3 2>&1 | q >> "$out/sim_bwd_times.txt"
python tools/scorer_bwd_times.py 2>&1 | q > "$out/scorer_bwd_times.txt"
python tools/sinkhorn_large_times.py 2>&1 | q > "$out/sinkhorn_large.txt"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/roof" -o r -- python3 "$repo/tools/roofline_launches.py" > "$out/roofline_launches.txt" 2>&1 )
cp "$(find "$out/roof" -name '*kernel_stats.csv' | head -1)" "$out/roofline_kernel_stats.csv" 2>/dev/null; rm -rf "$out/roof"
bash tools/pmc_passes.sh "$out/pmc" && python tools/pmc_to_json.py "$out/pmc" > "$out/pmc_sim.json"; rm -rf "$out/pmc" "$out"/pmc.*.log
python bench.py --steps 50 --no-cpu-baseline --e2e 2>/dev/null > "$out/bench_e2e.json"
ls -la "$out"
