#!/bin/bash
# What profiles/r05_* is made from (run on the GPU box from the repo root, one part per gpurun call: a call is limited to 20 min):
#   bash tools/round_evidence_r05.sh A|B|C   -> gpurun_out/r05/*
part="${1:-A}"; repo="$(pwd)"; out="$repo/gpurun_out/r05"; mkdir -p "$out"
q() { grep -v "amdgpu.ids" ; }
if [[ "$part" == A ]]; then
  python bench.py --steps 200 --backward 2> "$out/bench.err" > "$out/bench.json"
  cp gpurun_out/bench_step_kernel_stats.csv "$out/kernel_stats_pipelined.csv"; cp gpurun_out/bench_alone_kernel_stats.csv "$out/roofline_kernel_stats.csv"
  python bench.py --steps 20 --warmup 5 2>/dev/null > "$out/bench_20steps.json"
  cp gpurun_out/bench_step_kernel_stats.csv "$out/kernel_stats_20steps.csv"; cp gpurun_out/bench_alone_kernel_stats.csv "$out/roofline_kernel_stats_20steps.csv"
  python bench.py --steps 200 --no-cpu-baseline --no_kernel_profile --unroll 1 2>/dev/null > "$out/bench_unroll1.json"
  python bench.py --config 2 --steps 50 --warmup 10 2>/dev/null > "$out/bench_c2.json"; cp gpurun_out/bench_step_kernel_stats.csv "$out/c2_kernel_stats.csv"
  python bench.py --config 3 --steps 50 --warmup 10 2>/dev/null > "$out/bench_c3.json"; cp gpurun_out/bench_step_kernel_stats.csv "$out/c3_kernel_stats.csv"
fi
if [[ "$part" == B ]]; then
  for b in 128 16 1024; do echo "B = $b"; python tools/cluster_times.py $b 2>&1 | q; done > "$out/cluster_times.txt"
  bash tools/cluster_profile.sh r05/cl 128 > /dev/null 2>&1; cp "$repo/gpurun_out/r05/cl_cluster_kernel_stats.csv" "$out/cluster_kernel_stats.csv"
  python tools/microbench.py 2>&1 | q > "$out/microbench.txt"
  python tools/chain_times.py 2>&1 | q > "$out/chain_times.txt"
  bash tools/bench_profile.sh r05/step --unroll 1 > /dev/null 2>&1
  NR_PROF_TIMELINE_STEPS=3 bash tools/bench_profile.sh r05/pipe > /dev/null 2>&1
  python tools/sinkhorn_large_times.py 2>&1 | q > "$out/sinkhorn_large.txt"
  bash tools/ab_tail_edge.sh > "$out/ab_tail_edge.txt" 2>&1
fi
if [[ "$part" == C ]]; then
  python tools/rank_local_times.py --worlds 2 4 8 --out "gpurun_out/r05/rank_local.txt" > "$out/rank_local.log" 2>&1
  bash tools/pmc_passes.sh "$out/pmc" && python tools/pmc_to_json.py "$out/pmc" > "$out/pmc_sim.json"; rm -rf "$out/pmc" "$out"/pmc.*.log
  python bench.py --steps 50 --no-cpu-baseline --no_kernel_profile --e2e 2>/dev/null > "$out/bench_e2e.json"
  python tools/train_times.py 2>&1 | q > "$out/train_times.txt"
fi
ls -la "$out"
