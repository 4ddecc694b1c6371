#!/usr/bin/env python
"""profiles/rNN_pmc_sim.json from the rocprofv3 --pmc passes of tools/sim_pmc.py (directories a..e under the given root):
per sim kernel the mean counters, the derived MFMA-busy fraction and the fabric bytes per launch."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nr_sim_" in r.get("Kernel_Name", ""):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(root + "/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nr_sim_" in r.get("Kernel_Name", ""):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"source": "rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/sim_pmc.py (separate passes per counter group; "
                 "MI355X, the step's three products of configs[1], 20 eager launches each)",
       "units": "SQ_BUSY_CYCLES: cycles summed over the 32 shader engines; SQ_VALU_MFMA_BUSY_CYCLES: matrix-pipe busy cycles summed "
                "over the 1024 SIMDs (16 per v_mfma_f32_16x16x32_bf16); SQ_WAVE_CYCLES / SQ_WAIT_*: quad-cycles summed over waves; "
                "FETCH_SIZE / WRITE_SIZE: KiB, FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
       "per_kernel": {}}
for name, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    k = {"counters_mean": {c: round(v, 1) for c, v in sorted(m.items())}}
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        kernel_cycles = m["SQ_BUSY_CYCLES"] / 32.0
        k["kernel_cycles_per_shader_engine"] = round(kernel_cycles, 1)
        k["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / kernel_cycles, 4)
    if "SQ_WAVE_CYCLES" in m:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c in m:
                k[c.lower() + "_frac_of_wave_cycles"] = round(m[c] / m["SQ_WAVE_CYCLES"], 4)
    if m.get("SQ_INSTS_MFMA"):
        # what a change to the kernels' fragment reads has to move: LDS instructions / LDS-active cycles per MFMA (wave-instruction
        # counts summed over the launch)
        for c, key in (("SQ_INSTS_LDS", "lds_insts_per_mfma"), ("SQ_LDS_IDX_ACTIVE", "lds_idx_active_cycles_per_mfma"),
                       ("SQ_INSTS_VALU", "valu_insts_per_mfma")):
            if c in m:
                k[key] = round(m[c] / m["SQ_INSTS_MFMA"], 4)
    if "FETCH_SIZE" in m:
        k["fabric_read_bytes_per_launch"] = round(m["FETCH_SIZE"] * 1024 * 2)
    if "WRITE_SIZE" in m:
        k["fabric_write_bytes_per_launch"] = round(m["WRITE_SIZE"] * 1024)
    if dur.get(name):
        k["profiled_avg_duration_us"] = round(sum(dur[name]) / len(dur[name]) / 1e3, 2)
    out["per_kernel"][name] = k
# the step launches the two bank products as ONE launch of chained tile pairs ("nr_sim_pair_kernel": 2 x 19.33 GF) -- or, when
# that kernel is absent from the passes, the bank kernel (192 x 384 blocks, "<6, 6, ...") twice -- and the split-bf16 batch
# kernel ("<3, ...": 3 x 4.83 GF issued) once
# (the step as shipped launches the three products one by one: the pair kernel's counters are reported, not folded into
# the step's figures, unless asked for with a second argument "pair")
pair = [k for n, k in out["per_kernel"].items() if "pair" in n] if sys.argv[2:3] == ["pair"] else []
bank = [k for n, k in out["per_kernel"].items() if "<6, 6" in n]
x3 = [k for n, k in out["per_kernel"].items() if "<3, " in n]
if (pair or bank) and x3:
    b, x = (pair or bank)[0], x3[0]
    nb = 1 if pair else 2                       # launches that carry the bank flops
    if "fabric_read_bytes_per_launch" in b and "fabric_read_bytes_per_launch" in x:
        tot = lambda k: k["fabric_read_bytes_per_launch"] + k.get("fabric_write_bytes_per_launch", 0)      # noqa: E731
        out["bytes_per_launch_avg_over_step"] = round((nb * tot(b) + tot(x)) / (nb + 1))
    if "mfma_busy_frac" in b and "mfma_busy_frac" in x:
        fb, fx = 2 * 19.33, 3 * 4.83            # issued MFMA GF: two bank products, the 3-pass batch product
        out["mfma_busy_frac_flops_weighted"] = round((fb * b["mfma_busy_frac"] + fx * x["mfma_busy_frac"]) / (fb + fx), 4)
    out["step_uses"] = "nr_sim_pair_kernel + the split-bf16 nr_sim_reg_kernel" if pair else "nr_sim_reg_kernel x 3"
json.dump(out, sys.stdout, indent=1)
