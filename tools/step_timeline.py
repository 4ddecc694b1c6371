#!/usr/bin/env python
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV of bench.py (…_kernel_trace.csv):
start / end / duration (us) relative to the step's first kernel (nr_step_prologue), HW queue, grid, kernel name;
and the end of the step's critical stream (the queue that carries the prologue)."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                 int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)))
rows.sort(key=lambda r: r[1])
starts = [i for i, r in enumerate(rows) if r[0].startswith("nr_step_prologue")]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # consecutive steps to print (pipelined graphs: 2 shows the overlap)
a, b = starts[-2 - n_steps], starts[-2]          # steady-state step(s) well inside the timed loop
t0 = rows[a][1]
crit_q = rows[a][3]
crit_end = 0
for r in rows[a:b]:
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:6.1f} q{r[3]} g{r[4] // max(r[5], 1):5d}x{r[5]:4d} {r[0][:88]}")
    if r[3] == crit_q:
        crit_end = max(crit_end, r[2] - t0)
print(f"step span {max(r[2] for r in rows[a:b]) - t0:.0f} ns over {b - a} kernels; the critical stream (queue {crit_q}: prologue -> "
      f"clustering -> logits -> Sinkhorn) ends at {crit_end / 1e3:.1f} us; next step's prologue starts at {(rows[b][1] - t0) / 1e3:.1f} us")
