#!/usr/bin/env python
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace database of bench.py:
start / end / duration (us) relative to the step's first kernel, HW queue, grid, kernel name."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id,grid_x,workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if r[0].startswith("nr_bank_ring")]
a, b = idx[-3] + 1, idx[-2] + 1
t0 = rows[a][1]
for r in rows[a:b]:
    print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:6.1f} q{r[3]} g{r[4] // max(r[5], 1):5d}x{r[5]:4d} {r[0][:72]}")
print("step span", (rows[b - 1][2] - t0) / 1e3, "kernels", b - a)
