#!/usr/bin/env python
"""Per-kernel summary (name, grid, calls, avg/min us) of a rocprofv3 rocpd database; optional name filter."""
import sqlite3
import sys

db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
q = ("select name, grid_x/workgroup_x, workgroup_x, count(*), avg(end-start), min(end-start) from kernels "
     "where name like ? group by name, grid_x order by name, grid_x")
for r in c.execute(q, (f"%{pat}%",)):
    print(f"{r[0][:70]:70s} wg={r[1]:5d}x{r[2]:4d} n={r[3]:5d} avg={r[4] / 1e3:7.1f} min={r[5] / 1e3:7.1f}")
