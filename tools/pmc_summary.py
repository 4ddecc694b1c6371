#!/usr/bin/env python
"""Per-kernel means of the counters in a rocprofv3 --pmc csv output directory (counter_collection.csv files)."""
import csv
import glob
import sys
from collections import defaultdict

root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if pat in name:
            key = (name[:90], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in sorted(acc.items()):
    print(key[0], "grid", key[1], "wg", key[2])
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
