#!/usr/bin/env python
"""Prints a rocprofv3 kernel_stats.csv: name (shortened), calls, average us, share.  python tools/kstats.py file.csv [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:n]:
    print(f"{r['Name'][:118]:118s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1000:8.2f} us {float(r['Percentage']):6.2f} %")
